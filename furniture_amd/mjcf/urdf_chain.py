"""The serial chain the reference's IK controller solves on: `bullet_data/sawyer_description/urdf/sawyer_arm.urdf`, loaded
by pybullet in `controllers/sawyer_ik_controller.py:112-125`; end effector = link index 6 (`:162-163, 194`), whose pybullet
"link state" is the CENTRE-OF-MASS frame of right_l6 (getLinkState()[0:2]).  Parsed at asset-compile time into flat tables
that ship inside the compiled model (the GPU box has neither the URDF nor pybullet)."""
import math
import xml.etree.ElementTree as ET

import numpy as np


def _rpy_quat(r, p, y):
    cr, sr, cp, sp, cy, sy = math.cos(r / 2), math.sin(r / 2), math.cos(p / 2), math.sin(p / 2), math.cos(y / 2), math.sin(y / 2)
    # R = Rz(y) Ry(p) Rx(r), wxyz
    return np.array([cr * cp * cy + sr * sp * sy, sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy])


def load_chain(path, n_joints=7, eef_link="right_l6"):
    root = ET.parse(path).getroot()
    joints = [j for j in root.findall("joint") if j.get("type") == "revolute"][:n_joints]
    assert len(joints) == n_joints
    pos, quat, lim = [], [], []
    parent = None
    for j in joints:
        if parent is not None:
            assert j.find("parent").get("link") == parent, "the arm must be one serial chain"
        parent = j.find("child").get("link")
        o = j.find("origin")
        pos.append([float(v) for v in o.get("xyz").split()])
        quat.append(_rpy_quat(*[float(v) for v in o.get("rpy").split()]))
        assert [float(v) for v in j.find("axis").get("xyz").split()] == [0.0, 0.0, 1.0]
        l = j.find("limit")
        lim.append([float(l.get("lower")), float(l.get("upper"))])
    assert parent == eef_link
    link = [l for l in root.findall("link") if l.get("name") == eef_link][0]
    io = link.find("inertial").find("origin")
    assert [float(v) for v in io.get("rpy").split()] == [0.0, 0.0, 0.0]
    return dict(ik_joint_pos=np.array(pos), ik_joint_quat=np.array(quat), ik_limits=np.array(lim),
                ik_eef_pos=np.array([float(v) for v in io.get("xyz").split()]))


def _qmul(a, b):
    w1, x1, y1, z1 = a
    w2, x2, y2, z2 = b
    return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2])


def _qrot(q, v):
    w, u = q[0], np.asarray(q[1:])
    t = 2.0 * np.cross(u, v)
    return np.asarray(v) + w * t + np.cross(u, t)


def pybullet_joint_order(root):
    """Joint (= child link) indices as pybullet assigns them: depth-first from the base link, children in file order.  Checked
    against the constants the reference hard-codes for baxter_mod.urdf (controllers/baxter_ik_controller.py:124-137: effectors 27 /
    45, arm joints 13-17, 19, 20 / 31-35, 37, 38) in tests/test_ik.py."""
    joints = root.findall("joint")
    children = {}
    for j in joints:
        children.setdefault(j.find("parent").get("link"), []).append(j)
    child_links = {j.find("child").get("link") for j in joints}
    base = [l.get("name") for l in root.findall("link") if l.get("name") not in child_links]
    assert len(base) == 1
    order = []

    def dfs(link):
        for j in children.get(link, []):
            order.append(j)
            dfs(j.find("child").get("link"))
    dfs(base[0])
    return order


def load_tree_chain(path, eef_index, joint_indices):
    """Chain from the base to pybullet link `eef_index` of a URDF with fixed joints: the revolute joints `joint_indices` (pybullet
    numbering) with every fixed transform before them folded into their origin, and the fixed tail from the last revolute joint
    to the centre-of-mass frame of the end link (pos, quat)."""
    root = ET.parse(path).getroot()
    order = pybullet_joint_order(root)
    parent_of = {j.find("child").get("link"): j for j in order}
    chain = []
    link = order[eef_index].find("child").get("link")
    end_link = link
    while link in parent_of:
        chain.append(parent_of[link])
        link = parent_of[link].find("parent").get("link")
    chain.reverse()
    idx = {id(j): i for i, j in enumerate(order)}
    pos, quat, lim = [], [], []
    cp, cq = np.zeros(3), np.array([1.0, 0, 0, 0])   # accumulated fixed transform since the last revolute joint
    for j in chain:
        o = j.find("origin")
        xyz = [float(v) for v in o.get("xyz").split()] if o is not None else [0.0, 0, 0]
        rpy = [float(v) for v in o.get("rpy").split()] if o is not None and o.get("rpy") else [0.0, 0, 0]
        cp = cp + _qrot(cq, xyz)
        cq = _qmul(cq, _rpy_quat(*rpy))
        if j.get("type") == "revolute":
            assert idx[id(j)] == joint_indices[len(pos)], (j.get("name"), idx[id(j)])
            assert [float(v) for v in j.find("axis").get("xyz").split()] == [0.0, 0.0, 1.0]
            l = j.find("limit")
            pos.append(cp); quat.append(cq); lim.append([float(l.get("lower")), float(l.get("upper"))])
            cp, cq = np.zeros(3), np.array([1.0, 0, 0, 0])
        else:
            assert j.get("type") == "fixed", j.get("type")
    assert len(pos) == len(joint_indices)
    link_el = [l for l in root.findall("link") if l.get("name") == end_link][0]
    inertial = link_el.find("inertial")
    if inertial is not None and inertial.find("origin") is not None:
        io = inertial.find("origin")
        cp = cp + _qrot(cq, [float(v) for v in io.get("xyz").split()])
        cq = _qmul(cq, _rpy_quat(*[float(v) for v in (io.get("rpy") or "0 0 0").split()]))
    return dict(joint_pos=np.array(pos), joint_quat=np.array(quat), limits=np.array(lim), eef_pos=cp, eef_quat=cq)
