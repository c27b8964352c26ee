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
