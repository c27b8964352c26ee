"""Scene assembly: arena + robot (+gripper) + furniture parts + welds -> one MJCF tree.

This is the host-side, run-once-per-furniture-id input stage of the hot path
("scene assembly").  It restates what the reference does
with ``MujocoXML.merge`` (furniture/env/models/base.py:76-101), ``FloorTask``
(tasks/floor_task.py:18-77), ``Robot.add_gripper`` (robots/robot.py:15-46),
``MujocoXMLObject.get_collision`` (objects/objects.py:186-206) and
``FurnitureEnv._load_model_robot/_arena/_object`` (furniture.py:1889-2003),
but produces a plain ``xml.etree`` tree that ``compile.py`` flattens; it never
goes through mujoco-py.

Needs the reference's *asset* directory (MJCF files are input data, not code).
It is only used in the build container; the GPU box loads the pre-compiled
tables under ``furniture_amd/assets/compiled``.
"""

import copy
import glob
import os
import xml.etree.ElementTree as ET

import numpy as np

_MERGED_SECTIONS = ("actuator", "equality", "sensor", "contact", "default")


def default_assets_root():
    root = os.environ.get("FURNITURE_ASSETS_ROOT")
    if root:
        return root
    cand = "/root/reference/furniture/env/models/assets"
    return cand if os.path.isdir(cand) else None


def furniture_table(assets_root):
    """(xml relative paths, names, name->id); ids follow sorted file names
    (ref furniture/env/models/__init__.py:10-21)."""
    files = sorted(glob.glob(os.path.join(assets_root, "objects", "*.xml")))
    base = [os.path.basename(f) for f in files]
    names = [b.split(".")[0] for b in base]
    return ["objects/" + b for b in base], names, {n: i for i, n in enumerate(names)}


def _fmt(arr):
    return " ".join("{}".format(x) for x in arr)


class XmlDoc:
    """One MJCF file with the sections the reference's merge() tracks."""

    def __init__(self, path):
        self.path = path
        self.tree = ET.parse(path)
        self.root = self.tree.getroot()
        for sec in ("worldbody", "actuator", "asset", "equality", "sensor", "contact", "default"):
            if self.root.find(sec) is None:
                self.root.append(ET.Element(sec))
            setattr(self, sec, self.root.find(sec))

    def merge(self, other, merge_body=True):
        # ref base.py:76-101 -- worldbody children, then the tracked sections.
        if merge_body:
            for body in list(other.worldbody):
                self.worldbody.append(body)
        for sec in _MERGED_SECTIONS:
            dst = getattr(self, sec)
            for child in list(getattr(other, sec)):
                dst.append(child)


# ---------------------------------------------------------------------------
# robots / grippers (ref models/robots/*.py, models/grippers/*.py)
# ---------------------------------------------------------------------------

SAWYER = dict(
    xml="robots/sawyer/robot.xml",
    xml_torque="robots/sawyer/robot_torque.xml",
    bottom_offset=np.array([0.0, 0.0, -0.913]),
    init_qpos=np.array([-0.28, -0.60, 0.00, 1.86, 0.00, 0.3, 1.57]),
    joints=["right_j%d" % i for i in range(7)],
    dof=7,
    contact_geoms=[
        "pedestal_collision", "right_arm_base_link_collision", "right_l0_collision",
        "head_collision", "screen_collision", "right_l1_collision", "right_l2_collision",
        "right_l3_collision", "right_l4_collision", "right_l5_collision",
        "right_l6_collision", "right_l4_2_collision", "right_l2_2_collision",
        "right_l1_2_collision",
    ],
    arms=["right"],
)

BAXTER = dict(
    xml="robots/baxter/robot.xml",
    xml_torque="robots/baxter/robot_torque.xml",
    bottom_offset=np.array([0.0, 0.0, -0.913]),
    # ref baxter_robot.py:44-46 (right arm then left arm)
    init_qpos=np.array([0.814, -0.44, -0.07, 0.5, 0, 1.641, -1.57629266,
                        -0.872, -0.39, 0.07, 0.5, 0, 1.641, -1.57629197]),
    joints=["right_s0", "right_s1", "right_e0", "right_e1", "right_w0", "right_w1", "right_w2",
            "left_s0", "left_s1", "left_e0", "left_e1", "left_w0", "left_w1", "left_w2"],
    dof=14,
    contact_geoms=None,  # filled from the reference list in load_robot()
    arms=["right", "left"],
)

GRIPPERS = {
    "TwoFingerGripper": dict(
        xml="grippers/two_finger_gripper.xml",
        init_qpos=np.array([0.020833, -0.020833]),
        joints=["r_gripper_l_finger_joint", "r_gripper_r_finger_joint"],
        contact_geoms=["r_finger_g0", "r_finger_g1", "l_finger_g0", "l_finger_g1",
                       "r_fingertip_g0", "l_fingertip_g0", "right_gripper_base_collision"],
        left_finger_geoms=["l_finger_g0", "l_finger_g1", "l_fingertip_g0"],
        right_finger_geoms=["r_finger_g0", "r_finger_g1", "r_fingertip_g0"],
        grip_site="grip_site",
    ),
    "LeftTwoFingerGripper": dict(
        xml="grippers/left_two_finger_gripper.xml",
        init_qpos=np.array([0.020833, -0.020833]),
        joints=["l_gripper_l_finger_joint", "l_gripper_r_finger_joint"],
        contact_geoms=["l_g_r_finger_g0", "l_g_r_finger_g1", "l_g_l_finger_g0", "l_g_l_finger_g1",
                       "l_g_r_fingertip_g0", "l_g_l_fingertip_g0", "left_gripper_base_collision"],
        left_finger_geoms=["l_g_l_finger_g0", "l_g_l_finger_g1", "l_g_l_fingertip_g0"],
        right_finger_geoms=["l_g_r_finger_g0", "l_g_r_finger_g1", "l_g_r_fingertip_g0"],
        grip_site="l_g_grip_site",
    ),
}

# ref models/robots/baxter_robot.py:67-82
_BAXTER_CONTACT_GEOMS = [
    "right_upper_shoulder_collision", "right_lower_shoulder_collision",
    "right_upper_elbow_collision", "right_lower_elbow_collision",
    "right_upper_forearm_collision", "right_lower_forearm_collision",
    "right_wrist_collision",
    "left_upper_shoulder_collision", "left_lower_shoulder_collision",
    "left_upper_elbow_collision", "left_lower_elbow_collision",
    "left_upper_forearm_collision", "left_lower_forearm_collision",
]


def _add_gripper(robot_doc, mount_body, gripper_doc):
    # ref robots/robot.py:15-46
    arm = robot_doc.worldbody.find(".//body[@name='%s']" % mount_body)
    if arm is None:
        raise ValueError("no mount body %r" % mount_body)
    for act in gripper_doc.actuator:
        nm = act.get("name")
        if nm is None or not nm.startswith("gripper"):
            raise ValueError("gripper actuator name %r lacks 'gripper' prefix" % nm)
    for body in list(gripper_doc.worldbody):
        arm.append(body)
    robot_doc.merge(gripper_doc, merge_body=False)


def load_robot(assets_root, agent, use_torque=False, move_speed=0.1):
    """Returns (XmlDoc, info dict)."""
    info = {"agent": agent}
    if agent == "Sawyer":
        spec = SAWYER
        doc = XmlDoc(os.path.join(assets_root, spec["xml_torque" if use_torque else "xml"]))
        g = GRIPPERS["TwoFingerGripper"]
        _add_gripper(doc, "right_hand", XmlDoc(os.path.join(assets_root, g["xml"])))
        grippers = {"right": g}
    elif agent == "Baxter":
        spec = dict(BAXTER, contact_geoms=_BAXTER_CONTACT_GEOMS)
        doc = XmlDoc(os.path.join(assets_root, spec["xml_torque" if use_torque else "xml"]))
        gr, gl = GRIPPERS["TwoFingerGripper"], GRIPPERS["LeftTwoFingerGripper"]
        _add_gripper(doc, "right_hand", XmlDoc(os.path.join(assets_root, gr["xml"])))
        _add_gripper(doc, "left_hand", XmlDoc(os.path.join(assets_root, gl["xml"])))
        grippers = {"right": gr, "left": gl}
    elif agent == "Cursor":
        doc = XmlDoc(os.path.join(assets_root, "robots/cursor/robot.xml"))
        size = move_speed / 2.0
        for nm in ("cursor0", "cursor1"):
            # ref robots/cursor.py:14-29, furniture.py:1945-1950
            doc.worldbody.find("./body[@name='%s']" % nm).set("pos", _fmt([0, 0, size]))
            g = doc.worldbody.find("./body/geom[@name='%s']" % nm)
            g.set("size", _fmt([size] * 3))
            g.set("margin", _fmt([size]))
        info.update(joints=[], init_qpos=np.zeros(0), dof=14, contact_geoms=["cursor0", "cursor1"],
                    grippers={}, arms=[])
        return doc, info
    else:
        raise NotImplementedError("agent %r is outside the hot-path scope (Sawyer/Baxter/Cursor)" % agent)

    base = doc.worldbody.find("./body[@name='base']")
    # ref furniture.py:1901-1902 / sawyer_robot.py:24-36
    base.set("pos", _fmt(np.array([0, 0.65, -0.7]) - spec["bottom_offset"]))
    base.set("quat", _fmt([1, 0, 0, -1]))
    info.update(joints=list(spec["joints"]), init_qpos=spec["init_qpos"].copy(), dof=spec["dof"],
                contact_geoms=list(spec["contact_geoms"]), grippers=grippers, arms=list(spec["arms"]))
    return doc, info


# ---------------------------------------------------------------------------
# furniture
# ---------------------------------------------------------------------------

def load_furniture(assets_root, furniture_name):
    path = os.path.join(assets_root, "objects", furniture_name + ".xml")
    doc = XmlDoc(path)
    # part order = document order of <body> (ref models/base.py:159-167)
    part_names = [b.get("name") for b in doc.root.iter("body")]
    init_qpos = {}
    custom = doc.root.find("custom")
    if custom is not None:
        # ref objects/objects.py:149-164
        for num in custom:
            nm = num.attrib.get("name", "")
            if "initpos" in nm:
                key = "_".join(nm.split("_")[0:-1])
                if key in part_names:
                    init_qpos[key] = [float(x) for x in num.attrib["data"].split()]
    hradius = {}
    for p in part_names:
        s = doc.worldbody.find("./body/site[@name='%s_horizontal_radius_site']" % p)
        hradius[p] = float(s.get("size")) if s is not None else 0.0
    return doc, part_names, init_qpos, hradius


def _part_collision_body(doc, name, friction=(1, 10, 0.5)):
    # ref objects/objects.py:186-206 (get_collision(name, site=True))
    body = copy.deepcopy(doc.worldbody.find("./body[@name='%s']" % name))
    for i, g in enumerate(body.findall("geom")):
        gname = g.get("name")
        if not (gname.startswith("noviz") or gname.startswith("collision")):
            g.set("name", "{}-{}".format(name, i))
        g.set("friction", _fmt(friction))
    body.append(ET.Element("site", attrib={"pos": "0 0 0", "size": "0.002 0.002 0.002",
                                          "rgba": "1 0 0 0", "type": "sphere", "name": name}))
    # ref tasks/floor_task.py:66
    body.append(ET.Element("joint", attrib={"name": name, "type": "free", "damping": "0.0001"}))
    return body


def assemble_scene(assets_root, agent, furniture_name, control_type="impedance",
                   move_speed=0.1, no_collision=False):
    """Build the merged MJCF tree the reference hands to mujoco (furniture.py:1812-1838).

    Returns (root Element, info) where info carries the name tables the env needs.
    """
    use_torque = control_type in ("torque", "position", "position_orientation", "joint_impedance", "joint_torque", "joint_velocity")
    world = XmlDoc(os.path.join(assets_root, "base.xml"))

    arena = XmlDoc(os.path.join(assets_root, "arenas/floor_arena.xml"))
    floor = arena.worldbody.find("./geom[@name='FLOOR']")
    # ref furniture.py:1967-1977, arenas/arena.py (FloorArena)
    floor.set("size", _fmt(np.array([1.5, 1.0, 0.125]) / 2))
    floor.set("friction", _fmt((2.0, 0.005, 0.0001)))

    robot, rinfo = load_robot(assets_root, agent, use_torque=use_torque, move_speed=move_speed)
    if no_collision:
        for g in robot.worldbody.findall(".//geom"):
            g.set("conaffinity", "0")
            g.set("contype", "0")

    furn, part_names, init_qpos, hradius = load_furniture(assets_root, furniture_name)

    # FloorTask.__init__ order: arena, robot, objects, equality (floor_task.py:33-36)
    world.merge(arena)
    world.merge(robot)
    # (ref tasks/floor_task.py merge_objects -> merge_asset: the furniture's meshes travel with its bodies; three furniture collide them)
    have = {a_.get("name") for a_ in world.asset}
    for a_ in list(furn.asset):
        if a_.get("name") not in have:
            world.asset.append(copy.deepcopy(a_))
    for p in part_names:
        world.worldbody.append(_part_collision_body(furn, p))
    for eq in list(furn.equality):
        world.equality.append(eq)

    recipe_path = os.path.join(assets_root, "recipes", furniture_name + ".yaml")
    info = dict(rinfo)
    info.update(furniture_name=furniture_name, part_names=part_names, part_init_qpos=init_qpos,
                horizontal_radius=hradius, control_type=control_type,
                recipe_path=recipe_path if os.path.exists(recipe_path) else None)
    return world.root, info
