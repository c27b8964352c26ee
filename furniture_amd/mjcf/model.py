"""CompiledModel: flat physics tables + the env's index tables, with (de)serialisation.

``build_model`` = assemble_scene + compile_mjcf + the name->id reference tables
the env logic needs (ref FurnitureEnv._get_reference furniture.py:2696-2721,
FurnitureSawyerEnv._get_reference furniture_sawyer.py:157-212,
FurnitureBaxterEnv._get_reference furniture_baxter.py:167-244).

Two on-disk forms:
* ``.npz``  -- what ships in ``furniture_amd/assets/compiled`` (generated from the
  reference's MJCF assets by ``scripts/compile_assets.py``; the GPU box has no
  /root/reference).
* blob      -- ``to_blob()``: a self-describing little-endian byte string
  (name table + raw arrays) that crosses the C-ABI in ``fsim_create``.
"""

import io
import json
import os
import struct

import numpy as np

from . import assemble as _asm
from . import compile as _cmp

_COMPILED_DIR = os.path.join(os.path.dirname(os.path.dirname(__file__)), "assets", "compiled")

BLOB_MAGIC = b"FSIMBLOB"
BLOB_VERSION = 3
MAX_ANGLES = 8


class CompiledModel:
    def __init__(self, arrays, meta):
        self.arrays = arrays  # name -> ndarray
        self.meta = meta      # json-able dict (names, strings)
        for k, v in arrays.items():
            setattr(self, k, v)
        for k in ("nq", "nv", "nu", "nbody", "njnt", "ngeom", "nsite", "neq", "npair", "nM", "nparts"):
            setattr(self, k, int(arrays["dims"][_DIMS.index(k)]))

    # -- names -------------------------------------------------------------
    def body_name2id(self, n):
        return self.meta["body_names"].index(n)

    def geom_name2id(self, n):
        return self.meta["geom_names"].index(n)

    def site_name2id(self, n):
        return self.meta["site_names"].index(n)

    def joint_name2id(self, n):
        return self.meta["joint_names"].index(n)

    # -- io ------------------------------------------------------------------
    def save(self, path):
        buf = dict(self.arrays)
        buf["__meta__"] = np.frombuffer(json.dumps(self.meta).encode(), dtype=np.uint8)
        np.savez_compressed(path, **buf)

    @staticmethod
    def load(path):
        z = np.load(path, allow_pickle=False)
        arrays = {k: z[k] for k in z.files if k != "__meta__"}
        meta = json.loads(bytes(z["__meta__"]).decode())
        return CompiledModel(arrays, meta)

    def _cursor_tables(self):
        """Tables the device Cursor agent needs, derived from the stored arrays and names (furniture.py:3290-3310
        on_collision is a SUBSTRING match of 'cursorK' / the part name on the two geom names of a contact):
        cg_cursor[g] = 1 + K for the colliding geom of cursor K (else 0); cursor_pos0 = model body_pos of the two
        cursor bodies; cg_namepart[g] = bitmask of the parts whose name is a substring of colliding geom g's name."""
        A = self.arrays
        ncg = len(A["cg_orig"])
        cur = np.zeros(ncg, dtype=np.int32)
        pos0 = np.zeros(6, dtype=np.float64)
        names = self.meta.get("geom_names", [])
        parts = self.meta.get("part_names", [])
        namepart = np.zeros(ncg, dtype=np.int32)
        for g in range(ncg):
            nm = names[int(A["cg_orig"][g])] if names else ""
            for k in range(2):
                if ("cursor%d" % k) in nm:
                    cur[g] |= 1 << k
            for i, pn in enumerate(parts):
                if pn in nm:
                    namepart[g] |= 1 << i
        if "cursor_bodyid" in A and "body_pos" in A:
            pos0 = np.asarray(A["body_pos"], dtype=np.float64).reshape(-1, 3)[np.asarray(A["cursor_bodyid"])].reshape(-1)
        return {"cg_cursor": cur, "cursor_pos0": pos0, "cg_namepart": namepart}

    def to_blob(self):
        """magic, version, n, then n x {name[48], dtype(i32: 0=f64,1=i32), count(i64), offset(i64)}, data."""
        arrays = dict(self.arrays)
        arrays.update(self._cursor_tables())
        names = sorted(arrays.keys())
        head = 8 + 4 + 4 + len(names) * (48 + 4 + 4 + 8 + 8)
        off = (head + 63) // 64 * 64
        table, chunks = [], []
        for n in names:
            a = arrays[n]
            if a.dtype.kind == "f":
                a, code = np.ascontiguousarray(a, dtype="<f8"), 0
            else:
                a, code = np.ascontiguousarray(a, dtype="<i4"), 1
            raw = a.tobytes()
            table.append(struct.pack("<48siiqq", n.encode(), code, 0, a.size, off))
            pad = (-len(raw)) % 64
            chunks.append((off, raw + b"\0" * pad))
            off += len(raw) + pad
        out = io.BytesIO()
        out.write(BLOB_MAGIC)
        out.write(struct.pack("<ii", BLOB_VERSION, len(names)))
        for t in table:
            out.write(t)
        out.write(b"\0" * (chunks[0][0] - out.tell() if chunks else 0))
        for o, raw in chunks:
            assert out.tell() == o
            out.write(raw)
        return out.getvalue()


_DIMS = ["nq", "nv", "nu", "nbody", "njnt", "ngeom", "nsite", "neq", "npair", "nM", "nparts",
         "nrobot_dof", "narm", "nconn", "ndof_action", "agent_code", "nrobot_geom", "npart_geom"]
_AGENT_CODE = {"Sawyer": 0, "Baxter": 1, "Cursor": 2}


# furniture.py:41-47 (NEW_CONTROLLERS) and :1893: these control types load the robot's motor-actuated MJCF (robot_torque.xml)
TORQUE_CONTROL_TYPES = ("torque", "position", "position_orientation", "joint_impedance", "joint_torque", "joint_velocity")


def compiled_name(agent, furniture_name, control_type="impedance"):
    ct = "torque" if control_type in TORQUE_CONTROL_TYPES else "vel"
    return "%s__%s__%s" % (agent, furniture_name, ct)


def load_compiled(agent, furniture_name, control_type="impedance"):
    """Load the shipped tables; fall back to compiling from MJCF assets if available."""
    path = os.path.join(_COMPILED_DIR, compiled_name(agent, furniture_name, control_type) + ".npz")
    if os.path.exists(path):
        return CompiledModel.load(path)
    root = _asm.default_assets_root()
    if root is None:
        raise FileNotFoundError(
            "no compiled model %s and no MJCF assets (set FURNITURE_ASSETS_ROOT)" % path)
    return build_model(agent, furniture_name, control_type=control_type, assets_root=root)


def build_model(agent, furniture_name, control_type="impedance", assets_root=None, move_speed=0.1):
    assets_root = assets_root or _asm.default_assets_root()
    root, info = _asm.assemble_scene(assets_root, agent, furniture_name, control_type, move_speed)
    m = _cmp.compile_mjcf(root, mesh_root=os.path.join(assets_root, "objects"))
    A = {}
    for k, v in m.__dict__.items():
        if isinstance(v, np.ndarray) and not k.startswith("_"):
            A[k] = v
    A["opt"] = np.array([m.timestep, m.gravity[0], m.gravity[1], m.gravity[2], m.impratio])

    parts = info["part_names"]
    nparts = len(parts)
    part_body = np.array([m.body_names.index(p) for p in parts], dtype=np.int32)
    part_jnt = np.array([m.joint_names.index(p) for p in parts], dtype=np.int32)
    A["part_bodyid"] = part_body
    A["part_qposadr"] = m.jnt_qposadr[part_jnt].astype(np.int32)
    A["part_dofadr"] = m.jnt_dofadr[part_jnt].astype(np.int32)
    A["part_siteid"] = np.array([m.site_names.index(p) for p in parts], dtype=np.int32)
    body2part = -np.ones(m.nbody, dtype=np.int32)
    body2part[part_body] = np.arange(nparts)
    A["body_partid"] = body2part
    A["part_hradius"] = np.array([info["horizontal_radius"][p] for p in parts])
    init = np.zeros((nparts, 7))
    has_init = np.zeros(nparts, dtype=np.int32)
    for i, p in enumerate(parts):
        if p in info["part_init_qpos"]:
            init[i] = info["part_init_qpos"][p]
            has_init[i] = 1
        else:
            init[i] = [0, 0, 0, 1, 0, 0, 0]
    A["part_initqpos"], A["part_hasinit"] = init, has_init

    # welds in part indices (order = XML order, drives _get_next_subtask furniture.py:2723-2736)
    A["eq_part1"] = body2part[m.eq_obj1id].astype(np.int32) if m.neq else np.zeros(0, np.int32)
    A["eq_part2"] = body2part[m.eq_obj2id].astype(np.int32) if m.neq else np.zeros(0, np.int32)

    # geoms the env toggles
    is_part_geom = np.array([body2part[b] >= 0 for b in m.geom_bodyid])
    part_col = np.array([bool(is_part_geom[g] and "collision" in m.geom_names[g]) for g in range(m.ngeom)])
    A["geom_is_partcol"] = part_col.astype(np.int32)
    robot_names = set(info["contact_geoms"])
    for g in info["grippers"].values():
        robot_names.update(g["contact_geoms"])
    robot_geom = np.array([(not is_part_geom[g]) and (m.geom_names[g] in robot_names) for g in range(m.ngeom)])
    A["geom_is_robot"] = robot_geom.astype(np.int32)
    A["floor_geomid"] = np.array([m.geom_names.index("FLOOR")], dtype=np.int32)

    # finger role per geom: bit (2*arm) = left finger set, bit (2*arm+1) = right finger set
    role = np.zeros(m.ngeom, dtype=np.int32)
    arms = info["arms"]
    for ai, arm in enumerate(arms):
        g = info["grippers"][arm]
        for n in g["left_finger_geoms"]:
            role[m.geom_names.index(n)] |= 1 << (2 * ai)
        for n in g["right_finger_geoms"]:
            role[m.geom_names.index(n)] |= 1 << (2 * ai + 1)
    A["geom_fingerrole"] = role

    # robot joints (arm joints in robot order, then gripper joints per arm)
    jq = lambda names: np.array([m.jnt_qposadr[m.joint_names.index(n)] for n in names], dtype=np.int32)
    jd = lambda names: np.array([m.jnt_dofadr[m.joint_names.index(n)] for n in names], dtype=np.int32)
    A["arm_qposadr"], A["arm_dofadr"] = jq(info["joints"]), jd(info["joints"])
    gj = []
    ginit = []
    for arm in arms:
        gj += info["grippers"][arm]["joints"]
        ginit += list(info["grippers"][arm]["init_qpos"])
    A["grip_qposadr"], A["grip_dofadr"] = jq(gj), jd(gj)
    A["arm_initqpos"] = np.asarray(info["init_qpos"], dtype=np.float64)
    A["grip_initqpos"] = np.asarray(ginit, dtype=np.float64)
    if arms:
        A["eef_siteid"] = np.array([m.site_names.index(info["grippers"][a]["grip_site"]) for a in arms], dtype=np.int32)
        A["hand_bodyid"] = np.array([m.body_names.index(a + "_hand") for a in arms], dtype=np.int32)
    else:
        A["eef_siteid"] = np.zeros(0, np.int32)
        A["hand_bodyid"] = np.zeros(0, np.int32)
    if agent in ("Sawyer", "Baxter"):
        # IK controller tables (control_type "ik" / "ik_quaternion"): the pybullet URDF chain(s) the reference's IK controller solves
        # on, per arm (controllers/sawyer_ik_controller.py:112-125, baxter_ik_controller.py:118-137), its rest pose / limits / gains,
        # and the world pose of the MJCF body "base" (pose_in_base_from_name, furniture.py:3380-3396; no joint above it: static)
        from .urdf_chain import load_chain, load_tree_chain
        from ..transform_utils import Quaternion
        bd = os.path.join(assets_root, "bullet_data")
        if agent == "Sawyer":
            c = load_chain(os.path.join(bd, "sawyer_description", "urdf", "sawyer_arm.urdf"))
            chains = [dict(joint_pos=c["ik_joint_pos"], joint_quat=c["ik_joint_quat"], eef_pos=c["ik_eef_pos"], eef_quat=np.array([1.0, 0, 0, 0]),
                           rest=np.array([0, -1.18, 0.00, 2.18, 0.00, 0.57, 3.3161]),                            # :196, :263
                           lower=np.array([-3.05, -3.82, -3.05, -3.05, -2.98, -2.98, -4.71]),                      # :207
                           upper=np.array([3.05, 2.28, 3.05, 3.05, 2.98, 2.98, 4.71]))]                            # :208
            # user_sensitivity (:47), P gain (:82), rest pose: fixed (0) / current joints (1), Rz(-90 deg) end-effector convention (:248-254)
            params = np.array([0.3, 5.0, 0.0, 1.0])
        else:
            urdf = os.path.join(bd, "baxter_description", "urdf", "baxter_mod.urdf")
            chains = []
            for eff, act in ((27, [13, 14, 15, 16, 17, 19, 20]), (45, [31, 32, 33, 34, 35, 37, 38])):  # baxter_ik_controller.py:124-137
                c = load_tree_chain(urdf, eff, act)
                chains.append(dict(joint_pos=c["joint_pos"], joint_quat=c["joint_quat"], eef_pos=c["eef_pos"], eef_quat=c["eef_quat"],
                                   rest=np.zeros(7), lower=c["limits"][:, 0], upper=c["limits"][:, 1]))  # lower / upper: getJointInfo (:146-152)
            params = np.array([1.0, 2.0, 1.0, 0.0])  # user_sensitivity (:43), P gain -2 (:92), rest = current joints (:321), no Rz convention
        A["ik_joint_pos"] = np.concatenate([c["joint_pos"] for c in chains])          # [narm * 7, 3]
        A["ik_joint_quat"] = np.concatenate([c["joint_quat"] for c in chains])        # [narm * 7, 4] wxyz
        A["ik_eef_pos"] = np.stack([c["eef_pos"] for c in chains])                    # [narm, 3]
        A["ik_eef_quat"] = np.stack([c["eef_quat"] for c in chains])                  # [narm, 4]
        A["ik_rest"] = np.stack([c["rest"] for c in chains])
        A["ik_lower"], A["ik_upper"] = np.stack([c["lower"] for c in chains]), np.stack([c["upper"] for c in chains])
        A["ik_params"] = params
        b_ = m.body_names.index("base")
        pos, quat = np.zeros(3), Quaternion([1, 0, 0, 0])
        up = []
        while b_ > 0:
            up.append(b_)
            b_ = int(m.body_parentid[b_])
        for b_ in reversed(up):
            pos = pos + quat.rotate(m.body_pos[b_])
            quat = quat * Quaternion(m.body_quat[b_])
        A["ik_base_pos"], A["ik_base_quat"] = np.asarray(pos, dtype=np.float64), np.array(list(quat), dtype=np.float64)
        # one flat table for the device (csrc/fsim_ik.hpp IKT_*): per arm [joint_pos 21 | joint_quat 28 | eef_pos 3 | eef_quat 4 | rest 7 |
        # lower 7 | upper 7] = 77 floats, then [base_pos 3 | base_quat 4 | params 4]
        blocks = [np.concatenate([c["joint_pos"].reshape(-1), c["joint_quat"].reshape(-1), c["eef_pos"], c["eef_quat"], c["rest"], c["lower"], c["upper"]])
                  for c in chains]
        A["ik_table"] = np.concatenate(blocks + [A["ik_base_pos"], A["ik_base_quat"], params])
    if agent == "Cursor":
        A["cursor_bodyid"] = np.array([m.body_names.index("cursor0"), m.body_names.index("cursor1")], dtype=np.int32)
        A["cursor_geomid"] = np.array([m.geom_names.index("cursor0"), m.geom_names.index("cursor1")], dtype=np.int32)

    # connector sites (ref furniture.py:955-988, 1065-1067): pair key = name.split(",")[0].split("-")
    conn = [s for s in range(m.nsite) if "conn_site" in m.site_names[s]]
    keys = {}

    def key_id(tok):
        return keys.setdefault(tok, len(keys))

    cs_site, cs_part, cs_a, cs_b, cs_nang, cs_ang = [], [], [], [], [], []
    for s in conn:
        nm = m.site_names[s]
        toks = nm.split(",")[0].split("-")
        cs_site.append(s)
        cs_part.append(body2part[m.site_bodyid[s]])
        # names are "<A>-<B>,..."; anything else can never satisfy pairs1 == pairs2[::-1] with 2 tokens
        cs_a.append(key_id(toks[0]))
        cs_b.append(key_id(toks[1]) if len(toks) > 1 else -1)
        angs = [float(x) for x in nm.split(",")[1:-1] if x]
        if len(angs) > MAX_ANGLES:
            raise NotImplementedError("more than %d allowed angles on %s" % (MAX_ANGLES, nm))
        cs_nang.append(len(angs))
        cs_ang.append(angs + [0.0] * (MAX_ANGLES - len(angs)))
    A["conn_siteid"] = np.array(cs_site, dtype=np.int32)
    A["conn_partid"] = np.array(cs_part, dtype=np.int32)
    A["conn_keya"] = np.array(cs_a, dtype=np.int32)
    A["conn_keyb"] = np.array(cs_b, dtype=np.int32)
    A["conn_nangle"] = np.array(cs_nang, dtype=np.int32)
    A["conn_angles"] = np.array(cs_ang, dtype=np.float64).reshape(len(conn), MAX_ANGLES)

    # action -> ctrl (ref furniture.py:3359-3367)
    if m.nu:
        cr = m.actuator_ctrlrange
        A["ctrl_bias"] = 0.5 * (cr[:, 1] + cr[:, 0])
        A["ctrl_weight"] = 0.5 * (cr[:, 1] - cr[:, 0])
    else:
        A["ctrl_bias"], A["ctrl_weight"] = np.zeros(0), np.zeros(0)

    from .reduce import reduce_model
    A["geom_names_list"] = list(m.geom_names)  # (names for reduce_model's diagnostics; not stored)
    red = reduce_model(A)
    del A["geom_names_list"]
    A.update(red)
    A["flags"] = np.array([1 if info["recipe_path"] is not None else 0], dtype=np.int32)

    ndof_action = {"Sawyer": 9, "Baxter": 17, "Cursor": 15}[agent]
    dims = dict(nq=m.nq, nv=m.nv, nu=m.nu, nbody=m.nbody, njnt=m.njnt, ngeom=m.ngeom, nsite=m.nsite,
                neq=m.neq, npair=m.npair, nM=m.nM, nparts=nparts,
                nrobot_dof=len(A["arm_dofadr"]) + len(A["grip_dofadr"]), narm=len(arms), nconn=len(conn),
                ndof_action=ndof_action, agent_code=_AGENT_CODE[agent],
                nrobot_geom=int(robot_geom.sum()), npart_geom=int(part_col.sum()))
    A["dims"] = np.array([dims[k] for k in _DIMS], dtype=np.int32)

    meta = dict(agent=agent, furniture_name=furniture_name, control_type=control_type,
                part_names=parts, arms=arms, body_names=m.body_names, joint_names=m.joint_names,
                geom_names=m.geom_names, site_names=m.site_names, actuator_names=m.actuator_names,
                has_recipe=info["recipe_path"] is not None, move_speed=move_speed,
                conn_keys=sorted(keys, key=keys.get))
    if info["recipe_path"] is not None:
        rec = _read_recipe(info["recipe_path"])
        meta["site_recipe"] = [list(x) for x in rec.get("site_recipe", [])]
        # what the dense-reward env reads from the recipe (furniture_sawyer_dense.py:149-216, 243)
        meta["recipe"] = {k: rec[k] for k in ("recipe", "waypoints", "grip_init_pos", "z_finedist", "num_connects") if k in rec}
    return CompiledModel(A, meta)


def _read_recipe(path):
    """the furniture's recipe yaml; site_recipe entries are [[site1, site2, angle?], ...] (ref util/__init__.py:54-61 loader)."""
    import yaml

    class _L(yaml.SafeLoader):
        pass

    _L.add_constructor("tag:yaml.org,2002:python/tuple", lambda l, n: list(l.construct_sequence(n)))
    with open(path) as f:
        rec = yaml.load(f, Loader=_L)
    return rec
