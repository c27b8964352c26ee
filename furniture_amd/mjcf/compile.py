"""MJCF tree -> flat model tables (the "model compiler").

Restates, for the feature subset the hot-path models use (SURVEY.md appendix
C.1), the compile-time derivations MuJoCo 2.0 performs when the reference calls
``load_model_from_xml`` (furniture/env/models/base.py:113-115): body/joint/dof/
geom/site numbering (depth-first, elements grouped by body), quaternion
normalisation, inertia from primitive geoms, default-class application, contact
pair filtering (same weld body, parent-child, <exclude>), bounding radii, and
the ``invweight0`` quantities MuJoCo uses to scale constraint regularisation.

Everything is float64 numpy here; the device library down-converts to fp32.
"""

import math
import os
import xml.etree.ElementTree as ET

import numpy as np

# MuJoCo enum values (mjtGeom, mjtJoint) kept so ids/types read the same.
GEOM_PLANE, GEOM_HFIELD, GEOM_SPHERE, GEOM_CAPSULE, GEOM_ELLIPSOID, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH = range(8)
JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = range(4)
_GEOM_TYPES = dict(plane=0, hfield=1, sphere=2, capsule=3, ellipsoid=4, cylinder=5, box=6, mesh=7)
_JNT_TYPES = dict(free=0, ball=1, slide=2, hinge=3)

MJ_MINVAL = 1e-15

# MuJoCo 2.0 built-in defaults (mjmodel / XML reference)
_DEF_SOLREF = (0.02, 1.0)
_DEF_SOLIMP = (0.9, 0.95, 0.001, 0.5, 2.0)
_DEF_FRICTION = (1.0, 0.005, 0.0001)


def _floats(s, n=None, default=None):
    if s is None:
        return None if default is None else np.array(default, dtype=np.float64)
    v = np.array([float(x) for x in s.split()], dtype=np.float64)
    if n is not None and v.size < n and default is not None:
        out = np.array(default, dtype=np.float64)
        out[: v.size] = v
        return out
    return v


def qmul(a, b):
    return np.array(
        [
            a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
            a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
            a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
            a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0],
        ]
    )


def qconj(q):
    return np.array([q[0], -q[1], -q[2], -q[3]])


def q2m(q):
    w, x, y, z = q
    return np.array(
        [
            [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
            [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
            [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
        ]
    )


def m2q(R):
    """Rotation matrix -> unit quaternion (w>=0 branch by largest diagonal)."""
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    if tr > 0:
        s = math.sqrt(tr + 1.0) * 2
        q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = math.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = [(R[2, 1] - R[1, 2]) / s, 0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s]
    elif R[1, 1] > R[2, 2]:
        s = math.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = [(R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s]
    else:
        s = math.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = [(R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s]
    q = np.array(q)
    return q / np.linalg.norm(q)


def _axis_quat(axis, ang):
    s = math.sin(ang / 2)
    return np.array([math.cos(ang / 2), axis[0] * s, axis[1] * s, axis[2] * s])


def _orientation(attrib):
    """quat / euler (radians, intrinsic xyz -- base.xml:3 sets angle=radian)."""
    if "quat" in attrib:
        q = _floats(attrib["quat"])
        n = np.linalg.norm(q)
        return q / n if n > 0 else np.array([1.0, 0, 0, 0])
    if "euler" in attrib:
        e = _floats(attrib["euler"])
        q = np.array([1.0, 0, 0, 0])
        for ax, ang in zip(np.eye(3), e):
            q = qmul(q, _axis_quat(ax, ang))
        return q
    for bad in ("axisangle", "xyaxes", "zaxis", "fromto"):
        if bad in attrib:
            raise NotImplementedError("orientation spec %r not used by in-scope models" % bad)
    return np.array([1.0, 0, 0, 0])


class _Defaults:
    """<default> classes: tag -> attribute dict, nested classes inherit."""

    def __init__(self, root):
        self.classes = {"main": {}}
        top = root.find("default")
        if top is not None:
            self._walk(top, "main", {})

    def _walk(self, node, name, inherited):
        cur = {k: dict(v) for k, v in inherited.items()}
        for child in node:
            if child.tag != "default":
                cur.setdefault(child.tag, {}).update(child.attrib)
        self.classes[name] = cur
        for child in node:
            if child.tag == "default":
                self._walk(child, child.get("class", "main"), cur)

    def resolve(self, elem, childclass):
        cls = elem.get("class") or childclass or "main"
        base = dict(self.classes.get(cls, {}).get(elem.tag, {}))
        base.update(elem.attrib)
        return base


def load_stl(path):
    """Triangles [n, 3, 3] of a binary (or ASCII) STL file."""
    raw = open(path, "rb").read()
    if len(raw) >= 84:
        n = int(np.frombuffer(raw, dtype="<u4", count=1, offset=80)[0])
        if len(raw) == 84 + 50 * n:
            rec = np.frombuffer(raw, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]), count=n, offset=84)
            return np.array(rec["v"], dtype=np.float64)
    tri = [[float(x) for x in ln.split()[1:4]] for ln in raw.decode("ascii", "ignore").splitlines() if ln.strip().startswith("vertex")]
    return np.array(tri, dtype=np.float64).reshape(-1, 3, 3)


def mesh_properties(tri):
    """(volume, centre of mass, inertia tensor about the centre of mass per unit density) of the closed surface `tri` [n, 3, 3], by
    signed tetrahedra from the origin -- what MuJoCo's compiler does with a mesh geom's triangles (the surface is taken as closed and
    consistently oriented; a surface wound inside out gives a negative volume and is flipped)."""
    a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]
    det = np.einsum("ij,ij->i", a, np.cross(b, c))  # 6 x signed volume of (0, a, b, c)
    vol = det.sum() / 6.0
    if vol < 0:
        det, vol = -det, -vol
    com = (det[:, None] * (a + b + c)).sum(0) / (24.0 * vol)
    # second moments of a tetrahedron (0, a, b, c): integral x_i x_j dV = det / 120 * (sum over vertex pairs ...)
    s = a + b + c
    C = (np.einsum("n,ni,nj->ij", det, a, a) + np.einsum("n,ni,nj->ij", det, b, b) + np.einsum("n,ni,nj->ij", det, c, c) + np.einsum("n,ni,nj->ij", det, s, s)) / 120.0
    C -= vol * np.outer(com, com)  # about the centre of mass
    I = np.trace(C) * np.eye(3) - C
    return vol, com, I


def mesh_hull(tri):
    """vertices [k, 3] of the convex hull of the mesh: what MuJoCo collides a mesh geom with"""
    from scipy.spatial import ConvexHull
    pts = np.unique(tri.reshape(-1, 3), axis=0)
    h = ConvexHull(pts)
    return pts[np.sort(h.vertices)]


def _geom_mass_inertia(gtype, size, density):
    """mass and diagonal inertia (geom frame, about geom centre) of a primitive."""
    if gtype == GEOM_BOX:
        hx, hy, hz = size[:3]
        m = density * 8 * hx * hy * hz
        I = m / 3.0 * np.array([hy * hy + hz * hz, hx * hx + hz * hz, hx * hx + hy * hy])
    elif gtype == GEOM_SPHERE:
        r = size[0]
        m = density * 4.0 / 3.0 * math.pi * r ** 3
        I = np.full(3, 0.4 * m * r * r)
    elif gtype == GEOM_CYLINDER:
        r, h = size[0], size[1]
        m = density * math.pi * r * r * 2 * h
        ix = m * (3 * r * r + 4 * h * h) / 12.0
        I = np.array([ix, ix, 0.5 * m * r * r])
    elif gtype == GEOM_CAPSULE:
        r, h = size[0], size[1]
        mc = density * math.pi * r * r * 2 * h
        ms = density * 4.0 / 3.0 * math.pi * r ** 3
        m = mc + ms
        iz = 0.5 * mc * r * r + 0.4 * ms * r * r
        ix = mc * (3 * r * r + 4 * h * h) / 12.0 + ms * (0.4 * r * r + 0.75 * r * h + h * h)
        I = np.array([ix, ix, iz])
    elif gtype == GEOM_ELLIPSOID:
        a, b, c = size[:3]
        m = density * 4.0 / 3.0 * math.pi * a * b * c
        I = m / 5.0 * np.array([b * b + c * c, a * a + c * c, a * a + b * b])
    elif gtype in (GEOM_PLANE, GEOM_HFIELD):
        return 0.0, np.zeros(3)
    elif gtype == GEOM_MESH:
        # Visual meshes carry density 0 everywhere except toy_table's part1 (density="01",
        # SURVEY C.1): 1 kg/m^3 x ~1e-4 m^3 of STL volume ~ 0.1 g next to the part's primitive
        # colliders; that mass is dropped (documented in DESIGN.md).  Meshes that collide or weigh more are handled by the caller
        # (compile_mjcf: volume and inertia from the triangles).
        return 0.0, np.zeros(3)
    else:
        raise ValueError(gtype)
    return m, I


def _rbound(gtype, size):
    if gtype == GEOM_SPHERE:
        return size[0]
    if gtype == GEOM_CAPSULE:
        return size[0] + size[1]
    if gtype == GEOM_CYLINDER:
        return math.sqrt(size[0] ** 2 + size[1] ** 2)
    if gtype in (GEOM_BOX, GEOM_ELLIPSOID):
        return float(np.linalg.norm(size[:3])) if gtype == GEOM_BOX else float(max(size[:3]))
    return 0.0  # plane: unbounded, broadphase always passes


class Flat:
    """Plain attribute bag of numpy arrays + name lists."""

    def as_dict(self):
        return {k: v for k, v in self.__dict__.items()}


def compile_mjcf(root, mesh_root=None):
    """Flatten an assembled MJCF tree.  Returns a ``Flat``.  mesh_root: directory the <mesh file=...> paths are relative to (only
    needed by models whose mesh geoms collide or carry mass: three furniture)."""
    dfl = _Defaults(root)
    mesh_assets = {}
    for a_ in root.iter("asset"):
        for me in a_.findall("mesh"):
            mesh_assets[me.get("name")] = (me.get("file"), _floats(me.get("scale"), 3, (1, 1, 1)))
    mesh_cache = {}

    def mesh_of(name):
        """(hull vertices, volume, centre of mass, inertia about it per unit density) of mesh asset `name`, in the mesh geom's frame"""
        if name not in mesh_cache:
            if mesh_root is None or name not in mesh_assets:
                raise NotImplementedError("mesh geom %r collides / carries mass, but its STL file cannot be found (mesh_root)" % name)
            f, sc = mesh_assets[name]
            tri = load_stl(os.path.join(mesh_root, f)) * sc
            vol, com, I = mesh_properties(tri)
            mesh_cache[name] = (mesh_hull(tri), vol, com, I)
        return mesh_cache[name]
    opt = root.find("option")
    optattr = dict(opt.attrib) if opt is not None else {}
    if optattr.get("cone", "pyramidal") != "elliptic":
        raise NotImplementedError("only cone=elliptic (base.xml:4) is implemented")
    m = Flat()
    m.timestep = float(optattr.get("timestep", 0.002))
    m.gravity = _floats(optattr.get("gravity"), 3, (0, 0, -9.81))
    m.impratio = float(optattr.get("impratio", 1.0))

    bodies, joints, geoms, sites = [], [], [], []
    world = root.find("worldbody")

    def add_body(elem, parent, childclass):
        bid = len(bodies)
        cc = elem.get("childclass") or childclass
        if elem.tag == "worldbody":
            rec = dict(name="world", parent=0, pos=np.zeros(3), quat=np.array([1.0, 0, 0, 0]), inertial=None)
        else:
            rec = dict(name=elem.get("name"), parent=parent, pos=_floats(elem.get("pos"), 3, (0, 0, 0)),
                       quat=_orientation(elem.attrib), inertial=elem.find("inertial"))
        rec["id"] = bid
        rec["joints"], rec["geoms"], rec["sites"] = [], [], []
        bodies.append(rec)
        for ch in elem:
            if ch.tag == "joint" or ch.tag == "freejoint":
                a = dfl.resolve(ch, cc) if ch.tag == "joint" else dict(ch.attrib, type="free")
                rec["joints"].append(a)
            elif ch.tag == "geom":
                rec["geoms"].append(dfl.resolve(ch, cc))
            elif ch.tag == "site":
                rec["sites"].append(dfl.resolve(ch, cc))
        for ch in elem:
            if ch.tag == "body":
                add_body(ch, bid, cc)

    add_body(world, 0, None)
    nbody = len(bodies)

    # ---- joints / dofs ---------------------------------------------------
    qpos0 = []
    jnt = dict(type=[], qposadr=[], dofadr=[], bodyid=[], pos=[], axis=[], limited=[], range=[],
               margin=[], solref=[], solimp=[], name=[])
    dof = dict(bodyid=[], jntid=[], parentid=[], armature=[], damping=[])
    body_jntadr, body_jntnum, body_dofadr, body_dofnum = [], [], [], []
    last_dof_of_body = {}
    for b in bodies:
        body_jntadr.append(len(jnt["type"]) if b["joints"] else -1)
        body_jntnum.append(len(b["joints"]))
        body_dofadr.append(len(dof["bodyid"]) if b["joints"] else -1)
        nd0 = len(dof["bodyid"])
        # parent dof = last dof of nearest ancestor with dofs
        anc = b["parent"]
        pd = -1
        if b["id"] != 0:
            a = anc
            while True:
                if a in last_dof_of_body:
                    pd = last_dof_of_body[a]
                    break
                if a == 0:
                    break
                a = bodies[a]["parent"]
        for ja in b["joints"]:
            jt = _JNT_TYPES[ja.get("type", "hinge")]
            jid = len(jnt["type"])
            jnt["type"].append(jt)
            jnt["name"].append(ja.get("name", ""))
            jnt["qposadr"].append(len(qpos0))
            jnt["dofadr"].append(len(dof["bodyid"]))
            jnt["bodyid"].append(b["id"])
            jnt["pos"].append(_floats(ja.get("pos"), 3, (0, 0, 0)))
            ax = _floats(ja.get("axis"), 3, (0, 0, 1))
            jnt["axis"].append(ax / max(np.linalg.norm(ax), MJ_MINVAL))
            jnt["limited"].append(1 if ja.get("limited", "false") == "true" else 0)
            jnt["range"].append(_floats(ja.get("range"), 2, (0, 0)))
            jnt["margin"].append(float(ja.get("margin", 0)))
            jnt["solref"].append(_floats(ja.get("solreflimit"), 2, _DEF_SOLREF))
            jnt["solimp"].append(_floats(ja.get("solimplimit"), 5, _DEF_SOLIMP))
            damping = float(ja.get("damping", 0))
            arm = float(ja.get("armature", 0))
            if float(ja.get("stiffness", 0)) != 0 or float(ja.get("frictionloss", 0)) != 0:
                raise NotImplementedError("joint stiffness/frictionloss unused by in-scope models")
            if jt == JNT_FREE:
                qpos0.extend(list(b["pos"]) + list(b["quat"]))
                nd = 6
            elif jt == JNT_BALL:
                raise NotImplementedError("ball joints unused by in-scope models")
            else:
                qpos0.append(float(ja.get("ref", 0)))
                nd = 1
            for k in range(nd):
                dof["bodyid"].append(b["id"])
                dof["jntid"].append(jid)
                dof["parentid"].append(pd)
                dof["armature"].append(arm)
                dof["damping"].append(damping)
                pd = len(dof["bodyid"]) - 1
        body_dofnum.append(len(dof["bodyid"]) - nd0)
        if len(dof["bodyid"]) > nd0:
            last_dof_of_body[b["id"]] = len(dof["bodyid"]) - 1

    nq, nv, njnt = len(qpos0), len(dof["bodyid"]), len(jnt["type"])

    # ---- bodies ----------------------------------------------------------
    body_parent = np.array([b["parent"] for b in bodies], dtype=np.int32)
    body_pos = np.array([b["pos"] for b in bodies])
    body_quat = np.array([b["quat"] for b in bodies])
    body_rootid = np.zeros(nbody, dtype=np.int32)
    body_weldid = np.zeros(nbody, dtype=np.int32)
    for b in bodies[1:]:
        i = b["id"]
        body_rootid[i] = i if b["parent"] == 0 else body_rootid[b["parent"]]
        body_weldid[i] = i if b["joints"] else body_weldid[b["parent"]]

    # ---- geoms -----------------------------------------------------------
    G = dict(name=[], type=[], bodyid=[], contype=[], conaffinity=[], condim=[], size=[], pos=[], quat=[],
             friction=[], solref=[], solimp=[], margin=[], gap=[], rbound=[], solmix=[], priority=[], density=[], mesh=[])
    for b in bodies:
        for ga in b["geoms"]:
            gt = _GEOM_TYPES[ga.get("type", "sphere")]
            G["name"].append(ga.get("name", ""))
            G["type"].append(gt)
            G["bodyid"].append(b["id"])
            G["contype"].append(int(ga.get("contype", 1)))
            G["conaffinity"].append(int(ga.get("conaffinity", 1)))
            G["condim"].append(int(ga.get("condim", 3)))
            size = _floats(ga.get("size"), 3, (0, 0, 0))
            if size.size > 3:
                size = size[:3]
            G["size"].append(size)
            G["pos"].append(_floats(ga.get("pos"), 3, (0, 0, 0)))
            G["quat"].append(_orientation(ga))
            G["friction"].append(_floats(ga.get("friction"), 3, _DEF_FRICTION))
            G["solref"].append(_floats(ga.get("solref"), 2, _DEF_SOLREF))
            G["solimp"].append(_floats(ga.get("solimp"), 5, _DEF_SOLIMP))
            G["margin"].append(float(ga.get("margin", 0)))
            G["gap"].append(float(ga.get("gap", 0)))
            G["mesh"].append(ga.get("mesh") if gt == GEOM_MESH else None)
            G["rbound"].append(_rbound(gt, size))
            G["solmix"].append(float(ga.get("solmix", 1)))
            G["priority"].append(int(ga.get("priority", 0)))
            G["density"].append(float(ga.get("density", 1000)))
    ngeom = len(G["type"])
    # mesh geoms that collide or carry mass: convex hull (what they collide with), bounding radius about the geom origin.  Visual meshes
    # (contype = conaffinity = 0, density <= 1) stay what they were: massless and absent from the tables.
    mesh_adr, mesh_num, mesh_vert = [-1] * ngeom, [0] * ngeom, []
    for g in range(ngeom):
        if G["type"][g] != GEOM_MESH or not (G["contype"][g] or G["conaffinity"][g]):
            continue
        hull = mesh_of(G["mesh"][g])[0]
        # the portal routine takes the geom ORIGIN as the interior point of the hull (csrc/fsim_collide.hpp np_mpr; MuJoCo re-centres a
        # mesh on its centre of mass, here the vertices stay where the file puts them): the origin must lie strictly inside
        from scipy.spatial import ConvexHull
        eq = ConvexHull(hull).equations  # rows [n, d]: n . x + d <= 0 inside
        if not (eq[:, 3] < -1e-6).all():
            raise NotImplementedError("mesh %s: the geom origin is not strictly inside the convex hull (needs re-centring)" % G["mesh"][g])
        if len(hull) > 32767:
            raise NotImplementedError("mesh %s: more than 32767 hull vertices" % G["mesh"][g])
        mesh_adr[g], mesh_num[g] = sum(len(v) for v in mesh_vert), len(hull)
        mesh_vert.append(hull)
        G["rbound"][g] = float(np.linalg.norm(hull, axis=1).max())
        G["size"][g] = np.abs(hull).max(axis=0)
    if any(c not in (1, 3) for c, t, ct, ca in zip(G["condim"], G["type"], G["contype"], G["conaffinity"]) if ct or ca):
        raise NotImplementedError("only condim 1/3 contacts are implemented")

    # ---- sites -----------------------------------------------------------
    S = dict(name=[], bodyid=[], pos=[], quat=[])
    for b in bodies:
        for sa in b["sites"]:
            S["name"].append(sa.get("name", ""))
            S["bodyid"].append(b["id"])
            S["pos"].append(_floats(sa.get("pos"), 3, (0, 0, 0)))
            S["quat"].append(_orientation(sa))
    nsite = len(S["name"])

    # ---- inertial properties --------------------------------------------
    body_mass = np.zeros(nbody)
    body_ipos = np.zeros((nbody, 3))
    body_iquat = np.tile(np.array([1.0, 0, 0, 0]), (nbody, 1))
    body_inertia = np.zeros((nbody, 3))
    gi = 0
    geom_index_of_body = {}
    for b in bodies:
        geom_index_of_body[b["id"]] = list(range(gi, gi + len(b["geoms"])))
        gi += len(b["geoms"])
    for b in bodies[1:]:
        i = b["id"]
        ine = b["inertial"]
        if ine is not None:
            body_mass[i] = float(ine.get("mass"))
            body_ipos[i] = _floats(ine.get("pos"), 3, (0, 0, 0))
            body_iquat[i] = _orientation(ine.attrib)
            if ine.get("fullinertia") is not None:
                raise NotImplementedError("fullinertia unused by in-scope models")
            body_inertia[i] = _floats(ine.get("diaginertia"), 3, (0, 0, 0))
            continue
        # inertiafromgeom="auto": no <inertial> -> accumulate the body's geoms
        tot, com = 0.0, np.zeros(3)
        parts = []
        for g in geom_index_of_body[i]:
            if G["type"][g] == GEOM_MESH and (mesh_adr[g] >= 0 or G["density"][g] > 1.0) and G["density"][g] > 0:
                # the mesh's own volume, centre of mass and inertia tensor (signed tetrahedra of its triangles), entered as a primitive
                # sitting at that centre of mass with its principal axes
                _, vol, cm, It = mesh_of(G["mesh"][g])
                mg = G["density"][g] * vol
                wI, VI = np.linalg.eigh(It * G["density"][g])
                if np.linalg.det(VI) < 0:
                    VI[:, 2] = -VI[:, 2]
                Rg = q2m(G["quat"][g])
                pg = G["pos"][g] + Rg @ cm
                parts.append((mg, wI, pg, Rg @ VI))
                tot += mg
                com += mg * pg
                continue
            mg, Ig = _geom_mass_inertia(G["type"][g], G["size"][g], G["density"][g])
            if mg <= 0:
                continue
            parts.append((mg, Ig, G["pos"][g], q2m(G["quat"][g])))
            tot += mg
            com += mg * G["pos"][g]
        if tot <= 0:
            continue
        com /= tot
        I = np.zeros((3, 3))
        for mg, Ig, p, R in parts:
            d = p - com
            I += R @ np.diag(Ig) @ R.T + mg * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
        w, V = np.linalg.eigh(I)
        order = np.argsort(-w)  # MuJoCo stores principal inertias in descending order
        w, V = w[order], V[:, order]
        if np.linalg.det(V) < 0:
            V[:, 2] = -V[:, 2]
        if np.allclose(I, np.diag(np.diag(I)), atol=1e-14 * max(1.0, np.abs(I).max())):
            # already diagonal: keep the geom-aligned frame (no gratuitous axis permutation)
            w, V = np.diag(I).copy(), np.eye(3)
        body_mass[i] = tot
        body_ipos[i] = com
        body_iquat[i] = m2q(V)
        body_inertia[i] = w
    for b in bodies[1:]:
        i = b["id"]
        if b["joints"] and (body_mass[i] < MJ_MINVAL) and body_weldid[i] == i:
            # MuJoCo raises "mass and inertia of moving bodies must be larger than mjMINVAL"
            # only if the whole moving subtree is massless; children carry the mass here.
            pass

    # ---- actuators -------------------------------------------------------
    A = dict(name=[], jntid=[], gain=[], bias=[], ctrllimited=[], ctrlrange=[], forcelimited=[], forcerange=[], gear=[])
    act = root.find("actuator")
    for a in (list(act) if act is not None else []):
        attr = dfl.resolve(a, None)
        jname = attr.get("joint")
        if jname is None:
            raise NotImplementedError("only joint transmissions are used by in-scope models")
        jid = jnt["name"].index(jname)
        if a.tag == "motor":
            gain, bias = 1.0, (0.0, 0.0, 0.0)
        elif a.tag == "position":
            kp = float(attr.get("kp", 1))
            gain, bias = kp, (0.0, -kp, 0.0)
        elif a.tag == "velocity":
            kv = float(attr.get("kv", 1))
            gain, bias = kv, (0.0, 0.0, -kv)
        else:
            raise NotImplementedError("actuator <%s>" % a.tag)
        A["name"].append(attr.get("name", ""))
        A["jntid"].append(jid)
        A["gain"].append(gain)
        A["bias"].append(bias)
        A["ctrllimited"].append(1 if attr.get("ctrllimited", "false") == "true" else 0)
        A["ctrlrange"].append(_floats(attr.get("ctrlrange"), 2, (0, 0)))
        A["forcelimited"].append(1 if attr.get("forcelimited", "false") == "true" else 0)
        A["forcerange"].append(_floats(attr.get("forcerange"), 2, (0, 0)))
        A["gear"].append(float((attr.get("gear", "1")).split()[0]))
    nu = len(A["name"])

    # ---- equality (welds only) ------------------------------------------
    E = dict(obj1=[], obj2=[], active=[], data=[], solref=[], solimp=[])
    body_names = [b["name"] for b in bodies]
    eqs = root.find("equality")
    for e in (list(eqs) if eqs is not None else []):
        if e.tag != "weld":
            raise NotImplementedError("equality <%s>" % e.tag)
        attr = dfl.resolve(e, None)
        b1 = body_names.index(attr["body1"])
        b2 = body_names.index(attr["body2"]) if attr.get("body2") else 0
        E["obj1"].append(b1)
        E["obj2"].append(b2)
        E["active"].append(0 if attr.get("active", "true") == "false" else 1)
        E["solref"].append(_floats(attr.get("solref"), 2, _DEF_SOLREF))
        E["solimp"].append(_floats(attr.get("solimp"), 5, _DEF_SOLIMP))
        E["data"].append(None)  # filled after qpos0 kinematics
    neq = len(E["obj1"])

    # ---- contact excludes --------------------------------------------------
    excl = set()
    con = root.find("contact")
    for c in (list(con) if con is not None else []):
        if c.tag == "exclude":
            i1, i2 = body_names.index(c.get("body1")), body_names.index(c.get("body2"))
            excl.add((min(i1, i2), max(i1, i2)))
        elif c.tag == "pair":
            raise NotImplementedError("explicit contact pairs unused by in-scope models")

    # ---- pack ------------------------------------------------------------
    m.nq, m.nv, m.nu, m.nbody, m.njnt, m.ngeom, m.nsite, m.neq = nq, nv, nu, nbody, njnt, ngeom, nsite, neq
    m.qpos0 = np.array(qpos0)
    m.body_parentid = body_parent
    m.body_rootid = body_rootid
    m.body_weldid = body_weldid
    m.body_jntadr = np.array(body_jntadr, dtype=np.int32)
    m.body_jntnum = np.array(body_jntnum, dtype=np.int32)
    m.body_dofadr = np.array(body_dofadr, dtype=np.int32)
    m.body_dofnum = np.array(body_dofnum, dtype=np.int32)
    m.body_pos, m.body_quat = body_pos, body_quat
    m.body_ipos, m.body_iquat, m.body_mass, m.body_inertia = body_ipos, body_iquat, body_mass, body_inertia
    m.jnt_type = np.array(jnt["type"], dtype=np.int32)
    m.jnt_qposadr = np.array(jnt["qposadr"], dtype=np.int32)
    m.jnt_dofadr = np.array(jnt["dofadr"], dtype=np.int32)
    m.jnt_bodyid = np.array(jnt["bodyid"], dtype=np.int32)
    m.jnt_pos = np.array(jnt["pos"]).reshape(njnt, 3)
    m.jnt_axis = np.array(jnt["axis"]).reshape(njnt, 3)
    m.jnt_limited = np.array(jnt["limited"], dtype=np.int32)
    m.jnt_range = np.array(jnt["range"]).reshape(njnt, 2)
    m.jnt_margin = np.array(jnt["margin"])
    m.jnt_solref = np.array(jnt["solref"]).reshape(njnt, 2)
    m.jnt_solimp = np.array(jnt["solimp"]).reshape(njnt, 5)
    m.dof_bodyid = np.array(dof["bodyid"], dtype=np.int32)
    m.dof_jntid = np.array(dof["jntid"], dtype=np.int32)
    m.dof_parentid = np.array(dof["parentid"], dtype=np.int32)
    m.dof_armature = np.array(dof["armature"])
    m.dof_damping = np.array(dof["damping"])
    m.geom_type = np.array(G["type"], dtype=np.int32)
    m.geom_bodyid = np.array(G["bodyid"], dtype=np.int32)
    m.geom_contype = np.array(G["contype"], dtype=np.int32)
    m.geom_conaffinity = np.array(G["conaffinity"], dtype=np.int32)
    m.geom_condim = np.array(G["condim"], dtype=np.int32)
    m.geom_size = np.array(G["size"]).reshape(ngeom, 3)
    m.geom_pos = np.array(G["pos"]).reshape(ngeom, 3)
    m.geom_quat = np.array(G["quat"]).reshape(ngeom, 4)
    m.geom_friction = np.array(G["friction"]).reshape(ngeom, 3)
    m.geom_solref = np.array(G["solref"]).reshape(ngeom, 2)
    m.geom_solimp = np.array(G["solimp"]).reshape(ngeom, 5)
    m.geom_margin = np.array(G["margin"])
    m.geom_gap = np.array(G["gap"])
    m.geom_rbound = np.array(G["rbound"])
    m.geom_meshadr = np.array(mesh_adr, dtype=np.int32)  # first hull vertex of a colliding mesh geom in mesh_vert, -1 = none
    m.geom_meshnum = np.array(mesh_num, dtype=np.int32)
    m.mesh_vert = np.concatenate(mesh_vert).reshape(-1, 3) if mesh_vert else np.zeros((0, 3))  # hull vertices, geom frame
    m.geom_solmix = np.array(G["solmix"])
    m.geom_priority = np.array(G["priority"], dtype=np.int32)
    m.site_bodyid = np.array(S["bodyid"], dtype=np.int32)
    m.site_pos = np.array(S["pos"]).reshape(nsite, 3)
    m.site_quat = np.array(S["quat"]).reshape(nsite, 4)
    m.actuator_jntid = np.array(A["jntid"], dtype=np.int32)
    m.actuator_gain = np.array(A["gain"])
    m.actuator_bias = np.array(A["bias"]).reshape(nu, 3)
    m.actuator_ctrllimited = np.array(A["ctrllimited"], dtype=np.int32)
    m.actuator_ctrlrange = np.array(A["ctrlrange"]).reshape(nu, 2)
    m.actuator_forcelimited = np.array(A["forcelimited"], dtype=np.int32)
    m.actuator_forcerange = np.array(A["forcerange"]).reshape(nu, 2)
    m.actuator_gear = np.array(A["gear"])
    m.eq_obj1id = np.array(E["obj1"], dtype=np.int32)
    m.eq_obj2id = np.array(E["obj2"], dtype=np.int32)
    m.eq_active0 = np.array(E["active"], dtype=np.int32)
    m.eq_solref = np.array(E["solref"]).reshape(neq, 2)
    m.eq_solimp = np.array(E["solimp"]).reshape(neq, 5)
    m.body_names = body_names
    m.joint_names = jnt["name"]
    m.geom_names = G["name"]
    m.site_names = S["name"]
    m.actuator_names = A["name"]

    # sparse-M addressing (mj: dof_Madr; row i holds M[i, i], M[i, parent(i)], ...)
    madr, n = [], 0
    for i in range(nv):
        madr.append(n)
        j = i
        while j >= 0:
            n += 1
            j = m.dof_parentid[j]
    m.dof_Madr = np.array(madr, dtype=np.int32)
    m.nM = n

    _derive_at_qpos0(m)

    # weld relpose at qpos0 (overwritten by the env at attach time, furniture.py:2772)
    xpos, xquat = m._xpos0, m._xquat0
    data = np.zeros((neq, 7))
    for k in range(neq):
        b1, b2 = m.eq_obj1id[k], m.eq_obj2id[k]
        R1 = q2m(xquat[b1])
        data[k, :3] = R1.T @ (xpos[b2] - xpos[b1])
        data[k, 3:] = qmul(qconj(xquat[b1]), xquat[b2])
    m.eq_data0 = data

    # candidate collision pairs (g1 < g2), static filters only; the mutable
    # contype/conaffinity test is applied per env at run time (furniture.py:866-878)
    pairs = []
    for g1 in range(ngeom):
        for g2 in range(g1 + 1, ngeom):
            b1, b2 = m.geom_bodyid[g1], m.geom_bodyid[g2]
            w1, w2 = body_weldid[b1], body_weldid[b2]
            if w1 == w2:
                continue
            wp1, wp2 = body_weldid[body_parent[w1]], body_weldid[body_parent[w2]]
            if w1 != 0 and w2 != 0 and (w1 == wp2 or w2 == wp1):
                continue
            if (min(b1, b2), max(b1, b2)) in excl:
                continue
            ct1, ca1, ct2, ca2 = G["contype"][g1], G["conaffinity"][g1], G["contype"][g2], G["conaffinity"][g2]
            part_like = lambda nm: "collision" in nm  # geoms the env may re-enable (furniture.py:1455-1461)
            static_ok = bool((ct1 & ca2) or (ct2 & ca1))
            if not static_ok:
                # furniture colliders are forced to 1/1 at reset, so keep pairs that become
                # legal after that rewrite
                e1 = (1, 1) if (part_like(G["name"][g1]) and b1 in _part_body_ids(m)) else (ct1, ca1)
                e2 = (1, 1) if (part_like(G["name"][g2]) and b2 in _part_body_ids(m)) else (ct2, ca2)
                if not ((e1[0] & e2[1]) or (e2[0] & e1[1])):
                    continue
            if (G["type"][g1] == GEOM_MESH and mesh_adr[g1] < 0) or (G["type"][g2] == GEOM_MESH and mesh_adr[g2] < 0):
                continue  # a visual mesh (contype = conaffinity = 0): never collides
            pairs.append((g1, g2))
    m.pair_geom = np.array(pairs, dtype=np.int32).reshape(-1, 2)
    m.npair = len(pairs)
    return m


def _part_body_ids(m):
    """bodies that carry a free joint (furniture parts)."""
    if not hasattr(m, "_part_bodies"):
        m._part_bodies = set(int(m.jnt_bodyid[j]) for j in range(m.njnt) if m.jnt_type[j] == JNT_FREE)
    return m._part_bodies


# ---------------------------------------------------------------------------
# qpos0-dependent constants: kinematics, mass matrix, invweight0
# ---------------------------------------------------------------------------

def kinematics(m, qpos):
    """Reference-order forward kinematics in numpy (used at compile time and by tests)."""
    nb = m.nbody
    xpos = np.zeros((nb, 3))
    xquat = np.tile(np.array([1.0, 0, 0, 0]), (nb, 1))
    xanchor = np.zeros((m.njnt, 3))
    xaxis = np.zeros((m.njnt, 3))
    for i in range(1, nb):
        p = m.body_parentid[i]
        jn, ja = m.body_jntnum[i], m.body_jntadr[i]
        if jn == 1 and m.jnt_type[ja] == JNT_FREE:
            a = m.jnt_qposadr[ja]
            pos = qpos[a:a + 3].copy()
            quat = qpos[a + 3:a + 7] / np.linalg.norm(qpos[a + 3:a + 7])
            xanchor[ja] = pos
            xaxis[ja] = np.array([0, 0, 1.0])
        else:
            R = q2m(xquat[p])
            pos = xpos[p] + R @ m.body_pos[i]
            quat = qmul(xquat[p], m.body_quat[i])
            for j in range(ja, ja + jn):
                Rb = q2m(quat)
                xanchor[j] = pos + Rb @ m.jnt_pos[j]
                xaxis[j] = Rb @ m.jnt_axis[j]
                q = qpos[m.jnt_qposadr[j]] - m.qpos0[m.jnt_qposadr[j]]
                if m.jnt_type[j] == JNT_SLIDE:
                    pos = pos + xaxis[j] * q
                elif m.jnt_type[j] == JNT_HINGE:
                    quat = qmul(quat, _axis_quat(m.jnt_axis[j], q))
                    pos = xanchor[j] - q2m(quat) @ m.jnt_pos[j]
        xpos[i] = pos
        xquat[i] = quat / np.linalg.norm(quat)
    return xpos, xquat, xanchor, xaxis


def _skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def body_jacobians(m, qpos, points=None):
    """Dense (6*nbody, nv) Jacobian [lin; ang] of each body at its CoM (or given points)."""
    xpos, xquat, xanchor, xaxis = kinematics(m, qpos)
    nb, nv = m.nbody, m.nv
    J = np.zeros((nb, 6, nv))
    for b in range(1, nb):
        Rb = q2m(xquat[b])
        pt = xpos[b] + Rb @ m.body_ipos[b] if points is None else points[b]
        a = b
        while a != 0:
            for j in range(m.body_jntadr[a], m.body_jntadr[a] + m.body_jntnum[a]) if m.body_jntnum[a] else []:
                d = m.jnt_dofadr[j]
                if m.jnt_type[j] == JNT_FREE:
                    Ra = q2m(xquat[a])
                    J[b, 0:3, d:d + 3] = np.eye(3)
                    for k in range(3):
                        ax = Ra[:, k]
                        J[b, 3:6, d + 3 + k] = ax
                        J[b, 0:3, d + 3 + k] = np.cross(ax, pt - xpos[a])
                elif m.jnt_type[j] == JNT_SLIDE:
                    J[b, 0:3, d] = xaxis[j]
                else:
                    J[b, 3:6, d] = xaxis[j]
                    J[b, 0:3, d] = np.cross(xaxis[j], pt - xanchor[j])
            a = m.body_parentid[a]
    return J, xpos, xquat


def mass_matrix(m, qpos):
    """Dense M(q) from body Jacobians (slow, compile-time / test use)."""
    J, xpos, xquat = body_jacobians(m, qpos)
    M = np.diag(m.dof_armature.astype(np.float64).copy())
    for b in range(1, m.nbody):
        if m.body_mass[b] == 0 and not np.any(m.body_inertia[b]):
            continue
        Ri = q2m(qmul(xquat[b], m.body_iquat[b]))
        Iw = Ri @ np.diag(m.body_inertia[b]) @ Ri.T
        Jl, Ja = J[b, 0:3], J[b, 3:6]
        M += m.body_mass[b] * Jl.T @ Jl + Ja.T @ Iw @ Ja
    return M


def _derive_at_qpos0(m):
    xpos, xquat, _, _ = kinematics(m, m.qpos0)
    m._xpos0, m._xquat0 = xpos, xquat
    nv = m.nv
    m.body_invweight0 = np.zeros((m.nbody, 2))
    m.dof_invweight0 = np.zeros(nv)
    m.trace_M0 = np.array([1.0])
    if nv == 0:
        return
    M = mass_matrix(m, m.qpos0)
    m.trace_M0 = np.array([np.trace(M)])
    Minv = np.linalg.inv(M)
    J, _, _ = body_jacobians(m, m.qpos0)
    for b in range(1, m.nbody):
        if m.body_weldid[b] == 0:
            continue
        A = J[b] @ Minv @ J[b].T
        m.body_invweight0[b, 0] = max(MJ_MINVAL, np.trace(A[0:3, 0:3]) / 3)
        m.body_invweight0[b, 1] = max(MJ_MINVAL, np.trace(A[3:6, 3:6]) / 3)
    d = np.diag(Minv).copy()
    for j in range(m.njnt):
        a = m.jnt_dofadr[j]
        if m.jnt_type[j] == JNT_FREE:
            d[a:a + 3] = d[a:a + 3].mean()
            d[a + 3:a + 6] = d[a + 3:a + 6].mean()
    m.dof_invweight0 = np.maximum(d, MJ_MINVAL)
