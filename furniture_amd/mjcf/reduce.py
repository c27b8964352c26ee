"""Weld folding: collapse every body that has no joint into its nearest moving
ancestor, producing the *reduced* articulated model the HIP kernels integrate.

A body without joints is rigidly attached to its parent, so its mass/inertia can
be composed into the parent's and its geoms/sites re-expressed in the parent's
frame without changing the dynamics (Sawyer + table_lack: 36 bodies -> 14 moving
bodies, kinematic depth 13 -> 8).  The CPU oracle integrates the *unreduced*
model, so GPU-vs-oracle agreement also checks this transformation.

All arrays produced here are prefixed ``r_`` (reduced bodies), ``cg_`` (colliding
geoms), ``cp_`` (candidate pairs) or ``s_`` (sites) and are appended to the
CompiledModel's array table.
"""

import numpy as np

from .compile import (GEOM_BOX, GEOM_CAPSULE, GEOM_CYLINDER, GEOM_MESH, GEOM_PLANE, GEOM_SPHERE, JNT_FREE, JNT_HINGE, JNT_SLIDE, q2m, qmul)

# pair-type codes consumed by the narrow phase kernel
PT_PLANE_SPHERE, PT_PLANE_BOX, PT_PLANE_CYL, PT_SPHERE_SPHERE, PT_SPHERE_BOX, PT_SPHERE_CYL, PT_BOX_BOX, PT_CYL_BOX, PT_CYL_CYL, PT_PLANE_CAP, PT_CONVEX, PT_PLANE_MESH = range(12)
_PAIR_CODE = {
    (GEOM_PLANE, GEOM_SPHERE): PT_PLANE_SPHERE, (GEOM_PLANE, GEOM_BOX): PT_PLANE_BOX, (GEOM_PLANE, GEOM_CYLINDER): PT_PLANE_CYL,
    (GEOM_SPHERE, GEOM_SPHERE): PT_SPHERE_SPHERE, (GEOM_SPHERE, GEOM_BOX): PT_SPHERE_BOX, (GEOM_SPHERE, GEOM_CYLINDER): PT_SPHERE_CYL,
    (GEOM_BOX, GEOM_BOX): PT_BOX_BOX, (GEOM_CYLINDER, GEOM_BOX): PT_CYL_BOX, (GEOM_CYLINDER, GEOM_CYLINDER): PT_CYL_CYL,
    # capsules (round 5; one in the in-scope assets: Baxter's pedestal, robots/baxter/robot.xml:61): the two end spheres against a plane,
    # every other pair through the Minkowski-portal routine with the capsule's support function (a sphere swept along a segment)
    (GEOM_PLANE, GEOM_CAPSULE): PT_PLANE_CAP, (GEOM_SPHERE, GEOM_CAPSULE): PT_CONVEX, (GEOM_CAPSULE, GEOM_CAPSULE): PT_CONVEX,
    (GEOM_CAPSULE, GEOM_CYLINDER): PT_CONVEX, (GEOM_CAPSULE, GEOM_BOX): PT_CONVEX,
    # convex meshes (round 5; three furniture collide mesh geoms: chair_agne_0010, chair_bertil_0148, shelf_liden_0922): the mesh's convex
    # hull -- vertices in the geom frame, tables mesh_vert / cg_meshadr / cg_meshnum -- against a plane (its lowest vertices), every other
    # pair through the portal routine with the hull's support function (arg max over its vertices)
    (GEOM_PLANE, GEOM_MESH): PT_PLANE_MESH, (GEOM_SPHERE, GEOM_MESH): PT_CONVEX, (GEOM_CAPSULE, GEOM_MESH): PT_CONVEX,
    (GEOM_CYLINDER, GEOM_MESH): PT_CONVEX, (GEOM_BOX, GEOM_MESH): PT_CONVEX, (GEOM_MESH, GEOM_MESH): PT_CONVEX,
}
PAIR_MAXCON = [1, 4, 4, 1, 1, 1, 8, 1, 1, 2, 1, 4]


# Colliders whose pairs are waived (not collided) instead of failing the compilation: none since round 5 (Baxter's pedestal capsule,
# robots/baxter/robot.xml:61, collides like every other primitive; rounds 2-4 waived it by name).
WAIVED_COLLIDERS = set()


def reduce_model(A):
    """A: dict of compiled arrays (see compile.compile_mjcf / model.build_model). Returns dict of new arrays."""
    nbody = len(A["body_parentid"])
    parent = A["body_parentid"]
    jntnum, jntadr = A["body_jntnum"], A["body_jntadr"]
    if np.any(jntnum > 1):
        raise NotImplementedError("reduced model assumes at most one joint per body")

    # pose of every body in the frame of its weld (moving) ancestor, at qpos0-independent level
    moving = [b for b in range(1, nbody) if jntnum[b] == 1]
    red_of = -np.ones(nbody, dtype=np.int32)   # reduced index of the body each body is folded into
    red_of[0] = 0
    rel_pos = np.zeros((nbody, 3))
    rel_quat = np.tile(np.array([1.0, 0, 0, 0]), (nbody, 1))
    ridx = {0: 0}
    for k, b in enumerate(moving):
        ridx[b] = k + 1
    for b in range(1, nbody):
        if jntnum[b] == 1:
            red_of[b] = ridx[b]
        else:
            p = parent[b]
            red_of[b] = red_of[p]
            R = q2m(rel_quat[p])
            rel_pos[b] = rel_pos[p] + R @ A["body_pos"][b]
            rel_quat[b] = qmul(rel_quat[p], A["body_quat"][b])
    nr = len(moving) + 1

    r_parent = np.zeros(nr, dtype=np.int32)
    r_pos = np.zeros((nr, 3))
    r_quat = np.tile(np.array([1.0, 0, 0, 0]), (nr, 1))
    r_jtype = -np.ones(nr, dtype=np.int32)
    r_jaxis = np.zeros((nr, 3))
    r_jpos = np.zeros((nr, 3))
    r_qposadr = np.zeros(nr, dtype=np.int32)
    r_dofadr = np.zeros(nr, dtype=np.int32)
    r_dofnum = np.zeros(nr, dtype=np.int32)
    r_depth = np.zeros(nr, dtype=np.int32)
    r_orig = np.zeros(nr, dtype=np.int32)
    for b in moving:
        k = ridx[b]
        p = parent[b]
        r_parent[k] = red_of[p]
        R = q2m(rel_quat[p])
        r_pos[k] = rel_pos[p] + R @ A["body_pos"][b]
        r_quat[k] = qmul(rel_quat[p], A["body_quat"][b])
        j = jntadr[b]
        r_jtype[k] = A["jnt_type"][j]
        r_jaxis[k] = A["jnt_axis"][j]
        r_jpos[k] = A["jnt_pos"][j]
        r_qposadr[k] = A["jnt_qposadr"][j]
        r_dofadr[k] = A["jnt_dofadr"][j]
        r_dofnum[k] = 6 if A["jnt_type"][j] == JNT_FREE else 1
        r_depth[k] = r_depth[r_parent[k]] + 1
        r_orig[k] = b
        if A["jnt_type"][j] in (JNT_HINGE, JNT_SLIDE) and A["qpos0"][A["jnt_qposadr"][j]] != 0:
            raise NotImplementedError("nonzero joint ref")

    # composite inertia of each reduced body, in its own frame about its own CoM
    r_mass = np.zeros(nr)
    r_ipos = np.zeros((nr, 3))
    r_inertia = np.zeros((nr, 6))  # xx yy zz xy xz yz
    accum = [[] for _ in range(nr)]
    for b in range(1, nbody):
        k = red_of[b]
        if k == 0 or A["body_mass"][b] <= 0:
            continue
        Rb = q2m(rel_quat[b]) if jntnum[b] == 0 else np.eye(3)
        pb = rel_pos[b] if jntnum[b] == 0 else np.zeros(3)
        com = pb + Rb @ A["body_ipos"][b]
        Ri = Rb @ q2m(A["body_iquat"][b])
        I = Ri @ np.diag(A["body_inertia"][b]) @ Ri.T
        accum[k].append((A["body_mass"][b], com, I))
    for k in range(1, nr):
        tot = sum(m for m, _, _ in accum[k])
        if tot <= 0:
            continue
        c = sum(m * com for m, com, _ in accum[k]) / tot
        I = np.zeros((3, 3))
        for m, com, Ib in accum[k]:
            d = com - c
            I += Ib + m * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
        r_mass[k], r_ipos[k] = tot, c
        r_inertia[k] = [I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]]

    # kinematic trees over reduced bodies (root = child of world), dof ranges are contiguous
    r_tree = np.zeros(nr, dtype=np.int32)
    tree_root = []
    for k in range(1, nr):
        if r_parent[k] == 0:
            tree_root.append(k)
            r_tree[k] = len(tree_root) - 1
        else:
            r_tree[k] = r_tree[r_parent[k]]
    ntree = len(tree_root)
    nv = len(A["dof_bodyid"])
    dof_tree = np.array([r_tree[red_of[A["dof_bodyid"][d]]] for d in range(nv)], dtype=np.int32)
    tree_dofadr = np.array([int(np.argmax(dof_tree == t)) for t in range(ntree)], dtype=np.int32)
    tree_dofnum = np.array([int((dof_tree == t).sum()) for t in range(ntree)], dtype=np.int32)
    for t in range(ntree):
        assert np.all(dof_tree[tree_dofadr[t]:tree_dofadr[t] + tree_dofnum[t]] == t)
    dof_rbody = red_of[A["dof_bodyid"]].astype(np.int32)

    # per-dof limit tables (hinge/slide only)
    dof_limited = np.zeros(nv, dtype=np.int32)
    dof_range = np.zeros((nv, 2))
    dof_lmargin = np.zeros(nv)
    dof_lsolref = np.zeros((nv, 2))
    dof_lsolimp = np.zeros((nv, 5))
    dof_qposadr = np.zeros(nv, dtype=np.int32)
    for j in range(len(A["jnt_type"])):
        d = A["jnt_dofadr"][j]
        if A["jnt_type"][j] in (JNT_HINGE, JNT_SLIDE):
            dof_limited[d] = A["jnt_limited"][j]
            dof_range[d] = A["jnt_range"][j]
            dof_lmargin[d] = A["jnt_margin"][j]
            dof_lsolref[d] = A["jnt_solref"][j]
            dof_lsolimp[d] = A["jnt_solimp"][j]
            dof_qposadr[d] = A["jnt_qposadr"][j]
        else:
            dof_qposadr[d:d + 6] = A["jnt_qposadr"][j]

    # colliding geoms (anything a candidate pair references)
    pair = A["pair_geom"]
    used = sorted(set(pair.reshape(-1).tolist()))
    cg_of = {g: i for i, g in enumerate(used)}
    ncg = len(used)
    cg = dict(orig=np.array(used, dtype=np.int32))
    gb = A["geom_bodyid"]
    cg["body"] = np.array([red_of[gb[g]] for g in used], dtype=np.int32)
    cpos, cquat = np.zeros((ncg, 3)), np.zeros((ncg, 4))
    for i, g in enumerate(used):
        b = gb[g]
        if jntnum[b] == 1:
            cpos[i], cquat[i] = A["geom_pos"][g], A["geom_quat"][g]
        else:
            R = q2m(rel_quat[b])
            cpos[i] = rel_pos[b] + R @ A["geom_pos"][g]
            cquat[i] = qmul(rel_quat[b], A["geom_quat"][g])
    cg["pos"], cg["quat"] = cpos, cquat
    # convex-mesh colliders: where each colliding geom's hull vertices sit in mesh_vert (geom frame: unchanged by the body reduction)
    if "geom_meshadr" in A:
        cg["meshadr"] = A["geom_meshadr"][used].astype(np.int32)
        cg["meshnum"] = A["geom_meshnum"][used].astype(np.int32)
    else:  # (tables compiled before round 5)
        cg["meshadr"] = -np.ones(ncg, dtype=np.int32)
        cg["meshnum"] = np.zeros(ncg, dtype=np.int32)
    for nm in ("type", "condim"):
        cg[nm] = A["geom_" + nm][used].astype(np.int32)
    for nm in ("size", "rbound", "friction", "solref", "solimp", "margin", "gap", "solmix"):
        cg[nm] = A["geom_" + nm][used]
    cg["invweight"] = A["body_invweight0"][gb[used], 0]
    cg["partid"] = A["body_partid"][gb[used]].astype(np.int32)
    cg["fingerrole"] = A["geom_fingerrole"][used].astype(np.int32)
    cg["isfloor"] = (np.array(used) == A["floor_geomid"][0]).astype(np.int32)
    cg["isrobot"] = A["geom_is_robot"][used].astype(np.int32)
    cg["ispartcol"] = A["geom_is_partcol"][used].astype(np.int32)
    cg["contype0"] = A["geom_contype"][used].astype(np.int32)
    cg["conaffinity0"] = A["geom_conaffinity"][used].astype(np.int32)

    # candidate pairs, lower geom type first (contact normal points geom1 -> geom2)
    cp, dropped = [], []
    for g1, g2 in pair:
        t1, t2 = A["geom_type"][g1], A["geom_type"][g2]
        if t1 > t2:
            g1, g2, t1, t2 = g2, g1, t2, t1
        code = _PAIR_CODE.get((int(t1), int(t2)))
        if code is None:
            # a primitive pair the narrow phase has no routine for (capsule / ellipsoid / mesh).  Never dropped silently: the
            # only colliders waived are named here, with the reason; anything else fails the compilation like the other limits do.
            names = A.get("geom_names_list")
            nm = [str(names[g]) if names is not None else "geom%d" % g for g in (g1, g2)]
            if not any(n in WAIVED_COLLIDERS for n in nm):
                raise NotImplementedError("collision pair %s (type %d) x %s (type %d): no narrow-phase routine for this primitive pair "
                                          "(built: plane, sphere, cylinder, box)" % (nm[0], t1, nm[1], t2))
            dropped.append((nm[0], nm[1]))
            continue
        cp.append((cg_of[int(g1)], cg_of[int(g2)], code))
    cp = np.array(cp, dtype=np.int32).reshape(-1, 3)

    # sites -> reduced body frames
    ns = len(A["site_bodyid"])
    s_body = red_of[A["site_bodyid"]].astype(np.int32)
    s_pos, s_quat = np.zeros((ns, 3)), np.zeros((ns, 4))
    for k in range(ns):
        b = A["site_bodyid"][k]
        if jntnum[b] == 1:
            s_pos[k], s_quat[k] = A["site_pos"][k], A["site_quat"][k]
        else:
            R = q2m(rel_quat[b])
            s_pos[k] = rel_pos[b] + R @ A["site_pos"][k]
            s_quat[k] = qmul(rel_quat[b], A["site_quat"][k])

    # ---- tables the device kernels index directly --------------------------------------
    dof_parent = A["dof_parentid"]
    chain_adr, chain_len, chain = np.zeros(nr, np.int32), np.zeros(nr, np.int32), []
    ancmask = np.zeros(nr, dtype=np.int64)
    if nr > 32:  # (the world body is one of them: bit 0 is never set, bit 31 is the last body's)
        raise NotImplementedError("more than 31 moving bodies (ancestor bitmask is 32 bits)")
    for k in range(1, nr):
        lst = []
        d = r_dofadr[k] + r_dofnum[k] - 1
        while d >= 0:
            lst.append(d)
            d = dof_parent[d]
        lst.reverse()
        chain_adr[k], chain_len[k] = len(chain), len(lst)
        chain += lst
        a = k
        while a != 0:
            ancmask[k] |= 1 << a
            a = r_parent[a]
    tree_bodyadr = np.array([tree_root[t] for t in range(ntree)], dtype=np.int32)
    tree_bodynum = np.array([int((r_tree[1:] == t).sum()) for t in range(ntree)], dtype=np.int32)
    for t in range(ntree):
        assert np.all(r_tree[tree_bodyadr[t]:tree_bodyadr[t] + tree_bodynum[t]] == t)
    M_i, M_j = [], []
    for i in range(nv):
        j = i
        while j >= 0:
            M_i.append(i)
            M_j.append(j)
            j = dof_parent[j]
    lim = [d for d in range(nv) if dof_limited[d]]
    from .compile import q2m as _q2m
    cg_mat = np.array([_q2m(q).reshape(9) for q in cquat]).reshape(ncg, 9)
    part_body = A["part_bodyid"]
    part_sites, psa, psn = [], [], []
    for pb in part_body:
        ss = [k for k in range(ns) if A["site_bodyid"][k] == pb]
        psa.append(len(part_sites))
        psn.append(len(ss))
        part_sites += ss
    extra = dict(
        r_chainadr=chain_adr, r_chainlen=chain_len, chain_dofs=np.array(chain, dtype=np.int32),
        r_ancmask=ancmask.astype(np.uint32).view(np.int32), tree_bodyadr=tree_bodyadr, tree_bodynum=tree_bodynum,
        M_i=np.array(M_i, dtype=np.int32), M_j=np.array(M_j, dtype=np.int32),
        lim_dof=np.array(lim, dtype=np.int32), lim_range=dof_range[lim].reshape(-1, 2), lim_margin=dof_lmargin[lim],
        lim_solref=dof_lsolref[lim].reshape(-1, 2), lim_solimp=dof_lsolimp[lim].reshape(-1, 5),
        cg_mat=cg_mat, part_rbody=red_of[part_body].astype(np.int32), part_mass=A["body_mass"][part_body],
        part_site_adr=np.array(psa, dtype=np.int32), part_site_num=np.array(psn, dtype=np.int32),
        part_sites=np.array(part_sites, dtype=np.int32),
        hand_rbody=red_of[A["hand_bodyid"]].astype(np.int32) if len(A["hand_bodyid"]) else np.zeros(0, np.int32),
    )

    out = dict(
        r_parent=r_parent, r_pos=r_pos, r_quat=r_quat, r_jtype=r_jtype, r_jaxis=r_jaxis, r_jpos=r_jpos,
        r_qposadr=r_qposadr, r_dofadr=r_dofadr, r_dofnum=r_dofnum, r_depth=r_depth, r_orig=r_orig,
        r_mass=r_mass, r_ipos=r_ipos, r_inertia=r_inertia, r_tree=r_tree,
        tree_dofadr=tree_dofadr, tree_dofnum=tree_dofnum, tree_root=np.array(tree_root, dtype=np.int32),
        dof_tree=dof_tree, dof_rbody=dof_rbody, dof_limited=dof_limited, dof_range=dof_range, dof_lmargin=dof_lmargin,
        dof_lsolref=dof_lsolref, dof_lsolimp=dof_lsolimp, dof_qposadr=dof_qposadr,
        body_red=red_of, body_relpos=rel_pos, body_relquat=rel_quat,
        cp=cp, s_body=s_body, s_pos=s_pos, s_quat=s_quat,
        rdims=np.array([nr, ntree, ncg, len(cp), int(r_depth.max())], dtype=np.int32),
    )
    out.update(extra)
    for k, v in cg.items():
        out["cg_" + k] = v
    # welds between reduced bodies
    if len(A["eq_obj1id"]):
        out["eq_rbody1"] = red_of[A["eq_obj1id"]].astype(np.int32)
        out["eq_rbody2"] = red_of[A["eq_obj2id"]].astype(np.int32)
        iw = A["body_invweight0"]
        out["eq_invweight"] = np.stack([iw[A["eq_obj1id"], 0] + iw[A["eq_obj2id"], 0],
                                        iw[A["eq_obj1id"], 1] + iw[A["eq_obj2id"], 1]], axis=1)
    else:
        out["eq_rbody1"] = np.zeros(0, np.int32)
        out["eq_rbody2"] = np.zeros(0, np.int32)
        out["eq_invweight"] = np.zeros((0, 2))
    return out
