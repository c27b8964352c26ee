"""Shared test scenarios (pure numpy; no GPU, no oracle import here)."""

import numpy as np

from furniture_amd import transform_utils as T


def quat_from_axes(x, y, z):
    """wxyz quaternion of the rotation whose columns are (x, y, z)."""
    R = np.stack([x, y, z], axis=1)
    q = T.mat2quat(R)  # xyzw
    return np.array([q[3], q[0], q[1], q[2]])


def pinch_attach_state(m, qpos, body_xpos, body_xquat, leg=0, table_conn=4, leg_conn=0, gap=0.02):
    """Build a state in which the gripper pinches furniture part ``leg`` whose connector already faces the matching
    connector of the table (Sawyer + table_lack_0825): all parts float (xfrc = +m g), the leg lies between the nearly
    closed fingers, and the table hovers with its connector ``gap`` metres away along the connector axis.

    qpos: post-reset qpos (nq,), body_xpos/xquat: poses of ORIGINAL bodies from the last forward pass.
    The table's collider is given contype = conaffinity = 2 so that it is a ghost for the robot, the floor and the leg
    (the 0.64 m top would otherwise hit arm links in most arm poses); _connect() rewrites the masks of both groups anyway.
    Returns (qpos, xfrc_applied[nparts*6], {geom id: (contype, conaffinity)})."""
    names = m.meta["body_names"]
    gb = names.index("right_gripper_base")
    Rg = T.quat2mat_wxyz(body_xquat[gb] / np.linalg.norm(body_xquat[gb]))
    pg = body_xpos[gb]
    gx, gy, gz = Rg[:, 0], Rg[:, 1], Rg[:, 2]
    q = qpos.copy()
    # fingers almost closed on a 0.03 m wide leg
    q[m.grip_qposadr[0]] = -0.002   # l finger joint
    q[m.grip_qposadr[1]] = 0.002    # r finger joint
    # leg: long axis (local z) along gripper x, centred between the finger tips
    leg_q = quat_from_axes(gy, gz, gx)  # local x->gy, y->gz, z->gx
    leg_p = pg + gz * 0.08
    a = m.part_qposadr[leg]
    q[a:a + 3], q[a + 3:a + 7] = leg_p, leg_q
    Rl = T.quat2mat_wxyz(leg_q)
    leg_site_w = leg_p + Rl @ m.site_pos[m.conn_siteid[leg_conn]]
    up = Rl[:, 2]
    # table: same "up" (local z) as the leg, local x chosen among the leg's +-x/+-y so that the 0.64 m long top
    # extends upwards, away from the floor (forward vectors then differ by a multiple of 90 degrees, which is allowed)
    tpart = int(m.conn_partid[table_conn])
    ta = m.part_qposadr[tpart]
    cands = [gy, -gy, gz, -gz]
    tx = cands[int(np.argmax([c[2] for c in cands]))]
    Rt = np.stack([tx, np.cross(up, tx), up], axis=1)
    table_q = quat_from_axes(Rt[:, 0], Rt[:, 1], Rt[:, 2])
    table_site_w = leg_site_w + gap * up
    q[ta:ta + 3] = table_site_w - Rt @ m.site_pos[m.conn_siteid[table_conn]]
    q[ta + 3:ta + 7] = table_q
    # park the other parts far away, floating
    k = 0
    for i in range(m.nparts):
        if i in (leg, tpart):
            continue
        b = m.part_qposadr[i]
        q[b:b + 3] = [1.0 + 0.3 * k, -1.0, 0.5]
        q[b + 3:b + 7] = [1, 0, 0, 0]
        k += 1
    xfrc = np.zeros((m.nparts, 6))
    for i in range(m.nparts):
        xfrc[i, 2] = 9.81 * m.body_mass[m.part_bodyid[i]]
    masks = {g: (2, 2) for g in range(m.ngeom) if m.geom_is_partcol[g] and m.body_partid[m.geom_bodyid[g]] == tpart}
    return q, xfrc.reshape(-1), masks


def counter_actions(seed, env_index, t, dof):
    """U(-1,1)^dof float32 keyed by (seed, env, t) -- deterministic and independent of batch size / GPU count."""
    rs = np.random.RandomState((seed * 1000003 + env_index * 7919 + t * 104729) % (2 ** 31 - 1))
    return rs.uniform(-1, 1, dof).astype(np.float32)


def cursor_attach_state(m, qpos, leg=0, table_conn=4, leg_conn=0, gap=0.03):
    """Cursor + table_lack_0825: part ``leg`` hovers horizontally at (0, 0.3, 0.5) and the table hovers with its matching
    connector ``gap`` metres from the leg's connector along the connector axis, forward vectors a multiple of 90 degrees
    apart (an aligned pair for _is_aligned).  The other parts keep their poses.  Returns (qpos, table part index)."""
    gx, gy, gz = np.eye(3)
    q = qpos.copy()
    leg_q = quat_from_axes(gy, gz, gx)  # leg's local z (its connector's up axis) along world x
    leg_p = np.array([0.0, 0.3, 0.5])
    a = m.part_qposadr[leg]
    q[a:a + 3], q[a + 3:a + 7] = leg_p, leg_q
    Rl = T.quat2mat_wxyz(leg_q)
    leg_site_w = leg_p + Rl @ m.site_pos[m.conn_siteid[leg_conn]]
    up = Rl[:, 2]
    tpart = int(m.conn_partid[table_conn])
    ta = m.part_qposadr[tpart]
    tx = gz
    Rt = np.stack([tx, np.cross(up, tx), up], axis=1)
    table_q = quat_from_axes(Rt[:, 0], Rt[:, 1], Rt[:, 2])
    table_site_w = leg_site_w + gap * up
    q[ta:ta + 3] = table_site_w - Rt @ m.site_pos[m.conn_siteid[table_conn]]
    q[ta + 3:ta + 7] = table_q
    return q, tpart


def spread_layout(m, x0=0.9, gap=0.03):
    """Part poses [nparts, 7] that lay every part flat on the floor without touching another part or the robot: for furniture whose XML
    stacks all parts at the origin (bookcase_grevback_0484: fourteen planks, no *_initpos entries), where the reference's sampler -- 5 mm
    placement radii, 2 cm jitter -- starts the episode with the planks inside each other.  Each part is turned so that the thinnest
    extent of its first collision box points up and put on a grid row by row, rows along y, starting x0 metres in front of the robot."""
    cg_body, cg_pos = np.asarray(m.cg_body), np.asarray(m.cg_pos).reshape(-1, 3)
    cg_mat, cg_size, cg_rb = np.asarray(m.cg_mat).reshape(-1, 3, 3), np.asarray(m.cg_size).reshape(-1, 3), np.asarray(m.cg_rbound)
    out = np.zeros((m.nparts, 7))
    x, y, row_h = x0, -1.2, 0.0
    for p in range(m.nparts):
        gs = [g for g in range(len(cg_body)) if cg_body[g] == m.part_rbody[p]]
        g0 = gs[0]
        thin = int(np.argmin(cg_size[g0]))
        # rotation taking the geom's thinnest local axis to world z (and the other two to x, y)
        axes = [a for a in range(3) if a != thin] + [thin]
        Rg = cg_mat[g0][:, axes]            # columns: geom axes (in the body frame) that shall become world x, y, z
        if np.linalg.det(Rg) < 0:
            Rg[:, 0] = -Rg[:, 0]
        Rb = Rg.T                           # body -> world
        q = quat_from_axes(Rb[:, 0], Rb[:, 1], Rb[:, 2])
        rad = max(np.linalg.norm(cg_pos[g]) + cg_rb[g] for g in gs)
        if y + 2 * rad > 1.2 and y > -1.2:
            x, y, row_h = x + row_h + gap, -1.2, 0.0
        centre = Rb @ cg_pos[g0]
        out[p, :3] = [x + rad - centre[0], y + rad - centre[1], cg_size[g0][thin] + 0.002 - centre[2]]
        out[p, 3:] = q
        y += 2 * rad + gap
        row_h = max(row_h, 2 * rad)
    return out


def stacked_layout(m, x=1.0, y=0.0, gap=0.001):
    """Part poses [nparts, 7] that put every part flat ON TOP of the previous one, one pile a metre in front of the robot: every part touches its
    neighbours, so the whole furniture is ONE constraint island (6 x nparts dofs) with a few contacts per part -- for bookcase_billy_0191 66 dofs and
    about 45 contacts: a system that fits the contact slots of its kernel but not the 64 lanes of its island map (orientation as in spread_layout)."""
    lay = spread_layout(m)
    cg_body, cg_size = np.asarray(m.cg_body), np.asarray(m.cg_size).reshape(-1, 3)
    out, z = lay.copy(), 0.0
    for p in range(m.nparts):
        g0 = [g for g in range(len(cg_body)) if cg_body[g] == m.part_rbody[p]][0]
        h = cg_size[g0][int(np.argmin(cg_size[g0]))]
        out[p, 0], out[p, 1] = x, y
        out[p, 2] = lay[p, 2] - (h + 0.002) + z + h + gap  # (spread_layout put the first box's centre at height h + 0.002)
        z += 2 * h + gap
    return out
