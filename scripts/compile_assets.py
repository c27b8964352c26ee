"""Compile the reference's MJCF assets into the flat tables shipped in furniture_amd/assets/compiled.

Run in the build container (needs /root/reference or FURNITURE_ASSETS_ROOT); the GPU box has no
reference tree, so the hot path loads these .npz files.  Usage:
    python scripts/compile_assets.py            # the BASELINE configs + stress models
    python scripts/compile_assets.py --all      # every furniture id for Sawyer
    python scripts/compile_assets.py --all --agents Sawyer,Baxter,Cursor   # ... for every agent (furniture/tests/test_furniture_init.py:14-55 is Baxter x 64)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from furniture_amd.mjcf import assemble, model  # noqa: E402

DEFAULT = [
    ("Sawyer", "table_lack_0825"), ("Sawyer", "swivel_chair_0700"), ("Sawyer", "toy_table"),
    ("Sawyer", "chair_agne_0007"), ("Sawyer", "shelf_ivar_0678"), ("Baxter", "desk_mikael_1064"),
    ("Baxter", "table_lack_0825"), ("Baxter", "bench_bjoderna_0208"),  # furniture id 1: IKEABaxter-v0's default (furniture/env/__init__.py:47-57)
    ("Cursor", "bed_dalselv_0270"),  # furniture id 0: IKEACursor-v0's default (furniture/env/__init__.py:19-29)
    ("Cursor", "toy_table"), ("Cursor", "table_lack_0825"), ("Cursor", "swivel_chair_0700"),
    # motor-actuated robot (robot_torque.xml) for the torque-level arm controllers (furniture.py:1893)
    ("Sawyer", "table_lack_0825", "joint_torque"), ("Sawyer", "swivel_chair_0700", "joint_torque"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--all", action="store_true")
    ap.add_argument("--agents", default="Sawyer", help="comma-separated agents that --all compiles every furniture for")
    args = ap.parse_args()
    root = assemble.default_assets_root()
    if root is None:
        raise SystemExit("no MJCF assets: set FURNITURE_ASSETS_ROOT")
    todo = list(DEFAULT)
    if args.all:
        _, names, _ = assemble.furniture_table(root)
        for agent in args.agents.split(","):
            todo += [(agent, n) for n in names if (agent, n) not in todo]
    out = model._COMPILED_DIR
    os.makedirs(out, exist_ok=True)
    for item in todo:
        agent, furn = item[0], item[1]
        ctype = item[2] if len(item) > 2 else "impedance"
        try:
            m = model.build_model(agent, furn, control_type=ctype, assets_root=root)
        except NotImplementedError as e:
            print("skip %s/%s: %s" % (agent, furn, e))
            continue
        path = os.path.join(out, model.compiled_name(agent, furn, ctype) + ".npz")
        m.save(path)
        print("%-8s %-28s nbody=%d nq=%d nv=%d ngeom=%d npair=%d  -> %s (%d KB)" % (
            agent, furn, m.nbody, m.nq, m.nv, m.ngeom, m.npair, os.path.basename(path), os.path.getsize(path) // 1024))
    # furniture id table (ids follow the sorted asset file names, furniture/env/models/__init__.py:10-21)
    _, names, _ = assemble.furniture_table(root)
    with open(os.path.join(out, "furniture_names.txt"), "w") as f:
        f.write("\n".join(names) + "\n")


if __name__ == "__main__":
    main()
