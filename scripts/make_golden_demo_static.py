"""Generate tests/golden/demo_static.npz: frames of the reference's bundled demos/Sawyer_7.pkl (recorded from MuJoCo: Sawyer +
swivel_chair_0700) in which the three chair parts lie at rest -- frame 10 and, 500 physics substeps later, frame 20.  The
recorded resting poses (base 7.0 mm, lying column 14.9 mm, seat 146 mm above the floor) are MuJoCo's equilibrium of geometry +
soft contact for these parts; tests/test_model_independent.py checks that today's compiled model has the same equilibrium.
Runs only in the build container (needs /root/reference)."""
import os
import pickle

import numpy as np

d = pickle.load(open("/root/reference/demos/Sawyer_7.pkl", "rb"))
q = d["qpos"]
parts = ["1_chair_base", "2_chair_column", "3_chair_seat"]
frames = [10, 20, 50]
out = dict(frames=np.array(frames), part_names=np.array(parts),
           parts=np.array([[q[f][p] for p in parts] for f in frames]),
           arm=np.array([q[f]["sawyer_qpos"] for f in frames]), grip=np.array([q[f]["l_gripper"] for f in frames]))
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "demo_static.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, {k: np.shape(v) for k, v in out.items()})
