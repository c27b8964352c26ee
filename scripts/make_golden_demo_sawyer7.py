"""Generate tests/golden/demo_sawyer7.npz from the reference's bundled demos/Sawyer_7.pkl: 460 frames RECORDED FROM MuJoCo of a Sawyer
assembling swivel_chair_0700 under IK control (one frame per env step = 3 x 50 physics substeps): arm joints, finger joints and the
pose of every part.  tests/test_demo_sawyer_replay.py drives the robot along the recorded joints and compares what the parts do
(held, released, dropped, at rest; grasped, lifted 35 cm, carried for 45 s) with what they did in MuJoCo.
demos/Baxter_0.pkl (369 frames) was recorded with a two-block furniture that is no longer among the reference's assets: not replayable.
Runs only in the build container (needs /root/reference)."""
import os
import pickle

import numpy as np

d = pickle.load(open("/root/reference/demos/Sawyer_7.pkl", "rb"))
q = d["qpos"]
parts = ["1_chair_base", "2_chair_column", "3_chair_seat"]
out = dict(part_names=np.array(parts), parts=np.array([[f[p] for p in parts] for f in q]), arm=np.array([f["sawyer_qpos"] for f in q]),
           grip=np.array([f["l_gripper"] for f in q]), actions=np.array(d["actions"], dtype=np.float64))
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "demo_sawyer7.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, {k: np.shape(v) for k, v in out.items()})
