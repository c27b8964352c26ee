"""Generate tests/golden/demo_cursor7.npz from the reference's bundled demos/Cursor_7.pkl.

That pickle is the only material in the reference tree RECORDED FROM MuJoCo ITSELF: 172 frames of part / cursor poses and
the 171 Cursor-agent actions that produced them (swivel chair, move_speed 0.025, an older revision of the assets and env).
The first 62 frames (cursors wander, both select a part by contact, ten approach steps, the connect) are kept: up to there
the recording and today's assets agree on everything but the connector height and the yaw snapping of the seat.
Runs only in the build container (needs /root/reference)."""
import os
import pickle

import numpy as np

d = pickle.load(open("/root/reference/demos/Cursor_7.pkl", "rb"))
q, a = d["qpos"], np.array(d["actions"], dtype=np.float64)
N = 62
parts = ["1_chair_base", "2_chair_column", "3_chair_seat"]
NX = 92  # ... and 30 more frames in which the cursor carries the welded column + seat about (frames 62-91); at frame ~95 the recording
         # and today's env part company by two cursor steps (5 cm), so the replay ends there
ext = dict(actions_ext=a[:NX - 1], cursor0_ext=np.array([f["cursor0"] for f in q[:NX]]), cursor1_ext=np.array([f["cursor1"] for f in q[:NX]]),
           parts_ext=np.array([[f[p] for p in parts] for f in q[:NX]]))
out = dict(actions=a[:N - 1], cursor0=np.array([f["cursor0"] for f in q[:N]]), cursor1=np.array([f["cursor1"] for f in q[:N]]),
           parts=np.array([[f[p] for p in parts] for f in q[:N]]), part_names=np.array(parts))
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "demo_cursor7.npz")
out.update(ext)
np.savez_compressed(dst, **out)
print("wrote", dst, {k: np.shape(v) for k, v in out.items()})
