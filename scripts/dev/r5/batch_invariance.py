import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from tests.test_lds_clean_gpu import _run, ROOT
tool = ctypes.CDLL(os.path.join(ROOT, "tests", "liblds_poison.so"))
for agent, furn in [("Cursor", "toy_table"), ("Cursor", "table_lack_0825"), ("Sawyer", "toy_table"), ("Baxter", "desk_mikael_1064"), ("Sawyer", "chair_bertil_0148")]:
    a, k1 = _run(agent, furn, 40, 9, 0, tool)
    b, k2 = _run(agent, furn, 8, 9, 0x7FC00000, tool)
    bad = [(t - 1, np.nonzero((x[:8].view(np.uint32) != y.view(np.uint32)).any(axis=1))[0].tolist()) for t, (x, y) in enumerate(zip(a, b)) if (x[:8].view(np.uint32) != y.view(np.uint32)).any()]
    print(agent, furn, k1, "|", k2, "->", "first 8 envs identical in a batch of 40 and a batch of 8" if not bad else "DIFFER %s" % bad[:4])
