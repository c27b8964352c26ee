"""Minimal ``gym.spaces`` stand-ins (Box / Dict) used only when gym is not installed.

The reference builds its spaces with gym (furniture/env/furniture.py:215-310); gym is absent
from this image, so the drop-in classes fall back to these two shape/dtype carriers.
"""

from collections import OrderedDict

import numpy as np

try:  # pragma: no cover - exercised only where gym exists
    from gym.spaces import Box, Dict  # type: ignore
except Exception:  # noqa: BLE001

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.shape = tuple(shape) if shape is not None else np.shape(low)
            self.dtype = np.dtype(dtype)
            self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape)
            self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape)

        def sample(self, rng=None):
            rng = rng or np.random
            lo = np.where(np.isfinite(self.low), self.low, -1.0)
            hi = np.where(np.isfinite(self.high), self.high, 1.0)
            return rng.uniform(lo, hi).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self):
            return "Box%s" % (self.shape,)

    class Dict:
        def __init__(self, spaces):
            self.spaces = OrderedDict(spaces)

        def sample(self, rng=None):
            return OrderedDict((k, s.sample(rng)) for k, s in self.spaces.items())

        def __repr__(self):
            return "Dict(%s)" % ", ".join("%s: %r" % kv for kv in self.spaces.items())


def flatdim(space):
    if isinstance(space, Dict):
        return int(sum(flatdim(s) for s in space.spaces.values()))
    return int(np.prod(space.shape))
