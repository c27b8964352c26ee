"""furniture_amd — MI355X-native batched FurnitureEnv.step() hot path.

Only the path named in BASELINE.json's north_star lives here:

* ``mjcf``      host-side model compiler (MJCF -> flat tables / blob)
* ``csrc``      HIP kernels + the C-ABI shared library (libfsim.so)
* ``sim``       ctypes binding to the C-ABI (no torch types cross it)
* ``envs``      host-side mirror of the reference's gym.Env surface
* ``transform_utils`` quaternion helpers the env logic needs

The product path never imports anything from ``oracle/``.
"""

__version__ = "0.1.0"
