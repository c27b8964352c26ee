"""furniture_amd — MI355X-native batched FurnitureEnv.step() hot path.

Only the path named in BASELINE.json's north_star lives here:

* ``mjcf``      host-side model compiler (MJCF -> flat tables / blob)
* ``csrc``      HIP kernels + the C-ABI shared library (libfsim.so)
* ``sim``       ctypes binding to the C-ABI (no torch types cross it)
* ``envs``      host-side mirror of the reference's gym.Env surface (batched env, single-env classes, gym ids)
* ``vec_env``   SubprocVecEnv-shaped wrapper (numpy in / out, per-env infos)
* ``mixed``     mixed-furniture batch (BASELINE config 5)
* ``dense``     tables of the dense 8-phase reward
* ``dist``      env sharding + the per-step RCCL observation all-gather
* ``scripted``  scripted pick-and-attach policy under ik_quaternion (scenario generator, cf. furniture_sawyer_gen.py)
* ``transform_utils`` quaternion helpers the env logic needs

The product path never imports anything from ``oracle/``.
"""

__version__ = "0.1.0"
