"""Every furniture shipped compiled for the Sawyer agent (61 of the reference's 64: three carry mesh geoms with a density and
need mesh volumes) runs reset + random steps on the device, or is refused with a clear error at fsim_create -- models beyond
64 dofs (one solver lane per dof).  The parity tests cover the BASELINE configs' models; this one is breadth: the generic
kernels, the model compiler's tables and the host-side samplers on models nobody looked at individually."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_every_compiled_sawyer_furniture_resets_and_steps():
    import torch
    from furniture_amd.envs import make_vec_env
    from furniture_amd.mjcf.model import _COMPILED_DIR, load_compiled
    from furniture_amd.sim import FsimError

    names = sorted(os.path.basename(p)[len("Sawyer__"):-len("__vel.npz")] for p in glob.glob(os.path.join(_COMPILED_DIR, "Sawyer__*__vel.npz")))
    assert len(names) >= 60
    ran, refused, unplaceable = [], [], []
    for name in names:
        m = load_compiled("Sawyer", name)
        try:
            env = make_vec_env("Sawyer", 4, furniture_name=name, max_episode_steps=3, seed=11, record_vid=False, unity=False, control_type="impedance")
        except FsimError as e:
            assert m.nv > 64 and "64" in str(e), (name, m.nv, str(e))
            refused.append(name)
            continue
        assert m.nv <= 64, name
        try:
            ob = env.reset()
        except RuntimeError as e:
            # the reference's UniformRandomSampler raises RandomizationError for the same furniture and seeds (checked by running
            # it: cabinet_akurum_0021, table_hemnes_0539 never place with the default jitter) -- not a device matter
            assert "Cannot place all objects" in str(e), (name, str(e))
            unplaceable.append(name)
            env.close()
            continue
        assert ob["object_ob"].shape == (4, 7 * m.nparts)
        g = torch.Generator(device=env.sim.device)
        g.manual_seed(1)
        for t in range(4):  # crosses an in-kernel auto-reset (max_episode_steps = 3)
            a = torch.empty((4, 9), device=env.sim.device).uniform_(-1, 1, generator=g)
            ob, rew, done, info = env.step(a)
            assert bool(torch.isfinite(ob["object_ob"]).all()) and bool(torch.isfinite(ob["robot_ob"]).all()) and bool(torch.isfinite(rew).all()), (name, t)
            assert bool(done.all()) == (t == 2), (name, t)
        # the parts rest on the floor after the reset: no part centre below it, none flung away
        z = ob["object_ob"].reshape(4, m.nparts, 7)[:, :, 2]
        assert float(z.min()) > -0.01 and float(ob["object_ob"].reshape(4, m.nparts, 7)[:, :, :3].abs().max()) < 3.0, name
        env.close()
        ran.append(name)
    print("ran %d furniture models, refused %d (> 64 dofs): %s; placement sampler gives up (as the reference's does) on %s" % (len(ran), len(refused), refused, unplaceable))
    assert len(ran) >= 45 and len(unplaceable) <= 3
