"""Host side of config.reset_robot_after_attach (furniture_amd/envs.py, _attach_*): the env's RandomState must be consumed exactly as the
reference consumes it -- reset draws (placement + 101 joint-noise rows), ONE joint-noise row per attach (furniture.py:919-925), the reset an
unstable simulation triggers inside step(), the worker's reset of a finished episode -- while the device is handed speculative draws ahead
of every step.  The device is replaced by a recorder here (no GPU): a scripted sequence of step outcomes (attached / failed / done) is
played and every table the host uploads is compared with a plain per-env replay of the stream in reference order."""
import numpy as np

from furniture_amd.envs import FurnitureBatchEnv, ResetTableSampler, make_config
from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import INFO_CONNECTED_THIS_STEP, INFO_DIM, INFO_FAIL


class _Arr:
    def __init__(self, a):
        self.a = a

    def cpu(self):
        return self

    def numpy(self):
        return self.a


class _Torch:
    @staticmethod
    def as_tensor(x, device=None):
        return np.asarray(x)


class _Sim:
    """records what the host uploads / asks for"""
    torch, device = _Torch, "cpu"

    def __init__(self, n):
        self.n, self.tables, self.attach, self.resets = n, {}, {}, []

    def set_reset_tables(self, parts, noise, mask=None):
        for i in np.nonzero(np.ones(self.n, bool) if mask is None else mask)[0]:
            self.tables[i] = (parts[i].copy(), noise[i].copy())

    def set_attach_noise(self, noise, mask=None):
        for i in np.nonzero(np.ones(self.n, bool) if mask is None else mask)[0]:
            self.attach[i] = noise[i].copy()

    def reset(self, mask, obs):
        self.resets.append(np.asarray(mask).astype(bool).copy())

    def sync(self):
        pass


def _env(n, seed):
    m = load_compiled("Sawyer", "table_lack_0825")
    cfg = make_config(unity=False, record_vid=False, control_type="impedance", furniture_name="table_lack_0825", seed=seed, reset_robot_after_attach=True)
    env = FurnitureBatchEnv.__new__(FurnitureBatchEnv)  # (no device: only the stream bookkeeping is exercised)
    env.config, env.model, env.num_envs, env._attach_mode, env._auto_reset = cfg, m, n, True, True
    env._sampler = ResetTableSampler(m, cfg, seed, 0, n)
    env.sim = _Sim(n)
    env._obs = None
    env._split = lambda obs, sub=None: obs
    env._info = _Arr(np.zeros((n, INFO_DIM), dtype=np.int32))
    env._done = _Arr(np.zeros(n, dtype=np.uint8))
    return env, m, cfg


def test_attach_mode_consumes_the_stream_like_the_reference():
    n, seed = 3, 77
    env, m, cfg = _env(n, seed)
    narm, a = len(m.arm_qposadr), cfg.agent_xyz_rand
    # the reference order, env by env: a twin sampler's generators are advanced by hand
    twin = ResetTableSampler(m, cfg, seed, 0, n)
    ref = twin.rngs

    def ref_reset(i):
        mask = np.zeros(n, bool)
        mask[i] = True
        parts, noise = twin._draw_python(mask)
        return parts[i], noise[i]

    def ref_peek(i):  # what the device must hold ahead of the next step: the next reset table and the next attach row, both from the CURRENT position
        st = ref[i].get_state()
        tab = ref_reset(i)
        ref[i].set_state(st)
        att = ref[i].uniform(low=-a, high=a, size=narm).astype(np.float32)
        ref[i].set_state(st)
        return tab, att

    def check_pending():
        for i in range(n):
            tab, att = ref_peek(i)
            assert np.array_equal(env.sim.tables[i][0], tab[0]) and np.array_equal(env.sim.tables[i][1], tab[1]), i
            assert np.array_equal(env.sim.attach[i][:narm], att), i

    env.reset()
    for i in range(n):
        ref_reset(i)  # consumed by the reset
    assert len(env.sim.resets) == 1 and env.sim.resets[0].all()
    check_pending()
    # scripted outcomes: (attached, failed, done) per env
    script = [
        ([1, 0, 0], [0, 0, 0], [0, 0, 0]),  # env 0 attaches
        ([0, 0, 0], [0, 1, 0], [0, 0, 0]),  # env 1: unstable simulation, reset inside step()
        ([1, 0, 1], [0, 0, 0], [1, 0, 0]),  # env 0 attaches AND finishes (success); env 2 attaches
        ([0, 0, 0], [0, 1, 0], [0, 1, 1]),  # env 1 fails and is done (the worker resets it again); env 2 times out
        ([0, 0, 0], [0, 0, 0], [0, 0, 0]),
    ]
    for att, fail, done in script:
        env._info.a[:] = 0
        env._info.a[:, INFO_CONNECTED_THIS_STEP] = att
        env._info.a[:, INFO_FAIL] = fail
        env._done.a[:] = done
        nres = len(env.sim.resets)
        env._attach_after_step()
        for i in range(n):
            if att[i] and not fail[i]:
                ref[i].uniform(low=-a, high=a, size=narm)  # _connect -> _initialize_robot_pos
            if fail[i]:
                ref_reset(i)                               # the reset inside step()
            if done[i]:
                ref_reset(i)                               # the vec-env worker's reset
        if any(done):
            assert len(env.sim.resets) == nres + 1 and np.array_equal(env.sim.resets[-1], np.array(done, bool))
        else:
            assert len(env.sim.resets) == nres
        check_pending()
        for i in range(n):  # the committed generators are where the reference's are
            s1, s2 = env._sampler.rngs[i].get_state(), ref[i].get_state()
            assert s1[2] == s2[2] and np.array_equal(s1[1], s2[1]), i


def test_in_reset_attach_draws_sit_between_placement_and_robot_initialisation():
    """config.reset_robot_after_attach with config.preassembled on a furniture with a recipe (round 4): the reset's own _connect calls take
    one draw each (furniture.py:919-925 inside :1542-1557) between the placement's draws and the 101 of the robot initialisation.  The
    host sampler against the oracle env's recorded draws, two resets, one and two recipe steps: placement, the 101 noise rows, and the
    in-reset rows behind them (rows 101..); the generators end in the same state."""
    import numpy as np
    from furniture_amd.envs import ResetTableSampler, make_config
    from furniture_amd.mjcf.model import load_compiled
    from oracle.oracle_env import FurnitureEnvOracle, OracleConfig
    m = load_compiled("Sawyer", "table_lack_0825")
    for pre in ([0], [0, 1]):
        cfg = make_config(unity=False, record_vid=False, control_type="impedance", furniture_name="table_lack_0825", max_episode_steps=3, seed=53,
                          reset_robot_after_attach=True, preassembled=pre)
        s = ResetTableSampler(m, cfg, 53, 0, 1)
        assert s.n_attach_in_reset == len(pre)
        o = FurnitureEnvOracle(m, OracleConfig(max_episode_steps=3, seed=53, solver_tolerance=1e-10, reset_robot_after_attach=True, preassembled=pre))
        for rep in range(2):
            parts, noise = s.draw()
            o.reset()
            assert np.abs(parts[0] - o.reset_draws["part_qpos"].reshape(-1)).max() < 1e-6
            assert np.abs(noise[0][:707] - np.stack(o.reset_draws["noise"]).reshape(-1)).max() < 1e-9
            assert np.abs(noise[0][707:] - np.concatenate(o.attach_draws[-len(pre):])).max() < 1e-9
            a, b = s.rngs[0].get_state(), o._rng.get_state()
            assert a[2] == b[2] and np.array_equal(a[1], b[1])
