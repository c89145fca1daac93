"""-m gpu: the outer Python boundary (make_mqe_env / ENV_DICT / task wrappers / OpenRL adapter) on the DEFAULT engine
-- the HIP one behind the C ABI -- i.e. exactly what `openrl_ws/train.py` would drive.  tests/test_env_api.py holds the same
checks on the oracle-backed engine (CPU)."""
import types

import numpy as np
import pytest
import torch

from mqe.engine import abi
from mqe.envs.go1.go1 import Go1
from mqe.envs.utils import ENV_DICT, make_mqe_env, custom_cfg
from mqe.utils.helpers import finish_args

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _restore_registry(monkeypatch):
    monkeypatch.setattr(Go1, "shard", None)
    saved = {k: v["config"].env.num_envs for k, v in ENV_DICT.items()}
    yield
    for k, v in ENV_DICT.items():
        v["config"].env.num_envs = saved[k]


def args_for(task, n, seed=0):
    return finish_args(types.SimpleNamespace(task=task, num_envs=n, seed=seed, headless=True, record_video=False,
                                             sim_device="cuda:0", pipeline="gpu", subscenes=0, num_threads=0))


@pytest.mark.parametrize("task,A,Aw,D", [("go1gate", 2, 2, 16), ("go1sheep-hard", 2, 2, 34), ("go1seesaw", 2, 2, 14),
                                         ("go1football-defender", 3, 2, 20), ("go1pushbox", 2, 2, 22)])
def test_make_mqe_env_surface_on_the_hip_engine(task, A, Aw, D):
    a = args_for(task, 8)
    env, cfg = make_mqe_env(task, a, custom_cfg(a))
    from mqe.engine.hip_engine import HipEngine
    assert isinstance(env.env.engine, HipEngine) and str(env.device).startswith("cuda")
    assert cfg is ENV_DICT[task]["config"] and cfg.env.num_envs == 8
    assert env.num_envs == 8 and env.env.num_agents == A and env.num_agents == Aw
    assert env.observation_space.shape == (D,) and env.action_space.shape == (3,)
    assert env.dt == pytest.approx(0.02) and env.max_episode_length == np.ceil(cfg.env.episode_length_s / 0.02)
    obs = env.reset()
    assert obs.is_cuda and obs.shape == (8, Aw, D) and obs.dtype == torch.float32
    assert torch.equal(obs[:, :, :Aw].cpu(), torch.eye(Aw).expand(8, Aw, Aw))
    kept = []
    for t in range(3):
        obs, rew, done, info = env.step(torch.rand(8, Aw, 3, device="cuda") * 2 - 1)
        kept.append((obs, rew, done, obs.clone(), rew.clone(), done.clone()))
    assert obs.shape == (8, Aw, D) and rew.shape == (8, Aw) and done.shape == (8,) and done.dtype == torch.bool
    # every step hands out fresh tensors (the reference builds new ones per step): nothing a caller kept was overwritten
    torch.cuda.synchronize()
    for o, r, d, oc, rc, dc in kept:
        assert torch.equal(o, oc) and torch.equal(r, rc) and torch.equal(d, dc)
    assert len({k[0].data_ptr() for k in kept}) == 3 and len({k[2].data_ptr() for k in kept}) == 3
    assert isinstance(info, dict) and "time_outs" in info
    assert env.reward_buffer["step count"] == 3
    for name in ("root_states_npc", "env_origins", "collide_buf", "reset_ids", "r_term_buff", "p_term_buff", "base_init_state",
                 "BarrierTrack_kwargs", "all_dof_states", "npc_indices", "env_agent_indices", "agent_origins", "obs_buf"):
        assert getattr(env, name) is not None
    assert env.obs_buf.base_pos.shape == (8 * A, 3) and env.obs_buf.base_rpy.shape == (8 * A, 3)
    assert env.root_states.shape == (8 * A, 13) and env.dof_pos.shape == (8, 12 * A)
    assert env.history_locomotion_obs.shape == (8 * A, 2100)
    assert env.reset_ids.dtype == torch.int64
    sums = env.env.engine.tensor(abi.T_REWARD_SUMS)
    assert sums.is_cuda and torch.isfinite(sums).all()
    env.close()


@pytest.mark.parametrize("task", ["go1gate", "go1football-defender"])
def test_unfused_step_equals_fused_on_the_hip_engine(task):
    """wrapper.step (one fused mqe_step) == the reference's call pattern: the wrapper clips and scales, Go1.step runs policy,
    4 x (torques, simulate, post_decimation_step) and the post-physics step through the Isaac-Gym-shaped entry points; for the
    defender task Go1.step also appends the scripted command (mqe_defender_command, go1_football_defender.py:25-31)."""
    N = 16
    a = args_for(task, N)
    e1, _ = make_mqe_env(task, a, custom_cfg(a))
    e2, _ = make_mqe_env(task, a, custom_cfg(a))
    e1.reset(); e2.reset()
    Aw = e1.num_agents
    g = torch.Generator().manual_seed(3)
    for t in range(4):
        act = (torch.rand(N, Aw, 3, generator=g) * 3 - 1.5).cuda()
        o1, r1, d1, _ = e1.step(act)
        cmd = (act.clip(-1, 1) * e2.action_scale).reshape(-1, 3)
        ob, rew, reset, extras = e2.env.step(cmd)
        torch.cuda.synchronize()
        assert torch.allclose(e1.env.root_states, e2.env.root_states, atol=1e-6)
        assert torch.allclose(e1.env.dof_pos, e2.env.dof_pos, atol=1e-6)
        assert torch.equal(d1, reset)
        assert torch.allclose(ob.base_pos, e1.env.obs_buf.base_pos, atol=1e-6)
        assert (rew == 0).all()
        # the per-substep logs of post_decimation_step (legged_robot.py:112-115): fused == unfused
        for kind in (abi.T_SUBSTEP_TORQUES, abi.T_SUBSTEP_DOF_VEL, abi.T_SUBSTEP_EXCEED_DOF_POS_LIMITS):
            x1, x2 = e1.env.engine.tensor(kind), e2.env.engine.tensor(kind)
            assert torch.allclose(x1.float(), x2.float(), atol=2e-5), kind
    e1.close(); e2.close()


def test_openrl_adapter_on_the_hip_engine():
    from openrl_ws.utils import make_env
    a = args_for("go1gate", 8)
    env, cfg = make_env(a, custom_cfg(a))
    assert env.agent_num == 2 and env.parallel_env_num == 8
    o = env.reset()
    assert isinstance(o, np.ndarray) and o.shape == (8, 2, 16)
    o, r, d, infos = env.step(np.random.RandomState(0).uniform(-2, 2, (8, 2, 3)))
    assert o.shape == (8, 2, 16) and r.shape == (8, 2, 1) and d.shape == (8, 2) and d.dtype == bool and len(infos) == 8
    br = env.batch_rewards(None)
    assert "average step reward" in br and "target reward" in br
    assert float(env.env.reward_buffer["target reward"]) == 0.0 and env.env.reward_buffer["step count"] == 0   # cleared by the read
    env.close()


def test_device_resident_hand_off():
    """`step_torch` (tensors stay on the GPU, fresh per step) and `step_dlpack` (any `__dlpack__` producer in, DLPack
    exporters out) against the numpy adapter of the reference (openrl_ws/utils.py:53-67) on a twin env."""
    from openrl_ws.utils import make_env
    N = 8
    a = args_for("go1gate", N)
    e_np, _ = make_env(a, custom_cfg(a))
    e_t, _ = make_env(a, custom_cfg(a))
    e_d, _ = make_env(a, custom_cfg(a))
    for e in (e_np, e_t, e_d):
        e.reset()
    rs = np.random.RandomState(5)
    held = []
    for t in range(5):
        act = rs.uniform(-2, 2, (N, 2, 3)).astype(np.float32)
        o, r, d, _ = e_np.step(act)
        ot, rt, dt_ = e_t.step_torch(torch.from_numpy(act).cuda())
        assert ot.is_cuda and rt.is_cuda and dt_.is_cuda and rt.shape == (N, 2, 1) and dt_.shape == (N, 2) and dt_.dtype == torch.bool
        assert np.array_equal(ot.cpu().numpy(), o) and np.array_equal(rt.cpu().numpy(), r) and np.array_equal(dt_.cpu().numpy(), d)
        held.append((ot, ot.clone()))

        class Producer:                     # a foreign array: only the DLPack protocol is visible
            def __init__(self, t): self._t = t
            def __dlpack__(self, stream=None): return self._t.__dlpack__(stream=stream) if stream is not None else self._t.__dlpack__()
            def __dlpack_device__(self): return self._t.__dlpack_device__()
        od, rd, dd = e_d.step_dlpack(Producer(torch.from_numpy(act).cuda()))
        for x in (od, rd, dd):
            assert hasattr(x, "__dlpack__") and x.__dlpack_device__()[0] in (2, 10)          # kDLCUDA / kDLROCM
        o2 = torch.from_dlpack(od)
        assert o2.data_ptr() == od.data_ptr()                                                   # zero-copy
        assert np.array_equal(o2.cpu().numpy(), o) and np.array_equal(torch.from_dlpack(rd).cpu().numpy(), r)
        assert np.array_equal(torch.from_dlpack(dd).cpu().numpy().astype(bool), d)
    torch.cuda.synchronize()
    for kept, copy in held:
        assert torch.equal(kept, copy)
    for e in (e_np, e_t, e_d):
        e.close()


def test_seed_reaches_the_engine():
    """make_env hands --seed to the engine (every in-engine draw is keyed by it): same seed -> identical reset states, another
    seed -> other joint ratios / base velocities"""
    def reset_state(seed):
        a = args_for("go1gate", 16, seed=seed)
        env, _ = make_mqe_env("go1gate", a, custom_cfg(a))
        assert env.env.engine.desc.seed == seed
        env.reset()
        st = (env.env.dof_pos.clone(), env.env.root_states.clone())
        env.close()
        return st
    q0, r0 = reset_state(0)
    q0b, r0b = reset_state(0)
    q7, r7 = reset_state(7)
    assert torch.equal(q0, q0b) and torch.equal(r0, r0b)
    assert not torch.equal(q0, q7) and not torch.equal(r0[:, 7:], r7[:, 7:])


def _sheep_draws(make_engine, seed, steps=3, N=64):
    """N(0,1) draws of the sheep script recovered through the unfused entry point: (increment in hash mode - increment with a
    scripted zero noise) / (2 * randomness), from identical states (velocities zeroed before every call; no physics in between,
    so the deterministic part -- cohesion + repulsion, functions of positions -- is the same every time)"""
    from helpers import make_desc
    d0, k0, _ = make_desc("go1sheep-hard", N, noise_mode=1)
    d1, k1, _ = make_desc("go1sheep-hard", N)
    d1.seed = d0.seed = seed
    assert d1.sheep_movement_randomness == pytest.approx(0.1) and d1.noise_mode == 0 and d0.noise_mode == 1
    es, eh = make_engine(d0, k0), make_engine(d1, k1)
    es.reset_all(); eh.reset_all()
    zs = []
    for t in range(steps):
        for e in (es, eh):
            e.tensor(abi.T_ROOT_STATE)[:, 2:, 7:10] = 0
            e.post_physics_step()
        if eh.tensor(abi.T_ROOT_STATE).is_cuda:
            torch.cuda.synchronize()
        assert not bool(eh.tensor(abi.T_RESET_BUF).any()) and not bool(es.tensor(abi.T_RESET_BUF).any())
        zs.append(((eh.tensor(abi.T_ROOT_STATE)[:, 2:, 7:9] - es.tensor(abi.T_ROOT_STATE)[:, 2:, 7:9]) / 0.2).cpu().clone())
    es.close(); eh.close()
    return torch.stack(zs)


def test_sheep_random_walk_is_drawn_in_the_engine():
    """go1sheep-hard: `sheep_movement_randomness * randn * 2` is drawn fresh every step (go1_sheep.py:43) -- in MQE_NOISE_HASH mode
    by the engine itself, keyed by (seed, global env id, step): standard normal, different every step and for every seed, the
    same again for the same seed, and the same numbers as the CPU oracle draws"""
    from helpers import hip_engine, oracle_engine
    a, b, c = _sheep_draws(hip_engine, 0), _sheep_draws(hip_engine, 0), _sheep_draws(hip_engine, 1)
    assert torch.equal(a, b)
    assert (a[0] - a[1]).abs().mean() > 0.5 and (a[0] - c[0]).abs().mean() > 0.5
    assert abs(float(a.mean())) < 0.1 and 0.9 < float(a.std()) < 1.1
    o = _sheep_draws(oracle_engine, 0)
    assert (a - o).abs().max() < 1e-4


def test_perlin_track_through_the_plugin_api():
    """SURVEY 8(f)4 end to end: a user switches the Perlin relief on in `custom_cfg` (BarrierTrack_kwargs of the reference's config
    tree), make_mqe_env builds the terrain with the reference's generator stream and the HIP engine walks on it"""
    from mqe.envs.configs.go1_gate_config import Go1GateCfg
    a = args_for("go1gate", 64)
    old_kw, had = dict(Go1GateCfg.terrain.BarrierTrack_kwargs), hasattr(Go1GateCfg.terrain, "TerrainPerlin_kwargs")
    old_tp = getattr(Go1GateCfg.terrain, "TerrainPerlin_kwargs", None)

    def with_perlin(cfg):
        cfg = custom_cfg(a)(cfg)
        cfg.terrain.BarrierTrack_kwargs = dict(cfg.terrain.BarrierTrack_kwargs, add_perlin_noise=True, border_perlin_noise=True)
        cfg.terrain.TerrainPerlin_kwargs = dict(zScale=0.05, frequency=10)
        return cfg
    try:
        env, cfg = make_mqe_env("go1gate", a, with_perlin)
        t = env.env.terrain
        assert t.ground_height is not None and t.ground_z == 0.0 and bool(env.env.engine.desc.ground_height)
        env.reset()
        for _ in range(40):
            obs, rew, done, info = env.step(torch.zeros(64, 2, 3, device="cuda"))
        torch.cuda.synchronize()
        z = env.env.root_states[:, 2]
        assert torch.isfinite(env.env.root_states).all() and float(z.min()) > 0.15 and float(z.max()) < 0.6
        feet = env.env.contact_forces.reshape(64, 2, 17, 3)[:, :, [4, 8, 12, 16], 2].sum(-1)
        assert float((feet > 30).float().mean()) > 0.8          # standing on the bumps
        env.close()
    finally:
        Go1GateCfg.terrain.BarrierTrack_kwargs = old_kw
        if had:
            Go1GateCfg.terrain.TerrainPerlin_kwargs = old_tp
        else:
            try:
                del Go1GateCfg.terrain.TerrainPerlin_kwargs
            except AttributeError:
                pass


def test_terrain_classes_and_subclass_overrides_on_the_hip_engine():
    """round-3 additions on the product engine: (1) an env on `TerrainPerlin` and on the legacy `Terrain` (registry, relief as the
    ground) keeps its robots on the surface; (2) a Go1 subclass with its own `_compute_torques` (torch PD law) is stepped with it and
    reproduces the engine's control type "P"."""
    from helpers import task_cfg
    base_t = task_cfg("go1gate").terrain
    variants = {
        "TerrainPerlin": type("PerlinClassTerrain", (base_t,), dict(selected="TerrainPerlin", num_rows=2, num_cols=2, terrain_length=4.0, terrain_width=4.0,
                                                                     TerrainPerlin_kwargs=dict(zScale=0.1, frequency=5))),
        "Terrain": type("LegacyTerrain", (base_t,), dict(selected="Terrain", mesh_type="trimesh", num_rows=2, num_cols=2, terrain_length=8.0, terrain_width=8.0,
                                                         border_size=2.0, horizontal_scale=0.1, vertical_scale=0.005, curriculum=False,
                                                         terrain_proportions=[1.0, 0.0, 0.0, 0.0, 0.0], slope_treshold=0.75)),
    }
    base_plane = ENV_DICT["go1plane"]["config"]
    for name, tcfg in variants.items():
        a = args_for("go1plane", 6)
        np.random.seed(0)
        try:
            env, _ = make_mqe_env("go1plane", a, lambda c: type("Go1PlaneOn" + name, (custom_cfg(a)(c),), {"terrain": tcfg}))
        finally:
            ENV_DICT["go1plane"]["config"] = base_plane
        assert type(env.env.terrain).__name__ == name
        env.reset()
        A = env.env.num_agents
        for _ in range(60):
            env.step(torch.zeros(6, A, 3, device="cuda"))
        torch.cuda.synchronize()
        rs = env.env.root_states.cpu()
        gh, hsc = env.env.terrain.ground_height, tcfg.horizontal_scale
        under = np.array([gh[int(round(float(x) / hsc)), int(round(float(y) / hsc))] for x, y in rs[:, :2]])
        hgt = rs[:, 2].numpy() - under
        # on the relief, not through it and not hovering (a pyramid's flank under a tilted robot reads up to a few dm at the base's xy)
        assert np.isfinite(rs.numpy()).all() and (hgt > 0.05).all() and (hgt < 0.8).all(), (name, hgt)
        env.close()

    from mqe.envs.configs.go1_gate_config import Go1GateCfg

    class MyGo1(Go1):
        def _compute_torques(self, actions):
            tau = 20.0 * (actions * self.cfg.control.action_scale + self.default_dof_pos - self.dof_pos) - 0.5 * self.dof_vel
            return torch.clip(tau, -self.torque_limits, self.torque_limits)
    old, saved_cls = Go1GateCfg.control.control_type, ENV_DICT["go1gate"]["class"]
    Go1GateCfg.control.control_type = "P"
    try:
        a = args_for("go1gate", 8)
        ref_env, _ = make_mqe_env("go1gate", a, custom_cfg(a))
        ENV_DICT["go1gate"]["class"] = MyGo1
        my_env, _ = make_mqe_env("go1gate", a, custom_cfg(a))
        ref_env.reset(); my_env.reset()
        g = torch.Generator().manual_seed(4)
        for t in range(5):
            act = ((torch.rand(16, 12, generator=g) - 0.5) * 2.0).cuda()
            ref_env.env.step(act); my_env.env.step(act)
            torch.cuda.synchronize()
            assert torch.allclose(my_env.env.dof_pos, ref_env.env.dof_pos, atol=5e-5) and torch.allclose(my_env.env.root_states, ref_env.env.root_states, atol=5e-5)
        with pytest.raises(NotImplementedError, match="overrides"):
            my_env.step(torch.zeros(8, 2, 3, device="cuda"))
        ref_env.close(); my_env.close()
    finally:
        Go1GateCfg.control.control_type = old
        ENV_DICT["go1gate"]["class"] = saved_cls


@pytest.mark.parametrize("task,N", [("go1gate", 192), ("go1sheep-hard", 12), ("go1seesaw", 16)])
def test_checkpoint_and_resume_is_bit_exact(task, N):
    """get_state() / set_state() (mqe_state_save / _load; the reference has no save / restore of the simulation): 40 steps --
    the history ring has turned over, episodes of 25 steps have timed out, frames after resets sit in the compact layer-0 operand --
    then a checkpoint; 15 more steps; back to the checkpoint, in the same handle and in a FRESH one made from the same configuration:
    the continuation is bit for bit the first one (returned batch, root and joint states, every step).  go1gate at 192 envs runs
    layer 0 on the split-f16 path only with MQE_GEMM_SPLIT=1; it is forced so that the derived operand is part of what is restored."""
    import os
    os.environ["MQE_GEMM_SPLIT"] = "1"
    try:
        a = args_for(task, N)

        def short_episodes(cfg):
            cfg = custom_cfg(a)(cfg)
            return type(cfg.__name__ + "Short", (cfg,), {"env": type("env", (cfg.env,), {"episode_length_s": 0.5})})
        base = ENV_DICT[task]["config"]
        env, _ = make_mqe_env(task, a, short_episodes)
        env2, _ = make_mqe_env(task, a, short_episodes)
        ENV_DICT[task]["config"] = base
    finally:
        del os.environ["MQE_GEMM_SPLIT"]
    env.reset(); env2.reset()
    g = torch.Generator().manual_seed(41)
    A = env.num_agents
    acts = [(torch.rand(N, A, 3, generator=g) * 2 - 1).cuda() for _ in range(55)]
    n_done = 0
    for t in range(40):
        _, _, done, _ = env.step(acts[t])
        n_done += int(done.sum())
    assert n_done >= N                                # every env has been through a reset
    ck = env.env.get_state()

    def run(e):
        out = []
        for t in range(40, 55):
            obs, rew, done, _ = e.step(acts[t])
            out.append((obs.clone(), rew.clone(), done.clone(), e.env.root_states.clone(), e.env.dof_state.clone()))
        return out
    first = run(env)
    env.env.set_state(ck)
    again = run(env)
    env2.env.set_state(ck)
    fresh = run(env2)
    for t, (x, y, z) in enumerate(zip(first, again, fresh)):
        for k in range(5):
            assert torch.equal(x[k], y[k]), (task, "same handle", t, k)
            assert torch.equal(x[k], z[k]), (task, "fresh handle", t, k)
    env.close(); env2.close()


def test_go1_level_step_is_the_fused_step_command():
    """Go1.step() for control type C runs as mqe_step_command (five launches) unless a subclass overrides a piece of the loop; the
    stage-by-stage entry points (what the trace replays drive) give the same state bit for bit."""
    from helpers import make_desc, hip_engine
    d1, k1, _ = make_desc("go1gate", 48)
    d2, k2, _ = make_desc("go1gate", 48)
    ea, eb = hip_engine(d1, k1), hip_engine(d2, k2)
    ea.reset_all(); eb.reset_all()
    g = torch.Generator().manual_seed(3)
    for t in range(12):
        cmd = ((torch.rand(96, 3, generator=g) * 2 - 1) * torch.tensor([2.0, 0.5, 0.5])).cuda().contiguous()
        ea.step_command(cmd)
        eb.policy_step(cmd)
        for k in range(d2.decimation):
            eb.compute_torques(); eb.simulate(); eb.post_decimation_step(k)
        eb.post_physics_step()
        torch.cuda.synchronize()
        for kind in (abi.T_ROOT_STATE, abi.T_DOF_STATE, abi.T_OBS_BAG, abi.T_LAST_LOCO_ACTION, abi.T_RESET_BUF, abi.T_GAIT_INDICES):
            assert torch.equal(ea.tensor(kind), eb.tensor(kind)), (t, kind)
    ea.close(); eb.close()


def test_terrain_curriculum_moves_envs_on_the_hip_engine():
    """the run-time terrain curriculum (legged_robot.py:479-503; pinned to upstream's reset_idx by fullstep_pushbox_curriculum) through
    make_mqe_env on the HIP engine, fused wrapper steps: envs whose row of the agents' root states lies a track length away from their
    origin move one level up at the reset, their LIVE origin follows (the box respawns on the new track, the push-box wrapper subtracts
    it), the robots and the copies made at construction keep the first track; time-outs move them again."""
    a = args_for("go1pushbox", 4)
    base = ENV_DICT["go1pushbox"]["config"]

    def edit(cfg):
        cfg = custom_cfg(a)(cfg)
        ter = type("terrain", (cfg.terrain,), {"curriculum": True, "num_rows": 3, "num_cols": 2, "max_init_terrain_level": 0})
        return type("Go1PushboxCurCfg", (cfg,), {"terrain": ter, "env": type("env", (cfg.env,), {"episode_length_s": 0.1})})     # time-outs every 5 steps
    try:
        env, _ = make_mqe_env("go1pushbox", a, edit)
        g = env.env
        assert g.engine.desc.terrain_curriculum == 1 and g.terrain_levels.tolist() == [0, 0, 0, 0]
        g._root3[1, :2, 0] += float(g.terrain.env_length)           # rows 2 and 3 = the robots of env 1
        torch.cuda.synchronize()
        obs = env.reset()
        torch.cuda.synchronize()
        # env 0 measures its own robot 0 (at its origin: stays); env 1 measures robot 1 of env 0 -- the neighbouring track, 5 m > env_length / 2
        # away -- and envs 2, 3 the shifted robots of env 1: all three move up
        assert g.terrain_levels.tolist() == [0, 1, 1, 1]
        assert torch.equal(g.env_origins, g.terrain_origins[g.terrain_levels.long(), g.terrain_types.long()])
        box = g.root_states_npc.view(4, 13)
        init = torch.tensor(np.ctypeslib.as_array(g.engine.desc.npc_init_state, shape=(1, 13)).copy(), device="cuda")
        assert torch.allclose(box[:, :3] - g.env_origins, init[:, :3].expand(4, 3), atol=0.6)               # the box sits on each env's LIVE track (+ init_npc_base_pos_range)
        assert torch.allclose(obs[:, 0, -6:-4], (box[:, :2] - g.env_origins[:, :2]), atol=1e-5)             # ... and the wrapper's box position is relative to it
        assert torch.allclose(g.root_states.view(4, 2, 13)[:, :, :2] - g.agent_origins[:, :, :2], g.base_init_state.view(4, 2, 13)[:, :, :2], atol=0.6)   # robots: still their first track
        lv = g.terrain_levels.clone()
        for t in range(12):            # two waves of time-outs: rows far from their env's origin keep moving up, past the last level a level is drawn
            env.step(torch.zeros(4, 2, 3, device="cuda"))
        torch.cuda.synchronize()
        assert (g.terrain_levels >= 0).all() and (g.terrain_levels < 3).all() and not torch.equal(g.terrain_levels, lv)
        assert torch.equal(g.env_origins, g.terrain_origins[g.terrain_levels.long(), g.terrain_types.long()])
        env.close()
    finally:
        ENV_DICT["go1pushbox"]["config"] = base


def test_termination_npc_reset_and_observation_overrides_on_the_hip_engine():
    """check_termination / _step_npc / reset_idx / compute_observations of a Go1 subclass, on the HIP engine (mqe_post_physics_stage)"""
    from test_env_api import _plugin_points_check
    _plugin_points_check(args_for, "cuda")
