#!/usr/bin/env python3
"""Generate tests/golden/*.npz|json by IMPORTING the Python reference in this container.

Runs only where /root/reference exists (the build container).  The reference never travels: what is
committed are inputs + expected outputs (data), plus this script.  Physics (Isaac Gym / PhysX) is absent,
so every vector here pins the code AROUND the physics: MLPs, torque pipeline, command observation,
history, gait clock, termination, reset bookkeeping, observation bag, task wrappers, NPC scripts,
BarrierTrack terrain, config tree, OpenRL adapter.  Where the reference needs the simulator we drive it
with a scripted stand-in `FakeGym` whose `simulate()` overwrites the state tensors from arrays that are
stored in the fixture, so the product can replay exactly the same states.

usage:  python tools/gen_golden.py [--only name,...] [--out DIR]

One invocation regenerates everything.  Every stage runs in a forked child process: the reference's env constructors
assign tensors onto its config CLASSES (class attributes, shared by every later user of the class), so a stage that ran
after another one used to see a config tree that `class_to_dict` could not walk any more (RecursionError); a child
starts from the state right after import and leaves nothing behind.  `--out DIR` writes fixtures and product assets
under DIR/golden and DIR/assets instead of the committed places (tests/test_golden_regeneration.py compares them).
"""
import argparse
import importlib.util
import json
import os
import sys
import traceback
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")
PKG = os.path.join(ROOT, "multiagent-quadruped-environment_amd")
ASSETS = os.path.join(PKG, "assets")

sys.path.insert(0, os.path.join(HERE, "refstub"))
sys.path.insert(0, REF)
os.chdir(REF)  # the reference uses "./resources/..." relative paths (go1_config.py:122-123)

torch.set_grad_enabled(False)
torch.set_num_threads(1)


def _load_pkg_module(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(PKG, rel))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


pw = _load_pkg_module("_pw", "mqe/utils/policy_weights.py")

# the reference calls .cuda()/device="cuda" in wrappers; run everything on CPU instead
torch.Tensor.cuda = lambda self, *a, **k: self
_orig_tensor = torch.tensor


def _dev_cpu(fn):
    def w(*a, **k):
        if k.get("device", None) is not None and "cuda" in str(k["device"]):
            k["device"] = "cpu"
        return fn(*a, **k)
    return w


for _n in ("tensor", "zeros", "ones", "eye", "arange", "rand", "randn", "zeros_like", "randn_like", "empty"):
    setattr(torch, _n, _dev_cpu(getattr(torch, _n)))

from mqe.envs.go1.go1 import Go1  # noqa: E402
from mqe.envs.base.legged_robot import LeggedRobot  # noqa: E402
from mqe.envs.field.legged_robot_field import LeggedRobotField  # noqa: E402
from mqe.envs.npc.go1_sheep import Go1Sheep  # noqa: E402
from mqe.envs.npc.go1_football_defender import Go1FootballDefender  # noqa: E402
from mqe.envs.npc.go1_object import Go1Object  # noqa: E402
from mqe.envs import utils as ref_utils  # noqa: E402
from mqe.utils.helpers import class_to_dict  # noqa: E402
from mqe.utils.terrain.barrier_track import BarrierTrack  # noqa: E402
import mqe.envs.base.legged_robot as ref_lr  # noqa: E402

DEV = "cpu"


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print("wrote", name, {k: v.shape for k, v in out.items()})


# --------------------------------------------------------------------------------------------
# 1/2. MLP weights (product assets) + known-answer vectors
# --------------------------------------------------------------------------------------------
def export_mlp(jit_path, out_path):
    m = torch.jit.load(jit_path, map_location="cpu")
    sd = m.state_dict()
    keys = sorted({int(k.split(".")[0]) for k in sd})
    d = {}
    for i, k in enumerate(keys):
        d[f"W{i}"] = sd[f"{k}.weight"].numpy().astype(np.float32)
        d[f"b{i}"] = sd[f"{k}.bias"].numpy().astype(np.float32)
    np.savez(out_path, **d)
    return m


def gen_mlps():
    act = export_mlp(os.path.join(REF, "resources/actuator_nets/unitree_go1.pt"),
                     os.path.join(ASSETS, "actuator_net_unitree_go1.npz"))
    ada = export_mlp(os.path.join(REF, "mqe/utils/locomotion_checkpoints/walk_these_ways/adaptation_module_latest.jit"),
                     os.path.join(ASSETS, "adaptation_module.npz"))
    g = torch.Generator().manual_seed(0)
    x = torch.randn(256, 6, generator=g)
    x[:, :3] *= 0.4   # joint position errors [rad]
    x[:, 3:] *= 8.0   # joint velocities [rad/s]
    y = act(x)
    torch.manual_seed(0)
    spot_x = torch.randn(4, 6)
    save("actuator_net", x=x, y=y.flatten(), spot_x=spot_x, spot_y=act(spot_x).flatten())
    h = torch.randn(16, 2100, generator=g) * 0.5
    h[0] = 0
    save("adaptation_module", x=h, y=ada(h))
    return act, ada


# --------------------------------------------------------------------------------------------
# scripted simulator stand-in
# --------------------------------------------------------------------------------------------
class FakeGym:
    """simulate() copies the next scripted dof state into the tensor the env holds views of."""

    def __init__(self):
        self.dof_script = None   # [K, ...] successive dof states
        self.k = 0
        self.forces = []

    def set_dof_actuation_force_tensor(self, sim, t):
        self.forces.append(t.clone())

    def simulate(self, sim):
        self.env.all_dof_states.copy_(self.dof_script[self.k].reshape(self.env.all_dof_states.shape))
        self.k += 1

    def fetch_results(self, *a):
        pass

    def refresh_dof_state_tensor(self, sim):
        pass

    def refresh_actor_root_state_tensor(self, sim):
        if self.root_script is not None:
            self.env.all_root_states.copy_(self.root_script[self.step_idx])

    def refresh_net_contact_force_tensor(self, sim):
        if self.contact_script is not None:
            self.env.contact_forces.copy_(self.contact_script[self.step_idx].reshape(self.env.contact_forces.shape))

    def refresh_rigid_body_state_tensor(self, sim):
        pass

    def set_dof_state_tensor_indexed(self, *a):
        self.calls.append(("dof_indexed", a[2].clone()))

    def set_actor_root_state_tensor_indexed(self, *a):
        self.calls.append(("root_indexed", a[2].clone()))

    calls = []
    root_script = None
    contact_script = None
    step_idx = 0


def body_policy(Ws, bs):
    Wt = [torch.from_numpy(w) for w in Ws]
    bt = [torch.from_numpy(b) for b in bs]

    def f(x):
        for i, (w, b) in enumerate(zip(Wt, bt)):
            x = torch.nn.functional.linear(x, w, b)
            if i < len(Wt) - 1:
                x = torch.nn.functional.elu(x)
        return x
    return f


def make_env(cls, cfg, N, seed=0, P_dofs=0):
    """Build a reference env object WITHOUT a simulator: run the reference's own buffer/init code on an
    instance created with object.__new__, feeding it the handful of values Isaac Gym would have supplied."""
    A = cfg.env.num_agents
    P = getattr(cfg.env, "num_npcs", 0)
    env = object.__new__(cls)
    env.cfg = cfg
    env.sim_params = types.SimpleNamespace(dt=cfg.sim.dt)
    env.sim = None
    env.gym = FakeGym()
    env.gym.env = env
    env.gym.calls = []
    env.device = DEV
    env.sim_device = DEV
    env.headless = True
    env.num_envs, env.num_agents, env.num_npcs = N, A, P
    env.num_obs = cfg.env.num_observations
    env.num_privileged_obs = None
    env.num_action = cfg.env.num_actions
    env.num_actions = A * cfg.env.num_actions
    env.num_actions_npc = getattr(cfg.env, "num_actions_npc", 0) * P
    env.decimation = cfg.control.decimation
    env.num_dof = 12
    env.num_actuated_dof = 12 * A
    env.num_bodies = 17
    env.debug_viz = False
    env.record_now = False
    env.init_done = False
    env.viewer = None
    env.up_axis_idx = 2
    env.custom_origins = True
    env.envs = [None] * N
    env.sensor_handles = [[{} for _ in range(A)] for _ in range(N)]
    LeggedRobot._parse_cfg(env, cfg)
    # leg order fixed by the build: FL, FR, RL, RR
    env.dof_names = [f"{leg}_{j}_joint" for leg in ("FL", "FR", "RL", "RR") for j in ("hip", "thigh", "calf")]
    env.obs_buf = torch.zeros(N, env.num_obs)
    env.rew_buf = torch.zeros(N * A)
    env.reset_buf = torch.ones(N, dtype=torch.long)
    env.episode_length_buf = torch.zeros(N, dtype=torch.long)
    env.time_out_buf = torch.zeros(N, dtype=torch.bool)
    env.collide_buf = torch.zeros(N, dtype=torch.bool)
    env.privileged_obs_buf = None
    env.extras = {}
    # state tensors the simulator would own
    env._all_root = torch.zeros(N * (A + P), 13)
    env._all_root[:, 6] = 1.0
    env._all_dof = torch.zeros(N * (12 * A + env.num_actions_npc), 2)
    env._contact = torch.zeros(N * (17 * A + P_bodies(cfg)), 3)
    env._rigid = torch.zeros(N * (17 * A + P_bodies(cfg)), 13)
    g = env.gym
    g.acquire_actor_root_state_tensor = lambda sim: env._all_root
    g.acquire_dof_state_tensor = lambda sim: env._all_dof
    g.acquire_net_contact_force_tensor = lambda sim: env._contact
    g.acquire_rigid_body_state_tensor = lambda sim: env._rigid
    g.render_all_camera_sensors = lambda sim: None
    # indices the env creation loop would have produced (legged_robot.py:845-902)
    env.env_agent_indices = torch.arange(N * A).reshape(N, A)
    env.env_npc_indices = torch.arange(N * P).reshape(N, P)
    env.actor_indices = torch.arange(N * (A + P), dtype=torch.int32).reshape(N, A + P)
    env.agent_indices = env.actor_indices[:, :A].clone()
    env.npc_indices = env.actor_indices[:, A:].clone()
    env.feet_indices = torch.tensor([4, 8, 12, 16])
    env.termination_contact_indices = torch.tensor([0] if len(cfg.asset.terminate_after_contacts_on) else [], dtype=torch.long)
    env.penalised_contact_indices = torch.tensor([0, 2, 6, 10, 14])
    tl = cfg.control.torque_limits
    env.torque_limits = torch.tensor(tl * (env.num_actuated_dof // len(tl)), dtype=torch.float)
    lim = torch.tensor([[-0.802851455917, 0.802851455917], [-1.0471975512, 4.18879020479], [-2.69653369433, -0.916297857297]] * (4 * A))
    m = lim.mean(1)
    r = lim[:, 1] - lim[:, 0]
    env.dof_pos_limits = torch.stack([m - 0.5 * r * cfg.rewards.soft_dof_pos_limit, m + 0.5 * r * cfg.rewards.soft_dof_pos_limit], 1)
    env.default_friction, env.default_restitution = 1.0, 0.0
    return env


def P_bodies(cfg):
    name = getattr(cfg.asset, "name_npc", "")
    per = {"": 0, "ball": 1, "sheep": 1, "seesaw": 2, "box": 1, "rotation": 2, "bridge": 3, "wrestling": 9, "circular": 2}[name]
    return per * getattr(cfg.env, "num_npcs", 0)


def finish_env(env, terrain_origins, agent_origins, env_info, Ws, bs, ada):
    """Origins (from BarrierTrack) + buffers + policy, all through the reference's own methods."""
    cfg = env.cfg
    N, A = env.num_envs, env.num_agents
    env.env_origins = terrain_origins.clone()
    env.env_origins_repeat = env.env_origins.unsqueeze(1).repeat(1, A, 1).reshape(-1, 3)
    env.agent_origins = agent_origins.clone()
    if env_info is not None:
        env.env_info = env_info
    # base init states (legged_robot.py:815-831)
    lst = []
    for st in cfg.init_state.init_states:
        lst.append(torch.tensor(st.pos + st.rot + st.lin_vel + st.ang_vel, dtype=torch.float))
    env.base_init_state = torch.stack(lst, 0).repeat(N, 1)
    Go1._init_custom_buffers__(env)
    if isinstance(env, (Go1Object, Go1FootballDefender)) and env.num_npcs:
        # _prepare_npc minus asset loading (go1_object.py:27-51)
        if hasattr(cfg.init_state, "default_npc_joint_angles"):
            env.default_dof_pos_npc = torch.tensor(cfg.init_state.default_npc_joint_angles, dtype=torch.float).reshape(1, -1)
        else:
            env.default_dof_pos_npc = torch.zeros(env.num_actions_npc).unsqueeze(0)
        l2 = [torch.tensor(s.pos + s.rot + s.lin_vel + s.ang_vel, dtype=torch.float) for s in cfg.init_state.init_states_npc]
        env.base_init_state_npc = torch.stack(l2, 0).repeat(N, 1)
    # run the reference's buffer init (legged_robot.py:549-645, field:185-223, go1.py:357-387)
    Go1._init_buffers(env)
    Go1._prepare_reward_function(env)
    env.init_done = True
    from copy import copy
    env.obs_buf = copy(cfg.obs)
    env.privileged_obs_buf = copy(cfg.privileged_obs)
    env.last_locomotion_action = torch.zeros(N * A, 12)
    env.last_two_locomotion_action = torch.zeros(N * A, 12)
    # _prepare_locomotion_policy with the body injected (go1.py:389-409; body_latest.jit is missing)
    env.locomotion_obs = Go1._fill_command_obs(env).repeat([N * A, 1])
    env.history_locomotion_obs = torch.zeros(N * A, 2100)
    body = body_policy(Ws, bs)

    def policy(obs, info={}):
        latent = ada.forward(obs)
        return body(torch.cat((obs, latent), dim=-1))
    env.locomotion_policy = policy
    env.render = lambda *a, **k: None
    env._render_headless = lambda *a, **k: None
    env.store_recording = lambda *a, **k: None
    return env


def hash_u01(seed, genv, count, k):
    """The engine's reset RNG (include/mqe_hip.h MQE_NOISE_HASH; oracle/mqe_oracle.c mqo_u01), in numpy."""
    with np.errstate(over="ignore"):
        x = (np.uint32(seed) * np.uint32(0x9E3779B1)) ^ (np.uint32(genv) * np.uint32(0x85EBCA77)) ^ \
            (np.uint32(count) * np.uint32(0xC2B2AE3D)) ^ (np.uint32(k) * np.uint32(0x27D4EB2F))
        x ^= x >> np.uint32(16); x *= np.uint32(0x85EBCA6B); x ^= x >> np.uint32(13); x *= np.uint32(0xC2B2AE35); x ^= x >> np.uint32(16)
    return np.float32(x >> np.uint32(8)) * np.float32(1.0 / 16777216.0)


class ScriptedRand:
    """torch_rand_float replacement used while generating traces: returns what the ENGINE's counter-based reset
    RNG would return for (seed 0, env id, per-env reset count, stream), by looking at the caller's `env_ids`
    (legged_robot.py:394-470).  Streams: dof (a*12+j); base x 64+a, y 72+a; base vel 80+6a+c; npc x 128+p, y 160+p.
    This makes resets inside the trace reproducible by the product without any special test mode."""

    def __init__(self, env):
        self.env = env
        self.counts = {}
        self.frames = {}

    def __call__(self, lower, upper, shape, device):
        import sys as _s
        f = _s._getframe(1)
        name = f.f_code.co_name
        env_ids = [int(e) for e in f.f_locals["env_ids"]]
        A, P = self.env.num_agents, self.env.num_npcs
        cfg = self.env.cfg
        lo, hi = np.float32(lower), np.float32(upper)
        out = np.zeros(shape, np.float32)
        if name == "_reset_dofs":
            for i, e in enumerate(env_ids):
                for c in range(shape[1]):
                    out[i, c] = (hi - lo) * hash_u01(0, e, self.counts.get(e, 0), c) + lo
            return torch.from_numpy(out)
        assert name == "_reset_root_states", name
        kinds = []
        if getattr(cfg.domain_rand, "init_base_pos_range", None) is not None:
            kinds += ["bx", "by"]
        if getattr(cfg.domain_rand, "init_npc_base_pos_range", None) is not None:
            kinds += ["nx", "ny"]
        kinds += ["vel"]
        n = self.frames.get(id(f), 0)
        self.frames[id(f)] = n + 1
        kind = kinds[n]
        for i in range(shape[0]):
            if kind in ("bx", "by", "vel"):
                e, a = env_ids[i // A], i % A
            else:
                e, a = env_ids[i // max(P, 1)], i % max(P, 1)
            cnt = self.counts.get(e, 0)
            for c in range(shape[1]):
                k = {"bx": 64 + a, "by": 72 + a, "nx": 128 + a, "ny": 160 + a, "vel": 80 + a * 6 + c}[kind]
                out[i, c] = (hi - lo) * hash_u01(0, e, cnt, k) + lo
        if kind == "vel":
            for e in env_ids:
                self.counts[e] = self.counts.get(e, 0) + 1
            self.frames.pop(id(f), None)
        return torch.from_numpy(out)


def barrier_track_for(cfg, N, seed=0):
    np.random.seed(seed)
    BarrierTrack.track_kwargs = dict(BarrierTrack_defaults)  # undo class-level state leak (barrier_track.py:62)
    t = BarrierTrack(cfg.terrain, N, cfg.env.num_agents)
    fake = types.SimpleNamespace(add_triangle_mesh=lambda *a, **k: None)
    t.add_terrain_to_sim(fake, None, "cpu")
    return t


BarrierTrack_defaults = dict(BarrierTrack.track_kwargs)


def origins_from_terrain(t, cfg, N, seed=0):
    """legged_robot.py:972-997 (_get_env_origins) with terrain_levels from the global RNG stream."""
    torch.manual_seed(seed)
    max_init_level = cfg.terrain.num_rows - 1 if not cfg.terrain.curriculum else cfg.terrain.max_init_terrain_level
    levels = torch.randint(0, max_init_level + 1, (N,))
    types_ = torch.arange(N) % cfg.terrain.num_cols
    to = torch.from_numpy(t.env_origins).float()
    ao = torch.from_numpy(t.agent_origins).float()
    info = {k: v[levels, types_] for k, v in t.env_info.items()} if getattr(t, "env_info", None) else None
    return to[levels, types_], ao[levels, types_], info, levels, types_


def obs_fields(ob):
    d = {}
    for k in ("base_pos", "base_quat", "dof_pos", "dof_vel", "lin_vel", "ang_vel", "last_action",
              "last_last_action", "projected_gravity", "clock_inputs", "base_rpy"):
        v = getattr(ob, k, None)
        if isinstance(v, torch.Tensor):
            d[k] = v.clone()
    return d


def gen_fullstep(name, cls, cfg, N, T, act, ada, action_gain=1.0, seed=0, ctrl="C", nd=3, curriculum=False):
    """Drive the reference's own step() (go1.py:35-62 / go1_football_defender.py:25-54) for T steps with the
    scripted simulator; record everything a replay needs and everything it must reproduce."""
    A = cfg.env.num_agents
    P = getattr(cfg.env, "num_npcs", 0)
    Ws, bs = pw.synthetic_body(0)
    cfg.env.num_envs = N
    t = barrier_track_for(cfg, N)
    eo, ao, info, levels, types_ = origins_from_terrain(t, cfg, N)
    env = make_env(cls, cfg, N)
    if curriculum:
        # the actors' spawn poses (legged_robot.py:864-869: env origin + U(+-x_init_range, +-y_init_range) per robot): the simulator's
        # state when _init_buffers wraps it (:566-568) and when the first reset() -- which already runs the curriculum, init_done is
        # set before it -- measures the rows; counter RNG streams 200 / 208
        xr, yr = np.float32(cfg.terrain.x_init_range), np.float32(cfg.terrain.y_init_range)
        spawn = env._all_root.view(N, A + P, 13)
        for i in range(N):
            for j in range(A):
                spawn[i, j, :3] = eo[i]
                spawn[i, j, 0] += float((np.float32(2) * hash_u01(0, i, 0xC0DE, 200 + j) - np.float32(1)) * xr)
                spawn[i, j, 1] += float((np.float32(2) * hash_u01(0, i, 0xC0DE, 208 + j) - np.float32(1)) * yr)
        spawn_root = env._all_root.clone()
    finish_env(env, eo, ao, info, Ws, bs, ada)
    if cls is Go1Sheep:
        env.sheep_movement_scale = cfg.asset.sheep_movement_scale
        env.sheep_movement_randomness = cfg.asset.sheep_movement_randomness
        env.sheep_movement_range = cfg.asset.sheep_movement_range
        np.random.seed(1)
        # _prepare_npc init-state grid (go1_sheep.py:84-118) without the asset
        nr, nc, dis = cfg.asset.num_rows, cfg.asset.num_cols, cfg.asset.dis_sheep
        kw = cfg.terrain.BarrierTrack_kwargs
        so = np.array([kw["init"]["block_length"] + kw["plane"]["block_length"] / 2 - nr // 2 * dis[0], -(nc // 2) * dis[1], 0.3])
        pos = so.copy()
        lst = []
        for i in range(nr):
            for j in range(nc):
                st = np.concatenate((pos, np.array([0., 0., 0., 1.]) + np.random.randn(4) * np.array([0, 0, np.pi, 1]), np.zeros(3), np.zeros(3)))
                lst.append(torch.tensor(st, dtype=torch.float))
                pos[1] += dis[1]
            pos[0] += dis[0]
            pos[1] = so[1]
        env.base_init_state_npc = torch.stack(lst, 0).repeat(N, 1)
        env.npc_env_origins = env.env_origins.unsqueeze(1).repeat(1, P, 1)
    rng = np.random.RandomState(seed + 7)
    sr = ScriptedRand(env)
    ref_lr.torch_rand_float = sr
    if curriculum:
        # what _get_env_origins (legged_robot.py:980-993) leaves behind for the run-time terrain curriculum (:479-503), which this trace
        # drives through the reference's own reset_idx (go1.py:123-125): levels / types / the origin table / the terrain (env_length)
        assert cfg.terrain.curriculum and cfg.terrain.num_rows > 1
        env.terrain = t
        env.terrain_levels, env.terrain_types = levels.clone(), types_.clone()
        env.max_terrain_level = cfg.terrain.num_rows
        env.terrain_origins = torch.from_numpy(t.env_origins).float()

        def scripted_randint_like(x, high):      # torch.randint_like(levels[env_ids], max_level) -> the engine's counter RNG, stream 250
            import sys as _s
            ids = [int(e) for e in _s._getframe(1).f_locals["env_ids"]]
            return torch.tensor([int(hash_u01(0, e, sr.counts.get(e, 0), 250) * np.float32(high)) for e in ids], dtype=x.dtype)
        torch.randint_like = scripted_randint_like
    noise_script = rng.standard_normal((T, N, P, 3)).astype(np.float32) if P else None
    _state = {"t": 0}
    if cls is Go1Sheep:
        import mqe.envs.npc.go1_sheep as ref_sheep
        ref_sheep.torch.randn_like = lambda x, **k: torch.from_numpy(noise_script[_state["t"]])

    ndof_env = 12 * A + env.num_actions_npc
    nb_env = env._contact.shape[0] // N
    # scripted post-substep dof states: smooth random walk around the default pose
    q0 = env.default_dof_pos[0].numpy()
    dof_script = np.zeros((T, 4, N, ndof_env, 2), np.float32)
    q = np.tile(np.concatenate([q0, np.zeros(env.num_actions_npc, np.float32)]), (N, 1)).astype(np.float32)
    for ti in range(T):
        for k in range(4):
            qd = rng.standard_normal(q.shape).astype(np.float32) * 2.0
            q = q + 0.005 * qd
            dof_script[ti, k, :, :, 0] = q
            dof_script[ti, k, :, :, 1] = qd
    # scripted root states after each policy step: random walk incl. occasional big roll -> termination
    root_script = np.zeros((T, N * (A + P), 13), np.float32)
    base = env.base_init_state.numpy().reshape(N, A, 13).copy()
    base[:, :, :3] += ao.numpy()
    allr = np.zeros((N, A + P, 13), np.float32)
    allr[:, :A] = base
    if P:
        allr[:, A:] = env.base_init_state_npc.numpy().reshape(N, P, 13)
        allr[:, A:, :3] += eo.numpy()[:, None, :]
    contact_script = np.zeros((T, N, nb_env, 3), np.float32)
    for ti in range(T):
        allr[:, :, 0:3] += rng.standard_normal((N, A + P, 3)).astype(np.float32) * 0.02
        dq = rng.standard_normal((N, A + P, 4)).astype(np.float32) * 0.05
        if ti == 4:
            dq[1 % N, 0, 0] += 1.2   # tip one robot over: roll termination on env 1
        qq = allr[:, :, 3:7] + dq
        allr[:, :, 3:7] = qq / np.linalg.norm(qq, axis=-1, keepdims=True)
        allr[:, :, 7:13] = rng.standard_normal((N, A + P, 6)).astype(np.float32) * 0.3
        root_script[ti] = allr.reshape(-1, 13)
        if ti == 7 and len(cfg.asset.terminate_after_contacts_on):
            contact_script[ti, 2 % N, 17 * (A - 1), :] = [0.5, 0.2, 3.0]  # base of last agent touches: collide
    actions = (rng.uniform(-1.3, 1.3, (T, N * (A if cls is not Go1FootballDefender else A - 1), nd)) * action_gain).astype(np.float32)     # nd > 3: command.cfg adds action columns (go1.py:64-93)
    if ctrl != "C":     # low-level control types: joint-space actions, some beyond clip_actions
        actions = rng.uniform(-1.0, 1.0, (T, N * A, 12)).astype(np.float32) * np.where(rng.rand(T, N * A, 12) < 0.02, 150.0, 1.0).astype(np.float32)

    g = env.gym
    g.dof_script = torch.from_numpy(dof_script.reshape(T * 4, N * ndof_env, 2))
    g.root_script = torch.from_numpy(root_script)
    g.contact_script = torch.from_numpy(contact_script.reshape(T, -1, 3))
    env.contact_forces = env._contact.view(N, -1, 3)

    # reset() (go1.py:147-151)
    ob = cls.reset(env)
    rec = {"reset_" + k: v for k, v in obs_fields(ob).items()}
    rec["reset_all_root"] = env.all_root_states.clone()
    rec["reset_all_dof"] = env.all_dof_states.clone()
    if curriculum:
        rec.update(spawn_all_root=spawn_root, reset_env_origins=env.env_origins.clone(), reset_terrain_levels=env.terrain_levels.clone(),
                   terrain_origins=env.terrain_origins.clone())
    out = {k: [] for k in ("torques", "reset_buf", "collide_buf", "time_out", "episode_length", "gait_indices",
                           "locomotion_obs", "history_tail", "history_sum", "loco_action", "actions_clipped",
                           "post_all_root", "post_all_dof", "r_term", "p_term", "rew")}
    obs_out = {}
    extra = {}
    # a long episode would be needed for time-outs; shorten so that one occurs inside the trace
    env.max_episode_length = 9
    for ti in range(T):
        g.step_idx = ti
        _state["t"] = ti
        g.forces = []
        ob, rew, reset_buf, extras = cls.step(env, torch.from_numpy(actions[ti]))
        out["torques"].append(torch.stack(g.forces, 0))
        out["reset_buf"].append(reset_buf.clone().bool())
        out["collide_buf"].append(env.collide_buf.clone().bool() if isinstance(env.collide_buf, torch.Tensor) else torch.zeros(N, dtype=torch.bool))
        out["time_out"].append(env.time_out_buf.clone())
        out["episode_length"].append(env.episode_length_buf.clone())
        out["gait_indices"].append(env.gait_indices.clone())
        out["locomotion_obs"].append(env.locomotion_obs.clone())
        out["history_tail"].append(env.history_locomotion_obs[:, -140:].clone())
        out["history_sum"].append(env.history_locomotion_obs.double().sum(1).float())
        out["loco_action"].append(env.last_locomotion_action.clone())
        out["actions_clipped"].append(env.actions.clone())
        out["post_all_root"].append(env.all_root_states.clone())
        out["post_all_dof"].append(env.all_dof_states.clone())
        out["r_term"].append(getattr(env, "r_term_buff", torch.zeros(N, dtype=torch.bool)).clone())
        out["p_term"].append(getattr(env, "p_term_buff", torch.zeros(N, dtype=torch.bool)).clone())
        out["rew"].append(rew.clone())
        for k, v in obs_fields(ob).items():
            obs_out.setdefault("obs_" + k, []).append(v)
        if curriculum:
            extra.setdefault("live_env_origins", []).append(env.env_origins.clone())
            extra.setdefault("live_terrain_levels", []).append(env.terrain_levels.clone())
        if cls is Go1Sheep:
            extra.setdefault("sheep_pos_avg", []).append(env.sheep_pos_avg.clone())
            extra.setdefault("sheep_pos_var", []).append(env.sheep_pos_var.clone())
    for k, v in list(out.items()) + list(obs_out.items()) + list(extra.items()):
        rec[k] = torch.stack(v, 0)
    rec.update(actions=actions, dof_script=dof_script, root_script=root_script, contact_script=contact_script,
               env_origins=eo, agent_origins=ao, terrain_levels=levels, terrain_types=types_,
               max_episode_length=np.int64(9), N=np.int64(N), A=np.int64(A), P=np.int64(P),
               base_init_state=env.base_init_state, default_dof_pos=env.default_dof_pos)
    if P:
        rec["base_init_state_npc"] = env.base_init_state_npc
        if noise_script is not None:
            rec["noise_script"] = noise_script
    if info is not None:
        for k, v in info.items():
            rec["env_info_" + k] = v
    save(name, **rec)
    return env


# --------------------------------------------------------------------------------------------
# 5. gait clock, long run (go1.py:240-279)
# --------------------------------------------------------------------------------------------
def gen_gait_clock():
    from mqe.envs.configs.go1_gate_config import Go1GateCfg
    from mqe.envs.configs.go1_seesaw_config import Go1SeesawCfg
    for nm, cfg in (("trot", Go1GateCfg), ("pace", Go1SeesawCfg)):
        env = object.__new__(Go1)
        env.cfg = cfg
        env.device = DEV
        env.dt = 0.02
        env.num_envs, env.num_agents = 1, 1
        env.locomotion_obs = Go1._fill_command_obs(env)
        env.gait_indices = torch.zeros(1)
        env.clock_inputs = torch.zeros(1, 4)
        env.doubletime_clock_inputs = torch.zeros(1, 4)
        env.halftime_clock_inputs = torch.zeros(1, 4)
        cl, gi = [], []
        for i in range(200):
            Go1._step_contact_targets(env)
            cl.append(env.clock_inputs.clone()[0])
            gi.append(env.gait_indices.clone()[0])
        save("gait_clock_" + nm, command_obs=env.locomotion_obs[0], clock=torch.stack(cl), gait=torch.stack(gi))


# --------------------------------------------------------------------------------------------
# 8. task wrappers on synthetic env attributes (wrappers/*.py)
# --------------------------------------------------------------------------------------------
class FakeEnvForWrapper:
    """Carries exactly the attributes the reference wrappers read."""

    def __init__(self, cfg, N, A, P):
        self.cfg, self.num_envs, self.num_agents, self.num_npcs = cfg, N, A, P
        self.device = DEV
        self.script = None
        self.t = -1

    def _apply(self, t):
        for k, v in self.script[t].items():
            setattr(self, k, v)

    def reset(self):
        self._apply(0)
        return self.obs_buf

    def step(self, action):
        self.last_action_in = action.clone()
        self.t += 1
        self._apply(self.t + 1)
        return self.obs_buf, None, self.reset_buf, {}


def gen_wrappers(only_pushbox=False):
    from mqe.envs.configs.go1_sheep_config import NineSheepCfg, SingleSheepCfg
    from mqe.envs.configs.go1_pushbox_config import Go1PushboxCfg
    from mqe.envs.wrappers.go1_pushbox_wrapper import Go1PushboxWrapper
    from mqe.envs.configs.go1_seesaw_config import Go1SeesawCfg
    from mqe.envs.configs.go1_football_config import Go1FootballDefenderCfg
    from mqe.envs.wrappers.go1_sheep_wrapper import Go1SheepWrapper
    from mqe.envs.wrappers.go1_seesaw_wrapper import Go1SeesawWrapper
    from mqe.envs.wrappers.go1_football_wrapper import Go1FootballDefenderWrapper
    rng = np.random.RandomState(11)
    T, N = 6, 5
    tasks = (("sheep_hard", NineSheepCfg, Go1SheepWrapper), ("sheep_easy", SingleSheepCfg, Go1SheepWrapper),
             ("seesaw", Go1SeesawCfg, Go1SeesawWrapper), ("football_defender", Go1FootballDefenderCfg, Go1FootballDefenderWrapper))
    if only_pushbox:   # own random stream so that the older fixtures stay byte-identical
        tasks = (("pushbox", Go1PushboxCfg, Go1PushboxWrapper),)
        rng = np.random.RandomState(31)
    for name, cfg, W in tasks:
        A, P = cfg.env.num_agents, cfg.env.num_npcs
        fe = FakeEnvForWrapper(cfg, N, A, P)
        eo = torch.tensor(rng.uniform(0, 3, (N, 3)).astype(np.float32))
        eo[:, 2] = 0
        fe.env_origins = eo
        fe.npc_env_origins = eo.unsqueeze(1).repeat(1, P, 1)
        gate_dev = torch.tensor(rng.uniform(-0.4, 0.4, (N, 2)).astype(np.float32))
        gate_dev[:, 0] = 0
        fe.gate_pos_env = None
        if name == "football_defender":
            fe.gate_pos = eo.clone()
            fe.gate_pos[:, 0] += cfg.terrain.BarrierTrack_kwargs["init"]["block_length"] + cfg.terrain.BarrierTrack_kwargs["plane"]["block_length"]
        script = []
        rec = {"env_origins": eo, "gate_deviation": gate_dev.clone()}
        for t in range(T + 1):
            ob = types.SimpleNamespace()
            ob.base_pos = torch.tensor(rng.uniform(-1, 9, (N * A, 3)).astype(np.float32))
            ob.base_pos[:, 2] = torch.tensor(rng.uniform(0.2, 1.5, N * A).astype(np.float32))
            if t == 3:
                ob.base_pos[1 * A + 1, :2] = ob.base_pos[1 * A, :2] + 0.2  # agents close: distance punishment
            ob.base_rpy = torch.tensor(rng.uniform(0, 6.28, (N * A, 3)).astype(np.float32))
            ob.env_info = {"gate_deviation": gate_dev.clone()}
            npc = torch.tensor(rng.uniform(-1, 12, (N * P, 13)).astype(np.float32))
            npc[:, :3] += fe.npc_env_origins.reshape(-1, 3)
            d = dict(obs_buf=ob, root_states_npc=npc,
                     reset_buf=torch.tensor(rng.rand(N) < 0.3),
                     collide_buf=torch.tensor(rng.rand(N) < 0.3),
                     r_term_buff=torch.tensor(rng.rand(N) < 0.3), p_term_buff=torch.tensor(rng.rand(N) < 0.2),
                     sheep_pos_avg=torch.tensor(rng.uniform(0, 8, (N, 2)).astype(np.float32)),
                     sheep_pos_var=torch.tensor(rng.uniform(0, 3, (N,)).astype(np.float32)))
            d["reset_ids"] = d["reset_buf"].nonzero(as_tuple=False).flatten()
            script.append(d)
            for k in ("root_states_npc", "reset_buf", "collide_buf", "r_term_buff", "p_term_buff", "sheep_pos_avg", "sheep_pos_var"):
                rec.setdefault(k, []).append(d[k])
            rec.setdefault("base_pos", []).append(ob.base_pos)
            rec.setdefault("base_rpy", []).append(ob.base_rpy)
        fe.script = script
        w = W(fe)
        Aw = w.num_agents
        obs0 = w.reset()
        acts = rng.uniform(-1.5, 1.5, (T, N, Aw, 3)).astype(np.float32)
        obs_l, rew_l, act_l = [], [], []
        for t in range(T):
            o, r, term, info = w.step(torch.from_numpy(acts[t]))
            obs_l.append(o.clone())
            rew_l.append(r.clone())
            act_l.append(fe.last_action_in)
        for k in list(rec.keys()):
            if isinstance(rec[k], list):
                rec[k] = torch.stack(rec[k], 0)
        rb = {k: float(v) for k, v in w.reward_buffer.items()}
        save("wrapper_" + name, obs_reset=obs0, obs=torch.stack(obs_l), reward=torch.stack(rew_l), env_action=torch.stack(act_l),
             actions=acts, reward_buffer_keys=np.array(list(rb.keys())), reward_buffer_vals=np.array(list(rb.values()), np.float64),
             obs_dim=np.int64(w.observation_space.shape[0]), **rec)


# --------------------------------------------------------------------------------------------
# 10/11. BarrierTrack terrain + config dump
# --------------------------------------------------------------------------------------------
def gen_gate_wrapper():
    """W1: the gate wrapper's observation / reward code is COMMENTED OUT upstream (go1_gate_wrapper.py:41-54,64-67,78-154; the
    live methods return 0).  The commented block is executable Python: it is activated here IN MEMORY -- '# ' stripped from the
    code lines of those ranges (prose comments inside it are '# # ...' and stay comments), the two `obs = 0` overrides dropped --
    compiled as a subclass body under the stub, and driven like the other wrappers.  Nothing of the reference's text is stored:
    the fixture holds the scripted env attributes and the outputs."""
    import re
    from mqe.envs.configs.go1_gate_config import Go1GateCfg
    src = open(os.path.join(REF, "mqe/envs/wrappers/go1_gate_wrapper.py")).read().split("\n")
    live = []
    for ln, line in enumerate(src, 1):
        in_code = (41 <= ln <= 54) or (64 <= ln <= 67) or (78 <= ln <= 154)
        mm = re.match(r"^(\s*)# (.*)$", line)
        if in_code and mm and not mm.group(2).startswith("#"):
            line = mm.group(1) + mm.group(2)
        if line.strip() in ("obs = 0", "obs, reward = 0, 0"):
            continue
        live.append(line)
    mod = types.ModuleType("go1_gate_wrapper_live")
    exec(compile("\n".join(live), "go1_gate_wrapper.py[uncommented]", "exec"), mod.__dict__)
    W = mod.Go1GateWrapper
    cfg = Go1GateCfg
    rng = np.random.RandomState(53)
    T, N = 7, 6
    A, P = cfg.env.num_agents, 0
    fe = FakeEnvForWrapper(cfg, N, A, P)
    eo = torch.tensor(rng.uniform(0, 3, (N, 3)).astype(np.float32))
    eo[:, 2] = 0
    fe.env_origins = eo
    gate_dev = torch.tensor(rng.uniform(-0.4, 0.4, (N, 2)).astype(np.float32))
    gate_dev[:, 0] = 0
    script = []
    rec = {"env_origins": eo, "gate_deviation": gate_dev.clone()}
    for t in range(T + 1):
        ob = types.SimpleNamespace()
        ob.base_pos = torch.tensor(rng.uniform(-1, 6, (N * A, 3)).astype(np.float32))
        ob.base_pos[:, 2] = torch.tensor(rng.uniform(0.2, 0.5, N * A).astype(np.float32))
        if t in (2, 5):
            ob.base_pos[1 * A + 1, :2] = ob.base_pos[1 * A, :2] + 0.2      # agents close: distance punishment
            ob.base_pos[4 * A + 1, :2] = ob.base_pos[4 * A, :2] - 0.1
        ob.base_rpy = torch.tensor(rng.uniform(0, 6.28, (N * A, 3)).astype(np.float32))
        ob.lin_vel = torch.tensor(rng.uniform(-1, 1, (N * A, 3)).astype(np.float32))
        ob.env_info = {"gate_deviation": gate_dev.clone()}
        d = dict(obs_buf=ob, reset_buf=torch.tensor(rng.rand(N) < 0.3), collide_buf=torch.tensor(rng.rand(N) < 0.3),
                 root_states_npc=torch.zeros(0, 13), r_term_buff=torch.zeros(N, dtype=torch.bool), p_term_buff=torch.zeros(N, dtype=torch.bool),
                 sheep_pos_avg=torch.zeros(N, 2), sheep_pos_var=torch.zeros(N))
        d["reset_ids"] = d["reset_buf"].nonzero(as_tuple=False).flatten()
        script.append(d)
        for k in ("root_states_npc", "reset_buf", "collide_buf", "r_term_buff", "p_term_buff", "sheep_pos_avg", "sheep_pos_var"):
            rec.setdefault(k, []).append(d[k])
        rec.setdefault("base_pos", []).append(ob.base_pos)
        rec.setdefault("base_rpy", []).append(ob.base_rpy)
    fe.script = script
    w = W(fe)
    obs0 = w.reset()
    assert isinstance(obs0, torch.Tensor) and obs0.shape == (N, A, 14 + A), "the commented-out observation code did not run"
    acts = rng.uniform(-1.5, 1.5, (T, N, A, 3)).astype(np.float32)
    obs_l, rew_l, act_l = [], [], []
    for t in range(T):
        o, r, term, info = w.step(torch.from_numpy(acts[t]))
        obs_l.append(o.clone()); rew_l.append(r.clone()); act_l.append(fe.last_action_in)
    for k in list(rec.keys()):
        if isinstance(rec[k], list):
            rec[k] = torch.stack(rec[k], 0)
    rb = {k: float(v) for k, v in w.reward_buffer.items()}
    assert abs(rb["agent distance punishment"]) > 0 and abs(rb["success reward"]) > 0 and abs(rb["contact punishment"]) > 0
    save("wrapper_gate", obs_reset=obs0, obs=torch.stack(obs_l), reward=torch.stack(rew_l), env_action=torch.stack(act_l),
         actions=acts, reward_buffer_keys=np.array(list(rb.keys())), reward_buffer_vals=np.array(list(rb.values()), np.float64),
         obs_dim=np.int64(w.observation_space.shape[0]), **rec)


def gen_game_wrapper():
    """Go1FootballGameWrapper (go1_football_wrapper.py:93-156) is a stub upstream: None observations, zero reward (N, 4);
    what it does define is the action clip + scale handed to Go1.step."""
    from mqe.envs.configs.go1_football_config import Go1Football1vs1Cfg, Go1Football2vs2Cfg
    from mqe.envs.wrappers.go1_football_wrapper import Go1FootballGameWrapper
    rng = np.random.RandomState(23)
    T, N = 4, 3
    for name, cfg in (("football_game_1v1", Go1Football1vs1Cfg), ("football_game_2v2", Go1Football2vs2Cfg)):
        A, P = cfg.env.num_agents, cfg.env.num_npcs
        fe = FakeEnvForWrapper(cfg, N, A, P)
        fe.env_origins = torch.zeros(N, 3)
        script = []
        for t in range(T + 1):
            ob = types.SimpleNamespace()
            ob.base_pos = torch.tensor(rng.uniform(-1, 9, (N * A, 3)).astype(np.float32))
            ob.base_rpy = torch.tensor(rng.uniform(0, 6.28, (N * A, 3)).astype(np.float32))
            script.append(dict(obs_buf=ob, root_states_npc=torch.tensor(rng.uniform(-1, 12, (N * P, 13)).astype(np.float32)),
                               reset_buf=torch.tensor(rng.rand(N) < 0.3)))
        fe.script = script
        w = Go1FootballGameWrapper(fe)
        obs0 = w.reset()
        assert obs0 is None
        acts = rng.uniform(-1.5, 1.5, (T, N, A, 3)).astype(np.float32)
        rew_l, act_l, term_l = [], [], []
        for t in range(T):
            o, r, term, info = w.step(torch.from_numpy(acts[t]))
            assert o is None
            rew_l.append(r.clone()); act_l.append(fe.last_action_in); term_l.append(term.clone())
        save("wrapper_" + name, reward=torch.stack(rew_l), env_action=torch.stack(act_l), actions=acts, termination=torch.stack(term_l),
             obs_dim=np.int64(w.observation_space.shape[0]), step_count=np.int64(w.reward_buffer["step count"]))


def gen_rotation_wrapper():
    """Go1RotationWrapper (go1_rotation_wrapper.py).  Its distance term only broadcasts for num_envs in {1, 2}: N = 2."""
    from mqe.envs.configs.go1_rotation_config import Go1RotationCfg
    from mqe.envs.wrappers.go1_rotation_wrapper import Go1RotationWrapper
    rng = np.random.RandomState(41)
    T, N = 8, 2
    cfg = Go1RotationCfg
    A, P = cfg.env.num_agents, cfg.env.num_npcs
    fe = FakeEnvForWrapper(cfg, N, A, P)
    script, rec = [], {}
    for t in range(T + 1):
        ob = types.SimpleNamespace()
        ob.base_pos = torch.tensor(rng.uniform(0, 6, (N * A, 3)).astype(np.float32))
        ob.base_pos[:, 2] = torch.tensor(rng.uniform(0.2, 0.5, N * A).astype(np.float32))
        ob.base_rpy = torch.tensor(rng.uniform(0, 6.28, (N * A, 3)).astype(np.float32))
        script.append(dict(obs_buf=ob, reset_buf=torch.tensor(rng.rand(N) < 0.3)))
        rec.setdefault("base_pos", []).append(ob.base_pos); rec.setdefault("base_rpy", []).append(ob.base_rpy)
        rec.setdefault("reset_buf", []).append(script[-1]["reset_buf"])
    fe.script = script
    w = Go1RotationWrapper(fe)
    obs0 = w.reset()
    acts = rng.uniform(-1.5, 1.5, (T, N, A, 3)).astype(np.float32)
    obs_l, rew_l, act_l = [], [], []
    for t in range(T):
        a_in = torch.from_numpy(acts[t].copy())
        o, r, term, info = w.step(a_in)
        obs_l.append(o.clone()); rew_l.append(r.clone()); act_l.append(fe.last_action_in)
    rb = {k: float(v) for k, v in w.reward_buffer.items()}
    save("wrapper_rotation", obs_reset=obs0, obs=torch.stack(obs_l), reward=torch.stack(rew_l), env_action=torch.stack(act_l), actions=acts,
         reward_buffer_keys=np.array(list(rb.keys())), reward_buffer_vals=np.array(list(rb.values()), np.float64),
         obs_dim=np.int64(w.observation_space.shape[0]), **{k: torch.stack(v, 0) for k, v in rec.items()})


def gen_scenery_wrappers():
    """Go1BridgeWrapper / Go1WrestlingWrapper (fixed scenery tasks): the wrappers read base pos / rpy / quat only."""
    from isaacgym.torch_utils import quat_from_euler_xyz
    from mqe.envs.configs.go1_bridge_config import Go1BridgeCfg
    from mqe.envs.configs.go1_wrestling_config import Go1WrestlingCfg
    from mqe.envs.wrappers.go1_bridge_wrapper import Go1BridgeWrapper
    from mqe.envs.wrappers.go1_wrestling_wrapper import Go1WrestlingWrapper
    T, N = 8, 3
    for name, cfg, W, seed in (("bridge", Go1BridgeCfg, Go1BridgeWrapper, 51), ("wrestling", Go1WrestlingCfg, Go1WrestlingWrapper, 52)):
        rng = np.random.RandomState(seed)
        A, P = cfg.env.num_agents, cfg.env.num_npcs
        fe = FakeEnvForWrapper(cfg, N, A, P)
        fe.env_agent_indices = torch.arange(N * A).reshape(N, A)
        fe.base_init_state = torch.tensor(rng.uniform(0, 2, (N * A, 13)).astype(np.float32))
        script, rec = [], {}
        for t in range(T + 1):
            ob = types.SimpleNamespace()
            ob.base_pos = torch.tensor(rng.uniform(0, 9, (N * A, 3)).astype(np.float32))
            ob.base_pos[:, 2] = torch.tensor(rng.uniform(0.2, 1.5, N * A).astype(np.float32))
            rpy = rng.uniform(-3.1, 3.1, (N * A, 3)).astype(np.float32)
            rpy[rng.rand(N * A) < 0.5, :2] *= 0.1                      # half of the robots upright
            q = quat_from_euler_xyz(torch.tensor(rpy[:, 0]), torch.tensor(rpy[:, 1]), torch.tensor(rpy[:, 2]))
            ob.base_quat = q
            from isaacgym.torch_utils import get_euler_xyz
            r_, p_, y_ = get_euler_xyz(q)
            ob.base_rpy = torch.stack([r_, p_, y_], dim=-1)
            script.append(dict(obs_buf=ob, reset_buf=torch.tensor(rng.rand(N) < 0.3)))
            for k in ("base_pos", "base_rpy", "base_quat"):
                rec.setdefault(k, []).append(getattr(ob, k))
            rec.setdefault("reset_buf", []).append(script[-1]["reset_buf"])
        fe.script = script
        w = W(fe)
        obs0 = w.reset()
        acts = rng.uniform(-1.5, 1.5, (T, N, A, 3)).astype(np.float32)
        obs_l, rew_l, act_l = [], [], []
        for t in range(T):
            o, r, term, info = w.step(torch.from_numpy(acts[t].copy()))
            obs_l.append(o.clone()); rew_l.append(r.clone()); act_l.append(fe.last_action_in)
        rb = {k: float(v) for k, v in w.reward_buffer.items()}
        save("wrapper_" + name, obs_reset=obs0, obs=torch.stack(obs_l), reward=torch.stack(rew_l), env_action=torch.stack(act_l), actions=acts,
             reward_buffer_keys=np.array(list(rb.keys())), reward_buffer_vals=np.array(list(rb.values()), np.float64),
             obs_dim=np.int64(w.observation_space.shape[0]), **{k: torch.stack(v, 0) for k, v in rec.items()})


def gen_tug_wrapper():
    """Go1TugWrapper: scripted base pos / rpy, slider dof state, env resets; the gym call it makes is stubbed."""
    from mqe.envs.configs.go1_tug_config import Go1TugCfg
    from mqe.envs.wrappers.go1_tug_wrapper import Go1TugWrapper
    rng = np.random.RandomState(61)
    T, N = 24, 1          # upstream `reward[:, 0] += success_reward[:, 0]` ((N,1) += (N,)) only broadcasts for num_envs = 1
    cfg = Go1TugCfg
    A, P = cfg.env.num_agents, cfg.env.num_npcs
    fe = FakeEnvForWrapper(cfg, N, A, P)
    fe.npc_indices = torch.arange(N).reshape(N, 1)
    fe.all_dof_states = torch.zeros(N * (12 * A + 1), 2)
    fe.sim = None
    zero_log = []
    fe.gym = types.SimpleNamespace(set_dof_state_tensor_indexed=lambda *a: zero_log.append(fe.dof_state_npc[:, 0, :].clone()))
    script, rec = [], {}
    for t in range(T + 1):
        ob = types.SimpleNamespace()
        ob.base_pos = torch.tensor(rng.uniform(0, 3, (N * A, 3)).astype(np.float32))
        ob.base_pos[:, 1] = torch.tensor(rng.uniform(-3, 3, N * A).astype(np.float32))
        ob.base_rpy = torch.tensor(rng.uniform(0, 6.28, (N * A, 3)).astype(np.float32))
        npc = torch.tensor(rng.uniform(-1.5, 1.5, (N, 1, 2)).astype(np.float32))
        rb = torch.tensor(rng.rand(N) < 0.25)
        dct = dict(obs_buf=ob, dof_state_npc=npc, reset_buf=rb)
        dct["reset_ids"] = rb.nonzero(as_tuple=False).flatten()
        script.append(dct)
        rec.setdefault("base_pos", []).append(ob.base_pos); rec.setdefault("base_rpy", []).append(ob.base_rpy)
        rec.setdefault("dof_state_npc", []).append(npc.clone()); rec.setdefault("reset_buf", []).append(rb)
    fe.script = script
    w = Go1TugWrapper(fe)
    obs0 = w.reset()
    acts = rng.uniform(-1.5, 1.5, (T, N, A, 3)).astype(np.float32)
    obs_l, rew_l, act_l, dic_l, npc_after = [], [], [], [], []
    for t in range(T):
        o, r, term, info = w.step(torch.from_numpy(acts[t].copy()))
        obs_l.append(o.clone()); rew_l.append(r.clone()); act_l.append(fe.last_action_in); dic_l.append(w.reset_dic.clone())
    rb = {k: float(v) for k, v in w.reward_buffer.items()}
    save("wrapper_tug", obs_reset=obs0, obs=torch.stack(obs_l), reward=torch.stack(rew_l), env_action=torch.stack(act_l), actions=acts,
         reset_dic=torch.stack(dic_l), reward_buffer_keys=np.array(list(rb.keys())), reward_buffer_vals=np.array(list(rb.values()), np.float64),
         obs_dim=np.int64(w.observation_space.shape[0]), **{k: torch.stack(v, 0) for k, v in rec.items()})


def rle_rows(hf):
    """two-level heightfield -> per-row run-length list (value, start, stop)"""
    runs = []
    for r in range(hf.shape[0]):
        row = hf[r]
        ch = np.flatnonzero(np.diff(row)) + 1
        st = np.concatenate([[0], ch])
        en = np.concatenate([ch, [row.size]])
        for s, e in zip(st, en):
            if row[s] != 0:
                runs.append((r, s, e, row[s]))
    return np.array(runs, np.float32).reshape(-1, 4)


def perlin_terrain_cfgs():
    """Terrain configs that switch on what every shipped task leaves off: the whole-map Perlin relief and
    curriculum rows.  Returns {name: (terrain cfg class, num_agents)}; shared with tests/test_terrain_configs.py by name."""
    from mqe.envs.configs.go1_gate_config import Go1GateCfg
    base = Go1GateCfg.terrain
    kw = dict(base.BarrierTrack_kwargs)

    def variant(name, cfg_over, kw_over):
        k2 = dict(kw); k2.update(kw_over)
        return type(name, (base,), dict(cfg_over, BarrierTrack_kwargs=k2))
    out = {}
    out["perlin_map"] = (variant("PerlinMap", dict(num_rows=2, num_cols=2, border_size=1, TerrainPerlin_kwargs=dict(zScale=0.12, frequency=10)),
                                 dict(add_perlin_noise=True, border_perlin_noise=True)), 2)
    out["perlin_curriculum"] = (variant("PerlinCurriculum", dict(num_rows=2, num_cols=1, border_size=1, curriculum=True,
                                                                  TerrainPerlin_kwargs=dict(zScale=[0.05, 0.1], frequency=10)),
                                        dict(add_perlin_noise=True, border_perlin_noise=True, curriculum_perlin=True, no_perlin_threshold=0.06,
                                             border_height=0.3)), 2)
    out["perlin_tracks_only"] = (variant("PerlinTracksOnly", dict(num_rows=2, num_cols=1, border_size=1, TerrainPerlin_kwargs=dict(zScale=[0.05, 0.1], frequency=10)),
                                         dict(add_perlin_noise=True, border_perlin_noise=False)), 2)
    # a (lo, hi) wall_height: every block painter draws its own wall height (barrier_track.py:167-173,191-199,218-239)
    out["wall_heights"] = (variant("WallHeights", dict(num_rows=2, num_cols=2, border_size=1), dict(wall_height=(0.3, 0.7))), 2)
    return out


def gen_perlin_terrain():
    """BarrierTrack with Perlin noise / curriculum at np.random.seed(0): every third heightfield sample + whole-field sums, origins,
    gate deviations (barrier_track.py:372-393,421-459,635-638; perlin.py:33-72)"""
    for name, (tcfg, A) in perlin_terrain_cfgs().items():
        t = barrier_track_for(types.SimpleNamespace(terrain=tcfg, env=types.SimpleNamespace(num_agents=A)), 4)
        hf = np.asarray(t.heightfield_raw, np.float64)
        save("terrain_" + name, shape=np.array(hf.shape), sub=hf[::3, ::3].astype(np.float32), total=np.float64(hf.sum()), total_sq=np.float64((hf ** 2).sum()),
             row_sums=hf.sum(1), env_origins=t.env_origins, agent_origins=t.agent_origins,
             gate_deviation=np.asarray(t.env_info["gate_deviation"]) if t.env_info else np.zeros(0))


def gen_perlin_class():
    """TerrainPerlin (perlin.py:9-32,95-117; selectable through the registry, __init__.py:6) at np.random.seed(0): the int16 height
    samples (every fourth + sums) and the env origins of a square 2 x 2 map"""
    from mqe.utils.terrain import get_terrain_cls
    from mqe.envs.configs.go1_gate_config import Go1GateCfg
    base = Go1GateCfg.terrain
    tcfg = type("PerlinClassTerrain", (base,), dict(selected="TerrainPerlin", num_rows=2, num_cols=2, terrain_length=4.0, terrain_width=4.0,
                                                    TerrainPerlin_kwargs=dict(zScale=0.1, frequency=5)))
    np.random.seed(0)
    t = get_terrain_cls("TerrainPerlin")(tcfg, 4, 1)
    t.add_terrain_to_sim(types.SimpleNamespace(add_triangle_mesh=lambda *a, **k: None), "sim", "cpu")
    hs = np.asarray(t.heightsamples)
    save("terrain_perlin_class", shape=np.array(hs.shape), sub=hs[::4, ::4].astype(np.int16), total=np.int64(hs.astype(np.int64).sum()),
         row_sums=hs.astype(np.int64).sum(1), env_origins=np.asarray(t.env_origins, np.float64))


def gen_urdf_facts():
    """go1.urdf read with nothing but xml.etree -- every <inertial>, <collision> and <joint> as plain numbers -- so that the product's
    own URDF reader (mqe/utils/urdf_model.py, which feeds BOTH engines) is checked against an independent reading
    (tests/test_models_oracle.py recombines the fixed-joint subtrees itself)."""
    import xml.etree.ElementTree as ET

    def nums(t, d):
        return [float(x) for x in (t if t is not None else d).split()]
    root = ET.parse(os.path.join(REF, "resources/robots/go1/urdf/go1.urdf")).getroot()
    links, joints = {}, {}
    for l in root.findall("link"):
        e = {"collisions": []}
        ine = l.find("inertial")
        if ine is not None:
            o = ine.find("origin")
            e["mass"] = float(ine.find("mass").get("value"))
            e["com_xyz"] = nums(o.get("xyz") if o is not None else None, "0 0 0")
            e["com_rpy"] = nums(o.get("rpy") if o is not None else None, "0 0 0")
            i = ine.find("inertia")
            e["inertia"] = [float(i.get(k)) for k in ("ixx", "ixy", "ixz", "iyy", "iyz", "izz")]
        for c in l.findall("collision"):
            o, g = c.find("origin"), c.find("geometry")
            k = list(g)[0]
            e["collisions"].append({"type": k.tag, "params": {a: nums(v, "0") for a, v in k.attrib.items() if a != "filename"},
                                    "xyz": nums(o.get("xyz") if o is not None else None, "0 0 0"),
                                    "rpy": nums(o.get("rpy") if o is not None else None, "0 0 0")})
        links[l.get("name")] = e
    for j in root.findall("joint"):
        o, ax, lim = j.find("origin"), j.find("axis"), j.find("limit")
        joints[j.get("name")] = {"type": j.get("type"), "parent": j.find("parent").get("link"), "child": j.find("child").get("link"),
                                 "xyz": nums(o.get("xyz") if o is not None else None, "0 0 0"), "rpy": nums(o.get("rpy") if o is not None else None, "0 0 0"),
                                 "axis": nums(ax.get("xyz") if ax is not None else None, "1 0 0"),
                                 "limit": {k: float(v) for k, v in lim.attrib.items()} if lim is not None else {},
                                 "dont_collapse": j.get("dont_collapse") == "true"}
    with open(os.path.join(GOLD, "go1_urdf_facts.json"), "w") as f:
        json.dump({"links": links, "joints": joints}, f, indent=0, sort_keys=True)
    print("wrote go1_urdf_facts.json", len(links), "links", len(joints), "joints")


def gen_terrain_and_configs():
    cfgd = {}
    for task in ("go1gate", "go1sheep-easy", "go1sheep-hard", "go1seesaw", "go1football-defender", "go1football-1vs1", "go1football-2vs2", "go1pushbox", "go1revolvingdoor", "go1bridge", "go1wrestling", "go1tug"):
        cfg = ref_utils.ENV_DICT[task]["config"]
        t = barrier_track_for(cfg, 8)
        hf = t.heightfield_raw
        levels = np.unique(hf)
        save("terrain_" + task, runs=rle_rows(hf), shape=np.array(hf.shape), levels=levels,
             env_origins=t.env_origins, agent_origins=t.agent_origins,
             gate_deviation=(t.env_info["gate_deviation"].numpy() if getattr(t, "env_info", None) and "gate_deviation" in t.env_info else np.zeros((0,))),
             track_resolution=np.array(t.track_resolution), border=np.int64(t.border),
             vertical_scale=np.float32(cfg.terrain.vertical_scale), horizontal_scale=np.float32(cfg.terrain.horizontal_scale))
        d = class_to_dict(cfg)
        cfgd[task] = json.loads(json.dumps(d, default=lambda o: class_to_dict(o) if hasattr(o, "__dict__") else str(o)))
    with open(os.path.join(GOLD, "configs.json"), "w") as f:
        json.dump(cfgd, f, indent=1, sort_keys=True)
    print("wrote configs.json")


# --------------------------------------------------------------------------------------------
# 12. OpenRL adapter transforms (openrl_ws/utils.py:40-90)
# --------------------------------------------------------------------------------------------
def gen_adapter():
    sys.path.insert(0, os.path.join(REF))
    from openrl_ws.utils import mqe_openrl_wrapper, SingleAgentWrapper
    rng = np.random.RandomState(5)
    N, A, D = 4, 2, 7

    class E:
        num_envs, num_agents = N, A
        action_space = types.SimpleNamespace(shape=(3,))
        observation_space = types.SimpleNamespace(shape=(D,))
        reward_buffer = {"x reward": torch.tensor(3.0), "y punishment": torch.tensor(-1.0), "other": 2.0, "step count": 5}

        def reset(self):
            return torch.tensor(obs[0])

        def step(self, a):
            self.a = a.clone()
            return torch.tensor(obs[1]), torch.tensor(rew), torch.tensor(done), {}

    obs = rng.standard_normal((2, N, A, D)).astype(np.float32)
    rew = rng.standard_normal((N, A)).astype(np.float32)
    done = rng.rand(N) < 0.5
    e = E()
    w = mqe_openrl_wrapper(e)
    w.num_envs = N
    o0 = w.reset()
    act = rng.uniform(-3, 3, (N, A, 3)).astype(np.float32)
    o1, r1, d1, infos = w.step(act)
    br = w.batch_rewards(None)
    save("openrl_adapter", obs=obs, rew=rew, done=done, act=act, env_action=e.a, o0=o0, o1=o1, r1=r1, d1=d1,
         n_infos=np.int64(len(infos)), br_keys=np.array(list(br.keys())), br_vals=np.array([float(v) for v in br.values()]))


def run_stage(fn, *args, **kw):
    """Run one generator stage in a forked child (see the module docstring); the parent only waits for it."""
    sys.stdout.flush(); sys.stderr.flush()
    pid = os.fork()
    if pid == 0:
        code = 0
        try:
            fn(*args, **kw)
        except BaseException:
            traceback.print_exc()
            code = 1
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(code)
    _, status = os.waitpid(pid, 0)
    if status != 0:
        raise SystemExit(f"gen_golden: stage {fn.__name__}{args[:1]} failed (status {status})")


def main():
    global GOLD, ASSETS
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--out", default="", help="write under DIR/golden and DIR/assets instead of tests/golden and the package assets")
    a = ap.parse_args()
    only = set(a.only.split(",")) if a.only else None
    if a.out:
        GOLD = os.path.join(os.path.abspath(a.out), "golden")
        ASSETS = os.path.join(os.path.abspath(a.out), "assets")
    os.makedirs(GOLD, exist_ok=True)
    os.makedirs(ASSETS, exist_ok=True)
    act, ada = gen_mlps()

    def want(n):
        return only is None or n in only
    if want("gait"):
        run_stage(gen_gait_clock)
    if want("fullstep"):
        from mqe.envs.configs.go1_gate_config import Go1GateCfg
        from mqe.envs.configs.go1_sheep_config import NineSheepCfg
        from mqe.envs.configs.go1_seesaw_config import Go1SeesawCfg
        from mqe.envs.configs.go1_football_config import Go1FootballDefenderCfg
        run_stage(gen_fullstep, "fullstep_gate", Go1, Go1GateCfg, N=3, T=12, act=act, ada=ada)
        run_stage(gen_fullstep, "fullstep_seesaw", Go1Object, Go1SeesawCfg, N=3, T=12, act=act, ada=ada)
        run_stage(gen_fullstep, "fullstep_football", Go1FootballDefender, Go1FootballDefenderCfg, N=3, T=12, act=act, ada=ada)
        run_stage(gen_fullstep, "fullstep_sheep", Go1Sheep, NineSheepCfg, N=3, T=12, act=act, ada=ada)
    if want("fullstep_game"):     # the two free-play football tasks (Go1Object + ball, 2 and 4 robots)
        from mqe.envs.configs.go1_football_config import Go1Football1vs1Cfg, Go1Football2vs2Cfg
        run_stage(gen_fullstep, "fullstep_football1v1", Go1Object, Go1Football1vs1Cfg, N=3, T=12, act=act, ada=ada)
        run_stage(gen_fullstep, "fullstep_football2v2", Go1Object, Go1Football2vs2Cfg, N=2, T=10, act=act, ada=ada)
    if want("fullstep_pushbox"):
        from mqe.envs.configs.go1_pushbox_config import Go1PushboxCfg
        run_stage(gen_fullstep, "fullstep_pushbox", Go1Object, Go1PushboxCfg, N=3, T=12, act=act, ada=ada)
    if want("wrappers"):
        run_stage(gen_wrappers)
    if want("wrapper_game"):
        run_stage(gen_game_wrapper)
    if want("wrapper_gate"):
        run_stage(gen_gate_wrapper)
    if want("wrapper_pushbox"):
        run_stage(gen_wrappers, only_pushbox=True)
    if want("fullstep_rotation"):
        from mqe.envs.configs.go1_rotation_config import Go1RotationCfg
        run_stage(gen_fullstep, "fullstep_rotation", Go1Object, Go1RotationCfg, N=2, T=12, act=act, ada=ada)
    if want("wrapper_rotation"):
        run_stage(gen_rotation_wrapper)
    if want("fullstep_scenery"):
        from mqe.envs.configs.go1_bridge_config import Go1BridgeCfg
        from mqe.envs.configs.go1_wrestling_config import Go1WrestlingCfg
        run_stage(gen_fullstep, "fullstep_bridge", Go1Object, Go1BridgeCfg, N=2, T=12, act=act, ada=ada)
        run_stage(gen_fullstep, "fullstep_wrestling", Go1Object, Go1WrestlingCfg, N=2, T=12, act=act, ada=ada)
    if want("wrapper_scenery"):
        run_stage(gen_scenery_wrappers)
    if want("fullstep_pvt"):       # Go1.step's else-branch (go1.py:42-44): control types P / V / T on the gate scene
        from mqe.envs.configs.go1_gate_config import Go1GateCfg
        for c in ("P", "V", "T"):
            ctl = type("control", (Go1GateCfg.control,), {"control_type": c})
            cfg_c = type("Go1Gate" + c + "Cfg", (Go1GateCfg,), {"control": ctl})
            run_stage(gen_fullstep, "fullstep_gate_" + c, Go1, cfg_c, N=3, T=10, act=act, ada=ada, ctrl=c)
    if want("fullstep_cmd"):       # command.cfg beyond the velocity command (go1.py:64-93, slots of _fill_command_obs :411-479): 11 action columns per robot
        from mqe.envs.configs.go1_gate_config import Go1GateCfg
        cc = type("cfg", (Go1GateCfg.command.cfg,), dict(body_height=True, gait_freq=True, footswing_height=True, body_pose=True, stance_width=True, stance_length=True, aux_reward=True))
        cmd = type("command", (Go1GateCfg.command,), {"cfg": cc})
        cfg_c = type("Go1GateCmdCfg", (Go1GateCfg,), {"command": cmd})
        run_stage(gen_fullstep, "fullstep_gate_cmd", Go1, cfg_c, N=3, T=12, act=act, ada=ada, nd=11)
    if want("fullstep_curriculum"):       # the run-time terrain curriculum (legged_robot.py:479-503 through go1.py:123-125) on the push-box scene: 3 rows x 2 columns of tracks
        from mqe.envs.configs.go1_pushbox_config import Go1PushboxCfg
        ter = type("terrain", (Go1PushboxCfg.terrain,), dict(num_rows=3, num_cols=2, curriculum=True, max_init_terrain_level=1))
        cfg_c = type("Go1PushboxCurriculumCfg", (Go1PushboxCfg,), {"terrain": ter})
        run_stage(gen_fullstep, "fullstep_pushbox_curriculum", Go1Object, cfg_c, N=6, T=14, act=act, ada=ada, curriculum=True)
    if want("fullstep_tug"):
        from mqe.envs.configs.go1_tug_config import Go1TugCfg
        run_stage(gen_fullstep, "fullstep_tug", Go1Object, Go1TugCfg, N=2, T=12, act=act, ada=ada)
    if want("wrapper_tug"):
        run_stage(gen_tug_wrapper)
    if want("terrain"):
        run_stage(gen_terrain_and_configs)
    if want("terrain_perlin"):
        run_stage(gen_perlin_terrain)
    if want("terrain_perlin_class"):
        run_stage(gen_perlin_class)
    if want("urdf_facts"):
        run_stage(gen_urdf_facts)
    if want("adapter"):
        run_stage(gen_adapter)


if __name__ == "__main__":
    main()
