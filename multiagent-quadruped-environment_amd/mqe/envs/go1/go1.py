"""Go1: the vectorised multi-agent Go1 environment, host side.

Mirrors the surface of the reference class stack Go1 -> LeggedRobotField -> LeggedRobot -> BaseTask
(mqe/envs/go1/go1.py:19-62,147-151; mqe/envs/base/legged_robot.py:54-157,549-645; mqe/envs/base/base_task.py:40-105)
but owns no simulation logic: every per-step computation runs in the HIP engine behind the C ABI
(include/mqe_hip.h).  This class (1) builds the scene description once -- BarrierTrack terrain, env/agent origins,
robot model, MLP weights -- and (2) exposes the long-lived state tensors the reference's wrappers and users reach
for (`root_states`, `dof_pos`, `obs_buf.base_pos`, `collide_buf`, ...) as zero-copy views of engine memory.
"""
import types

import numpy as np
import torch

from mqe.engine import abi
from mqe.engine.desc import build_desc, task_kind, REWARD_TERMS
from mqe.utils.helpers import class_to_dict, engine_seed
from mqe.utils.terrain import get_terrain_cls


class ObsBag:
    """Attribute bag returned by Go1.step()/reset() (reference: obs_buf = copy(cfg.obs), go1.py:26,153-196).
    Every field is a live view into the engine's observation buffer."""

    def __init__(self, cfg_obs, bag, env_info):
        self.cfgs = cfg_obs.cfgs
        self.scales = getattr(cfg_obs, "scales", None)
        for name, (a, b) in abi.BAG.items():
            setattr(self, name, bag[:, a:b])
        if env_info is not None and getattr(cfg_obs.cfgs, "env_info", False):
            self.env_info = env_info

    def keys(self):
        return [k for k in dir(self.cfgs) if getattr(self.cfgs, k) is True]


def _default_engine_factory(desc, keep, device):
    from mqe.engine.hip_engine import HipEngine
    return HipEngine(desc, keep, device=device)


class Go1:
    # engine_factory(desc, keepalive, device) -> engine; the default (and only product) engine is the HIP one.
    engine_factory = staticmethod(_default_engine_factory)
    # sharding of a global batch across processes / GPUs: (global_num_envs, first_global_env) or None
    shard = None

    def __init__(self, cfg, sim_params, physics_engine, sim_device, headless):
        self.cfg = cfg
        self.env_name = getattr(cfg.env, "env_name", "go1")
        self.sim_params = sim_params
        self.physics_engine = physics_engine
        self.sim_device = sim_device
        self.headless = headless
        self.device = sim_device if getattr(sim_params, "use_gpu_pipeline", True) else "cpu"
        self.num_envs = cfg.env.num_envs
        self.num_agents = getattr(cfg.env, "num_agents", 1)
        self.num_npcs = getattr(cfg.env, "num_npcs", 0)
        self.num_obs = cfg.env.num_observations
        self.num_privileged_obs = cfg.env.num_privileged_obs
        self.num_action = cfg.env.num_actions
        self.num_actions = self.num_agents * cfg.env.num_actions
        self.num_actions_npc = getattr(cfg.env, "num_actions_npc", 0) * self.num_npcs
        self.num_dof = 12
        self.num_actuated_dof = 12 * self.num_agents
        self.num_bodies = abi.NREP
        self.decimation = cfg.control.decimation
        self._parse_cfg()
        self._create_scene()
        self._init_buffers()
        self.init_done = True

    # ---- configuration (reference legged_robot.py:1013-1024) -------------------------------------------------
    def _parse_cfg(self):
        cfg = self.cfg
        self.dt = cfg.control.decimation * self.sim_params.dt
        self.obs_scales = cfg.normalization.obs_scales
        self.reward_scales = class_to_dict(cfg.rewards.scales)
        self.max_episode_length_s = cfg.env.episode_length_s
        self.max_episode_length = np.ceil(self.max_episode_length_s / self.dt)
        cfg.env.max_episode_length = self.max_episode_length
        if getattr(cfg.terrain, "curriculum", False) and cfg.terrain.num_rows > 1 and self.shard is not None and self.shard[0] != self.num_envs:
            # reset_idx -> _update_terrain_curriculum (legged_robot.py:479-503) measures the walked distance on root_states[env_ids]: ROW e
            # of the agents' tensor is robot e % A of env e // A, another shard's env for almost every e
            raise NotImplementedError("cfg.terrain.curriculum = True with num_rows > 1 on an env-sharded batch: upstream's run-time terrain "
                                      "curriculum indexes the agents' root states with env ids across the whole batch; run it on one GPU")

    # ---- scene construction (reference create_sim, legged_robot.py:255-261,754-923,972-997) ---------------------
    def _create_scene(self):
        cfg = self.cfg
        N, A = self.num_envs, self.num_agents
        gN, g0 = self.shard if self.shard is not None else (N, 0)
        terrain_cls = get_terrain_cls(cfg.terrain.selected)
        self.terrain = terrain_cls(cfg.terrain, N, A).build()
        self.custom_origins = True
        t = self.terrain
        max_init_level = cfg.terrain.max_init_terrain_level if cfg.terrain.curriculum else cfg.terrain.num_rows - 1
        # levels/types are functions of the GLOBAL env index so that a sharded run sees the same scene
        levels = torch.randint(0, max_init_level + 1, (gN,))[g0:g0 + N]
        types_ = (torch.arange(gN) % cfg.terrain.num_cols)[g0:g0 + N]
        self.terrain_levels, self.terrain_types = levels, types_
        self.max_terrain_level = cfg.terrain.num_rows
        eo = torch.from_numpy(t.env_origins).float()[levels, types_]
        ao = torch.from_numpy(t.agent_origins).float()[levels, types_]
        info = {k: torch.from_numpy(v).float()[levels, types_] for k, v in (t.env_info or {}).items()}
        self._env_origins_np = eo.numpy().copy()
        self._agent_origins_np = ao.numpy().copy()
        self._env_info_np = {k: v.numpy().copy() for k, v in info.items()}
        self.task = task_kind(cfg)
        gate_pos = self._task_gate_pos()
        desc, keep = build_desc(cfg, N, t, self._env_origins_np, self._agent_origins_np, gate_pos=gate_pos,
                                env_id_offset=g0, seed=engine_seed(), task=self.task, terrain_levels=levels.numpy(), terrain_types=types_.numpy())
        self.body_is_synthetic = dict(k for k in keep if isinstance(k, tuple)).get("body_is_synthetic", True)
        self.engine = type(self).engine_factory(desc, keep, self.device)
        self.device = str(self.engine.torch_device) if hasattr(self.engine, "torch_device") else self.device

    def _task_gate_pos(self):
        """(N,2) per-env gate position the task wrapper / scripted defender uses, or None."""
        kw = self.cfg.terrain.BarrierTrack_kwargs
        gd = self._env_info_np.get("gate_deviation")
        if gd is None and self.task in ("gate", "pushbox", "sheep"):
            raise ValueError(f"task '{self.task}' reads the gate position from the terrain's env_info (a BarrierTrack with a gate block); "
                             f"{type(self.terrain).__name__} has none")
        if self.task in ("gate", "pushbox"):      # go1_gate_wrapper.py:41-42, go1_pushbox_wrapper.py:29-30
            g = gd.copy()
            g[:, 0] += kw["init"]["block_length"] + kw["gate"]["block_length"] / 2
            return g
        if self.task == "sheep":     # go1_sheep_wrapper.py:31-33
            g = gd.copy()
            g[:, 0] += kw["init"]["block_length"] + kw["plane"]["block_length"] + kw["gate"]["block_length"] / 2
            return g
        if self.task == "football_defender":   # go1_football_defender.py:61-63 (absolute frame)
            g = self._env_origins_np[:, :2].copy()
            g[:, 0] += kw["init"]["block_length"] + kw["plane"]["block_length"]
            return g
        return None

    # ---- buffers: zero-copy views (reference _init_buffers, legged_robot.py:549-645; go1.py:357-387) ------------
    def _init_buffers(self):
        e, N, A, P = self.engine, self.num_envs, self.num_agents, self.num_npcs
        T = e.tensor
        dev = e.torch_device
        self.all_root_states = T(abi.T_ROOT_STATE).view(N * (A + P), 13)
        self._root3 = T(abi.T_ROOT_STATE)
        self.all_dof_states = T(abi.T_DOF_STATE).view(-1, 2)
        self.dof_state = T(abi.T_DOF_STATE)[:, :12 * A, :]
        self.dof_pos = self.dof_state[:, :, 0]
        self.dof_vel = self.dof_state[:, :, 1]
        if self.num_actions_npc > 0:
            self.dof_state_npc = T(abi.T_DOF_STATE)[:, 12 * A:, :]
            self.dof_pos_npc = self.dof_state_npc[:, :, 0]
            self.dof_vel_npc = self.dof_state_npc[:, :, 1]
        self.contact_forces = T(abi.T_CONTACT_FORCE)
        self.torques = T(abi.T_TORQUES)
        self.actions = T(abi.T_ACTIONS)
        self.last_actions = T(abi.T_LAST_ACTIONS)
        self.locomotion_obs = T(abi.T_LOCOMOTION_OBS)[:, :70]
        self.last_locomotion_action = T(abi.T_LAST_LOCO_ACTION)
        self.last_two_locomotion_action = T(abi.T_LAST_TWO_LOCO_ACTION)
        ah = T(abi.T_ACT_HIST)
        self.joint_pos_err_last, self.joint_pos_err_last_last, self.joint_vel_last, self.joint_vel_last_last = ah[0], ah[1], ah[2], ah[3]
        self.gait_indices = T(abi.T_GAIT_INDICES)
        self.clock_inputs = T(abi.T_CLOCK_INPUTS)
        self.base_lin_vel = T(abi.T_BASE_LIN_VEL)
        self.base_ang_vel = T(abi.T_BASE_ANG_VEL)
        self.projected_gravity = T(abi.T_PROJECTED_GRAVITY)
        self.base_quat = T(abi.T_BASE_QUAT)
        self.episode_length_buf = T(abi.T_EPISODE_LENGTH)
        self.reset_buf = T(abi.T_RESET_BUF).view(torch.bool)
        self.collide_buf = T(abi.T_COLLIDE_BUF).view(torch.bool)
        self.time_out_buf = T(abi.T_TIME_OUT_BUF).view(torch.bool)
        self.r_term_buff = T(abi.T_R_TERM).view(torch.bool)
        self.p_term_buff = T(abi.T_P_TERM).view(torch.bool)
        self.z_high_term_buff = T(abi.T_Z_HIGH_TERM).view(torch.bool)
        self.substep_torques = T(abi.T_SUBSTEP_TORQUES)
        self.sheep_pos_avg = T(abi.T_SHEEP_POS_AVG)
        self.sheep_pos_var = T(abi.T_SHEEP_POS_VAR)
        self.npc_noise = T(abi.T_NPC_NOISE)
        # diagnostic, not in the reference: per env the number of substeps whose bounded contact list was truncated (cumulative
        # over the handle's life).  A device tensor -- reading it is the caller's sync, the step never does.
        self.contact_overflow = T(abi.T_CONTACT_OVERFLOW)
        self.rew_buf = torch.zeros(N * A, device=dev)                    # Go1 registers no reward functions (go1.py:198-219)
        # env_origins is the engine's LIVE tensor (the run-time terrain curriculum rewrites rows of it, legged_robot.py:495); the copies
        # upstream makes at construction -- env_origins_repeat (:992), the sheep task's npc_env_origins (go1_sheep.py:120) -- stay copies
        self.env_origins = T(abi.T_ENV_ORIGINS)
        self.env_origins_repeat = self.env_origins.clone().unsqueeze(1).repeat(1, A, 1).reshape(-1, 3)
        self.agent_origins = torch.from_numpy(self._agent_origins_np).to(dev)
        self.npc_env_origins = self.env_origins.clone().unsqueeze(1).repeat(1, max(P, 1), 1)[:, :P]
        self.terrain_levels = T(abi.T_TERRAIN_LEVELS) if self.engine.desc.terrain_curriculum else self.terrain_levels.to(dev)
        self.terrain_types = self.terrain_types.to(dev)
        self.terrain_origins = torch.from_numpy(np.asarray(self.terrain.env_origins, np.float32)).to(dev)
        if self.engine.desc.terrain_curriculum:
            # what the first reset()'s curriculum step measures (init_done is set before it): the actors' spawn poses, env origin +
            # U(+-x_init_range, +-y_init_range) per robot (legged_robot.py:864-869: torch_rand_float on the global generator)
            xr, yr = float(getattr(self.cfg.terrain, "x_init_range", 0.0)), float(getattr(self.cfg.terrain, "y_init_range", 0.0))
            spawn = self.env_origins.clone().unsqueeze(1).repeat(1, A, 1)
            spawn[..., 0] += (torch.rand(N, A, device=dev) * 2 - 1) * xr
            spawn[..., 1] += (torch.rand(N, A, device=dev) * 2 - 1) * yr
            self._root3[:, :A, :3] = spawn
        self.env_info = {k: torch.from_numpy(v).to(dev) for k, v in self._env_info_np.items()}
        self.default_dof_pos = torch.tensor([self.engine.desc.default_dof_pos[j] for j in range(12)] * A, device=dev).unsqueeze(0)
        self.torque_limits = torch.tensor([self.engine.desc.torque_limits[j] for j in range(12)] * A, device=dev)
        self.base_init_state = torch.from_numpy(np.ctypeslib.as_array(self.engine.desc.base_init_state, shape=(A, 13)).copy()).to(dev).repeat(N, 1)
        self.env_agent_indices = torch.arange(N * A, device=dev).reshape(N, A)
        self.env_npc_indices = torch.arange(N * P, device=dev).reshape(N, P)
        self.actor_indices = torch.arange(N * (A + P), dtype=torch.int32, device=dev).reshape(N, A + P)
        self.agent_indices = self.actor_indices[:, :A]
        self.npc_indices = self.actor_indices[:, A:]
        self.obs_buf = ObsBag(self.cfg.obs, T(abi.T_OBS_BAG), self.env_info if self.env_info else None)
        self.privileged_obs_buf = None
        # extras["episode"]: a plain dict.  With the run-time terrain curriculum on, "terrain_level" is a REAL 0-dim device tensor (upstream
        # stores torch.mean(terrain_levels.float()), _fill_extras, legged_robot.py:1069-1071): a FRESH tensor after every reset / step, as
        # upstream assigns one per reset (refreshed in place, a logger that kept the object saw its past entries change), so
        # .items(), .get(), dict(...) and ** all see the current value -- two tiny launches, on that path only
        self.extras = {"time_outs": self.time_out_buf, "episode": {}, "contact_overflow": self.contact_overflow}
        if self.engine.desc.terrain_curriculum:
            self.extras["episode"]["terrain_level"] = torch.zeros((), device=dev)
            self._refresh_extras()
        self.common_step_counter = 0
        if self.task == "football_defender":
            self.gate_pos = torch.zeros(N, 3, device=dev)
            self.gate_pos[:, :2] = torch.from_numpy(self._task_gate_pos()).to(dev)
            self.gate_pos[:, 2] = self.env_origins[:, 2]

    def _refresh_extras(self):
        ep = self.extras["episode"]
        if "terrain_level" in ep:
            ep["terrain_level"] = torch.mean(self.terrain_levels.float())

    # ---- derived views ------------------------------------------------------------------------------------------
    @property
    def root_states(self):
        """(N*A, 13): agents' rows of the actor root-state tensor (legged_robot.py:130)."""
        r = self._root3[:, :self.num_agents, :]
        return r.reshape(-1, 13)

    @property
    def root_states_npc(self):
        return self._root3[:, self.num_agents:, :].reshape(-1, 13)

    @property
    def base_pos(self):
        return self.root_states[:, 0:3]

    @property
    def reset_ids(self):
        """Indices of the envs reset by the last step (legged_robot.py:145-147).  Materialising them is a
        device->host sync; the fused path never needs it."""
        return self.reset_buf.nonzero(as_tuple=False).flatten()

    @property
    def history_locomotion_obs(self):
        """(R, 2100) time-ordered history (go1.py:102,395) gathered from the engine's ring buffer."""
        h = self.engine.tensor(abi.T_HISTORY)
        pos = self._hist_pos()
        idx = (torch.arange(abi.HIST, device=h.device) + pos) % abi.HIST
        return h[:, idx, :70].reshape(h.shape[0], -1)

    def _hist_pos(self):
        return getattr(self, "_steps_policy", 0) % abi.HIST

    @property
    def obs_history_ring(self):
        """The engine's own history tensor (R, 30, 72), ring order (slot `_hist_pos()` holds the oldest frame), zero copy.  It may be
        WRITTEN -- the reference's obs_history is a plain tensor (go1.py:102,145) -- followed by history_written()."""
        return self.engine.tensor(abi.T_HISTORY)

    def history_written(self):
        """after a write into obs_history_ring: what the engine derives from the ring (the compact operand of the policy's first layer)
        is rebuilt from it (mqe_history_sync)"""
        self.engine.history_sync()

    # ---- onboard sensors (legged_robot_field.py:23-93,196-223) ------------------------------------------------------------------
    def forward_depth(self, far=20.0):
        """(num_envs, num_agents, H, W) forward depth images from the CURRENT state, Isaac Gym's IMAGE_DEPTH convention (negative depth
        along the optical axis, -inf = nothing within `far` metres); camera = cfg.sensor.forward_camera (resolution, position, ZYX rotation
        on the base link, horizontal_fov in degrees if present, else Isaac Gym's default 90).  A ray caster over the collision geometry
        (mqe_render_depth), not the reference's rasteriser."""
        cam = self.cfg.sensor.forward_camera
        H, W = int(cam.resolution[0]), int(cam.resolution[1])
        fov = getattr(cam, "horizontal_fov", 90.0)
        if isinstance(fov, (tuple, list)):
            fov = 0.5 * (float(fov[0]) + float(fov[1]))          # upstream draws one value per camera from the range; the engine's cameras share the middle
        pos, rot = cam.position, cam.rotation
        if isinstance(pos, dict):
            pos = pos["mean"]
        if isinstance(rot, dict):
            rot = [0.5 * (lo + hi) for lo, hi in zip(rot["lower"], rot["upper"])]
        img = self.engine.render_depth(H, W, float(fov), pos, rot, far)
        return img.view(self.num_envs, self.num_agents, H, W)

    @property
    def sensor_tensor_dict(self):
        """as upstream's (legged_robot_field.py:196-223): {"forward_depth": [one (num_agents, H, W) tensor per env]} when
        cfg.obs.cfgs.depth_image is set, empty otherwise; rendered when read (upstream refreshes its image tensors inside
        LeggedRobotField.compute_observations, which Go1 overrides without calling it: go1.py:153)."""
        from collections import defaultdict
        out = defaultdict(list)
        if getattr(self.cfg.obs.cfgs, "depth_image", False):
            out["forward_depth"] = list(self.forward_depth().unbind(0))
        if getattr(self.cfg.obs.cfgs, "rgb_image", False):
            raise NotImplementedError("colour images need a rasteriser; only the depth camera exists (mqe_render_depth)")
        return out

    # ---- reference API --------------------------------------------------------------------------------------------
    def reset(self):
        """Reset all robots (go1.py:147-151): no physics step, observations recomputed."""
        self.engine.reset_all()
        self._refresh_extras()
        return self.obs_buf

    # ---- plugin points of the reference's class stack ----------------------------------------------------------------------------
    # The reference lets a subclass of LeggedRobot / Go1 replace pieces of the step: `_compute_torques` (legged_robot.py:368-392),
    # `_post_physics_step_callback` (:153-157), `check_termination` (:159-169), `compute_reward` (:202-219), `_step_npc` (:146, the NPC
    # tasks' scripts), `reset_idx` (:171-200) and `compute_observations` (go1.py:153-196).  Here those pieces run inside the engine; a
    # subclass that OVERRIDES any of the methods below is honoured on the Go1-level path: `step()` then runs the decimation loop unfused
    # and the post-physics step in the reference's stages (mqe_post_physics_stage), calling each override where the reference calls it.
    # The engine's own piece has already run when an override is entered (so `super().check_termination()` etc. are no-ops that keep
    # the reference's call pattern working), except `_step_npc`: an override REPLACES the engine's script, as a subclass's does upstream.
    # The fused wrapper-level step (`step_fused`, what the task wrappers call) refuses to run with overrides in place.
    def _compute_torques(self, actions):
        """(N, 12 A) joint-space actions -> (N, 12 A) torques.  Default: the engine's law for cfg.control.control_type."""
        return None

    def compute_reward(self):
        """fill self.rew_buf (legged_robot.py:202-219); Go1 registers no reward functions (go1.py:198-219): zeros"""

    def _post_physics_step_callback(self):
        """after the frame quantities, before check_termination (legged_robot.py:141, :153-157; go1.py:221-238: the gait clock, in the engine)"""

    def check_termination(self):
        """legged_robot.py:159-169 + legged_robot_field.py:121-146, evaluated by the engine before an override is called: reset_buf,
        time_out_buf, collide_buf, r/p/z term buffers hold this step's flags; an override edits them in place (`self.reset_buf |= ...`)"""

    def _step_npc(self):
        """the task's NPC script (legged_robot.py:146; go1_sheep.py:35-64 in the engine).  An override replaces it: it moves
        `root_states_npc` (a live view of the engine's root-state tensor)"""

    def reset_idx(self, env_ids):
        """legged_robot.py:171-200 / go1.py:110-145, done by the engine (in-kernel) for the envs whose reset_buf is set when the stage
        runs; an override is called afterwards with those env ids, like a subclass that calls super().reset_idx(env_ids) first"""

    def compute_observations(self):
        """go1.py:153-196: the engine fills obs_buf; an override runs afterwards and may add to / edit it"""

    _PLUGIN_POINTS = ("_compute_torques", "compute_reward", "_post_physics_step_callback", "check_termination", "_step_npc", "reset_idx",
                      "compute_observations")

    def _overridden(self, name):
        return getattr(type(self), name) is not getattr(Go1, name)

    @property
    def has_overrides(self):
        return any(self._overridden(n) for n in self._PLUGIN_POINTS)

    def _decimation_loop(self):
        e = self.engine
        custom_tau = self._overridden("_compute_torques")
        for dec_i in range(self.decimation):
            if custom_tau:
                self.torques.copy_(self._compute_torques(self.actions).reshape(self.torques.shape))
            else:
                e.compute_torques()
            e.simulate()
            e.post_decimation_step(dec_i)
        # post_physics_step in the reference's order (legged_robot.py:117-157): frame quantities -> callback -> check_termination ->
        # compute_reward -> _step_npc -> reset_idx -> compute_observations
        e.post_physics_stage(abi.POST_FRAME)
        self.common_step_counter += 1
        if self._overridden("_post_physics_step_callback"):
            self._post_physics_step_callback()
        if self._overridden("check_termination"):
            self.check_termination()
        if self._overridden("compute_reward"):
            self.compute_reward()
        if self._overridden("_step_npc"):
            self._step_npc()
        else:
            e.post_physics_stage(abi.POST_NPC)
        e.post_physics_stage(abi.POST_RESET)
        if self._overridden("reset_idx"):
            self.reset_idx(self.reset_buf.nonzero(as_tuple=False).flatten())
        e.post_physics_stage(abi.POST_OBS)
        if self._overridden("compute_observations"):
            self.compute_observations()
        e.post_physics_stage(abi.POST_WRAPPER)

    def step(self, action):
        """One policy step from already-scaled commands (go1.py:35-62); action: (N*A, 3) or (N, A, 3)."""
        if self.cfg.control.control_type != "C":
            # low-level control (go1.py:42-44): joint-space actions (N*A, 12) / (N, A*12), clipped to clip_actions inside
            # the engine (legged_robot.py:108-110); PD / torque law, 4 substeps and the post-physics step are one fused call
            a = action.reshape(-1, 12).to(self.engine.torch_device, torch.float32).contiguous()
            if self.has_overrides:            # pre_physics_step (legged_robot.py:108-110), then the loop with the subclass's pieces
                c = self.cfg.normalization.clip_actions
                self.actions.copy_(torch.clip(a, -c, c).reshape(self.actions.shape))
                self._decimation_loop()
                self._refresh_extras()
                return self.obs_buf, self.rew_buf, self.reset_buf, self.extras
            self.engine.step_joint(a)
            self.common_step_counter += 1
            self._refresh_extras()
            return self.obs_buf, self.rew_buf, self.reset_buf, self.extras
        cmd = action.reshape(-1, self.engine.desc.num_command_dims).to(self.engine.torch_device, torch.float32).contiguous()   # 3 unless command.cfg says otherwise (go1.py:64-93)
        e = self.engine
        if self.task == "football_defender" and cmd.shape[0] == self.num_envs * 2:
            # Go1FootballDefender.step (go1_football_defender.py:25-31): the scripted defender's command is appended to the two
            # learners' commands (mqe_defender_command evaluates _get_defender_action, :56-80, from the current state)
            if e.desc.num_command_dims != 3:
                raise NotImplementedError("the scripted defender issues (x, y, yaw) commands (go1_football_defender.py:56-80): command.cfg columns "
                                          "beyond the velocity command are not defined for this task")
            dc = torch.empty(self.num_envs, 3, device=cmd.device)
            e.defender_command(dc)
            cmd = torch.cat([cmd.view(self.num_envs, 2, 3), dc.unsqueeze(1)], dim=1).reshape(-1, 3).contiguous()
        if self.has_overrides:           # the subclass's pieces run where the reference calls them: stage by stage
            e.policy_step(cmd)
            self._steps_policy = getattr(self, "_steps_policy", 0) + 1
            self._decimation_loop()
        else:                            # the same step as the engine's five fused launches (mqe_step_command)
            e.step_command(cmd)
            self._steps_policy = getattr(self, "_steps_policy", 0) + 1
            self.common_step_counter += 1
        self._refresh_extras()
        return self.obs_buf, self.rew_buf, self.reset_buf, self.extras

    def step_fused(self, actions):
        """Wrapper-level step: raw actions (N, A', 3) in [-1,1]; clip, task action scale, policy, 4 substeps,
        post-step, task observation and reward all inside the engine (mqe_step)."""
        if self.has_overrides:
            raise NotImplementedError("the fused wrapper-level step runs entirely inside the engine: a Go1 subclass that overrides any of "
                                      + " / ".join(self._PLUGIN_POINTS) + " is stepped through Go1.step()")
        if self.engine.desc.num_command_dims != 3:
            raise NotImplementedError("wrapper-level steps carry (N, A', 3) velocity commands; a config whose command.cfg adds action "
                                      "columns (go1.py:64-93) is stepped through Go1.step()")
        a = actions.to(self.engine.torch_device, torch.float32).contiguous()
        hooks = (getattr(self, "between_policy_and_physics", None), getattr(self, "before_policy_tail", None))   # the env-sharded runner's (bench.py)
        if hooks[1] is not None:
            self.engine.step(a, hooks[0], hooks[1])
        else:
            self.engine.step(a, hooks[0])
        self._steps_policy = getattr(self, "_steps_policy", 0) + 1
        self.common_step_counter += 1
        self._refresh_extras()

    def get_state(self):
        """Checkpoint of the whole simulation state: the engine's blob (every state tensor, history ring and its position, lag ring, wrapper
        bookkeeping, RNG reset counters; mqe_state_save) + the host-side step counters.  The reference has no save / restore of the
        simulation; a rollout continued after set_state() is bit for bit the uninterrupted one."""
        return {"engine": self.engine.save_state(), "steps_policy": getattr(self, "_steps_policy", 0), "common_step_counter": self.common_step_counter}

    def set_state(self, state):
        self.engine.load_state(state["engine"])
        self._steps_policy = int(state["steps_policy"])
        self.common_step_counter = int(state["common_step_counter"])

    def get_observations(self):
        return self.obs_buf

    def get_privileged_observations(self):
        return None

    def render(self, *a, **k):
        return None

    def close(self):
        self.engine.close()

    # indexed setters of the Isaac Gym tensor API are no-ops here: state tensors are live engine memory
    def set_dof_state_tensor_indexed(self, *a):
        return True

    def set_actor_root_state_tensor_indexed(self, *a):
        return True


__all__ = ["Go1", "ObsBag", "REWARD_TERMS"]
