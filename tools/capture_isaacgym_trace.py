#!/usr/bin/env python3
"""Capture what row H of SURVEY.md section 8 cannot get inside this repository: trajectories of the REFERENCE's physics.

Runs ONLY on a machine that has NVIDIA Isaac Gym Preview 4 and a checkout of ziyanx02/multiagent-quadruped-environment
(never in the build container, never on the AMD box):

    cd <MQE checkout> && python <this repo>/tools/capture_isaacgym_trace.py --out <this repo>/tests/golden \
           [--tasks go1gate,go1sheep-hard,go1seesaw,go1football-defender] [--num_envs 4] [--steps 40] [--device cuda:0]

For every task it builds the reference environment itself (`make_mqe_env`: the reference's own create_sim / BarrierTrack /
asset options / actor order, mqe/envs/base/legged_robot.py:255-261,754-923), switches every source of randomness off
(init_dof_pos_ratio_range = [1, 1], init_base_vel_range = (0, 0), no NPC jitter), resets, and then drives the simulator through
exactly the calls of the substep body of Go1.step (mqe/envs/go1/go1.py:48-58):

    set_dof_actuation_force_tensor(tau_k) -> simulate -> fetch_results -> refresh_dof_state_tensor
    (+ refresh_actor_root_state_tensor / refresh_net_contact_force_tensor, legged_robot.py:122-124, to read the result)

with SCRIPTED joint torques tau_k (a PD law towards a slowly swaying stance, evaluated by this script from the recorded
state, so the file carries the torques themselves and no controller has to be reproduced).  Written per task:

    tests/golden/isaacgym_<task>.npz
        root   [K+1, N, A+P, 13]   actor root states before substep k (pos, quat xyzw, lin vel, ang vel), agents first
        dof    [K+1, N, D, 2]      joint position / velocity, the robots' 12 A first, NPC dofs after (legged_robot.py:577-585)
        tau    [K,   N, 12 A]      torques applied in substep k
        cf     [K,   N, B, 3]      net contact forces after substep k (refresh_net_contact_force_tensor)
        dof_names, body_names      Isaac Gym's own order (pins the leg order SURVEY 8 could only guess)
        env_origins, agent_origins, sim_dt, meta (json: task, isaacgym / PhysX parameters, seed)

`tests/test_isaacgym_trace.py` (CPU oracle) and `tests/test_gpu_isaacgym_trace.py` (HIP engine) replay any such file that is
present: substep by substep from the recorded state (one-step error) and free-running from the first state (trajectory error),
against the tolerances stated there; they skip when no capture exists.  Until somebody runs this script, row H stays
"parity unpinned" (DESIGN.md section 4).
"""
import argparse
import json
import os
import sys


def scripted_torques(q, qd, q0, k, dt, limits):
    """PD towards the default stance with a slow sway of thigh and calf (0.15 rad at 1 Hz, hips 0.05 rad): keeps the robots on
    their feet, loads every joint, and drives the feet into and out of contact.  Pure function of the recorded state."""
    import torch
    t = k * dt
    n = q.shape[-1] // 12
    sway = torch.zeros(12, device=q.device)
    sway[0::3] = 0.05
    sway[1::3] = 0.15
    sway[2::3] = -0.15
    phase = torch.tensor([0.0, 3.14159265, 3.14159265, 0.0], device=q.device).repeat_interleave(3)       # trot: diagonal legs together
    tgt = (q0 + sway * torch.sin(2 * 3.14159265 * 1.0 * t + phase)).repeat(n)
    tau = 20.0 * (tgt - q) - 0.5 * qd                                                                     # Kp, Kd of go1_config.py:110-111
    lim = limits.repeat(n)
    return torch.maximum(torch.minimum(tau, lim), -lim)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--tasks", default="go1gate,go1sheep-hard,go1seesaw,go1football-defender")
    ap.add_argument("--num_envs", type=int, default=4)
    ap.add_argument("--steps", type=int, default=40, help="policy steps; 4 substeps each are recorded")
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    try:
        import isaacgym  # noqa: F401  (must be imported before torch)
        from isaacgym import gymtorch
    except ImportError:
        sys.exit("capture_isaacgym_trace.py needs NVIDIA Isaac Gym Preview 4 (not available in the build container or on AMD hardware)")
    import numpy as np
    import torch
    sys.path.insert(0, os.getcwd())
    from mqe.envs.utils import make_mqe_env, custom_cfg
    from mqe.utils import get_args

    for task in a.tasks.split(","):
        sys.argv = [sys.argv[0], "--task", task, "--num_envs", str(a.num_envs), "--headless", "--sim_device", a.device, "--seed", str(a.seed)]
        args = get_args()

        def deterministic(cfg, _hook=custom_cfg(args)):
            cfg = _hook(cfg)
            dr = cfg.domain_rand
            dr.init_dof_pos_ratio_range = [1.0, 1.0]
            dr.init_base_vel_range = (0.0, 0.0)
            for name in ("init_base_pos_range", "init_npc_base_pos_range", "init_npc_base_rpy_range"):
                if getattr(dr, name, None) is not None:
                    setattr(dr, name, None)
            for name in ("randomize_friction", "randomize_base_mass", "randomize_com", "push_robots", "randomize_motor", "randomize_lag_timesteps"):
                if hasattr(dr, name):
                    setattr(dr, name, False)
            if hasattr(cfg.asset, "sheep_movement_randomness"):
                cfg.asset.sheep_movement_randomness = 0.0
            return cfg
        wrapped, cfg = make_mqe_env(task, args, deterministic)
        env = wrapped.env
        gym, sim = env.gym, env.sim
        wrapped.reset()
        N, A = env.num_envs, env.num_agents
        K = a.steps * env.decimation
        q0 = env.default_dof_pos.reshape(-1)[:12].clone()
        limits = env.torque_limits.reshape(-1)[:12].clone()
        root, dof, tau_log, cf_log = [], [], [], []

        def snapshot():
            gym.refresh_actor_root_state_tensor(sim)
            gym.refresh_dof_state_tensor(sim)
            root.append(env.all_root_states.view(N, -1, 13).detach().cpu().numpy().copy())
            dof.append(env.all_dof_states.view(N, -1, 2).detach().cpu().numpy().copy())
        snapshot()
        for k in range(K):
            st = env.all_dof_states.view(N, -1, 2)[:, :12 * A]
            tau = scripted_torques(st[..., 0], st[..., 1], q0, k, env.sim_params.dt, limits).contiguous()
            full = torch.cat((tau, torch.zeros((N, env.num_actions_npc), dtype=tau.dtype, device=tau.device)), dim=1) if env.num_actions_npc else tau
            gym.set_dof_actuation_force_tensor(sim, gymtorch.unwrap_tensor(full.contiguous()))       # go1.py:52
            gym.simulate(sim)                                                                       # go1.py:53
            gym.fetch_results(sim, True)                                                            # go1.py:54-55
            gym.refresh_net_contact_force_tensor(sim)                                               # legged_robot.py:124
            tau_log.append(tau.detach().cpu().numpy().copy())
            cf_log.append(env.contact_forces.view(N, -1, 3).detach().cpu().numpy().copy() if hasattr(env, "contact_forces") else np.zeros((N, 0, 3), np.float32))
            snapshot()
        h0 = env.actor_handles[0] if hasattr(env, "actor_handles") else 0
        px = env.sim_params.physx
        meta = dict(task=task, seed=a.seed, num_envs=N, num_agents=A, num_npcs=env.num_npcs, decimation=env.decimation, sim_dt=env.sim_params.dt,
                    physx=dict(solver_type=px.solver_type, num_position_iterations=px.num_position_iterations, num_velocity_iterations=px.num_velocity_iterations,
                               contact_offset=px.contact_offset, rest_offset=px.rest_offset, bounce_threshold_velocity=px.bounce_threshold_velocity,
                               max_depenetration_velocity=px.max_depenetration_velocity),
                    script="PD to the default stance + 1 Hz trot sway, tools/capture_isaacgym_trace.py::scripted_torques")
        os.makedirs(a.out, exist_ok=True)
        np.savez_compressed(os.path.join(a.out, f"isaacgym_{task}.npz"), root=np.stack(root), dof=np.stack(dof), tau=np.stack(tau_log), cf=np.stack(cf_log),
                            dof_names=np.array(gym.get_actor_dof_names(env.envs[0], h0)), body_names=np.array(gym.get_actor_rigid_body_names(env.envs[0], h0)),
                            env_origins=env.env_origins.detach().cpu().numpy(), agent_origins=env.agent_origins.detach().cpu().numpy(),
                            sim_dt=np.float32(env.sim_params.dt), meta=np.array(json.dumps(meta)))
        print("wrote", os.path.join(a.out, f"isaacgym_{task}.npz"), "substeps:", K)
        env.gym.destroy_sim(sim)


if __name__ == "__main__":
    main()
