"""Replay of the `fullstep_*` golden traces (tools/gen_golden.py: the reference's own Go1.step driven by a scripted
simulator) through an engine's unfused entry points.  Used for the CPU oracle (not gpu) and the HIP engine (gpu)."""
import numpy as np
import torch

from mqe.engine import abi
from helpers import golden, make_desc, bag, to_dev, close, curriculum_terrain

TASK_OF = {"gate": "go1gate", "seesaw": "go1seesaw", "football": "go1football-defender", "sheep": "go1sheep-hard",
           "football1v1": "go1football-1vs1", "football2v2": "go1football-2vs2", "pushbox": "go1pushbox", "rotation": "go1revolvingdoor",
           "bridge": "go1bridge", "wrestling": "go1wrestling", "tug": "go1tug", "gate_cmd": "go1gate", "pushbox_curriculum": "go1pushbox"}
# fullstep_pushbox_curriculum: the push-box scene on 3 rows x 2 columns of tracks with terrain.curriculum = True -- the reference's reset_idx moves
# envs between the rows at run time (legged_robot.py:479-503); the trace also holds the spawn poses the first reset() measures
# fullstep_gate_cmd: the gate scene with every implemented command.cfg switch on (go1.py:64-93): 11 action columns per robot
CMD_FLAGS = dict(body_height=True, gait_freq=True, footswing_height=True, body_pose=True, stance_width=True, stance_length=True, aux_reward=True)

# tolerances: MLP outputs go through libm expm1/ELU and a different accumulation order than torch's GEMM
TOL_POLICY = dict(atol=2e-5, rtol=1e-4)
TOL_TORQUE = dict(atol=5e-5, rtol=1e-4)
TOL_EXACT = dict(atol=1e-6, rtol=1e-6)


def replay(name, make_engine):
    z = golden("fullstep_" + name)
    N, A, P = int(z["N"]), int(z["A"]), int(z["P"])
    T = z["actions"].shape[0]
    d, keep, ctx = make_desc(TASK_OF[name], N, levels=z["terrain_levels"], types=z["terrain_types"],
                             max_episode_length=int(z["max_episode_length"]),
                             npc_init=z["base_init_state_npc"][:P] if P else None,
                             noise_mode=1 if name == "sheep" else 0,      # MQE_NOISE_SCRIPTED: the recorded randn sequence is injected
                             command_flags=CMD_FLAGS if name == "gate_cmd" else None,
                             terrain_cfg=curriculum_terrain(TASK_OF[name]) if name.endswith("_curriculum") else None)
    np.testing.assert_allclose(ctx["env_origins"], z["env_origins"], atol=0)
    np.testing.assert_allclose(ctx["agent_origins"], z["agent_origins"], atol=0)
    e = make_engine(d, keep)
    Tn = e.tensor
    R = N * A
    curr = "spawn_all_root" in z.files
    if curr:
        assert d.terrain_curriculum == 1
        np.testing.assert_allclose(np.ctypeslib.as_array(d.terrain_origins, shape=z["terrain_origins"].shape), z["terrain_origins"], atol=0)
        Tn(abi.T_ROOT_STATE).copy_(to_dev(e, z["spawn_all_root"].reshape(N, A + P, 13)))     # the simulator's state before the first reset()
    e.reset_all()
    if curr:
        assert (Tn(abi.T_TERRAIN_LEVELS).cpu().numpy() == z["reset_terrain_levels"]).all(), (Tn(abi.T_TERRAIN_LEVELS), z["reset_terrain_levels"])
        close(Tn(abi.T_ENV_ORIGINS), z["reset_env_origins"], what="env origins after the first reset", **TOL_EXACT)
    close(Tn(abi.T_ROOT_STATE).reshape(-1, 13), z["reset_all_root"], what="reset root", **TOL_EXACT)
    close(Tn(abi.T_DOF_STATE).reshape(-1, 2), z["reset_all_dof"], what="reset dof", **TOL_EXACT)
    for k in ("base_pos", "base_quat", "dof_pos", "dof_vel", "lin_vel", "ang_vel", "last_action", "last_last_action",
              "projected_gravity", "clock_inputs", "base_rpy"):
        close(bag(e, k), z["reset_" + k], what="reset obs " + k, **TOL_EXACT)
    root, dof, cf = Tn(abi.T_ROOT_STATE), Tn(abi.T_DOF_STATE), Tn(abi.T_CONTACT_FORCE)
    for t in range(T):
        cmd = z["actions"][t]
        if name == "football":
            dc = torch.zeros(N, 3, device=e.torch_device)
            e.defender_command(dc)
            cmd = np.concatenate([cmd.reshape(N, A - 1, 3), dc.cpu().numpy().reshape(N, 1, 3)], 1).reshape(-1, 3)
        e.policy_step(to_dev(e, cmd))
        close(Tn(abi.T_LOCOMOTION_OBS)[:, :70], z["locomotion_obs"][t], what=f"t{t} locomotion_obs", **TOL_EXACT)
        close(Tn(abi.T_LAST_LOCO_ACTION), z["loco_action"][t], what=f"t{t} policy output", **TOL_POLICY)
        close(Tn(abi.T_ACTIONS), z["actions_clipped"][t], what=f"t{t} clipped actions", **TOL_POLICY)
        for k in range(4):
            e.compute_torques()
            close(Tn(abi.T_TORQUES), z["torques"][t, k][:, :12 * A], what=f"t{t} substep {k} torques", **TOL_TORQUE)
            dof.copy_(to_dev(e, z["dof_script"][t, k]))          # scripted simulate()
            e.post_decimation_step(k)
        root.copy_(to_dev(e, z["root_script"][t].reshape(N, A + P, 13)))
        cf.copy_(to_dev(e, z["contact_script"][t]))
        if name == "sheep":
            Tn(abi.T_NPC_NOISE).copy_(to_dev(e, z["noise_script"][t]))
        e.post_physics_step()
        hist = e.history()   # recorded by the generator after the full step (resets zero it, go1.py:145)
        close(hist[:, -140:], z["history_tail"][t], what=f"t{t} history tail", **TOL_EXACT)
        close(hist.double().sum(1).float(), z["history_sum"][t], what=f"t{t} history sum", atol=1e-3)
        for key, kind in (("reset_buf", abi.T_RESET_BUF), ("time_out", abi.T_TIME_OUT_BUF), ("r_term", abi.T_R_TERM),
                          ("p_term", abi.T_P_TERM)):
            got = Tn(kind).cpu().numpy().astype(bool)
            assert (got == z[key][t]).all(), (t, key, got, z[key][t])
        if d.terminate_on_base_contact:
            assert (Tn(abi.T_COLLIDE_BUF).cpu().numpy().astype(bool) == z["collide_buf"][t]).all(), (t, "collide")
        assert (Tn(abi.T_EPISODE_LENGTH).cpu().numpy() == z["episode_length"][t]).all(), (t, "episode_length")
        if curr:
            assert (Tn(abi.T_TERRAIN_LEVELS).cpu().numpy() == z["live_terrain_levels"][t]).all(), (t, Tn(abi.T_TERRAIN_LEVELS), z["live_terrain_levels"][t])
            close(Tn(abi.T_ENV_ORIGINS), z["live_env_origins"][t], what=f"t{t} live env origins", **TOL_EXACT)
        close(Tn(abi.T_GAIT_INDICES), z["gait_indices"][t], what=f"t{t} gait", atol=2e-6)
        # sheep script: pow(x, 1.4) / norms differ in the last bits between torch (SLEEF) and libm / the GPU
        tol_root = dict(atol=3e-5, rtol=1e-5) if name == "sheep" else TOL_EXACT
        close(root.reshape(-1, 13), z["post_all_root"][t], what=f"t{t} root after post-step", **tol_root)
        close(dof.reshape(-1, 2), z["post_all_dof"][t], what=f"t{t} dof after post-step", **TOL_EXACT)
        for k in ("base_pos", "base_quat", "dof_pos", "dof_vel", "lin_vel", "ang_vel", "last_action", "last_last_action",
                  "projected_gravity", "base_rpy"):
            tol = TOL_POLICY if "action" in k else dict(atol=3e-6, rtol=1e-5)
            close(bag(e, k), z["obs_" + k][t], what=f"t{t} obs {k}", **tol)
        close(bag(e, "clock_inputs"), z["obs_clock_inputs"][t], what=f"t{t} clock", atol=2e-5)
        if name == "sheep":
            close(Tn(abi.T_SHEEP_POS_AVG), z["sheep_pos_avg"][t], what=f"t{t} sheep avg", atol=1e-5)
            close(Tn(abi.T_SHEEP_POS_VAR), z["sheep_pos_var"][t], what=f"t{t} sheep var", atol=1e-4, rtol=1e-5)
    e.close()
    return True


def replay_joint(ctrl, make_engine):
    """`fullstep_gate_{P,V,T}`: the reference's Go1.step else-branch (go1.py:42-44; PD / velocity / torque laws of
    legged_robot.py:380-392) on the gate scene with a scripted simulator, replayed through the unfused entry points."""
    z = golden("fullstep_gate_" + ctrl)
    N, A, P = int(z["N"]), int(z["A"]), int(z["P"])
    T = z["actions"].shape[0]
    d, keep, ctx = make_desc("go1gate", N, levels=z["terrain_levels"], types=z["terrain_types"], max_episode_length=int(z["max_episode_length"]))
    d.control_type = abi.CTRL[ctrl]
    e = make_engine(d, keep)
    Tn = e.tensor
    e.reset_all()
    close(Tn(abi.T_ROOT_STATE).reshape(-1, 13), z["reset_all_root"], what="reset root", **TOL_EXACT)
    root, dof, cf, act = Tn(abi.T_ROOT_STATE), Tn(abi.T_DOF_STATE), Tn(abi.T_CONTACT_FORCE), Tn(abi.T_ACTIONS)
    clip = d.clip_actions
    for t in range(T):
        a = np.clip(z["actions"][t].reshape(N, A * 12), -clip, clip)               # pre_physics_step (legged_robot.py:108-110)
        act.copy_(to_dev(e, a))
        close(act, z["actions_clipped"][t], what=f"t{t} clipped actions", **TOL_EXACT)
        for k in range(4):
            e.compute_torques()
            close(Tn(abi.T_TORQUES), z["torques"][t, k][:, :12 * A], what=f"t{t} substep {k} torques", atol=2e-5, rtol=1e-5)
            dof.copy_(to_dev(e, z["dof_script"][t, k]))
            e.post_decimation_step(k)
        root.copy_(to_dev(e, z["root_script"][t].reshape(N, A + P, 13)))
        cf.copy_(to_dev(e, z["contact_script"][t]))
        e.post_physics_step()
        for key, kind in (("reset_buf", abi.T_RESET_BUF), ("time_out", abi.T_TIME_OUT_BUF), ("r_term", abi.T_R_TERM), ("p_term", abi.T_P_TERM)):
            assert (Tn(kind).cpu().numpy().astype(bool) == z[key][t]).all(), (t, key)
        assert (Tn(abi.T_EPISODE_LENGTH).cpu().numpy() == z["episode_length"][t]).all(), (t, "episode_length")
        close(root.reshape(-1, 13), z["post_all_root"][t], what=f"t{t} root after post-step", **TOL_EXACT)
        close(dof.reshape(-1, 2), z["post_all_dof"][t], what=f"t{t} dof after post-step", **TOL_EXACT)
        for k in ("base_pos", "base_quat", "dof_pos", "dof_vel", "lin_vel", "ang_vel", "last_action", "last_last_action", "projected_gravity", "base_rpy"):
            if "obs_" + k in z.files:
                close(bag(e, k), z["obs_" + k][t], what=f"t{t} obs {k}", atol=3e-6, rtol=1e-5)
    e.close()
    return True
