"""Task wrappers (rows W2-W4): feed the synthetic env attributes of tests/golden/wrapper_*.npz (produced by running
the reference's wrapper classes) into an engine and compare observation, reward and the reward log."""
import numpy as np
import torch

from mqe.engine import abi
from mqe.engine.desc import REWARD_TERMS
from helpers import golden, make_desc, to_dev, close

TASK_OF = {"gate": "go1gate", "sheep_hard": "go1sheep-hard", "sheep_easy": "go1sheep-easy", "seesaw": "go1seesaw", "football_defender": "go1football-defender",
           "pushbox": "go1pushbox", "rotation": "go1revolvingdoor",
           "bridge": "go1bridge", "wrestling": "go1wrestling", "tug": "go1tug"}


def _rotation_replay(z, d, keep, make_engine, name="rotation"):
    """go1revolvingdoor / go1bridge / go1wrestling: the fixtures script only base pos / rpy (the wrappers read nothing else;
    wrestling's roll / pitch come from base_quat, of which base_rpy is the Euler form)."""
    T, N = z["obs"].shape[0], z["obs"].shape[1]
    A = d.num_agents
    e = make_engine(d, keep)
    Tn = e.tensor
    bagt = Tn(abi.T_OBS_BAG)

    def load(t):
        bagt[:, 0:3] = to_dev(e, z["base_pos"][t])
        bagt[:, 3:6] = to_dev(e, z["base_rpy"][t])
        Tn(abi.T_RESET_BUF).copy_(to_dev(e, z["reset_buf"][t].astype(np.uint8), torch.uint8))
    load(0)
    e.wrapper_eval(1)
    close(Tn(abi.T_WRAPPER_OBS), z["obs_reset"], what="reset obs", atol=1e-6)
    for t in range(T):
        load(t + 1)
        e.wrapper_eval(0)
        close(Tn(abi.T_WRAPPER_OBS), z["obs"][t], what=f"t{t} obs", atol=1e-6)
        close(Tn(abi.T_WRAPPER_REWARD), z["reward"][t].reshape(N, A), what=f"t{t} reward", atol=1e-6)
    sums = Tn(abi.T_REWARD_SUMS).double().sum(0).cpu().numpy()
    want = dict(zip([str(k) for k in z["reward_buffer_keys"]], z["reward_buffer_vals"]))
    for i, (_, n) in enumerate(REWARD_TERMS[name]):
        assert abs(sums[i] - want[n]) <= 1e-4, (n, sums[i], want[n])
    # the action the reference wrapper handed down: agent 1 mirrored in y / yaw, clipped, scaled
    act = torch.from_numpy(z["actions"][0].copy())
    act[:, 1, 1:] = -act[:, 1, 1:]
    ref = (act.clip(-1, 1) * torch.tensor([2.0, 0.5, 0.5])).reshape(-1, 3)
    assert torch.allclose(ref, torch.from_numpy(z["env_action"][0]), atol=1e-6)
    e.close()
    return True


def _tug_replay(z, d, keep, make_engine):
    """go1tug (upstream runs with num_envs = 1 only): base pos / rpy, the slider's dof state and the env resets are scripted."""
    T, N = z["obs"].shape[0], z["obs"].shape[1]
    A = d.num_agents
    e = make_engine(d, keep)
    Tn = e.tensor
    bagt, dof = Tn(abi.T_OBS_BAG), Tn(abi.T_DOF_STATE)

    def load(t):
        bagt[:, 0:3] = to_dev(e, z["base_pos"][t])
        bagt[:, 3:6] = to_dev(e, z["base_rpy"][t])
        dof[:, 12 * A, :] = to_dev(e, z["dof_state_npc"][t].reshape(N, 2))
        Tn(abi.T_RESET_BUF).copy_(to_dev(e, z["reset_buf"][t].astype(np.uint8), torch.uint8))
    load(0)
    e.wrapper_eval(1)
    close(Tn(abi.T_WRAPPER_OBS), z["obs_reset"], what="reset obs", atol=1e-6)
    for t in range(T):
        load(t + 1)
        e.wrapper_eval(0)
        close(Tn(abi.T_WRAPPER_OBS), z["obs"][t], what=f"t{t} obs", atol=1e-6)
        close(Tn(abi.T_WRAPPER_REWARD), z["reward"][t].reshape(N, A), what=f"t{t} reward", atol=2e-5, rtol=1e-5)
        # reset_dic (go1_tug_wrapper.py:61-71): the wrapper zeroes the slider at the start of the NEXT step while the counter,
        # decremented, is still positive, i.e. iff it reads >= 2 now; the engine applies that right after this observation
        now = dof[:, 12 * A, :].cpu().numpy()
        for n in range(N):
            if z["reset_dic"][t][n] >= 2:
                assert (now[n] == 0).all(), (t, n, now[n])
            else:
                np.testing.assert_allclose(now[n], z["dof_state_npc"][t + 1].reshape(N, 2)[n], atol=0)
    sums = Tn(abi.T_REWARD_SUMS).double().sum(0).cpu().numpy()
    want = dict(zip([str(k) for k in z["reward_buffer_keys"]], z["reward_buffer_vals"]))
    for i, (_, n) in enumerate(REWARD_TERMS["tug"]):
        assert abs(sums[i] - want[n]) <= 2e-3 + 1e-5 * abs(want[n]), (n, sums[i], want[n])
    act = torch.from_numpy(z["actions"][0].copy())
    act[:, 1, 1:] = -act[:, 1, 1:]
    ref = (act * torch.tensor([2.0, 0.5, 0.5])).reshape(-1, 3)            # no clip before the scale in this wrapper
    assert torch.allclose(ref, torch.from_numpy(z["env_action"][0]), atol=1e-6)
    e.close()
    return True


def wrapper_replay(name, make_engine):
    z = golden("wrapper_" + name)
    T, N = z["obs"].shape[0], z["obs"].shape[1]
    d, keep, ctx = make_desc(TASK_OF[name], N)
    A, P = d.num_agents, d.num_npcs
    # the fixture uses synthetic origins / gate positions: patch the descriptor's per-env constants
    kw = ctx["cfg"].terrain.BarrierTrack_kwargs
    if name == "tug":
        return _tug_replay(z, d, keep, make_engine)
    if name in ("rotation", "bridge", "wrestling"):
        return _rotation_replay(z, d, keep, make_engine, name)
    eo = np.ascontiguousarray(z["env_origins"], np.float32)
    gate = np.ascontiguousarray(z["gate_deviation"], np.float32).copy()
    if name.startswith("sheep"):
        gate[:, 0] += kw["init"]["block_length"] + kw["plane"]["block_length"] + kw["gate"]["block_length"] / 2
    elif name in ("pushbox", "gate"):       # go1_gate_wrapper.py:41-42 (commented upstream), go1_pushbox_wrapper.py:29-30
        gate[:, 0] += kw["init"]["block_length"] + kw["gate"]["block_length"] / 2
    elif name == "football_defender":
        gate = eo[:, :2].copy()
        gate[:, 0] += kw["init"]["block_length"] + kw["plane"]["block_length"]
    keep += [eo, gate]
    d.env_origins = eo.ctypes.data_as(abi.FP)
    d.gate_pos = gate.ctypes.data_as(abi.FP)
    e = make_engine(d, keep)
    Tn = e.tensor
    bagt, root = Tn(abi.T_OBS_BAG), Tn(abi.T_ROOT_STATE)
    R = N * A

    def load(t):
        bagt[:, 0:3] = to_dev(e, z["base_pos"][t])
        bagt[:, 3:6] = to_dev(e, z["base_rpy"][t])
        root[:, A:, :] = to_dev(e, z["root_states_npc"][t].reshape(N, P, 13))
        Tn(abi.T_RESET_BUF).copy_(to_dev(e, z["reset_buf"][t].astype(np.uint8), torch.uint8))
        Tn(abi.T_COLLIDE_BUF).copy_(to_dev(e, z["collide_buf"][t].astype(np.uint8), torch.uint8))
        Tn(abi.T_R_TERM).copy_(to_dev(e, z["r_term_buff"][t].astype(np.uint8), torch.uint8))
        Tn(abi.T_P_TERM).copy_(to_dev(e, z["p_term_buff"][t].astype(np.uint8), torch.uint8))
        Tn(abi.T_SHEEP_POS_AVG).copy_(to_dev(e, z["sheep_pos_avg"][t]))
        Tn(abi.T_SHEEP_POS_VAR).copy_(to_dev(e, z["sheep_pos_var"][t]))
    load(0)
    e.wrapper_eval(1)
    close(Tn(abi.T_WRAPPER_OBS), z["obs_reset"], what="reset obs", atol=1e-6)
    for t in range(T):
        load(t + 1)
        e.wrapper_eval(0)
        close(Tn(abi.T_WRAPPER_OBS), z["obs"][t], what=f"t{t} obs", atol=1e-6)
        close(Tn(abi.T_WRAPPER_REWARD), z["reward"][t], what=f"t{t} reward", atol=2e-5, rtol=1e-5)
    sums = Tn(abi.T_REWARD_SUMS).double().sum(0).cpu().numpy()
    names = [n for _, n in REWARD_TERMS[{"sheep_hard": "sheep", "sheep_easy": "sheep"}.get(name, name)]]
    want = dict(zip([str(k) for k in z["reward_buffer_keys"]], z["reward_buffer_vals"]))
    for i, n in enumerate(names):
        if n is not None:
            assert abs(sums[i] - want[n]) <= 1e-3 + 1e-5 * abs(want[n]), (n, sums[i], want[n])
    e.close()
    return True
