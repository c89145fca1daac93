"""Scene sizes beyond the shipped tasks (VERDICT r4 missing 5; the reference builds whatever cfg.env.num_agents / num_npcs name,
legged_robot.py:854-902, go1_sheep.py:84-118): a flock of 16 sheep -- 2 robots + 16 sheep = 42 bodies, 84 generalized velocities, 153 actor
pairs in one wavefront -- on the CPU specification here and on the HIP engine against it under -m gpu.  What stays a limit, and is refused
with its reason: a fifth robot (13 body lanes each: 65 > the 64 lanes of the wavefront that owns an env)."""
import numpy as np
import pytest
import torch

from helpers import make_desc, oracle_engine, hip_engine, flock_cfg, close
from mqe.engine import abi


def test_sixteen_sheep_flock_on_the_specification():
    N = 6
    d, k, _ = make_desc("go1sheep-hard", N, cfg=flock_cfg(4, 4))
    assert d.num_npcs == 16 and abi.MAX_NPCS >= 16
    e = oracle_engine(d, k)
    e.reset_all()
    assert e.tensor(abi.T_WRAPPER_OBS).shape == (N, 2, 14 + 2 * 16 + 2)          # go1_sheep_wrapper.py:10
    g = torch.Generator().manual_seed(0)
    root = e.tensor(abi.T_ROOT_STATE)
    start = root[:, 2:, :2].clone()
    for t in range(40):
        e.step(torch.rand(N, 2, 3, generator=g) * 2 - 1)
    assert torch.isfinite(root).all() and int(e.tensor(abi.T_CONTACT_OVERFLOW).sum()) == 0
    z = root[:, 2:, 2]
    assert ((z > 0.2) & (z <= 0.3 + 1e-6)).all()                               # on the ground, clipped by the script (go1_sheep.py:60)
    assert (root[:, 2:, :2] - start).norm(dim=-1).max() > 0.05                 # the script moves them (randomness 0.1 x 2 per step)
    avg = e.tensor(abi.T_SHEEP_POS_AVG)
    # logged one step behind: the statistics of the positions the last script call saw = before its own update is integrated
    assert torch.isfinite(avg).all() and torch.isfinite(e.tensor(abi.T_SHEEP_POS_VAR)).all()
    _, _, c = e.debug_dynamics(0, 0)
    per_actor = np.bincount(c[c[:, 2] < 0, 0].astype(int), minlength=18)
    assert (per_actor[2:] >= 1).all() and per_actor[2:].max() <= 2              # every sheep stands on the ground


@pytest.mark.gpu
def test_sixteen_sheep_flock_hip_matches_the_specification(solver):
    """the rollout bounds of test_fused_rollout_matches_oracle on the 18-actor scene (lane sweep, 3 x 64 actor pairs in the broad phase,
    8 passes of NPC-NPC sphere pairs, the fused epilogue with 16 NPC rows staged)"""
    N = 24
    d1, k1, _ = make_desc("go1sheep-hard", N, cfg=flock_cfg(4, 4))
    d2, k2, _ = make_desc("go1sheep-hard", N, cfg=flock_cfg(4, 4))
    eh, eo = hip_engine(d1, k1), oracle_engine(d2, k2)
    eh.reset_all(); eo.reset_all()
    g = torch.Generator().manual_seed(5)
    dev, mism = [], 0
    for t in range(20):
        a = torch.rand(N, 2, 3, generator=g) * 2 - 1
        eh.step(a.cuda().contiguous()); eo.step(a)
        torch.cuda.synchronize()
        rh, ro = eh.tensor(abi.T_ROOT_STATE).cpu(), eo.tensor(abi.T_ROOT_STATE)
        dev.append((rh[..., :3] - ro[..., :3]).abs().amax(dim=(1, 2)))
        mism += int((eh.tensor(abi.T_RESET_BUF).cpu() != eo.tensor(abi.T_RESET_BUF)).sum())
        if t == 0:
            close(eh.tensor(abi.T_WRAPPER_OBS), eo.tensor(abi.T_WRAPPER_OBS), atol=2e-4, what="wrapper obs (48 columns) after 1 step")
    dev = torch.stack(dev)
    # (the deviation is the maximum over the env's 18 actors of an ABSOLUTE position difference, and the flock stands 6-12 m from the origin, where
    # one f32 ulp is 1e-6 m: measured after 20 steps median 3.8e-6 = 4 ulp, max 1.2e-5)
    assert float(dev[4].median()) < 2e-6 and float(dev[4].max()) < 2e-5, (dev[4].median(), dev[4].max())
    assert float(dev[19].median()) < 1e-5 and float(dev[19].max()) < 5e-4, (dev[19].median(), dev[19].max())
    assert mism == 0 and int(eh.tensor(abi.T_CONTACT_OVERFLOW).sum()) == int(eo.tensor(abi.T_CONTACT_OVERFLOW).sum()) == 0
    # two sheep pushed into each other and into a robot: the NPC-NPC and robot-NPC pair passes on the 18-actor scene, identical lists
    ro, rh = eo.tensor(abi.T_ROOT_STATE), eh.tensor(abi.T_ROOT_STATE)
    ro[:, 3, :2] = ro[:, 2, :2] + torch.tensor([0.33, 0.0])
    ro[:, 17, :2] = ro[:, 16, :2] + torch.tensor([0.0, 0.35])
    ro[:, 10, :3] = ro[:, 0, :3] + torch.tensor([0.45, 0.0, -0.05])
    rh.copy_(ro.cuda())
    torch.cuda.synchronize()
    for env in (0, N - 1):
        _, ch = eh.debug_dynamics(env, 0)
        _, _, co = eo.debug_dynamics(env, 0)
        assert ch.shape == co.shape and (ch[:, :4] == co[:, :4]).all(), (env, ch[:, :4], co[:, :4])
        assert (co[:, 2] >= 2).sum() >= 2                                         # sheep-sheep contacts are in the list
        close(ch[:, 4:], co[:, 4:], atol=2e-5, what="contact separation / normal")


def _pile_up(e, N):
    """both robots lying on their sides in the middle of the flock, the sheep packed around them: every actor's one-sided contacts at their caps"""
    root = e.tensor(abi.T_ROOT_STATE)
    r = root.cpu().clone() if root.is_cuda else root.clone()
    c = r[:, 2:, :2].mean(dim=1)                                   # flock centre
    for a in range(2):
        r[:, a, 0] = c[:, 0] + (0.35 if a else -0.35); r[:, a, 1] = c[:, 1]; r[:, a, 2] = 0.14
        r[:, a, 3:7] = torch.tensor([0.70710678, 0.0, 0.0, 0.70710678])      # rolled 90 degrees about x: lying on the side
        r[:, a, 7:] = 0.0
    for p in range(16):                                              # a 4 x 4 grid at 0.36 m pitch (sheep radius 0.2: neighbours overlap), robots on top of it
        r[:, 2 + p, 0] = c[:, 0] + ((p % 4) - 1.5) * 0.36; r[:, 2 + p, 1] = c[:, 1] + ((p // 4) - 1.5) * 0.36
        r[:, 2 + p, 7:] = 0.0
    root.copy_(r.to(root.device))


def test_packed_flock_under_fallen_robots_keeps_every_contact_class():
    """2 robots + 16 sheep have 8 * 2 + 2 * 16 = 48 one-sided slots -- round 5 clipped the list at 40, so with both robots down
    and every sheep on the ground the last sheep lost their ground contacts and no robot-sheep contact fitted.  The list now holds the sum of the
    per-actor caps plus eight slots only two-actor contacts can take (mqe_maxc)."""
    N = 4
    d, k, _ = make_desc("go1sheep-hard", N, cfg=flock_cfg(4, 4))
    e = oracle_engine(d, k)
    e.reset_all()
    for t in range(10):
        e.step(torch.zeros(N, 2, 3))                                  # the sheep settle on the ground
    _pile_up(e, N)
    _, _, c = e.debug_dynamics(0, 0)
    one = c[c[:, 2] < 0]
    per_actor = np.bincount(one[:, 0].astype(int), minlength=18)
    assert (per_actor[2:] >= 1).all(), per_actor                     # every sheep keeps its ground contact, the last ones included
    assert per_actor[0] == 8 and per_actor[1] == 8, per_actor        # both robots at their cap (lying on the side: > 8 feature points touch)
    pairs = c[c[:, 2] >= 0]
    assert ((pairs[:, 0] < 2) & (pairs[:, 2] >= 2)).sum() >= 1, pairs[:, :4]      # robot-sheep contacts survive
    assert len(pairs) >= 8 and len(c) <= 56                           # the eight pair-only slots at least (here the sheep leave more of the pool free)
    for t in range(3):
        e.step(torch.zeros(N, 2, 3))
    assert torch.isfinite(e.tensor(abi.T_ROOT_STATE)).all()


@pytest.mark.gpu
def test_packed_flock_under_fallen_robots_hip_matches_the_specification():
    N = 4
    d1, k1, _ = make_desc("go1sheep-hard", N, cfg=flock_cfg(4, 4))
    d2, k2, _ = make_desc("go1sheep-hard", N, cfg=flock_cfg(4, 4))
    eh, eo = hip_engine(d1, k1), oracle_engine(d2, k2)
    eh.reset_all(); eo.reset_all()
    for t in range(10):
        eo.step(torch.zeros(N, 2, 3))
    _pile_up(eo, N)
    eh.tensor(abi.T_ROOT_STATE).copy_(eo.tensor(abi.T_ROOT_STATE).cuda())
    eh.tensor(abi.T_DOF_STATE).copy_(eo.tensor(abi.T_DOF_STATE).cuda())
    torch.cuda.synchronize()
    for env in (0, N - 1):
        _, ch = eh.debug_dynamics(env, 0)
        _, _, co = eo.debug_dynamics(env, 0)
        assert ch.shape == co.shape and (ch[:, :4] == co[:, :4]).all(), (env, ch[:, :4], co[:, :4])
        close(ch[:, 4:], co[:, 4:], atol=2e-5, what="contact separation / normal")


def _four_robots_and_a_box_cfg():
    """go1football-2vs2's four robots with go1pushbox's free box instead of the ball: five actors, one of them a 6-dof NPC with four contact slots --
    the scene shape that takes the hybrid sweep's general NPC lane step (kernels_physics.hpp) on the generic kernel"""
    from helpers import task_cfg
    base, box = task_cfg("go1football-2vs2"), task_cfg("go1pushbox")
    asset = type("asset", (base.asset,), {"file_npc": box.asset.file_npc, "name_npc": box.asset.name_npc})
    init = type("init_state", (base.init_state,), {"init_states_npc": box.init_state.init_states_npc})
    return type("FourRobotsAndABoxCfg", (base,), {"asset": asset, "init_state": init})


def test_four_robots_and_a_box_on_the_specification():
    N = 3
    d, k, _ = make_desc("go1football-2vs2", N, cfg=_four_robots_and_a_box_cfg())
    assert d.num_agents == 4 and d.num_npcs == 1 and d.npc_kind == abi.NPC["box"]
    e = oracle_engine(d, k)
    e.reset_all()
    g = torch.Generator().manual_seed(2)
    for t in range(15):
        e.step(torch.rand(N, e.tensor(abi.T_WRAPPER_OBS).shape[1], 3, generator=g) * 2 - 1)
    root = e.tensor(abi.T_ROOT_STATE)
    assert torch.isfinite(root).all() and int(e.tensor(abi.T_CONTACT_OVERFLOW).sum()) == 0
    assert (root[:, 4, 2] > 0.3).all()                               # the box rests on the ground (half height 0.5 m)


@pytest.mark.gpu
def test_four_robots_and_a_box_hip_matches_the_specification(solver):
    N = 16
    d1, k1, _ = make_desc("go1football-2vs2", N, cfg=_four_robots_and_a_box_cfg())
    d2, k2, _ = make_desc("go1football-2vs2", N, cfg=_four_robots_and_a_box_cfg())
    eh, eo = hip_engine(d1, k1), oracle_engine(d2, k2)
    eh.reset_all(); eo.reset_all()
    # two robots pushed against the box so that robot-box pairs and the box's own ground contacts are in the list from the first step
    ro = eo.tensor(abi.T_ROOT_STATE)
    ro[:, 0, :2] = ro[:, 4, :2] + torch.tensor([-0.78, 0.0]); ro[:, 2, :2] = ro[:, 4, :2] + torch.tensor([0.0, 0.80])
    eh.tensor(abi.T_ROOT_STATE).copy_(ro.cuda())
    g = torch.Generator().manual_seed(9)
    Aw = eo.tensor(abi.T_WRAPPER_OBS).shape[1]
    dev = []
    for t in range(12):
        a = torch.rand(N, Aw, 3, generator=g) * 2 - 1
        eh.step(a.cuda().contiguous()); eo.step(a)
        torch.cuda.synchronize()
        dev.append((eh.tensor(abi.T_ROOT_STATE).cpu()[..., :3] - eo.tensor(abi.T_ROOT_STATE)[..., :3]).abs().amax(dim=(1, 2)))
        assert (eh.tensor(abi.T_RESET_BUF).cpu() == eo.tensor(abi.T_RESET_BUF)).all()
    dev = torch.stack(dev)
    assert float(dev[3].median()) < 5e-6 and float(dev[3].max()) < 2e-4, (dev[3].median(), dev[3].max())
    assert float(dev[11].median()) < 2e-5 and float(dev[11].max()) < 2e-3, (dev[11].median(), dev[11].max())
    assert int(eh.tensor(abi.T_CONTACT_OVERFLOW).sum()) == int(eo.tensor(abi.T_CONTACT_OVERFLOW).sum())
    for env in (0, N - 1):
        _, ch = eh.debug_dynamics(env, 0)
        _, _, co = eo.debug_dynamics(env, 0)
        assert ch.shape == co.shape and (ch[:, :4] == co[:, :4]).all(), (env, ch[:, :4], co[:, :4])


@pytest.mark.gpu
def test_a_fifth_robot_is_refused_with_the_reason():
    d, k, _ = make_desc("go1gate", 4)
    d.num_agents = 5
    with pytest.raises(RuntimeError, match="64-lane wavefront"):
        hip_engine(d, k)
    d, k, _ = make_desc("go1sheep-hard", 4)
    d.num_npcs = 17
    with pytest.raises(RuntimeError, match="num_npcs"):
        hip_engine(d, k)
