"""-m gpu: the HIP engine (through the C ABI) against (1) the reference's golden traces for everything around the
physics and (2) the CPU oracle for the physics, on identical seeded states."""
import os

import numpy as np
import pytest
import torch

from mqe.engine import abi
from helpers import hip_engine, oracle_engine, make_desc, to_dev, close
from replay import replay
from wrapper_replay import wrapper_replay

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["gate", "seesaw", "football", "sheep", "football1v1", "football2v2", "pushbox", "rotation", "bridge", "wrestling", "tug", "gate_cmd", "pushbox_curriculum"])
def test_hip_matches_reference_trace(name):
    assert replay(name, hip_engine)


@pytest.mark.parametrize("name", ["gate", "sheep_hard", "sheep_easy", "seesaw", "football_defender", "pushbox", "rotation", "bridge", "wrestling", "tug"])
def test_hip_wrappers_match_reference(name):
    assert wrapper_replay(name, hip_engine)


def _pair(task, N, **kw):
    d1, k1, ctx = make_desc(task, N, **kw)
    d2, k2, _ = make_desc(task, N, **kw)
    return hip_engine(d1, k1), oracle_engine(d2, k2), d1


def _randomize(eh, eo, seed, drop=0.0, spread=1.0):
    """identical perturbed states in both engines: reset, then jitter poses / velocities"""
    eh.reset_all(); eo.reset_all()
    torch.cuda.synchronize()
    g = torch.Generator().manual_seed(seed)
    ro, do = eo.tensor(abi.T_ROOT_STATE), eo.tensor(abi.T_DOF_STATE)
    ro[:, :, 2] -= drop
    ro[:, :, 0:2] += (torch.rand(ro[:, :, 0:2].shape, generator=g) - 0.5) * 0.2 * spread
    q = ro[:, :, 3:7] + torch.randn(ro[:, :, 3:7].shape, generator=g) * 0.08 * spread
    ro[:, :, 3:7] = q / q.norm(dim=-1, keepdim=True)
    ro[:, :, 7:13] = torch.randn(ro[:, :, 7:13].shape, generator=g) * 0.3 * spread
    do[:, :, 1] = torch.randn(do[:, :, 1].shape, generator=g) * 1.0 * spread
    eh.tensor(abi.T_ROOT_STATE).copy_(ro.cuda()); eh.tensor(abi.T_DOF_STATE).copy_(do.cuda())
    tau = (torch.rand(eo.tensor(abi.T_TORQUES).shape, generator=g) - 0.5) * 10
    eo.tensor(abi.T_TORQUES).copy_(tau); eh.tensor(abi.T_TORQUES).copy_(tau.cuda())


@pytest.mark.parametrize("task,N", [("go1gate", 64), ("go1football-defender", 16), ("go1sheep-hard", 14)])
def test_mass_matrix_inverse_and_contacts(task, N):
    eh, eo, d = _pair(task, N)
    _randomize(eh, eo, 3, drop=0.12)
    for env in (0, N // 2, N - 1):
        for robot in range(d.num_agents):
            mh, ch = eh.debug_dynamics(env, robot)
            _, mo, co = eo.debug_dynamics(env, robot)
            close(mh, mo, atol=2e-4, rtol=2e-3, what=f"Minv env {env} robot {robot}")
            assert ch.shape == co.shape, (ch.shape, co.shape)
            assert (ch[:, :4] == co[:, :4]).all(), "contact list (actor, body/sphere ids) must be identical and ordered"
            close(ch[:, 4:], co[:, 4:], atol=2e-5, what="contact separation / normal")


@pytest.mark.parametrize("task,N", [("go1gate", 256), ("go1football-defender", 32), ("go1sheep-hard", 21), ("go1football-2vs2", 12), ("go1pushbox", 32), ("go1bridge", 24), ("go1wrestling", 24)])
def test_single_substep_matches_oracle(task, N, solver):
    """one 5 ms simulate() from identical states: tolerance 2e-4 abs on positions/velocities (float32 both sides;
    the HIP kernel sums in a different order: CRBA + Schur complement vs per-body Jacobian sums + Cholesky)"""
    eh, eo, d = _pair(task, N)
    for seed, drop in ((1, 0.0), (2, 0.11), (3, 0.2)):
        _randomize(eh, eo, seed, drop=drop)
        eh.simulate(); eo.simulate()
        torch.cuda.synchronize()
        if solver == "tgs":
            # The temporal solver turns a separation into a velocity within dt / 4: the 1-ulp differences of the two engines' contact
            # geometry (2e-7 m at 2 m from the origin) become 1.5e-4 m/s of bias, and in these rough states (robots dropped up to 20 cm
            # INTO the ground, a dozen saturated contacts per env) a barely touching contact can tip the whole solve: measured 1 env of
            # 768 off by 1.3e-4 rad / 0.1 rad/s, the others as below.  Held per env: all but 1 % inside the strict bounds, every env
            # inside loose ones.
            def per_env(a, b, atol, rtol=0.0):
                a, b = a.detach().cpu().double().reshape(N, -1), b.detach().cpu().double().reshape(N, -1)
                return ((a - b).abs() / (atol + rtol * b.abs())).max(dim=1).values
            worst = torch.stack([per_env(eh.tensor(abi.T_DOF_STATE)[..., 0], eo.tensor(abi.T_DOF_STATE)[..., 0], 2e-5),
                                 per_env(eh.tensor(abi.T_DOF_STATE)[..., 1], eo.tensor(abi.T_DOF_STATE)[..., 1], 5e-3, 1e-3),
                                 per_env(eh.tensor(abi.T_ROOT_STATE)[..., :7], eo.tensor(abi.T_ROOT_STATE)[..., :7], 2e-5),
                                 per_env(eh.tensor(abi.T_ROOT_STATE)[..., 7:], eo.tensor(abi.T_ROOT_STATE)[..., 7:], 2e-3, 1e-3),
                                 per_env(eh.tensor(abi.T_CONTACT_FORCE), eo.tensor(abi.T_CONTACT_FORCE), 0.5, 2e-2)]).max(dim=0).values
            assert int((worst > 1.0).sum()) <= max(1, N // 100) and float(worst.max()) < 50.0, (seed, worst.topk(min(4, N)))
            continue
        close(eh.tensor(abi.T_DOF_STATE)[..., 0], eo.tensor(abi.T_DOF_STATE)[..., 0], atol=2e-5, what="dof pos")
        close(eh.tensor(abi.T_DOF_STATE)[..., 1], eo.tensor(abi.T_DOF_STATE)[..., 1], atol=5e-3, rtol=1e-3, what="dof vel")
        close(eh.tensor(abi.T_ROOT_STATE)[..., :7], eo.tensor(abi.T_ROOT_STATE)[..., :7], atol=2e-5, what="root pose")
        close(eh.tensor(abi.T_ROOT_STATE)[..., 7:], eo.tensor(abi.T_ROOT_STATE)[..., 7:], atol=2e-3, rtol=1e-3, what="root vel")
        close(eh.tensor(abi.T_CONTACT_FORCE), eo.tensor(abi.T_CONTACT_FORCE), atol=0.5, rtol=2e-2, what="net contact force")


@pytest.mark.parametrize("split,N", [("1", 64), ("0", 64), ("1", 101)])     # 101 envs: 202 rows = one full + one ragged 128-row tile
def test_policy_layer0_paths_are_f32_equivalent(monkeypatch, split, N):
    """Layer 0 of the locomotion policy runs on the f16 matrix cores with two-plane split operands (k_gemm_h2: hh + hl + lh,
    large batches by default) or on the exact-f32 MFMA kernel (MQE_GEMM_SPLIT=0; the engine picks by batch size otherwise).  Either way the joint targets must agree with the CPU
    oracle's f32 fmaf chain to |diff| <= 5e-5 on O(1) outputs after a history of 12 random steps (tolerance = f32
    accumulation-order noise; a plain f16, bf16 or TF32 product would miss it by two orders of magnitude)."""
    monkeypatch.setenv("MQE_GEMM_SPLIT", split)
    eh, eo, d = _pair("go1gate", N)
    monkeypatch.delenv("MQE_GEMM_SPLIT")
    eh.reset_all(); eo.reset_all()
    g = torch.Generator().manual_seed(5)
    for t in range(12):
        a = torch.rand(N, 2, 3, generator=g) * 2 - 1
        eh.step(a.cuda().contiguous()); eo.step(a)
        # keep both engines on the same trajectory so that step t compares the policy on identical histories
        for k in (abi.T_ROOT_STATE, abi.T_DOF_STATE):
            eh.tensor(k).copy_(eo.tensor(k).cuda())
        if t in (0, 3, 11):
            close(eh.tensor(abi.T_ACTIONS), eo.tensor(abi.T_ACTIONS), atol=5e-5, what=f"policy actions step {t}")


def test_layer0_half_tiles_equal_full_tiles_bit_for_bit(monkeypatch):
    """k_gemm_h2_mix (round 4): whole rounds of 128 x 192 tiles and a remainder of at most half a round as 64 x 192 tiles.  A half tile walks K
    in the same order with the same three plane products per element, so its outputs equal a full tile's bit for bit: two HIP engines at a
    batch that mixes both (4132 envs = 8264 rows: 64 full M-tiles + 2 half tiles, the second ragged) against MQE_GEMM_HALF=0 (65 full tiles),
    and at 2048 envs (32 full tiles = exactly half a round -> 64 half tiles)."""
    for N in (4132, 2048):
        d1, k1, _ = make_desc("go1gate", N)
        e1 = hip_engine(d1, k1)
        monkeypatch.setenv("MQE_GEMM_HALF", "0")
        d2, k2, _ = make_desc("go1gate", N)
        e2 = hip_engine(d2, k2)
        monkeypatch.delenv("MQE_GEMM_HALF")
        e1.reset_all(); e2.reset_all()
        g = torch.Generator(device="cuda").manual_seed(3)
        for t in range(4):
            a = torch.rand(N, 2, 3, device="cuda", generator=g) * 2 - 1
            e1.step(a); e2.step(a)
        torch.cuda.synchronize()
        for k in (abi.T_ACTIONS, abi.T_ROOT_STATE, abi.T_DOF_STATE, abi.T_WRAPPER_OBS):
            assert torch.equal(e1.tensor(k), e2.tensor(k)), (N, k)
        del e1, e2


def test_history_written_by_the_host_reaches_the_policy(monkeypatch):
    """MQE_T_HISTORY is the reference's obs_history: a host may write it.  Large batches run layer 0 on a compact split-f16 copy of the ring,
    which mqe_history_sync rebuilds from the ring (presence flags, carrier columns, continuity bits included): after an edit of the ring --
    frames rescaled, one frame zeroed as a reset would, the action columns of one frame changed so that it no longer continues its
    predecessor -- the split path's joint targets equal those of the exact-f32 path, which reads the ring itself, to the usual 5e-5."""
    N = 48
    monkeypatch.setenv("MQE_GEMM_SPLIT", "1")
    d1, k1, _ = make_desc("go1gate", N)
    e1 = hip_engine(d1, k1)
    monkeypatch.setenv("MQE_GEMM_SPLIT", "0")
    d2, k2, _ = make_desc("go1gate", N)
    e2 = hip_engine(d2, k2)
    monkeypatch.delenv("MQE_GEMM_SPLIT")
    e1.reset_all(); e2.reset_all()
    g = torch.Generator(device="cuda").manual_seed(9)
    for t in range(34):                                   # the ring turns over once
        a = torch.rand(N, 2, 3, device="cuda", generator=g) * 2 - 1
        e1.step(a); e2.step(a)
        for k in (abi.T_ROOT_STATE, abi.T_DOF_STATE, abi.T_LAST_LOCO_ACTION, abi.T_LAST_TWO_LOCO_ACTION, abi.T_ACTIONS, abi.T_LAST_ACTIONS, abi.T_HISTORY):
            e2.tensor(k).copy_(e1.tensor(k))             # both engines on one trajectory and one ring
    h = e1.tensor(abi.T_HISTORY).clone()
    scale = 0.5 + torch.rand(h.shape[0], abi.HIST, 1, device="cuda", generator=g)
    keep = torch.ones(72, device="cuda"); keep[6:18] = 0   # the gait parameters stay the scene's constants
    h = h * (1 + (scale - 1) * keep)
    h[::3, 7] = 0.0                                        # a frame of zeros in every third robot
    h[1::3, 11, 54:66] += 0.25                             # a frame that does not continue its predecessor
    h[2::3, 20, 42:54] -= 0.125                            # ... and a predecessor changed under its successor
    for e in (e1, e2):
        e.tensor(abi.T_HISTORY).copy_(h)
    e1.history_sync()
    a = torch.rand(N, 2, 3, device="cuda", generator=g) * 2 - 1
    e1.step(a); e2.step(a)
    torch.cuda.synchronize()
    close(e1.tensor(abi.T_ACTIONS), e2.tensor(abi.T_ACTIONS), atol=5e-5, what="joint targets after a host edit of the history ring")
    # and without the sync the compact copy is stale: the same comparison must FAIL (the test would be vacuous otherwise)
    for e in (e1, e2):
        e.tensor(abi.T_HISTORY).copy_(h * 0.5 * keep + h * (1 - keep))
    e1.step(a); e2.step(a)
    torch.cuda.synchronize()
    assert float((e1.tensor(abi.T_ACTIONS) - e2.tensor(abi.T_ACTIONS)).abs().max()) > 1e-3


@pytest.mark.parametrize("split", ["1", "0"])
def test_policy_layer0_compact_history_with_resets(monkeypatch, split):
    """The split-f16 operand of layer 0 does not store last_two_locomotion_action: frame p's copy is frame p-1's
    last_locomotion_action, its weights ride on that frame's columns (mqe_common.hpp, MQE_H2_FRAME).  The two places where the
    identity has no partner -- the oldest frame of the ring, and the first frame after a reset (go1.py:139-145 zeroes the history but
    not the action registers) -- are carried separately (carrier columns; residual term in k_gemm_h2's epilogue).  45 steps (the
    ring turns over 1.5 times) with forced time-outs, two of them 2 steps apart in one env: the joint targets agree with the oracle's
    plain f32 chain over the full 2100-column history at every step -- on the split path and on the exact-f32 path (whose history
    operand is the ring itself).  This is also the only test that runs a policy long enough to fill the ring: the real adaptation
    module's hidden activations exceed 1000 on a full history (kernels_tail.hpp, TL_ASCALE)."""
    N = 16
    monkeypatch.setenv("MQE_GEMM_SPLIT", split)
    eh, eo, d = _pair("go1gate", N)
    monkeypatch.delenv("MQE_GEMM_SPLIT")
    eh.reset_all(); eo.reset_all()
    g = torch.Generator().manual_seed(17)
    forced = {5: [3], 20: [7], 22: [7, 8], 33: [0, 15], 34: [0]}
    n_reset = 0
    for t in range(45):
        a = torch.rand(N, 2, 3, generator=g) * 2 - 1
        for e in forced.get(t, []):
            eh.tensor(abi.T_EPISODE_LENGTH)[e] = 10 ** 6
            eo.tensor(abi.T_EPISODE_LENGTH)[e] = 10 ** 6
        eh.step(a.cuda().contiguous()); eo.step(a)
        torch.cuda.synchronize()
        rb = eo.tensor(abi.T_RESET_BUF)
        assert bool((eh.tensor(abi.T_RESET_BUF).cpu() == rb).all())
        for e in forced.get(t, []):
            assert int(rb[e]) == 1
        n_reset += int(rb.sum())
        close(eh.tensor(abi.T_ACTIONS), eo.tensor(abi.T_ACTIONS), atol=2e-5, what=f"policy actions step {t}")
        # same trajectory AND the same action registers (they feed back into the history): step t compares the policy on identical histories
        for k in (abi.T_ROOT_STATE, abi.T_DOF_STATE, abi.T_LAST_LOCO_ACTION, abi.T_LAST_TWO_LOCO_ACTION, abi.T_ACTIONS, abi.T_ACT_HIST,
                  abi.T_OBS_BAG, abi.T_GAIT_INDICES, abi.T_CLOCK_INPUTS, abi.T_LAST_ACTIONS):
            eh.tensor(k).copy_(eo.tensor(k).cuda())
    assert n_reset >= 7


@pytest.mark.parametrize("task,N", [("go1gate", 96), ("go1football-defender", 16), ("go1plane", 130), ("go1pushbox", 16)])
def test_exact_collision_model_matches_oracle(task, N):
    """collision_model = "exact" (thigh / calf as the URDF's boxes, 60 feature points per robot: one robot per terrain pass, the
    self-collision candidates in two chunks, no env pairing): identical contact lists from identical rough states, one substep and a
    12-step fused rollout against the oracle."""
    eh, eo, d = _pair(task, N, collision_model="exact")
    assert d.robot.n_spheres == 60
    _randomize(eh, eo, 5, drop=0.12)
    for env in (0, N // 2, N - 1):
        _, ch = eh.debug_dynamics(env, 0)
        _, _, co = eo.debug_dynamics(env, 0)
        assert ch.shape == co.shape and (ch[:, :4] == co[:, :4]).all(), (env, ch[:, :4], co[:, :4])
        close(ch[:, 4:], co[:, 4:], atol=2e-5, what="contact separation / normal")
    eh.simulate(); eo.simulate()
    torch.cuda.synchronize()
    err = (eh.tensor(abi.T_DOF_STATE)[..., 0].cpu() - eo.tensor(abi.T_DOF_STATE)[..., 0]).abs().reshape(N, -1).max(dim=1).values
    assert int((err > 2e-5).sum()) <= max(1, N // 50) and float(err.max()) < 5e-3, err.topk(3)
    eh.reset_all(); eo.reset_all()
    g = torch.Generator().manual_seed(3)
    Aw = eo.tensor(abi.T_WRAPPER_OBS).shape[1]
    for t in range(12):
        a = torch.rand(N, Aw, 3, generator=g) * 2 - 1
        eh.step(a.cuda().contiguous()); eo.step(a)
    torch.cuda.synchronize()
    dev = (eh.tensor(abi.T_ROOT_STATE).cpu()[..., :3] - eo.tensor(abi.T_ROOT_STATE)[..., :3]).abs().amax(dim=(1, 2))
    # (round 5, split-f16 actuator layer: torques differ from the oracle's by ~1e-6, and one env of go1pushbox's 16 -- a foot that one engine
    # lets slide and the other holds -- reads 1.5e-4 m after 12 steps; the median is unchanged at 2.4e-7)
    assert float(dev.median()) < 2e-6 and float(dev.quantile(0.99)) < 5e-4 and float(dev.max()) < 2e-3, (dev.median(), dev.max())
    assert torch.equal(eh.tensor(abi.T_RESET_BUF).cpu(), eo.tensor(abi.T_RESET_BUF))
    assert int(eh.tensor(abi.T_CONTACT_OVERFLOW).sum()) == int(eo.tensor(abi.T_CONTACT_OVERFLOW).sum())


def test_box_corner_contacts_match_oracle(solver):
    """go1pushbox: the free box balanced on one corner over the trunks of floating robots (its corners against the robots' primitives,
    round 4): identical contact lists -- box against base link -- and the same motion over 8 substeps."""
    N = 8
    eh, eo, d = _pair("go1pushbox", N)
    eh.reset_all(); eo.reset_all()
    torch.cuda.synchronize()
    ro, do = eo.tensor(abi.T_ROOT_STATE), eo.tensor(abi.T_DOF_STATE)
    g = torch.Generator().manual_seed(2)
    A = 2
    h = np.array([d.npc_box_half[0], d.npc_box_half[1], d.npc_box_half[2]], np.float64)
    v = -h / np.linalg.norm(h); t = np.array([0.0, 0.0, -1.0])
    ax = np.cross(v, t); s_, c_ = np.linalg.norm(ax), float(v @ t); ax /= s_
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + s_ * K + (1 - c_) * K @ K
    ang = np.arctan2(s_, c_)
    quat = torch.tensor(np.concatenate([ax * np.sin(ang / 2), [np.cos(ang / 2)]]), dtype=torch.float32)
    ro[:, 0, 2] = 2.0; ro[:, 1, 2] = 2.0; ro[:, 1, 1] += 3.0
    ro[:, :, 3:7] = torch.tensor([0.0, 0.0, 0.0, 1.0]); ro[:, :, 7:] = 0; do[..., 1] = 0
    top = ro[:, 0, :3].clone()
    top[:, 0] += (torch.rand(N, generator=g) - 0.5) * 0.2
    top[:, 1] += (torch.rand(N, generator=g) - 0.5) * 0.05
    top[:, 2] += 0.057 + 0.002                                    # the trunk box's half height (go1.urdf:56) + 2 mm
    ro[:, A, :3] = top - torch.tensor(R @ (-h), dtype=torch.float32)
    ro[:, A, 3:7] = quat
    eh.tensor(abi.T_ROOT_STATE).copy_(ro.cuda()); eh.tensor(abi.T_DOF_STATE).copy_(do.cuda())
    eh.tensor(abi.T_TORQUES).zero_(); eo.tensor(abi.T_TORQUES).zero_()
    saw = 0
    for env in range(N):
        _, ch = eh.debug_dynamics(env, 0)
        _, _, co = eo.debug_dynamics(env, 0)
        assert ch.shape == co.shape and (ch[:, :4] == co[:, :4]).all(), (env, ch, co)
        close(ch[:, 4:], co[:, 4:], atol=2e-5, what="contact separation / normal")
        saw += int(((co[:, 0] == 0) & (co[:, 1] == 0) & (co[:, 2] == A)).sum())
    assert saw >= N, "every env must hold the corner-on-trunk contact"
    for k in range(8):
        eh.simulate(); eo.simulate()
    torch.cuda.synchronize()
    close(eh.tensor(abi.T_ROOT_STATE)[..., :7], eo.tensor(abi.T_ROOT_STATE)[..., :7], atol=2e-4, what="poses after 8 substeps")


def test_edge_contacts_match_oracle():
    """edge contacts (round 4; include/mqe_hip.h edge_contacts bits 1, 2 and 4): robots scattered around the gate posts at every yaw -- post
    corners against the sides of trunks, heads and legs -- and robots leaning over the free box's edges: the contact lists of the two
    engines are identical (ids and order exact, separations 5e-5, normals 2e-2 where the golden-section search ends on a flat stretch), some
    contact is an edge contact, and a substep agrees."""
    box_edge_contacts = 0
    for task, N, mask in (("go1gate", 96, 3), ("go1pushbox", 48, 3), ("go1seesaw", 32, 3), ("go1pushbox", 48, 7), ("go1seesaw", 32, 7), ("go1bridge", 32, 7)):
        # (mask 7 adds bit 4, the scene boxes' EDGES against the robots' box primitives -- the trunk's side on the box's vertical edge, the
        # plank's long edge under the trunk -- which the HIP engine refused until round 4's last session)
        eh, eo, d = _pair(task, N, edge_contacts=mask)
        eh.reset_all(); eo.reset_all()
        torch.cuda.synchronize()
        ro, do = eo.tensor(abi.T_ROOT_STATE), eo.tensor(abi.T_DOF_STATE)
        g = torch.Generator().manual_seed(21)
        A = 2
        yaw = torch.rand(N, A, generator=g) * 6.283
        ro[:, :A, 3] = 0; ro[:, :A, 4] = 0; ro[:, :A, 5] = torch.sin(yaw / 2); ro[:, :A, 6] = torch.cos(yaw / 2)
        if task == "go1gate":
            gate = torch.tensor(np.ctypeslib.as_array(d.gate_pos, shape=(N, 2)).copy())
            eo_ = torch.tensor(np.ctypeslib.as_array(d.env_origins, shape=(N, 3)).copy())
            centre = eo_[:, :2] + gate                      # the gate's middle in the world; the posts stand +-0.29 m beside it
            for a in range(A):
                ro[:, a, 0] = centre[:, 0] + (torch.rand(N, generator=g) - 0.5) * 0.5
                ro[:, a, 1] = centre[:, 1] + (1 if a else -1) * (0.29 + 0.05 + torch.rand(N, generator=g) * 0.12)
                ro[:, a, 2] = 0.30 + torch.rand(N, generator=g) * 0.05
        elif task == "go1pushbox":
            box = ro[:, A, :3].clone()
            ro[:, A, 2] = d.ground_z + d.npc_box_half[2]
            ro[:, A, 5] = 0.3827; ro[:, A, 6] = 0.9239                               # the box turned by 45 degrees: its vertical edges point outwards
            for a in range(A):                                                       # robot 0 at the +x edge, robot 1 at the -x edge
                sx = 1.0 if a == 0 else -1.0
                ro[:, a, 0] = box[:, 0] + sx * (d.npc_box_half[0] * 1.4142 + 0.02 + torch.rand(N, generator=g) * 0.2)
                ro[:, a, 1] = box[:, 1] + (torch.rand(N, generator=g) - 0.5) * 0.4
                ro[:, a, 2] = 0.31
        elif task == "go1bridge":                            # static scenery: robots hanging over the long top edges of the first box
            nb = ro[:, A, :3].clone()
            c0 = torch.tensor([d.static_box_center[0][k] for k in range(3)]); h0 = torch.tensor([d.static_box_half[0][k] for k in range(3)])
            for a in range(A):
                ro[:, a, 0] = nb[:, 0] + c0[0] + (torch.rand(N, generator=g) - 0.5) * 1.6 * h0[0]
                ro[:, a, 1] = nb[:, 1] + c0[1] + (1 if a else -1) * (h0[1] + 0.02 + torch.rand(N, generator=g) * 0.1)
                ro[:, a, 2] = nb[:, 2] + c0[2] + h0[2] + 0.03 + torch.rand(N, generator=g) * 0.2
        else:
            hinge = ro[:, A, :3].clone()
            for a in range(A):
                ro[:, a, 0] = hinge[:, 0] - 2.4 + (torch.rand(N, generator=g) - 0.5) * 3.0
                ro[:, a, 1] = hinge[:, 1] + (1 if a else -1) * (0.5 + 0.02 + torch.rand(N, generator=g) * 0.1)      # beside the plank's long edges
                ro[:, a, 2] = 0.545 + 0.05 + torch.rand(N, generator=g) * 0.2
        ro[:, :, 7:] = 0; do[..., 1] = 0
        do[:, :12 * A, 0] += (torch.rand(N, 12 * A, generator=g) - 0.5) * 0.6
        eh.tensor(abi.T_ROOT_STATE).copy_(ro.cuda()); eh.tensor(abi.T_DOF_STATE).copy_(do.cuda())
        eh.tensor(abi.T_TORQUES).zero_(); eo.tensor(abi.T_TORQUES).zero_()
        d0, k0, _ = make_desc(task, N, edge_contacts=0 if mask == 3 else 3)           # the same scene without edge contacts (mask 7: without bit 4): what they add
        e0 = oracle_engine(d0, k0)
        e0.reset_all()
        e0.tensor(abi.T_ROOT_STATE).copy_(ro); e0.tensor(abi.T_DOF_STATE).copy_(do)
        edges = 0
        for env in range(N):
            _, ch = eh.debug_dynamics(env, 0)
            _, _, co = eo.debug_dynamics(env, 0)
            assert ch.shape == co.shape and (ch[:, :4] == co[:, :4]).all(), (task, env, ch[:, :5], co[:, :5])
            close(ch[:, 4], co[:, 4], atol=5e-5, what="contact separation")
            # (an edge contact's normal: where the separation is flat along the axis to the rounding of the distance function -- 1e-6 m at
            # 3 m from the origin -- the search may stop anywhere within sqrt(2 d 1e-6) = 0.2 mm, i.e. 1e-2 rad at d = 2 cm)
            close(ch[:, 5:], co[:, 5:], atol=2e-2, what="contact normal")
            edges += len(co) - len(e0.debug_dynamics(env, 0)[2])
        if mask == 3:
            assert edges >= N // 16, (task, edges)                # the scatter does produce edge contacts
        else:
            box_edge_contacts += edges
        eh.simulate(); eo.simulate()
        torch.cuda.synchronize()
        err = (eh.tensor(abi.T_ROOT_STATE)[..., :7].cpu() - eo.tensor(abi.T_ROOT_STATE)[..., :7]).abs().reshape(N, -1).max(dim=1).values
        assert int((err > 5e-5).sum()) <= max(1, N // 30) and float(err.max()) < 5e-3, (task, err.topk(3))
        _record("edge_contacts", {"task": task, "N": N, "mask": mask, "contacts_on_links_without_feature_points": edges})
    assert box_edge_contacts >= 2, box_edge_contacts              # ... and box edges on box primitives


@pytest.mark.parametrize("task", ["go1sheep-hard", "go1pushbox", "go1gate"])
def test_staged_post_physics_is_the_single_launch(task):
    """the five stages of mqe_post_physics_stage (k_post_staged: a thread per env) in sequence == mqe_post_physics_step (k_post_physics, the
    fused kernel): flags and counters exactly, floats to the last bit or two (the same formulas, compiled in two kernels): NPC script, in-kernel
    resets with history zeroing, observations, wrapper, over 9 steps with time-outs"""
    N = 70
    engs = []
    for _ in range(2):
        d, k, _c = make_desc(task, N, max_episode_length=4)
        e = hip_engine(d, k); e.reset_all(); engs.append(e)
    g = torch.Generator().manual_seed(0)
    for t in range(9):
        cmd = (torch.rand(N * 2, 3, generator=g) * 2 - 1).cuda()
        for i, e in enumerate(engs):
            e.policy_step(cmd)
            for k_ in range(4):
                e.compute_torques(); e.simulate(); e.post_decimation_step(k_)
            if i == 0:
                e.post_physics_step()
            else:
                for st in (abi.POST_FRAME, abi.POST_NPC, abi.POST_RESET, abi.POST_OBS, abi.POST_WRAPPER):
                    e.post_physics_stage(st)
        torch.cuda.synchronize()
        for kind in (abi.T_ROOT_STATE, abi.T_DOF_STATE, abi.T_OBS_BAG, abi.T_WRAPPER_OBS, abi.T_WRAPPER_REWARD, abi.T_RESET_BUF, abi.T_EPISODE_LENGTH,
                     abi.T_HISTORY, abi.T_GAIT_INDICES, abi.T_CLOCK_INPUTS, abi.T_BASE_LIN_VEL, abi.T_LAST_ACTIONS, abi.T_REWARD_SUMS):
            a, b = engs[0].tensor(kind), engs[1].tensor(kind)
            if a.dtype != torch.float32:
                assert torch.equal(a, b), (task, t, kind)                  # flags, counters: exact
            else:       # the same formulas compiled in two kernels: the compiler contracts them differently (a yaw angle's last bit)
                assert torch.allclose(a, b, rtol=2e-6, atol=1e-7), (task, t, kind, (a - b).abs().max())
    assert int(engs[0].tensor(abi.T_RESET_COUNT).sum()) > N


def _record(kind, obj):
    """measured deviations, appended to gpurun_out/test_measurements.jsonl when that directory exists (what the bounds are set from)"""
    import json
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "test_measurements.jsonl"), "a") as f:
            f.write(json.dumps(dict(obj, kind=kind)) + "\n")


@pytest.mark.parametrize("task,N", [("go1gate", 128), ("go1football-defender", 16), ("go1sheep-hard", 14), ("go1seesaw", 16), ("go1football-2vs2", 8), ("go1football-1vs1", 16), ("go1pushbox", 16), ("go1revolvingdoor", 16), ("go1bridge", 16), ("go1wrestling", 16), ("go1tug", 16)])
def test_fused_rollout_matches_oracle(task, N, solver):
    """20 fused step() calls (policy + 4 substeps + post-step + wrapper) from the seeded reset distribution, nothing re-synchronised.
    Contact dynamics amplify rounding differences (a contact that closes one substep earlier), so the bound is a distribution
    over envs -- median, 99th percentile AND maximum of the base-position deviation -- and the reset flags must agree in every env
    at every step.  The bounds sit about 10 x above what is measured (this test on MI355X, round 3: worst task after 5 steps
    median 3e-8 / p99 7e-7 / max 1e-6 m, after 20 steps 6e-8 / 1.5e-6 / 2.9e-6 m; the 256-env sweep profiles/r03_parity_sweep.json:
    worst max 8e-5 m after 20 steps): a regression of one order of magnitude fails (round 2's bounds were 1000 x looser)."""
    eh, eo, d = _pair(task, N)
    eh.reset_all(); eo.reset_all()
    g = torch.Generator().manual_seed(11)
    Aw = eo.tensor(abi.T_WRAPPER_OBS).shape[1]
    dev_pos, mism = [], 0
    for t in range(20):
        a = torch.rand(N, Aw, 3, generator=g) * 2 - 1
        eh.step(a.cuda().contiguous()); eo.step(a)
        torch.cuda.synchronize()
        rh, ro = eh.tensor(abi.T_ROOT_STATE).cpu(), eo.tensor(abi.T_ROOT_STATE)
        dev_pos.append((rh[..., :3] - ro[..., :3]).abs().max(dim=-1).values.flatten())
        mism += int((eh.tensor(abi.T_RESET_BUF).cpu() != eo.tensor(abi.T_RESET_BUF)).sum())
        if t == 0:
            close(eh.tensor(abi.T_WRAPPER_OBS), eo.tensor(abi.T_WRAPPER_OBS), atol=2e-4, what="wrapper obs after 1 step")
            close(eh.tensor(abi.T_ACTIONS), eo.tensor(abi.T_ACTIONS), atol=5e-5, what="policy actions step 0")
    dev = torch.stack(dev_pos)
    assert torch.isfinite(dev).all()
    meas = {}
    BOUNDS = ((4, 5e-7, 1e-5, 2e-5), (19, 2e-6, 5e-5, 5e-4))
    for step, med, p99, mx in BOUNDS:
        got = (float(dev[step].median()), float(dev[step].quantile(0.99)), float(dev[step].max()))
        meas[step + 1] = got
    _record("rollout_dev", {"task": task, "N": N, "dev": meas})
    for step, med, p99, mx in BOUNDS:
        got = meas[step + 1]
        assert got[0] < med and got[1] < p99 and got[2] < mx, f"{task}: base position deviation after {step + 1} steps (median, p99, max) = {got}"
    assert mism == 0, f"{mism} reset-flag mismatches"
    assert int(eh.tensor(abi.T_CONTACT_OVERFLOW).sum()) == int(eo.tensor(abi.T_CONTACT_OVERFLOW).sum())


def test_fused_rollout_past_the_history_horizon():
    """The 20-step rollouts above stop before the 30-frame history has filled; this one runs go1gate for 60 fused steps with nothing
    re-synchronised, i.e. through the regime the policy spends its life in (full ring, hidden activations of the adaptation module
    beyond 1000, time-outs and falls -> frames after a reset).  Chaotic growth is what it is (contacts), so the bounds are on the
    distribution: median and 90th percentile of the base-position deviation, and on the reset flags of the envs whose robots are
    still within 1 mm of the oracle's."""
    N = 128
    eh, eo, d = _pair("go1gate", N)
    eh.reset_all(); eo.reset_all()
    g = torch.Generator().manual_seed(23)
    dev = None
    agree = torch.ones(N, dtype=torch.bool)
    mism = 0
    for t in range(60):
        a = torch.rand(N, 2, 3, generator=g) * 2 - 1
        eh.step(a.cuda().contiguous()); eo.step(a)
        torch.cuda.synchronize()
        rh, ro = eh.tensor(abi.T_ROOT_STATE).cpu(), eo.tensor(abi.T_ROOT_STATE)
        rbh, rbo = eh.tensor(abi.T_RESET_BUF).cpu(), eo.tensor(abi.T_RESET_BUF)
        mism += int(((rbh != rbo) & agree).sum())
        dev = (rh[..., :3] - ro[..., :3]).abs().amax(dim=(-1, -2))
        agree &= (dev < 1e-3) & (rbh == rbo)
        if t == 39:
            d40 = (float(dev.median()), float(dev.quantile(0.9)))
    d60 = (float(dev.median()), float(dev.quantile(0.9)))
    _record("rollout_long", {"task": "go1gate", "N": N, "median_p90_after_40": d40, "median_p90_after_60": d60, "envs_still_within_1mm": int(agree.sum())})
    assert torch.isfinite(dev).all()
    assert mism == 0, f"{mism} reset-flag mismatches in envs that were still within 1 mm"
    # measured (MI355X, round 3): median / p90 after 40 steps 2.4e-7 / 4.8e-7 m, after 60 steps 3.6e-7 / 5.1e-7 m, all 128 envs within 1 mm
    assert d40[0] < 5e-6 and d40[1] < 2e-5, d40
    assert d60[0] < 1e-5 and d60[1] < 5e-5, d60
    assert int(agree.sum()) >= N - N // 8, int(agree.sum())


@pytest.mark.parametrize("task,N", [("go1gate", 64), ("go1sheep-hard", 10)])
def test_time_outs_bring_both_engines_back_to_the_same_state(task, N):
    """Episodes of 35 steps, 110 fused steps: every env times out three times (on top of whatever falls in between).  The in-kernel
    reset draws its state from the hash RNG keyed by (seed, global env id, the env's reset count), so whatever the two trajectories
    did to each other before a time-out, a few steps after it both engines are back within rounding of each other: the deviation of
    the base positions is checked 4 steps after each wave of time-outs, the reset flags at every step for the envs that still agree."""
    eh, eo, d = _pair(task, N, max_episode_length=35)
    eh.reset_all(); eo.reset_all()
    g = torch.Generator().manual_seed(29)
    Aw = eo.tensor(abi.T_WRAPPER_OBS).shape[1]
    agree = torch.ones(N, dtype=torch.bool)
    after, n_time_outs, mism = [], 0, 0
    since = torch.full((N,), 1000)
    for t in range(110):
        a = torch.rand(N, Aw, 3, generator=g) * 2 - 1
        eh.step(a.cuda().contiguous()); eo.step(a)
        torch.cuda.synchronize()
        rh, ro = eh.tensor(abi.T_ROOT_STATE).cpu(), eo.tensor(abi.T_ROOT_STATE)
        rbh, rbo = eh.tensor(abi.T_RESET_BUF).cpu().bool(), eo.tensor(abi.T_RESET_BUF).bool()
        dev = (rh[..., :3] - ro[..., :3]).abs().amax(dim=(-1, -2))
        mism += int(((rbh != rbo) & agree).sum())
        n_time_outs += int(eo.tensor(abi.T_TIME_OUT_BUF).sum())
        both = rbh & rbo
        since = torch.where(both, torch.zeros_like(since), since + 1)
        agree = torch.where(both, torch.ones_like(agree), agree & (dev < 1e-3) & (rbh == rbo))
        sel = since == 4
        if sel.any():
            after.append(dev[sel])
    after = torch.cat(after)
    _record("time_out_resync", {"task": task, "N": N, "time_outs": n_time_outs, "checked": int(after.numel()), "median": float(after.median()), "max": float(after.max())})
    assert n_time_outs >= 2 * N and after.numel() >= 2 * N
    assert mism == 0, f"{mism} reset-flag mismatches in envs that still agreed"
    assert float(after.median()) < 1e-6 and float(after.quantile(0.95)) < 1e-5, (float(after.median()), float(after.max()))


@pytest.mark.parametrize("task,N", [("go1gate", 256), ("go1seesaw", 64), ("go1tug", 32), ("go1sheep-hard", 48), ("go1football-2vs2", 32), ("go1football-defender", 32)])
def test_row_sweep_and_lane_sweep_agree(monkeypatch, task, N, solver):
    """The contact sweep has two lane mappings (kernels_physics.hpp: one DPP row per actor -- round 6: per ROBOT, with the free NPCs of a
    scene of more than four actors stepped by a lane each: go1sheep-hard, go1football-2vs2 -- or one lane per contact).  MQE_LANE_SWEEP=1
    sends a scene down the other one (generic kernels, lane sweep, other LDS layout): same records, same Gauss-Seidel order, different
    summation trees -> 8 fused steps agree to rounding and every reset flag and contact-overflow count is identical."""
    d, keep, _ = make_desc(task, N)
    er = hip_engine(d, keep)
    monkeypatch.setenv("MQE_LANE_SWEEP", "1")
    d2, keep2, _ = make_desc(task, N)
    el = hip_engine(d2, keep2)
    monkeypatch.delenv("MQE_LANE_SWEEP")
    er.reset_all(); el.reset_all()
    g = torch.Generator().manual_seed(5)
    Aw = er.tensor(abi.T_WRAPPER_OBS).shape[1]
    for t in range(8):
        a = (torch.rand(N, Aw, 3, generator=g) * 2 - 1).cuda().contiguous()
        er.step(a); el.step(a)
        torch.cuda.synchronize()
        assert (er.tensor(abi.T_RESET_BUF) == el.tensor(abi.T_RESET_BUF)).all()
        if t in (0, 3):
            close(er.tensor(abi.T_ROOT_STATE), el.tensor(abi.T_ROOT_STATE), atol=2e-5 if t == 0 else 2e-4, what=f"root state step {t}")
            close(er.tensor(abi.T_DOF_STATE), el.tensor(abi.T_DOF_STATE), atol=2e-4 if t == 0 else 5e-3, what=f"dof state step {t}")
    dev = (er.tensor(abi.T_ROOT_STATE)[..., :3] - el.tensor(abi.T_ROOT_STATE)[..., :3]).abs().max(dim=-1).values.flatten()
    assert float(dev.median()) < 1e-5 and float(dev.quantile(0.99)) < 1e-4 and float(dev.max()) < 2e-3, (float(dev.median()), float(dev.max()))
    assert int(er.tensor(abi.T_CONTACT_OVERFLOW).sum()) == int(el.tensor(abi.T_CONTACT_OVERFLOW).sum())


@pytest.mark.parametrize("task,N,dr", [("go1gate", 256, False), ("go1gate", 37, False), ("go1gate", 1, False), ("go1plane", 65, False), ("go1plane", 131, True), ("go1gate", 50, True)])
def test_two_envs_per_wavefront_is_bit_identical(monkeypatch, task, N, dr):
    """k_substeps<.., EPW = 2>: each half-wave of 32 lanes runs an env of its own (robot-only scenes of <= 2 robots; the engine picks
    it for single-robot scenes of >= 4096 envs, MQE_ENVS_PER_WAVE = 1 / 2 forces either form).  Per env it is the same arithmetic in the same order as the one-env form, so
    15 fused steps -- resets, contacts between the robots, joint limits, an odd batch whose last half-wave has no env -- agree BIT
    FOR BIT in every state tensor, log and returned batch.  `dr`: with every domain-randomisation hook on (friction buckets, added mass,
    CoM shift, 5 substeps of action lag, a push every 4th step) -- per-env parameters and the lag ring are addressed per half-wave."""
    def mk():
        d, k, _ = make_desc(task, N)
        if dr:
            d.rand_friction, d.friction_lo, d.friction_hi = 1, 0.3, 1.5
            d.rand_base_mass, d.added_mass_lo, d.added_mass_hi = 1, -1.0, 3.0
            d.rand_com = 1
            for c, (lo, hi) in enumerate(((-0.05, 0.15), (-0.1, 0.1), (-0.05, 0.05))):
                d.com_lo[c], d.com_hi[c] = lo, hi
            d.lag_timesteps, d.push_interval, d.max_push_vel_xy = 5, 4, 1.0
        return d, k
    monkeypatch.setenv("MQE_ENVS_PER_WAVE", "1")
    d1, k1 = mk()
    e1 = hip_engine(d1, k1)
    monkeypatch.setenv("MQE_ENVS_PER_WAVE", "2")
    d2, k2 = mk()
    e2 = hip_engine(d2, k2)
    monkeypatch.delenv("MQE_ENVS_PER_WAVE")
    e1.reset_all(); e2.reset_all()
    g = torch.Generator().manual_seed(3)
    Aw = e1.tensor(abi.T_WRAPPER_OBS).shape[1]
    A = d1.num_agents
    kinds = (abi.T_ROOT_STATE, abi.T_DOF_STATE, abi.T_TORQUES, abi.T_CONTACT_FORCE, abi.T_ACT_HIST, abi.T_SUBSTEP_TORQUES, abi.T_SUBSTEP_DOF_VEL,
             abi.T_SUBSTEP_EXCEED_DOF_POS_LIMITS, abi.T_RESET_BUF, abi.T_CONTACT_OVERFLOW, abi.T_OBS_BAG, abi.T_WRAPPER_OBS, abi.T_WRAPPER_REWARD)
    resets = 0
    for t in range(15):
        a = (torch.rand(N, Aw, 3, generator=g) * 2 - 1).cuda().contiguous()
        if t == 4 and A == 2:            # push the two robots of every third env into each other: robot-robot and self contacts
            for e in (e1, e2):
                r = e.tensor(abi.T_ROOT_STATE)
                r[::3, 1, :3] = r[::3, 0, :3] + torch.tensor([0.05, 0.22, 0.0], device="cuda")
        if t == 8:                       # and fold some legs far beyond their stops: joint-limit impulses in one half-wave only
            for e in (e1, e2):
                q = e.tensor(abi.T_DOF_STATE)
                q[1::4, 2, 0] = -3.2; q[1::4, 2, 1] = -20.0
        e1.step(a); e2.step(a)
        torch.cuda.synchronize()
        for kind in kinds:
            x1, x2 = e1.tensor(kind), e2.tensor(kind)
            assert torch.equal(x1.view(torch.uint8) if x1.dtype != torch.float32 else x1.view(torch.int32), x2.view(torch.uint8) if x2.dtype != torch.float32 else x2.view(torch.int32)), (t, kind)
        resets += int(e1.tensor(abi.T_RESET_BUF).sum())
    assert torch.isfinite(e2.tensor(abi.T_ROOT_STATE)).all()
    if N >= 37 and A == 2:
        assert resets > 0, "the rollout must include resets"


@pytest.mark.parametrize("N,split", [(1, "0"), (1, "1"), (3, "1"), (37, "0"), (37, "1")])
def test_tiny_and_ragged_batches(monkeypatch, N, split):
    """batch sizes far below a tile of any kernel (1 env = 2 robots: 2 of the 128 GEMM rows, 2 of the tail's 32, one physics
    wave) and not a multiple of anything, with either layer-0 kernel: 6 fused steps against the oracle"""
    monkeypatch.setenv("MQE_GEMM_SPLIT", split)
    eh, eo, d = _pair("go1gate", N)
    monkeypatch.delenv("MQE_GEMM_SPLIT")
    eh.reset_all(); eo.reset_all()
    g = torch.Generator().manual_seed(31 + N)
    for t in range(6):
        a = torch.rand(N, 2, 3, generator=g) * 2 - 1
        eh.step(a.cuda().contiguous()); eo.step(a)
        torch.cuda.synchronize()
        if t == 0:
            close(eh.tensor(abi.T_ACTIONS), eo.tensor(abi.T_ACTIONS), atol=5e-5, what="policy actions step 0")
            close(eh.tensor(abi.T_WRAPPER_OBS), eo.tensor(abi.T_WRAPPER_OBS), atol=2e-4, what="wrapper obs step 0")
    close(eh.tensor(abi.T_ROOT_STATE)[..., :3], eo.tensor(abi.T_ROOT_STATE)[..., :3], atol=2e-3, what="root position after 6 steps")
    assert (eh.tensor(abi.T_RESET_BUF).cpu() == eo.tensor(abi.T_RESET_BUF)).all()


def test_domain_randomisation_matches_oracle():
    """every hook of tests/test_domain_rand.py switched on at once (friction buckets, added mass, CoM shift, 6 substeps of action
    lag, a push every 3rd step): same draws in both engines (hash RNG keyed by the global env id), fused rollout within the
    bounds of test_fused_rollout_matches_oracle, pushed velocities identical."""
    N = 32

    def mk():
        d, k, _ = make_desc("go1gate", N)
        d.rand_friction, d.friction_lo, d.friction_hi = 1, 0.3, 1.5
        d.rand_base_mass, d.added_mass_lo, d.added_mass_hi = 1, -1.0, 3.0
        d.rand_com = 1
        for c, (lo, hi) in enumerate(((-0.05, 0.15), (-0.1, 0.1), (-0.05, 0.05))):
            d.com_lo[c], d.com_hi[c] = lo, hi
        d.lag_timesteps, d.push_interval, d.max_push_vel_xy = 6, 3, 1.0
        return d, k
    eh, eo = hip_engine(*mk()), oracle_engine(*mk())
    assert torch.equal(eh.tensor(abi.T_DOMAIN_PARAMS).cpu(), eo.tensor(abi.T_DOMAIN_PARAMS))
    assert eo.tensor(abi.T_DOMAIN_PARAMS)[:, 1].std() > 0.5
    eh.reset_all(); eo.reset_all()
    g = torch.Generator().manual_seed(21)
    dev_pos, mism = [], 0
    for t in range(1, 13):
        a = torch.rand(N, 2, 3, generator=g) * 2 - 1
        eh.step(a.cuda().contiguous()); eo.step(a)
        torch.cuda.synchronize()
        rh, ro = eh.tensor(abi.T_ROOT_STATE).cpu(), eo.tensor(abi.T_ROOT_STATE)
        dev_pos.append((rh[..., :3] - ro[..., :3]).abs().max(dim=-1).values.flatten())
        mism += int((eh.tensor(abi.T_RESET_BUF).cpu() != eo.tensor(abi.T_RESET_BUF)).sum())
        if t == 1:
            close(eh.tensor(abi.T_SUBSTEP_TORQUES), eo.tensor(abi.T_SUBSTEP_TORQUES), atol=2e-2, rtol=1e-3, what="substep torques under lag")
        if t % 3 == 0:
            keep = (eo.tensor(abi.T_RESET_BUF) == 0) & (eh.tensor(abi.T_RESET_BUF).cpu() == 0)
            assert torch.equal(rh[keep][:, :, 7:9], ro[keep][:, :, 7:9]), "pushed base velocities"
            assert ro[keep][:, :, 7:9].abs().max() <= 1.0 and ro[keep][:, :, 7:9].abs().mean() > 0.3
    dev = torch.stack(dev_pos)
    assert torch.isfinite(dev).all()
    assert dev[4].median() < 1e-4 and dev[-1].median() < 5e-3, (dev[4].median(), dev[-1].median())
    assert mism <= 2, f"{mism} reset-flag mismatches"


def test_self_contacts_match_oracle(solver):
    """asset.self_collisions = 0: contacts between two links of one robot (both contact sides on the same actor: one
    Jacobian row over one set of dofs, coupling blocks with all four side pairings).  Robots in free fall with crossed /
    folded legs, perturbed per env: identical contact lists, then 40 substeps tracked in joint space."""
    N = 16
    eh, eo, d = _pair("go1gate", N)
    assert d.self_collision == 1 and d.robot.n_self_pairs == 144        # (feature point, primitive) candidates of the model file
    eh.reset_all(); eo.reset_all()
    torch.cuda.synchronize()
    ro, do = eo.tensor(abi.T_ROOT_STATE), eo.tensor(abi.T_DOF_STATE)
    g = torch.Generator().manual_seed(9)
    crossed = torch.tensor([-0.6, 0.8, -2.2, 0.6, 0.8, -2.2, 0.1, 1.0, -1.5, -0.1, 1.0, -1.5])       # front feet 23 mm into each other
    folded = torch.tensor([-0.8, 0.8, -1.5, 0.8, 0.8, -1.5, -0.6, 1.0, -2.2, 0.6, 1.0, -2.2])        # front thighs + hind feet
    do[:, :12, 0] = crossed + (torch.rand(N, 12, generator=g) - 0.5) * 0.06
    do[:, 12:24, 0] = folded + (torch.rand(N, 12, generator=g) - 0.5) * 0.06
    do[..., 1] = (torch.rand(do[..., 1].shape, generator=g) - 0.5) * 0.5
    ro[:, :, 2] = 2.0; ro[:, 1, 1] += 1.0                                  # free fall, far from each other
    ro[:, :, 7:] = 0
    eh.tensor(abi.T_ROOT_STATE).copy_(ro.cuda()); eh.tensor(abi.T_DOF_STATE).copy_(do.cuda())
    eh.tensor(abi.T_TORQUES).zero_(); eo.tensor(abi.T_TORQUES).zero_()
    nself = 0
    for k in range(40):
        if k in (0, 10, 25):
            torch.cuda.synchronize()
            close(eh.tensor(abi.T_DOF_STATE)[..., 0], eo.tensor(abi.T_DOF_STATE)[..., 0], atol=2e-4, what=f"joint angles at substep {k}")
            eh.tensor(abi.T_ROOT_STATE).copy_(eo.tensor(abi.T_ROOT_STATE).cuda()); eh.tensor(abi.T_DOF_STATE).copy_(eo.tensor(abi.T_DOF_STATE).cuda())
            for env in range(N):
                _, ch = eh.debug_dynamics(env, 0)
                _, _, co = eo.debug_dynamics(env, 0)
                assert ch.shape == co.shape and (ch[:, :4] == co[:, :4]).all()
                close(ch[:, 4:], co[:, 4:], atol=1e-4, what="contact separation / normal")     # centres 1-2 cm apart at z = 2 m: the unit normal carries ~1e-5 of f32 rounding
                nself += int((co[:, 0] == co[:, 2]).sum())
        eh.simulate(); eo.simulate()
    torch.cuda.synchronize()
    assert nself >= 2 * N, "test must exercise self-contacts on both robots"
    close(eh.tensor(abi.T_CONTACT_FORCE), eo.tensor(abi.T_CONTACT_FORCE), atol=0.5, rtol=2e-2, what="net contact force")
    assert torch.isfinite(eh.tensor(abi.T_ROOT_STATE)).all() and torch.isfinite(eh.tensor(abi.T_DOF_STATE)).all()


def test_box_contacts_match_oracle(solver):
    """go1pushbox: robots dropped onto / against the free box (sphere vs oriented box narrow phase, box corners vs the
    ground): identical contact lists from identical states, then 60 substeps tracked by the box pose."""
    N = 16
    eh, eo, d = _pair("go1pushbox", N)
    eh.reset_all(); eo.reset_all()
    torch.cuda.synchronize()
    ro, do = eo.tensor(abi.T_ROOT_STATE), eo.tensor(abi.T_DOF_STATE)
    g = torch.Generator().manual_seed(4)
    A = 2
    box = ro[:, A, :3].clone()
    ro[:, A, 2] = d.ground_z + d.npc_box_half[2] + 0.002
    # robot 0 on the lid, robot 1 leaning against a side face
    ro[:, 0, :2] = box[:, :2] + (torch.rand(N, 2, generator=g) - 0.5) * 0.3
    ro[:, 0, 2] = d.ground_z + 2 * d.npc_box_half[2] + 0.34
    ro[:, 1, 0] = box[:, 0] - d.npc_box_half[0] - 0.30 + (torch.rand(N, generator=g) - 0.5) * 0.1
    ro[:, 1, 1] = box[:, 1] + (torch.rand(N, generator=g) - 0.5) * 0.4
    ro[:, 1, 2] = 0.34
    ro[:, :, 7:] = 0
    ro[:, 1, 7] = 0.8                                               # walking into the box
    do[..., 1] = 0
    eh.tensor(abi.T_ROOT_STATE).copy_(ro.cuda()); eh.tensor(abi.T_DOF_STATE).copy_(do.cuda())
    eh.tensor(abi.T_TORQUES).zero_(); eo.tensor(abi.T_TORQUES).zero_()
    saw_box = False
    for k in range(60):
        if k in (0, 20, 45):
            torch.cuda.synchronize()
            # (limp robots tumbling over the box for 20-25 substeps without re-synchronisation: the temporal solver, which turns every
            # separation into a velocity within dt / 4, lets the two engines' rounding differences grow ~3 x faster; measured 1.0e-3)
            close(eh.tensor(abi.T_ROOT_STATE)[:, A, :7], eo.tensor(abi.T_ROOT_STATE)[:, A, :7], atol=5e-4 if solver == "pgs" else 3e-3, what=f"box pose at substep {k}")
            eh.tensor(abi.T_ROOT_STATE).copy_(eo.tensor(abi.T_ROOT_STATE).cuda()); eh.tensor(abi.T_DOF_STATE).copy_(eo.tensor(abi.T_DOF_STATE).cuda())
            for env in range(N):
                _, ch = eh.debug_dynamics(env, 0)
                _, _, co = eo.debug_dynamics(env, 0)
                assert ch.shape == co.shape and (ch[:, :4] == co[:, :4]).all()
                close(ch[:, 4:], co[:, 4:], atol=2e-5, what="contact separation / normal")
                saw_box |= bool((co[:, 2] == A).any())
        eh.simulate(); eo.simulate()
    torch.cuda.synchronize()
    assert saw_box, "test must exercise robot-box contacts"
    assert torch.isfinite(eh.tensor(abi.T_ROOT_STATE)).all()


def test_revolving_door_matches_oracle(solver):
    """go1revolvingdoor: robots in the sweep of the spinning door (vertical hinge): identical contact lists from identical
    states, hinge angle tracked over 70 substeps."""
    N = 16
    eh, eo, d = _pair("go1revolvingdoor", N)
    eh.reset_all(); eo.reset_all()
    torch.cuda.synchronize()
    ro, do = eo.tensor(abi.T_ROOT_STATE), eo.tensor(abi.T_DOF_STATE)
    g = torch.Generator().manual_seed(9)
    A = 2
    hinge = ro[:, A, :3].clone()
    do[:, 24, 0] = (torch.rand(N, generator=g) - 0.5) * 1.0
    do[:, 24, 1] = 1.0 + torch.rand(N, generator=g) * 2.0
    for r, sy in ((0, 1.0), (1, -1.0)):
        ro[:, r, 0] = hinge[:, 0] + (torch.rand(N, generator=g) - 0.5) * 0.9
        ro[:, r, 1] = hinge[:, 1] + sy * (0.35 + torch.rand(N, generator=g) * 0.5)
        ro[:, r, 2] = 0.33
    ro[:, :, 7:] = 0
    do[:, :24, 1] = 0
    eh.tensor(abi.T_ROOT_STATE).copy_(ro.cuda()); eh.tensor(abi.T_DOF_STATE).copy_(do.cuda())
    eh.tensor(abi.T_TORQUES).zero_(); eo.tensor(abi.T_TORQUES).zero_()
    saw_door = False
    for k in range(70):
        if k in (0, 30, 60):
            torch.cuda.synchronize()
            close(eh.tensor(abi.T_DOF_STATE)[:, 24, 0], eo.tensor(abi.T_DOF_STATE)[:, 24, 0], atol=5e-4 if solver == "pgs" else 3e-3, what=f"door angle at substep {k}")   # (temporal solver: measured 1.1e-3 after 30 free substeps; see the box test)
            eh.tensor(abi.T_ROOT_STATE).copy_(eo.tensor(abi.T_ROOT_STATE).cuda()); eh.tensor(abi.T_DOF_STATE).copy_(eo.tensor(abi.T_DOF_STATE).cuda())
            for env in range(N):
                _, ch = eh.debug_dynamics(env, 0)
                _, _, co = eo.debug_dynamics(env, 0)
                assert ch.shape == co.shape and (ch[:, :4] == co[:, :4]).all()
                saw_door |= bool((co[:, 2] == A).any())
        eh.simulate(); eo.simulate()
    torch.cuda.synchronize()
    assert saw_door, "test must exercise door contacts"
    assert torch.isfinite(eh.tensor(abi.T_ROOT_STATE)).all() and torch.isfinite(eh.tensor(abi.T_DOF_STATE)).all()


def test_tug_slider_matches_oracle(solver):
    """go1tug: robots at the rim of the sliding disc (prismatic joint, sphere vs upright cylinder): identical contact lists
    from identical states, slider position tracked over 70 substeps."""
    N = 16
    eh, eo, d = _pair("go1tug", N)
    eh.reset_all(); eo.reset_all()
    torch.cuda.synchronize()
    ro, do = eo.tensor(abi.T_ROOT_STATE), eo.tensor(abi.T_DOF_STATE)
    g = torch.Generator().manual_seed(12)
    A = 2
    hinge = ro[:, A, :3].clone()
    do[:, 24, 0] = (torch.rand(N, generator=g) - 0.5) * 0.6
    do[:, 24, 1] = (torch.rand(N, generator=g) - 0.5) * 1.6
    R = d.seesaw_plank_half[0]
    for r, sy in ((0, 1.0), (1, -1.0)):
        ro[:, r, 0] = hinge[:, 0] + (torch.rand(N, generator=g) - 0.5) * 1.0
        ro[:, r, 1] = hinge[:, 1] + do[:, 24, 0] + sy * (R + 0.25 + torch.rand(N, generator=g) * 0.15)
        ro[:, r, 2] = 0.32
        ro[:, r, 3:7] = torch.tensor([0.0, 0.0, 0.0, 1.0])
    ro[:, :, 7:] = 0
    do[:, :24, 1] = 0
    eh.tensor(abi.T_ROOT_STATE).copy_(ro.cuda()); eh.tensor(abi.T_DOF_STATE).copy_(do.cuda())
    eh.tensor(abi.T_TORQUES).zero_(); eo.tensor(abi.T_TORQUES).zero_()
    saw = False
    for k in range(70):
        if k in (0, 30, 60):
            torch.cuda.synchronize()
            close(eh.tensor(abi.T_DOF_STATE)[:, 24, 0], eo.tensor(abi.T_DOF_STATE)[:, 24, 0], atol=5e-4, what=f"slider position at substep {k}")
            eh.tensor(abi.T_ROOT_STATE).copy_(eo.tensor(abi.T_ROOT_STATE).cuda()); eh.tensor(abi.T_DOF_STATE).copy_(eo.tensor(abi.T_DOF_STATE).cuda())
            for env in range(N):
                _, ch = eh.debug_dynamics(env, 0)
                _, _, co = eo.debug_dynamics(env, 0)
                assert ch.shape == co.shape and (ch[:, :4] == co[:, :4]).all()
                saw |= bool((co[:, 2] == A).any())
        eh.simulate(); eo.simulate()
    torch.cuda.synchronize()
    assert saw, "test must exercise disc contacts"
    assert torch.isfinite(eh.tensor(abi.T_ROOT_STATE)).all() and torch.isfinite(eh.tensor(abi.T_DOF_STATE)).all()


@pytest.mark.parametrize("ctrl", ["P", "V", "T"])
def test_low_level_control_matches_reference_and_oracle(ctrl, solver):
    """Control types P / V / T: (1) the reference's trace through the unfused entry points; (2) 12 fused mqe_step_joint calls
    (PD / torque law inside k_substeps) against the oracle from the seeded reset distribution."""
    from replay import replay_joint
    assert replay_joint(ctrl, hip_engine)
    N = 64
    d1, k1, _ = make_desc("go1gate", N); d2, k2, _ = make_desc("go1gate", N)
    d1.control_type = d2.control_type = abi.CTRL[ctrl]
    eh, eo = hip_engine(d1, k1), oracle_engine(d2, k2)
    eh.reset_all(); eo.reset_all()
    g = torch.Generator().manual_seed(3)
    dev = []
    together = torch.ones(N, dtype=torch.bool)          # envs whose two trajectories still agree to float noise (10 um)
    early = 0
    for t in range(12):
        a = (torch.rand(N * 2, 12, generator=g) * 2 - 1) * (1.0 if ctrl != "T" else 8.0)
        eh.step_joint(a.cuda().contiguous()); eo.step_joint(a)
        torch.cuda.synchronize()
        if t == 0:
            close(eh.tensor(abi.T_SUBSTEP_TORQUES)[:, 0], eo.tensor(abi.T_SUBSTEP_TORQUES)[:, 0], atol=2e-4, rtol=1e-4, what="first substep torques")
        # random joint-space actions throw the robots over within a few steps; a fall is chaotic, so a flag may only differ in an
        # env whose trajectories had already separated -- in one that is still on the common trajectory only when a termination
        # test sits exactly on its threshold (base contact force 1 N): at most one such event in the whole rollout
        flags_h, flags_o = eh.tensor(abi.T_RESET_BUF).cpu() != 0, eo.tensor(abi.T_RESET_BUF) != 0
        early += int(((flags_h != flags_o) & together).sum())
        d_now = (eh.tensor(abi.T_ROOT_STATE).cpu()[..., :3] - eo.tensor(abi.T_ROOT_STATE)[..., :3]).abs().max(dim=-1).values
        dev.append(d_now.flatten())
        together &= (d_now.max(dim=-1).values < 1e-5) & ~(flags_h | flags_o)
    dev = torch.stack(dev)
    assert torch.isfinite(dev).all() and dev[3].median() < 1e-4 and dev[-1].median() < 5e-3, (dev[3].median(), dev[-1].median())
    assert early <= 1, f"{early} reset flags differ in envs that had not diverged"
    assert int((eh.tensor(abi.T_RESET_BUF).cpu() != eo.tensor(abi.T_RESET_BUF)).sum()) <= N // 8


def test_seesaw_plank_matches_oracle(solver):
    """go1seesaw: robots dropped onto the plank / the platform / next to the column: contact lists identical, then
    110 substeps of coupled robot-plank dynamics (hinge angle tracked to 2e-4 rad)."""
    N = 24
    eh, eo, d = _pair("go1seesaw", N)
    eh.reset_all(); eo.reset_all()
    torch.cuda.synchronize()
    ro, do = eo.tensor(abi.T_ROOT_STATE), eo.tensor(abi.T_DOF_STATE)
    g = torch.Generator().manual_seed(2)
    base = ro[:, 2, :3].clone()                                    # seesaw platform centre
    xs = torch.rand(N, 2, generator=g) * 4.5 - 4.2                  # along the plank ... up to the platform
    ro[:, :2, 0] = base[:, None, 0] + xs
    ro[:, :2, 1] = base[:, None, 1] + (torch.rand(N, 2, generator=g) - 0.5) * 0.9
    ro[:, :2, 2] = 1.35
    ro[:, :2, 7:] = 0
    do[:, :24, 1] = 0
    do[:, 24, 0] = (torch.rand(N, generator=g) - 0.5) * 0.3
    theta0 = do[:, 24, 0].clone()
    eh.tensor(abi.T_ROOT_STATE).copy_(ro.cuda()); eh.tensor(abi.T_DOF_STATE).copy_(do.cuda())
    eh.tensor(abi.T_TORQUES).zero_(); eo.tensor(abi.T_TORQUES).zero_()
    saw_plank = False
    for k in range(110):
        if k in (70, 100):
            # contact lists are compared from IDENTICAL states (a sphere sitting exactly at the contact margin would
            # otherwise flip on a 1e-7 state difference): hinge angle first, then re-synchronise
            torch.cuda.synchronize()
            close(eh.tensor(abi.T_DOF_STATE)[:, 24, 0], eo.tensor(abi.T_DOF_STATE)[:, 24, 0], atol=2e-4, what=f"hinge angle at substep {k}")
            eh.tensor(abi.T_ROOT_STATE).copy_(eo.tensor(abi.T_ROOT_STATE).cuda()); eh.tensor(abi.T_DOF_STATE).copy_(eo.tensor(abi.T_DOF_STATE).cuda())
            for env in range(N):
                _, ch = eh.debug_dynamics(env, 0)
                _, _, co = eo.debug_dynamics(env, 0)
                assert ch.shape == co.shape and (ch[:, :4] == co[:, :4]).all()
                saw_plank |= bool((co[:, 2] == 2).any())
        eh.simulate(); eo.simulate()
    torch.cuda.synchronize()
    assert saw_plank, "test must exercise plank contacts"
    th_h, th_o = eh.tensor(abi.T_DOF_STATE)[:, 24, 0].cpu(), eo.tensor(abi.T_DOF_STATE)[:, 24, 0]
    assert (th_o - theta0).abs().max() > 0.01, "plank must have moved"
    close(th_h, th_o, atol=2e-4, what="hinge angle after 110 substeps")
    dz = (eh.tensor(abi.T_ROOT_STATE)[:, :2, 2].cpu() - eo.tensor(abi.T_ROOT_STATE)[:, :2, 2]).abs()
    assert dz.median() < 1e-4 and dz.max() < 5e-2


def test_full_size_invariants():
    """BASELINE config 1 (go1gate, 4096 envs x 2 agents): properties that do not need the oracle."""
    d, k, ctx = make_desc("go1gate", 4096)
    e = hip_engine(d, k)
    e.reset_all()
    g = torch.Generator(device="cuda").manual_seed(1234)
    for t in range(30):
        a = torch.rand(4096, 2, 3, device="cuda", generator=g) * 2 - 1
        e.step(a)
    torch.cuda.synchronize()
    root = e.tensor(abi.T_ROOT_STATE)
    assert torch.isfinite(root).all() and torch.isfinite(e.tensor(abi.T_DOF_STATE)).all()
    qn = root[..., 3:7].norm(dim=-1)
    assert (qn - 1).abs().max() < 1e-4                                # unit quaternions
    z = root[..., 2]
    assert z.min() > 0.0 and z.max() < 1.0                             # nobody fell through the ground / flew away
    obs = e.tensor(abi.T_WRAPPER_OBS)
    assert obs.shape == (4096, 2, 16) and torch.isfinite(obs).all()
    assert (obs[:, 0, 0] == 1).all() and (obs[:, 1, 1] == 1).all()     # one-hot ids
    # joint limits hold (to solver tolerance)
    q = e.tensor(abi.T_DOF_STATE)[..., 0].reshape(4096, 2, 12)
    lo = torch.tensor([d.robot.dof_lower[j] for j in range(12)], device="cuda")
    hi = torch.tensor([d.robot.dof_upper[j] for j in range(12)], device="cuda")
    assert (q > lo - 0.05).all() and (q < hi + 0.05).all()


ALL_TASKS = ["go1plane", "go1gate", "go1sheep-easy", "go1sheep-hard", "go1football-defender", "go1football-1vs1", "go1football-2vs2",
             "go1seesaw", "go1pushbox", "go1tug", "go1wrestling", "go1revolvingdoor", "go1bridge"]


@pytest.mark.parametrize("task", ALL_TASKS)
def test_every_task_survives_a_long_random_rollout(task):
    """All 13 registered tasks, 256 envs, 250 fused steps (5 s of simulated time, several episodes for the short ones) with
    random commands: every state stays finite and bounded, quaternions stay unit, objects stay in the arena."""
    N = 256
    d, k, ctx = make_desc(task, N)
    e = hip_engine(d, k)
    e.reset_all()
    A, P = d.num_agents, d.num_npcs
    Aw = e.tensor(abi.T_WRAPPER_OBS).shape[1]
    g = torch.Generator(device="cuda").manual_seed(7)
    worst = 0.0
    for t in range(250):
        e.step(torch.rand(N, Aw, 3, device="cuda", generator=g) * 3 - 1.5)
        if t % 50 == 49:
            root, dof = e.tensor(abi.T_ROOT_STATE), e.tensor(abi.T_DOF_STATE)
            assert torch.isfinite(root).all() and torch.isfinite(dof).all() and torch.isfinite(e.tensor(abi.T_WRAPPER_OBS)).all()
            assert torch.isfinite(e.tensor(abi.T_WRAPPER_REWARD)).all() and torch.isfinite(e.tensor(abi.T_CONTACT_FORCE)).all()
            rob = root[:, :A]
            assert ((rob[..., 3:7].norm(dim=-1) - 1).abs() < 1e-3).all()
            assert rob[..., 2].min() > -0.05 and rob[..., 2].max() < 3.0
            assert rob[..., 7:10].abs().max() < 30.0 and dof[:, :12 * A, 1].abs().max() < 200.0
            worst = max(worst, float(rob[..., 7:10].abs().max()))
            eo = torch.as_tensor(ctx["env_origins"], device="cuda")[:, None, :2]
            assert (root[:, :, :2] - eo).abs().max() < 40.0                      # nothing left the arena
    assert int(e.tensor(abi.T_RESET_COUNT).min()) >= 1
    e.close()


def test_general_shape_policy_tail(tmp_path):
    """Row D: a body file whose depth / widths are not the stand-in's (2102-128-64-12, TorchScript written and read back through
    load_body) takes the general k_gemm_f32 chain on the GPU: joint targets == the oracle == the TorchScript module (plain torch
    fp32) on identical histories, for both layer-0 kernels."""
    import os
    from test_models_oracle import scripted_body, torch_adaptation
    net, Ws, bs = scripted_body(tmp_path)
    ada = torch_adaptation()
    N = 48
    for split in ("1", "0"):
        os.environ["MQE_GEMM_SPLIT"] = split
        try:
            d1, k1, _ = make_desc("go1gate", N, body=(Ws, bs))
            d2, k2, _ = make_desc("go1gate", N, body=(Ws, bs))
            eh, eo = hip_engine(d1, k1), oracle_engine(d2, k2)
        finally:
            del os.environ["MQE_GEMM_SPLIT"]
        eh.reset_all(); eo.reset_all()
        g = torch.Generator().manual_seed(5)
        for t in range(8):
            a = torch.rand(N, 2, 3, generator=g) * 2 - 1
            eh.step(a.cuda().contiguous()); eo.step(a)
            for k in (abi.T_ROOT_STATE, abi.T_DOF_STATE):
                eh.tensor(k).copy_(eo.tensor(k).cuda())
            if t in (0, 7):
                close(eh.tensor(abi.T_ACTIONS), eo.tensor(abi.T_ACTIONS), atol=5e-5, what=f"policy actions step {t} (split {split})")
        hist = eh.history().cpu()
        with torch.no_grad():
            want = net(torch.cat([hist, ada(hist)], dim=1))
        close(eh.tensor(abi.T_LAST_LOCO_ACTION), want, atol=1e-4, rtol=1e-4, what="policy output vs the TorchScript module")
        eh.close()


def test_unfused_tail_on_the_default_shape(monkeypatch):
    """MQE_NO_FUSED_TAIL=1: the reference network shapes through the general chain (five k_gemm_f32 launches + k_body_l0_finish +
    k_post_policy) instead of k_policy_tail: same joint targets as the oracle, and as the fused tail to f32 rounding"""
    N = 64
    monkeypatch.setenv("MQE_NO_FUSED_TAIL", "1")
    eh, eo, d = _pair("go1gate", N)
    monkeypatch.delenv("MQE_NO_FUSED_TAIL")
    d3, k3, _ = make_desc("go1gate", N)
    ef = hip_engine(d3, k3)
    for e in (eh, eo, ef):
        e.reset_all()
    g = torch.Generator().manual_seed(6)
    for t in range(6):
        a = torch.rand(N, 2, 3, generator=g) * 2 - 1
        eh.step(a.cuda().contiguous()); eo.step(a); ef.step(a.cuda().contiguous())
        for k in (abi.T_ROOT_STATE, abi.T_DOF_STATE):
            eh.tensor(k).copy_(eo.tensor(k).cuda()); ef.tensor(k).copy_(eo.tensor(k).cuda())
        close(eh.tensor(abi.T_ACTIONS), eo.tensor(abi.T_ACTIONS), atol=5e-5, what=f"unfused tail vs oracle, step {t}")
        close(eh.tensor(abi.T_ACTIONS), ef.tensor(abi.T_ACTIONS), atol=5e-5, what=f"unfused vs fused tail, step {t}")


def test_out_of_range_observations_do_not_poison_the_policy():
    """The split-f16 operands carry 64 x: beyond |x| = 1023 the value saturates (clamped to the f16 range in split2, never inf / NaN).
    A blown-up joint velocity (1e4 rad/s in the history) must leave the policy output finite and bounded, on the other robots
    untouched, and the exact-f32 path must agree with the oracle there."""
    import os
    N = 32
    outs = {}
    for split in ("1", "0"):
        os.environ["MQE_GEMM_SPLIT"] = split
        try:
            d1, k1, _ = make_desc("go1gate", N)
            e = hip_engine(d1, k1)
        finally:
            del os.environ["MQE_GEMM_SPLIT"]
        e.reset_all()
        a = torch.zeros(N, 2, 3, device="cuda")
        e.step(a)
        dof = e.tensor(abi.T_DOF_STATE)
        dof[3, :12, 1] = 4.0e5                       # robot 0 of env 3: obs dof_vel = 0.05 * 4e5 = 2e4 >> 1023
        e.post_physics_step()                        # refreshes the observation bag from the poisoned state
        dof[3, :12, 1] = 0.0
        e.step(a)
        torch.cuda.synchronize()
        act = e.tensor(abi.T_LAST_LOCO_ACTION).cpu()
        assert torch.isfinite(act).all() and torch.isfinite(e.tensor(abi.T_ACTIONS)).all()
        assert act.abs().max() < 1e6
        outs[split] = act
        e.close()
    others = [i for i in range(2 * N) if i != 6]
    assert (outs["1"][others] - outs["0"][others]).abs().max() < 5e-5          # nobody else noticed
    assert outs["0"][6].abs().max() > 10 * outs["0"][others].abs().max()        # the exact path sees the full 2e4


@pytest.mark.parametrize("task,N,zs", [("go1gate", 64, 0.06), ("go1football-defender", 16, 0.12)])
def test_perlin_terrain_matches_oracle(task, N, zs):
    """heightfield ground (Perlin relief as a second terrain map, no slab): contact lists from identical perturbed
    states are identical incl. the tilted ground normals, a substep agrees, and a 20-step fused rollout stays within the bounds of
    the flat scenes"""
    from helpers import perlin_terrain
    tc = perlin_terrain(task, zScale=zs)
    d1, k1, _ = make_desc(task, N, terrain_cfg=tc)
    d2, k2, _ = make_desc(task, N, terrain_cfg=tc)
    assert d1.ground_z == 0.0 and bool(d1.ground_height)
    eh, eo = hip_engine(d1, k1), oracle_engine(d2, k2)
    _randomize(eh, eo, 5, drop=0.08)
    tilted = 0
    for env in (0, N // 2, N - 1):
        mh, ch = eh.debug_dynamics(env, 0)
        _, mo, co = eo.debug_dynamics(env, 0)
        assert ch.shape == co.shape and (ch[:, :4] == co[:, :4]).all()
        close(ch[:, 4:], co[:, 4:], atol=3e-5, what="contact separation / normal on the relief")
        tilted += int(((co[:, 2] < 0) & (np.abs(co[:, 7]) < 0.999) & (co[:, 7] > 0.3)).sum())
    assert tilted > 0, "the test must see ground contacts with tilted normals"
    eh.simulate(); eo.simulate()
    torch.cuda.synchronize()
    close(eh.tensor(abi.T_ROOT_STATE)[..., :7], eo.tensor(abi.T_ROOT_STATE)[..., :7], atol=2e-5, what="root pose after a substep")
    close(eh.tensor(abi.T_ROOT_STATE)[..., 7:], eo.tensor(abi.T_ROOT_STATE)[..., 7:], atol=2e-3, rtol=1e-3, what="root vel after a substep")
    close(eh.tensor(abi.T_CONTACT_FORCE), eo.tensor(abi.T_CONTACT_FORCE), atol=0.5, rtol=2e-2, what="net contact force")
    eh.reset_all(); eo.reset_all()
    g = torch.Generator().manual_seed(13)
    Aw = eo.tensor(abi.T_WRAPPER_OBS).shape[1]
    dev, mism = [], 0
    for t in range(20):
        a = torch.rand(N, Aw, 3, generator=g) * 2 - 1
        eh.step(a.cuda().contiguous()); eo.step(a)
        torch.cuda.synchronize()
        dev.append((eh.tensor(abi.T_ROOT_STATE).cpu()[..., :3] - eo.tensor(abi.T_ROOT_STATE)[..., :3]).abs().max(dim=-1).values.flatten())
        mism += int((eh.tensor(abi.T_RESET_BUF).cpu() != eo.tensor(abi.T_RESET_BUF)).sum())
    dev = torch.stack(dev)
    assert torch.isfinite(dev).all() and float(dev[4].median()) < 1e-5 and float(dev[-1].median()) < 1e-4 and float(dev[-1].quantile(0.99)) < 5e-3
    assert mism <= 1


def test_walls_of_different_heights_match_oracle():
    """a (lo, hi) wall_height: per-cell wall tops (`wall_top`, ABI v11) on the HIP engine == oracle -- contact lists of perturbed
    states incl. contacts with the top faces and edges of walls of different heights, one substep, and a short fused rollout"""
    from helpers import wall_heights_terrain
    task, N = "go1gate", 48
    np.random.seed(0)
    d1, k1, c1 = make_desc(task, N, terrain_cfg=wall_heights_terrain(task))
    np.random.seed(0)
    d2, k2, _ = make_desc(task, N, terrain_cfg=wall_heights_terrain(task))
    assert bool(d1.wall_top) and c1["terrain"].wall_top is not None
    eh, eo = hip_engine(d1, k1), oracle_engine(d2, k2)
    eh.reset_all(); eo.reset_all()
    # half of the robots onto / next to walls: base above a wall pixel at a height between the lowest and the tallest top
    t = c1["terrain"]
    hs = d1.horizontal_scale
    wi, wj = np.nonzero(t.wall)
    g = torch.Generator().manual_seed(21)
    ro = eo.tensor(abi.T_ROOT_STATE)
    sel = torch.randint(0, len(wi), (N,), generator=g)
    for env in range(0, N, 2):
        ro[env, 0, 0] = float(wi[sel[env]] * hs) + 0.03
        ro[env, 0, 1] = float(wj[sel[env]] * hs) - 0.02
        ro[env, 0, 2] = float(t.wall_top[wi[sel[env]], wj[sel[env]]]) + 0.12 + 0.1 * float(torch.rand((), generator=g))
    eh.tensor(abi.T_ROOT_STATE).copy_(ro.cuda())
    wall_contacts = 0
    for env in (0, 2, 10, N - 2):
        mh, ch = eh.debug_dynamics(env, 0)
        _, mo, co = eo.debug_dynamics(env, 0)
        assert ch.shape == co.shape and (ch[:, :4] == co[:, :4]).all()
        close(ch[:, 4:], co[:, 4:], atol=3e-5, what="contact separation / normal at walls of different heights")
        wall_contacts += len(co)
    assert wall_contacts > 0
    for _ in range(3):
        eh.simulate(); eo.simulate()
    torch.cuda.synchronize()
    close(eh.tensor(abi.T_ROOT_STATE)[..., :7], eo.tensor(abi.T_ROOT_STATE)[..., :7], atol=5e-5, what="root pose after 3 substeps")
    eh.reset_all(); eo.reset_all()
    dev = []
    for s_ in range(10):
        a = torch.rand(N, 2, 3, generator=g) * 2 - 1
        eh.step(a.cuda().contiguous()); eo.step(a)
        torch.cuda.synchronize()
        dev.append((eh.tensor(abi.T_ROOT_STATE).cpu()[..., :3] - eo.tensor(abi.T_ROOT_STATE)[..., :3]).abs().max(dim=-1).values.flatten())
        assert (eh.tensor(abi.T_RESET_BUF).cpu() == eo.tensor(abi.T_RESET_BUF)).all()
    dev = torch.stack(dev)
    assert float(dev[-1].median()) < 1e-4 and float(dev[-1].quantile(0.99)) < 1e-3


@pytest.mark.parametrize("task,N", [("go1gate", 96), ("go1sheep-hard", 24), ("go1football-defender", 40)])
def test_phase_timed_kernels_are_bit_identical_and_their_taps_ordered(monkeypatch, task, N):
    """MQE_PHASE_TIMES=1 swaps k_substeps for the instantiation whose phase taps are live (tools/dev/phase_walltimes.py: where the time of a
    full launch goes).  It is the same arithmetic: 8 fused steps agree bit for bit with the product kernel in every state tensor, and the
    wall-clock stamps every wavefront leaves are ordered -- tap after tap, substep after substep, entry before the first, exit after the last."""
    import ctypes as C
    import numpy as np
    d1, k1, _ = make_desc(task, N)
    e1 = hip_engine(d1, k1)
    monkeypatch.setenv("MQE_PHASE_TIMES", "1")
    d2, k2, _ = make_desc(task, N)
    e2 = hip_engine(d2, k2)
    monkeypatch.delenv("MQE_PHASE_TIMES")
    e1.reset_all(); e2.reset_all()
    g = torch.Generator().manual_seed(11)
    Aw = e1.tensor(abi.T_WRAPPER_OBS).shape[1]
    kinds = (abi.T_ROOT_STATE, abi.T_DOF_STATE, abi.T_TORQUES, abi.T_CONTACT_FORCE, abi.T_ACT_HIST, abi.T_SUBSTEP_TORQUES, abi.T_SUBSTEP_DOF_VEL,
             abi.T_RESET_BUF, abi.T_OBS_BAG, abi.T_WRAPPER_OBS, abi.T_WRAPPER_REWARD)
    taps = np.zeros((N, 4, 16), np.int64)
    span = np.zeros((N, 4), np.int64)
    for t in range(8):
        a = (torch.rand(N, Aw, 3, generator=g) * 2 - 1).cuda().contiguous()
        e1.step(a); e2.step(a)
        torch.cuda.synchronize()
        for kind in kinds:
            x1, x2 = e1.tensor(kind), e2.tensor(kind)
            assert torch.equal(x1.view(torch.uint8) if x1.dtype != torch.float32 else x1.view(torch.int32), x2.view(torch.uint8) if x2.dtype != torch.float32 else x2.view(torch.int32)), (t, kind)
        e2._call("debug_phase_times", C.c_void_p(taps.ctypes.data))
        e2._call("debug_wave_times", C.c_void_p(span.ctypes.data))
        flat = taps.reshape(N, 64)
        assert (np.diff(flat, axis=1) >= 0).all(), "a wavefront's taps run backwards"
        assert (flat[:, 0] >= span[:, 0]).all() and (span[:, 1] >= flat[:, -1]).all()
        us = (flat[:, -1] - flat[:, 0]) * 0.01                 # 100 MHz wall clock
        assert 20.0 < float(np.median(us)) < 2000.0, float(np.median(us))
    with pytest.raises(RuntimeError):                          # a handle created without the switch has no tap buffer
        e1._call("debug_phase_times", C.c_void_p(taps.ctypes.data))
