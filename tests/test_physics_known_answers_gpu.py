"""Closed-form answers asked of the HIP ENGINE itself (not of the oracle): the rigid-body row (H) has no reference to be pinned
against, so besides HIP == oracle (tests/test_gpu_parity.py) the product path is held directly to three answers that come from
mechanics alone -- a ball rolling down a ramp built from the relief map, a sliding ball's transition to rolling, and the weight
carried by the feet of standing robots.  The oracle versions live in tests/test_physics_known_answers.py / test_physics_oracle.py."""
import numpy as np
import pytest
import torch

from helpers import make_desc, hip_engine, perlin_terrain
from mqe.engine import abi

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("solver")]      # every known answer under both contact solvers (conftest.py)
G = 9.81


def test_hip_ball_rolls_down_a_ramp_with_the_closed_form_acceleration():
    slope = 0.1
    d, k, ctx = make_desc("go1football-defender", 4, terrain_cfg=perlin_terrain("go1football-defender", zScale=0.01))
    hs = d.horizontal_scale
    ramp = np.ascontiguousarray(np.tile((slope * np.arange(d.sdf_ny) * hs).astype(np.float32), (d.sdf_nx, 1)))
    k.append(ramp)
    d.ground_height = ramp.ctypes.data_as(abi.FP)
    e = hip_engine(d, k)
    e.reset_all()
    root = e.tensor(abi.T_ROOT_STATE)
    A, r = d.num_agents, d.npc_sphere_radius[0]
    root[:, :A, 2] += 30.0                       # the robots out of the way (falling, far above)
    root[:, A, 0] += 1.0
    th = np.arctan(slope)
    root[:, A, 2] = d.ground_z + slope * root[:, A, 1] + r / np.cos(th)
    root[:, A, 7:13] = 0
    a = G * np.sin(th) / (1.0 + d.npc_inertia / (d.npc_mass * r * r))
    T = 100
    for t in range(T):
        e.simulate()
    torch.cuda.synchronize()
    tt = T * d.dt
    v = root[:, A, 7:10].cpu().numpy()
    np.testing.assert_allclose(v[:, 1], -a * tt * np.cos(th), rtol=0.03)
    np.testing.assert_allclose(v[:, 2], -a * tt * np.sin(th), rtol=0.05)
    assert np.abs(v[:, 0]).max() < 1e-3
    np.testing.assert_allclose(root[:, A, 10].cpu().numpy(), a * tt / r, rtol=0.04)
    gap = (root[:, A, 2] - d.ground_z - slope * root[:, A, 1] - r / np.cos(th)).cpu().numpy()
    assert np.abs(gap).max() < 3e-3, gap


def test_hip_ball_slides_then_rolls_at_the_closed_form_speed():
    d, k, ctx = make_desc("go1football-defender", 4)
    e = hip_engine(d, k)
    e.reset_all()
    root = e.tensor(abi.T_ROOT_STATE)
    A, r = d.num_agents, d.npc_sphere_radius[0]
    root[:, A, 0] += 1.0                        # clear of the robots
    root[:, A, 2] = d.ground_z + r
    root[:, A, 7:13] = 0
    v0 = 2.0
    root[:, A, 8] = v0
    for t in range(200):
        e.simulate()
    torch.cuda.synchronize()
    want = v0 / (1.0 + d.npc_inertia / (d.npc_mass * r * r))
    np.testing.assert_allclose(root[:, A, 8].cpu().numpy(), want, rtol=0.02)
    np.testing.assert_allclose(root[:, A, 10].cpu().numpy(), -want / r, rtol=0.03)
    assert float((root[:, A, 2] - (d.ground_z + r)).abs().max()) < 2e-3


def test_hip_standing_robots_put_their_weight_on_their_feet():
    d, k, ctx = make_desc("go1gate", 64)
    e = hip_engine(d, k)
    e.reset_all()
    root, dof = e.tensor(abi.T_ROOT_STATE), e.tensor(abi.T_DOF_STATE)
    dof[:, :24, 0] = torch.tensor([d.default_dof_pos[j] for j in range(12)] * 2, device="cuda")
    dof[..., 1] = 0
    root[..., 7:] = 0
    a = torch.zeros(64, 2, 3, device="cuda")
    for t in range(120):
        e.step(a)
    torch.cuda.synchronize()
    cf = e.tensor(abi.T_CONTACT_FORCE).reshape(64, 2, 17, 3).cpu()
    mt = sum(d.robot.mass[b] for b in range(13))
    fz = cf[:, :, [4, 8, 12, 16], 2].sum(-1)
    assert torch.allclose(fz, torch.full((64, 2), mt * G), rtol=0.08), (fz.min(), fz.max(), mt * G)
    assert (cf[:, :, 0].norm(dim=-1) < 1e-6).all()                             # trunk does not touch
    z = e.tensor(abi.T_ROOT_STATE)[:, :, 2].cpu()
    assert ((z > 0.27) & (z < 0.34)).all()


def _hip_flight(N, seed=0, dt=None):
    """go1gate on the HIP engine, robots far above the ground and apart, random base orientations"""
    d, k, ctx = make_desc("go1gate", N)
    if dt is not None:
        d.dt = dt
    e = hip_engine(d, k)
    e.reset_all()
    root, dof = e.tensor(abi.T_ROOT_STATE), e.tensor(abi.T_DOF_STATE)
    g = torch.Generator().manual_seed(seed)
    root[:, :, 2] = 50.0
    root[:, 1, 1] += 5.0
    qt = torch.randn(N, 2, 4, generator=g)
    root[:, :, 3:7] = (qt / qt.norm(dim=-1, keepdim=True)).cuda()
    e.tensor(abi.T_TORQUES).zero_()
    return e, d, root, dof, g


def _state(root, dof, env, r):
    import rigid_ref as rr
    rc, dc = root.cpu(), dof.cpu()
    return rr.split_state(rc[env, r].numpy(), dc[env, r * 12:(r + 1) * 12, 0].numpy(), dc[env, r * 12:(r + 1) * 12, 1].numpy())


def test_hip_inverse_mass_matrix_inverts_the_kinetic_energy_hessian():
    """M^-1 as the kernel holds it (assembled from its factors by mqe_debug_dynamics) times d2 KE / dv dv from differenced poses of
    the independent float64 kinematics (tests/rigid_ref.py) is the identity -- random base orientations and joint angles"""
    import rigid_ref as rr
    m = rr.load_model()
    e, d, root, dof, g = _hip_flight(3)
    lo = torch.tensor([-0.8, -1.0, -2.6] * 4); hi = torch.tensor([0.8, 4.0, -0.95] * 4)
    for r in range(2):
        dof[:, r * 12:(r + 1) * 12, 0] = (lo + (hi - lo) * torch.rand(3, 12, generator=g)).cuda()
    dof[..., 1] = 0
    root[..., 7:] = 0
    torch.cuda.synchronize()
    for env in range(3):
        for r in range(2):
            Minv = np.asarray(e.debug_dynamics(env, r)[0], dtype=np.float64).reshape(18, 18)
            p0, R0, q, gv = _state(root, dof, env, r)
            Mfd = rr.mass_matrix_fd(m, p0, R0, q)
            np.testing.assert_allclose(Minv @ Mfd, np.eye(18), atol=3e-3)


def test_hip_momentum_budget_in_free_flight_under_joint_torques():
    """0.5 s of free flight on the HIP engine with random joint torques and every internal impulse on: linear momentum changes by
    m g t, angular momentum about the centre of mass stays, the centre of mass follows its parabola (recomputed from the poses
    by the independent kinematics; bounds = the oracle test's, widened for float32)"""
    import rigid_ref as rr
    m = rr.load_model()
    e, d, root, dof, g = _hip_flight(2, seed=3)
    dof[:, :, 1] = (torch.randn(2, 24, generator=g) * 1.0).cuda()
    root[:, :, 7:10] = torch.randn(2, 2, 3, generator=g).cuda()
    root[:, :, 10:13] = (torch.randn(2, 2, 3, generator=g) * 2.0).cuda()
    tau = e.tensor(abi.T_TORQUES)
    torch.cuda.synchronize()
    start = {(env, r): rr.momenta(m, *_state(root, dof, env, r)) for env in range(2) for r in range(2)}
    n = 100
    for kk in range(n):
        if kk % 10 == 0:
            tau.copy_(((torch.rand(2, 24, generator=g) - 0.5) * 0.5).cuda())
        e.simulate()
    torch.cuda.synchronize()
    t = n * d.dt
    for (env, r), (mt, C0, P0, L0) in start.items():
        _, C1, P1, L1 = rr.momenta(m, *_state(root, dof, env, r))
        eP = np.abs(P1 - (P0 + mt * np.array([0, 0, -G]) * t)).max() / mt
        eL = np.abs(L1 - L0).max()
        eC = np.abs(C1 - (C0 + P0 / mt * t + 0.5 * np.array([0, 0, -G]) * t * (t + d.dt))).max()
        assert eP < 5e-3 and eL < 4e-2 and eC < 3e-3, (env, r, eP, eL, eC)


def test_hip_penetration_recovery_follows_the_solvers_rule():
    """the depenetration rule of each contact solver (tests/test_physics_oracle.py::check_penetration_recovery), asked of the HIP engine"""
    from test_physics_oracle import _buried_ball, check_penetration_recovery
    d, k, ctx = make_desc("go1football-defender", 3)
    e = hip_engine(d, k)
    e.reset_all()
    z, v = _buried_ball(e, d, e.tensor(abi.T_ROOT_STATE), (0.001, 0.003, 0.06), 40)
    check_penetration_recovery(d, z, v)
