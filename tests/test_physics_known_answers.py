"""Known answers for the rigid-body row (H) that do NOT share the build's derivation: the oracle's mass
matrix against a finite-difference kinetic energy, momentum and energy budgets in free flight computed by tests/rigid_ref.py
(float64 forward kinematics from the model file, velocities by differencing poses), a torsional pendulum and a ball's
slide-to-roll transition against their closed forms, and the URDF joint velocity limit as a momentum-conserving impulse.
The HIP kernel is held to the same oracle by tests/test_gpu_parity.py (-m gpu)."""
import numpy as np
import pytest
import torch

import rigid_ref as rr
from helpers import make_desc, oracle_engine
from mqe.engine import abi

G = 9.81
pytestmark = pytest.mark.usefixtures("solver")      # every test under both contact solvers (conftest.py)


def flight(N=1, f64=True, widen=True, seed=0, gravity=None, dt=None, **kw):
    """go1gate with the robots far above the ground and apart: no contact can form; optionally without joint stops / speed limits /
    self-collision so that only the smooth dynamics act"""
    d, k, ctx = make_desc("go1gate", N, **kw)
    if widen:
        for j in range(12):
            d.robot.dof_lower[j], d.robot.dof_upper[j], d.robot.dof_vel_limit[j] = -100.0, 100.0, 0.0
        d.self_collision = 0
    if gravity is not None:
        d.gravity_z = gravity
    if dt is not None:
        d.dt = dt
    e = oracle_engine(d, k, f64=f64)
    e.reset_all()
    root, dof = e.tensor(abi.T_ROOT_STATE), e.tensor(abi.T_DOF_STATE)
    g = torch.Generator().manual_seed(seed)
    root[:, :, 2] = 50.0
    root[:, 1, 1] += 5.0
    qt = torch.randn(N, 2, 4, generator=g)
    root[:, :, 3:7] = qt / qt.norm(dim=-1, keepdim=True)
    e.tensor(abi.T_TORQUES).zero_()
    return e, d, root, dof, g


def state_of(root, dof, env, r):
    return rr.split_state(root[env, r].numpy(), dof[env, r * 12:(r + 1) * 12, 0].numpy(), dof[env, r * 12:(r + 1) * 12, 1].numpy())


def test_mass_matrix_equals_the_hessian_of_the_kinetic_energy():
    """M(q) from the oracle (per-body Jacobian sums) == d2 KE / dv dv with KE = sum 1/2 m |v_c|^2 + 1/2 w.I w evaluated from
    DIFFERENCED poses of an independent forward kinematics -- at random base orientations and joint angles"""
    m = rr.load_model()
    e, d, root, dof, g = flight(N=3)
    lo = torch.tensor([-0.8, -1.0, -2.6] * 4); hi = torch.tensor([0.8, 4.0, -0.95] * 4)
    for r in range(2):
        dof[:, r * 12:(r + 1) * 12, 0] = lo + (hi - lo) * torch.rand(3, 12, generator=g)
    for env in range(3):
        for r in range(2):
            M, _, _ = e.debug_dynamics(env, r)
            p0, R0, q, gv = state_of(root, dof, env, r)
            Mfd = rr.mass_matrix_fd(m, p0, R0, q)
            scale = np.sqrt(np.outer(np.diag(Mfd), np.diag(Mfd)))
            assert np.abs(M - Mfd).max() < 2e-5 and (np.abs(M - Mfd) / scale).max() < 2e-4, np.abs(M - Mfd).max()


def _momentum_errors(dt, torque, qd0):
    """0.5 s of free flight with random joint torques (uniform in +-torque / 2, re-drawn ten times) with every internal impulse
    switched on (joint stops, speed limits, self-contacts): largest deviation over 4 robots of the linear momentum from
    P0 + m g t, of the angular momentum about the centre of mass from L0, and of the centre of mass from its parabola"""
    m = rr.load_model()
    d, k, ctx = make_desc("go1gate", 2)
    d.dt = dt
    e = oracle_engine(d, k, f64=True)
    e.reset_all()
    root, dof = e.tensor(abi.T_ROOT_STATE), e.tensor(abi.T_DOF_STATE)
    g = torch.Generator().manual_seed(3)
    root[:, :, 2] = 50.0
    root[:, 1, 1] += 5.0
    qt = torch.randn(2, 2, 4, generator=g)
    root[:, :, 3:7] = qt / qt.norm(dim=-1, keepdim=True)
    dof[:, :, 1] = torch.randn(2, 24, generator=g) * qd0
    root[:, :, 7:10] = torch.randn(2, 2, 3, generator=g)
    root[:, :, 10:13] = torch.randn(2, 2, 3, generator=g) * 2.0
    tau = e.tensor(abi.T_TORQUES)
    start = {(env, r): rr.momenta(m, *state_of(root, dof, env, r)) for env in range(2) for r in range(2)}
    n = int(round(0.5 / dt))
    for kk in range(n):
        if kk % (n // 10) == 0:
            tau.copy_((torch.rand(2, 24, generator=g) - 0.5) * torque)
        e.simulate()
    t = n * dt
    eP = eL = eC = 0.0
    for (env, r), (mt, C0, P0, L0) in start.items():
        _, C1, P1, L1 = rr.momenta(m, *state_of(root, dof, env, r))
        eP = max(eP, np.abs(P1 - (P0 + mt * np.array([0, 0, -G]) * t)).max() / mt)
        eL = max(eL, np.abs(L1 - L0).max())
        eC = max(eC, np.abs(C1 - (C0 + P0 / mt * t + 0.5 * np.array([0, 0, -G]) * t * (t + dt))).max())   # the parabola of semi-implicit Euler
    return eP, eL, eC, float(dof[..., 1].abs().max())


def test_momentum_budget_in_free_flight_under_joint_torques():
    """No external force but gravity: the linear momentum changes by m g t, the angular momentum about the centre of mass not at
    all, whatever the joints do -- both recomputed from differenced poses of the independent kinematics.  The time stepping is
    first order (explicit velocity-product terms), so (a) at walking-gait joint speeds the budget closes to a few mm/s per robot,
    and (b) with the legs flailing at their 28 / 50 rad/s speed limits, into their stops and into each other (all internal
    impulses) the residual is integration error: it shrinks with the time step."""
    eP, eL, eC, qd = _momentum_errors(0.005, 0.5, 1.0)
    ePq, eLq, eCq, _ = _momentum_errors(0.00125, 0.5, 1.0)
    assert qd > 3.0
    assert eP < 3e-3 and eL < 2.5e-2 and eC < 2e-3, (eP, eL, eC)                      # m/s of the whole robot, kg m^2/s, m
    assert ePq < 0.4 * eP and eLq < 0.4 * eL, ((eP, ePq), (eL, eLq))                # first-order convergence
    eP1, eL1, eC1, qd1 = _momentum_errors(0.005, 20.0, 2.0)
    eP4, eL4, eC4, qd4 = _momentum_errors(0.00125, 20.0, 2.0)
    assert qd1 > 25.0 and qd4 > 25.0, "the legs must really have been thrown around"
    assert eP1 < 0.2 and eL1 < 0.5, (eP1, eL1)
    # different (chaotic) trajectories -- which links hit each other, and when, changes with the step: no clean factor 4, but smaller
    assert eP4 < 0.9 * eP1 and eL4 < 0.9 * eL1, ((eP1, eP4), (eL1, eL4))


def test_energy_budget_in_free_flight():
    """passive drift in zero gravity (no torques, tumbling base, swinging legs): the kinetic energy stays put over 1 s = 200
    substeps; driven: the energy gained equals the work of the joint torques, sum tau . qd dt.  (With gravity on, free fall adds
    the integrator's known 1/2 m g^2 t dt to the budget, 2.7 J after 1 s, which would bury the 1 J of internal motion.)"""
    m = rr.load_model()
    e, d, root, dof, g = flight(N=1, seed=5, gravity=0.0)
    dof[:, :, 1] = torch.randn(1, 24, generator=g) * 1.5
    root[:, :, 10:13] = torch.randn(1, 2, 3, generator=g)

    def energy(r):
        p0, R0, q, gv = state_of(root, dof, 0, r)
        return rr.kinetic_energy(m, p0, R0, q, gv)
    E0 = [energy(r) for r in range(2)]
    ke0 = [rr.kinetic_energy(m, *state_of(root, dof, 0, r)) for r in range(2)]
    worst = 0.0
    for k in range(200):
        e.simulate()
        if k % 20 == 19:
            worst = max(worst, max(abs(energy(r) - E0[r]) / ke0[r] for r in range(2)))
    assert worst < 0.05, f"energy wandered by {worst:.3f} of the initial kinetic energy"
    # driven: constant small torques for 20 substeps (the limits are off in this scene: a large torque would spin a 0.2 kg calf up
    # to hundreds of rad/s, where a 5 ms step no longer resolves the motion)
    tau = e.tensor(abi.T_TORQUES)
    tau.copy_((torch.rand(1, 24, generator=g) - 0.5) * 1.0)
    E1 = [energy(r) for r in range(2)]
    work = [0.0, 0.0]
    for k in range(20):
        qd_old = dof[0, :, 1].clone()
        e.simulate()
        qd_mid = 0.5 * (qd_old + dof[0, :, 1])
        for r in range(2):
            work[r] += float((tau[0, r * 12:(r + 1) * 12] * qd_mid[r * 12:(r + 1) * 12]).sum()) * d.dt
    for r in range(2):
        assert abs(work[r]) > 0.1
        assert energy(r) - E1[r] == pytest.approx(work[r], rel=0.05, abs=0.02), (energy(r) - E1[r], work[r])


def test_torsional_pendulum_has_the_closed_form_period():
    """Zero gravity, free flight, control type "P" with kd = 0: a joint is a torsional spring.  With every other body made 1000 x
    heavier the FL calf swings about its knee alone: w^2 = kp / (I_yy,cm + m |c|^2) from the model file, and symplectic Euler turns
    w into w_d with cos(w_d dt) = 1 - (w dt)^2 / 2.  Period from the zero crossings of 10 oscillations."""
    m = rr.load_model()
    d, k, ctx = make_desc("go1gate", 1)
    d.control_type = abi.CTRL["P"]
    d.gravity_z = 0.0
    d.kp, d.kd = 4.0, 0.0
    d.self_collision = 0
    calf = 3                                            # body index of FL_calf (base, FL hip, thigh, calf, ...)
    for j in range(12):
        d.robot.dof_lower[j], d.robot.dof_upper[j], d.robot.dof_vel_limit[j] = -100.0, 100.0, 0.0
    for b in range(13):
        if b != calf:
            d.robot.mass[b] *= 1000.0
            for c in range(6):
                d.robot.inertia[b][c] *= 1000.0
    e = oracle_engine(d, k, f64=True)
    e.reset_all()
    root, dof = e.tensor(abi.T_ROOT_STATE), e.tensor(abi.T_DOF_STATE)
    root[:, :, 2] = 50.0; root[:, 1, 1] += 5.0; root[:, :, 7:] = 0
    q0 = torch.tensor([d.default_dof_pos[j] for j in range(12)] * 2)
    dof[0, :, 0] = q0; dof[0, :, 1] = 0
    amp = 0.3
    dof[0, 2, 0] += amp                                 # FL calf joint displaced from its spring's rest angle
    axis = m["joint_axis"][calf]
    c = m["com"][calf]
    I_pivot = axis @ m["inertia"][calf] @ axis + m["mass"][calf] * (c @ c - (axis @ c) ** 2)
    w = np.sqrt(d.kp / I_pivot)
    w_d = np.arccos(1 - (w * d.dt) ** 2 / 2) / d.dt
    e.tensor(abi.T_ACTIONS).zero_()                     # action 0 -> target = default pose (legged_robot.py:385)
    xs = []
    for step in range(int(10.5 * 2 * np.pi / w_d / d.dt) + 1):      # the decimation loop's body (go1.py:48-56), no post-step / resets
        e.compute_torques()
        e.simulate()
        xs.append(float(dof[0, 2, 0] - q0[2]))
    xs = np.array(xs)
    up = [i for i in range(1, len(xs)) if xs[i - 1] < 0 <= xs[i]]
    t_cross = [(i - 1 + (0 - xs[i - 1]) / (xs[i] - xs[i - 1])) * d.dt for i in up]
    period = (t_cross[-1] - t_cross[0]) / (len(t_cross) - 1)
    assert len(t_cross) >= 9
    assert period == pytest.approx(2 * np.pi / w_d, rel=2e-3), (period, 2 * np.pi / w_d, 2 * np.pi / w)
    assert 0.9 * amp < np.abs(xs).max() < 1.1 * amp    # undamped
    assert abs(float(dof[0, 1, 0] - q0[1])) < 2e-3      # the heavy thigh stayed put


def test_ball_slides_then_rolls_at_the_closed_form_speed():
    """A ball launched sliding without spin: Coulomb friction decelerates it (mu g) and spins it up until the contact point sticks;
    from then on it rolls at v0 / (1 + I / (m r^2)) -- independent of mu and of how the solver got there."""
    d, k, ctx = make_desc("go1football-defender", 1)
    e = oracle_engine(d, k, f64=True)
    e.reset_all()
    root = e.tensor(abi.T_ROOT_STATE)
    A, r = d.num_agents, d.npc_sphere_radius[0]
    root[0, :A, 0] -= 0.0
    root[0, A, 0] += 1.0                        # clear of the robots
    root[0, A, 2] = d.ground_z + r
    root[0, A, 7:13] = 0
    v0 = 2.0
    root[0, A, 8] = v0                          # along +y, across the pitch
    for t in range(200):
        e.simulate()
    want = v0 / (1.0 + d.npc_inertia / (d.npc_mass * r * r))
    assert float(root[0, A, 8]) == pytest.approx(want, rel=0.02), (float(root[0, A, 8]), want)
    assert float(root[0, A, 10]) == pytest.approx(-want / r, rel=0.03)        # rolling: w_x = -v_y / r
    assert abs(float(root[0, A, 2]) - (d.ground_z + r)) < 2e-3


def test_ball_rolls_down_a_ramp_with_the_closed_form_acceleration():
    """The relief map as an inclined plane h = s y (exact under the bilinear sampling): a ball released at rest rolls without
    slipping (mu = 1 >> 2/7 tan(theta)) with a = g sin(theta) / (1 + I / (m r^2)) along the slope -- contact normal from the map's
    gradient, friction, and the rolling constraint through the contact solver, none of it shared with a flat-ground test."""
    from helpers import perlin_terrain
    slope = 0.1
    d, k, ctx = make_desc("go1football-defender", 1, terrain_cfg=perlin_terrain("go1football-defender", zScale=0.01))
    hs = d.horizontal_scale
    ramp = np.ascontiguousarray(np.tile((slope * np.arange(d.sdf_ny) * hs).astype(np.float32), (d.sdf_nx, 1)))
    k.append(ramp)
    d.ground_height = ramp.ctypes.data_as(abi.FP)
    e = oracle_engine(d, k, f64=True)
    e.reset_all()
    root = e.tensor(abi.T_ROOT_STATE)
    A, r = d.num_agents, d.npc_sphere_radius[0]
    root[0, :A, 2] += 30.0                       # the robots out of the way (falling, far above)
    root[0, A, 0] += 1.0
    th = np.arctan(slope)
    y0 = float(root[0, A, 1])
    root[0, A, 2] = d.ground_z + slope * y0 + r / np.cos(th)      # touching: centre r above the plane along its normal
    root[0, A, 7:13] = 0
    a = G * np.sin(th) / (1.0 + d.npc_inertia / (d.npc_mass * r * r))
    T = 100
    for t in range(T):
        e.simulate()
    tt = T * d.dt
    v = root[0, A, 7:10].numpy()
    assert float(v[1]) == pytest.approx(-a * tt * np.cos(th), rel=0.03), (v, a * tt)
    assert float(v[2]) == pytest.approx(-a * tt * np.sin(th), rel=0.05)
    assert abs(float(v[0])) < 1e-3
    assert float(root[0, A, 10]) == pytest.approx(a * tt / r, rel=0.04)          # rolling: w_x = +|v| / r for motion along -y
    gap = float(root[0, A, 2]) - d.ground_z - slope * float(root[0, A, 1]) - r / np.cos(th)
    assert abs(gap) < 3e-3, gap                  # stays on the plane


def test_joint_velocity_limit_is_an_internal_impulse():
    """go1.urdf:115,157,185 (50 / 28 / 28 rad/s): a joint thrown faster than its limit is braked to the limit within the substep,
    by an impulse along the joint -- so the robot's linear and angular momentum do not notice.  (Time step 10 us: the impulse does
    not depend on it, while the first-order integration error of a 60 rad/s leg, which would mask the budget, vanishes.)"""
    m = rr.load_model()
    e, d, root, dof, g = flight(N=1, widen=False, seed=2, dt=1e-5)
    assert [round(d.robot.dof_vel_limit[j]) for j in range(3)] == [50, 28, 28]
    d0 = torch.tensor([d.default_dof_pos[j] for j in range(12)] * 2)
    dof[0, :, 0] = d0
    dof[0, :, 1] = 0
    dof[0, 1, 1] = 60.0          # FL thigh, limit 28
    dof[0, 12 + 3, 1] = -80.0    # robot 1 FR hip, limit 50
    before = [rr.momenta(m, *state_of(root, dof, 0, r)) for r in range(2)]
    e.simulate()
    assert 26.5 < float(dof[0, 1, 1]) <= 28.0 + 1e-4 and -50.0 - 1e-4 <= float(dof[0, 15, 1]) < -48.0     # at the bound, up to the coupling with the neighbours
    vl = torch.tensor([d.robot.dof_vel_limit[j] for j in range(12)] * 2)
    assert (dof[0, :, 1].abs() <= vl + 1e-4).all()
    for r in range(2):
        mt, C0, P0, L0 = before[r]
        _, C1, P1, L1 = rr.momenta(m, *state_of(root, dof, 0, r))
        assert np.abs(P1 - P0 - mt * np.array([0, 0, -G]) * d.dt).max() < 2e-3 * mt
        assert np.abs(L1 - L0).max() < 0.02 * (np.abs(L0).max() + 0.1), (L0, L1)
