"""Row H on contact-rich motion (VERDICT r4 "Next" 5): evidence that does not come from the build's own derivation, on the CPU
specification (float64 oracle), under both contact solvers.  The HIP kernel runs the same scenarios under -m gpu
(tests/test_physics_contact_rich_gpu.py).  Scenarios and what is independent about each: tests/contact_rich.py.

(a) time-step convergence of a trotting robot: dt = 5 / 2.5 / 1.25 / 0.625 ms from identical states for 0.5 s -- the position
    difference to the finest run shrinks about linearly with dt (a first-order scheme with non-smooth contact events), base height,
    fall count and the vertical contact impulse (against the momentum theorem, momenta from independent float64 kinematics) agree
    across the sweep.  tools/dt_convergence.py writes the same sweep, at more sizes, to profiles/r05_dt_convergence.json.
(b1) a Go1 dropped 5 cm onto its four feet loses the fall's kinetic energy (restitution 0), does not rebound, and stands on m g.
(b2) a box on a ramp sticks below the friction limit and slides above it with g (sin - mu cos) -- along the friction frame's axis
     the limit is tan(theta) = mu; across it (slope along the diagonal) the friction BOX holds up to 1.30 mu: the anisotropy of the
     two-row box friction, computed from the tangent frame's definition.
(b3) two robots colliding in free flight exchange 5.7 kg m/s and keep their total linear and angular momentum."""
import numpy as np
import pytest

import contact_rich as cr
from helpers import oracle_engine

pytestmark = pytest.mark.usefixtures("solver")      # every test under both contact solvers (conftest.py)
DTS = (0.005, 0.0025, 0.00125, 0.000625)


def f64(d, k):
    return oracle_engine(d, k, f64=True)


def test_trot_converges_with_the_time_step(solver):
    res = {dt: cr.trot(f64, dt) for dt in DTS}
    ref = res[DTS[-1]]
    err = {dt: np.linalg.norm(res[dt]["pos"] - ref["pos"], axis=-1).ravel() for dt in DTS[:-1]}
    med = [float(np.median(err[dt])) for dt in DTS[:-1]]
    # measured (tgs / pgs): 2.1e-3 / 2.3e-3, 7.0e-4 / 9.3e-4, 3.6e-4 / 3.5e-4 m: halving dt divides the distance to the finest run by 2-3
    assert med[0] < 4e-3 and max(float(e.max()) for e in err.values()) < 1e-2, (med, [float(e.max()) for e in err.values()])
    assert med[1] < 0.6 * med[0] and med[2] < 0.7 * med[1], med
    qerr = [float(np.median(np.abs(res[dt]["q"] - ref["q"]).max(-1))) for dt in DTS[:-1]]
    assert qerr[0] < 5e-2 and qerr[2] < 0.4 * qerr[0], qerr
    for dt in DTS:
        r = res[dt]
        assert not r["fell"].any()                                                      # the same fall count at every step size: none
        assert abs(float(r["height"].mean()) - float(ref["height"].mean())) < 1e-3      # mean base height over the run
        # momentum theorem: sum of the vertical contact impulses = m g T + P_z(T) - P_z(0).  Velocity-level solver: first-order exact
        # (5.6e-4 at 5 ms, 7e-5 at 0.625 ms).  Temporal solver: 4.5e-3 at EVERY step size -- its positions move with the accumulated
        # motion of the sub-steps, i.e. a depenetration that is not kept as velocity changes the configuration-dependent momentum
        # without an impulse (the solver's own property; PhysX's TGS does the same)
        rel = np.abs(r["impulse"] / r["impulse_expected"] - 1.0).max()
        assert rel < (1e-2 if solver == "tgs" else 2.5e-1 * dt), (dt, rel)


def test_hopping_keeps_its_aggregates_across_the_time_step(solver):
    """2.5 x the amplitude: the robots hop (mean base height 0.39 m), trajectories part by centimetres within 0.5 s -- contact dynamics
    amplify any difference -- but height, fall count and impulse budget do not depend on the step size"""
    res = {dt: cr.trot(f64, dt, amp_scale=2.5) for dt in (0.005, 0.00125)}
    a, b = res[0.005], res[0.00125]
    assert not a["fell"].any() and not b["fell"].any()
    assert abs(float(a["height"].mean()) - float(b["height"].mean())) < 1e-2
    for r in (a, b):
        assert np.abs(r["impulse"] / r["impulse_expected"] - 1.0).max() < 1.5e-2


def test_a_robot_dropped_onto_its_feet_loses_the_falls_energy_and_stands_on_its_weight(solver):
    r = cr.drop(f64)
    assert r["ke"].max() > 0.5 * r["ke_fall"]                          # it did fall (contact begins 1 cm early: the contact offset)
    assert r["ke"][-40:].max() < 2e-3 * r["ke_fall"], r["ke"][-40:].max()      # measured 3e-3 J of 5.5 J (the PD loop still settling)
    assert np.allclose(r["fz_end"], r["weight"], rtol=5e-3), (r["fz_end"], r["weight"])
    assert r["rebound"] < 1e-4                                         # restitution 0: the base never rises after the first touch
    assert np.abs(r["z_end"]).max() < 1e-3                             # back at the standing height
    assert np.abs(r["impulse_balance"]).max() < 1e-4                   # momentum theorem over the whole landing (measured 3e-5 / 3e-6)
    assert r["trunk_force"] == 0.0


def test_box_on_a_ramp_sticks_below_the_friction_limit_and_slides_above_it(solver):
    mu = 0.5
    lim_axis, lim_diag = cr.friction_frame_limit(mu, False), cr.friction_frame_limit(mu, True)
    assert lim_axis == mu and lim_diag == pytest.approx(1.305 * mu, rel=5e-3)        # the box friction's anisotropy: DESIGN.md section 4
    for diag, lim in ((False, lim_axis), (True, lim_diag)):
        stick = cr.box_on_ramp(f64, 0.93 * lim, diag, mu)
        assert abs(stick["slid"]) < 1e-2 and abs(stick["speed"]) < 2e-2, (diag, stick)       # measured <= 6 mm of settling, 6 mm/s
        slide = cr.box_on_ramp(f64, 1.1 * lim, diag, mu)
        assert slide["speed"] > 0.1 and slide["slid"] > 0.02, (diag, slide)
        assert abs(slide["gap"]) < 1e-3
    # along the axis the slide is Coulomb's: a = g (sin theta - mu cos theta)
    for slope in (0.6, 0.8):
        r = cr.box_on_ramp(f64, slope, False, mu)
        a = cr.G * (np.sin(r["theta"]) - mu * np.cos(r["theta"]))
        assert r["speed"] == pytest.approx(a * 0.5, rel=0.03), (slope, r, a * 0.5)
        assert r["sideways"] < 2e-3
    # a slope between the two limits: slides along the axis, holds on the diagonal
    mid = 0.5 * (lim_axis + lim_diag)
    assert cr.box_on_ramp(f64, mid, False, mu)["speed"] > 0.1 and abs(cr.box_on_ramp(f64, mid, True, mu)["speed"]) < 2e-2


def test_two_robots_colliding_in_free_flight_keep_their_total_momentum(solver):
    r = cr.collide_in_flight(f64)
    assert r["max_contact_force"] > 50.0 and np.abs(r["dP_each"][:, 1]).min() > 3.0       # they did collide: 5.7 kg m/s exchanged
    assert np.abs(r["dP"]).max() < 3e-3, r["dP"]                                         # measured <= 1.1e-3 kg m/s (2e-4 of the exchange)
    assert np.abs(r["dL"]).max() < 5e-3, r["dL"]                                         # measured <= 1.5e-3 kg m^2/s
    q = cr.collide_in_flight(f64, dt=0.00125)
    assert np.abs(q["dP"]).max() < 1e-3 and np.abs(q["dL"]).max() < 1e-3, (q["dP"], q["dL"])
