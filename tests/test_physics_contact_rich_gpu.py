"""The contact-rich scenarios of tests/test_physics_contact_rich.py on the HIP engine (k_simulate / k_compute_torques through the C
ABI), held (1) to the same physical answers and (2) to the float64 CPU specification's own time-step limit: the kernel's f32 trajectory
at dt = 5 ms must lie as close to the oracle's FINEST run as the oracle's own 5 ms run does."""
import numpy as np
import pytest

import contact_rich as cr
from helpers import hip_engine, oracle_engine

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("solver")]


def f64(d, k):
    return oracle_engine(d, k, f64=True)


def test_hip_trot_converges_to_the_specifications_time_step_limit(solver):
    ref = cr.trot(f64, 0.000625)
    o5 = cr.trot(f64, 0.005)
    h = {dt: cr.trot(hip_engine, dt) for dt in (0.005, 0.00125)}
    same = np.linalg.norm(h[0.005]["pos"] - o5["pos"], axis=-1)
    assert same.max() < 5e-4, same.max()                              # kernel (f32) vs specification (f64) at the same step size: measured on the CPU pair 5e-5
    e_h5 = np.median(np.linalg.norm(h[0.005]["pos"] - ref["pos"], axis=-1))
    e_o5 = np.median(np.linalg.norm(o5["pos"] - ref["pos"], axis=-1))
    e_h1 = np.median(np.linalg.norm(h[0.00125]["pos"] - ref["pos"], axis=-1))
    assert e_h5 < 1.2 * e_o5 + 1e-4 and e_h1 < 0.35 * e_h5, (e_h5, e_o5, e_h1)
    for dt, r in h.items():
        assert not r["fell"].any() and abs(float(r["height"].mean()) - float(ref["height"].mean())) < 1e-3
        rel = np.abs(r["impulse"] / r["impulse_expected"] - 1.0).max()
        assert rel < (1e-2 if solver == "tgs" else 2.5e-1 * dt + 1e-4), (dt, rel)


def test_hip_robot_dropped_onto_its_feet_loses_the_falls_energy_and_stands_on_its_weight(solver):
    r = cr.drop(hip_engine)
    assert r["ke"].max() > 0.5 * r["ke_fall"] and r["ke"][-40:].max() < 2e-3 * r["ke_fall"], (r["ke"].max(), r["ke"][-40:].max())
    assert np.allclose(r["fz_end"], r["weight"], rtol=5e-3), (r["fz_end"], r["weight"])
    assert r["rebound"] < 1e-4 and np.abs(r["z_end"]).max() < 1e-3 and r["trunk_force"] == 0.0
    assert np.abs(r["impulse_balance"]).max() < 3e-4, r["impulse_balance"]


def test_hip_box_on_a_ramp_sticks_below_the_friction_limit_and_slides_above_it(solver):
    mu = 0.5
    for diag in (False, True):
        lim = cr.friction_frame_limit(mu, diag)
        stick = cr.box_on_ramp(hip_engine, 0.93 * lim, diag, mu)
        assert abs(stick["slid"]) < 1e-2 and abs(stick["speed"]) < 2e-2, (diag, stick)
        slide = cr.box_on_ramp(hip_engine, 1.1 * lim, diag, mu)
        assert slide["speed"] > 0.1 and slide["slid"] > 0.02 and abs(slide["gap"]) < 1e-3, (diag, slide)
    r = cr.box_on_ramp(hip_engine, 0.7, False, mu)
    assert r["speed"] == pytest.approx(cr.G * (np.sin(r["theta"]) - mu * np.cos(r["theta"])) * 0.5, rel=0.03), r
    o = cr.box_on_ramp(f64, 0.7, False, mu)
    assert r["slid"] == pytest.approx(o["slid"], abs=2e-4) and r["speed"] == pytest.approx(o["speed"], abs=2e-3), (r, o)


def test_hip_two_robots_colliding_in_free_flight_keep_their_total_momentum(solver):
    r = cr.collide_in_flight(hip_engine)
    assert r["max_contact_force"] > 50.0 and np.abs(r["dP_each"][:, 1]).min() > 3.0
    assert np.abs(r["dP"]).max() < 3e-3 and np.abs(r["dL"]).max() < 5e-3, (r["dP"], r["dL"])
