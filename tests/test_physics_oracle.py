"""Known-answer tests of the rigid-body step (row H).  The reference has no physics of its own to compare with
(Isaac Gym), so these pin the build's CPU specification physically; the HIP kernel is then held to it (-m gpu)."""
import numpy as np
import pytest
import torch

from helpers import make_desc, oracle_engine, self_gap
from mqe.engine import abi

G = 9.81
pytestmark = pytest.mark.usefixtures("solver")      # every test under both contact solvers (conftest.py)


def fresh(task="go1gate", N=2, f64=False, **kw):
    d, k, ctx = make_desc(task, N, **kw)
    e = oracle_engine(d, k, f64=f64)
    e.reset_all()
    root, dof = e.tensor(abi.T_ROOT_STATE), e.tensor(abi.T_DOF_STATE)
    q0 = torch.tensor([d.default_dof_pos[j] for j in range(12)] * d.num_agents)
    dof[:, :12 * d.num_agents, 0] = q0
    dof[..., 1] = 0
    root[..., 7:] = 0
    e.tensor(abi.T_TORQUES).zero_()
    return e, d, root, dof


def test_free_fall_is_rigid():
    """no contacts, no torques, zero velocity: every link accelerates with g => base dv = g dt, joint dv = 0"""
    e, d, root, dof = fresh()
    root[:, :, 2] = 2.0
    e.simulate()
    assert torch.allclose(root[:, :, 9], torch.full((2, 2), -G * d.dt), atol=2e-6)
    assert root[:, :, 7:9].abs().max() < 1e-6 and root[:, :, 10:13].abs().max() < 2e-5
    assert dof[..., 1].abs().max() < 2e-4
    assert torch.allclose(root[:, :, 2], torch.full((2, 2), 2.0 - G * d.dt * d.dt), atol=1e-6)   # semi-implicit Euler


def test_mass_matrix_base_block_and_symmetry():
    e, d, root, dof = fresh()
    M, Minv, _ = e.debug_dynamics(0, 0)
    mt = sum(d.robot.mass[b] for b in range(13))
    np.testing.assert_allclose(M[:3, :3], mt * np.eye(3), atol=1e-4)
    np.testing.assert_allclose(M, M.T, atol=1e-5)
    assert np.all(np.linalg.eigvalsh(M.astype(np.float64)) > 0)
    np.testing.assert_allclose(M.astype(np.float64) @ Minv.astype(np.float64), np.eye(18), atol=2e-3)


def test_static_stand_supports_weight():
    e, d, root, dof = fresh()
    a = torch.zeros(2, 2, 3)
    for t in range(120):
        e.step(a)
    cf = e.tensor(abi.T_CONTACT_FORCE).reshape(2, 2, 17, 3)
    mt = sum(d.robot.mass[b] for b in range(13))
    fz = cf[:, :, [4, 8, 12, 16], 2].sum(-1)
    assert torch.allclose(fz, torch.full((2, 2), mt * G), rtol=0.08)          # feet carry the weight
    assert (cf[:, :, 0].norm(dim=-1) < 1e-6).all()                             # trunk does not touch
    z = root[:, :, 2]
    assert ((z > 0.27) & (z < 0.34)).all()
    assert (e.tensor(abi.T_RESET_COUNT) == 1).all()                            # nobody terminated
    # spheres do not sink: foot centre height >= ground + r - small penetration
    assert root[:, :, 7:10].abs().max() < 0.15


def test_ball_comes_to_rest_on_the_ground():
    e, d, root, dof = fresh("go1football-defender", 2)
    A = d.num_agents
    root[:, A, 2] = 0.5
    for t in range(150):
        e.simulate()
    r = d.npc_sphere_radius[0]
    assert torch.allclose(root[:, A, 2], torch.full((2,), d.ground_z + r), atol=3e-3)
    assert root[:, A, 7:10].abs().max() < 0.05


def test_sliding_ball_decelerates_with_mu_g():
    e, d, root, dof = fresh("go1football-defender", 1)
    A = d.num_agents
    root[0, A, 2] = d.ground_z + d.npc_sphere_radius[0]
    root[0, A, 7] = 3.0
    root[0, A, 10:13] = 0
    v = []
    for t in range(10):
        e.simulate()
        v.append(root[0, A, 7].item())
    # sliding (not yet rolling): dv/dt = -mu g until the contact point sticks
    dec = (v[0] - v[4]) / (4 * d.dt)
    assert dec == pytest.approx(d.friction * G, rel=0.1)


def test_float32_and_float64_oracles_agree():
    outs = []
    for f64 in (False, True):
        e, d, root, dof = fresh(N=3, f64=f64)
        g = torch.Generator().manual_seed(0)
        for t in range(10):
            e.step(torch.rand(3, 2, 3, generator=g) * 2 - 1)
        outs.append((root.clone(), dof.clone()))
    assert (outs[0][0][..., :3] - outs[1][0][..., :3]).abs().max() < 2e-3
    assert (outs[0][1][..., 0] - outs[1][1][..., 0]).abs().max() < 2e-2


def test_robots_collide_with_walls_and_each_other():
    e, d, root, dof = fresh(N=1)
    # drive robot 0 sideways into robot 1 and the side wall: positions must stay separated / inside the track
    for t in range(60):
        root[0, 0, 8] = 1.0
        e.simulate()
    dxy = (root[0, 0, :2] - root[0, 1, :2]).norm()
    assert dxy > 0.12, "trunk spheres (r=0.057) may not interpenetrate"
    assert torch.isfinite(root).all()


def test_seesaw_plank_carries_robots_and_obeys_hinge_limits():
    """go1seesaw: two robots standing on the right arm: the plank takes their weight and turns at the URDF velocity
    limit (0.2 rad/s, seesaw.urdf:65); an unloaded spinning plank stops where its end meets the ground slab."""
    e, d, root, dof = fresh("go1seesaw", 2)
    base = root[:, 2, :3].clone()
    for r, (dx, dy) in enumerate(((-1.2, -0.2), (-0.8, 0.25))):
        root[:, r, 0] = base[:, 0] + dx
        root[:, r, 1] = base[:, 1] + dy
        root[:, r, 2] = 1.25
    a = torch.zeros(2, 2, 3)
    thetas, fz = [], []
    for t in range(70):
        e.step(a)
        thetas.append(dof[0, 24, 0].item())
        fz.append(e.tensor(abi.T_CONTACT_FORCE)[0, 2 * 17 + 1, 2].item())
        assert abs(dof[:, 24, 1]).max() <= d.seesaw_vel_limit + 1e-6
    assert (e.tensor(abi.T_RESET_COUNT) == 1).all()
    mt = sum(d.robot.mass[b] for b in range(13))
    rate = (thetas[60] - thetas[30]) / (30 * 0.02)
    assert rate == pytest.approx(d.seesaw_vel_limit, rel=0.02)                 # loaded arm goes down at the limit
    assert np.mean(fz[30:60]) == pytest.approx(-2 * mt * G, rel=0.1)           # plank carries both robots
    # end stops: plank alone, spinning either way
    for sign, stop in ((1.0, d.seesaw_theta_hi), (-1.0, d.seesaw_theta_lo)):
        e2, d2, root2, dof2 = fresh("go1seesaw", 1)
        root2[:, :2, 0] -= 5.0        # robots far away from the plank
        dof2[0, 24, 0] = 0.0
        for k in range(400):
            dof2[0, 24, 1] = sign * 0.2
            e2.simulate()
        assert dof2[0, 24, 0].item() == pytest.approx(stop, abs=1e-5)


def test_box_rests_on_its_face_and_carries_a_robot():
    """go1pushbox: the 6 kg unit box (box.urdf) dropped from its spawn height settles upright on four corner contacts
    carrying m g; a robot set down on top of it (sphere vs oriented box) stands there and its weight reaches the ground
    through the box."""
    e, d, root, dof = fresh("go1pushbox", 2)
    A = d.num_agents
    for t in range(200):
        e.simulate()
    hz = d.npc_box_half[2]
    assert torch.allclose(root[:, A, 2], torch.full((2,), d.ground_z + hz), atol=4e-3)
    # upright, at rest.  Yaw creeps under either solver (four corner contacts solved one after the other from a cold start, 4 sweeps:
    # 0.006 rad/s with the velocity-level sweeps, 0.018 rad/s with the temporal ones, whose normal impulses still move in the last
    # sub-step) -- 1.2 cm/s at a corner of the 1 m box; PhysX warm-starts its contacts, this engine does not
    assert root[:, A, 3:5].abs().max() < 2e-3 and root[:, A, 5].abs().max() < 5e-3 and root[:, A, 7:13].abs().max() < 0.05
    fz = e.tensor(abi.T_CONTACT_FORCE)[:, A * 17, 2]
    assert torch.allclose(fz, torch.full((2,), d.npc_mass * G), rtol=0.05)
    root[:, 0, :2] = root[:, A, :2]
    root[:, 0, 2] = d.ground_z + 2 * hz + 0.36
    root[:, 1, 1] = root[:, A, 1] + 3.0                                                        # the other robot out of the way
    a = torch.zeros(2, 2, 3)
    for t in range(40):                     # ~2x the settling time; the stand-in body network is not a balance controller
        e.step(a)
    assert (e.tensor(abi.T_RESET_COUNT) == 1).all()
    top = d.ground_z + 2 * hz
    assert ((root[:, 0, 2] > top + 0.25) & (root[:, 0, 2] < top + 0.36)).all(), root[:, 0, 2]   # standing on the lid
    assert torch.allclose(root[:, A, 2], torch.full((2,), d.ground_z + hz), atol=6e-3)         # the box neither sinks nor tips
    assert root[:, A, 3:5].abs().max() < 5e-3                                                  # no tip (yaw is free)
    mt = sum(d.robot.mass[b] for b in range(13))
    fz = e.tensor(abi.T_CONTACT_FORCE)[:, A * 17, 2]
    # net CONTACT force on the box balances its own weight only: ground pushes up with (m_box + m_robot) g, the feet down with m_robot g
    assert torch.allclose(fz, torch.full((2,), d.npc_mass * G), atol=0.15 * mt * G)
    feet = e.tensor(abi.T_CONTACT_FORCE).reshape(2, -1, 3)[:, [4, 8, 12, 16], 2].sum(-1)
    assert torch.allclose(feet, torch.full((2,), mt * G), rtol=0.15)                           # ... which is what the feet report
    assert torch.isfinite(root).all()


def test_revolving_door_spins_freely_and_is_stopped_by_a_robot():
    """go1revolvingdoor: the door (rotation_door.urdf) is a 1-dof link on a vertical hinge without drive, damping or range:
    it keeps its angular velocity; a robot standing in its sweep takes the hit through sphere-vs-oriented-box contacts and
    the door loses (most of) its spin."""
    e, d, root, dof = fresh("go1revolvingdoor", 2)
    A = d.num_agents
    assert d.seesaw_axis == 2 and d.num_npcs == 1
    root[:, :A, 0] += 5.0                                   # both robots far away
    dof[:, 12 * A, 1] = 1.0
    for t in range(50):
        e.simulate()
    assert torch.allclose(dof[:, 12 * A, 1], torch.full((2,), 1.0), atol=1e-5)
    assert torch.allclose(dof[:, 12 * A, 0], torch.full((2,), 50 * d.dt), atol=1e-4)
    # robot 0 in the path of the +y half of the door (hinge at the NPC base, door half width 0.975 along y at angle 0)
    hinge = root[:, A, :3].clone()
    dof[:, 12 * A, 0] = 0.0
    dof[:, 12 * A, 1] = 2.0                                 # +z spin: the +y half moves towards -x
    root[:, 0, 0] = hinge[:, 0] - 0.45
    root[:, 0, 1] = hinge[:, 1] + 0.6
    root[:, 0, 2] = 0.32
    root[:, :, 7:] = 0
    x0 = root[:, 0, 0].clone()
    for t in range(80):
        e.simulate()
    assert torch.isfinite(root).all() and torch.isfinite(dof).all()
    assert (dof[:, 12 * A, 1] < 1.0).all(), dof[:, 12 * A, 1]                  # the door was braked by the impact
    assert (root[:, 0, 0] < x0 - 0.02).all(), root[:, 0, 0] - x0              # and the robot was pushed along -x


def test_robots_stand_on_the_bridge_and_fall_beside_it():
    """go1bridge / go1wrestling: fixed scenery = world-aligned boxes (the STL collision meshes are boxes).  Robots dropped at
    their spawn points come to rest on the end blocks (top 1.02 m) / the field (top 0.5 m); one set down beside the 0.7 m
    wide deck has nothing under its feet and ends on the ground."""
    e, d, root, dof = fresh("go1bridge", 2)
    assert d.n_static_boxes == 3 and d.npc_reported_bodies == 3
    top = root[0, 2, 2].item() + d.static_box_center[0][2] + d.static_box_half[0][2]
    assert abs(top - 1.02) < 1e-5
    a = torch.zeros(2, 2, 3)
    root[1, 0, 0] = root[1, 2, 0]                                   # env 1: robot 0 next to the middle of the deck
    root[1, 0, 1] = root[1, 2, 1] + 1.0
    for t in range(60):
        e.step(a)
    assert ((root[0, :2, 2] > top + 0.24) & (root[0, :2, 2] < top + 0.36)).all(), root[0, :2, 2]   # standing on the end blocks
    assert abs(root[1, 1, 2] - root[0, 1, 2]) < 0.02
    assert e.tensor(abi.T_RESET_COUNT)[0] == 1 and e.tensor(abi.T_RESET_COUNT)[1] >= 2           # the one beside the deck fell (z_low)
    e2, d2, root2, dof2 = fresh("go1wrestling", 2)
    assert d2.n_static_boxes == 1 and d2.npc_reported_bodies == 9
    for t in range(60):
        e2.step(a)
    assert ((root2[:, :2, 2] > 0.5 + 0.24) & (root2[:, :2, 2] < 0.5 + 0.36)).all(), root2[:, :2, 2]
    q = root2[:, :2, 3:7]
    assert torch.allclose(q.norm(dim=-1), torch.ones(2, 2), atol=1e-5)                           # (0,0,-+1,1) start quaternions are normalised by the integrator
    assert (e2.tensor(abi.T_RESET_COUNT) == 1).all()


def test_tug_slider_translates_and_pushes_a_robot():
    """go1tug: the 3 kg disc (cylinder.urdf) slides along +y without friction or drive, capped at the joint's 1 m/s; a robot
    standing at its rim is shoved along (sphere vs upright cylinder) and the disc gives up momentum."""
    e, d, root, dof = fresh("go1tug", 2)
    A = d.num_agents
    assert d.seesaw_axis == 3 and d.seesaw_link_cylinder == 1
    root[:, :A, 0] += 6.0                                   # robots out of the way
    dof[0, 12 * A, 1] = 0.5
    dof[1, 12 * A, 1] = 3.0                                 # above the velocity limit
    for t in range(40):
        e.simulate()
    assert abs(dof[0, 12 * A, 1] - 0.5) < 1e-6 and abs(dof[0, 12 * A, 0] - 40 * d.dt * 0.5) < 1e-5
    assert abs(dof[1, 12 * A, 1] - d.seesaw_vel_limit) < 1e-6
    hinge = root[:, A, :3].clone()
    dof[:, 12 * A, 0] = 0.0
    dof[:, 12 * A, 1] = 0.9
    root[:, 0, 0] = hinge[:, 0]
    root[:, 0, 1] = hinge[:, 1] + d.seesaw_plank_half[0] + 0.32
    root[:, 0, 2] = 0.32
    root[:, 0, 3:7] = torch.tensor([0.0, 0.0, 0.0, 1.0])
    root[:, :, 7:] = 0
    y0 = root[:, 0, 1].clone()
    vmax = torch.zeros(2)
    for t in range(80):
        e.simulate()
        vmax = torch.maximum(vmax, root[:, 0, 8])
    assert torch.isfinite(root).all() and torch.isfinite(dof).all()
    assert (dof[:, 12 * A, 1] < 0.8).all(), dof[:, 12 * A, 1]                  # momentum went into the robot
    # which was shoved along +y: 2.7 N s of disc momentum would give its 12.6 kg 0.2 m/s (the unpowered robot is collapsing onto
    # its belly meanwhile and mu = 1 friction stops it within millimetres, so the base POSITION says little: its roll moves it more)
    assert (vmax > 0.05).all(), vmax
    assert (root[:, 0, 1] - (hinge[:, 1] + dof[:, 12 * A, 0]) > d.seesaw_plank_half[0] - 0.05).all()   # and never ended up inside the disc


CROSSED_FRONT_FEET = [-0.6, 0.8, -2.2, 0.6, 0.8, -2.2, 0.1, 1.0, -1.5, -0.1, 1.0, -1.5]      # spheres 0 / 1 (FL / FR foot) overlap by 23 mm


def _crossed(self_collision):
    d, k, ctx = make_desc("go1gate", 1)
    d.self_collision = self_collision
    e = oracle_engine(d, k)
    e.reset_all()
    root, dof = e.tensor(abi.T_ROOT_STATE), e.tensor(abi.T_DOF_STATE)
    dof[0, :12, 0] = torch.tensor(CROSSED_FRONT_FEET)
    dof[..., 1] = 0
    root[..., 7:] = 0
    root[0, 0, 2] = 2.0                     # free fall: no other contact, and without contact forces the pose would not change
    e.tensor(abi.T_TORQUES).zero_()
    return e, d, root, dof


def test_links_of_one_robot_collide():
    """asset.self_collisions = 0 (go1_config.py:73): the front feet, posed 23 mm into each other, are pushed apart by a contact
    whose two sides are the same actor; with the flag off nothing happens.  The contact force is internal: the robot's
    momentum is not changed (the base keeps falling with g up to the reaction of the legs' relative motion)."""
    assert self_gap(CROSSED_FRONT_FEET, 0, 1) < -0.02
    e, d, root, dof = _crossed(1)
    _, _, con = e.debug_dynamics(0, 0)
    # every contact has robot 0 on both sides and joins the two front legs (links 1-3 and 4-6): foot against foot (taken once: the
    # deepest) and each foot against the other calf's end cap, which ends inside its foot sphere
    assert len(con) == 3 and (con[:, 0] == 0).all() and (con[:, 2] == 0).all(), con
    assert all({int(c[1]) <= 3, int(c[3]) <= 3} == {True, False} and 1 <= min(c[1], c[3]) and max(c[1], c[3]) <= 6 for c in con), con
    assert abs(float(con[:, 4].min()) - self_gap(CROSSED_FRONT_FEET, 0, 1)) < 1e-5
    for _ in range(40):
        e.simulate()
    assert self_gap(dof[0, :12, 0].numpy(), 0, 1) > -1e-3, "the penetration must be resolved"
    assert torch.isfinite(root).all() and torch.isfinite(dof).all()
    assert abs(float(root[0, 0, 9]) + G * d.dt * 40) < 0.5, "an internal force cannot stop the fall"
    e2, d2, root2, dof2 = _crossed(0)
    assert len(e2.debug_dynamics(0, 0)[2]) == 0
    for _ in range(40):
        e2.simulate()
    assert abs(self_gap(dof2[0, :12, 0].numpy(), 0, 1) - self_gap(CROSSED_FRONT_FEET, 0, 1)) < 1e-4, "free fall keeps the pose"


def test_perlin_relief_is_the_ground():
    """on a Perlin track the walkable surface is the heightfield (no 2 cm slab).  A ball dropped onto it rides ON the
    relief -- never through it, never gaining energy from it, hugging it once the bounces have died down on gentle bumps --
    and robots spawned above it settle with their feet on it."""
    from helpers import perlin_terrain
    for zs, settle in ((0.12, False), (0.03, True)):
        d, k, ctx = make_desc("go1football-defender", 3, terrain_cfg=perlin_terrain("go1football-defender", zScale=zs))
        t = ctx["terrain"]
        assert d.ground_z == 0.0 and t.ground_height is not None and t.ground_height.max() > 0.6 * zs
        e = oracle_engine(d, k)
        e.reset_all()
        root = e.tensor(abi.T_ROOT_STATE)
        A, r, hs = d.num_agents, d.npc_sphere_radius[0], d.horizontal_scale

        def relief(xy):
            fx, fy = xy[0] / hs, xy[1] / hs
            ix, iy = int(fx), int(fy)
            tx, ty = fx - ix, fy - iy
            g = t.ground_height
            return float((g[ix, iy] * (1 - ty) + g[ix, iy + 1] * ty) * (1 - tx) + (g[ix + 1, iy] * (1 - ty) + g[ix + 1, iy + 1] * ty) * tx)

        def energy(env):
            v = root[env, A, 7:10]
            return 0.5 * d.npc_mass * float(v @ v) + 0.5 * d.npc_inertia * float(root[env, A, 10:13] @ root[env, A, 10:13]) + d.npc_mass * G * float(root[env, A, 2])
        for env in range(3):
            root[env, A, 0] += 0.5 + 0.37 * env                  # three different spots of the pitch, clear of the robots
            root[env, A, 1] += 0.3 * env
            root[env, A, 2] = relief(root[env, A, :2].tolist()) + r + 0.05
            root[env, A, 7:13] = 0
        e0 = [energy(env) for env in range(3)]
        for step in range(800):
            e.simulate()
            if step % 25 == 0:
                for env in range(3):
                    gap = float(root[env, A, 2]) - r - relief(root[env, A, :2].tolist())
                    assert gap > -0.006, (zs, env, step, gap)     # never through the surface (contact margin + first-order distance)
        for env in range(3):
            assert energy(env) < e0[env] + 1e-3, "contacts with the relief may only dissipate"
            if settle:
                gap = float(root[env, A, 2]) - r - relief(root[env, A, :2].tolist())
                # gentle bumps: it rolls on (nothing but sliding friction dissipates on this surface), hugging the relief
                assert gap < 0.006 and root[env, A, 7:10].abs().max() < 0.5, (env, gap, root[env, A, 7:10])
    # robots dropped from their spawn height stand on the bumps
    e2 = oracle_engine(*make_desc("go1gate", 2, terrain_cfg=perlin_terrain("go1gate", zScale=0.06))[:2])
    e2.reset_all()
    for step in range(60):
        e2.step(torch.zeros(2, 2, 3))
    r2 = e2.tensor(abi.T_ROOT_STATE)
    assert torch.isfinite(r2).all() and ((r2[:, :, 2] > 0.2) & (r2[:, :, 2] < 0.45)).all(), r2[:, :, 2]
    cf = e2.tensor(abi.T_CONTACT_FORCE).reshape(2, 2, 17, 3)
    assert (cf[:, :, [4, 8, 12, 16], 2].sum(-1) > 40).all()                                    # feet carry (most of) the 113 N


def test_walls_of_different_heights_carry_a_ball_at_their_own_top():
    """a (lo, hi) wall_height (one draw per block): the engine's terrain is the wall SDF plus, per cell, the top of the nearest wall.
    A ball set down on the middle of the lowest and of the tallest wall rests at THAT wall's top, and a ball beside the tall wall,
    above the low one's top, is still stopped by it"""
    from helpers import wall_heights_terrain
    np.random.seed(0)
    d, k, ctx = make_desc("go1football-defender", 2, terrain_cfg=wall_heights_terrain("go1football-defender"))
    t = ctx["terrain"]
    assert t.wall_top is not None
    e = oracle_engine(d, k, f64=True)
    e.reset_all()
    root = e.tensor(abi.T_ROOT_STATE)
    A, r, hs = d.num_agents, d.npc_sphere_radius[0], d.horizontal_scale
    deep = t.wall_sdf < -1.4 * hs                        # well inside a wall: the contact is with the top face only
    assert deep.any()
    tops = np.where(deep, t.wall_top, np.nan)
    lo_ij = np.unravel_index(np.nanargmin(tops), tops.shape)
    hi_ij = np.unravel_index(np.nanargmax(tops), tops.shape)
    want = [float(t.wall_top[lo_ij]), float(t.wall_top[hi_ij])]
    assert want[1] - want[0] > 0.1
    root[:, :A, 2] += 30.0                               # robots out of the way
    for env, ij in enumerate((lo_ij, hi_ij)):
        root[env, A, 0] = ij[0] * hs
        root[env, A, 1] = ij[1] * hs
        root[env, A, 2] = want[env] + r + 0.01
        root[env, A, 7:13] = 0
    for _ in range(120):
        e.simulate()
    for env in range(2):
        assert abs(float(root[env, A, 2]) - (want[env] + r)) < 2e-3, (env, float(root[env, A, 2]), want[env] + r)
        assert root[env, A, 7:10].abs().max() < 1e-2


def _buried_ball(e, d, root, depths, steps):
    """the football of env i starts `depths[i]` below its rest height, at rest; returns per step its height above rest and its vertical speed"""
    A, r = d.num_agents, d.npc_sphere_radius[0]
    root[:, :A, 2] += 30.0                       # the robots out of the way
    root[:, A, 0] += 1.0
    root[..., 7:] = 0
    for i, dep in enumerate(depths):
        root[i, A, 2] = d.ground_z + r - dep
    z, v = [], []
    for t in range(steps):
        e.simulate()
        z.append((root[:, A, 2] - d.ground_z - r).clone()); v.append(root[:, A, 9].clone())
    return torch.stack(z).cpu().numpy().astype(np.float64), torch.stack(v).cpu().numpy().astype(np.float64)


def check_penetration_recovery(d, z, v):
    """What each contact solver's rule says about a sphere that starts 1 mm, 3 mm and 60 mm inside the ground, at rest.
    Temporal Gauss-Seidel: a penetration leaves at no more than max_depenetration_velocity and within the sub-steps it needs -- 1 mm and 3 mm
    are gone inside the first step (one resp. three sub-steps of 1.25 mm) and, the later sub-steps asking for nothing, NO velocity is left;
    60 mm take twelve whole steps at exactly 1 m/s, and that speed is what the ball leaves with (PhysX's pop-out, which the cap bounds).
    Velocity-level sweeps with erp = 0.2: a fifth of the penetration per step, as velocity erp * depth / dt (capped at the same 1 m/s)."""
    dt, cap = d.dt, d.max_depenetration_velocity
    if d.solver_type == 1:
        assert np.abs(z[:, :2]).max() < 2e-5 and np.abs(v[:, :2]).max() < 1e-3, (z[0], v[0])         # resolved in the first step, at rest ever after
        np.testing.assert_allclose(z[:12, 2], -0.06 + cap * dt * np.arange(1, 13), atol=2e-5)          # 5 mm per step
        np.testing.assert_allclose(v[:12, 2], cap, atol=1e-3)
        np.testing.assert_allclose(v[12:16, 2], cap - G * dt * np.arange(1, 5), atol=2e-3)             # then a free flight that starts at the cap
        assert z[:, 2].max() < cap * cap / (2 * G) + 1e-3                                              # and never rises beyond its ballistic height
    else:
        for i, dep in enumerate((0.001, 0.003)):
            np.testing.assert_allclose(z[:8, i], -dep * 0.8 ** np.arange(1, 9), rtol=2e-2, atol=2e-6)
            np.testing.assert_allclose(v[:8, i], 0.2 * dep * 0.8 ** np.arange(8) / dt, rtol=2e-2, atol=1e-4)
        np.testing.assert_allclose(v[:8, 2], cap, atol=1e-3)                                           # 0.2 * 60 mm / dt = 2.4 m/s asks for more than the cap
    assert np.isfinite(z).all() and np.isfinite(v).all()


def test_penetration_recovery_follows_the_solvers_rule():
    e, d, root, dof = fresh("go1football-defender", 3)
    z, v = _buried_ball(e, d, root, (0.001, 0.003, 0.06), 40)
    check_penetration_recovery(d, z, v)
