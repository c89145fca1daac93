"""Geometric known answers of the forward depth camera (legged_robot_field.py:23-93,196-223), engine-agnostic: `make(d, keep)` is the scalar
ray caster of the CPU specification (oracle/: mqo_render_depth, tests/test_camera_oracle.py) or the HIP kernel (mqe_render_depth,
tests/test_camera_gpu.py).  The reference renders with Isaac Gym's rasteriser, which cannot run here: the flat ground from a known height, a
wall found by marching the scene's own signed-distance map on the host, another robot's trunk at a known distance, a ball -- in Isaac Gym's
IMAGE_DEPTH convention (negative depth along the optical axis, -inf where nothing is hit)."""
import math

import numpy as np
import torch

from helpers import make_desc
from mqe.engine import abi

H = W = 16
POS, ROT = [0.26, 0.0, 0.03], [0.0, 0.0, 0.0]            # cfg.sensor.forward_camera (legged_robot_field_config.py:72-76)


def _upright(e, A):
    ro, do = e.tensor(abi.T_ROOT_STATE).clone(), e.tensor(abi.T_DOF_STATE).clone()
    ro[:, :A, 3:7] = torch.tensor([0.0, 0.0, 0.0, 1.0], device=ro.device)
    ro[:, :A, 7:] = 0
    return ro, do


def flat_ground_from_a_known_height(make):
    N = 4
    d, k, _ = make_desc("go1plane", N)
    e = make(d, k)
    e.reset_all()
    ro, do = _upright(e, 1)
    ro[:, 0, 2] = d.ground_z + 0.40
    e.tensor(abi.T_ROOT_STATE).copy_(ro)
    img = e.render_depth(H, W, 90.0, POS, ROT, far=50.0).cpu().view(N, H, W)
    cam_h = 0.40 + POS[2]
    for i in range(H):
        zc = -(2 * (i + 0.5) / H - 1)                     # tan(45 deg) = 1, square image
        want = cam_h / -zc if zc < 0 else None            # depth along the axis at which a ray of slope zc reaches the ground
        row = img[:, i, 5:11]                             # (the outer columns see the track's side walls first)
        if want is not None and want < 3.0:               # (farther out the track's border walls may come first)
            assert torch.allclose(row, torch.full_like(row, -want), atol=2e-4), (i, want, row[0])
        elif zc > 0:
            assert (row <= -0.5).all() or torch.isinf(row).all()      # above the horizon: walls far away or nothing


def a_wall_where_the_signed_distance_map_says(make):
    N = 8
    d, k, _ = make_desc("go1gate", N)
    e = make(d, k)
    e.reset_all()
    ro, do = _upright(e, 2)
    nx, ny, hs = d.sdf_nx, d.sdf_ny, d.horizontal_scale
    sdf = np.ctypeslib.as_array(d.wall_sdf, shape=(nx, ny)).copy()
    # robot 0 of every env looks along +x from its reset position at mid wall height; robot 1 is moved out of the way
    ro[:, 0, 2] = d.ground_z + 0.15
    ro[:, 1, 1] += 50.0 * hs
    e.tensor(abi.T_ROOT_STATE).copy_(ro)
    img = e.render_depth(H, W, 90.0, [0.26, 0.0, 0.0], ROT, far=30.0).cpu().view(N, 2, H, W)
    checked = 0
    for env in range(N):
        ox, oy = float(ro[env, 0, 0]) + 0.26, float(ro[env, 0, 1])
        # host march of the same map along the central ray (two central columns straddle it: yc = -+1/16)
        t, hit = 0.0, None
        while t < 20.0:
            fx, fy = (ox + t) / hs, oy / hs
            if fx >= nx - 1 or fy >= ny - 1 or fx < 0 or fy < 0:
                break
            ix, iy = int(fx), int(fy)
            tx, ty = fx - ix, fy - iy
            s = (sdf[ix, iy] * (1 - ty) + sdf[ix, iy + 1] * ty) * (1 - tx) + (sdf[ix + 1, iy] * (1 - ty) + sdf[ix + 1, iy + 1] * ty) * tx
            if s <= 0.002:
                hit = t
                break
            t += max(s, 0.002)
        if hit is None or hit < 0.3:
            continue
        # the four central pixels look 1/16 of the half width off axis and 1/16 below / above it: a wall face normal to x gives the same depth
        c = -img[env, 0, H // 2 - 1:H // 2 + 1, W // 2 - 1:W // 2 + 1]
        assert torch.isfinite(c).all() and float((c - hit).abs().max()) < 3 * hs + 0.02 * hit, (env, hit, c)
        checked += 1
    assert checked >= 2


def another_robots_trunk_and_a_ball(make):
    N = 2
    d, k, _ = make_desc("go1football-1vs1", N)
    e = make(d, k)
    e.reset_all()
    ro, do = _upright(e, 2)
    do[:, :, 0] = torch.tensor(np.ctypeslib.as_array(d.default_dof_pos, shape=(12,)).copy(), device=do.device).repeat(2)[None, :] if do.shape[1] >= 24 else do[:, :, 0]
    base = ro[:, 0, :3].clone()
    base[:, 2] = d.ground_z + 0.32
    ro[:, 0, :3] = base
    ro[:, 1, :3] = base + torch.tensor([1.0, 0.0, 0.0], device=ro.device)           # robot 1 one metre ahead, same heading
    ball_r = float(d.npc_sphere_radius[0])
    ro[:, 2, :3] = base + torch.tensor([0.9, 0.6, 0.0], device=ro.device)           # the ball up and to the left, at camera height
    ro[:, 2, 2] = d.ground_z + 0.32 + POS[2]
    e.tensor(abi.T_ROOT_STATE).copy_(ro); e.tensor(abi.T_DOF_STATE).copy_(do)
    img = e.render_depth(64, 64, 90.0, POS, ROT, far=10.0).cpu().view(N, 2, 64, 64)
    # central pixels of robot 0's camera: the rear face of robot 1's trunk box (go1.urdf:56: 0.3762 long, centred on the base)
    want = 1.0 - 0.3762 / 2 - POS[0]
    c = -img[:, 0, 31:33, 31:33]
    assert float((c - want).abs().max()) < 2e-3, (want, c)
    # the ball: its nearest point along the ray through its centre; the pixel that looks at the centre
    bx, by = 0.9 - POS[0], 0.6
    j = int((1 - by / bx) / 2 * 64)                       # yc = by / bx -> column
    depth_centre = bx - ball_r * bx / math.hypot(bx, by)  # axial depth of the sphere's nearest point on that ray
    got = -img[:, 0, 31:33, j - 1:j + 2]
    assert float((got - depth_centre).abs().min()) < 0.01, (depth_centre, got)
    # robot 1 looks away from both: nothing but ground / far walls in its centre
    assert not torch.isfinite(img[:, 1, 31, 31]).any() or float((-img[:, 1, 31, 31]).min()) > 1.5


