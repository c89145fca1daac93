"""The forward depth camera of the CPU specification (oracle/: mqo_render_depth -- the scalar ray caster that DEFINES the image, since the
reference's rasteriser is closed; legged_robot_field.py:23-93,196-223) against the geometric known answers of tests/camera_cases.py.  The HIP
kernel is held to this caster pixel by pixel in tests/test_camera_gpu.py."""
import numpy as np
import torch

import camera_cases as cc
from helpers import make_desc, oracle_engine
from mqe.engine import abi


def f32(d, k):
    return oracle_engine(d, k)


def f64(d, k):
    return oracle_engine(d, k, f64=True)


def test_flat_ground_from_a_known_height():
    cc.flat_ground_from_a_known_height(f32)
    cc.flat_ground_from_a_known_height(f64)


def test_a_wall_where_the_signed_distance_map_says():
    cc.a_wall_where_the_signed_distance_map_says(f64)


def test_another_robots_trunk_and_a_ball():
    cc.another_robots_trunk_and_a_ball(f32)
    cc.another_robots_trunk_and_a_ball(f64)


def test_f32_and_f64_casters_agree_on_a_scattered_scene():
    N = 4
    imgs = []
    for f in (False, True):
        d, k, _ = make_desc("go1sheep-hard", N)
        e = oracle_engine(d, k, f64=f)
        e.reset_all()
        g = torch.Generator().manual_seed(0)
        ro = e.tensor(abi.T_ROOT_STATE)
        yaw = torch.rand(N, 2, generator=g) * 6.283
        ro[:, :2, 3:7] = torch.stack([torch.zeros_like(yaw), torch.zeros_like(yaw), torch.sin(yaw / 2), torch.cos(yaw / 2)], -1)
        imgs.append(e.render_depth(24, 32, 87.0, cc.POS, cc.ROT, 20.0).numpy().astype(np.float64))
    a, b = imgs
    ma, mb = np.isfinite(a), np.isfinite(b)
    assert (ma != mb).mean() < 2e-3 and ma.mean() > 0.5
    assert np.percentile(np.abs(a[ma & mb] - b[ma & mb]), 99.5) < 1e-4
