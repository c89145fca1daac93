"""The forward depth camera on the HIP engine (mqe_render_depth, csrc/kernels_camera.hpp; reference legged_robot_field.py:23-93,196-223):
(1) the geometric known answers of tests/camera_cases.py, (2) HIP == the CPU specification's scalar ray caster (oracle/: mqo_render_depth)
pixel by pixel on scenes with every kind of surface -- walls, relief, other robots, sheep, the free box, the plank, scenery -- after a
few random steps, (3) the sensor dictionary through the plugin API."""
import numpy as np
import pytest
import torch

import camera_cases as cc
from helpers import make_desc, hip_engine, oracle_engine, perlin_terrain
from mqe.engine import abi

pytestmark = pytest.mark.gpu


def test_flat_ground_from_a_known_height():
    cc.flat_ground_from_a_known_height(hip_engine)


def test_a_wall_where_the_signed_distance_map_says():
    cc.a_wall_where_the_signed_distance_map_says(hip_engine)


def test_another_robots_trunk_and_a_ball():
    cc.another_robots_trunk_and_a_ball(hip_engine)


@pytest.mark.parametrize("task,relief", [("go1gate", False), ("go1sheep-hard", False), ("go1pushbox", False), ("go1seesaw", False), ("go1bridge", False),
                                         ("go1football-defender", False), ("go1tug", False), ("go1gate", True)])
def test_hip_depth_image_is_the_specifications(task, relief):
    """VERDICT r4 "Next" 7: the kernel against the oracle's ray caster on the SAME state (copied from the HIP engine after 6 random steps,
    robots scattered and tilted so that they see each other, the NPCs and the walls): identical hit / miss masks up to silhouette pixels
    (<= 0.2 % may differ: a ray grazing an edge or a march step falling on the other side of a threshold in f32 vs f64), depths to 1e-4 m
    + 1e-5 relative on the rest."""
    N = 16
    kw = dict(terrain_cfg=perlin_terrain(task, zScale=0.08)) if relief else {}
    d1, k1, _ = make_desc(task, N, **kw)
    d2, k2, _ = make_desc(task, N, **kw)
    eh, eo = hip_engine(d1, k1), oracle_engine(d2, k2, f64=True)
    eh.reset_all(); eo.reset_all()
    g = torch.Generator().manual_seed(3)
    Aw = eh.tensor(abi.T_WRAPPER_OBS).shape[1]
    for t in range(6):
        eh.step((torch.rand(N, Aw, 3, generator=g) * 2 - 1).cuda())
    A = d1.num_agents
    ro, do = eh.tensor(abi.T_ROOT_STATE), eh.tensor(abi.T_DOF_STATE)
    # robots turned towards each other / the NPCs, some pitched and rolled
    yaw = torch.rand(N, A, generator=g) * 6.283
    pitch = (torch.rand(N, A, generator=g) - 0.5) * 0.6
    roll = (torch.rand(N, A, generator=g) - 0.5) * 0.4
    cy, sy, cp, sp, cr_, sr = torch.cos(yaw / 2), torch.sin(yaw / 2), torch.cos(pitch / 2), torch.sin(pitch / 2), torch.cos(roll / 2), torch.sin(roll / 2)
    q = torch.stack([sr * cp * cy - cr_ * sp * sy, cr_ * sp * cy + sr * cp * sy, cr_ * cp * sy - sr * sp * cy, cr_ * cp * cy + sr * sp * sy], -1)
    ro[:, :A, 3:7] = q.cuda()
    torch.cuda.synchronize()
    eo.tensor(abi.T_ROOT_STATE).copy_(ro.cpu()); eo.tensor(abi.T_DOF_STATE).copy_(do.cpu())
    for (hh, ww, fov, pos, rpy, far) in ((24, 32, 87.0, [0.26, 0.0, 0.03], [0.0, 0.0, 0.0], 20.0), (16, 16, 60.0, [0.2, 0.05, 0.1], [0.1, 0.35, -0.4], 6.0)):
        ih = eh.render_depth(hh, ww, fov, pos, rpy, far).cpu().numpy().astype(np.float64)
        io = eo.render_depth(hh, ww, fov, pos, rpy, far).numpy().astype(np.float64)
        mh, mo = np.isfinite(ih), np.isfinite(io)
        assert mo.mean() > 0.3                                      # the cameras do see something
        both = mh & mo
        off = np.abs(ih[both] - io[both]) > 1e-4 + 1e-5 * np.abs(io[both])
        bad = (mh != mo).sum() + off.sum()
        assert bad <= 2e-3 * ih.size, (task, relief, int((mh != mo).sum()), int(off.sum()), ih.size, float(np.abs(ih[both] - io[both]).max()))
        assert np.median(np.abs(ih[both] - io[both])) < 2e-6


def test_sensor_dict_through_the_plugin_api():
    import types
    from mqe.envs.utils import make_mqe_env, custom_cfg, ENV_DICT
    from mqe.utils.helpers import finish_args
    a = finish_args(types.SimpleNamespace(task="go1gate", num_envs=4, seed=1, headless=True, record_video=False, sim_device="cuda:0", pipeline="gpu",
                                          subscenes=0, num_threads=0))
    base = custom_cfg(a)

    def with_depth(cfg):
        cfg = base(cfg)
        cfg.obs.cfgs.depth_image = True
        return cfg
    saved_n = ENV_DICT["go1gate"]["config"].env.num_envs
    try:
        env, cfg = make_mqe_env("go1gate", a, with_depth)
        env.reset()
        sd = env.env.sensor_tensor_dict
        assert len(sd["forward_depth"]) == 4 and sd["forward_depth"][0].shape == (2, 16, 16)
        low = sd["forward_depth"][0][:, -1, :]                # the bottom row looks at the ground right in front
        assert torch.isfinite(low).all() and (low < 0).all() and float((-low).max()) < 1.0
    finally:
        ENV_DICT["go1gate"]["config"].obs.cfgs.depth_image = False
        ENV_DICT["go1gate"]["config"].env.num_envs = saved_n
