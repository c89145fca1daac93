"""BarrierTrack + config tree against the reference's outputs (tests/golden/terrain_*.npz, configs.json)."""
import json
import os

import numpy as np
import pytest

from helpers import GOLD, golden, task_cfg
from mqe.utils.helpers import class_to_dict
from mqe.utils.terrain import BarrierTrack

TASKS = ["go1gate", "go1sheep-easy", "go1sheep-hard", "go1seesaw", "go1football-defender", "go1football-1vs1", "go1football-2vs2", "go1pushbox", "go1revolvingdoor", "go1bridge", "go1wrestling", "go1tug"]


def unrle(runs, shape):
    hf = np.zeros(shape, np.float32)
    for r, s, e, v in runs:
        hf[int(r), int(s):int(e)] = v
    return hf


@pytest.mark.parametrize("task", TASKS)
def test_barrier_track_matches_reference(task):
    z = golden("terrain_" + task)
    cfg = task_cfg(task)
    np.random.seed(0)
    t = BarrierTrack(cfg.terrain, 8, cfg.env.num_agents).build()
    assert np.array_equal(unrle(z["runs"], tuple(z["shape"])), t.heightfield_raw)
    assert np.array_equal(t.env_origins, z["env_origins"])
    assert np.array_equal(t.agent_origins, z["agent_origins"])
    if z["gate_deviation"].size:
        assert np.array_equal(t.env_info["gate_deviation"], z["gate_deviation"])
    # signed distance field: negative exactly on wall pixels, 1-Lipschitz in units of the pixel pitch
    wall = t.heightfield_raw > 0
    assert ((t.wall_sdf < 0) == wall).all()
    hs = cfg.terrain.horizontal_scale
    assert np.abs(np.diff(t.wall_sdf, axis=0)).max() <= hs * 1.0001 and np.abs(np.diff(t.wall_sdf, axis=1)).max() <= hs * 1.0001
    assert t.ground_z == pytest.approx(0.02)


def test_instances_do_not_leak_state():
    """the reference mutates a class-level kwargs dict (barrier_track.py:62); ours is per instance"""
    a = BarrierTrack(task_cfg("go1football-defender").terrain, 1, 3).build()
    b = BarrierTrack(task_cfg("go1gate").terrain, 1, 2).build()
    assert a.track_kwargs["track_width"] == 9.0 and b.track_kwargs["track_width"] == 3.0


@pytest.mark.parametrize("task", TASKS)
def test_config_tree_matches_reference(task):
    gold = json.load(open(os.path.join(GOLD, "configs.json")))[task]
    mine = json.loads(json.dumps(class_to_dict(task_cfg(task)), default=lambda o: class_to_dict(o) if hasattr(o, "__dict__") else str(o)))
    # fields the factory itself mutates on the config CLASS at construction (mqe/envs/utils.py:127, legged_robot.py:1022)
    for vol in ("num_envs", "max_episode_length"):
        mine["env"].pop(vol, None)
        gold["env"].pop(vol, None)
    assert mine == gold


def test_custom_cfg_plugin_point():
    from mqe.envs.utils import custom_cfg
    import types
    cfg = task_cfg("go1gate")
    old = cfg.env.num_envs
    out = custom_cfg(types.SimpleNamespace(num_envs=77, record_video=False))(cfg)
    assert out is cfg and cfg.env.num_envs == 77
    cfg.env.num_envs = old
