"""go1plane: one robot on an open track (reference mqe/envs/configs/go1_plane_config.py)."""
from mqe.utils.helpers import merge_dict
from mqe.envs.go1.go1_config import Go1Cfg
from ._common import two_agents_at_origin


class Go1PlaneCfg(Go1Cfg):
    class env(Go1Cfg.env):
        env_name = "go1plane"
        num_envs = 1
        num_agents = 1
        episode_length_s = 10

    class terrain(Go1Cfg.terrain):
        num_rows = 1
        num_cols = 1
        BarrierTrack_kwargs = merge_dict(Go1Cfg.terrain.BarrierTrack_kwargs, dict(
            options=["init", "plane", "wall"],
            track_width=3.0,
            init=dict(block_length=2.0, room_size=(1.0, 1.5), border_width=0.00, offset=(0, 0)),
            plane=dict(block_length=5.0),
            wall=dict(block_length=0.1),
            wall_height=0.5,
            virtual_terrain=False,
            no_perlin_threshold=0.06,
            add_perlin_noise=False,
        ))

    class command(Go1Cfg.command):
        class cfg(Go1Cfg.command.cfg):
            vel = True

    class init_state(Go1Cfg.init_state):
        multi_init_state = True
        init_state_class = Go1Cfg.init_state
        init_states = two_agents_at_origin(1)

    class domain_rand(Go1Cfg.domain_rand):
        init_base_pos_range = None

    class rewards(Go1Cfg.rewards):
        class scales:
            pass
