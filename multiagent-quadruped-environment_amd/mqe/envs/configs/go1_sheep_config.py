"""go1sheep-easy / go1sheep-hard: herd scripted sheep through a gate
(values: reference mqe/envs/configs/go1_sheep_config.py:5-130 and :132-256)."""
from mqe.utils.helpers import merge_dict
from mqe.envs.go1.go1_config import Go1Cfg
from ._common import two_agents_at_origin


def _track(track_width, init, gate, plane_len):
    return merge_dict(Go1Cfg.terrain.BarrierTrack_kwargs, dict(
        options=["init", "plane", "gate", "plane", "wall"],
        track_width=track_width,
        init=init,
        gate=gate,
        plane=dict(block_length=plane_len),
        wall=dict(block_length=0.1),
        wall_height=0.5,
        virtual_terrain=False,
        no_perlin_threshold=0.06,
        add_perlin_noise=False,
    ))


class _SheepCommon(Go1Cfg):
    class command(Go1Cfg.command):
        class cfg(Go1Cfg.command.cfg):
            vel = True

    class init_state(Go1Cfg.init_state):
        multi_init_state = True
        init_state_class = Go1Cfg.init_state
        init_states = two_agents_at_origin()

    class termination(Go1Cfg.termination):
        check_obstacle_conditioned_threshold = False
        termination_terms = ["roll", "pitch"]

    class domain_rand(Go1Cfg.domain_rand):
        init_base_pos_range = dict(x=[-0.1, 0.1], y=[-0.1, 0.1])
        init_npc_base_pos_range = dict(x=[-0.3, 0.3], y=[-0.3, 0.3])

    class viewer(Go1Cfg.viewer):
        pos = [0.0, 3.0, 5.0]
        lookat = [4.0, 3.0, 0.0]


class SingleSheepCfg(_SheepCommon):
    class env(Go1Cfg.env):
        env_name = "go1sheep"
        num_envs = 1
        num_agents = 2
        num_npcs = 1
        episode_length_s = 15

    class asset(Go1Cfg.asset):
        file_npc = "{LEGGED_GYM_ROOT_DIR}/resources/objects/sheep.urdf"
        name_npc = "sheep"
        num_rows = 1
        num_cols = 1
        dis_sheep = (1.5, 1.5)
        sheep_movement_scale = 0.2
        sheep_movement_randomness = 0.0
        sheep_movement_range = [2.0, 2.0, 0]

    class terrain(Go1Cfg.terrain):
        num_rows = 1
        num_cols = 1
        BarrierTrack_kwargs = _track(
            4.0,
            dict(block_length=1.5, room_size=(1.0, 1.95), border_width=0.00, offset=(0.5, 0)),
            dict(block_length=1.0, width=0.8, depth=0.1, offset=(0, 0), random=(0, 0.5)),
            3.0)

    class rewards(Go1Cfg.rewards):
        class scales:
            success_reward_scale = 1
            contact_punishment_scale = 0
            sheep_movement_reward_scale = 2
            mixed_sheep_reward_scale = 0
            sheep_pos_var_exp_punishment_scale = 0
            sheep_pos_var_lin_punishment_scale = 0


class NineSheepCfg(_SheepCommon):
    class env(Go1Cfg.env):
        env_name = "go1sheep"
        num_envs = 35
        num_agents = 2
        num_npcs = 9
        episode_length_s = 15

    class asset(Go1Cfg.asset):
        file_npc = "{LEGGED_GYM_ROOT_DIR}/resources/objects/sheep.urdf"
        name_npc = "sheep"
        num_rows = 3
        num_cols = 3
        dis_sheep = (1.5, 1.5)
        sheep_movement_scale = 0.2
        sheep_movement_randomness = 0.1
        sheep_movement_range = [2.0, 2.0, 0]

    class terrain(Go1Cfg.terrain):
        num_rows = 5
        num_cols = 7
        BarrierTrack_kwargs = _track(
            6.0,
            dict(block_length=2, room_size=(1.0, 3), border_width=0.00, offset=(0.5, 0)),
            dict(block_length=1.0, width=1.5, depth=0.1, offset=(0, 0), random=(0, 1)),
            6.0)

    class rewards(Go1Cfg.rewards):
        class scales:
            success_reward_scale = 0
            contact_punishment_scale = 0
            sheep_movement_reward_scale = 0
            mixed_sheep_reward_scale = 1
            sheep_pos_var_exp_punishment_scale = 0
            sheep_pos_var_lin_punishment_scale = 0
