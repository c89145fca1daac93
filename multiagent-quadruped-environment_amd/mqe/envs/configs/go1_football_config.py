"""go1football-defender: two learners vs a scripted defender (values: reference
mqe/envs/configs/go1_football_config.py:5-132)."""
from mqe.utils.helpers import merge_dict
from mqe.envs.go1.go1_config import Go1Cfg
from ._common import state


class Go1FootballDefenderCfg(Go1Cfg):
    class env(Go1Cfg.env):
        env_name = "go1football"
        num_envs = 1
        num_agents = 3
        num_npcs = 1
        episode_length_s = 20

    class asset(Go1Cfg.asset):
        file_npc = "{LEGGED_GYM_ROOT_DIR}/resources/objects/ball.urdf"
        name_npc = "ball"
        terminate_after_contacts_on = []
        npc_collision = True
        fix_npc_base_link = False
        npc_gravity = True

    class terrain(Go1Cfg.terrain):
        num_rows = 1
        num_cols = 1
        BarrierTrack_kwargs = merge_dict(Go1Cfg.terrain.BarrierTrack_kwargs, dict(
            options=["init", "gate", "plane", "gate", "wall"],
            track_width=9.0,
            init=dict(block_length=1.0, room_size=(0, 3.0), border_width=0.00, offset=(0.5, 0)),
            plane=dict(block_length=10.0),
            gate=dict(block_length=1.0, width=2.0, depth=1.0, offset=(0, 0), random=(0, 0.0)),
            wall=dict(block_length=0.1),
            wall_height=1.0,
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
        init_states = [state([3.0, 1.0, 0.42]), state([3.0, 2.0, 0.42]), state([9.0, -3.0, 0.42], rot=[0.0, 0.0, 1.0, 0.0])]
        init_states_npc = [state([5.0, -2.1, 0.3])]

    class control(Go1Cfg.control):
        control_type = "C"

    class termination(Go1Cfg.termination):
        check_obstacle_conditioned_threshold = False
        termination_terms = ["roll", "pitch"]

    class domain_rand(Go1Cfg.domain_rand):
        init_base_pos_range = dict(x=[-0.1, 0.1], y=[-0.1, 0.1])

    class rewards(Go1Cfg.rewards):
        class scales:
            goal_reward_scale = 10
            ball_gate_distance_reward_scale = 3

    class viewer(Go1Cfg.viewer):
        pos = [2.0, 2.0, 2.0]
        lookat = [6.0, 5.0, 0.0]


def _game_terrain(room_size):
    """terrain of the two free-play football tasks (reference go1_football_config.py:151-190, 264-303)"""
    return merge_dict(Go1Cfg.terrain.BarrierTrack_kwargs, dict(
        options=["init", "gate", "plane", "gate", "wall"],
        track_width=9.0,
        init=dict(block_length=1.0, room_size=room_size, border_width=0.00, offset=(0.5, 0)),
        plane=dict(block_length=10.0),
        gate=dict(block_length=1.0, width=2.0, depth=1.0, offset=(0, 0), random=(0, 0.0)),
        wall=dict(block_length=0.1),
        wall_height=1.0,
        virtual_terrain=False,
        no_perlin_threshold=0.06,
        add_perlin_noise=False,
    ))


class Go1Football1vs1Cfg(Go1Cfg):
    """go1football-1vs1: one robot per side and a free ball (values: reference go1_football_config.py:134-245)."""

    class env(Go1Cfg.env):
        env_name = "go1football"
        num_envs = 1
        num_agents = 2
        num_npcs = 1
        episode_length_s = 1

    class asset(Go1Cfg.asset):
        file_npc = "{LEGGED_GYM_ROOT_DIR}/resources/objects/ball.urdf"
        name_npc = "ball"
        terminate_after_contacts_on = []
        npc_collision = True
        fix_npc_base_link = False
        npc_gravity = True

    class terrain(Go1Cfg.terrain):
        num_rows = 1
        num_cols = 1
        BarrierTrack_kwargs = _game_terrain((0.0, 0.0))

    class command(Go1Cfg.command):
        class cfg(Go1Cfg.command.cfg):
            vel = True

    class init_state(Go1Cfg.init_state):
        multi_init_state = True
        init_state_class = Go1Cfg.init_state
        init_states = [state([3.0, 0.0, 0.42]), state([9.0, 0.0, 0.42], rot=[0.0, 0.0, 1.0, 0.0])]
        init_states_npc = [state([7.0, 0.0, 0.2])]

    class control(Go1Cfg.control):
        control_type = "C"

    class termination(Go1Cfg.termination):
        check_obstacle_conditioned_threshold = False
        termination_terms = ["roll", "pitch"]

    class domain_rand(Go1Cfg.domain_rand):
        init_base_pos_range = dict(x=[-0.1, 0.1], y=[-0.1, 0.1])

    class rewards(Go1Cfg.rewards):
        class scales:
            goal_reward_scale = 1

    class viewer(Go1Cfg.viewer):
        pos = [2.0, 2.0, 2.0]
        lookat = [6.0, 5.0, 0.0]


class Go1Football2vs2Cfg(Go1Cfg):
    """go1football-2vs2: two robots per side and a free ball (values: reference go1_football_config.py:247-372)."""

    class env(Go1Cfg.env):
        env_name = "go1football"
        num_envs = 1
        num_agents = 4
        num_npcs = 1
        episode_length_s = 20

    class asset(Go1Cfg.asset):
        file_npc = "{LEGGED_GYM_ROOT_DIR}/resources/objects/ball.urdf"
        name_npc = "ball"
        terminate_after_contacts_on = []
        npc_collision = True
        fix_npc_base_link = False
        npc_gravity = True

    class terrain(Go1Cfg.terrain):
        num_rows = 1
        num_cols = 1
        BarrierTrack_kwargs = _game_terrain((0, 0))

    class command(Go1Cfg.command):
        class cfg(Go1Cfg.command.cfg):
            vel = True

    class init_state(Go1Cfg.init_state):
        multi_init_state = True
        init_state_class = Go1Cfg.init_state
        init_states = [state([3.0, 2.0, 0.42]), state([3.0, -2.0, 0.42]),
                       state([9.0, 2.0, 0.42], rot=[0.0, 0.0, 1.0, 0.0]), state([9.0, -2.0, 0.42], rot=[0.0, 0.0, 1.0, 0.0])]
        init_states_npc = [state([7.0, 0.0, 0.2])]

    class control(Go1Cfg.control):
        control_type = "C"

    class termination(Go1Cfg.termination):
        check_obstacle_conditioned_threshold = False
        termination_terms = ["roll", "pitch"]

    class domain_rand(Go1Cfg.domain_rand):
        init_base_pos_range = dict(x=[-0.1, 0.1], y=[-0.1, 0.1])

    class rewards(Go1Cfg.rewards):
        class scales:
            goal_reward_scale = 1

    class viewer(Go1Cfg.viewer):
        pos = [2.0, 2.0, 2.0]
        lookat = [6.0, 5.0, 0.0]
