"""go1seesaw: climb a tilting plank onto a platform (values: reference mqe/envs/configs/go1_seesaw_config.py:5-136)."""
from mqe.utils.helpers import merge_dict
from mqe.envs.go1.go1_config import Go1Cfg
from ._common import two_agents_at_origin, state


class Go1SeesawCfg(Go1Cfg):
    class env(Go1Cfg.env):
        env_name = "go1seesaw"
        num_envs = 1
        num_agents = 2
        num_npcs = 1
        num_actions_npc = 1
        episode_length_s = 10

    class asset(Go1Cfg.asset):
        file_npc = "{LEGGED_GYM_ROOT_DIR}/resources/objects/seesaw.urdf"
        name_npc = "seesaw"
        npc_collision = True
        fix_npc_base_link = True
        npc_gravity = True

    class terrain(Go1Cfg.terrain):
        num_rows = 1
        num_cols = 1
        BarrierTrack_kwargs = merge_dict(Go1Cfg.terrain.BarrierTrack_kwargs, dict(
            options=["init", "plane", "wall"],
            track_width=3.0,
            init=dict(block_length=2.0, room_size=(1.0, 1.5), border_width=0.00, offset=(0, 0)),
            plane=dict(block_length=8.0),
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
        init_states = two_agents_at_origin()
        init_states_npc = [state([8.0, 0.0, 1.0])]
        default_npc_joint_angles = [-0.2]

    class control(Go1Cfg.control):
        control_type = "C"

        class default_command(Go1Cfg.control.default_command):
            gait = "pacing"

    class termination(Go1Cfg.termination):
        check_obstacle_conditioned_threshold = False
        termination_terms = ["roll", "pitch", "z_low"]

    class domain_rand(Go1Cfg.domain_rand):
        init_base_pos_range = dict(x=[-0.1, 0.1], y=[-0.1, 0.1])
        init_npc_base_pos_range = None

    class obs(Go1Cfg.obs):
        class cfgs(Go1Cfg.obs.cfgs):
            env_info = False

    class rewards(Go1Cfg.rewards):
        class scales:
            height_reward_scale = 1
            success_reward_scale = 10
            contact_punishment_scale = -2
            agent_distance_punishment_scale = -0.25
            x_movement_reward_scale = 5
            fall_punishment_scale = -2
            y_punishment_scale = -0.5

    class viewer(Go1Cfg.viewer):
        pos = [0.0, -2.0, 4.0]
        lookat = [4.0, 2.0, 0.0]
