"""go1gate: two robots pass a narrow gate (values: reference mqe/envs/configs/go1_gate_config.py:5-130)."""
from mqe.utils.helpers import merge_dict
from mqe.envs.go1.go1_config import Go1Cfg
from ._common import two_agents_at_origin


class Go1GateCfg(Go1Cfg):
    class env(Go1Cfg.env):
        env_name = "go1gate"
        num_envs = 1
        num_agents = 2
        episode_length_s = 10

    class terrain(Go1Cfg.terrain):
        num_rows = 1
        num_cols = 1
        BarrierTrack_kwargs = merge_dict(Go1Cfg.terrain.BarrierTrack_kwargs, dict(
            options=["init", "gate", "plane", "wall"],
            track_width=3.0,
            init=dict(block_length=2.0, room_size=(1.0, 1.5), border_width=0.00, offset=(0, 0)),
            gate=dict(block_length=3.0, width=0.6, depth=0.1, offset=(0, 0), random=(0.5, 0.5)),
            plane=dict(block_length=1.0),
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

    class control(Go1Cfg.control):
        control_type = "C"

    class termination(Go1Cfg.termination):
        check_obstacle_conditioned_threshold = False
        termination_terms = ["roll", "pitch", "z_low", "z_high"]

    class domain_rand(Go1Cfg.domain_rand):
        init_base_pos_range = None

    class rewards(Go1Cfg.rewards):
        class scales:
            target_reward_scale = 1
            success_reward_scale = 5
            lin_vel_x_reward_scale = 0
            approach_frame_punishment_scale = 0
            agent_distance_punishment_scale = -0.025
            contact_punishment_scale = -2
            lin_vel_y_punishment_scale = 0
            command_value_punishment_scale = 0

    class viewer(Go1Cfg.viewer):
        pos = [-2.0, 2.5, 4.0]
        lookat = [4.0, 2.5, 0.0]
