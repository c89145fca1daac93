"""Every task configuration of the package as data.  `mqe.envs.configs._build.cfg(name)` turns an entry into the class tree the
environments read (cfg.env.num_agents, cfg.terrain.BarrierTrack_kwargs, ...).  An entry lists only what it changes relative to
its `base`: SEC(inherit, attrs) is a nested section (inherit = derived from the base entry's section of the same name, otherwise
a fresh one that replaces it), S(...) an actor start state, REF(entry, section) a reference to another entry's section class.
Values: the reference's mqe/envs/base/legged_robot_config.py, mqe/envs/field/legged_robot_field_config.py,
mqe/envs/go1/go1_config.py and mqe/envs/configs/go1_*_config.py (dict-valued options are spelled out in full here)."""
from ._build import SEC, S, REF

SPEC = {
    'LeggedRobotCfg': ('BaseConfig', {
        'asset': SEC(False, {
            'angular_damping': 0.0,
            'armature': 0.0,
            'collapse_fixed_joints': True,
            'default_dof_drive_mode': 3,
            'density': 0.001,
            'disable_gravity': False,
            'file': '',
            'file_npc': '',
            'fix_base_link': False,
            'flip_visual_attachments': True,
            'foot_name': 'None',
            'linear_damping': 0.0,
            'max_angular_velocity': 1000.0,
            'max_linear_velocity': 1000.0,
            'name': 'legged_robot',
            'name_npc': '',
            'penalize_contacts_on': [],
            'replace_cylinder_with_capsule': True,
            'self_collisions': 0,
            'terminate_after_contacts_on': [],
            'thickness': 0.01
        }),
        'commands': SEC(False, {
            'curriculum': False,
            'heading_command': True,
            'max_curriculum': 1.0,
            'num_commands': 4,
            'ranges': SEC(False, {
                'ang_vel_yaw': [-1, 1],
                'heading': [-3.14, 3.14],
                'lin_vel_x': [-1.0, 1.0],
                'lin_vel_y': [-1.0, 1.0]
            }),
            'resampling_time': 10.0
        }),
        'control': SEC(False, {
            'action_scale': 0.5,
            'control_type': 'P',
            'damping': {'joint_a': 1.0, 'joint_b': 1.5},
            'decimation': 4,
            'stiffness': {'joint_a': 10.0, 'joint_b': 15.0}
        }),
        'curriculum': SEC(False, {}),
        'domain_rand': SEC(False, {
            'added_mass_range': [-1.0, 1.0],
            'friction_range': [0.5, 1.25],
            'init_dof_pos_ratio_range': [0.5, 1.5],
            'max_push_vel_ang': 0.0,
            'max_push_vel_xy': 1.0,
            'push_interval_s': 15,
            'push_robots': True,
            'randomize_base_mass': False,
            'randomize_friction': True
        }),
        'env': SEC(False, {
            'env_spacing': 3.0,
            'episode_length_s': 20,
            'num_actions': 12,
            'num_actions_npc': 0,
            'num_envs': 4096,
            'num_npcs': 0,
            'num_observations': 235,
            'num_privileged_obs': None,
            'send_timeouts': True,
            'use_lin_vel': True
        }),
        'init_state': SEC(False, {
            'ang_vel': [0.0, 0.0, 0.0],
            'default_joint_angles': {'joint_a': 0.0, 'joint_b': 0.0},
            'lin_vel': [0.0, 0.0, 0.0],
            'pos': [0.0, 0.0, 1.0],
            'rot': [0.0, 0.0, 0.0, 1.0]
        }),
        'noise': SEC(False, {
            'add_noise': True,
            'noise_level': 1.0,
            'noise_scales': SEC(False, {
                'ang_vel': 0.2,
                'dof_pos': 0.01,
                'dof_vel': 1.5,
                'gravity': 0.05,
                'height_measurements': 0.1,
                'lin_vel': 0.1
            })
        }),
        'normalization': SEC(False, {
            'clip_actions': 100.0,
            'clip_observations': 100.0,
            'obs_scales': SEC(False, {
                'ang_vel': 0.25,
                'dof_pos': 1.0,
                'dof_vel': 0.05,
                'height_measurements': 5.0,
                'lin_vel': 2.0
            })
        }),
        'rewards': SEC(False, {
            'base_height_target': 1.0,
            'max_contact_force': 100.0,
            'only_positive_rewards': True,
            'scales': SEC(False, {
                'action_rate': -0.01,
                'ang_vel_xy': -0.05,
                'base_height': -0.0,
                'collision': -1.0,
                'dof_acc': -2.5e-07,
                'dof_vel': -0.0,
                'feet_air_time': 1.0,
                'feet_stumble': -0.0,
                'lin_vel_z': -2.0,
                'orientation': -0.0,
                'stand_still': -0.0,
                'termination': -0.0,
                'torques': -1e-05,
                'tracking_ang_vel': 0.5,
                'tracking_lin_vel': 1.0
            }),
            'soft_dof_pos_limit': 1.0,
            'soft_dof_vel_limit': 1.0,
            'soft_torque_limit': 1.0,
            'tracking_sigma': 0.25
        }),
        'sim': SEC(False, {
            'dt': 0.005,
            'gravity': [0.0, 0.0, -9.81],
            'no_camera': True,
            'physx': SEC(False, {
                'bounce_threshold_velocity': 0.5,
                'contact_collection': 2,
                'contact_offset': 0.01,
                'default_buffer_size_multiplier': 5,
                'max_depenetration_velocity': 1.0,
                'max_gpu_contact_pairs': 8388608,
                'num_position_iterations': 4,
                'num_threads': 10,
                'num_velocity_iterations': 0,
                'rest_offset': 0.0,
                'solver_type': 1
            }),
            'substeps': 1,
            'up_axis': 1
        }),
        'terrain': SEC(False, {
            'border_size': 0,
            'curriculum': True,
            'difficulty_scale': 1.0,
            'dynamic_friction': 1.0,
            'horizontal_scale': 0.1,
            'max_init_terrain_level': 5,
            'max_platform_height': 0.2,
            'measure_heights': True,
            'measured_points_x': [-0.8, -0.7, -0.6, -0.5, -0.4, -0.3, -0.2, -0.1, 0.0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8],
            'measured_points_y': [-0.5, -0.4, -0.3, -0.2, -0.1, 0.0, 0.1, 0.2, 0.3, 0.4, 0.5],
            'mesh_type': 'trimesh',
            'num_cols': 20,
            'num_rows': 10,
            'restitution': 0.0,
            'selected': False,
            'slope_treshold': 0.75,
            'static_friction': 1.0,
            'terrain_kwargs': None,
            'terrain_length': 8.0,
            'terrain_proportions': [0.1, 0.1, 0.35, 0.25, 0.2],
            'terrain_smoothness': 0.005,
            'terrain_width': 8.0,
            'vertical_scale': 0.005,
            'x_init_offset': 0.0,
            'x_init_range': 1.0,
            'y_init_offset': 0.0,
            'y_init_range': 1.0,
            'yaw_init_range': 0.0
        }),
        'viewer': SEC(False, {
            'lookat': [11.0, 5, 3.0],
            'pos': [10, 0, 6],
            'ref_env': 0
        })
    }),
    'LeggedRobotFieldCfg': ('LeggedRobotCfg', {
        'sensor': SEC(False, {
            'forward_camera': SEC(False, {
                'position': [0.26, 0.0, 0.03],
                'resolution': [16, 16],
                'rotation': [0.0, 0.0, 0.0]
            }),
            'proprioception': SEC(False, {
                'delay_action_obs': False,
                'latency_range': [0.0, 0.0],
                'latency_resample_time': 2.0
            })
        }),
        'terrain': SEC(True, {
            'BarrierTrack_kwargs': {'add_perlin_noise': False, 'border_height': 0.0, 'border_perlin_noise': False, 'curriculum_perlin': False, 'engaging_next_threshold': 1.2, 'gate': {'block_length': 1.6, 'depth': 0.1, 'offset': (0.4, 0), 'random': (0.0, 0.0), 'width': 0.5}, 'init': {'block_length': 3.0, 'border_width': 0.0, 'offset': (0, 0), 'room_size': (1.0, 1.0)}, 'no_perlin_threshold': 0.06, 'options': ['init', 'gate', 'wall', 'plane'], 'plane': {'block_length': 3.0}, 'track_width': 2.0, 'virtual_terrain': False, 'wall': {'block_length': 3.0}, 'wall_height': 0.5, 'wall_thickness': 0.04},
            'TerrainPerlin_kwargs': {'frequency': 10, 'zScale': 0.12},
            'border_size': 1,
            'curriculum': False,
            'horizontal_scale': 0.025,
            'max_init_terrain_level': 0,
            'num_cols': 50,
            'num_rows': 20,
            'pad_unavailable_info': True,
            'selected': 'BarrierTrack',
            'slope_treshold': 100.0
        })
    }),
    'Go1Cfg': ('LeggedRobotFieldCfg', {
        'asset': SEC(False, {
            'angular_damping': 0.0,
            'armature': 0.0,
            'collapse_fixed_joints': True,
            'default_dof_drive_mode': 3,
            'density': 0.001,
            'disable_gravity': False,
            'file': '{LEGGED_GYM_ROOT_DIR}/resources/robots/go1/urdf/go1.urdf',
            'files': ['{LEGGED_GYM_ROOT_DIR}/resources/robots/go1/urdf/go1 blue.urdf', '{LEGGED_GYM_ROOT_DIR}/resources/robots/go1/urdf/go1 green.urdf', '{LEGGED_GYM_ROOT_DIR}/resources/robots/go1/urdf/go1 red.urdf', '{LEGGED_GYM_ROOT_DIR}/resources/robots/go1/urdf/go1 orange.urdf'],
            'fix_base_link': False,
            'flip_visual_attachments': False,
            'foot_name': 'foot',
            'linear_damping': 0.0,
            'max_angular_velocity': 1000.0,
            'max_linear_velocity': 1000.0,
            'name': 'go1',
            'penalize_contacts_on': ['base', 'thigh'],
            'replace_cylinder_with_capsule': True,
            'self_collisions': 0,
            'terminate_after_contacts_on': ['base'],
            'thickness': 0.01
        }),
        'command': SEC(False, {
            'cfg': SEC(False, {
                'aux_reward': False,
                'body_height': False,
                'body_pose': False,
                'footswing_height': False,
                'gait': False,
                'gait_freq': False,
                'stance_length': False,
                'stance_width': False,
                'vel': False
            }),
            'curriculum': False,
            'gaits': {'bounding': [0, 0.5, 0], 'pacing': [0, 0, 0.5], 'pronking': [0, 0, 0], 'trotting': [0.5, 0, 0]},
            'heading_command': True,
            'max_curriculum': 1.0,
            'num_commands': 4,
            'ranges': SEC(False, {
                'ang_vel_yaw': [-1, 1],
                'heading': [-3.14, 3.14],
                'lin_vel_x': [-1.0, 1.0],
                'lin_vel_y': [-1.0, 1.0]
            }),
            'resampling_time': 10.0
        }),
        'control': SEC(True, {
            'action_scale': 0.25,
            'actuator_network_path': './resources/actuator_nets',
            'computer_clip_torque': True,
            'control_type': 'C',
            'damping': {'joint': 0.5},
            'decimation': 4,
            'default_command': SEC(False, {
                'ang_vel': -0.0,
                'aux_reward': 0.0,
                'body_height': 0.0,
                'body_pitch': 0.0,
                'body_roll': 0.0,
                'footswing_height': 0.08,
                'gait': 'trotting',
                'gait_freq': 3.0,
                'lin_vel_x': 1.0,
                'lin_vel_y': -0.0,
                'stance_length': 0.428,
                'stance_width': 0.25
            }),
            'hip_scale_reduction': 0.5,
            'locomotion_policy_dir': './mqe/utils/locomotion_checkpoints/walk_these_ways',
            'motor_clip_torque': False,
            'obs_scales': SEC(False, {
                'ang_vel': 0.25,
                'aux_reward': 1.0,
                'body_height': 2.0,
                'body_pitch': 0.3,
                'body_roll': 0.3,
                'compliance': 1.0,
                'dof_pos': 1.0,
                'dof_vel': 0.05,
                'footswing_height': 0.15,
                'gait_freq': 1.0,
                'gait_phase': 1.0,
                'lin_vel': 2.0,
                'stance_length': 1.0,
                'stance_width': 1.0
            }),
            'stiffness': {'joint': 20.0},
            'torque_limits': [20.0, 20.0, 25.0, 20.0, 20.0, 25.0, 20.0, 20.0, 25.0, 20.0, 20.0, 25.0]
        }),
        'domain_rand': SEC(True, {
            'added_mass_range': [-1.0, 3.0],
            'com_range': SEC(False, {
                'x': [-0.05, 0.15],
                'y': [-0.1, 0.1],
                'z': [-0.05, 0.05]
            }),
            'friction_range': [0.05, 4.5],
            'init_base_pos_range': {'x': [0.1, 0.1], 'y': [-0.1, 0.1]},
            'init_dof_pos_ratio_range': [0.7, 1.3],
            'init_npc_base_pos_range': {'x': [-0.2, 0.2], 'y': [-0.2, 0.2]},
            'lag_timesteps': 6,
            'leg_motor_strength_range': [0.9, 1.1],
            'push_robots': False,
            'randomize_base_mass': False,
            'randomize_com': False,
            'randomize_friction': False,
            'randomize_lag_timesteps': False,
            'randomize_motor': False
        }),
        'env': SEC(True, {
            'env_spacing': 3.0,
            'episode_length_s': 5,
            'num_actions': 12,
            'num_envs': 256,
            'num_observations': 235,
            'num_privileged_obs': None,
            'record_actor_id': 0,
            'record_video': False,
            'recording_height_px': 240,
            'recording_mode': 'COLOR',
            'recording_width_px': 360,
            'send_timeouts': True,
            'use_lin_vel': True
        }),
        'init_state': SEC(True, {
            'default_joint_angles': {'FL_calf_joint': -1.5, 'FL_hip_joint': 0.1, 'FL_thigh_joint': 0.8, 'FR_calf_joint': -1.5, 'FR_hip_joint': -0.1, 'FR_thigh_joint': 0.8, 'RL_calf_joint': -1.5, 'RL_hip_joint': 0.1, 'RL_thigh_joint': 1.0, 'RR_calf_joint': -1.5, 'RR_hip_joint': -0.1, 'RR_thigh_joint': 1.0},
            'pos': [0.0, 0.0, 0.42]
        }),
        'normalization': SEC(True, {
            'clip_actions': 10.0
        }),
        'obs': SEC(False, {
            'cfgs': SEC(False, {
                'ang_vel': True,
                'base_pos': True,
                'base_quat': True,
                'base_rpy': True,
                'clock_inputs': False,
                'command': True,
                'contact_states': False,
                'depth_image': False,
                'dof_pos': True,
                'dof_vel': True,
                'env_info': True,
                'gait_commands': False,
                'height_command': False,
                'imu': False,
                'last_action': True,
                'last_last_action': True,
                'lin_vel': True,
                'projected_gravity': True,
                'rgb_image': False,
                'timing_parameter': False
            }),
            'scales': SEC(False, {
                'base_pos': 1.0,
                'base_quat': 1.0,
                'depth_image': 1.0,
                'rgb_image': 1.0,
                'segmentation_image': 1.0
            })
        }),
        'privileged_obs': SEC(False, {
            'cfgs': SEC(False, {})
        }),
        'rewards': SEC(True, {
            'base_height_target': 0.25,
            'scales': SEC(True, {
                'dof_pos_limits': -10.0,
                'torques': -0.0002
            }),
            'soft_dof_pos_limit': 0.9
        }),
        'termination': SEC(False, {
            'out_of_track_kwargs': {'threshold': 1.0},
            'pitch_kwargs': {'threshold': 1.6},
            'roll_kwargs': {'threshold': 0.8},
            'termination_terms': ['roll', 'pitch', 'z_low', 'z_high'],
            'z_high_kwargs': {'threshold': 1.5},
            'z_low_kwargs': {'threshold': 0.08}
        }),
        'viewer': SEC(True, {
            'lookat': [4.0, 11.0, 0.0],
            'pos': [0.0, 11.0, 5.0]
        })
    }),
    'Go1PlaneCfg': ('Go1Cfg', {
        'command': SEC(True, {
            'cfg': SEC(True, {
                'vel': True
            })
        }),
        'domain_rand': SEC(True, {
            'init_base_pos_range': None
        }),
        'env': SEC(True, {
            'env_name': 'go1plane',
            'episode_length_s': 10,
            'num_agents': 1,
            'num_envs': 1
        }),
        'init_state': SEC(True, {
            'init_state_class': REF("Go1Cfg", "init_state"),
            'init_states': [S(pos=[0.0, 0.0, 0.42], rot=[0.0, 0.0, 0.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0])],
            'multi_init_state': True
        }),
        'rewards': SEC(True, {
            'scales': SEC(False, {})
        }),
        'terrain': SEC(True, {
            'BarrierTrack_kwargs': {'add_perlin_noise': False, 'border_height': 0.0, 'border_perlin_noise': False, 'curriculum_perlin': False, 'engaging_next_threshold': 1.2, 'gate': {'block_length': 1.6, 'depth': 0.1, 'offset': (0.4, 0), 'random': (0.0, 0.0), 'width': 0.5}, 'init': {'block_length': 2.0, 'border_width': 0.0, 'offset': (0, 0), 'room_size': (1.0, 1.5)}, 'no_perlin_threshold': 0.06, 'options': ['init', 'plane', 'wall'], 'plane': {'block_length': 5.0}, 'track_width': 3.0, 'virtual_terrain': False, 'wall': {'block_length': 0.1}, 'wall_height': 0.5, 'wall_thickness': 0.04},
            'num_cols': 1,
            'num_rows': 1
        })
    }),
    'Go1GateCfg': ('Go1Cfg', {
        'command': SEC(True, {
            'cfg': SEC(True, {
                'vel': True
            })
        }),
        'control': SEC(True, {
            'control_type': 'C'
        }),
        'domain_rand': SEC(True, {
            'init_base_pos_range': None
        }),
        'env': SEC(True, {
            'env_name': 'go1gate',
            'episode_length_s': 10,
            'num_agents': 2,
            'num_envs': 1
        }),
        'init_state': SEC(True, {
            'init_state_class': REF("Go1Cfg", "init_state"),
            'init_states': [S(pos=[0.0, 0.0, 0.42], rot=[0.0, 0.0, 0.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0]), S(pos=[0.0, 0.0, 0.42], rot=[0.0, 0.0, 0.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0])],
            'multi_init_state': True
        }),
        'rewards': SEC(True, {
            'scales': SEC(False, {
                'agent_distance_punishment_scale': -0.025,
                'approach_frame_punishment_scale': 0,
                'command_value_punishment_scale': 0,
                'contact_punishment_scale': -2,
                'lin_vel_x_reward_scale': 0,
                'lin_vel_y_punishment_scale': 0,
                'success_reward_scale': 5,
                'target_reward_scale': 1
            })
        }),
        'termination': SEC(True, {
            'check_obstacle_conditioned_threshold': False,
            'termination_terms': ['roll', 'pitch', 'z_low', 'z_high']
        }),
        'terrain': SEC(True, {
            'BarrierTrack_kwargs': {'add_perlin_noise': False, 'border_height': 0.0, 'border_perlin_noise': False, 'curriculum_perlin': False, 'engaging_next_threshold': 1.2, 'gate': {'block_length': 3.0, 'depth': 0.1, 'offset': (0, 0), 'random': (0.5, 0.5), 'width': 0.6}, 'init': {'block_length': 2.0, 'border_width': 0.0, 'offset': (0, 0), 'room_size': (1.0, 1.5)}, 'no_perlin_threshold': 0.06, 'options': ['init', 'gate', 'plane', 'wall'], 'plane': {'block_length': 1.0}, 'track_width': 3.0, 'virtual_terrain': False, 'wall': {'block_length': 0.1}, 'wall_height': 0.5, 'wall_thickness': 0.04},
            'num_cols': 1,
            'num_rows': 1
        }),
        'viewer': SEC(True, {
            'lookat': [4.0, 2.5, 0.0],
            'pos': [-2.0, 2.5, 4.0]
        })
    }),
    '_SheepCommon': ('Go1Cfg', {
        'command': SEC(True, {
            'cfg': SEC(True, {
                'vel': True
            })
        }),
        'domain_rand': SEC(True, {
            'init_base_pos_range': {'x': [-0.1, 0.1], 'y': [-0.1, 0.1]},
            'init_npc_base_pos_range': {'x': [-0.3, 0.3], 'y': [-0.3, 0.3]}
        }),
        'init_state': SEC(True, {
            'init_state_class': REF("Go1Cfg", "init_state"),
            'init_states': [S(pos=[0.0, 0.0, 0.42], rot=[0.0, 0.0, 0.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0]), S(pos=[0.0, 0.0, 0.42], rot=[0.0, 0.0, 0.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0])],
            'multi_init_state': True
        }),
        'termination': SEC(True, {
            'check_obstacle_conditioned_threshold': False,
            'termination_terms': ['roll', 'pitch']
        }),
        'viewer': SEC(True, {
            'lookat': [4.0, 3.0, 0.0],
            'pos': [0.0, 3.0, 5.0]
        })
    }),
    'SingleSheepCfg': ('_SheepCommon', {
        'asset': SEC(True, {
            'dis_sheep': (1.5, 1.5),
            'file_npc': '{LEGGED_GYM_ROOT_DIR}/resources/objects/sheep.urdf',
            'name_npc': 'sheep',
            'num_cols': 1,
            'num_rows': 1,
            'sheep_movement_randomness': 0.0,
            'sheep_movement_range': [2.0, 2.0, 0],
            'sheep_movement_scale': 0.2
        }),
        'env': SEC(True, {
            'env_name': 'go1sheep',
            'episode_length_s': 15,
            'num_agents': 2,
            'num_envs': 1,
            'num_npcs': 1
        }),
        'rewards': SEC(True, {
            'scales': SEC(False, {
                'contact_punishment_scale': 0,
                'mixed_sheep_reward_scale': 0,
                'sheep_movement_reward_scale': 2,
                'sheep_pos_var_exp_punishment_scale': 0,
                'sheep_pos_var_lin_punishment_scale': 0,
                'success_reward_scale': 1
            })
        }),
        'terrain': SEC(True, {
            'BarrierTrack_kwargs': {'add_perlin_noise': False, 'border_height': 0.0, 'border_perlin_noise': False, 'curriculum_perlin': False, 'engaging_next_threshold': 1.2, 'gate': {'block_length': 1.0, 'depth': 0.1, 'offset': (0, 0), 'random': (0, 0.5), 'width': 0.8}, 'init': {'block_length': 1.5, 'border_width': 0.0, 'offset': (0.5, 0), 'room_size': (1.0, 1.95)}, 'no_perlin_threshold': 0.06, 'options': ['init', 'plane', 'gate', 'plane', 'wall'], 'plane': {'block_length': 3.0}, 'track_width': 4.0, 'virtual_terrain': False, 'wall': {'block_length': 0.1}, 'wall_height': 0.5, 'wall_thickness': 0.04},
            'num_cols': 1,
            'num_rows': 1
        })
    }),
    'NineSheepCfg': ('_SheepCommon', {
        'asset': SEC(True, {
            'dis_sheep': (1.5, 1.5),
            'file_npc': '{LEGGED_GYM_ROOT_DIR}/resources/objects/sheep.urdf',
            'name_npc': 'sheep',
            'num_cols': 3,
            'num_rows': 3,
            'sheep_movement_randomness': 0.1,
            'sheep_movement_range': [2.0, 2.0, 0],
            'sheep_movement_scale': 0.2
        }),
        'env': SEC(True, {
            'env_name': 'go1sheep',
            'episode_length_s': 15,
            'num_agents': 2,
            'num_envs': 35,
            'num_npcs': 9
        }),
        'rewards': SEC(True, {
            'scales': SEC(False, {
                'contact_punishment_scale': 0,
                'mixed_sheep_reward_scale': 1,
                'sheep_movement_reward_scale': 0,
                'sheep_pos_var_exp_punishment_scale': 0,
                'sheep_pos_var_lin_punishment_scale': 0,
                'success_reward_scale': 0
            })
        }),
        'terrain': SEC(True, {
            'BarrierTrack_kwargs': {'add_perlin_noise': False, 'border_height': 0.0, 'border_perlin_noise': False, 'curriculum_perlin': False, 'engaging_next_threshold': 1.2, 'gate': {'block_length': 1.0, 'depth': 0.1, 'offset': (0, 0), 'random': (0, 1), 'width': 1.5}, 'init': {'block_length': 2, 'border_width': 0.0, 'offset': (0.5, 0), 'room_size': (1.0, 3)}, 'no_perlin_threshold': 0.06, 'options': ['init', 'plane', 'gate', 'plane', 'wall'], 'plane': {'block_length': 6.0}, 'track_width': 6.0, 'virtual_terrain': False, 'wall': {'block_length': 0.1}, 'wall_height': 0.5, 'wall_thickness': 0.04},
            'num_cols': 7,
            'num_rows': 5
        })
    }),
    'Go1FootballDefenderCfg': ('Go1Cfg', {
        'asset': SEC(True, {
            'file_npc': '{LEGGED_GYM_ROOT_DIR}/resources/objects/ball.urdf',
            'fix_npc_base_link': False,
            'name_npc': 'ball',
            'npc_collision': True,
            'npc_gravity': True,
            'terminate_after_contacts_on': []
        }),
        'command': SEC(True, {
            'cfg': SEC(True, {
                'vel': True
            })
        }),
        'control': SEC(True, {
            'control_type': 'C'
        }),
        'domain_rand': SEC(True, {
            'init_base_pos_range': {'x': [-0.1, 0.1], 'y': [-0.1, 0.1]}
        }),
        'env': SEC(True, {
            'env_name': 'go1football',
            'episode_length_s': 20,
            'num_agents': 3,
            'num_envs': 1,
            'num_npcs': 1
        }),
        'init_state': SEC(True, {
            'init_state_class': REF("Go1Cfg", "init_state"),
            'init_states': [S(pos=[3.0, 1.0, 0.42], rot=[0.0, 0.0, 0.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0]), S(pos=[3.0, 2.0, 0.42], rot=[0.0, 0.0, 0.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0]), S(pos=[9.0, -3.0, 0.42], rot=[0.0, 0.0, 1.0, 0.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0])],
            'init_states_npc': [S(pos=[5.0, -2.1, 0.3], rot=[0.0, 0.0, 0.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0])],
            'multi_init_state': True
        }),
        'rewards': SEC(True, {
            'scales': SEC(False, {
                'ball_gate_distance_reward_scale': 3,
                'goal_reward_scale': 10
            })
        }),
        'termination': SEC(True, {
            'check_obstacle_conditioned_threshold': False,
            'termination_terms': ['roll', 'pitch']
        }),
        'terrain': SEC(True, {
            'BarrierTrack_kwargs': {'add_perlin_noise': False, 'border_height': 0.0, 'border_perlin_noise': False, 'curriculum_perlin': False, 'engaging_next_threshold': 1.2, 'gate': {'block_length': 1.0, 'depth': 1.0, 'offset': (0, 0), 'random': (0, 0.0), 'width': 2.0}, 'init': {'block_length': 1.0, 'border_width': 0.0, 'offset': (0.5, 0), 'room_size': (0, 3.0)}, 'no_perlin_threshold': 0.06, 'options': ['init', 'gate', 'plane', 'gate', 'wall'], 'plane': {'block_length': 10.0}, 'track_width': 9.0, 'virtual_terrain': False, 'wall': {'block_length': 0.1}, 'wall_height': 1.0, 'wall_thickness': 0.04},
            'num_cols': 1,
            'num_rows': 1
        }),
        'viewer': SEC(True, {
            'lookat': [6.0, 5.0, 0.0],
            'pos': [2.0, 2.0, 2.0]
        })
    }),
    'Go1Football1vs1Cfg': ('Go1Cfg', {
        'asset': SEC(True, {
            'file_npc': '{LEGGED_GYM_ROOT_DIR}/resources/objects/ball.urdf',
            'fix_npc_base_link': False,
            'name_npc': 'ball',
            'npc_collision': True,
            'npc_gravity': True,
            'terminate_after_contacts_on': []
        }),
        'command': SEC(True, {
            'cfg': SEC(True, {
                'vel': True
            })
        }),
        'control': SEC(True, {
            'control_type': 'C'
        }),
        'domain_rand': SEC(True, {
            'init_base_pos_range': {'x': [-0.1, 0.1], 'y': [-0.1, 0.1]}
        }),
        'env': SEC(True, {
            'env_name': 'go1football',
            'episode_length_s': 1,
            'num_agents': 2,
            'num_envs': 1,
            'num_npcs': 1
        }),
        'init_state': SEC(True, {
            'init_state_class': REF("Go1Cfg", "init_state"),
            'init_states': [S(pos=[3.0, 0.0, 0.42], rot=[0.0, 0.0, 0.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0]), S(pos=[9.0, 0.0, 0.42], rot=[0.0, 0.0, 1.0, 0.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0])],
            'init_states_npc': [S(pos=[7.0, 0.0, 0.2], rot=[0.0, 0.0, 0.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0])],
            'multi_init_state': True
        }),
        'rewards': SEC(True, {
            'scales': SEC(False, {
                'goal_reward_scale': 1
            })
        }),
        'termination': SEC(True, {
            'check_obstacle_conditioned_threshold': False,
            'termination_terms': ['roll', 'pitch']
        }),
        'terrain': SEC(True, {
            'BarrierTrack_kwargs': {'add_perlin_noise': False, 'border_height': 0.0, 'border_perlin_noise': False, 'curriculum_perlin': False, 'engaging_next_threshold': 1.2, 'gate': {'block_length': 1.0, 'depth': 1.0, 'offset': (0, 0), 'random': (0, 0.0), 'width': 2.0}, 'init': {'block_length': 1.0, 'border_width': 0.0, 'offset': (0.5, 0), 'room_size': (0.0, 0.0)}, 'no_perlin_threshold': 0.06, 'options': ['init', 'gate', 'plane', 'gate', 'wall'], 'plane': {'block_length': 10.0}, 'track_width': 9.0, 'virtual_terrain': False, 'wall': {'block_length': 0.1}, 'wall_height': 1.0, 'wall_thickness': 0.04},
            'num_cols': 1,
            'num_rows': 1
        }),
        'viewer': SEC(True, {
            'lookat': [6.0, 5.0, 0.0],
            'pos': [2.0, 2.0, 2.0]
        })
    }),
    'Go1Football2vs2Cfg': ('Go1Cfg', {
        'asset': SEC(True, {
            'file_npc': '{LEGGED_GYM_ROOT_DIR}/resources/objects/ball.urdf',
            'fix_npc_base_link': False,
            'name_npc': 'ball',
            'npc_collision': True,
            'npc_gravity': True,
            'terminate_after_contacts_on': []
        }),
        'command': SEC(True, {
            'cfg': SEC(True, {
                'vel': True
            })
        }),
        'control': SEC(True, {
            'control_type': 'C'
        }),
        'domain_rand': SEC(True, {
            'init_base_pos_range': {'x': [-0.1, 0.1], 'y': [-0.1, 0.1]}
        }),
        'env': SEC(True, {
            'env_name': 'go1football',
            'episode_length_s': 20,
            'num_agents': 4,
            'num_envs': 1,
            'num_npcs': 1
        }),
        'init_state': SEC(True, {
            'init_state_class': REF("Go1Cfg", "init_state"),
            'init_states': [S(pos=[3.0, 2.0, 0.42], rot=[0.0, 0.0, 0.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0]), S(pos=[3.0, -2.0, 0.42], rot=[0.0, 0.0, 0.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0]), S(pos=[9.0, 2.0, 0.42], rot=[0.0, 0.0, 1.0, 0.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0]), S(pos=[9.0, -2.0, 0.42], rot=[0.0, 0.0, 1.0, 0.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0])],
            'init_states_npc': [S(pos=[7.0, 0.0, 0.2], rot=[0.0, 0.0, 0.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0])],
            'multi_init_state': True
        }),
        'rewards': SEC(True, {
            'scales': SEC(False, {
                'goal_reward_scale': 1
            })
        }),
        'termination': SEC(True, {
            'check_obstacle_conditioned_threshold': False,
            'termination_terms': ['roll', 'pitch']
        }),
        'terrain': SEC(True, {
            'BarrierTrack_kwargs': {'add_perlin_noise': False, 'border_height': 0.0, 'border_perlin_noise': False, 'curriculum_perlin': False, 'engaging_next_threshold': 1.2, 'gate': {'block_length': 1.0, 'depth': 1.0, 'offset': (0, 0), 'random': (0, 0.0), 'width': 2.0}, 'init': {'block_length': 1.0, 'border_width': 0.0, 'offset': (0.5, 0), 'room_size': (0, 0)}, 'no_perlin_threshold': 0.06, 'options': ['init', 'gate', 'plane', 'gate', 'wall'], 'plane': {'block_length': 10.0}, 'track_width': 9.0, 'virtual_terrain': False, 'wall': {'block_length': 0.1}, 'wall_height': 1.0, 'wall_thickness': 0.04},
            'num_cols': 1,
            'num_rows': 1
        }),
        'viewer': SEC(True, {
            'lookat': [6.0, 5.0, 0.0],
            'pos': [2.0, 2.0, 2.0]
        })
    }),
    'Go1SeesawCfg': ('Go1Cfg', {
        'asset': SEC(True, {
            'file_npc': '{LEGGED_GYM_ROOT_DIR}/resources/objects/seesaw.urdf',
            'fix_npc_base_link': True,
            'name_npc': 'seesaw',
            'npc_collision': True,
            'npc_gravity': True
        }),
        'command': SEC(True, {
            'cfg': SEC(True, {
                'vel': True
            })
        }),
        'control': SEC(True, {
            'control_type': 'C',
            'default_command': SEC(True, {
                'gait': 'pacing'
            })
        }),
        'domain_rand': SEC(True, {
            'init_base_pos_range': {'x': [-0.1, 0.1], 'y': [-0.1, 0.1]},
            'init_npc_base_pos_range': None
        }),
        'env': SEC(True, {
            'env_name': 'go1seesaw',
            'episode_length_s': 10,
            'num_actions_npc': 1,
            'num_agents': 2,
            'num_envs': 1,
            'num_npcs': 1
        }),
        'init_state': SEC(True, {
            'default_npc_joint_angles': [-0.2],
            'init_state_class': REF("Go1Cfg", "init_state"),
            'init_states': [S(pos=[0.0, 0.0, 0.42], rot=[0.0, 0.0, 0.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0]), S(pos=[0.0, 0.0, 0.42], rot=[0.0, 0.0, 0.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0])],
            'init_states_npc': [S(pos=[8.0, 0.0, 1.0], rot=[0.0, 0.0, 0.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0])],
            'multi_init_state': True
        }),
        'obs': SEC(True, {
            'cfgs': SEC(True, {
                'env_info': False
            })
        }),
        'rewards': SEC(True, {
            'scales': SEC(False, {
                'agent_distance_punishment_scale': -0.25,
                'contact_punishment_scale': -2,
                'fall_punishment_scale': -2,
                'height_reward_scale': 1,
                'success_reward_scale': 10,
                'x_movement_reward_scale': 5,
                'y_punishment_scale': -0.5
            })
        }),
        'termination': SEC(True, {
            'check_obstacle_conditioned_threshold': False,
            'termination_terms': ['roll', 'pitch', 'z_low']
        }),
        'terrain': SEC(True, {
            'BarrierTrack_kwargs': {'add_perlin_noise': False, 'border_height': 0.0, 'border_perlin_noise': False, 'curriculum_perlin': False, 'engaging_next_threshold': 1.2, 'gate': {'block_length': 1.6, 'depth': 0.1, 'offset': (0.4, 0), 'random': (0.0, 0.0), 'width': 0.5}, 'init': {'block_length': 2.0, 'border_width': 0.0, 'offset': (0, 0), 'room_size': (1.0, 1.5)}, 'no_perlin_threshold': 0.06, 'options': ['init', 'plane', 'wall'], 'plane': {'block_length': 8.0}, 'track_width': 3.0, 'virtual_terrain': False, 'wall': {'block_length': 0.1}, 'wall_height': 0.5, 'wall_thickness': 0.04},
            'num_cols': 1,
            'num_rows': 1
        }),
        'viewer': SEC(True, {
            'lookat': [4.0, 2.0, 0.0],
            'pos': [0.0, -2.0, 4.0]
        })
    }),
    'Go1PushboxCfg': ('Go1Cfg', {
        'asset': SEC(True, {
            'file_npc': '{LEGGED_GYM_ROOT_DIR}/resources/objects/box.urdf',
            'fix_npc_base_link': False,
            'name_npc': 'box',
            'npc_collision': True,
            'npc_gravity': True,
            'terminate_after_contacts_on': []
        }),
        'command': SEC(True, {
            'cfg': SEC(True, {
                'vel': True
            })
        }),
        'control': SEC(True, {
            'control_type': 'C'
        }),
        'domain_rand': SEC(True, {
            'init_base_pos_range': {'x': [-0.1, 0.1], 'y': [-0.1, 0.1]},
            'init_npc_base_pos_range': {'x': [-0.5, 0.5], 'y': [-0.5, 0.5]},
            'push_robots': False
        }),
        'env': SEC(True, {
            'env_name': 'go1pushbox',
            'episode_length_s': 15,
            'num_agents': 2,
            'num_envs': 1,
            'num_npcs': 1
        }),
        'init_state': SEC(True, {
            'init_state_class': REF("Go1Cfg", "init_state"),
            'init_states': [S(pos=[0.0, 0.0, 0.42], rot=[0.0, 0.0, 0.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0]), S(pos=[0.0, 0.0, 0.42], rot=[0.0, 0.0, 0.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0])],
            'init_states_npc': [S(pos=[2.5, 0.0, 0.6], rot=[0.0, 0.0, 0.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0])],
            'multi_init_state': True
        }),
        'rewards': SEC(True, {
            'scales': SEC(False, {
                'box_x_movement_reward_scale': 10
            })
        }),
        'termination': SEC(True, {
            'check_obstacle_conditioned_threshold': False,
            'termination_terms': ['roll', 'pitch']
        }),
        'terrain': SEC(True, {
            'BarrierTrack_kwargs': {'add_perlin_noise': False, 'border_height': 0.0, 'border_perlin_noise': False, 'curriculum_perlin': False, 'engaging_next_threshold': 1.2, 'gate': {'block_length': 5.0, 'depth': 0.1, 'offset': (0, 0), 'random': (0, 0.5), 'width': 1.5}, 'init': {'block_length': 2.0, 'border_width': 0.0, 'offset': (0, 0), 'room_size': (1.0, 2.5)}, 'no_perlin_threshold': 0.06, 'options': ['init', 'gate', 'wall'], 'plane': {'block_length': 3.0}, 'track_width': 5.0, 'virtual_terrain': False, 'wall': {'block_length': 0.1}, 'wall_height': 0.5, 'wall_thickness': 0.04},
            'num_cols': 1,
            'num_rows': 1
        }),
        'viewer': SEC(True, {
            'lookat': [4.0, 6.0, 0.0],
            'pos': [0.0, 6.0, 5.0]
        })
    }),
    'Go1RotationCfg': ('Go1Cfg', {
        'asset': SEC(True, {
            'file_npc': '{LEGGED_GYM_ROOT_DIR}/resources/objects/rotation_door.urdf',
            'fix_npc_base_link': True,
            'name_npc': 'rotation',
            'npc_collision': True,
            'terminate_after_contacts_on': []
        }),
        'command': SEC(True, {
            'cfg': SEC(True, {
                'vel': True
            })
        }),
        'control': SEC(True, {
            'control_type': 'C'
        }),
        'domain_rand': SEC(True, {
            'init_base_pos_range': None,
            'init_npc_base_pos_range': None
        }),
        'env': SEC(True, {
            'env_name': 'go1rotationCfg',
            'episode_length_s': 5,
            'num_actions_npc': 1,
            'num_agents': 2,
            'num_envs': 1,
            'num_npcs': 1
        }),
        'init_state': SEC(True, {
            'init_state_class': REF("Go1Cfg", "init_state"),
            'init_states': [S(pos=[0.5, -1.0, 0.42], rot=[0.0, 0.0, 0.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0]), S(pos=[0.5, 1.0, 0.42], rot=[0.0, 0.0, 0.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0])],
            'init_states_npc': [S(pos=[2.59, -0.01, 0.04], rot=[0.0, 0.0, 0.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0])],
            'multi_init_state': True
        }),
        'rewards': SEC(True, {
            'scales': SEC(False, {
                'distance_reward_scale': 1,
                'punishment_scale': 1,
                'success_reward_scale': 10
            })
        }),
        'termination': SEC(True, {
            'termination_terms': ['roll', 'pitch', 'z_low', 'z_high']
        }),
        'terrain': SEC(True, {
            'BarrierTrack_kwargs': {'add_perlin_noise': False, 'border_height': 0.0, 'border_perlin_noise': False, 'curriculum_perlin': False, 'engaging_next_threshold': 1.2, 'gate': {'block_length': 5.0, 'depth': 0.1, 'offset': (0, 0), 'random': (0, 0), 'width': 2.0}, 'init': {'block_length': 0, 'border_width': 0.0, 'offset': (0, 0), 'room_size': (0.0, 0.0)}, 'no_perlin_threshold': 0.06, 'options': ['init', 'wall', 'gate', 'wall'], 'plane': {'block_length': 3.0}, 'randomize_obstacle_order': False, 'rotation': {'block_length': 5, 'depth': 0.1, 'offset': (0, 0), 'wide_px': (0.84, 0.2)}, 'track_width': 3.5, 'virtual_terrain': False, 'wall': {'block_length': 0.1}, 'wall_height': 0.85, 'wall_thickness': 0.04},
            'num_cols': 1,
            'num_rows': 1,
            'x_limits': [5.0],
            'y_limits': [-1.5, 1.5]
        }),
        'viewer': SEC(True, {
            'lookat': [13.0, 20.0, 0.0],
            'pos': [12.0, 20.0, 20.0]
        })
    }),
    'Go1TugCfg': ('Go1Cfg', {
        'asset': SEC(True, {
            'file_npc': '{LEGGED_GYM_ROOT_DIR}/resources/objects/cylinder.urdf',
            'fix_npc_base_link': True,
            'name_npc': 'circular',
            'terminate_after_contacts_on': []
        }),
        'command': SEC(True, {
            'cfg': SEC(True, {
                'vel': True
            })
        }),
        'control': SEC(True, {
            'control_type': 'C'
        }),
        'domain_rand': SEC(True, {
            'init_base_pos_range': {'x': [-1.0, 1.0], 'y': [-0.0, 0.0]},
            'init_dof_pos_ratio_range': None,
            'init_npc_base_pos_range': None,
            'push_robots': False
        }),
        'env': SEC(True, {
            'env_name': 'go1tug',
            'env_type': 1,
            'episode_length_s': 15,
            'num_actions_npc': 1,
            'num_agents': 2,
            'num_envs': 1,
            'num_npcs': 1
        }),
        'init_state': SEC(True, {
            'init_state_class': REF("Go1Cfg", "init_state"),
            'init_states': [S(pos=[1.6, 2.5, 0.34], rot=[0.0, 0.0, -1.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0]), S(pos=[1.6, -2.5, 0.34], rot=[0.0, 0.0, 1.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0])],
            'init_states_npc': [S(pos=[1.6, 0.0, 0.0], rot=[0.0, 0.0, 0.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0])],
            'multi_init_state': True
        }),
        'rewards': SEC(True, {
            'scales': SEC(False, {
                'pos_punishment_scale': 2,
                'pos_reward_scale': 2,
                'punishment_reward_scale': 10,
                'success_reward_scale': 10
            })
        }),
        'termination': SEC(True, {
            'termination_terms': ['roll', 'pitch', 'z_low', 'z_high']
        }),
        'terrain': SEC(True, {
            'BarrierTrack_kwargs': {'add_perlin_noise': False, 'border_height': 0.0, 'border_perlin_noise': False, 'curriculum_perlin': False, 'engaging_next_threshold': 1.2, 'gate': {'block_length': 1.6, 'depth': 0.1, 'offset': (0.4, 0), 'random': (0.0, 0.0), 'width': 0.5}, 'init': {'block_length': 0.0, 'border_width': 0.0, 'offset': (0, 0), 'room_size': (0.0, 0.0)}, 'no_perlin_threshold': 0.06, 'options': ['init', 'wall', 'plane', 'wall'], 'plane': {'block_length': 3.0}, 'randomize_obstacle_order': False, 'track_width': 6.0, 'virtual_terrain': False, 'wall': {'block_length': 0.1}, 'wall_height': 1.0, 'wall_thickness': 0.04},
            'TerrainPerlin_kwargs': {'frequency': 10, 'zScale': [0.05, 0.1]},
            'num_cols': 1,
            'num_rows': 1
        }),
        'viewer': SEC(True, {
            'lookat': [4.0, 11.0, 0.0],
            'pos': [0.0, 11.0, 5.0]
        })
    }),
    'Go1BridgeCfg': ('Go1Cfg', {
        'asset': SEC(True, {
            'file_npc': '{LEGGED_GYM_ROOT_DIR}/resources/objects/bridge/urdf/bridge.urdf',
            'fix_npc_base_link': True,
            'name_npc': 'bridge',
            'terminate_after_contacts_on': []
        }),
        'command': SEC(True, {
            'cfg': SEC(True, {
                'vel': True
            })
        }),
        'control': SEC(True, {
            'control_type': 'C'
        }),
        'domain_rand': SEC(True, {
            'init_base_pos_range': {'x': [-0.1, 0.1], 'y': [-0.1, 0.1]},
            'init_dof_pos_ratio_range': None,
            'init_npc_base_pos_range': None,
            'push_robots': False
        }),
        'env': SEC(True, {
            'env_name': 'go1bridge',
            'env_type': 1,
            'episode_length_s': 20,
            'num_agents': 2,
            'num_envs': 1,
            'num_npcs': 1
        }),
        'init_state': SEC(True, {
            'init_state_class': REF("Go1Cfg", "init_state"),
            'init_states': [S(pos=[2.0, 0.0, 1.4], rot=[0.0, 0.0, 0.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0]), S(pos=[7.5, 0.0, 1.4], rot=[0.0, 0.0, 1.0, 0.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0])],
            'init_states_npc': [S(pos=[5.0, 0.0, 0.72], rot=[0.0, 0.0, 0.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0])],
            'multi_init_state': True
        }),
        'rewards': SEC(True, {
            'scales': SEC(False, {
                'punishment_scale': 1,
                'success_reward_scale': 10,
                'target_reward_scale': 1
            })
        }),
        'termination': SEC(True, {
            'z_low_kwargs': {'threshold': 0.3}
        }),
        'terrain': SEC(True, {
            'BarrierTrack_kwargs': {'add_perlin_noise': False, 'border_height': 0.0, 'border_perlin_noise': False, 'curriculum_perlin': False, 'engaging_next_threshold': 1.2, 'gate': {'block_length': 1.6, 'depth': 0.1, 'offset': (0.4, 0), 'random': (0.0, 0.0), 'width': 0.5}, 'init': {'block_length': 0.5, 'border_width': 0.0, 'offset': (0, 0), 'room_size': (0.0, 0.0)}, 'no_perlin_threshold': 0.06, 'options': ['init', 'wall', 'plane', 'wall'], 'plane': {'block_length': 10.0}, 'randomize_obstacle_order': False, 'track_width': 6, 'virtual_terrain': False, 'wall': {'block_length': 0.1}, 'wall_height': 0.01, 'wall_thickness': 0.04},
            'TerrainPerlin_kwargs': {'frequency': 10, 'zScale': [0.05, 0.1]},
            'num_cols': 1,
            'num_rows': 1
        }),
        'viewer': SEC(True, {
            'lookat': [4.0, 3.0, 0.0],
            'pos': [0.0, 3.0, 5.0]
        })
    }),
    'Go1WrestlingCfg': ('Go1Cfg', {
        'asset': SEC(True, {
            'file_npc': '{LEGGED_GYM_ROOT_DIR}/resources/objects/wrestling_field/urdf/wrestling.urdf',
            'fix_npc_base_link': True,
            'name_npc': 'wrestling',
            'terminate_after_contacts_on': []
        }),
        'command': SEC(True, {
            'cfg': SEC(True, {
                'vel': True
            })
        }),
        'control': SEC(True, {
            'control_type': 'C'
        }),
        'domain_rand': SEC(True, {
            'init_base_pos_range': {'x': [-0.1, 0.1], 'y': [-0.1, 0.1]},
            'init_dof_pos_ratio_range': None,
            'init_npc_base_pos_range': None,
            'push_robots': False
        }),
        'env': SEC(True, {
            'env_name': 'go1wrestling',
            'env_type': 1,
            'episode_length_s': 15,
            'num_agents': 2,
            'num_envs': 1,
            'num_npcs': 1
        }),
        'init_state': SEC(True, {
            'init_state_class': REF("Go1Cfg", "init_state"),
            'init_states': [S(pos=[3.1, 1.0, 0.74], rot=[0.0, 0.0, -1.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0]), S(pos=[3.1, -1.0, 0.74], rot=[0.0, 0.0, 1.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0])],
            'init_states_npc': [S(pos=[3.1, 0.0, 0.0], rot=[0.0, 0.0, 0.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0])],
            'multi_init_state': True
        }),
        'rewards': SEC(True, {
            'scales': SEC(False, {
                'punishment_scale': 1,
                'success_reward_scale': 10
            })
        }),
        'termination': SEC(True, {
            'termination_terms': ['roll', 'pitch', 'z_low'],
            'z_low_kwargs': {'threshold': 0.3}
        }),
        'terrain': SEC(True, {
            'BarrierTrack_kwargs': {'add_perlin_noise': False, 'border_height': 0.0, 'border_perlin_noise': False, 'curriculum_perlin': False, 'engaging_next_threshold': 1.2, 'gate': {'block_length': 1.6, 'depth': 0.1, 'offset': (0.4, 0), 'random': (0.0, 0.0), 'width': 0.5}, 'init': {'block_length': 0.0, 'border_width': 0.0, 'offset': (0, 0), 'room_size': (0.0, 0.0)}, 'no_perlin_threshold': 0.06, 'options': ['init', 'plane'], 'plane': {'block_length': 7}, 'randomize_obstacle_order': False, 'track_width': 6, 'virtual_terrain': False, 'wall': {'block_length': 0.1}, 'wall_height': 0.001, 'wall_thickness': 0.04},
            'TerrainPerlin_kwargs': {'frequency': 10, 'zScale': [0.05, 0.1]},
            'num_cols': 1,
            'num_rows': 1
        }),
        'viewer': SEC(True, {
            'lookat': [4.0, 3.0, 0.0],
            'pos': [0.0, 3.0, 5.0]
        })
    }),
}
