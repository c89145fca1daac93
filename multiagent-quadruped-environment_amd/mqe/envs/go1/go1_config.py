"""Go1 robot + hierarchical "C" control defaults (values: reference mqe/envs/go1/go1_config.py:34-311)."""
from mqe.envs.base.legged_robot_config import LeggedRobotCfg  # noqa: F401
from mqe.envs.field.legged_robot_field_config import LeggedRobotFieldCfg

_LEGS_FRONT, _LEGS_REAR = ("FL", "FR"), ("RL", "RR")


class Go1Cfg(LeggedRobotFieldCfg):
    class env(LeggedRobotFieldCfg.env):
        use_lin_vel = True
        num_envs = 256
        num_observations = 235
        num_privileged_obs = None
        num_actions = 12
        env_spacing = 3.0
        send_timeouts = True
        episode_length_s = 5
        record_video = False
        record_actor_id = 0
        recording_width_px = 360
        recording_height_px = 240
        recording_mode = "COLOR"

    class asset:
        file = "{LEGGED_GYM_ROOT_DIR}/resources/robots/go1/urdf/go1.urdf"
        files = ["{LEGGED_GYM_ROOT_DIR}/resources/robots/go1/urdf/go1 %s.urdf" % c for c in ("blue", "green", "red", "orange")]
        name = "go1"
        foot_name = "foot"
        penalize_contacts_on = ["base", "thigh"]
        terminate_after_contacts_on = ["base"]
        disable_gravity = False
        collapse_fixed_joints = True
        fix_base_link = False
        default_dof_drive_mode = 3
        self_collisions = 0
        replace_cylinder_with_capsule = True
        flip_visual_attachments = False
        density = 0.001
        angular_damping = 0.0
        linear_damping = 0.0
        max_angular_velocity = 1000.0
        max_linear_velocity = 1000.0
        armature = 0.0
        thickness = 0.01

    class init_state(LeggedRobotFieldCfg.init_state):
        pos = [0.0, 0.0, 0.42]
        default_joint_angles = {}
        for _l in ("FR", "FL", "RR", "RL"):
            default_joint_angles[_l + "_hip_joint"] = 0.1 if _l[1] == "L" else -0.1
        for _l in ("FL", "RL", "FR", "RR"):
            default_joint_angles[_l + "_thigh_joint"] = 0.8 if _l[0] == "F" else 1.0
        for _l in ("FL", "RL", "FR", "RR"):
            default_joint_angles[_l + "_calf_joint"] = -1.5
        del _l

    class normalization(LeggedRobotFieldCfg.normalization):
        clip_actions = 10.0

    class control(LeggedRobotFieldCfg.control):
        control_type = "C"
        stiffness = {"joint": 20.0}
        damping = {"joint": 0.5}
        action_scale = 0.25
        torque_limits = [20.0, 20.0, 25.0] * 4
        computer_clip_torque = True
        motor_clip_torque = False
        decimation = 4
        hip_scale_reduction = 0.5
        locomotion_policy_dir = "./mqe/utils/locomotion_checkpoints/walk_these_ways"
        actuator_network_path = "./resources/actuator_nets"

        class default_command:
            lin_vel_x = 1.0
            lin_vel_y = -0.0
            ang_vel = -0.0
            body_height = 0.0
            gait_freq = 3.0
            gait = "trotting"
            footswing_height = 0.08
            body_pitch = 0.0
            body_roll = 0.0
            stance_width = 0.25
            stance_length = 0.428
            aux_reward = 0.0

        class obs_scales:
            lin_vel = 2.0
            ang_vel = 0.25
            dof_pos = 1.0
            dof_vel = 0.05
            body_height = 2.0
            gait_phase = 1.0
            gait_freq = 1.0
            footswing_height = 0.15
            body_pitch = 0.3
            body_roll = 0.3
            aux_reward = 1.0
            compliance = 1.0
            stance_width = 1.0
            stance_length = 1.0

    class command:
        gaits = {"pronking": [0, 0, 0], "trotting": [0.5, 0, 0], "bounding": [0, 0.5, 0], "pacing": [0, 0, 0.5]}
        curriculum = False
        max_curriculum = 1.0
        num_commands = 4
        resampling_time = 10.0
        heading_command = True

        class cfg:
            vel = False
            body_height = False
            body_pose = False
            gait_freq = False
            gait = False
            footswing_height = False
            stance_width = False
            stance_length = False
            aux_reward = False

        class ranges:
            lin_vel_x = [-1.0, 1.0]
            lin_vel_y = [-1.0, 1.0]
            ang_vel_yaw = [-1, 1]
            heading = [-3.14, 3.14]

    class termination:
        termination_terms = ["roll", "pitch", "z_low", "z_high"]
        roll_kwargs = dict(threshold=0.8)
        pitch_kwargs = dict(threshold=1.6)
        z_low_kwargs = dict(threshold=0.08)
        z_high_kwargs = dict(threshold=1.5)
        out_of_track_kwargs = dict(threshold=1.0)

    class domain_rand(LeggedRobotFieldCfg.domain_rand):
        randomize_com = False

        class com_range:
            x = [-0.05, 0.15]
            y = [-0.1, 0.1]
            z = [-0.05, 0.05]

        randomize_motor = False
        leg_motor_strength_range = [0.9, 1.1]
        randomize_base_mass = False
        added_mass_range = [-1.0, 3.0]
        randomize_friction = False
        friction_range = [0.05, 4.5]
        randomize_lag_timesteps = False
        lag_timesteps = 6
        init_base_pos_range = dict(x=[0.1, 0.1], y=[-0.1, 0.1])
        init_dof_pos_ratio_range = [0.7, 1.3]
        init_npc_base_pos_range = dict(x=[-0.2, 0.2], y=[-0.2, 0.2])
        push_robots = False

    class obs:
        class cfgs:
            base_pos = True
            base_quat = True
            dof_pos = True
            dof_vel = True
            lin_vel = True
            ang_vel = True
            projected_gravity = True
            base_rpy = True
            contact_states = False
            command = True
            height_command = False
            gait_commands = False
            timing_parameter = False
            clock_inputs = False
            last_action = True
            last_last_action = True
            imu = False
            depth_image = False
            rgb_image = False
            env_info = True

        class scales:
            base_pos = 1.0
            base_quat = 1.0
            segmentation_image = 1.0
            rgb_image = 1.0
            depth_image = 1.0

        def keys(self):
            return [k for k in dir(self.cfgs) if getattr(self.cfgs, k) == True and k]  # noqa: E712

    class privileged_obs:
        class cfgs:
            pass

        def keys(self):
            return [k for k in dir(self.cfgs) if getattr(self.cfgs, k) == True and k]  # noqa: E712

    class rewards(LeggedRobotFieldCfg.rewards):
        soft_dof_pos_limit = 0.9
        base_height_target = 0.25

        class scales(LeggedRobotFieldCfg.rewards.scales):
            torques = -0.0002
            dof_pos_limits = -10.0

    class viewer(LeggedRobotFieldCfg.viewer):
        pos = [0.0, 11.0, 5.0]
        lookat = [4.0, 11.0, 0.0]
