"""ctypes mirror of include/mqe_hip.h (keep in sync; tests/test_abi.py checks sizes against the built library)."""
import ctypes as C

ABI_VERSION = 16
MAX_SPHERES, NBODY, NREP, NDOF = 64, 13, 17, 12
MAX_SELF_PAIRS = 384
MAX_PRIMS = 20
PRIM_SPHERE, PRIM_CAPSULE, PRIM_BOX = 0, 1, 2
MAX_AGENTS, MAX_NPCS, FRAME, HIST, MAX_LAYERS, MAX_REWARD_TERMS = 4, 16, 72, 30, 6, 12
OBS_BAG = 74

TASK = dict(plain=0, gate=1, sheep=2, seesaw=3, football_defender=4, pushbox=5, rotation=6, bridge=7, wrestling=8, tug=9)
NPC = dict(none=0, ball=1, sheep=2, seesaw=3, box=4, rotation=3, bridge=5, wrestling=5, circular=3)     # the revolving door shares the seesaw's fixed-base + 1-dof-link structure
CTRL = dict(C=0, P=1, V=2, T=3)
TERM = dict(roll=1, pitch=2, z_low=4, z_high=8)
POST_FRAME, POST_NPC, POST_RESET, POST_OBS, POST_WRAPPER, POST_ALL, POST_WRAPPER_LEVEL = 1, 2, 4, 8, 16, 31, 32      # mqe_post_physics_stage

(T_ROOT_STATE, T_DOF_STATE, T_CONTACT_FORCE, T_TORQUES, T_ACTIONS, T_LAST_ACTIONS, T_LOCOMOTION_OBS, T_HISTORY,
 T_LAST_LOCO_ACTION, T_LAST_TWO_LOCO_ACTION, T_ACT_HIST, T_GAIT_INDICES, T_CLOCK_INPUTS, T_BASE_LIN_VEL,
 T_BASE_ANG_VEL, T_PROJECTED_GRAVITY, T_BASE_QUAT, T_EPISODE_LENGTH, T_RESET_BUF, T_COLLIDE_BUF, T_TIME_OUT_BUF,
 T_R_TERM, T_P_TERM, T_Z_HIGH_TERM, T_OBS_BAG, T_WRAPPER_OBS, T_WRAPPER_REWARD, T_REWARD_SUMS, T_SHEEP_POS_AVG,
 T_SHEEP_POS_VAR, T_RESET_COUNT, T_SUBSTEP_TORQUES, T_NPC_NOISE, T_WRAPPER_PACKED, T_DOMAIN_PARAMS, T_SUBSTEP_DOF_VEL,
 T_SUBSTEP_EXCEED_DOF_POS_LIMITS, T_CONTACT_OVERFLOW, T_ENV_ORIGINS, T_TERRAIN_LEVELS, T_CONTACT_REDUCED, T_COUNT) = range(42)

# slices of one OBS_BAG row (compute_observations, reference go1.py:153-196)
BAG = dict(base_pos=(0, 3), base_rpy=(3, 6), dof_pos=(6, 18), dof_vel=(18, 30), lin_vel=(30, 33), ang_vel=(33, 36),
           last_action=(36, 48), last_last_action=(48, 60), projected_gravity=(60, 63), clock_inputs=(63, 67),
           base_quat=(67, 71))

f32, i32 = C.c_float, C.c_int32
FP = C.POINTER(C.c_float)


class Mlp(C.Structure):
    _fields_ = [("n_layers", i32), ("dims", i32 * (MAX_LAYERS + 1)), ("W", FP * MAX_LAYERS), ("b", FP * MAX_LAYERS)]


class RobotModel(C.Structure):
    _fields_ = [
        ("mass", f32 * NBODY), ("com", (f32 * 3) * NBODY), ("inertia", (f32 * 6) * NBODY),
        ("joint_offset", (f32 * 3) * NBODY), ("joint_axis", (f32 * 3) * NBODY),
        ("dof_lower", f32 * NDOF), ("dof_upper", f32 * NDOF), ("dof_vel_limit", f32 * NDOF),
        ("n_spheres", i32), ("sphere_body", i32 * MAX_SPHERES), ("sphere_reported", i32 * MAX_SPHERES), ("sphere_prim", i32 * MAX_SPHERES),
        ("sphere_center", (f32 * 3) * MAX_SPHERES), ("sphere_radius", f32 * MAX_SPHERES),
        ("n_prims", i32), ("prim_type", i32 * MAX_PRIMS), ("prim_body", i32 * MAX_PRIMS), ("prim_reported", i32 * MAX_PRIMS),
        ("prim_center", (f32 * 3) * MAX_PRIMS), ("prim_axis", (f32 * 3) * MAX_PRIMS), ("prim_half", (f32 * 3) * MAX_PRIMS),
        ("prim_bound", f32 * MAX_PRIMS), ("feature_reach", f32),
        ("n_self_pairs", i32), ("self_pair", C.c_uint16 * MAX_SELF_PAIRS),
        ("self_safe_lo", f32 * NDOF), ("self_safe_hi", f32 * NDOF),
    ]


class SimDesc(C.Structure):
    _fields_ = [
        ("abi_version", i32),
        ("num_envs", i32), ("num_agents", i32), ("num_npcs", i32), ("npc_kind", i32), ("task", i32),
        ("env_id_offset", i32), ("seed", i32),
        ("dt", f32), ("decimation", i32), ("gravity_z", f32), ("solver_iterations", i32),
        ("contact_offset", f32), ("max_depenetration_velocity", f32), ("friction", f32), ("erp", f32), ("solver_type", i32), ("velocity_iterations", i32),
        ("robot", RobotModel),
        ("npc_mass", f32), ("npc_inertia", f32), ("npc_n_spheres", i32),
        ("npc_sphere_center", (f32 * 3) * 8), ("npc_sphere_radius", f32 * 8), ("npc_box_half", f32 * 3), ("npc_contact_cap", i32),
        ("seesaw_joint_offset", f32 * 3), ("seesaw_plank_center", f32 * 3), ("seesaw_plank_half", f32 * 3),
        ("seesaw_base_half", f32 * 3),
        ("seesaw_plank_mass", f32), ("seesaw_plank_inertia_yy", f32), ("seesaw_vel_limit", f32), ("seesaw_default_angle", f32),
        ("seesaw_column_radius", f32), ("seesaw_column_length", f32), ("seesaw_theta_lo", f32), ("seesaw_theta_hi", f32),
        ("n_static_boxes", i32), ("npc_reported_bodies", i32), ("self_collision", i32), ("static_box_center", (f32 * 3) * 4), ("static_box_half", (f32 * 3) * 4),
        ("seesaw_axis", i32), ("seesaw_link_cylinder", i32),
        ("control_type", i32), ("action_scale", f32), ("hip_scale_reduction", f32), ("clip_actions", f32),
        ("torque_limits", f32 * NDOF), ("kp", f32), ("kd", f32), ("default_dof_pos", f32 * NDOF),
        ("command_obs", f32 * 70), ("cmd_lin_scale", f32), ("cmd_ang_scale", f32), ("clip_command", i32),
        ("num_command_dims", i32), ("command_src", i32 * 18), ("command_scale", f32 * 18),
        ("wall_sdf", FP), ("sdf_nx", i32), ("sdf_ny", i32),
        ("horizontal_scale", f32), ("wall_height", f32), ("ground_z", f32), ("ground_height", FP), ("wall_top", FP), ("edge_contacts", i32), ("wall_corner", FP), ("soft_dof_pos_limit", f32),
        ("env_origins", FP), ("agent_origins", FP), ("base_init_state", FP), ("npc_init_state", FP), ("gate_pos", FP),
        ("terrain_curriculum", i32), ("terrain_num_rows", i32), ("terrain_num_cols", i32), ("terrain_env_length", f32),
        ("terrain_origins", FP), ("terrain_levels", C.POINTER(i32)), ("terrain_types", C.POINTER(i32)),
        ("termination_flags", i32), ("terminate_on_base_contact", i32), ("max_episode_length", i32),
        ("roll_threshold", f32), ("pitch_threshold", f32), ("z_low_threshold", f32), ("z_high_threshold", f32),
        ("noise_mode", i32), ("dof_ratio_lo", f32), ("dof_ratio_hi", f32),
        ("has_base_pos_range", i32), ("has_npc_pos_range", i32),
        ("base_pos_x_lo", f32), ("base_pos_x_hi", f32), ("base_pos_y_lo", f32), ("base_pos_y_hi", f32),
        ("npc_pos_x_lo", f32), ("npc_pos_x_hi", f32), ("npc_pos_y_lo", f32), ("npc_pos_y_hi", f32),
        ("base_vel_lo", f32), ("base_vel_hi", f32),
        ("rand_friction", i32), ("friction_lo", f32), ("friction_hi", f32),
        ("rand_base_mass", i32), ("added_mass_lo", f32), ("added_mass_hi", f32),
        ("rand_com", i32), ("com_lo", f32 * 3), ("com_hi", f32 * 3),
        ("lag_timesteps", i32), ("push_interval", i32), ("max_push_vel_xy", f32),
        ("sheep_movement_scale", f32), ("sheep_movement_randomness", f32),
        ("reward_scale", f32 * MAX_REWARD_TERMS), ("wrapper_param", f32 * 8),
        ("actuator", Mlp), ("adaptation", Mlp), ("body", Mlp),
    ]


class TensorView(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("ndim", i32), ("shape", C.c_int64 * 4), ("dtype", i32)]


def bind(lib, prefix):
    """Declare argtypes/restypes of the entry points of include/mqe_hip.h for a loaded library."""
    H = C.c_void_p

    def fn(name, *args, res=C.c_int):
        f = getattr(lib, prefix + name)
        f.argtypes, f.restype = list(args), res
        return f
    api = {}
    api["last_error"] = fn("last_error", res=C.c_char_p)
    api["sim_create"] = fn("sim_create", C.POINTER(SimDesc), C.POINTER(H))
    api["sim_destroy"] = fn("sim_destroy", H)
    api["sim_tensor"] = fn("sim_tensor", H, C.c_int, C.POINTER(TensorView))
    return api
