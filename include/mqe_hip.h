/* mqe_hip.h -- C ABI of the MI355X-native MQE rollout engine (drop-in boundary).
 *
 * The reference (ziyanx02/multiagent-quadruped-environment) is pure Python on top of the Isaac Gym tensor API;
 * this library replaces that *inner* boundary and adds a fused fast path.  Each entry point cites the reference
 * call it stands in for.  Conventions: every function returns 0 on success or a negative code and records a
 * message retrievable with mqe_last_error(); no exceptions cross the ABI; the handle owns all device memory it
 * allocates; caller owns buffers it passes in; one handle per GPU; calls on one handle are serialised by the
 * caller; all work is enqueued on the `stream` argument (a hipStream_t passed as void*; NULL = default stream).
 *
 * Memory contract (reference mqe/envs/base/legged_robot.py:549-645, "_init_buffers"): all tensors are float32,
 * env-major:
 *   ROOT_STATE     [N, A+P, 13]  pos3, quat4 (xyzw), linvel3, angvel3 in world frame; agents first   (:567-574)
 *   DOF_STATE      [N, 12A+Dn, 2] (pos, vel); agent dofs first, NPC dofs after                        (:577-585)
 *   CONTACT_FORCE  [N, 17A+Bn, 3] net contact force per reported rigid body, agents first             (:595)
 *   TORQUES        [N, 12A]                                                                            (:605)
 * Leg/DOF order inside one robot: FL, FR, RL, RR x (hip, thigh, calf); bodies: base, then per leg hip, thigh,
 * calf, foot.
 */
#ifndef MQE_HIP_H
#define MQE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped on EVERY change of a limit, an export, a descriptor field or the meaning of a call (a binding compiled against another header is
 * refused by mqe_sim_create and can compare mqe_abi_limits() with the constants it was built with).  v15 (round 6) over v14: mqe_abi_limits
 * (new export); MQE_MAX_NPCS 16 and mqe_debug_tail_times / mqe_profile_enable(on = N) as shipped late in round 5 under v14; an env's contact
 * list holds the sum of its per-actor caps (+ 8 two-actor slots in scenes of more than four actors, at most 64) instead of min(40, .), and
 * mqe_sim_create refuses a scene whose caps exceed 64; edge_contacts bit 8 + MQE_T_CONTACT_REDUCED (new tensor): optional manifold reduction of a
 * robot's one-sided contacts to its DEEPEST eight instead of the first eight in feature order.  v16 (round 6) over v15: mqe_debug_epilogue_times
 * (new export). */
#define MQE_ABI_VERSION 16
#define MQE_MAX_SPHERES 64    /* feature points of one robot (the capsule model has 32, the exact one 60) */
#define MQE_MAX_PRIMS 20      /* collision primitives of one robot (Go1: 18) */
#define MQE_MAX_SELF_PAIRS 384
#define MQE_NBODY 13      /* dynamic bodies of one Go1 after fixed-joint collapsing */
#define MQE_NREP 17       /* reported rigid bodies of one Go1 (feet kept, go1.urdf dont_collapse) */
#define MQE_NDOF 12
#define MQE_MAX_AGENTS 4  /* one env = one 64-lane wavefront of the physics kernel, a body per lane: 13 per Go1 (+ the NPCs): a fifth robot does not fit */
#define MQE_MAX_NPCS 16   /* round 5 (was 9): a 4 x 4 sheep flock = 18 actors, 42 bodies, 84 generalized velocities per env */
#define MQE_FRAME 72      /* 70-float locomotion observation padded to 72 (16-byte rows) */
#define MQE_HIST 30       /* frames of history fed to the locomotion policy (go1.py:395) */
#define MQE_MAX_LAYERS 6
#define MQE_MAX_REWARD_TERMS 12

/* tasks whose wrapper observation / reward are evaluated in-kernel (reference mqe/envs/wrappers) */
enum { MQE_TASK_PLAIN = 0, MQE_TASK_GATE = 1, MQE_TASK_SHEEP = 2, MQE_TASK_SEESAW = 3, MQE_TASK_FOOTBALL_DEFENDER = 4, MQE_TASK_PUSHBOX = 5,
       MQE_TASK_ROTATION = 6, MQE_TASK_BRIDGE = 7, MQE_TASK_WRESTLING = 8, MQE_TASK_TUG = 9 };
/* NPC kinds (reference resources/objects/{ball,sheep,seesaw}.urdf) */
enum { MQE_NPC_NONE = 0, MQE_NPC_BALL = 1, MQE_NPC_SHEEP = 2, MQE_NPC_SEESAW = 3, MQE_NPC_BOX = 4, MQE_NPC_STATIC = 5 };
/* collision primitive types of mqe_robot_model */
enum { MQE_PRIM_SPHERE = 0, MQE_PRIM_CAPSULE = 1, MQE_PRIM_BOX = 2 };
/* control types (reference legged_robot.py:368-392 "P","V","T"; go1.py:315-354 "C") */
enum { MQE_CTRL_C = 0, MQE_CTRL_P = 1, MQE_CTRL_V = 2, MQE_CTRL_T = 3 };
/* termination terms (reference legged_robot_field.py:121-146) */
enum { MQE_TERM_ROLL = 1, MQE_TERM_PITCH = 2, MQE_TERM_Z_LOW = 4, MQE_TERM_Z_HIGH = 8 };
/* reset-noise modes: hash RNG keyed by (seed, global env id, reset count) | scripted (tests: the sequence
 * tools/gen_golden.py's ScriptedRand produces) */
enum { MQE_NOISE_HASH = 0, MQE_NOISE_SCRIPTED = 1 };

typedef struct {
  int32_t n_layers;                       /* Linear layers */
  int32_t dims[MQE_MAX_LAYERS + 1];       /* in, hidden..., out */
  const float* W[MQE_MAX_LAYERS];         /* host pointers, W[l] is (dims[l+1], dims[l]) row-major (torch Linear) */
  const float* b[MQE_MAX_LAYERS];
} mqe_mlp;

typedef struct {
  /* floating-base tree of 13 bodies: 0 = base, 1+3k+{0,1,2} = hip, thigh, calf(+foot) of leg k */
  float mass[MQE_NBODY];
  float com[MQE_NBODY][3];                /* body frame */
  float inertia[MQE_NBODY][6];            /* xx, yy, zz, xy, xz, yz about the COM, body axes */
  float joint_offset[MQE_NBODY][3];       /* joint origin in the parent frame (entry 0 unused) */
  float joint_axis[MQE_NBODY][3];         /* in the child (= parent at q=0) frame */
  float dof_lower[MQE_NDOF], dof_upper[MQE_NDOF];
  float dof_vel_limit[MQE_NDOF];          /* URDF <limit velocity> (go1.urdf:115,157,185: 50 / 28 / 28 rad/s; props["velocity"],
                                             legged_robot.py:315): the solver keeps |joint speed| below it; <= 0 = unlimited */
  /* Collision model (go1.urdf <collision> elements; ABI v12).  PRIMITIVES are the URDF's shapes themselves -- what other bodies
   * collide WITH: sphere (foot), capsule (hip cylinder as `replace_cylinder_with_capsule` makes it, go1_config.py:75; thigh and
   * calf bars as the best-fitting capsule), box (trunk go1.urdf:56, head :80; aligned with the link frame).  FEATURE POINTS
   * ("spheres": centre + radius, rigidly on a link) are what is tested AGAINST the terrain maps, the scenery, the 1-dof link,
   * the free box and the other actors' primitives: the foot spheres, the capsules' end points (capsule radius) and the boxes'
   * corners (radius 0) -- a convex body's outermost point against a plane is always one of them.  Order = priority in the
   * bounded contact list (feet, trunk, head, knees, thigh tops, hips).
   * Two models ship (assets/go1_model.json; desc builder `collision_model`): "capsule" (default; the bars as best-fitting capsules,
   * 32 feature points, surface within 6 mm of the URDF's) and "exact" (round 4: the thigh and calf bars as the URDF's own
   * link-aligned boxes, go1.urdf:170,198, and every box corner that can be outermost -- 60 feature points; the union's support
   * function to < 1 mm, tests/test_models_oracle.py). */
  int32_t n_spheres;
  int32_t sphere_body[MQE_MAX_SPHERES];
  int32_t sphere_reported[MQE_MAX_SPHERES];
  int32_t sphere_prim[MQE_MAX_SPHERES];   /* the primitive the feature point belongs to */
  float sphere_center[MQE_MAX_SPHERES][3];
  float sphere_radius[MQE_MAX_SPHERES];
  int32_t n_prims;
  int32_t prim_type[MQE_MAX_PRIMS];       /* MQE_PRIM_SPHERE / _CAPSULE / _BOX */
  int32_t prim_body[MQE_MAX_PRIMS];
  int32_t prim_reported[MQE_MAX_PRIMS];
  float prim_center[MQE_MAX_PRIMS][3];    /* link frame */
  float prim_axis[MQE_MAX_PRIMS][3];      /* capsule: half of its segment (centre +- axis), link frame; otherwise 0 */
  float prim_half[MQE_MAX_PRIMS][3];      /* box: half extents along the link axes; sphere / capsule: [0] = radius */
  float prim_bound[MQE_MAX_PRIMS];        /* radius of the bounding sphere about prim_center */
  float feature_reach;                    /* no feature point (its sphere included) is ever farther from the base origin than this,
                                             whatever the joint angles: the broad phase between two robots */
  /* self-collision candidates (asset.self_collisions = 0, go1_config.py:73 / legged_robot.py:874): (feature point, primitive)
   * pairs whose links are neither the same nor parent and child and that some pose inside the joint limits brings within 3 cm
   * (mqe/utils/urdf_model.py::_self_pair_candidates), ascending; entry = feature | primitive << 8 */
  int32_t n_self_pairs;
  uint16_t self_pair[MQE_MAX_SELF_PAIRS];
  /* a box of joint angles around the stance inside which no candidate pair comes within 4 cm (dense sampling when the model file is
   * built): a robot whose joints are all inside it skips the self-collision phase (lo > hi: no such box, never skip) */
  float self_safe_lo[MQE_NDOF], self_safe_hi[MQE_NDOF];
} mqe_robot_model;

typedef struct {
  int32_t abi_version;
  /* sizes */
  int32_t num_envs, num_agents, num_npcs, npc_kind, task;
  int32_t env_id_offset;                  /* global index of local env 0 (env sharding across GPUs) */
  int32_t seed;
  /* simulation (reference legged_robot_config.py:211-229) */
  float dt;                               /* 0.005 */
  int32_t decimation;                     /* 4 (go1_config.py:119) */
  float gravity_z;                        /* -9.81 */
  int32_t solver_iterations;              /* sim.physx.num_position_iterations (:221): sweeps over the contact list (solver type 0) resp.
                                             sub-steps of the temporal Gauss-Seidel (solver type 1) */
  float contact_offset, max_depenetration_velocity, friction;
  float erp;                              /* solver type 0 only: share of a penetration asked back per step (velocity-level bias) */
  /* sim.physx.solver_type (legged_robot_config.py:219, "0: pgs, 1: tgs"; 1 in every config of the reference):
   *   0 = projected Gauss-Seidel on velocities with the erp bias (the scheme of rounds 1-3, kept);
   *   1 = temporal Gauss-Seidel as PhysX publishes it: `solver_iterations` sub-steps of dt / n, every contact's separation re-evaluated
   *       from the motion accumulated so far, penetrations pushed out within the sub-step at <= max_depenetration_velocity (no erp)
   *       and the push-out speed taken back once the penetration is gone, positions integrated with the accumulated motion.
   * velocity_iterations = sim.physx.num_velocity_iterations (:222; 0 in every config): further sweeps that see penetrations as
   * touching and move nothing. */
  int32_t solver_type, velocity_iterations;
  /* robot */
  mqe_robot_model robot;
  /* NPC free bodies (ball / sheep): mass, isotropic inertia, spheres in the body frame */
  float npc_mass, npc_inertia;
  int32_t npc_n_spheres;
  float npc_sphere_center[8][3];   /* vs terrain; a box (MQE_NPC_BOX) carries its 8 corners here */
  float npc_sphere_radius[8];
  float npc_box_half[3];           /* MQE_NPC_BOX: half extents of the oriented box the robots' spheres collide with */
  int32_t npc_contact_cap;         /* one-sided (terrain) contacts kept per NPC: 2, a box resting on its face needs 4 */
  /* seesaw (fixed base + revolute plank), reference resources/objects/seesaw.urdf */
  float seesaw_joint_offset[3], seesaw_plank_center[3], seesaw_plank_half[3], seesaw_base_half[3];
  float seesaw_plank_mass, seesaw_plank_inertia_yy, seesaw_vel_limit, seesaw_default_angle;
  float seesaw_column_radius, seesaw_column_length;   /* static column under the platform (seesaw.urdf:90-108) */
  float seesaw_theta_lo, seesaw_theta_hi;             /* plank angles at which an end touches the ground slab */
  /* MQE_NPC_STATIC (bridge.urdf, wrestling.urdf: fixed-base scenery whose collision meshes are boxes): up to 4 world-aligned
     static boxes, centres relative to the NPC root position; the actor reports `npc_reported_bodies` rigid bodies */
  int32_t n_static_boxes, npc_reported_bodies;
  int32_t self_collision;                 /* 1: contacts between the links of one robot (the self_pair list) */
  float static_box_center[4][3], static_box_half[4][3];
  int32_t seesaw_axis;   /* joint of the 1-dof link: 1 = revolute +y (seesaw plank), 2 = revolute +z (revolving door,
                            rotation_door.urdf:44-50), 3 = prismatic +y (the tug-of-war cylinder, cylinder.urdf:37-43); same
                            fixed-base + one-link structure, `seesaw_plank_inertia_yy` holds the inertia about the axis resp.
                            the link mass */
  int32_t seesaw_link_cylinder;   /* 1: the link is an upright cylinder, radius = seesaw_plank_half[0], half height = [2] */
  /* control (reference go1_config.py:108-155) */
  int32_t control_type;
  float action_scale, hip_scale_reduction, clip_actions;
  float torque_limits[MQE_NDOF];
  float kp, kd;
  float default_dof_pos[MQE_NDOF];
  float command_obs[70];                  /* _fill_command_obs (go1.py:411-479) */
  float cmd_lin_scale, cmd_ang_scale;     /* 2.0, 0.25 */
  int32_t clip_command;                   /* 1: Go1.step re-clips to +-1 (go1.py:38); 0: defender variant */
  /* Which column of the command row feeds entry c (0..17) of the locomotion observation (Go1.preprocess_action, go1.py:64-93, with the
   * slots that _fill_command_obs assigns for command.cfg.{vel, body_height, gait_freq, footswing_height, body_pose, stance_width,
   * stance_length, aux_reward}, go1.py:411-479): -1 = the constant command_obs[c]; otherwise row[command_src[c]] * command_scale[c]
   * (control.obs_scales).  Shipped configs: entries 3, 4, 5 <- columns 0, 1, 2, everything else constant, num_command_dims = 3.
   * num_command_dims = width of the rows handed to mqe_policy_step / mqe_step_begin.  Anything but the default is served by the
   * unfused entry points and the exact-f32 layer-0 kernel; the fused wrapper-level mqe_step refuses it (the task wrappers'
   * (N, A, 3) action scaling cannot carry more columns upstream either). */
  int32_t num_command_dims;
  int32_t command_src[18];
  float command_scale[18];
  /* terrain: 2D signed distance [m] to the wall set, raster entry (i, j) at the world point (i hs, j hs) = the vertices of the BarrierTrack heightfield mesh */
  const float* wall_sdf;                  /* host pointer, [sdf_nx][sdf_ny] */
  int32_t sdf_nx, sdf_ny;
  float horizontal_scale, wall_height, ground_z;
  /* low relief of the walkable surface (Perlin noise, barrier_track.py:372-393,421-439; perlin.py:33-72): height [m] above
   * ground_z at the same raster points as wall_sdf, [sdf_nx][sdf_ny] host pointer, or NULL for the flat slab */
  const float* ground_height;
  /* walls of different heights in one scene (barrier_track.py:167-173,191-199,218-239: a (lo, hi) wall_height draws one height per
   * block): top [m] of the wall nearest to each raster point, [sdf_nx][sdf_ny] host pointer, or NULL = wall_height everywhere */
  const float* wall_top;
  /* Edge contacts (round 4): what a test of feature POINTS against shapes cannot see -- an edge cutting into a primitive between its
   * feature points.  edge_contacts is a bit mask: 1 = the vertical edges of the wall prisms (wall_corner: per raster point the world
   * (x, y) of the nearest convex corner of the wall set, [sdf_nx][sdf_ny][2] host pointer, or NULL) against the robots' primitives --
   * a capsule's axis against the edge, the edge against a box primitive's faces; 2 = the robots' capsule axes against the oriented
   * boxes of the scene (1-dof link plank / door, free box, scenery boxes): the closest point of the whole segment, not of its two end
   * points; 4 = the twelve edges of those boxes against the robots' box primitives (off by default: desc builder default 3).  The closest approach of a segment to a convex box
   * is a one-dimensional convex minimisation: both engines run the same 18-evaluation golden-section search.  Such a contact is kept
   * when it lies BETWEEN feature points (segment parameter inside 5 .. 95 %); 0 = round 3's behaviour.
   * Bit 8 (round 6, off by default; not an edge contact, it shares the contact-generation option word): manifold reduction -- a robot that
   * touches the static world at more points than its eight one-sided slots keeps every contact penetrating by more than 1 mm first (deepest
   * 2 mm class first) and fills the rest in feature order, instead of the first eight in feature order whatever their depth; counted in
   * MQE_T_CONTACT_REDUCED instead of MQE_T_CONTACT_OVERFLOW.  Built in both engines and parity-tested; not the default because a limp robot
   * collapsing onto its belly comes to rest in another pose with it. */
  int32_t edge_contacts;
  const float* wall_corner;
  float soft_dof_pos_limit;               /* rewards.soft_dof_pos_limit (legged_robot.py:317-321): fraction of the URDF joint range
                                             outside of which MQE_T_SUBSTEP_EXCEED_DOF_POS_LIMITS flags a joint; 0 = 1.0 */
  /* per-env constants, host pointers */
  const float* env_origins;               /* [N,3] */
  const float* agent_origins;             /* [N,A,3] */
  const float* base_init_state;           /* [A,13] */
  const float* npc_init_state;            /* [P,13] */
  const float* gate_pos;                  /* [N,2] task specific (wrappers' gate_pos / football gate) or NULL */
  /* Run-time terrain curriculum (terrain.curriculum with several rows; legged_robot.py:479-503, called first in reset_idx,
   * go1.py:123-125), upstream's code restated with its accidents (fixture tests/golden/fullstep_pushbox_curriculum.npz): when env e
   * is reset -- the first reset() included, init_done is set before it -- the distance walked is measured on ROW e of the agents'
   * root-state tensor (robot e % A of env e / A: single-agent code left in place) against env e's origin, as the rows stood before
   * any reset of this step; farther than terrain_env_length / 2 moves the env one level up; the commands Go1 never samples are zero,
   * so nothing ever moves down; past the last level a level is re-drawn uniformly; then ONLY env_origins[e] is re-read from the
   * origin table -- the live tensor behind the NPCs' respawn (:437) and the football / push-box wrappers; agent_origins (where the
   * robots respawn), the env_origins_repeat copy behind obs.base_pos, env_info and the copies made at construction (sheep
   * wrapper, gate positions) keep the first track.  terrain_origins [rows][cols][3], terrain_levels / terrain_types [N] are host
   * pointers read at creation; the live values are MQE_T_TERRAIN_LEVELS / MQE_T_ENV_ORIGINS.  A shard of a larger batch
   * (env_id_offset != 0) cannot evaluate it: row e belongs to another shard's env. */
  int32_t terrain_curriculum, terrain_num_rows, terrain_num_cols;
  float terrain_env_length;
  const float* terrain_origins;
  const int32_t* terrain_levels;
  const int32_t* terrain_types;
  /* termination (reference go1_config.py:187-207, legged_robot.py:159-169) */
  int32_t termination_flags, terminate_on_base_contact, max_episode_length;
  float roll_threshold, pitch_threshold, z_low_threshold, z_high_threshold;
  /* reset distribution (reference legged_robot.py:394-470) */
  int32_t noise_mode;
  float dof_ratio_lo, dof_ratio_hi;
  int32_t has_base_pos_range, has_npc_pos_range;
  float base_pos_x_lo, base_pos_x_hi, base_pos_y_lo, base_pos_y_hi;
  float npc_pos_x_lo, npc_pos_x_hi, npc_pos_y_lo, npc_pos_y_hi;
  float base_vel_lo, base_vel_hi;
  /* domain randomisation -- every switch is off in the reference's task configs (go1_config.py:216-247).  Values are drawn
   * once per handle from the hash RNG keyed by the GLOBAL env id and exposed as MQE_T_DOMAIN_PARAMS:
   *   friction (legged_robot.py:283-294): 64 buckets U(lo, hi), one bucket per env, applied to the robots' shapes; a contact
   *     uses the average of the robot's coefficient and `friction` (PhysX combine mode average with terrain / objects);
   *   added base mass (legged_robot.py:332-334) and base CoM shift (legged_robot_field.py:324-334): per robot, body 0;
   *   lag_timesteps (go1.py:337-339, control type C): the joint targets are delayed by this many SUBSTEPS (the reference
   *     shifts its lag buffer in every _compute_torques call); the buffer is never reset;
   *   push_interval / max_push_vel_xy (go1.py:237, legged_robot.py:470-476): every push_interval-th step the base linear
   *     velocity x, y of every robot is re-drawn from U(-max, max) after the step's observations were taken. */
  int32_t rand_friction; float friction_lo, friction_hi;
  int32_t rand_base_mass; float added_mass_lo, added_mass_hi;
  int32_t rand_com; float com_lo[3], com_hi[3];
  int32_t lag_timesteps;
  int32_t push_interval; float max_push_vel_xy;
  /* sheep script (reference go1_sheep.py:35-64) */
  float sheep_movement_scale, sheep_movement_randomness;
  /* wrapper parameters: reward scales in the order documented in mqe/envs/wrappers of this package */
  float reward_scale[MQE_MAX_REWARD_TERMS];
  float wrapper_param[8];
  /* networks */
  mqe_mlp actuator, adaptation, body;
} mqe_sim_desc;

/* tensor kinds for mqe_sim_tensor (device pointers into handle-owned memory, valid for the handle's lifetime;
 * the zero-copy analogue of gym.acquire_*_tensor + gymtorch.wrap_tensor, legged_robot.py:554-567,595) */
enum {
  MQE_T_ROOT_STATE = 0, MQE_T_DOF_STATE, MQE_T_CONTACT_FORCE, MQE_T_TORQUES, MQE_T_ACTIONS, MQE_T_LAST_ACTIONS,
  MQE_T_LOCOMOTION_OBS, MQE_T_HISTORY, MQE_T_LAST_LOCO_ACTION, MQE_T_LAST_TWO_LOCO_ACTION,
                           /* MQE_T_HISTORY [R][MQE_HIST][72] is the f32 ring (slot order; the newest frame sits in the slot written last).  It is an
                              OUTPUT view: large batches run layer 0 on a compact split-f16 copy that the engine maintains frame by frame
                              (csrc/mqe_common.hpp, MQE_H2_FRAME), so host writes into the ring do not reach the policy there.  The action
                              registers, the observation bag and every other state tensor ARE inputs of the next step.  (Handles created
                              with MQE_GEMM_SPLIT=0 read the ring itself.)  A host that WRITES the ring calls mqe_history_sync afterwards:
                              the compact operand is rebuilt from it (columns 6..17 of a written frame must hold the scene's constants,
                              desc.command_obs -- the compact form carries them on the frame's presence flag). */
  MQE_T_ACT_HIST,          /* [4][R][12]: pos_err_last, pos_err_last_last, vel_last, vel_last_last */
  MQE_T_GAIT_INDICES, MQE_T_CLOCK_INPUTS,
  MQE_T_BASE_LIN_VEL, MQE_T_BASE_ANG_VEL, MQE_T_PROJECTED_GRAVITY, MQE_T_BASE_QUAT,
  MQE_T_EPISODE_LENGTH,    /* int32 [N] */
  MQE_T_RESET_BUF, MQE_T_COLLIDE_BUF, MQE_T_TIME_OUT_BUF, MQE_T_R_TERM, MQE_T_P_TERM, MQE_T_Z_HIGH_TERM, /* uint8 [N] */
  MQE_T_OBS_BAG,           /* [R][74]: base_pos3 base_rpy3 dof_pos12 dof_vel12 lin_vel3 ang_vel3 last_action12
                              last_last_action12 projected_gravity3 clock_inputs4 base_quat4, filled by
                              compute_observations (go1.py:153-196) */
  MQE_T_WRAPPER_OBS, MQE_T_WRAPPER_REWARD, MQE_T_REWARD_SUMS, /* [N,A',D], [N,A'], [MQE_MAX_REWARD_TERMS] */
  MQE_T_SHEEP_POS_AVG, MQE_T_SHEEP_POS_VAR,
  MQE_T_RESET_COUNT,       /* int32 [N] */
  MQE_T_SUBSTEP_TORQUES,   /* [N,4,12A] (legged_robot.py:112-115) */
  MQE_T_NPC_NOISE,         /* [N,P,3] injected N(0,1) for the sheep script when noise_mode is SCRIPTED */
  MQE_T_WRAPPER_PACKED,    /* f32 [N*Aw*D + N*Aw + ceil(N/4)] everything a step returns as ONE contiguous buffer: wrapper observation,
                              reward, then MQE_T_RESET_BUF once more as N BYTES (0 / 1; pad bytes of the last word are never written) --
                              the caller views them as bool in place, and the whole buffer is what the env-sharded runner all-gathers */
  MQE_T_DOMAIN_PARAMS,     /* [R][8]: shape friction of the robot's env, added base mass, base CoM shift xyz, 3 unused; read by every
                              physics step, writable (tests / curricula) */
  MQE_T_SUBSTEP_DOF_VEL,   /* [N,4,12A] joint velocities after each substep (legged_robot.py:114) */
  MQE_T_SUBSTEP_EXCEED_DOF_POS_LIMITS,   /* uint8 [N,4,12A] joint outside its soft position limits after each substep (legged_robot.py:115) */
  MQE_T_CONTACT_OVERFLOW,  /* int32 [N]: substeps so far in which the bounded contact list of the env dropped a touching pair (per-actor
                              cap or list end); 0 everywhere = no contact was ever truncated */
  MQE_T_ENV_ORIGINS,       /* [N,3] the LIVE env origins (legged_robot.py:495): what the NPCs respawn around and the football / push-box
                              wrappers subtract; equal to desc.env_origins unless the terrain curriculum has moved an env */
  MQE_T_TERRAIN_LEVELS,    /* int32 [N] terrain level of each env (legged_robot.py:983,490) */
  MQE_T_CONTACT_REDUCED,   /* int32 [N]: with edge_contacts bit 8: substeps so far in which a robot of the env touched the static world at more
                            * points than its eight one-sided slots and the set was reduced (penetrations of more than 1 mm first, deepest 2 mm class first, then feature
                            * order: feet first); MQE_T_CONTACT_OVERFLOW then counts only what is dropped in list order (NPC caps, the two-actor
                            * share, the end of the list).  Without the bit (default): always zero, robots' extra contacts count as overflow */
  MQE_T_COUNT
};

typedef struct {
  void* ptr;               /* device pointer */
  int32_t ndim;
  int64_t shape[4];
  int32_t dtype;           /* 0 f32, 1 i32, 2 u8 */
} mqe_tensor_view;

typedef struct mqe_sim mqe_sim;

const char* mqe_last_error(void);
int mqe_abi_version(void);
/* the compile-time limits of the library, for a binding to compare with the header it was built against: writes up to n of
 * { MQE_ABI_VERSION, MQE_MAX_AGENTS, MQE_MAX_NPCS, MQE_MAX_SPHERES, MQE_MAX_PRIMS, MQE_MAX_SELF_PAIRS, MQE_MAX_LAYERS, MQE_MAX_REWARD_TERMS,
 *   MQE_NBODY, MQE_NREP, MQE_NDOF, MQE_FRAME, MQE_HIST, MQE_T_COUNT } and returns how many there are (14) */
int mqe_abi_limits(int32_t* out, int n);
int mqe_sizeof_desc(void);   /* sizeof(mqe_sim_desc): lets a foreign-language binding verify its struct mirror */

/* gym.create_sim + load_asset + create_env/create_actor + add_triangle_mesh + prepare_sim
 * (reference legged_robot.py:255-261,754-923; barrier_track.py:395-410; base_task.py:91) */
int mqe_sim_create(const mqe_sim_desc* desc, mqe_sim** out);
int mqe_sim_destroy(mqe_sim* s);
/* gym.acquire_*_tensor + gymtorch.wrap_tensor (legged_robot.py:554-567) */
int mqe_sim_tensor(mqe_sim* s, int kind, mqe_tensor_view* out);

/* ---- unfused, Isaac-Gym-shaped entry points (inner boundary) ---- */
/* Go1.preprocess_action (go1.py:64-108): command -> locomotion obs -> history -> adaptation+body MLP ->
 * clipped joint-target actions.  `command` is [R,3] device memory (already scaled by the task wrapper). */
int mqe_policy_step(mqe_sim* s, const float* command, void* stream);
/* Go1FootballDefender._get_defender_action (go1_football_defender.py:56-80): scripted (x, y, yaw) command of agent 2
 * from the current state -> out_dev [N,3] */
int mqe_defender_command(mqe_sim* s, float* out_dev, void* stream);
/* Go1._compute_torques (go1.py:315-354) / LeggedRobot._compute_torques (legged_robot.py:368-392) */
int mqe_compute_torques(mqe_sim* s, void* stream);
/* gym.set_dof_actuation_force_tensor + gym.simulate + refresh_dof_state_tensor (go1.py:52-56): one dt of rigid-body
 * dynamics with the torques currently in MQE_T_TORQUES; also refreshes the net contact forces */
int mqe_simulate(mqe_sim* s, void* stream);
/* post_decimation_step (legged_robot.py:112-115) */
int mqe_post_decimation_step(mqe_sim* s, int dec_i, void* stream);
/* post_physics_step (legged_robot_field.py:117-119 -> legged_robot.py:117-157) incl. termination, NPC script,
 * in-kernel reset, compute_observations, and the task wrapper's observation / reward */
int mqe_post_physics_step(mqe_sim* s, void* stream);
/* The same in the stages the reference's method has (legged_robot.py:117-157), for a host that runs code of its own between them -- a
 * subclass's check_termination / _step_npc / reset_idx / compute_observations: any OR of
 *   MQE_POST_FRAME    episode counter, body-frame velocities, gravity, gait clock, the wrapper's copy of the NPC rows (:126-139) and the
 *                     default check_termination (:159-169, legged_robot_field.py:121-146) -> MQE_T_RESET_BUF & co;
 *   MQE_POST_NPC      the task's NPC script (:146: the sheep's flocking walk);
 *   MQE_POST_RESET    reset_idx of the envs whose MQE_T_RESET_BUF is set WHEN THIS STAGE RUNS (:147-148), terrain curriculum included;
 *   MQE_POST_OBS      compute_observations, last_actions, last_dof_vel (:149-152);
 *   MQE_POST_WRAPPER  the task wrapper's observation / reward and the periodic push; the step counter advances with this stage.
 * Stages of one call run in this order; mqe_post_physics_step == MQE_POST_ALL in one launch (the staged form is a plain
 * thread-per-env kernel per call: the same arithmetic -- results equal to the last bit or two -- not the fast path). */
enum { MQE_POST_FRAME = 1, MQE_POST_NPC = 2, MQE_POST_RESET = 4, MQE_POST_OBS = 8, MQE_POST_WRAPPER = 16, MQE_POST_ALL = 31,
       MQE_POST_WRAPPER_LEVEL = 32 };   /* modifier of MQE_POST_WRAPPER: the call comes from a task wrapper's step() (go1tug re-poses its slider then) */
int mqe_post_physics_stage(mqe_sim* s, int stages, void* stream);
/* task wrapper observation + reward only (mqe/envs/wrappers/go1_<task>_wrapper.py step()/reset() bodies) from the current contents of
 * MQE_T_OBS_BAG, MQE_T_ROOT_STATE (NPC rows), the termination flags and MQE_T_SHEEP_POS_*; part of
 * mqe_post_physics_step / mqe_reset_all, exposed separately so the wrappers can be checked against golden vectors */
int mqe_wrapper_eval(mqe_sim* s, int is_reset_call, void* stream);
/* gym.set_actor_root_state_tensor_indexed / set_dof_state_tensor_indexed (legged_robot.py:419-421,468-470):
 * state tensors are live device memory, so these only invalidate cached per-env data for the listed actors */
int mqe_set_actor_root_state_indexed(mqe_sim* s, const int32_t* actor_ids_dev, int n, void* stream);
int mqe_set_dof_state_indexed(mqe_sim* s, const int32_t* actor_ids_dev, int n, void* stream);
/* Go1.reset (go1.py:147-151): reset_idx(all) + compute_observations, no physics step */
int mqe_reset_all(mqe_sim* s, void* stream);

/* ---- fused fast path: one Go1.step (go1.py:35-62) + task wrapper ----
 * actions: [N, A', 3] raw policy actions in [-1,1] (the wrapper's clip and action_scale are applied inside);
 * results land in MQE_T_WRAPPER_OBS / MQE_T_WRAPPER_REWARD / MQE_T_RESET_BUF. */
int mqe_step(mqe_sim* s, const float* actions, void* stream);
/* mqe_step in two halves, for a host that has launches of its own to place between them: _begin enqueues the wrapper head
 * and the locomotion policy (Go1.step up to go1.py:41), _end the decimation loop, post-physics step and wrapper
 * (go1.py:46-62).  mqe_step == _begin; _end.  The env-sharded runner issues the all-gather of the PREVIOUS batch between
 * the two, so that the collective shares the GPU with the physics kernel (wave-granular, tolerant of a few displaced
 * CUs) and not with the policy GEMM (one workgroup per CU: any displaced workgroup costs a whole extra round). */
int mqe_step_begin(mqe_sim* s, const float* actions, void* stream);
int mqe_step_end(mqe_sim* s, void* stream);
/* mqe_step_begin in two parts, for a host that wants to place launches of its own between layer 0 of the policy and the rest of it:
 * mqe_step_head = wrapper head + history frame + layer 0 (k_pre_policy, k_gemm_*), mqe_step_tail = the rest of the policy
 * (k_policy_tail); then mqe_step_end as usual.  The env-sharded runner issues the previous batch's all-gather after the head -- the
 * RCCL kernel then shares the GPU with the policy tail, which leaves every CU three quarters empty, instead of with either of the two
 * kernels that fill the machine exactly -- and makes the physics kernel wait for it (bench.py --gather tail). */
int mqe_step_head(mqe_sim* s, const float* actions, void* stream);
int mqe_step_tail(mqe_sim* s, void* stream);
/* Where the following launches write what a step returns (the MQE_T_WRAPPER_PACKED layout: obs | reward | done): a device
 * buffer of the caller, at least as large as MQE_T_WRAPPER_PACKED, or NULL for the engine's own buffer (the one the
 * MQE_T_WRAPPER_* views show).  Host-side switch only; a launch uses the buffer that was set when it was enqueued.  The
 * wrappers hand every step a fresh tensor this way (the reference returns new tensors each step) instead of copying. */
int mqe_set_return_buffer(mqe_sim* s, float* packed_dev);
/* Go1.step itself (go1.py:35-62) as one call for control type "C": `command` [R, num_command_dims] = the per-robot command rows
 * Go1.step receives (already scaled by a task wrapper, if any; re-clipped to +-1 inside as go1.py:38 does unless the scene is the
 * defender variant), policy, decimation loop and post-physics step as the five fused launches.  No wrapper head and no wrapper
 * evaluation: the task wrapper's observation, reward and bookkeeping belong to mqe_step.  Equivalent to mqe_policy_step + decimation x
 * (mqe_compute_torques, mqe_simulate, mqe_post_decimation_step) + mqe_post_physics_step. */
int mqe_step_command(mqe_sim* s, const float* command, void* stream);
/* The same for the low-level control types "P" / "V" / "T" (Go1.step's else branch, go1.py:42-44 -> pre_physics_step,
 * legged_robot.py:108-110, and the PD / torque laws of legged_robot.py:380-392): actions [R, 12] joint-space actions, clipped
 * to clip_actions inside; no locomotion policy runs.  The decimation loop, post-physics step and (plain) wrapper are the
 * fused ones. */
int mqe_step_joint(mqe_sim* s, const float* actions12, void* stream);

/* The onboard forward depth camera of LeggedRobotField (legged_robot_field.py:23-93: create_camera_sensor + attach_camera_to_body on the
 * base link, :196-223: get_camera_image_gpu_tensor(IMAGE_DEPTH)) for every robot, from the CURRENT state: out_dev [R][height][width]
 * device floats, Isaac Gym's IMAGE_DEPTH convention -- the NEGATIVE distance along the optical axis to the first surface, -inf where
 * nothing lies within far_m.  cam_pos3 / cam_rpy3 (host pointers) = cfg.sensor.forward_camera.position / .rotation (ZYX Euler) in the base
 * link; the camera looks along its +x, +z up; pixel (0, 0) is the top-left corner.  What is seen is what the physics collides with: the
 * ground (slab or relief), the wall prisms, the OTHER robots' collision primitives, free NPCs, the 1-dof link, the scenery boxes -- a ray
 * caster (csrc/kernels_camera.hpp), not the reference's rasteriser, which is closed: the image is SPECIFIED by the scalar caster of the CPU
 * oracle (oracle/mqe_oracle.c: mqo_render_depth -- conventions, surfaces, marching rules), which passes geometric known answers on its own
 * (tests/test_camera_oracle.py); the kernel is held to it per pixel on 8 scenes (tests/test_camera_gpu.py: <= 0.2 % of the pixels -- silhouettes -- may differ in hit / miss or depth, the rest agree to 1e-4 m + 1e-5 relative).  Colour images (IMAGE_COLOR) are not offered. */
int mqe_render_depth(mqe_sim* s, float* out_dev, int height, int width, float horizontal_fov_deg, const float* cam_pos3, const float* cam_rpy3,
                     float far_m, void* stream);

/* After host writes into MQE_T_HISTORY (obs_history of the reference, go1.py:102,145): rebuilds what the engine derives from the ring --
 * the compact split-f16 operand of layer 0, the presence flags (a frame of 70 zeros is absent), the carrier columns, the continuity
 * bits.  A no-op for handles whose layer 0 reads the ring itself.  Enqueued on `stream`. */
int mqe_history_sync(mqe_sim* s, void* stream);

/* debug taps (tests): M^-1 (18 x 18) of one robot and the contact list of one env ([<= 64][8]: actor A, link A, actor B (-1 static),
 * link B, separation, normal xyz) from the CURRENT state, without advancing it; outputs are host pointers */
int mqe_debug_dynamics(mqe_sim* s, int env, int robot, float* minv_out_host, int* nc_out_host, float* contacts_out_host);
int mqe_debug_times(long long* out16);   /* clock64 stamps of the phases of the last mqe_debug_dynamics launch */
/* per-phase counter runs (tools/phase_counters.py): the following mqe_simulate launches leave the wavefront after phase tap `tap`
 * and write nothing back (tap < 0: normal launches again; tap >= 100: the same specialised kernel run to the end, state written).  Two-robot scenes without objects only; any other scene is refused.
 * The environment variable MQE_DEBUG_STOP_PHASE sets the same thing when the handle is created. */
int mqe_debug_stop_phase(mqe_sim* s, int tap);
/* handles created with MQE_WAVE_TIMES=1 in the environment: wall-clock stamps (100 MHz) of every wavefront of the last fused
 * decimation launch at entry and exit + its HW_ID and XCC_ID registers, [num wavefronts][4] (tools/dev/wave_times.py: the spread
 * of the wavefronts' run times and where the dispatcher put them) */
int mqe_debug_wave_times(mqe_sim* s, long long* out_host);
/* handles created with MQE_TAIL_TIMES=1 (fused policy tail): wall-clock stamps (100 MHz) of every workgroup of the last k_policy_tail launch at
 * its entry [0], after each of its eight barriers [1 .. 8] and at its end [9], [row blocks of 32 robots][16] (tools/dev/tail_times.py); slots
 * [10 .. 15] of the same rows carry the split-f16 layer-0 kernel's (k_gemm_h2) stamps of the same step, workgroup by workgroup: the wall clock at
 * entry / end of the K loop / exit, then the shader clock (s_memtime) at the same three points */
int mqe_debug_tail_times(mqe_sim* s, long long* out_host);
/* handles created with MQE_PHASE_TIMES=1 (two robots without objects, two robots + a flock, three robots + ball): the fused decimation launch runs with its phase taps
 * live; [num_envs][4 substeps][16] wall-clock stamps (100 MHz) of the last launch: taps 0..14 of the substep (kernels_physics.hpp
 * TSTAMP), [15] = its end (tools/dev/phase_walltimes.py: where the time of a full launch goes, phase by phase) */
int mqe_debug_phase_times(mqe_sim* s, long long* out_host);
/* the same handles (scenes whose post-physics step is the decimation kernel's epilogue): [num_envs][16] stamps of the epilogue of the last launch --
 * [0] after the actuator history's write-back, [1] actions staged, [2] loads + frame quantities, [3] flag / frame stores, [4] NPC rows staged,
 * [5] NPC script, [6] reset, [7] observation rows staged, [8] task wrapper (one lane per env), [9] rows flushed, [10] end
 * (tools/dev/epilogue_taps.py, round 6: the wrapper's serialised load -> store chains were half of the epilogue) */
int mqe_debug_epilogue_times(mqe_sim* s, long long* out_host);

/* Checkpoint / resume of the simulation state (the reference has none: SURVEY 5).  The blob holds every state buffer of the handle --
 * the tensors of mqe_sim_tensor and the internal ones (the compact history operand and its ring position, the action-lag ring, wrapper
 * bookkeeping, the reset counters that key the RNG streams, the domain parameters) -- and NOT the scene: load it into a handle created
 * from the same descriptor (same shape and layer-0 path; checked).  A rollout continued after mqe_state_load is bit for bit the
 * rollout that was not interrupted.  Both calls synchronise `stream`; `host_blob` is host memory of mqe_state_size() bytes. */
long long mqe_state_size(mqe_sim* s);
int mqe_state_save(mqe_sim* s, void* host_blob, void* stream);
int mqe_state_load(mqe_sim* s, const void* host_blob, void* stream);

/* bookkeeping for benchmarks: time of every kernel class measured with HIP events on `stream`; `on` = N > 0 brackets the launches of one step in every N
 * (the third step of each period: the first one after a synchronisation starts on an idle GPU), 0 switches it off */
int mqe_profile_enable(mqe_sim* s, int on);
int mqe_profile_read(mqe_sim* s, float* ms_per_kernel, int n, int* n_launches);

#ifdef __cplusplus
}
#endif
#endif
