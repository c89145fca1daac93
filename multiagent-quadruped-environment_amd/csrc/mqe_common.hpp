// mqe_common.hpp -- device-side data model of the MI355X MQE engine (gfx950 only).
//
// One environment = one 64-lane wavefront in the physics kernel; everything else is thread-per-robot /
// thread-per-joint / thread-per-env streaming work.  All state lives in HBM between kernels in the env-major layout
// of include/mqe_hip.h (the Isaac Gym tensor contract of the reference, legged_robot.py:549-645), which is already
// the coalesced layout for a wave that owns one env: its root/dof/contact rows are contiguous.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/mqe_hip.h"

#define MQE_RD 18          // generalized velocities of one robot: 3 lin + 3 ang + 12 joints
#define MQE_OBS_BAG 74
#define MQE_WAVE 64

struct DevMlp {
  int n_layers;
  int dims[MQE_MAX_LAYERS + 1];
  const float* W[MQE_MAX_LAYERS];   // device, (out,in) row-major as given
  const float* b[MQE_MAX_LAYERS];
};

// constants of one simulation, resident in device memory; every kernel receives a pointer to it (uniform loads)
struct DevModel {
  int N, A, P, R, ND, NBR, Aw, D;          // envs, agents, npcs, robots, dofs/env, reported bodies/env, wrapper dims
  int npc_kind, task, npc_dofs_each, npc_lin_only, n_npc_dyn;   // dynamic NPC bodies (ball/sheep) handled by the solver
  int env_id_offset, seed, noise_mode;
  float dt; int decimation; float gravity_z; int solver_iterations;
  float contact_offset, max_depen, friction, erp;
  int solver_type, vel_iters;                      // desc.solver_type (0: velocity-level sweeps with erp, 1: temporal Gauss-Seidel), desc.velocity_iterations
  mqe_robot_model robot;
  float npc_mass, npc_inertia; int npc_n_spheres; float npc_sphere_center[8][3]; float npc_sphere_radius[8];
  int self_collision;                              // contacts between the links of one robot (rm.self_pair)
  int rowgs;                                       // contact sweep variant of the physics kernel (PhysShape::rowgs): 1 = row sweep, scenes of <= 4 actors
  int lag_steps; float max_push;                   // domain randomisation: action lag in substeps (0 = off), push velocity bound
  int has_box, cap_npc; float npc_box_half[3];     // MQE_NPC_BOX: robots' spheres vs the oriented box; terrain contacts kept per NPC
  float seesaw_default_angle;
  int n_static; float sb_center[4][3], sb_half[4][3];     // MQE_NPC_STATIC: world-aligned scenery boxes on the NPC root
  int has_seesaw, ss_axis, ss_link_cyl; float ss_joint_offset[3], ss_plank_center[3], ss_plank_half[3], ss_base_half[3];
  float ss_inertia, ss_vel_limit, ss_col_radius, ss_col_length, ss_theta_lo, ss_theta_hi;
  int control_type; float action_scale, hip_scale_reduction, clip_actions; float torque_limits[12]; float kp, kd;
  float default_dof_pos[12]; float command_obs[70]; float cmd_lin_scale, cmd_ang_scale; int clip_command;
  int cmd_dims, cmd_general; int cmd_src[18]; float cmd_scale[18];      // desc.command_src / command_scale; cmd_general: not the shipped (x, y, yaw) layout
  const float* wall_sdf; int sdf_nx, sdf_ny; float hs, wall_height, ground_z;
  const float* wall_top;                           // per-cell wall top [m] (walls of different heights), or nullptr = wall_height
  const float* wall_corner;                        // per raster point the (x, y) of the nearest convex corner of the wall set, or nullptr (edge contacts)
  int edge_mask;                                   // desc.edge_contacts: 1 wall edges, 2 capsule axes against the scene's boxes
  float prim_feat_t[MQE_MAX_PRIMS][2];             // axis parameters (0 .. 1) of the feature points that sit ON a capsule primitive's axis between its ends (-9: none)
  const float* ground_height;                      // relief of the walkable surface above ground_z at the SDF's raster points, or nullptr
  float soft_lo[12], soft_hi[12];                  // soft joint position limits (legged_robot.py:317-321) of MQE_T_SUBSTEP_EXCEED_DOF_POS_LIMITS
  const float *env_origins, *agent_origins, *base_init, *npc_init, *gate_pos;   // env_origins: the values at construction (obs.base_pos, sheep wrapper, gate height); the live ones are DevState::env_origins_live
  int curriculum, terrain_rows, terrain_cols; float terrain_env_length;            // run-time terrain curriculum (include/mqe_hip.h terrain_curriculum)
  const float* terrain_origins; const int32_t* terrain_types;                      // [rows][cols][3]; [N]
  int termination_flags, terminate_on_base_contact, max_episode_length;
  float roll_thr, pitch_thr, zlow_thr, zhigh_thr;
  float dof_ratio_lo, dof_ratio_hi; int has_base_pos_range, has_npc_pos_range;
  float base_pos_x_lo, base_pos_x_hi, base_pos_y_lo, base_pos_y_hi, npc_pos_x_lo, npc_pos_x_hi, npc_pos_y_lo, npc_pos_y_hi;
  float base_vel_lo, base_vel_hi;
  float sheep_scale, sheep_rand;
  float reward_scale[MQE_MAX_REWARD_TERMS]; float wrapper_param[8];
  // The wave-uniform constants the physics kernel reads in every substep, once more as ONE 64-entry table (bit patterns; pointers as two
  // entries): k_substeps loads it with one coalesced request per launch -- lane i holds entry i -- and reads an entry with v_readlane
  // where the field itself would be an s_load from this struct in every substep (the compiler does not keep ~40 of them in SGPRs across
  // the loop: 106 are in use), i.e. a scalar-cache round trip at the head of most phases of a wavefront whose time is its chain of waits.
  uint32_t hot[64];
  DevMlp actuator;
  // the actuator network's weights in the order k_substeps' lanes consume them, [fragment][lane] (64 floats = one coalesced 256 B load per
  // fragment instead of 64 lanes x a 128 B stride): fragments 0-15 = layer 2 rows (W1[(lane & 31) * 32 + u(r, lane >> 5)]), 16-18 = layer 1
  const float* act_frag;
  // ... and layer 2 once more for the f16 matrix cores: two f16 planes of 2^14 W1 in the fragment order of v_mfma_f32_32x32x16_f16,
  // [k-step 2][plane 2][lane 64][8]: element i of lane (row, g) in k-step s = W1[row][(i & 3) + 16 s + 8 (i >> 2) + 4 g] -- the order in
  // which layer 1's accumulator registers hold the hidden units (kernels_physics.hpp, k_substeps).  act_f16: use it (MQE_ACT_F32=1: no)
  const uint16_t* act_frag16; int act_f16;
  // physics kernel geometry
  int nbody_env, ndof_env, nsph_env, nprim_env, maxc;
  unsigned long long feat_sphere_mask;                   // bit f: feature point f of the robot model belongs to a sphere primitive (a foot)
};

enum {   // DevModel::hot
  HOT_DT = 0, HOT_GRAVITY_Z, HOT_CONTACT_OFFSET, HOT_FRICTION, HOT_MAX_DEPEN, HOT_ERP, HOT_HS, HOT_WALL_HEIGHT, HOT_GROUND_Z, HOT_FEATURE_REACH,
  HOT_SDF_NX, HOT_SDF_NY, HOT_EDGE_MASK, HOT_SOLVER_TYPE, HOT_SOLVER_ITERATIONS, HOT_VEL_ITERS, HOT_SELF_COLLISION, HOT_NSPH_ENV, HOT_NPRIM_ENV,
  HOT_N, HOT_NBR, HOT_N_SPHERES, HOT_N_PRIMS, HOT_N_SELF_PAIRS, HOT_CAP_NPC, HOT_NPC_N_SPHERES, HOT_FEAT_MASK_LO, HOT_FEAT_MASK_HI,
  HOT_WALL_SDF_LO, HOT_WALL_SDF_HI, HOT_GROUND_HEIGHT_LO, HOT_GROUND_HEIGHT_HI, HOT_WALL_TOP_LO, HOT_WALL_TOP_HI, HOT_WALL_CORNER_LO, HOT_WALL_CORNER_HI,
  HOT_NPC_PAIR_REACH,      // two free NPCs whose origins are farther apart than this cannot touch: 2 x (largest |sphere centre| + radius) + contact offset
  HOT_COUNT
};
static_assert(HOT_COUNT <= 64, "DevModel::hot");
__host__ inline void mqe_fill_hot(struct DevModel& m);

// device pointers of all state tensors (kernel argument by value)
struct DevState {
  float *root, *dof, *cf, *torques, *actions, *last_actions, *loco_obs, *hist, *last_loco, *last_two_loco, *act_hist;
  float *gait, *clock, *blv, *bav, *pg, *bquat, *obs_bag, *wobs, *wrew, *rsum, *sheep_avg, *sheep_var, *sub_tau, *npc_noise;
  float *w_last, *w_last2, *cmd, *last_dof_vel;
  float* sub_dof_vel; uint8_t* sub_exceed; int32_t* overflow;   // per-substep logs (legged_robot.py:114-115); truncated-contact-list counter
  float *dparams, *lag_buf;                 // [R][8] friction / added mass / CoM shift (MQE_T_DOMAIN_PARAMS); [(lag + 1)][R][12] scaled actions
  uint16_t* hist2;                          // compact split-f16 copy of the history ring: [R][180 units][2 planes][8] (k_gemm_h2; MQE_H2_FRAME)
  long long* wave_times;                    // debug (MQE_WAVE_TIMES=1): [waves of k_substeps][4] wall_clock64 at entry / exit, HW_ID, XCC_ID, else null
  uint32_t* hist_irr;                       // [R] bit s: the frame in ring slot s does not continue its predecessor's actions (MQE_H2_FRAME)
  int32_t *ep_len, *reset_count;
  uint8_t *reset_buf, *collide_buf, *time_out, *r_term, *p_term, *zh_term, *w_have_last, *w_delayed_reset;
  float* npc_pre;                           // [N][P][13] the wrapper's copy of the NPC rows (staged post-physics path only)
  float *env_origins_live, *curr_xy;        // MQE_T_ENV_ORIGINS [N][3]; pre-reset xy of the agents' root-state rows 0 .. N-1 (terrain curriculum)
  int32_t* terrain_levels;                  // MQE_T_TERRAIN_LEVELS [N]
  uint8_t* wdone;         // the reset flags once more, as the byte tail of the packed return batch (obs | reward | done)
};

// A kernel that takes DevState by value gets its ~60 pointers as a few 16-register kernarg loads, and under register pressure the
// allocator spills and reloads those TUPLES whole: 16 v_readlane for the one pointer a store needs (k_substeps<2, 0>: 1750 static
// v_readlane, a sixth of its vector instructions).  own_global gives a pointer a scalar register pair of its own (the inline asm cuts it
// off the tuple) and keeps it a GLOBAL pointer for the address-space inference (through an integer, so that the accesses stay
// global_load / global_store instead of becoming flat_*, whose completion also holds up every LDS wait).
template <class T> __device__ __forceinline__ void own_global(T*& p) {
  unsigned long long u;
  asm volatile("s_mov_b64 %0, %1" : "=s"(u) : "s"((unsigned long long)p));     // (a tied "+s" operand is coalesced back into the tuple)
  p = (T*)(__attribute__((address_space(1))) T*)u;
}
// a pointer read out of a structure in memory (DevModel's tables): tell the compiler it is a global one
template <class T> __device__ __forceinline__ const T* as_global(const T* p) {
  return (const T*)(const __attribute__((address_space(1))) T*)(unsigned long long)p;
}
static_assert(sizeof(DevState) == 52 * sizeof(void*), "DevState is pointers only, and own_state() below lists every one of them");
__device__ __forceinline__ void own_state(DevState& st) {
  own_global(st.root); own_global(st.dof); own_global(st.cf); own_global(st.torques); own_global(st.actions); own_global(st.last_actions); own_global(st.loco_obs);
  own_global(st.hist); own_global(st.last_loco); own_global(st.last_two_loco); own_global(st.act_hist); own_global(st.gait); own_global(st.clock); own_global(st.blv);
  own_global(st.bav); own_global(st.pg); own_global(st.bquat); own_global(st.obs_bag); own_global(st.wobs); own_global(st.wrew); own_global(st.rsum);
  own_global(st.sheep_avg); own_global(st.sheep_var); own_global(st.sub_tau); own_global(st.npc_noise); own_global(st.w_last); own_global(st.w_last2); own_global(st.cmd);
  own_global(st.last_dof_vel); own_global(st.sub_dof_vel); own_global(st.sub_exceed); own_global(st.overflow); own_global(st.dparams); own_global(st.lag_buf);
  own_global(st.hist2); own_global(st.wave_times); own_global(st.hist_irr); own_global(st.ep_len); own_global(st.reset_count); own_global(st.reset_buf);
  own_global(st.collide_buf); own_global(st.time_out); own_global(st.r_term); own_global(st.p_term); own_global(st.zh_term); own_global(st.w_have_last);
  own_global(st.w_delayed_reset); own_global(st.npc_pre); own_global(st.env_origins_live); own_global(st.curr_xy); own_global(st.terrain_levels); own_global(st.wdone);
}

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// Layer 2 of the actuator network (32 x 32, unitree_go1.pt; go1.py:345) on the f16 matrix cores, shared by k_substeps and the stand-alone
// k_compute_torques_mfma so that the fused and the staged step stay bit for bit the same: acc2 (= the bias on entry) += W1 s1 with both
// operands as two f16 planes (x 2^14: |w| <= 1.06, |s1| < 1), products hh + hl + lh on v_mfma_f32_32x32x16_f16 (each exact, f32
// accumulation; the dropped ll term is <= 2^-22 of a product): 6 MFMAs of 32 cycles instead of the 16 x 64 cycles of the f32 chain.
// Layer 1's accumulator IS the B operand: register r of lane (joint, h) holds hidden unit (r & 3) + 8 (r >> 2) + 4 h, i.e. registers
// 8 s .. 8 s + 7 are the eight k of half h in k-step s once the weight columns are permuted the same way (DevModel::act_frag16).
// wbits: this lane's four weight fragments [k-step][plane] as 16 dwords.  k-step by k-step: eight activations split, three MFMAs; the next
// eight are split while these run.
typedef _Float16 mqe_f16x8 __attribute__((ext_vector_type(8)));
typedef float mqe_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int mqe_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ mqe_f32x16 act_layer2_f16(const mqe_f32x16& s1, mqe_f32x16 acc2, const float* wbits) {
#pragma unroll
  for (int r = 0; r < 16; r++) acc2[r] *= 268435456.0f;                 // the bias in the products' scale, 2^28 (exact)
#pragma unroll
  for (int s2 = 0; s2 < 2; s2++) {
    mqe_f16x8 bh, bl;
#pragma unroll
    for (int i8 = 0; i8 < 8; i8++) {
      const float y = s1[8 * s2 + i8] * 16384.0f;
      const _Float16 hh = (_Float16)y;
      bh[i8] = hh;
      bl[i8] = (_Float16)(y - (float)hh);
    }
    mqe_u32x4 th, tl;
    th.x = __float_as_uint(wbits[8 * s2]); th.y = __float_as_uint(wbits[8 * s2 + 1]); th.z = __float_as_uint(wbits[8 * s2 + 2]); th.w = __float_as_uint(wbits[8 * s2 + 3]);
    tl.x = __float_as_uint(wbits[8 * s2 + 4]); tl.y = __float_as_uint(wbits[8 * s2 + 5]); tl.z = __float_as_uint(wbits[8 * s2 + 6]); tl.w = __float_as_uint(wbits[8 * s2 + 7]);
    const mqe_f16x8 ah = __builtin_bit_cast(mqe_f16x8, th), al = __builtin_bit_cast(mqe_f16x8, tl);
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc2, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc2, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc2, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int r = 0; r < 16; r++) acc2[r] *= 3.7252902984619140625e-9f;    // 2^-28
  return acc2;
}

// counter-based RNG of the reset distribution and the domain randomisation: keyed by (seed, GLOBAL env id, count, stream)
__host__ __device__ __forceinline__ uint32_t mqe_hash(uint32_t seed, uint32_t genv, uint32_t count, uint32_t k) {
  uint32_t x = seed * 0x9E3779B1u ^ genv * 0x85EBCA77u ^ count * 0xC2B2AE3Du ^ k * 0x27D4EB2Fu;
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
__host__ __device__ __forceinline__ float mqe_u01(uint32_t seed, uint32_t genv, uint32_t count, uint32_t k) {
  return (float)(mqe_hash(seed, genv, count, k) >> 8) * (1.0f / 16777216.0f);
}
#define MQE_RNG_CREATE 0xD0D0D0D0u         // `count` of the draws made once per handle (domain parameters)
#define MQE_RNG_PUSH 0x50000000u           // + ordinal of the push
#define MQE_RNG_NPC 0x60000000u            // + ordinal of the post-physics step: the sheep's per-step N(0,1) draws


// f32 -> two f16 planes of scale * x (h + l == scale * x to 22 significand bits); see k_gemm_h2 in kernels_gemm.hpp
// The split-f16 copy of the history (k_gemm_h2's A operand) is COMPACT: per frame 48 columns = 46 entries of the 70-float frame
// + one "frame is present" flag + one carrier column.  K = 30 x 48 = 1440 instead of 30 x 72 padded to 2208.  What is not stored:
//   * columns 6..17 of a frame, the gait parameters: constants of the scene (desc.command_obs, go1.py:411-479: only the velocity
//     command is an action), so all twelve ride on the flag column -- its weight is sum_c W[., c] * command_obs[c], it is 1 in a
//     written frame and 0 in a frame zeroed by a reset, exactly like the constants themselves;
//   * columns 54..65, last_two_locomotion_action (go1.py:99): frame p's copy IS frame p-1's last_locomotion_action (columns
//     42..53; go1.py:106-107 shifts the one into the other), so its weights are added onto those of frame p-1's columns 42..53.
//     Two places where that identity has no partner: (a) the oldest frame of the ring (its predecessor has left): its twelve
//     values ride on the carrier columns of the frames at logical positions 18..29 (rewritten every step with the frame, weight
//     = W[., frame 0, 54 + j] in frame 18 + j's carrier, zero in the others); (b) a frame whose predecessor in the ring does not hold
//     its values -- the first frame after a reset (the history is zeroed, last_locomotion_action is not: go1.py:139-145) or
//     registers written by the host: the push compares bit for bit and records the ring slot in
//     DevState::hist_irr, and k_gemm_h2's epilogue adds W[., frame p, 54..65] (a2(p) - a1(p-1)) in f32 for those rows.
#define MQE_H2_FRAME 48
#define MQE_H2_FLAG_COL 46
#define MQE_H2_CARRIER_COL 47
#define MQE_H2_CARRIER0 18                 // logical positions MQE_H2_CARRIER0 .. +11 carry the oldest frame's last_two_locomotion_action
// frame column -> compact column (-1: constant on the flag column, -2: folded onto the previous frame's columns 42..53)
__host__ __device__ __forceinline__ int h2_col(int c) { return c < 6 ? c : (c < 18 ? -1 : (c < 54 ? c - 12 : (c < 66 ? -2 : c - 24))); }
#define MQE_H2_ASCALE 64.0f                 // activation scale c_a: |x| <= 1023 representable, beyond that the value saturates
__host__ __device__ __forceinline__ uint16_t f16_bits(_Float16 h) { union { _Float16 f; uint16_t u; } v; v.f = h; return v.u; }
__host__ __device__ __forceinline__ void split2(float x, float scale, uint16_t& h, uint16_t& l) {
  float y = x * scale;
  y = y > 65504.0f ? 65504.0f : (y < -65504.0f ? -65504.0f : y);       // NaN falls through (and propagates, as in f32)
  const _Float16 hh = (_Float16)y;                                      // round to nearest even
  h = f16_bits(hh);
  l = f16_bits((_Float16)(y - (float)hh));
}
// element offset of (row-local element k, plane p) in the plane-interleaved layout of k_gemm_h2
__host__ __device__ __forceinline__ size_t h2_index(size_t k, int p) { return ((k >> 3) * 2 + p) * 8 + (k & 7); }

__host__ inline void mqe_fill_hot(DevModel& m) {
  auto f = [](float x) { uint32_t u; memcpy(&u, &x, 4); return u; };
  auto lo = [](const void* p) { return (uint32_t)((uintptr_t)p & 0xFFFFFFFFull); };
  auto hi = [](const void* p) { return (uint32_t)((uintptr_t)p >> 32); };
  memset(m.hot, 0, sizeof m.hot);
  m.hot[HOT_DT] = f(m.dt); m.hot[HOT_GRAVITY_Z] = f(m.gravity_z); m.hot[HOT_CONTACT_OFFSET] = f(m.contact_offset); m.hot[HOT_FRICTION] = f(m.friction);
  m.hot[HOT_MAX_DEPEN] = f(m.max_depen); m.hot[HOT_ERP] = f(m.erp); m.hot[HOT_HS] = f(m.hs); m.hot[HOT_WALL_HEIGHT] = f(m.wall_height);
  m.hot[HOT_GROUND_Z] = f(m.ground_z); m.hot[HOT_FEATURE_REACH] = f(m.robot.feature_reach);
  m.hot[HOT_SDF_NX] = (uint32_t)m.sdf_nx; m.hot[HOT_SDF_NY] = (uint32_t)m.sdf_ny; m.hot[HOT_EDGE_MASK] = (uint32_t)m.edge_mask;
  m.hot[HOT_SOLVER_TYPE] = (uint32_t)m.solver_type; m.hot[HOT_SOLVER_ITERATIONS] = (uint32_t)m.solver_iterations; m.hot[HOT_VEL_ITERS] = (uint32_t)m.vel_iters;
  m.hot[HOT_SELF_COLLISION] = (uint32_t)m.self_collision; m.hot[HOT_NSPH_ENV] = (uint32_t)m.nsph_env; m.hot[HOT_NPRIM_ENV] = (uint32_t)m.nprim_env;
  m.hot[HOT_N] = (uint32_t)m.N; m.hot[HOT_NBR] = (uint32_t)m.NBR; m.hot[HOT_N_SPHERES] = (uint32_t)m.robot.n_spheres; m.hot[HOT_N_PRIMS] = (uint32_t)m.robot.n_prims;
  m.hot[HOT_N_SELF_PAIRS] = (uint32_t)m.robot.n_self_pairs; m.hot[HOT_CAP_NPC] = (uint32_t)m.cap_npc; m.hot[HOT_NPC_N_SPHERES] = (uint32_t)m.npc_n_spheres;
  m.hot[HOT_FEAT_MASK_LO] = (uint32_t)(m.feat_sphere_mask & 0xFFFFFFFFull); m.hot[HOT_FEAT_MASK_HI] = (uint32_t)(m.feat_sphere_mask >> 32);
  m.hot[HOT_WALL_SDF_LO] = lo(m.wall_sdf); m.hot[HOT_WALL_SDF_HI] = hi(m.wall_sdf); m.hot[HOT_GROUND_HEIGHT_LO] = lo(m.ground_height); m.hot[HOT_GROUND_HEIGHT_HI] = hi(m.ground_height);
  {
    float bnd = 0.0f;
    for (int i = 0; i < m.npc_n_spheres && i < 8; i++) {
      const float* c = m.npc_sphere_center[i];
      const float r = sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]) + m.npc_sphere_radius[i];
      bnd = r > bnd ? r : bnd;
    }
    m.hot[HOT_NPC_PAIR_REACH] = f(2.0f * bnd + m.contact_offset + 1e-3f);
  }
  m.hot[HOT_WALL_TOP_LO] = lo(m.wall_top); m.hot[HOT_WALL_TOP_HI] = hi(m.wall_top); m.hot[HOT_WALL_CORNER_LO] = lo(m.wall_corner); m.hot[HOT_WALL_CORNER_HI] = hi(m.wall_corner);
}
