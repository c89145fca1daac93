// kernels_step.hpp -- the streaming kernels around the physics: command/history write (row B), post-policy
// bookkeeping, actuator-net torques (rows E/F/G), post-physics step (rows J,K,L,S,M,N) and the task wrappers (W1-W4).
// Each kernel cites the reference lines it replaces; arithmetic order follows the reference / the CPU oracle so that
// integer and flag results are identical and float results agree to the last bits where libm allows.
#pragma once
#include <type_traits>
#include "mqe_common.hpp"

// ----------------------------------------------------------------------------------------------------------------
// small float helpers with the exact operation order of isaacgym.torch_utils (see oracle/mqe_oracle.c)
__device__ __forceinline__ void quat_rotate_inverse_f(const float* q, const float* v, float* o) {
  float qw = q[3];
  float s = 2.0f * qw * qw - 1.0f;
  float cx = q[1] * v[2] - q[2] * v[1], cy = q[2] * v[0] - q[0] * v[2], cz = q[0] * v[1] - q[1] * v[0];
  float dt = q[0] * v[0] + q[1] * v[1] + q[2] * v[2];
  o[0] = v[0] * s - cx * qw * 2.0f + q[0] * dt * 2.0f;
  o[1] = v[1] * s - cy * qw * 2.0f + q[1] * dt * 2.0f;
  o[2] = v[2] * s - cz * qw * 2.0f + q[2] * dt * 2.0f;
}
__device__ __forceinline__ float wrap2pi(float a) {
  const float T = 6.2831855f;
  float r = fmodf(a, T);
  if (r < 0) r += T;
  return r;
}
__device__ __forceinline__ void euler_xyz_f(const float* q, float* rpy) {
  float x = q[0], y = q[1], z = q[2], w = q[3];
  float sinr = 2.0f * (w * x + y * z), cosr = w * w - x * x - y * y + z * z;
  float sinp = 2.0f * (w * y - z * x);
  float siny = 2.0f * (w * z + x * y), cosy = w * w + x * x - y * y - z * z;
  float roll = atan2f(sinr, cosr);
  float pitch = fabsf(sinp) >= 1.0f ? copysignf(1.5707964f, sinp) : asinf(sinp);
  float yaw = atan2f(siny, cosy);
  rpy[0] = wrap2pi(roll); rpy[1] = wrap2pi(pitch); rpy[2] = wrap2pi(yaw);
}

__device__ __forceinline__ float mqe_rand(const DevModel* m, int env, int count, uint32_t k, float lo, float hi) {
  float u = mqe_u01((uint32_t)m->seed, (uint32_t)(env + m->env_id_offset), (uint32_t)count, k);
  return (hi - lo) * u + lo;
}
// joint target delayed by m->lag_steps substeps (go1.py:337-339): ring [(L + 1)][R * 12] of scaled actions, slot `pos` is
// written, the oldest slot (pos + 1) mod (L + 1) is read.  Both half-waves of an MFMA lane pair write the same value.
__device__ __forceinline__ float lag_target(const DevModel* m, const DevState& st, size_t idx, float as, int pos) {
  if (m->lag_steps <= 0) return as;
  const int n = m->lag_steps + 1;
  const size_t R12 = (size_t)m->R * 12;
  st.lag_buf[(size_t)pos * R12 + idx] = as;
  const int rd = pos + 1 >= n ? 0 : pos + 1;
  return st.lag_buf[(size_t)rd * R12 + idx];
}

// ----------------------------------------------------------------------------------------------------------------
// wrapper.step head: clip(action,-1,1) * action_scale (go1_sheep_wrapper.py:55-56) -> per-robot command; the scripted
// defender command of go1football-defender (go1_football_defender.py:56-80) fills agent 2.  One thread per env.
// a / b and atan(a / b) with the IEEE results of the reference's torch expressions spelt out for b == 0 (the engine is built with
// -fno-honor-infinities / -fno-honor-nans, under which a division by zero is undefined): +-inf resp. +-pi/2; 0 / 0 (NaN in torch) -> 0
__device__ __forceinline__ float div_ieee(float a, float b) { return b != 0.0f ? a / b : (a > 0.0f ? 3.0e38f : (a < 0.0f ? -3.0e38f : 0.0f)); }
__device__ __forceinline__ float atan_ratio(float a, float b) { return b != 0.0f ? atanf(a / b) : (a > 0.0f ? 1.5707964f : (a < 0.0f ? -1.5707964f : 0.0f)); }

__device__ __forceinline__ void defender_command_dev(const DevModel* m, const DevState& st, int e, float* cmd3) {
  int A = m->A, P = m->P;
  const float* root = st.root + (size_t)e * (A + P) * 13;
  const float* dp = root + 2 * 13;
  const float* bp = root + A * 13;
  float gate[3] = {as_global(m->gate_pos)[e * 2], as_global(m->gate_pos)[e * 2 + 1], as_global(m->env_origins)[e * 3 + 2]};
  float tp[3];
  for (int k = 0; k < 3; k++) tp[k] = 0.6f * bp[k] + 0.4f * gate[k];
  float yaw = st.obs_bag[(size_t)(e * A + 2) * MQE_OBS_BAG + 5];
  float yaw_to_gate = 3.1415927f + atan_ratio(gate[1] - dp[1], gate[0] - dp[0]);
  float yc = clampf(yaw_to_gate - yaw, -0.3f, 0.3f) / 0.3f;
  float tdg = sqrtf((tp[0] - gate[0]) * (tp[0] - gate[0]) + (tp[1] - gate[1]) * (tp[1] - gate[1]));
  float ddg = sqrtf((dp[0] - gate[0]) * (dp[0] - gate[0]) + (dp[1] - gate[1]) * (dp[1] - gate[1]));
  float xc = clampf(tdg - ddg, -0.5f, 0.5f);
  float yy = -clampf(gate[1] + div_ieee((tp[1] - gate[1]) * (dp[0] - gate[0]), tp[0] - gate[0]) - dp[1], -0.5f, 0.5f);
  cmd3[0] = xc; cmd3[1] = yy; cmd3[2] = yc;
}

__global__ void k_defender_command(const DevModel* m, DevState st, float* out) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m->N) return;
  float c[3];
  defender_command_dev(m, st, e, c);
  out[e * 3] = c[0]; out[e * 3 + 1] = c[1]; out[e * 3 + 2] = c[2];
}

__global__ void k_wrapper_command(const DevModel* m, DevState st, const float* __restrict__ actions) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m->N) return;
  int A = m->A, Aw = m->Aw;
  const float scale[3] = {2.0f, 0.5f, 0.5f};
  for (int a = 0; a < Aw; a++)
    for (int k = 0; k < 3; k++) {
      float v = actions[((size_t)e * Aw + a) * 3 + k];
      if (m->task != MQE_TASK_TUG) v = clampf(v, -1.0f, 1.0f);          // the tug wrapper does not clip before scaling
      st.cmd[((size_t)e * A + a) * 3 + k] = m->task == MQE_TASK_PLAIN ? v : v * scale[k];
    }
  if (m->task == MQE_TASK_FOOTBALL_DEFENDER) defender_command_dev(m, st, e, st.cmd + ((size_t)e * A + 2) * 3);
}

// ----------------------------------------------------------------------------------------------------------------
// Go1.preprocess_action up to the history update (go1.py:64-102).  One thread per (robot, frame element): the 70-float
// frame is assembled and written once to the robot's ring slot (280 B) instead of re-concatenating 2100 floats.
// `wrapper_actions` != nullptr (fused mqe_step): the wrapper head (k_wrapper_command) is evaluated here for the three
// command columns instead of in a launch of its own; `command` is then ignored.
// Element c (0..71) of robot i's frame in two halves: everything that is READ (pre_policy_load) and everything that is WRITTEN
// (pre_policy_store).  Every write is a pure function of state that this launch does not change (the action registers, the
// observation bag, the other ring slots).  (Round 3 tried to let k_gemm_h2 write the frame of its 128 rows in its prologue and save
// this launch: with 4 wavefronts per CU the loads and stores of 32 robots per wavefront serialise into four memory round trips,
// +17 us on the GEMM against the 12 us of this kernel with its 590 k threads -- measured, dropped.)
struct PreVal { float v, aux; unsigned mk; };
__device__ __forceinline__ PreVal pre_policy_load(const DevModel* m, const DevState& st, const float* __restrict__ command, int hist_slot,
                                                  const float* __restrict__ wrapper_actions, int i, int c) {
  const float* ob = st.obs_bag + (size_t)i * MQE_OBS_BAG;
  PreVal r; r.aux = 0.0f; r.mk = 0u;
  // Most entries of the frame are plain copies: their source ADDRESS is selected first (no memory access in the selection, so the
  // compiler turns it into conditional moves) and ONE load follows -- the if-chain over the columns with a load in every arm made a
  // wavefront, whose 64 lanes span 64 of the 72 columns, walk eight dependent memory round trips one after the other (10.4 us kernel).
  const bool cmd_col = c >= 3 && (c < 6 || (c < 18 && m->cmd_general));      // filled from the command / the wrapper's actions below
  const float* src = nullptr;
  if (c < 3) src = ob + 60 + c;                                  // projected gravity      :95
  else if (c < 18) src = cmd_col ? nullptr : m->command_obs + c;  // fixed gait parameters (constants of the scene: desc.command_obs)
  else if (c < 42) src = ob + (c - 12);                          // dof_pos :96 (ob[6 ..]), dof_vel :97 (ob[18 ..])
  else if (c < 54) src = st.last_loco + (size_t)i * 12 + (c - 42);           //             :98
  else if (c < 66) src = st.last_two_loco + (size_t)i * 12 + (c - 54);       //             :99
  else if (c < 70) src = ob + (c - 3);                           // clock inputs           :100 (ob[63 ..])
  float v = src != nullptr ? *src : 0.0f;
  if (cmd_col && m->cmd_general) {                               // command.cfg beyond / without the velocity command (go1.py:66-93): desc.command_src
    const int srcc = m->cmd_src[c];
    if (srcc < 0) v = m->command_obs[c];
    else {
      float x = command[(size_t)i * m->cmd_dims + srcc];
      if (m->clip_command) x = clampf(x, -1.0f, 1.0f);           // Go1.step clips the whole action row (go1.py:38)
      v = x * m->cmd_scale[c];
    }
  } else if (cmd_col) {                                          // velocity command       :67-68 (+ clip :38)
    float x;
    if (wrapper_actions) {
      const int A = m->A, Aw = m->Aw, e = i / A, a = i - e * A, k = c - 3;
      if (m->task == MQE_TASK_FOOTBALL_DEFENDER && a == 2) {
        float c3[3];
        defender_command_dev(m, st, e, c3);
        x = c3[k];
      } else if (a < Aw) {
        float w = wrapper_actions[((size_t)e * Aw + a) * 3 + k];
        if (m->task != MQE_TASK_TUG) w = clampf(w, -1.0f, 1.0f);        // the tug wrapper does not clip before scaling
        x = m->task == MQE_TASK_PLAIN ? w : w * (k == 0 ? 2.0f : 0.5f);
      } else x = st.cmd[i * 3 + k];
      r.aux = x;                                               // -> st.cmd
    } else x = command[i * 3 + (c - 3)];
    if (m->clip_command) x = clampf(x, -1.0f, 1.0f);
    v = x * (c < 5 ? m->cmd_lin_scale : m->cmd_ang_scale);
  }
  if (c >= 54 && c < 66 && st.hist2) {     // the oldest frame's copy of this column once this frame is in (carrier columns, mqe_common.hpp)
    const int oldest = hist_slot + 1 >= MQE_HIST ? 0 : hist_slot + 1;
    r.aux = st.hist[((size_t)i * MQE_HIST + oldest) * MQE_FRAME + c];
  }
  if (c == 71 && st.hist2) {
    // does this frame continue its predecessor (bit for bit)?  One bit per RING SLOT, so that the update does not depend on the
    // bit's old value (the element may be written twice)
    const int prev = hist_slot > 0 ? hist_slot - 1 : MQE_HIST - 1;
    const float* pa = st.hist + ((size_t)i * MQE_HIST + prev) * MQE_FRAME + 42;
    const float* na = st.last_two_loco + (size_t)i * 12;
    unsigned diff = 0;
#pragma unroll
    for (int j = 0; j < 12; j++) diff |= __float_as_uint(pa[j]) ^ __float_as_uint(na[j]);
    const unsigned bit = 1u << hist_slot;
    const unsigned mk = st.hist_irr[i];
    r.mk = diff ? (mk | bit) : (mk & ~bit);
  }
  r.v = v;
  return r;
}
__device__ __forceinline__ void pre_policy_store(const DevModel* m, const DevState& st, int hist_slot, bool have_wrapper_actions, int i, int c, const PreVal& r) {
  const float v = r.v;
  st.loco_obs[(size_t)i * MQE_FRAME + c] = v;
  st.hist[((size_t)i * MQE_HIST + hist_slot) * MQE_FRAME + c] = v;   // :102
  if (have_wrapper_actions && c >= 3 && c < 6) st.cmd[i * 3 + (c - 3)] = r.aux;
  if (st.hist2) {      // the split-f16 GEMM's operand copy: two f16 planes, compact frames (mqe_common.hpp: MQE_H2_FRAME)
    const int cc = c < 70 ? h2_col(c) : (c == 70 ? MQE_H2_FLAG_COL : -1);      // element 70 of the frame writes the presence flag
    uint16_t* row2 = st.hist2 + (size_t)i * (2 * MQE_HIST * MQE_H2_FRAME);
    if (cc >= 0) {
      uint16_t h, l;
      split2(c < 70 ? v : 1.0f, MQE_H2_ASCALE, h, l);
      const size_t k = (size_t)hist_slot * MQE_H2_FRAME + cc;
      row2[h2_index(k, 0)] = h; row2[h2_index(k, 1)] = l;
    } else if (cc == -2) {
      // last_two_locomotion_action is not stored (it is the previous frame's last_locomotion_action); the oldest frame's copy has no
      // previous frame: component j rides on the carrier column of the frame at logical position MQE_H2_CARRIER0 + j (the newest twelve
      // frames: the K range k_gemm_h2 reaches last, long after its own prologue has written them)
      const int oldest = hist_slot + 1 >= MQE_HIST ? 0 : hist_slot + 1;            // ring slot of logical frame 0 once this frame is in
      int slot = oldest + MQE_H2_CARRIER0 + (c - 54); if (slot >= MQE_HIST) slot -= MQE_HIST;
      uint16_t h, l;
      split2(r.aux, MQE_H2_ASCALE, h, l);
      const size_t k = (size_t)slot * MQE_H2_FRAME + MQE_H2_CARRIER_COL;
      row2[h2_index(k, 0)] = h; row2[h2_index(k, 1)] = l;
    } else if (c == 71) st.hist_irr[i] = r.mk;
  }
}
__device__ __forceinline__ void pre_policy_element(const DevModel* m, const DevState& st, const float* __restrict__ command, int hist_slot,
                                                   const float* __restrict__ wrapper_actions, int i, int c) {
  const PreVal r = pre_policy_load(m, st, command, hist_slot, wrapper_actions, i, c);
  pre_policy_store(m, st, hist_slot, wrapper_actions != nullptr, i, c, r);
}

// One WAVEFRONT per robot: lane l holds element l of the frame, lanes 0..7 element 64 + l as well.  Every load of the wavefront is issued
// before the first wait -- the source of the plain copies, the wrapper's action / the command, the oldest frame's copy of the action columns
// and the previous frame's action column the "continues its predecessor" bit compares -- through ADDRESSES selected without memory accesses
// (a lane that does not need a load reads its robot's observation row instead).  One thread per (robot, element) with the loads in the arms of
// the column if-chain walked ~10 memory round trips one after the other (the wavefront's 64 lanes span 64 of the 72 columns, so every arm runs,
// and the 24 + 1 loads of element 71 sat behind all of them): 8.8 us for a kernel that moves 7 MB.  With the robot in one wavefront the bit is a
// ballot over the lanes that hold the action columns.  The general command layout (desc.command_src) and the scripted defender's robot keep the
// per-element form (pre_policy_element), a wavefront-uniform choice.
__global__ void __launch_bounds__(256) k_pre_policy(const DevModel* m, DevState st, const float* __restrict__ command, int hist_slot,
                                                    const float* __restrict__ wrapper_actions) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (i >= m->R) return;
  const int A = m->A, Aw = m->Aw, task = m->task, clip_command = m->clip_command;
  const float lin = m->cmd_lin_scale, ang = m->cmd_ang_scale;
  const int e = i / A, a = i - e * A;
  const bool wrap = wrapper_actions != nullptr;
  if (m->cmd_general || (wrap && task == MQE_TASK_FOOTBALL_DEFENDER && a == 2)) {      // wavefront-uniform
    pre_policy_element(m, st, command, hist_slot, wrapper_actions, i, lane);
    if (lane < MQE_FRAME - 64) pre_policy_element(m, st, command, hist_slot, wrapper_actions, i, 64 + lane);
    return;
  }
  const bool h2 = st.hist2 != nullptr;
  const float* ob = st.obs_bag + (size_t)i * MQE_OBS_BAG;
  const int oldest = hist_slot + 1 >= MQE_HIST ? 0 : hist_slot + 1, prev = hist_slot > 0 ? hist_slot - 1 : MQE_HIST - 1;
  const float* hist_i = st.hist + (size_t)i * MQE_HIST * MQE_FRAME;
  const unsigned mk = h2 ? st.hist_irr[i] : 0u;
  float x1[2], x2[2], x3[2], x4[2];
  bool on[2], cmdc[2], actc[2];
  int cc[2];
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const int c = q * 64 + lane;
    cc[q] = c;
    on[q] = c < MQE_FRAME;
    cmdc[q] = c >= 3 && c < 6;
    actc[q] = c >= 54 && c < 66;                                // last_two_locomotion_action
    const float* a1 = ob;                                       // (a load nobody reads)
    if (c < 3) a1 = ob + 60 + c;                                // projected gravity      :95
    else if (c < 6) a1 = ob;
    else if (c < 18) a1 = m->command_obs + c;                   // fixed gait parameters (constants of the scene: desc.command_obs)
    else if (c < 42) a1 = ob + (c - 12);                        // dof_pos :96 (ob[6 ..]), dof_vel :97 (ob[18 ..])
    else if (c < 54) a1 = st.last_loco + (size_t)i * 12 + (c - 42);            //             :98
    else if (c < 66) a1 = st.last_two_loco + (size_t)i * 12 + (c - 54);        //             :99
    else if (c < 70) a1 = ob + (c - 3);                         // clock inputs           :100 (ob[63 ..])
    const float* a2 = ob;
    if (cmdc[q]) a2 = wrap ? (a < Aw ? wrapper_actions + ((size_t)e * Aw + a) * 3 + (c - 3) : st.cmd + i * 3 + (c - 3)) : command + i * 3 + (c - 3);
    const float* a3 = ob;
    const float* a4 = ob;
    if (actc[q] && h2) { a3 = hist_i + oldest * MQE_FRAME + c; a4 = hist_i + prev * MQE_FRAME + (c - 12); }
    x1[q] = on[q] ? *a1 : 0.0f; x2[q] = on[q] ? *a2 : 0.0f; x3[q] = on[q] ? *a3 : 0.0f; x4[q] = on[q] ? *a4 : 0.0f;
  }
  // ---- values
  PreVal r[2];
  bool differs = false;
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const int c = cc[q];
    float v = (c < 70 && !cmdc[q]) ? x1[q] : 0.0f;
    r[q].aux = 0.0f; r[q].mk = 0u;
    if (cmdc[q]) {                                              // velocity command       :67-68 (+ clip :38)
      float x = x2[q];
      if (wrap) {
        if (a < Aw) {
          if (task != MQE_TASK_TUG) x = clampf(x, -1.0f, 1.0f);            // the tug wrapper does not clip before scaling
          x = task == MQE_TASK_PLAIN ? x : x * (c == 3 ? 2.0f : 0.5f);
        }
        r[q].aux = x;                                           // -> st.cmd
      }
      if (clip_command) x = clampf(x, -1.0f, 1.0f);
      v = x * (c < 5 ? lin : ang);
    }
    if (actc[q] && h2) {
      r[q].aux = x3[q];                                         // the oldest frame's copy of this column once this frame is in
      differs = differs || (on[q] && __float_as_uint(x4[q]) != __float_as_uint(x1[q]));
    }
    r[q].v = v;
  }
  // does this frame continue its predecessor (bit for bit)?  One bit per RING SLOT: the update does not depend on the bit's old value
  const bool diff = __ballot(differs) != 0ull;
  const unsigned bit = 1u << hist_slot;
  if (lane == MQE_FRAME - 1 - 64) r[1].mk = diff ? (mk | bit) : (mk & ~bit);
  pre_policy_store(m, st, hist_slot, wrap, i, lane, r[0]);
  if (lane < MQE_FRAME - 64) pre_policy_store(m, st, hist_slot, wrap, i, 64 + lane, r[1]);
}

// go1.py:106-107 + :40-41: shift the last-action registers and clip the new joint targets.  act: [R, ld]
__global__ void k_post_policy(const DevModel* m, DevState st, const float* __restrict__ act, int ld) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= m->R * 12) return;
  int i = idx / 12, k = idx - i * 12;
  float a = act[(size_t)i * ld + k];
  st.last_two_loco[idx] = st.last_loco[idx];
  st.last_loco[idx] = a;
  st.actions[idx] = clampf(a, -m->clip_actions, m->clip_actions);
}

// body layer 0 finish: v = (hist . W + b) + lat0*w0 + lat1*w1 ; ELU   (go1.py:404: body(cat(history, latent)))
__global__ void k_body_l0_finish(float* __restrict__ P1, int ldp, int col0, int ncols, const float* __restrict__ lat, int ldl,
                                 const float* __restrict__ w_lat0, const float* __restrict__ w_lat1, int R) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= R * ncols) return;
  int i = idx / ncols, c = idx - i * ncols;
  float v = P1[(size_t)i * ldp + col0 + c];
  v = fmaf(lat[(size_t)i * ldl], w_lat0[c], v);
  v = fmaf(lat[(size_t)i * ldl + 1], w_lat1[c], v);
  P1[(size_t)i * ldp + col0 + c] = v > 0 ? v : expm1f(v);
}

// ----------------------------------------------------------------------------------------------------------------
// Go1._compute_torques (go1.py:315-354) incl. the actuator network (go1.py:367-382): one thread per joint; the 1313
// weights are wave-uniform, so they arrive through the scalar cache and every FMA is v_fmac(vgpr, sgpr).
__device__ __forceinline__ float softsign_f(float x) { return x / (1.0f + fabsf(x)); }

__global__ void __launch_bounds__(256) k_compute_torques(const DevModel* m, DevState st, int dec_i, int lag_pos) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  int R = m->R, A = m->A;
  if (idx >= R * 12) return;
  int i = idx / 12, j = idx - i * 12;
  int env = i / A, a = i - env * A;
  const float* ds = st.dof + ((size_t)env * m->ND + a * 12 + j) * 2;
  float q = ds[0], qd = ds[1];
  float as = st.actions[idx] * m->action_scale;
  float tau;
  if (m->control_type == MQE_CTRL_C) {
    if (j % 3 == 0) as *= m->hip_scale_reduction;
    float target = lag_target(m, st, (size_t)idx, as, lag_pos) + m->default_dof_pos[j];
    float err = q - target;
    float* e1 = st.act_hist; float* e2 = e1 + (size_t)R * 12; float* v1 = e2 + (size_t)R * 12; float* v2 = v1 + (size_t)R * 12;
    float x[6] = {err, e1[idx], e2[idx], qd, v1[idx], v2[idx]};
    const float* W0 = m->actuator.W[0]; const float* b0 = m->actuator.b[0];
    const float* W1 = m->actuator.W[1]; const float* b1 = m->actuator.b[1];
    const float* W2 = m->actuator.W[2]; const float* b2 = m->actuator.b[2];
    float h1[32], h2[32];
#pragma unroll
    for (int o = 0; o < 32; o++) {
      float acc = 0.0f;
#pragma unroll
      for (int k = 0; k < 6; k++) acc = fmaf(W0[o * 6 + k], x[k], acc);
      h1[o] = softsign_f(acc + b0[o]);
    }
#pragma unroll
    for (int o = 0; o < 32; o++) {
      float acc = 0.0f;
#pragma unroll
      for (int k = 0; k < 32; k++) acc = fmaf(W1[o * 32 + k], h1[k], acc);
      h2[o] = softsign_f(acc + b1[o]);
    }
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 32; k++) acc = fmaf(W2[k], h2[k], acc);
    tau = acc + b2[0];
    e2[idx] = e1[idx]; e1[idx] = err;
    v2[idx] = v1[idx]; v1[idx] = qd;
  } else if (m->control_type == MQE_CTRL_V) {       // legged_robot.py:387: velocity targets, D term on the change per policy step
    tau = m->kp * (as - qd) - m->kd * (qd - st.last_dof_vel[idx]) / m->dt;
  } else if (m->control_type == MQE_CTRL_P) {
    tau = m->kp * (as + m->default_dof_pos[j] - q) - m->kd * qd;       // legged_robot.py:385
  } else if (m->control_type == MQE_CTRL_T) {
    tau = as;
  } else {
    tau = 0.0f;
  }
  float lim = m->torque_limits[j];
  tau = clampf(tau, -lim, lim);
  st.torques[idx] = tau;
  if (dec_i >= 0) st.sub_tau[((size_t)env * 4 + dec_i) * 12 * A + a * 12 + j] = tau;   // post_decimation_step :113
}

// MFMA form of the same computation for control type "C" (the batched actuator-MLP GEMM of the north star):
// one wavefront = 32 joints.  All three layers are evaluated TRANSPOSED (units x joints) with v_mfma_f32_32x32x2_f32,
// so the accumulator layout of one layer (lane = joint column, 16 registers = 16 hidden units, split over the two
// half-waves) is exactly the B-operand layout of the next: step r of layer 2 consumes hidden unit u(r,h) =
// (r&3)+8(r>>2)+4h from register r of lane (joint, h) -- no LDS transpose, no shuffles; the weights are the A operand,
// pre-permuted per lane into 3 + 16 VGPRs.  The 32->1 output layer is 16 FMAs per lane + one cross-half add.
typedef float f32x16_t __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(256) k_compute_torques_mfma(const DevModel* m, DevState st, int dec_i, int lag_pos) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j32 = lane & 31, h = lane >> 5;
  const int R = m->R, A = m->A;
  const int nj = R * 12;
  const int idx = (blockIdx.x * 4 + wave) * 32 + j32;          // joint handled by this lane pair
  const bool valid = idx < nj;
  const int ii = valid ? idx : nj - 1;
  const int i = ii / 12, j = ii - i * 12;
  const int env = i / A, a = i - env * A;
  const float* W0 = m->actuator.W[0]; const float* b0 = m->actuator.b[0];
  const float* W1 = m->actuator.W[1]; const float* b1 = m->actuator.b[1];
  const float* W2 = m->actuator.W[2]; const float* b2 = m->actuator.b[2];
  // per-lane weight fragments (A operands): row = hidden unit j32, k = the half-wave's element of each k-pair
  float a1[3], a2[16], w3[16];
  f32x16_t acc1, acc2;
#pragma unroll
  for (int s2 = 0; s2 < 3; s2++) a1[s2] = W0[j32 * 6 + 2 * s2 + h];
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int u = (r & 3) + 8 * (r >> 2) + 4 * h;
    a2[r] = W1[j32 * 32 + u];             // (act_f16: overwritten below with the four f16 fragments)
    w3[r] = W2[u];
    acc1[r] = b0[u];
    acc2[r] = b1[u];
  }
  float* e1 = st.act_hist; float* e2 = e1 + (size_t)R * 12; float* v1 = e2 + (size_t)R * 12; float* v2 = v1 + (size_t)R * 12;
  const float* ds = st.dof + ((size_t)env * m->ND + a * 12 + j) * 2;
  const float q = ds[0], qd = ds[1];
  float as = st.actions[ii] * m->action_scale;
  if (j % 3 == 0) as *= m->hip_scale_reduction;
  const float err = q - (lag_target(m, st, (size_t)ii, as, lag_pos) + m->default_dof_pos[j]);
  const float x0 = err, x1 = e1[ii], x2 = e2[ii], x3 = qd, x4 = v1[ii], x5 = v2[ii];
  // layer 1 (K = 6): B operand = input (2s + h) of this lane's joint
  acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[0], h ? x1 : x0, acc1, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[1], h ? x3 : x2, acc1, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[2], h ? x5 : x4, acc1, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 16; r++) acc1[r] = softsign_f(acc1[r]);
  // layer 2 (K = 32): step r consumes hidden units u(r,0), u(r,1); or -- DevModel::act_f16, the default -- the split-f16 form k_substeps uses
  // (mqe_common.hpp: act_layer2_f16; the same instruction sequence on the same bits, so the staged step stays the fused one bit for bit)
  if (m->act_f16) {
    const mqe_u32x4* f16w = reinterpret_cast<const mqe_u32x4*>(m->act_frag16) + lane;
#pragma unroll
    for (int f = 0; f < 4; f++) {
      const mqe_u32x4 t = f16w[f * 64];
      a2[4 * f] = __uint_as_float(t.x); a2[4 * f + 1] = __uint_as_float(t.y); a2[4 * f + 2] = __uint_as_float(t.z); a2[4 * f + 3] = __uint_as_float(t.w);
    }
    acc2 = act_layer2_f16(acc1, acc2, a2);
  } else {
#pragma unroll
    for (int r = 0; r < 16; r++) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[r], acc1[r], acc2, 0, 0, 0);
  }
  float part = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; r++) part = fmaf(w3[r], softsign_f(acc2[r]), part);
  float tau = part + __shfl_xor(part, 32, 64) + b2[0];
  if (valid && h == 0) {
    e2[idx] = x1; e1[idx] = err;
    v2[idx] = x4; v1[idx] = qd;
    const float lim = m->torque_limits[j];
    tau = clampf(tau, -lim, lim);
    st.torques[idx] = tau;
    if (dec_i >= 0) st.sub_tau[((size_t)env * 4 + dec_i) * 12 * A + a * 12 + j] = tau;
  }
}

// ----------------------------------------------------------------------------------------------------------------
// post_physics_step (legged_robot.py:117-157, legged_robot_field.py:117-146, go1.py:153-279,110-145, go1_sheep.py:35-64)
// One thread per env (a few hundred flops each; ~10 kB/env of traffic only when the env resets).
__device__ __forceinline__ void compute_observations_env(const DevModel* m, const DevState& st, int e, int in_step) {
  int A = m->A;
  for (int a = 0; a < A; a++) {
    int i = e * A + a;
    float* ob = st.obs_bag + (size_t)i * MQE_OBS_BAG;
    const float* rs = st.root + ((size_t)e * (A + m->P) + a) * 13;
    for (int k = 0; k < 3; k++) ob[k] = rs[k] - as_global(m->env_origins)[e * 3 + k];
    euler_xyz_f(st.bquat + i * 4, ob + 3);
    for (int j = 0; j < 12; j++) {
      const float* ds = st.dof + ((size_t)e * m->ND + a * 12 + j) * 2;
      ob[6 + j] = (ds[0] - m->default_dof_pos[j]) * 1.0f;
      ob[18 + j] = ds[1] * 0.05f;
      float act = st.actions[i * 12 + j];
      ob[36 + j] = act;
      ob[48 + j] = in_step ? act : st.last_actions[i * 12 + j];    // view aliasing, see oracle
    }
    for (int k = 0; k < 3; k++) {
      ob[30 + k] = st.blv[i * 3 + k] * 2.0f;
      ob[33 + k] = st.bav[i * 3 + k] * 0.25f;
      ob[60 + k] = st.pg[i * 3 + k];
    }
    for (int k = 0; k < 4; k++) { ob[63 + k] = st.clock[i * 4 + k]; ob[67 + k] = st.bquat[i * 4 + k]; }
  }
}

// Run-time terrain curriculum (legged_robot.py:479-503; include/mqe_hip.h terrain_curriculum).  k_curriculum_snapshot copies the xy of the
// agents' root-state ROWS 0 .. N-1 (row e = robot e % A of env e / A: upstream indexes that tensor with env ids) before the post-physics
// kernel resets anything; the reset of env e then moves the env's level and re-reads ITS origin only.
__global__ void __launch_bounds__(256) k_curriculum_snapshot(const DevModel* m, DevState st) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m->N) return;
  const float* rs = st.root + ((size_t)(e / m->A) * (m->A + m->P) + e % m->A) * 13;
  st.curr_xy[e * 2] = rs[0]; st.curr_xy[e * 2 + 1] = rs[1];
}
__device__ __forceinline__ void curriculum_move_dev(const DevModel* m, const DevState& st, int e) {
  const float dx = st.curr_xy[e * 2] - st.env_origins_live[e * 3], dy = st.curr_xy[e * 2 + 1] - st.env_origins_live[e * 3 + 1];
  const float distance = sqrtf(dx * dx + dy * dy);
  const int move_up = distance > m->terrain_env_length / 2 ? 1 : 0;      // (move_down: distance < |commands| * ... with commands == 0: never)
  int lvl = st.terrain_levels[e] + move_up;
  if (lvl >= m->terrain_rows) lvl = (int)(mqe_u01((uint32_t)m->seed, (uint32_t)(e + m->env_id_offset), (uint32_t)st.reset_count[e], 250u) * (float)m->terrain_rows);
  else if (lvl < 0) lvl = 0;
  if (lvl >= m->terrain_rows) lvl = m->terrain_rows - 1;
  st.terrain_levels[e] = lvl;
  for (int k = 0; k < 3; k++) st.env_origins_live[e * 3 + k] = as_global(m->terrain_origins)[((size_t)lvl * m->terrain_cols + as_global(m->terrain_types)[e]) * 3 + k];
}

__device__ __forceinline__ void reset_env_dev(const DevModel* m, const DevState& st, int e) {
  int A = m->A, P = m->P;
  float* root = st.root + (size_t)e * (A + P) * 13;
  float* dofs = st.dof + (size_t)e * m->ND * 2;
  int cnt = st.reset_count[e];
  if (m->curriculum) curriculum_move_dev(m, st, e);          // go1.py:123-125: first thing in reset_idx
  for (int a = 0; a < A; a++)
    for (int j = 0; j < 12; j++) {
      float ratio = mqe_rand(m, e, cnt, (uint32_t)(a * 12 + j), m->dof_ratio_lo, m->dof_ratio_hi);
      dofs[(a * 12 + j) * 2] = m->default_dof_pos[j] * ratio;
      dofs[(a * 12 + j) * 2 + 1] = 0.0f;
    }
  for (int k = 12 * A; k < m->ND; k++) { dofs[k * 2] = m->seesaw_default_angle; dofs[k * 2 + 1] *= 0.0f; }
  for (int a = 0; a < A; a++) {
    float* rs = root + a * 13;
    for (int k = 0; k < 13; k++) rs[k] = as_global(m->base_init)[a * 13 + k];
    for (int k = 0; k < 3; k++) rs[k] += as_global(m->agent_origins)[((size_t)e * A + a) * 3 + k];
  }
  for (int p = 0; p < P; p++) {
    float* rs = root + (A + p) * 13;
    for (int k = 0; k < 13; k++) rs[k] = as_global(m->npc_init)[p * 13 + k];
    for (int k = 0; k < 3; k++) rs[k] += st.env_origins_live[e * 3 + k];
  }
  if (m->has_base_pos_range)
    for (int a = 0; a < A; a++) {
      root[a * 13 + 0] += mqe_rand(m, e, cnt, (uint32_t)(64 + a), m->base_pos_x_lo, m->base_pos_x_hi);
      root[a * 13 + 1] += mqe_rand(m, e, cnt, (uint32_t)(72 + a), m->base_pos_y_lo, m->base_pos_y_hi);
    }
  if (m->has_npc_pos_range)
    for (int p = 0; p < P; p++) {
      root[(A + p) * 13 + 0] += mqe_rand(m, e, cnt, (uint32_t)(128 + p), m->npc_pos_x_lo, m->npc_pos_x_hi);
      root[(A + p) * 13 + 1] += mqe_rand(m, e, cnt, (uint32_t)(160 + p), m->npc_pos_y_lo, m->npc_pos_y_hi);
    }
  for (int a = 0; a < A; a++)
    for (int c = 0; c < 6; c++) root[a * 13 + 7 + c] = mqe_rand(m, e, cnt, (uint32_t)(80 + a * 6 + c), m->base_vel_lo, m->base_vel_hi);
  for (int k = 0; k < 12 * A; k++) st.last_actions[(size_t)e * 12 * A + k] = 0.0f;
  st.ep_len[e] = 0;
  st.reset_buf[e] = 1;
  st.wdone[e] = 1;
  for (int a = 0; a < A; a++) st.gait[e * A + a] = 0.0f;
  // history zeroing (go1.py:145) is done by k_reset_history, 16 B per thread
  st.reset_count[e] = cnt + 1;
}

// N(0,1) of the sheep's random walk (go1_sheep.py:43: randn_like per step) in MQE_NOISE_HASH mode: Box-Muller over the counter
// RNG keyed by (seed, GLOBAL env id, ordinal of the post-physics step, sheep * 3 + axis) -- independent of the GPU count
__device__ __forceinline__ float mqe_randn(const DevModel* m, int e, int step_no, uint32_t k) {
  const uint32_t cnt = MQE_RNG_NPC + (uint32_t)step_no, genv = (uint32_t)(e + m->env_id_offset);
  const float u1 = mqe_u01((uint32_t)m->seed, genv, cnt, 2u * k), u2 = mqe_u01((uint32_t)m->seed, genv, cnt, 2u * k + 1u);
  return sqrtf(-2.0f * logf(1.0f - u1)) * cosf(6.2831855f * u2);
}

// The sheep script (go1_sheep.py:35-64) in three pieces so that k_post_physics can spread an env's sheep over lanes: flock mean
// (+ the two logged statistics), one sheep's velocity increment from the pre-update state, and its write-back.  All increments are
// formed before any row is written, as in the reference's vectorised update.
// npc_rows: the env's NPC rows [P][13] before the update, rob_rows: its robots' rows [A][13] -- global memory, or the caller's LDS copies
// (post_body: a row read back from global memory behind a store costs a full round trip, and the script had six of them in a row)
typedef const __attribute__((address_space(3))) float* lds_cptr;
template <class NP>
__device__ __forceinline__ void sheep_flock_mean(const DevModel* m, NP npc_rows, float* avg) {
  const int P = m->P;
  avg[0] = 0; avg[1] = 0; avg[2] = 0;
  for (int p = 0; p < P; p++) for (int k = 0; k < 3; k++) avg[k] += npc_rows[p * 13 + k];
  for (int k = 0; k < 3; k++) avg[k] /= (float)P;
}
template <class NP>
__device__ __forceinline__ void sheep_flock_stats(const DevModel* m, const DevState& st, int e, NP npc_rows, const float* avg) {
  const int P = m->P;
  float var = 0;
  for (int k = 0; k < 2; k++) {
    float acc = 0;
    for (int p = 0; p < P; p++) { float t = npc_rows[p * 13 + k] - avg[k]; acc += t * t; }
    var += acc / (float)P;
  }
  st.sheep_avg[e * 2] = avg[0]; st.sheep_avg[e * 2 + 1] = avg[1];
  st.sheep_var[e] = var;
}
template <class NP, class RP>
__device__ __forceinline__ void sheep_increment(const DevModel* m, const DevState& st, int e, int p, NP npc_rows, RP rob_rows, const float* avg, int step_no, float* dv) {
  const int A = m->A, P = m->P;
  NP sp = npc_rows + p * 13;
  for (int k = 0; k < 3; k++) {
    // MQE_NOISE_SCRIPTED: the injected sequence (golden traces); otherwise a fresh draw every step, as the reference's randn_like
    const float z = m->noise_mode == MQE_NOISE_SCRIPTED ? st.npc_noise[((size_t)e * P + p) * 3 + k]
                                                        : (m->sheep_rand != 0.0f ? mqe_randn(m, e, step_no, (uint32_t)(p * 3 + k)) : 0.0f);
    dv[k] = m->sheep_rand * z * 2.0f;
  }
  if (P != 1) {
    float rel[3] = {avg[0] - sp[0], avg[1] - sp[1], avg[2] - sp[2]};
    float nr = sqrtf(rel[0] * rel[0] + rel[1] * rel[1] + rel[2] * rel[2]);
    // a sheep exactly on the flock mean (the centre of the 3 x 3 grid at reset): torch gives 0 / 0 = NaN, which the clip of
    // go1_sheep.py:59 passes on; the engine is built with -fno-honor-nans, so the cohesion term is dropped there instead
    if (nr > 0.0f) for (int k = 0; k < 3; k++) dv[k] += m->sheep_rand * rel[k] / nr / 1.5f;
  }
  for (int a = 0; a < A; a++) {
    RP dp = rob_rows + a * 13;
    float rel[3] = {sp[0] - dp[0], sp[1] - dp[1], sp[2] - dp[2]};
    float sq[3] = {rel[0] * rel[0], rel[1] * rel[1], rel[2] * rel[2]};
    float dis = sqrtf(sq[0] * sq[0] + sq[1] * sq[1] + sq[2] * sq[2]);
    float den = powf(dis, 1.4f);
    for (int k = 0; k < 3; k++) { float t = rel[k] / den; if (dis > 9.0f) t = 0.0f; dv[k] += m->sheep_scale * t; }
  }
  dv[2] = 0.0f;
}
// the sheep's new row entries from its row before the update (npc_rows) -- stores only
template <class NP>
__device__ __forceinline__ void sheep_apply(float* root, int A, int p, NP npc_rows, const float* dv) {
  float* sp = root + (A + p) * 13;
  NP old = npc_rows + p * 13;
  const float v0 = clampf(old[7] + dv[0], -2.0f, 2.0f), v1 = clampf(old[8] + dv[1], -2.0f, 2.0f), v2 = old[9] + dv[2];
  sp[7] = v0; sp[8] = v1; sp[9] = v2;
  sp[2] = clampf(old[2], 0.0f, 0.3f);
  sp[3] = 0.0f; sp[4] = 0.0f;
}
// wrapper observation + reward for env e.  npc = the `root_states_npc` rows the wrapper sees ([P][13]); see oracle.
// side_effects: 1 on the wrapper-level paths (mqe_step, mqe_wrapper_eval); the Go1-level mqe_post_physics_step passes 0 so that the
// state stays exactly what Go1.step leaves (go1tug re-poses its slider from the wrapper)
// BAG_LDS: `bag_in` is the caller's LDS copy of the env's observation rows (post_body), read as LDS (ds_read) -- as a generic pointer
// (null = "read the tensor") the rows were fetched with flat_load, whose completion counts on vmcnt like the stores in front of it: every
// element of the observation copy below waited for the previous element's store to be acknowledged (5.6 of the epilogue's 10.7 us on go1gate).
template <bool BAG_LDS = false>
__device__ __forceinline__ void wrapper_env_dev(const DevModel* m, const DevState& st, int e, int is_reset_call, const float* npc, int side_effects = 1, const float* bag_in = nullptr) {
  typedef typename std::conditional<BAG_LDS, const __attribute__((address_space(3))) float*, const float*>::type bag_ptr;
  bag_ptr bag;
  if constexpr (BAG_LDS) bag = (bag_ptr)bag_in;
  else bag = bag_in ? bag_in : st.obs_bag + (size_t)e * m->A * MQE_OBS_BAG;   // this env's rows
  int A = m->A, P = m->P, Aw = m->Aw, D = m->D;
  const int task = m->task;
  float* obs = st.wobs + (size_t)e * Aw * D;
  float* rew = st.wrew + (size_t)e * Aw;
  float* rs = st.rsum + (size_t)e * MQE_MAX_REWARD_TERMS;
  float sc[7];
#pragma unroll
  for (int k = 0; k < 7; k++) sc[k] = m->reward_scale[k];
  // Loads first, and no load behind a store (see post_body): the per-env constants the rows and the reward read, once; then -- for the tasks
  // whose reward does not touch the observation rows -- the reward (its loads are the first memory operations of the call), then the
  // observation rows, which read nothing but LDS and registers.  The rows and the reward share no tensor, so the order changes no value.
  const bool early = task == MQE_TASK_GATE || task == MQE_TASK_SHEEP || task == MQE_TASK_SEESAW || task == MQE_TASK_PUSHBOX || task == MQE_TASK_FOOTBALL_DEFENDER;
  float gp0 = 0.0f, gp1 = 0.0f, eol[3] = {0.0f, 0.0f, 0.0f}, eo0 = 0.0f, eo1 = 0.0f;
  if (task == MQE_TASK_GATE || task == MQE_TASK_SHEEP || task == MQE_TASK_PUSHBOX || (task == MQE_TASK_FOOTBALL_DEFENDER && !is_reset_call && (sc[0] != 0 || sc[1] != 0))) {
    gp0 = as_global(m->gate_pos)[e * 2]; gp1 = as_global(m->gate_pos)[e * 2 + 1];
  }
  if (task == MQE_TASK_PUSHBOX || task == MQE_TASK_FOOTBALL_DEFENDER)
    for (int k = 0; k < 3; k++) eol[k] = st.env_origins_live[e * 3 + k];
  if (task == MQE_TASK_SHEEP) { eo0 = as_global(m->env_origins)[e * 3]; eo1 = as_global(m->env_origins)[e * 3 + 1]; }
  if (early) {
    if (is_reset_call) { for (int a = 0; a < Aw; a++) rew[a] = 0; }
    else {
      float r_env = 0.0f;
      uint8_t was_reset = st.reset_buf[e];
      float rsv[7];                               // the env's reward sums: read once, accumulated in registers, written once (`rs[k] += v` is a
#pragma unroll
      for (int k = 0; k < 7; k++) rsv[k] = rs[k];   // load behind the previous term's store: one memory round trip per reward term)
      if (task == MQE_TASK_GATE) {
        // loads first (see k_post_physics), then the same sums in the same order as the straightforward loop nest
        float r_ag[MQE_MAX_AGENTS] = {0, 0, 0, 0}, bx[MQE_MAX_AGENTS] = {0, 0, 0, 0}, by[MQE_MAX_AGENTS] = {0, 0, 0, 0}, wl[MQE_MAX_AGENTS] = {0, 0, 0, 0};
        const uint8_t have = st.w_have_last[e];
        const float gx = gp0, colf = (float)st.collide_buf[e];
        float rs0 = rsv[0], rs1 = rsv[1], rs2 = rsv[2], rs3 = rsv[3];
        const float wp0 = m->wrapper_param[0], wp1 = m->wrapper_param[1];
#pragma unroll
        for (int a = 0; a < MQE_MAX_AGENTS; a++)
          if (a < A) {
            bag_ptr ob = bag + (a) * MQE_OBS_BAG;
            bx[a] = ob[0]; by[a] = ob[1];
            wl[a] = st.w_last[e * MQE_MAX_AGENTS + a];
          }
        float tsum = 0;
#pragma unroll
        for (int a = 0; a < MQE_MAX_AGENTS; a++)
          if (a < A) {
            const float tx = wp0, ty = (a == 0 ? 1.0f : -1.0f) * wp1;
            const float dist = sqrtf((bx[a] - tx) * (bx[a] - tx) + (by[a] - ty) * (by[a] - ty));
            const float last = have ? wl[a] : dist;
            tsum += last - dist;
            st.w_last[e * MQE_MAX_AGENTS + a] = dist;
          }
        st.w_have_last[e] = 1;
        if (was_reset) tsum = 0;
        tsum *= sc[0];
        const float col = sc[1] * colf;
        rs0 += tsum;
        rs1 += col;
#pragma unroll
        for (int a = 0; a < MQE_MAX_AGENTS; a++)
          if (a < A) {
            r_ag[a] += tsum;
            r_ag[a] += col;
            if (bx[a] > gx + 0.25f) { r_ag[a] += sc[2]; rs2 += sc[2]; }
            const int pa = A - 1 - a;
            const float px = pa == 0 ? bx[0] : (pa == 1 ? bx[1] : (pa == 2 ? bx[2] : bx[3]));
            const float py = pa == 0 ? by[0] : (pa == 1 ? by[1] : (pa == 2 ? by[2] : by[3]));
            const float d2 = (bx[a] - px) * (bx[a] - px) + (by[a] - py) * (by[a] - py);
            if (d2 < 0.25f) { const float pn = div_ieee(sc[3], d2); r_ag[a] += pn; rs3 += pn; }
          }
        rsv[0] = rs0; rsv[1] = rs1; rsv[2] = rs2; rsv[3] = rs3;
        float tot = 0;
#pragma unroll
        for (int a = 0; a < MQE_MAX_AGENTS; a++) if (a < A) tot += r_ag[a];
        for (int a = 0; a < A; a++) rew[a] = tot;
      } else if (task == MQE_TASK_SHEEP) {
        float gate_x = gp0;
        const float colf = (float)st.collide_buf[e], avg0 = st.sheep_avg[e * 2], avg1 = st.sheep_avg[e * 2 + 1], wl20 = st.w_last2[e * 2], svar = st.sheep_var[e];
        const uint8_t have = st.w_have_last[e], delayed = st.w_delayed_reset[e];
        if (sc[0] != 0) {
          int cnt = 0;
          for (int p = 0; p < P; p++) if ((npc[p * 13] - eo0) - gate_x > 0) cnt++;
          r_env = (float)cnt;
          rsv[0] += (float)cnt;
        }
        if (sc[1] != 0) { float c = sc[1] * colf; r_env += c; rsv[1] += c; }
        if (sc[2] != 0) {
          if (have) {
            float xm = avg0 - wl20;
            if (delayed) xm = 0;
            float v = sc[2] * xm;
            r_env += v; rsv[2] += v;
          }
          st.w_last2[e * 2] = avg0; st.w_last2[e * 2 + 1] = avg1;
          st.w_have_last[e] = 1;
        }
        if (sc[3] != 0) {
          float acc = 0;
          for (int p = 0; p < P; p++) {
            float x = npc[p * 13] - eo0, y = npc[p * 13 + 1] - eo1;
            float dg = sqrtf((x - gate_x) * (x - gate_x) + (y - gp1) * (y - gp1));
            float v = expf(-dg / 2.0f) * sc[3];
            if (x >= gate_x) v = sc[3];
            acc += v;
          }
          r_env += acc; rsv[3] += acc;
        }
        if (sc[4] != 0 || sc[5] != 0) {
          float v = sc[5] * (svar - 1.0f) + sc[4] * expf(svar / 2.0f - 1.0f);
          r_env += v; rsv[4] += v;
        }
        st.w_delayed_reset[e] = was_reset;
        for (int a = 0; a < Aw; a++) rew[a] = r_env;
      } else if (task == MQE_TASK_SEESAW) {
        float xs = 0, zs = 0, y2 = 0, wl[MQE_MAX_AGENTS] = {0, 0, 0, 0};
        const uint8_t have = st.w_have_last[e];
        const float colf = (float)st.collide_buf[e];
        const bool fell = sc[6] != 0 && (st.r_term[e] | st.p_term[e]) != 0;
#pragma unroll
        for (int a = 0; a < MQE_MAX_AGENTS; a++) if (a < A) wl[a] = st.w_last[e * MQE_MAX_AGENTS + a];
#pragma unroll
        for (int a = 0; a < MQE_MAX_AGENTS; a++)
          if (a < A) {
            bag_ptr ob = bag + (a) * MQE_OBS_BAG;
            const float x = ob[0];
            xs += x - (have ? wl[a] : x);             // (no last position yet: it is taken to be this one)
            st.w_last[e * MQE_MAX_AGENTS + a] = x;
            zs += ob[2]; y2 += ob[1] * ob[1];
          }
        st.w_have_last[e] = 1;
        if (sc[0] != 0) { if (was_reset) xs = 0; xs *= sc[0]; r_env += xs; rsv[0] += xs; }
        if (sc[1] != 0) { float v = sc[1] * (zs - 0.56f); r_env += v; rsv[1] += v; }
        if (sc[2] != 0) { float v = sc[2] * (y2 - 0.5f); r_env += v; rsv[2] += v; }
        if (sc[3] != 0) { float v = sc[3] * colf; r_env += v; rsv[3] += v; }
        if (sc[4] != 0) {
          bag_ptr o0 = bag; bag_ptr o1 = bag + (A - 1) * MQE_OBS_BAG;
          float d2 = (o0[0] - o1[0]) * (o0[0] - o1[0]) + (o0[1] - o1[1]) * (o0[1] - o1[1]);
          if (d2 < 0.25f) { float v = div_ieee(sc[4], d2); r_env += v; rsv[4] += v; }
        }
        if (sc[5] != 0) {
          int cnt = 0;
          for (int a = 0; a < A; a++) { bag_ptr ob = bag + (a) * MQE_OBS_BAG; if (ob[0] > 7.7f && ob[2] > 1.3f) cnt++; }
          float v = sc[5] * (float)cnt; r_env += v; rsv[5] += v;
        }
        if (fell) { r_env += sc[6]; rsv[6] += sc[6]; }
        for (int a = 0; a < Aw; a++) rew[a] = r_env;
      } else if (task == MQE_TASK_PUSHBOX) {                // go1_pushbox_wrapper.py:52-88
        const float bx = npc[0] - eol[0], wl20 = st.w_last2[e * 2];
        const uint8_t have = st.w_have_last[e];
        if (sc[0] != 0 && have) {
          float xm = bx - wl20;
          if (was_reset) xm = 0;                        // x_movement[reset_ids] = 0
          const float v = sc[0] * xm;
          r_env += v; rsv[0] += v;
        }
        st.w_last2[e * 2] = bx;
        st.w_have_last[e] = 1;
        for (int a = 0; a < Aw; a++) rew[a] = r_env;
      } else if (task == MQE_TASK_FOOTBALL_DEFENDER) {
        float bx = npc[0] - eol[0], by = npc[1] - eol[1];
        if (sc[0] != 0) { if (bx > gp0) { r_env += sc[0]; rsv[0] += sc[0]; } }
        if (sc[1] != 0) {
          float dg = sqrtf((bx - gp0) * (bx - gp0) + (by - gp1) * (by - gp1));
          float v = sc[1] * expf(-dg / 3.0f);
          r_env += v; rsv[1] += v;
        }
        for (int a = 0; a < Aw; a++) rew[a] = r_env;
      }
#pragma unroll
      for (int k = 0; k < 7; k++) rs[k] = rsv[k];
    }
  }
  for (int a = 0; a < Aw; a++) {
    float* o = obs + a * D;
    int c = 0;
    if (task == MQE_TASK_TUG) {                // go1_tug_wrapper.py:47-57: [base info, slider (pos, vel), distance to it, slider pos]
      const float npos = st.dof[((size_t)e * m->ND + 12 * A) * 2], nvel = st.dof[((size_t)e * m->ND + 12 * A) * 2 + 1];
      const float sgn = a == 1 ? -1.0f : 1.0f;    // agent 1 sees the mirrored scene: entries 1, 4, 6, 9 negated
      bag_ptr ob = bag + (a) * MQE_OBS_BAG;
      float v6[6];
      for (int k = 0; k < 6; k++) v6[k] = ob[k];
      const float dx = v6[0] - 1.6f, dy = v6[1] - npos;
      v6[1] *= sgn; v6[4] *= sgn;
      for (int k = 0; k < 6; k++) o[k] = v6[k];
      o[6] = sgn * npos; o[7] = nvel; o[8] = sqrtf(dx * dx + dy * dy); o[9] = sgn * npos;
      continue;
    }
    if (task != MQE_TASK_ROTATION && task != MQE_TASK_BRIDGE && task != MQE_TASK_WRESTLING)
      for (int k = 0; k < Aw; k++) o[c++] = (k == a) ? 1.0f : 0.0f;
    bag_ptr ob = bag + (a) * MQE_OBS_BAG;
    for (int k = 0; k < 6; k++) o[c++] = ob[k];
    if (task != MQE_TASK_PLAIN) {
      bag_ptr ob2 = bag + ((Aw - 1 - a)) * MQE_OBS_BAG;
      for (int k = 0; k < 6; k++) o[c++] = ob2[k];
    }
    if (task == MQE_TASK_GATE || task == MQE_TASK_SHEEP || task == MQE_TASK_PUSHBOX) { o[c++] = gp0; o[c++] = gp1; }
    if (task == MQE_TASK_PUSHBOX) {              // go1_pushbox_wrapper.py:44-48: box xy rel. env origin, box quaternion
      o[c++] = npc[0] - eol[0]; o[c++] = npc[1] - eol[1];
      for (int k = 0; k < 4; k++) o[c++] = npc[3 + k];
    }
    if (task == MQE_TASK_SHEEP)
      for (int p = 0; p < P; p++) { o[c++] = npc[p * 13] - eo0; o[c++] = npc[p * 13 + 1] - eo1; }
    if (task == MQE_TASK_FOOTBALL_DEFENDER) {
      for (int k = 0; k < 3; k++) o[c++] = npc[k] - eol[k];
      for (int k = 0; k < 3; k++) o[c++] = npc[7 + k];
    }
  }
  if (early) return;
  if (task == MQE_TASK_TUG) {                  // go1_tug_wrapper.py:59-136
    float* nd = st.dof + ((size_t)e * m->ND + 12 * A) * 2;
    const float npos = nd[0];
    bag_ptr ob0 = bag;
    bag_ptr ob1 = bag + (1) * MQE_OBS_BAG;
    const float x0 = ob0[0], y0 = ob0[1], x1 = ob1[0], y1 = ob1[1];
    if (is_reset_call) {                          // _init_extras (:37-40)
      st.w_last[e * MQE_MAX_AGENTS] = x0; st.w_last[e * MQE_MAX_AGENTS + 1] = y0;
      st.w_last2[e * 2] = npos;
      st.w_delayed_reset[e] = 0;
      for (int a = 0; a < Aw; a++) rew[a] = 0;
      return;
    }
    const float last_npc = st.w_last2[e * 2];
    const float lx = st.w_last[e * MQE_MAX_AGENTS] - 1.6f, ly = st.w_last[e * MQE_MAX_AGENTS + 1] - npos;
    const float cx = x0 - 1.6f, cy = y0 - npos;
    const float last_dis = sqrtf(lx * lx + ly * ly), dis = sqrtf(cx * cx + cy * cy);
    float r0 = 0.0f, sr = 0.0f, pu = 0.0f, pr = 0.0f, pp = 0.0f;
    if (sc[0] != 0) { if (npos < 0) sr = sc[0] * -npos; if (last_npc <= npos) sr /= 2; r0 += sr; rs[0] += sr; }
    if (sc[1] != 0) { if (npos > 0) pu = sc[1] * npos; if (last_npc > npos) pu /= 2; r0 -= pu; rs[1] += pu; }
    if (sc[2] != 0) { if (dis < last_dis) pr = (last_dis - dis) * sc[2]; r0 += pr; rs[2] += pr; }
    if (sc[3] != 0) { if (dis >= last_dis) pp = powf(2.0f, dis) * sc[3]; r0 -= pp; rs[3] += pp; }
    rs[4] += npos; rs[5] += pr + sr - pu; rs[6] += x0; rs[7] += y0; rs[8] += x1; rs[9] += y1;
    st.w_last[e * MQE_MAX_AGENTS] = x0; st.w_last[e * MQE_MAX_AGENTS + 1] = y0;
    st.w_last2[e * 2] = npos;
    rew[0] = r0;
    for (int a = 1; a < Aw; a++) rew[a] = 0;
    // slider re-zeroed for the two steps that follow an env reset (reset_dic, :61-71), see the oracle
    if (side_effects) {
      uint8_t cnt = st.reset_buf[e] ? 2 : st.w_delayed_reset[e];
      if (cnt > 0) { nd[0] = 0.0f; nd[1] = 0.0f; cnt--; }
      st.w_delayed_reset[e] = cnt;
    }
    return;
  }
  if (task == MQE_TASK_BRIDGE) {               // go1_bridge_wrapper.py
    bag_ptr ob0 = bag;
    bag_ptr ob1 = bag + (1) * MQE_OBS_BAG;
    const float x0 = ob0[0], z0 = ob0[2], x1 = ob1[0], z1 = ob1[2];
    float S = st.w_last[e * MQE_MAX_AGENTS], tgt = st.w_last[e * MQE_MAX_AGENTS + 1];
    if (is_reset_call) {                          // _init_extras (:27-29): target_pos = flip(base_pos at reset)
      S = fabsf(x1 + x0); tgt = x1;
      st.w_last[e * MQE_MAX_AGENTS] = S; st.w_last[e * MQE_MAX_AGENTS + 1] = tgt;
    }
    float* o1 = obs + 1 * D;                      // agent 1 walks the bridge the other way (:37-40, :76-79)
    o1[0] = S - o1[0]; o1[4] = -o1[4]; o1[6] = S - o1[6]; o1[10] = -o1[10];
    if (is_reset_call) { for (int a = 0; a < Aw; a++) rew[a] = 0; return; }
    float r0 = 0.0f;
    if (sc[0] != 0 && z1 < 0.5f) { r0 += sc[0]; rs[0] += sc[0]; }
    if (sc[1] != 0 && z0 < 0.5f) { r0 -= sc[1]; rs[1] += sc[1]; }
    if (sc[2] != 0 && x0 > tgt) { r0 += sc[2]; rs[2] += sc[2]; }
    rew[0] = r0;
    for (int a = 1; a < Aw; a++) rew[a] = 0;
    return;
  }
  if (task == MQE_TASK_WRESTLING) {            // go1_wrestling_wrapper.py
    float* o1 = obs + 1 * D;
    o1[1] = -o1[1]; o1[4] = -o1[4]; o1[7] = -o1[7]; o1[10] = -o1[10];
    if (is_reset_call) { for (int a = 0; a < Aw; a++) rew[a] = 0; return; }
    float r0 = 0.0f;
    bool down0, down1;
    {
      bag_ptr ob = bag;
      float r = ob[3], p = ob[4];
      if (r > 3.1415927f) r -= 6.2831855f;
      if (p > 3.1415927f) p -= 6.2831855f;
      down0 = fabsf(p) > 3.1415927f * 0.9f || fabsf(r) >= 3.1415927f * 0.4f;
      ob += MQE_OBS_BAG;
      r = ob[3]; p = ob[4];
      if (r > 3.1415927f) r -= 6.2831855f;
      if (p > 3.1415927f) p -= 6.2831855f;
      down1 = fabsf(p) > 3.1415927f * 0.9f || fabsf(r) >= 3.1415927f * 0.4f;
    }
    if (sc[0] != 0 && down1) { r0 += sc[0]; rs[0] += sc[0]; }
    if (sc[1] != 0 && down0) { r0 -= sc[1]; rs[1] += sc[1]; }
    rew[0] = r0;
    for (int a = 1; a < Aw; a++) rew[a] = 0;
    return;
  }
  if (task == MQE_TASK_ROTATION) {             // go1_rotation_wrapper.py:46-50,90-93: agent 1 sees the mirrored scene
    float* o1 = obs + 1 * D;
    o1[1] = -o1[1]; o1[4] = -o1[4]; o1[7] = -o1[7]; o1[10] = -o1[10];
    const float tgt = m->wrapper_param[0];
    bag_ptr ob0 = bag;
    bag_ptr ob1 = bag + (1) * MQE_OBS_BAG;
    const float x0 = ob0[0], y0 = ob0[1], x1 = ob1[0];
    if (is_reset_call) {                          // _init_extras (:30-38): only x is shifted by the target here
      st.w_last[e * MQE_MAX_AGENTS] = sqrtf((x0 - tgt) * (x0 - tgt) + y0 * y0);
      for (int a = 0; a < Aw; a++) rew[a] = 0;
      return;
    }
    float r0 = 0.0f;
    if (sc[0] != 0 && x0 > tgt) { r0 += sc[0]; rs[0] += sc[0]; }
    if (sc[1] != 0 && x1 > tgt) { r0 -= sc[1]; rs[1] += sc[1]; }
    if (sc[2] != 0) {     // the upstream broadcast subtracts the target from x AND y (:74); kept, see the oracle
      const float dis = sqrtf((x0 - tgt) * (x0 - tgt) + (y0 - tgt) * (y0 - tgt));
      if (dis < st.w_last[e * MQE_MAX_AGENTS]) { r0 += sc[2]; rs[2] += sc[2]; }
      st.w_last[e * MQE_MAX_AGENTS] = dis;
    }
    rew[0] = r0;
    for (int a = 1; a < Aw; a++) rew[a] = 0;
    return;
  }
  for (int a = 0; a < Aw; a++) rew[a] = 0;         // the remaining tasks (reset call or not)
}

// Launch geometry: POST_EPW envs per 64-lane wavefront, lane = (agent, env): lane a * POST_EPW + le holds robot a of the
// block's env le, so the per-robot work (state loads, body-frame quantities, gait clock, termination tests, observation
// row) runs once per lane instead of A times per env lane; the env-level work (flags, NPC script, reset, wrapper, push)
// stays on the agent-0 lane, which collects the other robots' termination bits with wave shuffles.  The body is latency-
// bound (row-strided accesses, one wave per SIMD), so what counts is the length of the per-lane instruction chain.
// Within a lane the global accesses are ordered LOADS FIRST: every input of the step (root row, joint states, actions,
// gait parameters, contact force, origins) is requested before the first store, because a load behind a store that may
// alias cannot be hoisted by the compiler and each such load costs a full memory round trip (the first version
// interleaved them per robot and per buffer: 37 us of s_waitcnt).
#ifndef POST_EPW
#define POST_EPW 8
#endif

// One block's rows of a [R][W] tensor (contiguous for the block's consecutive envs) from LDS to HBM, 16 B per lane.
__device__ __forceinline__ void post_flush_rows(float* __restrict__ g, const float* lds, const int n, const int tid) {
  const int n4 = n >> 2;
  for (int i = tid; i < n4; i += 64) reinterpret_cast<float4*>(g)[i] = reinterpret_cast<const float4*>(lds)[i];
  for (int i = (n4 << 2) + tid; i < n; i += 64) g[i] = lds[i];
}

// The body of the post-physics step for ONE wavefront that serves PEPW consecutive envs (blk = their group's index): the stand-alone
// kernel k_post_physics runs it with PEPW = POST_EPW and static LDS; the fused decimation kernel k_substeps runs it as its epilogue
// with PEPW = its envs per wavefront and the physics' dead LDS (mqe_step: one launch less and no second pass over the state).
// AM: compile-time bound of the agent lanes (>= m->A; 2 for the two-robot tasks, MQE_MAX_AGENTS otherwise); tid = lane;
// s_bag / s_la / s_npc: LDS of PEPW * AM * 74, PEPW * AM * 24 and PEPW * npc_stride floats, 16 B aligned.
// root_l / dof_l / act_l (fused epilogue only, else nullptr): the env's root rows, joint states and actions in LDS -- [PEPW] x ([A + P][13],
// [ND][2], [12 A]) at the given strides -- so that the robot lanes' 49 scattered global loads become LDS reads (in the epilogue every
// wavefront has 2 active lanes per vector-memory instruction: 8 x the instructions of the stand-alone kernel for the same bytes).
// LDS_STATE: root_l / dof_l / act_l are given -- a compile-time flag, not a null test: a pointer that is "LDS or global" at run time is a
// generic one, and its flat_load waits for every global store in front of it as well as for the LDS.
template <int AM, int PEPW, bool LDS_STATE = false>
__device__ __forceinline__ void post_body(const DevModel* m, const DevState& st, const int blk, const int tid, float* s_bag, float* s_la, float* s_npc,
                                          int wrapper_level, int push_count, int step_no,
                                          const float* root_l = nullptr, const float* dof_l = nullptr, const float* act_l = nullptr, int lds_env_stride = 0, int act_env_stride = 0,
                                          const int npc_stride = MQE_MAX_NPCS * 13, long long* taps = nullptr) {      // floats between two envs' NPC rows in s_npc (the fused epilogue packs them: P * 13 rounded up to 4)
  static_assert(PEPW * AM <= 64 && (PEPW & (PEPW - 1)) == 0, "agent lanes of PEPW envs must fit one wavefront");
#define ETAP(i) do { if (taps != nullptr && tid == 0) taps[i] = (long long)wall_clock64(); } while (0)
  const int A = m->A, P = m->P;
  const int le = tid & (PEPW - 1), a = tid / PEPW;
  const int e = blk * PEPW + le;
  const bool mine = e < m->N && a < A;          // this lane's robot exists
  const bool lead = e < m->N && a == 0;         // this lane does its env's env-level work
  const int i = e * A + a;
  const int nrow = min(PEPW, m->N - blk * PEPW) * A;   // robots of this block
  const float dtp = m->dt * (float)m->decimation;
  float* root = st.root + (size_t)e * (A + P) * 13;
  float* bag = s_bag + le * A * MQE_OBS_BAG;    // this env's rows, agent-major like the tensor
  typedef typename std::conditional<LDS_STATE, lds_cptr, const float*>::type state_ptr;
  state_ptr root_src;                           // the env's root rows as the physics left them
  if constexpr (LDS_STATE) root_src = (lds_cptr)(root_l + le * lds_env_stride); else root_src = root;
  float* la = s_la + le * A * 12;
  // ---- loads --------------------------------------------------------------------------------------------------------------
  float rs[13], gpar[5], gi0 = 0.f, f3[3], aoz = 0.f, dq[24], act[12], eo[3];
  int ep = 0;
  if (mine) {
    ep = st.ep_len[e] + 1;
#pragma unroll
    for (int k = 0; k < 3; k++) eo[k] = as_global(m->env_origins)[e * 3 + k];
#pragma unroll
    for (int k = 0; k < 13; k++) rs[k] = root_src[a * 13 + k];
    const float* lo = st.loco_obs + (size_t)i * MQE_FRAME;
#pragma unroll
    for (int k = 0; k < 5; k++) gpar[k] = lo[7 + k];
    gi0 = st.gait[i];
    const float* cf3 = st.cf + ((size_t)e * m->NBR + a * MQE_NREP) * 3;
#pragma unroll
    for (int k = 0; k < 3; k++) f3[k] = cf3[k];
    aoz = as_global(m->agent_origins)[(size_t)i * 3 + 2];
    state_ptr ds, as;
    if constexpr (LDS_STATE) { ds = (lds_cptr)(dof_l + le * lds_env_stride + a * 24); as = (lds_cptr)(act_l + le * act_env_stride + a * 12); }
    else { ds = st.dof + ((size_t)e * m->ND + a * 12) * 2; as = st.actions + (size_t)i * 12; }
#pragma unroll
    for (int k = 0; k < 24; k++) dq[k] = ds[k];
#pragma unroll
    for (int k = 0; k < 12; k++) act[k] = as[k];
  }
  // ---- body-frame quantities, gait clock, termination (registers only) ------------------------------------------------------
  float bq[4], lv[3], av[3], pgr[3], clk[4], gi1 = 0.f;
  float rpy[3] = {0.f, 0.f, 0.f};               // Euler angles of bq: the termination test's and the observation's (one evaluation: ~170 instructions a second one costs)
  unsigned fl = 0;                              // bit 0 base contact, 1 roll, 2 pitch, 3 z high, 4 z low
  if (mine) {
    const float q[4] = {rs[3], rs[4], rs[5], rs[6]}, v[3] = {rs[7], rs[8], rs[9]}, w[3] = {rs[10], rs[11], rs[12]};
    const float g3[3] = {0.0f, 0.0f, -1.0f};
#pragma unroll
    for (int k = 0; k < 4; k++) bq[k] = q[k];
    quat_rotate_inverse_f(q, v, lv);
    quat_rotate_inverse_f(q, w, av);
    quat_rotate_inverse_f(q, g3, pgr);
    const float f = gpar[0], ph = gpar[1], off = gpar[2], bnd = gpar[3], dur = gpar[4];
    float gi = gi0 + dtp * f;
    gi = gi - floorf(gi);
    gi1 = gi;
    float fi[4] = {gi + ph + off + bnd, gi + off, gi + bnd, gi + ph};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float r = fi[k] - floorf(fi[k]);
      if (r < dur) fi[k] = r * (0.5f / dur);
      else if (r > dur) fi[k] = 0.5f + (r - dur) * (0.5f / (1.0f - dur));
      clk[k] = sinf(6.2831855f * fi[k]);
    }
    if (m->terminate_on_base_contact && sqrtf(f3[0] * f3[0] + f3[1] * f3[1] + f3[2] * f3[2]) > 1.0f) fl |= 1u;
    euler_xyz_f(q, rpy);
    float r = rpy[0], p = rpy[1];
    if (r > 3.1415927f) r -= 6.2831855f;
    if (p > 3.1415927f) p -= 6.2831855f;
    const float z = rs[2] - aoz;
    if ((m->termination_flags & MQE_TERM_ROLL) && fabsf(r) > m->roll_thr) fl |= 2u;
    if ((m->termination_flags & MQE_TERM_PITCH) && fabsf(p) > m->pitch_thr) fl |= 4u;
    if ((m->termination_flags & MQE_TERM_Z_HIGH) && z > m->zhigh_thr) fl |= 8u;
    if ((m->termination_flags & MQE_TERM_Z_LOW) && z < m->zlow_thr) fl |= 16u;
  }
  ETAP(2);
  // any robot of the env: OR over the agent lanes (lane ^ PEPW, ^ 2 PEPW stay inside the PEPW * AM robot lanes)
#pragma unroll
  for (int d = PEPW; d < PEPW * AM; d <<= 1) fl |= (unsigned)__shfl_xor((int)fl, d);
  const uint8_t to = ep > m->max_episode_length, rterm = (fl >> 1) & 1, pterm = (fl >> 2) & 1, zh = (fl >> 3) & 1;
  const uint8_t reset = (uint8_t)(mine && (fl != 0 || to));       // the same value on all robot lanes of the env
  // ---- stores of the frame quantities and flags ------------------------------------------------------------------------------
  if (lead) {
    st.ep_len[e] = ep;
    st.time_out[e] = to;
    if (m->termination_flags & MQE_TERM_ROLL) st.r_term[e] = rterm;
    if (m->termination_flags & MQE_TERM_PITCH) st.p_term[e] = pterm;
    if (m->termination_flags & MQE_TERM_Z_HIGH) st.zh_term[e] = zh;
    st.reset_buf[e] = reset;
    st.wdone[e] = reset;                        // the flag once more, in the packed return batch (byte tail: a torch.bool view, no kernel)
    if (m->terminate_on_base_contact) st.collide_buf[e] = reset;
  }
  if (mine) {
#pragma unroll
    for (int k = 0; k < 3; k++) { st.blv[i * 3 + k] = lv[k]; st.bav[i * 3 + k] = av[k]; st.pg[i * 3 + k] = pgr[k]; }
#pragma unroll
    for (int k = 0; k < 4; k++) { st.bquat[i * 4 + k] = bq[k]; st.clock[i * 4 + k] = clk[k]; }
    st.gait[i] = gi1;
  }
  ETAP(3);
  // wrapper's view of root_states_npc: copy taken before the NPC script (legged_robot.py:136); xy/vel only are read
  // (staged in LDS by the whole wavefront: as a per-lane array of P * 13 floats it lived in scratch memory)
  float* npc_pre = s_npc + le * npc_stride;
  {
    const int e0 = blk * PEPW, nenv = min(PEPW, m->N - e0), per = P * 13;
    for (int t = tid; t < nenv * per; t += 64) {
      const int sl = t / per, r = t - sl * per;
      if constexpr (LDS_STATE) s_npc[sl * npc_stride + r] = ((lds_cptr)(root_l + sl * lds_env_stride))[A * 13 + r];
      else s_npc[sl * npc_stride + r] = st.root[((size_t)(e0 + sl) * (A + P) + A) * 13 + r];
    }
    __syncthreads();
  }
  ETAP(4);
  if (m->npc_kind == MQE_NPC_SHEEP) {           // wave-uniform.  The 64 / PEPW lanes of an env share its sheep (lane a: sheep a, a + 8, ..)
    constexpr int LPE = 64 / PEPW, NPASS = (MQE_MAX_NPCS + LPE - 1) / LPE;
    float dvs[NPASS][3];
    if (e < m->N) {
      float avg[3];
      // (the flock before the update = the LDS copy just taken; the robots' rows: the physics' LDS state or the tensor)
      sheep_flock_mean(m, (lds_cptr)npc_pre, avg);           // every lane of the env, the same loop -> the same bits
      if (lead) sheep_flock_stats(m, st, e, (lds_cptr)npc_pre, avg);
#pragma unroll
      for (int q = 0; q < NPASS; q++)
        if (a + q * LPE < P) sheep_increment(m, st, e, a + q * LPE, (lds_cptr)npc_pre, root_src, avg, step_no, dvs[q]);
    }
    __syncthreads();                            // every increment is formed from the pre-update flock
    if (e < m->N) {
#pragma unroll
      for (int q = 0; q < NPASS; q++)
        if (a + q * LPE < P) sheep_apply(root, A, a + q * LPE, (lds_cptr)npc_pre, dvs[q]);
    }
    // (a WORKGROUP-scope fence: writer and readers are lanes of this one wavefront, i.e. one CU and one vector L1.  The device-scope
    // __threadfence() that stood here writes the XCD's L2 back and invalidates it on gfx950 -- its L2s are not coherent with each other --
    // which cost every step of every sheep task ~15 us: go1sheep-hard k_post_physics 38.6 -> see profiles/r03_bench_go1sheep-hard.json)
    __threadfence_block();
    __syncthreads();                            // the sheep rows are final before the lead lane's reset / wrapper reads
  }
  ETAP(5);
  if (lead) {
    if (reset) {                                // rare: the reset writes memory, the robot lanes refresh their registers from it
      reset_env_dev(m, st, e);
      for (int k = 0; k < P * 13; k++) npc_pre[k] = root[A * 13 + k];
    }
  }
  if (__ballot(reset != 0)) {                   // wave-uniform
    __threadfence_block();                      // the lead lane's stores are visible to the other robot lanes of this wavefront (same CU, same L1)
    __syncthreads();
    if (reset) {
#pragma unroll
      for (int k = 0; k < 13; k++) rs[k] = root[a * 13 + k];
      const float* ds = st.dof + ((size_t)e * m->ND + a * 12) * 2;
#pragma unroll
      for (int k = 0; k < 24; k++) dq[k] = ds[k];
      if (P == 0) {
#pragma unroll
        for (int k = 0; k < 4; k++) { bq[k] = rs[3 + k]; st.bquat[i * 4 + k] = bq[k]; }
        euler_xyz_f(bq, rpy);
      }
    }
  }
  ETAP(6);
  // ---- compute_observations (legged_robot_field.py:117-146) from registers; obs.last_last_action aliases the current action
  if (mine) {
    float* ob = bag + a * MQE_OBS_BAG;
#pragma unroll
    for (int k = 0; k < 3; k++) { ob[k] = rs[k] - eo[k]; ob[3 + k] = rpy[k]; }
#pragma unroll
    for (int j = 0; j < 12; j++) {
      ob[6 + j] = (dq[2 * j] - m->default_dof_pos[j]) * 1.0f;
      ob[18 + j] = dq[2 * j + 1] * 0.05f;
      ob[36 + j] = act[j];
      ob[48 + j] = act[j];
      la[a * 12 + j] = act[j];
      la[PEPW * AM * 12 + a * 12 + j] = dq[2 * j + 1];             // legged_robot.py:152
    }
#pragma unroll
    for (int k = 0; k < 3; k++) { ob[30 + k] = lv[k] * 2.0f; ob[33 + k] = av[k] * 0.25f; ob[60 + k] = pgr[k]; }
#pragma unroll
    for (int k = 0; k < 4; k++) { ob[63 + k] = clk[k]; ob[67 + k] = bq[k]; }
#pragma unroll
    for (int k = 71; k < MQE_OBS_BAG; k++) ob[k] = 0.0f;               // row padding (the LDS copy is stored whole)
  }
  __syncthreads();
  ETAP(7);
  if (lead) {
    wrapper_env_dev<true>(m, st, e, 0, npc_pre, wrapper_level, bag);
    // _push_robots (go1.py:237, legged_robot.py:470-476): after this step's frame quantities were taken, before reset_idx --
    // whose U(-0.5, 0.5) base velocities replace the push in the envs that reset.  One draw per robot (the reference draws
    // (num_envs, 2) for a (num_envs * num_agents, 2) slice, which only broadcasts for a single agent).
    if (push_count > 0 && !reset)
      for (int b = 0; b < A; b++) {
        float* rv = root + b * 13 + 7;
        rv[0] = mqe_rand(m, e, (int)(MQE_RNG_PUSH + (uint32_t)push_count), (uint32_t)(2 * b), -m->max_push, m->max_push);
        rv[1] = mqe_rand(m, e, (int)(MQE_RNG_PUSH + (uint32_t)push_count), (uint32_t)(2 * b + 1), -m->max_push, m->max_push);
      }
  }
  ETAP(8);
  {
    const size_t r0 = (size_t)blk * PEPW * A;
    post_flush_rows(st.obs_bag + r0 * MQE_OBS_BAG, s_bag, nrow * MQE_OBS_BAG, tid);
    post_flush_rows(st.last_actions + r0 * 12, s_la, nrow * 12, tid);
    post_flush_rows(st.last_dof_vel + r0 * 12, s_la + PEPW * AM * 12, nrow * 12, tid);
  }
  // go1.py:145: history[agent_ids] = 0 for the envs that reset this step -- rare, so the whole wavefront zeroes them,
  // 16 B per lane per request: the f32 ring and, when present, its two f16 planes
  ETAP(9);
  unsigned long long rm = __ballot(lead && reset != 0);
  while (rm) {
    const int l = __ffsll((long long)rm) - 1;
    rm &= rm - 1;
    const int er = blk * PEPW + l;
    const int per = m->A * (MQE_HIST * MQE_FRAME / 4);                 // float4 units of this env's robots (contiguous)
    float4* h4 = reinterpret_cast<float4*>(st.hist) + (size_t)er * per;
    for (int k = tid; k < per; k += 64) h4[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (st.hist2) {                                                    // same robots, 2 planes interleaved: the same bytes
      uint4* p4 = reinterpret_cast<uint4*>(st.hist2 + (size_t)er * m->A * (2 * MQE_HIST * MQE_H2_FRAME));
      const int per2 = m->A * (2 * MQE_HIST * MQE_H2_FRAME / 8);
      for (int k = tid; k < per2; k += 64) p4[k] = make_uint4(0u, 0u, 0u, 0u);
      if ((int)tid < m->A) st.hist_irr[(size_t)er * m->A + tid] = 0u;      // all frames zero: every frame continues its predecessor
    }
  }
}

template <int AM>
__global__ void __launch_bounds__(64) k_post_physics(const DevModel* m, DevState st, int wrapper_level, int push_count, int step_no) {
  // The per-robot rows this block produces (obs bag 74, last action 12, last dof velocity 12) are contiguous in HBM over
  // the block's envs: the robot lanes write them to LDS (the wrapper reads the obs rows back from there, not through L2)
  // and the whole wavefront stores them 16 B per lane.
  __shared__ float4 s_bag4[POST_EPW * AM * MQE_OBS_BAG / 4], s_la4[POST_EPW * AM * 24 / 4];
  __shared__ float s_npc[POST_EPW * MQE_MAX_NPCS * 13];
  post_body<AM, POST_EPW>(m, st, blockIdx.x, threadIdx.x, reinterpret_cast<float*>(s_bag4), reinterpret_cast<float*>(s_la4), s_npc, wrapper_level, push_count, step_no);
}

// go1.py:145: history[agent_ids] = 0 for envs that reset this step.  One float4 per thread, R*540 threads.
__global__ void k_reset_history(const DevModel* m, DevState st) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int per = MQE_HIST * MQE_FRAME / 4;
  int i = idx / per;
  if (i >= m->R) return;
  if (!st.reset_buf[i / m->A]) return;
  reinterpret_cast<float4*>(st.hist)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (st.hist2) {      // the robot's compact f16 planes: 2 x 30 x 48 values = 360 16-byte words, one per thread of the first 360
    const int w = idx - i * per;
    if (w < 2 * MQE_HIST * MQE_H2_FRAME / 8)
      reinterpret_cast<uint4*>(st.hist2 + (size_t)i * (2 * MQE_HIST * MQE_H2_FRAME))[w] = make_uint4(0u, 0u, 0u, 0u);
    if (w == 0) st.hist_irr[i] = 0u;
  }
}

// mqe_history_sync: the compact split-f16 operand of layer 0 (and the continuity bits) rebuilt from the f32 ring, for a host that has
// WRITTEN MQE_T_HISTORY (the engine itself keeps the two in step frame by frame: pre_policy_store).  One thread per (robot, ring slot):
// the slot's 46 stored columns, its presence flag (a frame is present unless all of its 70 entries are zero: what a reset leaves), its
// carrier column (the oldest frame's last_two_locomotion_action, component j on the frame at logical position MQE_H2_CARRIER0 + j, zero
// elsewhere) and its continuity bit (columns 54..65 against the previous slot's 42..53, bit for bit).  `oldest` = ring slot of logical
// frame 0 = the slot the next step overwrites.
__global__ void k_hist2_rebuild(const DevModel* m, DevState st, int oldest) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = idx / MQE_HIST, slot = idx - i * MQE_HIST;
  if (i >= m->R || !st.hist2) return;
  const float* fr = st.hist + ((size_t)i * MQE_HIST + slot) * MQE_FRAME;
  uint16_t* row2 = st.hist2 + (size_t)i * (2 * MQE_HIST * MQE_H2_FRAME);
  unsigned any = 0u;
  for (int c = 0; c < 70; c++) {
    const float v = fr[c];
    any |= __float_as_uint(v) & 0x7FFFFFFFu;
    const int cc = h2_col(c);
    if (cc >= 0) {
      uint16_t h, l;
      split2(v, MQE_H2_ASCALE, h, l);
      const size_t k = (size_t)slot * MQE_H2_FRAME + cc;
      row2[h2_index(k, 0)] = h; row2[h2_index(k, 1)] = l;
    }
  }
  uint16_t h, l;
  split2(any ? 1.0f : 0.0f, MQE_H2_ASCALE, h, l);
  size_t k = (size_t)slot * MQE_H2_FRAME + MQE_H2_FLAG_COL;
  row2[h2_index(k, 0)] = h; row2[h2_index(k, 1)] = l;
  int p = slot - oldest; if (p < 0) p += MQE_HIST;             // logical position of this slot's frame
  float carry = 0.0f;
  if (p >= MQE_H2_CARRIER0 && p < MQE_H2_CARRIER0 + 12) carry = st.hist[((size_t)i * MQE_HIST + oldest) * MQE_FRAME + 54 + (p - MQE_H2_CARRIER0)];
  split2(carry, MQE_H2_ASCALE, h, l);
  k = (size_t)slot * MQE_H2_FRAME + MQE_H2_CARRIER_COL;
  row2[h2_index(k, 0)] = h; row2[h2_index(k, 1)] = l;
  const int prev = slot > 0 ? slot - 1 : MQE_HIST - 1;
  const float* pa = st.hist + ((size_t)i * MQE_HIST + prev) * MQE_FRAME + 42;
  unsigned diff = 0u;
  for (int j = 0; j < 12; j++) diff |= __float_as_uint(pa[j]) ^ __float_as_uint(fr[54 + j]);
  if (diff) atomicOr(&st.hist_irr[i], 1u << slot);
}

// post_physics_step in the stages the reference's method has (include/mqe_hip.h mqe_post_physics_stage; oracle: post_stages): one thread
// per env, the per-env device functions of the fused kernel in the same arithmetic (results equal k_post_physics' to the last bit or two).
// Not the fast path: it exists so that a subclass's check_termination / _step_npc / reset_idx / compute_observations can run in between.
__global__ void __launch_bounds__(64) k_post_staged(const DevModel* m, DevState st, int stages, int wrapper_level, int push_count, int step_no) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m->N) return;
  const int A = m->A, P = m->P;
  float* root = st.root + (size_t)e * (A + P) * 13;
  float* npc_pre = st.npc_pre + (size_t)e * (P ? P : 1) * 13;
  const float dtp = m->dt * (float)m->decimation;
  if (stages & MQE_POST_FRAME) {
    const int ep = st.ep_len[e] + 1;
    unsigned fl = 0;
    for (int a = 0; a < A; a++) {
      const int i = e * A + a;
      float rs[13];
      for (int k = 0; k < 13; k++) rs[k] = root[a * 13 + k];
      const float q[4] = {rs[3], rs[4], rs[5], rs[6]}, v[3] = {rs[7], rs[8], rs[9]}, w[3] = {rs[10], rs[11], rs[12]};
      const float g3[3] = {0.0f, 0.0f, -1.0f};
      float lv[3], av[3], pgr[3], clk[4];
      quat_rotate_inverse_f(q, v, lv);
      quat_rotate_inverse_f(q, w, av);
      quat_rotate_inverse_f(q, g3, pgr);
      const float* lo = st.loco_obs + (size_t)i * MQE_FRAME;
      const float f = lo[7], ph = lo[8], off = lo[9], bnd = lo[10], dur = lo[11];
      float gi = st.gait[i] + dtp * f;
      gi = gi - floorf(gi);
      float fi[4] = {gi + ph + off + bnd, gi + off, gi + bnd, gi + ph};
      for (int k = 0; k < 4; k++) {
        const float r = fi[k] - floorf(fi[k]);
        if (r < dur) fi[k] = r * (0.5f / dur);
        else if (r > dur) fi[k] = 0.5f + (r - dur) * (0.5f / (1.0f - dur));
        clk[k] = sinf(6.2831855f * fi[k]);
      }
      const float* f3 = st.cf + ((size_t)e * m->NBR + a * MQE_NREP) * 3;
      if (m->terminate_on_base_contact && sqrtf(f3[0] * f3[0] + f3[1] * f3[1] + f3[2] * f3[2]) > 1.0f) fl |= 1u;
      float rpy[3];
      euler_xyz_f(q, rpy);
      float r = rpy[0], p = rpy[1];
      if (r > 3.1415927f) r -= 6.2831855f;
      if (p > 3.1415927f) p -= 6.2831855f;
      const float z = rs[2] - as_global(m->agent_origins)[(size_t)i * 3 + 2];
      if ((m->termination_flags & MQE_TERM_ROLL) && fabsf(r) > m->roll_thr) fl |= 2u;
      if ((m->termination_flags & MQE_TERM_PITCH) && fabsf(p) > m->pitch_thr) fl |= 4u;
      if ((m->termination_flags & MQE_TERM_Z_HIGH) && z > m->zhigh_thr) fl |= 8u;
      if ((m->termination_flags & MQE_TERM_Z_LOW) && z < m->zlow_thr) fl |= 16u;
      for (int k = 0; k < 3; k++) { st.blv[i * 3 + k] = lv[k]; st.bav[i * 3 + k] = av[k]; st.pg[i * 3 + k] = pgr[k]; }
      for (int k = 0; k < 4; k++) { st.bquat[i * 4 + k] = q[k]; st.clock[i * 4 + k] = clk[k]; }
      st.gait[i] = gi;
    }
    const uint8_t to = ep > m->max_episode_length, reset = (uint8_t)(fl != 0 || to);
    st.ep_len[e] = ep;
    st.time_out[e] = to;
    if (m->termination_flags & MQE_TERM_ROLL) st.r_term[e] = (fl >> 1) & 1;
    if (m->termination_flags & MQE_TERM_PITCH) st.p_term[e] = (fl >> 2) & 1;
    if (m->termination_flags & MQE_TERM_Z_HIGH) st.zh_term[e] = (fl >> 3) & 1;
    st.reset_buf[e] = reset;
    st.wdone[e] = reset;
    if (m->terminate_on_base_contact) st.collide_buf[e] = reset;
    for (int k = 0; k < P * 13; k++) npc_pre[k] = root[A * 13 + k];       // the wrapper's copy of the NPC rows, before the NPC script (legged_robot.py:136)
  }
  if ((stages & MQE_POST_NPC) && m->npc_kind == MQE_NPC_SHEEP) {
    float avg[3], dvs[MQE_MAX_NPCS][3];
    const float* nrows = root + A * 13;          // the live rows (a subclass hook may have changed them since the FRAME stage): all increments first, then the rows
    sheep_flock_mean(m, nrows, avg);
    sheep_flock_stats(m, st, e, nrows, avg);
    for (int p = 0; p < P; p++) sheep_increment(m, st, e, p, nrows, (const float*)root, avg, step_no, dvs[p]);    // every increment from the pre-update flock
    for (int p = 0; p < P; p++) sheep_apply(root, A, p, nrows, dvs[p]);
  }
  if (stages & MQE_POST_RESET) {
    const uint8_t reset = st.reset_buf[e];       // as it stands NOW: a subclass's check_termination may have changed it since FRAME
    st.wdone[e] = reset;
    if (m->terminate_on_base_contact) st.collide_buf[e] = reset;      // upstream's collide_buf IS reset_buf then (legged_robot.py:165): an override's edit shows in both
    if (reset) {
      reset_env_dev(m, st, e);
      for (int k = 0; k < P * 13; k++) npc_pre[k] = root[A * 13 + k];
      if (P == 0)
        for (int a = 0; a < A; a++) for (int k = 0; k < 4; k++) st.bquat[(e * A + a) * 4 + k] = root[a * 13 + 3 + k];
      // go1.py:145: the history of the env's robots (the f32 ring and, when present, its compact f16 planes)
      float4* h4 = reinterpret_cast<float4*>(st.hist) + (size_t)e * A * (MQE_HIST * MQE_FRAME / 4);
      for (int k = 0; k < A * (MQE_HIST * MQE_FRAME / 4); k++) h4[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (st.hist2) {
        uint4* p4 = reinterpret_cast<uint4*>(st.hist2 + (size_t)e * A * (2 * MQE_HIST * MQE_H2_FRAME));
        for (int k = 0; k < A * (2 * MQE_HIST * MQE_H2_FRAME / 8); k++) p4[k] = make_uint4(0u, 0u, 0u, 0u);
        for (int a = 0; a < A; a++) st.hist_irr[(size_t)e * A + a] = 0u;
      }
    }
  }
  if (stages & MQE_POST_OBS) {
    compute_observations_env(m, st, e, 1);
    for (int k = 0; k < 12 * A; k++) {
      st.last_actions[(size_t)e * 12 * A + k] = st.actions[(size_t)e * 12 * A + k];
      st.last_dof_vel[(size_t)e * 12 * A + k] = st.dof[((size_t)e * m->ND + k) * 2 + 1];
    }
  }
  if (stages & MQE_POST_WRAPPER) {
    wrapper_env_dev(m, st, e, 0, npc_pre, wrapper_level);
    if (push_count > 0 && !st.reset_buf[e])
      for (int b = 0; b < A; b++) {
        float* rv = root + b * 13 + 7;
        rv[0] = mqe_rand(m, e, (int)(MQE_RNG_PUSH + (uint32_t)push_count), (uint32_t)(2 * b), -m->max_push, m->max_push);
        rv[1] = mqe_rand(m, e, (int)(MQE_RNG_PUSH + (uint32_t)push_count), (uint32_t)(2 * b + 1), -m->max_push, m->max_push);
      }
  }
}

__global__ void __launch_bounds__(64) k_reset_all(const DevModel* m, DevState st, int no_post_step_yet) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m->N) return;
  int A = m->A, P = m->P;
  reset_env_dev(m, st, e);
  const float* root = st.root + (size_t)e * (A + P) * 13;
  if (P == 0 || no_post_step_yet)
    for (int a = 0; a < A; a++) for (int k = 0; k < 4; k++) st.bquat[(e * A + a) * 4 + k] = root[a * 13 + 3 + k];
  compute_observations_env(m, st, e, 0);
  st.w_have_last[e] = 0;
  st.w_delayed_reset[e] = 0;
  wrapper_env_dev(m, st, e, 1, root + A * 13);
}

__global__ void __launch_bounds__(64) k_wrapper_eval(const DevModel* m, DevState st, int is_reset_call) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m->N) return;
  if (is_reset_call) { st.w_have_last[e] = 0; st.w_delayed_reset[e] = 0; }
  wrapper_env_dev(m, st, e, is_reset_call, st.root + ((size_t)e * (m->A + m->P) + m->A) * 13);
}
