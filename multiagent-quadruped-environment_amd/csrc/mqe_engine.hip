// mqe_engine.hip -- C ABI (include/mqe_hip.h) of the MI355X-native MQE rollout engine: handle, device memory,
// kernel launches.  gfx950 only; built in-tree by __graft_entry__.build():
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC mqe_engine.hip -o libmqe_hip.so
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mqe_common.hpp"
#include "kernels_step.hpp"
#include "kernels_tail.hpp"
#include "kernels_gemm.hpp"
#include "kernels_physics.hpp"
#include "kernels_camera.hpp"

static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, const char* a = "") {
  snprintf(g_err, sizeof g_err, fmt, a);
  return code;
}
#define HIPCHK(x)                                                        \
  do {                                                                   \
    hipError_t _e = (x);                                                 \
    if (_e != hipSuccess) return fail(-100, "HIP error: %s", hipGetErrorString(_e)); \
  } while (0)

struct GemmLayer {
  float* Wt;      // device [Kpad][Npad]            (exact-f32 path, k_gemm_f32)
  float* bias;    // device [Npad]
  int K, N, Kpad, Npad;
  uint16_t* W2 = nullptr;   // device, 2 f16 planes of wscale * [Npad][Kpad3] (split-f16 path, k_gemm_h2)
  float wscale = 1.0f;
  uint16_t* Wfrag = nullptr;   // device, the same two planes in MFMA-fragment order (k_policy_tail), scaled by fscale
  float fscale = 1.0f;
  int Kpad3 = 0;
  std::vector<float> hW;    // host staging [Npad][Kpad3] until finalize_layer
};

enum { PROF_GEMM_L0 = 0, PROF_GEMM_REST, PROF_TORQUES, PROF_SIMULATE, PROF_POST, PROF_MISC, PROF_N };

struct mqe_sim {
  mqe_sim_desc d;
  DevModel hm;            // host copy
  DevModel* dm = nullptr; // device copy
  DevState st;
  std::vector<void*> allocs;
  std::vector<std::pair<void*, size_t>> state_bufs;     // the buffers of DevState, in allocation order (mqe_state_save / _load)
  bool reg_state = false;
  void* tens[MQE_T_COUNT];
  int N, A, P, R, ND, NBR, Aw, D;
  int hist_pos = 0, n_post_steps = 0;
  void (*substeps_fn)(const DevModel*, DevState, int, int, PostArgs) = nullptr;    // the k_substeps specialisation of this scene
  int substeps_shape = 0;             // SubShape
  int substeps_epw = 1;               // envs per wavefront of that kernel (2: two-robot scenes without objects at large batches)
  bool a2_scene = false;              // two robots, no objects: the scene k_simulate_a2 (phase taps) is compiled for
  int lag_pos = 0;                    // write slot of the action-lag ring (domain randomisation), advances per substep
  // policy network (fused first layer: [adaptation L0 | body L0 history part])
  GemmLayer l0;                       // K = 30*72 (ring), N = ada_h0 + body_h0
  std::vector<GemmLayer> ada_rest, body_rest;
  float *w_lat0 = nullptr, *w_lat1 = nullptr;   // body L0 rows for the two latent inputs
  int ada_h0, body_h0;
  float *P1 = nullptr, *bufA = nullptr, *bufB = nullptr, *lat = nullptr, *act_out = nullptr;
  int ldP1, ldbuf, ldlat, ldact;
  bool gemm_split = false;
  bool gemm_half = true;              // k_gemm_h2_mix: half tiles for a remainder of at most half a round (MQE_GEMM_HALF=0: full tiles only)
  bool cmd_general = false;           // command layout other than (x, y, yaw) -> entries 3-5 (desc.command_src): unfused entry points, exact-f32 layer 0
  bool phase_timed = false;           // MQE_PHASE_TIMES=1 at creation: k_substeps runs with its phase taps live (mqe_debug_phase_times)
  bool tail_fused = false;            // k_policy_tail: the reference network shapes (256-128-2 / 512-256-128-12 after layer 0)
  long long* tail_times = nullptr;    // MQE_TAIL_TIMES=1: [row blocks][16] stage stamps of the last k_policy_tail launch (mqe_debug_tail_times)
  size_t phys_lds_bytes = 0;
  bool fuse_substeps = true;
  bool fuse_post = true;              // the post-physics step as the epilogue of k_substeps (mqe_step & co; MQE_NO_FUSE_POST=1: its own launch)
  int dbg_stop_phase = -1;            // MQE_DEBUG_STOP_PHASE, read once at creation (tools/phase_counters.py)
  // profiling
  bool prof = false, prof_now = false;   // prof_now: this call is one of the sampled ones
  int step_open = 0;                     // 0: no step in flight; 1: after mqe_step_head; 2: after mqe_step_tail / mqe_step_begin (mqe_step_end closes)
  int prof_every = 1; long prof_step = 0;
  std::vector<hipEvent_t> ev0[PROF_N], ev1[PROF_N];
  float prof_ms[PROF_N];
  int prof_cnt[PROF_N];
};

extern "C" const char* mqe_last_error(void) { return g_err; }
extern "C" int mqe_abi_version(void) { return MQE_ABI_VERSION; }
extern "C" int mqe_abi_limits(int32_t* out, int n) {
  const int32_t v[] = {MQE_ABI_VERSION, MQE_MAX_AGENTS, MQE_MAX_NPCS, MQE_MAX_SPHERES, MQE_MAX_PRIMS, MQE_MAX_SELF_PAIRS, MQE_MAX_LAYERS, MQE_MAX_REWARD_TERMS,
                       MQE_NBODY, MQE_NREP, MQE_NDOF, MQE_FRAME, MQE_HIST, MQE_T_COUNT};
  const int cnt = (int)(sizeof v / sizeof v[0]);
  for (int i = 0; i < cnt && i < n; i++) out[i] = v[i];
  return cnt;
}
extern "C" int mqe_sizeof_desc(void) { return (int)sizeof(mqe_sim_desc); }

template <typename T>
static int dalloc(mqe_sim* s, T** p, size_t n, int fill_zero = 1) {
  void* q = nullptr;
  size_t bytes = (n ? n : 1) * sizeof(T);
  if (hipMalloc(&q, bytes) != hipSuccess) return -1;
  if (fill_zero && hipMemset(q, 0, bytes) != hipSuccess) return -1;
  s->allocs.push_back(q);
  if (s->reg_state) s->state_bufs.push_back({q, bytes});
  *p = (T*)q;
  return 0;
}
static int upload(mqe_sim* s, const float** dst, const float* src, size_t n) {
  if (!src || !n) { *dst = nullptr; return 0; }
  float* q;
  if (dalloc(s, &q, n, 0)) return -1;
  if (hipMemcpy(q, src, n * 4, hipMemcpyHostToDevice) != hipSuccess) return -1;
  *dst = q;
  return 0;
}
static inline int rup(int v, int m) { return (v + m - 1) / m * m; }

// W (out,in) row-major host -> Wt [Kpad][Npad] device at column offset; rows remapped by `rowmap` (-1 = zero row)
static int make_layer(mqe_sim* s, GemmLayer* L, int K, int N) {
  L->K = K; L->N = N; L->Kpad = rup(K, GB_K); L->Npad = rup(N, GB_N); L->Kpad3 = rup(K, H2_KMULT);
  L->hW.assign((size_t)L->Npad * L->Kpad3, 0.0f);
  if (dalloc(s, &L->Wt, (size_t)L->Kpad * L->Npad)) return -1;
  if (dalloc(s, &L->bias, L->Npad)) return -1;
  return 0;
}
// krow: input k -> row of the exact-f32 operand (ring-row order).  k2row / k2scale (layer 0 only): input k -> row of the split-f16
// operand in its own, compact K order; inputs that share a row (the constant columns on the flag column) are accumulated with
// their constant as the factor.
static int fill_layer(GemmLayer* L, int col0, const float* W, const float* b, int out, int in_used, int ldw_in,
                      const std::vector<int>* krow = nullptr, const std::vector<int>* k2row = nullptr, const std::vector<float>* k2scale = nullptr) {
  std::vector<float> tmp((size_t)L->Kpad * out, 0.0f);
  std::vector<double> acc;
  if (k2row) acc.assign((size_t)out * L->Kpad3, 0.0);
  for (int o = 0; o < out; o++)
    for (int k = 0; k < in_used; k++) {
      int kr = krow ? (*krow)[k] : k;
      tmp[(size_t)kr * out + o] = W[(size_t)o * ldw_in + k];
      if (k2row) acc[(size_t)o * L->Kpad3 + (*k2row)[k]] += (double)W[(size_t)o * ldw_in + k] * (double)(*k2scale)[k];
      else L->hW[(size_t)(col0 + o) * L->Kpad3 + kr] = W[(size_t)o * ldw_in + k];
    }
  if (k2row)
    for (int o = 0; o < out; o++)
      for (int k = 0; k < L->Kpad3; k++) L->hW[(size_t)(col0 + o) * L->Kpad3 + k] = (float)acc[(size_t)o * L->Kpad3 + k];
  if (hipMemcpy2D(L->Wt + col0, (size_t)L->Npad * 4, tmp.data(), (size_t)out * 4, (size_t)out * 4, L->Kpad, hipMemcpyHostToDevice) != hipSuccess) return -1;
  if (b && hipMemcpy(L->bias + col0, b, (size_t)out * 4, hipMemcpyHostToDevice) != hipSuccess) return -1;
  return 0;
}

// split the staged weights into the two f16 planes (scaled by a power of two that puts max|w| in [16384, 32768]) and upload them
static int finalize_layer(mqe_sim* s, GemmLayer* L) {
  const size_t n = (size_t)L->Npad * L->Kpad3;
  float wmax = 0.0f;
  for (size_t i = 0; i < n; i++) wmax = std::max(wmax, std::fabs(L->hW[i]));
  if (!(wmax < 1e30f)) return -1;
  int e = wmax > 0.0f ? (int)std::floor(std::log2(32768.0f / wmax)) : 0;
  e = std::min(std::max(e, -100), 100);
  L->wscale = std::ldexp(1.0f, e);
  std::vector<uint16_t> pl(2 * n);
  for (int r = 0; r < L->Npad; r++)
    for (int k = 0; k < L->Kpad3; k++) {
      uint16_t* row = pl.data() + (size_t)r * 2 * L->Kpad3;
      split2(L->hW[(size_t)r * L->Kpad3 + k], L->wscale, row[h2_index(k, 0)], row[h2_index(k, 1)]);
    }
  if (dalloc(s, &L->W2, 2 * n, 0)) return -1;
  if (hipMemcpy(L->W2, pl.data(), 2 * n * 2, hipMemcpyHostToDevice) != hipSuccess) return -1;
  L->hW.clear(); L->hW.shrink_to_fit();
  return 0;
}

// weights of a tail layer -> two f16 planes in MFMA-fragment order: block (column tile ct, 16-k step s, plane p) = 512 values,
// lane l holds the 8 k = 16 s + 8 (l / 32) .. of column 32 ct + l % 32 (kernels_tail.hpp)
static int finalize_frag(mqe_sim* s, GemmLayer* L) {
  if (L->K % 16 || L->Npad % 32) return -1;
  float wmax = 0.0f;
  for (float v : L->hW) wmax = std::max(wmax, std::fabs(v));
  if (!(wmax < 1e30f)) return -1;
  int e = wmax > 0.0f ? (int)std::floor(std::log2(32768.0f / wmax)) : 0;
  e = std::min(std::max(e, -100), 100);
  L->fscale = std::ldexp(1.0f, e);
  const int S = L->K / 16, CT = L->Npad / 32;
  std::vector<uint16_t> pl((size_t)CT * S * 2 * 512);
  for (int ct = 0; ct < CT; ct++)
    for (int st = 0; st < S; st++)
      for (int l = 0; l < 64; l++)
        for (int q = 0; q < 8; q++) {
          const int col = ct * 32 + (l & 31), k = st * 16 + (l >> 5) * 8 + q;
          uint16_t h, lo;
          split2(L->hW[(size_t)col * L->Kpad3 + k], L->fscale, h, lo);
          const size_t blk = ((size_t)ct * S + st) * 2;
          pl[(blk + 0) * 512 + l * 8 + q] = h;
          pl[(blk + 1) * 512 + l * 8 + q] = lo;
        }
  if (dalloc(s, &L->Wfrag, pl.size(), 0)) return -1;
  if (hipMemcpy(L->Wfrag, pl.data(), pl.size() * 2, hipMemcpyHostToDevice) != hipSuccess) return -1;
  return 0;
}

// k_substeps is compiled for the env shapes of the shipped tasks (kernels_physics.hpp: PhysShape); everything else takes the
// runtime form
typedef void (*substeps_fn_t)(const DevModel*, DevState, int, int, PostArgs);
enum SubShape { SH_A2, SH_A1, SH_A2_NOPAD, SH_A2_LINK, SH_A2_NPC_FEW, SH_A2_BOX_FEW, SH_A2_STATIC_FEW, SH_A3_NPC_ROW, SH_A2_NPC, SH_A4_NPC, SH_A2_GEN, SH_GEN };
static SubShape pick_shape(const DevModel& m, size_t lds_bytes, int pad) {
  const int feat = (m.has_seesaw ? PS_F_LINK : 0) | (m.n_npc_dyn > 0 ? PS_F_NPC : 0) | (m.has_box ? PS_F_BOX : 0) | (m.n_static > 0 ? PS_F_STATIC : 0);
  // the small class (row sweep compiled in, 128 VGPRs, every env resident): <= 4 actors and 16 LDS footprints per CU
  if (m.rowgs && lds_bytes <= 10240 && m.maxc <= 32 && m.A + m.n_npc_dyn + (m.has_seesaw ? 1 : 0) <= 4) {                       // (MQE_LANE_SWEEP=1 sends these scenes to the kernels below; <= 32 contacts: two record passes)
    if (feat == 0 && m.P == 0 && pad) {
      if (m.A == 2) return SH_A2;                                            // go1gate
      if (m.A == 1) return SH_A1;                                            // go1plane
    }
    if (feat == 0 && m.P == 0 && !pad && m.A == 2) return SH_A2_NOPAD;       // go1gate with the exact collision model (unpadded records)
    if (m.A == 2 && feat == PS_F_LINK) return SH_A2_LINK;                    // go1seesaw, go1revolvingdoor, go1tug
    if (m.A == 2 && feat == PS_F_NPC) return SH_A2_NPC_FEW;                  // go1football-1vs1, a single sheep
    if (m.A == 2 && feat == (PS_F_NPC | PS_F_BOX)) return SH_A2_BOX_FEW;     // go1pushbox
    if (m.A == 2 && feat == PS_F_STATIC) return SH_A2_STATIC_FEW;            // go1bridge, go1wrestling
  }
  // larger scenes: 2 waves per SIMD; the sweep variant is compiled in (the kernel's LDS layout has to be the one computed from m.rowgs)
  if (m.A == 3 && feat == PS_F_NPC && m.rowgs && m.maxc <= 32 && m.A + m.n_npc_dyn <= 4) return SH_A3_NPC_ROW;         // go1football-defender
  if (m.A == 2 && feat == PS_F_NPC && m.rowgs) return SH_A2_NPC;             // go1sheep-* (flocks): robots in rows, sheep in lanes
  if (m.A == 4 && feat == PS_F_NPC && m.rowgs) return SH_A4_NPC;             // go1football-2vs2
  if (m.A == 2) return SH_A2_GEN;
  return SH_GEN;
}
// the kernel of a shape: envs per wavefront (2: SH_A2 / SH_A1 only), phase taps live (TIMED: three shapes, f16 actuator only), and the
// actuator network's layer 2 as the exact f32 MFMA chain (ACT32: MQE_ACT_F32=1) instead of the split-f16 form
template <bool ACT32>
static substeps_fn_t shape_fn(SubShape sh, int epw) {
  switch (sh) {
    case SH_A2: return epw == 2 ? (substeps_fn_t)k_substeps<2, 0, 2, false, ACT32> : (substeps_fn_t)k_substeps<2, 0, 1, false, ACT32>;
    case SH_A1: return epw == 2 ? (substeps_fn_t)k_substeps<1, 0, 2, false, ACT32> : (substeps_fn_t)k_substeps<1, 0, 1, false, ACT32>;
    case SH_A2_NOPAD: return k_substeps<2, PS_F_FEW, 1, false, ACT32>;
    case SH_A2_LINK: return k_substeps<2, PS_F_LINK, 1, false, ACT32>;
    case SH_A2_NPC_FEW: return k_substeps<2, PS_F_NPC | PS_F_FEW, 1, false, ACT32>;
    case SH_A2_BOX_FEW: return k_substeps<2, PS_F_NPC | PS_F_BOX | PS_F_FEW, 1, false, ACT32>;
    case SH_A2_STATIC_FEW: return k_substeps<2, PS_F_STATIC | PS_F_FEW, 1, false, ACT32>;
    case SH_A3_NPC_ROW: return k_substeps<3, PS_F_NPC | PS_F_ROW, 1, false, ACT32>;
    case SH_A2_NPC: return k_substeps<2, PS_F_NPC, 1, false, ACT32>;
    case SH_A4_NPC: return k_substeps<4, PS_F_NPC, 1, false, ACT32>;
    case SH_A2_GEN: return k_substeps<2, -1, 1, false, ACT32>;
    default: return k_substeps<0, -1, 1, false, ACT32>;
  }
}
static substeps_fn_t shape_fn_timed(SubShape sh) {
  if (sh == SH_A2) return k_substeps<2, 0, 1, true>;
  if (sh == SH_A2_NPC) return k_substeps<2, PS_F_NPC, 1, true>;
  if (sh == SH_A3_NPC_ROW) return k_substeps<3, PS_F_NPC | PS_F_ROW, 1, true>;
  return nullptr;
}

static int wrapper_dims(const mqe_sim_desc* d, int* Aw, int* D) {
  int A = d->num_agents, P = d->num_npcs;
  switch (d->task) {
    case MQE_TASK_GATE: *Aw = A; *D = 14 + A; break;
    case MQE_TASK_SHEEP: *Aw = A; *D = 14 + 2 * P + A; break;
    case MQE_TASK_SEESAW: *Aw = A; *D = 12 + A; break;
    case MQE_TASK_FOOTBALL_DEFENDER: *Aw = 2; *D = 20; break;
    case MQE_TASK_PUSHBOX: *Aw = A; *D = 20 + A; break;
    case MQE_TASK_ROTATION: case MQE_TASK_BRIDGE: case MQE_TASK_WRESTLING: *Aw = A; *D = 12; break;
    case MQE_TASK_TUG: *Aw = A; *D = 10; break;
    default: *Aw = A; *D = 6 + A; break;
  }
  return 0;
}

// M-tiles of k_gemm_h2 for M rows and ntn column blocks: whole rounds of full tiles (a round = 256 / ntn M-tiles, one block per CU); the
// remainder as full tiles too when it fills more than half a round, else as half tiles (MQE_GEMM_HALF=0: never).  A half tile costs a CU
// ~0.6 of a full one, so a remainder of r <= half a round takes 0.6 of a round's time on 2 r tiles instead of a whole round on r.
static void h2_tiling(int M, int ntn, bool use_half, int* full_tiles, int* half_tiles) {
  const int ntm = (M + H2_M - 1) / H2_M, per_round = 256 / ntn > 0 ? 256 / ntn : 1;
  const int rem = ntm % per_round;
  *full_tiles = ntm; *half_tiles = 0;
  if (use_half && rem != 0 && 2 * rem <= per_round) {
    *full_tiles = ntm - rem;
    *half_tiles = (M - *full_tiles * H2_M + H2_M / 2 - 1) / (H2_M / 2);
  }
}
extern "C" int mqe_sim_create(const mqe_sim_desc* d, mqe_sim** out) {
  if (!d || !out) return fail(-1, "null argument");
  if (d->abi_version != MQE_ABI_VERSION) return fail(-1, "abi version mismatch");
  if (d->num_agents < 1 || d->num_agents > MQE_MAX_AGENTS)
    return fail(-2, "num_agents must be 1 .. 4: one env is one 64-lane wavefront in the physics kernel (a body per lane: 13 per Go1 + the NPCs), a fifth robot does not fit");
  if (d->num_npcs < 0 || d->num_npcs > MQE_MAX_NPCS) return fail(-2, "num_npcs must be 0 .. 16 (MQE_MAX_NPCS)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(-3, "no HIP device: the engine has no CPU path");
  mqe_sim* s = new mqe_sim();
  // every error exit below releases the handle and whatever it has allocated so far; success disarms the guard
  struct Guard { mqe_sim* s; ~Guard() { if (s) mqe_sim_destroy(s); } } guard{s};
  s->d = *d;
  const int N = s->N = d->num_envs, A = s->A = d->num_agents, P = s->P = d->num_npcs;
  const int R = s->R = N * A;
  const int seesaw = d->npc_kind == MQE_NPC_SEESAW;
  s->ND = 12 * A + (seesaw ? P : 0);
  s->NBR = MQE_NREP * A + (seesaw ? 2 * P : (d->npc_kind == MQE_NPC_STATIC ? d->npc_reported_bodies * P : P));
  wrapper_dims(d, &s->Aw, &s->D);
  memset(s->tens, 0, sizeof s->tens);
  memset(s->prof_ms, 0, sizeof s->prof_ms);
  memset(s->prof_cnt, 0, sizeof s->prof_cnt);
  DevModel& m = s->hm;
  memset(&m, 0, sizeof m);
  m.N = N; m.A = A; m.P = P; m.R = R; m.ND = s->ND; m.NBR = s->NBR; m.Aw = s->Aw; m.D = s->D;
  m.npc_kind = d->npc_kind; m.task = d->task;
  m.n_npc_dyn = (d->npc_kind == MQE_NPC_BALL || d->npc_kind == MQE_NPC_SHEEP || d->npc_kind == MQE_NPC_BOX) ? P : 0;
  m.self_collision = d->self_collision != 0 && d->robot.n_self_pairs > 0;
  m.lag_steps = (d->control_type == MQE_CTRL_C && d->lag_timesteps > 0) ? d->lag_timesteps : 0; m.max_push = d->max_push_vel_xy;
  m.has_box = d->npc_kind == MQE_NPC_BOX; m.cap_npc = d->npc_contact_cap > 0 ? d->npc_contact_cap : 2;
  memcpy(m.npc_box_half, d->npc_box_half, sizeof m.npc_box_half);
  m.npc_lin_only = d->npc_kind == MQE_NPC_SHEEP;
  m.npc_dofs_each = seesaw ? 1 : (m.npc_lin_only ? 3 : 6);
  m.env_id_offset = d->env_id_offset; m.seed = d->seed; m.noise_mode = d->noise_mode;
  m.dt = d->dt; m.decimation = d->decimation; m.gravity_z = d->gravity_z; m.solver_iterations = d->solver_iterations;
  m.contact_offset = d->contact_offset; m.max_depen = d->max_depenetration_velocity; m.friction = d->friction; m.erp = d->erp;
  m.solver_type = d->solver_type == 1 ? 1 : 0; m.vel_iters = d->velocity_iterations > 0 ? d->velocity_iterations : 0;
  if (m.solver_type == 1 && d->solver_iterations < 1) return fail(-6, "solver_type = 1 (temporal Gauss-Seidel) needs num_position_iterations >= 1: its sub-step is dt / n");
  if (d->solver_iterations < 0 || d->solver_iterations > 64) return fail(-6, "solver_iterations out of range (0 .. 64)");
  m.robot = d->robot;
  m.npc_mass = d->npc_mass; m.npc_inertia = d->npc_inertia; m.npc_n_spheres = d->npc_n_spheres;
  memcpy(m.npc_sphere_center, d->npc_sphere_center, sizeof m.npc_sphere_center);
  memcpy(m.npc_sphere_radius, d->npc_sphere_radius, sizeof m.npc_sphere_radius);
  m.seesaw_default_angle = d->seesaw_default_angle;
  m.n_static = d->npc_kind == MQE_NPC_STATIC ? d->n_static_boxes : 0;
  memcpy(m.sb_center, d->static_box_center, sizeof m.sb_center); memcpy(m.sb_half, d->static_box_half, sizeof m.sb_half);
  m.has_seesaw = seesaw; m.ss_axis = (d->seesaw_axis == 2 || d->seesaw_axis == 3) ? d->seesaw_axis : 1; m.ss_link_cyl = d->seesaw_link_cylinder;
  memcpy(m.ss_joint_offset, d->seesaw_joint_offset, 12); memcpy(m.ss_plank_center, d->seesaw_plank_center, 12);
  memcpy(m.ss_plank_half, d->seesaw_plank_half, 12); memcpy(m.ss_base_half, d->seesaw_base_half, 12);
  m.ss_inertia = d->seesaw_plank_inertia_yy; m.ss_vel_limit = d->seesaw_vel_limit;
  m.ss_col_radius = d->seesaw_column_radius; m.ss_col_length = d->seesaw_column_length;
  m.ss_theta_lo = d->seesaw_theta_lo; m.ss_theta_hi = d->seesaw_theta_hi;
  m.control_type = d->control_type; m.action_scale = d->action_scale; m.hip_scale_reduction = d->hip_scale_reduction;
  m.clip_actions = d->clip_actions; memcpy(m.torque_limits, d->torque_limits, sizeof m.torque_limits);
  m.kp = d->kp; m.kd = d->kd; memcpy(m.default_dof_pos, d->default_dof_pos, sizeof m.default_dof_pos);
  memcpy(m.command_obs, d->command_obs, sizeof m.command_obs);
  m.cmd_lin_scale = d->cmd_lin_scale; m.cmd_ang_scale = d->cmd_ang_scale; m.clip_command = d->clip_command;
  m.cmd_dims = d->num_command_dims; m.cmd_general = d->num_command_dims != 3;
  for (int c = 0; c < 18; c++) {
    m.cmd_src[c] = d->command_src[c]; m.cmd_scale[c] = d->command_scale[c];
    if (d->command_src[c] >= d->num_command_dims) return fail(-6, "command_src points past num_command_dims");
    if (d->command_src[c] != (c >= 3 && c < 6 ? c - 3 : -1)) m.cmd_general = 1;
  }
  if (d->num_command_dims < 1) return fail(-6, "num_command_dims must be >= 1");
  s->cmd_general = m.cmd_general != 0;
  m.sdf_nx = d->sdf_nx; m.sdf_ny = d->sdf_ny; m.hs = d->horizontal_scale; m.wall_height = d->wall_height; m.ground_z = d->ground_z;
  m.termination_flags = d->termination_flags; m.terminate_on_base_contact = d->terminate_on_base_contact; m.max_episode_length = d->max_episode_length;
  m.roll_thr = d->roll_threshold; m.pitch_thr = d->pitch_threshold; m.zlow_thr = d->z_low_threshold; m.zhigh_thr = d->z_high_threshold;
  m.dof_ratio_lo = d->dof_ratio_lo; m.dof_ratio_hi = d->dof_ratio_hi; m.has_base_pos_range = d->has_base_pos_range; m.has_npc_pos_range = d->has_npc_pos_range;
  m.base_pos_x_lo = d->base_pos_x_lo; m.base_pos_x_hi = d->base_pos_x_hi; m.base_pos_y_lo = d->base_pos_y_lo; m.base_pos_y_hi = d->base_pos_y_hi;
  m.npc_pos_x_lo = d->npc_pos_x_lo; m.npc_pos_x_hi = d->npc_pos_x_hi; m.npc_pos_y_lo = d->npc_pos_y_lo; m.npc_pos_y_hi = d->npc_pos_y_hi;
  m.base_vel_lo = d->base_vel_lo; m.base_vel_hi = d->base_vel_hi;
  m.sheep_scale = d->sheep_movement_scale; m.sheep_rand = d->sheep_movement_randomness;
  memcpy(m.reward_scale, d->reward_scale, sizeof m.reward_scale); memcpy(m.wrapper_param, d->wrapper_param, sizeof m.wrapper_param);
  // physics kernel geometry
  m.nbody_env = A * MQE_NBODY + m.n_npc_dyn;
  m.ndof_env = A * MQE_RD + m.n_npc_dyn * m.npc_dofs_each + (seesaw ? 1 : 0);
  m.nsph_env = A * d->robot.n_spheres + m.n_npc_dyn * d->npc_n_spheres;
  m.nprim_env = A * d->robot.n_prims;
  m.feat_sphere_mask = 0ull;
  for (int f = 0; f < d->robot.n_spheres; f++)
    if (d->robot.prim_type[d->robot.sphere_prim[f]] == MQE_PRIM_SPHERE) m.feat_sphere_mask |= 1ull << f;
  if (d->robot.n_spheres > MQE_MAX_SPHERES || d->robot.n_prims > MQE_MAX_PRIMS || d->robot.n_self_pairs > MQE_MAX_SELF_PAIRS) return fail(-6, "robot model: too many feature points / primitives / self-collision pairs");
  m.maxc = mqe_maxc(A, P, m.cap_npc);
  if (mqe_maxc_uncapped(A, P, m.cap_npc) > 64) return fail(-4, "the per-actor contact caps of this scene (8 per robot + cap per NPC + 8 two-actor slots) exceed the 64 contact lanes of the wavefront that owns an env");
  if (m.ndof_env > 128 || m.nbody_env > 64) { return fail(-4, "env has more than 64 bodies or 128 generalized velocities: does not fit one wavefront"); }
  m.rowgs = 1;      // one 16-lane row per actor (<= 4 actors), or (round 6) per ROBOT with the free NPCs' one-sided contacts stepped by a lane each; the compiled shapes assume it
  if (getenv("MQE_LANE_SWEEP")) m.rowgs = 0;      // tests: the other lane mapping of the contact sweep on the same scene (tests/test_gpu_parity.py)
  // padded link / contact records (kernels_physics.hpp: PhysPad): the robot-only kernels k_substeps<A, 0>, which the scene gets iff ...
  int pad = (m.rowgs && m.P == 0 && !m.has_seesaw && m.n_npc_dyn == 0 && !m.has_box && m.n_static == 0 && (A == 1 || A == 2)) ? 1 : 0;
  PhysLds L = phys_lds_layout(A, P, s->ND, m.nbody_env, m.ndof_env, m.nsph_env, m.nprim_env, m.maxc, m.rowgs, pad);
  // ... and whose padded layout still lets 16 envs sit on a CU.  The "exact" collision model (60 feature points per robot) needs 10768 B
  // padded and 10096 B unpadded: it takes the unpadded robots-only kernel (k_substeps<2, PS_F_FEW>: SH_A2_NOPAD) and stays in the 16-envs-per-CU
  // class (fused epilogue, 4 wavefronts per SIMD) instead of falling to the generic kernel at 2 per SIMD
  if (pad && (size_t)L.total * 4 > 10240) {
    pad = 0;
    L = phys_lds_layout(A, P, s->ND, m.nbody_env, m.ndof_env, m.nsph_env, m.nprim_env, m.maxc, m.rowgs, pad);
  }
  s->phys_lds_bytes = (size_t)L.total * 4;
  if (getenv("MQE_VERBOSE")) fprintf(stderr, "mqe: physics LDS %zu B per env (wavefront)\n", s->phys_lds_bytes);
  if (const char* pad = getenv("MQE_PHYS_LDS_PAD")) s->phys_lds_bytes += (size_t)atoi(pad);   // experiments: caps the physics kernel's waves per CU
  // the records the physics kernel moves as 16 B words must start on 16 B (kernels_physics.hpp)
  if ((L.total | L.body | L.sph | L.prim | L.con | L.side | L.leg | L.legc | L.basei | L.sinv | L.fcol | L.acc | L.rhs | L.phi | L.srec | L.wacc) & 3) { return fail(-4, "physics LDS layout: a 16 B record area is misaligned"); }
  if (s->phys_lds_bytes > 160 * 1024) { return fail(-4, "physics LDS footprint exceeds 160 KiB"); }
  {   // the actuator network's layer 2 on the f16 matrix cores (k_substeps<..., ACT32 = false>) unless MQE_ACT_F32=1 asks for the exact f32 chain
      // or the weights do not fit the f16 planes: |w| 2^14 must stay inside f16 (65504); the shipped net's largest layer-2 weight is 1.06
    float w1max = 0.0f;
    if (d->actuator.n_layers == 3 && d->actuator.dims[1] == 32 && d->actuator.dims[2] == 32 && d->actuator.W[1])
      for (int i = 0; i < 32 * 32; i++) w1max = std::max(w1max, std::fabs(d->actuator.W[1][i]));
    m.act_f16 = (getenv("MQE_ACT_F32") == nullptr && w1max < 3.99f) ? 1 : 0;
  }
  const SubShape shape = pick_shape(m, s->phys_lds_bytes, pad);
  s->substeps_shape = (int)shape;
  s->a2_scene = shape == SH_A2;
  const bool act32 = m.act_f16 == 0;                 // MQE_ACT_F32=1 (or a network whose weights do not fit the f16 planes): the exact f32 chain
  substeps_fn_t timed_fn = nullptr;                   // tools/dev/phase_walltimes.py: the same kernel with its phase taps live (go1gate and the two large scenes)
  if (getenv("MQE_PHASE_TIMES") && !act32) timed_fn = shape_fn_timed(shape);
  const bool phase_timed = timed_fn != nullptr;
  {
    // Two envs per wavefront (kernels_physics.hpp, EPW): for robot-only scenes of <= 2 robots each half-wave runs an env of its own --
    // 44 % fewer VALU instructions per env (the dynamics and sweep phases are shared, only contact generation runs per env), but half
    // as many wavefronts, each with twice the LDS traffic per instruction.  For two-robot envs it does NOT pay: the kernel is bound by the dependency
    // chain of a wavefront through the LDS (a lone one-env wavefront takes 82 us for its 58 k cycles of issue; LDS 60 % busy per
    // CU), and the two-env wavefront's chain is 1.5 x as long -- 2 of them per SIMD take what 4 one-env wavefronts take.  Kept as
    // a selectable, bit-identical variant (MQE_ENVS_PER_WAVE=1 / 2; tests hold the two forms against each other); the default for
    // single-robot scenes at full batches only (below).
    const bool can = (shape == SH_A2 || shape == SH_A1) && 2 * s->phys_lds_bytes <= 64 * 1024 && d->robot.n_spheres <= 32;      // (a half-wave tests 32 feature points per pass)
    // measured (MI355X, k_substeps us, one / two envs per wavefront): go1gate (two robots per env) 4096 envs 122 / 128, 8192 envs 238 / 237;
    // go1plane (ONE robot per env: a pair is exactly the lane population of a go1gate wavefront) 4096 envs 108.8 / 85.6 -- the pairing
    // pays there once the batch fills the machine (4 one-env wavefronts per SIMD), so single-robot scenes of >= 4096 envs take it
    int want = (m.A == 1 && N >= 4096) ? 2 : 1;
    if (const char* ev = getenv("MQE_ENVS_PER_WAVE")) want = atoi(ev);
    s->substeps_epw = (can && want == 2) ? 2 : 1;
  }
  s->substeps_fn = act32 ? shape_fn<true>(shape, s->substeps_epw) : shape_fn<false>(shape, s->substeps_epw);
  {   // the dynamic LDS of each launch as it is actually requested: one env's layout for k_simulate, substeps_epw of them for k_substeps
    const size_t lds_sub = s->phys_lds_bytes * (size_t)s->substeps_epw;
    if (s->phys_lds_bytes > 48 * 1024 && hipFuncSetAttribute((const void*)k_simulate, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->phys_lds_bytes) != hipSuccess)
      return fail(-4, "cannot raise dynamic LDS limit");
    if (lds_sub > 48 * 1024 && hipFuncSetAttribute((const void*)s->substeps_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sub) != hipSuccess)
      return fail(-4, "cannot raise dynamic LDS limit");
    if (timed_fn && lds_sub > 48 * 1024 && hipFuncSetAttribute((const void*)timed_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sub) != hipSuccess)
      return fail(-4, "cannot raise dynamic LDS limit");
  }
  if (phase_timed && s->substeps_epw == 1) { s->substeps_fn = timed_fn; s->phase_timed = true; }
  s->fuse_substeps = getenv("MQE_NO_FUSE_SUBSTEPS") == nullptr && d->decimation <= 4;
  // the run-time terrain curriculum needs its snapshot launch between the physics and the resets: no epilogue fusion there
  // ... and only the kernels of the small class carry the epilogue (ShapeClass<TP>::small: go1gate, go1plane, the two-robot tasks with one more object)
  {
    const bool robots_only = shape == SH_A2 || shape == SH_A1;
    // ... and, since every DevState pointer has a register pair of its own (own_state), the other kernels of the 16-envs-per-CU class: no spills
    const bool few = shape == SH_A2_NOPAD || shape == SH_A2_LINK || shape == SH_A2_NPC_FEW || shape == SH_A2_BOX_FEW || shape == SH_A2_STATIC_FEW;
    // ... and the flock and 2-vs-2 shapes (one round at 2 wavefronts per SIMD, registers to spare: go1sheep-hard +3.2 %, go1football-2vs2
    // +1.5 %).  NOT go1football-defender, whose 4096 envs run in two rounds: both pay the epilogue's ~10 us chain, -1 % (MQE_FUSE_POST_ALL=1 tries it)
    const bool flock = shape == SH_A2_NPC || shape == SH_A4_NPC;
    const bool defender = shape == SH_A3_NPC_ROW;
    s->fuse_post = s->fuse_substeps && getenv("MQE_NO_FUSE_POST") == nullptr && !m.curriculum &&
                   (robots_only || (getenv("MQE_FUSE_POST_ROBOTS_ONLY") == nullptr &&
                                    (few || (getenv("MQE_FUSE_POST_SMALL_ONLY") == nullptr && (flock || (getenv("MQE_FUSE_POST_ALL") != nullptr && defender))))));
  }
  if (s->fuse_post) {
    // the epilogue stages its observation / last-action / NPC rows and the env's actions in the link-record area (k_substeps: `sb`):
    // EPW * AMP * (MQE_OBS_BAG + 24) + EPW * (P * 13 rounded up to 4) + EPW * 12 A floats from L.body on.  With one env per wavefront they must
    // end inside the env's own layout; with two, before the SECOND env's root rows (which post_body still reads).  A layout that does
    // not leave that room (another stride, a larger bag, more NPC rows) keeps the separate launch instead of overwriting live state.
    const int epw = s->substeps_epw, amp = (A == 1 || A == 2) ? 2 : MQE_MAX_AGENTS;
    const int need = post_staging_floats(epw, amp, 12 * A, P);
    const int room = (epw == 1 ? L.total : L.total + L.root) - L.body;
    if (need > room) {
      if (getenv("MQE_VERBOSE")) fprintf(stderr, "mqe: post-physics epilogue needs %d floats of staging, the layout has %d: separate launch\n", need, room);
      s->fuse_post = false;
    }
  }
  // Debug / experiment switches are read HERE, once per handle, never on the launch path; MQE_VERBOSE lists the ones in effect.
  if (const char* sp = getenv("MQE_DEBUG_STOP_PHASE")) {
    // per-phase counter runs: the wavefront leaves k_simulate_a2 after that phase tap WITHOUT writing the state back, so the
    // unfused path stops advancing.  Only the two-robot, no-object scene has that kernel: refuse everywhere else.
    if (!s->a2_scene) { return fail(-4, "MQE_DEBUG_STOP_PHASE applies to two-robot scenes without objects only (k_simulate_a2)"); }
    s->dbg_stop_phase = atoi(sp);
  }
  if (getenv("MQE_VERBOSE")) {
    const char* names[] = {"MQE_LANE_SWEEP", "MQE_PHYS_LDS_PAD", "MQE_NO_FUSE_SUBSTEPS", "MQE_GEMM_SPLIT", "MQE_NO_FUSED_TAIL", "MQE_DEBUG_STOP_PHASE", "MQE_ENVS_PER_WAVE", "MQE_ACT_F32"};
    fprintf(stderr, "mqe: k_substeps runs %d env(s) per wavefront\n", s->substeps_epw);
    for (const char* n : names)
      if (const char* v = getenv(n)) fprintf(stderr, "mqe: override in effect: %s=%s\n", n, v);
  }
#define UP(dst, src, n) if (upload(s, &(dst), (src), (n))) { return fail(-5, "device upload failed"); }
  UP(m.wall_sdf, d->wall_sdf, (size_t)d->sdf_nx * d->sdf_ny);
  UP(m.ground_height, d->ground_height, (size_t)d->sdf_nx * d->sdf_ny);
  UP(m.wall_top, d->wall_top, (size_t)d->sdf_nx * d->sdf_ny);
  if (d->edge_contacts & ~15) return fail(-6, "edge_contacts: bits 1 (wall edges), 2 (capsule axes against the scene's boxes), 4 (box edges against box primitives) and 8 (manifold reduction to the deepest contacts)");
  m.edge_mask = d->edge_contacts & 15;
  m.wall_corner = nullptr;
  if ((m.edge_mask & 1) && d->wall_corner) { UP(m.wall_corner, d->wall_corner, (size_t)d->sdf_nx * d->sdf_ny * 2); }
  for (int q = 0; q < MQE_MAX_PRIMS; q++) {       // feature points on a capsule's axis strictly between its ends (the thigh's middle): an edge contact next to one would duplicate it
    m.prim_feat_t[q][0] = m.prim_feat_t[q][1] = -9.0f;
    if (q >= d->robot.n_prims || d->robot.prim_type[q] != MQE_PRIM_CAPSULE) continue;
    const float* ax = d->robot.prim_axis[q];
    const float aa = ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2];
    int k = 0;
    for (int f = 0; f < d->robot.n_spheres && aa > 0.0f; f++)
      if (d->robot.sphere_prim[f] == q) {
        const float df[3] = {d->robot.sphere_center[f][0] - d->robot.prim_center[q][0], d->robot.sphere_center[f][1] - d->robot.prim_center[q][1], d->robot.sphere_center[f][2] - d->robot.prim_center[q][2]};
        const float tf = 0.5f + 0.5f * (df[0] * ax[0] + df[1] * ax[1] + df[2] * ax[2]) / aa;
        if (tf > -0.05f && tf < 1.05f && k < 2) m.prim_feat_t[q][k++] = tf;       // (the ends themselves are outside the 5 .. 95 % window anyway)
      }
  }
  {
    const float soft = d->soft_dof_pos_limit > 0.0f ? d->soft_dof_pos_limit : 1.0f;
    for (int j = 0; j < MQE_NDOF; j++) {          // legged_robot.py:317-321
      const float mid = (d->robot.dof_lower[j] + d->robot.dof_upper[j]) / 2, r = d->robot.dof_upper[j] - d->robot.dof_lower[j];
      m.soft_lo[j] = mid - 0.5f * r * soft; m.soft_hi[j] = mid + 0.5f * r * soft;
    }
  }
  UP(m.env_origins, d->env_origins, (size_t)N * 3);
  UP(m.agent_origins, d->agent_origins, (size_t)N * A * 3);
  m.curriculum = d->terrain_curriculum ? 1 : 0; m.terrain_rows = d->terrain_num_rows; m.terrain_cols = d->terrain_num_cols; m.terrain_env_length = d->terrain_env_length;
  m.terrain_origins = nullptr; m.terrain_types = nullptr;
  if (m.curriculum) {
    if (d->env_id_offset != 0) return fail(-6, "terrain curriculum: not defined for a shard of a larger batch (row e of the agents' root states belongs to another shard's env)");
    if (!d->terrain_origins || !d->terrain_levels || !d->terrain_types || d->terrain_num_rows < 1 || d->terrain_num_cols < 1) return fail(-6, "terrain curriculum: origin table / levels / types missing");
    UP(m.terrain_origins, d->terrain_origins, (size_t)d->terrain_num_rows * d->terrain_num_cols * 3);
    const float* tt = nullptr;
    UP(tt, reinterpret_cast<const float*>(d->terrain_types), (size_t)N);
    m.terrain_types = reinterpret_cast<const int32_t*>(tt);
  }
  UP(m.base_init, d->base_init_state, (size_t)A * 13);
  UP(m.npc_init, d->npc_init_state, (size_t)P * 13);
  UP(m.gate_pos, d->gate_pos, (size_t)N * 2);
  m.actuator.n_layers = d->actuator.n_layers;
  memcpy(m.actuator.dims, d->actuator.dims, sizeof m.actuator.dims);
  if (d->actuator.n_layers != 3 || d->actuator.dims[0] != 6 || d->actuator.dims[1] != 32 || d->actuator.dims[2] != 32 || d->actuator.dims[3] != 1)
    return fail(-6, "actuator network must be 6-32-32-1 (unitree_go1.pt)");
  for (int l = 0; l < 3; l++) {
    UP(m.actuator.W[l], d->actuator.W[l], (size_t)d->actuator.dims[l] * d->actuator.dims[l + 1]);
    UP(m.actuator.b[l], d->actuator.b[l], (size_t)d->actuator.dims[l + 1]);
  }
  {   // the same weights in k_substeps' fragment order (DevModel::act_frag): [19][64]
    std::vector<float> fr((size_t)19 * 64);
    for (int lane = 0; lane < 64; lane++) {
      const int j32 = lane & 31, h = lane >> 5;
      for (int r = 0; r < 16; r++) fr[(size_t)r * 64 + lane] = d->actuator.W[1][j32 * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
      for (int s2 = 0; s2 < 3; s2++) fr[(size_t)(16 + s2) * 64 + lane] = d->actuator.W[0][j32 * 6 + 2 * s2 + h];
    }
    UP(m.act_frag, fr.data(), fr.size());
    // layer 2 for the f16 matrix cores: two planes of 2^14 W1 in k_substeps' B-operand order (DevModel::act_frag16)
    std::vector<uint16_t> f16((size_t)2 * 2 * 64 * 8);
    float w1max = 0.0f;
    for (int i = 0; i < 32 * 32; i++) w1max = std::max(w1max, std::fabs(d->actuator.W[1][i]));
    for (int s2 = 0; s2 < 2; s2++)
      for (int lane = 0; lane < 64; lane++)
        for (int i8 = 0; i8 < 8; i8++) {
          const int row = lane & 31, g = lane >> 5, u = (i8 & 3) + 16 * s2 + 8 * (i8 >> 2) + 4 * g;
          uint16_t hh, ll;
          split2(d->actuator.W[1][row * 32 + u], 16384.0f, hh, ll);
          f16[((size_t)(s2 * 2 + 0) * 64 + lane) * 8 + i8] = hh;
          f16[((size_t)(s2 * 2 + 1) * 64 + lane) * 8 + i8] = ll;
        }
    const float* t16;
    UP(t16, reinterpret_cast<const float*>(f16.data()), f16.size() / 2);
    m.act_frag16 = reinterpret_cast<const uint16_t*>(t16);
    (void)w1max;
  }
  // ---- policy network: fused layer 0 over the ring-buffer history ------------------------------------------------
  const mqe_mlp& ad = d->adaptation; const mqe_mlp& bd = d->body;
  if (ad.dims[0] != 2100 || bd.dims[0] != 2102 || bd.dims[bd.n_layers] != 12 || ad.dims[ad.n_layers] != 2)
    return fail(-6, "policy I/O must be adaptation 2100->2, body 2102->12 (go1.py:395,404,29)");
  s->ada_h0 = ad.dims[1]; s->body_h0 = bd.dims[1];
  if (s->ada_h0 % GB_N || s->body_h0 % GB_N) return fail(-6, "first hidden sizes must be multiples of 64");
  std::vector<int> krow(2100);
  for (int k = 0; k < 2100; k++) krow[k] = (k / 70) * MQE_FRAME + (k % 70);
  // the split-f16 operand has its own K order: compact frames of MQE_H2_FRAME columns (mqe_common.hpp): the twelve constant gait
  // parameters of a frame folded onto its presence-flag column, its last_two_locomotion_action onto the previous frame's
  // last_locomotion_action (the oldest frame's: component j onto the carrier column of frame MQE_H2_CARRIER0 + j)
  std::vector<int> k2row(2100);
  std::vector<float> k2scale(2100);
  for (int k = 0; k < 2100; k++) {
    const int f = k / 70, c = k % 70, cc = h2_col(c);
    if (cc == -2) k2row[k] = f > 0 ? (f - 1) * MQE_H2_FRAME + h2_col(c - 12) : (MQE_H2_CARRIER0 + c - 54) * MQE_H2_FRAME + MQE_H2_CARRIER_COL;
    else k2row[k] = f * MQE_H2_FRAME + (cc >= 0 ? cc : MQE_H2_FLAG_COL);
    k2scale[k] = cc == -1 ? d->command_obs[c] : 1.0f;
  }
  if (make_layer(s, &s->l0, MQE_HIST * MQE_FRAME, s->ada_h0 + s->body_h0)) return fail(-5, "alloc");
  s->l0.Kpad3 = rup(MQE_HIST * MQE_H2_FRAME, H2_KMULT);
  s->l0.hW.assign((size_t)s->l0.Npad * s->l0.Kpad3, 0.0f);
  if (fill_layer(&s->l0, 0, ad.W[0], ad.b[0], s->ada_h0, 2100, 2100, &krow, &k2row, &k2scale)) return fail(-5, "upload");
  if (fill_layer(&s->l0, s->ada_h0, bd.W[0], bd.b[0], s->body_h0, 2100, 2102, &krow, &k2row, &k2scale)) return fail(-5, "upload");
  {
    std::vector<float> w0(s->body_h0), w1(s->body_h0);
    for (int o = 0; o < s->body_h0; o++) { w0[o] = bd.W[0][(size_t)o * 2102 + 2100]; w1[o] = bd.W[0][(size_t)o * 2102 + 2101]; }
    const float* t;
    UP(t, w0.data(), w0.size()); s->w_lat0 = (float*)t;
    UP(t, w1.data(), w1.size()); s->w_lat1 = (float*)t;
  }
  // Layer 0 runs on the f16 matrix cores with two-plane split operands (f32-class accuracy, k_gemm_h2) whenever
  // its width tiles by 192; MQE_GEMM_SPLIT=0 selects the exact-f32 MFMA kernel instead (0.255 ms vs 0.080 ms at 8192 rows).
  {
    // Which kernel for layer 0?  k_gemm_h2 owns a whole CU per 128 x 192 tile: its time is a staircase in R (80 us per
    // started round of 256 tiles), the exact-f32 kernel scales linearly (255 us at R = 8192).  Pick the faster one for this
    // batch unless MQE_GEMM_SPLIT forces a choice.
    const char* f = getenv("MQE_GEMM_SPLIT");
    int ft = 0, ht = 0;
    const int ntn0 = s->l0.Npad / H2_N > 0 ? s->l0.Npad / H2_N : 1;
    s->gemm_half = getenv("MQE_GEMM_HALF") == nullptr || atoi(getenv("MQE_GEMM_HALF")) != 0;      // read per handle (tests build both forms in one process)
    h2_tiling(R, ntn0, s->gemm_half, &ft, &ht);
    const double rounds = std::ceil(ft * (double)ntn0 / 256.0) + (ht ? 0.6 : 0.0);       // a (partial) round of half tiles: ~0.6 of a full one
    const bool faster = rounds * 80.0 < 255.0 * R / 8192.0;
    s->gemm_split = s->l0.Npad % H2_N == 0 && (f ? atoi(f) != 0 : faster) && !s->cmd_general;      // the compact operand folds entries 6-17 into its weights
  }
  if (s->gemm_split) {
    if (finalize_layer(s, &s->l0)) return fail(-5, "upload");
    if (hipFuncSetAttribute((const void*)k_gemm_h2, hipFuncAttributeMaxDynamicSharedMemorySize, H2_LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_gemm_h2_mix, hipFuncAttributeMaxDynamicSharedMemorySize, H2_LDS_BYTES) != hipSuccess)
      return fail(-4, "cannot raise dynamic LDS limit");
  }
  for (int l = 1; l < ad.n_layers; l++) {
    GemmLayer g;
    if (make_layer(s, &g, ad.dims[l], ad.dims[l + 1]) || fill_layer(&g, 0, ad.W[l], ad.b[l], ad.dims[l + 1], ad.dims[l], ad.dims[l])) return fail(-5, "upload");
    s->ada_rest.push_back(g);
  }
  for (int l = 1; l < bd.n_layers; l++) {
    GemmLayer g;
    if (make_layer(s, &g, bd.dims[l], bd.dims[l + 1]) || fill_layer(&g, 0, bd.W[l], bd.b[l], bd.dims[l + 1], bd.dims[l], bd.dims[l])) return fail(-5, "upload");
    s->body_rest.push_back(g);
  }
  s->tail_fused = getenv("MQE_NO_FUSED_TAIL") == nullptr && ad.n_layers == 3 && bd.n_layers == 4 && ad.dims[1] == 256 && ad.dims[2] == 128 &&
                  bd.dims[1] == 512 && bd.dims[2] == 256 && bd.dims[3] == 128;
  if (s->tail_fused) {
    if (hipFuncSetAttribute((const void*)k_policy_tail, hipFuncAttributeMaxDynamicSharedMemorySize, TL_LDS_BYTES) != hipSuccess)
      return fail(-4, "cannot raise dynamic LDS limit");
    for (auto& g : s->ada_rest) if (finalize_frag(s, &g)) return fail(-5, "upload");
    for (auto& g : s->body_rest) if (finalize_frag(s, &g)) return fail(-5, "upload");
  }
  if (s->tail_fused && getenv("MQE_TAIL_TIMES")) { if (dalloc(s, &s->tail_times, (size_t)16 * ((R + TL_ROWS - 1) / TL_ROWS))) return fail(-5, "alloc"); }
  int maxw = 64;
  for (auto& g : s->ada_rest) maxw = std::max(maxw, g.Npad);
  for (auto& g : s->body_rest) maxw = std::max(maxw, g.Npad);
  s->ldP1 = s->l0.Npad; s->ldbuf = maxw; s->ldlat = 64; s->ldact = 64;
#define DA(p, n) if (dalloc(s, &(p), (n))) return fail(-5, "device alloc failed");
  DA(s->P1, (size_t)R * s->ldP1); DA(s->bufA, (size_t)R * s->ldbuf); DA(s->bufB, (size_t)R * s->ldbuf);
  DA(s->lat, (size_t)R * s->ldlat); DA(s->act_out, (size_t)R * s->ldact);
  // ---- state ---------------------------------------------------------------------------------------------------------
  DevState& st = s->st;
  s->reg_state = true;
  DA(st.root, (size_t)N * (A + P) * 13); DA(st.dof, (size_t)N * s->ND * 2); DA(st.cf, (size_t)N * s->NBR * 3);
  DA(st.torques, (size_t)N * 12 * A); DA(st.actions, (size_t)N * 12 * A); DA(st.last_actions, (size_t)N * 12 * A);
  DA(st.loco_obs, (size_t)R * MQE_FRAME); DA(st.hist, (size_t)R * MQE_HIST * MQE_FRAME);
  st.hist2 = nullptr;
  st.hist_irr = nullptr;
  st.wave_times = nullptr;
  if (getenv("MQE_WAVE_TIMES") || getenv("MQE_PHASE_TIMES")) { DA(st.wave_times, (size_t)(4 + 64 + 16) * N); }      // [N][4] entry / exit / ids, then [N][4 substeps][16 taps]
  if (s->gemm_split) { DA(st.hist2, (size_t)2 * R * MQE_HIST * MQE_H2_FRAME); DA(st.hist_irr, (size_t)R); }
  DA(st.last_loco, (size_t)R * 12); DA(st.last_two_loco, (size_t)R * 12); DA(st.act_hist, (size_t)4 * R * 12);
  DA(st.gait, R); DA(st.clock, (size_t)R * 4); DA(st.blv, (size_t)R * 3); DA(st.bav, (size_t)R * 3); DA(st.pg, (size_t)R * 3);
  DA(st.bquat, (size_t)R * 4); DA(st.obs_bag, (size_t)R * MQE_OBS_BAG); DA(st.wobs, (size_t)N * s->Aw * s->D + (size_t)N * s->Aw + (N + 3) / 4); st.wrew = st.wobs + (size_t)N * s->Aw * s->D; st.wdone = (uint8_t*)(st.wrew + (size_t)N * s->Aw);   // one buffer: obs | reward | done (N bytes)
  DA(st.rsum, (size_t)N * MQE_MAX_REWARD_TERMS); DA(st.sheep_avg, (size_t)N * 2); DA(st.sheep_var, N);
  DA(st.sub_dof_vel, (size_t)N * 4 * 12 * A); DA(st.sub_exceed, (size_t)N * 4 * 12 * A); DA(st.overflow, 2 * (size_t)N);      /* [0, N): MQE_T_CONTACT_OVERFLOW, [N, 2 N): MQE_T_CONTACT_REDUCED */
  DA(st.sub_tau, (size_t)N * 4 * 12 * A); DA(st.npc_noise, (size_t)N * (P ? P : 1) * 3);
  DA(st.w_last, (size_t)N * MQE_MAX_AGENTS); DA(st.w_last2, (size_t)N * 2); DA(st.cmd, (size_t)R * 3);
  DA(st.ep_len, N); DA(st.reset_count, N); DA(st.last_dof_vel, (size_t)R * 12);
  DA(st.env_origins_live, (size_t)N * 3); DA(st.curr_xy, (size_t)N * 2); DA(st.terrain_levels, N); DA(st.npc_pre, (size_t)N * (P ? P : 1) * 13);
  if (hipMemcpy(st.env_origins_live, d->env_origins, (size_t)N * 12, hipMemcpyHostToDevice) != hipSuccess) return fail(-5, "upload");
  if (d->terrain_curriculum && hipMemcpy(st.terrain_levels, d->terrain_levels, (size_t)N * 4, hipMemcpyHostToDevice) != hipSuccess) return fail(-5, "upload");
  // domain parameters (include/mqe_hip.h): drawn once, keyed by the global env id so that a sharded run sees the same robots
  {
    std::vector<float> dp((size_t)R * 8, 0.0f);
    const uint32_t seed = (uint32_t)d->seed;
    for (int e = 0; e < N; e++) {
      const uint32_t genv = (uint32_t)(e + d->env_id_offset);
      float mu = d->friction;
      if (d->rand_friction) {            // legged_robot.py:283-294: 64 buckets, one per env
        const uint32_t bucket = mqe_hash(seed, genv, MQE_RNG_CREATE, 0) % 64u;
        mu = d->friction_lo + (d->friction_hi - d->friction_lo) * mqe_u01(seed, bucket, MQE_RNG_CREATE + 1u, 0);
      }
      for (int a = 0; a < A; a++) {
        float* p = dp.data() + ((size_t)e * A + a) * 8;
        p[0] = mu;
        if (d->rand_base_mass) p[1] = d->added_mass_lo + (d->added_mass_hi - d->added_mass_lo) * mqe_u01(seed, genv, MQE_RNG_CREATE, 16u + (uint32_t)a);
        if (d->rand_com)
          for (int k = 0; k < 3; k++) p[2 + k] = d->com_lo[k] + (d->com_hi[k] - d->com_lo[k]) * mqe_u01(seed, genv, MQE_RNG_CREATE, 32u + (uint32_t)(a * 3 + k));
      }
    }
    DA(st.dparams, (size_t)R * 8);
    if (hipMemcpy(st.dparams, dp.data(), dp.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return fail(-5, "upload");
  }
  st.lag_buf = nullptr;
  if (s->hm.lag_steps > 0) { DA(st.lag_buf, (size_t)(s->hm.lag_steps + 1) * R * 12); }
  DA(st.reset_buf, N); DA(st.collide_buf, N); DA(st.time_out, N); DA(st.r_term, N); DA(st.p_term, N); DA(st.zh_term, N);
  DA(st.w_have_last, N); DA(st.w_delayed_reset, N);
  s->reg_state = false;
  {
    // initial values the reference's buffers start from (base_task.py:77-84, legged_robot.py:567-622)
    std::vector<float> h((size_t)N * (A + P) * 13, 0.0f);
    for (size_t i = 0; i < (size_t)N * (A + P); i++) h[i * 13 + 6] = 1.0f;
    HIPCHK(hipMemcpy(st.root, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> lo((size_t)R * MQE_FRAME, 0.0f), pg((size_t)R * 3, 0.0f), bq((size_t)R * 4, 0.0f);
    for (int i = 0; i < R; i++) { memcpy(&lo[(size_t)i * MQE_FRAME], d->command_obs, 70 * 4); pg[i * 3 + 2] = -1.0f; bq[i * 4 + 3] = 1.0f; }
    HIPCHK(hipMemcpy(st.loco_obs, lo.data(), lo.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(st.pg, pg.data(), pg.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(st.bquat, bq.data(), bq.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(st.reset_buf, 1, N));
  }
  DevModel* dmp;
  if (dalloc(s, &dmp, 1)) return fail(-5, "alloc");
  mqe_fill_hot(m);
  HIPCHK(hipMemcpy(dmp, &m, sizeof m, hipMemcpyHostToDevice));
  s->dm = dmp;
  void** t = s->tens;
  t[MQE_T_ROOT_STATE] = st.root; t[MQE_T_DOF_STATE] = st.dof; t[MQE_T_CONTACT_FORCE] = st.cf; t[MQE_T_TORQUES] = st.torques;
  t[MQE_T_ACTIONS] = st.actions; t[MQE_T_LAST_ACTIONS] = st.last_actions; t[MQE_T_LOCOMOTION_OBS] = st.loco_obs;
  t[MQE_T_HISTORY] = st.hist; t[MQE_T_LAST_LOCO_ACTION] = st.last_loco; t[MQE_T_LAST_TWO_LOCO_ACTION] = st.last_two_loco;
  t[MQE_T_ACT_HIST] = st.act_hist; t[MQE_T_GAIT_INDICES] = st.gait; t[MQE_T_CLOCK_INPUTS] = st.clock;
  t[MQE_T_BASE_LIN_VEL] = st.blv; t[MQE_T_BASE_ANG_VEL] = st.bav; t[MQE_T_PROJECTED_GRAVITY] = st.pg; t[MQE_T_BASE_QUAT] = st.bquat;
  t[MQE_T_EPISODE_LENGTH] = st.ep_len; t[MQE_T_RESET_BUF] = st.reset_buf; t[MQE_T_COLLIDE_BUF] = st.collide_buf;
  t[MQE_T_TIME_OUT_BUF] = st.time_out; t[MQE_T_R_TERM] = st.r_term; t[MQE_T_P_TERM] = st.p_term; t[MQE_T_Z_HIGH_TERM] = st.zh_term;
  t[MQE_T_OBS_BAG] = st.obs_bag; t[MQE_T_WRAPPER_OBS] = st.wobs; t[MQE_T_WRAPPER_REWARD] = st.wrew; t[MQE_T_REWARD_SUMS] = st.rsum;
  t[MQE_T_SHEEP_POS_AVG] = st.sheep_avg; t[MQE_T_SHEEP_POS_VAR] = st.sheep_var; t[MQE_T_RESET_COUNT] = st.reset_count;
  t[MQE_T_SUBSTEP_TORQUES] = st.sub_tau; t[MQE_T_NPC_NOISE] = st.npc_noise; t[MQE_T_WRAPPER_PACKED] = st.wobs;
  t[MQE_T_DOMAIN_PARAMS] = st.dparams;
  t[MQE_T_SUBSTEP_DOF_VEL] = st.sub_dof_vel; t[MQE_T_SUBSTEP_EXCEED_DOF_POS_LIMITS] = st.sub_exceed; t[MQE_T_CONTACT_OVERFLOW] = st.overflow;
  t[MQE_T_ENV_ORIGINS] = st.env_origins_live; t[MQE_T_TERRAIN_LEVELS] = st.terrain_levels;
  t[MQE_T_CONTACT_REDUCED] = st.overflow + s->N;
  HIPCHK(hipDeviceSynchronize());
  guard.s = nullptr;
  *out = s;
  return 0;
}

extern "C" int mqe_sim_destroy(mqe_sim* s) {
  if (!s) return 0;
  hipDeviceSynchronize();
  for (void* p : s->allocs) hipFree(p);
  for (int k = 0; k < PROF_N; k++) {
    for (auto e : s->ev0[k]) hipEventDestroy(e);
    for (auto e : s->ev1[k]) hipEventDestroy(e);
  }
  delete s;
  return 0;
}

extern "C" int mqe_sim_tensor(mqe_sim* s, int kind, mqe_tensor_view* v) {
  if (!s || !v || kind < 0 || kind >= MQE_T_COUNT) return fail(-1, "bad tensor kind");
  memset(v, 0, sizeof *v);
  v->ptr = s->tens[kind];
  const int N = s->N, A = s->A, P = s->P, R = s->R;
#define SH(nd, a, b, c, e, dt_) do { v->ndim = nd; v->shape[0] = a; v->shape[1] = b; v->shape[2] = c; v->shape[3] = e; v->dtype = dt_; } while (0)
  switch (kind) {
    case MQE_T_ROOT_STATE: SH(3, N, A + P, 13, 0, 0); break;
    case MQE_T_DOF_STATE: SH(3, N, s->ND, 2, 0, 0); break;
    case MQE_T_CONTACT_FORCE: SH(3, N, s->NBR, 3, 0, 0); break;
    case MQE_T_TORQUES: case MQE_T_ACTIONS: case MQE_T_LAST_ACTIONS: SH(2, N, 12 * A, 0, 0, 0); break;
    case MQE_T_LOCOMOTION_OBS: SH(2, R, MQE_FRAME, 0, 0, 0); break;
    case MQE_T_HISTORY: SH(3, R, MQE_HIST, MQE_FRAME, 0, 0); break;
    case MQE_T_LAST_LOCO_ACTION: case MQE_T_LAST_TWO_LOCO_ACTION: SH(2, R, 12, 0, 0, 0); break;
    case MQE_T_ACT_HIST: SH(3, 4, R, 12, 0, 0); break;
    case MQE_T_GAIT_INDICES: SH(1, R, 0, 0, 0, 0); break;
    case MQE_T_CLOCK_INPUTS: case MQE_T_BASE_QUAT: SH(2, R, 4, 0, 0, 0); break;
    case MQE_T_BASE_LIN_VEL: case MQE_T_BASE_ANG_VEL: case MQE_T_PROJECTED_GRAVITY: SH(2, R, 3, 0, 0, 0); break;
    case MQE_T_EPISODE_LENGTH: case MQE_T_RESET_COUNT: SH(1, N, 0, 0, 0, 1); break;
    case MQE_T_RESET_BUF: case MQE_T_COLLIDE_BUF: case MQE_T_TIME_OUT_BUF: case MQE_T_R_TERM: case MQE_T_P_TERM:
    case MQE_T_Z_HIGH_TERM: SH(1, N, 0, 0, 0, 2); break;
    case MQE_T_OBS_BAG: SH(2, R, MQE_OBS_BAG, 0, 0, 0); break;
    case MQE_T_WRAPPER_OBS: SH(3, N, s->Aw, s->D, 0, 0); break;
    case MQE_T_WRAPPER_REWARD: SH(2, N, s->Aw, 0, 0, 0); break;
    case MQE_T_REWARD_SUMS: SH(2, N, MQE_MAX_REWARD_TERMS, 0, 0, 0); break;
    case MQE_T_SHEEP_POS_AVG: SH(2, N, 2, 0, 0, 0); break;
    case MQE_T_SHEEP_POS_VAR: SH(1, N, 0, 0, 0, 0); break;
    case MQE_T_SUBSTEP_TORQUES: case MQE_T_SUBSTEP_DOF_VEL: SH(3, N, 4, 12 * A, 0, 0); break;
    case MQE_T_SUBSTEP_EXCEED_DOF_POS_LIMITS: SH(3, N, 4, 12 * A, 0, 2); break;
    case MQE_T_CONTACT_OVERFLOW: case MQE_T_TERRAIN_LEVELS: case MQE_T_CONTACT_REDUCED: SH(1, N, 0, 0, 0, 1); break;
    case MQE_T_ENV_ORIGINS: SH(2, N, 3, 0, 0, 0); break;
    case MQE_T_NPC_NOISE: SH(3, N, P, 3, 0, 0); break;
    case MQE_T_WRAPPER_PACKED: SH(1, N * s->Aw * s->D + N * s->Aw + (N + 3) / 4, 0, 0, 0, 0); break;
    case MQE_T_DOMAIN_PARAMS: SH(2, s->R, 8, 0, 0, 0); break;
  }
  return 0;
}

// ---- profiling helpers ---------------------------------------------------------------------------------------------------
// which step of every `every` is bracketed with events: the third, not the first -- the first step after a synchronisation starts on an idle GPU, and the
// ~35 us its twelve event records add would sit on the exposed launch latency (bench.py mirrors this rule to count the bracketed steps)
static inline long prof_phase(int every) { return every > 2 ? 2 : 0; }
struct ProfScope {
  mqe_sim* s; int k; hipStream_t q; hipEvent_t e1;
  ProfScope(mqe_sim* s_, int k_, hipStream_t q_) : s(s_), k(k_), q(q_), e1(nullptr) {
    if (!s->prof_now) return;
    hipEvent_t e0;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, q);
    s->ev0[k].push_back(e0); s->ev1[k].push_back(e1);
  }
  ~ProfScope() { if (e1) hipEventRecord(e1, q); }
};

// on = 0: off; on = k > 0: the kernel classes of every k-th fused step are bracketed with HIP events on the launch stream (an
// event pair costs ~4 us of GPU timeline, 5 classes = 7 % of a 0.5 ms step: sampling keeps the measurement out of the result)
extern "C" int mqe_profile_enable(mqe_sim* s, int on) {
  if (!s) return fail(-1, "null engine handle");
  s->prof = on != 0;
  s->prof_every = on > 0 ? on : 1;
  s->prof_step = 0;
  s->prof_now = s->prof;
  return 0;
}
extern "C" int mqe_profile_read(mqe_sim* s, float* ms, int n, int* n_launches) {
  if (!s) return fail(-1, "null engine handle");
  HIPCHK(hipDeviceSynchronize());
  for (int k = 0; k < PROF_N; k++) {
    for (size_t i = 0; i < s->ev0[k].size(); i++) {
      float t = 0;
      hipEventElapsedTime(&t, s->ev0[k][i], s->ev1[k][i]);
      s->prof_ms[k] += t; s->prof_cnt[k] += 1;
      hipEventDestroy(s->ev0[k][i]); hipEventDestroy(s->ev1[k][i]);
    }
    s->ev0[k].clear(); s->ev1[k].clear();
  }
  for (int k = 0; k < PROF_N && k < n; k++) ms[k] = s->prof_ms[k];
  for (int k = 0; k < PROF_N && PROF_N + k < n; k++) ms[PROF_N + k] = (float)s->prof_cnt[k];
  if (n_launches) *n_launches = s->prof_cnt[PROF_GEMM_L0];
  memset(s->prof_ms, 0, sizeof s->prof_ms);
  memset(s->prof_cnt, 0, sizeof s->prof_cnt);
  return 0;
}

// ---- launches ---------------------------------------------------------------------------------------------------------------
static void launch_gemm(hipStream_t q, const float* A, int lda, int rot4, int ring4, const GemmLayer& L, float* C, int ldc, int M, int act_cols) {
  GemmArgs g;
  g.A = A; g.lda = lda; g.a_rot4 = rot4; g.a_ring4 = ring4; g.Wt = L.Wt; g.ldw = L.Npad; g.bias = L.bias;
  g.C = C; g.ldc = ldc; g.M = M; g.N = L.Npad; g.K = L.Kpad; g.act_cols = act_cols;
  int grid = ((M + GB_M - 1) / GB_M) * (L.Npad / GB_N);
  hipLaunchKernelGGL(k_gemm_f32, dim3(grid), dim3(256), 0, q, g);
}

static void launch_gemm2(hipStream_t q, const uint16_t* A, int lda, int rot8, int ring8, const GemmLayer& L,
                         float* C, int ldc, int M, int act_cols, const unsigned* irr = nullptr, const float* ring = nullptr, int ring_pos = 0, bool use_half = true, long long* times = nullptr, int times_blocks = 0) {
  Gemm2Args g;
  g.times = times; g.times_blocks = times_blocks;
  g.irr = irr; g.ring = ring; g.ring_pos = ring_pos; g.Wt32 = L.Wt; g.ldwt = L.Npad;
  g.A = A; g.lda = lda; g.a_rot8 = rot8; g.a_ring8 = ring8;
  g.W = L.W2; g.ldw = 2 * L.Kpad3; g.bias = L.bias;
  g.C = C; g.ldc = ldc; g.M = M; g.N = L.Npad; g.K = L.Kpad3; g.act_cols = act_cols;
  g.descale = 1.0f / (MQE_H2_ASCALE * L.wscale);
  int full_tiles, half_tiles;
  h2_tiling(M, L.Npad / H2_N, use_half, &full_tiles, &half_tiles);
  g.full_blocks = full_tiles * (L.Npad / H2_N); g.full_rows = full_tiles * H2_M;
  const int grid = (full_tiles + half_tiles) * (L.Npad / H2_N);
  if (half_tiles == 0) hipLaunchKernelGGL(k_gemm_h2, dim3(grid), dim3(H2_THREADS), H2_LDS_BYTES, q, g);
  else hipLaunchKernelGGL(k_gemm_h2_mix, dim3(grid), dim3(H2_THREADS), H2_LDS_BYTES, q, g);
}

static int policy_tail(mqe_sim* s, hipStream_t q);
static void policy_head(mqe_sim* s, const float* command, hipStream_t q, const float* wrapper_actions);
static int policy_step(mqe_sim* s, const float* command, hipStream_t q, const float* wrapper_actions = nullptr) {
  policy_head(s, command, q, wrapper_actions);
  return policy_tail(s, q);
}
// first half of the policy: the frame of this step into the history ring (wrapper head included) and layer 0 of both networks
static void policy_head(mqe_sim* s, const float* command, hipStream_t q, const float* wrapper_actions) {
  const int R = s->R;
  const int slot = s->hist_pos;
  s->hist_pos = (s->hist_pos + 1) % MQE_HIST;     // ring slot of the oldest frame
  {
    ProfScope ps(s, PROF_MISC, q);
    hipLaunchKernelGGL(k_pre_policy, dim3((R + 3) / 4), dim3(256), 0, q, s->dm, s->st, command, slot, wrapper_actions);      // one wavefront per robot
  }
  {
    ProfScope ps(s, PROF_GEMM_L0, q);
    // fused layer 0 of both networks over the ring: ELU on the adaptation columns only
    if (s->gemm_split)
      launch_gemm2(q, s->st.hist2, 2 * MQE_HIST * MQE_H2_FRAME, s->hist_pos * (MQE_H2_FRAME / 8), MQE_HIST * MQE_H2_FRAME / 8, s->l0,
                   s->P1, s->ldP1, R, s->ada_h0, s->st.hist_irr, s->st.hist, s->hist_pos, s->gemm_half, s->tail_times, (R + TL_ROWS - 1) / TL_ROWS);
    else
      launch_gemm(q, s->st.hist, MQE_HIST * MQE_FRAME, s->hist_pos * (MQE_FRAME / 4), MQE_HIST * MQE_FRAME / 4, s->l0, s->P1, s->ldP1, R, s->ada_h0);
  }
}
// second half: everything after layer 0 (latent, body MLP, post-policy registers)
static int policy_tail(mqe_sim* s, hipStream_t q) {
  const int R = s->R;
  ProfScope ps(s, PROF_GEMM_REST, q);
  if (s->tail_fused) {
    TailArgs t;
    t.P1 = s->P1; t.ldp = s->ldP1; t.ada_h0 = s->ada_h0;
    auto tl = [](const GemmLayer& L) { TailLayer r; r.W = L.Wfrag; r.bias = L.bias; r.descale = 1.0f / (TL_ASCALE * L.fscale); return r; };
    t.a1 = tl(s->ada_rest[0]); t.a2 = tl(s->ada_rest[1]);
    t.wl0 = s->w_lat0; t.wl1 = s->w_lat1;
    t.b1 = tl(s->body_rest[0]); t.b2 = tl(s->body_rest[1]); t.b3 = tl(s->body_rest[2]);
    t.lat = s->lat; t.ldl = s->ldlat; t.act = s->act_out; t.lda = s->ldact;
    t.last_loco = s->st.last_loco; t.last_two_loco = s->st.last_two_loco; t.actions = s->st.actions; t.clip_actions = s->hm.clip_actions;
    t.R = R;
    t.block0 = (int)(((long long)s->d.env_id_offset * s->A) / TL_ROWS);
    t.times = s->tail_times;
    hipLaunchKernelGGL(k_policy_tail, dim3((R + TL_ROWS - 1) / TL_ROWS), dim3(TL_THREADS), TL_LDS_BYTES, q, t);
    return 0;
  }
  // adaptation tail -> latent
  const float* x = s->P1; int ldx = s->ldP1;
  for (size_t l = 0; l < s->ada_rest.size(); l++) {
    const bool last = l + 1 == s->ada_rest.size();
    float* y = last ? s->lat : ((l & 1) ? s->bufB : s->bufA);
    int ldy = last ? s->ldlat : s->ldbuf;
    launch_gemm(q, x, ldx, 0, 0, s->ada_rest[l], y, ldy, R, last ? 0 : s->ada_rest[l].Npad);
    x = y; ldx = ldy;
  }
  {
    int n = R * s->body_h0;
    hipLaunchKernelGGL(k_body_l0_finish, dim3((n + 255) / 256), dim3(256), 0, q, s->P1, s->ldP1, s->ada_h0, s->body_h0,
                       (const float*)s->lat, s->ldlat, (const float*)s->w_lat0, (const float*)s->w_lat1, R);
  }
  x = s->P1 + s->ada_h0; ldx = s->ldP1;
  for (size_t l = 0; l < s->body_rest.size(); l++) {
    const bool last = l + 1 == s->body_rest.size();
    float* y = last ? s->act_out : ((l & 1) ? s->bufB : s->bufA);
    int ldy = last ? s->ldact : s->ldbuf;
    launch_gemm(q, x, ldx, 0, 0, s->body_rest[l], y, ldy, R, last ? 0 : s->body_rest[l].Npad);
    x = y; ldx = ldy;
  }
  int n = R * 12;
  hipLaunchKernelGGL(k_post_policy, dim3((n + 255) / 256), dim3(256), 0, q, s->dm, s->st, (const float*)s->act_out, s->ldact);
  return 0;
}

static void advance_lag(mqe_sim* s, int n) { if (s->d.lag_timesteps > 0) s->lag_pos = (s->lag_pos + n) % (s->d.lag_timesteps + 1); }
static void launch_torques(mqe_sim* s, int dec_i, hipStream_t q) {
  ProfScope ps(s, PROF_TORQUES, q);
  int n = s->R * 12;
  if (s->d.control_type == MQE_CTRL_C)
    hipLaunchKernelGGL(k_compute_torques_mfma, dim3((n + 127) / 128), dim3(256), 0, q, s->dm, s->st, dec_i, s->lag_pos);
  else
    hipLaunchKernelGGL(k_compute_torques, dim3((n + 255) / 256), dim3(256), 0, q, s->dm, s->st, dec_i, s->lag_pos);
  advance_lag(s, 1);
}
static void launch_simulate(mqe_sim* s, hipStream_t q) {
  ProfScope ps(s, PROF_SIMULATE, q);
  PhysDebug dbg = {nullptr, nullptr, nullptr, 0, nullptr, -1};
  if (s->dbg_stop_phase >= 0) {             // tools/phase_counters.py: the wavefront leaves after that phase tap (validated at creation)
    dbg.stop_after = s->dbg_stop_phase;      // a tap >= 100 never fires: the whole a2 substep, state written back (the "full" launch of the tool)
    hipLaunchKernelGGL(k_simulate_a2, dim3(s->N), dim3(64), s->phys_lds_bytes, q, s->dm, s->st, 0, s->dbg_stop_phase < 100 ? 1 : 0, dbg);
    return;
  }
  hipLaunchKernelGGL(k_simulate, dim3(s->N), dim3(64), s->phys_lds_bytes, q, s->dm, s->st, 0, 0, dbg);
}
static void launch_post(mqe_sim* s, hipStream_t q, int wrapper_level) {
  ProfScope ps(s, PROF_POST, q);
  s->n_post_steps++;                          // = common_step_counter after its increment (legged_robot.py:127)
  const int push = (s->d.push_interval > 0 && s->n_post_steps % s->d.push_interval == 0) ? (int)(s->n_post_steps / s->d.push_interval) : 0;
  if (s->hm.curriculum) hipLaunchKernelGGL(k_curriculum_snapshot, dim3((s->N + 255) / 256), dim3(256), 0, q, s->dm, s->st);   // the rows every reset of this step measures
  if (s->hm.A <= 2)
    hipLaunchKernelGGL(k_post_physics<2>, dim3((s->N + POST_EPW - 1) / POST_EPW), dim3(64), 0, q, s->dm, s->st, wrapper_level, push, s->n_post_steps);
  else
    hipLaunchKernelGGL(k_post_physics<MQE_MAX_AGENTS>, dim3((s->N + POST_EPW - 1) / POST_EPW), dim3(64), 0, q, s->dm, s->st, wrapper_level, push, s->n_post_steps);   // incl. history zeroing
}

// ---- checkpoint / resume: every buffer of DevState + the host-side counters -------------------------------------------------
struct StateHeader { uint32_t magic, abi; int32_t N, A, P, nbuf; int32_t hist_pos, n_post_steps, lag_pos, reserved; uint64_t bytes; };
static const uint32_t STATE_MAGIC = 0x5345514du;      // "MQES"
extern "C" long long mqe_state_size(mqe_sim* s) {
  if (!s) { fail(-1, "null engine handle"); return 0; }
  size_t n = sizeof(StateHeader);
  for (auto& b : s->state_bufs) n += (b.second + 15) / 16 * 16;
  return (long long)n;
}
extern "C" int mqe_state_save(mqe_sim* s, void* host_blob, void* stream) {
  if (!s || !host_blob) return fail(-1, "mqe_state_save: null argument");
  if (s->step_open) return fail(-8, "mqe_state_save inside an open step (mqe_step_begin / _head without mqe_step_end)");
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  StateHeader h = {STATE_MAGIC, (uint32_t)MQE_ABI_VERSION, s->N, s->A, s->P, (int32_t)s->state_bufs.size(), s->hist_pos, s->n_post_steps, s->lag_pos, 0,
                   (uint64_t)mqe_state_size(s)};
  char* o = (char*)host_blob;
  memcpy(o, &h, sizeof h); o += sizeof h;
  for (auto& b : s->state_bufs) {
    HIPCHK(hipMemcpy(o, b.first, b.second, hipMemcpyDeviceToHost));
    o += (b.second + 15) / 16 * 16;
  }
  return 0;
}
extern "C" int mqe_state_load(mqe_sim* s, const void* host_blob, void* stream) {
  if (!s || !host_blob) return fail(-1, "mqe_state_load: null argument");
  if (s->step_open) return fail(-8, "mqe_state_load inside an open step");
  StateHeader h;
  memcpy(&h, host_blob, sizeof h);
  if (h.magic != STATE_MAGIC || h.abi != (uint32_t)MQE_ABI_VERSION) return fail(-6, "mqe_state_load: not a state blob of this ABI version");
  if (h.N != s->N || h.A != s->A || h.P != s->P || h.nbuf != (int32_t)s->state_bufs.size() || h.bytes != (uint64_t)mqe_state_size(s))
    return fail(-6, "mqe_state_load: the blob was saved from a handle of another shape (envs, agents, NPCs, layer-0 path)");
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  const char* o = (const char*)host_blob + sizeof h;
  for (auto& b : s->state_bufs) {
    HIPCHK(hipMemcpy(b.first, o, b.second, hipMemcpyHostToDevice));
    o += (b.second + 15) / 16 * 16;
  }
  s->hist_pos = h.hist_pos; s->n_post_steps = h.n_post_steps; s->lag_pos = h.lag_pos;
  return 0;
}
extern "C" int mqe_render_depth(mqe_sim* s, float* out_dev, int height, int width, float horizontal_fov_deg, const float* cam_pos3, const float* cam_rpy3,
                                float far_m, void* stream) {
  if (!s || !out_dev || !cam_pos3 || !cam_rpy3) return fail(-1, "null argument");
  if (height <= 0 || width <= 0 || height * width > 1 << 16) return fail(-6, "camera resolution out of range");
  if (!(horizontal_fov_deg > 1.0f && horizontal_fov_deg < 179.0f) || !(far_m > 0.0f)) return fail(-6, "camera field of view / far plane out of range");
  CamArgs ca;
  ca.out = out_dev; ca.H = height; ca.W = width; ca.tan_half_h = tanf(0.5f * horizontal_fov_deg * 3.14159265358979f / 180.0f);
  for (int k = 0; k < 3; k++) { ca.pos[k] = cam_pos3[k]; ca.rpy[k] = cam_rpy3[k]; }
  ca.far_ = far_m;
  hipLaunchKernelGGL(k_depth_camera, dim3(s->N), dim3(256), 0, (hipStream_t)stream, s->dm, s->st, ca);
  return hipGetLastError() == hipSuccess ? 0 : fail(-4, "k_depth_camera launch failed");
}
extern "C" int mqe_history_sync(mqe_sim* s, void* stream) {
  if (!s) return fail(-1, "null engine handle");
  if (!s->st.hist2) return 0;                    // the exact-f32 layer 0 reads the ring itself
  hipStream_t q = (hipStream_t)stream;
  if (hipMemsetAsync(s->st.hist_irr, 0, (size_t)s->R * sizeof(uint32_t), q) != hipSuccess) return fail(-4, "memset failed");
  const int n = s->R * MQE_HIST;
  hipLaunchKernelGGL(k_hist2_rebuild, dim3((n + 255) / 256), dim3(256), 0, q, s->dm, s->st, s->hist_pos);
  return hipGetLastError() == hipSuccess ? 0 : fail(-4, "k_hist2_rebuild launch failed");
}
extern "C" int mqe_debug_phase_times(mqe_sim* s, long long* out_host) {
  if (!s) return fail(-1, "null engine handle");
  if (!s->st.wave_times || !s->phase_timed) return fail(-4, "create the handle with MQE_PHASE_TIMES=1 (go1gate-, go1sheep- or go1football-defender-shaped scene)");
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out_host, s->st.wave_times + (size_t)4 * s->N, (size_t)64 * s->N * sizeof(long long), hipMemcpyDeviceToHost));
  return 0;
}
extern "C" int mqe_debug_epilogue_times(mqe_sim* s, long long* out_host) {
  if (!s) return fail(-1, "null engine handle");
  if (!s->st.wave_times || !s->phase_timed) return fail(-4, "create the handle with MQE_PHASE_TIMES=1 (go1gate-, go1sheep- or go1football-defender-shaped scene)");
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out_host, s->st.wave_times + (size_t)68 * s->N, (size_t)16 * s->N * sizeof(long long), hipMemcpyDeviceToHost));
  return 0;
}
extern "C" int mqe_debug_tail_times(mqe_sim* s, long long* out_host) {
  if (!s) return fail(-1, "null engine handle");
  if (!s->tail_times) return fail(-4, "create the handle with MQE_TAIL_TIMES=1 (fused policy tail)");
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out_host, s->tail_times, (size_t)16 * ((s->R + TL_ROWS - 1) / TL_ROWS) * sizeof(long long), hipMemcpyDeviceToHost));
  return 0;
}
extern "C" int mqe_debug_wave_times(mqe_sim* s, long long* out_host) {
  if (!s) return fail(-1, "null engine handle");
  if (!s->st.wave_times) return fail(-4, "create the handle with MQE_WAVE_TIMES=1");
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out_host, s->st.wave_times, (size_t)4 * s->N * sizeof(long long), hipMemcpyDeviceToHost));
  return 0;
}
extern "C" int mqe_debug_stop_phase(mqe_sim* s, int tap) {
  if (!s) return fail(-1, "null engine handle");
  if (tap >= 0 && !s->a2_scene) return fail(-4, "phase taps exist for two-robot scenes without objects only (k_simulate_a2)");
  s->dbg_stop_phase = tap < 0 ? -1 : tap;
  return 0;
}
extern "C" int mqe_policy_step(mqe_sim* s, const float* command, void* stream) {
  if (!s) return fail(-1, "null engine handle");
  policy_step(s, command, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return 0;
}
extern "C" int mqe_defender_command(mqe_sim* s, float* out_dev, void* stream) {
  if (!s) return fail(-1, "null engine handle");
  if (s->d.task != MQE_TASK_FOOTBALL_DEFENDER) return fail(-7, "defender command needs the football-defender task");
  hipLaunchKernelGGL(k_defender_command, dim3((s->N + 63) / 64), dim3(64), 0, (hipStream_t)stream, s->dm, s->st, out_dev);
  HIPCHK(hipGetLastError());
  return 0;
}
extern "C" int mqe_compute_torques(mqe_sim* s, void* stream) {
  if (!s) return fail(-1, "null engine handle");
  launch_torques(s, -1, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return 0;
}
extern "C" int mqe_simulate(mqe_sim* s, void* stream) {
  if (!s) return fail(-1, "null engine handle");
  launch_simulate(s, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return 0;
}
// post_decimation_step (legged_robot.py:112-115) of the unfused path: torques, joint velocities and soft-limit flags of substep dec_i
__global__ void k_post_decimation(const DevModel* m, DevState st, int dec_i) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int nj = 12 * m->A;
  if (idx >= m->N * nj) return;
  const int e = idx / nj, jt = idx - e * nj, j = jt % 12;
  const size_t o = ((size_t)e * 4 + dec_i) * nj + jt;
  const float q = st.dof[((size_t)e * m->ND + jt) * 2], qd = st.dof[((size_t)e * m->ND + jt) * 2 + 1];
  st.sub_tau[o] = st.torques[idx];
  st.sub_dof_vel[o] = qd;
  st.sub_exceed[o] = (q < m->soft_lo[j]) | (q > m->soft_hi[j]);
}

extern "C" int mqe_post_decimation_step(mqe_sim* s, int dec_i, void* stream) {
  if (!s) return fail(-1, "null engine handle");
  if (dec_i < 0 || dec_i >= 4) return fail(-1, "dec_i out of range");
  const int n = s->N * 12 * s->A;
  hipLaunchKernelGGL(k_post_decimation, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, s->dm, s->st, dec_i);
  HIPCHK(hipGetLastError());
  return 0;
}
extern "C" int mqe_post_physics_step(mqe_sim* s, void* stream) {
  if (!s) return fail(-1, "null engine handle");
  launch_post(s, (hipStream_t)stream, 0);
  HIPCHK(hipGetLastError());
  return 0;
}
extern "C" int mqe_post_physics_stage(mqe_sim* s, int stages, void* stream) {
  if (!s) return fail(-1, "null engine handle");
  if ((stages & ~(MQE_POST_ALL | MQE_POST_WRAPPER_LEVEL)) != 0 || (stages & MQE_POST_ALL) == 0) return fail(-2, "mqe_post_physics_stage: stages must be an OR of MQE_POST_*");
  hipStream_t q = (hipStream_t)stream;
  const int step_no = s->n_post_steps + 1;           // = common_step_counter after its increment (legged_robot.py:127), the same for every stage of the step
  const int push = (s->d.push_interval > 0 && step_no % s->d.push_interval == 0) ? (int)(step_no / s->d.push_interval) : 0;
  if ((stages & MQE_POST_RESET) && s->hm.curriculum) hipLaunchKernelGGL(k_curriculum_snapshot, dim3((s->N + 255) / 256), dim3(256), 0, q, s->dm, s->st);
  hipLaunchKernelGGL(k_post_staged, dim3((s->N + 63) / 64), dim3(64), 0, q, s->dm, s->st, stages & MQE_POST_ALL, (stages & MQE_POST_WRAPPER_LEVEL) ? 1 : 0, push, step_no);
  if (stages & MQE_POST_WRAPPER) s->n_post_steps++;
  HIPCHK(hipGetLastError());
  return 0;
}
extern "C" int mqe_wrapper_eval(mqe_sim* s, int is_reset_call, void* stream) {
  if (!s) return fail(-1, "null engine handle");
  hipLaunchKernelGGL(k_wrapper_eval, dim3((s->N + 63) / 64), dim3(64), 0, (hipStream_t)stream, s->dm, s->st, is_reset_call);
  HIPCHK(hipGetLastError());
  return 0;
}
extern "C" int mqe_set_actor_root_state_indexed(mqe_sim*, const int32_t*, int, void*) { return 0; }
extern "C" int mqe_set_dof_state_indexed(mqe_sim*, const int32_t*, int, void*) { return 0; }

extern "C" int mqe_reset_all(mqe_sim* s, void* stream) {
  if (!s) return fail(-1, "null engine handle");
  hipStream_t q = (hipStream_t)stream;
  if (s->hm.curriculum) hipLaunchKernelGGL(k_curriculum_snapshot, dim3((s->N + 255) / 256), dim3(256), 0, q, s->dm, s->st);
  hipLaunchKernelGGL(k_reset_all, dim3((s->N + 63) / 64), dim3(64), 0, q, s->dm, s->st, s->n_post_steps == 0 ? 1 : 0);
  int n = s->R * (MQE_HIST * MQE_FRAME / 4);
  hipLaunchKernelGGL(k_reset_history, dim3((n + 255) / 256), dim3(256), 0, q, s->dm, s->st);
  HIPCHK(hipGetLastError());
  return 0;
}

static int run_substeps_and_post(mqe_sim* s, hipStream_t q, int wrapper_level = 1);
__global__ void k_post_decimation(const DevModel* m, DevState st, int dec_i);

// clip to clip_actions (legged_robot.py:108-110) -> st.actions
__global__ void k_set_joint_actions(const DevModel* m, DevState st, const float* __restrict__ a12) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < m->R * 12) st.actions[idx] = clampf(a12[idx], -m->clip_actions, m->clip_actions);
}

extern "C" int mqe_step_joint(mqe_sim* s, const float* actions12, void* stream) {
  if (!s) return fail(-1, "null engine handle");
  if (s->d.control_type == MQE_CTRL_C) return fail(-7, "mqe_step_joint drives control types P / V / T; use mqe_step for the hierarchical controller");
  hipStream_t q = (hipStream_t)stream;
  s->prof_now = s->prof && (s->prof_step++ % s->prof_every == prof_phase(s->prof_every));
  {
    ProfScope ps(s, PROF_MISC, q);
    const int n = s->R * 12;
    hipLaunchKernelGGL(k_set_joint_actions, dim3((n + 255) / 256), dim3(256), 0, q, s->dm, s->st, actions12);
  }
  return run_substeps_and_post(s, q);
}

extern "C" int mqe_step_command(mqe_sim* s, const float* command, void* stream) {
  if (!s) return fail(-1, "null engine handle");
  hipStream_t q = (hipStream_t)stream;
  if (s->d.control_type != MQE_CTRL_C) return fail(-7, "mqe_step_command drives the hierarchical controller (control type C); use mqe_step_joint for P / V / T");
  if (s->step_open) return fail(-8, "mqe_step_command inside an open step");
  s->prof_now = s->prof && (s->prof_step++ % s->prof_every == prof_phase(s->prof_every));
  policy_step(s, command, q);                    // Go1-level commands: no wrapper head
  return run_substeps_and_post(s, q, 0);         // ... and no wrapper evaluation: the wrapper's bookkeeping belongs to wrapper-level steps
}

extern "C" int mqe_step(mqe_sim* s, const float* actions, void* stream) {
  if (!s) return fail(-1, "null engine handle");
  hipStream_t q = (hipStream_t)stream;
  if (s->d.control_type != MQE_CTRL_C) return fail(-7, "mqe_step drives the hierarchical controller (control type C); use mqe_step_joint for P / V / T");
  if (s->cmd_general) return fail(-7, "mqe_step takes wrapper-level (N, A', 3) actions; this handle's command layout (desc.command_src) is served by mqe_policy_step + the stage entry points");
  s->prof_now = s->prof && (s->prof_step++ % s->prof_every == prof_phase(s->prof_every));
  policy_step(s, s->st.cmd, q, actions);         // wrapper head (clip, task action scale, scripted defender) inside k_pre_policy
  return run_substeps_and_post(s, q);
}

extern "C" int mqe_step_begin(mqe_sim* s, const float* actions, void* stream) {
  if (!s) return fail(-1, "null engine handle");
  hipStream_t q = (hipStream_t)stream;
  if (s->d.control_type != MQE_CTRL_C) return fail(-7, "mqe_step_begin drives the hierarchical controller (control type C)");
  if (s->step_open) return fail(-8, "mqe_step_begin: the previous step was not closed with mqe_step_end");
  if (s->cmd_general) return fail(-7, "mqe_step_begin takes wrapper-level (N, A', 3) actions; this handle's command layout (desc.command_src) is served by mqe_policy_step + the stage entry points");
  s->prof_now = s->prof && (s->prof_step++ % s->prof_every == prof_phase(s->prof_every));
  policy_step(s, s->st.cmd, q, actions);
  s->step_open = 2;
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int mqe_step_head(mqe_sim* s, const float* actions, void* stream) {
  if (!s) return fail(-1, "null engine handle");
  if (s->d.control_type != MQE_CTRL_C) return fail(-7, "mqe_step_head drives the hierarchical controller (control type C)");
  if (s->step_open) return fail(-8, "mqe_step_head: the previous step was not closed with mqe_step_end");
  if (s->cmd_general) return fail(-7, "mqe_step_head takes wrapper-level (N, A', 3) actions; this handle's command layout (desc.command_src) is served by mqe_policy_step + the stage entry points");
  s->prof_now = s->prof && (s->prof_step++ % s->prof_every == prof_phase(s->prof_every));
  policy_head(s, s->st.cmd, (hipStream_t)stream, actions);
  s->step_open = 1;
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int mqe_step_tail(mqe_sim* s, void* stream) {
  if (!s) return fail(-1, "null engine handle");
  if (s->step_open != 1) return fail(-8, "mqe_step_tail without mqe_step_head");
  policy_tail(s, (hipStream_t)stream);
  s->step_open = 2;
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int mqe_set_return_buffer(mqe_sim* s, float* packed_dev) {
  if (!s) return fail(-1, "mqe_set_return_buffer: no engine");
  float* base = packed_dev ? packed_dev : (float*)s->tens[MQE_T_WRAPPER_PACKED];
  s->st.wobs = base;
  s->st.wrew = base + (size_t)s->N * s->Aw * s->D;
  s->st.wdone = (uint8_t*)(s->st.wrew + (size_t)s->N * s->Aw);
  return 0;
}

extern "C" int mqe_step_end(mqe_sim* s, void* stream) {
  if (!s) return fail(-1, "null engine handle");
  if (s->step_open != 2) return fail(-8, "mqe_step_end without mqe_step_begin (or mqe_step_head + mqe_step_tail)");
  s->step_open = 0;
  return run_substeps_and_post(s, (hipStream_t)stream);
}

static int run_substeps_and_post(mqe_sim* s, hipStream_t q, int wrapper_level) {
  if (s->fuse_substeps) {
    // decimation loop in one launch: state stays in LDS; actuator net on MFMA (C) or the PD / torque law (P, V, T) inside the wavefront
    ProfScope ps(s, PROF_SIMULATE, q);
    PostArgs pa = {0, 0, 0, 0};
    if (s->fuse_post) {                        // the post-physics step rides along as the kernel's epilogue (launch_post's bookkeeping here)
      s->n_post_steps++;
      pa.on = 1; pa.wrapper_level = wrapper_level; pa.step_no = s->n_post_steps;
      pa.push_count = (s->d.push_interval > 0 && s->n_post_steps % s->d.push_interval == 0) ? (int)(s->n_post_steps / s->d.push_interval) : 0;
    }
    hipLaunchKernelGGL(s->substeps_fn, dim3((s->N + s->substeps_epw - 1) / s->substeps_epw), dim3(64), s->phys_lds_bytes * s->substeps_epw, q, s->dm, s->st, s->d.decimation, s->lag_pos, pa);
    advance_lag(s, s->d.decimation);
    if (s->fuse_post) {
      s->prof_now = s->prof;
      HIPCHK(hipGetLastError());
      return 0;
    }
  } else {
    for (int k = 0; k < s->d.decimation; k++) {
      launch_torques(s, k < 4 ? k : 3, q);
      launch_simulate(s, q);
      const int n = s->N * 12 * s->A;
      hipLaunchKernelGGL(k_post_decimation, dim3((n + 255) / 256), dim3(256), 0, q, s->dm, s->st, k < 4 ? k : 3);
    }
  }
  launch_post(s, q, wrapper_level);
  s->prof_now = s->prof;                      // the unfused entry points are always bracketed when profiling is on
  HIPCHK(hipGetLastError());
  return 0;
}

// debug: M^-1 of one robot and the contact list of one env from the current state, without advancing it
static long long g_dbg_times[16];
extern "C" int mqe_debug_times(long long* out16) { memcpy(out16, g_dbg_times, sizeof g_dbg_times); return 0; }
extern "C" int mqe_debug_dynamics(mqe_sim* s, int env, int robot, float* minv_out_host, int* nc_out_host, float* contacts_out_host) {
  if (!s) return fail(-1, "null engine handle");
  float *dm_, *dc; int* dn; long long* dtm;
  HIPCHK(hipMalloc(&dm_, 324 * 4)); HIPCHK(hipMalloc(&dc, 64 * 8 * 4)); HIPCHK(hipMalloc(&dn, 4)); HIPCHK(hipMalloc(&dtm, 16 * 8));
  HIPCHK(hipMemset(dtm, 0, 16 * 8));
  PhysDebug dbg = {dm_, dn, dc, robot, dtm, -1};
  hipLaunchKernelGGL(k_simulate, dim3(1), dim3(64), s->phys_lds_bytes, 0, s->dm, s->st, env, 1, dbg);
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(minv_out_host, dm_, 324 * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(nc_out_host, dn, 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(contacts_out_host, dc, 64 * 8 * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(g_dbg_times, dtm, 16 * 8, hipMemcpyDeviceToHost));
  hipFree(dtm);
  hipFree(dm_); hipFree(dc); hipFree(dn);
  return 0;
}
