/* mqe_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the env.step() hot path of ziyanx02/multiagent-quadruped-environment (MQE) used as
 * the parity checker for the HIP engine (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg only).
 *
 * Pinning status:
 *   - everything AROUND the physics (policy MLPs, torque pipeline, command observation/history, gait clock,
 *     termination, reset bookkeeping, observation bag, task wrappers, NPC scripts) restates the reference's
 *     Python line by line (citations at each function) and is pinned by the .npz files of tests/golden, which were produced
 *     by importing that Python (tools/gen_golden.py).
 *   - the rigid-body physics (mqo_simulate) has NO reference to restate: in MQE it lives inside Isaac Gym
 *     Preview 4 / PhysX (closed source, CUDA only; call sites go1.py:52-56).  PARITY UNPINNED for that row: the
 *     algorithm here is the build's own specification (DESIGN.md "physics"), written the textbook way
 *     (per-body Jacobian sums for the mass matrix, dense Cholesky, row-wise projected Gauss-Seidel) so that it
 *     is an independent check of the wavefront-parallel HIP formulation (CRBA, Schur-complement inverse).
 *     It is pinned by physical known-answer tests (free fall, momentum, energy, static stand) in tests/.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).  -DREAL=double builds the float64 variant.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/mqe_hip.h"

#ifndef REAL
#define REAL float
#endif
typedef REAL real;

#define MAXA MQE_MAX_AGENTS
#define MAXP MQE_MAX_NPCS
#define NB MQE_NBODY
#define RD 18                       /* dofs of one robot: 3 lin + 3 ang + 12 joints */
#define MAXDOF (MAXA * RD + MAXP * 6 + 1)
#define MAXC 64                     /* storage; the active bound is env_maxc() */
static inline int env_maxc(int A, int P, int cap_npc) { int v = 8 * A + cap_npc * P + (A + P > 4 ? 8 : 0); return v > 64 ? 64 : v; } /* = mqe_maxc() of the engine: per-actor caps + eight pair-only slots in scenes of more than four actors */
#define LIMIT_PASSES 4 /* Gauss-Seidel passes of the joint position / speed limits per substep (= MQE_LIMIT_PASSES of the engine) */
#define CAP_ROBOT 8   /* terrain / static-object contacts kept per robot (spheres are priority ordered: feet first) */
#define FR MQE_FRAME
#define OBS_BAG 74

static char g_err[512];
const char* mqo_last_error(void) { return g_err; }
int mqo_sizeof_desc(void) { return (int)sizeof(mqe_sim_desc); }
int mqo_num_threads(void);
void mqo_set_num_threads(int n);
/* the checker's own team size (omp_set_num_threads would change every other OpenMP user of the process -- torch's CPU ops share
 * libgomp): 0 = the runtime's default; the parallel regions below carry it as their num_threads clause */
static int g_team = 0;
static inline int mqo_team(void);

typedef struct {
  int n_layers;
  int dims[MQE_MAX_LAYERS + 1];
  float* W[MQE_MAX_LAYERS];
  float* b[MQE_MAX_LAYERS];
} mlp_t;

typedef struct mqo_sim {
  mqe_sim_desc d;
  int N, A, P, R, ND, NBR, Aw, D;   /* ND dofs per env in DOF_STATE, NBR reported bodies per env, Aw wrapper agents */
  int npc_dofs, npc_bodies;
  int wrapper_side_effects;   /* 1 while a wrapper-level call runs (mqo_step / mqo_wrapper_eval): go1tug re-poses its slider */
  mlp_t act, ada, body;
  float* sdf;
  float* ground_height;             /* relief of the walkable surface at the SDF's raster points, or NULL (flat slab) */
  float* wall_top;                  /* per-cell wall top (walls of different heights), or NULL */
  float* wall_corner;               /* per raster point the (x, y) of the nearest convex corner of the wall set, or NULL (edge contacts) */
  float *env_origins, *agent_origins, *base_init, *npc_init, *gate_pos;
  float* pre_npc;                   /* [N][P][13] the wrapper's copy of the NPC rows, taken before the NPC script (legged_robot.py:136) */
  float *env_origins_live, *terrain_origins, *curr_xy;   /* terrain curriculum: MQE_T_ENV_ORIGINS, the origin table, pre-reset xy of the robot rows */
  int32_t *terrain_levels, *terrain_types;
  /* state */
  float *root, *dof, *cf, *torques, *actions, *last_actions, *loco_obs, *hist, *last_loco, *last_two_loco;
  float *act_hist, *gait, *clock, *blv, *bav, *pg, *bquat, *obs_bag, *wobs, *wrew, *rsum, *sheep_avg, *sheep_var;
  float *sub_tau, *npc_noise, *last_dof_vel;
  float* sub_dof_vel; uint8_t* sub_exceed; int32_t* overflow;   /* legged_robot.py:114-115 logs; truncated-contact-list counter */
  float soft_lo[12], soft_hi[12];
  float *dparams, *lag_buf;         /* [R][8] MQE_T_DOMAIN_PARAMS; [(lag + 1)][R][12] scaled actions (domain randomisation, include/mqe_hip.h) */
  int lag_pos;
  int32_t *ep_len, *reset_count;
  uint8_t *reset_buf, *collide_buf, *time_out, *r_term, *p_term, *zh_term;
  /* wrapper memory */
  float *w_last, *w_last2;          /* per env: previous distances / x positions */
  uint8_t *w_have_last, *w_delayed_reset;
  uint8_t* wdone;                   /* byte tail of the packed return batch */
  int hist_pos;                     /* ring slot that holds the OLDEST frame == next write slot */
  int n_post_steps;                 /* post_physics_step calls so far (base_quat aliasing, see mqo_reset_all) */
  void* tens[MQE_T_COUNT];
  /* EXPERIMENT (MQO_WARM_START=<factor> in the environment at creation; not part of the specification, not in the HIP engine): contact
   * impulses of the previous substep as the sweep's starting point, matched by (kind, actors, links) and position (tools/warm_start_delta.py) */
  float warm; int32_t* wc_n; int32_t* wc_key; float* wc_p; float* wc_lam;
} mqo_sim;

/* ------------------------------------------------------------------------------------------ small math */
static inline void cross3(const real* a, const real* b, real* o) {
  real x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
static inline real dot3(const real* a, const real* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void mat3_vec(const real* M, const real* v, real* o) {
  real x = M[0] * v[0] + M[1] * v[1] + M[2] * v[2], y = M[3] * v[0] + M[4] * v[1] + M[5] * v[2],
       z = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static inline void mat3_mul(const real* A, const real* B, real* C) {
  real t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) t[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
  memcpy(C, t, sizeof t);
}
static void quat_to_mat(const real* q, real* R) { /* xyzw */
  real x = q[0], y = q[1], z = q[2], w = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
}
static void axis_angle_mat(const real* a, real th, real* R) {
  real c = (real)cos((double)th), s = (real)sin((double)th), t = 1 - c;
  R[0] = t * a[0] * a[0] + c; R[1] = t * a[0] * a[1] - s * a[2]; R[2] = t * a[0] * a[2] + s * a[1];
  R[3] = t * a[0] * a[1] + s * a[2]; R[4] = t * a[1] * a[1] + c; R[5] = t * a[1] * a[2] - s * a[0];
  R[6] = t * a[0] * a[2] - s * a[1]; R[7] = t * a[1] * a[2] + s * a[0]; R[8] = t * a[2] * a[2] + c;
}

/* quat_rotate_inverse of isaacgym.torch_utils as used at legged_robot.py:133-135 (float32, same op order) */
static void quat_rotate_inverse_f(const float* q, const float* v, float* o) {
  float qw = q[3];
  float s = 2.0f * qw * qw - 1.0f;
  float cx = q[1] * v[2] - q[2] * v[1], cy = q[2] * v[0] - q[0] * v[2], cz = q[0] * v[1] - q[1] * v[0];
  float dt = q[0] * v[0] + q[1] * v[1] + q[2] * v[2];
  o[0] = v[0] * s - cx * qw * 2.0f + q[0] * dt * 2.0f;
  o[1] = v[1] * s - cy * qw * 2.0f + q[1] * dt * 2.0f;
  o[2] = v[2] * s - cz * qw * 2.0f + q[2] * dt * 2.0f;
}
static float wrap2pi(float a) { /* python float32 % (2*pi) */
  const float T = 6.2831855f;
  float r = fmodf(a, T);
  if (r < 0) r += T;
  return r;
}
/* get_euler_xyz of isaacgym.torch_utils (go1.py:193, legged_robot_field.py:125): each angle in [0, 2pi) */
static void euler_xyz_f(const float* q, float* rpy) {
  float x = q[0], y = q[1], z = q[2], w = q[3];
  float sinr = 2.0f * (w * x + y * z), cosr = w * w - x * x - y * y + z * z;
  float sinp = 2.0f * (w * y - z * x);
  float siny = 2.0f * (w * z + x * y), cosy = w * w + x * x - y * y - z * z;
  float roll = atan2f(sinr, cosr);
  float pitch = fabsf(sinp) >= 1.0f ? copysignf(1.5707964f, sinp) : asinf(sinp);
  float yaw = atan2f(siny, cosy);
  rpy[0] = wrap2pi(roll); rpy[1] = wrap2pi(pitch); rpy[2] = wrap2pi(yaw);
}

/* ------------------------------------------------------------------------------------------ RNG (resets) */
static inline uint32_t mqo_hash(uint32_t seed, uint32_t genv, uint32_t count, uint32_t k) {
  uint32_t x = seed * 0x9E3779B1u ^ genv * 0x85EBCA77u ^ count * 0xC2B2AE3Du ^ k * 0x27D4EB2Fu;
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
static inline float mqo_u01(uint32_t seed, uint32_t genv, uint32_t count, uint32_t k) {
  return (float)(mqo_hash(seed, genv, count, k) >> 8) * (1.0f / 16777216.0f);
}
static inline float mqo_rand(const mqo_sim* s, int env, uint32_t k, float lo, float hi) {
  float u = mqo_u01((uint32_t)s->d.seed, (uint32_t)(env + s->d.env_id_offset), (uint32_t)s->reset_count[env], k);
  return (hi - lo) * u + lo; /* torch_rand_float: (upper-lower)*rand + lower */
}
float mqo_debug_u01(uint32_t seed, uint32_t genv, uint32_t count, uint32_t k) { return mqo_u01(seed, genv, count, k); }

/* ------------------------------------------------------------------------------------------ MLPs */
static void mlp_copy(mlp_t* dst, const mqe_mlp* src) {
  dst->n_layers = src->n_layers;
  memcpy(dst->dims, src->dims, sizeof dst->dims);
  for (int l = 0; l < src->n_layers; l++) {
    size_t nw = (size_t)src->dims[l] * src->dims[l + 1];
    dst->W[l] = (float*)malloc(nw * 4);
    memcpy(dst->W[l], src->W[l], nw * 4);
    dst->b[l] = (float*)malloc((size_t)src->dims[l + 1] * 4);
    memcpy(dst->b[l], src->b[l], (size_t)src->dims[l + 1] * 4);
  }
}
static inline float softsign(float x) { return x / (1.0f + fabsf(x)); }
static inline float elu(float x) { return x > 0 ? x : expm1f(x); }
/* y = W x (+ b afterwards): k-ordered fmaf chain from 0, bias added last -- the accumulation order the HIP
 * kernels use (an f32 MFMA is bitwise a k-ordered fmaf chain on gfx950) */
static inline float dot_chain(const float* w, const float* x, int K) {
  float acc = 0.0f;
  for (int k = 0; k < K; k++) acc = fmaf(w[k], x[k], acc);
  return acc;
}
/* y[o] = chain(W[o,:], x) for O outputs: identical per-output k-ordered chains, eight outputs interleaved so that the
 * CPU pipelines them (the dependency chain of one output is the bottleneck otherwise).  Two clones, picked by the loader from the
 * host's CPUID: with FMA3 every fmaf is one vfmadd instruction, without it the libm call -- the same correctly rounded result
 * either way (5 x the checker's speed on the policy layers, which is what bench.py's cpu_baseline spends its time in) */
#if defined(__x86_64__) && defined(__GNUC__) && !defined(MQO_NO_FMA_CLONE)    /* (tests build the checker once without the clone and compare bit for bit) */
__attribute__((target_clones("fma", "default")))
#endif
static void matvec_chain(const float* W, const float* x, int K, int O, int ldw, float* y) {
  int o = 0;
  for (; o + 8 <= O; o += 8) {
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    const float *w0 = W + (size_t)o * ldw, *w1 = w0 + ldw, *w2 = w1 + ldw, *w3 = w2 + ldw, *w4 = w3 + ldw, *w5 = w4 + ldw, *w6 = w5 + ldw, *w7 = w6 + ldw;
    for (int k = 0; k < K; k++) {
      float xv = x[k];
      a0 = fmaf(w0[k], xv, a0); a1 = fmaf(w1[k], xv, a1); a2 = fmaf(w2[k], xv, a2); a3 = fmaf(w3[k], xv, a3);
      a4 = fmaf(w4[k], xv, a4); a5 = fmaf(w5[k], xv, a5); a6 = fmaf(w6[k], xv, a6); a7 = fmaf(w7[k], xv, a7);
    }
    y[o] = a0; y[o + 1] = a1; y[o + 2] = a2; y[o + 3] = a3; y[o + 4] = a4; y[o + 5] = a5; y[o + 6] = a6; y[o + 7] = a7;
  }
  for (; o < O; o++) y[o] = dot_chain(W + (size_t)o * ldw, x, K);
}

/* actuator net, reference go1.py:367-382 (unitree_go1.pt: Linear(6,32) softsign Linear(32,32) softsign Linear(32,1)) */
static float actuator_net(const mlp_t* m, const float* x6) {
  float h1[32], h2[32];
  for (int i = 0; i < 32; i++) h1[i] = softsign(dot_chain(m->W[0] + i * 6, x6, 6) + m->b[0][i]);
  for (int i = 0; i < 32; i++) h2[i] = softsign(dot_chain(m->W[1] + i * 32, h1, 32) + m->b[1][i]);
  return dot_chain(m->W[2], h2, 32) + m->b[2][0];
}
float mqo_actuator_net(mqo_sim* s, const float* x6) { return actuator_net(&s->act, x6); }

/* logical (time-ordered, 2100) history of robot i gathered from the ring */
static void gather_history(const mqo_sim* s, int i, float* out2100) {
  for (int f = 0; f < MQE_HIST; f++) {
    int slot = (s->hist_pos + f) % MQE_HIST;
    memcpy(out2100 + f * 70, s->hist + ((size_t)i * MQE_HIST + slot) * FR, 70 * 4);
  }
}
void mqo_history(mqo_sim* s, float* out) {
  for (int i = 0; i < s->R; i++) gather_history(s, i, out + (size_t)i * 2100);
}
/* adaptation module + body, reference go1.py:400-407 */
static void policy_forward(const mqo_sim* s, const float* h2100, float* latent2, float* act12) {
  const mlp_t* a = &s->ada;
  float buf0[1024], buf1[1024];
  const float* x = h2100;
  int K = 2100;
  for (int l = 0; l < a->n_layers; l++) {
    float* y = (l & 1) ? buf1 : buf0;
    int O = a->dims[l + 1];
    matvec_chain(a->W[l], x, K, O, K, y);
    for (int o = 0; o < O; o++) {
      float v = y[o] + a->b[l][o];
      y[o] = (l < a->n_layers - 1) ? elu(v) : v;
    }
    x = y; K = O;
  }
  latent2[0] = x[0]; latent2[1] = x[1];
  const mlp_t* b = &s->body;
  /* layer 0: 2100 history columns as a chain, + bias, then the two latent columns */
  int O = b->dims[1], K0 = b->dims[0];
  float* y = buf0;
  matvec_chain(b->W[0], h2100, 2100, O, K0, y);
  for (int o = 0; o < O; o++) {
    const float* w = b->W[0] + (size_t)o * K0;
    float v = y[o] + b->b[0][o];
    v = fmaf(latent2[0], w[2100], v);
    v = fmaf(latent2[1], w[2101], v);
    y[o] = elu(v);
  }
  x = y; K = O;
  for (int l = 1; l < b->n_layers; l++) {
    float* yy = (l & 1) ? buf1 : buf0;
    int OO = b->dims[l + 1];
    matvec_chain(b->W[l], x, K, OO, K, yy);
    for (int o = 0; o < OO; o++) {
      float v = yy[o] + b->b[l][o];
      yy[o] = (l < b->n_layers - 1) ? elu(v) : v;
    }
    x = yy; K = OO;
  }
  for (int j = 0; j < 12; j++) act12[j] = x[j];
}
void mqo_policy_forward(mqo_sim* s, const float* h2100, float* latent2, float* act12) { policy_forward(s, h2100, latent2, act12); }

/* ------------------------------------------------------------------------------------------ create / tensors */
#define ALLOCF(n) ((float*)calloc((size_t)(n) > 0 ? (size_t)(n) : 1, sizeof(float)))
static void* dupmem(const void* p, size_t n) {
  if (!p) return NULL;
  void* q = malloc(n);
  memcpy(q, p, n);
  return q;
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
    default: *Aw = A; *D = 6 + A; break; /* plain: [id, base_pos, base_rpy] */
  }
  return 0;
}

int mqo_sim_create(const mqe_sim_desc* d, mqo_sim** out) {
  if (d->abi_version != MQE_ABI_VERSION) { snprintf(g_err, sizeof g_err, "abi version mismatch"); return -1; }
  if (d->num_agents > MAXA || d->num_npcs > MAXP) { snprintf(g_err, sizeof g_err, "too many agents/npcs"); return -2; }
  if (d->solver_type == 1 && d->solver_iterations < 1) { snprintf(g_err, sizeof g_err, "solver_type = 1 (temporal Gauss-Seidel) needs num_position_iterations >= 1"); return -6; }
  if (d->solver_iterations < 0 || d->solver_iterations > 64) { snprintf(g_err, sizeof g_err, "solver_iterations out of range (0 .. 64)"); return -6; }
  mqo_sim* s = (mqo_sim*)calloc(1, sizeof *s);
  s->d = *d;
  int N = s->N = d->num_envs, A = s->A = d->num_agents, P = s->P = d->num_npcs;
  s->R = N * A;
  s->npc_dofs = d->npc_kind == MQE_NPC_SEESAW ? 1 : 0;
  s->npc_bodies = d->npc_kind == MQE_NPC_SEESAW ? 2 * P : (d->npc_kind == MQE_NPC_STATIC ? d->npc_reported_bodies * P : P);
  s->ND = 12 * A + s->npc_dofs;
  s->NBR = MQE_NREP * A + s->npc_bodies;
  wrapper_dims(d, &s->Aw, &s->D);
  mlp_copy(&s->act, &d->actuator);
  mlp_copy(&s->ada, &d->adaptation);
  mlp_copy(&s->body, &d->body);
  s->sdf = (float*)dupmem(d->wall_sdf, (size_t)d->sdf_nx * d->sdf_ny * 4);
  s->ground_height = d->ground_height ? (float*)dupmem(d->ground_height, (size_t)d->sdf_nx * d->sdf_ny * 4) : NULL;
  s->wall_top = d->wall_top ? (float*)dupmem(d->wall_top, (size_t)d->sdf_nx * d->sdf_ny * 4) : NULL;
  s->wall_corner = d->wall_corner ? (float*)dupmem(d->wall_corner, (size_t)d->sdf_nx * d->sdf_ny * 8) : NULL;
  s->env_origins = (float*)dupmem(d->env_origins, (size_t)N * 3 * 4);
  s->env_origins_live = (float*)dupmem(d->env_origins, (size_t)N * 3 * 4);
  s->terrain_levels = (int32_t*)calloc((size_t)N, 4);
  s->terrain_types = (int32_t*)calloc((size_t)N, 4);
  s->curr_xy = (float*)calloc((size_t)N * 2, 4);
  if (d->terrain_curriculum) {
    if (d->env_id_offset != 0) { snprintf(g_err, sizeof g_err, "terrain curriculum: not defined for a shard of a larger batch"); return -6; }
    if (!d->terrain_origins || !d->terrain_levels || !d->terrain_types || d->terrain_num_rows < 1 || d->terrain_num_cols < 1) { snprintf(g_err, sizeof g_err, "terrain curriculum: origin table / levels / types missing"); return -6; }
    s->terrain_origins = (float*)dupmem(d->terrain_origins, (size_t)d->terrain_num_rows * d->terrain_num_cols * 3 * 4);
    memcpy(s->terrain_levels, d->terrain_levels, (size_t)N * 4);
    memcpy(s->terrain_types, d->terrain_types, (size_t)N * 4);
  }
  s->agent_origins = (float*)dupmem(d->agent_origins, (size_t)N * A * 3 * 4);
  s->base_init = (float*)dupmem(d->base_init_state, (size_t)A * 13 * 4);
  s->npc_init = (float*)dupmem(d->npc_init_state, (size_t)P * 13 * 4);
  s->gate_pos = (float*)dupmem(d->gate_pos, (size_t)N * 2 * 4);
  int R = s->R;
  s->root = ALLOCF((size_t)N * (A + P) * 13);
  for (int i = 0; i < N * (A + P); i++) s->root[i * 13 + 6] = 1.0f;
  s->dof = ALLOCF((size_t)N * s->ND * 2);
  s->cf = ALLOCF((size_t)N * s->NBR * 3);
  s->torques = ALLOCF((size_t)N * 12 * A);
  s->actions = ALLOCF((size_t)N * 12 * A);
  s->last_actions = ALLOCF((size_t)N * 12 * A);
  s->loco_obs = ALLOCF((size_t)R * FR);
  for (int i = 0; i < R; i++) memcpy(s->loco_obs + (size_t)i * FR, d->command_obs, 70 * 4);
  s->hist = ALLOCF((size_t)R * MQE_HIST * FR);
  s->last_loco = ALLOCF((size_t)R * 12);
  s->last_two_loco = ALLOCF((size_t)R * 12);
  s->act_hist = ALLOCF((size_t)4 * R * 12);
  s->gait = ALLOCF(R);
  s->clock = ALLOCF((size_t)R * 4);
  s->blv = ALLOCF((size_t)R * 3);
  s->bav = ALLOCF((size_t)R * 3);
  s->pg = ALLOCF((size_t)R * 3);
  for (int i = 0; i < R; i++) s->pg[i * 3 + 2] = -1.0f;
  s->bquat = ALLOCF((size_t)R * 4);
  for (int i = 0; i < R; i++) s->bquat[i * 4 + 3] = 1.0f;
  s->obs_bag = ALLOCF((size_t)R * OBS_BAG);
  s->wobs = ALLOCF((size_t)N * s->Aw * s->D + (size_t)N * s->Aw + (size_t)(N + 3) / 4);
  s->wrew = s->wobs + (size_t)N * s->Aw * s->D;        /* one buffer, as in the engine (MQE_T_WRAPPER_PACKED): obs | reward | done */
  s->wdone = (uint8_t*)(s->wrew + (size_t)N * s->Aw);
  s->rsum = ALLOCF((size_t)N * MQE_MAX_REWARD_TERMS);
  s->sheep_avg = ALLOCF((size_t)N * 2);
  s->sheep_var = ALLOCF(N);
  s->sub_tau = ALLOCF((size_t)N * 4 * 12 * A);
  s->sub_dof_vel = ALLOCF((size_t)N * 4 * 12 * A);
  s->sub_exceed = (uint8_t*)calloc((size_t)N * 4 * 12 * A, 1);
  s->overflow = (int32_t*)calloc(2 * (size_t)N, 4);      /* [0, N): MQE_T_CONTACT_OVERFLOW, [N, 2 N): MQE_T_CONTACT_REDUCED */
  if (getenv("MQO_WARM_START") && atof(getenv("MQO_WARM_START")) > 0) {      /* EXPERIMENT, see mqo_sim::warm */
    s->warm = (float)atof(getenv("MQO_WARM_START"));
    s->wc_n = (int32_t*)calloc(N, 4); s->wc_key = (int32_t*)calloc((size_t)N * MAXC * 5, 4);
    s->wc_p = (float*)calloc((size_t)N * MAXC * 3, 4); s->wc_lam = (float*)calloc((size_t)N * MAXC * 3, 4);
  }
  {
    const float soft = d->soft_dof_pos_limit > 0.0f ? d->soft_dof_pos_limit : 1.0f;
    for (int j = 0; j < 12; j++) {                           /* legged_robot.py:317-321 */
      const float mid = (d->robot.dof_lower[j] + d->robot.dof_upper[j]) / 2, r = d->robot.dof_upper[j] - d->robot.dof_lower[j];
      s->soft_lo[j] = mid - 0.5f * r * soft; s->soft_hi[j] = mid + 0.5f * r * soft;
    }
  }
  /* domain parameters: drawn once, keyed by the global env id (the engine makes the same draws on its host side) */
  s->dparams = ALLOCF((size_t)R * 8);
  for (int e = 0; e < N; e++) {
    uint32_t genv = (uint32_t)(e + d->env_id_offset), seed = (uint32_t)d->seed;
    float mu = d->friction;
    if (d->rand_friction) {                  /* legged_robot.py:283-294: 64 buckets, one per env */
      uint32_t bucket = mqo_hash(seed, genv, 0xD0D0D0D0u, 0) % 64u;
      mu = d->friction_lo + (d->friction_hi - d->friction_lo) * mqo_u01(seed, bucket, 0xD0D0D0D1u, 0);
    }
    for (int a = 0; a < A; a++) {
      float* p = s->dparams + ((size_t)e * A + a) * 8;
      p[0] = mu;
      if (d->rand_base_mass) p[1] = d->added_mass_lo + (d->added_mass_hi - d->added_mass_lo) * mqo_u01(seed, genv, 0xD0D0D0D0u, 16u + (uint32_t)a);
      if (d->rand_com)
        for (int k = 0; k < 3; k++) p[2 + k] = d->com_lo[k] + (d->com_hi[k] - d->com_lo[k]) * mqo_u01(seed, genv, 0xD0D0D0D0u, 32u + (uint32_t)(a * 3 + k));
    }
  }
  s->lag_buf = (d->control_type == MQE_CTRL_C && d->lag_timesteps > 0) ? ALLOCF((size_t)(d->lag_timesteps + 1) * R * 12) : NULL;
  s->npc_noise = ALLOCF((size_t)N * (P ? P : 1) * 3);
  s->last_dof_vel = ALLOCF((size_t)s->R * 12);
  s->ep_len = (int32_t*)calloc(N, 4);
  s->reset_count = (int32_t*)calloc(N, 4);
  s->reset_buf = (uint8_t*)calloc(N, 1);
  memset(s->reset_buf, 1, N);
  s->collide_buf = (uint8_t*)calloc(N, 1);
  s->time_out = (uint8_t*)calloc(N, 1);
  s->r_term = (uint8_t*)calloc(N, 1);
  s->p_term = (uint8_t*)calloc(N, 1);
  s->zh_term = (uint8_t*)calloc(N, 1);
  s->w_last = ALLOCF((size_t)N * MAXA);
  s->w_last2 = ALLOCF((size_t)N * 2);
  s->w_have_last = (uint8_t*)calloc(N, 1);
  s->w_delayed_reset = (uint8_t*)calloc(N, 1);
  s->hist_pos = 0;
  void** t = s->tens;
  t[MQE_T_ROOT_STATE] = s->root; t[MQE_T_DOF_STATE] = s->dof; t[MQE_T_CONTACT_FORCE] = s->cf; t[MQE_T_TORQUES] = s->torques;
  t[MQE_T_ACTIONS] = s->actions; t[MQE_T_LAST_ACTIONS] = s->last_actions; t[MQE_T_LOCOMOTION_OBS] = s->loco_obs;
  t[MQE_T_HISTORY] = s->hist; t[MQE_T_LAST_LOCO_ACTION] = s->last_loco; t[MQE_T_LAST_TWO_LOCO_ACTION] = s->last_two_loco;
  t[MQE_T_ACT_HIST] = s->act_hist; t[MQE_T_GAIT_INDICES] = s->gait; t[MQE_T_CLOCK_INPUTS] = s->clock;
  t[MQE_T_BASE_LIN_VEL] = s->blv; t[MQE_T_BASE_ANG_VEL] = s->bav; t[MQE_T_PROJECTED_GRAVITY] = s->pg; t[MQE_T_BASE_QUAT] = s->bquat;
  t[MQE_T_EPISODE_LENGTH] = s->ep_len; t[MQE_T_RESET_BUF] = s->reset_buf; t[MQE_T_COLLIDE_BUF] = s->collide_buf;
  t[MQE_T_TIME_OUT_BUF] = s->time_out; t[MQE_T_R_TERM] = s->r_term; t[MQE_T_P_TERM] = s->p_term; t[MQE_T_Z_HIGH_TERM] = s->zh_term;
  t[MQE_T_OBS_BAG] = s->obs_bag; t[MQE_T_WRAPPER_OBS] = s->wobs; t[MQE_T_WRAPPER_REWARD] = s->wrew; t[MQE_T_REWARD_SUMS] = s->rsum;
  t[MQE_T_SHEEP_POS_AVG] = s->sheep_avg; t[MQE_T_SHEEP_POS_VAR] = s->sheep_var; t[MQE_T_RESET_COUNT] = s->reset_count;
  t[MQE_T_SUBSTEP_TORQUES] = s->sub_tau; t[MQE_T_NPC_NOISE] = s->npc_noise; t[MQE_T_WRAPPER_PACKED] = s->wobs;
  t[MQE_T_DOMAIN_PARAMS] = s->dparams;
  t[MQE_T_SUBSTEP_DOF_VEL] = s->sub_dof_vel; t[MQE_T_SUBSTEP_EXCEED_DOF_POS_LIMITS] = s->sub_exceed; t[MQE_T_CONTACT_OVERFLOW] = s->overflow;
  t[MQE_T_ENV_ORIGINS] = s->env_origins_live; t[MQE_T_TERRAIN_LEVELS] = s->terrain_levels;
  t[MQE_T_CONTACT_REDUCED] = s->overflow + s->N;
  *out = s;
  return 0;
}

int mqo_sim_destroy(mqo_sim* s) { free(s); return 0; } /* test helper: leaks the arrays on purpose (short-lived processes) */

int mqo_sim_tensor(mqo_sim* s, int kind, mqe_tensor_view* v) {
  if (kind < 0 || kind >= MQE_T_COUNT) { snprintf(g_err, sizeof g_err, "bad tensor kind %d", kind); return -1; }
  memset(v, 0, sizeof *v);
  v->ptr = s->tens[kind];
  int N = s->N, A = s->A, P = s->P, R = s->R;
#define SH(nd, a, b, c, e, dt) do { v->ndim = nd; v->shape[0] = a; v->shape[1] = b; v->shape[2] = c; v->shape[3] = e; v->dtype = dt; } while (0)
  switch (kind) {
    case MQE_T_ROOT_STATE: SH(3, N, A + P, 13, 0, 0); break;
    case MQE_T_DOF_STATE: SH(3, N, s->ND, 2, 0, 0); break;
    case MQE_T_CONTACT_FORCE: SH(3, N, s->NBR, 3, 0, 0); break;
    case MQE_T_TORQUES: case MQE_T_ACTIONS: case MQE_T_LAST_ACTIONS: SH(2, N, 12 * A, 0, 0, 0); break;
    case MQE_T_LOCOMOTION_OBS: SH(2, R, FR, 0, 0, 0); break;
    case MQE_T_HISTORY: SH(3, R, MQE_HIST, FR, 0, 0); break;
    case MQE_T_LAST_LOCO_ACTION: case MQE_T_LAST_TWO_LOCO_ACTION: SH(2, R, 12, 0, 0, 0); break;
    case MQE_T_ACT_HIST: SH(3, 4, R, 12, 0, 0); break;
    case MQE_T_GAIT_INDICES: SH(1, R, 0, 0, 0, 0); break;
    case MQE_T_CLOCK_INPUTS: case MQE_T_BASE_QUAT: SH(2, R, 4, 0, 0, 0); break;
    case MQE_T_BASE_LIN_VEL: case MQE_T_BASE_ANG_VEL: case MQE_T_PROJECTED_GRAVITY: SH(2, R, 3, 0, 0, 0); break;
    case MQE_T_EPISODE_LENGTH: case MQE_T_RESET_COUNT: SH(1, N, 0, 0, 0, 1); break;
    case MQE_T_RESET_BUF: case MQE_T_COLLIDE_BUF: case MQE_T_TIME_OUT_BUF: case MQE_T_R_TERM: case MQE_T_P_TERM:
    case MQE_T_Z_HIGH_TERM: SH(1, N, 0, 0, 0, 2); break;
    case MQE_T_OBS_BAG: SH(2, R, OBS_BAG, 0, 0, 0); break;
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
int mqo_hist_pos(mqo_sim* s) { return s->hist_pos; }

/* ------------------------------------------------------------------------------------------ row A/B: policy */
/* Go1.step head (go1.py:37-41) + preprocess_action (go1.py:64-108).  command: [R,3] */
int mqo_policy_step(mqo_sim* s, const float* command) {
  int R = s->R;
  const mqe_sim_desc* d = &s->d;
  for (int i = 0; i < R; i++) {
    float* lo = s->loco_obs + (size_t)i * FR;
    const float* ob = s->obs_bag + (size_t)i * OBS_BAG;
    int general = d->num_command_dims != 3;
    for (int c = 0; c < 18; c++) if (d->command_src[c] != (c >= 3 && c < 6 ? c - 3 : -1)) general = 1;
    if (!general) {
      float c[3];
      for (int k = 0; k < 3; k++) {
        c[k] = command[i * 3 + k];
        if (d->clip_command) c[k] = fminf(fmaxf(c[k], -1.0f), 1.0f);   /* go1.py:38 */
      }
      lo[3] = c[0] * d->cmd_lin_scale; lo[4] = c[1] * d->cmd_lin_scale; lo[5] = c[2] * d->cmd_ang_scale; /* :67-68 */
    } else {                                                       /* command.cfg beyond / without the velocity command: go1.py:66-93 */
      for (int c = 3; c < 18; c++) {
        const int src = d->command_src[c];
        if (src < 0) { lo[c] = d->command_obs[c]; continue; }
        float x = command[(size_t)i * d->num_command_dims + src];
        if (d->clip_command) x = fminf(fmaxf(x, -1.0f), 1.0f);
        lo[c] = x * d->command_scale[c];
      }
    }
    for (int k = 0; k < 3; k++) lo[k] = ob[60 + k];              /* projected gravity  :95 */
    for (int k = 0; k < 12; k++) lo[18 + k] = ob[6 + k];         /* dof_pos            :96 */
    for (int k = 0; k < 12; k++) lo[30 + k] = ob[18 + k];        /* dof_vel            :97 */
    for (int k = 0; k < 12; k++) lo[42 + k] = s->last_loco[i * 12 + k];      /* :98 */
    for (int k = 0; k < 12; k++) lo[54 + k] = s->last_two_loco[i * 12 + k];  /* :99 */
    for (int k = 0; k < 4; k++) lo[66 + k] = ob[63 + k];         /* clock inputs       :100 */
    memcpy(s->hist + ((size_t)i * MQE_HIST + s->hist_pos) * FR, lo, FR * 4);  /* :102 (ring write) */
  }
  s->hist_pos = (s->hist_pos + 1) % MQE_HIST;
#pragma omp parallel for schedule(static) num_threads(mqo_team())
  for (int i = 0; i < R; i++) {
    float h[2100], lat[2], a12[12];
    gather_history(s, i, h);
    policy_forward(s, h, lat, a12);                               /* :104 */
    for (int k = 0; k < 12; k++) {
      s->last_two_loco[i * 12 + k] = s->last_loco[i * 12 + k];    /* :106 */
      s->last_loco[i * 12 + k] = a12[k];                          /* :107 */
      s->actions[i * 12 + k] = fminf(fmaxf(a12[k], -d->clip_actions), d->clip_actions); /* :40-41 */
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------ row E/F/G: torques */
int mqo_compute_torques(mqo_sim* s) {
  const mqe_sim_desc* d = &s->d;
  int R = s->R, A = s->A;
  float* e1 = s->act_hist; float* e2 = e1 + (size_t)R * 12; float* v1 = e2 + (size_t)R * 12; float* v2 = v1 + (size_t)R * 12;
#pragma omp parallel for schedule(static) num_threads(mqo_team())
  for (int i = 0; i < R; i++) {
    int env = i / A, a = i % A;
    for (int j = 0; j < 12; j++) {
      float act = s->actions[i * 12 + j];
      float q = s->dof[((size_t)env * s->ND + a * 12 + j) * 2], qd = s->dof[((size_t)env * s->ND + a * 12 + j) * 2 + 1];
      float as = act * d->action_scale;                                 /* go1.py:329 */
      float tau;
      if (d->control_type == MQE_CTRL_C) {
        if (j % 3 == 0) as *= d->hip_scale_reduction;                     /* :331 */
        if (s->lag_buf) {                                                  /* :337-339: buffer = buffer[1:] + [as]; target from buffer[0] */
          int n = d->lag_timesteps + 1, rd = s->lag_pos + 1 >= n ? 0 : s->lag_pos + 1;
          s->lag_buf[(size_t)s->lag_pos * R * 12 + i * 12 + j] = as;
          as = s->lag_buf[(size_t)rd * R * 12 + i * 12 + j];
        }
        float target = as + d->default_dof_pos[j];                         /* :341 */
        float err = q - target;                                            /* :343 */
        float x[6] = {err, e1[i * 12 + j], e2[i * 12 + j], qd, v1[i * 12 + j], v2[i * 12 + j]};
        tau = actuator_net(&s->act, x);                                    /* :345 */
        e2[i * 12 + j] = e1[i * 12 + j]; e1[i * 12 + j] = err;             /* :347-348 */
        v2[i * 12 + j] = v1[i * 12 + j]; v1[i * 12 + j] = qd;              /* :349-350 */
      } else if (d->control_type == MQE_CTRL_P) {
        tau = d->kp * (as + d->default_dof_pos[j] - q) - d->kd * qd;       /* legged_robot.py:385 */
      } else if (d->control_type == MQE_CTRL_V) {                          /* :387: velocity targets, D term on the change per policy step */
        tau = d->kp * (as - qd) - d->kd * (qd - s->last_dof_vel[i * 12 + j]) / d->dt;
      } else if (d->control_type == MQE_CTRL_T) {
        tau = as;                                                          /* :389 */
      } else {
        tau = 0;
      }
      float lim = d->torque_limits[j];
      s->torques[i * 12 + j] = fminf(fmaxf(tau, -lim), lim);               /* go1.py:352 */
    }
  }
  if (s->lag_buf) s->lag_pos = (s->lag_pos + 1) % (d->lag_timesteps + 1);
  return 0;
}

int mqo_post_decimation_step(mqo_sim* s, int dec_i) { /* legged_robot.py:112-115 */
  int n = 12 * s->A;
  for (int e = 0; e < s->N; e++) {
    memcpy(s->sub_tau + ((size_t)e * 4 + dec_i) * n, s->torques + (size_t)e * n, n * 4);                     /* :113 */
    for (int jt = 0; jt < n; jt++) {
      const float q = s->dof[((size_t)e * s->ND + jt) * 2], qd = s->dof[((size_t)e * s->ND + jt) * 2 + 1];
      s->sub_dof_vel[((size_t)e * 4 + dec_i) * n + jt] = qd;                                                  /* :114 */
      s->sub_exceed[((size_t)e * 4 + dec_i) * n + jt] = (q < s->soft_lo[jt % 12]) | (q > s->soft_hi[jt % 12]);  /* :115 */
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------ row H: physics */
typedef struct {
  real R[9], p[3], c[3], Iw[9], w[3], vp[3], a[3], al[3], ap[3];
} bodyk_t;

typedef struct {
  int kind;           /* 0 terrain, 1 two actors, 2 plank */
  int actA, actB;     /* actor index in env: 0..A-1 robots, A.. npcs; actB=-1 static */
  int bodyA, bodyB;   /* touching link of a robot side (0 = base; 0 for NPC sides) */
  int repA, repB;     /* reported rigid body of each side (index into the env's net-contact-force rows) */
  real p[3], n[3], t1[3], t2[3], sd;
  real J[3][MAXDOF], B[3][MAXDOF], K[3][3], lam[3];
} contact_t;

static void sym6_to_mat(const float* s6, real* I) {
  I[0] = s6[0]; I[4] = s6[1]; I[8] = s6[2]; I[1] = I[3] = s6[3]; I[2] = I[6] = s6[4]; I[5] = I[7] = s6[5];
}

/* bilinear sample of a terrain map (wall SDF / ground relief) at world (x,y) + gradient.  Raster entry (i, j) sits at the world point
 * (i hs, j hs): upstream hands PhysX convert_heightfield_to_trimesh(heightfield, hs, vs, slope_treshold) (barrier_track.py:483-491,
 * isaacgym.terrain_utils -- third party, not in the snapshot: vertices on linspace(0, (n-1) hs, n)) placed at the track's pixel origin
 * times hs (:492-497).  With the field configs' slope_treshold = 100 (legged_robot_field_config.py:13) no slope is corrected to a
 * vertical face, so a raised pixel is a frustum whose flanks run from the neighbouring low vertices to its own: the wall prism of
 * the signed-distance map (pixel = square centred on its vertex) cuts those flanks at mid height. */
static real map_sample(const mqo_sim* s, const float* map, real x, real y, real* gx, real* gy) {
  const mqe_sim_desc* d = &s->d;
  real hs = d->horizontal_scale;
  real fx = x / hs, fy = y / hs;                          /* samples sit at the raster's vertices */
  int nx = d->sdf_nx, ny = d->sdf_ny;
  if (!(fx >= 0)) fx = 0; if (!(fy >= 0)) fy = 0;     /* also catches NaN (a diverged state must not index out of the map) */
  if (fx > nx - 1) fx = (real)(nx - 1); if (fy > ny - 1) fy = (real)(ny - 1);
  int ix = (int)fx, iy = (int)fy;
  if (ix > nx - 2) ix = nx - 2; if (iy > ny - 2) iy = ny - 2;
  real tx = fx - ix, ty = fy - iy;
  real s00 = map[(size_t)ix * ny + iy], s01 = map[(size_t)ix * ny + iy + 1];
  real s10 = map[(size_t)(ix + 1) * ny + iy], s11 = map[(size_t)(ix + 1) * ny + iy + 1];
  real a0 = s00 + (s01 - s00) * ty, a1 = s10 + (s11 - s10) * ty;
  *gx = (a1 - a0) / hs;
  *gy = ((s01 - s00) + ((s11 - s10) - (s01 - s00)) * tx) / hs;
  return a0 + (a1 - a0) * tx;
}
static real sdf_sample(const mqo_sim* s, real x, real y, real* gx, real* gy) { return map_sample(s, s->sdf, x, y, gx, gy); }
/* top of the wall next to (x, y): one height per scene, or -- walls of different heights (barrier_track.py:167-173: a (lo, hi)
 * wall_height draws one per block) -- the value stored at the nearest raster vertex = the height of the wall nearest to it */
static real wall_top_at(const mqo_sim* s, real x, real y) {
  const mqe_sim_desc* d = &s->d;
  if (!s->wall_top) return d->wall_height;
  real hs = d->horizontal_scale;
  real fx = x / hs, fy = y / hs;
  int nx = d->sdf_nx, ny = d->sdf_ny;
  if (!(fx >= 0)) fx = 0; if (!(fy >= 0)) fy = 0;
  if (fx > nx - 1) fx = (real)(nx - 1); if (fy > ny - 1) fy = (real)(ny - 1);
  int ix = (int)fx, iy = (int)fy;
  if (ix > nx - 2) ix = nx - 2; if (iy > ny - 2) iy = ny - 2;
  real tx = fx - ix, ty = fy - iy;
  return s->wall_top[(size_t)(tx < (real)0.5 ? ix : ix + 1) * ny + (ty < (real)0.5 ? iy : iy + 1)];
}

/* sphere (centre c, radius r) vs box (centre bc, rotation R row-major, half extents h): signed distance and world
 * normal pointing from the box to the sphere */
/* sphere (centre c, radius r) vs upright solid cylinder (centre bc, radius rc, half height hh): signed distance, normal cylinder->sphere */
static real sphere_vcyl(const real* c, real r, const real* bc, real rc, real hh, real* n) {
  real dx = c[0] - bc[0], dy = c[1] - bc[1], dz = c[2] - bc[2];
  real rho = (real)sqrt((double)(dx * dx + dy * dy));
  real ux = rho > (real)1e-9 ? dx / rho : 1, uy = rho > (real)1e-9 ? dy / rho : 0;
  real er = rho - rc, ez = (dz < 0 ? -dz : dz) - hh, sz = dz < 0 ? (real)-1 : (real)1;
  if (er <= 0 && ez <= 0) {                /* centre inside: leave through the nearer surface */
    if (er > ez) { n[0] = ux; n[1] = uy; n[2] = 0; return er - r; }
    n[0] = 0; n[1] = 0; n[2] = sz; return ez - r;
  }
  real pr = er > 0 ? er : 0, pz = ez > 0 ? ez : 0;
  real dist = (real)sqrt((double)(pr * pr + pz * pz));
  n[0] = ux * pr / dist; n[1] = uy * pr / dist; n[2] = sz * pz / dist;
  return dist - r;
}
static real sphere_box(const real* c, real r, const real* bc, const real* R, const real* h, real* n) {
  real d[3] = {c[0] - bc[0], c[1] - bc[1], c[2] - bc[2]}, pl[3], q[3], dl[3];
  for (int k = 0; k < 3; k++) pl[k] = R[k] * d[0] + R[3 + k] * d[1] + R[6 + k] * d[2];     /* R^T d */
  int inside = 1;
  for (int k = 0; k < 3; k++) {
    q[k] = pl[k] < -h[k] ? -h[k] : (pl[k] > h[k] ? h[k] : pl[k]);
    dl[k] = pl[k] - q[k];
    if (dl[k] != 0) inside = 0;
  }
  real nl[3] = {0, 0, 0}, sd;
  if (!inside) {
    real dist = (real)sqrt((double)(dl[0] * dl[0] + dl[1] * dl[1] + dl[2] * dl[2]));
    for (int k = 0; k < 3; k++) nl[k] = dl[k] / dist;
    sd = dist - r;
  } else {
    int ax = 0; real best = h[0] - (real)fabs((double)pl[0]);
    for (int k = 1; k < 3; k++) { real m = h[k] - (real)fabs((double)pl[k]); if (m < best) { best = m; ax = k; } }
    nl[ax] = pl[ax] >= 0 ? (real)1 : (real)-1;
    sd = -best - r;
  }
  mat3_vec(R, nl, n);
  return sd;
}

/* Closest approach of the segment p0 + t (p1 - p0), t in [0, 1], swept by the radius r, to a box: the signed distance of a point to a
 * convex set is convex along a line, so a golden-section search finds its minimum -- two first evaluations and sixteen refinements (the bracket ends at 0.05 % of the segment), the
 * same sequence in the engine (csrc/kernels_physics.hpp seg_box).  Returns the signed distance at the final t, *tb = that t, n = the
 * normal from the box to the point. */
#define SEGBOX_ITERS 16
static real seg_box(const real* p0, const real* p1, real r, const real* bc, const real* R, const real* h, real* tb, real* n, real* pt) {
  const real gr = (real)0.6180339887498949;
  real a = 0, b = 1, nn[3], x[3];
  real c = b - gr * (b - a), dd = a + gr * (b - a);
  for (int k = 0; k < 3; k++) x[k] = p0[k] + c * (p1[k] - p0[k]);
  real fc = sphere_box(x, r, bc, R, h, nn);
  for (int k = 0; k < 3; k++) x[k] = p0[k] + dd * (p1[k] - p0[k]);
  real fd = sphere_box(x, r, bc, R, h, nn);
  for (int it = 0; it < SEGBOX_ITERS; it++) {
    if (fc < fd) { b = dd; dd = c; fd = fc; c = b - gr * (b - a); for (int k = 0; k < 3; k++) x[k] = p0[k] + c * (p1[k] - p0[k]); fc = sphere_box(x, r, bc, R, h, nn); }
    else { a = c; c = dd; fc = fd; dd = a + gr * (b - a); for (int k = 0; k < 3; k++) x[k] = p0[k] + dd * (p1[k] - p0[k]); fd = sphere_box(x, r, bc, R, h, nn); }
  }
  real t = (real)0.5 * (a + b);
  for (int k = 0; k < 3; k++) pt[k] = p0[k] + t * (p1[k] - p0[k]);
  *tb = t;
  return sphere_box(pt, r, bc, R, h, n);
}
/* the twelve edges of a box (centre bc, rotation R, half extents h): edge e = axis e / 4, the two other coordinates' signs from bits 0, 1 */
static void box_edge(const real* bc, const real* R, const real* h, int e, real* p0, real* p1) {
  int ax = e >> 2, o1 = (ax + 1) % 3, o2 = (ax + 2) % 3;
  real l0[3], l1[3];
  l0[ax] = -h[ax]; l1[ax] = h[ax];
  l0[o1] = l1[o1] = (e & 1) ? h[o1] : -h[o1];
  l0[o2] = l1[o2] = (e & 2) ? h[o2] : -h[o2];
  mat3_vec(R, l0, p0); mat3_vec(R, l1, p1);
  for (int k = 0; k < 3; k++) { p0[k] += bc[k]; p1[k] += bc[k]; }
}

static void make_tangents(const real* n, real* t1, real* t2) {
  real a[3] = {0, 0, 1};
  if (fabs((double)n[2]) > 0.7) { a[0] = 1; a[2] = 0; }
  cross3(a, n, t1);
  real l = (real)sqrt((double)dot3(t1, t1));
  t1[0] /= l; t1[1] /= l; t1[2] /= l;
  cross3(n, t1, t2);
}

static int chol(real* Aa, int n, int ld) { /* in place lower Cholesky */
  for (int j = 0; j < n; j++) {
    real d = Aa[j * ld + j];
    for (int k = 0; k < j; k++) d -= Aa[j * ld + k] * Aa[j * ld + k];
    if (d <= 0) return -1;
    d = (real)sqrt((double)d);
    Aa[j * ld + j] = d;
    for (int i = j + 1; i < n; i++) {
      real v = Aa[i * ld + j];
      for (int k = 0; k < j; k++) v -= Aa[i * ld + k] * Aa[j * ld + k];
      Aa[i * ld + j] = v / d;
    }
  }
  return 0;
}
static void chol_solve(const real* L, int n, int ld, real* b) {
  for (int i = 0; i < n; i++) {
    real v = b[i];
    for (int k = 0; k < i; k++) v -= L[i * ld + k] * b[k];
    b[i] = v / L[i * ld + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    real v = b[i];
    for (int k = i + 1; k < n; k++) v -= L[k * ld + i] * b[k];
    b[i] = v / L[i * ld + i];
  }
}

typedef struct {
  bodyk_t bk[MAXA][NB];
  real L[MAXA][RD * RD];       /* Cholesky factor of each robot's mass matrix */
  real sph_c[MAXA + MAXP][MQE_MAX_SPHERES][3];     /* feature points (robots) / collision spheres (free NPCs), world frame */
  real sph_r[MAXA + MAXP][MQE_MAX_SPHERES];
  int sph_n[MAXA + MAXP];
  real prim_c[MAXA][MQE_MAX_PRIMS][3], prim_u[MAXA][MQE_MAX_PRIMS][3];   /* robots' primitives: centre, capsule half-segment (world) */
  real npcR[MAXP][9];
  real v[MAXDOF], tau[MAXDOF];
  contact_t con[MAXC + 320];          /* + room for a robot's one-sided candidates before their reduction to CAP_ROBOT */
  int nc;
} envwork_t;

static int is_ancestor_or_self(const int* parent, int anc, int b) {
  while (b >= 0) { if (b == anc) return 1; b = parent[b]; }
  return 0;
}
static const int g_parent[NB] = {-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11};

/* point-velocity Jacobian column block of actor `act` for a point p rigidly attached to (robot body b | npc) */
static void fill_jac(const mqo_sim* s, const envwork_t* w, int act, int body, const real* p, real sign, const real dirs[3][3],
                     real J[3][MAXDOF], const real npc_pos[][3]) {
  int A = s->A;
  const int lin_only = s->d.npc_kind == MQE_NPC_SHEEP;
  if (act < A) {
    int o = act * RD;
    const bodyk_t* bk = w->bk[act];
    for (int k = 0; k < 3; k++) {
      real e[3] = {0, 0, 0}; e[k] = 1;
      real r[3] = {p[0] - bk[0].p[0], p[1] - bk[0].p[1], p[2] - bk[0].p[2]};
      real wv[3]; cross3(e, r, wv);
      for (int q = 0; q < 3; q++) { J[q][o + k] += sign * dirs[q][k]; J[q][o + 3 + k] += sign * dot3(dirs[q], wv); }
    }
    for (int j = 1; j < NB; j++) {
      if (!is_ancestor_or_self(g_parent, j, body)) continue;
      real r[3] = {p[0] - bk[j].p[0], p[1] - bk[j].p[1], p[2] - bk[j].p[2]};
      real wv[3]; cross3(bk[j].a, r, wv);
      for (int q = 0; q < 3; q++) J[q][o + 6 + (j - 1)] += sign * dot3(dirs[q], wv);
    }
  } else if (s->d.npc_kind == MQE_NPC_SEESAW) {
    real ay[3] = {0, s->d.seesaw_axis == 2 ? 0 : 1, s->d.seesaw_axis == 2 ? 1 : 0};     /* joint axis: +y plank / slider, +z door */
    real r[3] = {p[0] - npc_pos[0][0], p[1] - npc_pos[0][1], p[2] - npc_pos[0][2]};
    real wv[3]; cross3(ay, r, wv);
    if (s->d.seesaw_axis == 3) { wv[0] = ay[0]; wv[1] = ay[1]; wv[2] = ay[2]; }         /* prismatic: the point moves with the axis */
    for (int q = 0; q < 3; q++) J[q][A * RD] += sign * dot3(dirs[q], wv);
  } else {
    int pi = act - A, o = A * RD + pi * 6;
    for (int k = 0; k < 3; k++) {
      real e[3] = {0, 0, 0}; e[k] = 1;
      real r[3] = {p[0] - npc_pos[pi][0], p[1] - npc_pos[pi][1], p[2] - npc_pos[pi][2]};
      real wv[3]; cross3(e, r, wv);
      for (int q = 0; q < 3; q++) { J[q][o + k] += sign * dirs[q][k]; if (!lin_only) J[q][o + 3 + k] += sign * dot3(dirs[q], wv); }
    }
  }
}

/* feature point (sphere: centre c, radius r) against primitive q of robot `rob` (go1.urdf <collision> shapes, include/mqe_hip.h):
 * signed distance and the unit normal from the primitive to the sphere.  Sphere / capsule: the closest point of the capsule's
 * segment centre +- u (u = 0: a sphere); box: the link-aligned box.  Returns 0 when the centre lies on the segment itself (no
 * direction), the caller skips the pair. */
static int feat_vs_prim(const mqe_robot_model* m, const envwork_t* w, int rob, int q, const real* c, real r, real* sd, real* n) {
  const real* cq = w->prim_c[rob][q];
  if (m->prim_type[q] == MQE_PRIM_BOX) {
    real hb[3] = {m->prim_half[q][0], m->prim_half[q][1], m->prim_half[q][2]};
    *sd = sphere_box(c, r, cq, w->bk[rob][m->prim_body[q]].R, hb, n);
    return 1;
  }
  const real* u = w->prim_u[rob][q];
  real dq[3] = {c[0] - cq[0], c[1] - cq[1], c[2] - cq[2]};
  real uu = dot3(u, u), t = 0;
  if (uu > 0) { t = dot3(dq, u) / uu; if (t < -1) t = -1; if (t > 1) t = 1; }
  real e[3] = {dq[0] - t * u[0], dq[1] - t * u[1], dq[2] - t * u[2]};
  real dist = (real)sqrt((double)dot3(e, e));
  if (!(dist > (real)1e-9)) return 0;
  *sd = dist - r - m->prim_half[q][0];
  n[0] = e[0] / dist; n[1] = e[1] / dist; n[2] = e[2] / dist;
  return 1;
}

/* Edge contact between primitive q of robot `rob` and a convex box (centre bc, rotation R, half extents h; a wall's vertical edge is the
 * degenerate box h = (0, 0, L / 2) with n_edges = 1: its own axis): a capsule's AXIS against the box (mask bit 2), the box's edges against
 * a box primitive (mask bit 4) -- the deepest edge.  Kept when the closest approach lies between the segment's end points (5 .. 95 %:
 * the ends are feature points of the robot resp. corners).  Returns 1 with sd, n (from the obstacle to the robot), pa (the point on the
 * robot's side: the capsule's surface point resp. the edge point). */
static int edge_vs_box(const mqo_sim* s, const envwork_t* w, int rob, int q, const real* bc, const real* R, const real* h, int n_edges, int mask, real* sd, real* n, real* pa) {
  const mqe_robot_model* m = &s->d.robot;
  const real* cq = w->prim_c[rob][q];
  if (m->prim_type[q] == MQE_PRIM_CAPSULE) {
    const real* u = w->prim_u[rob][q];
    if (!(mask & 2) || dot3(u, u) <= 0) return 0;
    real p0[3] = {cq[0] - u[0], cq[1] - u[1], cq[2] - u[2]}, p1[3] = {cq[0] + u[0], cq[1] + u[1], cq[2] + u[2]}, tb, nn[3], pt[3];
    real r = m->prim_half[q][0];
    real v = seg_box(p0, p1, r, bc, R, h, &tb, nn, pt);
    if (!(tb > (real)0.05 && tb < (real)0.95)) return 0;
    for (int f = 0; f < m->n_spheres; f++)          /* ... nor next to a feature point that sits ON the axis (the thigh's middle) */
      if (m->sphere_prim[f] == q) {
        real df[3] = {m->sphere_center[f][0] - m->prim_center[q][0], m->sphere_center[f][1] - m->prim_center[q][1], m->sphere_center[f][2] - m->prim_center[q][2]};
        real ax[3] = {m->prim_axis[q][0], m->prim_axis[q][1], m->prim_axis[q][2]};
        real tf = (real)0.5 + (real)0.5 * dot3(df, ax) / dot3(ax, ax);
        if (tb > tf - (real)0.1 && tb < tf + (real)0.1) return 0;
      }
    if (v + r < 0) return 0;      /* the axis itself is inside the box: a link that has tunnelled through a thin plate has no meaningful normal here */
    *sd = v;
    for (int k = 0; k < 3; k++) { n[k] = nn[k]; pa[k] = pt[k] - r * nn[k]; }
    return 1;
  }
  if (m->prim_type[q] != MQE_PRIM_BOX || !(mask & 4)) return 0;
  real hb[3] = {m->prim_half[q][0], m->prim_half[q][1], m->prim_half[q][2]};
  const real* Rb = w->bk[rob][m->prim_body[q]].R;
  int found = 0;
  for (int e = 0; e < n_edges; e++) {
    real p0[3], p1[3], tb, nn[3], pt[3];
    if (n_edges == 1) { for (int k = 0; k < 3; k++) { p0[k] = bc[k] - (R[k * 3 + 2] * h[2]); p1[k] = bc[k] + (R[k * 3 + 2] * h[2]); } }
    else box_edge(bc, R, h, e, p0, p1);
    real v = seg_box(p0, p1, (real)0, cq, Rb, hb, &tb, nn, pt);
    if (n_edges > 1 && !(tb > (real)0.05 && tb < (real)0.95)) continue;
    if (v < (real)-0.02) continue;  /* an edge deeper than 2 cm inside the primitive: tunnelled, left to the feature points */
    if (!found || v < *sd) { found = 1; *sd = v; for (int k = 0; k < 3; k++) { n[k] = -nn[k]; pa[k] = pt[k]; } }
  }
  return found;
}

static void simulate_env(mqo_sim* s, int env, envwork_t* w) {
  const mqe_sim_desc* d = &s->d;
  const mqe_robot_model* m = &d->robot;
  int A = s->A, P = (d->npc_kind == MQE_NPC_BALL || d->npc_kind == MQE_NPC_SHEEP || d->npc_kind == MQE_NPC_BOX) ? s->P : 0;
  const int cap_npc = d->npc_contact_cap > 0 ? d->npc_contact_cap : 2;
  const int maxc = env_maxc(A, s->P, cap_npc);
  const int BOX = d->npc_kind == MQE_NPC_BOX;     /* robots' spheres vs the oriented box; its corners vs the terrain */
  /* sheep: translation-only bodies (orientation is scripted: go1_sheep.py:61 zeroes quat x,y every step) */
  const int lin_only = d->npc_kind == MQE_NPC_SHEEP;
  real dt = d->dt;
  float* root = s->root + (size_t)env * (A + s->P) * 13;
  float* dofs = s->dof + (size_t)env * s->ND * 2;
  const int SS = d->npc_kind == MQE_NPC_SEESAW;   /* fixed base + 1-dof plank (seesaw.urdf) */
  const int sdof = A * RD;
  int ndof = A * RD + P * 6 + SS;
  real g[3] = {0, 0, d->gravity_z};
  real ssR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, ssC[3] = {0, 0, 0}, ssB[3] = {0, 0, 0}, ssTheta = 0;
  real npc_pos[MAXP][3];

  /* ---- forward kinematics, mass matrix, bias, unconstrained velocity per robot */
  for (int r = 0; r < A; r++) {
    bodyk_t* bk = w->bk[r];
    const float* rs = root + r * 13;
    real q[4] = {rs[3], rs[4], rs[5], rs[6]};
    {  /* a reset copies the configured quaternion verbatim and go1_wrestling_config.py:68,74 gives (0,0,-+1,1): the physics
        * reads it normalised (the integrator writes unit quaternions from then on) */
      real nq = (real)sqrt((double)(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]));
      for (int k = 0; k < 4; k++) q[k] /= nq;
    }
    quat_to_mat(q, bk[0].R);
    for (int k = 0; k < 3; k++) { bk[0].p[k] = rs[k]; bk[0].vp[k] = rs[7 + k]; bk[0].w[k] = rs[10 + k]; bk[0].al[k] = 0; bk[0].ap[k] = 0; bk[0].a[k] = 0; }
    for (int b = 1; b < NB; b++) {
      int pb = g_parent[b];
      real qj = dofs[(r * 12 + b - 1) * 2], qdj = dofs[(r * 12 + b - 1) * 2 + 1];
      real off[3] = {m->joint_offset[b][0], m->joint_offset[b][1], m->joint_offset[b][2]}, dd[3];
      mat3_vec(bk[pb].R, off, dd);
      for (int k = 0; k < 3; k++) bk[b].p[k] = bk[pb].p[k] + dd[k];
      real ax[3] = {m->joint_axis[b][0], m->joint_axis[b][1], m->joint_axis[b][2]}, Rj[9];
      axis_angle_mat(ax, qj, Rj);
      mat3_mul(bk[pb].R, Rj, bk[b].R);
      mat3_vec(bk[pb].R, ax, bk[b].a);
      real t[3], t2[3];
      for (int k = 0; k < 3; k++) bk[b].w[k] = bk[pb].w[k] + bk[b].a[k] * qdj;
      cross3(bk[pb].w, dd, t);
      for (int k = 0; k < 3; k++) bk[b].vp[k] = bk[pb].vp[k] + t[k];
      /* bias accelerations (all generalized accelerations zero) */
      real aq[3] = {bk[b].a[0] * qdj, bk[b].a[1] * qdj, bk[b].a[2] * qdj};
      cross3(bk[pb].w, aq, t);
      for (int k = 0; k < 3; k++) bk[b].al[k] = bk[pb].al[k] + t[k];
      cross3(bk[pb].al, dd, t);
      real wd[3]; cross3(bk[pb].w, dd, wd); cross3(bk[pb].w, wd, t2);
      for (int k = 0; k < 3; k++) bk[b].ap[k] = bk[pb].ap[k] + t[k] + t2[k];
    }
    for (int b = 0; b < NB; b++) {
      /* domain parameters: added base mass and base CoM shift act on body 0 only, inertia about the CoM unchanged */
      const float* dp = s->dparams + ((size_t)env * A + r) * 8;
      real cl[3] = {m->com[b][0] + (b == 0 ? dp[2] : 0), m->com[b][1] + (b == 0 ? dp[3] : 0), m->com[b][2] + (b == 0 ? dp[4] : 0)}, cw[3], Il[9], T[9], Rt[9];
      mat3_vec(bk[b].R, cl, cw);
      for (int k = 0; k < 3; k++) bk[b].c[k] = bk[b].p[k] + cw[k];
      sym6_to_mat(m->inertia[b], Il);
      mat3_mul(bk[b].R, Il, T);
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rt[i * 3 + j] = bk[b].R[j * 3 + i];
      mat3_mul(T, Rt, bk[b].Iw);
    }
    /* M = sum_b m Jv^T Jv + Jw^T I Jw ; h = sum_b Jv.f + Jw.n */
    real M[RD * RD], h[RD];
    memset(M, 0, sizeof M); memset(h, 0, sizeof h);
    for (int b = 0; b < NB; b++) {
      real Jv[RD][3], Jw[RD][3];
      memset(Jv, 0, sizeof Jv); memset(Jw, 0, sizeof Jw);
      real rc0[3] = {bk[b].c[0] - bk[0].p[0], bk[b].c[1] - bk[0].p[1], bk[b].c[2] - bk[0].p[2]};
      for (int k = 0; k < 3; k++) {
        Jv[k][k] = 1;
        real e[3] = {0, 0, 0}; e[k] = 1;
        Jw[3 + k][k] = 1;
        cross3(e, rc0, Jv[3 + k]);
      }
      for (int j = 1; j < NB; j++) {
        if (!is_ancestor_or_self(g_parent, j, b)) continue;
        real rr[3] = {bk[b].c[0] - bk[j].p[0], bk[b].c[1] - bk[j].p[1], bk[b].c[2] - bk[j].p[2]};
        for (int k = 0; k < 3; k++) Jw[6 + j - 1][k] = bk[j].a[k];
        cross3(bk[j].a, rr, Jv[6 + j - 1]);
      }
      real mb = m->mass[b] + (b == 0 ? s->dparams[((size_t)env * A + r) * 8 + 1] : 0);
      for (int i = 0; i < RD; i++) {
        real IJ[3]; mat3_vec(bk[b].Iw, Jw[i], IJ);
        for (int j = 0; j < RD; j++) M[i * RD + j] += mb * dot3(Jv[i], Jv[j]) + dot3(IJ, Jw[j]);
      }
      real rc[3] = {bk[b].c[0] - bk[b].p[0], bk[b].c[1] - bk[b].p[1], bk[b].c[2] - bk[b].p[2]};
      real t[3], t2[3], wr[3], ac[3], f[3], nn[3], Iw_[3], Ial[3];
      cross3(bk[b].al, rc, t); cross3(bk[b].w, rc, wr); cross3(bk[b].w, wr, t2);
      for (int k = 0; k < 3; k++) { ac[k] = bk[b].ap[k] + t[k] + t2[k]; f[k] = mb * (ac[k] - g[k]); }
      mat3_vec(bk[b].Iw, bk[b].w, Iw_); cross3(bk[b].w, Iw_, t); mat3_vec(bk[b].Iw, bk[b].al, Ial);
      for (int k = 0; k < 3; k++) nn[k] = Ial[k] + t[k];
      for (int i = 0; i < RD; i++) h[i] += dot3(Jv[i], f) + dot3(Jw[i], nn);
    }
    memcpy(w->L[r], M, sizeof M);
    if (chol(w->L[r], RD, RD) != 0) { fprintf(stderr, "mqe_oracle: mass matrix not SPD (env %d robot %d)\n", env, r); }
    real rhs[RD];
    for (int i = 0; i < 6; i++) rhs[i] = -h[i];
    for (int j = 0; j < 12; j++) rhs[6 + j] = (real)s->torques[(size_t)env * 12 * A + r * 12 + j] - h[6 + j];
    chol_solve(w->L[r], RD, RD, rhs);
    real* v = w->v + r * RD;
    for (int k = 0; k < 3; k++) { v[k] = bk[0].vp[k] + dt * rhs[k]; v[3 + k] = bk[0].w[k] + dt * rhs[3 + k]; }
    for (int j = 0; j < 12; j++) v[6 + j] = (real)dofs[(r * 12 + j) * 2 + 1] + dt * rhs[6 + j];
    /* collision spheres */
    w->sph_n[r] = m->n_spheres;
    for (int si = 0; si < m->n_spheres; si++) {
      int b = m->sphere_body[si];
      real cl[3] = {m->sphere_center[si][0], m->sphere_center[si][1], m->sphere_center[si][2]}, cw[3];
      mat3_vec(bk[b].R, cl, cw);
      for (int k = 0; k < 3; k++) w->sph_c[r][si][k] = bk[b].p[k] + cw[k];
      w->sph_r[r][si] = m->sphere_radius[si];
    }
    for (int q = 0; q < m->n_prims; q++) {
      int b = m->prim_body[q];
      real cl[3] = {m->prim_center[q][0], m->prim_center[q][1], m->prim_center[q][2]}, ul[3] = {m->prim_axis[q][0], m->prim_axis[q][1], m->prim_axis[q][2]}, cw[3];
      mat3_vec(bk[b].R, cl, cw);
      for (int k = 0; k < 3; k++) w->prim_c[r][q][k] = bk[b].p[k] + cw[k];
      mat3_vec(bk[b].R, ul, w->prim_u[r][q]);
    }
  }
  /* ---- free NPC bodies (ball / sheep): isotropic inertia => no gyroscopic term */
  for (int p = 0; p < P; p++) {
    const float* rs = root + (A + p) * 13;
    real q[4] = {rs[3], rs[4], rs[5], rs[6]};
    quat_to_mat(q, w->npcR[p]);
    real* v = w->v + A * RD + p * 6;
    for (int k = 0; k < 3; k++) { npc_pos[p][k] = rs[k]; v[k] = rs[7 + k] + dt * g[k]; v[3 + k] = rs[10 + k]; }
    w->sph_n[A + p] = d->npc_n_spheres;
    for (int si = 0; si < d->npc_n_spheres; si++) {
      real cl[3] = {d->npc_sphere_center[si][0], d->npc_sphere_center[si][1], d->npc_sphere_center[si][2]}, cw[3];
      mat3_vec(w->npcR[p], cl, cw);
      for (int k = 0; k < 3; k++) w->sph_c[A + p][si][k] = npc_pos[p][k] + cw[k];
      w->sph_r[A + p][si] = d->npc_sphere_radius[si];
    }
  }

  if (SS) {
    const float* rs = root + A * 13;
    for (int k = 0; k < 3; k++) { ssB[k] = rs[k]; npc_pos[0][k] = rs[k] + d->seesaw_joint_offset[k]; }   /* npc_pos[0] = hinge */
    ssTheta = dofs[(12 * A) * 2];
    real c = (real)cos((double)ssTheta), sn = (real)sin((double)ssTheta);
    if (d->seesaw_axis == 3) { c = 1; sn = 0; npc_pos[0][1] += ssTheta; }             /* slider: translation along +y, no rotation */
    if (d->seesaw_axis == 2) { ssR[0] = c; ssR[1] = -sn; ssR[3] = sn; ssR[4] = c; }   /* door: rotation about +z */
    else { ssR[0] = c; ssR[2] = sn; ssR[6] = -sn; ssR[8] = c; }                      /* plank: rotation about +y */
    real pc[3] = {d->seesaw_plank_center[0], d->seesaw_plank_center[1], d->seesaw_plank_center[2]}, pw[3];
    mat3_vec(ssR, pc, pw);
    for (int k = 0; k < 3; k++) ssC[k] = npc_pos[0][k] + pw[k];
    w->v[sdof] = dofs[(12 * A) * 2 + 1];                               /* COM on the hinge: no gravity torque, no drive */
  }

  /* ---- contact generation (canonical order: terrain contacts actor by actor, sphere by sphere, ground before
   * wall; then sphere pairs for actor pairs (a<b), outer loop over b's spheres, inner over a's) */
  int nact = A + P;
  int ovf = 0;                          /* a touching pair did not fit the bounded list (MQE_T_CONTACT_OVERFLOW) */
  int red = 0;                          /* a robot's one-sided contacts were reduced to the deepest CAP_ROBOT (MQE_T_CONTACT_REDUCED) */
  w->nc = 0;
  for (int act = 0; act < nact; act++) {
    int mine = 0;                       /* no actor may starve the ones after it */
    const int cap = act < A ? CAP_ROBOT : cap_npc;
    /* manifold reduction (round 6; desc.edge_contacts bit 8, off by default): a ROBOT's one-sided contacts are collected first -- in the canonical order: feature by feature (ground,
     * wall, platform, column), then the edge contacts primitive by primitive -- and when there are more than its slots the DEEPEST are kept:
     * penetrations of more than 1 mm first, the deepest 2 mm class first; everything shallower -- resting and speculative contacts -- ties (a body lying flat keeps the feature order's spread and the same set from substep to substep), ties in the canonical order (feet first).  The kept ones enter the list in the canonical order.  (Rounds
     * 1-5 kept the first CAP_ROBOT and counted the rest as overflow; an NPC's cap still works that way.)  The engine does the same with a
     * ranking over the wavefront's candidate lanes (kernels_physics.hpp "keeps the DEEPEST ones"). */
    const int robot_first = w->nc;
    const int robot_cap = cap;
    const int reduce = (d->edge_contacts & 8) != 0;      /* optional (include/mqe_hip.h edge_contacts bit 8); off: the first CAP_ROBOT in feature order, the rest is overflow */
    const int cap_eff = (act < A && reduce) ? MAXC + 320 : cap;      /* robots then collect without a cap and reduce */
#define cap cap_eff
    for (int si = 0; si < w->sph_n[act]; si++) {
      const real* c = w->sph_c[act][si];
      real r = w->sph_r[act][si];
      const int npass = act < A ? (SS ? 4 : (d->n_static_boxes > 0 ? 3 : 2)) : 2;
      for (int pass = 0; pass < npass; pass++) {
        real n[3], sd;
        if (pass == 0) {
          sd = c[2] - d->ground_z - r; n[0] = 0; n[1] = 0; n[2] = 1;
          if (s->ground_height) {      /* heightfield ground (Perlin relief): first-order distance to the surface along its normal */
            real hx, hy;
            real h = map_sample(s, s->ground_height, c[0], c[1], &hx, &hy);
            real inl = 1 / (real)sqrt((double)(hx * hx + hy * hy + 1));
            n[0] = -hx * inl; n[1] = -hy * inl; n[2] = inl;
            sd = (c[2] - d->ground_z - h) * inl - r;
          }
        }
        else if (pass == 2 && !SS) { /* static scenery: the world-aligned box with the smallest signed distance */
          real I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
          const float* nb = root + A * 13;
          sd = (real)1e3; n[0] = 0; n[1] = 0; n[2] = 1;
          for (int bx = 0; bx < d->n_static_boxes; bx++) {
            real bc[3] = {nb[0] + d->static_box_center[bx][0], nb[1] + d->static_box_center[bx][1], nb[2] + d->static_box_center[bx][2]};
            real hb[3] = {d->static_box_half[bx][0], d->static_box_half[bx][1], d->static_box_half[bx][2]}, nn[3];
            real sdb = sphere_box(c, r, bc, I3, hb, nn);
            if (sdb < sd) { sd = sdb; n[0] = nn[0]; n[1] = nn[1]; n[2] = nn[2]; }
          }
        }
        else if (pass == 2) {      /* seesaw platform: static axis-aligned box */
          real I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, hb[3] = {d->seesaw_base_half[0], d->seesaw_base_half[1], d->seesaw_base_half[2]};
          sd = hb[0] > 0 ? sphere_box(c, r, ssB, I3, hb, n) : (real)1e3;     /* no platform: tug-of-war slider */
        } else if (pass == 3) {    /* column under the platform: static vertical cylinder, lateral surface only */
          real dx = c[0] - ssB[0], dy = c[1] - ssB[1];
          real rho = (real)sqrt((double)(dx * dx + dy * dy));
          if (c[2] < ssB[2] && c[2] > ssB[2] - d->seesaw_column_length && rho > (real)1e-6) { sd = rho - d->seesaw_column_radius - r; n[0] = dx / rho; n[1] = dy / rho; n[2] = 0; }
          else { sd = (real)1e3; n[0] = 0; n[1] = 0; n[2] = 1; }
        }
        else {
          real gx, gy;
          real sh = sdf_sample(s, c[0], c[1], &gx, &gy);
          real gl = (real)sqrt((double)(gx * gx + gy * gy));
          if (gl < (real)1e-6) { gx = 1; gy = 0; gl = 1; }
          gx /= gl; gy /= gl;
          real dz = c[2] - wall_top_at(s, c[0], c[1]);
          if (dz <= 0) {               /* beside (or inside) the wall prism: lateral contact */
            if (sh <= 0 && -sh > -dz) { sd = dz - r; n[0] = 0; n[1] = 0; n[2] = 1; }  /* deep inside, closer to the top */
            else { sd = sh - r; n[0] = gx; n[1] = gy; n[2] = 0; }
          } else if (sh <= 0) {        /* above the top face */
            sd = dz - r; n[0] = 0; n[1] = 0; n[2] = 1;
          } else {                     /* near the top edge */
            real dist = (real)sqrt((double)(sh * sh + dz * dz));
            sd = dist - r; n[0] = gx * sh / dist; n[1] = gy * sh / dist; n[2] = dz / dist;
          }
        }
        if (sd < d->contact_offset && !(w->nc < ((act < A && reduce) ? MAXC + 320 : maxc) && mine < cap)) ovf = 1;
        if (sd < d->contact_offset && w->nc < ((act < A && reduce) ? MAXC + 320 : maxc) && mine < cap) {
          mine++;
          contact_t* ct = &w->con[w->nc++];
          memset(ct, 0, sizeof *ct);
          ct->kind = 0; ct->actA = act; ct->actB = -1; ct->sd = sd; ct->repB = -1;
          ct->bodyA = act < A ? m->sphere_body[si] : 0;
          ct->repA = act < A ? act * MQE_NREP + m->sphere_reported[si] : A * MQE_NREP + (act - A);
          for (int k = 0; k < 3; k++) { ct->n[k] = n[k]; ct->p[k] = c[k] - r * n[k]; }
        }
      }
    }
    /* edge contacts of a robot with the static world (include/mqe_hip.h edge_contacts), primitive by primitive after its feature points:
     * per primitive the deepest of the nearest vertical wall edge and the static scenery boxes */
    if (act < A && (d->edge_contacts & 7))
      for (int q = 0; q < m->n_prims; q++) {
        if (m->prim_type[q] == MQE_PRIM_SPHERE) continue;
        const real* cq = w->prim_c[act][q];
        real sd = (real)1e3, n[3] = {0, 0, 1}, pa[3] = {0, 0, 0};
        int got = 0;
        real I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        if ((d->edge_contacts & 1) && s->wall_corner) {
          real hs = d->horizontal_scale;
          int ix = (int)floor((double)(cq[0] / hs) + 0.5), iy = (int)floor((double)(cq[1] / hs) + 0.5);
          if (ix < 0) ix = 0; if (iy < 0) iy = 0; if (ix > d->sdf_nx - 1) ix = d->sdf_nx - 1; if (iy > d->sdf_ny - 1) iy = d->sdf_ny - 1;
          real cx = s->wall_corner[((size_t)ix * d->sdf_ny + iy) * 2], cy = s->wall_corner[((size_t)ix * d->sdf_ny + iy) * 2 + 1];
          real dxy2 = (cq[0] - cx) * (cq[0] - cx) + (cq[1] - cy) * (cq[1] - cy);
          real reach = m->prim_bound[q] + d->contact_offset;
          if (dxy2 < reach * reach) {          /* the edge passes the primitive's bounding sphere */
            real z0 = d->ground_z, z1 = wall_top_at(s, cx, cy);
            real bc[3] = {cx, cy, (real)0.5 * (z0 + z1)}, hh[3] = {0, 0, (real)0.5 * (z1 - z0)};
            got = edge_vs_box(s, w, act, q, bc, I3, hh, 1, 6, &sd, n, pa);
          }
        }
        if ((d->edge_contacts & 6) && d->n_static_boxes > 0 && !SS) {
          const float* nb = root + A * 13;
          for (int bx = 0; bx < d->n_static_boxes; bx++) {
            real bc[3] = {nb[0] + d->static_box_center[bx][0], nb[1] + d->static_box_center[bx][1], nb[2] + d->static_box_center[bx][2]};
            real hb[3] = {d->static_box_half[bx][0], d->static_box_half[bx][1], d->static_box_half[bx][2]}, sdb, nn[3], pp[3];
            if (edge_vs_box(s, w, act, q, bc, I3, hb, 12, d->edge_contacts, &sdb, nn, pp) && (!got || sdb < sd)) { got = 1; sd = sdb; for (int k = 0; k < 3; k++) { n[k] = nn[k]; pa[k] = pp[k]; } }
          }
        }
        if (!got) continue;
        if (sd < d->contact_offset && !(w->nc < ((act < A && reduce) ? MAXC + 320 : maxc) && mine < cap)) ovf = 1;
        if (sd < d->contact_offset && w->nc < ((act < A && reduce) ? MAXC + 320 : maxc) && mine < cap) {
          mine++;
          contact_t* ct = &w->con[w->nc++];
          memset(ct, 0, sizeof *ct);
          ct->kind = 0; ct->actA = act; ct->actB = -1; ct->sd = sd; ct->repB = -1;
          ct->bodyA = m->prim_body[q]; ct->repA = act * MQE_NREP + m->prim_reported[q];
          for (int k = 0; k < 3; k++) { ct->n[k] = n[k]; ct->p[k] = pa[k]; }
        }
      }
#undef cap
    if (act < A && reduce && mine > robot_cap) {
      int keep[MAXC + 320];
      for (int i = 0; i < mine; i++) {
        const float sdi = (float)w->con[robot_first + i].sd;
        const int bi = sdi < -1e-3f ? (int)floor((double)((sdi + 1e-3f) * 500.0f)) : 0;
        int rank = 0;
        for (int j = 0; j < mine; j++) {
          const float sdj = (float)w->con[robot_first + j].sd;
          const int bj = sdj < -1e-3f ? (int)floor((double)((sdj + 1e-3f) * 500.0f)) : 0;
          if (bj < bi || (bj == bi && j < i)) rank++;
        }
        keep[i] = rank < robot_cap;
      }
      int o = robot_first;
      for (int i = 0; i < mine; i++)
        if (keep[i]) { if (o != robot_first + i) w->con[o] = w->con[robot_first + i]; o++; }
      w->nc = o;
      red = 1;
    }
  }
  /* two-actor contacts (plank, sphere pairs): at most maxc/2 of them (the engine slot-allocates their second side) */
  const int pair_lim = w->nc + maxc / 2 < maxc ? w->nc + maxc / 2 : maxc;
  if (SS)   /* robot spheres vs the plank (dynamic: couples the robots through the hinge) */
    for (int act = 0; act < A; act++) {
      int mine = 0;                     /* per-robot share of the two-actor budget */
      for (int si = 0; si < w->sph_n[act]; si++) {
        const real* c = w->sph_c[act][si];
        real r = w->sph_r[act][si], n[3];
        real hp[3] = {d->seesaw_plank_half[0], d->seesaw_plank_half[1], d->seesaw_plank_half[2]};
        real sd = d->seesaw_link_cylinder ? sphere_vcyl(c, r, ssC, hp[0], hp[2], n) : sphere_box(c, r, ssC, ssR, hp, n);
        if (sd < d->contact_offset && !(w->nc < pair_lim && mine < (maxc / 2) / A)) ovf = 1;
        if (sd < d->contact_offset && w->nc < pair_lim && mine < (maxc / 2) / A) {
          mine++;
          contact_t* ct = &w->con[w->nc++];
          memset(ct, 0, sizeof *ct);
          ct->kind = 2; ct->actA = act; ct->actB = A; ct->sd = sd;
          ct->bodyA = m->sphere_body[si]; ct->repA = act * MQE_NREP + m->sphere_reported[si]; ct->bodyB = 0; ct->repB = A * MQE_NREP + 1;
          for (int k = 0; k < 3; k++) { ct->n[k] = n[k]; ct->p[k] = c[k] - r * n[k]; }
        }
      }
      /* ... and the plank's / door's edges against the robot's primitives between their feature points (edge_contacts bits 2, 4) */
      if ((d->edge_contacts & 6) && !d->seesaw_link_cylinder)
        for (int q = 0; q < m->n_prims; q++) {
          real hp[3] = {d->seesaw_plank_half[0], d->seesaw_plank_half[1], d->seesaw_plank_half[2]}, sd, n[3], pa[3];
          if (!edge_vs_box(s, w, act, q, ssC, ssR, hp, 12, d->edge_contacts, &sd, n, pa)) continue;
          if (sd < d->contact_offset && !(w->nc < pair_lim && mine < (maxc / 2) / A)) ovf = 1;
          if (sd < d->contact_offset && w->nc < pair_lim && mine < (maxc / 2) / A) {
            mine++;
            contact_t* ct = &w->con[w->nc++];
            memset(ct, 0, sizeof *ct);
            ct->kind = 2; ct->actA = act; ct->actB = A; ct->sd = sd;
            ct->bodyA = m->prim_body[q]; ct->repA = act * MQE_NREP + m->prim_reported[q]; ct->bodyB = 0; ct->repB = A * MQE_NREP + 1;
            for (int k = 0; k < 3; k++) { ct->n[k] = n[k]; ct->p[k] = pa[k]; }
          }
        }
    }
  for (int a = 0; a < nact; a++)
    for (int b = a + 1; b < nact; b++) {
      const real* pa = a < A ? w->bk[a][0].p : npc_pos[a - A];
      const real* pb = b < A ? w->bk[b][0].p : npc_pos[b - A];
      real dd[3] = {pa[0] - pb[0], pa[1] - pb[1], pa[2] - pb[2]};
      if (BOX && b >= A) {                              /* robot spheres vs the box (sphere_box narrow phase) */
        if (a >= A || dot3(dd, dd) > (real)(1.8 * 1.8)) continue;
        real hb[3] = {d->npc_box_half[0], d->npc_box_half[1], d->npc_box_half[2]};
        for (int sa = 0; sa < w->sph_n[a]; sa++) {
          const real* ca = w->sph_c[a][sa];
          real n[3];
          real sd = sphere_box(ca, w->sph_r[a][sa], npc_pos[b - A], w->npcR[b - A], hb, n);
          if (sd < d->contact_offset && !(w->nc < pair_lim)) ovf = 1;
          if (sd < d->contact_offset && w->nc < pair_lim) {
            contact_t* ct = &w->con[w->nc++];
            memset(ct, 0, sizeof *ct);
            ct->kind = 1; ct->actA = a; ct->actB = b; ct->sd = sd;
            ct->bodyA = m->sphere_body[sa]; ct->repA = a * MQE_NREP + m->sphere_reported[sa]; ct->bodyB = 0; ct->repB = A * MQE_NREP + (b - A);
            for (int k = 0; k < 3; k++) { ct->n[k] = n[k]; ct->p[k] = ca[k] - n[k] * (w->sph_r[a][sa] + (real)0.5 * sd); }
          }
        }
        /* ... and the box's own eight corners (radius 0) against the robot's primitives: what a feature point of the robot cannot
         * see -- a corner of the box pressing into a face of the trunk or into a bar between its ends.  Corner by corner (-,-,-),
         * (-,-,+), ... (+,+,+), primitive by primitive; normal from B (the box) to A */
        for (int cn = 0; cn < 8; cn++) {
          real cl[3] = {(cn & 4 ? hb[0] : -hb[0]), (cn & 2 ? hb[1] : -hb[1]), (cn & 1 ? hb[2] : -hb[2])}, cw[3], cc[3];
          mat3_vec(w->npcR[b - A], cl, cw);
          for (int k = 0; k < 3; k++) cc[k] = npc_pos[b - A][k] + cw[k];
          real db[3] = {cc[0] - pa[0], cc[1] - pa[1], cc[2] - pa[2]};
          real reach = m->feature_reach + d->contact_offset;
          if (dot3(db, db) > reach * reach) continue;          /* no primitive reaches farther from the base than the feature points do */
          for (int q = 0; q < m->n_prims; q++) {
            real n[3], sd;
            if (!feat_vs_prim(m, w, a, q, cc, (real)0, &sd, n)) continue;
            if (sd < d->contact_offset && !(w->nc < pair_lim)) ovf = 1;
            if (sd < d->contact_offset && w->nc < pair_lim) {
              contact_t* ct = &w->con[w->nc++];
              memset(ct, 0, sizeof *ct);
              ct->kind = 1; ct->actA = a; ct->actB = b; ct->sd = sd;
              ct->bodyA = m->prim_body[q]; ct->repA = a * MQE_NREP + m->prim_reported[q]; ct->bodyB = 0; ct->repB = A * MQE_NREP + (b - A);
              for (int k = 0; k < 3; k++) { ct->n[k] = -n[k]; ct->p[k] = cc[k] - n[k] * ((real)0.5 * sd); }
            }
          }
        }
        /* ... and its edges against the robot's primitives between their feature points (edge_contacts bits 2, 4) */
        if (d->edge_contacts & 6)
          for (int q = 0; q < m->n_prims; q++) {
            real sd, n[3], pa[3];
            if (!edge_vs_box(s, w, a, q, npc_pos[b - A], w->npcR[b - A], hb, 12, d->edge_contacts, &sd, n, pa)) continue;
            if (sd < d->contact_offset && !(w->nc < pair_lim)) ovf = 1;
            if (sd < d->contact_offset && w->nc < pair_lim) {
              contact_t* ct = &w->con[w->nc++];
              memset(ct, 0, sizeof *ct);
              ct->kind = 1; ct->actA = a; ct->actB = b; ct->sd = sd;
              ct->bodyA = m->prim_body[q]; ct->repA = a * MQE_NREP + m->prim_reported[q]; ct->bodyB = 0; ct->repB = A * MQE_NREP + (b - A);
              for (int k = 0; k < 3; k++) { ct->n[k] = n[k]; ct->p[k] = pa[k]; }
            }
          }
        continue;
      }
      if (dot3(dd, dd) > (real)(1.2 * 1.2)) continue;   /* broad phase: actors farther apart than 1.2 m cannot touch */
      if (a < A && b < A) {
        /* two robots: the feature points of one against the primitives of the other, both ways (outer loop over the primitives,
         * inner over the feature points); a foot against a foot is the same sphere pair both ways and is taken the first time only */
        for (int dir = 0; dir < 2; dir++) {
          const int fa = dir == 0 ? a : b, qa = dir == 0 ? b : a;     /* feature points of fa, primitives of qa */
          for (int q = 0; q < m->n_prims; q++)
            for (int f = 0; f < m->n_spheres; f++) {
              if (dir == 1 && m->prim_type[q] == MQE_PRIM_SPHERE && m->prim_type[m->sphere_prim[f]] == MQE_PRIM_SPHERE) continue;
              real n[3], sd;
              if (!feat_vs_prim(m, w, qa, q, w->sph_c[fa][f], w->sph_r[fa][f], &sd, n)) continue;
              if (sd < d->contact_offset && !(w->nc < pair_lim)) ovf = 1;
              if (sd < d->contact_offset && w->nc < pair_lim) {
                contact_t* ct = &w->con[w->nc++];
                memset(ct, 0, sizeof *ct);
                ct->kind = 1; ct->actA = fa; ct->actB = qa; ct->sd = sd;
                ct->bodyA = m->sphere_body[f]; ct->repA = fa * MQE_NREP + m->sphere_reported[f];
                ct->bodyB = m->prim_body[q]; ct->repB = qa * MQE_NREP + m->prim_reported[q];
                for (int k = 0; k < 3; k++) { ct->n[k] = n[k]; ct->p[k] = w->sph_c[fa][f][k] - n[k] * (w->sph_r[fa][f] + (real)0.5 * sd); }
              }
            }
        }
        continue;
      }
      if (a < A) {
        /* robot a against the collision spheres of free NPC b (ball, sheep): every sphere of b against the robot's primitives */
        for (int sb = 0; sb < w->sph_n[b]; sb++)
          for (int q = 0; q < m->n_prims; q++) {
            real n[3], sd;
            if (!feat_vs_prim(m, w, a, q, w->sph_c[b][sb], w->sph_r[b][sb], &sd, n)) continue;     /* n: from the primitive to the sphere */
            if (sd < d->contact_offset && !(w->nc < pair_lim)) ovf = 1;
            if (sd < d->contact_offset && w->nc < pair_lim) {
              contact_t* ct = &w->con[w->nc++];
              memset(ct, 0, sizeof *ct);
              ct->kind = 1; ct->actA = a; ct->actB = b; ct->sd = sd;
              ct->bodyA = m->prim_body[q]; ct->repA = a * MQE_NREP + m->prim_reported[q]; ct->bodyB = 0; ct->repB = A * MQE_NREP + (b - A);
              for (int k = 0; k < 3; k++) { ct->n[k] = -n[k]; ct->p[k] = w->sph_c[b][sb][k] - n[k] * (w->sph_r[b][sb] + (real)0.5 * sd); }   /* normal from B (the NPC) to A */
            }
          }
        continue;
      }
      for (int sb = 0; sb < w->sph_n[b]; sb++)          /* two free NPCs: sphere pairs */
        for (int sa = 0; sa < w->sph_n[a]; sa++) {
          const real* ca = w->sph_c[a][sa]; const real* cb = w->sph_c[b][sb];
          real e[3] = {ca[0] - cb[0], ca[1] - cb[1], ca[2] - cb[2]};
          real dist = (real)sqrt((double)dot3(e, e));
          real sd = dist - w->sph_r[a][sa] - w->sph_r[b][sb];
          if (sd < d->contact_offset && dist > (real)1e-9 && !(w->nc < pair_lim)) ovf = 1;
          if (sd < d->contact_offset && w->nc < pair_lim && dist > (real)1e-9) {
            contact_t* ct = &w->con[w->nc++];
            memset(ct, 0, sizeof *ct);
            ct->kind = 1; ct->actA = a; ct->actB = b; ct->sd = sd;
            ct->bodyA = 0; ct->repA = A * MQE_NREP + (a - A); ct->bodyB = 0; ct->repB = A * MQE_NREP + (b - A);
            for (int k = 0; k < 3; k++) { ct->n[k] = e[k] / dist; ct->p[k] = cb[k] + ct->n[k] * (w->sph_r[b][sb] + (real)0.5 * sd); }
          }
        }
    }
  /* links of one robot against each other (asset.self_collisions = 0, go1_config.py:73): the candidate (feature point, primitive)
   * list of the robot model in order; both sides of such a contact are the same actor (fill_jac adds the two Jacobians into one
   * row) */
  if (d->self_collision)
    for (int a = 0; a < A; a++)
      for (int pi = 0; pi < m->n_self_pairs; pi++) {
        int f = m->self_pair[pi] & 255, q = m->self_pair[pi] >> 8;
        real n[3], sd;
        if (!feat_vs_prim(m, w, a, q, w->sph_c[a][f], w->sph_r[a][f], &sd, n)) continue;
        if (sd < d->contact_offset && !(w->nc < pair_lim)) ovf = 1;
        if (sd < d->contact_offset && w->nc < pair_lim) {
          contact_t* ct = &w->con[w->nc++];
          memset(ct, 0, sizeof *ct);
          ct->kind = 1; ct->actA = a; ct->actB = a; ct->sd = sd;
          ct->bodyA = m->sphere_body[f]; ct->repA = a * MQE_NREP + m->sphere_reported[f];
          ct->bodyB = m->prim_body[q]; ct->repB = a * MQE_NREP + m->prim_reported[q];
          for (int k = 0; k < 3; k++) { ct->n[k] = n[k]; ct->p[k] = w->sph_c[a][f][k] - n[k] * (w->sph_r[a][f] + (real)0.5 * sd); }
        }
      }

  /* ---- contact rows: J, B = Minv J^T, K = J B */
  for (int ci = 0; ci < w->nc; ci++) {
    contact_t* ct = &w->con[ci];
    make_tangents(ct->n, ct->t1, ct->t2);
    real dirs[3][3];
    for (int k = 0; k < 3; k++) { dirs[0][k] = ct->n[k]; dirs[1][k] = ct->t1[k]; dirs[2][k] = ct->t2[k]; }
    fill_jac(s, w, ct->actA, ct->bodyA, ct->p, (real)1, dirs, ct->J, npc_pos);
    if (ct->actB >= 0) {
      fill_jac(s, w, ct->actB, ct->bodyB, ct->p, (real)-1, dirs, ct->J, npc_pos);
    }
    for (int q = 0; q < 3; q++) {
      for (int r = 0; r < A; r++) {
        real tmp[RD];
        for (int i = 0; i < RD; i++) tmp[i] = ct->J[q][r * RD + i];
        chol_solve(w->L[r], RD, RD, tmp);
        for (int i = 0; i < RD; i++) ct->B[q][r * RD + i] = tmp[i];
      }
      for (int p = 0; p < P; p++) {
        int o = A * RD + p * 6;
        for (int k = 0; k < 3; k++) { ct->B[q][o + k] = ct->J[q][o + k] / d->npc_mass; ct->B[q][o + 3 + k] = ct->J[q][o + 3 + k] / d->npc_inertia; }
      }
      if (SS) ct->B[q][sdof] = ct->J[q][sdof] / d->seesaw_plank_inertia_yy;
    }
    for (int q = 0; q < 3; q++) for (int r2 = 0; r2 < 3; r2++) {
      real acc = 0;
      for (int i = 0; i < ndof; i++) acc += ct->J[q][i] * ct->B[r2][i];
      ct->K[q][r2] = acc;
    }
  }

  if (s->warm > 0) {          /* EXPERIMENT: warm start (see mqo_sim::warm) */
    const int n_old = s->wc_n[env];
    const int32_t* ok = s->wc_key + (size_t)env * MAXC * 5;
    const float* op = s->wc_p + (size_t)env * MAXC * 3;
    const float* ol = s->wc_lam + (size_t)env * MAXC * 3;
    unsigned char used[MAXC];
    for (int j = 0; j < n_old; j++) used[j] = 0;
    for (int ci = 0; ci < w->nc; ci++) {
      contact_t* ct = &w->con[ci];
      int best = -1; real bd = (real)(0.02 * 0.02);
      for (int j = 0; j < n_old; j++) {
        if (used[j] || ok[j * 5] != ct->kind || ok[j * 5 + 1] != ct->actA || ok[j * 5 + 2] != ct->bodyA || ok[j * 5 + 3] != ct->actB || ok[j * 5 + 4] != ct->bodyB) continue;
        real d2 = 0;
        for (int k = 0; k < 3; k++) { real dd = ct->p[k] - (real)op[j * 3 + k]; d2 += dd * dd; }
        if (d2 < bd) { bd = d2; best = j; }
      }
      if (best < 0) continue;
      used[best] = 1;
      for (int q = 0; q < 3; q++) ct->lam[q] = (real)s->warm * (real)ol[best * 3 + q];
      for (int i = 0; i < ndof; i++) w->v[i] += ct->B[0][i] * ct->lam[0] + ct->B[1][i] * ct->lam[1] + ct->B[2][i] * ct->lam[2];
    }
  }

  /* ---- contact solver.  d->solver_type follows sim.physx.solver_type (legged_robot_config.py:219: "0: pgs, 1: tgs"):
   *   0  projected Gauss-Seidel on velocities: `solver_iterations` sweeps over the contacts detected at the start-of-step pose, gaps
   *      and penetrations enter as a velocity bias (penetration: erp / dt, capped by max_depenetration_velocity), positions are
   *      integrated once with the final velocity;
   *   1  temporal Gauss-Seidel as PhysX 4/5 publish it (PxSolverType::eTGS; Macklin et al., "Small Steps in Physics Simulation",
   *      SCA 2019): the step is cut into `solver_iterations` (= num_position_iterations, :221) sub-steps of dt / n.  Every position
   *      iteration re-evaluates each contact's separation from the motion accumulated so far (sep = sd + J_n . Delta, J of the
   *      start-of-step pose), asks for  u_n >= -sep / (dt / n)  -- a gap may close within the sub-step, a penetration is pushed out
   *      within it, at most at max_depenetration_velocity; no erp -- with the impulse ACCUMULATED over the iterations clamped at
   *      zero, and then advances the accumulated motion by (dt / n) v.  A penetration that has been pushed out by iteration k asks
   *      for u_n >= 0 from then on, so the push-out speed is taken back inside the same step instead of staying in the velocity;
   *      `velocity_iterations` (= num_velocity_iterations, :222; 0 in every config) further sweeps see penetrations as touching
   *      (sep = max(sep, 0)) and move nothing.  Positions are integrated with the accumulated motion, velocities are the last ones.
   * the shapes of a robot carry the env's (randomised) coefficient, averaged with the other shape's (PhysX combine mode);
   * contacts without a robot keep d->friction */
  const int TGS = d->solver_type == 1;
  const int npos = d->solver_iterations, nvel = d->velocity_iterations > 0 ? d->velocity_iterations : 0;
  const real sdt = TGS && npos > 0 ? dt / (real)npos : dt;
  real Dacc[MAXDOF];                   /* TGS: generalized displacement accumulated over the position iterations */
  for (int i = 0; i < ndof; i++) Dacc[i] = 0;
  real mu_robot = (real)0.5 * (s->dparams[(size_t)env * A * 8] + d->friction);
  for (int it = 0; it < npos + nvel; it++) {
    const int vel_it = it >= npos;
    for (int ci = 0; ci < w->nc; ci++) {
      contact_t* ct = &w->con[ci];
      real mu = (ct->actA < A || (ct->actB >= 0 && ct->actB < A)) ? mu_robot : (real)d->friction;
      real u[3];
      for (int q = 0; q < 3; q++) { real acc = 0; for (int i = 0; i < ndof; i++) acc += ct->J[q][i] * w->v[i]; u[q] = acc; }
      real bias;
      if (TGS) {
        real sep = ct->sd;
        { real acc = 0; for (int i = 0; i < ndof; i++) acc += ct->J[0][i] * Dacc[i]; sep += acc; }
        if (vel_it && sep < 0) sep = 0;
        bias = -sep / sdt;
        if (sep < 0 && bias > d->max_depenetration_velocity) bias = d->max_depenetration_velocity;
      } else {
        bias = ct->sd >= 0 ? -ct->sd / dt : (vel_it ? (real)0 : fminf((float)(-ct->sd * d->erp / dt), d->max_depenetration_velocity));
      }
      real dl[3];
      /* normal row */
      real ln = ct->lam[0] - (u[0] - bias) / ct->K[0][0];
      if (ln < 0) ln = 0;
#ifdef MQO_TRACE_ENV
      if (env == MQO_TRACE_ENV) fprintf(stderr, "ora k %d c %d sd %g bias %g u0 %g lam %g -> %g\n", it, ci, (double)ct->sd, (double)bias, (double)u[0], (double)ct->lam[0], (double)ln);
#endif
      dl[0] = ln - ct->lam[0]; ct->lam[0] = ln;
      u[1] += ct->K[1][0] * dl[0]; u[2] += ct->K[2][0] * dl[0];
      /* tangent rows, box friction |lt| <= mu ln */
      real lim = mu * ct->lam[0];
      real l1 = ct->lam[1] - u[1] / ct->K[1][1];
      if (l1 > lim) l1 = lim; if (l1 < -lim) l1 = -lim;
      dl[1] = l1 - ct->lam[1]; ct->lam[1] = l1;
      u[2] += ct->K[2][1] * dl[1];
      real l2 = ct->lam[2] - u[2] / ct->K[2][2];
      if (l2 > lim) l2 = lim; if (l2 < -lim) l2 = -lim;
      dl[2] = l2 - ct->lam[2]; ct->lam[2] = l2;
      for (int i = 0; i < ndof; i++) w->v[i] += ct->B[0][i] * dl[0] + ct->B[1][i] * dl[1] + ct->B[2][i] * dl[2];
    }
    if (TGS && !vel_it) for (int i = 0; i < ndof; i++) Dacc[i] += sdt * w->v[i];
  }
  if (s->warm > 0) {          /* EXPERIMENT: remember this substep's impulses */
    int32_t* ok = s->wc_key + (size_t)env * MAXC * 5;
    float* op = s->wc_p + (size_t)env * MAXC * 3;
    float* ol = s->wc_lam + (size_t)env * MAXC * 3;
    s->wc_n[env] = w->nc;
    for (int ci = 0; ci < w->nc; ci++) {
      const contact_t* ct = &w->con[ci];
      ok[ci * 5] = ct->kind; ok[ci * 5 + 1] = ct->actA; ok[ci * 5 + 2] = ct->bodyA; ok[ci * 5 + 3] = ct->actB; ok[ci * 5 + 4] = ct->bodyB;
      for (int k = 0; k < 3; k++) { op[ci * 3 + k] = (float)ct->p[k]; ol[ci * 3 + k] = (float)ct->lam[k]; }
    }
  }
  /* what the positions are integrated with beyond the final velocity: voff = (accumulated motion) / dt - v  (0 for solver type 0).
   * The joint-limit impulses below act on the velocity AND on the motion of the step: they keep voff. */
  real voff[MAXDOF];
  for (int i = 0; i < ndof; i++) voff[i] = (TGS && npos > 0) ? Dacc[i] / dt - w->v[i] : (real)0;
  {
    /* joint limits: q + dt*(qd + voff) within [lower, upper] and |qd| <= the URDF velocity limit (go1.urdf:115,157,185; legged_robot.py:315;
     * PhysX maxJointVelocity); one pass after the contact iterations, violations removed by an impulse along the joint */
    /* Gauss-Seidel over the joints, repeated while something still violates (an impulse on one joint changes its neighbours'
     * speeds), at most LIMIT_PASSES times; then the bound is enforced exactly */
    for (int pass = 0; pass <= LIMIT_PASSES; pass++) {
      int any = 0;
      for (int r = 0; r < A; r++)
        for (int j = 0; j < 12; j++) {
          real q = dofs[(r * 12 + j) * 2];
          real* v = w->v + r * RD;
          real lo = (m->dof_lower[j] - q) / dt - voff[r * RD + 6 + j], hi = (m->dof_upper[j] - q) / dt - voff[r * RD + 6 + j];
          if (m->dof_vel_limit[j] > 0) { real vl = m->dof_vel_limit[j]; if (lo < -vl) lo = -vl; if (hi > vl) hi = vl; }
          if (v[6 + j] < lo || v[6 + j] > hi) any = 1;
          if (pass == LIMIT_PASSES) { if (v[6 + j] < lo) v[6 + j] = lo; if (v[6 + j] > hi) v[6 + j] = hi; }
        }
      if (!any || pass == LIMIT_PASSES) break;
      for (int r = 0; r < A; r++)
        for (int j = 0; j < 12; j++) {
          real q = dofs[(r * 12 + j) * 2];
          real* v = w->v + r * RD;
          real lo = (m->dof_lower[j] - q) / dt - voff[r * RD + 6 + j], hi = (m->dof_upper[j] - q) / dt - voff[r * RD + 6 + j];
          if (m->dof_vel_limit[j] > 0) { real vl = m->dof_vel_limit[j]; if (lo < -vl) lo = -vl; if (hi > vl) hi = vl; }
          real viol = 0;
          if (v[6 + j] < lo) viol = lo - v[6 + j];
          else if (v[6 + j] > hi) viol = hi - v[6 + j];
          if (viol != 0) {
            /* impulse along e_j: dv = Minv e_j * lambda with (Minv)_jj lambda = viol */
            real col[RD];
            memset(col, 0, sizeof col); col[6 + j] = 1;
            chol_solve(w->L[r], RD, RD, col);
            real lam = viol / col[6 + j];
            for (int i = 0; i < RD; i++) v[i] += col[i] * lam;
          }
        }
    }
    if (SS) {   /* hinge: velocity limit (seesaw.urdf:65), then the geometric end stops */
      real vv = w->v[sdof], vl = d->seesaw_vel_limit;
      if (vv > vl) vv = vl; if (vv < -vl) vv = -vl;
      real lo = (d->seesaw_theta_lo - ssTheta) / dt - voff[sdof], hi = (d->seesaw_theta_hi - ssTheta) / dt - voff[sdof];
      if (vv < lo) vv = lo; if (vv > hi) vv = hi;
      w->v[sdof] = vv;
    }
  }

  if (ovf) s->overflow[env] += 1;
  if (red) s->overflow[s->N + env] += 1;
  /* ---- net contact forces per reported body (gym.refresh_net_contact_force_tensor analogue) */
  float* cf = s->cf + (size_t)env * s->NBR * 3;
  memset(cf, 0, (size_t)s->NBR * 3 * 4);
  for (int ci = 0; ci < w->nc; ci++) {
    contact_t* ct = &w->con[ci];
    real F[3];
    for (int k = 0; k < 3; k++) F[k] = (ct->lam[0] * ct->n[k] + ct->lam[1] * ct->t1[k] + ct->lam[2] * ct->t2[k]) / dt;
    for (int k = 0; k < 3; k++) cf[ct->repA * 3 + k] += (float)F[k];
    if (ct->actB >= 0) {
      for (int k = 0; k < 3; k++) cf[ct->repB * 3 + k] -= (float)F[k];
    }
  }

  /* ---- integrate (semi-implicit Euler; quaternion: first-order update + renormalise) */
  if (SS) { float* ds = dofs + (12 * A) * 2; ds[0] = (float)(ds[0] + dt * (w->v[sdof] + voff[sdof])); ds[1] = (float)w->v[sdof]; }
  for (int act = 0; act < nact; act++) {
    float* rs = root + act * 13;
    real* v = act < A ? w->v + act * RD : w->v + A * RD + (act - A) * 6;
    const real* vo = act < A ? voff + act * RD : voff + A * RD + (act - A) * 6;      /* positions move with v + voff (TGS: the accumulated motion) */
    for (int k = 0; k < 3; k++) { rs[k] = (float)(rs[k] + dt * (v[k] + vo[k])); rs[7 + k] = (float)v[k]; rs[10 + k] = (float)v[3 + k]; }
    if (act >= A && lin_only) { for (int k = 0; k < 3; k++) rs[10 + k] = (float)w->v[A * RD + (act - A) * 6 + 3 + k]; continue; }
    real q[4] = {rs[3], rs[4], rs[5], rs[6]}, wq[3] = {v[3] + vo[3], v[4] + vo[4], v[5] + vo[5]};
    /* qdot = 0.5 * (w,0) * q */
    real dq[4];
    dq[0] = (real)0.5 * (wq[0] * q[3] + wq[1] * q[2] - wq[2] * q[1]);
    dq[1] = (real)0.5 * (-wq[0] * q[2] + wq[1] * q[3] + wq[2] * q[0]);
    dq[2] = (real)0.5 * (wq[0] * q[1] - wq[1] * q[0] + wq[2] * q[3]);
    dq[3] = (real)0.5 * (-wq[0] * q[0] - wq[1] * q[1] - wq[2] * q[2]);
    real nq = 0;
    for (int k = 0; k < 4; k++) { q[k] += dt * dq[k]; nq += q[k] * q[k]; }
    nq = (real)sqrt((double)nq);
    for (int k = 0; k < 4; k++) rs[3 + k] = (float)(q[k] / nq);
    if (act < A)
      for (int j = 0; j < 12; j++) {
        float* ds = dofs + (act * 12 + j) * 2;
        ds[0] = (float)(ds[0] + dt * (v[6 + j] + vo[6 + j]));
        ds[1] = (float)v[6 + j];
      }
  }
}

int mqo_simulate(mqo_sim* s) {
#pragma omp parallel num_threads(mqo_team())
  {
    envwork_t* w = (envwork_t*)malloc(sizeof(envwork_t));
#pragma omp for schedule(static)
    for (int e = 0; e < s->N; e++) simulate_env(s, e, w);
    free(w);
  }
  return 0;
}

/* debug: mass matrix / bias / contact list of one env for fine-grained HIP parity */
int mqo_debug_dynamics(mqo_sim* s, int env, int robot, float* M_out /*18x18*/, float* Minv_out, int* nc_out, float* contacts_out /*[MAXC][8]: actA,sphA,actB,sphB,sd,n*/) {
  envwork_t* w = (envwork_t*)malloc(sizeof(envwork_t));
  /* run on a copy of the state so that nothing moves */
  int A = s->A;
  size_t nr = (size_t)(A + s->P) * 13, ndf = (size_t)s->ND * 2, ncf = (size_t)s->NBR * 3;
  float* r0 = (float*)dupmem(s->root + env * nr, nr * 4);
  float* d0 = (float*)dupmem(s->dof + env * ndf, ndf * 4);
  float* c0 = (float*)dupmem(s->cf + env * ncf, ncf * 4);
  const int32_t ov0 = s->overflow[env], rd0 = s->overflow[s->N + env];
  simulate_env(s, env, w);
  s->overflow[env] = ov0; s->overflow[s->N + env] = rd0;
  memcpy(s->root + env * nr, r0, nr * 4); memcpy(s->dof + env * ndf, d0, ndf * 4); memcpy(s->cf + env * ncf, c0, ncf * 4);
  free(r0); free(d0); free(c0);
  const real* L = w->L[robot];
  for (int i = 0; i < RD; i++) for (int j = 0; j < RD; j++) {
    real acc = 0;
    for (int k = 0; k <= (i < j ? i : j); k++) acc += L[i * RD + k] * L[j * RD + k];
    M_out[i * RD + j] = (float)acc;
  }
  for (int j = 0; j < RD; j++) {
    real col[RD]; memset(col, 0, sizeof col); col[j] = 1;
    chol_solve(L, RD, RD, col);
    for (int i = 0; i < RD; i++) Minv_out[i * RD + j] = (float)col[i];
  }
  *nc_out = w->nc;
  for (int c = 0; c < w->nc; c++) {
    float* o = contacts_out + c * 8;
    const contact_t* ct = &w->con[c];   /* (actor, dynamic body) pairs, -1/0 for static geometry */
    o[0] = (float)ct->actA; o[1] = (float)ct->bodyA;
    o[2] = (float)ct->actB; o[3] = (float)(ct->actB >= 0 ? ct->bodyB : 0);
    o[4] = (float)w->con[c].sd; o[5] = (float)w->con[c].n[0]; o[6] = (float)w->con[c].n[1]; o[7] = (float)w->con[c].n[2];
  }
  free(w);
  return 0;
}

/* ------------------------------------------------------------------------------------------ rows J..N: post-physics */
/* in_step: obs_buf.last_last_action is a VIEW of self.last_actions (go1.py:183), and legged_robot.py:151 overwrites
 * last_actions in place right after compute_observations -- so what any caller of step() observes is the new value */
static void compute_observations_env(mqo_sim* s, int e, int in_step) { /* go1.py:153-196 */
  int A = s->A;
  const mqe_sim_desc* d = &s->d;
  for (int a = 0; a < A; a++) {
    int i = e * A + a;
    float* ob = s->obs_bag + (size_t)i * OBS_BAG;
    const float* rs = s->root + ((size_t)e * (A + s->P) + a) * 13;
    for (int k = 0; k < 3; k++) ob[k] = rs[k] - s->env_origins[e * 3 + k];          /* base_pos :162 */
    euler_xyz_f(s->bquat + i * 4, ob + 3);                                          /* base_rpy :193 */
    for (int j = 0; j < 12; j++) {
      const float* ds = s->dof + ((size_t)e * s->ND + a * 12 + j) * 2;
      ob[6 + j] = (ds[0] - d->default_dof_pos[j]) * 1.0f;                           /* dof_pos :168 */
      ob[18 + j] = ds[1] * 0.05f;                                                   /* dof_vel :171 */
      ob[36 + j] = s->actions[i * 12 + j];                                          /* last_action :180 */
      ob[48 + j] = in_step ? s->actions[i * 12 + j] : s->last_actions[i * 12 + j];  /* last_last_action :183 (+ :151 aliasing) */
    }
    for (int k = 0; k < 3; k++) {
      ob[30 + k] = s->blv[i * 3 + k] * 2.0f;                                        /* lin_vel :174 */
      ob[33 + k] = s->bav[i * 3 + k] * 0.25f;                                       /* ang_vel :177 */
      ob[60 + k] = s->pg[i * 3 + k];                                                /* projected_gravity :186 */
    }
    for (int k = 0; k < 4; k++) { ob[63 + k] = s->clock[i * 4 + k]; ob[67 + k] = s->bquat[i * 4 + k]; }  /* :190, :165 */
  }
}
/* NOTE on the bag layout used above: [0:3] base_pos, [3:6] base_rpy, [6:18] dof_pos, [18:30] dof_vel, [30:33] lin_vel,
 * [33:36] ang_vel, [36:48] last_action, [48:60] last_last_action, [60:63] projected_gravity, [63:67] clock_inputs,
 * [67:71] base_quat, [71:74] pad.  mqo_policy_step reads gravity/clock at these offsets. */

/* pre-reset xy of the agents' root-state ROWS 0 .. N-1 (row e = robot e % A of env e / A): what _get_terrain_curriculum_move indexes
 * with env ids (legged_robot.py:498), taken before any env of the step is reset (the curriculum runs first in reset_idx) */
static void curriculum_snapshot(mqo_sim* s) {
  if (!s->d.terrain_curriculum) return;
  int A = s->A, P = s->P;
  for (int e = 0; e < s->N; e++) {
    const float* rs = s->root + ((size_t)(e / A) * (A + P) + e % A) * 13;
    s->curr_xy[e * 2] = rs[0]; s->curr_xy[e * 2 + 1] = rs[1];
  }
}
/* _update_terrain_curriculum (legged_robot.py:479-503) for one env that is being reset */
static void curriculum_move(mqo_sim* s, int e) {
  const mqe_sim_desc* d = &s->d;
  float dx = s->curr_xy[e * 2] - s->env_origins_live[e * 3], dy = s->curr_xy[e * 2 + 1] - s->env_origins_live[e * 3 + 1];
  float distance = sqrtf(dx * dx + dy * dy);                                   /* :498 */
  int move_up = distance > d->terrain_env_length / 2;                          /* :500 */
  int move_down = (distance < 0.0f * 0.5f) && !move_up;                        /* :502 with the commands Go1 never samples (zero) */
  int lvl = s->terrain_levels[e] + move_up - move_down;                        /* :490 */
  if (lvl >= d->terrain_num_rows)                                              /* :492-494: past the last level -> a random one */
    lvl = (int)(mqo_u01((uint32_t)d->seed, (uint32_t)(e + d->env_id_offset), (uint32_t)s->reset_count[e], 250u) * (float)d->terrain_num_rows);
  else if (lvl < 0) lvl = 0;
  if (lvl >= d->terrain_num_rows) lvl = d->terrain_num_rows - 1;
  s->terrain_levels[e] = lvl;
  for (int k = 0; k < 3; k++) s->env_origins_live[e * 3 + k] = s->terrain_origins[((size_t)lvl * d->terrain_num_cols + s->terrain_types[e]) * 3 + k];   /* :495 */
}

static void reset_env(mqo_sim* s, int e) { /* go1.py:110-145, legged_robot.py:394-470,647-652 */
  const mqe_sim_desc* d = &s->d;
  if (s->warm > 0) s->wc_n[e] = 0;
  int A = s->A, P = s->P;
  float* root = s->root + (size_t)e * (A + P) * 13;
  float* dofs = s->dof + (size_t)e * s->ND * 2;
  if (d->terrain_curriculum) curriculum_move(s, e);                            /* go1.py:123-125: first thing in reset_idx */
  for (int a = 0; a < A; a++)
    for (int j = 0; j < 12; j++) {
      float ratio = mqo_rand(s, e, (uint32_t)(a * 12 + j), d->dof_ratio_lo, d->dof_ratio_hi);
      dofs[(a * 12 + j) * 2] = d->default_dof_pos[j] * ratio;     /* legged_robot.py:403 */
      dofs[(a * 12 + j) * 2 + 1] = 0.0f;                          /* :411 */
    }
  for (int k = 0; k < s->npc_dofs; k++) { dofs[(12 * A + k) * 2] = d->seesaw_default_angle; dofs[(12 * A + k) * 2 + 1] *= 0.0f; } /* :414-415 */
  for (int a = 0; a < A; a++) {
    float* rs = root + a * 13;
    memcpy(rs, s->base_init + a * 13, 13 * 4);                     /* :433 */
    for (int k = 0; k < 3; k++) rs[k] += s->agent_origins[((size_t)e * A + a) * 3 + k];  /* :434 */
  }
  for (int p = 0; p < P; p++) {
    float* rs = root + (A + p) * 13;
    memcpy(rs, s->npc_init + p * 13, 13 * 4);                      /* :436 */
    for (int k = 0; k < 3; k++) rs[k] += s->env_origins_live[e * 3 + k];  /* :437 (the live tensor) */
  }
  if (d->has_base_pos_range)
    for (int a = 0; a < A; a++) {
      root[a * 13 + 0] += mqo_rand(s, e, (uint32_t)(64 + a), d->base_pos_x_lo, d->base_pos_x_hi);   /* :441 */
      root[a * 13 + 1] += mqo_rand(s, e, (uint32_t)(72 + a), d->base_pos_y_lo, d->base_pos_y_hi);   /* :442 */
    }
  if (d->has_npc_pos_range)
    for (int p = 0; p < P; p++) {
      root[(A + p) * 13 + 0] += mqo_rand(s, e, (uint32_t)(128 + p), d->npc_pos_x_lo, d->npc_pos_x_hi);  /* :445 */
      root[(A + p) * 13 + 1] += mqo_rand(s, e, (uint32_t)(160 + p), d->npc_pos_y_lo, d->npc_pos_y_hi);  /* :446 */
    }
  for (int a = 0; a < A; a++)
    for (int c = 0; c < 6; c++) root[a * 13 + 7 + c] = mqo_rand(s, e, (uint32_t)(80 + a * 6 + c), d->base_vel_lo, d->base_vel_hi); /* :458 */
  /* _reset_buffers: legged_robot.py:647-652, go1.py:141-145 */
  for (int k = 0; k < 12 * A; k++) s->last_actions[(size_t)e * 12 * A + k] = 0.0f;
  s->ep_len[e] = 0;
  s->reset_buf[e] = 1;
  s->wdone[e] = 1;
  for (int a = 0; a < A; a++) {
    int i = e * A + a;
    s->gait[i] = 0.0f;
    memset(s->hist + (size_t)i * MQE_HIST * FR, 0, (size_t)MQE_HIST * FR * 4);
  }
  s->reset_count[e] += 1;
}

/* sheep flocking script, go1_sheep.py:14-18,35-64.  noise: injected N(0,1) [P][3] */
/* MQE_NOISE_HASH: N(0,1) by Box-Muller over the counter RNG, keyed by (seed, global env id, post-step ordinal, sheep * 3 + axis);
 * same formula as mqe_randn() of the engine (kernels_step.hpp) */
static float mqo_randn(const mqo_sim* s, int e, uint32_t k) {
  const mqe_sim_desc* d = &s->d;
  const uint32_t cnt = 0x60000000u + (uint32_t)(s->n_post_steps + 1), genv = (uint32_t)(e + d->env_id_offset);
  const float u1 = mqo_u01((uint32_t)d->seed, genv, cnt, 2u * k), u2 = mqo_u01((uint32_t)d->seed, genv, cnt, 2u * k + 1u);
  return sqrtf(-2.0f * logf(1.0f - u1)) * cosf(6.2831855f * u2);
}

static void step_sheep_env(mqo_sim* s, int e) {
  const mqe_sim_desc* d = &s->d;
  int A = s->A, P = s->P;
  float* root = s->root + (size_t)e * (A + P) * 13;
  float avg[3] = {0, 0, 0};
  for (int p = 0; p < P; p++) for (int k = 0; k < 3; k++) avg[k] += root[(A + p) * 13 + k];
  for (int k = 0; k < 3; k++) avg[k] /= (float)P;
  s->sheep_avg[e * 2] = avg[0]; s->sheep_avg[e * 2 + 1] = avg[1];
  float var = 0;
  for (int k = 0; k < 2; k++) {
    float acc = 0;
    for (int p = 0; p < P; p++) { float t = root[(A + p) * 13 + k] - avg[k]; acc += t * t; }
    var += acc / (float)P;
  }
  s->sheep_var[e] = var;
  float dvs[MAXP][3];
  for (int p = 0; p < P; p++) {
    const float* sp = root + (A + p) * 13;
    float dv[3];
    for (int k = 0; k < 3; k++) {                                               /* :43 randn_like: scripted (fixtures) or a fresh draw per step */
      float z = d->noise_mode == MQE_NOISE_SCRIPTED ? s->npc_noise[((size_t)e * P + p) * 3 + k]
                                                    : (d->sheep_movement_randomness != 0.0f ? mqo_randn(s, e, (uint32_t)(p * 3 + k)) : 0.0f);
      dv[k] = d->sheep_movement_randomness * z * 2.0f;
    }
    if (P != 1) {
      float rel[3] = {avg[0] - sp[0], avg[1] - sp[1], avg[2] - sp[2]};
      float nr = sqrtf(rel[0] * rel[0] + rel[1] * rel[1] + rel[2] * rel[2]);
      /* :47; a sheep exactly on the flock mean would be 0 / 0 in torch: the cohesion term is dropped there (engine: no-NaN math) */
      if (nr > 0.0f) for (int k = 0; k < 3; k++) dv[k] += d->sheep_movement_randomness * rel[k] / nr / 1.5f;
    }
    for (int a = 0; a < A; a++) {
      const float* dp = root + a * 13;
      float rel[3] = {sp[0] - dp[0], sp[1] - dp[1], sp[2] - dp[2]};
      float sq[3] = {rel[0] * rel[0], rel[1] * rel[1], rel[2] * rel[2]};
      float dis = sqrtf(sq[0] * sq[0] + sq[1] * sq[1] + sq[2] * sq[2]);    /* norm of the element-wise square, :15 */
      float den = powf(dis, 1.4f);
      for (int k = 0; k < 3; k++) { float t = rel[k] / den; if (dis > 9.0f) t = 0.0f; dv[k] += d->sheep_movement_scale * t; } /* :16-17,52 */
    }
    dv[2] = 0.0f;                                                             /* :54 */
    for (int k = 0; k < 3; k++) dvs[p][k] = dv[k];
  }
  for (int p = 0; p < P; p++) {
    float* sp = root + (A + p) * 13;
    for (int k = 0; k < 3; k++) sp[7 + k] += dvs[p][k];                         /* :58 */
    for (int k = 0; k < 2; k++) sp[7 + k] = fminf(fmaxf(sp[7 + k], -2.0f), 2.0f);  /* :59 */
    sp[2] = fminf(fmaxf(sp[2], 0.0f), 0.3f);                                    /* :60 */
    sp[3] = 0.0f; sp[4] = 0.0f;                                                 /* :61 */
  }
}

static void wrapper_env(mqo_sim* s, int e, int is_reset_call, const float* pre_npc);
/* a / b and atan(a / b) as the engine evaluates them (kernels_step.hpp: div_ieee / atan_ratio): torch's IEEE results for b == 0
 * spelt out with a large finite value in place of +-inf, 0 / 0 -> 0 */
static inline float div_ieee(float a, float b) { return b != 0.0f ? a / b : (a > 0.0f ? 3.0e38f : (a < 0.0f ? -3.0e38f : 0.0f)); }
static inline float atan_ratio(float a, float b) { return b != 0.0f ? atanf(a / b) : (a > 0.0f ? 1.5707964f : (a < 0.0f ? -1.5707964f : 0.0f)); }

/* post_physics_step in the stages the reference's own method has (legged_robot.py:117-157), so that a subclass's check_termination /
 * _step_npc / reset_idx / compute_observations can run between them (include/mqe_hip.h mqe_post_physics_stage): FRAME = episode counter,
 * body-frame quantities, gait clock, the wrapper's copy of the NPC rows (:126-139) + the default check_termination (:159-169, field:121-146);
 * NPC = the task's NPC script (:146); RESET = reset_idx of the envs whose reset_buf is set NOW (:147-148); OBS = compute_observations +
 * last_actions / last_dof_vel (:149-152); WRAPPER = the task wrapper's observation / reward + the periodic push; the step counter
 * advances with it.  mqo_post_physics_step = all of them. */
static int post_stages(mqo_sim* s, int stages) {
  const mqe_sim_desc* d = &s->d;
  int N = s->N, A = s->A, P = s->P;
  float dtp = d->dt * (float)d->decimation;     /* self.dt = decimation * sim dt (legged_robot.py:1014) */
  if (P && !s->pre_npc) s->pre_npc = (float*)malloc((size_t)N * P * 13 * 4);
  float* pre_npc = P ? s->pre_npc : NULL;
  if (stages & MQE_POST_FRAME)
  for (int e = 0; e < N; e++) {
    float* root = s->root + (size_t)e * (A + P) * 13;
    s->ep_len[e] += 1;                                                           /* legged_robot.py:126 */
    for (int a = 0; a < A; a++) {
      int i = e * A + a;
      const float* rs = root + a * 13;
      float g3[3] = {0.0f, 0.0f, -1.0f};
      for (int k = 0; k < 4; k++) s->bquat[i * 4 + k] = rs[3 + k];               /* :132 */
      quat_rotate_inverse_f(rs + 3, rs + 7, s->blv + i * 3);                     /* :133 */
      quat_rotate_inverse_f(rs + 3, rs + 10, s->bav + i * 3);                    /* :134 */
      quat_rotate_inverse_f(rs + 3, g3, s->pg + i * 3);                          /* :135 */
      /* gait clock, go1.py:240-279 */
      const float* lo = s->loco_obs + (size_t)i * FR;
      float f = lo[7], ph = lo[8], off = lo[9], bnd = lo[10], dur = lo[11];
      float gi = s->gait[i] + dtp * f;
      gi = gi - floorf(gi);                                                       /* torch.remainder(x, 1.0) */
      s->gait[i] = gi;
      float fi[4] = {gi + ph + off + bnd, gi + off, gi + bnd, gi + ph};
      for (int k = 0; k < 4; k++) {
        float r = fi[k] - floorf(fi[k]);
        if (r < dur) fi[k] = r * (0.5f / dur);
        else if (r > dur) fi[k] = 0.5f + (r - dur) * (0.5f / (1.0f - dur));
        s->clock[i * 4 + k] = sinf(6.2831855f * fi[k]);
      }
    }
    /* check_termination: legged_robot.py:159-169 + legged_robot_field.py:121-146 */
    uint8_t reset = 0, collide = 0, rterm = 0, pterm = 0, zh = 0;
    if (d->terminate_on_base_contact) {
      for (int a = 0; a < A; a++) {
        const float* f3 = s->cf + ((size_t)e * s->NBR + a * MQE_NREP) * 3;
        if (sqrtf(f3[0] * f3[0] + f3[1] * f3[1] + f3[2] * f3[2]) > 1.0f) collide = 1;
      }
      reset = collide;
    }
    uint8_t to = s->ep_len[e] > d->max_episode_length;
    s->time_out[e] = to;
    reset |= to;
    for (int a = 0; a < A; a++) {
      int i = e * A + a;
      float rpy[3];
      euler_xyz_f(s->bquat + i * 4, rpy);
      float r = rpy[0], p = rpy[1];
      if (r > 3.1415927f) r -= 6.2831855f;
      if (p > 3.1415927f) p -= 6.2831855f;
      float z = root[a * 13 + 2] - s->agent_origins[((size_t)e * A + a) * 3 + 2];
      if ((d->termination_flags & MQE_TERM_ROLL) && fabsf(r) > d->roll_threshold) rterm = 1;
      if ((d->termination_flags & MQE_TERM_PITCH) && fabsf(p) > d->pitch_threshold) pterm = 1;
      if ((d->termination_flags & MQE_TERM_Z_LOW) && z < d->z_low_threshold) reset = 1;
      if ((d->termination_flags & MQE_TERM_Z_HIGH) && z > d->z_high_threshold) zh = 1;
    }
    if (d->termination_flags & MQE_TERM_ROLL) s->r_term[e] = rterm;
    if (d->termination_flags & MQE_TERM_PITCH) s->p_term[e] = pterm;
    if (d->termination_flags & MQE_TERM_Z_HIGH) s->zh_term[e] = zh;
    reset |= rterm | pterm | zh;
    s->reset_buf[e] = reset;
    s->wdone[e] = (uint8_t)reset;
    /* reset_buf aliases collide_buf when contact termination is on (legged_robot.py:165) */
    if (d->terminate_on_base_contact) s->collide_buf[e] = reset;
    /* the wrapper's view of root_states_npc: the copy taken before the NPC script (:136) */
    if (P) memcpy(pre_npc + (size_t)e * P * 13, s->root + ((size_t)e * (A + P) + A) * 13, (size_t)P * 13 * 4);
  }
  /* NPC script before resets (legged_robot.py:146), sees this step's post-physics states */
  if ((stages & MQE_POST_NPC) && d->npc_kind == MQE_NPC_SHEEP) for (int e = 0; e < N; e++) step_sheep_env(s, e);
  if (stages & MQE_POST_RESET) curriculum_snapshot(s);
  for (int e = 0; e < N; e++) {
    if (stages & MQE_POST_RESET) {
      s->wdone[e] = s->reset_buf[e];            /* (a subclass's check_termination may have changed the flag since FRAME) */
      if (d->terminate_on_base_contact) s->collide_buf[e] = s->reset_buf[e];   /* ... and collide_buf aliases it (legged_robot.py:165) */
      if (s->reset_buf[e]) {
        reset_env(s, e);                                                            /* :148 */
        if (P) memcpy(pre_npc + (size_t)e * P * 13, s->root + ((size_t)e * (A + P) + A) * 13, (size_t)P * 13 * 4);
        if (P == 0) /* root_states is a live view when there are no NPCs: base_quat shows the post-reset value */
          for (int a = 0; a < A; a++) for (int k = 0; k < 4; k++) s->bquat[(e * A + a) * 4 + k] = s->root[((size_t)e * A + a) * 13 + 3 + k];
      }
    }
    if (stages & MQE_POST_OBS) {
      compute_observations_env(s, e, 1);                                            /* :149 */
      for (int k = 0; k < 12 * A; k++) s->last_actions[(size_t)e * 12 * A + k] = s->actions[(size_t)e * 12 * A + k];  /* :151 */
      for (int k = 0; k < 12 * A; k++) s->last_dof_vel[(size_t)e * 12 * A + k] = s->dof[((size_t)e * s->ND + k) * 2 + 1];    /* :152 */
    }
    if (stages & MQE_POST_WRAPPER) {
      const int keep = s->wrapper_side_effects;
      if (stages & MQE_POST_WRAPPER_LEVEL) s->wrapper_side_effects = 1;
      wrapper_env(s, e, 0, pre_npc ? pre_npc + (size_t)e * P * 13 : NULL);
      s->wrapper_side_effects = keep;
      /* _push_robots (go1.py:237, legged_robot.py:470-476): after this step's frame quantities were taken, before reset_idx,
       * whose base velocities replace the push in the envs that reset; one draw per robot */
      if (d->push_interval > 0 && (s->n_post_steps + 1) % d->push_interval == 0 && !s->reset_buf[e]) {
        uint32_t cnt = 0x50000000u + (uint32_t)((s->n_post_steps + 1) / d->push_interval);
        for (int a = 0; a < A; a++)
          for (int c = 0; c < 2; c++)
            s->root[((size_t)e * (A + P) + a) * 13 + 7 + c] =
                2 * d->max_push_vel_xy * mqo_u01((uint32_t)d->seed, (uint32_t)(e + d->env_id_offset), cnt, (uint32_t)(2 * a + c)) + -d->max_push_vel_xy;
      }
    }
  }
  if (stages & MQE_POST_WRAPPER) s->n_post_steps++;
  return 0;
}
int mqo_post_physics_step(mqo_sim* s) { return post_stages(s, MQE_POST_ALL); }
int mqo_post_physics_stage(mqo_sim* s, int stages) { return post_stages(s, stages); }

int mqo_reset_all(mqo_sim* s) { /* Go1.reset go1.py:147-151 + wrapper.reset() */
  curriculum_snapshot(s);
  for (int e = 0; e < s->N; e++) {
    reset_env(s, e);
    /* base_quat is a view of the root_states tensor made in _init_buffers (legged_robot.py:568-570).  Without NPCs
     * that tensor aliases the simulator state for ever; with NPCs it is a private copy that reset_idx still writes
     * to only until the first post_physics_step rebinds self.root_states (:130) */
    if (s->P == 0 || s->n_post_steps == 0)
      for (int a = 0; a < s->A; a++) for (int k = 0; k < 4; k++) s->bquat[(e * s->A + a) * 4 + k] = s->root[((size_t)e * (s->A + s->P) + a) * 13 + 3 + k];
    compute_observations_env(s, e, 0);
    s->w_have_last[e] = 0;
    s->w_delayed_reset[e] = 0;
    wrapper_env(s, e, 1, s->P ? s->root + ((size_t)e * (s->A + s->P) + s->A) * 13 : NULL);
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------ rows W1-W4: task wrappers
 * npc: the (N*P,13) `root_states_npc` the wrapper sees = copy taken before the NPC script, overwritten by reset
 * (legged_robot.py:136,436).  Reward terms are also accumulated per env into rsum[e][term]. */
static void base_info(const mqo_sim* s, int i, float* o6) {
  const float* ob = s->obs_bag + (size_t)i * OBS_BAG;
  for (int k = 0; k < 6; k++) o6[k] = ob[k];
}
static void wrapper_env(mqo_sim* s, int e, int is_reset_call, const float* npc) {
  const mqe_sim_desc* d = &s->d;
  int A = s->A, P = s->P, Aw = s->Aw, D = s->D;
  float* obs = s->wobs + (size_t)e * Aw * D;
  float* rew = s->wrew + (size_t)e * Aw;
  float* rs = s->rsum + (size_t)e * MQE_MAX_REWARD_TERMS;
  const float* sc = d->reward_scale;
  for (int a = 0; a < Aw; a++) {
    float* o = obs + a * D;
    int c = 0;
    if (d->task == MQE_TASK_TUG) {                /* go1_tug_wrapper.py:47-53: [base info, slider (pos, vel), distance to it, slider pos] */
      const float npos = s->dof[((size_t)e * s->ND + 12 * A) * 2], nvel = s->dof[((size_t)e * s->ND + 12 * A) * 2 + 1];
      const float sgn = a == 1 ? -1.0f : 1.0f;    /* agent 1 sees the mirrored scene: entries 1, 4, 6, 9 negated (:54-57) */
      base_info(s, e * A + a, o);
      const float dx = o[0] - 1.6f, dy = o[1] - npos;
      o[1] *= sgn; o[4] *= sgn;
      o[6] = sgn * npos; o[7] = nvel; o[8] = sqrtf(dx * dx + dy * dy); o[9] = sgn * npos;   /* last_npc_pos was just refreshed (:120) */
      continue;
    }
    if (d->task != MQE_TASK_ROTATION && d->task != MQE_TASK_BRIDGE && d->task != MQE_TASK_WRESTLING)
      for (int k = 0; k < Aw; k++) o[c++] = (k == a) ? 1.0f : 0.0f;           /* obs_ids (empty_wrapper.py:18) */
    base_info(s, e * A + a, o + c); c += 6;
    if (d->task != MQE_TASK_PLAIN) { base_info(s, e * A + (Aw - 1 - a), o + c); c += 6; }  /* torch.flip(base_info,[1]) */
    if (d->task == MQE_TASK_GATE || d->task == MQE_TASK_SHEEP || d->task == MQE_TASK_PUSHBOX) { o[c++] = s->gate_pos[e * 2]; o[c++] = s->gate_pos[e * 2 + 1]; }
    if (d->task == MQE_TASK_PUSHBOX) {              /* go1_pushbox_wrapper.py:44-48: box xy (rel. env origin), box quaternion */
      o[c++] = npc[0] - s->env_origins_live[e * 3]; o[c++] = npc[1] - s->env_origins_live[e * 3 + 1];
      for (int k = 0; k < 4; k++) o[c++] = npc[3 + k];
    }
    if (d->task == MQE_TASK_SHEEP)
      for (int p = 0; p < P; p++) { o[c++] = npc[p * 13] - s->env_origins[e * 3]; o[c++] = npc[p * 13 + 1] - s->env_origins[e * 3 + 1]; }
    if (d->task == MQE_TASK_FOOTBALL_DEFENDER) {
      for (int k = 0; k < 3; k++) o[c++] = npc[k] - s->env_origins_live[e * 3 + k];
      for (int k = 0; k < 3; k++) o[c++] = npc[7 + k];
    }
  }
  if (d->task == MQE_TASK_TUG) {                  /* go1_tug_wrapper.py:59-136 */
    float* nd = s->dof + ((size_t)e * s->ND + 12 * A) * 2;
    const float npos = nd[0];
    const float* ob0 = s->obs_bag + (size_t)(e * A) * OBS_BAG;
    const float* ob1 = s->obs_bag + (size_t)(e * A + 1) * OBS_BAG;
    if (is_reset_call) {                          /* _init_extras (:37-40) */
      s->w_last[e * MAXA] = ob0[0]; s->w_last[e * MAXA + 1] = ob0[1];
      s->w_last2[e * 2] = npos;
      s->w_delayed_reset[e] = 0;
      for (int a = 0; a < Aw; a++) rew[a] = 0;
      return;
    }
    const float last_npc = s->w_last2[e * 2];
    const float lx = s->w_last[e * MAXA] - 1.6f, ly = s->w_last[e * MAXA + 1] - npos;      /* last base pos vs the CURRENT slider pos (:74-77) */
    const float cx = ob0[0] - 1.6f, cy = ob0[1] - npos;
    const float last_dis = sqrtf(lx * lx + ly * ly), dis = sqrtf(cx * cx + cy * cy);
    float r0 = 0.0f, sr = 0.0f, pu = 0.0f, pr = 0.0f, pp = 0.0f;
    if (sc[0] != 0) { if (npos < 0) sr = sc[0] * -npos; if (last_npc <= npos) sr /= 2; r0 += sr; rs[0] += sr; }
    if (sc[1] != 0) { if (npos > 0) pu = sc[1] * npos; if (last_npc > npos) pu /= 2; r0 -= pu; rs[1] += pu; }
    if (sc[2] != 0) { if (dis < last_dis) pr = (last_dis - dis) * sc[2]; r0 += pr; rs[2] += pr; }
    if (sc[3] != 0) { if (dis >= last_dis) pp = powf(2.0f, dis) * sc[3]; r0 -= pp; rs[3] += pp; }
    rs[4] += npos; rs[5] += pr + sr - pu; rs[6] += ob0[0]; rs[7] += ob0[1]; rs[8] += ob1[0]; rs[9] += ob1[1];   /* running logs (:122-127) */
    s->w_last[e * MAXA] = ob0[0]; s->w_last[e * MAXA + 1] = ob0[1];
    s->w_last2[e * 2] = npos;
    rew[0] = r0;
    for (int a = 1; a < Aw; a++) rew[a] = 0;
    /* the wrapper re-zeroes the slider at the start of the two steps that follow an env reset (reset_dic, :61-69, :71);
     * done here, right after the observation of this step, which is the same instant as far as the simulation goes */
    if (s->wrapper_side_effects) {               /* not part of Go1.step (mqo_post_physics_step alone leaves the state as simulated) */
      if (s->reset_buf[e]) s->w_delayed_reset[e] = 2;
      if (s->w_delayed_reset[e] > 0) { nd[0] = 0.0f; nd[1] = 0.0f; s->w_delayed_reset[e]--; }
    }
    return;
  }
  if (d->task == MQE_TASK_BRIDGE) {               /* go1_bridge_wrapper.py */
    const float* ob0 = s->obs_bag + (size_t)(e * A) * OBS_BAG;
    const float* ob1 = s->obs_bag + (size_t)(e * A + 1) * OBS_BAG;
    if (is_reset_call) {                          /* _init_extras (:27-29): target_pos = flip(base_pos at reset) */
      s->w_last[e * MAXA] = fabsf(ob1[0] + ob0[0]);     /* |target_pos[:,0,0] + target_pos[:,1,0]| */
      s->w_last[e * MAXA + 1] = ob1[0];                 /* target_pos[:,0,0]: where the opponent started */
    }
    const float S = s->w_last[e * MAXA];
    float* o1 = obs + 1 * D;                      /* agent 1 walks the bridge the other way (:37-40, :76-79) */
    o1[0] = S - o1[0]; o1[4] = -o1[4]; o1[6] = S - o1[6]; o1[10] = -o1[10];
    if (is_reset_call) { for (int a = 0; a < Aw; a++) rew[a] = 0; return; }
    float r0 = 0.0f;
    if (sc[0] != 0 && ob1[2] < 0.5f) { r0 += sc[0]; rs[0] += sc[0]; }          /* the opponent fell off */
    if (sc[1] != 0 && ob0[2] < 0.5f) { r0 -= sc[1]; rs[1] += sc[1]; }          /* agent 0 fell off */
    if (sc[2] != 0 && ob0[0] > s->w_last[e * MAXA + 1]) { r0 += sc[2]; rs[2] += sc[2]; }   /* reached the other side */
    rew[0] = r0;
    for (int a = 1; a < Aw; a++) rew[a] = 0;
    return;
  }
  if (d->task == MQE_TASK_WRESTLING) {            /* go1_wrestling_wrapper.py */
    float* o1 = obs + 1 * D;
    o1[1] = -o1[1]; o1[4] = -o1[4]; o1[7] = -o1[7]; o1[10] = -o1[10];
    if (is_reset_call) { for (int a = 0; a < Aw; a++) rew[a] = 0; return; }
    float r0 = 0.0f;
    int down[2];
    for (int a = 0; a < 2; a++) {                 /* roll / pitch of base_quat, wrapped to (-pi, pi] (:58-62) */
      const float* ob = s->obs_bag + (size_t)(e * A + a) * OBS_BAG;
      float r = ob[3], p = ob[4];
      if (r > 3.14159265358979f) r -= 6.28318530717959f;
      if (p > 3.14159265358979f) p -= 6.28318530717959f;
      down[a] = fabsf(p) > 3.14159265358979f * 0.9f || fabsf(r) >= 3.14159265358979f * 0.4f;
    }
    if (sc[0] != 0 && down[1]) { r0 += sc[0]; rs[0] += sc[0]; }                /* the opponent is on its side / back */
    if (sc[1] != 0 && down[0]) { r0 -= sc[1]; rs[1] += sc[1]; }
    rew[0] = r0;
    for (int a = 1; a < Aw; a++) rew[a] = 0;
    return;
  }
  if (d->task == MQE_TASK_ROTATION) {             /* go1_rotation_wrapper.py:46-50,90-93: agent 1 sees the mirrored scene */
    float* o1 = obs + 1 * D;
    o1[1] = -o1[1]; o1[4] = -o1[4]; o1[7] = -o1[7]; o1[10] = -o1[10];
    const float tgt = d->wrapper_param[0];
    const float* ob0 = s->obs_bag + (size_t)(e * A) * OBS_BAG;
    const float* ob1 = s->obs_bag + (size_t)(e * A + 1) * OBS_BAG;
    if (is_reset_call) {                          /* _init_extras (:30-38): only x is shifted by the target here */
      s->w_last[e * MAXA] = sqrtf((ob0[0] - tgt) * (ob0[0] - tgt) + ob0[1] * ob0[1]);
      for (int a = 0; a < Aw; a++) rew[a] = 0;
      return;
    }
    float r0 = 0.0f;
    if (sc[0] != 0 && ob0[0] > tgt) { r0 += sc[0]; rs[0] += sc[0]; }          /* success: agent 0 past the door line */
    if (sc[1] != 0 && ob1[0] > tgt) { r0 -= sc[1]; rs[1] += sc[1]; }          /* punishment: the opponent got there */
    if (sc[2] != 0) {
      /* `dis[:, :] -= self.target_pos` (:74) broadcasts the (num_envs,) target over the LAST axis, i.e. it is subtracted
       * from x AND y (and only runs for num_envs in {1, 2} upstream); kept */
      const float dis = sqrtf((ob0[0] - tgt) * (ob0[0] - tgt) + (ob0[1] - tgt) * (ob0[1] - tgt));
      if (dis < s->w_last[e * MAXA]) { r0 += sc[2]; rs[2] += sc[2]; }
      s->w_last[e * MAXA] = dis;
    }
    rew[0] = r0;
    for (int a = 1; a < Aw; a++) rew[a] = 0;
    return;
  }
  if (is_reset_call) { for (int a = 0; a < Aw; a++) rew[a] = 0; return; }
  float r_env = 0.0f;            /* the (N,1) reward column */
  float r_ag[MAXA] = {0, 0, 0, 0};
  uint8_t was_reset = s->reset_buf[e];
  if (d->task == MQE_TASK_GATE) {
    /* documented-but-commented semantics of go1_gate_wrapper.py:78-154 (live code returns 0: SURVEY F7) */
    float tsum = 0;
    for (int a = 0; a < A; a++) {
      const float* ob = s->obs_bag + (size_t)(e * A + a) * OBS_BAG;
      float tx = d->wrapper_param[0], ty = (a == 0 ? 1.0f : -1.0f) * d->wrapper_param[1];
      float dist = sqrtf((ob[0] - tx) * (ob[0] - tx) + (ob[1] - ty) * (ob[1] - ty));
      if (!s->w_have_last[e]) s->w_last[e * MAXA + a] = dist;
      tsum += s->w_last[e * MAXA + a] - dist;
      s->w_last[e * MAXA + a] = dist;
    }
    s->w_have_last[e] = 1;
    if (was_reset) tsum = 0;
    tsum *= sc[0];
    for (int a = 0; a < A; a++) r_ag[a] += tsum;
    rs[0] += tsum;
    float col = sc[1] * (float)s->collide_buf[e];
    for (int a = 0; a < A; a++) r_ag[a] += col;
    rs[1] += col;
    for (int a = 0; a < A; a++) {
      const float* ob = s->obs_bag + (size_t)(e * A + a) * OBS_BAG;
      if (ob[0] > s->gate_pos[e * 2] + 0.25f) { r_ag[a] += sc[2]; rs[2] += sc[2]; }
      const float* ob2 = s->obs_bag + (size_t)(e * A + (A - 1 - a)) * OBS_BAG;
      float d2 = (ob[0] - ob2[0]) * (ob[0] - ob2[0]) + (ob[1] - ob2[1]) * (ob[1] - ob2[1]);
      if (d2 < 0.25f) { float pn = div_ieee(sc[3], d2); r_ag[a] += pn; rs[3] += pn; }
    }
    float tot = 0;
    for (int a = 0; a < A; a++) tot += r_ag[a];
    for (int a = 0; a < A; a++) rew[a] = tot;                                  /* sum over agents, broadcast (:154) */
    return;
  }
  if (d->task == MQE_TASK_SHEEP) {   /* go1_sheep_wrapper.py:54-118 */
    float gate_x = s->gate_pos[e * 2];
    if (sc[0] != 0) {
      int cnt = 0;
      for (int p = 0; p < P; p++) if ((npc[p * 13] - s->env_origins[e * 3]) - gate_x > 0) cnt++;
      r_env = (float)cnt;                                                       /* assigned, unscaled (:73) */
      rs[0] += (float)cnt;
    }
    if (sc[1] != 0) { float c = sc[1] * (float)s->collide_buf[e]; r_env += c; rs[1] += c; }
    if (sc[2] != 0) {
      if (s->w_have_last[e]) {
        float xm = s->sheep_avg[e * 2] - s->w_last2[e * 2];
        if (s->w_delayed_reset[e]) xm = 0;
        float v = sc[2] * xm;
        r_env += v; rs[2] += v;
      }
      s->w_last2[e * 2] = s->sheep_avg[e * 2]; s->w_last2[e * 2 + 1] = s->sheep_avg[e * 2 + 1];
      s->w_have_last[e] = 1;
    }
    if (sc[3] != 0) {
      float acc = 0;
      for (int p = 0; p < P; p++) {
        float x = npc[p * 13] - s->env_origins[e * 3], y = npc[p * 13 + 1] - s->env_origins[e * 3 + 1];
        float dg = sqrtf((x - gate_x) * (x - gate_x) + (y - s->gate_pos[e * 2 + 1]) * (y - s->gate_pos[e * 2 + 1]));
        float v = expf(-dg / 2.0f) * sc[3];
        if (x >= gate_x) v = sc[3];
        acc += v;
      }
      r_env += acc; rs[3] += acc;
    }
    if (sc[4] != 0 || sc[5] != 0) {
      float v = sc[5] * (s->sheep_var[e] - 1.0f) + sc[4] * expf(s->sheep_var[e] / 2.0f - 1.0f);
      r_env += v; rs[4] += v;
    }
    s->w_delayed_reset[e] = was_reset;
    for (int a = 0; a < Aw; a++) rew[a] = r_env;
    return;
  }
  if (d->task == MQE_TASK_SEESAW) {   /* go1_seesaw_wrapper.py:48-120 */
    float xs = 0, zs = 0, y2 = 0;
    for (int a = 0; a < A; a++) {
      const float* ob = s->obs_bag + (size_t)(e * A + a) * OBS_BAG;
      if (!s->w_have_last[e]) s->w_last[e * MAXA + a] = ob[0];
      xs += ob[0] - s->w_last[e * MAXA + a];
      s->w_last[e * MAXA + a] = ob[0];
      zs += ob[2]; y2 += ob[1] * ob[1];
    }
    s->w_have_last[e] = 1;
    if (sc[0] != 0) { if (was_reset) xs = 0; xs *= sc[0]; r_env += xs; rs[0] += xs; }
    if (sc[1] != 0) { float v = sc[1] * (zs - 0.56f); r_env += v; rs[1] += v; }
    if (sc[2] != 0) { float v = sc[2] * (y2 - 0.5f); r_env += v; rs[2] += v; }
    if (sc[3] != 0) { float v = sc[3] * (float)s->collide_buf[e]; r_env += v; rs[3] += v; }
    if (sc[4] != 0) {
      const float* o0 = s->obs_bag + (size_t)(e * A) * OBS_BAG; const float* o1 = s->obs_bag + (size_t)(e * A + A - 1) * OBS_BAG;
      float d2 = (o0[0] - o1[0]) * (o0[0] - o1[0]) + (o0[1] - o1[1]) * (o0[1] - o1[1]);
      if (d2 < 0.25f) { float v = div_ieee(sc[4], d2); r_env += v; rs[4] += v; }
    }
    if (sc[5] != 0) {
      int cnt = 0;
      for (int a = 0; a < A; a++) { const float* ob = s->obs_bag + (size_t)(e * A + a) * OBS_BAG; if (ob[0] > 7.7f && ob[2] > 1.3f) cnt++; }
      float v = sc[5] * (float)cnt; r_env += v; rs[5] += v;
    }
    if (sc[6] != 0) { if (s->r_term[e] | s->p_term[e]) { r_env += sc[6]; rs[6] += sc[6]; } }
    for (int a = 0; a < Aw; a++) rew[a] = r_env;
    return;
  }
  if (d->task == MQE_TASK_PUSHBOX) {              /* go1_pushbox_wrapper.py:52-88 */
    float bx = npc[0] - s->env_origins_live[e * 3];
    if (sc[0] != 0 && s->w_have_last[e]) {
      float xm = bx - s->w_last2[e * 2];
      if (was_reset) xm = 0;                      /* x_movement[reset_ids] = 0 */
      float v = sc[0] * xm;
      r_env += v; rs[0] += v;
    }
    s->w_last2[e * 2] = bx;
    s->w_have_last[e] = 1;
    for (int a = 0; a < Aw; a++) rew[a] = r_env;
    return;
  }
  if (d->task == MQE_TASK_FOOTBALL_DEFENDER) {   /* go1_football_wrapper.py:57-91 */
    float bx = npc[0] - s->env_origins_live[e * 3], by = npc[1] - s->env_origins_live[e * 3 + 1];
    if (sc[0] != 0) { if (bx > s->gate_pos[e * 2]) { r_env += sc[0]; rs[0] += sc[0]; } }
    if (sc[1] != 0) {
      float dg = sqrtf((bx - s->gate_pos[e * 2]) * (bx - s->gate_pos[e * 2]) + (by - s->gate_pos[e * 2 + 1]) * (by - s->gate_pos[e * 2 + 1]));
      float v = sc[1] * expf(-dg / 3.0f);
      r_env += v; rs[1] += v;
    }
    for (int a = 0; a < Aw; a++) rew[a] = r_env;
    return;
  }
  for (int a = 0; a < Aw; a++) rew[a] = 0;
}

int mqo_wrapper_eval(mqo_sim* s, int is_reset_call) {
  s->wrapper_side_effects = 1;
  for (int e = 0; e < s->N; e++) {
    if (is_reset_call) { s->w_have_last[e] = 0; s->w_delayed_reset[e] = 0; }
    wrapper_env(s, e, is_reset_call, s->P ? s->root + ((size_t)e * (s->A + s->P) + s->A) * 13 : NULL);
  }
  s->wrapper_side_effects = 0;
  return 0;
}

/* scripted defender command, go1_football_defender.py:56-80 */
static void defender_command(const mqo_sim* s, int e, float* cmd3) {
  int A = s->A, P = s->P;
  const float* root = s->root + (size_t)e * (A + P) * 13;
  const float* dp = root + 2 * 13;
  const float* bp = root + A * 13;
  float gate[3] = {s->gate_pos[e * 2], s->gate_pos[e * 2 + 1], s->env_origins[e * 3 + 2]};
  float tp[3];
  for (int k = 0; k < 3; k++) tp[k] = 0.6f * bp[k] + 0.4f * gate[k];
  float yaw = s->obs_bag[(size_t)(e * A + 2) * OBS_BAG + 5];
  float yaw_to_gate = 3.1415927f + atan_ratio(gate[1] - dp[1], gate[0] - dp[0]);
  float yc = fminf(fmaxf(yaw_to_gate - yaw, -0.3f), 0.3f) / 0.3f;
  float tdg = sqrtf((tp[0] - gate[0]) * (tp[0] - gate[0]) + (tp[1] - gate[1]) * (tp[1] - gate[1]));
  float ddg = sqrtf((dp[0] - gate[0]) * (dp[0] - gate[0]) + (dp[1] - gate[1]) * (dp[1] - gate[1]));
  float xc = fminf(fmaxf(tdg - ddg, -0.5f), 0.5f);
  float yy = -fminf(fmaxf(gate[1] + div_ieee((tp[1] - gate[1]) * (dp[0] - gate[0]), tp[0] - gate[0]) - dp[1], -0.5f), 0.5f);
  cmd3[0] = xc; cmd3[1] = yy; cmd3[2] = yc;
}
int mqo_defender_command(mqo_sim* s, float* out /*[N,3]*/) { for (int e = 0; e < s->N; e++) defender_command(s, e, out + e * 3); return 0; }

/* ------------------------------------------------------------------------------------------ fused step */
/* wrapper.step + Go1.step: actions [N, Aw, 3] */
int mqo_step(mqo_sim* s, const float* actions) {
  const mqe_sim_desc* d = &s->d;
  int N = s->N, A = s->A, Aw = s->Aw;
  if (d->num_command_dims != 3) { snprintf(g_err, sizeof g_err, "mqo_step takes wrapper-level (N, A', 3) actions; use mqo_policy_step for this command layout"); return -7; }
  float* cmd = (float*)malloc((size_t)s->R * 3 * 4);
  static const float scale[3] = {2.0f, 0.5f, 0.5f};
  for (int e = 0; e < N; e++) {
    for (int a = 0; a < Aw; a++)
      for (int k = 0; k < 3; k++) {
        float v = actions[((size_t)e * Aw + a) * 3 + k];
        if (d->task != MQE_TASK_TUG) v = fminf(fmaxf(v, -1.0f), 1.0f);   /* wrapper clip (go1_sheep_wrapper.py:55); the tug wrapper has none */
        cmd[((size_t)e * A + a) * 3 + k] = d->task == MQE_TASK_PLAIN ? v : v * scale[k];
      }
    if (d->task == MQE_TASK_FOOTBALL_DEFENDER) defender_command(s, e, cmd + ((size_t)e * A + 2) * 3);
  }
  mqo_policy_step(s, cmd);
  free(cmd);
  for (int k = 0; k < d->decimation; k++) {
    mqo_compute_torques(s);
    mqo_simulate(s);
    mqo_post_decimation_step(s, k);
  }
  s->wrapper_side_effects = 1;
  mqo_post_physics_step(s);
  s->wrapper_side_effects = 0;
  return 0;
}

/* Go1.step for control types P / V / T (go1.py:42-44): clip to clip_actions (legged_robot.py:108-110), no policy */
int mqo_step_joint(mqo_sim* s, const float* actions12) {
  const mqe_sim_desc* d = &s->d;
  for (size_t i = 0; i < (size_t)s->R * 12; i++) s->actions[i] = fminf(fmaxf(actions12[i], -d->clip_actions), d->clip_actions);
  for (int k = 0; k < d->decimation; k++) {
    mqo_compute_torques(s);
    mqo_simulate(s);
    mqo_post_decimation_step(s, k);
  }
  s->wrapper_side_effects = 1;
  mqo_post_physics_step(s);
  s->wrapper_side_effects = 0;
  return 0;
}


/* ------------------------------------------------------------------------------------------ (f)4: the forward depth camera
 * legged_robot_field.py:23-93 (create_camera_sensor + attach_camera_to_body on the base link, FOLLOW_TRANSFORM) and :196-223
 * (get_camera_image_gpu_tensor(..., IMAGE_DEPTH)): Isaac Gym's rasteriser is closed, so the image is DEFINED here as a ray cast over
 * the geometry the physics collides with -- per pixel the distance along the optical axis to the first surface, reported negative,
 * -inf where nothing lies within `far` -- and this scalar caster is what the HIP kernel (csrc/kernels_camera.hpp) is held to.
 * Conventions (the spec of the image): camera frame = body-local (position, ZYX Euler) on the base link, optical axis +x, +z up; pixel
 * (0, 0) top-left; ray direction (1, yc, zc), yc = -(2 (j + 1/2) / W - 1) tan(hfov / 2), zc = -(2 (i + 1/2) / H - 1) tan(hfov / 2) H / W,
 * so the ray parameter IS the depth.  Surfaces: ground slab (or the relief map, marched in half cells and bisected six times), the
 * wall prisms (sphere tracing over the signed-distance map; inside a footprint the wall reaches from the ground to its top), the
 * OTHER robots' collision primitives, free NPC spheres / the free box, the 1-dof link (plank / door box or the tug's upright cylinder
 * as a capsule) and its platform, the scenery boxes.  An eye inside a closed shape does not see that shape. */
static real ray_sphere_o(const real* o, const real* dir, const real* c, real r, real best) {
  real oc[3] = {o[0] - c[0], o[1] - c[1], o[2] - c[2]};
  real a = dot3(dir, dir), b = dot3(dir, oc), cc = dot3(oc, oc) - r * r;
  real disc = b * b - a * cc;
  if (disc < 0 || cc < 0) return best;
  real t = (-b - (real)sqrt((double)disc)) / a;
  return (t > (real)1e-4 && t < best) ? t : best;
}
static real ray_capsule_o(const real* o, const real* dir, const real* c, const real* u, real r, real best) {
  /* segment c - u .. c + u swept by r: the infinite cylinder's quadratic where the foot falls between the ends, else the end spheres */
  real pa[3] = {c[0] - u[0], c[1] - u[1], c[2] - u[2]}, pb[3] = {c[0] + u[0], c[1] + u[1], c[2] + u[2]};
  real ba[3] = {2 * u[0], 2 * u[1], 2 * u[2]}, oa[3] = {o[0] - pa[0], o[1] - pa[1], o[2] - pa[2]};
  real baba = dot3(ba, ba);
  if (baba < (real)1e-12) return ray_sphere_o(o, dir, c, r, best);
  real bard = dot3(ba, dir), baoa = dot3(ba, oa), rdoa = dot3(dir, oa), oaoa = dot3(oa, oa), dd = dot3(dir, dir);
  real a = baba * dd - bard * bard, b = baba * rdoa - baoa * bard, cc = baba * oaoa - baoa * baoa - r * r * baba;
  real disc = b * b - a * cc;
  if (disc >= 0 && a > (real)1e-12) {
    real t = (-b - (real)sqrt((double)disc)) / a, y = baoa + t * bard;
    if (y > 0 && y < baba) return (t > (real)1e-4 && t < best) ? t : best;
  }
  return ray_sphere_o(o, dir, pb, r, ray_sphere_o(o, dir, pa, r, best));
}
static real ray_box_o(const real* o, const real* dir, const real* c, const real* R, const real* h, real best) {
  real oc[3] = {o[0] - c[0], o[1] - c[1], o[2] - c[2]}, ol[3], dl[3];
  for (int k = 0; k < 3; k++) { ol[k] = R[k] * oc[0] + R[3 + k] * oc[1] + R[6 + k] * oc[2]; dl[k] = R[k] * dir[0] + R[3 + k] * dir[1] + R[6 + k] * dir[2]; }
  real t0 = (real)1e-4, t1 = best;
  for (int k = 0; k < 3; k++) {
    if (fabs((double)dl[k]) < 1e-9) { if (fabs((double)ol[k]) > h[k]) return best; continue; }
    real ta = (-h[k] - ol[k]) / dl[k], tb = (h[k] - ol[k]) / dl[k];
    if (ta > tb) { real x = ta; ta = tb; tb = x; }
    if (ta > t0) t0 = ta;
    if (tb < t1) t1 = tb;
    if (t0 > t1) return best;
  }
  if (fabs((double)ol[0]) <= h[0] && fabs((double)ol[1]) <= h[1] && fabs((double)ol[2]) <= h[2]) return best;
  return t0;
}
static void link_frames(const mqo_sim* s, int env, int r, real R[NB][9], real p[NB][3]) {
  const mqe_robot_model* m = &s->d.robot;
  const float* rs = s->root + ((size_t)env * (s->A + s->P) + r) * 13;
  const float* dofs = s->dof + (size_t)env * s->ND * 2;
  real q[4] = {rs[3], rs[4], rs[5], rs[6]};
  real nq = (real)sqrt((double)(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]));
  for (int k = 0; k < 4; k++) q[k] /= nq;
  quat_to_mat(q, R[0]);
  for (int k = 0; k < 3; k++) p[0][k] = rs[k];
  for (int b = 1; b < NB; b++) {
    int pb = g_parent[b];
    real off[3] = {m->joint_offset[b][0], m->joint_offset[b][1], m->joint_offset[b][2]}, dd[3];
    mat3_vec(R[pb], off, dd);
    for (int k = 0; k < 3; k++) p[b][k] = p[pb][k] + dd[k];
    real ax[3] = {m->joint_axis[b][0], m->joint_axis[b][1], m->joint_axis[b][2]}, Rj[9];
    axis_angle_mat(ax, (real)dofs[(r * 12 + b - 1) * 2], Rj);
    mat3_mul(R[pb], Rj, R[b]);
  }
}
int mqo_render_depth(mqo_sim* s, float* out, int H, int W, float hfov_deg, const float* cam_pos3, const float* cam_rpy3, float far_m) {
  if (!s) { snprintf(g_err, sizeof g_err, "mqo_render_depth: null handle"); return -1; }
  const mqe_sim_desc* d = &s->d;
  const mqe_robot_model* m = &d->robot;
  if (!out || H <= 0 || W <= 0 || H * W > (1 << 16) || !(hfov_deg > 1.0f && hfov_deg < 179.0f) || !(far_m > 0.0f)) { snprintf(g_err, sizeof g_err, "mqo_render_depth: bad argument"); return -6; }
  const int A = s->A, P = s->P, N = s->N, npix = H * W;
  const real tan_h = (real)tan(0.5 * (double)hfov_deg * 3.14159265358979 / 180.0), tan_v = tan_h * (real)H / (real)W;
  const real hs = d->horizontal_scale;
  const int nx = d->sdf_nx, ny = d->sdf_ny;
  const int n_free = (d->npc_kind == MQE_NPC_BALL || d->npc_kind == MQE_NPC_SHEEP || d->npc_kind == MQE_NPC_BOX) ? P : 0;
  real Rc[9];
  {
    real Rz[9], Ry[9], Rx[9], T[9], ez[3] = {0, 0, 1}, ey[3] = {0, 1, 0}, ex[3] = {1, 0, 0};
    axis_angle_mat(ez, cam_rpy3[2], Rz); axis_angle_mat(ey, cam_rpy3[1], Ry); axis_angle_mat(ex, cam_rpy3[0], Rx);
    mat3_mul(Rz, Ry, T); mat3_mul(T, Rx, Rc);
  }
#pragma omp parallel for schedule(static) num_threads(mqo_team())
  for (int e = 0; e < N; e++) {
    real LR[MAXA][NB][9], Lp[MAXA][NB][3];
    for (int r = 0; r < A; r++) link_frames(s, e, r, LR[r], Lp[r]);
    const float* root = s->root + (size_t)e * (A + P) * 13;
    for (int a = 0; a < A; a++) {
      real Rw[9], o[3], cp[3] = {cam_pos3[0], cam_pos3[1], cam_pos3[2]}, t3[3];
      mat3_mul(LR[a][0], Rc, Rw);
      mat3_vec(LR[a][0], cp, t3);
      for (int k = 0; k < 3; k++) o[k] = Lp[a][0][k] + t3[k];
      for (int pix = 0; pix < npix; pix++) {
        const int pi = pix / W, pj = pix - pi * W;
        real loc[3] = {1, -((real)2 * ((real)pj + (real)0.5) / (real)W - 1) * tan_h, -((real)2 * ((real)pi + (real)0.5) / (real)H - 1) * tan_v}, dir[3];
        mat3_vec(Rw, loc, dir);
        real best = far_m;
        const real dxy = (real)sqrt((double)(dir[0] * dir[0] + dir[1] * dir[1]));
        /* ground */
        if (!s->ground_height) {
          if (dir[2] < (real)-1e-9) { real t = (d->ground_z - o[2]) / dir[2]; if (t > (real)1e-4 && t < best) best = t; }
        } else {
          const real st = dxy > (real)1e-6 ? (real)0.5 * hs / dxy : (real)0.05;
          real tp = 0;
          for (real t = st; t < best; t += st) {
            real gx, gy, pz = o[2] + t * dir[2];
            if (pz < d->ground_z + map_sample(s, s->ground_height, o[0] + t * dir[0], o[1] + t * dir[1], &gx, &gy)) {
              real lo = tp, hi = t;
              for (int it = 0; it < 6; it++) {
                real mid = (real)0.5 * (lo + hi);
                if (o[2] + mid * dir[2] < d->ground_z + map_sample(s, s->ground_height, o[0] + mid * dir[0], o[1] + mid * dir[1], &gx, &gy)) hi = mid; else lo = mid;
              }
              best = hi;
              break;
            }
            tp = t;
          }
        }
        /* wall prisms */
        if (dxy > (real)1e-6 && s->sdf) {
          real t = (real)1e-4;
          for (int it = 0; it < 400 && t < best; it++) {
            real px = o[0] + t * dir[0], py = o[1] + t * dir[1], pz = o[2] + t * dir[2];
            if (px / hs < 0 || py / hs < 0 || px / hs > (real)(nx - 1) || py / hs > (real)(ny - 1)) break;
            real gx, gy, sd = sdf_sample(s, px, py, &gx, &gy);
            if (sd <= (real)0.002) {
              if (pz <= wall_top_at(s, px, py) && pz >= d->ground_z - (real)1e-3) { best = t; break; }
              t += (real)0.25 * hs / dxy;
            } else t += (sd > (real)0.002 ? sd : (real)0.002) / dxy;
          }
        }
        /* the other robots' primitives */
        for (int r = 0; r < A; r++) {
          if (r == a) continue;
          for (int q = 0; q < m->n_prims; q++) {
            const int b = m->prim_body[q];
            real pc[3] = {m->prim_center[q][0], m->prim_center[q][1], m->prim_center[q][2]}, c[3];
            mat3_vec(LR[r][b], pc, c);
            for (int k = 0; k < 3; k++) c[k] += Lp[r][b][k];
            if (m->prim_type[q] == MQE_PRIM_BOX) {
              real hb[3] = {m->prim_half[q][0], m->prim_half[q][1], m->prim_half[q][2]};
              best = ray_box_o(o, dir, c, LR[r][b], hb, best);
            } else if (m->prim_type[q] == MQE_PRIM_CAPSULE) {
              real ax[3] = {m->prim_axis[q][0], m->prim_axis[q][1], m->prim_axis[q][2]}, u[3];
              mat3_vec(LR[r][b], ax, u);
              best = ray_capsule_o(o, dir, c, u, m->prim_half[q][0], best);
            } else best = ray_sphere_o(o, dir, c, m->prim_half[q][0], best);
          }
        }
        /* free NPCs */
        for (int p = 0; p < n_free; p++) {
          const float* ns = root + (A + p) * 13;
          real q[4] = {ns[3], ns[4], ns[5], ns[6]}, Rb[9], pb[3] = {ns[0], ns[1], ns[2]};
          real nq = (real)sqrt((double)(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]));
          for (int k = 0; k < 4; k++) q[k] /= nq;
          quat_to_mat(q, Rb);
          if (d->npc_kind == MQE_NPC_BOX) {
            real hb[3] = {d->npc_box_half[0], d->npc_box_half[1], d->npc_box_half[2]};
            best = ray_box_o(o, dir, pb, Rb, hb, best);
          } else
            for (int k = 0; k < d->npc_n_spheres; k++) {
              real sc[3] = {d->npc_sphere_center[k][0], d->npc_sphere_center[k][1], d->npc_sphere_center[k][2]}, c[3];
              mat3_vec(Rb, sc, c);
              for (int i = 0; i < 3; i++) c[i] += pb[i];
              best = ray_sphere_o(o, dir, c, d->npc_sphere_radius[k], best);
            }
        }
        const real I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        if (d->npc_kind == MQE_NPC_SEESAW) {
          real sb[3] = {root[A * 13], root[A * 13 + 1], root[A * 13 + 2]};
          if (d->seesaw_base_half[0] > 0.0f) { real hb[3] = {d->seesaw_base_half[0], d->seesaw_base_half[1], d->seesaw_base_half[2]}; best = ray_box_o(o, dir, sb, I3, hb, best); }
          real piv[3] = {sb[0] + d->seesaw_joint_offset[0], sb[1] + d->seesaw_joint_offset[1], sb[2] + d->seesaw_joint_offset[2]};
          const real th = s->dof[((size_t)e * s->ND + 12 * A) * 2];
          real Rp[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
          if (d->seesaw_axis == 3) piv[1] += th;
          else { real ax[3] = {0, d->seesaw_axis == 2 ? 0 : 1, d->seesaw_axis == 2 ? 1 : 0}; axis_angle_mat(ax, th, Rp); }
          real pcl[3] = {d->seesaw_plank_center[0], d->seesaw_plank_center[1], d->seesaw_plank_center[2]}, pc[3];
          mat3_vec(Rp, pcl, pc);
          for (int k = 0; k < 3; k++) pc[k] += piv[k];
          if (d->seesaw_link_cylinder) {
            real hz = d->seesaw_plank_half[2] - d->seesaw_plank_half[0];
            real u[3] = {0, 0, hz > 0 ? hz : 0};
            best = ray_capsule_o(o, dir, pc, u, d->seesaw_plank_half[0], best);
          } else { real hb[3] = {d->seesaw_plank_half[0], d->seesaw_plank_half[1], d->seesaw_plank_half[2]}; best = ray_box_o(o, dir, pc, Rp, hb, best); }
        }
        if (d->npc_kind == MQE_NPC_STATIC)
          for (int bx = 0; bx < d->n_static_boxes; bx++) {
            real c[3] = {root[A * 13] + d->static_box_center[bx][0], root[A * 13 + 1] + d->static_box_center[bx][1], root[A * 13 + 2] + d->static_box_center[bx][2]};
            real hb[3] = {d->static_box_half[bx][0], d->static_box_half[bx][1], d->static_box_half[bx][2]};
            best = ray_box_o(o, dir, c, I3, hb, best);
          }
        out[((size_t)e * A + a) * npix + pix] = best < far_m ? -(float)best : -INFINITY;
      }
    }
  }
  return 0;
}

#ifdef _OPENMP
#include <omp.h>
static inline int mqo_team(void) { return g_team > 0 ? g_team : omp_get_max_threads(); }
int mqo_num_threads(void) { return mqo_team(); }
void mqo_set_num_threads(int n) { if (n > 0) g_team = n; }
#else
static inline int mqo_team(void) { return 1; }
int mqo_num_threads(void) { return 1; }
void mqo_set_num_threads(int n) { (void)n; (void)g_team; }
#endif
