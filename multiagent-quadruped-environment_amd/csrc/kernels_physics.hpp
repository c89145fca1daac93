// kernels_physics.hpp -- k_substeps / k_simulate: articulated rigid-body dynamics with contact, one 5 ms substep at a time.
// Replaces gym.set_dof_actuation_force_tensor + gym.simulate + refresh_* (reference go1.py:52-56, legged_robot.py:
// 122-124); the reference delegates this to Isaac Gym / PhysX, so the algorithm is this build's own.
//
// Mapping: ONE ENVIRONMENT PER 64-LANE WAVEFRONT (one wave per workgroup).  Lanes take different roles per phase:
//   body lanes    (A*13 + P)   forward kinematics by tree level, spatial inertia / bias wrench about the base origin,
//                              composite sums up the 3-link leg chains; a child's parent is the neighbouring lane, so frames,
//                              composites and joint axes travel through DPP wave shifts, not LDS
//   hip lanes     (A*4)        3x3 leg block: Mll^-1, its Cholesky factor, G = Mbl Mll^-1;   (A*6) lanes: 6x6 base Schur complement,
//                              Cholesky-factored -- the inverse mass matrix exists only as these factors (M^-1 = T T^T, see SIDE_STRIDE)
//   sphere lanes  (2*27 / P)   collision spheres vs ground (plane or relief map) / wall signed-distance field / static scenery boxes /
//                              1-dof link (plank, door, disc) / free box / other actors' spheres, compacted with ballots
//                              into a bounded, canonically ordered contact list
//   contact lanes (<= maxc)    the side records Phi = J T (28 floats) and the contact's own 3x3 block
//   sweep                      projected Gauss-Seidel on w = sum Phi^T lambda (one float per generalized coordinate): either one
//                              DPP row of 16 lanes per ACTOR (lane = coordinate; scenes of <= 4 actors) or one lane per contact
//   dof lanes     (<= 2 x 64)  unconstrained velocity, dv = T w, joint limits, integration (velocities in LDS)
// Link frames, the factors, contact and side records live in LDS (layout: phys_lds_layout: 9-10 KiB for two robots and one more
// object = 16 envs per CU); the records that are moved whole are laid out in 16 B words and accessed with ds_read_b128 /
// ds_write_b128.  With k_substeps the state is read from and written to HBM once per env.step(), coalesced (the env-major rows
// of one env are contiguous), and the actuator network / PD law of every substep runs inside the same wavefront.
#pragma once
#include "mqe_common.hpp"

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ V3 ld3(const float* p) { return v3(p[0], p[1], p[2]); }
__device__ __forceinline__ void st3(float* p, V3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
__device__ __forceinline__ V3 mat_vec(const float* R, V3 v) {
  return v3(R[0] * v.x + R[1] * v.y + R[2] * v.z, R[3] * v.x + R[4] * v.y + R[5] * v.z, R[6] * v.x + R[7] * v.y + R[8] * v.z);
}
// symmetric 3x3 stored (xx, yy, zz, xy, xz, yz)
__device__ __forceinline__ V3 sym_vec(const float* S, V3 v) {
  return v3(S[0] * v.x + S[3] * v.y + S[4] * v.z, S[3] * v.x + S[1] * v.y + S[5] * v.z, S[4] * v.x + S[5] * v.y + S[2] * v.z);
}

// value of the neighbouring lane through the DPP wave shifts of the VALU (a modifier of v_mov, no LDS crossbar round trip as
// in ds_bpermute / __shfl): lane i reads lane i + 1 (wave_shl:1) resp. lane i - 1 (wave_shr:1); the end lanes read zero (bound_ctrl: no
// "old" value to keep, so no copy in front of the shift, and a value with a single consumer becomes that instruction's DPP operand)
template <int CTRL> __device__ __forceinline__ float dpp_take(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_next(float x) { return dpp_take<0x130>(x); }
__device__ __forceinline__ float lane_prev(float x) { return dpp_take<0x138>(x); }

// sum over the 16 lanes of a DPP row, delivered to every lane of the row: four butterfly steps (lane ^ 1, lane ^ 2 as quad
// permutations, then the mirror inside each half row and inside the row).  Both partners of a step add the same two operands, so
// all 16 lanes end with bitwise the same sum.
__device__ __forceinline__ float row16_sum(float x) {
  x += dpp_take<0xB1>(x);      // quad_perm [1,0,3,2]
  x += dpp_take<0x4E>(x);      // quad_perm [2,3,0,1]
  x += dpp_take<0x141>(x);     // row_half_mirror
  x += dpp_take<0x140>(x);     // row_mirror
  return x;
}
// maximum over the wave of a non-negative value, delivered to every lane: DPP butterfly inside each row of 16, then the four rows
// through scalar registers
// (HALF = true: over each half-wave of 32 lanes separately -- the two rows of a half through the xor-16 swizzle, no lane of the other
// half is read, so the two halves may sit in different branches)
template <bool HALF> __device__ __forceinline__ float wave_max_nonneg(float x) {
  x = fmaxf(x, dpp_take<0xB1>(x)); x = fmaxf(x, dpp_take<0x4E>(x)); x = fmaxf(x, dpp_take<0x141>(x)); x = fmaxf(x, dpp_take<0x140>(x));
  if (HALF) return fmaxf(x, __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), 0x401F)));
  const int xi = __float_as_int(x);
  const float a = __int_as_float(__builtin_amdgcn_readlane(xi, 0)), b = __int_as_float(__builtin_amdgcn_readlane(xi, 16));
  const float c = __int_as_float(__builtin_amdgcn_readlane(xi, 32)), d = __int_as_float(__builtin_amdgcn_readlane(xi, 48));
  return fmaxf(fmaxf(a, b), fmaxf(c, d));
}
// value of lane ^ 16 (the same position in the neighbouring row): ds_swizzle in bit-mask mode, xor 0x10 -- no LDS access
__device__ __forceinline__ float row_partner(float x) {
  return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), 0x401F));
}

// sphere (centre c, radius r) vs box (centre bc, rotation R, half extents h): signed distance, world normal box->sphere
__device__ __forceinline__ float sphere_box(V3 c, float r, V3 bc, const float* R, V3 h, V3& n) {
  const V3 d = c - bc;
  const float pl[3] = {R[0] * d.x + R[3] * d.y + R[6] * d.z, R[1] * d.x + R[4] * d.y + R[7] * d.z, R[2] * d.x + R[5] * d.y + R[8] * d.z};
  const float hh[3] = {h.x, h.y, h.z};
  float dl[3], nl[3] = {0, 0, 0};
  bool inside = true;
  for (int k = 0; k < 3; k++) {
    const float q = fminf(fmaxf(pl[k], -hh[k]), hh[k]);
    dl[k] = pl[k] - q;
    if (dl[k] != 0.0f) inside = false;
  }
  float sd;
  if (!inside) {
    const float dist = sqrtf(dl[0] * dl[0] + dl[1] * dl[1] + dl[2] * dl[2]);
    nl[0] = dl[0] / dist; nl[1] = dl[1] / dist; nl[2] = dl[2] / dist;
    sd = dist - r;
  } else {
    int ax = 0; float best = hh[0] - fabsf(pl[0]);
    for (int k = 1; k < 3; k++) { const float mm = hh[k] - fabsf(pl[k]); if (mm < best) { best = mm; ax = k; } }
    nl[0] = ax == 0 ? (pl[0] >= 0 ? 1.0f : -1.0f) : 0.0f;
    nl[1] = ax == 1 ? (pl[1] >= 0 ? 1.0f : -1.0f) : 0.0f;
    nl[2] = ax == 2 ? (pl[2] >= 0 ? 1.0f : -1.0f) : 0.0f;
    sd = -best - r;
  }
  n = mat_vec(R, v3(nl[0], nl[1], nl[2]));
  return sd;
}

// sphere (centre c, radius r) vs upright solid cylinder (centre bc, radius rc, half height hh): signed distance, normal cylinder->sphere
__device__ __forceinline__ float sphere_vcyl(V3 c, float r, V3 bc, float rc, float hh, V3& n) {
  const float dx = c.x - bc.x, dy = c.y - bc.y, dz = c.z - bc.z;
  const float rho = sqrtf(dx * dx + dy * dy);
  const float ux = rho > 1e-9f ? dx / rho : 1.0f, uy = rho > 1e-9f ? dy / rho : 0.0f;
  const float er = rho - rc, ez = fabsf(dz) - hh, sz = dz < 0 ? -1.0f : 1.0f;
  if (er <= 0 && ez <= 0) {                // centre inside: leave through the nearer surface
    if (er > ez) { n = v3(ux, uy, 0); return er - r; }
    n = v3(0, 0, sz); return ez - r;
  }
  const float pr = er > 0 ? er : 0.0f, pz = ez > 0 ? ez : 0.0f;
  const float dist = sqrtf(pr * pr + pz * pz);
  n = v3(ux * pr / dist, uy * pr / dist, sz * pz / dist);
  return dist - r;
}

// sphere (centre c, radius r) vs capsule (centre cq, half-segment u, radius rq; u = 0: a sphere): signed distance and the unit normal
// from the capsule to the sphere through the closest point of the segment; false when the centre lies on the segment itself
__device__ __forceinline__ bool sphere_capsule(V3 c, float r, V3 cq, V3 u, float rq, float& sd, V3& n) {
  const V3 d = c - cq;
  const float uu = dot(u, u);
  float t = 0.0f;
  if (uu > 0.0f) t = fminf(fmaxf(dot(d, u) / uu, -1.0f), 1.0f);
  const V3 e = d - t * u;
  const float dist = sqrtf(dot(e, e));
  sd = dist - r - rq;
  n = (1.0f / dist) * e;
  return dist > 1e-9f;
}

// Closest approach of the segment p0 + t (p1 - p0), t in [0, 1], swept by the radius r, to a box (centre bc, rotation R, half extents h):
// the signed distance to a convex set is convex along a line -> golden-section search, two first evaluations and sixteen refinements (the bracket ends at 0.05 % of the segment), the
// sequence of oracle/mqe_oracle.c::seg_box.  Returns the signed distance at the final t; tb = that t, n = box -> point, pt = the point.
__device__ __forceinline__ float seg_box_dev(V3 p0, V3 p1, float r, V3 bc, const float* R, V3 h, float& tb, V3& n, V3& pt) {
  const float gr = 0.6180339887498949f;
  const V3 dp = p1 - p0;
  float a = 0.0f, b = 1.0f;
  V3 nn;
  float c = b - gr * (b - a), dd = a + gr * (b - a);
  float fc = sphere_box(p0 + c * dp, r, bc, R, h, nn);
  float fd = sphere_box(p0 + dd * dp, r, bc, R, h, nn);
#pragma clang loop unroll(disable)
  for (int it = 0; it < 16; it++) {
    const bool left = fc < fd;
    float tn;
    if (left) { b = dd; dd = c; fd = fc; c = b - gr * (b - a); tn = c; }
    else { a = c; c = dd; fc = fd; dd = a + gr * (b - a); tn = dd; }
    const float fn = sphere_box(p0 + tn * dp, r, bc, R, h, nn);
    if (left) fc = fn; else fd = fn;
  }
  tb = 0.5f * (a + b);
  pt = p0 + tb * dp;
  return sphere_box(pt, r, bc, R, h, n);
}
// Edge contact of one robot primitive (lane-local data) with a convex box: a capsule's AXIS against the box; a box primitive against the
// box's own axis segment when the box is degenerate (a wall's vertical edge: h = (0, 0, L / 2)).  The rules of oracle/mqe_oracle.c::edge_vs_box:
// kept between the end points only (5 .. 95 % of a capsule's axis), not next to a feature point on the axis, not when tunnelled.
// ptype: MQE_PRIM_CAPSULE / MQE_PRIM_BOX; cq / uq: centre, half-axis; rq: capsule radius; hb / Rb: box primitive's half extents / rotation.
__device__ __forceinline__ bool edge_vs_box_dev(int ptype, V3 cq, V3 uq, float rq, V3 hb, const float* Rb, float tf0, float tf1,
                                                V3 bc, const float* R, V3 h, bool is_wall_edge, float& sd, V3& n, V3& pa) {
  const bool cap = ptype == MQE_PRIM_CAPSULE;
  if (!cap && !is_wall_edge) return false;
  // one call site: the capsule's axis against the obstacle box, or the wall edge (the obstacle's own z axis) against the box primitive
  const V3 ez = v3(R[2] * h.z, R[5] * h.z, R[8] * h.z);
  const V3 p0 = cap ? cq - uq : bc - ez, p1 = cap ? cq + uq : bc + ez;
  float Rs[9];
#pragma unroll
  for (int k = 0; k < 9; k++) Rs[k] = cap ? R[k] : Rb[k];
  float tb; V3 nn, pt;
  const float v = seg_box_dev(p0, p1, cap ? rq : 0.0f, cap ? bc : cq, Rs, cap ? h : hb, tb, nn, pt);
  if (cap) {
    if (!(tb > 0.05f && tb < 0.95f)) return false;
    if ((tb > tf0 - 0.1f && tb < tf0 + 0.1f) || (tb > tf1 - 0.1f && tb < tf1 + 0.1f)) return false;
    if (v + rq < 0.0f) return false;
    sd = v; n = nn; pa = pt - rq * nn;
    return true;
  }
  if (v < -0.02f) return false;
  sd = v; n = v3(-nn.x, -nn.y, -nn.z); pa = pt;
  return true;
}

// desc.edge_contacts bit 4 (oracle/mqe_oracle.c::edge_vs_box, box primitives): the twelve edges of a convex box (centre bc, rotation R, half
// extents h) against a BOX primitive of a robot (centre cq, rotation Rb, half extents hb) -- an edge of the plank / the free box / a scenery
// box cutting into the trunk between its corners.  Per edge the closest approach of the segment to the primitive (seg_box_dev, radius 0),
// kept between the edge's end points (5 .. 95 %: the ends are the box's corners) and not deeper than 2 cm (tunnelled: left to the feature
// points); the deepest edge wins.  n points from the obstacle to the robot, pa is the edge point.
__device__ __forceinline__ bool box_edges_vs_prim_dev(V3 cq, V3 hb, const float* Rb, V3 bc, const float* R, V3 h, float& sd, V3& n, V3& pa) {
  bool found = false;
#pragma clang loop unroll(disable)
  for (int e = 0; e < 12; e++) {
    const int ax = e >> 2;                                    // the edge runs along axis ax; bits 0, 1: the signs of the two other coordinates
    const float s1 = (e & 1) ? 1.0f : -1.0f, s2 = (e & 2) ? 1.0f : -1.0f;
    // component c of the end points: -+h_c along the edge's axis, s1 h_c on axis (ax + 1) % 3, s2 h_c on axis (ax + 2) % 3
    const float x0 = ax == 0 ? -h.x : (ax == 2 ? s1 : s2) * h.x, x1 = ax == 0 ? h.x : x0;
    const float y0 = ax == 1 ? -h.y : (ax == 0 ? s1 : s2) * h.y, y1 = ax == 1 ? h.y : y0;
    const float z0 = ax == 2 ? -h.z : (ax == 1 ? s1 : s2) * h.z, z1 = ax == 2 ? h.z : z0;
    const V3 p0 = bc + mat_vec(R, v3(x0, y0, z0)), p1 = bc + mat_vec(R, v3(x1, y1, z1));
    float tb; V3 nn, pt;
    const float v = seg_box_dev(p0, p1, 0.0f, cq, Rb, hb, tb, nn, pt);
    if (!(tb > 0.05f && tb < 0.95f)) continue;
    if (v < -0.02f) continue;
    if (!found || v < sd) { found = true; sd = v; n = v3(-nn.x, -nn.y, -nn.z); pa = pt; }
  }
  return found;
}

// sin and cos of a joint angle in ~25 instructions instead of libm's ~100 + ~100 (each with its own argument reduction and a
// Payne-Hanek path that a joint angle never takes): one reduction to r = q - k pi/2 (three Cody-Waite steps, exact for |k| <= 5), the
// single-precision minimax kernels on |r| <= pi/4, quadrant swap / sign.  <= 1.5 ulp (9e-8 absolute) against the double-precision
// functions on |q| <= 8 (measured on a 1e-5 grid; ocml's sinf / cosf are 1-2 ulp); beyond that -- no joint of the URDF gets there, a
// state written through the API could -- libm.
__device__ __forceinline__ void joint_sincos(float q, float& s, float& c) {
  if (__builtin_expect(fabsf(q) > 8.0f, 0)) { s = sinf(q); c = cosf(q); return; }
  const float kf = __builtin_rintf(q * 0.63661977236758134f);
  float r = __builtin_fmaf(kf, -1.57079601287841796875f, q);
  r = __builtin_fmaf(kf, -3.1391647326017846353352069854736328125e-7f, r);
  r = __builtin_fmaf(kf, -5.390302529957764765544681040410068817436695098876953125e-15f, r);
  const float r2 = r * r;
  float sp = __builtin_fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
  sp = __builtin_fmaf(sp, r2, -1.6666654611e-1f);
  const float sr = __builtin_fmaf(sp * r2, r, r);
  float cp = __builtin_fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
  cp = __builtin_fmaf(cp, r2, 4.166664568298827e-2f);
  const float cr = __builtin_fmaf(cp * r2, r2, __builtin_fmaf(r2, -0.5f, 1.0f));
  const int k = (int)kf;
  const float ss = (k & 1) ? cr : sr, cc = (k & 1) ? sr : cr;
  s = (k & 2) ? -ss : ss;
  c = ((k + 1) & 2) ? -cc : cc;
}

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}

// Record strides are multiples of 4 floats (16 B words) that are NOT multiples of 32: with 32 banks of 4 B a stride of 128 B puts the
// same word of every lane's record into the same banks (26 body lanes storing a float4 each = a 26-way conflict; PMC, round 3: 31 % of
// the LDS-busy cycles of k_substeps were bank-conflict stalls, the LDS the busiest unit of the CU at 60 %).  36 / 20 / 28 floats walk
// through all 32 banks in 8 lanes: 16 B accesses of neighbouring lanes are conflict-free.  The records that live in the scratch
// union or on top of the link records (LEGC, FCOL, SREC) are padded for free; the link and contact records cost LDS (+ 672 B for two
// robots), which only robot-only scenes can afford inside the 10 KiB that keep 16 envs on a CU: `pad` in phys_lds_layout /
// PhysPad<TP> (go1gate k_substeps 127.6 -> 122.2 us).
template <int TP> struct PhysPad { static constexpr int on = TP == 0 ? 1 : 0; };
#define BODY_STRIDE_OF(pad) ((pad) ? 36 : 32)
#define CON_STRIDE_OF(pad) ((pad) ? 20 : 16)
// leg block record in 16 B words (lives to the end of the substep): Mll^-1 (sym6: 00,11,22,10,20,21) at 0, its Cholesky factor
// Lm (Mll^-1 = Lm Lm^T; l00,l10,l11,l20,l21,l22) at 6, G = Mbl Mll^-1 (6 x 3) at 12; the Schur term C = G Mbl^T (upper
// triangle, 21 values, 6 words) is scratch of its own (LEGC_STRIDE)
#define LEG_STRIDE 32
#define LEG_MI 0
#define LEG_LM 6
#define LEG_G 12
#define LEGC_STRIDE 28
#define FCOL_STRIDE 20      // one hip's composite (16 floats) on its way to the base lane
// Contact side record (7 words): Phi = the side's Jacobian in coordinates in which the actor's inverse mass matrix is the
// identity.  Robot: M^-1 = [S^-1, -S^-1 G; -G^T S^-1, Mll^-1 + G^T S^-1 G] = T T^T with T = [F 0; -G^T F, Lm] (S^-1 = F F^T, F upper
// triangular; Mll^-1 = Lm Lm^T per leg), so Phi = J T = [ U | Z' ],  U = (J_base - J_leg G^T) F (3 x 6),  Z' = J_leg Lm (3 x 3, the
// <= 3 joints of the chain to the touching link).  Free body / 1-dof link: U = J M^-1/2, no Z'.  With these
//   K(c, c') = Phi_c Phi_c'^T   and   M^-1 J^T lambda = T Phi^T lambda,
// so the projected Gauss-Seidel sweep runs on  w = sum_c Phi_c^T lambda_c  (one float per generalized coordinate, LDS) and needs
// neither the coupling blocks K (5 kB on go1gate, 21 kB on go1sheep-hard) nor M^-1 J^T, and  dv = T w  falls out at the end.
//   floats 0-17: U[q][m] at q * 6 + m;  18-26: Z'[q][i] at 18 + q * 3 + i;  27: (leg + 1) | columns << 4
#define SIDE_STRIDE 28
#define SIDE_Z 18
#define SIDE_INFO 27
// link record: rotation (9), origin (3), joint axis (3) = four 16 B words, what the collision and Jacobian phases read back; the
// base records also carry angular velocity, origin velocity and the two bias accelerations for their hips (the other links
// hand those to their children through registers)
enum { B_R = 0, B_P = 9, B_A = 12, B_W = 16, B_VP = 19, B_AL = 22, B_AP = 25 };
__device__ __forceinline__ void body_store(float* rec, const float* R, V3 p, V3 a) {
  float4* r4 = reinterpret_cast<float4*>(rec);
  r4[0] = make_float4(R[0], R[1], R[2], R[3]); r4[1] = make_float4(R[4], R[5], R[6], R[7]);
  r4[2] = make_float4(R[8], p.x, p.y, p.z); r4[3] = make_float4(a.x, a.y, a.z, 0.0f);
}
// contact record in 16 B words: [actor A, link A, actor B (-1: static), link B] [point, separation] [normal, reported body A]
// [contact force on A (written after the sweep of the last substep), reported body B]; the tangent frame is a function of the normal
enum { C_IDS = 0, C_P = 4, C_SD = 7, C_N = 8, C_REPA = 11, C_F = 12, C_REPB = 15 };
__device__ __forceinline__ void contact_tangents(V3 n, V3& t1, V3& t2) {
  const V3 aa = fabsf(n.z) > 0.7f ? v3(1, 0, 0) : v3(0, 0, 1);
  t1 = cross(aa, n);
  t1 = (1.0f / sqrtf(dot(t1, t1))) * t1;
  t2 = cross(n, t1);
}
__device__ __forceinline__ void con_store(float* cr, int a, int linkA, int b, int linkB, V3 p, V3 n, float sd, int repA, int repB) {
  float4* c4 = reinterpret_cast<float4*>(cr);
  c4[0] = make_float4(__int_as_float(a), __int_as_float(linkA), __int_as_float(b), __int_as_float(linkB));
  c4[1] = make_float4(p.x, p.y, p.z, sd);
  c4[2] = make_float4(n.x, n.y, n.z, __int_as_float(repA));
  c4[3] = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(repB));
}

// Contact slots of an env = the sum of the per-actor caps of its one-sided contacts (8 per robot, cap_npc per NPC) -- the pool the two-actor
// contacts share (at most half of it) in the scenes of at most four actors -- plus, for the larger scenes (flocks, 2 vs 2 + ball), eight slots
// that only two-actor contacts can take: a packed flock under two fallen robots fills every one-sided slot and still keeps its robot-sheep
// contacts.  One lane per contact in the sweep: at most 64; mqe_sim_create refuses a scene whose caps do not fit (round 5 clipped the
// sum at 40, so the last sheep of a 16-sheep flock could lose their ground contacts).
__host__ __device__ inline int mqe_maxc_uncapped(int A, int P, int cap_npc) { return 8 * A + cap_npc * P + (A + P > 4 ? 8 : 0); }
__host__ __device__ inline int mqe_maxc(int A, int P, int cap_npc) { const int v = mqe_maxc_uncapped(A, P, cap_npc); return v > 64 ? 64 : v; }
#define MQE_LIMIT_PASSES 4   // Gauss-Seidel passes of the joint position / speed limits per substep (oracle: the same constant)
#define CAP_ROBOT 8     // terrain / static-object contacts kept per robot (spheres are priority ordered: feet first)
// NPCs keep m->cap_npc one-sided contacts each (2; a box resting on a face 4): per-actor caps so that no actor starves the ones after it

struct PhysLds {   // float offsets into dynamic LDS
  int root, dof, tau, body, rhs, acc, acth, fcol, leg, legc, basei, sinv, sph, prim, con, side, phi, srec, wacc, total;
};
// Row sweep (scenes of <= 4 actors): ONE record per contact, in the order a sweep step reads it -- three 16 B words that every lane of the
// row reads (broadcast), then ten 3-float slots, one per lane of the row:
//   [0..3] mu, 1/d00, 1/d11, 1/d22   [4..7] d10, d20, d21, info   [8..11] lambda (3), running separation
//   [12 + 3 k + q], k = 0..8: column k of side A's Phi (k < 6: U[q][k]; 6..8: Z'[q][k - 6]) for the rows q = normal, tangent 1, tangent 2
//   [12 + 27 + q]: slot 9 = (u*_n - bias, u*_t1, u*_t2) -- lane 9 multiplies it by 1, so the butterfly sums arrive as u* - bias + Phi w
//   [42] bias, [43] u*_n (the contact's own lane keeps both for the separation updates of the temporal solver)
// info = first-joint offset of the leg (0, 3, 6, 9) | lanes << 8 | first coordinate << 16.  Side B of a two-actor contact: its ten slots
// (the tenth zero) and its info word at [30] in the pair area.  Round 6: the record used to be two (side record 28 + solve record 20 floats,
// column k at q * 6 + k resp. 18 + q * 3 + k - 6, the leg offset in a word of its own); a step now forms two addresses instead of
// seven and adds u* - bias inside the butterfly -- 52 -> 37 vector instructions per step.
#define RS_STRIDE 44
#define RS_SLOT 12
#define RS_BIAS 42
#define RS_USN 43
#define RSB_STRIDE 32
#define RSB_INFO 30
__host__ __device__ inline int mqe_maxpair(int maxc) { return maxc / 2; }    // two-actor contacts kept per env
__host__ __device__ inline PhysLds phys_lds_layout(int A, int P, int ND, int nbody, int ndof, int nsph, int nprim, int maxc, int rowgs, int pad) {
  const int BODY_STRIDE = BODY_STRIDE_OF(pad), CON_STRIDE = CON_STRIDE_OF(pad);
  // Regions that live to the end of the substep first; then the link records; then ONE scratch area used three times over: by the
  // CRBA / Schur scratch (dead once the factors exist), by the collision geometry in world coordinates -- the feature points of the
  // robots / collision spheres of the NPCs (16 B each) and the robots' primitives (32 B each: centre + bounding radius, capsule
  // half-segment + radius) -- live during contact generation only, and by side B of the two-actor contacts, written afterwards.
  // Contact sides are slot-allocated: side A of contact c -> slot c, side B of the k-th two-actor contact -> slot k of its own area
  // (terrain contacts have no B).  The mass-matrix inverse is kept in factored form only (per leg Mll^-1, its Cholesky factor and G;
  // per robot S^-1 and its factor F) and the contact problem in the Phi form of SIDE_STRIDE -- no coupling blocks.
  PhysLds L; int o = 0;
  L.root = o; o += (A + P) * 13;
  L.dof = o; o += ND * 2;
  L.tau = o; o += 12 * A;
  o = (o + 3) & ~3;
  L.rhs = o; o += (ndof + 3) & ~3;                          // generalized bias, later v* / the solved velocity (one per dof)
  L.acc = o; o += (ndof + 3) & ~3;                          // one scratch float per dof: reduced right-hand sides, J^T lambda sums
  L.acth = o; o += 4 * 12 * A;                              // k_substeps: actuator-net history of every joint (two past errors, two past velocities)
  L.leg = o; o += A * 4 * LEG_STRIDE;
  L.sinv = o; o += A * 72;                                  // per robot: S^-1 (36), then F with S^-1 = F F^T (upper triangular, stored 6 x 6)
  L.con = o; o += maxc * CON_STRIDE;
  L.body = o; o += nbody * BODY_STRIDE;
  // row sweep (scenes of <= 4 actors): side A of every contact and the per-contact solve record, written over the link records once the
  // last Jacobian row has been read
  L.phi = L.body; L.srec = L.phi;
  if (rowgs && L.phi + maxc * RS_STRIDE > o) o = L.phi + maxc * RS_STRIDE;
  const int scratch = o;
  L.fcol = o; o += A * 5 * FCOL_STRIDE;                                  // the four hip composites of every robot and the base body's own, on their way into the base block
  L.legc = o; o += A * 4 * LEGC_STRIDE;
  L.basei = o; o += A * 24;                                 // per robot: upper triangle of the base block, then of the Schur complement (21 values)
  L.sph = scratch; L.prim = scratch + nsph * 4;
  if (L.prim + nprim * 8 > o) o = L.prim + nprim * 8;     // two arrays of 16 B words: [centre, bounding radius] x nprim, then [half-segment, radius] x nprim
  L.side = scratch;                                         // side B of the two-actor contacts only (side A: registers / the phi area)
  if (scratch + mqe_maxpair(maxc) * RSB_STRIDE > o) o = scratch + mqe_maxpair(maxc) * RSB_STRIDE;      // (RSB_STRIDE >= SIDE_STRIDE: either sweep's side B fits)
  // temporal Gauss-Seidel: W = sum over the finished position iterations of w (one float per generalized coordinate), behind side B in
  // the scratch area (the factorisation scratch and the collision geometry are dead when the sweep starts); afterwards voff = what the
  // positions move with beyond the final velocity
  L.wacc = scratch + mqe_maxpair(maxc) * RSB_STRIDE;
  if (L.wacc + ((ndof + 3) & ~3) > o) o = L.wacc + ((ndof + 3) & ~3);
  L.total = o;
  return L;
}

// floats the post-physics epilogue of k_substeps stages from L.body on (obs-bag rows, last-action rows, NPC rows, the actions)
__host__ __device__ inline int post_npc_stride(int P) { return (P * 13 + 3) & ~3; }
__host__ __device__ inline int post_staging_floats(int epw, int amp, int nj, int P) { return epw * amp * (MQE_OBS_BAG + 24) + epw * post_npc_stride(P) + epw * nj; }
struct PhysDebug { float* minv; int* nc; float* contacts; int robot; long long* times; int stop_after; };
// phase tap: the 100 MHz wall clock at lane 0 and, for per-phase counter runs (tools/phase_counters.py), an early exit of the whole wavefront
#define TSTAMP(i) do { if (dbg.times != nullptr && lane == 0) dbg.times[i] = (long long)wall_clock64(); if (dbg.stop_after == (i)) return; } while (0)

// flags of one physics substep executed by a wavefront
enum { PS_LOAD_STATE = 1, PS_LOAD_TAU = 2, PS_STORE_STATE = 4, PS_WRITE_CF = 8 };

// Compile-time shape of an env (specialisations of the runtime values below; measured on go1gate: -8 % kernel time --
// loops over the agents unroll, the NPC / seesaw / box / scenery branches disappear, the LDS layout becomes constants):
//   TA > 0: number of agents, TA = 0: m->A;
//   TP >= 0: which kinds of non-robot objects the scene has -- PS_F_LINK (fixed base + 1-dof link: seesaw, door, tug),
//            PS_F_NPC (free bodies: ball, sheep, box), PS_F_BOX (the free body is the oriented box), PS_F_STATIC (scenery
//            boxes); 0 = robots only.  TP = -1: everything read from the model at run time.
enum { PS_F_LINK = 1, PS_F_NPC = 2, PS_F_BOX = 4, PS_F_STATIC = 8,
       PS_F_FEW = 16,     // at most 4 actors and at most 10 KiB of LDS per env: the scene runs the row sweep with all envs resident (SubstepsClass)
       PS_F_ROW = 32 };   // at most 4 actors: row sweep compiled in (without it a TP >= 0 kernel of a larger scene has the lane sweep only)
template <int TP> struct ShapeClass {
  static constexpr bool small = TP == 0 || TP == PS_F_LINK || (TP > 0 && (TP & PS_F_FEW) != 0);
  static constexpr int sweep = TP >= 0 ? 1 : -1;    // 1 row (round 6: also the flocks and 2 vs 2 + ball -- robots in rows, NPCs in lanes), -1 the model says (generic kernels)
  static constexpr int npq = (small || (TP > 0 && (TP & PS_F_ROW) != 0)) ? 2 : 4;      // passes of 16 contacts when the side records are built: <= 32 contacts, else <= 64
};
template <int TA, int TP>
struct PhysShape {
  const int A, P, PD, npcdof, ND, nbody, ndof, maxc, n_static;
  const bool has_seesaw, has_box;
  const bool rowgs;      // contact sweep with one DPP row per actor (scenes of <= 4 actors) instead of one lane per contact
  __device__ __forceinline__ explicit PhysShape(const DevModel* m)
      : A(TA > 0 ? TA : m->A), P(TP == 0 ? 0 : m->P), PD((TP < 0 || (TP & PS_F_NPC)) ? m->n_npc_dyn : 0),
        npcdof((TP < 0 || (TP & PS_F_NPC)) ? m->npc_dofs_each : 0),
        ND((TA > 0 && TP == 0) ? 12 * TA : m->ND), nbody((TA > 0 && TP == 0) ? TA * MQE_NBODY : m->nbody_env),
        ndof((TA > 0 && TP == 0) ? TA * MQE_RD : m->ndof_env), maxc((TA > 0 && TP == 0) ? mqe_maxc(TA, 0, 2) : m->maxc),
        n_static((TP < 0 || (TP & PS_F_STATIC)) ? m->n_static : 0),
        has_seesaw(TP < 0 ? m->has_seesaw != 0 : (TP & PS_F_LINK) != 0), has_box(TP < 0 ? m->has_box != 0 : (TP & PS_F_BOX) != 0),
        rowgs(ShapeClass<TP>::sweep >= 0 ? ShapeClass<TP>::sweep == 1 : m->rowgs != 0) {}
};

// EPW = environments per wavefront.  1: the 64 lanes work for one env.  2 (two-robot scenes without objects): each half-wave of 32
// lanes runs its own env -- body lanes 0-25, one 16-lane sweep row per robot, 32 feature points per pass -- so the dynamics and
// contact phases, which keep a third of 64 lanes busy for one env, do the work of two envs with the same instruction stream.
// Everything below is written for "the env's lanes": `lane` is the lane WITHIN the group, `lds` the group's own state, ballots are
// the group's bits; the few places where a wave-wide decision is needed (barriers inside loops whose trip count depends on the
// env) say so.  The arithmetic per env is the same either way: both forms produce bit-identical states (tests/test_gpu_parity.py).
template <int TA, int TP, int EPW = 1>
__device__ __forceinline__ void phys_substep(const DevModel* __restrict__ m, const DevState& st, float* lds_wave, const int e_first, const int lane_wave,
                                             const int flags, const int no_write, const PhysDebug& dbg, const int hotv) {
  static_assert(EPW == 1 || (EPW == 2 && TP == 0 && (TA == 1 || TA == 2)), "two envs per wavefront: robot-only scenes of at most two robots");
  // the model's wave-uniform constants from the table k_substeps loaded once per launch (DevModel::hot: lane i of `hotv` = entry i)
  auto HI = [&](int i) -> int { return __builtin_amdgcn_readlane(hotv, i); };
  auto HF = [&](int i) -> float { return __int_as_float(__builtin_amdgcn_readlane(hotv, i)); };
  auto HP = [&](int i) -> const float* {
    return as_global(reinterpret_cast<const float*>(((unsigned long long)(unsigned)__builtin_amdgcn_readlane(hotv, i + 1) << 32) | (unsigned long long)(unsigned)__builtin_amdgcn_readlane(hotv, i)));
  };
  auto HMASK = [&]() -> unsigned long long {
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(hotv, HOT_FEAT_MASK_HI) << 32) | (unsigned long long)(unsigned)__builtin_amdgcn_readlane(hotv, HOT_FEAT_MASK_LO);
  };
  constexpr int LW = 64 / EPW;                                   // lanes of one env
  const int grp = EPW == 1 ? 0 : lane_wave / LW;                 // which env of the wavefront this lane works for
  const int lane = EPW == 1 ? lane_wave : lane_wave - grp * LW;  // lane within the env's group
  const bool evalid = EPW == 1 || e_first + grp < HI(HOT_N);          // an odd batch leaves the last wavefront's second half without an env:
  const int e = evalid ? e_first + grp : HI(HOT_N) - 1;               // it recomputes the last env and stores nothing
  auto gballot = [&](bool p) -> unsigned long long {             // the group's bits of a ballot, in the low LW bits
    const unsigned long long b = __ballot(p);
    return EPW == 1 ? b : ((b >> (grp * LW)) & ((1ull << (LW & 63)) - 1ull));
  };
  auto wave_max_of_groups = [&](int x) -> int {                  // a value that is uniform within each group -> its maximum over the wave
    return EPW == 1 ? x : max(__builtin_amdgcn_readlane(x, 0), __builtin_amdgcn_readlane(x, 32));
  };
  const PhysShape<TA, TP> shp(m);
  const int A = shp.A, P = shp.P, PD = shp.PD, npcdof = shp.npcdof;
  const int nbody = shp.nbody, ndof = shp.ndof, nsph = HI(HOT_NSPH_ENV), maxc = shp.maxc;
  constexpr int BODY_STRIDE = BODY_STRIDE_OF(PhysPad<TP>::on), CON_STRIDE = CON_STRIDE_OF(PhysPad<TP>::on);
  const PhysLds L = phys_lds_layout(A, P, shp.ND, nbody, ndof, nsph, HI(HOT_NPRIM_ENV), maxc, shp.rowgs, PhysPad<TP>::on);
  float* lds = lds_wave + grp * L.total;
  const float dt = HF(HOT_DT);
  const mqe_robot_model& rm = m->robot;
  float* g_root = st.root + (size_t)e * (A + P) * 13;
  float* g_dof = st.dof + (size_t)e * shp.ND * 2;

  // this lane's self-collision candidates, one per pass of 64 (requested here, consumed after the terrain contacts: the
  // table sits in global memory and a load inside the pass loop put its full latency on every pass)
  // (192 candidates at a time: the capsule model has 144; the exact one, 350, takes a second chunk whose table entries are loaded when it runs)
  constexpr int SELF_CHUNK = 192;
  constexpr int NSP = (SELF_CHUNK + LW - 1) / LW;
  int selfp[NSP];
#pragma unroll
  for (int k = 0; k < NSP; k++) selfp[k] = (HI(HOT_SELF_COLLISION) && k * LW + lane < HI(HOT_N_SELF_PAIRS)) ? (int)rm.self_pair[k * LW + lane] : -1;
  // ... and this lane's joint of the self-collision "safe box" (lanes 0-11): a robot whose joint angles are all inside it cannot touch itself
  const float ss_lo = (HI(HOT_SELF_COLLISION) && lane < MQE_NDOF) ? rm.self_safe_lo[lane] : -1e30f;
  const float ss_hi = (HI(HOT_SELF_COLLISION) && lane < MQE_NDOF) ? rm.self_safe_hi[lane] : 1e30f;
  TSTAMP(0);
  // ---- coalesced state load (first substep of a launch only; afterwards the state stays in LDS) -----------------
  if (flags & PS_LOAD_STATE) {
    for (int i = lane; i < (A + P) * 13; i += LW) lds[L.root + i] = g_root[i];
    for (int i = lane; i < shp.ND * 2; i += LW) lds[L.dof + i] = g_dof[i];
  }
  if (flags & PS_LOAD_TAU)
    for (int i = lane; i < 12 * A; i += LW) lds[L.tau + i] = st.torques[(size_t)e * 12 * A + i];
  __syncthreads();

  TSTAMP(1);
  // ---- forward kinematics by tree level (body lanes) ---------------------------------------------------------
  const bool is_body = lane < nbody;
  const bool is_rbody = lane < A * MQE_NBODY;
  const int br = is_rbody ? lane / MQE_NBODY : 0;              // robot of this body lane
  const int bb = is_rbody ? lane - br * MQE_NBODY : 0;         // body index in the robot
  const int depth = (is_rbody && bb > 0) ? ((bb - 1) % 3 + 1) : 0;
  // domain parameters (MQE_T_DOMAIN_PARAMS, global memory) are requested here and consumed after the kinematics / in the
  // contact solver, so that their latency is off the critical path: added base mass and base CoM shift of this lane's robot --
  // body 0 only, inertia about the CoM unchanged (legged_robot.py:332-334, legged_robot_field.py:324-334 edit mass and com of
  // props[0] only) -- and the env's shape friction
  const float* dp = st.dparams + ((size_t)e * A + br) * 8;
  const bool dbase = is_rbody && bb == 0;
  const float dp_mass = dbase ? dp[1] : 0.0f, dp_cx = dbase ? dp[2] : 0.0f, dp_cy = dbase ? dp[3] : 0.0f, dp_cz = dbase ? dp[4] : 0.0f;
  const float mu_env = st.dparams[(size_t)e * A * 8];
  // (no initial values for what is only read where it has been written -- Rm, bp, bw, bvp of a body lane; the joint frame of a joint lane: a
  // constant costs one v_mov per register and substep for the whole wavefront.  The base lanes do read bax / bal / bap = 0.)
  // (ADVICE r5 asked for defined values here.  Both ways of giving them one were built in round 6 and cost what the zeros cost: a frozen value
  // (__builtin_nondeterministic_value) +66, the output of an empty asm statement +63 static vector instructions in the substep -- the register
  // allocator materialises either.  Left as they are: every consumer of these registers is guarded by the lane's role, the wave shifts and the
  // unconditional cross product at "Sv =" only move / combine bits that nobody reads.)
  float Rm[9];
  V3 bp, bw, bvp, bax = v3(0, 0, 0), bal = bax, bap = bax;
  float* myrec = lds + L.body + lane * BODY_STRIDE;
  if (is_body && depth == 0) {
    const float* rs = lds + L.root + (is_rbody ? br : (A + (lane - A * MQE_NBODY))) * 13;
    float x = rs[3], y = rs[4], z = rs[5], w = rs[6];
    {   // a reset copies the configured quaternion verbatim and go1_wrestling_config.py:68,74 gives (0,0,-+1,1): read it normalised
      const float inq = 1.0f / sqrtf(x * x + y * y + z * z + w * w);
      x *= inq; y *= inq; z *= inq; w *= inq;
    }
    Rm[0] = 1 - 2 * (y * y + z * z); Rm[1] = 2 * (x * y - z * w); Rm[2] = 2 * (x * z + y * w);
    Rm[3] = 2 * (x * y + z * w); Rm[4] = 1 - 2 * (x * x + z * z); Rm[5] = 2 * (y * z - x * w);
    Rm[6] = 2 * (x * z - y * w); Rm[7] = 2 * (y * z + x * w); Rm[8] = 1 - 2 * (x * x + y * y);
    bp = ld3(rs); bvp = ld3(rs + 7); bw = ld3(rs + 10);
    body_store(myrec, Rm, bp, bax);
    reinterpret_cast<float4*>(myrec)[4] = make_float4(bw.x, bw.y, bw.z, bvp.x);
    reinterpret_cast<float4*>(myrec)[5] = make_float4(bvp.y, bvp.z, bal.x, bal.y);
    reinterpret_cast<float4*>(myrec)[6] = make_float4(bal.z, bap.x, bap.y, bap.z);
  }
  // joint-local quantities do not depend on the parent: computed once for all joint lanes, outside the level loop
  // (whose divergent body would otherwise run the sin/cos code three times)
  float Rj[9], qdj;
  V3 joff, jax;
  if (depth > 0) {
    const int j = bb - 1;
    const float qj = lds[L.dof + (br * 12 + j) * 2];
    qdj = lds[L.dof + (br * 12 + j) * 2 + 1];
    joff = v3(rm.joint_offset[bb][0], rm.joint_offset[bb][1], rm.joint_offset[bb][2]);
    jax = v3(rm.joint_axis[bb][0], rm.joint_axis[bb][1], rm.joint_axis[bb][2]);
    const V3 ax = jax;
    float c, s;
    joint_sincos(qj, s, c);
    const float t = 1 - c;
    Rj[0] = t * ax.x * ax.x + c; Rj[1] = t * ax.x * ax.y - s * ax.z; Rj[2] = t * ax.x * ax.z + s * ax.y;
    Rj[3] = t * ax.x * ax.y + s * ax.z; Rj[4] = t * ax.y * ax.y + c; Rj[5] = t * ax.y * ax.z - s * ax.x;
    Rj[6] = t * ax.x * ax.z - s * ax.y; Rj[7] = t * ax.y * ax.z + s * ax.x; Rj[8] = t * ax.z * ax.z + c;
  }
  __syncthreads();
#pragma unroll
  for (int lev = 1; lev <= 3; lev++) {
    // the parent of a thigh / calf lane is lane - 1, which computed its frame in the previous level: its registers arrive through
    // DPP wave shifts (executed by every lane: a DPP read needs an active source lane); hips read the base record from LDS
    float QR[9];
    V3 qp, qw, qvp, qal, qap;
    if (lev >= 2) {
#pragma unroll
      for (int k = 0; k < 9; k++) QR[k] = lane_prev(Rm[k]);
      qp = v3(lane_prev(bp.x), lane_prev(bp.y), lane_prev(bp.z)); qw = v3(lane_prev(bw.x), lane_prev(bw.y), lane_prev(bw.z));
      qvp = v3(lane_prev(bvp.x), lane_prev(bvp.y), lane_prev(bvp.z)); qal = v3(lane_prev(bal.x), lane_prev(bal.y), lane_prev(bal.z));
      qap = v3(lane_prev(bap.x), lane_prev(bap.y), lane_prev(bap.z));
    }
    if (depth == lev) {
      float PR[9];
      V3 pp, pw, pvp, pal, pap;
      if (lev == 1) {
        const float* pr = lds + L.body + br * MQE_NBODY * BODY_STRIDE;
        const float4* p4 = reinterpret_cast<const float4*>(pr);
        const float4 b0 = p4[0], b1 = p4[1], b2 = p4[2], b4 = p4[4], b5 = p4[5], b6 = p4[6];     // words 0-2: R, origin; 4-6: w, vp, al, ap
        PR[0] = b0.x; PR[1] = b0.y; PR[2] = b0.z; PR[3] = b0.w; PR[4] = b1.x; PR[5] = b1.y; PR[6] = b1.z; PR[7] = b1.w; PR[8] = b2.x;
        pp = v3(b2.y, b2.z, b2.w); pw = v3(b4.x, b4.y, b4.z); pvp = v3(b4.w, b5.x, b5.y); pal = v3(b5.z, b5.w, b6.x); pap = v3(b6.y, b6.z, b6.w);
      } else {
#pragma unroll
        for (int k = 0; k < 9; k++) PR[k] = QR[k];
        pp = qp; pw = qw; pvp = qvp; pal = qal; pap = qap;
      }
      V3 dd = mat_vec(PR, joff);
      bp = pp + dd;
      for (int r = 0; r < 3; r++)
        for (int cc = 0; cc < 3; cc++) Rm[r * 3 + cc] = PR[r * 3] * Rj[cc] + PR[r * 3 + 1] * Rj[3 + cc] + PR[r * 3 + 2] * Rj[6 + cc];
      bax = mat_vec(PR, jax);
      bw = pw + qdj * bax;
      bvp = pvp + cross(pw, dd);
      bal = pal + cross(pw, qdj * bax);
      bap = pap + cross(pal, dd) + cross(pw, cross(pw, dd));
      body_store(myrec, Rm, bp, bax);
    }
  }
  __syncthreads();
  TSTAMP(2);
  // ---- spatial inertia + bias wrench about o = base origin; composite sums up each leg ------------------------
  // X[0]=m, X[1:4]=h=m*(c-o), X[4:10]=Ibar (sym6), X[10:13]=moment about o, X[13:16]=force
  float X[16];
#pragma unroll
  for (int k = 0; k < 16; k++) X[k] = 0.0f;
  V3 o = v3(0, 0, 0);
  // world COM / inertia of robot bodies, then their share of the composite -- ONE block (round 6: two blocks under the same condition carried
  // Iw, the mass and the COM across their join as zero-initialised values: 18 moves per substep)
  if (is_rbody) {
    float Iw[6];
    const float bmass = rm.mass[bb] + dp_mass;
    const V3 bc = bp + mat_vec(Rm, v3(rm.com[bb][0] + dp_cx, rm.com[bb][1] + dp_cy, rm.com[bb][2] + dp_cz));
    const float* S = rm.inertia[bb];
    float Il[9] = {S[0], S[3], S[4], S[3], S[1], S[5], S[4], S[5], S[2]}, T[9];
    for (int r = 0; r < 3; r++)
      for (int cc = 0; cc < 3; cc++) T[r * 3 + cc] = Rm[r * 3] * Il[cc] + Rm[r * 3 + 1] * Il[3 + cc] + Rm[r * 3 + 2] * Il[6 + cc];
    // Iw = T * R^T (symmetric)
    Iw[0] = T[0] * Rm[0] + T[1] * Rm[1] + T[2] * Rm[2];
    Iw[1] = T[3] * Rm[3] + T[4] * Rm[4] + T[5] * Rm[5];
    Iw[2] = T[6] * Rm[6] + T[7] * Rm[7] + T[8] * Rm[8];
    Iw[3] = T[0] * Rm[3] + T[1] * Rm[4] + T[2] * Rm[5];
    Iw[4] = T[0] * Rm[6] + T[1] * Rm[7] + T[2] * Rm[8];
    Iw[5] = T[3] * Rm[6] + T[4] * Rm[7] + T[5] * Rm[8];
    o = ld3(lds + L.body + br * MQE_NBODY * BODY_STRIDE + B_P);
    V3 rc = bc - o;
    X[0] = bmass; X[1] = bmass * rc.x; X[2] = bmass * rc.y; X[3] = bmass * rc.z;
    float r2 = dot(rc, rc);
    X[4] = Iw[0] + bmass * (r2 - rc.x * rc.x); X[5] = Iw[1] + bmass * (r2 - rc.y * rc.y); X[6] = Iw[2] + bmass * (r2 - rc.z * rc.z);
    X[7] = Iw[3] - bmass * rc.x * rc.y; X[8] = Iw[4] - bmass * rc.x * rc.z; X[9] = Iw[5] - bmass * rc.y * rc.z;
    V3 rcb = bc - bp;
    V3 ac = bap + cross(bal, rcb) + cross(bw, cross(bw, rcb));
    V3 f = bmass * (ac - v3(0, 0, HF(HOT_GRAVITY_Z)));
    V3 nc = sym_vec(Iw, bal) + cross(bw, sym_vec(Iw, bw));
    V3 no = nc + cross(rc, f);
    X[10] = no.x; X[11] = no.y; X[12] = no.z; X[13] = f.x; X[14] = f.y; X[15] = f.z;
  }
  {
    // thigh += calf, then hip += thigh: x + 1.0 * t rounds like x + t and x + 0.0 * t is x, so the lane's depth enters as a factor and each
    // term is ONE instruction -- v_fmac_f32 with the wave shift (lane + 1; zero past the end) as the operand's DPP modifier -- instead of
    // shift + add + select.  Written out: the compiler keeps the shift as a v_mov_dpp of its own.  (s_nop 1: a DPP operand must not have
    // been written by the two instructions before; inside the block the second round reads what the first wrote 16 instructions earlier.)
    const float on2 = depth == 2 ? 1.0f : 0.0f, on1 = depth == 1 ? 1.0f : 0.0f;
#define MQE_FD(i, o) "v_fmac_f32_dpp %" #i ", %" #i ", %" #o " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
#define MQE_FD16(o) MQE_FD(0, o) MQE_FD(1, o) MQE_FD(2, o) MQE_FD(3, o) MQE_FD(4, o) MQE_FD(5, o) MQE_FD(6, o) MQE_FD(7, o) \
                    MQE_FD(8, o) MQE_FD(9, o) MQE_FD(10, o) MQE_FD(11, o) MQE_FD(12, o) MQE_FD(13, o) MQE_FD(14, o) MQE_FD(15, o)
    asm volatile("s_nop 1\n\t" MQE_FD16(16) MQE_FD16(17)
                 : "+v"(X[0]), "+v"(X[1]), "+v"(X[2]), "+v"(X[3]), "+v"(X[4]), "+v"(X[5]), "+v"(X[6]), "+v"(X[7]),
                   "+v"(X[8]), "+v"(X[9]), "+v"(X[10]), "+v"(X[11]), "+v"(X[12]), "+v"(X[13]), "+v"(X[14]), "+v"(X[15])
                 : "v"(on2), "v"(on1));
#undef MQE_FD16
#undef MQE_FD
  }
  {
    // The base block wants the robot's TOTAL composite = the base body's own + the four hip composites.  It is not summed on the base lane
    // (16 x 16 B loads and 64 additions with two lanes of the wavefront active, round 5): the five records go to LDS here -- 4 x 16 B
    // stores per lane -- and the lanes that assemble the Schur complement further down gather their entry's five terms themselves.
    float* hx = lds + L.fcol;
    if (is_rbody && depth <= 1) {
      float4* w = reinterpret_cast<float4*>(hx + (br * 5 + (bb == 0 ? 4 : (bb - 1) / 3)) * FCOL_STRIDE);
#pragma unroll
      for (int k = 0; k < 4; k++) w[k] = make_float4(X[4 * k], X[4 * k + 1], X[4 * k + 2], X[4 * k + 3]);
    }
  }
  // ---- mass-matrix columns (CRBA in the common frame) and generalized bias ------------------------------------
  // joint lane: S = (w: a, v_o: (p - o) x a);  F = Ic S = (f, n)
  V3 Sa = bax, Sv = cross(bp - o, bax);
  V3 Ff = X[0] * Sv + cross(Sa, v3(X[1], X[2], X[3]));
  V3 Fn = sym_vec(X + 4, Sa) + cross(v3(X[1], X[2], X[3]), Sv);
  {
    // ancestors' S through shuffles (lane-1: parent joint if depth>=2, lane-2: grandparent if depth==3)
    V3 Sa1 = v3(lane_prev(Sa.x), lane_prev(Sa.y), lane_prev(Sa.z));
    V3 Sv1 = v3(lane_prev(Sv.x), lane_prev(Sv.y), lane_prev(Sv.z));
    V3 Sa2 = v3(lane_prev(Sa1.x), lane_prev(Sa1.y), lane_prev(Sa1.z));
    V3 Sv2 = v3(lane_prev(Sv1.x), lane_prev(Sv1.y), lane_prev(Sv1.z));
    // joint lane: own diagonal entry, couplings with the parent (thigh, calf) and the grandparent (calf), bias force
    float mjj = 0.0f, cpl1 = 0.0f, cpl2 = 0.0f;
    if (depth >= 1) {
      const int j = bb - 1;
      mjj = dot(Sa, Fn) + dot(Sv, Ff);
      const float hj = dot(Sa, v3(X[10], X[11], X[12])) + dot(Sv, v3(X[13], X[14], X[15]));
      if (depth >= 2) cpl1 = dot(Sa1, Fn) + dot(Sv1, Ff);
      if (depth == 3) cpl2 = dot(Sa2, Fn) + dot(Sv2, Ff);
      lds[L.rhs + br * MQE_RD + 6 + j] = lds[L.tau + br * 12 + j] - hj;
    }
    // ---- leg blocks on the hip lanes: Mi = Mll^-1, G = Mbl Mi (6x3), C = G Mbl^T (6x6 sym).  The thigh's and the calf's
    // force columns and matrix entries come from lanes + 1 and + 2 through DPP wave shifts (every lane executes the shifts: a DPP
    // read needs an active source lane) -- no LDS round trip between the CRBA columns and the block inverse.
    float fc[18], t1v[9], t2v[9];
    {
      const float mine[9] = {Ff.x, Ff.y, Ff.z, Fn.x, Fn.y, Fn.z, mjj, cpl1, cpl2};
#pragma unroll
      for (int k = 0; k < 9; k++) { t1v[k] = lane_next(mine[k]); t2v[k] = lane_next(t1v[k]); }
#pragma unroll
      for (int k = 0; k < 6; k++) { fc[k] = mine[k]; fc[6 + k] = t1v[k]; fc[12 + k] = t2v[k]; }
    }
    if (depth == 1) {
      const int leg = (bb - 1) / 3;
      float* Ml = lds + L.leg + (br * 4 + leg) * LEG_STRIDE;
      const float a = mjj, b = t1v[6], c = t2v[6], d = t1v[7], ee = t2v[8], f = t2v[7];   // Mll = [[a,d,ee],[d,b,f],[ee,f,c]]
      float c00 = b * c - f * f, c01 = ee * f - d * c, c02 = d * f - ee * b;
      float c11 = a * c - ee * ee, c12 = d * ee - a * f, c22 = a * b - d * d;
      float idet = 1.0f / (a * c00 + d * c01 + ee * c02);
      float Mi[9] = {c00 * idet, c01 * idet, c02 * idet, c01 * idet, c11 * idet, c12 * idet, c02 * idet, c12 * idet, c22 * idet};
      float G[18];
      for (int mm = 0; mm < 6; mm++)
        for (int i = 0; i < 3; i++) G[mm * 3 + i] = fc[mm] * Mi[i] + fc[6 + mm] * Mi[3 + i] + fc[12 + mm] * Mi[6 + i];
      float Cv[24];
      int q = 0;
      for (int mm = 0; mm < 6; mm++)
        for (int n = mm; n < 6; n++) Cv[q++] = G[mm * 3] * fc[n] + G[mm * 3 + 1] * fc[6 + n] + G[mm * 3 + 2] * fc[12 + n];
      Cv[21] = 0.0f; Cv[22] = 0.0f; Cv[23] = 0.0f;
      float4* M4 = reinterpret_cast<float4*>(Ml);
      // Cholesky factor of Mll^-1 (3 x 3, SPD): Mll^-1 = Lm Lm^T
      const float l00 = sqrtf(Mi[0]), il00 = 1.0f / l00, l10 = Mi[1] * il00, l20 = Mi[2] * il00;
      const float l11 = sqrtf(Mi[4] - l10 * l10), l21 = (Mi[5] - l20 * l10) / l11, l22 = sqrtf(Mi[8] - l20 * l20 - l21 * l21);
      M4[0] = make_float4(Mi[0], Mi[4], Mi[8], Mi[1]); M4[1] = make_float4(Mi[2], Mi[5], l00, l10);   // sym6: 00,11,22,01,02,12; then Lm
      M4[2] = make_float4(l11, l20, l21, l22);
#pragma unroll
      for (int k = 0; k < 4; k++) M4[3 + k] = make_float4(G[4 * k], G[4 * k + 1], G[4 * k + 2], G[4 * k + 3]);
      M4[7] = make_float4(G[16], G[17], 0.0f, 0.0f);
      float4* C4w = reinterpret_cast<float4*>(lds + L.legc + (br * 4 + leg) * LEGC_STRIDE);
#pragma unroll
      for (int k = 0; k < 6; k++) C4w[k] = make_float4(Cv[4 * k], Cv[4 * k + 1], Cv[4 * k + 2], Cv[4 * k + 3]);
    }
  }
  __syncthreads();
  TSTAMP(3);
  TSTAMP(4);
  // ---- Schur complement S = M_bb - sum_legs C, one lane per entry of the upper triangle (the six lanes that factor it below would
  // otherwise each subtract all four C matrices), and the base part of the generalized bias.  M_bb = [m 1, -[h]x; [h]x, Ibar] and
  // -(force, moment) are entries of the robot's total composite X = base body + four hips (records of L.fcol; X[0] = m, X[1:4] = h,
  // X[4:10] = Ibar, X[10:13] = moment, X[13:16] = force), summed in the order the base lane used to: own + (((h0 + h1) + h2) + h3).
  // Entry t of a robot: 0-20 the upper triangle row by row, 21-26 the bias; which X it is, its sign, and whether it is a structural zero: 4 + 1 + 1 bits each.
  for (int t = lane; t < A * 27; t += LW) {
    const int r = (int)(t >= 27) + (int)(t >= 54) + (int)(t >= 81), q = t - r * 27;
    //                     q:  0  1  2  3  4  5   6  7  8  9 10  11 12 13 14  15 16 17  18 19  20 | 21 22 23 24 25 26
    //                     X:  0  -  -  -  3 -2   0  - -3  -  1   0  2 -1  -   4  7  8   5  9   6 |-13-14-15-10-11-12
    const int xi = (int)((q < 16 ? (0x4012010300230000ull >> (4 * q)) : (0xCBAFED69587ull >> (4 * (q - 16)))) & 15ull);
    const bool neg = ((0x7E02120u >> q) & 1u) != 0u, none = ((0x428Eu >> q) & 1u) != 0u;
    float v = 0.0f;
    if (!none) {
      const float* hb = lds + L.fcol + r * 5 * FCOL_STRIDE + xi;
      v = hb[4 * FCOL_STRIDE] + (((hb[0] + hb[FCOL_STRIDE]) + hb[2 * FCOL_STRIDE]) + hb[3 * FCOL_STRIDE]);
      v = neg ? -v : v;
    }
    if (q < 21) {
      const float* cq = lds + L.legc + r * 4 * LEGC_STRIDE + q;
      lds[L.basei + r * 24 + q] = (((v - cq[0]) - cq[LEGC_STRIDE]) - cq[2 * LEGC_STRIDE]) - cq[3 * LEGC_STRIDE];
    } else {
      lds[L.rhs + r * MQE_RD + (q - 21)] = v;
    }
  }
  __syncthreads();
  // ---- 6x6 Schur complement inverse: lane (robot, column) --------------------------------------------------------
  if (lane < A * 6) {
    const int r = lane / 6, col = lane - r * 6;
    float S[6][6];
    {
      const float4* s4 = reinterpret_cast<const float4*>(lds + L.basei + r * 24);
      float Su[24];
#pragma unroll
      for (int w = 0; w < 6; w++) { const float4 t = s4[w]; Su[4 * w] = t.x; Su[4 * w + 1] = t.y; Su[4 * w + 2] = t.z; Su[4 * w + 3] = t.w; }
      int q = 0;
      for (int i = 0; i < 6; i++) for (int j = i; j < 6; j++) S[i][j] = Su[q++];
    }
    for (int i = 0; i < 6; i++) for (int j = 0; j < i; j++) S[i][j] = S[j][i];
    // Cholesky (lower), then solve S x = e_col
    float Lc[6][6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
      float dgn = S[j][j];
#pragma unroll
      for (int k = 0; k < j; k++) dgn -= Lc[j][k] * Lc[j][k];
      dgn = sqrtf(dgn);
      Lc[j][j] = dgn;
      float inv = 1.0f / dgn;
#pragma unroll
      for (int i = j + 1; i < 6; i++) {
        float v = S[i][j];
#pragma unroll
        for (int k = 0; k < j; k++) v -= Lc[i][k] * Lc[j][k];
        Lc[i][j] = v * inv;
      }
    }
    float x[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
      float v = (i == col) ? 1.0f : 0.0f;
#pragma unroll
      for (int k = 0; k < i; k++) v -= Lc[i][k] * x[k];
      x[i] = v / Lc[i][i];
    }
    float* Si = lds + L.sinv + r * 72;
    // after the forward substitution x = Lc^-1 e_col, i.e. row `col` of F = Lc^-T (S^-1 = Lc^-T Lc^-1 = F F^T): zeros left of the diagonal
#pragma unroll
    for (int i = 0; i < 6; i++) Si[36 + col * 6 + i] = x[i];
#pragma unroll
    for (int i = 5; i >= 0; i--) {
      float v = x[i];
#pragma unroll
      for (int k = i + 1; k < 6; k++) v -= Lc[k][i] * x[k];
      x[i] = v / Lc[i][i];
    }
    for (int i = 0; i < 6; i++) Si[i * 6 + col] = x[i];
  }
  __syncthreads();
  TSTAMP(5);
  // ---- M^-1 in factored form.  With G_k = M_bl,k M_ll,k^-1 (6 x 3 per leg) and S = M_bb - sum_k G_k M_lb,k:
  //   M^-1 = [ S^-1            -S^-1 G            ]    so   x = M^-1 r  is  x_b = S^-1 (r_b - sum_k G_k r_l,k),
  //          [ -G^T S^-1   M_ll^-1 + G^T S^-1 G   ]                          x_l,k = M_ll,k^-1 r_l,k - G_k^T x_b
  // -- three short lane-parallel stages instead of the 18 x 18 inverse (which took 4.7 k cycles and 1.3 kB per robot to build).
  auto mi_at = [](const float* Mi, int i, int j) -> float {         // sym6 00,11,22,01,02,12: (0,1) -> 3, (0,2) -> 4, (1,2) -> 5
    return Mi[i == j ? i : 2 + i + j];
  };
  // row k of one robot's M^-1, element `e` (both 0..17): only for the joint-limit impulses and the debug tap
  auto minv_elem = [&](int r, int k, int e) -> float {
    const float* Si = lds + L.sinv + r * 72;
    const float* lg = lds + L.leg + r * 4 * LEG_STRIDE;
    float y[6];                                                     // k < 6: row k of S^-1; else S^-1 G_L[:, t]
    if (k < 6) {
#pragma unroll
      for (int mm = 0; mm < 6; mm++) y[mm] = Si[k * 6 + mm];
    } else {
      const float* G = lg + ((k - 6) / 3) * LEG_STRIDE + LEG_G;
      const int t = (k - 6) % 3;
#pragma unroll
      for (int mm = 0; mm < 6; mm++) {
        float acc = 0.0f;
#pragma unroll
        for (int n = 0; n < 6; n++) acc += Si[mm * 6 + n] * G[n * 3 + t];
        y[mm] = acc;
      }
    }
    if (e < 6) return k < 6 ? y[e] : -y[e];
    const int L2 = (e - 6) / 3, i = (e - 6) % 3;
    const float* G2 = lg + L2 * LEG_STRIDE + LEG_G;
    float acc = 0.0f;
#pragma unroll
    for (int mm = 0; mm < 6; mm++) acc += G2[mm * 3 + i] * y[mm];
    if (k < 6) return -acc;
    if (L2 == (k - 6) / 3) acc += mi_at(lg + L2 * LEG_STRIDE + LEG_MI, i, (k - 6) % 3);
    return acc;
  };
  float* accv = lds + L.acc;
  if (lane < A * 6) {                       // stage 1: t = r_b - sum_k G_k r_l,k
    const int r = lane / 6, mm = lane - r * 6;
    const float* rh = lds + L.rhs + r * MQE_RD;
    float t = rh[mm];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float* G = lds + L.leg + (r * 4 + k) * LEG_STRIDE + LEG_G + mm * 3;
      t -= G[0] * rh[6 + k * 3] + G[1] * rh[7 + k * 3] + G[2] * rh[8 + k * 3];
    }
    accv[r * MQE_RD + 6 + mm] = t;          // parked in the joint part of the scratch vector
  }
  __syncthreads();
  if (lane < A * 6) {                       // stage 2: x_b = S^-1 t
    const int r = lane / 6, mm = lane - r * 6;
    const float* Si = lds + L.sinv + r * 72 + mm * 6;
    const float* t = accv + r * MQE_RD + 6;
    accv[r * MQE_RD + mm] = Si[0] * t[0] + Si[1] * t[1] + Si[2] * t[2] + Si[3] * t[3] + Si[4] * t[4] + Si[5] * t[5];
  }
  __syncthreads();
  TSTAMP(6);
  // ---- unconstrained velocity v* = v + dt M^-1 (tau - h) per generalized velocity; a lane owns dofs lane and lane + LW
  // (4 robots + ball = 78).  Kept in registers until the bias vector it overwrites has been consumed by every lane.
  auto vstar = [&](int d) -> float {
    if (d < A * MQE_RD) {
      const int r = d / MQE_RD, k = d - r * MQE_RD;
      const float* xb = accv + r * MQE_RD;
      if (k < 6) return lds[L.root + r * 13 + 7 + k] + dt * xb[k];
      const int lg = (k - 6) / 3, i = (k - 6) - lg * 3;
      const float* rec = lds + L.leg + (r * 4 + lg) * LEG_STRIDE;
      const float* rl = lds + L.rhs + r * MQE_RD + 6 + lg * 3;
      const float* G = rec + LEG_G + i;
      float x = mi_at(rec + LEG_MI, i, 0) * rl[0] + mi_at(rec + LEG_MI, i, 1) * rl[1] + mi_at(rec + LEG_MI, i, 2) * rl[2];
#pragma unroll
      for (int mm = 0; mm < 6; mm++) x -= G[mm * 3] * xb[mm];
      return lds[L.dof + (r * 12 + k - 6) * 2 + 1] + dt * x;
    }
    if (shp.has_seesaw) return lds[L.dof + (12 * A) * 2 + 1];       // the plank's hinge: COM on the axis, no drive
    const int q = d - A * MQE_RD, p = q / npcdof, k = q - p * npcdof;
    float v = lds[L.root + (A + p) * 13 + 7 + k];
    if (k == 2) v += dt * HF(HOT_GRAVITY_Z);
    return v;
  };
  const float vs0 = lane < ndof ? vstar(lane) : 0.0f;
  const float vs1 = lane + LW < ndof ? vstar(lane + LW) : 0.0f;

  TSTAMP(7);
  // ---- collision spheres ----------------------------------------------------------------------------------------------
  const int nsr = HI(HOT_N_SPHERES);
  for (int s = lane; s < nsph; s += LW) {
    V3 c; float rad;
    if (s < A * nsr) {
      const int r = (int)(s >= nsr) + (int)(s >= 2 * nsr) + (int)(s >= 3 * nsr), si = s - r * nsr;       // s / nsr for at most four robots, without the division
      const float* rec = lds + L.body + (r * MQE_NBODY + rm.sphere_body[si]) * BODY_STRIDE;
      const float4 q0 = reinterpret_cast<const float4*>(rec)[0], q1 = reinterpret_cast<const float4*>(rec)[1], q2 = reinterpret_cast<const float4*>(rec)[2];
      const float Rr[9] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x};
      c = v3(q2.y, q2.z, q2.w) + mat_vec(Rr, v3(rm.sphere_center[si][0], rm.sphere_center[si][1], rm.sphere_center[si][2]));
      rad = rm.sphere_radius[si];
    } else {
      const int nsn_ = HI(HOT_NPC_N_SPHERES);
      const int q = s - A * nsr, p = nsn_ == 1 ? q : (nsn_ == 2 ? q >> 1 : q / nsn_), si = q - p * nsn_;      // (one or two spheres per NPC in every shipped scene: no run-time division)
      const float* rec = lds + L.body + (A * MQE_NBODY + p) * BODY_STRIDE;
      c = ld3(rec + B_P) + mat_vec(rec + B_R, v3(m->npc_sphere_center[si][0], m->npc_sphere_center[si][1], m->npc_sphere_center[si][2]));
      rad = m->npc_sphere_radius[si];
    }
    float* sp = lds + L.sph + s * 4;
    *reinterpret_cast<float4*>(sp) = make_float4(c.x, c.y, c.z, rad);
  }
  // Who can touch whom in this substep, decided before any pair geometry is built:
  //  * broad phase of ALL actor pairs at once, lane = pair (a < b, row-major): actors farther apart than 1.2 m (robot vs the box:
  //    1.8 m) cannot touch.  The 55 pairs of the 2 + 9 actors of go1sheep-hard used to cost 55 sequential LDS round trips per substep.
  //  * self-collision: a robot whose joint angles are all inside the model's safe box (a walking robot is) cannot touch itself.
  unsigned long long near0 = 0ull, near1 = 0ull, near2 = 0ull;       // 3 x 64 pairs: up to 19 actors (2 robots + 16 sheep: 153 pairs)
  unsigned int self_todo = 0u;                               // bit a: robot a is outside its safe box
  {
    const int nact = A + PD;
    const int npa = (nact * (nact - 1)) / 2;
    for (int t0 = 0; t0 < npa; t0 += LW) {
      const int t = t0 + lane;
      bool nr = false;
      if (t < npa) {
        int a = 0, rem = t;
        while (rem >= nact - 1 - a) { rem -= nact - 1 - a; a++; }
        const int b = a + 1 + rem;
        const V3 pa = ld3(lds + L.body + (a < A ? a * MQE_NBODY : A * MQE_NBODY + (a - A)) * BODY_STRIDE + B_P);
        const V3 pb = ld3(lds + L.body + (b < A ? b * MQE_NBODY : A * MQE_NBODY + (b - A)) * BODY_STRIDE + B_P);
        const V3 dd = pa - pb;
        if (shp.has_box && b >= A) nr = a < A && !(dot(dd, dd) > 1.8f * 1.8f);
        else if (a >= A) nr = !(dot(dd, dd) > HF(HOT_NPC_PAIR_REACH) * HF(HOT_NPC_PAIR_REACH));      // two free NPCs: their own size, not a robot's (round 6: sheep 1.5 m apart used to be "near")
        else nr = !(dot(dd, dd) > 1.2f * 1.2f);
      }
      const unsigned long long bm = gballot(nr);
      if (t0 == 0) near0 = bm; else if (t0 == LW) near1 = bm; else near2 |= bm;
    }
    if (HI(HOT_SELF_COLLISION))
      for (int a = 0; a < A; a++) {
        const float qj = lds[L.dof + (a * 12 + (lane < MQE_NDOF ? lane : 0)) * 2];
        if (gballot(qj < ss_lo || qj > ss_hi) != 0ull) self_todo |= 1u << a;
      }
  }
  // the robots' primitives in world coordinates (what the OTHER actors' feature points and spheres are tested against): centre and
  // bounding radius, capsule half-segment and radius (a sphere is a capsule with a zero segment; a box keeps its link's rotation
  // and is marked by a negative radius).  Only built when something can touch a primitive at all: two robots walking apart from
  // each other in ordinary poses skip it.
  const int npr = HI(HOT_N_PRIMS);
  // (only a near pair that HAS a robot needs them: the pair bits are a-major, the robots' pairs are the first A (nact - 1) - A (A - 1) / 2)
  const int n_rpairs = A * (A + PD - 1) - (A * (A - 1)) / 2;
  const unsigned long long rmask0 = n_rpairs >= 64 ? ~0ull : ((1ull << n_rpairs) - 1ull);
  const unsigned long long rmask1 = n_rpairs <= 64 ? 0ull : (n_rpairs >= 128 ? ~0ull : ((1ull << (n_rpairs - 64)) - 1ull));
  const bool near_robot = (near0 & rmask0) != 0ull || (near1 & rmask1) != 0ull || (n_rpairs > 128 && near2 != 0ull);
  const bool near_npcs = (near0 & ~rmask0) != 0ull || (near1 & ~rmask1) != 0ull || near2 != 0ull;
  const bool need_prims = near_robot || self_todo != 0u;
  for (int t = lane; need_prims && t < A * npr; t += LW) {
    const int r = (int)(t >= npr) + (int)(t >= 2 * npr) + (int)(t >= 3 * npr), q = t - r * npr;
    const float* rec = lds + L.body + (r * MQE_NBODY + rm.prim_body[q]) * BODY_STRIDE;
    const float4 q0 = reinterpret_cast<const float4*>(rec)[0], q1 = reinterpret_cast<const float4*>(rec)[1], q2 = reinterpret_cast<const float4*>(rec)[2];
    const float Rr[9] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x};
    const V3 c = v3(q2.y, q2.z, q2.w) + mat_vec(Rr, v3(rm.prim_center[q][0], rm.prim_center[q][1], rm.prim_center[q][2]));
    const V3 u = mat_vec(Rr, v3(rm.prim_axis[q][0], rm.prim_axis[q][1], rm.prim_axis[q][2]));
    float4* pw0 = reinterpret_cast<float4*>(lds + L.prim) + t;
    float4* pw1 = pw0 + HI(HOT_NPRIM_ENV);
    pw0[0] = make_float4(c.x, c.y, c.z, rm.prim_bound[q]);
    // radius; a box carries MINUS the radius of its bounding capsule about its longest edge (the sign marks it; screens use |.|)
    float rad = rm.prim_half[q][0];
    if (rm.prim_type[q] == MQE_PRIM_BOX) {
      const V3 hb = v3(rm.prim_half[q][0], rm.prim_half[q][1], rm.prim_half[q][2]), al = v3(rm.prim_axis[q][0], rm.prim_axis[q][1], rm.prim_axis[q][2]);
      rad = -sqrtf(fmaxf(dot(hb, hb) - dot(al, al), 1e-12f));
    }
    pw1[0] = make_float4(u.x, u.y, u.z, rad);
  }
  __syncthreads();

  TSTAMP(8);
  // seesaw geometry (uniform): platform centre, hinge, plank rotation about +y and centre
  const bool SS = shp.has_seesaw;
  V3 ssB = v3(0, 0, 0), ssPiv = v3(0, 0, 0), ssC = v3(0, 0, 0);
  float ssR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, ssTheta = 0.0f;
  if (SS) {
    ssB = ld3(lds + L.root + A * 13);
    ssPiv = ssB + v3(m->ss_joint_offset[0], m->ss_joint_offset[1], m->ss_joint_offset[2]);
    ssTheta = lds[L.dof + (12 * A) * 2];
    float cth, sth;
    joint_sincos(ssTheta, sth, cth);                    // (libm beyond |theta| = 8: a revolving door goes round)
    if (m->ss_axis == 3) ssPiv.y += ssTheta;                                                // slider: translation along +y, no rotation
    else if (m->ss_axis == 2) { ssR[0] = cth; ssR[1] = -sth; ssR[3] = sth; ssR[4] = cth; }  // door: rotation about +z
    else { ssR[0] = cth; ssR[2] = sth; ssR[6] = -sth; ssR[8] = cth; }                       // plank: rotation about +y
    ssC = ssPiv + mat_vec(ssR, v3(m->ss_plank_center[0], m->ss_plank_center[1], m->ss_plank_center[2]));
  }
  // ---- contact generation: terrain (ground plane, wall SDF), canonical order --------------------------------------------
  int nc = 0;
  int ovf = 0;                      // wave-uniform: a touching pair did not fit the bounded list (per-actor cap or list end)
  int red = 0;                      // per env: a robot's one-sided contacts were reduced to the deepest CAP_ROBOT (MQE_T_CONTACT_REDUCED)
  // Passes over whole actors, lane = sphere: two robots per pass (2 x 27 spheres), all single-sphere NPCs in one pass
  // (multi-sphere NPCs: one per pass).  The list order stays canonical -- actor by actor, sphere by sphere, ground /
  // wall / platform / column -- because a lane's slot = contacts of earlier groups (capped) + its rank in its group.
  const int rpp = (2 * nsr <= LW) ? 2 : 1;
  const int n_rpass = rpp == 2 ? (A + 1) >> 1 : A;          // (a division by a run-time value is ~25 vector instructions, wave-uniform or not)
  const int nsn = HI(HOT_NPC_N_SPHERES);
  const bool npc_one = PD * nsn <= LW;                     // every sphere of every free NPC in ONE pass, lane = (npc, sphere): 9 sheep x 2
  const int n_pass = n_rpass + (PD > 0 ? (npc_one ? 1 : PD) : 0);
  const unsigned long long mns = (nsn < 64) ? ((1ull << nsn) - 1ull) : ~0ull;
  for (int pass = 0; pass < n_pass; pass++) {
    int act = -1, sidx = 0, s = nsph, sub = 0;
    unsigned long long gm = ~0ull;                         // lanes of my group (= my actor)
    const bool rob = pass < n_rpass;
    const int cap = rob ? CAP_ROBOT : HI(HOT_CAP_NPC);
    const unsigned long long m0 = (nsr < 64) ? ((1ull << nsr) - 1ull) : ~0ull;
    if (rob) {
      sub = (rpp == 2 && lane >= nsr) ? 1 : 0;
      const int l = lane - sub * nsr, r = pass * rpp + sub;
      if (l < nsr && r < A) { act = r; sidx = l; s = r * nsr + l; }
      gm = sub ? (m0 << nsr) : m0;
    } else if (npc_one) {
      const int pl = lane < PD * nsn ? lane / nsn : 0;
      if (lane < PD * nsn) { act = A + pl; sidx = lane - pl * nsn; s = A * nsr + lane; }
      gm = mns << (pl * nsn);
    } else {
      const int p = pass - n_rpass;
      if (lane < HI(HOT_NPC_N_SPHERES)) { act = A + p; sidx = lane; s = A * nsr + p * HI(HOT_NPC_N_SPHERES) + lane; }
    }
    // ---- edge contacts of the pass's robots with the static world (desc.edge_contacts; oracle: "edge contacts of a robot with the static
    // world"): lane = primitive of the lane's robot; per primitive the deepest of the nearest vertical wall edge and the scenery boxes.
    // Computed first, ranked behind the robot's feature contacts below.
    bool eflag = false; float esd = 1e3f; V3 en = v3(0, 0, 1), epa = v3(0, 0, 0); int ebody = 0, erep = 0;
    bool edge_pass = rob && (HI(HOT_EDGE_MASK) & 6) != 0 && shp.n_static > 0;
    if (rob && !edge_pass && (HI(HOT_EDGE_MASK) & 1) != 0 && HP(HOT_WALL_CORNER_LO) != nullptr) {
      // is any robot of the pass within reach of a wall edge at all?  One look-up per robot (its base's raster point): no primitive
      // reaches farther from the base than feature_reach, so a corner beyond that + the margin of the map's nearest-of-four choice is out
      const int l = lane - sub * nsr, r = pass * rpp + sub;
      bool nearw = false;
      if (l == 0 && r < A) {
        const V3 pbase = ld3(lds + L.body + r * MQE_NBODY * BODY_STRIDE + B_P);
        int ix = (int)floorf(pbase.x / HF(HOT_HS) + 0.5f), iy = (int)floorf(pbase.y / HF(HOT_HS) + 0.5f);
        ix = min(max(ix, 0), HI(HOT_SDF_NX) - 1); iy = min(max(iy, 0), HI(HOT_SDF_NY) - 1);
        const float2 cc = reinterpret_cast<const float2*>(HP(HOT_WALL_CORNER_LO))[(size_t)ix * HI(HOT_SDF_NY) + iy];
        const float dx = pbase.x - cc.x, dy = pbase.y - cc.y, rr = 2.0f * HF(HOT_FEATURE_REACH) + 0.1f;
        nearw = dx * dx + dy * dy < rr * rr;
      }
      edge_pass = gballot(nearw) != 0ull;
    }
    if (edge_pass) {
      const int l = lane - sub * nsr, r = pass * rpp + sub;
      const bool isp = l >= 0 && l < npr && r < A && rm.prim_type[l < npr ? (l < 0 ? 0 : l) : 0] != MQE_PRIM_SPHERE;
      const int q = isp ? l : 0;
      V3 cq = v3(0, 0, 0), uq = v3(0, 0, 0), hb = v3(0, 0, 0);
      float Rb[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      const int ptype = rm.prim_type[q];
      if (isp) {
        const float* rec = lds + L.body + (r * MQE_NBODY + rm.prim_body[q]) * BODY_STRIDE;
        const float4 q0 = reinterpret_cast<const float4*>(rec)[0], q1 = reinterpret_cast<const float4*>(rec)[1], q2 = reinterpret_cast<const float4*>(rec)[2];
        Rb[0] = q0.x; Rb[1] = q0.y; Rb[2] = q0.z; Rb[3] = q0.w; Rb[4] = q1.x; Rb[5] = q1.y; Rb[6] = q1.z; Rb[7] = q1.w; Rb[8] = q2.x;
        cq = v3(q2.y, q2.z, q2.w) + mat_vec(Rb, v3(rm.prim_center[q][0], rm.prim_center[q][1], rm.prim_center[q][2]));
        uq = mat_vec(Rb, v3(rm.prim_axis[q][0], rm.prim_axis[q][1], rm.prim_axis[q][2]));
        hb = v3(rm.prim_half[q][0], rm.prim_half[q][1], rm.prim_half[q][2]);
        ebody = rm.prim_body[q]; erep = r * MQE_NREP + rm.prim_reported[q];
      }
      const float reach = rm.prim_bound[q] + HF(HOT_CONTACT_OFFSET);
      const float I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      const int njobs = 1 + ((HI(HOT_EDGE_MASK) & 6) ? shp.n_static : 0);
      for (int j = 0; j < njobs; j++) {
        bool valid = false; V3 bc = v3(0, 0, 0), hh = v3(0, 0, 0);
        if (j == 0) {
          if (isp && (HI(HOT_EDGE_MASK) & 1) != 0 && HP(HOT_WALL_CORNER_LO) != nullptr) {
            const float hs = HF(HOT_HS);
            int ix = (int)floorf(cq.x / hs + 0.5f), iy = (int)floorf(cq.y / hs + 0.5f);
            ix = min(max(ix, 0), HI(HOT_SDF_NX) - 1); iy = min(max(iy, 0), HI(HOT_SDF_NY) - 1);
            const float2 cc = reinterpret_cast<const float2*>(HP(HOT_WALL_CORNER_LO))[(size_t)ix * HI(HOT_SDF_NY) + iy];
            const float dx = cq.x - cc.x, dy = cq.y - cc.y;
            if (dx * dx + dy * dy < reach * reach) {
              // the wall's top at the corner: one height per scene or the map's value at the raster point nearest to the corner
              float zt = HF(HOT_WALL_HEIGHT);
              if (HP(HOT_WALL_TOP_LO) != nullptr) {
                float fx = fminf(fmaxf(cc.x / hs, 0.0f), (float)(HI(HOT_SDF_NX) - 1)), fy = fminf(fmaxf(cc.y / hs, 0.0f), (float)(HI(HOT_SDF_NY) - 1));
                int jx = min((int)fx, HI(HOT_SDF_NX) - 2), jy = min((int)fy, HI(HOT_SDF_NY) - 2);
                zt = HP(HOT_WALL_TOP_LO)[(size_t)((fx - jx) < 0.5f ? jx : jx + 1) * HI(HOT_SDF_NY) + ((fy - jy) < 0.5f ? jy : jy + 1)];
              }
              bc = v3(cc.x, cc.y, 0.5f * (HF(HOT_GROUND_Z) + zt)); hh = v3(0, 0, 0.5f * (zt - HF(HOT_GROUND_Z)));
              valid = true;
            }
          }
        } else if (isp && (ptype == MQE_PRIM_CAPSULE ? (HI(HOT_EDGE_MASK) & 2) != 0 : (HI(HOT_EDGE_MASK) & 4) != 0)) {      // capsule axes (bit 2) / box primitives (bit 4)
          const V3 nb = ld3(lds + L.root + A * 13);
          bc = nb + v3(m->sb_center[j - 1][0], m->sb_center[j - 1][1], m->sb_center[j - 1][2]);
          hh = v3(m->sb_half[j - 1][0], m->sb_half[j - 1][1], m->sb_half[j - 1][2]);
          valid = fabsf(cq.x - bc.x) < hh.x + reach && fabsf(cq.y - bc.y) < hh.y + reach && fabsf(cq.z - bc.z) < hh.z + reach;
        }
        if (gballot(valid) == 0ull) continue;                 // (wave-wide skip: nobody near a wall edge / this box)
        if (valid) {
          float sdj; V3 nj, pj;
          const bool got = (j > 0 && ptype == MQE_PRIM_BOX) ? box_edges_vs_prim_dev(cq, hb, Rb, bc, I3, hh, sdj, nj, pj)
                                                            : edge_vs_box_dev(ptype, cq, uq, hb.x, hb, Rb, m->prim_feat_t[q][0], m->prim_feat_t[q][1], bc, I3, hh, j == 0, sdj, nj, pj);
          if (got && (!eflag || sdj < esd)) {
            eflag = true; esd = sdj; en = nj; epa = pj;
          }
        }
      }
      eflag = eflag && esd < HF(HOT_CONTACT_OFFSET);
    }
    bool gflag = false, wflag = false, bflag = false, cflag = false;   // ground, wall, seesaw platform, seesaw column
    float gsd = 0, wsd = 0, bsd = 0, csd = 0; V3 gn = v3(0, 0, 1), wn = v3(0, 0, 1), bn = v3(0, 0, 1), cn3 = v3(0, 0, 1); V3 c = v3(0, 0, 0); float rad = 0;
    int body = 0, rep = 0;
    if (act >= 0) {
      const float* sp = lds + L.sph + s * 4;
      { const float4 q = *reinterpret_cast<const float4*>(sp); c = v3(q.x, q.y, q.z); rad = q.w; }
      if (act < A) { body = rm.sphere_body[sidx]; rep = act * MQE_NREP + rm.sphere_reported[sidx]; }
      else { body = 0; rep = A * MQE_NREP + (act - A); }
      // terrain maps are sampled bilinearly; raster entry (i, j) sits at the world point (i hs, j hs) -- the vertices of upstream's
      // convert_heightfield_to_trimesh mesh (barrier_track.py:483-497): the wall set's signed distance and, when the scene has one,
      // the relief of the walkable surface (Perlin noise, barrier_track.py:372-393)
      const float hs = HF(HOT_HS);
      float fx = c.x / hs, fy = c.y / hs;
      const int nx = HI(HOT_SDF_NX), ny = HI(HOT_SDF_NY);
      fx = fminf(fmaxf(fx, 0.0f), (float)(nx - 1)); fy = fminf(fmaxf(fy, 0.0f), (float)(ny - 1));
      int ix = (int)fx, iy = (int)fy;
      if (ix > nx - 2) ix = nx - 2;
      if (iy > ny - 2) iy = ny - 2;
      const float tx = fx - ix, ty = fy - iy;
      gsd = c.z - HF(HOT_GROUND_Z) - rad;
      if (HP(HOT_GROUND_HEIGHT_LO) != nullptr) {       // heightfield ground: first-order distance to the surface along its normal
        const float* gh = HP(HOT_GROUND_HEIGHT_LO) + (size_t)ix * ny + iy;
        const float h00 = gh[0], h01 = gh[1], h10 = gh[ny], h11 = gh[ny + 1];
        const float b0 = h00 + (h01 - h00) * ty, b1 = h10 + (h11 - h10) * ty;
        const float hx = (b1 - b0) / hs, hy = ((h01 - h00) + ((h11 - h10) - (h01 - h00)) * tx) / hs;
        const float inl = 1.0f / sqrtf(hx * hx + hy * hy + 1.0f);
        gn = v3(-hx * inl, -hy * inl, inl);
        gsd = (c.z - HF(HOT_GROUND_Z) - (b0 + (b1 - b0) * tx)) * inl - rad;
      }
      gflag = gsd < HF(HOT_CONTACT_OFFSET);
      const float* sd = HP(HOT_WALL_SDF_LO) + (size_t)ix * ny + iy;
      const float s00 = sd[0], s01 = sd[1], s10 = sd[ny], s11 = sd[ny + 1];
      const float a0 = s00 + (s01 - s00) * ty, a1 = s10 + (s11 - s10) * ty;
      float gx = (a1 - a0) / hs;
      float gy = ((s01 - s00) + ((s11 - s10) - (s01 - s00)) * tx) / hs;
      const float sh = a0 + (a1 - a0) * tx;
      float gl = sqrtf(gx * gx + gy * gy);
      if (gl < 1e-6f) { gx = 1; gy = 0; gl = 1; }
      gx /= gl; gy /= gl;
      // top of the wall this sphere is next to: one height per scene, or (walls of different heights) that of the wall nearest to
      // the sphere's own cell
      const float wtop = HP(HOT_WALL_TOP_LO) != nullptr ? HP(HOT_WALL_TOP_LO)[(size_t)(tx < 0.5f ? ix : ix + 1) * ny + (ty < 0.5f ? iy : iy + 1)] : HF(HOT_WALL_HEIGHT);
      const float dz = c.z - wtop;
      if (dz <= 0) {
        if (sh <= 0 && -sh > -dz) { wsd = dz - rad; wn = v3(0, 0, 1); }
        else { wsd = sh - rad; wn = v3(gx, gy, 0); }
      } else if (sh <= 0) { wsd = dz - rad; wn = v3(0, 0, 1); }
      else { const float dist = sqrtf(sh * sh + dz * dz); wsd = dist - rad; wn = v3(gx * sh / dist, gy * sh / dist, dz / dist); }
      wflag = wsd < HF(HOT_CONTACT_OFFSET);
      if (shp.n_static > 0 && act < A) {     // static scenery: the world-aligned box with the smallest signed distance
        const float I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        const V3 nb = ld3(lds + L.root + A * 13);
        bsd = 1e3f;
        for (int bx = 0; bx < shp.n_static; bx++) {
          V3 nn;
          const float sdb = sphere_box(c, rad, nb + v3(m->sb_center[bx][0], m->sb_center[bx][1], m->sb_center[bx][2]), I3,
                                       v3(m->sb_half[bx][0], m->sb_half[bx][1], m->sb_half[bx][2]), nn);
          if (sdb < bsd) { bsd = sdb; bn = nn; }
        }
        bflag = bsd < HF(HOT_CONTACT_OFFSET);
      }
      if (SS && act < A) {
        const float I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        bsd = sphere_box(c, rad, ssB, I3, v3(m->ss_base_half[0], m->ss_base_half[1], m->ss_base_half[2]), bn);
        bflag = bsd < HF(HOT_CONTACT_OFFSET) && m->ss_base_half[0] > 0.0f;       // no platform: tug-of-war slider
        const float dx = c.x - ssB.x, dy = c.y - ssB.y, rho = sqrtf(dx * dx + dy * dy);
        if (c.z < ssB.z && c.z > ssB.z - m->ss_col_length && rho > 1e-6f) { csd = rho - m->ss_col_radius - rad; cn3 = v3(dx / rho, dy / rho, 0); cflag = csd < HF(HOT_CONTACT_OFFSET); }
      }
    }
    // ---- desc.edge_contacts bit 8 (round 6, off by default): a robot that touches the static world at more points than it has slots keeps the
    // DEEPEST ones (otherwise, as in rounds 1-5: the first ones in feature order, the rest counted as overflow).  Depth counts from 1 mm of penetration on, in classes of 2 mm; everything shallower -- resting and speculative contacts -- is one class, so that a body at rest keeps the feature order's spread (feet, knees, hips, trunk corners) and the same set from substep to substep; ties in the canonical order (feature by
    // feature: ground, wall, platform, column; then the edge contacts, primitive by primitive) -- the steps keep the choice the same in
    // the oracle and here when a body lies flat and many separations agree to rounding.  Rare (a fallen robot): one wave-uniform test per
    // pass; the ranking walks the wavefront's candidates with v_readlane (oracle/mqe_oracle.c "manifold reduction").
    if (rob && (HI(HOT_EDGE_MASK) & 8) != 0) {
      const unsigned long long wg = __ballot(gflag), ww = __ballot(wflag), wb = __ballot(bflag), wc = __ballot(cflag), we = (HI(HOT_EDGE_MASK) & 7) != 0 ? __ballot(eflag) : 0ull;
      const unsigned long long mine_m = EPW == 1 ? gm : (gm << (grp * LW));                  // my robot's lanes, as lanes of the wavefront
      const int cnt = __popcll(wg & mine_m) + __popcll(ww & mine_m) + __popcll(wb & mine_m) + __popcll(wc & mine_m) + __popcll(we & mine_m);
      const bool over = cnt > cap && act >= 0;
      if (__ballot(cnt > cap) != 0ull) {
        auto bucket = [](float sd) -> int { return sd < -1e-3f ? (int)floorf((sd + 1e-3f) * 500.0f) : 0; };
        const int lw = lane_wave;
        const int kb[5] = {bucket(gsd), bucket(wsd), bucket(bsd), bucket(csd), bucket(esd)};
        int rk[5] = {0, 0, 0, 0, 0};
        unsigned long long um = wg | ww | wb | wc | we;
        while (um != 0ull) {
          const int j = __ffsll((long long)um) - 1;                 // wave-uniform
          um &= um - 1ull;
          const bool same = ((mine_m >> j) & 1ull) != 0ull;          // candidate lane j belongs to my robot
          const int bj[5] = {__builtin_amdgcn_readlane(kb[0], j), __builtin_amdgcn_readlane(kb[1], j), __builtin_amdgcn_readlane(kb[2], j),
                             __builtin_amdgcn_readlane(kb[3], j), __builtin_amdgcn_readlane(kb[4], j)};
          const bool fj[5] = {((wg >> j) & 1ull) != 0ull, ((ww >> j) & 1ull) != 0ull, ((wb >> j) & 1ull) != 0ull, ((wc >> j) & 1ull) != 0ull, ((we >> j) & 1ull) != 0ull};
#pragma unroll
          for (int t2 = 0; t2 < 5; t2++) {
            if (!fj[t2]) continue;                                   // wave-uniform
            const int keyj = (t2 == 4 ? 4096 : 0) + j * 4 + (t2 == 4 ? 0 : t2);
#pragma unroll
            for (int t = 0; t < 5; t++) {
              const int keym = (t == 4 ? 4096 : 0) + lw * 4 + (t == 4 ? 0 : t);
              rk[t] += (int)(same && (bj[t2] < kb[t] || (bj[t2] == kb[t] && keyj < keym)));
            }
          }
        }
        if (over) {
          gflag = gflag && rk[0] < cap; wflag = wflag && rk[1] < cap; bflag = bflag && rk[2] < cap; cflag = cflag && rk[3] < cap; eflag = eflag && rk[4] < cap;
          red = 1;
        }
      }
    }
    const unsigned long long bg = gballot(gflag), bw2 = gballot(wflag), bb2 = gballot(bflag), bc2 = gballot(cflag);
    const unsigned long long be = HI(HOT_EDGE_MASK) != 0 ? gballot(eflag) : 0ull;     // edge contacts: behind ALL feature contacts of their robot
    const unsigned long long lower = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const unsigned long long gl = gm & lower;
    const int pre = __popcll(bg & gl) + __popcll(bw2 & gl) + __popcll(bb2 & gl) + __popcll(bc2 & gl);   // rank in my actor
    int tot0 = __popcll(bg & m0) + __popcll(bw2 & m0) + __popcll(bb2 & m0) + __popcll(bc2 & m0) + __popcll(be & m0);          // first robot of the pass
    int tot1 = 0;
    if (tot0 > cap) { tot0 = cap; if (rob || !npc_one) ovf = 1; }       // (the all-NPC pass recounts per actor below)
    int base = nc;
    if (rob) {
      if (rpp == 2) {
        const unsigned long long m1 = m0 << nsr;
        tot1 = __popcll(bg & m1) + __popcll(bw2 & m1) + __popcll(bb2 & m1) + __popcll(bc2 & m1) + __popcll(be & m1);
        if (tot1 > cap) { tot1 = cap; ovf = 1; }
      }
      base = nc + (sub ? tot0 : 0);
    } else if (npc_one) {                  // per actor: its (capped) count; a lane's base = the counts of the actors before its own
      const int myp = act >= 0 ? act - A : PD;
      tot0 = 0;
      for (int q = 0; q < PD; q++) {
        const unsigned long long mq = mns << (q * nsn);
        int tq = __popcll(bg & mq) + __popcll(bw2 & mq);
        if (tq > cap) { tq = cap; ovf = 1; }
        if (q < myp) base += tq;
        tot0 += tq;
      }
    }
    if (gflag) {
      const int slot = base + pre;
      if (slot < maxc && pre < cap) {
        float* cr = lds + L.con + slot * CON_STRIDE;
        con_store(cr, act, body, -1, 0, c - rad * gn, gn, gsd, rep, -1);
      }
    }
    if (wflag) {
      const int rk = pre + (gflag ? 1 : 0), slot = base + rk;
      if (slot < maxc && rk < cap) {
        float* cr = lds + L.con + slot * CON_STRIDE;
        con_store(cr, act, body, -1, 0, c - rad * wn, wn, wsd, rep, -1);
      }
    }
    if (bflag || cflag) {
      for (int which = 0; which < 2; which++) {
        if (which == 0 ? !bflag : !cflag) continue;
        const int rk = pre + (gflag ? 1 : 0) + (wflag ? 1 : 0) + (which == 1 && bflag ? 1 : 0), slot = base + rk;
        if (slot < maxc && rk < cap) {
          const V3 nn = which == 0 ? bn : cn3;
          float* cr = lds + L.con + slot * CON_STRIDE;
          con_store(cr, act, body, -1, 0, c - rad * nn, nn, which == 0 ? bsd : csd, rep, -1);
        }
      }
    }
    if (eflag) {                                   // behind every feature contact of my robot, primitive by primitive
      const int rk = __popcll(bg & gm) + __popcll(bw2 & gm) + __popcll(bb2 & gm) + __popcll(bc2 & gm) + __popcll(be & gl), slot = base + rk;
      if (slot < maxc && rk < cap) {
        float* cr = lds + L.con + slot * CON_STRIDE;
        con_store(cr, pass * rpp + sub, ebody, -1, 0, epa, en, esd, erep, -1);
      }
    }
    nc += tot0 + tot1;
    if (nc > maxc) { nc = maxc; ovf = 1; }
  }
  // robot spheres vs the seesaw plank (dynamic: couples the robots through the hinge); after all terrain contacts
  const int nc_terr = nc;                                   // one-sided contacts end here; two-actor contacts follow
  const int pair_lim = nc_terr + mqe_maxpair(maxc) < maxc ? nc_terr + mqe_maxpair(maxc) : maxc;
  // edge contacts of robot a's capsule primitives (lanes) with a moving box of the scene -- the plank / door, the free box -- (desc.edge_contacts
  // bit 2; oracle: "... and the plank's / door's edges", "... and its edges against the robot's primitives"): the closest approach of the
  // capsule's whole axis.  `room` = how many more contacts this robot may add (the plank's per-robot share), actor / reported body of the box.
  auto pair_edges = [&](int a, V3 bc, const float* Rbx, V3 hbx, int actB, int repB, int room) {
    const int ptq = rm.prim_type[lane < npr ? lane : 0];
    const bool isp = lane < npr && (ptq == MQE_PRIM_CAPSULE ? (HI(HOT_EDGE_MASK) & 2) != 0 : (ptq == MQE_PRIM_BOX && (HI(HOT_EDGE_MASK) & 4) != 0));
    const int q = isp ? lane : 0;
    V3 cq = v3(0, 0, 0), uq = v3(0, 0, 0);
    float Rr[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    bool valid = false;
    if (isp) {
      const float* rec = lds + L.body + (a * MQE_NBODY + rm.prim_body[q]) * BODY_STRIDE;
      const float4 q0 = reinterpret_cast<const float4*>(rec)[0], q1 = reinterpret_cast<const float4*>(rec)[1], q2 = reinterpret_cast<const float4*>(rec)[2];
      Rr[0] = q0.x; Rr[1] = q0.y; Rr[2] = q0.z; Rr[3] = q0.w; Rr[4] = q1.x; Rr[5] = q1.y; Rr[6] = q1.z; Rr[7] = q1.w; Rr[8] = q2.x;
      cq = v3(q2.y, q2.z, q2.w) + mat_vec(Rr, v3(rm.prim_center[q][0], rm.prim_center[q][1], rm.prim_center[q][2]));
      uq = mat_vec(Rr, v3(rm.prim_axis[q][0], rm.prim_axis[q][1], rm.prim_axis[q][2]));
      // screen: the primitive's bounding sphere against the box's (a point inside the box passes too)
      const V3 dd = cq - bc;
      const float reach = rm.prim_bound[q] + HF(HOT_CONTACT_OFFSET) + sqrtf(dot(hbx, hbx));
      valid = dot(dd, dd) < reach * reach;
    }
    if (gballot(valid) == 0ull) return;
    bool hit = false; float sd = 0; V3 n = v3(0, 0, 1), pa = v3(0, 0, 0);
    if (valid) {
      const float I9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      hit = (ptq == MQE_PRIM_BOX ? box_edges_vs_prim_dev(cq, v3(rm.prim_half[q][0], rm.prim_half[q][1], rm.prim_half[q][2]), Rr, bc, Rbx, hbx, sd, n, pa)
                                 : edge_vs_box_dev(MQE_PRIM_CAPSULE, cq, uq, rm.prim_half[q][0], v3(0, 0, 0), I9, m->prim_feat_t[q][0], m->prim_feat_t[q][1], bc, Rbx, hbx, false, sd, n, pa))
            && sd < HF(HOT_CONTACT_OFFSET);
    }
    const unsigned long long bh = gballot(hit);
    if (bh == 0ull) return;
    const unsigned long long lower = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int rk = __popcll(bh & lower), slot = nc + rk;
    if (hit && rk < room && slot < pair_lim) {
      float* cr = lds + L.con + slot * CON_STRIDE;
      con_store(cr, a, rm.prim_body[q], actB, 0, pa, n, sd, a * MQE_NREP + rm.prim_reported[q], repB);
    }
    int tot = __popcll(bh);
    if (tot > room) { tot = room < 0 ? 0 : room; ovf = 1; }
    nc += tot;
    if (nc > pair_lim) { nc = pair_lim; ovf = 1; }
  };
  if (SS) {
    const int capP = mqe_maxpair(maxc) / A;              // per robot, so that the first robot cannot starve the others
    for (int a = 0; a < A; a++) {                        // one robot per iteration: lanes = its spheres
      const int s = a * nsr + lane;
      bool hit = false; float sd = 0; V3 n = v3(0, 0, 1), c = v3(0, 0, 0); float rad = 0; int body = 0, rep = 0;
      if (lane < nsr) {
        const float* sp = lds + L.sph + s * 4;
        { const float4 q = *reinterpret_cast<const float4*>(sp); c = v3(q.x, q.y, q.z); rad = q.w; }
        body = rm.sphere_body[lane]; rep = a * MQE_NREP + rm.sphere_reported[lane];
        sd = m->ss_link_cyl ? sphere_vcyl(c, rad, ssC, m->ss_plank_half[0], m->ss_plank_half[2], n)
                            : sphere_box(c, rad, ssC, ssR, v3(m->ss_plank_half[0], m->ss_plank_half[1], m->ss_plank_half[2]), n);
        hit = sd < HF(HOT_CONTACT_OFFSET);
      }
      const unsigned long long bh = gballot(hit);
      const unsigned long long lower = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
      const int rk = __popcll(bh & lower), slot = nc + rk;
      if (hit && rk < capP && slot < pair_lim) {
        float* cr = lds + L.con + slot * CON_STRIDE;
        con_store(cr, a, body, A, 0, c - rad * n, n, sd, rep, A * MQE_NREP + 1);
      }
      int tot = __popcll(bh);
      if (tot > capP) { tot = capP; ovf = 1; }
      nc += tot;
      if (nc > pair_lim) { nc = pair_lim; ovf = 1; }
      if ((HI(HOT_EDGE_MASK) & 6) != 0 && !m->ss_link_cyl)
        pair_edges(a, ssC, ssR, v3(m->ss_plank_half[0], m->ss_plank_half[1], m->ss_plank_half[2]), A, A * MQE_NREP + 1, capP - tot);
    }
  }
  TSTAMP(9);
  // ---- sphere-sphere contacts between different actors (a < b; outer loop over b's spheres, lanes = a's spheres) -------
  {
    const int nact = A + PD;
    // pairs of two NPCs of one or two spheres (a flock) are tested lane-parallel, 64 sphere pairs per pass, after the robots' pairs:
    // in the canonical order (a, b, sphere of b, sphere of a) they come last anyway, and 36 wave-uniform iterations for 9 sheep
    // were a third of this phase
    const bool npc_pass = PD > 1 && HI(HOT_NPC_N_SPHERES) <= 2 && !shp.has_box;
    const int a_end = npc_pass ? A : nact;
    int tp = -1;
    for (int a = 0; a < a_end; a++)
      for (int b = a + 1; b < nact; b++) {
        tp++;
        if (!(((tp < LW ? near0 >> tp : (tp < 2 * LW ? near1 >> (tp - LW) : near2 >> (tp - 2 * LW))) & 1ull))) continue;          // uniform within the group
        const V3 pb = ld3(lds + L.body + (b < A ? b * MQE_NBODY : A * MQE_NBODY + (b - A)) * BODY_STRIDE + B_P);
        if (shp.has_box && b >= A) {                           // robot spheres vs the oriented box (NPC body record = its pose)
          const float* brec = lds + L.body + (A * MQE_NBODY + (b - A)) * BODY_STRIDE;
          bool hit = false; float sd = 0; V3 n = v3(0, 0, 1), c = v3(0, 0, 0); float ra = 0;
          if (lane < nsr) {
            const float* spa = lds + L.sph + (a * nsr + lane) * 4;
            { const float4 q = *reinterpret_cast<const float4*>(spa); c = v3(q.x, q.y, q.z); ra = q.w; }
            sd = sphere_box(c, ra, pb, brec + B_R, v3(m->npc_box_half[0], m->npc_box_half[1], m->npc_box_half[2]), n);
            hit = sd < HF(HOT_CONTACT_OFFSET);
          }
          const unsigned long long bh = gballot(hit);
          const unsigned long long lower = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
          const int slot = nc + __popcll(bh & lower);
          if (hit && slot < pair_lim) {
            float* cr = lds + L.con + slot * CON_STRIDE;
            con_store(cr, a, rm.sphere_body[lane], b, 0, c - (ra + 0.5f * sd) * n, n, sd, a * MQE_NREP + rm.sphere_reported[lane], A * MQE_NREP + (b - A));
          }
          nc += __popcll(bh);
          if (nc > pair_lim) { nc = pair_lim; ovf = 1; }
          // ... and the box's own eight corners (radius 0) against the robot's primitives (lanes = primitives): a corner of the box pressing
          // into a face of the trunk or into a bar between its ends.  Corners farther from the robot's base than any feature point reaches
          // are dropped with one ballot (lane = corner): a robot pushing a face of the 1 m box has none.
          unsigned long long cmask;
          {
            const V3 pbase = ld3(lds + L.body + a * MQE_NBODY * BODY_STRIDE + B_P);
            const V3 hbx = v3(m->npc_box_half[0], m->npc_box_half[1], m->npc_box_half[2]);
            bool nearc = false;
            if (lane < 8) {
              const V3 cw = pb + mat_vec(brec + B_R, v3((lane & 4) ? hbx.x : -hbx.x, (lane & 2) ? hbx.y : -hbx.y, (lane & 1) ? hbx.z : -hbx.z));
              const V3 db = cw - pbase;
              const float reach = HF(HOT_FEATURE_REACH) + HF(HOT_CONTACT_OFFSET);
              nearc = !(dot(db, db) > reach * reach);
            }
            cmask = gballot(nearc);
          }
          if (cmask != 0ull) {
            float4 w0 = make_float4(0, 0, 0, 0), w1 = w0; int pbody = 0, prep = 0; V3 ph = v3(0, 0, 0);
            if (lane < npr) {
              const float4* pw = reinterpret_cast<const float4*>(lds + L.prim) + (a * npr + lane);
              w0 = pw[0]; w1 = pw[HI(HOT_NPRIM_ENV)];
              pbody = rm.prim_body[lane]; prep = rm.prim_reported[lane];
              ph = v3(rm.prim_half[lane][0], rm.prim_half[lane][1], rm.prim_half[lane][2]);
            }
            const V3 hbx = v3(m->npc_box_half[0], m->npc_box_half[1], m->npc_box_half[2]);
            while (cmask != 0ull) {
              const int cn = __ffsll((long long)cmask) - 1;
              cmask &= cmask - 1ull;
              const V3 cc = pb + mat_vec(brec + B_R, v3((cn & 4) ? hbx.x : -hbx.x, (cn & 2) ? hbx.y : -hbx.y, (cn & 1) ? hbx.z : -hbx.z));
              bool hit2 = false; float sd2 = 0; V3 n2 = v3(0, 0, 1);
              if (lane < npr) {
                const V3 cq = v3(w0.x, w0.y, w0.z);
                if (w1.w < 0.0f) {
                  sd2 = sphere_box(cc, 0.0f, cq, lds + L.body + (a * MQE_NBODY + pbody) * BODY_STRIDE + B_R, ph, n2);
                  hit2 = sd2 < HF(HOT_CONTACT_OFFSET);
                } else {
                  const bool ok = sphere_capsule(cc, 0.0f, cq, v3(w1.x, w1.y, w1.z), w1.w, sd2, n2);
                  hit2 = ok && sd2 < HF(HOT_CONTACT_OFFSET);
                }
              }
              const unsigned long long bh2 = gballot(hit2);
              if (bh2 == 0ull) continue;
              const int slot2 = nc + __popcll(bh2 & lower);
              if (hit2 && slot2 < pair_lim) {
                float* cr = lds + L.con + slot2 * CON_STRIDE;
                con_store(cr, a, pbody, b, 0, cc - (0.5f * sd2) * n2, v3(-n2.x, -n2.y, -n2.z), sd2, a * MQE_NREP + prep, A * MQE_NREP + (b - A));
              }
              nc += __popcll(bh2);
              if (nc > pair_lim) { nc = pair_lim; ovf = 1; }
            }
          }
          if ((HI(HOT_EDGE_MASK) & 6) != 0)
            pair_edges(a, pb, brec + B_R, v3(m->npc_box_half[0], m->npc_box_half[1], m->npc_box_half[2]), b, A * MQE_NREP + (b - A), 64);
          continue;
        }
        if (b < A) {
          // two robots: the feature points of one (lanes) against the primitives of the other (wave-uniform loop), both ways; a foot
          // against a foot is the same sphere pair both ways and is taken the first time only.  Every primitive is screened with
          // its bounding sphere first, so a pair of robots that is merely close costs two compares per primitive.
          for (int dir = 0; dir < 2; dir++) {
            const int fa = dir == 0 ? a : b, qa = dir == 0 ? b : a;
            V3 c = v3(0, 0, 0); float ra = 0; bool foot = false;
            if (lane < nsr) {
              const float4 qf = *reinterpret_cast<const float4*>(lds + L.sph + (fa * nsr + lane) * 4);
              c = v3(qf.x, qf.y, qf.z); ra = qf.w;
              foot = ((HMASK() >> lane) & 1ull) != 0ull;
            }
            // which primitives of qa reach into the ball around fa's base that holds all of fa's feature points in THIS pose (its
            // radius: a maximum over the feature lanes; lane = primitive, one ballot): two robots walking side by side normally
            // have none, and the loop below does not run at all
            unsigned long long qmask;
            {
              const V3 pbase = ld3(lds + L.body + fa * MQE_NBODY * BODY_STRIDE + B_P);
              const V3 db = c - pbase;
              const float rfeat = wave_max_nonneg<EPW == 2>(lane < nsr ? sqrtf(dot(db, db)) + ra : 0.0f);
              bool act = false;
              if (lane < npr) {
                const float4 w0 = reinterpret_cast<const float4*>(lds + L.prim)[qa * npr + lane];
                const V3 df = v3(w0.x, w0.y, w0.z) - pbase;
                const float reach = w0.w + rfeat + HF(HOT_CONTACT_OFFSET);
                act = dot(df, df) < reach * reach;
              }
              qmask = gballot(act);
            }
            while (qmask != 0ull) {
              const int q = __ffsll((long long)qmask) - 1;
              qmask &= qmask - 1ull;
              const float4* pw = reinterpret_cast<const float4*>(lds + L.prim) + (qa * npr + q);
              const float4 w0 = pw[0], w1 = pw[HI(HOT_NPRIM_ENV)];
              const V3 cq = v3(w0.x, w0.y, w0.z);
              const bool qbox = w1.w < 0.0f, qsph = !qbox && w1.x == 0.0f && w1.y == 0.0f && w1.z == 0.0f;      // wave-uniform
              const V3 dq = c - cq;
              const float reach = ra + w0.w + HF(HOT_CONTACT_OFFSET);
              bool cand = lane < nsr && dot(dq, dq) < reach * reach && !(dir == 1 && foot && qsph);
              if (gballot(cand) == 0ull) continue;
              bool hit = false; float sd = 0; V3 n = v3(0, 0, 1);
              if (cand) {
                if (qbox) {
                  sd = sphere_box(c, ra, cq, lds + L.body + (qa * MQE_NBODY + rm.prim_body[q]) * BODY_STRIDE + B_R,
                                  v3(rm.prim_half[q][0], rm.prim_half[q][1], rm.prim_half[q][2]), n);
                  hit = sd < HF(HOT_CONTACT_OFFSET);
                } else {
                  const bool ok = sphere_capsule(c, ra, cq, v3(w1.x, w1.y, w1.z), w1.w, sd, n);
                  hit = ok && sd < HF(HOT_CONTACT_OFFSET);
                }
              }
              const unsigned long long bh = gballot(hit);
              if (bh == 0ull) continue;
              const unsigned long long lower = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
              const int slot = nc + __popcll(bh & lower);
              if (hit && slot < pair_lim) {
                float* cr = lds + L.con + slot * CON_STRIDE;
                con_store(cr, fa, rm.sphere_body[lane], qa, rm.prim_body[q], c - (ra + 0.5f * sd) * n, n, sd, fa * MQE_NREP + rm.sphere_reported[lane],
                          qa * MQE_NREP + rm.prim_reported[q]);
              }
              nc += __popcll(bh);
              if (nc > pair_lim) { nc = pair_lim; ovf = 1; }
            }
          }
          continue;
        }
        if (a < A) {
          // robot a against the collision spheres of free NPC b (ball, sheep): lanes = the robot's primitives, one sphere of b per
          // iteration; the contact's normal points from B (the NPC) to A
          const int nb = HI(HOT_NPC_N_SPHERES), ob = A * nsr + (b - A) * HI(HOT_NPC_N_SPHERES);
          float4 w0 = make_float4(0, 0, 0, 0), w1 = w0; int ptype = MQE_PRIM_SPHERE, pbody = 0, prep = 0; V3 ph = v3(0, 0, 0);
          if (lane < npr) {
            const float4* pw = reinterpret_cast<const float4*>(lds + L.prim) + (a * npr + lane);
            w0 = pw[0]; w1 = pw[HI(HOT_NPRIM_ENV)];
            ptype = w1.w < 0.0f ? MQE_PRIM_BOX : MQE_PRIM_CAPSULE; pbody = rm.prim_body[lane]; prep = rm.prim_reported[lane];
            ph = v3(rm.prim_half[lane][0], rm.prim_half[lane][1], rm.prim_half[lane][2]);
          }
          for (int sb = 0; sb < nb; sb++) {
            const float4 qb = *reinterpret_cast<const float4*>(lds + L.sph + (ob + sb) * 4);
            const V3 cb = v3(qb.x, qb.y, qb.z); const float rb = qb.w;
            bool hit = false; float sd = 0; V3 n = v3(0, 0, 1);
            if (lane < npr) {
              const V3 cq = v3(w0.x, w0.y, w0.z);
              if (ptype == MQE_PRIM_BOX) {
                sd = sphere_box(cb, rb, cq, lds + L.body + (a * MQE_NBODY + pbody) * BODY_STRIDE + B_R, ph, n);
                hit = sd < HF(HOT_CONTACT_OFFSET);
              } else {
                const bool ok = sphere_capsule(cb, rb, cq, v3(w1.x, w1.y, w1.z), w1.w, sd, n);
                hit = ok && sd < HF(HOT_CONTACT_OFFSET);
              }
            }
            const unsigned long long bh = gballot(hit);
            if (bh == 0ull) continue;
            const unsigned long long lower = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
            const int slot = nc + __popcll(bh & lower);
            if (hit && slot < pair_lim) {
              float* cr = lds + L.con + slot * CON_STRIDE;
              con_store(cr, a, pbody, b, 0, cb - (rb + 0.5f * sd) * n, v3(-n.x, -n.y, -n.z), sd, a * MQE_NREP + prep, A * MQE_NREP + (b - A));
            }
            nc += __popcll(bh);
            if (nc > pair_lim) { nc = pair_lim; ovf = 1; }
          }
          continue;
        }
        const int na = HI(HOT_NPC_N_SPHERES), nb = HI(HOT_NPC_N_SPHERES);        // two free NPCs: sphere pairs (flocks take the lane-parallel pass below)
        const int oa = A * nsr + (a - A) * HI(HOT_NPC_N_SPHERES);
        const int ob = A * nsr + (b - A) * HI(HOT_NPC_N_SPHERES);
        for (int sb = 0; sb < nb; sb++) {
          const float* spb = lds + L.sph + (ob + sb) * 4;
          const float4 qb = *reinterpret_cast<const float4*>(spb);
          const V3 cb = v3(qb.x, qb.y, qb.z); const float rb = qb.w;
          bool hit = false; float sd = 0, dist = 1; V3 ev = v3(0, 0, 0); float ra = 0;
          if (lane < na) {
            const float* spa = lds + L.sph + (oa + lane) * 4;
            { const float4 q = *reinterpret_cast<const float4*>(spa); ev = v3(q.x, q.y, q.z) - cb; ra = q.w; }
            dist = sqrtf(dot(ev, ev));
            sd = dist - ra - rb;
            hit = sd < HF(HOT_CONTACT_OFFSET) && dist > 1e-9f;
          }
          const unsigned long long bh = gballot(hit);
          if (bh == 0ull) continue;
          const unsigned long long lower = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
          const int slot = nc + __popcll(bh & lower);
          if (hit && slot < pair_lim) {
            float* cr = lds + L.con + slot * CON_STRIDE;
            const V3 n = (1.0f / dist) * ev;
            con_store(cr, a, 0, b, 0, cb + (rb + 0.5f * sd) * n, n, sd, A * MQE_NREP + (a - A), A * MQE_NREP + (b - A));
          }
          nc += __popcll(bh);
          if (nc > pair_lim) { nc = pair_lim; ovf = 1; }
        }
      }
    if (npc_pass && near_npcs) {                       // (wave-uniform: a flock in which no two sheep are within reach of each other skips its pair passes)
      const int ns = HI(HOT_NPC_N_SPHERES), ns2 = ns * ns;
      const int np2 = ((PD * (PD - 1)) / 2) * ns2;
      for (int t0 = 0; t0 < np2; t0 += LW) {
        const int t = t0 + lane;
        bool hit = false; float sd = 0, dist = 1, rb = 0; V3 ev = v3(0, 0, 0), cb = v3(0, 0, 0); int i = 0, j = 0;
        if (t < np2) {
          int rem = t / ns2;
          const int ss = t - rem * ns2, sb = ss / ns, sa = ss - sb * ns;
          while (rem >= PD - 1 - i) { rem -= PD - 1 - i; i++; }
          j = i + 1 + rem;
          const float4 qa = *reinterpret_cast<const float4*>(lds + L.sph + (A * nsr + i * ns + sa) * 4);
          const float4 qb = *reinterpret_cast<const float4*>(lds + L.sph + (A * nsr + j * ns + sb) * 4);
          cb = v3(qb.x, qb.y, qb.z); rb = qb.w;
          ev = v3(qa.x, qa.y, qa.z) - cb;
          dist = sqrtf(dot(ev, ev));
          sd = dist - qa.w - rb;
          hit = sd < HF(HOT_CONTACT_OFFSET) && dist > 1e-9f;
        }
        const unsigned long long bh = gballot(hit);
        if (bh == 0ull) continue;
        const unsigned long long lower = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        const int slot = nc + __popcll(bh & lower);
        if (hit && slot < pair_lim) {
          float* cr = lds + L.con + slot * CON_STRIDE;
          const V3 n = (1.0f / dist) * ev;
          con_store(cr, A + i, 0, A + j, 0, cb + (rb + 0.5f * sd) * n, n, sd, A * MQE_NREP + i, A * MQE_NREP + j);
        }
        nc += __popcll(bh);
        if (nc > pair_lim) { nc = pair_lim; ovf = 1; }
      }
    }
    // links of one robot against each other (asset.self_collisions = 0): lanes = the model's candidate (feature point, primitive)
    // pairs (links neither the same nor adjacent, reachable inside the joint limits), 64 per pass; both contact sides belong to the
    // same actor.  Last in the list: they only take the two-actor slots that the contacts with other actors left over.
    if (HI(HOT_SELF_COLLISION) && self_todo != 0u)
      for (int ac = 0; ac < A * ((HI(HOT_N_SELF_PAIRS) + SELF_CHUNK - 1) / SELF_CHUNK); ac++) {      // robot by robot, each robot chunk by chunk (list order)
        const int nchunk = (HI(HOT_N_SELF_PAIRS) + SELF_CHUNK - 1) / SELF_CHUNK;
        const int a = ac / nchunk, chunk0 = (ac - a * nchunk) * SELF_CHUNK;
        const int npairs = min(HI(HOT_N_SELF_PAIRS) - chunk0, SELF_CHUNK);          // candidates of this chunk
        // joint-space screen (above): inside the model's safe box of joint angles no candidate pair is closer than 4 cm
        if (!((self_todo >> a) & 1u)) continue;
        if (nchunk > 1) {                                                      // (a one-chunk model keeps the entries requested at the top)
#pragma unroll
          for (int k = 0; k < NSP; k++) selfp[k] = k * LW + lane < npairs ? (int)rm.self_pair[chunk0 + k * LW + lane] : -1;
        }
        // screen: all passes at once (independent 16 B loads, one ballot).  Bounding SPHERES (11 cm for a thigh or calf, 20 cm for the
        // trunk) would let the neighbouring legs and the thigh tops through in every substep, so the screen is the distance to the
        // capsule's segment itself (a box: its bounding capsule); robots rarely touch themselves and the compaction below normally
        // does not run at all
        const float* sp0 = lds + L.sph + a * nsr * 4;
        const float4* pp0 = reinterpret_cast<const float4*>(lds + L.prim) + a * npr;       // [q]: centre, bound; [nprim_env + q]: half-segment, radius
        float4 si4[NSP], sj4[NSP];
        int any = 0;
#pragma unroll
        for (int k = 0; k < NSP; k++) {                          // beyond the list: feature 0 against primitive 0, masked below
          const int pr = selfp[k] < 0 ? 0 : selfp[k];
          si4[k] = *reinterpret_cast<const float4*>(sp0 + (pr & 255) * 4);
          sj4[k] = pp0[pr >> 8];
          const float4 uj = pp0[HI(HOT_NPRIM_ENV) + (pr >> 8)];
          const V3 u = v3(uj.x, uj.y, uj.z), dq = v3(si4[k].x - sj4[k].x, si4[k].y - sj4[k].y, si4[k].z - sj4[k].z);
          const float uu = dot(u, u);
          const float t = uu > 0.0f ? fminf(fmaxf(dot(dq, u) * __builtin_amdgcn_rcpf(uu), -1.0f), 1.0f) : 0.0f;
          const V3 e = dq - t * u;
          const float lim = si4[k].w + fabsf(uj.w) + HF(HOT_CONTACT_OFFSET) + 1e-5f;                                       // (+ the screen's own rounding)
          any |= (int)(selfp[k] >= 0) & (int)(dot(e, e) < lim * lim);
        }
        if (gballot(any != 0) == 0ull) continue;
#pragma unroll
        for (int k = 0; k < NSP; k++) {
          if (k * LW >= npairs) continue;
          bool hit = false; float sd = 0; V3 n = v3(0, 0, 1), c = v3(0, 0, 0); float ra = 0; int f = 0, q = 0;
          if (selfp[k] >= 0) {
            f = selfp[k] & 255; q = selfp[k] >> 8;
            c = v3(si4[k].x, si4[k].y, si4[k].z); ra = si4[k].w;
            const V3 cq = v3(sj4[k].x, sj4[k].y, sj4[k].z);
            const float4 w1 = pp0[HI(HOT_NPRIM_ENV) + q];
            if (w1.w < 0.0f) {
              sd = sphere_box(c, ra, cq, lds + L.body + (a * MQE_NBODY + rm.prim_body[q]) * BODY_STRIDE + B_R,
                              v3(rm.prim_half[q][0], rm.prim_half[q][1], rm.prim_half[q][2]), n);
              hit = sd < HF(HOT_CONTACT_OFFSET);
            } else {
              const bool ok = sphere_capsule(c, ra, cq, v3(w1.x, w1.y, w1.z), w1.w, sd, n);
              hit = ok && sd < HF(HOT_CONTACT_OFFSET);
            }
          }
          const unsigned long long bh = gballot(hit);
          if (bh == 0ull) continue;
          const unsigned long long lower = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
          const int slot = nc + __popcll(bh & lower);
          if (hit && slot < pair_lim) {
            float* cr = lds + L.con + slot * CON_STRIDE;
            con_store(cr, a, rm.sphere_body[f], a, rm.prim_body[q], c - (ra + 0.5f * sd) * n, n, sd, a * MQE_NREP + rm.sphere_reported[f],
                      a * MQE_NREP + rm.prim_reported[q]);
          }
          nc += __popcll(bh);
          if (nc > pair_lim) { nc = pair_lim; ovf = 1; }
        }
      }
  }
  __syncthreads();

  TSTAMP(10);
  // ---- contact rows, lane = contact: sparse Jacobian (<= 9 columns per articulated side), B = M^-1 J^T, bias --------------------
  // Per side the columns are [base lin xyz, base ang xyz, the <=3 joints of the chain to the touching link] (robot) or
  // [lin xyz, (ang xyz)] (ball / sheep).  Column value in the contact frame: dirs . (axis x (p - anchor)) = (r x dirs) . axis.
  // friction of a contact: the shapes of a robot carry the env's (randomised) coefficient, averaged with the other shape's
  // (terrain / objects: HF(HOT_FRICTION)) as PhysX does; contacts without a robot keep HF(HOT_FRICTION)
  const float mu_robot = 0.5f * (mu_env + HF(HOT_FRICTION));
  float* Vm = lds + L.rhs;           // v* (unconstrained velocity) over the consumed bias vector, read through the sparse Jacobian rows
  if (lane < ndof) Vm[lane] = vs0;
  if (lane + LW < ndof) Vm[lane + LW] = vs1;
  __syncthreads();
  // Contact solver (include/mqe_hip.h solver_type, oracle/mqe_oracle.c "contact solver"): 0 = sweeps on velocities with the erp bias
  // of the start-of-step separation; 1 = temporal Gauss-Seidel -- npos sub-steps of sdt = dt / npos, each re-evaluating every contact's
  // separation from the motion accumulated so far.  In the Phi form that motion is sdt (k v* + T W_k), W_k = the sum of w at the end
  // of the k finished position iterations (L.wacc), so  sep = sd + sdt (k u*_n + Phi_n W_k):  one more row sum per sweep step.
  // The sweeps themselves are the same for both: a contact's normal row asks for u_n >= bias.  Solver 0 fixes the bias at the start;
  // solver 1 re-derives it between the sweeps, all contacts at once (lane = contact): sep += sdt (u*_n + Phi_n . w), bias = -sep / sdt
  // capped at the depenetration speed -- nothing is added to the 64 dependent steps of the sweep.  Velocity iterations (either solver)
  // see a penetration as touching.  W itself is only needed for the positions at the end: a register per lane (coordinates lane, lane + LW).
  const bool tgs = HI(HOT_SOLVER_TYPE) == 1;
  const int npos = HI(HOT_SOLVER_ITERATIONS), nsweeps = npos + HI(HOT_VEL_ITERS);
  const float sdt = tgs ? dt / (float)npos : dt, inv_sdt = 1.0f / sdt;
  float* waccv = lds + L.wacc;
  auto first_bias = [&](float sd) -> float {
    if (!tgs) return sd >= 0 ? -sd / dt : (npos > 0 ? fminf(-sd * HF(HOT_ERP) / dt, HF(HOT_MAX_DEPEN)) : 0.0f);
    const float s0 = npos > 0 ? sd : fmaxf(sd, 0.0f);
    const float b = -s0 * inv_sdt;
    return s0 < 0.0f ? fminf(b, HF(HOT_MAX_DEPEN)) : b;
  };
  // bias of sweep `it + 1` from the separation after sweep `it`; sep is advanced by the sub-step's normal motion sdt * un when sweep `it` was a position iteration
  auto next_bias = [&](float& sep, float un, float bias_now, int it) -> float {
    const bool next_vel = it + 1 >= npos;
    if (!tgs) return next_vel ? fminf(bias_now, 0.0f) : bias_now;
    if (it < npos) sep += sdt * un;
    const float s2 = next_vel ? fmaxf(sep, 0.0f) : sep;
    const float b = -s2 * inv_sdt;
    return s2 < 0.0f ? fminf(b, HF(HOT_MAX_DEPEN)) : b;
  };
  float Wacc0 = 0.0f, Wacc1 = 0.0f;
  const bool is_con = lane < nc;
  float us0 = 0, us1 = 0, us2 = 0, cl0 = 0, cl1 = 0, cl2 = 0, cbias = 0, csep = 0;      // relative velocity of the unconstrained motion, impulse, bias
  float d00 = 0, d10 = 0, d11 = 0, d20 = 0, d21 = 0, d22 = 0;                // my contact's own 3 x 3 block K(c, c) = sum over sides Phi Phi^T
  int myA = -2, myB = -2;                                                   // actors of my contact's sides
  int wA = 0, wB = 0, infoA = 0, infoB = 0;                                  // first generalized coordinate of each side's actor; the side's info word
  int lgA = 0, lgB = 0;                                                     // leg + 1 of each side's touching link (0: base / not a robot)
  float mu = HF(HOT_FRICTION);
  float fA[27];                                                             // Phi of my contact's side A (U 18, Z' 9): registers for the whole sweep
#pragma unroll
  for (int i = 0; i < 27; i++) fA[i] = 0.0f;
  for (int i = lane; i < ((ndof + 3) & ~3); i += LW) accv[i] = 0.0f;          // w = sum_c Phi_c^T lambda_c starts at zero (no warm start)
  if (shp.rowgs) {
    // ---- side records for the row sweep: lane = (contact, direction).  The three rows of a contact (normal, two tangents) are
    // independent of each other up to the contact's own 3 x 3 block, so four lanes share a contact (lane & 3 = row, the fourth
    // idles), each builds ONE row of Phi per side, and the block's couplings come from the neighbour's row through a quad
    // permutation: a third of the instructions of the one-lane-per-contact form below.  Rows of side A wait in registers until
    // every lane has read the link records (the records go on top of them); side B has its own area.
    constexpr int NPQ = EPW == 2 ? 2 : ShapeClass<TP>::npq;      // passes of LW / 4 contacts: scenes of <= 4 actors keep <= 32 contacts (two robots alone <= 16), flocks <= 64
    const int q = lane & 3;
    float rowA[NPQ][9], usq[NPQ], dqq[NPQ], dqn[NPQ], cbq[NPQ], muq[NPQ], sdq[NPQ];
    int infq[NPQ];
#pragma unroll
    for (int ps = 0; ps < NPQ; ps++) {
      // (what a pass leaves behind is read further down under the pass's own condition: nothing is initialised outside it -- a skipped pass, the
      // second one of a two-robot scene, then costs nothing; rowA / fan / infq / cbq / sdq are written by side A of every contact before they are read)
      const int c = ps * (LW / 4) + (lane >> 2);
      if (ps * (LW / 4) < nc && c < nc) {
        usq[ps] = 0.0f; dqq[ps] = 0.0f; dqn[ps] = 0.0f; muq[ps] = HF(HOT_FRICTION);
        const float* cr = lds + L.con + c * CON_STRIDE;
        const float4 w0 = reinterpret_cast<const float4*>(cr)[0], w1 = reinterpret_cast<const float4*>(cr)[1], w2 = reinterpret_cast<const float4*>(cr)[2];
        const int cA = __float_as_int(w0.x), cB = __float_as_int(w0.z);
        if (cA < A || (cB >= 0 && cB < A)) muq[ps] = mu_robot;
        const int bodyA = __float_as_int(w0.y), bodyB = __float_as_int(w0.w);
        const V3 p = v3(w1.x, w1.y, w1.z), n = v3(w2.x, w2.y, w2.z);
        V3 t1, t2;
        contact_tangents(n, t1, t2);
        const V3 dir = q == 0 ? n : (q == 1 ? t1 : t2);
        const float sd = w1.w;
        cbq[ps] = first_bias(sd);
        sdq[ps] = sd;
        float fan[9];                            // side A's NEXT row (two sides on one actor: cross terms)
        int lgA_ = 0;
        for (int side = 0; side < 2; side++) {
          const int act = side == 0 ? cA : cB, body = side == 0 ? bodyA : bodyB;
          const float sg = side == 0 ? 1.0f : -1.0f;
          if (act < 0) continue;
          float fq[9];                           // this lane's row of the record: U[q][0..5], Z'[q][0..2]
#pragma unroll
          for (int i = 0; i < 9; i++) fq[i] = 0.0f;
          int legi = -1, ncol = 6, wbase = 0;
          float uq = 0.0f;
          if (act < A) {
            wbase = act * MQE_RD;
            const float* brec = lds + L.body + act * MQE_NBODY * BODY_STRIDE;
            const V3 r0 = p - ld3(brec + B_P);
            const float* vb = Vm + act * MQE_RD;
            const V3 cq = cross(r0, dir);
            float Wq[6] = {sg * dir.x, sg * dir.y, sg * dir.z, sg * cq.x, sg * cq.y, sg * cq.z}, Zq[3];
            const int leg = body > 0 ? (body - 1) / 3 : 0, dep = body > 0 ? (body - 1) % 3 + 1 : 0;
#pragma unroll
            for (int t = 0; t < 3; t++) {
              const float* jrec = brec + (1 + leg * 3 + t) * BODY_STRIDE;
              const float4 jq2 = reinterpret_cast<const float4*>(jrec)[2], jq3 = reinterpret_cast<const float4*>(jrec)[3];
              const V3 ax = v3(jq3.x, jq3.y, jq3.z), rr = p - v3(jq2.y, jq2.z, jq2.w);
              Zq[t] = (t < dep ? sg : 0.0f) * dot(cross(rr, dir), ax);
            }
            {                                    // relative velocity of the unconstrained motion through the FULL Jacobian row
              const float* vl = vb + 6 + leg * 3;
#pragma unroll
              for (int mm = 0; mm < 6; mm++) uq += Wq[mm] * vb[mm];
              uq = uq + Zq[0] * vl[0] + Zq[1] * vl[1] + Zq[2] * vl[2];
            }
            if (dep > 0) {                       // reduce onto the base coordinates: W = J_b - J_l G^T, Z' = J_l Lm
              legi = leg;
              const float* lrec = lds + L.leg + (act * 4 + leg) * LEG_STRIDE;
#pragma unroll
              for (int mm = 0; mm < 6; mm++) Wq[mm] -= Zq[0] * lrec[LEG_G + mm * 3] + Zq[1] * lrec[LEG_G + mm * 3 + 1] + Zq[2] * lrec[LEG_G + mm * 3 + 2];
              const float4 la = reinterpret_cast<const float4*>(lrec)[1], lb = reinterpret_cast<const float4*>(lrec)[2];   // .., l00, l10 | l11, l20, l21, l22
              fq[6] = Zq[0] * la.z + Zq[1] * la.w + Zq[2] * lb.y;
              fq[7] = Zq[1] * lb.x + Zq[2] * lb.z;
              fq[8] = Zq[2] * lb.w;
            }
            const float* Fm = lds + L.sinv + act * 72 + 36;   // U = W F, F upper triangular
#pragma unroll
            for (int nn = 0; nn < 6; nn++)
#pragma unroll
              for (int mm = nn; mm < 6; mm++) fq[mm] += Wq[nn] * Fm[nn * 6 + mm];
          } else if (SS) {
            wbase = A * MQE_RD; ncol = 1;
            const V3 r0 = p - ssPiv;
            const V3 wy = m->ss_axis == 3 ? v3(0, 1, 0) : cross(m->ss_axis == 2 ? v3(0, 0, 1) : v3(0, 1, 0), r0);   // prismatic: the axis itself
            const float jq = sg * dot(dir, wy);
            fq[0] = sqrtf(1.0f / m->ss_inertia) * jq;
            uq = jq * Vm[A * MQE_RD];
          } else {
            const int pi = act - A;
            wbase = A * MQE_RD + pi * npcdof; ncol = npcdof;
            const V3 r0 = p - ld3(lds + L.root + (A + pi) * 13);
            const float sm = sqrtf(1.0f / m->npc_mass), si = sqrtf(1.0f / m->npc_inertia);
            const float* vn = Vm + wbase;
            const V3 cq = cross(r0, dir);
            const float jr[6] = {sg * dir.x, sg * dir.y, sg * dir.z, sg * cq.x, sg * cq.y, sg * cq.z};
#pragma unroll
            for (int mm = 0; mm < 6; mm++) {
              const bool on = mm < npcdof;
              fq[mm] = on ? (mm < 3 ? sm : si) * jr[mm] : 0.0f;
              if (on) uq += jr[mm] * vn[mm];
            }
          }
          usq[ps] += uq;
          // the contact's own block: |row|^2 and the coupling with the next row (0 -> 1, 1 -> 2, 2 -> 0) fetched from the neighbour lane
          float fn[9];
#pragma unroll
          for (int i = 0; i < 9; i++) fn[i] = dpp_take<0xC9>(fq[i]);          // quad_perm [1,2,0,3]
          float sqq = 0.0f, sqn = 0.0f;
#pragma unroll
          for (int i = 0; i < 9; i++) { sqq += fq[i] * fq[i]; sqn += fq[i] * fn[i]; }
          dqq[ps] += sqq; dqn[ps] += sqn;
          const int info = (legi > 0 ? legi * 3 : 0) | ((ncol == 6 && act < A ? 9 : ncol) << 8) | (wbase << 16);   // first joint offset | lanes | first coordinate
          if (side == 0) {
            infq[ps] = info; lgA_ = legi + 1;
#pragma unroll
            for (int i = 0; i < 9; i++) { rowA[ps][i] = fq[i]; fan[i] = fn[i]; }
          } else {                               // side B (two-actor contacts only): straight into its slot
            float* rec = lds + L.side + (c - nc_terr) * RSB_STRIDE;
            if (q < 3) {
#pragma unroll
              for (int i = 0; i < 9; i++) rec[i * 3 + q] = fq[i];
              rec[27 + q] = 0.0f;                // slot 9 (side A's carries u* - bias): lane 9 of side B's row multiplies it by zero
            }
            if (q == 0) rec[RSB_INFO] = __int_as_float(info);
            if (cB == cA) {                      // both sides on ONE actor (two links of a robot): the sides share coordinates -> cross terms
              const bool same_leg = lgA_ != 0 && lgA_ == legi + 1;
              float xqq = 0.0f, xqn = 0.0f, xnq = 0.0f;      // A_q . B_q,  A_q . B_next,  A_next . B_q
#pragma unroll
              for (int mm = 0; mm < 6; mm++) { xqq += rowA[ps][mm] * fq[mm]; xqn += rowA[ps][mm] * fn[mm]; xnq += fan[mm] * fq[mm]; }
              if (same_leg) {
#pragma unroll
                for (int i = 6; i < 9; i++) { xqq += rowA[ps][i] * fq[i]; xqn += rowA[ps][i] * fn[i]; xnq += fan[i] * fq[i]; }
              }
              dqq[ps] += 2.0f * xqq; dqn[ps] += xqn + xnq;
            }
          }
        }
      }
    }
    __syncthreads();                             // the link records are dead from here: side A and the solve records go on top of them
#pragma unroll
    for (int ps = 0; ps < NPQ; ps++) {
      const int c = ps * (LW / 4) + (lane >> 2);
      if (ps * (LW / 4) < nc && c < nc) {
        float* rc = lds + L.phi + c * RS_STRIDE;
        if (q < 3) {
#pragma unroll
          for (int i = 0; i < 9; i++) rc[RS_SLOT + i * 3 + q] = rowA[ps][i];
          rc[RS_SLOT + 27 + q] = q == 0 ? usq[ps] - cbq[ps] : usq[ps];
          rc[1 + q] = 1.0f / dqq[ps];
          rc[q == 0 ? 4 : (q == 1 ? 6 : 5)] = dqn[ps];           // d10 = row 0 . row 1, d21 = row 1 . row 2, d20 = row 2 . row 0
          if (q == 0) { rc[RS_USN] = usq[ps]; rc[RS_BIAS] = cbq[ps]; }
        } else {
          reinterpret_cast<float4*>(rc)[2] = make_float4(0.0f, 0.0f, 0.0f, sdq[ps]);      // lambda starts at zero; the running separation (temporal solver)
          rc[0] = muq[ps]; rc[7] = __int_as_float(infq[ps]);
        }
      }
    }
    if (is_con) {                                // the sweep groups the contacts by actor: lane = contact again
      const float4 w0 = reinterpret_cast<const float4*>(lds + L.con + lane * CON_STRIDE)[0];
      myA = __float_as_int(w0.x); myB = __float_as_int(w0.z);
    }
  } else if (is_con) {
    float* cr = lds + L.con + lane * CON_STRIDE;
    const float4 w0 = reinterpret_cast<const float4*>(cr)[0], w1 = reinterpret_cast<const float4*>(cr)[1], w2 = reinterpret_cast<const float4*>(cr)[2];
    myA = __float_as_int(w0.x); myB = __float_as_int(w0.z);
    if (myA < A || (myB >= 0 && myB < A)) mu = mu_robot;
    const int bodyA = __float_as_int(w0.y), bodyB = __float_as_int(w0.w);
    const V3 p = v3(w1.x, w1.y, w1.z), n = v3(w2.x, w2.y, w2.z);
    V3 t1, t2;
    contact_tangents(n, t1, t2);
    const float sd = w1.w;
    cbias = first_bias(sd);
    csep = sd;
    for (int side = 0; side < 2; side++) {
      const int act = side == 0 ? myA : myB, body = side == 0 ? bodyA : bodyB;
      const float sg = side == 0 ? 1.0f : -1.0f;
      if (act < 0) continue;
      float W[3][6], Z[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
      float f[SIDE_STRIDE];                                   // the record: U, Z', info
#pragma unroll
      for (int i = 0; i < SIDE_STRIDE; i++) f[i] = 0.0f;
      int legi = -1, ncol = 6, wbase = 0;
      const V3 dirs[3] = {n, t1, t2};
      if (act < A) {
        wbase = act * MQE_RD;
        const float* brec = lds + L.body + act * MQE_NBODY * BODY_STRIDE;
        const V3 r0 = p - ld3(brec + B_P);
        const float* vb = Vm + act * MQE_RD;
#pragma unroll
        for (int q = 0; q < 3; q++) {
          const V3 cq = cross(r0, dirs[q]);
          W[q][0] = sg * dirs[q].x; W[q][1] = sg * dirs[q].y; W[q][2] = sg * dirs[q].z;
          W[q][3] = sg * cq.x; W[q][4] = sg * cq.y; W[q][5] = sg * cq.z;
        }
        const int leg = body > 0 ? (body - 1) / 3 : 0, dep = body > 0 ? (body - 1) % 3 + 1 : 0;
#pragma unroll
        for (int t = 0; t < 3; t++) {
          const float* jrec = brec + (1 + leg * 3 + t) * BODY_STRIDE;
          const float4 jq2 = reinterpret_cast<const float4*>(jrec)[2], jq3 = reinterpret_cast<const float4*>(jrec)[3];
          const V3 ax = v3(jq3.x, jq3.y, jq3.z), rr = p - v3(jq2.y, jq2.z, jq2.w);
          const float on = t < dep ? sg : 0.0f;
#pragma unroll
          for (int q = 0; q < 3; q++) Z[q][t] = on * dot(cross(rr, dirs[q]), ax);
        }
        // relative velocity of the unconstrained motion through the FULL Jacobian row: base columns, then the chain's joints
        {
          const float* vl = vb + 6 + leg * 3;
          float uq[3];
#pragma unroll
          for (int q = 0; q < 3; q++) {
            float acc = 0.0f;
#pragma unroll
            for (int mm = 0; mm < 6; mm++) acc += W[q][mm] * vb[mm];
            uq[q] = acc + Z[q][0] * vl[0] + Z[q][1] * vl[1] + Z[q][2] * vl[2];
          }
          us0 += uq[0]; us1 += uq[1]; us2 += uq[2];
        }
        if (dep > 0) {                       // reduce onto the base coordinates: W = J_b - J_l G^T (five 16 B loads of G), Z' = J_l Lm
          legi = leg;
          const float* lrec = lds + L.leg + (act * 4 + leg) * LEG_STRIDE;
#pragma unroll
          for (int mm = 0; mm < 6; mm++) {   // G streamed row by row (3 floats): few registers live at a time
            const float g0 = lrec[LEG_G + mm * 3], g1 = lrec[LEG_G + mm * 3 + 1], g2 = lrec[LEG_G + mm * 3 + 2];
#pragma unroll
            for (int q = 0; q < 3; q++) W[q][mm] -= Z[q][0] * g0 + Z[q][1] * g1 + Z[q][2] * g2;
          }
          const float4 la = reinterpret_cast<const float4*>(lrec)[1], lb = reinterpret_cast<const float4*>(lrec)[2];   // .., l00, l10 | l11, l20, l21, l22
#pragma unroll
          for (int q = 0; q < 3; q++) {
            f[SIDE_Z + q * 3 + 0] = Z[q][0] * la.z + Z[q][1] * la.w + Z[q][2] * lb.y;
            f[SIDE_Z + q * 3 + 1] = Z[q][1] * lb.x + Z[q][2] * lb.z;
            f[SIDE_Z + q * 3 + 2] = Z[q][2] * lb.w;
          }
        }
        {                                    // U = W F (F upper triangular: nine 16 B loads)
          const float* Fm = lds + L.sinv + act * 72 + 36;
#pragma unroll
          for (int nn = 0; nn < 6; nn++) {   // F streamed row by row: row nn feeds the columns mm >= nn
            float fr[6];
#pragma unroll
            for (int mm = 0; mm < 6; mm++) fr[mm] = mm >= nn ? Fm[nn * 6 + mm] : 0.0f;
#pragma unroll
            for (int q = 0; q < 3; q++)
#pragma unroll
              for (int mm = nn; mm < 6; mm++) f[q * 6 + mm] += W[q][nn] * fr[mm];
          }
        }
      } else if (SS) {
        wbase = A * MQE_RD; ncol = 1;
        const V3 r0 = p - ssPiv;
        const V3 wy = m->ss_axis == 3 ? v3(0, 1, 0) : cross(m->ss_axis == 2 ? v3(0, 0, 1) : v3(0, 1, 0), r0);   // prismatic: the axis itself
        const float si = sqrtf(1.0f / m->ss_inertia), vv = Vm[A * MQE_RD];
#pragma unroll
        for (int q = 0; q < 3; q++) {
          const float jq = sg * dot(dirs[q], wy);
          f[q * 6] = si * jq;
          if (q == 0) us0 += jq * vv; else if (q == 1) us1 += jq * vv; else us2 += jq * vv;
        }
      } else {
        const int pi = act - A;
        wbase = A * MQE_RD + pi * npcdof; ncol = npcdof;
        const V3 r0 = p - ld3(lds + L.root + (A + pi) * 13);
        const float sm = sqrtf(1.0f / m->npc_mass), si = sqrtf(1.0f / m->npc_inertia);
        const float* vn = Vm + wbase;
#pragma unroll
        for (int q = 0; q < 3; q++) {
          const V3 cq = cross(r0, dirs[q]);
          const float jr[6] = {sg * dirs[q].x, sg * dirs[q].y, sg * dirs[q].z, sg * cq.x, sg * cq.y, sg * cq.z};
          float acc = 0.0f;
#pragma unroll
          for (int mm = 0; mm < 6; mm++) {
            const bool on = mm < npcdof;
            f[q * 6 + mm] = on ? (mm < 3 ? sm : si) * jr[mm] : 0.0f;
            if (on) acc += jr[mm] * vn[mm];
          }
          if (q == 0) us0 += acc; else if (q == 1) us1 += acc; else us2 += acc;
        }
      }
      // this side's share of the contact's own block
      {
        float s00 = 0, s10 = 0, s11 = 0, s20 = 0, s21 = 0, s22 = 0;
#pragma unroll
        for (int mm = 0; mm < 6; mm++) {
          s00 += f[mm] * f[mm]; s10 += f[6 + mm] * f[mm]; s11 += f[6 + mm] * f[6 + mm];
          s20 += f[12 + mm] * f[mm]; s21 += f[12 + mm] * f[6 + mm]; s22 += f[12 + mm] * f[12 + mm];
        }
#pragma unroll
        for (int i = 0; i < 3; i++) {
          s00 += f[SIDE_Z + i] * f[SIDE_Z + i]; s10 += f[SIDE_Z + 3 + i] * f[SIDE_Z + i]; s11 += f[SIDE_Z + 3 + i] * f[SIDE_Z + 3 + i];
          s20 += f[SIDE_Z + 6 + i] * f[SIDE_Z + i]; s21 += f[SIDE_Z + 6 + i] * f[SIDE_Z + 3 + i]; s22 += f[SIDE_Z + 6 + i] * f[SIDE_Z + 6 + i];
        }
        d00 += s00; d10 += s10; d11 += s11; d20 += s20; d21 += s21; d22 += s22;
      }
      const int info = (legi + 1) | (ncol << 4);           // the lane sweep's info word (the row sweep has its own: see above)
      if (side == 0) {                       // side A: registers for the whole sweep
        wA = wbase; infoA = info; lgA = legi + 1;
#pragma unroll
        for (int i = 0; i < 27; i++) fA[i] = f[i];
      } else {                               // side B (two-actor contacts only): its slot in LDS
        wB = wbase; infoB = info; lgB = legi + 1;
        f[SIDE_INFO] = __int_as_float(info);
        float4* sr = reinterpret_cast<float4*>(lds + L.side + (lane - nc_terr) * SIDE_STRIDE);
#pragma unroll
        for (int w = 0; w < SIDE_STRIDE / 4; w++) sr[w] = make_float4(f[4 * w], f[4 * w + 1], f[4 * w + 2], f[4 * w + 3]);
        if (myB == myA) {                    // both sides on ONE actor (two links of a robot): the sides share coordinates -> cross terms
          const bool same_leg = lgA != 0 && lgA == lgB;
          float x[3][3];
#pragma unroll
          for (int q = 0; q < 3; q++)
#pragma unroll
            for (int q2 = 0; q2 < 3; q2++) {
              float acc = 0.0f;
#pragma unroll
              for (int mm = 0; mm < 6; mm++) acc += fA[q * 6 + mm] * f[q2 * 6 + mm];
              if (same_leg) acc += fA[SIDE_Z + q * 3] * f[SIDE_Z + q2 * 3] + fA[SIDE_Z + q * 3 + 1] * f[SIDE_Z + q2 * 3 + 1] + fA[SIDE_Z + q * 3 + 2] * f[SIDE_Z + q2 * 3 + 2];
              x[q][q2] = acc;
            }
          d00 += 2.0f * x[0][0]; d10 += x[1][0] + x[0][1]; d11 += 2.0f * x[1][1];
          d20 += x[2][0] + x[0][2]; d21 += x[2][1] + x[1][2]; d22 += 2.0f * x[2][2];
        }
      }
    }
  }
  const float ik00 = is_con ? 1.0f / d00 : 0.0f, ik11 = is_con ? 1.0f / d11 : 0.0f, ik22 = is_con ? 1.0f / d22 : 0.0f;
  __syncthreads();
  TSTAMP(11);
  TSTAMP(12);
  // ---- projected Gauss-Seidel on  w = sum_c Phi_c^T lambda_c  ---------------------------------------------------------------
  // A step: the owning lane forms its relative velocity u = u* + Phi w (its actors' coordinates: 6 + 3 per robot side), solves its
  // three rows (normal >= 0, two friction rows boxed by mu lambda_n, the rows coupled through the contact's own block) and adds
  // Phi^T d(lambda) to w.  The one-sided contacts of one actor touch that actor's coordinates only, so the s-th contact of EVERY
  // actor is processed in the same step; contacts between two actors follow one by one.  Mathematically the sweep of the CPU
  // oracle (velocity space) and of the former coupling-block form (contact space); here w IS M^-1 J^T lambda in disguise: dv = T w.
  if (shp.rowgs) {
    // ---- row sweep (scenes of <= 4 actors): one DPP row of 16 lanes per ACTOR -------------------------------------------------
    // The same projected Gauss-Seidel on w = sum Phi^T lambda, but a step is spread over the lanes of a row instead of running on the
    // contact's one lane: lane k of the row holds coordinate k of the actor (6 base + the 3 joints of the touching leg), multiplies
    // Phi[:, k] by w[k], the three sums over the row come from four DPP butterfly steps each, every lane solves the three rows
    // (identical operands -> identical results) and updates its own w[k].  ~50 VALU instructions per step instead of ~90, and the
    // side records and per-contact constants sit in LDS (over the dead link records) instead of 45 registers of every lane.
    const bool is_terr = is_con && myB < 0, is_pair = is_con && myB >= 0;
    const int row = lane >> 4, k = lane & 15;
    int gstart = 0, glen = 0, maxlen = 0;                    // the one-sided contacts of my row's actor: first list index, count
    const int nact = A + PD + (SS ? 1 : 0);
    // Scenes of more than four actors (flocks, 2 vs 2 + ball; round 6 -- they ran the lane sweep before): the ROBOTS keep their rows, a free
    // NPC's one-sided contacts (<= 6 coordinates, no leg) are stepped by ONE lane each -- lane p = NPC p -- right behind the robots' step of
    // the same index: different actors, different coordinates of w.  Two-actor contacts take rows 0 and 1 as everywhere.
    const bool hyb = ShapeClass<TP>::npq == 4 && nact > 4;      // (the kernels of the 16-envs-per-CU class and go1football-defender's hold at most four actors: mqe_engine.hip pick_shape)
    int gstartN = 0, glenN = 0, maxlenN = 0;                 // hybrid: the one-sided contacts of NPC `lane`
    for (int a = 0; a < nact; a++) {
      const unsigned long long bm = gballot(is_terr && myA == a);
      const int len = __popcll(bm), start = bm ? __ffsll((long long)bm) - 1 : 0;
      if (!hyb || a < A) {
        maxlen = len > maxlen ? len : maxlen;
        if (row == a) { gstart = start; glen = len; }
      } else {
        maxlenN = len > maxlenN ? len : maxlenN;
        if (lane == a - A) { gstartN = start; glenN = len; }
      }
    }
    const int nrow_act = hyb ? A : nact;                     // actors that own a row
    const int npair = __popcll(gballot(is_pair));
    const int pair0 = nc - npair;
    // lane k of a row reads slot k of a record (three floats: its column of Phi for the three rows of the contact); lanes 6-8 own the joints
    // of the contact's leg (first-joint offset in the info word), lane 9 the u* - bias slot, the lanes beyond multiply slot 0 by zero
    const int k3 = (k < 10 ? k : 0) * 3;
    const int legm = (k >= 6 && k < 9) ? 63 : 0;
    const float wdef = k == 9 ? 1.0f : 0.0f;
    __syncthreads();
    // sums over the row of slot[q] * w -- three independent butterflies, interleaved; every lane of the row ends with bitwise the same sums
    auto row_sums = [&](float ph0, float ph1, float ph2, float wk, float& s0, float& s1, float& s2) {
      float a0 = ph0 * wk, a1 = ph1 * wk, a2 = ph2 * wk;
      asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2));    // products stay products: contracted into the first step they cost mov_dpp + fma each
      a0 += dpp_take<0xB1>(a0); a1 += dpp_take<0xB1>(a1); a2 += dpp_take<0xB1>(a2);
      a0 += dpp_take<0x4E>(a0); a1 += dpp_take<0x4E>(a1); a2 += dpp_take<0x4E>(a2);
      a0 += dpp_take<0x141>(a0); a1 += dpp_take<0x141>(a1); a2 += dpp_take<0x141>(a2);
      a0 += dpp_take<0x140>(a0); a1 += dpp_take<0x140>(a1); a2 += dpp_take<0x140>(a2);
      s0 = a0; s1 = a1; s2 = a2;
    };
    // the contact's three rows from the record's constants (q1: mu, 1 / d; q2: couplings; q3: lambda) and the sums u - bias (normal row),
    // u (tangents): the new impulses and their increments
    auto row_solve = [&](const float4 q1, const float4 q2, const float4 q3, float u0, float u1, float u2, float& ln, float& l1, float& l2, float& e0, float& e1, float& e2) {
      ln = fmaxf(q3.x - u0 * q1.y, 0.0f);
      e0 = ln - q3.x;
      const float lim = q1.x * ln;
      l1 = __builtin_amdgcn_fmed3f(q3.y - (u1 + q2.x * e0) * q1.z, -lim, lim);      // lim >= 0: the median IS the clamp, one instruction
      e1 = l1 - q3.y;
      l2 = __builtin_amdgcn_fmed3f(q3.z - (u2 + q2.y * e0 + q2.z * e1) * q1.w, -lim, lim);
      e2 = l2 - q3.z;
    };
    // temporal solver: what the contact's own lane keeps for the separation updates between the sweeps -- side A's normal row, u*_n, the
    // running separation, the bias (side B of a two-actor contact is re-read from its slot: rare)
    float nrow[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, usn = 0.0f, cbias_r = 0.0f;
    const float* nwb = accv; const float* nwl = accv;
    float* myrc = lds + L.phi + (is_con ? lane : 0) * RS_STRIDE;
    if (is_con) {
      const int info = __float_as_int(myrc[7]);
      if (tgs) {
#pragma unroll
        for (int i = 0; i < 9; i++) nrow[i] = myrc[RS_SLOT + i * 3];          // columns beyond the side's are stored as zeros
        nwb = accv + (info >> 16);
        nwl = nwb + (((info >> 8) & 15) == 9 ? 6 + (info & 63) : 0);          // (no leg columns: three zero weights on the base's first coordinates)
      }
      usn = myrc[RS_USN]; cbias_r = myrc[RS_BIAS]; csep = myrc[11];
    }
    // (barriers inside: with two envs per wavefront the trip counts are the larger of the two envs'; a row without work idles)
    const int maxlen_w = wave_max_of_groups(maxlen > maxlenN ? maxlen : maxlenN), npair_w = wave_max_of_groups(npair);
    const int rowk = row * MQE_RD + k;
    for (int it = 0; it < nsweeps; it++) {
      for (int sidx = 0; sidx < maxlen_w; sidx++) {
        if (hyb && lane < PD && sidx < glenN) {              // hybrid: the s-th one-sided contact of NPC `lane`, the whole step on this lane
          float* rc = lds + L.phi + (gstartN + sidx) * RS_STRIDE;
          const float4 q1 = reinterpret_cast<const float4*>(rc)[0], q2 = reinterpret_cast<const float4*>(rc)[1], q3 = reinterpret_cast<const float4*>(rc)[2];
          float* wb = accv + A * MQE_RD + lane * npcdof;
          float u0 = rc[RS_SLOT + 27], u1 = rc[RS_SLOT + 28], u2 = rc[RS_SLOT + 29];       // u* - bias, u*, u*
          float ln, l1, l2, e0, e1, e2;
          if (npcdof == 3) {                                 // wave-uniform: ball, sheep (translation only) -- nine record floats, three of w
            float wv[3], ph[9];
#pragma unroll
            for (int i = 0; i < 9; i++) ph[i] = rc[RS_SLOT + i];
#pragma unroll
            for (int mm = 0; mm < 3; mm++) {
              wv[mm] = wb[mm];
              u0 += ph[3 * mm] * wv[mm]; u1 += ph[3 * mm + 1] * wv[mm]; u2 += ph[3 * mm + 2] * wv[mm];
            }
            row_solve(q1, q2, q3, u0, u1, u2, ln, l1, l2, e0, e1, e2);
            reinterpret_cast<float4*>(rc)[2] = make_float4(ln, l1, l2, q3.w);
#pragma unroll
            for (int mm = 0; mm < 3; mm++) wb[mm] = wv[mm] + ph[3 * mm] * e0 + ph[3 * mm + 1] * e1 + ph[3 * mm + 2] * e2;
          } else {
            float wv[6], ph[18];
#pragma unroll
            for (int mm = 0; mm < 6; mm++) {
              const bool on = mm < npcdof;
              wv[mm] = on ? wb[on ? mm : 0] : 0.0f;
              ph[3 * mm] = rc[RS_SLOT + 3 * mm]; ph[3 * mm + 1] = rc[RS_SLOT + 3 * mm + 1]; ph[3 * mm + 2] = rc[RS_SLOT + 3 * mm + 2];      // (columns beyond the body's are stored as zeros)
              u0 += ph[3 * mm] * wv[mm]; u1 += ph[3 * mm + 1] * wv[mm]; u2 += ph[3 * mm + 2] * wv[mm];
            }
            row_solve(q1, q2, q3, u0, u1, u2, ln, l1, l2, e0, e1, e2);
            reinterpret_cast<float4*>(rc)[2] = make_float4(ln, l1, l2, q3.w);
#pragma unroll
            for (int mm = 0; mm < 6; mm++)
              if (mm < npcdof) wb[mm] = wv[mm] + ph[3 * mm] * e0 + ph[3 * mm + 1] * e1 + ph[3 * mm + 2] * e2;
          }
        }
        if (row < nrow_act && sidx < glen) {                 // the s-th one-sided contact of every actor that owns a row
          float* rc = lds + L.phi + (gstart + sidx) * RS_STRIDE;
          const float4 q1 = reinterpret_cast<const float4*>(rc)[0], q2 = reinterpret_cast<const float4*>(rc)[1], q3 = reinterpret_cast<const float4*>(rc)[2];
          const int info = __float_as_int(q2.w);
          // a robot's row (scenes whose actors 0 .. A-1 are robots): 9 coordinates, the first at row * 18 -- only the leg offset comes from the contact
          const bool on = TP == 0 ? k < 9 : k < ((info >> 8) & 15);
          const int widx = (TP == 0 ? rowk : (info >> 16) + k) + (info & legm);
          const float ww = accv[TP == 0 ? widx : (on ? widx : 0)];            // (a robot row's idle lanes read a neighbouring coordinate and drop it)
          const float ph0 = rc[RS_SLOT + k3], ph1 = rc[RS_SLOT + k3 + 1], ph2 = rc[RS_SLOT + k3 + 2];
          const float wk = on ? ww : wdef;
          float s0, s1, s2, ln, l1, l2, e0, e1, e2;
          row_sums(ph0, ph1, ph2, wk, s0, s1, s2);
          row_solve(q1, q2, q3, s0, s1, s2, ln, l1, l2, e0, e1, e2);
          if (k == 0) reinterpret_cast<float4*>(rc)[2] = make_float4(ln, l1, l2, q3.w);
          if (on) accv[widx] = wk + ph0 * e0 + ph1 * e1 + ph2 * e2;
        }
        __syncthreads();
      }
      for (int ip = 0; ip < npair_w; ip++) {                 // two-actor contacts one by one: row 0 = side A, row 1 = side B
        const int c = pair0 + ip;
        float e0 = 0.0f, e1 = 0.0f, e2 = 0.0f, fb0 = 0.0f, fb1 = 0.0f, fb2 = 0.0f;
        int widxB = 0;
        bool onB = false;
        if (row < 2 && c < nc) {
          float* rc = lds + L.phi + c * RS_STRIDE;
          const float* rb = lds + L.side + (c - nc_terr) * RSB_STRIDE;
          const float4 q1 = reinterpret_cast<const float4*>(rc)[0], q2 = reinterpret_cast<const float4*>(rc)[1], q3 = reinterpret_cast<const float4*>(rc)[2];
          const int info = row == 0 ? __float_as_int(q2.w) : __float_as_int(rb[RSB_INFO]);
          const float* slot = (row == 0 ? rc + RS_SLOT : rb) + k3;
          const bool on = k < ((info >> 8) & 15);
          const int widx = (info >> 16) + k + (info & legm);
          const float ww = accv[on ? widx : 0];
          const float ph0 = slot[0], ph1 = slot[1], ph2 = slot[2];
          const float wk = on ? ww : (row == 0 ? wdef : 0.0f);               // u* - bias enters once, through side A's row
          float s0, s1, s2, ln, l1, l2;
          row_sums(ph0, ph1, ph2, wk, s0, s1, s2);
          s0 += row_partner(s0); s1 += row_partner(s1); s2 += row_partner(s2);
          row_solve(q1, q2, q3, s0, s1, s2, ln, l1, l2, e0, e1, e2);          // both rows solve the same numbers; one lane records lambda
          if (lane == 0) reinterpret_cast<float4*>(rc)[2] = make_float4(ln, l1, l2, q3.w);
          if (row == 0 && on) accv[widx] = wk + ph0 * e0 + ph1 * e1 + ph2 * e2;
          onB = row == 1 && on; widxB = widx; fb0 = ph0; fb1 = ph1; fb2 = ph2;
        }
        __syncthreads();
        if (onB) accv[widxB] += fb0 * e0 + fb1 * e1 + fb2 * e2;                 // after side A's stores: the sides may share coordinates (self-contact)
        __syncthreads();
      }
      if (tgs && it < npos) { Wacc0 += lane < ndof ? accv[lane] : 0.0f; Wacc1 += lane + LW < ndof ? accv[lane + LW] : 0.0f; }    // (every step above ended with a barrier)
      if (it + 1 < nsweeps && (tgs || it + 1 >= npos)) {     // what the next sweep asks of the normal rows, lane = contact
        if (is_con) {
          float g = 0.0f;
          if (tgs && it < npos) {                            // Phi_n . w: side A from the lane's registers
            g = nrow[0] * nwb[0] + nrow[1] * nwb[1] + nrow[2] * nwb[2] + nrow[3] * nwb[3] + nrow[4] * nwb[4] + nrow[5] * nwb[5]
              + nrow[6] * nwl[0] + nrow[7] * nwl[1] + nrow[8] * nwl[2];
            if (is_pair) {
              const float* rec = lds + L.side + (lane - nc_terr) * RSB_STRIDE;
              const int info = __float_as_int(rec[RSB_INFO]);
              const int ncl = (info >> 8) & 15, jo = info & 63;
              const float* wb = accv + (info >> 16);
#pragma unroll
              for (int mm = 0; mm < 6; mm++) g += rec[mm * 3] * (mm < ncl ? wb[mm] : 0.0f);
              if (ncl == 9) g += rec[18] * wb[6 + jo] + rec[21] * wb[7 + jo] + rec[24] * wb[8 + jo];
            }
          }
          cbias_r = tgs ? next_bias(csep, usn + g, 0.0f, it) : next_bias(csep, 0.0f, cbias_r, it);
          myrc[RS_SLOT + 27] = usn - cbias_r;
        }
        __syncthreads();
      }
    }
    if (tgs) { if (lane < ndof) waccv[lane] = Wacc0; if (lane + LW < ndof) waccv[lane + LW] = Wacc1; }     // W = the sum of w over the position iterations
    if (is_con) { const float4 q3 = reinterpret_cast<const float4*>(myrc)[2]; cl0 = q3.x; cl1 = q3.y; cl2 = q3.z; }
  } else
    {
      const bool is_terr = is_con && myB < 0, is_pair = is_con && myB >= 0;
      int gstartA = 0, glenA = 0, maxlen = 0;
      const int nact = A + PD + (SS ? 1 : 0);
      for (int a = 0; a < nact; a++) {
        const unsigned long long bm = gballot(is_terr && myA == a);
        const int len = __popcll(bm), start = bm ? __ffsll((long long)bm) - 1 : 0;
        maxlen = len > maxlen ? len : maxlen;
        if (is_terr && myA == a) { gstartA = start; glenA = len; }
      }
      const int npair = __popcll(gballot(is_pair));
      const int pair0 = nc - npair;
      const int ncolA = (infoA >> 4) & 15, legA = (infoA & 15) - 1;
      float* wbA = accv + wA;
      float* wlA = accv + wA + 6 + (legA > 0 ? legA : 0) * 3;
      auto gs_update = [&]() {
        // side A from registers; its actor's coordinates are read once and written once
        float wv[9];
  #pragma unroll
        for (int mm = 0; mm < 6; mm++) wv[mm] = mm < ncolA ? wbA[mm] : 0.0f;
  #pragma unroll
        for (int i = 0; i < 3; i++) wv[6 + i] = legA >= 0 ? wlA[i] : 0.0f;
        float u0 = us0, u1 = us1, u2 = us2;
  #pragma unroll
        for (int mm = 0; mm < 6; mm++) { u0 += fA[mm] * wv[mm]; u1 += fA[6 + mm] * wv[mm]; u2 += fA[12 + mm] * wv[mm]; }
  #pragma unroll
        for (int i = 0; i < 3; i++) { u0 += fA[SIDE_Z + i] * wv[6 + i]; u1 += fA[SIDE_Z + 3 + i] * wv[6 + i]; u2 += fA[SIDE_Z + 6 + i] * wv[6 + i]; }
        float fb[SIDE_STRIDE];
        int ncolB = 0, legB = -1;
        if (is_pair) {                         // side B: record from its LDS slot, coordinates of the second actor
          const float4* r4 = reinterpret_cast<const float4*>(lds + L.side + (lane - nc_terr) * SIDE_STRIDE);
  #pragma unroll
          for (int w = 0; w < SIDE_STRIDE / 4; w++) { const float4 t = r4[w]; fb[4 * w] = t.x; fb[4 * w + 1] = t.y; fb[4 * w + 2] = t.z; fb[4 * w + 3] = t.w; }
          ncolB = (infoB >> 4) & 15; legB = (infoB & 15) - 1;
          const float* wb = accv + wB;
          const float* wl = accv + wB + 6 + (legB > 0 ? legB : 0) * 3;
  #pragma unroll
          for (int mm = 0; mm < 6; mm++)
            if (mm < ncolB) { const float x = wb[mm]; u0 += fb[mm] * x; u1 += fb[6 + mm] * x; u2 += fb[12 + mm] * x; }
          if (legB >= 0) {
  #pragma unroll
            for (int i = 0; i < 3; i++) { const float x = wl[i]; u0 += fb[SIDE_Z + i] * x; u1 += fb[SIDE_Z + 3 + i] * x; u2 += fb[SIDE_Z + 6 + i] * x; }
          }
        }
        const float ln = fmaxf(cl0 - (u0 - cbias) * ik00, 0.0f);
        const float e0 = ln - cl0;
        const float lim = mu * ln;
        const float l1 = __builtin_amdgcn_fmed3f(cl1 - (u1 + d10 * e0) * ik11, -lim, lim);
        const float e1 = l1 - cl1;
        const float l2 = __builtin_amdgcn_fmed3f(cl2 - (u2 + d20 * e0 + d21 * e1) * ik22, -lim, lim);
        const float e2 = l2 - cl2;
        cl0 = ln; cl1 = l1; cl2 = l2;
  #pragma unroll
        for (int mm = 0; mm < 6; mm++)
          if (mm < ncolA) wbA[mm] = wv[mm] + fA[mm] * e0 + fA[6 + mm] * e1 + fA[12 + mm] * e2;
        if (legA >= 0) {
  #pragma unroll
          for (int i = 0; i < 3; i++) wlA[i] = wv[6 + i] + fA[SIDE_Z + i] * e0 + fA[SIDE_Z + 3 + i] * e1 + fA[SIDE_Z + 6 + i] * e2;
        }
        if (is_pair) {                         // after side A's stores: the two sides may share coordinates (self-contact)
          float* wb = accv + wB;
          float* wl = accv + wB + 6 + (legB > 0 ? legB : 0) * 3;
  #pragma unroll
          for (int mm = 0; mm < 6; mm++)
            if (mm < ncolB) wb[mm] += fb[mm] * e0 + fb[6 + mm] * e1 + fb[12 + mm] * e2;
          if (legB >= 0) {
  #pragma unroll
            for (int i = 0; i < 3; i++) wl[i] += fb[SIDE_Z + i] * e0 + fb[SIDE_Z + 3 + i] * e1 + fb[SIDE_Z + 6 + i] * e2;
          }
        }
      };
      for (int it = 0; it < nsweeps; it++) {
        for (int sidx = 0; sidx < maxlen; sidx++) {
          if (is_terr && sidx < glenA && lane == gstartA + sidx) gs_update();
          __syncthreads();
        }
        for (int c = pair0; c < nc; c++) {
          if (lane == c) gs_update();
          __syncthreads();
        }
        if (tgs && it < npos) { Wacc0 += lane < ndof ? accv[lane] : 0.0f; Wacc1 += lane + LW < ndof ? accv[lane + LW] : 0.0f; }
        if (it + 1 < nsweeps && (tgs || it + 1 >= npos) && is_con) {       // what the next sweep asks of my normal row (reads w only: no barrier)
          float g = 0.0f;
          if (tgs && it < npos) {
  #pragma unroll
            for (int mm = 0; mm < 6; mm++) g += fA[mm] * (mm < ncolA ? wbA[mm] : 0.0f);
            if (legA >= 0) g += fA[SIDE_Z] * wlA[0] + fA[SIDE_Z + 1] * wlA[1] + fA[SIDE_Z + 2] * wlA[2];
            if (is_pair) {
              const float* rb = lds + L.side + (lane - nc_terr) * SIDE_STRIDE;
              const int ncolB = (infoB >> 4) & 15, legB = (infoB & 15) - 1;
              const float* wb = accv + wB;
              const float* wl = accv + wB + 6 + (legB > 0 ? legB : 0) * 3;
  #pragma unroll
              for (int mm = 0; mm < 6; mm++) g += rb[mm] * (mm < ncolB ? wb[mm] : 0.0f);
              if (legB >= 0) g += rb[SIDE_Z] * wl[0] + rb[SIDE_Z + 1] * wl[1] + rb[SIDE_Z + 2] * wl[2];
            }
          }
          cbias = next_bias(csep, us0 + g, cbias, it);
        }
      }
      if (tgs) { if (lane < ndof) waccv[lane] = Wacc0; if (lane + LW < ndof) waccv[lane + LW] = Wacc1; }
    }
  TSTAMP(13);
  if (is_con && (flags & PS_WRITE_CF)) {        // the contact's force on its side A, for the per-body sums further down
    float* cr = lds + L.con + lane * CON_STRIDE;
    const V3 n = ld3(cr + C_N);
    V3 t1, t2;
    contact_tangents(n, t1, t2);
    const V3 f = (1.0f / dt) * (cl0 * n + cl1 * t1 + cl2 * t2);
    cr[C_F] = f.x; cr[C_F + 1] = f.y; cr[C_F + 2] = f.z;
  }
  // impulses -> velocities: dv = T w.  Base: F w_b; leg: Lm w_l - G^T (F w_b); free body / 1-dof link: M^-1/2 w
  // Temporal solver: the positions move with the ACCUMULATED motion dt (v* + T W / npos), not with dt times the final velocity
  // v* + T w: what they move with beyond it, voff = T (W / npos - w), goes through the same factors (W is overwritten with
  // W / npos - w first) and replaces w in L.acc afterwards (the joint-limit impulses below change the velocity and keep voff).
  if (tgs) {
    const float inpos = 1.0f / (float)npos;
    for (int i = lane; i < ndof; i += LW) waccv[i] = waccv[i] * inpos - accv[i];
    __syncthreads();
  }
  // both through one pass over the factors (the coefficient loads are shared): dv = T w, vo = T (W / npos - w)
  auto t_apply2 = [&](int d) __attribute__((always_inline)) -> float2 {
    const float* w1 = accv; const float* w2 = waccv;
    float dv = 0.0f, vo = 0.0f;
    if (d < A * MQE_RD) {
      const int r = d / MQE_RD, k = d - r * MQE_RD;
      const float* Fm = lds + L.sinv + r * 72 + 36;
      const float* wb = w1 + r * MQE_RD; const float* xb = w2 + r * MQE_RD;
      if (k < 6) {
#pragma unroll
        for (int mm = 0; mm < 6; mm++) { const float c = Fm[k * 6 + mm]; dv += c * wb[mm]; if (tgs) vo += c * xb[mm]; }
      } else {
        const int lg = (k - 6) / 3, i = (k - 6) - lg * 3;
        const float* rec = lds + L.leg + (r * 4 + lg) * LEG_STRIDE;
        const float* wl = wb + 6 + lg * 3; const float* xl = xb + 6 + lg * 3;
        const float* Lm = rec + LEG_LM + (i * (i + 1)) / 2;
        dv = Lm[0] * wl[0]; if (tgs) vo = Lm[0] * xl[0];
        if (i >= 1) { dv += Lm[1] * wl[1]; if (tgs) vo += Lm[1] * xl[1]; }
        if (i >= 2) { dv += Lm[2] * wl[2]; if (tgs) vo += Lm[2] * xl[2]; }
        const float* G = rec + LEG_G + i;
        // (F w_b is recomputed by every joint lane on purpose: computed once per robot and shared through the LDS it is 60 VALU instructions
        // less and one LDS round trip more, and the round trip is what costs -- A/B round 4: no gain at 4 wavefronts per SIMD, +2 % on the
        // scenes that run at 2)
#pragma unroll
        for (int nn = 0; nn < 6; nn++) {          // F is upper triangular (stored 6 x 6 with its zeros): row nn starts at column nn -- the 15 skipped terms were exact zeros
          float dvb = 0.0f, vob = 0.0f;
#pragma unroll
          for (int mm = nn; mm < 6; mm++) { const float c = Fm[nn * 6 + mm]; dvb += c * wb[mm]; if (tgs) vob += c * xb[mm]; }
          dv -= G[nn * 3] * dvb; if (tgs) vo -= G[nn * 3] * vob;
        }
      }
    } else {
      float c;
      if (shp.has_seesaw) c = sqrtf(1.0f / m->ss_inertia);
      else { const int q = d - A * MQE_RD, k = q - (q / npcdof) * npcdof; c = sqrtf(1.0f / (k < 3 ? m->npc_mass : m->npc_inertia)); }
      dv = c * w1[d]; if (tgs) vo = c * w2[d];
    }
    return make_float2(dv, vo);
  };
  float vo0 = 0.0f, vo1 = 0.0f;                 // voff of this lane's coordinates (lane, lane + LW)
  if (lane < ndof) { const float2 t = t_apply2(lane); Vm[lane] += t.x; vo0 = t.y; }
  if (lane + LW < ndof) { const float2 t = t_apply2(lane + LW); Vm[lane + LW] += t.x; vo1 = t.y; }
  __syncthreads();
  float* voff = accv;                           // w is consumed: its place holds voff from here (zero for the velocity-level solver)
  if (lane < ndof) voff[lane] = vo0;
  if (lane + LW < ndof) voff[lane + LW] = vo1;
  __syncthreads();
  // joint limits: one sequential pass in (robot, joint) order after the contact solve, only when some joint violates.  The bound
  // of a joint's speed is the tighter of its position stops ((limit - q) / dt) and the URDF velocity limit (go1.urdf:115,157,185;
  // PhysX maxJointVelocity); a violation is removed by an impulse along the joint coordinate, so momentum is exchanged with the
  // rest of the robot instead of disappearing.
  {
    auto jbound = [&](int j, float q, float vo, float& lo, float& hi) {      // vo: what the joint's position moves with beyond its velocity
      lo = (rm.dof_lower[j] - q) / dt - vo; hi = (rm.dof_upper[j] - q) / dt - vo;
      const float vl = rm.dof_vel_limit[j];
      if (vl > 0.0f) { lo = fmaxf(lo, -vl); hi = fminf(hi, vl); }
    };
    // Gauss-Seidel over the violated joints: an impulse on one joint changes its neighbours' speeds, so the pass is repeated while
    // something still violates (at most MQE_LIMIT_PASSES times; normally none or one runs), then the bound is enforced exactly
    for (int pass = 0; pass <= MQE_LIMIT_PASSES; pass++) {
      bool viol = false;
      for (int d = lane; d < A * 12; d += LW) {
        const int r = d / 12, j = d - r * 12;
        const float q = lds[L.dof + d * 2], v = Vm[r * MQE_RD + 6 + j];
        float lo, hi;
        jbound(j, q, voff[r * MQE_RD + 6 + j], lo, hi);
        viol = viol || v < lo || v > hi;
        if (pass == MQE_LIMIT_PASSES) Vm[r * MQE_RD + 6 + j] = fminf(fmaxf(v, lo), hi);     // residual of the last pass (~1e-3 of the violation)
      }
      if (__ballot(viol) == 0ull || pass == MQE_LIMIT_PASSES) break;       // wave-wide: the passes below hold barriers
      for (int r = 0; r < A; r++)
        for (int j = 0; j < 12; j++) {
          const float q = lds[L.dof + (r * 12 + j) * 2], vj = Vm[r * MQE_RD + 6 + j];
          float lo, hi;
          jbound(j, q, voff[r * MQE_RD + 6 + j], lo, hi);
          float vio = 0.0f;
          if (vj < lo) vio = lo - vj; else if (vj > hi) vio = hi - vj;
          // uniform within the env's lanes: impulse along e_j, dv = M^-1 e_j lambda, (M^-1)_jj lambda = vio.  The barriers need the
          // whole wavefront: with two envs per wavefront both halves go through when either has a violation
          const bool mine = vio != 0.0f;
          if (EPW == 1 ? mine : (__ballot(mine) != 0ull)) {
            const float lam = mine ? vio / minv_elem(r, 6 + j, 6 + j) : 0.0f;
            __syncthreads();
            if (mine && lane < MQE_RD) Vm[r * MQE_RD + lane] += minv_elem(r, 6 + j, lane) * lam;
            __syncthreads();
          }
        }
    }
    __syncthreads();
  }
  if (SS && lane == 0) {              // hinge: velocity limit, then the geometric end stops
    float v = Vm[A * MQE_RD];
    v = clampf(v, -m->ss_vel_limit, m->ss_vel_limit);
    const float vo = voff[A * MQE_RD];
    v = clampf(v, (m->ss_theta_lo - ssTheta) / dt - vo, (m->ss_theta_hi - ssTheta) / dt - vo);
    Vm[A * MQE_RD] = v;
  }
  TSTAMP(14);
  if (dbg.minv != nullptr) {
    for (int i = lane; i < MQE_RD * MQE_RD; i += LW) dbg.minv[i] = minv_elem(dbg.robot, i / MQE_RD, i % MQE_RD);   // assembled from the factors
    if (lane == 0) *dbg.nc = nc;
    for (int c = lane; c < nc; c += LW) {
      const float* cr = lds + L.con + c * CON_STRIDE;
      float* o8 = dbg.contacts + c * 8;
      o8[0] = (float)__float_as_int(cr[C_IDS]); o8[1] = (float)__float_as_int(cr[C_IDS + 1]);
      o8[2] = (float)__float_as_int(cr[C_IDS + 2]); o8[3] = (float)__float_as_int(cr[C_IDS + 3]);
      o8[4] = cr[C_SD]; o8[5] = cr[C_N]; o8[6] = cr[C_N + 1]; o8[7] = cr[C_N + 2];
    }
  }
  if (no_write) return;
  if (ovf && lane == 0 && evalid) st.overflow[e] += 1;          // MQE_T_CONTACT_OVERFLOW: this substep's list was truncated
  if (__ballot(red != 0 && evalid) != 0ull) {                   // MQE_T_CONTACT_REDUCED (second half of the same buffer): per env of the wavefront
    if (gballot(red != 0) != 0ull && lane == 0 && evalid) st.overflow[HI(HOT_N) + e] += 1;
  }

  // ---- net contact force per reported body (deterministic: contact order) ---------------------------------------------------
  if ((flags & PS_WRITE_CF) && evalid) {
    float* g_cf = st.cf + (size_t)e * HI(HOT_NBR) * 3;
    for (int rb = lane; rb < HI(HOT_NBR); rb += LW) {
      V3 F = v3(0, 0, 0);
      for (int c = 0; c < nc; c++) {
        const float* cr = lds + L.con + c * CON_STRIDE;
        const int ra = __float_as_int(cr[C_REPA]), rb2 = __float_as_int(cr[C_REPB]);
        if (ra == rb || rb2 == rb) {
          const V3 f = ld3(cr + C_F);
          F = (ra == rb) ? F + f : F - f;
        }
      }
      g_cf[rb * 3] = F.x; g_cf[rb * 3 + 1] = F.y; g_cf[rb * 3 + 2] = F.z;
    }
  }
  // ---- integrate: semi-implicit Euler; quaternion first-order update + renormalisation -----------------------------------------
  __syncthreads();
  for (int d = lane; d < A * 12; d += LW) {
    const int r = d / 12, j = d - r * 12;
    const float v = Vm[r * MQE_RD + 6 + j];
    float* ds = lds + L.dof + d * 2;
    ds[0] = ds[0] + dt * (v + voff[r * MQE_RD + 6 + j]);
    ds[1] = v;
  }
  if (SS && lane == 0) {
    const float v = Vm[A * MQE_RD];
    lds[L.dof + (12 * A) * 2] = ssTheta + dt * (v + voff[A * MQE_RD]);
    lds[L.dof + (12 * A) * 2 + 1] = v;
  }
  if (lane < A + PD) {
    const int act = lane;
    float* rs = lds + L.root + act * 13;
    const int vbase = act < A ? act * MQE_RD : A * MQE_RD + (act - A) * npcdof;
    const float* v = lds + L.rhs + vbase;
    const float* vo = voff + vbase;
    const bool has_ang = act < A || npcdof == 6;
    float wx = has_ang ? v[3] : rs[10], wy = has_ang ? v[4] : rs[11], wz = has_ang ? v[5] : rs[12];
    for (int k = 0; k < 3; k++) { rs[k] = rs[k] + dt * (v[k] + vo[k]); rs[7 + k] = v[k]; }
    if (has_ang) {
      rs[10] = wx; rs[11] = wy; rs[12] = wz;
      wx += vo[3]; wy += vo[4]; wz += vo[5];          // the orientation moves with the accumulated rotation
      float q0 = rs[3], q1 = rs[4], q2 = rs[5], q3 = rs[6];
      const float d0 = 0.5f * (wx * q3 + wy * q2 - wz * q1);
      const float d1 = 0.5f * (-wx * q2 + wy * q3 + wz * q0);
      const float d2 = 0.5f * (wx * q1 - wy * q0 + wz * q3);
      const float d3 = 0.5f * (-wx * q0 - wy * q1 - wz * q2);
      q0 += dt * d0; q1 += dt * d1; q2 += dt * d2; q3 += dt * d3;
      const float nq = sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
      rs[3] = q0 / nq; rs[4] = q1 / nq; rs[5] = q2 / nq; rs[6] = q3 / nq;
    }
  }
  __syncthreads();
  if ((flags & PS_STORE_STATE) && evalid) {       // coalesced write-back
    for (int i = lane; i < (A + P) * 13; i += LW) g_root[i] = lds[L.root + i];
    for (int i = lane; i < shp.ND * 2; i += LW) g_dof[i] = lds[L.dof + i];
  }
}

__global__ void __launch_bounds__(64) k_simulate(const DevModel* __restrict__ m, DevState st, int env_base, int no_write, PhysDebug dbg) {
  extern __shared__ float lds[];
  own_state(st);
  phys_substep<0, -1>(m, st, lds, env_base + blockIdx.x, threadIdx.x, PS_LOAD_STATE | PS_LOAD_TAU | PS_STORE_STATE | PS_WRITE_CF, no_write, dbg, (int)m->hot[threadIdx.x]);
}
// the same substep compiled for the two-robot, no-NPC shape (what k_substeps<2,0> runs): only the per-phase counter tool launches it
__global__ void __launch_bounds__(64) k_simulate_a2(const DevModel* __restrict__ m, DevState st, int env_base, int no_write, PhysDebug dbg) {
  extern __shared__ float lds[];
  phys_substep<2, 0>(m, st, lds, env_base + blockIdx.x, threadIdx.x, PS_LOAD_STATE | PS_LOAD_TAU | PS_STORE_STATE | PS_WRITE_CF, no_write, dbg, (int)m->hot[threadIdx.x]);
}

// ----------------------------------------------------------------------------------------------------------------------
// k_substeps: the whole decimation loop of Go1.step (go1.py:48-58) for one env in one wavefront:
//   nsub x { actuator-net torques on MFMA (32 joints per tile, see k_compute_torques_mfma) -> physics substep }
// Root/dof state is loaded once, stays in LDS across the substeps and is written back once; the actuator history
// (two past position errors and velocities per joint) lives in registers for the whole launch; torques go straight
// into the LDS slot the physics reads.  Control type "C" only (the reference's hierarchical controller).
typedef float f32x16_p __attribute__((ext_vector_type(16)));
// 1 + |x| >= 1: the hardware reciprocal (1 ulp) needs none of the range scaling a general division carries
__device__ __forceinline__ float softsign_p(float x) { return x * __builtin_amdgcn_rcpf(1.0f + fabsf(x)); }

#define ACT_TILES 2       // 2 x 32 joints >= 12 * MQE_MAX_AGENTS(=4)... agents <= 4 need 48 joints

// Occupancy class of a scene shape.  Scenes of two robots and at most one more object (link, ball, box, scenery) need < 10 KiB of LDS
// per env, so 16 waves fit a CU: they are compiled for 128 VGPRs (4 waves per SIMD) and 4096 envs run as ONE round of 16 waves per
// CU.  Larger scenes need more LDS than that allows and stay at 2 waves per SIMD with the full register file.  (Overrides for
// experiments: -DMQE_SUBSTEPS_WAVES=n, -DMQE_LAUNDER=0..3.)
template <int TP> struct SubstepsClass {
  static constexpr bool small = ShapeClass<TP>::small;
#ifdef MQE_SUBSTEPS_WAVES
  static constexpr int waves = MQE_SUBSTEPS_WAVES;
#else
  static constexpr int waves = small ? 4 : 2;
#endif
#ifdef MQE_LAUNDER
  static constexpr int launder = MQE_LAUNDER;
#else
  static constexpr int launder = 2;       // bit 0: the model pointer, bit 1: the lane id.  (The larger scenes at 2 waves per SIMD spilled
                                          // 12-44 registers with everything hoisted; laundered they need 136-154 and none: -4 % kernel time)
#endif
};
// the post-physics step as this kernel's epilogue (kernels_step.hpp post_body): on = 0 leaves it to its own launch
struct PostArgs { int on, wrapper_level, push_count, step_no; };
// TIMED (MQE_PHASE_TIMES=1, tools/dev/phase_walltimes.py): the same kernel with the phase taps of phys_substep live -- every wavefront
// writes the wall clock at the 15 taps of each of its substeps (+ [15]: the substep's end) behind the entry / exit stamps of
// st.wave_times: where the time of a FULL launch goes, phase by phase, as opposed to the lone wavefront of tools/phase_times.py.
// ACT32: the actuator network's layer 2 as the exact f32 MFMA chain (MQE_ACT_F32=1: torques bit for bit the oracle's fmaf chain); default:
// two-plane split-f16 operands on v_mfma_f32_32x32x16_f16 (f32-class accuracy, 22 significand bits per operand).  A compile-time choice: with
// both forms behind a run-time branch the 128-register kernels spilled 20-50 registers.
template <int TA, int TP, int EPW = 1, bool TIMED = false, bool ACT32 = false>
__global__ void __launch_bounds__(64, EPW == 2 ? 2 : SubstepsClass<TP>::waves) k_substeps(const DevModel* __restrict__ m, DevState st, int nsub, int lag_pos, PostArgs pa) {
  extern __shared__ float lds_wave[];
  // EPW = 2 (phys_substep): the two halves of the wavefront run envs 2 b and 2 b + 1; `lane` / `lds` / `e` below are the group's.
  // 2048 wavefronts for 4096 envs = 2 per SIMD with the whole register file each: nothing is laundered, everything invariant
  // over the substeps is hoisted.
  constexpr int LW = 64 / EPW;
#ifdef MQE_LAUNDER2
  constexpr int launder = EPW == 2 ? MQE_LAUNDER2 : SubstepsClass<TP>::launder;
#else
  constexpr int launder = EPW == 2 ? 2 : SubstepsClass<TP>::launder;     // hoisting everything overflows even 256 VGPRs (43 spilled)
#endif
  const int lane_wave = threadIdx.x, e_first = blockIdx.x * EPW;
  own_state(st);                                  // every pointer in a register pair of its own (mqe_common.hpp)
  const int hotv = (int)m->hot[lane_wave];        // DevModel::hot: the physics' wave-uniform constants, one entry per lane, read by v_readlane
  if (st.wave_times && lane_wave == 0) {      // MQE_WAVE_TIMES (tools/dev/wave_times.py): entry / exit time, HW_ID and XCC_ID of the wavefront
    st.wave_times[4 * blockIdx.x] = (long long)wall_clock64();
    st.wave_times[4 * blockIdx.x + 2] = (long long)__builtin_amdgcn_s_getreg(0xF804);
    st.wave_times[4 * blockIdx.x + 3] = (long long)__builtin_amdgcn_s_getreg(0xF814);
  }
  const int grp = EPW == 1 ? 0 : lane_wave / LW, lane = EPW == 1 ? lane_wave : lane_wave - grp * LW;
  const bool evalid = EPW == 1 || e_first + grp < m->N;
  const int e = evalid ? e_first + grp : m->N - 1;
  const PhysShape<TA, TP> shp(m);
  const int A = shp.A, P = shp.P;
  const PhysLds L = phys_lds_layout(A, P, shp.ND, shp.nbody, shp.ndof, m->nsph_env, m->nprim_env, shp.maxc, shp.rowgs, PhysPad<TP>::on);
  float* lds = lds_wave + grp * L.total;
  const int nj = 12 * A;
  const int ctrl = m->control_type;
  const size_t R12 = (size_t)m->R * 12;
  // Nothing of the controller stays in registers across the physics body (it needs every one of the 168 a wave may hold at 3
  // waves per SIMD): the actuator history of each joint lives in LDS, the per-joint constants and the action are re-read (L1 / L2)
  // at the head of every substep.
  float* acth = lds + L.acth;                                    // [4][nj]: e1, e2, v1, v2
  for (int i = lane; i < 4 * nj; i += LW) {
    const int w = i / nj, jt = i - w * nj;
    acth[i] = st.act_hist[(size_t)w * R12 + (size_t)e * nj + jt];
  }
  float* g_root = st.root + (size_t)e * (A + P) * 13;
  float* g_dof = st.dof + (size_t)e * shp.ND * 2;
  for (int i = lane; i < (A + P) * 13; i += LW) lds[L.root + i] = g_root[i];
  for (int i = lane; i < shp.ND * 2; i += LW) lds[L.dof + i] = g_dof[i];
  __syncthreads();
  const PhysDebug nodbg = {nullptr, nullptr, nullptr, 0, nullptr, -1};
#pragma clang loop unroll(disable)
  for (int k = 0; k < nsub; k++) {
    const bool last = k + 1 == nsub;
    // Wave priority by progress.  At equal priority the SIMD's arbiter prefers its oldest wave: the four waves of a SIMD then finish
    // one after the other (lifetimes 81 .. 131 us measured) and the last one runs alone, with nobody to hide its latencies.  A wave
    // that is a substep behind goes first instead, so the four stay abreast and finish together: go1gate 136.6 -> 120.5 us.
    // (Finer steps inside the last substep -- 3,3,2,1 and 0 from the sweep on -- do not help: A/B 117.3 vs 117.2 us; the exits of a
    // SIMD's four waves still spread by +-5 us, tools/dev/wave_times.py, but the SIMD is busy until the last one leaves.)
    if (k == 0) __builtin_amdgcn_s_setprio(3); else if (k == 1) __builtin_amdgcn_s_setprio(2); else if (k == 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
    // Register budget vs. recomputation (measured, MI355X).  Everything the body derives from the lane id and
    // the model alone (indices, LDS addresses, masks, per-lane model constants: ~330 VALU instructions, ~100 values) is invariant over
    // the substeps; left alone the compiler hoists it out of this loop and the kernel needs ~250 VGPRs = 2 waves per SIMD.
    // Laundering the lane id (bit 1) once per substep keeps everything lane-derived inside the loop: the body then fits 128 VGPRs
    // with a handful of spills, at the price of re-deriving those values every substep; what depends on the model alone is
    // wave-uniform, lives in SGPRs and stays hoisted (laundering the model pointer too, bit 0, costs 5-12 %).  That pays exactly when
    // the LDS footprint lets 16 waves sit on a CU (SubstepsClass::small): 4096 envs then run as one round instead of two.
#ifdef MQE_DUMMY_VALU
    {   // experiment: MQE_DUMMY_VALU independent vector instructions per substep -- is the kernel bound by the number of VALU instructions it issues?
      int dummy;
      asm volatile(".rept %1\n v_mov_b32 %0, 0\n .endr" : "=v"(dummy) : "n"(MQE_DUMMY_VALU));
    }
#endif
    const DevModel* mk = m;
    int lane_k = lane_wave;
    if (launder & 1) asm volatile("" : "+s"(mk));
    if (launder & 2) {
      asm volatile("" : "+v"(lane_k));
      lane_k &= 63;                                  // gives the value range of threadIdx.x back to the optimiser
    }
    const int glane_k = EPW == 1 ? lane_k : lane_k - (lane_k / LW) * LW;        // lane within the env's group
    const int j32 = lane_k & 31, h = lane_k >> 5;
    if (ctrl != MQE_CTRL_C) {          // P / V / T (legged_robot.py:384-390): a few FMAs per joint lane instead of the actuator network
      for (int jt = glane_k; jt < nj; jt += LW) {
        const size_t gi = (size_t)e * nj + jt;
        const int j = jt % 12;
        const float asc = st.actions[gi] * m->action_scale;          // no hip reduction (legged_robot.py:380)
        const float q = lds[L.dof + jt * 2], qd = lds[L.dof + jt * 2 + 1], lim = m->torque_limits[j];
        float tau = asc;
        if (ctrl == MQE_CTRL_P) tau = m->kp * (asc + m->default_dof_pos[j] - q) - m->kd * qd;
        else if (ctrl == MQE_CTRL_V) tau = m->kp * (asc - qd) - m->kd * (qd - st.last_dof_vel[gi]) / m->dt;
        tau = clampf(tau, -lim, lim);
        lds[L.tau + jt] = tau;
        if (evalid) {
          st.sub_tau[((size_t)e * 4 + (k < 4 ? k : 3)) * nj + jt] = tau;
          if (last) st.torques[gi] = tau;
        }
      }
    } else {                           // control type C: actuator network (one call site of the physics body below: it is
                                       // inlined, and two copies of its ~9 k instructions would not fit the instruction cache)
      // weight fragments (A operands): row = hidden unit j32, k = this half-wave's element of each k pair; re-read every substep
      // (L1/L2 resident, 5 kB shared by every wave)
      const float* b0 = as_global(mk->actuator.b[0]);      // through the laundered pointer: the 70 fragment
      const float* b1 = as_global(mk->actuator.b[1]);      // loads below stay inside the substep loop (global_load, not flat_load:
      const float* W2 = as_global(mk->actuator.W[2]); const float* b2 = as_global(mk->actuator.b[2]);     // mqe_common.hpp as_global)
      constexpr bool act16 = !ACT32;
      float a1[3], a2[16], w3[16], bb0[16], bb1[16];
      // (the two matrix operands from the fragment-ordered copy: lane-contiguous, one 256 B request per fragment -- W1[j32 * 32 + u] itself
      // is 64 lanes x a 128 B stride, 64 cache lines per instruction, 16 instructions per substep and wavefront)
      const float* fragw = as_global(mk->act_frag) + lane_k;
#pragma unroll
      for (int s2 = 0; s2 < 3; s2++) a1[s2] = fragw[(16 + s2) * 64];
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int u = (r & 3) + 8 * (r >> 2) + 4 * h;
        if constexpr (!act16) a2[r] = fragw[r * 64];
        w3[r] = W2[u]; bb0[r] = b0[u]; bb1[r] = b1[u];
#ifdef MQE_ACT_DUPLOAD
        {   // experiment: the 48 two-address dword loads issued once more (volatile: kept) -- what do they cost at 4 wavefronts per SIMD?
          const float d0 = *(const volatile float*)(W2 + u), d1 = *(const volatile float*)(b0 + u), d2 = *(const volatile float*)(b1 + u);
          asm volatile("" :: "v"(d0), "v"(d1), "v"(d2));
        }
#endif
      }
      const float bout = b2[0];
      // layer 2 on the f16 matrix cores (DevModel::act_f16; MQE_ACT_F32=1 keeps the f32 MFMA chain above): W1 as two f16 planes of 2^14 w in
      // the fragment order of v_mfma_f32_32x32x16_f16 -- [k-step][plane][lane][8] -- one 16 B load per fragment
      // (the four fragments live in the sixteen registers the f32 chain keeps its layer-2 operand in: one set of registers for both forms)
      if constexpr (act16) {
        const h2_gvec* f16w = (const h2_gvec*)(unsigned long long)mk->act_frag16 + lane_k;
#pragma unroll
        for (int f = 0; f < 4; f++) {
          const h2_u32x4 t = f16w[f * 64];
          a2[4 * f] = __uint_as_float(t.x); a2[4 * f + 1] = __uint_as_float(t.y); a2[4 * f + 2] = __uint_as_float(t.z); a2[4 * f + 3] = __uint_as_float(t.w);
        }
      }
      int pos = 0;
      if (m->lag_steps > 0) {          // go1.py:337-339: the lag buffer shifts in every _compute_torques call, i.e. per substep
        const int n = m->lag_steps + 1;
        pos = lag_pos + k; pos -= (pos / n) * n;
      }
      // the network's columns are the joints of the WAVEFRONT: 12 A per env, the envs of the wavefront one after the other (EPW = 2:
      // 48 joints in two tiles of 32); a column's lanes read and write the LDS state of the env the joint belongs to
      const int njw = nj * EPW;
#pragma unroll
      for (int t = 0; t < ACT_TILES; t++) {
        if (t * 32 >= njw) break;                          // wave-uniform
        const int jw = t * 32 + j32;
        const bool ok = jw < njw;
        const int jwc = ok ? jw : 0;
        const int ge = EPW == 1 ? 0 : jwc / nj;            // env of the wavefront this column belongs to
        const int jt = jwc - ge * nj, jc = jt, j = jc % 12;
        const bool ev = EPW == 1 || e_first + ge < m->N;
        const int eg = ev ? e_first + ge : m->N - 1;
        float* ldsg = lds_wave + ge * L.total;
        float* acthg = ldsg + L.acth;
        const size_t gi = (size_t)eg * nj + jc;
        // the joint's inputs are requested together, right behind the weight fragments: the action, its rest angle and torque limit and --
        // with actuator lag -- the ring's oldest slot, which this substep's write (into ANOTHER slot: the ring has lag_steps + 1 >= 2) does
        // not change.  One after the other behind `if (lag && ok)` they were three memory round trips per substep that no other
        // wavefront hides: the four of a SIMD reach this phase together.
        const bool lagged = m->lag_steps > 0;                        // wave-uniform
        const float a_raw = st.actions[gi], ddp = m->default_dof_pos[j], lim = m->torque_limits[j];
        float lag_old = 0.0f;
        size_t lag_wr = 0;
        if (lagged) {                                                // go1.py:337-339 (kernels_step.hpp lag_target: the same ring)
          const size_t R12l = (size_t)m->R * 12;
          const int rd = pos + 1 >= m->lag_steps + 1 ? 0 : pos + 1;
          lag_old = st.lag_buf[(size_t)rd * R12l + gi];
          lag_wr = (size_t)pos * R12l + gi;
        }
        float as = a_raw * m->action_scale;
        if (j % 3 == 0) as *= m->hip_scale_reduction;
        float tgt = as + ddp;
        if (lagged && ok) tgt = lag_old + ddp;                    // (this substep's slot is written with the torque below: no store in front of a wait)
        const float q = ldsg[L.dof + jc * 2], qd = ldsg[L.dof + jc * 2 + 1];
        const float he1 = acthg[jc], he2 = acthg[nj + jc], hv1 = acthg[2 * nj + jc], hv2 = acthg[3 * nj + jc];
        const float err = q - tgt;
        f32x16_p acc1, acc2;
#pragma unroll
        for (int r = 0; r < 16; r++) { acc1[r] = bb0[r]; acc2[r] = bb1[r]; }
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[0], h ? he1 : err, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[1], h ? qd : he2, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[2], h ? hv2 : hv1, acc1, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; r++) acc1[r] = softsign_p(acc1[r]);
        if constexpr (act16) {
          // layer 2 on the f16 matrix cores (mqe_common.hpp: act_layer2_f16) -- the one phase of the substep that grew at 4 wavefronts per SIMD,
          // where all four reach it together and queue for the SIMD's one matrix pipe.  Torques then equal the oracle's f32 fmaf chain to
          // ~1e-6 instead of bit for bit.
          acc2 = act_layer2_f16(acc1, acc2, a2);
        } else {
#pragma unroll
          for (int r = 0; r < 16; r++) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[r], acc1[r], acc2, 0, 0, 0);
        }
        float part = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; r++) part = fmaf(w3[r], softsign_p(acc2[r]), part);
        float tau = part + __shfl_xor(part, 32, 64) + bout;
        tau = clampf(tau, -lim, lim);
        __syncthreads();                                   // every lane has read the history before it shifts
        if (ok && h == 0) {
          acthg[nj + jt] = he1; acthg[jt] = err; acthg[3 * nj + jt] = hv1; acthg[2 * nj + jt] = qd;      // go1.py:347-350
          ldsg[L.tau + jt] = tau;
          if (ev) {
            st.sub_tau[((size_t)eg * 4 + (k < 4 ? k : 3)) * nj + jt] = tau;          // post_decimation_step (legged_robot.py:113)
            if (last) st.torques[gi] = tau;
            if (lagged) st.lag_buf[lag_wr] = as;
          }
        }
      }
    }
    __syncthreads();
    if constexpr (TIMED) {
      long long* tt = st.wave_times + 4 * (size_t)gridDim.x + ((size_t)blockIdx.x * 4 + (k < 4 ? k : 3)) * 16;
      const PhysDebug tdbg = {nullptr, nullptr, nullptr, 0, tt, -1};
      phys_substep<TA, TP, EPW>(mk, st, lds_wave, e_first, lane_k, last ? (PS_STORE_STATE | PS_WRITE_CF) : 0, 0, tdbg, hotv);
      if (lane_wave == 0) tt[15] = (long long)wall_clock64();
    } else
    phys_substep<TA, TP, EPW>(mk, st, lds_wave, e_first, lane_k, last ? (PS_STORE_STATE | PS_WRITE_CF) : 0, 0, nodbg, hotv);
    // post_decimation_step (legged_robot.py:114-115): joint velocities and soft-limit flags after this substep, from the LDS state
    if (evalid)
      for (int jt = glane_k; jt < nj; jt += LW) {
        const float q = lds[L.dof + jt * 2], qd = lds[L.dof + jt * 2 + 1];
        const int j = jt % 12;
        const size_t o = ((size_t)e * 4 + (k < 4 ? k : 3)) * nj + jt;
        st.sub_dof_vel[o] = qd;
        st.sub_exceed[o] = (uint8_t)((q < m->soft_lo[j]) | (q > m->soft_hi[j]));
      }
  }
  __syncthreads();
  if (evalid)
    for (int i = lane; i < 4 * nj; i += LW) {
      const int w = i / nj, jt = i - w * nj;
      st.act_hist[(size_t)w * R12 + (size_t)e * nj + jt] = acth[i];
    }
  if constexpr (ShapeClass<TP>::small || TP > 0) if (pa.on) {      // (every kernel of the 16-envs-per-CU class; with the DevState pointers in register pairs of their own none of them spills)
    // ---- post_physics_step of this wavefront's env(s) (legged_robot.py:117-157 + the task wrapper), from the state just written: the
    // writer and the readers are lanes of this one wavefront (one CU, one vector L1), a workgroup-scope fence orders them
    long long* et = nullptr;
    if constexpr (TIMED) { et = st.wave_times + 68 * (size_t)gridDim.x + (size_t)blockIdx.x * 16; if (lane_wave == 0) et[0] = (long long)wall_clock64(); }
    __threadfence_block();
    __syncthreads();
    constexpr int AMP = (TA == 1 || TA == 2) ? 2 : MQE_MAX_AGENTS;
    // LDS: root and joint states are still where the physics kept them (L.root, L.dof of each env's layout); the staging rows of the
    // post step go into the dead torque / bias / history area behind them ... no: into the link-record area (L.body .. : dead since the sweep)
    // obs rows, last-action rows, NPC rows, the env's actions: post_staging_floats(EPW, AMP, nj, P) floats from L.body on -- 337 of the 936 there for
    // go1gate, 650 of the 740 in front of the second env's root rows for the paired go1plane kernel, 557 for four robots; mqe_sim_create
    // checks the actual layout against the same formula and keeps the separate launch when it does not fit
    float* sb = lds_wave + L.body;
    float* act_l = sb + EPW * AMP * (MQE_OBS_BAG + 24) + EPW * post_npc_stride(P);
    for (int i = lane_wave; i < EPW * nj; i += 64) {          // the wavefront's actions, one coalesced load
      const int ge = i / nj, jt = i - ge * nj;
      if (e_first + ge < m->N) act_l[ge * nj + jt] = st.actions[(size_t)(e_first + ge) * nj + jt];
    }
    __syncthreads();
    if constexpr (TIMED) { if (lane_wave == 0) et[1] = (long long)wall_clock64(); }
    post_body<AMP, EPW, true>(m, st, (int)blockIdx.x, lane_wave, sb, sb + EPW * AMP * MQE_OBS_BAG, sb + EPW * AMP * (MQE_OBS_BAG + 24), pa.wrapper_level, pa.push_count, pa.step_no,
                        lds_wave + L.root, lds_wave + L.dof, act_l, L.total, nj, post_npc_stride(P), et);
    if constexpr (TIMED) { if (lane_wave == 0) et[10] = (long long)wall_clock64(); }
  }
  if (st.wave_times && lane_wave == 0) st.wave_times[4 * blockIdx.x + 1] = (long long)wall_clock64();
}
