// kernels_tail.hpp -- k_policy_tail: everything of the locomotion policy after the fused first layer in ONE launch
// (reference go1.py:400-407 + :40-41,106-107):
//   adaptation  h1 = ELU(h0 Wa1 + b)  (256 -> 128),  latent = h1 Wa2 + b  (128 -> 2)
//   body        b0 = ELU(pre0 + latent . w_lat)  (the latent columns of layer 0),  b1 = ELU(b0 Wb1 + b)  (512 -> 256),
//               b2 = ELU(b1 Wb2 + b)  (256 -> 128),  joint targets = b2 Wb3 + b  (128 -> 12)
//   post        last-action registers shifted, targets clipped into `actions`
// Replaces five k_gemm_f32 launches + k_body_l0_finish + k_post_policy, which were launch/latency-bound (84 us for
// 3.3 GFLOP at R = 8192).  One 512-thread workgroup owns 32 robots (R = 8192 -> 256 workgroups = one per CU) and walks
// the six dependent stages with the activations in LDS.
//
// Arithmetic: the same two-plane split-f16 scheme as k_gemm_h2 (x = h + l in f16 after a power-of-two scale, products
// hh + hl + lh on v_mfma_f32_32x32x16_f16, f32 accumulation, exact rescale): f32-class accuracy at 3/16 of the f32-MFMA
// instruction cost -- the exact-f32 form of this kernel spent 21 of its 54 us in the matrix pipe alone.
//   * activations: two f16 planes [32 rows][K] in LDS (row stride 2 K + 16 B: the 16 B fragment reads of a ds_read_b128
//     lane group tile all banks), written by the previous stage's epilogue straight from the accumulators.
//   * weights: split once on the host into MFMA-fragment order -- block (column tile ct, 16-k step s, plane p) = 1 KiB,
//     lane l holds the 8 k of column ct * 32 + l % 32, k half l / 32 -- so a wave streams its column tiles from L2 with
//     fully coalesced 16 B per lane loads, straight into the B operand, through a ring of registers that runs 8-16 k
//     steps ahead (a CU hosts a single workgroup: memory-level parallelism has to come from the wave itself).
//   * no K split: wave w owns column tile w over the whole K (the 256-wide layer has eight tiles = all eight waves, the 128-wide ones four), so there are no partial
//     sums to reduce through LDS; the two narrow layers (2 and 12 outputs, padded to one column tile) run on wave 0.
//     The first ring fill of every stage is issued one stage early, so no stage starts by waiting for L2.
//   * every CU streams the same weights: each workgroup starts its k loop at a different step so that the 32 CUs of an
//     XCD do not pull the same cache line through the same L2 channel at the same moment (a sum over k does not care).
#pragma once
#include "mqe_common.hpp"
#include "kernels_gemm.hpp"

#define TL_ROWS 32
#ifndef TL_THREADS
#define TL_THREADS 512                   // eight wavefronts (round 5; -DTL_THREADS=256: the four of rounds 2-4): stage 4's eight column tiles one per wavefront, the
#endif                                   // elementwise stages on twice the lanes; the same sums in the same order, bit for bit.  A/B one box: 27.5 -> 23.8 us

#define TL_SX (2 * 512 + 16)             // row strides in bytes of the activation planes: b0 (K = 512)
#define TL_SY (2 * 256 + 16)             // h0, b1 (K = 256)
#define TL_SZ (2 * 128 + 16)             // h1, b2 (K = 128)
#define TL_PX (TL_ROWS * TL_SX)          // one plane
#define TL_PY (TL_ROWS * TL_SY)
#define TL_PZ (TL_ROWS * TL_SZ)
#define TL_C (64 + 640 + 1024 + 32 * 33)   // floats: latent [32][2], biases (128 + 64 + 256 + 128 + 64), latent weight columns 2 x 512, narrow-stage output [32][33]
#define TL_LDS_BYTES (2 * TL_PX + 2 * TL_PY + 2 * TL_PZ + TL_C * 4)
// activation scale: none.  The hidden activations of the REAL adaptation module are not O(1): on a walking robot with a full
// history its layer 0 reaches +1100 and layer 1 pre-activations of -12000 (measured on the oracle, go1gate rollout); the scale of 64
// used until round 3 saturated everything above 1023 once the ring had filled (step 28 of an episode: joint targets off by up to
// 1e-3; tests/test_gpu_parity.py::test_policy_layer0_compact_history_with_resets is the test that ran long enough to see it).
// Unscaled planes represent |x| <= 65504 with an error of max(2^-22 |x|, 2^-25): the absolute floor of 3e-8 is below the f32
// rounding of the O(1) sums these values enter.
#define TL_ASCALE 1.0f

struct TailLayer { const uint16_t* W; const float* bias; float descale; };   // fragment-ordered planes, see above

struct TailArgs {
  const float* P1; int ldp; int ada_h0;       // [R][ldp]: cols [0, 256) = ELU(adaptation h0); [256, 768) = body layer-0 pre-activation
  TailLayer a1, a2, b1, b2, b3;               // 256->128, 128->2, 512->256, 256->128, 128->12
  const float *wl0, *wl1;                     // [512] body layer-0 weights of the two latent inputs
  float* lat; int ldl;                        // out: latent [R][ldl]
  float* act; int lda;                        // out: raw joint targets [R][lda]
  float *last_loco, *last_two_loco, *actions; float clip_actions;   // post-policy registers [R][12]
  int R;
  int block0;                                 // index of this batch's first row block among the GLOBAL rows (env_id_offset * A / TL_ROWS)
  long long* times;                           // debug (MQE_TAIL_TIMES=1, tools/dev/tail_times.py): [blocks][16] wall clock at the stage boundaries, else null
};
#define TL_STAMP(i) do { if (g.times != nullptr && tid == 0) g.times[(size_t)blockIdx.x * 16 + (i)] = (long long)wall_clock64(); } while (0)

// ELU with exp(v) - 1 on the hardware exponential: absolute error <= 1.2e-7 (the branchy expm1f polynomial costs more than
// the matrix work around it)
__device__ __forceinline__ float elu_f(float v) { return v > 0 ? v : __expf(v) - 1.0f; }

// acc[t] += A[32 x K] * W[K x 32 columns of tile t], K = 16 S.  Ahi: this lane's fragment address in plane 0 at step 0
// (row lane % 32, k half lane / 32), plane 1 `plane` bytes further; Wf: this lane's 16 B in block (tile 0, step 0, plane 0),
// tiles `tile_u4` uint4 apart.  D k-steps of weights in flight; the step order is rotated by krot (< S).
template <int S, int NT, int D>
struct TlRing { h2_u32x4 w[D][NT][2]; };
// request the first D k-steps of a stage's weights; issued before the barrier / the work that precedes the stage, so that
// the L2 latency of every stage start is off the critical path
template <int S, int NT, int D>
__device__ __forceinline__ void tl_fill(const h2_gvec* Wf, int tile_u4, int krot, TlRing<S, NT, D>& q) {
#pragma unroll
  for (int i = 0; i < D; i++) {
    const int sr = (i + krot) & (S - 1);
#pragma unroll
    for (int t = 0; t < NT; t++) { q.w[i][t][0] = Wf[t * tile_u4 + sr * 128]; q.w[i][t][1] = Wf[t * tile_u4 + sr * 128 + 64]; }
  }
}
template <int S, int NT, int D>
__device__ __forceinline__ void tl_mm(const unsigned char* Ahi, int plane, const h2_gvec* Wf, int tile_u4, int krot, TlRing<S, NT, D>& q, f32x16* acc) {
  h2_u32x4 aq[2][2];
  {
    const int sr = krot & (S - 1);
    aq[0][0] = *reinterpret_cast<const h2_u32x4*>(Ahi + sr * 32); aq[0][1] = *reinterpret_cast<const h2_u32x4*>(Ahi + plane + sr * 32);
  }
#pragma unroll
  for (int s = 0; s < S; s++) {
    if (s + 1 < S) {
      const int sr = (s + 1 + krot) & (S - 1);
      aq[(s + 1) & 1][0] = *reinterpret_cast<const h2_u32x4*>(Ahi + sr * 32);
      aq[(s + 1) & 1][1] = *reinterpret_cast<const h2_u32x4*>(Ahi + plane + sr * 32);
    }
    const f16x8 ah = __builtin_bit_cast(f16x8, aq[s & 1][0]), al = __builtin_bit_cast(f16x8, aq[s & 1][1]);
#pragma unroll
    for (int t = 0; t < NT; t++) {
      const f16x8 wh = __builtin_bit_cast(f16x8, q.w[s % D][t][0]), wl = __builtin_bit_cast(f16x8, q.w[s % D][t][1]);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh, acc[t], 0, 0, 0);
    }
    if (s + D < S) {
      const int sr = (s + D + krot) & (S - 1);
#pragma unroll
      for (int t = 0; t < NT; t++) { q.w[s % D][t][0] = Wf[t * tile_u4 + sr * 128]; q.w[s % D][t][1] = Wf[t * tile_u4 + sr * 128 + 64]; }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// accumulator tile (columns col0 .. col0 + 31 of the layer) -> ELU(acc * descale + bias) as two f16 planes [row][k = column]
__device__ __forceinline__ void tl_store_act(const f32x16& acc, float descale, const float* bias, int col0, unsigned char* out, int stride, int plane, int lane) {
  const int col = col0 + (lane & 31);
  const float b = bias[col];
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    uint16_t h, l;
    split2(elu_f(fmaf(acc[r], descale, b)), TL_ASCALE, h, l);
    *reinterpret_cast<uint16_t*>(out + row * stride + col * 2) = h;
    *reinterpret_cast<uint16_t*>(out + plane + row * stride + col * 2) = l;
  }
}

__global__ void __launch_bounds__(TL_THREADS, 1) k_policy_tail(TailArgs g) {
  constexpr int TLT = TL_THREADS, NW = TLT / 64, NT4 = 8 / NW, QH = 2048 / TLT, QB = 4096 / TLT;
  constexpr int D1 = TLT == 256 ? 16 : 8;      // k-steps of weights in flight in the 16-step stages (eight wavefronts: 256 registers each)
  extern __shared__ __attribute__((aligned(16))) unsigned char tl_lds[];
  unsigned char* X = tl_lds;                  // b0 planes
  unsigned char* Y = X + 2 * TL_PX;           // h0, later b1
  unsigned char* Z = Y + 2 * TL_PY;           // h1, later b2
  float* latS = reinterpret_cast<float*>(Z + 2 * TL_PZ);     // [32][2]
  float* bA1 = latS + 64;  float* bA2 = bA1 + 128;  float* bB1 = bA2 + 64;  float* bB2 = bB1 + 256;  float* bB3 = bB2 + 128;
  float* wL0 = bB3 + 64;   float* wL1 = wL0 + 512;  float* nar = wL1 + 512;   // nar: [32][33] raw output of a narrow stage
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r0 = blockIdx.x * TL_ROWS;
  TL_STAMP(0);
  // k order rotated per row block to spread the L2 requests for the weights; keyed by the GLOBAL row block, so that an env's
  // summation order (hence its bits) does not depend on which shard of the batch it is computed in
  const int krot = ((blockIdx.x + g.block0) * 7) & 31;
  const int frow = lane & 31, fhalf = lane >> 5;
  const h2_gvec* Wa1 = (const h2_gvec*)(g.a1.W) + lane;
  const h2_gvec* Wa2 = (const h2_gvec*)(g.a2.W) + lane;
  const h2_gvec* Wb1 = (const h2_gvec*)(g.b1.W) + lane;
  const h2_gvec* Wb2 = (const h2_gvec*)(g.b2.W) + lane;
  const h2_gvec* Wb3 = (const h2_gvec*)(g.b3.W) + lane;
  // ---- weights of the first two stages, then all of this workgroup's P1 rows, are requested before anything else
  TlRing<16, 1, D1> q1;  TlRing<8, 1, 8> q2;
  if (wave < 4) tl_fill(Wa1 + wave * (16 * 128), 0, krot, q1);
  if (wave == 0) tl_fill(Wa2, 0, 0, q2);
  // ---- all of this workgroup's P1 rows (P1 was written by the previous launch:
  // far-memory latency); h0 is consumed right away, the body pre-activations after the latent exists (stage 3)
  float4 vh[QH], vb[QB];
#pragma unroll
  for (int q = 0; q < QH; q++) {
    const int idx = tid + TLT * q, row = idx >> 6, k4 = idx & 63;
    vh[q] = *reinterpret_cast<const float4*>(g.P1 + (size_t)min(r0 + row, g.R - 1) * g.ldp + k4 * 4);
  }
#pragma unroll
  for (int q = 0; q < QB; q++) {
    const int idx = tid + TLT * q, row = idx >> 7, k4 = idx & 127;
    vb[q] = *reinterpret_cast<const float4*>(g.P1 + (size_t)min(r0 + row, g.R - 1) * g.ldp + g.ada_h0 + k4 * 4);
  }
  // (the last-action register this thread shifts at the very end: an input of the launch, requested here instead of behind the target's store)
  float ll_pre = 0.0f;
  if constexpr (TLT >= TL_ROWS * 12) { const int row = tid / 12; if (tid < TL_ROWS * 12 && r0 + row < g.R) ll_pre = g.last_loco[(size_t)r0 * 12 + tid]; }
  // ---- the small vectors (biases, latent weight columns) -> LDS.  All of them are requested before the first is stored (clamped indices: no
  // branch around a load): as `if (tid < n) lds[tid] = g[tid]` blocks each block waited for its own load, five memory round trips one after
  // the other behind the P1 rows
  if constexpr (TLT == 512) {
    const float s0 = g.a1.bias[tid & 127], s1 = g.b2.bias[tid & 127], s2 = g.a2.bias[tid & 63], s3 = g.b3.bias[tid & 63], s4 = g.b1.bias[tid & 255],
                s5 = g.wl0[tid], s6 = g.wl1[tid];
    if (tid < 128) { bA1[tid] = s0; bB2[tid] = s1; }
    if (tid < 64) { bA2[tid] = s2; bB3[tid] = s3; }
    if (tid < 256) bB1[tid] = s4;
    wL0[tid] = s5; wL1[tid] = s6;
  } else {
    for (int i = tid; i < 128; i += TLT) { bA1[i] = g.a1.bias[i]; bB2[i] = g.b2.bias[i]; }
    for (int i = tid; i < 64; i += TLT) { bA2[i] = g.a2.bias[i]; bB3[i] = g.b3.bias[i]; }
    for (int i = tid; i < 256; i += TLT) bB1[i] = g.b1.bias[i];
    for (int i = tid; i < 512; i += TLT) { wL0[i] = g.wl0[i]; wL1[i] = g.wl1[i]; }
  }
  // ---- stage 0: h0 (adaptation layer-0 activations) -> Y as split planes [row][k]
#pragma unroll
  for (int q = 0; q < QH; q++) {
    const int idx = tid + TLT * q, row = idx >> 6, k4 = idx & 63;
    uint16_t h0, l0, h1, l1, h2, l2, h3, l3;
    split2(vh[q].x, TL_ASCALE, h0, l0); split2(vh[q].y, TL_ASCALE, h1, l1); split2(vh[q].z, TL_ASCALE, h2, l2); split2(vh[q].w, TL_ASCALE, h3, l3);
    *reinterpret_cast<uint2*>(Y + row * TL_SY + k4 * 8) = make_uint2(h0 | ((unsigned)h1 << 16), h2 | ((unsigned)h3 << 16));
    *reinterpret_cast<uint2*>(Y + TL_PY + row * TL_SY + k4 * 8) = make_uint2(l0 | ((unsigned)l1 << 16), l2 | ((unsigned)l3 << 16));
  }
  __syncthreads();
  TL_STAMP(1);
  f32x16 acc[2];
#define TL_ZERO(n_) _Pragma("unroll") for (int t_ = 0; t_ < (n_); t_++) _Pragma("unroll") for (int i_ = 0; i_ < 16; i_++) acc[t_][i_] = 0.0f;
  // ---- stage 1: h1 = ELU(h0 Wa1 + b): 256 -> 128, wave w = column tile w
  if (wave < 4) {
    TL_ZERO(1)
    tl_mm(Y + frow * TL_SY + fhalf * 16, TL_PY, Wa1 + wave * (16 * 128), 0, krot, q1, acc);
    tl_store_act(acc[0], g.a1.descale, bA1, wave * 32, Z, TL_SZ, TL_PZ, lane);
  }
#ifndef TL_D4
#define TL_D4 8
#endif
  TlRing<32, NT4, TL_D4> q4;                    // stage 4's first weights travel during stages 2 and 3
  tl_fill(Wb1 + (NT4 * wave) * (32 * 128), 32 * 128, krot, q4);
  __syncthreads();
  TL_STAMP(2);
  // ---- stage 2: latent = h1 Wa2 + b: 128 -> 2 (one column tile, wave 0)
  if (wave == 0) {
    TL_ZERO(1)
    tl_mm(Z + frow * TL_SZ + fhalf * 16, TL_PZ, Wa2, 0, 0, q2, acc);
#pragma unroll
    for (int r = 0; r < 16; r++) nar[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 33 + (lane & 31)] = acc[0][r];
  }
  __syncthreads();
  TL_STAMP(3);
  if (tid < TL_ROWS * 2) {
    const int row = tid >> 1, c = tid & 1;
    const float v = fmaf(nar[row * 33 + c], g.a2.descale, bA2[c]);
    latS[row * 2 + c] = v;
    if (r0 + row < g.R) g.lat[(size_t)(r0 + row) * g.ldl + c] = v;
  }
  __syncthreads();
  TL_STAMP(4);
  // ---- stage 3: b0 = ELU(pre0 + latent . w_lat) -> X
#pragma unroll
  for (int q = 0; q < QB; q++) {
    const int idx = tid + TLT * q, row = idx >> 7, k4 = idx & 127;
    const float4 w0 = *reinterpret_cast<const float4*>(wL0 + k4 * 4), w1 = *reinterpret_cast<const float4*>(wL1 + k4 * 4);
    const float l0 = latS[row * 2], l1 = latS[row * 2 + 1];
    uint16_t h0, q0, h1, q1, h2, q2, h3, q3;
    split2(elu_f(fmaf(l1, w1.x, fmaf(l0, w0.x, vb[q].x))), TL_ASCALE, h0, q0);
    split2(elu_f(fmaf(l1, w1.y, fmaf(l0, w0.y, vb[q].y))), TL_ASCALE, h1, q1);
    split2(elu_f(fmaf(l1, w1.z, fmaf(l0, w0.z, vb[q].z))), TL_ASCALE, h2, q2);
    split2(elu_f(fmaf(l1, w1.w, fmaf(l0, w0.w, vb[q].w))), TL_ASCALE, h3, q3);
    *reinterpret_cast<uint2*>(X + row * TL_SX + k4 * 8) = make_uint2(h0 | ((unsigned)h1 << 16), h2 | ((unsigned)h3 << 16));
    *reinterpret_cast<uint2*>(X + TL_PX + row * TL_SX + k4 * 8) = make_uint2(q0 | ((unsigned)q1 << 16), q2 | ((unsigned)q3 << 16));
  }
  __syncthreads();
  TL_STAMP(5);
  // ---- stage 4: b1 = ELU(b0 Wb1 + b): 512 -> 256, wave w = column tiles 2 w, 2 w + 1
  TL_ZERO(NT4)
  tl_mm(X + frow * TL_SX + fhalf * 16, TL_PX, Wb1 + (NT4 * wave) * (32 * 128), 32 * 128, krot, q4, acc);
  TlRing<16, 1, D1> q5;  TlRing<8, 1, 8> q6;
  if (wave < 4) tl_fill(Wb2 + wave * (16 * 128), 0, krot, q5);
  if (wave == 0) tl_fill(Wb3, 0, 0, q6);
#pragma unroll
  for (int t = 0; t < NT4; t++) tl_store_act(acc[t], g.b1.descale, bB1, (NT4 * wave + t) * 32, Y, TL_SY, TL_PY, lane);     // h0 is dead since stage 1
  __syncthreads();
  TL_STAMP(6);
  // ---- stage 5: b2 = ELU(b1 Wb2 + b): 256 -> 128
  if (wave < 4) {
    TL_ZERO(1)
    tl_mm(Y + frow * TL_SY + fhalf * 16, TL_PY, Wb2 + wave * (16 * 128), 0, krot, q5, acc);
    tl_store_act(acc[0], g.b2.descale, bB2, wave * 32, Z, TL_SZ, TL_PZ, lane);           // h1 is dead since stage 2
  }
  __syncthreads();
  TL_STAMP(7);
  // ---- stage 6: joint targets = b2 Wb3 + b: 128 -> 12 (wave 0); post-policy registers (go1.py:106-107, :40-41)
  if (wave == 0) {
    TL_ZERO(1)
    tl_mm(Z + frow * TL_SZ + fhalf * 16, TL_PZ, Wb3, 0, 0, q6, acc);
#pragma unroll
    for (int r = 0; r < 16; r++) nar[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 33 + (lane & 31)] = acc[0][r];
  }
#undef TL_ZERO
  __syncthreads();
  TL_STAMP(8);
  for (int idx = tid; idx < TL_ROWS * 12; idx += TLT) {
    const int row = idx / 12, c = idx - row * 12;
    if (r0 + row >= g.R) continue;
    const float v = fmaf(nar[row * 33 + c], g.b3.descale, bB3[c]);
    const size_t i = (size_t)(r0 + row);
    g.act[i * g.lda + c] = v;
    g.last_two_loco[i * 12 + c] = TLT >= TL_ROWS * 12 ? ll_pre : g.last_loco[i * 12 + c];
    g.last_loco[i * 12 + c] = v;
    g.actions[i * 12 + c] = clampf(v, -g.clip_actions, g.clip_actions);
  }
  TL_STAMP(9);
}
