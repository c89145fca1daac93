// kernels_tail.hpp -- k_policy_tail: everything of the locomotion policy after the fused first layer in ONE launch
// (reference go1.py:400-407 + :40-41,106-107):
//   adaptation  h1 = ELU(h0 Wa1 + b)  (256 -> 128),  latent = h1 Wa2 + b  (128 -> 2)
//   body        b0 = ELU(pre0 + latent . w_lat)  (the latent columns of layer 0),  b1 = ELU(b0 Wb1 + b)  (512 -> 256),
//               b2 = ELU(b1 Wb2 + b)  (256 -> 128),  joint targets = b2 Wb3 + b  (128 -> 12)
//   post        last-action registers shifted, targets clipped into `actions`
// Replaces five k_gemm_f32 launches + k_body_l0_finish + k_post_policy, which were launch/latency-bound (84 us for
// 3.3 GFLOP at R = 8192).  One 256-thread workgroup owns 32 robots (R = 8192 -> 256 workgroups = one per CU) and walks
// the six dependent stages with the activations in LDS (k-major, row stride 33: the MFMA A operand is one conflict-free
// ds_read_b32).  Weights stream from L2 straight into the B operand: a lane loads float4 = 4 consecutive output
// columns of one k, i.e. the B values of FOUR interleaved 32-column tiles (tile t = columns 4 j + t), so one 1 KiB load
// feeds four v_mfma_f32_32x32x2_f32.  The four waves split (column group) x (K range); K-partials are summed through
// LDS in a fixed order (deterministic).  Exact f32 (f32 MFMA == fmaf chain per partial).
#pragma once
#include "mqe_common.hpp"
#include "kernels_gemm.hpp"

#define TL_ROWS 32
#define TL_AS 33                         // row stride of the k-major activation arrays
#define TL_X (512 * TL_AS)               // floats: b0, and the K-partial buffers
#define TL_Y (256 * TL_AS)               // h0 -> b1
#define TL_Z (128 * TL_AS)               // h1 -> b2
#define TL_C (64 + 640 + 1024)             // latent [32][2], biases (128 + 64 + 256 + 128 + 64), latent weight columns 2 x 512
#define TL_LDS_BYTES ((TL_X + TL_Y + TL_Z + TL_C) * 4)

struct TailArgs {
  const float* P1; int ldp; int ada_h0;       // [R][ldp]: cols [0, 256) = ELU(adaptation h0); [256, 768) = body layer-0 pre-activation
  const float *Wa1, *ba1; int ldwa1;          // [256][128]
  const float *Wa2, *ba2; int ldwa2;          // [128][64] (2 used)
  const float *wl0, *wl1;                     // [512] body layer-0 weights of the two latent inputs
  const float *Wb1, *bb1; int ldwb1;          // [512][256]
  const float *Wb2, *bb2; int ldwb2;          // [256][128]
  const float *Wb3, *bb3; int ldwb3;          // [128][64] (12 used)
  float* lat; int ldl;                        // out: latent [R][ldl]
  float* act; int lda;                        // out: raw joint targets [R][lda]
  float *last_loco, *last_two_loco, *actions; float clip_actions;   // post-policy registers [R][12]
  int R;
};

// ELU with exp(v) - 1 on the hardware exponential: absolute error <= 1.2e-7 (the branchy expm1f polynomial costs more than
// the reductions around it and keeps the loops from being pipelined); the result feeds f32 GEMMs with ~1e-6 noise.
__device__ __forceinline__ float elu_f(float v) { return v > 0 ? v : __expf(v) - 1.0f; }

// acc[t] += A[32 x K-range] * Wt[K-range][group columns 4 j + t]; wave = (column group cg of 128, K part kp of KS).
// The weights are not cache-resident when the kernel starts (layer 0 streams 190 MB through L2 every step) and a CU
// hosts a single workgroup, so memory-level parallelism has to come from the wave itself: the weight stream runs 32
// k-pairs (32 KiB per wave, 128 KiB per CU) ahead of the MFMAs through a ring of named registers (an array carried
// around the loop would be spilled).
template <int K, int KS>
__device__ __forceinline__ void tl_wide(const float* __restrict__ As, const float* __restrict__ Wt, int ldw, int cg, int kp, int lane, f32x16 (&acc)[4]) {
#pragma unroll
  for (int t = 0; t < 4; t++)
#pragma unroll
    for (int i = 0; i < 16; i++) acc[t][i] = 0.0f;
  constexpr int KR = K / KS;                 // k range of this wave, multiple of 64
  const int kh = lane >> 5, j = lane & 31;
  const int kb = kp * KR;
  const float* ap = As + (kb + kh) * TL_AS + j;
  const float* wp = Wt + (size_t)(kb + kh) * ldw + cg * 128 + 4 * j;
  // every CU streams the same weights: rotate the k order per workgroup so that the 32 CUs of an XCD do not pull the same
  // cache line through the same L2 channel at the same moment (a sum over k does not care where it starts)
  const int krot = ((blockIdx.x * 7) & 31) * (KR / 32);            // even, < KR
  float4 W0, W1, W2, W3, W4, W5, W6, W7, W8, W9, W10, W11, W12, W13, W14, W15, W16, W17, W18, W19, W20, W21, W22, W23, W24, W25, W26, W27, W28, W29, W30, W31;
#define TL_ROT(c_) { c_ += krot; if (c_ >= KR) c_ -= KR; }
#define TL_LD(i_, kk_) { int c_ = (kk_); if (c_ > KR - 2) c_ = KR - 2; TL_ROT(c_) W##i_ = *reinterpret_cast<const float4*>(wp + (size_t)c_ * ldw); }
#define TL_STEP(i_, kk_)                                                                 \
  {                                                                                      \
    int ca_ = (kk_); TL_ROT(ca_)                                                         \
    const float a_ = ap[ca_ * TL_AS];                                                    \
    const float4 w_ = W##i_;                                                             \
    if (KR > 64) TL_LD(i_, (kk_) + 64)                                                   \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_, w_.x, acc[0], 0, 0, 0);            \
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_, w_.y, acc[1], 0, 0, 0);            \
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_, w_.z, acc[2], 0, 0, 0);            \
    acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_, w_.w, acc[3], 0, 0, 0);            \
  }
  TL_LD(0, 0) TL_LD(1, 2) TL_LD(2, 4) TL_LD(3, 6) TL_LD(4, 8) TL_LD(5, 10) TL_LD(6, 12) TL_LD(7, 14) TL_LD(8, 16) TL_LD(9, 18) TL_LD(10, 20) TL_LD(11, 22) TL_LD(12, 24) TL_LD(13, 26) TL_LD(14, 28) TL_LD(15, 30) TL_LD(16, 32) TL_LD(17, 34) TL_LD(18, 36) TL_LD(19, 38) TL_LD(20, 40) TL_LD(21, 42) TL_LD(22, 44) TL_LD(23, 46) TL_LD(24, 48) TL_LD(25, 50) TL_LD(26, 52) TL_LD(27, 54) TL_LD(28, 56) TL_LD(29, 58) TL_LD(30, 60) TL_LD(31, 62)
  for (int k2 = 0; k2 < KR; k2 += 64) {
    TL_STEP(0, k2 + 0) TL_STEP(1, k2 + 2) TL_STEP(2, k2 + 4) TL_STEP(3, k2 + 6)
    TL_STEP(4, k2 + 8) TL_STEP(5, k2 + 10) TL_STEP(6, k2 + 12) TL_STEP(7, k2 + 14)
    TL_STEP(8, k2 + 16) TL_STEP(9, k2 + 18) TL_STEP(10, k2 + 20) TL_STEP(11, k2 + 22)
    TL_STEP(12, k2 + 24) TL_STEP(13, k2 + 26) TL_STEP(14, k2 + 28) TL_STEP(15, k2 + 30)
    TL_STEP(16, k2 + 32) TL_STEP(17, k2 + 34) TL_STEP(18, k2 + 36) TL_STEP(19, k2 + 38)
    TL_STEP(20, k2 + 40) TL_STEP(21, k2 + 42) TL_STEP(22, k2 + 44) TL_STEP(23, k2 + 46)
    TL_STEP(24, k2 + 48) TL_STEP(25, k2 + 50) TL_STEP(26, k2 + 52) TL_STEP(27, k2 + 54)
    TL_STEP(28, k2 + 56) TL_STEP(29, k2 + 58) TL_STEP(30, k2 + 60) TL_STEP(31, k2 + 62)
  }
#undef TL_STEP
#undef TL_LD
#undef TL_ROT
}
// partial tile -> red[kp][row][N + 1]
template <int N>
__device__ __forceinline__ void tl_store_partial(float* red, int cg, int kp, int lane, const f32x16 (&acc)[4]) {
  float* base = red + kp * (TL_ROWS * (N + 1)) + cg * 128 + 4 * (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#pragma unroll
    for (int t = 0; t < 4; t++) base[row * (N + 1) + t] = acc[t][r];
  }
}
// out[col][row] (k-major) = ELU(sum_kp red + bias)
template <int N, int KS>
__device__ __forceinline__ void tl_reduce_act(const float* red, const float* bias, float* out, int tid) {
#pragma unroll 4
  for (int idx = tid; idx < TL_ROWS * N; idx += 256) {
    const int col = idx >> 5, row = idx & 31;
    float v = red[row * (N + 1) + col];
#pragma unroll
    for (int p = 1; p < KS; p++) v += red[p * (TL_ROWS * (N + 1)) + row * (N + 1) + col];
    out[col * TL_AS + row] = elu_f(v + bias[col]);
  }
}
// narrow stage (<= 32 output columns, Wt [128][ldw]): every wave one K quarter; partial -> red[wave][32][33]
__device__ __forceinline__ void tl_narrow(const float* __restrict__ As, const float* __restrict__ Wt, int ldw, int wave, int lane, float* red) {
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = 0.0f;
  const int kh = lane >> 5, j = lane & 31;
  const int kb = wave * 32;
  float b[16];
#pragma unroll
  for (int q = 0; q < 16; q++) b[q] = Wt[(size_t)(kb + 2 * q + kh) * ldw + j];      // all 16 requests first
#pragma unroll
  for (int q = 0; q < 16; q++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[(kb + 2 * q + kh) * TL_AS + j], b[q], acc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    red[wave * (TL_ROWS * 33) + row * 33 + j] = acc[r];
  }
}

__global__ void __launch_bounds__(256, 1) k_policy_tail(TailArgs g) {
  extern __shared__ float tl_lds[];
  float* X = tl_lds;
  float* Y = X + TL_X;
  float* Z = Y + TL_Y;
  float* latS = Z + TL_Z;                    // [32][2]
  float* bA1 = latS + 64;  float* bA2 = bA1 + 128;  float* bB1 = bA2 + 64;  float* bB2 = bB1 + 256;  float* bB3 = bB2 + 128;
  float* wL0 = bB3 + 64;   float* wL1 = wL0 + 512;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = blockIdx.x * TL_ROWS;
  // ---- the small vectors (biases, latent weight columns) -> LDS
  {
    for (int i = tid; i < 128; i += 256) { bA1[i] = g.ba1[i]; bB2[i] = g.bb2[i]; }
    for (int i = tid; i < 64; i += 256) { bA2[i] = g.ba2[i]; bB3[i] = g.bb3[i]; }
    for (int i = tid; i < 256; i += 256) bB1[i] = g.bb1[i];
    for (int i = tid; i < 512; i += 256) { wL0[i] = g.wl0[i]; wL1[i] = g.wl1[i]; }
  }
  // ---- stage 0: h0 (adaptation layer-0 activations) -> Y, k-major.  All eight 16 B requests of a thread go out before
  // the first is used (P1 was written by the previous launch: far-memory latency); a wave reads 1 KiB of one row.
  {
    float4 v[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int idx = tid + 256 * q, row = idx >> 6, k4 = idx & 63;
      v[q] = (r0 + row < g.R) ? *reinterpret_cast<const float4*>(g.P1 + (size_t)(r0 + row) * g.ldp + k4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int idx = tid + 256 * q, row = idx >> 6, k4 = idx & 63;
      float* o = Y + (k4 * 4) * TL_AS + row;
      o[0] = v[q].x; o[TL_AS] = v[q].y; o[2 * TL_AS] = v[q].z; o[3 * TL_AS] = v[q].w;
    }
  }
  __syncthreads();
  f32x16 acc[4];
  // ---- stage 1: h1 = ELU(h0 Wa1 + b): 256 -> 128, one column group, K split four ways
  tl_wide<256, 4>(Y, g.Wa1, g.ldwa1, 0, wave, lane, acc);
  tl_store_partial<128>(X, 0, wave, lane, acc);
  __syncthreads();
  tl_reduce_act<128, 4>(X, bA1, Z, tid);
  __syncthreads();
  // ---- stage 2: latent = h1 Wa2 + b: 128 -> 2
  tl_narrow(Z, g.Wa2, g.ldwa2, wave, lane, X);
  __syncthreads();
  if (tid < TL_ROWS * 2) {
    const int row = tid >> 1, c = tid & 1;
    float v = X[row * 33 + c];
#pragma unroll
    for (int p = 1; p < 4; p++) v += X[p * (TL_ROWS * 33) + row * 33 + c];
    v += bA2[c];
    latS[row * 2 + c] = v;
    if (r0 + row < g.R) g.lat[(size_t)(r0 + row) * g.ldl + c] = v;
  }
  __syncthreads();
  // ---- stage 3: b0 = ELU(pre0 + latent . w_lat) -> X (k-major)
  {
    float4 v[16];
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const int idx = tid + 256 * q, row = idx >> 7, k4 = idx & 127;
      v[q] = (r0 + row < g.R) ? *reinterpret_cast<const float4*>(g.P1 + (size_t)(r0 + row) * g.ldp + g.ada_h0 + k4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const int idx = tid + 256 * q, row = idx >> 7, k4 = idx & 127;
      const float4 w0 = *reinterpret_cast<const float4*>(wL0 + k4 * 4), w1 = *reinterpret_cast<const float4*>(wL1 + k4 * 4);
      const float l0 = latS[row * 2], l1 = latS[row * 2 + 1];
      float* o = X + (k4 * 4) * TL_AS + row;
      o[0] = elu_f(fmaf(l1, w1.x, fmaf(l0, w0.x, v[q].x)));
      o[TL_AS] = elu_f(fmaf(l1, w1.y, fmaf(l0, w0.y, v[q].y)));
      o[2 * TL_AS] = elu_f(fmaf(l1, w1.z, fmaf(l0, w0.z, v[q].z)));
      o[3 * TL_AS] = elu_f(fmaf(l1, w1.w, fmaf(l0, w0.w, v[q].w)));
    }
  }
  __syncthreads();
  // ---- stage 4: b1 = ELU(b0 Wb1 + b): 512 -> 256, two column groups x two K halves
  tl_wide<512, 2>(X, g.Wb1, g.ldwb1, wave & 1, wave >> 1, lane, acc);
  __syncthreads();                                   // every wave is done reading b0 before the partials overwrite it
  tl_store_partial<256>(X, wave & 1, wave >> 1, lane, acc);
  __syncthreads();
  tl_reduce_act<256, 2>(X, bB1, Y, tid);
  __syncthreads();
  // ---- stage 5: b2 = ELU(b1 Wb2 + b): 256 -> 128
  tl_wide<256, 4>(Y, g.Wb2, g.ldwb2, 0, wave, lane, acc);
  tl_store_partial<128>(X, 0, wave, lane, acc);
  __syncthreads();
  tl_reduce_act<128, 4>(X, bB2, Z, tid);
  __syncthreads();
  // ---- stage 6: joint targets = b2 Wb3 + b: 128 -> 12; post-policy registers (go1.py:106-107, :40-41)
  tl_narrow(Z, g.Wb3, g.ldwb3, wave, lane, X);
  __syncthreads();
  for (int idx = tid; idx < TL_ROWS * 12; idx += 256) {
    const int row = idx / 12, c = idx - row * 12;
    if (r0 + row >= g.R) continue;
    float v = X[row * 33 + c];
#pragma unroll
    for (int p = 1; p < 4; p++) v += X[p * (TL_ROWS * 33) + row * 33 + c];
    v += bB3[c];
    const size_t i = (size_t)(r0 + row);
    g.act[i * g.lda + c] = v;
    g.last_two_loco[i * 12 + c] = g.last_loco[i * 12 + c];
    g.last_loco[i * 12 + c] = v;
    g.actions[i * 12 + c] = clampf(v, -g.clip_actions, g.clip_actions);
  }
}
