// kernels_gemm.hpp -- f32 MFMA GEMM with fused bias + ELU epilogue for the locomotion-policy MLPs
// (adaptation module + body, reference go1.py:400-407).
//
//   C[M, N] = act( A[M, K] * Wt[K, N] + bias[N] )      A row-major (lda), Wt row-major [K][ldw] (pre-transposed)
//
// * v_mfma_f32_32x32x2_f32: exact f32, and on gfx950 bitwise equal to a k-ordered fmaf chain, which is exactly how
//   the CPU oracle accumulates -> the policy outputs can be compared tightly.
// * 256 threads = 4 wavefronts per workgroup, 64x64 output tile (each wave one 32x32 MFMA accumulator = 16 VGPRs),
//   K stepped by 16 through LDS.  A is staged k-major so that the MFMA operand read is one conflict-free ds_read_b32.
// * `a_rot`/`a_ring`: the A operand can be a ring buffer along K (the 30 x 72 history ring): logical k maps to
//   physical (k + a_rot) mod a_ring, in float4 units, so the newest frame is always the last 72 logical columns
//   without ever shifting 8.6 kB per robot per step (reference go1.py:102 re-concatenates the whole history).
// * tile->workgroup mapping is XCD-aware: consecutive workgroup ids land on different XCDs (id % 8), so ids are
//   remapped such that the N-tiles of one M-stripe share an XCD and re-read the A stripe from that XCD's L2.
#pragma once
#include "mqe_common.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GB_M 64
#define GB_N 64
#define GB_K 16

struct GemmArgs {
  const float* A; int lda; int a_rot4; int a_ring4;   // rotation / ring length in float4 units (0 = plain)
  const float* Wt; int ldw;
  const float* bias;
  float* C; int ldc;
  int M, N, K;          // N multiple of 64, K multiple of 16
  int act_cols;         // ELU applied to output columns [0, act_cols); identity beyond
};

__global__ void __launch_bounds__(256) k_gemm_f32(GemmArgs g) {
  __shared__ float As[GB_K][GB_M + 1];
  __shared__ float Bs[GB_K][GB_N + 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware remap: ntn consecutive N-tiles of one M-stripe go to one XCD
  const int ntn = g.N / GB_N, ntm = (g.M + GB_M - 1) / GB_M;
  int bid = blockIdx.x;
  const int total = ntn * ntm;
  {
    const int xcd = bid & 7, slot = bid >> 3;
    const int per = (total + 7) >> 3;
    int lin = xcd * per + slot;
    if ((total & 7) == 0) bid = lin;     // exact remap only when the grid divides evenly
  }
  const int tm = bid / ntn, tn = bid - tm * ntn;
  const int m0 = tm * GB_M, n0 = tn * GB_N;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = 0.0f;
  // global load assignment
  const int am = tid >> 2, akq = (tid & 3) * 4;          // A: row am, 4 consecutive k
  const int bk = tid >> 4, bnq = (tid & 15) * 4;         // B: row bk, 4 consecutive n
  const bool a_ok = (m0 + am) < g.M;
  const float* Arow = g.A + (size_t)(m0 + am) * g.lda;
  // register prefetch: the global loads of tile t+1 are issued before the MFMAs of tile t, so their latency hides
  // behind 8 x 64 cycles of matrix work instead of being exposed at the LDS store
  float4 av = make_float4(0.f, 0.f, 0.f, 0.f), bv;
  auto load_tile = [&](int k0) {
    if (a_ok) {
      int k4 = (k0 + akq) >> 2;
      if (g.a_ring4) { k4 += g.a_rot4; if (k4 >= g.a_ring4) k4 -= g.a_ring4; }
      av = *reinterpret_cast<const float4*>(Arow + (size_t)k4 * 4);
    }
    bv = *reinterpret_cast<const float4*>(g.Wt + (size_t)(k0 + bk) * g.ldw + n0 + bnq);
  };
  load_tile(0);
  for (int k0 = 0; k0 < g.K; k0 += GB_K) {
    __syncthreads();                     // previous tile fully consumed
    As[akq + 0][am] = av.x; As[akq + 1][am] = av.y; As[akq + 2][am] = av.z; As[akq + 3][am] = av.w;
    *reinterpret_cast<float4*>(&Bs[bk][bnq]) = bv;
    __syncthreads();
    if (k0 + GB_K < g.K) load_tile(k0 + GB_K);
#pragma unroll
    for (int kk = 0; kk < GB_K; kk += 2) {
      float a = As[kk + (lane >> 5)][wm * 32 + (lane & 31)];
      float b = Bs[kk + (lane >> 5)][wn * 32 + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  }
  // epilogue: D[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31]
  const int col = n0 + wn * 32 + (lane & 31);
  const float bias = g.bias ? g.bias[col] : 0.0f;
  const bool do_act = col < g.act_cols;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (row < g.M) {
      float v = acc[r] + bias;
      if (do_act) v = v > 0 ? v : expm1f(v);
      g.C[(size_t)row * g.ldc + col] = v;
    }
  }
}

// ----------------------------------------------------------------------------------------------------------------------
// k_gemm_b3: the same product on the bf16 matrix cores with f32-equivalent accuracy ("split-bf16", 3 planes).
//
// Every f32 operand x is carried as three bf16 planes x = h + l + s (h = rne(x), l = rne(x - h), s = rne(x - h - l):
// 3 x 8 = 24 significand bits, i.e. the f32 value itself; the f32 exponent range is kept, unlike fp16 splits).
// A product a*b is the six bf16 x bf16 terms down to 2^-24 relative (hh, hl, lh, ll, hs, sh; the dropped ls, sl, ss are
// <= 2^-24 |ab|), each exact in the MFMA and accumulated in f32.  Six v_mfma_f32_32x32x16_bf16 replace eight
// v_mfma_f32_32x32x2_f32 per 16 k: 16x the rate per instruction, 2.7x net of the extra terms.
//   * the producers write the planes: k_pre_policy (new history frame), this kernel's epilogue (hidden activations),
//     k_body_l0_finish (body layer 0 after the latent columns); weights are split once on the host.
//   * block 128 x 64, 4 waves as 2 x 2, wave tile 64 x 32 = 2 accumulators; K step 32 through LDS (row = 32 k of one
//     plane, padded to 80 B so that the 16 B fragment reads of 32 rows spread over the banks).
//   * A may be the history ring: 8-element units, logical unit u -> (u + rot) mod ring (frames are 72 = 9 units).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define G3_M 128
#define G3_N 64
#define G3_K 32
#define G3_ROWB 80          // bytes per LDS row (64 B of data + 16 B pad)

struct Gemm3Args {
  const uint16_t* A; int lda; size_t a_plane; int a_rot8, a_ring8;
  const uint16_t* W; int ldw; size_t w_plane;          // W [Npad][Kpad] per plane (K contiguous: the (out,in) layout)
  const float* bias;
  float* C; int ldc;                                   // f32 result (nullable)
  uint16_t* C3; int ldc3; size_t c_plane; int c3_cols;  // split result for the next layer, columns [0, c3_cols)
  int M, N, K;                                         // N multiple of 64, K multiple of 32
  int act_cols;
};

__global__ void __launch_bounds__(256) k_gemm_b3(Gemm3Args g) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[3 * (G3_M + G3_N) * G3_ROWB];
  unsigned char* As = lds;                                   // [plane][row 0..127][80 B]
  unsigned char* Bs = lds + 3 * G3_M * G3_ROWB;              // [plane][row 0..63][80 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = g.N / G3_N, ntm = (g.M + G3_M - 1) / G3_M;
  int bid = blockIdx.x;
  const int total = ntn * ntm;
  if ((total & 7) == 0) { const int xcd = bid & 7, slot = bid >> 3; bid = xcd * (total >> 3) + slot; }
  const int tm = bid / ntn, tn = bid - tm * ntn;
  const int m0 = tm * G3_M, n0 = tn * G3_N;
  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; t++)
#pragma unroll
    for (int i = 0; i < 16; i++) acc[t][i] = 0.0f;
  // staging assignment: 16 B units; A rows tid>>2 and 64 + tid>>2, unit tid&3; B row tid>>2, unit tid&3
  const int srow = tid >> 2, sunit = tid & 3;
  const bool a_ok0 = (m0 + srow) < g.M, a_ok1 = (m0 + 64 + srow) < g.M;
  const uint16_t* Ar0 = g.A + (size_t)(m0 + srow) * g.lda;
  const uint16_t* Ar1 = g.A + (size_t)(m0 + 64 + srow) * g.lda;
  const uint16_t* Wr = g.W + (size_t)(n0 + srow) * g.ldw;
  // prefetch registers (plain scalars: arrays here end up in scratch): tile t+1 is requested before tile t is multiplied.
  // (A second register set for t+2 was measured slower: 188 registers drop the CU to 2 resident blocks.)
  uint4 pa00, pa01, pa10, pa11, pa20, pa21, pb0, pb1, pb2;
  const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
#define G3_LOAD_TILE(P, k0_)                                                                                    \
  {                                                                                                             \
    const int u_ = ((k0_) >> 3) + sunit;                                                                        \
    int ua_ = u_;                                                                                               \
    if (g.a_ring8) { ua_ += g.a_rot8; if (ua_ >= g.a_ring8) ua_ -= g.a_ring8; if (ua_ >= g.a_ring8) ua_ -= g.a_ring8; } \
    const size_t oa_ = (size_t)ua_ * 8, ow_ = (size_t)u_ * 8;                                                    \
    P##a00 = a_ok0 ? *reinterpret_cast<const uint4*>(Ar0 + oa_) : z4;                                           \
    P##a01 = a_ok1 ? *reinterpret_cast<const uint4*>(Ar1 + oa_) : z4;                                           \
    P##a10 = a_ok0 ? *reinterpret_cast<const uint4*>(Ar0 + g.a_plane + oa_) : z4;                               \
    P##a11 = a_ok1 ? *reinterpret_cast<const uint4*>(Ar1 + g.a_plane + oa_) : z4;                               \
    P##a20 = a_ok0 ? *reinterpret_cast<const uint4*>(Ar0 + 2 * g.a_plane + oa_) : z4;                           \
    P##a21 = a_ok1 ? *reinterpret_cast<const uint4*>(Ar1 + 2 * g.a_plane + oa_) : z4;                           \
    P##b0 = *reinterpret_cast<const uint4*>(Wr + ow_);                                                          \
    P##b1 = *reinterpret_cast<const uint4*>(Wr + g.w_plane + ow_);                                              \
    P##b2 = *reinterpret_cast<const uint4*>(Wr + 2 * g.w_plane + ow_);                                          \
  }
#define G3_STORE_TILE(P)                                                                                        \
  {                                                                                                             \
    unsigned char* wa = As + srow * G3_ROWB + sunit * 16;                                                       \
    unsigned char* wb = Bs + srow * G3_ROWB + sunit * 16;                                                       \
    *reinterpret_cast<uint4*>(wa) = P##a00;                                                                     \
    *reinterpret_cast<uint4*>(wa + 64 * G3_ROWB) = P##a01;                                                      \
    *reinterpret_cast<uint4*>(wa + G3_M * G3_ROWB) = P##a10;                                                    \
    *reinterpret_cast<uint4*>(wa + (G3_M + 64) * G3_ROWB) = P##a11;                                             \
    *reinterpret_cast<uint4*>(wa + 2 * G3_M * G3_ROWB) = P##a20;                                                \
    *reinterpret_cast<uint4*>(wa + (2 * G3_M + 64) * G3_ROWB) = P##a21;                                         \
    *reinterpret_cast<uint4*>(wb) = P##b0;                                                                      \
    *reinterpret_cast<uint4*>(wb + G3_N * G3_ROWB) = P##b1;                                                     \
    *reinterpret_cast<uint4*>(wb + 2 * G3_N * G3_ROWB) = P##b2;                                                 \
  }
#define G3_COMPUTE()                                                                                            \
  _Pragma("unroll") for (int ks = 0; ks < 2; ks++) {                                                            \
    bf16x8 b[3], a[2][3];                                                                                       \
    _Pragma("unroll") for (int p = 0; p < 3; p++) {                                                             \
      b[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Bs + (p * G3_N + wn * 32 + frow) * G3_ROWB + ks * 32 + fk)); \
      _Pragma("unroll") for (int t = 0; t < 2; t++)                                                             \
        a[t][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(As + (p * G3_M + wm * 64 + t * 32 + frow) * G3_ROWB + ks * 32 + fk)); \
    }                                                                                                           \
    _Pragma("unroll") for (int t = 0; t < 2; t++) { /* small terms first */                                     \
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][2], b[0], acc[t], 0, 0, 0);                         \
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][0], b[2], acc[t], 0, 0, 0);                         \
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][1], b[1], acc[t], 0, 0, 0);                         \
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][1], b[0], acc[t], 0, 0, 0);                         \
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][0], b[1], acc[t], 0, 0, 0);                         \
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][0], b[0], acc[t], 0, 0, 0);                         \
    }                                                                                                           \
  }
  const int frow = lane & 31, fk = (lane >> 5) * 16;           // fragment: row, byte offset of its 8 k inside a 16-k step
  G3_LOAD_TILE(p, 0);
  for (int k0 = 0; k0 < g.K; k0 += G3_K) {
    __syncthreads();
    G3_STORE_TILE(p);
    __syncthreads();
    if (k0 + G3_K < g.K) G3_LOAD_TILE(p, k0 + G3_K);
    G3_COMPUTE();
  }
#undef G3_COMPUTE
#undef G3_STORE_TILE
#undef G3_LOAD_TILE
  const int col = n0 + wn * 32 + (lane & 31);
  const float bias = g.bias ? g.bias[col] : 0.0f;
  const bool do_act = col < g.act_cols;
  const bool do_c3 = g.C3 != nullptr && col < g.c3_cols;
#pragma unroll
  for (int t = 0; t < 2; t++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int row = m0 + wm * 64 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row < g.M) {
        float v = acc[t][r] + bias;
        if (do_act) v = v > 0 ? v : expm1f(v);
        if (g.C) g.C[(size_t)row * g.ldc + col] = v;
        if (do_c3) {
          uint16_t h, l, s;
          split3(v, h, l, s);
          uint16_t* o = g.C3 + (size_t)row * g.ldc3 + col;
          o[0] = h; o[g.c_plane] = l; o[2 * g.c_plane] = s;
        }
      }
    }
}
