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
//   * producers write the planes (interleaved per 8-element unit, see Gemm3Args): k_pre_policy (new history frame);
//     weights are split once on the host.
//   * with 2.7x less matrix time the LDS becomes the critical resource (a 64 x 32 wave tile reads 0.75 fragments per
//     MFMA and measured LDS-bound), so the tile is large: block 128 x 192, 4 waves as 2 x 2, wave tile 64 x 96 = 6
//     accumulators (0.42 fragment reads per MFMA).  M = 8192, N = 768 gives exactly 256 blocks = one per CU, so the
//     pipeline is explicit instead of relying on other resident blocks: LDS double buffer (2 x 77 KB), k-tile t+2 in
//     flight from global while tile t is multiplied and tile t+1 is written to the other buffer; one barrier per tile.
//   * LDS row = 32 k of one plane padded to 80 B, so the 16 B fragment reads of 16 consecutive rows tile all banks.
//   * A may be the history ring: 8-element units, logical unit u -> (u + rot) mod ring (frames are 72 = 9 units).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// explicit global address space: pointers selected between two kernel-argument buffers degrade to flat loads, whose
// completion is also counted by lgkmcnt, i.e. every LDS wait would wait for the prefetch as well
typedef unsigned int g3_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) g3_u32x4 g3_gvec;

#define G3_M 128
#define G3_N 192
#define G3_K 32
#define G3_ROWB 80                                  // bytes per LDS row (64 B of data + 16 B pad)
#define G3_ROWS (G3_M + G3_N)
#define G3_BUF (3 * G3_ROWS * G3_ROWB)              // one LDS buffer: [plane][A rows | W rows][80 B]
#define G3_LDS_BYTES (2 * G3_BUF)
#define G3_NLD ((3 * G3_ROWS * 4) / 256)            // 16 B units per thread per k-tile (= 15)

struct Gemm3Args {
  // operands are plane-interleaved per 8-element unit: row r, element k of plane p at r * ld + ((k / 8) * 3 + p) * 8 + k % 8
  // (ld = 3 * K elements), so the 32 k x 3 planes a row contributes to a k-tile are 192 contiguous bytes: whole 128 B
  // lines are consumed at once (separate planes half-used every line and fetched it twice through the 32 KB L1)
  const uint16_t* A; int lda; int a_rot8, a_ring8;
  const uint16_t* W; int ldw;                          // W rows = output units (the (out,in) layout), same interleaving
  const float* bias;
  float* C; int ldc;
  int M, N, K;                                         // N multiple of 192, K multiple of 64
  int act_cols;
};

__global__ void __launch_bounds__(256, 1) k_gemm_b3(Gemm3Args g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds3[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = g.N / G3_N, ntm = (g.M + G3_M - 1) / G3_M;
  int bid = blockIdx.x;
  const int total = ntn * ntm;
  if ((total & 7) == 0) { const int xcd = bid & 7, slot = bid >> 3; bid = xcd * (total >> 3) + slot; }
  const int tm = bid / ntn, tn = bid - tm * ntn;
  const int m0 = tm * G3_M, n0 = tn * G3_N;
  f32x16 acc00, acc01, acc02, acc10, acc11, acc12;
#pragma unroll
  for (int i = 0; i < 16; i++) { acc00[i] = 0.0f; acc01[i] = 0.0f; acc02[i] = 0.0f; acc10[i] = 0.0f; acc11[i] = 0.0f; acc12[i] = 0.0f; }
  // staging: per plane this thread moves A rows srow, 64 + srow and W rows srow, 64 + srow, 128 + srow, one 16 B unit
  // (8 k) each.  Everything but two per-thread byte offsets (A side with the ring rotation, W side) is wave-uniform, so
  // the 15 loads of a k-tile are SGPR base + VGPR offset.
  const int srow = tid >> 2, sunit = tid & 3;
  const bool a_ok0 = (m0 + srow) < g.M, a_ok1 = (m0 + 64 + srow) < g.M;
  const char* Abase = reinterpret_cast<const char*>(g.A) + (size_t)m0 * g.lda * 2;
  const char* Wbase = reinterpret_cast<const char*>(g.W) + (size_t)n0 * g.ldw * 2;
  const unsigned a_row = (unsigned)srow * g.lda * 2, a_row64 = 64u * g.lda * 2;
  const unsigned w_row = (unsigned)srow * g.ldw * 2, w_row64 = 64u * g.ldw * 2;
  // load q (0..2) of a row moves bytes [64 q, 64 q + 64) of the row's 192 B: chunk c = 4 q + sunit = (unit c / 3, plane c % 3)
  const int c0 = sunit, c1 = 4 + sunit, c2 = 8 + sunit;
  const int j0 = c0 / 3, j1 = c1 / 3, j2 = c2 / 3;
  const unsigned st_q0 = ((c0 % 3) * G3_ROWS + srow) * G3_ROWB + j0 * 16;
  const unsigned st_q1 = ((c1 % 3) * G3_ROWS + srow) * G3_ROWB + j1 * 16;
  const unsigned st_q2 = ((c2 % 3) * G3_ROWS + srow) * G3_ROWB + j2 * 16;
  const int frow = lane & 31, fk = (lane >> 5) * 16;           // fragment: row, byte offset of its 8 k inside a 16-k step
  const unsigned fa_ofs = (wm * 64 + frow) * G3_ROWB + fk, fb_ofs = (G3_M + wn * 96 + frow) * G3_ROWB + fk;
  const int nkt = g.K / G3_K;                                   // even (K is a multiple of 64)
  // All blocks walk K in the same order on purpose: the 64 blocks that share a W tile then touch the same 37 kB window of
  // W at about the same time and it stays L2-resident.  (Rotating the k order per M-tile, which cured an L2-channel hot
  // spot in k_policy_tail, was measured here: same kernel time, 4x the HBM-side fetch traffic -- W no longer fits in L2.)
  const int koff = 0;
  unsigned char* buf0 = lds3;
  unsigned char* buf1 = lds3 + G3_BUF;
  const g3_u32x4 z4 = {0u, 0u, 0u, 0u};
  // two prefetch register sets [load q][row block] and two fragment sets [tile][plane]; all named scalars
  g3_u32x4 Pa00, Pa01, Pa10, Pa11, Pa20, Pa21, Pw00, Pw01, Pw02, Pw10, Pw11, Pw12, Pw20, Pw21, Pw22;
  g3_u32x4 Qa00, Qa01, Qa10, Qa11, Qa20, Qa21, Qw00, Qw01, Qw02, Qw10, Qw11, Qw12, Qw20, Qw21, Qw22;
  g3_u32x4 f0a00, f0a01, f0a02, f0a10, f0a11, f0a12, f0b00, f0b01, f0b02, f0b10, f0b11, f0b12, f0b20, f0b21, f0b22;
  g3_u32x4 f1a00, f1a01, f1a02, f1a10, f1a11, f1a12, f1b00, f1b01, f1b02, f1b10, f1b11, f1b12, f1b20, f1b21, f1b22;
  unsigned aoff0, aoff1, aoff2, woff;
#define G3_WRAP(u_) { if (u_ >= g.a_ring8) u_ -= g.a_ring8; if (u_ >= g.a_ring8) u_ -= g.a_ring8; }
#define G3_ADDR(kt_)                                                                                            \
  {                                                                                                             \
    int kc_ = (kt_); if (kc_ > nkt - 1) kc_ = nkt - 1;      /* the last trips re-request the final tile */     \
    kc_ += koff; if (kc_ >= nkt) kc_ -= nkt;                /* per-block rotation of the k order, see koff */   \
    woff = w_row + (unsigned)kc_ * 192u + (unsigned)sunit * 16u;                                                \
    int u0_ = kc_ * 4 + j0, u1_ = kc_ * 4 + j1, u2_ = kc_ * 4 + j2;                                              \
    if (g.a_ring8) { u0_ += g.a_rot8; u1_ += g.a_rot8; u2_ += g.a_rot8; G3_WRAP(u0_) G3_WRAP(u1_) G3_WRAP(u2_) } \
    aoff0 = a_row + (unsigned)(u0_ * 3 + c0 % 3) * 16u;                                                         \
    aoff1 = a_row + (unsigned)(u1_ * 3 + c1 % 3) * 16u;                                                         \
    aoff2 = a_row + (unsigned)(u2_ * 3 + c2 % 3) * 16u;                                                         \
  }
#define G3_LDA(dst, ok_, q_, i_) dst = (ok_) ? *(const g3_gvec*)(Abase + (size_t)(i_) * a_row64 + aoff##q_) : z4;
#define G3_LDW(dst, q_, i_) dst = *(const g3_gvec*)(Wbase + (size_t)(i_) * w_row64 + (q_) * 64 + woff);
#define G3_ST(buf_, src_, q_, r_) *reinterpret_cast<g3_u32x4*>((buf_) + st_q##q_ + (r_) * G3_ROWB) = src_;
#define G3_RDA(buf_, p_, t_, ks_) *reinterpret_cast<const g3_u32x4*>((buf_) + fa_ofs + ((p_) * G3_ROWS + (t_) * 32) * G3_ROWB + (ks_) * 32)
#define G3_RDB(buf_, p_, u_, ks_) *reinterpret_cast<const g3_u32x4*>((buf_) + fb_ofs + ((p_) * G3_ROWS + (u_) * 32) * G3_ROWB + (ks_) * 32)
#define G3_BF(x_) __builtin_bit_cast(bf16x8, x_)
#include "kernels_gemm_b3_loop.inc"
#undef G3_ADDR
#undef G3_WRAP
#undef G3_LDA
#undef G3_LDW
#undef G3_ST
#undef G3_RDA
#undef G3_RDB
#undef G3_BF
#define G3_EPI(acc_, t_, u_)                                                                                    \
  {                                                                                                             \
    const int col = n0 + wn * 96 + (u_) * 32 + (lane & 31);                                                     \
    const float bias = g.bias ? g.bias[col] : 0.0f;                                                             \
    const bool do_act = col < g.act_cols;                                                                       \
    _Pragma("unroll") for (int r = 0; r < 16; r++) {                                                            \
      const int row = m0 + wm * 64 + (t_) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);                      \
      if (row < g.M) {                                                                                          \
        float v = acc_[r] + bias;                                                                               \
        if (do_act) v = v > 0 ? v : expm1f(v);                                                                  \
        g.C[(size_t)row * g.ldc + col] = v;                                                                     \
      }                                                                                                         \
    }                                                                                                           \
  }
  G3_EPI(acc00, 0, 0) G3_EPI(acc01, 0, 1) G3_EPI(acc02, 0, 2) G3_EPI(acc10, 1, 0) G3_EPI(acc11, 1, 1) G3_EPI(acc12, 1, 2)
#undef G3_EPI
}
