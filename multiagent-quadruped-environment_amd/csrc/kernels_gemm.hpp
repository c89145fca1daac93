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
// k_gemm_h2: the same product on the f16 matrix cores with f32-class accuracy ("split-f16", 2 planes, 3 terms).
//
// Every f32 operand x is carried as two f16 planes of c x (c = a power of two that centres the operand in the f16
// range): h = rne(c x), l = rne(c x - h), h + l = c x to 22 significand bits.  A product a*b is the three f16 x f16 terms
// hh + hl + lh (the dropped ll is <= 2^-22 |ab|), each exact in the MFMA, accumulated in f32 and rescaled by the exact
// 1 / (c_a c_w) in the epilogue: per-product error <= 3 * 2^-22 against the 2^-24 of an f32 multiply, i.e. below the
// rounding noise of the 2100-term f32 sum itself (tests: |diff| <= 5e-5 against the oracle's fmaf chain on O(1) outputs).
// Three v_mfma_f32_32x32x16_f16 replace eight v_mfma_f32_32x32x2_f32 per 16 k: 16x the rate per instruction, 5.3x net.
//   * scaling: activations c_a = 64 (|x| <= 1023 representable, the observation terms are O(1)..O(10); larger magnitudes
//     saturate), weights c_w = 2^floor(log2(32768 / max|w|)) per layer; residuals below the f16 normal range
//     (|c x| < 6e-5 * 2^11) keep an absolute error of 2^-25 / c, far below the terms' own rounding.
//   * producers write the planes (interleaved per 8-element unit, see Gemm2Args): k_pre_policy (new history frame);
//     weights are split once on the host.
//   * block 128 x 192 (M = 8192, N = 768: exactly 256 blocks = one per CU), 8 waves = 2 per SIMD with separate roles
//     (tools/gen_gemm_h2.py emits the loop): waves 0-3 multiply (wave tile 64 x 96 = 6 accumulators, 0.55 fragment reads
//     per MFMA, no vmcnt wait anywhere in their stream), waves 4-7 stage (global loads 3 k-tiles ahead into registers,
//     LDS stores 2 tiles ahead into a ring of three 50 KB buffers); one barrier per k-tile.
//   * what bounds it (standalone harness tools/dev/gemm_bench.hip, R = 8192, per launch): MFMAs alone 33 us + 12 us of
//     prologue / epilogue; the global -> register stream alone 35 us (2.8 MB per CU = 80 GB/s per CU, the tile shape
//     already minimises it); LDS stores and fragment reads slow each other down (1.6x the cycles of either alone); and
//     with the loads running the shader clock drops from 2.4 to ~1.65 GHz (same cycle count, 1.5x the time).  Measured
//     structures: every wave staging and multiplying, 4 waves 111 us (= the sum of the parts: each wait of the single
//     instruction stream idles the SIMD's matrix pipe), 8 waves 94 us; roles split 93 us; roles split with direct
//     global -> LDS loads (global_load_lds_dwordx4, 4 x 40 KB ring, XOR-swizzled rows) 101 us: the LDS-DMA path took
//     ~200 cycles per instruction, 20 B/clk per CU.
//   * LDS row = 32 k of one plane padded to 80 B, so the 16 B fragment reads of a ds_read_b128 lane group (16 rows)
//     tile all 64 banks; the planes are 64 B apart modulo the 128 B store banking, so the 8 lanes that stage one row's
//     128 contiguous global bytes (4 units x 2 planes) store conflict-free as well.
//   * A may be the history ring: 8-element units, logical unit u -> (u + rot) mod ring (compact frames of 64 = 8 units, mqe_common.hpp).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// explicit global address space: pointers selected between two kernel-argument buffers degrade to flat loads, whose
// completion is also counted by lgkmcnt, i.e. every LDS wait would wait for the prefetch as well
typedef unsigned int h2_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) h2_u32x4 h2_gvec;

#define H2_M 128
#define H2_N 192
#define H2_K 32
#define H2_KMULT 96                                 // K must be a multiple of three k-tiles (the loop is unrolled over the ring)
#define H2_THREADS 512
#define H2_ROWB 80                                  // bytes per LDS row (64 B of data + 16 B pad)
#define H2_ROWS (H2_M + H2_N)
#define H2_PLANE (H2_ROWS * H2_ROWB + 64)           // one plane of a buffer: [A rows | W rows][80 B] (+ 64 B: see above)
#define H2_BUF (2 * H2_PLANE)
#define H2_LDS_BYTES (3 * H2_BUF)
#define H2_EPS 196                                  // epilogue staging: floats per row of the 128 x 192 block tile (192 + 4)
static_assert(H2_M * H2_EPS * 4 <= H2_LDS_BYTES, "epilogue staging");

struct Gemm2Args {
  // operands are plane-interleaved per 8-element unit: row r, element k of plane p at r * ld + ((k / 8) * 2 + p) * 8 + k % 8
  // (ld = 2 * K elements), so the 32 k x 2 planes a row contributes to a k-tile are one 128 B line
  const uint16_t* A; int lda; int a_rot8, a_ring8;
  const uint16_t* W; int ldw;                          // W rows = output units (the (out,in) layout), same interleaving
  const float* bias;
  float* C; int ldc;
  int M, N, K;                                         // N multiple of 192, K multiple of 96
  int act_cols;                                        // multiple of 4
  float descale;                                       // 1 / (c_a c_w)
  // rows whose compact history has frames that do not continue their predecessor (mqe_common.hpp, MQE_H2_FRAME): bit s of irr[row]
  // = ring slot s; the epilogue adds W[., frame p, 54..65] (a2(p) - a1(p - 1)) from the f32 ring and the f32 weights for the frames
  // at logical positions p >= 1 (ring_pos = slot of the oldest frame).  irr may be null.
  const unsigned* irr; const float* ring; int ring_pos;
  const float* Wt32; int ldwt;                         // [frame * MQE_FRAME + column][ldwt] (GemmLayer::Wt)
  int full_blocks, full_rows;                          // blocks that run full 128-row tiles and the rows they cover; the rest: half tiles
  long long* times; int times_blocks;                  // debug (MQE_TAIL_TIMES=1, tools/dev/tail_times.py): slots 10..15 of the tail's [blocks][16] stamp table
                                                       // = wall clock (100 MHz) at entry / end of the K loop / exit, then the shader clock at the same points
};
#define H2_STAMP(i) do { if (g.times != nullptr && threadIdx.x == 0 && (int)blockIdx.x < g.times_blocks) {                          \
    g.times[(size_t)blockIdx.x * 16 + 10 + (i)] = (long long)wall_clock64(); g.times[(size_t)blockIdx.x * 16 + 13 + (i)] = (long long)clock64(); } } while (0)
// One block tile: TA = 2 the full 128 x 192 tile, TA = 1 the HALF tile of 64 x 192 (multiplier waves 2 x 2 on 32 x 96 wave tiles, two
// staging blocks of A; same LDS layout, same W tile, same epilogue) for the rows that do not fill a round of full tiles -- a round is
// 256 / (N / 192) M-tiles = 8192 rows of layer 0, and a started round costs the same whether one tile or 256 of them run in it.
template <int TA>
__device__ __forceinline__ void gemm_h2_tile(const Gemm2Args& g, const int m0, const int n0, unsigned char* lds2) {
  constexpr int HM = 64 * TA;                                   // rows of this block tile
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (wave >> 1) & 1, wn = wave & 1;               // multiplier waves 0..3 as 2 x 2
  f32x16 acc00, acc01, acc02, acc10, acc11, acc12;
  H2_STAMP(0);
#pragma unroll
  for (int i = 0; i < 16; i++) { acc00[i] = 0.0f; acc01[i] = 0.0f; acc02[i] = 0.0f; acc10[i] = 0.0f; acc11[i] = 0.0f; acc12[i] = 0.0f; }
  // staging (waves 4..7, 256 threads): 8 consecutive threads move the 128 B a row contributes to a k-tile (chunk c = unit
  // c / 2, plane c % 2), 32 rows per instruction: 4 instructions for the A tile, 6 for the W tile.  Rows past M re-read
  // row M - 1 (never stored) so that the loads are unconditional.
  const int stid = tid & 255, srow = stid >> 3, sc = stid & 7, sj = sc >> 1, sp = sc & 1;
  const char* Abase = reinterpret_cast<const char*>(g.A);
  const char* Wbase = reinterpret_cast<const char*>(g.W) + (size_t)n0 * g.ldw * 2;
  const unsigned a_row0 = (unsigned)min(m0 + srow, g.M - 1) * g.lda * 2, a_row1 = (unsigned)min(m0 + 32 + srow, g.M - 1) * g.lda * 2;
  const unsigned a_row2 = TA == 2 ? (unsigned)min(m0 + 64 + srow, g.M - 1) * g.lda * 2 : 0u, a_row3 = TA == 2 ? (unsigned)min(m0 + 96 + srow, g.M - 1) * g.lda * 2 : 0u;
  (void)a_row2; (void)a_row3;
  const unsigned w_row = (unsigned)srow * g.ldw * 2, w_row32 = 32u * g.ldw * 2;
  const unsigned st_ofs = sp * H2_PLANE + srow * H2_ROWB + sj * 16;
  const int frow = lane & 31, fk = (lane >> 5) * 16;           // fragment: row, byte offset of its 8 k inside a 16-k step
  const unsigned fa_ofs = (wm * 32 * TA + frow) * H2_ROWB + fk, fb_ofs = (H2_M + wn * 96 + frow) * H2_ROWB + fk;
  const int nkt = g.K / H2_K;                                   // multiple of 3
  // All blocks walk K in the same order on purpose: the 64 blocks that share a W tile then touch the same window of W at
  // about the same time and it stays L2-resident.  (Rotating the k order per M-tile, which cured an L2-channel hot spot
  // in k_policy_tail, was measured here: same kernel time, 4x the HBM-side fetch traffic -- W no longer fits in L2.)
  unsigned char* buf0 = lds2;
  unsigned char* buf1 = lds2 + H2_BUF;
  unsigned char* buf2 = lds2 + 2 * H2_BUF;
  // three prefetch register sets [row block] (staging waves) and two fragment sets [tile][plane] (multiplier waves); all
  // named scalars: arrays carried across the loop end up in scratch memory
  h2_u32x4 Pa0, Pa1, Pa2, Pa3, Pw0, Pw1, Pw2, Pw3, Pw4, Pw5;
  h2_u32x4 Qa0, Qa1, Qa2, Qa3, Qw0, Qw1, Qw2, Qw3, Qw4, Qw5;
  h2_u32x4 Ra0, Ra1, Ra2, Ra3, Rw0, Rw1, Rw2, Rw3, Rw4, Rw5;
  h2_u32x4 f0a00, f0a01, f0a10, f0a11, f0b00, f0b01, f0b10, f0b11, f0b20, f0b21;
  h2_u32x4 f1a00, f1a01, f1a10, f1a11, f1b00, f1b01, f1b10, f1b11, f1b20, f1b21;
  unsigned aoff, woff;
#define H2_ADDR(kt_)                                                                                            \
  {                                                                                                             \
    int kc_ = (kt_); if (kc_ > nkt - 1) kc_ = nkt - 1;      /* the last trips re-request the final tile */     \
    woff = w_row + (unsigned)kc_ * 128u + (unsigned)sc * 16u;                                                   \
    int u_ = kc_ * 4 + sj;                                                                                      \
    if (g.a_ring8) { u_ += g.a_rot8; if (u_ >= g.a_ring8) u_ -= g.a_ring8; if (u_ >= g.a_ring8) u_ -= g.a_ring8; } \
    aoff = (unsigned)(u_ * 2 + sp) * 16u;                                                                       \
  }
  // (tools/dev/gemm_bench.hip may predefine H2_LDA / H2_LDW / H2_ST / H2_MFMA to take one of the loop's streams out: what bounds the loop)
#ifndef H2_LDA
#define H2_LDA(dst, i_) dst = *(const h2_gvec*)(Abase + a_row##i_ + aoff);
#endif
#ifndef H2_LDW
#define H2_LDW(dst, i_) dst = *(const h2_gvec*)(Wbase + (size_t)(i_) * w_row32 + woff);
#endif
#ifndef H2_ST
#define H2_ST(buf_, src_, r_) *reinterpret_cast<h2_u32x4*>((buf_) + st_ofs + (r_) * H2_ROWB) = src_;
#endif
#define H2_RDA(buf_, p_, t_, ks_) *reinterpret_cast<const h2_u32x4*>((buf_) + fa_ofs + (p_) * H2_PLANE + (t_) * 32 * H2_ROWB + (ks_) * 32)
#define H2_RDB(buf_, p_, u_, ks_) *reinterpret_cast<const h2_u32x4*>((buf_) + fb_ofs + (p_) * H2_PLANE + (u_) * 32 * H2_ROWB + (ks_) * 32)
#define H2_F16(x_) __builtin_bit_cast(f16x8, x_)
#ifndef H2_MFMA
#define H2_MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x16_f16(H2_F16(a_), H2_F16(b_), c_, 0, 0, 0)
#endif
#define H2_PIN() __builtin_amdgcn_sched_barrier(0)
  // dev instrumentation of the loop (tools/dev/gemm_bench.hip -DH2_STAGER_TIMES / -DH2_MULT_TIMES): shader-clock sums between the hooks the
  // generator leaves in the loop, written to slots 0..5 of the stamp table; empty in the product
#if defined(H2_STAGER_TIMES) || defined(H2_MULT_TIMES)
  long long ts_prev = clock64(), ts_sum0 = 0, ts_sum1 = 0, ts_sum2 = 0, ts_sum3 = 0, tm_sum0 = 0, tm_sum1 = 0;
#endif
#ifdef H2_STAGER_TIMES
#define H2_TS(n) { const long long c_ = clock64(); ts_sum##n += c_ - ts_prev; ts_prev = c_; }
#define H2_TS_WAITV() asm volatile("s_waitcnt vmcnt(20)" ::: "memory")
#else
#define H2_TS(n)
#define H2_TS_WAITV()
#endif
#ifdef H2_MULT_TIMES
#define H2_TM(n) { const long long c_ = clock64(); tm_sum##n += c_ - ts_prev; ts_prev = c_; }
#else
#define H2_TM(n)
#endif
  if constexpr (TA == 2) {
#include "kernels_gemm_h2_loop.inc"
  } else {
#include "kernels_gemm_h2_loop_half.inc"
  }
#if defined(H2_STAGER_TIMES) || defined(H2_MULT_TIMES)
  if (g.times != nullptr && lane == 0 && (int)blockIdx.x < g.times_blocks) {
    long long* o = g.times + (size_t)blockIdx.x * 16;
    if (wave == 4) { o[0] = ts_sum0; o[1] = ts_sum1; o[2] = ts_sum2; o[3] = ts_sum3; }
    if (wave == 0) { o[4] = tm_sum0; o[5] = tm_sum1; }
  }
#endif
#undef H2_TS
#undef H2_TS_WAITV
#undef H2_TM
#undef H2_ADDR
#undef H2_LDA
#undef H2_LDW
#undef H2_ST
#undef H2_RDA
#undef H2_RDB
#undef H2_F16
#undef H2_MFMA
#undef H2_PIN
  // epilogue: the accumulators hold 4-row column slivers, so they go through the (now idle) LDS once and leave as 16 B per
  // lane = whole row segments, stored by all 8 waves (dword stores of 128 B half-rows measured 13 us for the 25 MB)
  H2_STAMP(1);
  __syncthreads();
  float* ep = reinterpret_cast<float*>(lds2);
#define H2_EPW(acc_, t_, u_)                                                                                    \
  _Pragma("unroll") for (int r = 0; r < 16; r++)                                                                \
    ep[(wm * 32 * TA + (t_) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * H2_EPS + wn * 96 + (u_) * 32 + (lane & 31)] = acc_[r];
  if (wave < 4) {
    H2_EPW(acc00, 0, 0) H2_EPW(acc01, 0, 1) H2_EPW(acc02, 0, 2)
    if constexpr (TA == 2) { H2_EPW(acc10, 1, 0) H2_EPW(acc11, 1, 1) H2_EPW(acc12, 1, 2) }
  }
#undef H2_EPW
  __syncthreads();
  // every bias segment and row mask of this thread's twelve (six) output segments is requested before the first segment is stored: with the
  // loads inside the store loop each iteration waited for its loads BEHIND the previous iteration's store (one counter for both on gfx9):
  // twelve store acknowledgements one after the other, 5.4 us of a 52 us kernel
  constexpr int EP_IT = (HM * H2_N / 4) / H2_THREADS;
  float4 bbs[EP_IT];
  unsigned mks[EP_IT];
#pragma unroll
  for (int it = 0; it < EP_IT; it++) {
    const int i = it * H2_THREADS + tid, row = i / (H2_N / 4), c4 = i - row * (H2_N / 4);
    bbs[it] = g.bias ? *reinterpret_cast<const float4*>(g.bias + n0 + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    mks[it] = g.irr && m0 + row < g.M ? g.irr[m0 + row] : 0u;
  }
#pragma unroll
  for (int it = 0; it < EP_IT; it++) {
    const int i = it * H2_THREADS + tid, row = i / (H2_N / 4), c4 = i - row * (H2_N / 4);
    const int col = n0 + c4 * 4, grow = m0 + row;
    float4 v = *reinterpret_cast<const float4*>(ep + row * H2_EPS + c4 * 4);
    const float4 bb = bbs[it];
    v.x = fmaf(v.x, g.descale, bb.x); v.y = fmaf(v.y, g.descale, bb.y); v.z = fmaf(v.z, g.descale, bb.z); v.w = fmaf(v.w, g.descale, bb.w);
    unsigned mk = mks[it];
    if (mk) {                     // rare: the first frame after a reset is still in the ring (30 steps per episode)
      float4 cr = make_float4(0.f, 0.f, 0.f, 0.f);
      do {
        const int s1 = __ffs((int)mk) - 1;
        mk &= mk - 1;
        const int p = s1 >= g.ring_pos ? s1 - g.ring_pos : s1 + MQE_HIST - g.ring_pos;
        if (p == 0) continue;                  // the oldest frame has no partner by construction: carrier columns
        const int s0 = s1 > 0 ? s1 - 1 : MQE_HIST - 1;
        const float* a2 = g.ring + ((size_t)grow * MQE_HIST + s1) * MQE_FRAME + 54;
        const float* a1 = g.ring + ((size_t)grow * MQE_HIST + s0) * MQE_FRAME + 42;
        const float* wr = g.Wt32 + (size_t)(p * MQE_FRAME + 54) * g.ldwt + col;
#pragma unroll 4
        for (int j = 0; j < 12; j++) {
          const float r = a2[j] - a1[j];
          const float4 w = *reinterpret_cast<const float4*>(wr + (size_t)j * g.ldwt);
          cr.x = fmaf(r, w.x, cr.x); cr.y = fmaf(r, w.y, cr.y); cr.z = fmaf(r, w.z, cr.z); cr.w = fmaf(r, w.w, cr.w);
        }
      } while (mk);
      v.x += cr.x; v.y += cr.y; v.z += cr.z; v.w += cr.w;
    }
    if (col < g.act_cols) {       // ELU
      v.x = v.x > 0 ? v.x : __expf(v.x) - 1.0f; v.y = v.y > 0 ? v.y : __expf(v.y) - 1.0f;
      v.z = v.z > 0 ? v.z : __expf(v.z) - 1.0f; v.w = v.w > 0 ? v.w : __expf(v.w) - 1.0f;
    }
    if (grow < g.M) {
      // streaming store: the 25 MB of C are read next by another kernel on (mostly) another XCD, i.e. through memory whatever this L2 keeps; stored
      // plainly they sit dirty in the eight L2s until the dispatch's release writes them back.  Back to back in tools/dev/gemm_bench.hip 58.8 ->
      // 57.3 us per launch (write-through sc0 sc1: 57.5), in the step +0.5 % (3 / 3 alternating runs, profiles/r06_ab_gemm_nt_store.txt);
      // -DH2_PLAIN_STORE keeps the plain form.  tools/dev/gap_bench.hip: what a dispatch costs beyond its wavefronts' run time, by dirty bytes.
#ifdef H2_PLAIN_STORE
      *reinterpret_cast<float4*>(g.C + (size_t)grow * g.ldc + col) = v;
#else
      typedef float h2_f4 __attribute__((ext_vector_type(4)));
      h2_f4 nv; nv.x = v.x; nv.y = v.y; nv.z = v.z; nv.w = v.w;
      __builtin_nontemporal_store(nv, reinterpret_cast<h2_f4*>(g.C + (size_t)grow * g.ldc + col));
#endif
    }
  }
  H2_STAMP(2);
}
// k_gemm_h2: full tiles only (what the headline batch runs: 8192 rows = exactly one round).  k_gemm_h2_mix: blocks [0, full_blocks) run
// full tiles over the rows [0, full_rows), the others half tiles over the rows behind them (launch_gemm2 picks the split: whole rounds
// of full tiles, and a remainder of at most half a round as half tiles, which then fill twice the CUs at 0.63 of a full tile's time).
// Two kernels because the one with both tile bodies allocates twice the registers and its full-tile path measured ~1 % slower.
// The XCD-aware numbering is applied inside each part.
__global__ void __launch_bounds__(H2_THREADS, 1) k_gemm_h2(Gemm2Args g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds2[];
  const int ntn = g.N / H2_N, ntm = (g.M + H2_M - 1) / H2_M;
  int bid = blockIdx.x;
  const int total = ntn * ntm;
  if ((total & 7) == 0) { const int xcd = bid & 7, slot = bid >> 3; bid = xcd * (total >> 3) + slot; }
  const int tm = bid / ntn, tn = bid - tm * ntn;
  gemm_h2_tile<2>(g, tm * H2_M, tn * H2_N, lds2);
}
__global__ void __launch_bounds__(H2_THREADS, 1) k_gemm_h2_mix(Gemm2Args g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds2[];
  const int ntn = g.N / H2_N;
  int bid = blockIdx.x;
  if (bid < g.full_blocks) {
    const int total = g.full_blocks;
    if ((total & 7) == 0) { const int xcd = bid & 7, slot = bid >> 3; bid = xcd * (total >> 3) + slot; }
    const int tm = bid / ntn, tn = bid - tm * ntn;
    gemm_h2_tile<2>(g, tm * H2_M, tn * H2_N, lds2);
  } else {
    bid -= g.full_blocks;
    const int total = (int)gridDim.x - g.full_blocks;
    if ((total & 7) == 0) { const int xcd = bid & 7, slot = bid >> 3; bid = xcd * (total >> 3) + slot; }
    const int tm = bid / ntn, tn = bid - tm * ntn;
    gemm_h2_tile<1>(g, g.full_rows + tm * (H2_M / 2), tn * H2_N, lds2);
  }
}
