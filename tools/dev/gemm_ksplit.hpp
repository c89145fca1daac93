// gemm_ksplit.hpp -- the K-SPLIT wave tile of k_gemm_h2 (development variant, built and measured by tools/dev/gemm_bench.hip
// -DH2_KSPLIT).  Block tile, LDS ring, staging waves, operand layout and epilogue store as in
// csrc/kernels_gemm.hpp::gemm_h2_tile<2>; the multiplier waves are 2 (N halves) x 2 (the two 16-k steps of a k-tile) on 128 x 96
// wave tiles: 14 fragment reads per 36 MFMAs instead of 20 (-30 % LDS read bytes, -20 % of the loop's LDS bytes with the stores).
// Loop: tools/dev/gen_gemm_ksplit.py -> gemm_ksplit_loop.inc.  The two k halves' partial sums are added in the epilogue's LDS
// staging (k half 0 writes, barrier, k half 1 adds), i.e. the f32 summation order differs from the product's (not bit-identical).
#pragma once
#include "../../multiagent-quadruped-environment_amd/csrc/kernels_gemm.hpp"

__device__ __forceinline__ void gemm_h2_tile_ks(const Gemm2Args& g, const int m0, const int n0, unsigned char* lds2) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = (wave >> 1) & 1, wn = wave & 1;               // multiplier waves 0..3: k half x N half
  f32x16 acc00, acc01, acc02, acc10, acc11, acc12, acc20, acc21, acc22, acc30, acc31, acc32;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    acc00[i] = 0.0f; acc01[i] = 0.0f; acc02[i] = 0.0f; acc10[i] = 0.0f; acc11[i] = 0.0f; acc12[i] = 0.0f;
    acc20[i] = 0.0f; acc21[i] = 0.0f; acc22[i] = 0.0f; acc30[i] = 0.0f; acc31[i] = 0.0f; acc32[i] = 0.0f;
  }
  const int stid = tid & 255, srow = stid >> 3, sc = stid & 7, sj = sc >> 1, sp = sc & 1;
  const char* Abase = reinterpret_cast<const char*>(g.A);
  const char* Wbase = reinterpret_cast<const char*>(g.W) + (size_t)n0 * g.ldw * 2;
  const unsigned a_row0 = (unsigned)min(m0 + srow, g.M - 1) * g.lda * 2, a_row1 = (unsigned)min(m0 + 32 + srow, g.M - 1) * g.lda * 2;
  const unsigned a_row2 = (unsigned)min(m0 + 64 + srow, g.M - 1) * g.lda * 2, a_row3 = (unsigned)min(m0 + 96 + srow, g.M - 1) * g.lda * 2;
  const unsigned w_row = (unsigned)srow * g.ldw * 2, w_row32 = 32u * g.ldw * 2;
  const unsigned st_ofs = sp * H2_PLANE + srow * H2_ROWB + sj * 16;
  const int frow = lane & 31, fk = (lane >> 5) * 16 + wk * 32;  // fragment: row, byte offset of its 8 k inside this wave's 16-k step
  const unsigned fa_ofs = frow * H2_ROWB + fk, fb_ofs = (H2_M + wn * 96 + frow) * H2_ROWB + fk;
  const int nkt = g.K / H2_K;                                   // multiple of 3
  unsigned char* buf0 = lds2;
  unsigned char* buf1 = lds2 + H2_BUF;
  unsigned char* buf2 = lds2 + 2 * H2_BUF;
  h2_u32x4 Pa0, Pa1, Pa2, Pa3, Pw0, Pw1, Pw2, Pw3, Pw4, Pw5;
  h2_u32x4 Qa0, Qa1, Qa2, Qa3, Qw0, Qw1, Qw2, Qw3, Qw4, Qw5;
  h2_u32x4 Ra0, Ra1, Ra2, Ra3, Rw0, Rw1, Rw2, Rw3, Rw4, Rw5;
  h2_u32x4 fa00, fa01, fa10, fa11, fa20, fa21, fa30, fa31, fb00, fb01, fb10, fb11, fb20, fb21;
  unsigned aoff, woff;
#define H2_ADDR(kt_)                                                                                            \
  {                                                                                                             \
    int kc_ = (kt_); if (kc_ > nkt - 1) kc_ = nkt - 1;                                                          \
    woff = w_row + (unsigned)kc_ * 128u + (unsigned)sc * 16u;                                                   \
    int u_ = kc_ * 4 + sj;                                                                                      \
    if (g.a_ring8) { u_ += g.a_rot8; if (u_ >= g.a_ring8) u_ -= g.a_ring8; if (u_ >= g.a_ring8) u_ -= g.a_ring8; } \
    aoff = (unsigned)(u_ * 2 + sp) * 16u;                                                                       \
  }
#define H2_LDA(dst, i_) dst = *(const h2_gvec*)(Abase + a_row##i_ + aoff);
#define H2_LDW(dst, i_) dst = *(const h2_gvec*)(Wbase + (size_t)(i_) * w_row32 + woff);
#define H2_ST(buf_, src_, r_) *reinterpret_cast<h2_u32x4*>((buf_) + st_ofs + (r_) * H2_ROWB) = src_;
#define H2_RDA(buf_, p_, t_) *reinterpret_cast<const h2_u32x4*>((buf_) + fa_ofs + (p_) * H2_PLANE + (t_) * 32 * H2_ROWB)
#define H2_RDB(buf_, p_, u_) *reinterpret_cast<const h2_u32x4*>((buf_) + fb_ofs + (p_) * H2_PLANE + (u_) * 32 * H2_ROWB)
#define H2_F16(x_) __builtin_bit_cast(f16x8, x_)
#define H2_MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x16_f16(H2_F16(a_), H2_F16(b_), c_, 0, 0, 0)
#define H2_PIN() __builtin_amdgcn_sched_barrier(0)
#ifdef H2_KS_SYNC
#define H2_BAR() __syncthreads()
#else
#define H2_BAR() asm volatile("s_barrier" ::: "memory")
#endif
  H2_STAMP(0);
#include "gemm_ksplit_loop.inc"
#undef H2_ADDR
#undef H2_LDA
#undef H2_LDW
#undef H2_ST
#undef H2_RDA
#undef H2_RDB
#undef H2_F16
#undef H2_MFMA
#undef H2_PIN
#undef H2_BAR
  H2_STAMP(1);
  __syncthreads();
  float* ep = reinterpret_cast<float*>(lds2);
#define H2_EPI(t_, u_, r) (((t_) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * H2_EPS + wn * 96 + (u_) * 32 + (lane & 31))
#define H2_EPW(acc_, t_, u_) _Pragma("unroll") for (int r = 0; r < 16; r++) ep[H2_EPI(t_, u_, r)] = acc_[r];
#define H2_EPA(acc_, t_, u_) _Pragma("unroll") for (int r = 0; r < 16; r++) ep[H2_EPI(t_, u_, r)] += acc_[r];
  // both k halves work in both phases: half 0 stores its rows 0..63 and half 1 its rows 64..127, barrier, then each adds its other
  // half onto the partner's (read + add + store: ds_add_f32 measured 34 us for this phase -- LDS float atomics run at ~1/30 of the store rate)
  if (wave < 4) {
    if (wk == 0) { H2_EPW(acc00, 0, 0) H2_EPW(acc01, 0, 1) H2_EPW(acc02, 0, 2) H2_EPW(acc10, 1, 0) H2_EPW(acc11, 1, 1) H2_EPW(acc12, 1, 2) }
    else         { H2_EPW(acc20, 2, 0) H2_EPW(acc21, 2, 1) H2_EPW(acc22, 2, 2) H2_EPW(acc30, 3, 0) H2_EPW(acc31, 3, 1) H2_EPW(acc32, 3, 2) }
  }
  __syncthreads();
  if (wave < 4) {
    if (wk == 1) { H2_EPA(acc00, 0, 0) H2_EPA(acc01, 0, 1) H2_EPA(acc02, 0, 2) H2_EPA(acc10, 1, 0) H2_EPA(acc11, 1, 1) H2_EPA(acc12, 1, 2) }
    else         { H2_EPA(acc20, 2, 0) H2_EPA(acc21, 2, 1) H2_EPA(acc22, 2, 2) H2_EPA(acc30, 3, 0) H2_EPA(acc31, 3, 1) H2_EPA(acc32, 3, 2) }
  }
#undef H2_EPA
#undef H2_EPI
#undef H2_EPW
  __syncthreads();
  constexpr int EP_IT = (H2_M * H2_N / 4) / H2_THREADS;
  float4 bbs[EP_IT];
#pragma unroll
  for (int it = 0; it < EP_IT; it++) {
    const int i = it * H2_THREADS + tid, row = i / (H2_N / 4), c4 = i - row * (H2_N / 4);
    bbs[it] = g.bias ? *reinterpret_cast<const float4*>(g.bias + n0 + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int it = 0; it < EP_IT; it++) {          // (the compact-history residual path of the product epilogue is not part of the harness)
    const int i = it * H2_THREADS + tid, row = i / (H2_N / 4), c4 = i - row * (H2_N / 4);
    const int col = n0 + c4 * 4, grow = m0 + row;
    float4 v = *reinterpret_cast<const float4*>(ep + row * H2_EPS + c4 * 4);
    const float4 bb = bbs[it];
    v.x = fmaf(v.x, g.descale, bb.x); v.y = fmaf(v.y, g.descale, bb.y); v.z = fmaf(v.z, g.descale, bb.z); v.w = fmaf(v.w, g.descale, bb.w);
    if (col < g.act_cols) {
      v.x = v.x > 0 ? v.x : __expf(v.x) - 1.0f; v.y = v.y > 0 ? v.y : __expf(v.y) - 1.0f;
      v.z = v.z > 0 ? v.z : __expf(v.z) - 1.0f; v.w = v.w > 0 ? v.w : __expf(v.w) - 1.0f;
    }
    if (grow < g.M) *reinterpret_cast<float4*>(g.C + (size_t)grow * g.ldc + col) = v;
  }
  H2_STAMP(2);
}
__global__ void __launch_bounds__(H2_THREADS, 1) k_gemm_h2_ks(Gemm2Args g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds2[];
  const int ntn = g.N / H2_N, ntm = (g.M + H2_M - 1) / H2_M;
  int bid = blockIdx.x;
  const int total = ntn * ntm;
  if ((total & 7) == 0) { const int xcd = bid & 7, slot = bid >> 3; bid = xcd * (total >> 3) + slot; }
  const int tm = bid / ntn, tn = bid - tm * ntn;
  gemm_h2_tile_ks(g, tm * H2_M, tn * H2_N, lds2);
}
