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
