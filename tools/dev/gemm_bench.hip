// Standalone timing / correctness harness for k_gemm_h2 (development tool, not part of the product or the tests).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DH2_...] tools/dev/gemm_bench.hip -o gemm_bench && ./gemm_bench
#include "../../multiagent-quadruped-environment_amd/csrc/kernels_gemm.hpp"
#ifdef H2_KSPLIT
#include "gemm_ksplit.hpp"          // the K-split wave tile (128 x 96 per multiplier wave), full tiles only
#endif
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 8192, N = 768, K = argc > 7 ? atoi(argv[7]) : 1440 /* multiple of 96 */, ring8 = K / 8, rot8 = (9 * 7) % ring8;
  const int KU = K;          // columns that carry data
  const int reps = argc > 2 ? atoi(argv[2]) : 200;
  const int lda = argc > 3 ? atoi(argv[3]) : 2 * K;
  const int ldc = argc > 5 ? atoi(argv[5]) : N;
  const int ldw = argc > 6 ? atoi(argv[6]) : 2 * K;
  std::vector<float> A((size_t)M * K), W((size_t)N * K), bias(N);
  srand(1);
  auto rnd = []() { return (float)rand() / RAND_MAX * 2.0f - 1.0f; };
  const char* dat = getenv("GEMM_DATA") ? getenv("GEMM_DATA") : "random";       // random | zeroA | zero | const: how much of the time is the data's switching activity
  for (auto& v : A) v = rnd() * 3.0f;
  for (auto& v : W) v = rnd() * 0.05f;
  for (auto& v : bias) v = rnd();
  if (!strcmp(dat, "zeroA") || !strcmp(dat, "zero")) for (auto& v : A) v = 0.0f;
  if (!strcmp(dat, "zero")) for (auto& v : W) v = 0.0f;
  if (!strcmp(dat, "const")) { for (auto& v : A) v = 1.2345678f; for (auto& v : W) v = 0.0123456f; }
  for (int r = 0; r < M; r++) for (int k = KU; k < K; k++) A[(size_t)r * K + k] = 0.0f;
  for (int r = 0; r < N; r++) for (int k = KU; k < K; k++) W[(size_t)r * K + k] = 0.0f;
  const float wscale = 262144.0f;      // 32768 / 0.05 -> 2^18 = 262144 (0.05 * 2^18 = 13107)
  std::vector<uint16_t> A2((size_t)M * lda), W2((size_t)N * ldw);
  // ring: logical unit u lives at physical unit (u + rot) % ring for u < ring; pad units stay in place
  for (int r = 0; r < M; r++) for (int k = 0; k < K; k++) {
    int u = k / 8, pu = u < ring8 ? (u + rot8) % ring8 : u;
    size_t pk = (size_t)pu * 8 + k % 8;
    uint16_t h, l; split2(A[(size_t)r * K + k], MQE_H2_ASCALE, h, l);
    if (u >= ring8) continue;   // the pad units alias ring data (zero weights)
    A2[(size_t)r * lda + h2_index(pk, 0)] = h; A2[(size_t)r * lda + h2_index(pk, 1)] = l;
  }
  for (int r = 0; r < N; r++) for (int k = 0; k < K; k++) {
    uint16_t h, l; split2(W[(size_t)r * K + k], wscale, h, l);
    W2[(size_t)r * ldw + h2_index(k, 0)] = h; W2[(size_t)r * ldw + h2_index(k, 1)] = l;
  }
  uint16_t *dA, *dW; float *dB, *dC;
  CK(hipMalloc(&dA, A2.size() * 2)); CK(hipMalloc(&dW, W2.size() * 2)); CK(hipMalloc(&dB, N * 4)); CK(hipMalloc(&dC, (size_t)M * ldc * 4));
  CK(hipMemcpy(dA, A2.data(), A2.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, W2.data(), W2.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, bias.data(), N * 4, hipMemcpyHostToDevice));
  Gemm2Args g;
  g.A = dA; g.lda = lda; g.a_rot8 = rot8; g.a_ring8 = ring8; g.W = dW; g.ldw = ldw; g.bias = dB; g.C = dC; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
  g.act_cols = argc > 4 ? atoi(argv[4]) : 256; g.descale = 1.0f / (MQE_H2_ASCALE * wscale);
  g.times = nullptr; g.times_blocks = 0;
  g.irr = nullptr; g.ring = nullptr; g.ring_pos = 0; g.Wt32 = nullptr; g.ldwt = 0;      // no compact-history residuals in the harness
  CK(hipFuncSetAttribute((const void*)k_gemm_h2, hipFuncAttributeMaxDynamicSharedMemorySize, H2_LDS_BYTES));
  CK(hipFuncSetAttribute((const void*)k_gemm_h2_mix, hipFuncAttributeMaxDynamicSharedMemorySize, H2_LDS_BYTES));
  // GEMM_MODE: "half" = every row in half tiles (64 x 192), "mix" = whole rounds of full tiles + a remainder of at most half a round as half tiles
  // (the engine's rule, mqe_engine.hip::h2_tiling), default = full tiles only
  const char* mode = getenv("GEMM_MODE") ? getenv("GEMM_MODE") : "full";
  const int ntn = N / H2_N, ntm = (M + H2_M - 1) / H2_M, per_round = 256 / ntn;
  int full_tiles = ntm, half_tiles = 0;
  if (!strcmp(mode, "half")) { full_tiles = 0; half_tiles = (M + 63) / 64; }
  else if (!strcmp(mode, "mix") && ntm % per_round != 0 && 2 * (ntm % per_round) <= per_round) {
    full_tiles = ntm - ntm % per_round; half_tiles = (M - full_tiles * H2_M + 63) / 64;
  }
  g.full_blocks = full_tiles * ntn; g.full_rows = full_tiles * H2_M;
  const int grid = (full_tiles + half_tiles) * ntn;
  printf("mode %s: %d full + %d half M-tiles\n", mode, full_tiles, half_tiles);
#ifdef H2_KSPLIT
  auto KERNEL = k_gemm_h2_ks;
  CK(hipFuncSetAttribute((const void*)k_gemm_h2_ks, hipFuncAttributeMaxDynamicSharedMemorySize, H2_LDS_BYTES));
#else
  auto KERNEL = half_tiles ? k_gemm_h2_mix : k_gemm_h2;
#endif
  hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(H2_THREADS), H2_LDS_BYTES, 0, g);
  CK(hipDeviceSynchronize());
  std::vector<float> C((size_t)M * ldc);
  CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
  double maxerr = 0;
  for (int t = 0; t < 4000; t++) {
    int r = rand() % M, c = rand() % N;
    if (t < 8) { r = t < 4 ? t * 37 % M : M - 1 - t; c = (t * 101) % N; }
    double acc = 0;
    for (int k = 0; k < KU; k++) acc += (double)A[(size_t)r * K + k] * (double)W[(size_t)c * K + k];
    acc += bias[c];
    if (c < 256) acc = acc > 0 ? acc : std::expm1(acc);
    maxerr = std::max(maxerr, std::fabs(acc - (double)C[(size_t)r * ldc + c]));
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; i++) hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(H2_THREADS), H2_LDS_BYTES, 0, g);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(H2_THREADS), H2_LDS_BYTES, 0, g);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps;
  {   // where the time goes: wall clock (100 MHz) and shader clock at entry / end of the K loop / exit of every workgroup, last of 5 stamped launches
    long long* dT; CK(hipMalloc(&dT, (size_t)grid * 16 * 8)); CK(hipMemset(dT, 0, (size_t)grid * 16 * 8));
    g.times = dT; g.times_blocks = grid;
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(H2_THREADS), H2_LDS_BYTES, 0, g);
    CK(hipDeviceSynchronize());
    std::vector<long long> T((size_t)grid * 16); CK(hipMemcpy(T.data(), dT, T.size() * 8, hipMemcpyDeviceToHost));
    double w01 = 0, w12 = 0, c01 = 0, c12 = 0; long long first = T[10], last = T[12], lastin = T[10];
    for (int b = 0; b < grid; b++) { const long long* t = &T[(size_t)b * 16 + 10];
      w01 += (t[1] - t[0]) * 0.01; w12 += (t[2] - t[1]) * 0.01; c01 += (double)(t[4] - t[3]); c12 += (double)(t[5] - t[4]);
      first = std::min(first, t[0]); last = std::max(last, t[2]); lastin = std::max(lastin, t[0]); }
    w01 /= grid; w12 /= grid; c01 /= grid; c12 /= grid;
    printf("stamps (mean over %d workgroups): K loop %.2f us = %.0f shader-clock ticks (%.3f GHz), epilogue %.2f us = %.0f ticks (%.3f GHz); first entry .. last exit %.2f us, entries spread over %.2f us\n",
           grid, w01, c01, c01 / w01 * 1e-3, w12, c12, c12 / w12 * 1e-3, (last - first) * 0.01, (lastin - first) * 0.01);
#if defined(H2_STAGER_TIMES) || defined(H2_MULT_TIMES)
    double q[6] = {0, 0, 0, 0, 0, 0};
    for (int b = 0; b < grid; b++) for (int i = 0; i < 6; i++) q[i] += (double)T[(size_t)b * 16 + i] / grid / (K / 32);
    printf("per k-tile, shader-clock ticks -- stager (wave 4): at the barrier %.0f, waiting for its loads %.0f, 10 LDS stores (to completion) %.0f, address + 10 loads issued %.0f;"
           " multiplier (wave 0): reads + MFMAs %.0f, at the barrier %.0f\n", q[0], q[1], q[2], q[3], q[4], q[5]);
#endif
    g.times = nullptr;
  }
  printf("K=%d ldw=%d lda=%d M=%d grid=%d  %.2f us/launch  %.1f TF (3-term f16)  %.1f TF f32-equivalent  max|err| = %.3e\n", K, ldw, lda, M, grid, us,
         3 * 2.0 * M * N * K / us * 1e-6, 2.0 * M * N * (double)K / us * 1e-6, maxerr);
  return 0;
}
