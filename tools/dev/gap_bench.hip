// gap_bench.hip -- what a dispatch costs beyond the time its wavefronts run (development tool).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dev/gap_bench.hip -o build/gap_bench && build/gap_bench
// Back-to-back launches of one kernel in one stream: (time per launch between two HIP events) - (first wavefront's entry .. last
// wavefront's exit, wall clock stamps) = what the dispatch itself costs: launch ramp, the release at its end (write-back of the dirty
// L2 lines: eight XCDs, eight L2s), the acquire at the start of the next.  Variants: how many bytes the kernel leaves dirty, and whether
// it stores them plainly, as streaming stores (nt) or write-through (sc0 sc1).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256) k_write(f4* out, size_t n4, int spin, long long* stamps) {
  const long long t0 = (long long)wall_clock64();
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  f4 v; v.x = (float)threadIdx.x; v.y = 1.f; v.z = 2.f; v.w = 3.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    if (MODE == 0) out[i] = v;
    else if (MODE == 1) __builtin_nontemporal_store(v, out + i);
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(out + i), "v"(v) : "memory");
  }
  // busy time that leaves nothing dirty: the kernel is "long" whatever it stores
  float a = v.x;
  for (int k = 0; k < spin; k++) a = __builtin_fmaf(a, 1.0001f, 0.5f);
  if (a == 12345.678f) out[0] = v;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = (long long)wall_clock64(); }
}

template <int MODE>
static void run(const char* name, f4* out, size_t bytes, int spin, long long* dS, int grid, int reps) {
  const size_t n4 = bytes / 16;
  for (int i = 0; i < 10; i++) hipLaunchKernelGGL(k_write<MODE>, dim3(grid), dim3(256), 0, 0, out, n4, spin, dS);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k_write<MODE>, dim3(grid), dim3(256), 0, 0, out, n4, spin, dS);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> S(2 * grid); CK(hipMemcpy(S.data(), dS, S.size() * 8, hipMemcpyDeviceToHost));
  long long first = S[0], last = S[1];
  for (int b = 0; b < grid; b++) { first = std::min(first, S[2 * b]); last = std::max(last, S[2 * b + 1]); }
  const double per = ms * 1e3 / reps, inside = (last - first) * 0.01;
  printf("%-14s %6.1f MB dirty, spin %6d: %7.2f us per launch, wavefronts busy %7.2f us, dispatch overhead %6.2f us\n", name, bytes / 1e6, spin, per, inside, per - inside);
}

// the shape of the dispatch: workgroup size, registers per wave (launch bounds), dynamic LDS -- nothing stored, `spin` FMAs of run time
template <int THREADS, int WAVES_PER_EU>
__global__ void __launch_bounds__(THREADS, WAVES_PER_EU) k_shape(float* out, int spin, long long* stamps) {
  extern __shared__ float lds_s[];
  const long long t0 = (long long)wall_clock64();
  float a = (float)threadIdx.x;
  for (int k = 0; k < spin; k++) a = __builtin_fmaf(a, 1.0001f, 0.5f);
  if (a == 12345.678f) { out[0] = a; lds_s[threadIdx.x] = a; }
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = (long long)wall_clock64(); }
}
template <int THREADS, int WAVES_PER_EU>
static void run_shape(const char* name, float* out, int lds_bytes, int spin, long long* dS, int grid, int reps) {
  auto K = k_shape<THREADS, WAVES_PER_EU>;
  CK(hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  for (int i = 0; i < 10; i++) hipLaunchKernelGGL(K, dim3(grid), dim3(THREADS), lds_bytes, 0, out, spin, dS);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL(K, dim3(grid), dim3(THREADS), lds_bytes, 0, out, spin, dS);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> S(2 * grid); CK(hipMemcpy(S.data(), dS, S.size() * 8, hipMemcpyDeviceToHost));
  long long first = S[0], last = S[1], lastin = S[0];
  for (int b = 0; b < grid; b++) { first = std::min(first, S[2 * b]); last = std::max(last, S[2 * b + 1]); lastin = std::max(lastin, S[2 * b]); }
  const double per = ms * 1e3 / reps, inside = (last - first) * 0.01;
  printf("%-34s grid %5d, LDS %6d B, spin %5d: %7.2f us per launch, wavefronts busy %7.2f us (entries spread over %5.2f), dispatch overhead %6.2f us\n", name, grid, lds_bytes, spin, per,
         inside, (lastin - first) * 0.01, per - inside);
}

int main(int argc, char** argv) {
  const int reps = 200, grid = 2048;
  f4* out; CK(hipMalloc(&out, 64u << 20));
  long long* dS; CK(hipMalloc(&dS, 2 * 4096 * 8));
  for (int spin : {0}) {
    for (size_t mb : {0, 1, 4, 14, 25, 50}) {
      const size_t bytes = mb << 20;
      run<0>("plain", out, bytes, spin, dS, grid, reps);
      if (mb) { run<1>("nt", out, bytes, spin, dS, grid, reps); run<2>("sc0 sc1", out, bytes, spin, dS, grid, reps); }
    }
  }
  float* fo = reinterpret_cast<float*>(out);
  for (int spin : {0, 2000}) {
    run_shape<64, 4>("64 threads, 4 waves / SIMD", fo, 0, spin, dS, 4096, reps);
    run_shape<64, 4>("64 threads, 4 waves / SIMD", fo, 10000, spin, dS, 4096, reps);          // k_substeps' shape
    run_shape<256, 1>("256 threads", fo, 0, spin, dS, 256, reps);
    run_shape<512, 1>("512 threads, 2 waves / SIMD", fo, 0, spin, dS, 256, reps);
    run_shape<512, 1>("512 threads, 2 waves / SIMD", fo, 132000, spin, dS, 256, reps);       // k_policy_tail's
    run_shape<512, 1>("512 threads, 2 waves / SIMD", fo, 153984, spin, dS, 256, reps);       // k_gemm_h2's
  }
  return 0;
}
