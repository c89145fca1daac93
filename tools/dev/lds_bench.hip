// LDS instruction cost on gfx950 as the physics kernel uses it: one 64-lane wavefront per workgroup, 16 workgroups per CU (10 KiB each),
// every wavefront issuing the same LDS read in a loop.  Reports LDS cycles per instruction per CU (kernel cycles * CUs-worth / instructions)
// for ds_read_b32 / b64 / b128 with 64 / 32 / 26 / 16 / 4 / 1 active lanes and three address patterns (unit stride, one address per
// 16-lane row = the sweep's record reads, records of 36 floats = the link records).
//   hipcc --offload-arch=gfx950 -O3 tools/dev/lds_bench.hip -o /tmp/lds_bench && /tmp/lds_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
template <int W>
__global__ void __launch_bounds__(64) k_read(float* out, int active, int pattern, int iters) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 2560; i += 64) lds[i] = (float)i;
  __syncthreads();
  int off;                                   // float offset of this lane's element
  if (pattern == 0) off = lane * W;          // unit stride
  else if (pattern == 1) off = (lane >> 4) * 20;   // one address per 16-lane row (solve records of 20 floats)
  else off = lane * 36;                      // link records, 36 floats apart
  off &= ~(W - 1);
  unsigned addr = (unsigned)(off * 4) + (unsigned)(size_t)lds * 0u;      // byte address (dynamic LDS starts at 0)
  float acc = 0.0f;
  if (lane < active) {
    for (int it = 0; it < iters; it++) {
      if (W == 1) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[u]) : "v"(addr), "n"(0));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 8; u++) acc += v[u];
      } else if (W == 2) {
        f2v v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) asm volatile("ds_read_b64 %0, %1" : "=v"(v[u]) : "v"(addr));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 8; u++) acc += v[u].x;
      } else {
        f4v v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) asm volatile("ds_read_b128 %0, %1" : "=v"(v[u]) : "v"(addr));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 8; u++) acc += v[u].x;
      }
    }
  }
  out[blockIdx.x * 64 + lane] = acc;
}
int main() {
  float* out; hipMalloc(&out, 4096 * 64 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000, blocks = 4096;
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const double clk = p.clockRate * 1e3;      // Hz
  const int lanes[] = {64, 32, 26, 16, 4, 1};
  const char* pn[] = {"unit-stride", "row-broadcast", "stride-36"};
  printf("CUs %d clock %.2f GHz; cycles per LDS instruction per CU (16 waves per CU all issuing)\n", p.multiProcessorCount, clk * 1e-9);
  for (int w = 0; w < 3; w++)
    for (int pat = 0; pat < 3; pat++)
      for (int a : lanes) {
        auto launch = [&]() {
          if (w == 0) hipLaunchKernelGGL(k_read<1>, dim3(blocks), dim3(64), 10240, 0, out, a, pat, iters);
          else if (w == 1) hipLaunchKernelGGL(k_read<2>, dim3(blocks), dim3(64), 10240, 0, out, a, pat, iters);
          else hipLaunchKernelGGL(k_read<4>, dim3(blocks), dim3(64), 10240, 0, out, a, pat, iters);
        };
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_cu = (double)blocks / p.multiProcessorCount * iters * 8;
        printf("ds_read_b%-3d %-14s active %2d : %6.2f cycles/instr/CU  (%.3f ms)\n", 32 * (1 << w), pn[pat], a, ms * 1e-3 * clk / instr_per_cu, ms);
      }
  return 0;
}
