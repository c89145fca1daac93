// What bounds k_gemm_h2's operand stream per CU?  (development tool)  One 256-thread workgroup per CU reads what a k_gemm_h2 block reads
// (128 A rows + 192 W rows x K = 1440, two f16 planes: 40 KB per k-tile of 32, 45 k-tiles) with global_load_dwordx4 and xors it together.
//   PAT 0: the engine's layout -- row-major operands, 8 lanes per 128 B row chunk, 8 rows (stride 5760 B) per wave instruction
//   PAT 1: k-tile-major ("pre-tiled") operands -- the 128 x 128 B of a k-tile are one contiguous 16 KB block: 1 KB contiguous per wave instruction
//   ROT  : byte offset added to every A chunk (0 / 64: the history ring's rotation leaves the 128 B chunks of every second step 64 B off a line)
//   DEPTH: k-tiles in flight per wave (10 loads each)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dev/stream_bench.hip -o build/stream_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u32x4 gvec;
constexpr int K2B = 5760, NKT = 45, MT = 128, NT = 192;
template <int PAT, int DEPTH, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k_stream(const char* A, const char* W, unsigned* out, long long* times, int rot, int ntm) {
  const int tid = threadIdx.x;
  int bid = blockIdx.x; const int total = gridDim.x;
  if ((total & 7) == 0) { const int xcd = bid & 7, slot = bid >> 3; bid = xcd * (total >> 3) + slot; }
  const int tm = bid / 4, tn = bid & 3;
  const int srow = tid >> 3, sc = tid & 7;                    // PAT 0: row within a block of WAVES * 8 rows, 16 B chunk
  constexpr int RPI = WAVES * 8;                              // rows per instruction of the workgroup
  constexpr int NA = MT / RPI, NW = NT / RPI;
  const char* Ab = A + (size_t)tm * MT * K2B; const char* Wb = W + (size_t)tn * NT * K2B;
  u32x4 acc = {0, 0, 0, 0};
  const long long t0 = wall_clock64(), c0 = clock64();
  u32x4 v[DEPTH][NA + NW];
  auto issue = [&](int kt, int d) {
    if (kt > NKT - 1) kt = NKT - 1;
#pragma unroll
    for (int i = 0; i < NA; i++) {
      size_t off = PAT == 0 ? (size_t)(i * RPI + srow) * K2B + (size_t)kt * 128 + sc * 16 + rot : (size_t)kt * (MT * 128) + (size_t)(i * RPI) * 128 + tid * 16 + rot;
      v[d][i] = *(const gvec*)(Ab + off);
    }
#pragma unroll
    for (int i = 0; i < NW; i++) {
      size_t off = PAT == 0 ? (size_t)(i * RPI + srow) * K2B + (size_t)kt * 128 + sc * 16 : (size_t)kt * (NT * 128) + (size_t)(i * RPI) * 128 + tid * 16;
      v[d][NA + i] = *(const gvec*)(Wb + off);
    }
  };
#pragma unroll
  for (int d = 0; d < DEPTH; d++) issue(d, d);
  for (int kt = 0; kt < NKT; kt += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
#pragma unroll
      for (int i = 0; i < NA + NW; i++) acc ^= v[d][i];
      issue(kt + d + DEPTH, d);
    }
  }
  const long long t1 = wall_clock64(), c1 = clock64();
  out[blockIdx.x * blockDim.x + tid] = acc.x ^ acc.y ^ acc.z ^ acc.w;
  if (tid == 0) { times[blockIdx.x * 2] = t1 - t0; times[blockIdx.x * 2 + 1] = c1 - c0; }
}
template <int PAT, int DEPTH, int WAVES>
void run(const char* name, const char* A, const char* W, unsigned* out, long long* dT, int rot, int grid) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; i++) hipLaunchKernelGGL((k_stream<PAT, DEPTH, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, A, W, out, dT, rot, grid / 4);
  CK(hipEventRecord(e0, 0));
  const int reps = 100;
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_stream<PAT, DEPTH, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, A, W, out, dT, rot, grid / 4);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> T(grid * 2); CK(hipMemcpy(T.data(), dT, T.size() * 8, hipMemcpyDeviceToHost));
  double w = 0, c = 0; for (int b = 0; b < grid; b++) { w += T[2 * b] * 0.01; c += (double)T[2 * b + 1]; }
  w /= grid; c /= grid;
  printf("%-44s rot %3d grid %3d: %.2f us/launch, in-kernel %.2f us = %.0f ticks (%.2f GHz) = %.1f B/tick/CU, %.2f TB/s\n", name, rot, grid, ms * 1e3 / reps, w, c, c / w * 1e-3,
         (double)(MT + NT) * K2B / c, (double)grid * (MT + NT) * K2B / (w * 1e-6) * 1e-12);
}
int main(int argc, char** argv) {
  const int M = 8192, N = 768;
  char *A, *W; unsigned* out; long long* dT;
  CK(hipMalloc(&A, (size_t)M * K2B + 4096)); CK(hipMalloc(&W, (size_t)N * K2B + 4096)); CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&dT, 256 * 16));
  CK(hipMemset(A, 1, (size_t)M * K2B + 4096)); CK(hipMemset(W, 2, (size_t)N * K2B + 4096));
  for (int grid : {256, 64}) {
    for (int rot : {0, 64}) {
      run<0, 3, 4>("row-major, 4 waves, 3 k-tiles in flight", A, W, out, dT, rot, grid);
      run<1, 3, 4>("k-tile-major, 4 waves, 3 k-tiles in flight", A, W, out, dT, rot, grid);
    }
    run<0, 5, 4>("row-major, 4 waves, 5 k-tiles in flight", A, W, out, dT, 0, grid);
    run<1, 5, 4>("k-tile-major, 4 waves, 5 k-tiles in flight", A, W, out, dT, 0, grid);
    run<0, 3, 8>("row-major, 8 waves, 3 k-tiles in flight", A, W, out, dT, 0, grid);
    run<1, 3, 8>("k-tile-major, 8 waves, 3 k-tiles in flight", A, W, out, dT, 0, grid);
    run<0, 1, 4>("row-major, 4 waves, 1 k-tile in flight", A, W, out, dT, 0, grid);
    run<1, 1, 4>("k-tile-major, 4 waves, 1 k-tile in flight", A, W, out, dT, 0, grid);
  }
  return 0;
}
