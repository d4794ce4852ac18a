// pure-read HBM streaming ceiling on MI355X: sum 16-byte loads over N GB
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int UNROLL, bool NT>
__global__ void __launch_bounds__(1024) k(const u32x4* p, u64 n16, u64* out) {
  u64 acc = 0;
  const u64 stride = (u64) gridDim.x * blockDim.x;
  u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
  }
  if (acc == 0x1234567) out[0] = acc;
}
template <int UNROLL, bool NT>
void run(const u32x4* d, u64 n16, u64* out, int grid, const char* name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<UNROLL, NT>), dim3(grid), dim3(1024), 0, 0, d, n16, out);
  hipEventRecord(e0);
  for (int it = 0; it < 5; ++it) hipLaunchKernelGGL((k<UNROLL, NT>), dim3(grid), dim3(1024), 0, 0, d, n16, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("%-28s grid %5d  %.3f ms  %.0f GB/s\n", name, grid, ms, n16 * 16.0 / ms / 1e6);
}
int main() {
  const u64 bytes = 32ull << 30;
  u32x4* d; u64* out;
  hipMalloc(&d, bytes); hipMalloc(&out, 8); hipMemset(d, 1, bytes);
  const u64 n16 = bytes / 16;
  for (int grid : {256, 512, 1024, 2048}) {
    run<4, false>(d, n16, out, grid, "unroll4");
    run<8, false>(d, n16, out, grid, "unroll8");
    run<4, true>(d, n16, out, grid, "unroll4 nontemporal");
    run<8, true>(d, n16, out, grid, "unroll8 nontemporal");
  }
  return 0;
}
