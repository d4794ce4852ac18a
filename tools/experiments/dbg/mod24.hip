// does gfx950 codegen compute (x & 0xffffff) % 13 correctly when the range is known?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;
__global__ void k(const u32* in, u64* outk, u64* outm, u32 n) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u32 v = in[i] & 0xffffffu;
  const u64 a = v;            // zero-extended, known 24 bits
  u64 t = 0;
  if (13ull == 0) { t = 1; } else { t = a % 13ull; }
  outk[i] = t;
  outm[i] = a;                // second use of the value (as in max(x))
}
int main() {
  const u32 n = 1u << 24;
  std::vector<u32> h(n);
  for (u32 i = 0; i < n; ++i) h[i] = i | 0x55000000u;
  u32* d; u64 *dk, *dm;
  hipMalloc(&d, n * 4); hipMalloc(&dk, n * 8); hipMalloc(&dm, n * 8);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, dk, dm, n);
  std::vector<u64> hk(n), hm(n);
  hipMemcpy(hk.data(), dk, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(hm.data(), dm, n * 8, hipMemcpyDeviceToHost);
  u64 bad = 0;
  for (u32 i = 0; i < n; ++i) {
    if (hk[i] != i % 13 || hm[i] != i) {
      if (bad < 5) printf("x=%u got k=%llu m=%llu exp %u\n", i, hk[i], hm[i], i % 13);
      ++bad;
    }
  }
  printf("mismatches: %llu\n", bad);
  return 0;
}
