// device-to-device copy variants (which one approaches the guide's 6.29 TB/s?): dev tool for fdg_copy16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2d __attribute__((ext_vector_type(2)));
template <int UNROLL, int NT>
__global__ void __launch_bounds__(256) k(const v2d *__restrict__ s, v2d *__restrict__ d, long n) {
  const long stride = (long)gridDim.x * 256L;
  long i = blockIdx.x * 256L + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
    v2d r[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) r[u] = NT ? __builtin_nontemporal_load(s + i + u * stride) : s[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { if (NT) __builtin_nontemporal_store(r[u], d + i + u * stride); else d[i + u * stride] = r[u]; }
  }
  for (; i < n; i += stride) d[i] = s[i];
}
// block-contiguous: each block copies a contiguous span
template <int NT>
__global__ void __launch_bounds__(256) kb(const v2d *__restrict__ s, v2d *__restrict__ d, long n) {
  const long per = (n + gridDim.x - 1) / gridDim.x;
  const long lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  for (long i = lo + threadIdx.x; i < hi; i += 1024) {
    v2d a = s[i], b = i + 256 < hi ? s[i + 256] : a, c = i + 512 < hi ? s[i + 512] : a, e = i + 768 < hi ? s[i + 768] : a;
    d[i] = a; if (i + 256 < hi) d[i + 256] = b; if (i + 512 < hi) d[i + 512] = c; if (i + 768 < hi) d[i + 768] = e;
  }
}
template <typename F> void run(const char *name, F launch, long bytes) {
  for (int w = 0; w < 3; ++w) launch();
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s %.1f GB/s (read + write)\n", name, 10.0 * 2 * bytes / (ms * 1e-3) / 1e9);
}
int main() {
  const long bytes = 2L << 30, n = bytes / 16;
  v2d *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes);
  hipMemset(a, 1, bytes);
  for (int grid : {2048, 4096, 8192, 16384, 65536}) {
    char nm[64];
    snprintf(nm, 64, "grid %d unroll 4", grid); run(nm, [&] { hipLaunchKernelGGL((k<4, 0>), dim3(grid), dim3(256), 0, 0, a, b, n); }, bytes);
    snprintf(nm, 64, "grid %d unroll 8", grid); run(nm, [&] { hipLaunchKernelGGL((k<8, 0>), dim3(grid), dim3(256), 0, 0, a, b, n); }, bytes);
    snprintf(nm, 64, "grid %d unroll 4 nontemporal", grid); run(nm, [&] { hipLaunchKernelGGL((k<4, 1>), dim3(grid), dim3(256), 0, 0, a, b, n); }, bytes);
    snprintf(nm, 64, "grid %d unroll 1", grid); run(nm, [&] { hipLaunchKernelGGL((k<1, 0>), dim3(grid), dim3(256), 0, 0, a, b, n); }, bytes);
    snprintf(nm, 64, "grid %d block-contiguous", grid); run(nm, [&] { hipLaunchKernelGGL((kb<0>), dim3(grid), dim3(256), 0, 0, a, b, n); }, bytes);
  }
  run("hipMemcpyDtoD", [&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }, bytes);
  return 0;
}
