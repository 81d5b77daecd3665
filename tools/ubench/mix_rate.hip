// Does fp64 VALU throughput drop when the same waves also stream from HBM?  Each wave (64 lanes, 2 waves/SIMD,
// 248 VGPRs claimed) loops over tiles: 8 column loads of 512 B issued one tile ahead, OPS fp64 ops per load.
// Reports time, measured shader clock (clock64 / wall_clock64), achieved T op/s and TB/s.  (dev tool)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int MODE>   // 0: no loads, 1: loads from HBM (distinct tiles), 2: loads that hit L2 (same tile every time)
__global__ void __launch_bounds__(64) k(const double *__restrict__ src, double *out, long ntile, long col_stride, int ops, unsigned long long *clk) {
  const long wave = blockIdx.x, nw = gridDim.x;
  double a0 = threadIdx.x * 1e-9 + 1.0, a1 = a0 + 1e-9, a2 = a0 + 2e-9, a3 = a0 + 3e-9;
  const double m = 1.0000001;
  double v[8];
  for (int c = 0; c < 8; ++c) v[c] = 0.0;
  unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (long t = wave; t < ntile; t += nw) {
    double nv[8];
    if (MODE) {
      const long tt = MODE == 1 ? t : wave % 64;
#pragma unroll
      for (int c = 0; c < 8; ++c) nv[c] = __builtin_nontemporal_load(src + c * col_stride + tt * 64 + threadIdx.x);
    }
    for (int i = 0; i < ops * 8 / 8; ++i) {
      asm volatile("v_mul_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
                   "v_mul_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_add_f64 %3, %3, %4"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));
    }
    if (MODE) {
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] += nv[c];
    }
  }
  unsigned long long c1 = clock64(), w1 = wall_clock64();
  double s = a0 + a1 + a2 + a3;
  for (int c = 0; c < 8; ++c) s += v[c];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
template <int MODE> void run(const char *name, const double *src, double *out, long ntile, long cs, int ops, unsigned long long *clk, int wps) {
  const int grid = 256 * 4 * wps;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(64), 0, 0, src, out, ntile / 8, cs, ops, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(64), 0, 0, src, out, ntile, cs, ops, clk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double nops = (double)ntile * ops * 8 * 64, bytes = MODE == 1 ? (double)ntile * 8 * 512 : 0.0;
  printf("%-10s waves/SIMD=%d ops/load=%3d  %.3f ms  %.2f T op/s  %.2f TB/s  shader clock %.0f MHz\n", name, wps, ops, ms, nops / ms / 1e9, bytes / ms / 1e9,
         (double)h[0] / (double)h[1] * 100.0);
}
int main() {
  const long ntile = 1 << 20;              // 1 Mi tiles x 8 columns x 512 B = 4.3 GB per pass
  const long cs = ntile * 64 + 1024;       // column stride in doubles
  double *src, *out; unsigned long long *clk;
  hipMalloc(&src, (size_t)cs * 8 * 8); hipMalloc(&out, 256 * 4 * 8 * 64 * 8 * 2); hipMalloc(&clk, 16);
  hipMemset(src, 0, (size_t)cs * 8 * 8);
  for (int wps : {2, 4, 8}) for (int ops : {4, 8, 16, 31, 62}) {
    run<0>("fp64 only", src, out, ntile, cs, ops, clk, wps);
    run<2>("fp64 + L2", src, out, ntile, cs, ops, clk, wps);
    run<1>("fp64 + HBM", src, out, ntile, cs, ops, clk, wps);
  }
  return 0;
}
