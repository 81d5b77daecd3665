// Does reading a tile ahead matter for the evaluator's pattern on a memory-bound graph (84 columns x 512 B per tile, ~14 fp64
// ops per load, 4 root stores per tile, two waves per SIMD)?  Variants: AHEAD = 0: a tile's loads are issued at its start, a
// quarter of the columns ahead of their use (what the evaluator does: look-ahead of 300 ops); AHEAD = 1: every column is read one
// whole tile ahead (registers for all columns).  (dev tool)
#include <hip/hip_runtime.h>
#include <cstdio>
#define OPS4(n) for (int i = 0; i < (n); i += 4) asm volatile("v_mul_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));
template <int NCOL, int AHEAD, int NT, int NTS = 0>
__global__ void __launch_bounds__(64, 2) k(const double *__restrict__ src, double *__restrict__ dst, long ntile, long col_stride, int ops, int nstore) {
  const long wave = blockIdx.x, nw = gridDim.x;
  double a0 = threadIdx.x * 1e-9 + 1.0, a1 = a0 + 1e-9, a2 = a0 + 2e-9, a3 = a0 + 3e-9;
  const double m = 1.0000001;
  auto ld = [&](const double *p) { return NT ? __builtin_nontemporal_load(p) : *p; };
  if (AHEAD) {
    double v[NCOL], s = 0.0;
#pragma unroll
    for (int c = 0; c < NCOL; ++c) v[c] = ld(src + c * col_stride + wave * 64 + threadIdx.x);
    for (long t = wave; t < ntile; t += nw) {
      const long tn = t + nw < ntile ? t + nw : t;
#pragma unroll
      for (int c = 0; c < NCOL; ++c) {
        s += v[c];
        v[c] = ld(src + c * col_stride + tn * 64 + threadIdx.x);
        OPS4(ops)
      }
      for (int r = 0; r < nstore; ++r) dst[r * col_stride + t * 64 + threadIdx.x] = s + a0 + r;
    }
  } else {
    constexpr int Q = NCOL / 4;            // look-ahead: a quarter of the tile's columns
    for (long t = wave; t < ntile; t += nw) {
      double v[Q], s = 0.0;
#pragma unroll
      for (int c = 0; c < Q; ++c) v[c] = ld(src + c * col_stride + t * 64 + threadIdx.x);
#pragma unroll
      for (int c = 0; c < NCOL; ++c) {
        s += v[c % Q];
        if (c + Q < NCOL) v[c % Q] = ld(src + (c + Q) * col_stride + t * 64 + threadIdx.x);
        OPS4(ops)
      }
      for (int r = 0; r < nstore; ++r) { if (NTS) __builtin_nontemporal_store(s + a0 + r, dst + r * col_stride + t * 64 + threadIdx.x); else dst[r * col_stride + t * 64 + threadIdx.x] = s + a0 + r; }
    }
  }
  if (a0 + a1 + a2 + a3 == 12345.678) dst[0] = a1;
}
template <int NCOL, int AHEAD, int NT, int NTS = 0> void run(const double *src, double *dst, long total_bytes, int ops, int nstore) {
  const long ntile = total_bytes / (NCOL * 512L);
  const long cs = ntile * 64;
  const int grid = 256 * 4 * 2;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<NCOL, AHEAD, NT, NTS>), dim3(grid), dim3(64), 0, 0, src, dst, ntile, cs, ops, nstore);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<NCOL, AHEAD, NT, NTS>), dim3(grid), dim3(64), 0, 0, src, dst, ntile, cs, ops, nstore);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("nt_store=%d columns=%3d ahead=%d nt=%d ops/load=%2d stores=%d  %.3f ms  %.2f TB/s (reads+writes)  %.2f T op/s\n", NTS, NCOL, AHEAD, NT, ops, nstore, ms,
         (double)ntile * (NCOL + nstore) * 512 / ms / 1e9, (double)ntile * NCOL * ops * 64 / ms / 1e9);
}
int main() {
  const long total = 32L << 30;
  double *src, *dst;
  hipMalloc(&src, total + (1 << 20)); hipMalloc(&dst, (total / 84) * 8 + (1 << 20));
  hipMemset(src, 0, total);
  for (int nstore : {0, 4, 1, 8}) for (int ops : {12}) {
    run<84, 0, 0, 0>(src, dst, total, ops, nstore);
    run<84, 0, 1, 0>(src, dst, total, ops, nstore);
    run<84, 0, 0, 1>(src, dst, total, ops, nstore);
    run<84, 0, 1, 1>(src, dst, total, ops, nstore);
  }
  printf("last error: %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
