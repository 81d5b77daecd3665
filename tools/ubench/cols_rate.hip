// HBM read rate of the evaluator's access pattern without the evaluator: every wave reads NCOL columns x 512 B
// per tile (leaf-major matrix), one tile ahead, and does OPS fp64 ops per load.  How does the achievable
// bandwidth depend on the number of concurrent column streams?  (dev tool)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NCOL>
__global__ void __launch_bounds__(64) k(const double *__restrict__ src, double *out, long ntile, long col_stride, int ops) {
  const long wave = blockIdx.x, nw = gridDim.x;
  double a0 = threadIdx.x * 1e-9 + 1.0, a1 = a0 + 1e-9, a2 = a0 + 2e-9, a3 = a0 + 3e-9;
  const double m = 1.0000001;
  double v[NCOL], s = 0.0;
#pragma unroll
  for (int c = 0; c < NCOL; ++c) v[c] = __builtin_nontemporal_load(src + c * col_stride + wave * 64 + threadIdx.x);
  for (long t = wave; t < ntile; t += nw) {
    const long tn = t + nw < ntile ? t + nw : t;
#pragma unroll
    for (int c = 0; c < NCOL; ++c) {
      s += v[c];
      v[c] = __builtin_nontemporal_load(src + c * col_stride + tn * 64 + threadIdx.x);
      for (int i = 0; i < ops; i += 4)
        asm volatile("v_mul_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_add_f64 %3, %3, %4"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = s + a0 + a1 + a2 + a3;
}
__global__ void fill_random(double *p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long x = i * 0x9E3779B97F4A7C15ull + 0x1234567ull;
    x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29;
    p[i] = (double)(x >> 11) * (1.0 / 9007199254740992.0);
  }
}
template <int NCOL> void run(const double *src, double *out, long total_bytes, int ops) {
  const long ntile = total_bytes / (NCOL * 512L);
  const long cs = ntile * 64;
  const int grid = 256 * 4 * 2;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NCOL>), dim3(grid), dim3(64), 0, 0, src, out, ntile, cs, ops);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NCOL>), dim3(grid), dim3(64), 0, 0, src, out, ntile, cs, ops);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("columns=%3d ops/load=%2d  %.3f ms  %.2f TB/s  %.2f T op/s\n", NCOL, ops, ms, (double)ntile * NCOL * 512 / ms / 1e9,
         (double)ntile * NCOL * ops * 64 / ms / 1e9);
}
int main(int argc, char **argv) {
  const long total = 4L << 30;
  double *src, *out;
  hipMalloc(&src, total + (1 << 20)); hipMalloc(&out, 256 * 4 * 2 * 64 * 8);
  hipMemset(src, 0, total);
  if (argc > 1) {   // profiling: only the evaluator-like point (110 columns, 8 ops per load), a few launches
    for (int i = 0; i < 4; ++i) run<110>(src, out, total, 8);
    return 0;
  }
  for (int pass = 0; pass < 2; ++pass) {
    printf("== source data: %s\n", pass ? "uniform random doubles" : "zeros");
    if (pass) { hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, src, (size_t)total / 8); hipDeviceSynchronize(); }
    for (int ops : {0, 8, 32}) {
      run<8>(src, out, total, ops); run<32>(src, out, total, ops); run<110>(src, out, total, ops);
    }
  }
  return 0;
}
