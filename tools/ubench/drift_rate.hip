// Does the HBM read rate of many column streams depend on the waves staying in phase?  Same loop as cols_rate
// (NCOL columns x 512 B per tile, one tile ahead, OPS fp64 ops per load), but (a) waves in lock step, (b) every wave
// starts with a different amount of extra work so that the waves of the chip sit at different columns at any time,
// (c) like (b) but the 8 waves of a workgroup (one CU) are kept in phase with a barrier per tile.  (dev tool)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NCOL, int MODE>
__global__ void __launch_bounds__(512) k(const double *__restrict__ src, double *out, long ntile, long col_stride, int ops) {
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwv = blockDim.x >> 6;
  const long wave = (long)blockIdx.x * nwv + wv, nw = (long)gridDim.x * nwv;
  double a0 = lane * 1e-9 + 1.0, a1 = a0 + 1e-9, a2 = a0 + 2e-9, a3 = a0 + 3e-9;
  const double m = 1.0000001;
  if (MODE >= 1) {   // de-phase: up to one tile's worth of extra work, different per wave (MODE 2: per workgroup)
    const long skew = ((MODE == 2 ? blockIdx.x : wave) * 2654435761u) % (unsigned)(NCOL * (ops > 0 ? ops : 4));
    for (long i = 0; i < skew; i += 4)
      asm volatile("v_mul_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));
  }
  double v[NCOL], s = 0.0;
#pragma unroll
  for (int c = 0; c < NCOL; ++c) v[c] = __builtin_nontemporal_load(src + c * col_stride + wave * 64 + lane);
  for (long t = wave; t < ntile; t += nw) {
    const long tn = t + nw < ntile ? t + nw : t;
#pragma unroll
    for (int c = 0; c < NCOL; ++c) {
      s += v[c];
      v[c] = __builtin_nontemporal_load(src + c * col_stride + tn * 64 + lane);
      for (int i = 0; i < ops; i += 4)
        asm volatile("v_mul_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));
    }
    if (MODE == 2) __syncthreads();
  }
  out[(size_t)wave * 64 + lane] = s + a0 + a1 + a2 + a3;
}
template <int NCOL, int MODE> void run(const char *name, const double *src, double *out, long total_bytes, int ops, int wg_waves) {
  const long ntile = total_bytes / (NCOL * 512L) / (256 * 8) * (256 * 8);   // whole rounds: every wave does the same number of tiles
  const long cs = ntile * 64;
  const int grid = 256 * 8 / wg_waves;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NCOL, MODE>), dim3(grid), dim3(64 * wg_waves), 0, 0, src, out, ntile, cs, ops);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NCOL, MODE>), dim3(grid), dim3(64 * wg_waves), 0, 0, src, out, ntile, cs, ops);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("columns=%3d ops/load=%2d %-46s %.3f ms  %.2f TB/s\n", NCOL, ops, name, ms, (double)ntile * NCOL * 512 / ms / 1e9);
}
int main() {
  const long total = 6L << 30;
  double *src, *out;
  hipMalloc(&src, total + (1 << 20)); hipMalloc(&out, 256 * 8 * 64 * 8);
  hipMemset(src, 0, total);
  for (int ops : {8, 24}) {
    run<48, 0>("waves in lock step", src, out, total, ops, 1);
    run<48, 1>("every wave de-phased", src, out, total, ops, 1);
    run<48, 1>("every wave de-phased, 8-wave workgroups", src, out, total, ops, 8);
    run<48, 2>("workgroups de-phased, 8 waves kept in phase", src, out, total, ops, 8);
  }
  return 0;
}
