// fp64 VALU issue-rate microbenchmark: dependent vs independent v_mul_f64 / v_add_f64
// chains at 1..8 waves per SIMD (dev tool; informs the ISA back end's scheduling).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int DEP, int ADD>
__global__ void __launch_bounds__(64) k(double *out, int iters) {
  double a0 = threadIdx.x * 1e-9 + 1.0, a1 = a0 + 1e-9, a2 = a0 + 2e-9, a3 = a0 + 3e-9, a4 = a0 + 4e-9, a5 = a0 + 5e-9, a6 = a0 + 6e-9, a7 = a0 + 7e-9;
  double m = 1.0000001;
  for (int i = 0; i < iters; ++i) {
    if (DEP == 1) {   // one chain: 8 dependent ops
      if (ADD) asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1" : "+v"(a0) : "v"(m));
      else asm volatile("v_mul_f64 %0, %0, %1\n v_mul_f64 %0, %0, %1\n v_mul_f64 %0, %0, %1\n v_mul_f64 %0, %0, %1\n v_mul_f64 %0, %0, %1\n v_mul_f64 %0, %0, %1\n v_mul_f64 %0, %0, %1\n v_mul_f64 %0, %0, %1" : "+v"(a0) : "v"(m));
    } else if (DEP == 2) {  // two interleaved chains
      asm volatile("v_mul_f64 %0, %0, %2\n v_mul_f64 %1, %1, %2\n v_mul_f64 %0, %0, %2\n v_mul_f64 %1, %1, %2\n v_mul_f64 %0, %0, %2\n v_mul_f64 %1, %1, %2\n v_mul_f64 %0, %0, %2\n v_mul_f64 %1, %1, %2" : "+v"(a0), "+v"(a1) : "v"(m));
    } else if (DEP == 4) {
      asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));
    } else {  // 8 independent
      asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int DEP, int ADD> void run(const char *name, double *d) {
  for (int wps : {1, 2, 4, 8}) {
    int grid = 256 * 4 * wps, iters = 20000;
    hipLaunchKernelGGL((k<DEP, ADD>), dim3(grid), dim3(64), 0, 0, d, 100);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<DEP, ADD>), dim3(grid), dim3(64), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double insts = (double)grid * iters * 8;
    printf("%-22s waves/SIMD=%d  %.3f ms  %.2f Tinst-lanes/s (=TFLOP/s non-FMA)  cycles/inst/SIMD @2.4GHz=%.2f\n", name, wps, ms,
           insts * 64 / ms / 1e9, ms * 1e-3 * 2.4e9 / (insts / 1024));
  }
}
int main() {
  double *d; hipMalloc(&d, 256 * 4 * 8 * 64 * 8);
  run<1, 0>("mul dependent x1", d);
  run<1, 1>("add dependent x1", d);
  run<2, 0>("mul 2 chains", d);
  run<4, 0>("mul 4 chains", d);
  run<8, 0>("mul 8 independent", d);
  return 0;
}
