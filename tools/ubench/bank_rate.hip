// Do VGPR bank conflicts cost fp64 issue slots?  v_mul_f64 d, a, b with the 64-bit operands a and b in register
// pairs of the same bank pair (index mod 4 equal) versus different bank pairs, at 1 / 2 / 4 waves per SIMD. (dev tool)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int SAME, int BIG>
__global__ void __launch_bounds__(64) k(double *out, int iters) {
  if (BIG) asm volatile("v_mov_b32 v247, 0" ::: "v247");   // claim 248 VGPRs like the evaluator: two waves per SIMD fill the register file
  // explicit registers: accumulators v[20:21] v[24:25] v[28:29] v[32:33] (banks 0,1), multiplier in v[40:41] (same banks) or v[42:43] (other banks)
  asm volatile("v_mov_b32 v20, 0\n v_mov_b32 v21, 0x3ff00000\n v_mov_b32 v24, 0\n v_mov_b32 v25, 0x3ff00000\n"
               "v_mov_b32 v28, 0\n v_mov_b32 v29, 0x3ff00000\n v_mov_b32 v32, 0\n v_mov_b32 v33, 0x3ff00000\n"
               "v_mov_b32 v40, 0x10000000\n v_mov_b32 v41, 0x3ff00000\n v_mov_b32 v42, 0x10000000\n v_mov_b32 v43, 0x3ff00000\n"
               ::: "v20","v21","v24","v25","v28","v29","v32","v33","v40","v41","v42","v43");
  for (int i = 0; i < iters; ++i) {
    if (SAME == 1)
      asm volatile("v_mul_f64 v[20:21], v[20:21], v[40:41]\n v_mul_f64 v[24:25], v[24:25], v[40:41]\n v_mul_f64 v[28:29], v[28:29], v[40:41]\n v_mul_f64 v[32:33], v[32:33], v[40:41]\n"
                   "v_mul_f64 v[20:21], v[20:21], v[40:41]\n v_mul_f64 v[24:25], v[24:25], v[40:41]\n v_mul_f64 v[28:29], v[28:29], v[40:41]\n v_mul_f64 v[32:33], v[32:33], v[40:41]"
                   ::: "v20","v21","v24","v25","v28","v29","v32","v33");
    else if (SAME == 0)
      asm volatile("v_mul_f64 v[20:21], v[20:21], v[42:43]\n v_mul_f64 v[24:25], v[24:25], v[42:43]\n v_mul_f64 v[28:29], v[28:29], v[42:43]\n v_mul_f64 v[32:33], v[32:33], v[42:43]\n"
                   "v_mul_f64 v[20:21], v[20:21], v[42:43]\n v_mul_f64 v[24:25], v[24:25], v[42:43]\n v_mul_f64 v[28:29], v[28:29], v[42:43]\n v_mul_f64 v[32:33], v[32:33], v[42:43]"
                   ::: "v20","v21","v24","v25","v28","v29","v32","v33");
    else   // destination in the other bank pair as well: d = a(0,1) * b(2,3) -> d(2,3); then back
      asm volatile("v_mul_f64 v[22:23], v[20:21], v[42:43]\n v_mul_f64 v[26:27], v[24:25], v[42:43]\n v_mul_f64 v[30:31], v[28:29], v[42:43]\n v_mul_f64 v[34:35], v[32:33], v[42:43]\n"
                   "v_mul_f64 v[20:21], v[22:23], v[40:41]\n v_mul_f64 v[24:25], v[26:27], v[40:41]\n v_mul_f64 v[28:29], v[30:31], v[40:41]\n v_mul_f64 v[32:33], v[34:35], v[40:41]"
                   ::: "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35");
  }
  double r;
  asm volatile("v_add_f64 %0, v[20:21], v[24:25]" : "=v"(r));
  out[blockIdx.x * 64 + threadIdx.x] = r;
}
template <int SAME, int BIG> void run(const char *name, double *d) {
  for (int wps : {1, 2, 4}) {
    if (BIG && wps > 2) continue;
    const int grid = 256 * 4 * wps, iters = 200000;
    hipLaunchKernelGGL((k<SAME, BIG>), dim3(grid), dim3(64), 0, 0, d, 1000);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SAME, BIG>), dim3(grid), dim3(64), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s waves/SIMD=%d  %.2f T op/s\n", name, wps, (double)grid * iters * 8 * 64 / ms / 1e9);
  }
}
int main() {
  double *d; hipMalloc(&d, 256 * 4 * 4 * 64 * 8);
  run<1, 0>("a, b in the same bank pair", d);
  run<0, 0>("a, b in different bank pairs", d);
  run<2, 0>("a, b different; d in b's banks", d);
  run<0, 1>("different banks, 248 VGPRs claimed", d);
  return 0;
}
