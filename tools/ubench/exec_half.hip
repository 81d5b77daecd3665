// (a) Does a half-masked wave64 fp64 op cost less than a full one?  (b) cost of an s_barrier epoch among the 4 waves of a
// CU that exchange values through LDS.  dev tool; informs the work mapping for graphs whose live set overflows one lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void __launch_bounds__(64) k_exec(double *out, int iters, uint64_t mask) {
  double a0 = threadIdx.x * 1e-9 + 1.0, a1 = a0 + 1e-9, a2 = a0 + 2e-9, a3 = a0 + 3e-9;
  double m = 1.0000001;
  for (int i = 0; i < iters; ++i) {
    asm volatile("s_mov_b64 exec, %5\n"
                 "v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                 "v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                 "v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                 "v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                 "s_mov_b64 exec, -1\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "s"(mask));
  }
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3;
}
// alternating halves: 8 ops under the lower half, 8 under the upper half
__global__ void __launch_bounds__(64) k_alt(double *out, int iters) {
  double a0 = threadIdx.x * 1e-9 + 1.0, a1 = a0 + 1e-9, a2 = a0 + 2e-9, a3 = a0 + 3e-9;
  double m = 1.0000001;
  const uint64_t lo = 0xffffffffull, hi = 0xffffffff00000000ull;
  for (int i = 0; i < iters; ++i) {
    asm volatile("s_mov_b64 exec, %5\n"
                 "v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                 "v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                 "s_mov_b64 exec, %6\n"
                 "v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                 "v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                 "s_mov_b64 exec, -1\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "s"(lo), "s"(hi));
  }
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3;
}
// 4 waves of one workgroup (one per SIMD: the LDS request keeps other workgroups off the CU): per epoch K dependent
// fp64 ops, M values written to the wave's mailbox, barrier, M values read from the next wave's mailbox.
template <int SYNC>
__global__ void __launch_bounds__(256) k_epoch(double *out, int epochs, int K, int M) {
  extern __shared__ double lds[];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double a = lane * 1e-9 + 1.0, m = 1.0000001, acc = 0.0;
  for (int e = 0; e < epochs; ++e) {
    for (int i = 0; i < K; i += 8)
      asm volatile("v_mul_f64 %0, %0, %1\n v_mul_f64 %0, %0, %1\n v_mul_f64 %0, %0, %1\n v_mul_f64 %0, %0, %1\n"
                   "v_mul_f64 %0, %0, %1\n v_mul_f64 %0, %0, %1\n v_mul_f64 %0, %0, %1\n v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(m));
    const int buf = (e & 1) * 4 * 32 * 64;
    for (int j = 0; j < M; ++j) lds[buf + (w * 32 + j) * 64 + lane] = a + j;
    if (SYNC) __syncthreads();
    for (int j = 0; j < M; ++j) acc += lds[buf + ((SYNC ? ((w + 1) & 3) : w) * 32 + j) * 64 + lane];
  }
  out[blockIdx.x * 256 + threadIdx.x] = a + acc;
}
static float timeit(void (*launch)(void)) { return 0; }
int main() {
  double *d; hipMalloc(&d, 256 * 4 * 8 * 64 * 8 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const uint64_t masks[] = {~0ull, 0xffffffffull, 0xffffffff00000000ull, 0xffffull, 0x0000ffff0000ffffull, 0x1ull, 0x5555555555555555ull};
  const char *names[] = {"full", "lower32", "upper32", "lower16", "rows0+2", "lane0", "even lanes"};
  for (int wps : {1, 2}) {
    const int grid = 256 * 4 * wps, iters = 20000;
    for (int v = 0; v < 7; ++v) {
      hipLaunchKernelGGL(k_exec, dim3(grid), dim3(64), 0, 0, d, 100, masks[v]);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_exec, dim3(grid), dim3(64), 0, 0, d, iters, masks[v]);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double insts = (double)grid * iters * 16;
      printf("exec %-10s waves/SIMD=%d  %.3f ms  cycles/inst/SIMD @2.4GHz=%.2f\n", names[v], wps, ms, ms * 1e-3 * 2.4e9 / (insts / 1024));
    }
    hipLaunchKernelGGL(k_alt, dim3(grid), dim3(64), 0, 0, d, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_alt, dim3(grid), dim3(64), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)grid * iters * 16;
    printf("exec %-10s waves/SIMD=%d  %.3f ms  cycles/inst/SIMD @2.4GHz=%.2f\n", "alternating", wps, ms, ms * 1e-3 * 2.4e9 / (insts / 1024));
  }
  hipFuncSetAttribute((const void *)k_epoch<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void *)k_epoch<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const size_t shm = 2 * 4 * 64 * 64 * 8;   // 256 KB?  no: 2 buffers x 4 waves x 64 slots x 512 B = 256 KB -- use 32 slots
  (void)shm;
  for (int K : {64, 256, 1024})
    for (int M : {0, 8, 32}) {
      for (int sync = 0; sync < 2; ++sync) {
        const int epochs = 200000 / (K + 8 * M + 16);
        auto go = [&](int ep) {
          if (sync) hipLaunchKernelGGL(k_epoch<1>, dim3(256), dim3(256), 144 * 1024, 0, d, ep, K, M);
          else hipLaunchKernelGGL(k_epoch<0>, dim3(256), dim3(256), 144 * 1024, 0, d, ep, K, M);
        };
        go(2); hipDeviceSynchronize();
        hipEventRecord(e0); go(epochs); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("epoch K=%4d M=%2d %s  %.3f ms  cycles/epoch @2.4GHz=%.0f  (K ops alone at 8 cyc = %d)\n", K, M, sync ? "barrier" : "nosync ", ms,
               ms * 1e-3 * 2.4e9 / epochs, K * 8);
      }
    }
  hipError_t err = hipGetLastError();
  printf("last error: %s\n", hipGetErrorString(err));
  return 0;
}
