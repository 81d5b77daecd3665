// Does the mere allocation of AGPRs change the rate at which a wave issues fp64 arithmetic?  (salu_vmem.hip / vmem_addtid.hip, whose
// kernels clobber a0..a3, run 32 independent v_mul_f64 in 168 cycles; vmem_issue.hip, without AGPRs, in 129.)
// A straight-line body of G groups of K v_mul_f64 (the same eight as the other microbenchmarks), one workgroup of 4 x W waves per CU, and
//   R0  248 VGPRs, no AGPR            R1  248 VGPRs + a0..a3 (accum_offset 248)      R2  256 VGPRs + a0..a255 (the one-wave configuration of fdg_isa_eval)
//   R3  120 VGPRs + a0..a3            R4  120 VGPRs, no AGPR
//   hipcc --offload-arch=gfx950 -O2 -o agpr_rate.bin agpr_rate.hip && ./agpr_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define MUL8 "v_mul_f64 v[44:45], v[20:21], v[22:23]\n v_mul_f64 v[46:47], v[24:25], v[26:27]\n v_mul_f64 v[48:49], v[28:29], v[30:31]\n v_mul_f64 v[50:51], v[32:33], v[34:35]\n" \
             "v_mul_f64 v[52:53], v[44:45], v[46:47]\n v_mul_f64 v[54:55], v[48:49], v[50:51]\n v_mul_f64 v[56:57], v[52:53], v[40:41]\n v_mul_f64 v[58:59], v[54:55], v[40:41]\n"
#define BASEC "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v40","v41", \
              "v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","s20","s22","s23","scc","memory"
#define INIT "v_mov_b32 v20, 0\n v_mov_b32 v21, 0x3ff00000\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0x3ff00000\n v_mov_b32 v24, 0\n v_mov_b32 v25, 0x3ff00000\n" \
             "v_mov_b32 v26, 0\n v_mov_b32 v27, 0x3ff00000\n v_mov_b32 v28, 0\n v_mov_b32 v29, 0x3ff00000\n v_mov_b32 v30, 0\n v_mov_b32 v31, 0x3ff00000\n" \
             "v_mov_b32 v32, 0\n v_mov_b32 v33, 0x3ff00000\n v_mov_b32 v34, 0\n v_mov_b32 v35, 0x3ff00000\n v_mov_b32 v40, 0x10000000\n v_mov_b32 v41, 0x3ff00000\n"
#define LOOP "s_mov_b32 s20, %2\n L0_%=:\n .rept %1\n .rept %0\n" MUL8 ".endr\n .endr\n s_sub_u32 s20, s20, 1\n s_cmp_eq_u32 s20, 0\n s_cbranch_scc1 L2_%=\n" \
             "s_getpc_b64 s[22:23]\n L1_%=:\n s_add_u32 s22, s22, L0_%=-L1_%=\n s_addc_u32 s23, s23, -1\n s_setpc_b64 s[22:23]\n L2_%=:\n"

template <int R, int KREP, int G>
__global__ void __launch_bounds__(512) k(double *out, long long *clk, int iters) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  long long c0, c1;
  if (R == 0) { asm volatile("v_mov_b32 v247, 0\n" INIT ::: "v247", BASEC); c0 = clock64(); asm volatile(LOOP :: "i"(KREP), "i"(G), "s"(iters) : "v247", BASEC); c1 = clock64(); }
  if (R == 1) { asm volatile("v_mov_b32 v247, 0\n v_accvgpr_write_b32 a3, 0\n" INIT ::: "v247", "a0", "a1", "a2", "a3", BASEC); c0 = clock64(); asm volatile(LOOP :: "i"(KREP), "i"(G), "s"(iters) : "v247", "a0", "a1", "a2", "a3", BASEC); c1 = clock64(); }
  if (R == 2) { asm volatile("v_mov_b32 v255, 0\n v_accvgpr_write_b32 a255, 0\n" INIT ::: "v255", "a255", BASEC); c0 = clock64(); asm volatile(LOOP :: "i"(KREP), "i"(G), "s"(iters) : "v255", "a255", BASEC); c1 = clock64(); }
  if (R == 3) { asm volatile("v_mov_b32 v119, 0\n v_accvgpr_write_b32 a3, 0\n" INIT ::: "v119", "a0", "a1", "a2", "a3", BASEC); c0 = clock64(); asm volatile(LOOP :: "i"(KREP), "i"(G), "s"(iters) : "v119", "a0", "a1", "a2", "a3", BASEC); c1 = clock64(); }
  if (R == 4) { asm volatile("v_mov_b32 v119, 0\n" INIT ::: "v119", BASEC); c0 = clock64(); asm volatile(LOOP :: "i"(KREP), "i"(G), "s"(iters) : "v119", BASEC); c1 = clock64(); }
  double r;
  asm volatile("v_add_f64 %0, v[56:57], v[58:59]" : "=v"(r));
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) clk[wave] = c1 - c0;
}

static double *d_out; static long long *d_clk;
static const char *names[] = {"R0 248 VGPR, no AGPR", "R1 248 VGPR + 4 AGPR", "R2 256 VGPR + 256 AGPR", "R3 120 VGPR + 4 AGPR", "R4 120 VGPR, no AGPR"};

template <int R, int KREP> void run(int wps) {
  constexpr int G = 128;
  const int block = 64 * 4 * wps, grid = 256, iters = 64;
  hipLaunchKernelGGL((k<R, KREP, G>), dim3(grid), dim3(block), 0, 0, d_out, d_clk, 4);
  if (hipDeviceSynchronize() != hipSuccess) { printf("%-26s waves/SIMD=%d: launch failed (%s)\n", names[R], wps, hipGetErrorString(hipGetLastError())); return; }
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<R, KREP, G>), dim3(grid), dim3(block), 0, 0, d_out, d_clk, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const int n_wave = grid * block / 64;
  std::vector<long long> h(n_wave);
  (void)hipMemcpy(h.data(), d_clk, sizeof(long long) * n_wave, hipMemcpyDeviceToHost);
  double cs = 0; for (int i = 0; i < n_wave; ++i) cs += h[i];
  const double ops = (double)iters * G * 8 * KREP;
  printf("%-26s K=%3d waves/SIMD=%d  %8.3f ms  wave cycles per op %.3f   %.2f e12 lane-ops/s\n", names[R], 8 * KREP, wps, ms, cs / n_wave / ops, ops * n_wave * 64 / ms / 1e9);
}

int main() {
  (void)hipMalloc(&d_out, sizeof(double) * 256 * 8 * 64);
  (void)hipMalloc(&d_clk, sizeof(long long) * 256 * 8);
  for (int rep = 0; rep < 2; ++rep) {
    run<0, 4>(1); run<1, 4>(1); run<2, 4>(1); run<3, 4>(1); run<4, 4>(1);
    run<0, 4>(2); run<1, 4>(2); run<3, 4>(2); run<4, 4>(2);
  }
  return 0;
}
