// fp64 vector-ALU issue rate without loop overhead (dev tool).  fp64_rate.hip unrolls 8 ops per loop iteration, so a
// taken branch rides on every 8 ops of a wave; here the loop body is N straight-line ops (N = 8 ... 65 536, i.e. 64 B ... 512 KB
// of code: the larger bodies exceed the 64 KB instruction cache a pair of CUs shares), occupancy is set by the workgroup shape
// (one workgroup of 4 x w waves per CU: w waves on every SIMD), and the shader clock is read next to the wall clock so that
// cycles are real cycles.
//   hipcc --offload-arch=gfx950 -O2 -o fp64_issue.bin fp64_issue.hip && ./fp64_issue.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define STR2(x) #x
#define STR(x) STR2(x)
// eight independent accumulators, multiplier in v[40:41]; pattern P: 0 = mul only, 1 = add/mul alternating, 2 = one dependent chain
#define BODY8_MUL "v_mul_f64 v[20:21], v[20:21], v[40:41]\n v_mul_f64 v[22:23], v[22:23], v[40:41]\n v_mul_f64 v[24:25], v[24:25], v[40:41]\n v_mul_f64 v[26:27], v[26:27], v[40:41]\n" \
                  "v_mul_f64 v[28:29], v[28:29], v[40:41]\n v_mul_f64 v[30:31], v[30:31], v[40:41]\n v_mul_f64 v[32:33], v[32:33], v[40:41]\n v_mul_f64 v[34:35], v[34:35], v[40:41]\n"
#define BODY8_MIX "v_mul_f64 v[20:21], v[20:21], v[40:41]\n v_add_f64 v[22:23], v[22:23], v[42:43]\n v_mul_f64 v[24:25], v[24:25], v[40:41]\n v_add_f64 v[26:27], v[26:27], v[42:43]\n" \
                  "v_mul_f64 v[28:29], v[28:29], v[40:41]\n v_add_f64 v[30:31], v[30:31], v[42:43]\n v_mul_f64 v[32:33], v[32:33], v[40:41]\n v_add_f64 v[34:35], v[34:35], v[42:43]\n"
#define BODY8_DEP "v_mul_f64 v[20:21], v[20:21], v[40:41]\n v_mul_f64 v[20:21], v[20:21], v[40:41]\n v_mul_f64 v[20:21], v[20:21], v[40:41]\n v_mul_f64 v[20:21], v[20:21], v[40:41]\n" \
                  "v_mul_f64 v[20:21], v[20:21], v[40:41]\n v_mul_f64 v[20:21], v[20:21], v[40:41]\n v_mul_f64 v[20:21], v[20:21], v[40:41]\n v_mul_f64 v[20:21], v[20:21], v[40:41]\n"
// an evaluator-like stream: three-address ops over a window of registers (destination differs from the sources)
#define BODY8_3AD "v_mul_f64 v[44:45], v[20:21], v[22:23]\n v_add_f64 v[46:47], v[24:25], v[26:27]\n v_mul_f64 v[48:49], v[28:29], v[30:31]\n v_add_f64 v[50:51], v[32:33], v[34:35]\n" \
                  "v_mul_f64 v[52:53], v[44:45], v[46:47]\n v_add_f64 v[54:55], v[48:49], v[50:51]\n v_mul_f64 v[56:57], v[52:53], v[40:41]\n v_add_f64 v[58:59], v[54:55], v[42:43]\n"

#define CLOBBER "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v40","v41","v42","v43", \
                "v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59"

template <int REPT, int P, int BIGREG>
__global__ void k(double *out, long long *clk, int iters, int dephase) {
  // dephase > 0: wave w of the CU (and every CU differently) idles w * dephase cycles first, so that the waves sharing an
  // instruction cache walk the body at different places (the evaluator's persistent waves drift apart the same way)
  if (dephase > 0) {
    const long long t0 = clock64(), wait = (long long)dephase * (long long)(((threadIdx.x >> 6) + 1) * 7 + (blockIdx.x % 13));
    while (clock64() - t0 < wait) __builtin_amdgcn_s_sleep(8);
  }
  if (BIGREG == 1) asm volatile("v_mov_b32 v247, 0" ::: "v247");      // 248 VGPRs: at most two waves per SIMD
  if (BIGREG == 2) asm volatile("v_mov_b32 v160, 0" ::: "v160");      // 168 VGPRs: at most three
  asm volatile("v_mov_b32 v20, 0\n v_mov_b32 v21, 0x3ff00000\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0x3ff00000\n v_mov_b32 v24, 0\n v_mov_b32 v25, 0x3ff00000\n"
               "v_mov_b32 v26, 0\n v_mov_b32 v27, 0x3ff00000\n v_mov_b32 v28, 0\n v_mov_b32 v29, 0x3ff00000\n v_mov_b32 v30, 0\n v_mov_b32 v31, 0x3ff00000\n"
               "v_mov_b32 v32, 0\n v_mov_b32 v33, 0x3ff00000\n v_mov_b32 v34, 0\n v_mov_b32 v35, 0x3ff00000\n"
               "v_mov_b32 v40, 0x10000000\n v_mov_b32 v41, 0x3ff00000\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0x3e000000\n" ::: CLOBBER);
  const long long c0 = clock64(), w0 = wall_clock64();
  // the loop in assembly: a body of more than 128 KB is out of reach of s_cbranch's 16-bit offset, so the back edge is s_setpc_b64
#define LOOPASM(BODY) asm volatile("s_mov_b32 s20, %1\n L0_%=:\n .rept %0\n" BODY ".endr\n s_sub_u32 s20, s20, 1\n s_cmp_eq_u32 s20, 0\n s_cbranch_scc1 L2_%=\n" \
                                   "s_getpc_b64 s[22:23]\n L1_%=:\n s_add_u32 s22, s22, L0_%=-L1_%=\n s_addc_u32 s23, s23, -1\n s_setpc_b64 s[22:23]\n L2_%=:\n" \
                                   :: "i"(REPT), "s"(iters) : CLOBBER, "s20", "s22", "s23", "scc")
  if (P == 0) LOOPASM(BODY8_MUL);
  if (P == 1) LOOPASM(BODY8_MIX);
  if (P == 2) LOOPASM(BODY8_DEP);
  if (P == 3) LOOPASM(BODY8_3AD);
  const long long c1 = clock64(), w1 = wall_clock64();
  double r;
  asm volatile("v_add_f64 %0, v[20:21], v[24:25]" : "=v"(r));
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  out[gid] = r;
  if ((threadIdx.x & 63) == 0) { clk[2 * (gid >> 6)] = c1 - c0; clk[2 * (gid >> 6) + 1] = w1 - w0; }
}

static double *d_out; static long long *d_clk;

template <int REPT, int P, int BIGREG> void run(const char *name, int wps, int dephase = 0) {
  const int block = 64 * 4 * (wps > 4 ? 4 : wps), grid = 256 * (wps > 4 ? wps / 4 : 1);
  const long long ops_target = 1 << 24;                      // ops per wave
  const int iters = (int)(ops_target / (8LL * REPT));
  hipLaunchKernelGGL((k<REPT, P, BIGREG>), dim3(grid), dim3(block), 0, 0, d_out, d_clk, iters > 16 ? iters / 16 : 1, 0);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<REPT, P, BIGREG>), dim3(grid), dim3(block), 0, 0, d_out, d_clk, iters, dephase);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const int n_wave = grid * block / 64;
  std::vector<long long> h(2 * n_wave);
  hipMemcpy(h.data(), d_clk, sizeof(long long) * 2 * n_wave, hipMemcpyDeviceToHost);
  double cs = 0, ws = 0; for (int i = 0; i < n_wave; ++i) { cs += h[2 * i]; ws += h[2 * i + 1]; }
  const double ghz = cs / ws * 0.1;                          // wall clock: 100 MHz
  const double ops_wave = (double)iters * 8 * REPT;
  const double cyc_per_op_simd = (cs / n_wave) / ops_wave / wps;          // shader-clock cycles a SIMD spends per op
  if (dephase) printf("dephased %6d: ", dephase);
  printf("%-26s body %6d ops (%4d KB)  waves/SIMD=%d  %8.3f ms  %6.2f T op/s  clock %.2f GHz  cycles/op/SIMD %.2f\n", name, 8 * REPT, 8 * REPT * 8 / 1024, wps, ms,
         ops_wave * n_wave * 64 / ms / 1e9, ghz, cyc_per_op_simd);
  hipEventDestroy(e0); hipEventDestroy(e1);
}

template <int REPT, int P> void sweep(const char *name) {
  for (int w : {1, 2, 3, 4, 8}) run<REPT, P, 0>(name, w);
}

int main() {
  hipMalloc(&d_out, sizeof(double) * 256 * 8 * 4 * 64);
  hipMalloc(&d_clk, sizeof(long long) * 2 * 256 * 8 * 4);
  sweep<1, 0>("mul x8 independent");
  sweep<8, 0>("mul x8 independent");
  sweep<64, 0>("mul x8 independent");
  sweep<512, 0>("mul x8 independent");
  sweep<2048, 0>("mul x8 independent");
  sweep<8192, 0>("mul x8 independent");
  sweep<64, 1>("mul/add alternating");
  sweep<64, 2>("mul dependent chain");
  sweep<64, 3>("three-address mix");
  sweep<2048, 3>("three-address mix");
  sweep<8192, 3>("three-address mix");
  for (int w : {1, 2}) run<64, 3, 1>("3-addr, 248 VGPRs", w);
  for (int w : {1, 2}) run<8192, 3, 1>("3-addr, 248 VGPRs", w);
  for (int w : {1, 2, 3}) run<64, 3, 2>("3-addr, 168 VGPRs", w);
  for (int w : {1, 2, 3}) run<8192, 3, 2>("3-addr, 168 VGPRs", w);
  // the instruction cache: the same bodies with the waves of a CU out of phase
  for (int w : {1, 2}) { run<64, 3, 1>("3-addr, 248 VGPRs", w, 3000); run<2048, 3, 1>("3-addr, 248 VGPRs", w, 3000); run<8192, 3, 1>("3-addr, 248 VGPRs", w, 3000); run<8192, 3, 1>("3-addr, 248 VGPRs", w, 30000); }
  return 0;
}
