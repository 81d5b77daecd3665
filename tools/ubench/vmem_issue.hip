// What does ONE vector-memory instruction cost the wave that issues it, when nothing waits for its data?  (in the
// kernels of the 4-loop vertex functions -- one wave per SIMD -- removing the leaf loads, which no instruction waits for any more
// once the arithmetic is garbage, saves 120-150 cycles per load.)  A straight-line body of G groups; a group is K independent
// v_mul_f64 and then one of
//   V0  nothing                                      (the arithmetic alone)
//   V1  the evaluator's address arithmetic alone: s_mul_i32, s_mul_hi_u32, s_add_u32, s_addc_u32
//   V2  global_load_dwordx2 from a fixed SGPR base   (512 B per wave; at most 16 outstanding: s_waitcnt vmcnt(15) in front)
//   V3  V1 + the load from the computed base         (what fdg_isa_eval issues per leaf)
//   V4  s_mov m0 + global_load_lds_dwordx4, low 32 lanes  (what fdg_isa_eval_pool issues per leaf; the load lands in LDS)
//   V5  V2 with the loads of FOUR groups issued back to back after 4 K ops
//   V6  global_load_dwordx4 of the low 32 lanes (the same 512 B; half the lanes' addresses)
// One workgroup of 4 x W waves per CU (W waves per SIMD), every wave reads its own region of `region` bytes round and round
// (L2 / Infinity-Cache resident for small regions, HBM for large ones).
//   hipcc --offload-arch=gfx950 -O2 -o vmem_issue.bin vmem_issue.hip && ./vmem_issue.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define MUL8 "v_mul_f64 v[44:45], v[20:21], v[22:23]\n v_mul_f64 v[46:47], v[24:25], v[26:27]\n v_mul_f64 v[48:49], v[28:29], v[30:31]\n v_mul_f64 v[50:51], v[32:33], v[34:35]\n" \
             "v_mul_f64 v[52:53], v[44:45], v[46:47]\n v_mul_f64 v[54:55], v[48:49], v[50:51]\n v_mul_f64 v[56:57], v[52:53], v[40:41]\n v_mul_f64 v[58:59], v[54:55], v[40:41]\n"
// s[24:25] = region base of the wave, s26 = running leaf index (0 .. n-1), s27 = leaf stride in bytes (512), s28 = n - 1 (mask)
#define STEP  "s_add_u32 s26, s26, 1\n s_and_b32 s26, s26, s28\n"
#define ADDR  "s_mul_i32 s30, s26, s27\n s_mul_hi_u32 s31, s26, s27\n s_add_u32 s30, s30, s24\n s_addc_u32 s31, s31, s25\n"
#define CLOBBER "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v40","v41", \
                "v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59", \
                "v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79", \
                "v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v247", \
                "s20","s22","s23","s24","s25","s26","s27","s28","s30","s31","scc","memory"

template <int V, int KREP, int G>
__global__ void __launch_bounds__(512) k(const double *src, double *out, long long *clk, int iters, unsigned region) {
  extern __shared__ double lds[];
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const unsigned long long base_v = (unsigned long long)(src + (size_t)wave * (region / 8));
  const unsigned base_lo = __builtin_amdgcn_readfirstlane((unsigned)base_v), base_hi = __builtin_amdgcn_readfirstlane((unsigned)(base_v >> 32));
  const unsigned lane_off = (threadIdx.x & 63) * 8u, lane_off16 = (threadIdx.x & 31) * 16u;
  asm volatile("v_mov_b32 v247, 0" ::: "v247");          // 248 VGPRs: at most two waves per SIMD
  asm volatile("v_mov_b32 v20, 0\n v_mov_b32 v21, 0x3ff00000\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0x3ff00000\n v_mov_b32 v24, 0\n v_mov_b32 v25, 0x3ff00000\n"
               "v_mov_b32 v26, 0\n v_mov_b32 v27, 0x3ff00000\n v_mov_b32 v28, 0\n v_mov_b32 v29, 0x3ff00000\n v_mov_b32 v30, 0\n v_mov_b32 v31, 0x3ff00000\n"
               "v_mov_b32 v32, 0\n v_mov_b32 v33, 0x3ff00000\n v_mov_b32 v34, 0\n v_mov_b32 v35, 0x3ff00000\n v_mov_b32 v40, 0x10000000\n v_mov_b32 v41, 0x3ff00000\n" ::: CLOBBER);
#define SETUP "s_mov_b32 s24, %3\n s_mov_b32 s25, %7\n s_mov_b32 s26, 0\n s_mov_b32 s27, 512\n s_mov_b32 s28, %4\n s_mov_b64 s[30:31], s[24:25]\n v_mov_b32 v90, %5\n v_mov_b32 v91, %6\n"
#define SETUP_ARGS "s"(base_lo), "s"(region / 512u - 1u), "v"(lane_off), "v"(lane_off16), "s"(base_hi)
  const long long c0 = clock64(), w0 = wall_clock64();
#define LOOPASM(GROUP) asm volatile(SETUP "s_mov_b32 s20, %2\n L0_%=:\n .rept %1\n .rept %0\n" MUL8 ".endr\n" GROUP ".endr\n s_sub_u32 s20, s20, 1\n s_cmp_eq_u32 s20, 0\n s_cbranch_scc1 L2_%=\n" \
                                    "s_getpc_b64 s[22:23]\n L1_%=:\n s_add_u32 s22, s22, L0_%=-L1_%=\n s_addc_u32 s23, s23, -1\n s_setpc_b64 s[22:23]\n L2_%=:\n s_waitcnt vmcnt(0)\n" \
                                    :: "i"(KREP), "i"(G), "s"(iters), SETUP_ARGS : CLOBBER)
  if (V == 0) LOOPASM("");
  if (V == 1) LOOPASM(STEP ADDR);
  if (V == 2) LOOPASM(STEP "s_waitcnt vmcnt(15)\n global_load_dwordx2 v[60:61], v90, s[24:25]\n");
  if (V == 3) LOOPASM(STEP ADDR "s_waitcnt vmcnt(15)\n global_load_dwordx2 v[60:61], v90, s[30:31]\n");
  if (V == 4) LOOPASM(STEP ADDR "s_mov_b64 exec, 0xffffffff\n s_mov_b32 m0, 0x2000\n s_nop 0\n global_load_lds_dwordx4 v91, s[30:31]\n s_mov_b64 exec, -1\n");
  if (V == 5) {
    asm volatile(SETUP "s_mov_b32 s20, %2\n L0_%=:\n .rept %1\n .rept %0\n" MUL8 MUL8 MUL8 MUL8 ".endr\n s_waitcnt vmcnt(12)\n"
                 STEP ADDR "global_load_dwordx2 v[60:61], v90, s[30:31]\n" STEP ADDR "global_load_dwordx2 v[62:63], v90, s[30:31]\n"
                 STEP ADDR "global_load_dwordx2 v[64:65], v90, s[30:31]\n" STEP ADDR "global_load_dwordx2 v[66:67], v90, s[30:31]\n"
                 ".endr\n s_sub_u32 s20, s20, 1\n s_cmp_eq_u32 s20, 0\n s_cbranch_scc1 L2_%=\n"
                 "s_getpc_b64 s[22:23]\n L1_%=:\n s_add_u32 s22, s22, L0_%=-L1_%=\n s_addc_u32 s23, s23, -1\n s_setpc_b64 s[22:23]\n L2_%=:\n s_waitcnt vmcnt(0)\n"
                 :: "i"(KREP), "i"(G / 4), "s"(iters), SETUP_ARGS : CLOBBER);
  }
  if (V == 6) LOOPASM(STEP ADDR "s_waitcnt vmcnt(15)\n s_mov_b64 exec, 0xffffffff\n global_load_dwordx4 v[60:63], v91, s[30:31]\n s_mov_b64 exec, -1\n");
  const long long c1 = clock64(), w1 = wall_clock64();
  double r;
  asm volatile("v_add_f64 %0, v[56:57], v[58:59]" : "=v"(r));
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  out[gid] = r + lds[threadIdx.x];
  if ((threadIdx.x & 63) == 0) { clk[2 * wave] = c1 - c0; clk[2 * wave + 1] = w1 - w0; }
}

static double *d_src, *d_out; static long long *d_clk;
static const char *names[] = {"V0 arithmetic alone", "V1 + address SALU", "V2 + load, fixed base", "V3 + SALU + load", "V4 + SALU + LDS-direct x4", "V5 loads in bursts of 4", "V6 + SALU + dwordx4 half"};

template <int V, int KREP> void run(int wps, unsigned region) {
  constexpr int G = 256;
  const int block = 64 * 4 * wps, grid = 256;
  const int iters = 64;
  const size_t shmem = 64 * 1024;
  hipFuncSetAttribute((const void *)k<V, KREP, G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  hipLaunchKernelGGL((k<V, KREP, G>), dim3(grid), dim3(block), shmem, 0, d_src, d_out, d_clk, 4, region);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<V, KREP, G>), dim3(grid), dim3(block), shmem, 0, d_src, d_out, d_clk, iters, region);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const int n_wave = grid * block / 64;
  std::vector<long long> h(2 * n_wave);
  hipMemcpy(h.data(), d_clk, sizeof(long long) * 2 * n_wave, hipMemcpyDeviceToHost);
  double cs = 0, ws = 0; for (int i = 0; i < n_wave; ++i) { cs += h[2 * i]; ws += h[2 * i + 1]; }
  const double groups = (double)iters * G;
  const double cyc_group = (cs / n_wave) / groups;
  const double bytes = (V >= 2 ? groups * n_wave * 512.0 : 0.0);
  printf("%-28s K=%3d ops/group  waves/SIMD=%d  region %8u B  %8.3f ms  clock %.2f GHz  cycles/group %7.1f  (%.1f per op)  %.2f TB/s\n", names[V], 8 * KREP, wps, region, ms,
         cs / ws * 0.1, cyc_group, cyc_group / (8 * KREP), bytes / ms / 1e9);
  hipEventDestroy(e0); hipEventDestroy(e1);
}

template <int KREP> void sweep(int wps, unsigned region) {
  run<0, KREP>(wps, region); run<1, KREP>(wps, region); run<2, KREP>(wps, region); run<3, KREP>(wps, region);
  run<4, KREP>(wps, region); run<5, KREP>(wps, region); run<6, KREP>(wps, region);
}

int main() {
  const size_t max_region = 4u << 20;
  hipMalloc(&d_src, max_region * 256 * 8);
  hipMemset(d_src, 0, max_region * 256 * 8);
  hipMalloc(&d_out, sizeof(double) * 256 * 8 * 64);
  hipMalloc(&d_clk, sizeof(long long) * 2 * 256 * 8);
  for (unsigned region : {16384u, 4u << 20}) {
    for (int wps : {1, 2}) { sweep<4>(wps, region); sweep<8>(wps, region); }
  }
  return 0;
}
