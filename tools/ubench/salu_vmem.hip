// Which part of a leaf load's address arithmetic costs the issuing wave its cycles, and what do the other non-arithmetic instructions cost?
// (vmem_addtid.hip: a load from a fixed base costs 10-13 cycles, the same load behind one s_mul_i32 that feeds it 35-40.)
// Same frame: groups of K independent v_mul_f64; PRE is issued in front of a group's arithmetic, POST behind it.
//   A0  POST: index step alone                              A1  POST: s_add_u32 / s_addc_u32 into the pointer + global_load from it   (what fdg_isa_eval issues per leaf)
//   A2  PRE: the same two adds, POST: the load  (hoisted)   A3  POST: s_mul_i32 + s_mul_hi_u32, no load
//   A4  POST: s_add_u32 + s_addc_u32, no load               A5  PRE: s_mul_i32, s_mul_hi_u32, s_add_u32, s_addc_u32, POST: the load  (hoisted)
//   A6  POST: 2 v_accvgpr_read                               A7  POST: 2 v_accvgpr_write
//   A8  POST: 4 v_mov_b32                                    A9  POST: ds_write_b64 + ds_read_b64
//   A10 POST: adds + global_load into an AGPR pair + 2 v_accvgpr_read of the pair loaded a group earlier
//   A11 POST: the four-instruction address arithmetic + load (vmem_issue.hip's V3)
//   hipcc --offload-arch=gfx950 -O2 -o salu_vmem.bin salu_vmem.hip && ./salu_vmem.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define MUL8 "v_mul_f64 v[44:45], v[20:21], v[22:23]\n v_mul_f64 v[46:47], v[24:25], v[26:27]\n v_mul_f64 v[48:49], v[28:29], v[30:31]\n v_mul_f64 v[50:51], v[32:33], v[34:35]\n" \
             "v_mul_f64 v[52:53], v[44:45], v[46:47]\n v_mul_f64 v[54:55], v[48:49], v[50:51]\n v_mul_f64 v[56:57], v[52:53], v[40:41]\n v_mul_f64 v[58:59], v[54:55], v[40:41]\n"
#define STEP  "s_add_u32 s26, s26, 1\n s_and_b32 s26, s26, s28\n"
#define ADDR  "s_mul_i32 s30, s26, s27\n s_mul_hi_u32 s31, s26, s27\n s_add_u32 s30, s30, s24\n s_addc_u32 s31, s31, s25\n"
// pointer += 512, then back to the region's base when the index wraps (s26 == 0): the two adds are the only arithmetic of the address
#define PADD  "s_add_u32 s30, s30, 0x200\n s_addc_u32 s31, s31, 0\n"
#define PWRAP "s_cmp_eq_u32 s26, 0\n s_cselect_b32 s30, s24, s30\n s_cselect_b32 s31, s25, s31\n"
#define CLOBBER "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v40","v41", \
                "v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59", \
                "v60","v61","v62","v63","v64","v65","v66","v67","v90","v91","v247","a0","a1","a2","a3", \
                "s20","s22","s23","s24","s25","s26","s27","s28","s30","s31","s32","s33","s34","s35","s36","scc","memory"

template <int V, int KREP, int G, int DEPTH>
__global__ void __launch_bounds__(512) k(const double *src, double *out, long long *clk, int iters, unsigned region) {
  extern __shared__ double lds[];
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const unsigned long long base_v = (unsigned long long)(src + (size_t)wave * (region / 8));
  const unsigned base_lo = __builtin_amdgcn_readfirstlane((unsigned)base_v), base_hi = __builtin_amdgcn_readfirstlane((unsigned)(base_v >> 32));
  const unsigned lane_off = (threadIdx.x & 63) * 8u;
  asm volatile("v_mov_b32 v247, 0" ::: "v247");          // 248 VGPRs: at most two waves per SIMD
  asm volatile("v_mov_b32 v20, 0\n v_mov_b32 v21, 0x3ff00000\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0x3ff00000\n v_mov_b32 v24, 0\n v_mov_b32 v25, 0x3ff00000\n"
               "v_mov_b32 v26, 0\n v_mov_b32 v27, 0x3ff00000\n v_mov_b32 v28, 0\n v_mov_b32 v29, 0x3ff00000\n v_mov_b32 v30, 0\n v_mov_b32 v31, 0x3ff00000\n"
               "v_mov_b32 v32, 0\n v_mov_b32 v33, 0x3ff00000\n v_mov_b32 v34, 0\n v_mov_b32 v35, 0x3ff00000\n v_mov_b32 v40, 0x10000000\n v_mov_b32 v41, 0x3ff00000\n" ::: CLOBBER);
#define SETUP "s_mov_b32 s24, %3\n s_mov_b32 s25, %6\n s_mov_b32 s26, 0\n s_mov_b32 s27, 512\n s_mov_b32 s28, %4\n s_mov_b64 s[30:31], s[24:25]\n v_mov_b32 v90, %5\n"
#define SETUP_ARGS "s"(base_lo), "s"(region / 512u - 1u), "v"(lane_off), "s"(base_hi)
  const long long c0 = clock64(), w0 = wall_clock64();
#define LOOP2(PRE, POST) asm volatile(SETUP "s_mov_b32 s20, %2\n L0_%=:\n .rept %1\n" PRE ".rept %0\n" MUL8 ".endr\n" POST ".endr\n s_sub_u32 s20, s20, 1\n s_cmp_eq_u32 s20, 0\n s_cbranch_scc1 L2_%=\n" \
                                    "s_getpc_b64 s[22:23]\n L1_%=:\n s_add_u32 s22, s22, L0_%=-L1_%=\n s_addc_u32 s23, s23, -1\n s_setpc_b64 s[22:23]\n L2_%=:\n s_waitcnt vmcnt(0) lgkmcnt(0)\n" \
                                    :: "i"(KREP), "i"(G), "s"(iters), SETUP_ARGS : CLOBBER)
#define WAITV "s_waitcnt vmcnt(47)\n"
#define LOADP "global_load_dwordx2 v[60:61], v90, s[30:31]\n"
  if (V == 0) LOOP2("", STEP);
  if (V == 1) LOOP2("", STEP PADD PWRAP WAITV LOADP);
  if (V == 2) LOOP2(STEP PADD PWRAP, WAITV LOADP);
  if (V == 3) LOOP2("", STEP "s_mul_i32 s32, s26, s27\n s_mul_hi_u32 s33, s26, s27\n");
  if (V == 4) LOOP2("", STEP "s_add_u32 s32, s26, s24\n s_addc_u32 s33, s27, s25\n");
  if (V == 5) LOOP2(STEP ADDR, WAITV LOADP);
  if (V == 6) LOOP2("", STEP "v_accvgpr_read_b32 v62, a0\n v_accvgpr_read_b32 v63, a1\n");
  if (V == 7) LOOP2("", STEP "v_accvgpr_write_b32 a2, v64\n v_accvgpr_write_b32 a3, v65\n");
  if (V == 8) LOOP2("", STEP "v_mov_b32 v62, v64\n v_mov_b32 v63, v65\n v_mov_b32 v66, v64\n v_mov_b32 v67, v65\n");
  if (V == 9) LOOP2("", STEP "s_waitcnt lgkmcnt(6)\n ds_write_b64 v90, v[64:65] offset:2048\n ds_read_b64 v[60:61], v90 offset:1024\n");
  if (V == 10) LOOP2("", STEP PADD PWRAP WAITV "v_accvgpr_read_b32 v62, a0\n v_accvgpr_read_b32 v63, a1\n global_load_dwordx2 a[0:1], v90, s[30:31]\n");
  if (V == 11) LOOP2("", STEP ADDR WAITV LOADP);
  const long long c1 = clock64(), w1 = wall_clock64();
  double r;
  asm volatile("v_add_f64 %0, v[56:57], v[58:59]" : "=v"(r));
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  out[gid] = r + lds[threadIdx.x];
  if ((threadIdx.x & 63) == 0) { clk[2 * wave] = c1 - c0; clk[2 * wave + 1] = w1 - w0; }
}

static double *d_src, *d_out; static long long *d_clk;
static const char *names[] = {"A0 index step alone", "A1 2 adds + load", "A2 2 adds hoisted, load", "A3 s_mul_i32 + s_mul_hi_u32", "A4 s_add_u32 + s_addc_u32", "A5 4 SALU hoisted, load",
                              "A6 2 v_accvgpr_read", "A7 2 v_accvgpr_write", "A8 4 v_mov_b32", "A9 ds_write_b64 + ds_read_b64", "A10 adds + load to AGPR + 2 reads", "A11 4 SALU + load"};

template <int V, int KREP> void run(int wps, unsigned region) {
  constexpr int G = 256, DEPTH = 47;
  const int block = 64 * 4 * wps, grid = 256;
  const int iters = 64;
  const size_t shmem = 64 * 1024;
  (void)hipFuncSetAttribute((const void *)k<V, KREP, G, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  hipLaunchKernelGGL((k<V, KREP, G, DEPTH>), dim3(grid), dim3(block), shmem, 0, d_src, d_out, d_clk, 4, region);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<V, KREP, G, DEPTH>), dim3(grid), dim3(block), shmem, 0, d_src, d_out, d_clk, iters, region);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const int n_wave = grid * block / 64;
  std::vector<long long> h(2 * n_wave);
  (void)hipMemcpy(h.data(), d_clk, sizeof(long long) * 2 * n_wave, hipMemcpyDeviceToHost);
  double cs = 0, ws = 0; for (int i = 0; i < n_wave; ++i) { cs += h[2 * i]; ws += h[2 * i + 1]; }
  const double groups = (double)iters * G;
  const double clock_ghz = cs / ws * 0.1;
  const double cyc_wave = (cs / n_wave) / groups;                     // the wave's own cycle counter
  const double cyc_wall = ms * 1e-3 * clock_ghz * 1e9 / groups;       // the launch's wall time in shader cycles
  printf("%-36s K=%3d waves/SIMD=%d region %8u B  %8.3f ms  clock %.2f GHz  cycles/group: wave %7.1f  launch %7.1f\n", names[V], 8 * KREP, wps, region, ms, clock_ghz, cyc_wave, cyc_wall);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
}

template <int KREP> void sweep(int wps, unsigned region) {
  run<0, KREP>(wps, region); run<1, KREP>(wps, region); run<2, KREP>(wps, region); run<3, KREP>(wps, region); run<4, KREP>(wps, region); run<5, KREP>(wps, region);
  run<6, KREP>(wps, region); run<7, KREP>(wps, region); run<8, KREP>(wps, region); run<9, KREP>(wps, region); run<10, KREP>(wps, region); run<11, KREP>(wps, region);
}

int main() {
  const size_t max_region = 4u << 20;
  (void)hipMalloc(&d_src, max_region * 256 * 8);
  (void)hipMalloc(&d_out, sizeof(double) * 256 * 8 * 64);
  (void)hipMalloc(&d_clk, sizeof(long long) * 2 * 256 * 8);
  (void)hipMemset(d_src, 0, max_region * 256 * 8);
  for (int rep = 0; rep < 2; ++rep)
    for (unsigned region : {16384u, 4u << 20})
      for (int wps : {1, 2}) { sweep<4>(wps, region); if (rep == 0 && wps == 1) sweep<2>(wps, region); }
  return 0;
}
