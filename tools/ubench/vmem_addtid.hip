// Does a vector-memory instruction WITHOUT a VGPR address cost its wave less?  (a leaf load costs the issuing wave of a
// one-wave-per-SIMD kernel 40-100 cycles.)  MUBUF with ADD_TID_ENABLE in the resource descriptor: the lane's address is
// base + soffset + inst_offset + 8 * lane, computed by the address unit -- no VGPR operand, one SGPR of offset.
// Same frame as vmem_issue.hip: groups of K independent v_mul_f64 followed by one of
//   V0  nothing                                              V1  global_load_dwordx2, SGPR base + VGPR lane offset, fixed base
//   V2  buffer_load_dwordx2 add_tid, fixed soffset           V3  buffer_load_dwordx2 offen (VGPR lane offset), fixed soffset
//   V4  s_mul_i32 + global address arithmetic (4 SALU) + global_load     (what fdg_isa_eval issues per leaf)
//   V5  s_mul_i32 soffset (1 SALU) + buffer_load add_tid
//   V6  ds_read_b64                                          V7  2 v_accvgpr_read + 2 v_accvgpr_write
//   V8  s_waitcnt vmcnt(15) alone                            V9 s_nop 0 alone
// DEPTH = the vmcnt the wave waits for in front of each load (15: at most 16 outstanding; 47: 48).
//   hipcc --offload-arch=gfx950 -O2 -o vmem_addtid.bin vmem_addtid.hip && ./vmem_addtid.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define MUL8 "v_mul_f64 v[44:45], v[20:21], v[22:23]\n v_mul_f64 v[46:47], v[24:25], v[26:27]\n v_mul_f64 v[48:49], v[28:29], v[30:31]\n v_mul_f64 v[50:51], v[32:33], v[34:35]\n" \
             "v_mul_f64 v[52:53], v[44:45], v[46:47]\n v_mul_f64 v[54:55], v[48:49], v[50:51]\n v_mul_f64 v[56:57], v[52:53], v[40:41]\n v_mul_f64 v[58:59], v[54:55], v[40:41]\n"
#define STEP  "s_add_u32 s26, s26, 1\n s_and_b32 s26, s26, s28\n"
#define ADDR  "s_mul_i32 s30, s26, s27\n s_mul_hi_u32 s31, s26, s27\n s_add_u32 s30, s30, s24\n s_addc_u32 s31, s31, s25\n"
#define CLOBBER "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v40","v41", \
                "v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59", \
                "v60","v61","v62","v63","v64","v65","v66","v67","v90","v91","v247","a0","a1","a2","a3", \
                "s20","s22","s23","s24","s25","s26","s27","s28","s30","s31","s32","s33","s34","s35","s36","scc","memory"
#define XSTR(x) #x
#define STR(x) XSTR(x)

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
  // s[32:35]: resource descriptor over the wave's region: stride 8, ADD_TID_ENABLE (word 3 bit 23), num_records = all ones
#define SETUP "s_mov_b32 s24, %3\n s_mov_b32 s25, %6\n s_mov_b32 s26, 0\n s_mov_b32 s27, 512\n s_mov_b32 s28, %4\n s_mov_b64 s[30:31], s[24:25]\n v_mov_b32 v90, %5\n" \
              "s_mov_b32 s32, s24\n s_or_b32 s33, s25, 0x80000\n s_mov_b32 s34, -1\n s_mov_b32 s35, %7\n s_mov_b32 s36, 0\n"
#define SETUP_ARGS(W3) "s"(base_lo), "s"(region / 512u - 1u), "v"(lane_off), "s"(base_hi), "s"(W3)
  const long long c0 = clock64(), w0 = wall_clock64();
#define LOOPASM(GROUP, W3) asm volatile(SETUP "s_mov_b32 s20, %2\n L0_%=:\n .rept %1\n .rept %0\n" MUL8 ".endr\n" GROUP ".endr\n s_sub_u32 s20, s20, 1\n s_cmp_eq_u32 s20, 0\n s_cbranch_scc1 L2_%=\n" \
                                    "s_getpc_b64 s[22:23]\n L1_%=:\n s_add_u32 s22, s22, L0_%=-L1_%=\n s_addc_u32 s23, s23, -1\n s_setpc_b64 s[22:23]\n L2_%=:\n s_waitcnt vmcnt(0) lgkmcnt(0)\n" \
                                    :: "i"(KREP), "i"(G), "s"(iters), SETUP_ARGS(W3) : CLOBBER)
#define WAIT "s_waitcnt vmcnt(" STR(DP) ")\n"
  constexpr unsigned TID = 0x00800000u, RAW = 0x00020000u;
#define BODY(DP_) \
  if (V == 0) LOOPASM("", RAW); \
  if (V == 1) LOOPASM(STEP "s_waitcnt vmcnt(" #DP_ ")\n global_load_dwordx2 v[60:61], v90, s[24:25]\n", RAW); \
  if (V == 2) LOOPASM(STEP "s_waitcnt vmcnt(" #DP_ ")\n buffer_load_dwordx2 v[60:61], off, s[32:35], s36\n", TID); \
  if (V == 3) LOOPASM(STEP "s_waitcnt vmcnt(" #DP_ ")\n buffer_load_dwordx2 v[60:61], v90, s[32:35], s36 offen\n", RAW); \
  if (V == 4) LOOPASM(STEP ADDR "s_waitcnt vmcnt(" #DP_ ")\n global_load_dwordx2 v[60:61], v90, s[30:31]\n", RAW); \
  if (V == 5) LOOPASM(STEP "s_mul_i32 s36, s26, s27\n s_waitcnt vmcnt(" #DP_ ")\n buffer_load_dwordx2 v[60:61], off, s[32:35], s36\n", TID); \
  if (V == 6) LOOPASM(STEP "s_waitcnt lgkmcnt(7)\n ds_read_b64 v[60:61], v90 offset:1024\n", RAW); \
  if (V == 7) LOOPASM(STEP "v_accvgpr_read_b32 v62, a0\n v_accvgpr_read_b32 v63, a1\n v_accvgpr_write_b32 a2, v64\n v_accvgpr_write_b32 a3, v65\n", RAW); \
  if (V == 8) LOOPASM(STEP "s_waitcnt vmcnt(" #DP_ ")\n", RAW); \
  if (V == 9) LOOPASM(STEP "s_nop 0\n", RAW);
  if (DEPTH == 15) { BODY(15) } else { BODY(47) }
  const long long c1 = clock64(), w1 = wall_clock64();
  double r;
  asm volatile("v_add_f64 %0, v[56:57], v[58:59]" : "=v"(r));
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  out[gid] = r + lds[threadIdx.x];
  if ((threadIdx.x & 63) == 0) { clk[2 * wave] = c1 - c0; clk[2 * wave + 1] = w1 - w0; }
}

// correctness of the descriptor: lane l of wave w must receive src[w * region / 8 + soff / 8 + l]
__global__ void chk(const double *src, double *out, unsigned soff) {
  const unsigned long long base_v = (unsigned long long)src;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)base_v), hi = __builtin_amdgcn_readfirstlane((unsigned)(base_v >> 32));
  double r;
  asm volatile("s_mov_b32 s32, %1\n s_or_b32 s33, %2, 0x80000\n s_mov_b32 s34, -1\n s_mov_b32 s35, 0x00800000\n s_mov_b32 s36, %3\n s_nop 4\n"
               "buffer_load_dwordx2 %0, off, s[32:35], s36 offset:16\n s_waitcnt vmcnt(0)\n" : "=v"(r) : "s"(lo), "s"(hi), "s"(soff) : "s32", "s33", "s34", "s35", "s36", "memory");
  out[threadIdx.x] = r;
}

static double *d_src, *d_out; static long long *d_clk;
static const char *names[] = {"V0 arithmetic alone", "V1 global_load fixed base", "V2 buffer_load add_tid fixed", "V3 buffer_load offen fixed", "V4 4 SALU + global_load",
                              "V5 1 SALU + buffer add_tid", "V6 ds_read_b64", "V7 4 accvgpr moves", "V8 s_waitcnt alone", "V9 s_nop alone"};

template <int V, int KREP, int DEPTH> void run(int wps, unsigned region) {
  constexpr int G = 256;
  const int block = 64 * 4 * wps, grid = 256;
  const int iters = 64;
  const size_t shmem = 64 * 1024;
  hipFuncSetAttribute((const void *)k<V, KREP, G, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  hipLaunchKernelGGL((k<V, KREP, G, DEPTH>), dim3(grid), dim3(block), shmem, 0, d_src, d_out, d_clk, 4, region);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<V, KREP, G, DEPTH>), dim3(grid), dim3(block), shmem, 0, d_src, d_out, d_clk, iters, region);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const int n_wave = grid * block / 64;
  std::vector<long long> h(2 * n_wave);
  hipMemcpy(h.data(), d_clk, sizeof(long long) * 2 * n_wave, hipMemcpyDeviceToHost);
  double cs = 0, ws = 0; for (int i = 0; i < n_wave; ++i) { cs += h[2 * i]; ws += h[2 * i + 1]; }
  const double groups = (double)iters * G;
  const double clock_ghz = cs / ws * 0.1;
  const double cyc_group = ms * 1e-3 * clock_ghz * 1e9 / groups;     // wall time of the launch in shader cycles per group
  const double bytes = (V >= 1 && V <= 5 ? groups * n_wave * 512.0 : 0.0);
  printf("%-30s K=%3d depth %2d waves/SIMD=%d region %8u B  %8.3f ms  clock %.2f GHz  cycles/group %7.1f  %.2f TB/s\n", names[V], 8 * KREP, DEPTH + 1, wps, region, ms,
         clock_ghz, cyc_group, bytes / ms / 1e9);
  hipEventDestroy(e0); hipEventDestroy(e1);
}

template <int KREP, int DEPTH> void sweep(int wps, unsigned region) {
  run<0, KREP, DEPTH>(wps, region); run<1, KREP, DEPTH>(wps, region); run<2, KREP, DEPTH>(wps, region); run<3, KREP, DEPTH>(wps, region);
  run<4, KREP, DEPTH>(wps, region); run<5, KREP, DEPTH>(wps, region);
  if (DEPTH == 15) { run<6, KREP, DEPTH>(wps, region); run<7, KREP, DEPTH>(wps, region); run<8, KREP, DEPTH>(wps, region); run<9, KREP, DEPTH>(wps, region); }
}

int main() {
  const size_t max_region = 4u << 20;
  hipMalloc(&d_src, max_region * 256 * 8);
  hipMalloc(&d_out, sizeof(double) * 256 * 8 * 64);
  hipMalloc(&d_clk, sizeof(long long) * 2 * 256 * 8);
  {   // descriptor check on a small ramp
    std::vector<double> ramp(4096); for (int i = 0; i < 4096; ++i) ramp[i] = i;
    hipMemcpy(d_src, ramp.data(), sizeof(double) * 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(chk, dim3(1), dim3(64), 0, 0, d_src, d_out, 1024u);
    std::vector<double> got(64); hipMemcpy(got.data(), d_out, sizeof(double) * 64, hipMemcpyDeviceToHost);
    int bad = 0; for (int l = 0; l < 64; ++l) if (got[l] != 1024 / 8 + 2 + l) bad++;
    printf("add_tid descriptor check: lane 0 -> %.0f, lane 1 -> %.0f, lane 63 -> %.0f (want %d, %d, %d): %s\n", got[0], got[1], got[63], 130, 131, 193, bad ? "WRONG" : "ok");
    if (bad) return 1;
  }
  hipMemset(d_src, 0, max_region * 256 * 8);
  for (unsigned region : {16384u, 4u << 20}) {
    for (int wps : {1, 2}) { sweep<4, 15>(wps, region); sweep<4, 47>(wps, region); }
  }
  return 0;
}
