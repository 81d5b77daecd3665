// What can a wave stream out of a ROW-MAJOR [B, L] matrix (compile_Python's layout) when lane = row?  (dev tool)
// A tile is 64 rows; a chunk is 16 consecutive leaves of those rows (64 x 128 B).  Two ways of getting a chunk to the lanes:
//   MODE 0  LDS-direct: 8 x global_load_lds_dwordx4 into one of NB staging buffers (the evaluator's row-major variant);
//           the bytes in flight live in LDS, so NB x waves/CU x 8 KB <= 160 KB bounds them.
//   MODE 1  register landing: 8 x global_load_dwordx4 into 32 VGPRs per chunk, NV chunks in flight in registers, then
//           ds_write_b128 (swizzled) into ONE 8 KB transposition buffer and 16 x ds_read_b64 with lane = row.
// Either way every leaf is then used by K dependent fp64 multiply-adds (K ~ 12 for the 4-loop self-energy) and the tile's
// "roots" (4 sums) are written row-major.  The (tile, chunk) stream is pipelined across tile boundaries.
// usage: rm_stream <rows> <L> <pitch> ; prints TB/s of matrix bytes for a few configurations
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));
#define WAITCNT(vm, lgkm) __builtin_amdgcn_s_waitcnt(((vm) & 15) | (((vm) >> 4) << 14) | (7 << 4) | ((lgkm) << 8))

// the tile's four "roots" per row, row-major [rows, 4]: ROOTS 1 = each lane its own 32 bytes (two 16-byte stores at a 32-byte
// stride), 2 = the same non-temporal, 3 = through LDS so that every store instruction writes 1 KB contiguous, 4 = 3 non-temporal
template <int ROOTS> __device__ __forceinline__ void put_roots(double *root, long tile, int l, double a0, double a1, double a2, double a3, char *scratch) {
  double *blk = root + tile * 64 * 4;
  if (ROOTS == 5) { double *r = root + ((long)blockIdx.x * 64 + l) * 4; r[0] = a0; r[1] = a1; r[2] = a2; r[3] = a3; }        // every wave its own 2 KB, over and over: stays in L2
  if (ROOTS == 6) { double *r = root + (((tile >> 3) & 0xffff) * 64 + l) * 4; r[0] = a0; r[1] = a1; r[2] = a2; r[3] = a3; }   // a 128 MB window: stays in the Infinity Cache
  if (ROOTS == 1) { double *r = blk + l * 4; r[0] = a0; r[1] = a1; r[2] = a2; r[3] = a3; }
  if (ROOTS == 2) { d2 *r = (d2 *)(blk + l * 4); __builtin_nontemporal_store(d2{a0, a1}, r); __builtin_nontemporal_store(d2{a2, a3}, r + 1); }
  if (ROOTS >= 3) {
    *(d2 *)(scratch + l * 32) = d2{a0, a1}; *(d2 *)(scratch + l * 32 + 16) = d2{a2, a3};
    const d2 x = *(const d2 *)(scratch + l * 16), y = *(const d2 *)(scratch + 1024 + l * 16);
    d2 *r = (d2 *)blk + l;
    if (ROOTS == 3) { r[0] = x; r[64] = y; } else { __builtin_nontemporal_store(x, r); __builtin_nontemporal_store(y, r + 64); }
  }
}

template <int K> __device__ __forceinline__ double work(double acc, double x) {
#pragma unroll
  for (int k = 0; k < K; ++k) acc = acc * 0.999 + x;       // two dependent fp64 ops per k (contraction off)
  return acc;
}

// source address (in doubles, relative to the tile's first row) of the 16-byte piece lane l fetches in instruction n of chunk c
__device__ __forceinline__ long src_off(int l, int n, int c, long pitch) {
  const int row = 8 * n + (l >> 3), piece = (l & 7) ^ ((row >> 1) & 7);
  return (long)row * pitch + c * 16 + piece * 2;
}
// LDS byte address of leaf j (0..15) of row r inside a staging buffer
__device__ __forceinline__ unsigned lds_off(int r, int j) { return r * 128 + ((((j >> 1) ^ ((r >> 1) & 7))) << 4) + ((j & 1) << 3); }

// the same bytes read linearly (lane l takes 16 B at l*16 of every KB of the tile): the harness's ceiling for this volume
template <int K, int ROOTS>
__global__ void __launch_bounds__(64) k_lin(const double *__restrict__ src, long pitch, long ntile, int nchunk, double *__restrict__ root) {
  const int l = threadIdx.x;
  __shared__ __attribute__((aligned(16))) char scratch[2048];
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  for (long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const d2 *base = (const d2 *)(src + tile * 64 * pitch) + l;
    const int nk = (int)(64 * pitch * 8 / 1024);
    for (int i = 0; i < nk; i += 8) {
      d2 v[8];
#pragma unroll
      for (int n = 0; n < 8; ++n) v[n] = i + n < nk ? __builtin_nontemporal_load(base + (i + n) * 64) : d2{0, 0};
#pragma unroll
      for (int n = 0; n < 8; n += 2) { a0 = work<K>(a0, v[n].x); a1 = work<K>(a1, v[n].y); a2 = work<K>(a2, v[n + 1].x); a3 = work<K>(a3, v[n + 1].y); }
    }
    if (ROOTS) { put_roots<ROOTS>(root, tile, l, a0, a1, a2, a3, scratch); a0 = a1 = a2 = a3 = 0; }
  }
  if (!ROOTS) root[blockIdx.x * 64 + l] = a0 + a1 + a2 + a3;
}

template <int NB, int K, int ROOTS = 1, int AUX = 0>
__global__ void __launch_bounds__(64) k_lds(const double *__restrict__ src, long pitch, long ntile, int nchunk, double *__restrict__ root) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int l = threadIdx.x;
  const long nstream = ((ntile - blockIdx.x + gridDim.x - 1) / gridDim.x) * nchunk;   // chunks this wave walks
  auto issue = [&](long s) {
    const long tile = blockIdx.x + (s / nchunk) * gridDim.x;
    const int c = (int)(s % nchunk), buf = (int)(s % NB);
    const double *base = src + tile * 64 * pitch;
#pragma unroll
    for (int n = 0; n < 8; ++n)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + src_off(l, n, c, pitch)),
                                       (__attribute__((address_space(3))) void *)(lds + buf * 8192 + n * 1024), 16, 0, AUX);
  };
  for (long s = 0; s < NB - 1 && s < nstream; ++s) issue(s);
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  double hold[ROOTS > 20 ? ROOTS - 20 : 1][4];
  for (long s = 0; s < nstream; ++s) {
    // (vmcnt counts loads and stores in issue order: right after a tile's two root stores the wait for the landed chunk must
    //  allow them to stay outstanding too, or the wave sits out the stores' whole latency -- ROOTS >= 10 model the naive wait)
    const bool after_store = (ROOTS % 10) != 0 && ROOTS < 10 && s > 0 && s % nchunk == 0;
    if (s + NB - 1 < nstream) {
      issue(s + NB - 1);
      if (NB == 2) { if (after_store) WAITCNT(10, 15); else WAITCNT(8, 15); }
      else if (NB == 3) { if (after_store) WAITCNT(18, 15); else WAITCNT(16, 15); }
      else { if (after_store) WAITCNT(26, 15); else WAITCNT(24, 15); }
    }
    else WAITCNT(0, 15);
    asm volatile("" ::: "memory");
    const int buf = (int)(s % NB);
    double v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = *(const double *)(lds + buf * 8192 + lds_off(l, j));
#pragma unroll
    for (int j = 0; j < 16; j += 4) { a0 = work<K>(a0, v[j]); a1 = work<K>(a1, v[j + 1]); a2 = work<K>(a2, v[j + 2]); a3 = work<K>(a3, v[j + 3]); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the buffer is refilled by the next issue
    if (ROOTS && ROOTS < 20 && s % nchunk == nchunk - 1) {
      const long tile = blockIdx.x + (s / nchunk) * gridDim.x;
      put_roots<ROOTS % 10>(root, tile, l, a0, a1, a2, a3, lds + NB * 8192);
      a0 = a1 = a2 = a3 = 0;
    }
    if (ROOTS > 20 && s % nchunk == nchunk - 1) {     // the roots of T = ROOTS - 20 tiles are kept in registers and written together
      constexpr int T = ROOTS > 20 ? ROOTS - 20 : 1;
      const long tl = s / nchunk;                    // this wave's tile counter
      const int slot = (int)(tl % T);
#pragma unroll
      for (int q = 0; q < T; ++q) if (q == slot) { hold[q][0] = a0; hold[q][1] = a1; hold[q][2] = a2; hold[q][3] = a3; }
      a0 = a1 = a2 = a3 = 0;
      if (slot == T - 1 || s == nstream - 1) {
#pragma unroll
        for (int q = 0; q < T; ++q) if (q <= slot) {
          const long tile = blockIdx.x + (tl - slot + q) * gridDim.x;
          put_roots<1>(root, tile, l, hold[q][0], hold[q][1], hold[q][2], hold[q][3], lds + NB * 8192);
        }
      }
    }
  }
  if (!ROOTS) root[blockIdx.x * 64 + l] = a0 + a1 + a2 + a3;
}

template <int NV, int K>
__global__ void __launch_bounds__(64) k_reg(const double *__restrict__ src, long pitch, long ntile, int nchunk, double *__restrict__ root) {
  __shared__ __attribute__((aligned(1024))) char lds[8192];
  const int l = threadIdx.x;
  const long nstream = ((ntile - blockIdx.x + gridDim.x - 1) / gridDim.x) * nchunk;
  d2 land[NV][8];
  auto issue = [&](long s, d2 (&dst)[8]) {
    const long tile = blockIdx.x + (s / nchunk) * gridDim.x;
    const int c = (int)(s % nchunk);
    const double *base = src + tile * 64 * pitch;
#pragma unroll
    for (int n = 0; n < 8; ++n) dst[n] = __builtin_nontemporal_load((const d2 *)(base + src_off(l, n, c, pitch)));
  };
#pragma unroll
  for (int i = 0; i < NV; ++i) if (i < nstream) issue(i, land[i]);
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  for (long s0 = 0; s0 < nstream; s0 += NV) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const long s = s0 + i;
      if (s >= nstream) break;
#pragma unroll
      for (int n = 0; n < 8; ++n) *(d2 *)(lds + n * 1024 + l * 16) = land[i][n];      // lane-linear image, as the LDS-direct load leaves it
      if (s + NV < nstream) issue(s + NV, land[i]);
      double v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = *(const double *)(lds + lds_off(l, j));
#pragma unroll
      for (int j = 0; j < 16; j += 4) { a0 = work<K>(a0, v[j]); a1 = work<K>(a1, v[j + 1]); a2 = work<K>(a2, v[j + 2]); a3 = work<K>(a3, v[j + 3]); }
      if (s % nchunk == nchunk - 1) {
        const long tile = blockIdx.x + (s / nchunk) * gridDim.x;
        double *r = root + (tile * 64 + l) * 4;
        r[0] = a0; r[1] = a1; r[2] = a2; r[3] = a3;
        a0 = a1 = a2 = a3 = 0;
      }
    }
  }
}

template <typename F> static void run(const char *name, F launch, long rows, int L) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  printf("%-52s %.3f ms  %.2f TB/s of matrix bytes  (%s)\n", name, ms, (double)rows * L * 8 / ms / 1e9, hipGetErrorString(hipGetLastError()));
}

int main(int argc, char **argv) {
  const long rows = argc > 1 ? atol(argv[1]) : 16000000;
  const int L = argc > 2 ? atoi(argv[2]) : 96;
  const long pitch = argc > 3 ? atol(argv[3]) : L;
  const int nchunk = L / 16;                     // (L a multiple of 16 here: the tail chunk is the evaluator's business)
  const long ntile = rows / 64;
  double *src, *root;
  hipMalloc(&src, (size_t)rows * pitch * 8 + 4096); hipMalloc(&root, (size_t)rows * 4 * 8);
  hipMemset(src, 0, (size_t)rows * pitch * 8 + 4096);
  printf("rows %ld  L %d  pitch %ld  (%d chunks per tile, K = 6: 12 fp64 ops per leaf)\n", rows, L, pitch, nchunk);
#define LDSRUN(NB, WPC) { char nm[96]; snprintf(nm, sizeof nm, "LDS-direct, %d buffers, %d waves/CU", NB, WPC); \
    hipFuncSetAttribute((const void *)k_lds<NB, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 / WPC); \
    run(nm, [&] { hipLaunchKernelGGL((k_lds<NB, 6>), dim3(256 * WPC), dim3(64), 160 * 1024 / WPC / 1024 * 1024, 0, src, pitch, ntile, nchunk, root); }, rows, L); }
  #define LDSRUN2(NB, WPC, KK, RR) { char nm[96]; snprintf(nm, sizeof nm, "LDS-direct, %d buffers, %d waves/CU, K=%d, roots %d", NB, WPC, KK, RR); \
    hipFuncSetAttribute((const void *)k_lds<NB, KK, RR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 / WPC); \
    run(nm, [&] { hipLaunchKernelGGL((k_lds<NB, KK, RR>), dim3(256 * WPC), dim3(64), 160 * 1024 / WPC / 1024 * 1024, 0, src, pitch, ntile, nchunk, root); }, rows, L); }
  LDSRUN2(2, 6, 6, 0) LDSRUN2(2, 6, 6, 1) LDSRUN2(2, 6, 6, 2)
#define LDSRUN3(NB, WPC, KK, RR, AUX) { char nm[96]; snprintf(nm, sizeof nm, "LDS-direct, %d buffers, %d waves/CU, roots %d, load policy bits %d", NB, WPC, RR, AUX); \
    hipFuncSetAttribute((const void *)k_lds<NB, KK, RR, AUX>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 / WPC); \
    run(nm, [&] { hipLaunchKernelGGL((k_lds<NB, KK, RR, AUX>), dim3(256 * WPC), dim3(64), 160 * 1024 / WPC / 1024 * 1024, 0, src, pitch, ntile, nchunk, root); }, rows, L); }
  LDSRUN3(2, 6, 6, 22, 0) LDSRUN3(2, 6, 6, 24, 0) LDSRUN3(2, 6, 6, 28, 0) LDSRUN3(2, 6, 6, 36, 0)
#define LINRUN(KK, RR, WPC) { char nm[96]; snprintf(nm, sizeof nm, "linear read of the same bytes, K=%d, roots %d, %d waves/CU", KK, RR, WPC); \
    run(nm, [&] { hipLaunchKernelGGL((k_lin<KK, RR>), dim3(256 * WPC), dim3(64), 0, 0, src, pitch, ntile, nchunk, root); }, rows, L); }
  LINRUN(6, 0, 8) LINRUN(6, 1, 8) LINRUN(6, 2, 8) LINRUN(6, 5, 8) LINRUN(6, 6, 8)
#define REGRUN(NV, WPC) { char nm[96]; snprintf(nm, sizeof nm, "register landing, %d chunks in flight, %d waves/CU", NV, WPC); \
    run(nm, [&] { hipLaunchKernelGGL((k_reg<NV, 6>), dim3(256 * WPC), dim3(64), 0, 0, src, pitch, ntile, nchunk, root); }, rows, L); }
  REGRUN(2, 8) REGRUN(2, 16)
  return 0;
}
