// Device runtime behind include/fdg.h: the table-walking interpreter kernel,
// the Philox leaf generator, the partial-sum reducers, the transposition kernel,
// the launch logic of the three back ends (ISA, HIP source, interpreter), the
// hiprtc/hipcc/assembler JIT with its on-device autotuner, and most C ABI entry
// points (leaf kernels and the fused step: fdg_leaf.hip; communicator: fdg_comm.cpp).
//
// Written for gfx950 only (wave64, 256 CUs, 160 KiB LDS per CU).  There is no
// CPU path here: every evaluation entry point needs a device.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#define FDG_RUNTIME_TU 1   // fdg_internal.h then also declares the helpers that need the HIP runtime types
#include "fdg_internal.h"
#include "fdg_opt.h"
#include "fdg_powi.h"

using namespace fdg;


// ============================================================================
// kernels
// ============================================================================
typedef const __attribute__((address_space(4))) uint32_t cword;  // constant AS: wave-uniform reads become s_load

__device__ __forceinline__ double fdg_block_sum256(double v, double *sh) {
  const int t = threadIdx.x;
  sh[t] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s) sh[t] = sh[t] + sh[t + s];
    __syncthreads();
  }
  double r = sh[0];
  __syncthreads();
  return r;
}

// One lane = one sample; the block walks the micro-op stream in lock step
// (no divergence: every branch below depends on stream words only).
// Per-sample values live in LDS columns lds[slot][lane] and, past the LDS
// budget, in the block's HBM panel ws[slot][lane] (both lane-contiguous, so
// every access of a wave is one coalesced 512-byte transaction).
template <int MODE, bool STAGED>
__global__ void __launch_bounds__(256)
fdg_interp(const uint32_t *code_, const double *__restrict__ leaf, long ss, long ls,
           double *__restrict__ root, long rs, long rk, const double *__restrict__ weight,
           double *__restrict__ partial, long B, double *__restrict__ ws, uint32_t lds_slots,
           uint32_t mem_slots, uint32_t L, uint32_t R) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  cword *code = (cword *)(uintptr_t)code_;
  const int t = threadIdx.x;
  const size_t per_block = (size_t)(mem_slots + (STAGED ? L : 0u) + (MODE ? R : 0u)) * 256u;
  double *mem = ws + (size_t)blockIdx.x * per_block + t;
  double *lpanel = mem + (size_t)mem_slots * 256u;
  double *accp = lpanel + (STAGED ? (size_t)L * 256u : 0u);
  double *ldsl = lds + t;
  if (MODE)
    for (uint32_t k = 0; k < R; ++k) accp[(size_t)k * 256u] = 0.0;

  const long nblk = (B + 255) / 256;
#pragma unroll 1
  for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const long b0 = blk * 256 + t;
    const bool valid = b0 < B;
    const long b = valid ? b0 : (B - 1);
    const double *lp = leaf + b * ss;
    if (STAGED) {
      // sample-major input: every lane streams its own row into the panel once,
      // so later leaf reads are lane-contiguous
      for (uint32_t i = 0; i < L; ++i) lpanel[(size_t)i * 256u] = lp[(long)i * ls];
    }
    double w = 0.0;
    if (MODE) w = valid ? (weight ? weight[b] : 1.0) : 0.0;

    auto fetch = [&](uint32_t loc) -> double {
      const uint32_t sp = loc >> 30, ix = loc & LOC_IDX_MASK;
      double v;
      // three explicit branches (the empty asm keeps the compiler from merging them into one flat load
      // through a selected generic pointer: LDS reads stay ds_read, panel reads stay global_load)
      if (sp == SP_LDS) { v = ldsl[(size_t)ix * 256u]; asm volatile("" : "+v"(v)); }
      else if (sp == SP_MEM) { v = mem[(size_t)ix * 256u]; asm volatile("" : "+v"(v)); }
      else { v = STAGED ? lpanel[(size_t)ix * 256u] : lp[(long)ix * ls]; asm volatile("" : "+v"(v)); }
      return v;
    };
    auto factor = [&](uint32_t pc) -> double {
      const uint64_t u = (uint64_t)code[pc] | ((uint64_t)code[pc + 1] << 32);
      return __longlong_as_double((long long)u);
    };

    uint32_t pc = 0;
#pragma unroll 1
    for (;;) {
      const uint32_t h = code[pc];
      const uint32_t opc = h & 15u, arg = h >> 4;
      if (opc == UOP_END) break;
      if (opc == UOP_ROOT) {
        const double v = fetch(code[pc + 1]);
        pc += 2;
        if (MODE) accp[(size_t)arg * 256u] = accp[(size_t)arg * 256u] + w * v;
        else if (valid) root[b * rs + (long)arg * rk] = v;
        continue;
      }
      if (opc == UOP_LEAF) {
        const uint32_t dst = code[pc + 1], li = code[pc + 2];
        pc += 3;
        ldsl[(size_t)(dst & LOC_IDX_MASK) * 256u] = STAGED ? lpanel[(size_t)li * 256u] : lp[(long)li * ls];
        continue;
      }
      const uint32_t dst = code[pc + 1];
      pc += 2;
      double acc;
      if (opc == UOP_POW) {
        const uint32_t w0 = code[pc++];
        acc = fdg_powi_impl(fetch(w0), (int32_t)arg - (1 << 27));
        if (w0 & LOC_FAC) { acc = acc * factor(pc); pc += 2; }
      } else {
        const uint32_t w0 = code[pc++];
        acc = fetch(w0);
        if (w0 & LOC_FAC) { acc = acc * factor(pc); pc += 2; }
        if (opc == UOP_SUM) {
#pragma unroll 1
          for (uint32_t j = 1; j < arg; ++j) {
            const uint32_t wj = code[pc++];
            double x = fetch(wj);
            if (wj & LOC_FAC) { x = x * factor(pc); pc += 2; }
            acc = acc + x;
          }
        } else if (opc == UOP_PRODI) {      // eval!'s product: prod(w_i * f_i), every operand scaled before it is folded (eval.jl:2)
#pragma unroll 1
          for (uint32_t j = 1; j < arg; ++j) {
            const uint32_t wj = code[pc++];
            double x = fetch(wj);
            if (wj & LOC_FAC) { x = x * factor(pc); pc += 2; }
            acc = acc * x;
          }
        } else {
#pragma unroll 1
          for (uint32_t j = 1; j < arg; ++j) {
            const uint32_t wj = code[pc++];
            acc = acc * fetch(wj);
            if (wj & LOC_FAC) { acc = acc * factor(pc); pc += 2; }
          }
        }
      }
      if ((dst >> 30) == SP_LDS) ldsl[(size_t)(dst & LOC_IDX_MASK) * 256u] = acc;
      else mem[(size_t)(dst & LOC_IDX_MASK) * 256u] = acc;
    }
  }
  if (MODE) {
    double *sh = lds + (size_t)lds_slots * 256u;
    for (uint32_t k = 0; k < R; ++k) {
      const double s = fdg_block_sum256(accp[(size_t)k * 256u], sh);
      if (t == 0) partial[(size_t)blockIdx.x * R + k] = s;
    }
  }
}

// acc[k] += sum over blocks of partial[blk][k], fixed order (deterministic for
// a given grid size)
__global__ void fdg_reduce_partials(const double *__restrict__ partial, uint32_t nblk, uint32_t R,
                                    double *__restrict__ acc) {
  __shared__ double sh[256];
  for (uint32_t k = blockIdx.x; k < R; k += gridDim.x) {
    double s = 0.0;
    for (uint32_t i = threadIdx.x; i < nblk; i += 256) s = s + partial[(size_t)i * R + k];
    s = fdg_block_sum256(s, sh);
    if (threadIdx.x == 0) acc[k] = acc[k] + s;
  }
}

int launch_reduce_partials(const double *partial, uint32_t nblk, uint32_t R, double *acc, hipStream_t st) {
  hipLaunchKernelGGL(fdg_reduce_partials, dim3(std::min<uint32_t>(R, 64u)), dim3(256), 0, st, partial, nblk, R, acc);
  HIP_TRY(hipGetLastError());
  return FDG_OK;
}

// acc[k] += sum over waves of partial[k][wave], fixed order (one block per root)
__global__ void __launch_bounds__(256)
fdg_reduce_lane_partials(const double *__restrict__ partial, uint32_t nwave, uint32_t R, double *__restrict__ acc) {
  __shared__ double sh[256];
  for (uint32_t k = blockIdx.x; k < R; k += gridDim.x) {
    double s = 0.0;
    for (uint32_t i = threadIdx.x; i < nwave; i += 256) s = s + partial[(size_t)k * nwave + i];
    s = fdg_block_sum256(s, sh);
    if (threadIdx.x == 0) acc[k] = acc[k] + s;
  }
}

// partial[seg][k] = sum over the segment's samples of w[b] * root_k[b]   (root k of sample b at root[k * ld + b]: the scratch
// matrix is kept column-major, so every pass over a root is a coalesced stream -- with row-major scratch the 180 roots of
// example/benchmark.jl's vertex function cost 180 strided passes and the reduction took as long as the evaluation).
// One block per (root, segment of the batch): a thread adds a dozen samples or more before the block sum (round 4; one block per 256
// samples and a block sum per root before: the 180 block sums per block made this pass a third of the accumulation's time).
__global__ void __launch_bounds__(256)
fdg_weighted_partials(const double *__restrict__ root, long ld, const double *__restrict__ weight, long B, uint32_t R, uint32_t n_seg,
                      double *__restrict__ partial, const uint8_t *__restrict__ live) {
  __shared__ double sh[256];
  const long seg_len = (((B + n_seg - 1) / n_seg) + 255) & ~255L;
  for (uint32_t id = blockIdx.x; id < R * n_seg; id += gridDim.x) {
    const uint32_t k = id % R, seg = id / R;             // neighbouring blocks: the same samples (and weights) of different roots
    if (live && !live[k]) {                              // FDG_NO_ROOT: the evaluator never wrote this column of the scratch; acc[k] stays as it is
      if (threadIdx.x == 0) partial[(size_t)seg * R + k] = 0.0;
      continue;
    }
    const double *rk = root + (size_t)k * (size_t)ld;
    const long b0 = (long)seg * seg_len, b1 = b0 + seg_len < B ? b0 + seg_len : B;
    double s = 0.0;
#pragma unroll 4
    for (long b = b0 + threadIdx.x; b < b1; b += 256L)
      s = s + (weight ? weight[b] : 1.0) * rk[b];
    s = fdg_block_sum256(s, sh);
    if (threadIdx.x == 0) partial[(size_t)seg * R + k] = s;
  }
}
// segments of the batch for fdg_weighted_partials: about 4096 blocks in all, 4096 samples per block or more, at most 2048 segments
static inline uint32_t weighted_segments(long B, uint32_t R) {
  const long by_size = (B + 4095) / 4096, by_blocks = std::max<long>(1, 4096 / std::max<uint32_t>(R, 1));
  return (uint32_t)std::max<long>(1, std::min<long>(std::min<long>(by_size, by_blocks), 2048));
}

// The same for rows that can be read 16 bytes per lane (leaf stride 1, even sample stride, 16-byte aligned base): 64 x 64 tiles, every row
// segment one 512-byte run (round 4: the 256-byte runs of the kernel below moved 2.3 TB/s read + written in front of the 4-loop vertex
// functions' kernels, which cannot overlap with it -- one wave per SIMD with 496 registers and all of the LDS).
typedef double fdg_pair_d __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256)
fdg_transpose_rows_wide(const double *__restrict__ src, long ss, double *__restrict__ dst, long ld, long n, uint32_t L) {
  __shared__ double tile[64][65];          // [column][row]
  const int t = threadIdx.x;
  const long ntile_s = (n + 63) / 64, ntile_l = (L + 63) / 64;
  for (long tid = blockIdx.x; tid < ntile_s * ntile_l; tid += gridDim.x) {
    const long s0 = (tid / ntile_l) * 64, l0 = (tid % ntile_l) * 64;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const long row = s0 + k * 8 + t / 32, col = l0 + 2 * (t % 32);
      if (row < n && col + 1 < (long)L) {
        const fdg_pair_d v = __builtin_nontemporal_load((const fdg_pair_d *)(src + row * ss + col));
        tile[2 * (t % 32)][k * 8 + t / 32] = v.x;
        tile[2 * (t % 32) + 1][k * 8 + t / 32] = v.y;
      } else if (row < n && col < (long)L) {
        tile[2 * (t % 32)][k * 8 + t / 32] = __builtin_nontemporal_load(src + row * ss + col);
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const long col = l0 + k * 4 + t / 64, row = s0 + t % 64;
      if (row < n && col < (long)L) dst[col * ld + row] = tile[k * 4 + t / 64][t % 64];
    }
    __syncthreads();
  }
}

// dst[b * rs + k * rk] = src[k * ld + b]  for b < n, k < R: column-major root scratch -> the caller's (usually row-major) root matrix,
// 64 x 32 tiles through LDS: 512-byte runs read along a root, 256-byte runs written along a row.
__global__ void __launch_bounds__(256)
fdg_transpose_from_leaf_major(const double *__restrict__ src, long ld, double *__restrict__ dst, long rs, long rk, long n, uint32_t R,
                              const uint8_t *__restrict__ live) {
  __shared__ double tile[32][65];
  const int t = threadIdx.x;
  const long ntile_s = (n + 63) / 64, ntile_k = (R + 31) / 32;
  for (long tid = blockIdx.x; tid < ntile_s * ntile_k; tid += gridDim.x) {
    const long s0 = (tid / ntile_k) * 64, k0 = (tid % ntile_k) * 32;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long k = k0 + j * 4 + t / 64, row = s0 + t % 64;
      if (row < n && k < (long)R) tile[j * 4 + t / 64][t % 64] = src[k * ld + row];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long row = s0 + j * 8 + t / 32, k = k0 + t % 32;
      if (row < n && k < (long)R && (!live || live[k])) dst[row * rs + k * rk] = tile[t % 32][j * 8 + t / 32];   // FDG_NO_ROOT: root[k] left untouched (fdg.h)
    }
    __syncthreads();
  }
}

// dst[l * ld + b] = src[b * ss + l * ls]  for b < n, l < L: (usually sample-major, ls = 1) rows -> leaf-major columns,
// 64 x 32 tiles through LDS so that both the reads (256 B runs along a row) and the writes
// (512 B runs along a column) are coalesced.
__global__ void __launch_bounds__(256)
fdg_transpose_to_leaf_major(const double *__restrict__ src, long ss, long ls, double *__restrict__ dst, long ld, long n,
                            uint32_t L) {
  __shared__ double tile[32][65];
  const int t = threadIdx.x;
  const long ntile_s = (n + 63) / 64, ntile_l = (L + 31) / 32;
  for (long tid = blockIdx.x; tid < ntile_s * ntile_l; tid += gridDim.x) {
    const long s0 = (tid / ntile_l) * 64, l0 = (tid % ntile_l) * 32;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const long row = s0 + k * 8 + t / 32, col = l0 + t % 32;
      if (row < n && col < L) tile[t % 32][k * 8 + t / 32] = __builtin_nontemporal_load(src + row * ss + col * ls);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const long col = l0 + k * 4 + t / 64, row = s0 + t % 64;
      if (row < n && col < L) dst[col * ld + row] = tile[k * 4 + t / 64][t % 64];
    }
    __syncthreads();
  }
}

// ---- a matrix of the reference's layouts <-> the tile-major batch (round 6: fdg_repack_tile_major / fdg_unpack_tile_major) -----------------
// tiled[t][c][l] (64 samples of column c of tile t contiguous) against a strided matrix m[b * ss + c * cs], b = 64 t + l.
// Column-major matrices (ss == 1: a Julia B x C Matrix) need no transposition: the 64 samples of (t, c) are one 512-byte run on both sides,
// 32 lanes of 16 bytes each, eight runs per block and pass (runs cut short by the matrix's end or not 16-byte aligned go lane by lane).
template <bool TO_TILED>
__global__ void __launch_bounds__(256)
fdg_repack_runs(double *__restrict__ mat, long cs, double *__restrict__ tiled, long n, uint32_t C) {
  const long ntile = (n + 63) / 64, nrun = ntile * (long)C;
  const int sub = threadIdx.x >> 5, l2 = (threadIdx.x & 31) * 2;
  const bool aligned = ((((uintptr_t)mat) & 15) == 0) && ((cs & 1) == 0);
  for (long r = (long)blockIdx.x * 8 + sub; r < nrun; r += (long)gridDim.x * 8) {
    const long t = r / C, c = r - t * C, b = 64 * t + l2;
    double *m = mat + c * cs + b, *q = tiled + r * 64 + l2;
    if (b + 1 < n && aligned) {
      if (TO_TILED) __builtin_nontemporal_store(__builtin_nontemporal_load((const fdg_pair_d *)m), (fdg_pair_d *)q);
      else __builtin_nontemporal_store(__builtin_nontemporal_load((const fdg_pair_d *)q), (fdg_pair_d *)m);
    } else {
      for (int k = 0; k < 2; ++k) if (b + k < n) { if (TO_TILED) q[k] = m[k]; else m[k] = q[k]; }
    }
  }
}
// Any other strides (row-major [B, C]: cs == 1) go through a 64 x 64 LDS tile: rows read / written along their columns (256-byte runs and
// up), tile-major runs of 512 bytes on the other side.
template <bool TO_TILED>
__global__ void __launch_bounds__(256)
fdg_repack_transpose(double *__restrict__ mat, long ss, long cs, double *__restrict__ tiled, long n, uint32_t C) {
  __shared__ double tile[64][65];          // [column][row]
  const int t = threadIdx.x;
  const long ntile_s = (n + 63) / 64, ntile_c = ((long)C + 63) / 64;
  const bool wide = cs == 1 && (ss & 1) == 0 && (((uintptr_t)mat) & 15) == 0;
  for (long tid = blockIdx.x; tid < ntile_s * ntile_c; tid += gridDim.x) {
    const long ts = tid / ntile_c, s0 = ts * 64, c0 = (tid % ntile_c) * 64;
    if (TO_TILED) {
      if (wide) {       // contiguous, 16-byte aligned rows: two columns per lane (512-byte runs along a row)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const long row = s0 + k * 8 + t / 32, col = c0 + 2 * (t % 32);
          if (row < n && col + 1 < (long)C) {
            const fdg_pair_d v = __builtin_nontemporal_load((const fdg_pair_d *)(mat + row * ss + col));
            tile[2 * (t % 32)][k * 8 + t / 32] = v.x;
            tile[2 * (t % 32) + 1][k * 8 + t / 32] = v.y;
          } else if (row < n && col < (long)C) {
            tile[2 * (t % 32)][k * 8 + t / 32] = __builtin_nontemporal_load(mat + row * ss + col);
          }
        }
      } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const long row = s0 + k * 4 + t / 64, col = c0 + t % 64;
        if (row < n && col < (long)C) tile[t % 64][k * 4 + t / 64] = __builtin_nontemporal_load(mat + row * ss + col * cs);
      }
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const long col = c0 + k * 4 + t / 64, row = s0 + t % 64;
        if (row < n && col < (long)C) tiled[(ts * (long)C + col) * 64 + t % 64] = tile[k * 4 + t / 64][t % 64];
      }
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const long col = c0 + k * 4 + t / 64, row = s0 + t % 64;
        if (row < n && col < (long)C) tile[k * 4 + t / 64][t % 64] = __builtin_nontemporal_load(tiled + (ts * (long)C + col) * 64 + t % 64);
      }
      __syncthreads();
      if (wide) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const long row = s0 + k * 8 + t / 32, col = c0 + 2 * (t % 32);
          if (row < n && col + 1 < (long)C) {
            fdg_pair_d v; v.x = tile[2 * (t % 32)][k * 8 + t / 32]; v.y = tile[2 * (t % 32) + 1][k * 8 + t / 32];
            __builtin_nontemporal_store(v, (fdg_pair_d *)(mat + row * ss + col));
          } else if (row < n && col < (long)C) {
            mat[row * ss + col] = tile[2 * (t % 32)][k * 8 + t / 32];
          }
        }
      } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const long row = s0 + k * 4 + t / 64, col = c0 + t % 64;
        if (row < n && col < (long)C) mat[row * ss + col * cs] = tile[t % 64][k * 4 + t / 64];
      }
      }
    }
    __syncthreads();
  }
}

// Philox4x32-10 (Salmon et al., SC'11), key = seed, counter = (sample, leaf)
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__global__ void __launch_bounds__(256)
fdg_fill_uniform(double *__restrict__ leaf, long B, uint32_t L, long ss, long ls, uint64_t seed,
                 uint64_t off, int leaf_fastest) {
  const long total = B * (long)L;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256L) {
    long b, i;
    if (leaf_fastest) { b = e / L; i = e - b * L; } else { i = e / B; b = e - i * B; }
    const uint64_t s = off + (uint64_t)b;
    uint32_t o[4];
    philox4x32_10((uint32_t)s, (uint32_t)(s >> 32), (uint32_t)i, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
    const uint64_t m = ((uint64_t)(o[0] >> 5) << 26) | (uint64_t)(o[1] >> 6);  // 53 random bits
    leaf[b * ss + i * ls] = (double)m * 0x1.0p-53;
  }
}

// the same values into a tile-major batch: sample b = 64 t + l lives at leaf[t * lts + l * ss + i * ls]
__global__ void __launch_bounds__(256)
fdg_fill_uniform_tiled(double *__restrict__ leaf, long B, uint32_t L, long ss, long ls, long lts, uint64_t seed, uint64_t off) {
  const long ntile = (B + 63) / 64, total = ntile * 64L * (long)L;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256L) {
    const long l = e & 63, ti = e >> 6, t = ti / L, i = ti - t * L, b = 64 * t + l;
    if (b >= B) continue;
    const uint64_t s = off + (uint64_t)b;
    uint32_t o[4];
    philox4x32_10((uint32_t)s, (uint32_t)(s >> 32), (uint32_t)i, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
    const uint64_t m = ((uint64_t)(o[0] >> 5) << 26) | (uint64_t)(o[1] >> 6);
    leaf[t * lts + l * ss + i * ls] = (double)m * 0x1.0p-53;
  }
}

// Harness only: the box's streaming ceiling.  16 bytes per lane; every workgroup copies its own contiguous span, four
// 4 KB pieces in flight per wave (of the variants in tools/ubench/copy_rate.hip this one streams fastest on MI355X:
// 5.5-5.7 TB/s read + write, against 4.7-5.3 for grid-strided loops and 5.05 for hipMemcpyDtoD; 5.9-6.0 with non-temporal
// loads and stores, the default).
typedef double fdg_v2d __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ void fdg_copy16_body(const fdg_v2d *__restrict__ src, fdg_v2d *__restrict__ dst, long n16) {
  const long per = (n16 + gridDim.x - 1) / gridDim.x;
  const long lo = blockIdx.x * per, hi = lo + per < n16 ? lo + per : n16;
  auto ld = [&](long i) { return NT ? __builtin_nontemporal_load(src + i) : src[i]; };
  auto st = [&](long i, fdg_v2d v) { if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v; };
  for (long i = lo + threadIdx.x; i < hi; i += 1024) {
    const fdg_v2d a = ld(i), b = i + 256 < hi ? ld(i + 256) : a, c = i + 512 < hi ? ld(i + 512) : a, d = i + 768 < hi ? ld(i + 768) : a;
    st(i, a);
    if (i + 256 < hi) st(i + 256, b);
    if (i + 512 < hi) st(i + 512, c);
    if (i + 768 < hi) st(i + 768, d);
  }
}
__global__ void __launch_bounds__(256) fdg_copy16(const fdg_v2d *__restrict__ src, fdg_v2d *__restrict__ dst, long n16) { fdg_copy16_body<false>(src, dst, n16); }
// Harness: a read-only stream (8 bytes per lane, non-temporal, eight loads in flight per lane) -- the memory system's ceiling for the evaluator's kind of traffic,
// which is 95 % reads (round 5: 6.65 TB/s where the copy reaches 5.5-6.0; tools/ubench/stream_power.hip).  Nothing is written unless the data says so.
__global__ void __launch_bounds__(256) fdg_read8_nt(const double *__restrict__ src, double *__restrict__ sink, long n) {
  const long stride = (long)gridDim.x * 256L;
  double acc = 0.0;
  long i = blockIdx.x * 256L + threadIdx.x;
  for (; i + 7 * stride < n; i += 8 * stride) {
    double r[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) r[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += r[u];
  }
  for (; i < n; i += stride) acc += __builtin_nontemporal_load(src + i);
  if (acc == 0x1.23456789abcdp+777) sink[0] = acc;
}
__global__ void __launch_bounds__(256) fdg_copy16_nt(const fdg_v2d *__restrict__ src, fdg_v2d *__restrict__ dst, long n16) { fdg_copy16_body<true>(src, dst, n16); }

// ============================================================================
// host side
// ============================================================================
int ensure_device(fdg_graph *g) {
  if (g->device >= 0) return FDG_OK;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    set_error("no HIP device available (the evaluator has no CPU fallback)");
    return FDG_E_NO_DEVICE;
  }
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, dev));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_error(std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    return FDG_E_NO_DEVICE;
  }
  g->device = dev;
  g->n_cu = prop.multiProcessorCount;
  return FDG_OK;
}

static void free_ws_set(fdg_ws_set &w) {
  if (w.d_ws) hipFree(w.d_ws);
  if (w.d_ws2) hipFree(w.d_ws2);
  if (w.d_ws3) hipFree(w.d_ws3);
  if (w.d_ws4) hipFree(w.d_ws4);
  if (w.s2) {
    hipStreamSynchronize((hipStream_t)w.s2);
    hipStreamDestroy((hipStream_t)w.s2);
    for (int i = 0; i < 2; ++i) { hipEventDestroy((hipEvent_t)w.ev_t[i]); hipEventDestroy((hipEvent_t)w.ev_k[i]); }
    hipEventDestroy((hipEvent_t)w.ev_in);
  }
  w = fdg_ws_set();
}
static void park_current_ws(fdg_graph *g, fdg_ws_set &w) {
  w.key = g->ws_key;
  w.d_ws = g->d_ws; w.ws_bytes = g->ws_bytes; w.d_ws2 = g->d_ws2; w.ws2_bytes = g->ws2_bytes;
  w.d_ws3 = g->d_ws3; w.ws3_bytes = g->ws3_bytes; w.d_ws4 = g->d_ws4; w.ws4_bytes = g->ws4_bytes;
  w.s2 = g->s2; w.ev_in = g->ev_in;
  for (int i = 0; i < 2; ++i) { w.ev_t[i] = g->ev_t[i]; w.ev_k[i] = g->ev_k[i]; }
  w.last_use = g->ws_clock;
}
static void load_ws(fdg_graph *g, const fdg_ws_set &w) {
  g->d_ws = w.d_ws; g->ws_bytes = w.ws_bytes; g->d_ws2 = w.d_ws2; g->ws2_bytes = w.ws2_bytes;
  g->d_ws3 = w.d_ws3; g->ws3_bytes = w.ws3_bytes; g->d_ws4 = w.d_ws4; g->ws4_bytes = w.ws4_bytes;
  g->s2 = w.s2; g->ev_in = w.ev_in;
  for (int i = 0; i < 2; ++i) { g->ev_t[i] = w.ev_t[i]; g->ev_k[i] = w.ev_k[i]; }
}
// One scratch set per caller stream (include/fdg.h: the device entry points may be called concurrently on different
// streams and from different threads; calls on one stream are ordered by the stream).  At most 8 sets are kept: the
// least recently used one is released (after a device synchronisation) when a ninth stream shows up.
int fdg_bind_stream_ws(fdg_graph *g, void *stream) {
  g->ws_clock++;
  if (g->ws_bound && g->ws_key == stream) return FDG_OK;
  if (g->ws_bound) {
    fdg_ws_set w;
    park_current_ws(g, w);
    g->ws_pool.push_back(w);
  }
  g->ws_bound = true;
  g->ws_key = stream;
  for (size_t i = 0; i < g->ws_pool.size(); ++i)
    if (g->ws_pool[i].key == stream) {
      load_ws(g, g->ws_pool[i]);
      g->ws_pool.erase(g->ws_pool.begin() + (long)i);
      return FDG_OK;
    }
  load_ws(g, fdg_ws_set());
  if (g->ws_pool.size() >= 8) {
    size_t lru = 0;
    for (size_t i = 1; i < g->ws_pool.size(); ++i) if (g->ws_pool[i].last_use < g->ws_pool[lru].last_use) lru = i;
    HIP_TRY(hipDeviceSynchronize());
    free_ws_set(g->ws_pool[lru]);
    g->ws_pool.erase(g->ws_pool.begin() + (long)lru);
  }
  return FDG_OK;
}

int ensure_ws(fdg_graph *g, size_t bytes) {
  if (g->ws_bytes >= bytes && g->d_ws) return FDG_OK;
  if (g->d_ws) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(g->d_ws)); g->d_ws = nullptr; g->ws_bytes = 0; }
  hipError_t e = hipMalloc(&g->d_ws, bytes);
  if (e != hipSuccess) { set_error("hipMalloc(workspace) failed: " + std::string(hipGetErrorString(e))); return FDG_E_NOMEM; }
  g->ws_bytes = bytes;
  return FDG_OK;
}

static int ensure_code(fdg_graph *g) {
  if (g->d_code) return FDG_OK;
  const size_t nb = g->prog.code.size() * sizeof(uint32_t);
  HIP_TRY(hipMalloc(&g->d_code, nb));
  HIP_TRY(hipMemcpy(g->d_code, g->prog.code.data(), nb, hipMemcpyHostToDevice));
  return FDG_OK;
}

// Device mask of the roots that exist (root_slot[k] != FDG_NO_ROOT), for the passes over the column-major root scratch: null when every root
// exists.  (the scratch's columns of missing roots are never written by the evaluator; copying or summing them would overwrite
// root[k] / add garbage to acc[k], which fdg.h promises to leave alone.)
static int root_live_mask(fdg_graph *g, const uint8_t **out) {
  *out = nullptr;
  const Lowered &p = g->prog;
  bool any = false;
  for (uint32_t k = 0; k < p.R; ++k) any = any || p.root_slot[k] == FDG_NO_ROOT;
  if (!any) return FDG_OK;
  if (!g->d_root_live) {
    std::vector<uint8_t> m(p.R);
    for (uint32_t k = 0; k < p.R; ++k) m[k] = p.root_slot[k] != FDG_NO_ROOT;
    HIP_TRY(hipMalloc(&g->d_root_live, p.R));
    HIP_TRY(hipMemcpy(g->d_root_live, m.data(), p.R, hipMemcpyHostToDevice));
  }
  *out = (const uint8_t *)g->d_root_live;
  return FDG_OK;
}

static int ensure_module(fdg_graph *g) {
  if (g->module || g->code_object.empty()) return FDG_OK;
  hipModule_t m;
  hipError_t e = hipModuleLoadData(&m, g->code_object.data());
  if (e != hipSuccess) { set_error("hipModuleLoadData failed: " + std::string(hipGetErrorString(e))); return FDG_E_JIT; }
  if (g->isa) {
    hipFunction_t f;
    HIP_TRY(hipModuleGetFunction(&f, m, "fdg_isa_eval"));
    g->module = m; g->fn_isa = f;
    { hipFunction_t fn; g->fn_isa_nt = hipModuleGetFunction(&fn, m, "fdg_isa_eval_nt") == hipSuccess ? (void *)fn : nullptr; (void)hipGetLastError(); }
    { hipFunction_t fn; g->fn_isa_acc_nt = (g->has_acc && hipModuleGetFunction(&fn, m, "fdg_isa_eval_acc_nt") == hipSuccess) ? (void *)fn : nullptr; (void)hipGetLastError(); }
    if (g->has_w2) { hipFunction_t f2; HIP_TRY(hipModuleGetFunction(&f2, m, "fdg_isa_eval_w2")); g->fn_isa_w2 = f2; }
    if (g->has_acc) { hipFunction_t f3; HIP_TRY(hipModuleGetFunction(&f3, m, "fdg_isa_eval_acc")); g->fn_isa_acc = f3; }
    if (g->has_rm) { hipFunction_t f4; HIP_TRY(hipModuleGetFunction(&f4, m, "fdg_isa_eval_rm")); g->fn_isa_rm = f4; }
    if (g->has_rm_acc) { hipFunction_t f6; HIP_TRY(hipModuleGetFunction(&f6, m, "fdg_isa_eval_rm_acc")); g->fn_isa_rm_acc = f6; }
    if (g->has_coop) { hipFunction_t f5; HIP_TRY(hipModuleGetFunction(&f5, m, "fdg_isa_eval_coop")); g->fn_isa_coop = f5; }
    if (g->has_pool) { hipFunction_t f6; HIP_TRY(hipModuleGetFunction(&f6, m, "fdg_isa_eval_pool")); g->fn_isa_pool = f6; }
    if (g->has_rl) { hipFunction_t f7; HIP_TRY(hipModuleGetFunction(&f7, m, "fdg_isa_eval_rl")); g->fn_isa_rl = f7; }
    if (g->has_rl_acc) { hipFunction_t f8; HIP_TRY(hipModuleGetFunction(&f8, m, "fdg_isa_eval_rl_acc")); g->fn_isa_rl_acc = f8; }
    return FDG_OK;
  }
  hipFunction_t f1, f2;
  HIP_TRY(hipModuleGetFunction(&f1, m, "fdg_spec_sm"));
  HIP_TRY(hipModuleGetFunction(&f2, m, "fdg_spec_gen"));
  g->module = m; g->fn_eval_sm = f1; g->fn_eval_gen = f2;
  int v = 0;
  if (hipFuncGetAttribute(&v, HIP_FUNC_ATTRIBUTE_NUM_REGS, f1) == hipSuccess) g->spec_vgpr = (uint32_t)v;
  if (hipFuncGetAttribute(&v, HIP_FUNC_ATTRIBUTE_SHARED_SIZE_BYTES, f1) == hipSuccess) g->spec_lds = (uint32_t)v;
  if (hipFuncGetAttribute(&v, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, f1) == hipSuccess) g->spec_scratch = (uint32_t)v;
  return FDG_OK;
}

// The launch path's view of the handle's options (fdg_internal.h: fdg_launch_cfg).  Called under g->mu (or before the handle is shared).
void parse_launch_cfg(fdg_graph *g) {
  fdg_launch_cfg c;
  auto get = [&](const char *n) -> const char * { auto it = g->knobs.find(n); return it == g->knobs.end() ? nullptr : it->second.c_str(); };
  c.no_rl = get("FDG_ISA_NO_RL") != nullptr;
  c.no_fused_acc = get("FDG_ISA_NO_FUSED_ACC") != nullptr;
  c.no_pool = get("FDG_ISA_NO_POOL") != nullptr;
  c.pool_no_acc = get("FDG_ISA_POOL_NO_ACC") != nullptr;
  c.no_streaming = get("FDG_ISA_NO_STREAMING") != nullptr;
  c.no_w2 = get("FDG_ISA_NO_W2") != nullptr;
  c.no_coop = get("FDG_ISA_NO_COOP") != nullptr;
  c.no_rm = get("FDG_ISA_NO_RM") != nullptr;
  c.transpose_narrow = get("FDG_TRANSPOSE_NARROW") != nullptr;
  if (const char *e = get("FDG_ROOT_SCRATCH_MIN")) c.root_scratch_min = (uint32_t)std::max(0, std::atoi(e));
  if (const char *e = get("FDG_ROOT_SCRATCH_MB")) c.root_scratch_mb = (uint64_t)std::max(1, std::atoi(e));
  if (const char *e = get("FDG_ISA_WAVES_PER_CU")) c.waves_per_cu = std::max(1, std::atoi(e));
  if (const char *e = get("FDG_ISA_OVERSUB")) c.oversub = std::max(1, std::atoi(e));
  if (const char *e = get("FDG_ISA_MEM_WAVES")) c.mem_waves = std::max(0l, std::atol(e));
  if (const char *e = get("FDG_ISA_MEM_OVERSUB")) c.mem_oversub = std::max(1l, std::atol(e));
  if (const char *e = get("FDG_ISA_MEM_RATIO")) c.mem_ratio = std::atof(e);
  if (const char *e = get("FDG_SM_CHUNK_MB")) c.sm_chunk_bytes = (uint64_t)std::max(1, std::atoi(e)) << 20;
  if (const char *e = get("FDG_EVAL_CHUNK")) c.eval_chunk = std::atoll(e);
  if (const char *e = get("FDG_MC_CHUNK")) c.mc_chunk = std::atoll(e);
  g->cfg = c;
}

// mode 0: roots -> d_root; mode 1: partial sums -> d_acc
static int run(fdg_graph *g, int mode, const double *d_leaf, int64_t ss, int64_t ls, double *d_root,
               int64_t rs, int64_t rk, const double *d_weight, double *d_acc, int64_t B, hipStream_t st,
               int64_t lts = 0, int64_t rts = 0) {
  if (!g) { set_error("null handle"); return FDG_E_INVALID; }
  if (B < 0) { set_error("n_sample < 0"); return FDG_E_INVALID; }
  if (B == 0) return FDG_OK;
  if ((g->prog.L && !d_leaf) || (mode == 0 && g->prog.R && !d_root) || (mode == 1 && !d_acc)) {
    set_error("null device buffer"); return FDG_E_INVALID;
  }
  if (mode == 1 && g->prog.R == 0) return FDG_OK;        // no roots: nothing to accumulate
  std::lock_guard<std::mutex> lk(g->mu);
  fdg::KnobScope knob_scope(&g->knobs);
  { const int rcb = fdg_bind_stream_ws(g, (void *)st); if (rcb) return rcb; }
  return fdg_run_locked(g, mode, d_leaf, ss, ls, d_root, rs, rk, d_weight, d_acc, B, st, lts, rts);
}

// launch of the compiler-scheduled per-graph kernels (fdg_spec_sm / fdg_spec_gen)
static int launch_hip_source(fdg_graph *g, hipFunction_t fn, int mode, const double *d_leaf, int64_t ss, int64_t ls, double *d_root,
                             int64_t rs, int64_t rk, const double *d_weight, double *d_acc, int64_t B, hipStream_t st) {
  const uint32_t R = g->prog.R;
  const long nblk = (long)((B + 255) / 256);
  const long grid = std::min<long>(nblk, (long)g->n_cu * 8);
  double *partial = nullptr;
  if (mode == 1) {
    int rc = ensure_ws(g, (size_t)grid * R * sizeof(double));
    if (rc) return rc;
    partial = (double *)g->d_ws;
  }
  long a_ss = ss, a_ls = ls, a_rs = rs, a_rk = rk, a_B = B;
  int a_mode = mode;
  void *args[] = {(void *)&d_leaf, &a_ss, &a_ls, (void *)&d_root, &a_rs, &a_rk, (void *)&d_weight,
                  (void *)&partial, &a_B, &a_mode};
  HIP_TRY(hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, 256, 1, 1, 0, st, args, nullptr));
  if (mode == 1) {
    hipLaunchKernelGGL(fdg_reduce_partials, dim3(std::min<uint32_t>(R, 64u)), dim3(256), 0, st, partial,
                       (uint32_t)grid, R, d_acc);
    HIP_TRY(hipGetLastError());
  }
  return FDG_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The launch path.  fdg_run_locked validates the call and hands it to ONE of the functions below -- one per kernel family / variant
// (the 400-line function this used to be is split; nothing here looks an option up by name, see fdg_launch_cfg).
// ---------------------------------------------------------------------------------------------------------------------------------
namespace {
// the arguments of one evaluation call (mode 0: roots -> d_root; mode 1: acc[k] += sum_b w_b root_k(b))
struct RunArgs {
  int mode;
  const double *d_leaf; int64_t ss, ls;
  double *d_root; int64_t rs, rk;
  const double *d_weight; double *d_acc;
  int64_t B; hipStream_t st;
  int64_t lts, rts;                 // tile strides of a tile-major batch (0: a plain strided matrix)
};
inline bool root_stride_ok(const RunArgs &a) { return !(a.rs < 0 || a.rs >= (1ll << 23)); }

int ensure_root_scratch(fdg_graph *g, size_t need) {
  if (g->ws2_bytes >= need) return FDG_OK;
  if (g->d_ws2) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(g->d_ws2)); g->d_ws2 = nullptr; g->ws2_bytes = 0; }
  if (hipMalloc(&g->d_ws2, need) != hipSuccess) { set_error("hipMalloc(root scratch) failed"); return FDG_E_NOMEM; }
  g->ws2_bytes = need;
  return FDG_OK;
}

// Roots into a ROW-MAJOR matrix (compile_Python's [B, R]) of a graph with many roots: a root store of the kernels is 64 lanes x 8 bytes, each in
// another row -- with R = 180 (example/benchmark.jl's vertex function) the stores doubled the kernel's time.  Such calls evaluate chunk by chunk
// into the column-major root scratch (every store one 512-byte run) and a transposition writes the caller's rows (round 4).
int run_through_root_scratch(fdg_graph *g, const RunArgs &a) {
  const uint32_t R = g->prog.R;
  long Bc = std::max<long>(64, (long)((g->cfg.root_scratch_mb << 20) / (8ull * R)) & ~63l);
  Bc = std::min<long>(Bc, (long)((a.B + 63) & ~(int64_t)63));
  int rc = ensure_root_scratch(g, (size_t)Bc * R * sizeof(double) + (size_t)2048 * R * sizeof(double));
  if (rc) return rc;
  double *scratch = (double *)g->d_ws2;
  const uint8_t *live = nullptr;
  rc = root_live_mask(g, &live);
  if (rc) return rc;
  for (long c0 = 0; c0 < (long)a.B; c0 += Bc) {
    const long n = std::min<long>(Bc, (long)a.B - c0);
    const double *lf = a.lts ? a.d_leaf + (size_t)(c0 / 64) * (size_t)a.lts : a.d_leaf + (size_t)c0 * (size_t)a.ss;
    rc = fdg_run_locked(g, 0, lf, a.ss, a.ls, scratch, 1, Bc, nullptr, nullptr, n, a.st, a.lts, 0);
    if (rc) return rc;
    const long ntile = ((n + 63) / 64) * ((R + 31) / 32);
    hipLaunchKernelGGL(fdg_transpose_from_leaf_major, dim3((unsigned)std::min<long>(ntile, (long)g->n_cu * 16)), dim3(256), 0, a.st,
                       scratch, Bc, a.d_root + (size_t)c0 * (size_t)a.rs, (long)a.rs, (long)a.rk, n, R, live);
    HIP_TRY(hipGetLastError());
  }
  return FDG_OK;
}

// sample-major input of an ISA handle that carries the HIP-source companion: its lanes read their own rows; no transposition pass
int run_companion(fdg_graph *g, const RunArgs &a) {
  if (!g->alt_module) {
    hipModule_t m; hipFunction_t f1, f2;
    hipError_t e = hipModuleLoadData(&m, g->alt_code.data());
    if (e != hipSuccess) { set_error("hipModuleLoadData failed: " + std::string(hipGetErrorString(e))); return FDG_E_JIT; }
    HIP_TRY(hipModuleGetFunction(&f1, m, "fdg_spec_sm"));
    HIP_TRY(hipModuleGetFunction(&f2, m, "fdg_spec_gen"));
    g->alt_module = m; g->fn_alt_sm = f1; g->fn_alt_gen = f2;
  }
  g->last_kernel = "fdg_spec_sm";
  return launch_hip_source(g, (hipFunction_t)g->fn_alt_sm, a.mode, a.d_leaf, a.ss, a.ls, a.d_root, a.rs, a.rk, a.d_weight, a.d_acc, a.B, a.st);
}

// the generic table interpreter (no JIT)
int run_interpreter(fdg_graph *g, const RunArgs &a) {
  const Lowered &p = g->prog;
  const uint32_t R = p.R;
  const int mode = a.mode;
  const long nblk = (long)((a.B + 255) / 256);
  g->last_kernel = "fdg_interp";
  int rc = ensure_code(g);
  if (rc) return rc;
  const bool staged = (a.ls == 1 && a.ss != 1 && p.L > 0);
  const size_t lds_bytes = ((size_t)p.lds_slots + (mode ? 1u : 0u)) * 256u * sizeof(double);
  int per_cu = lds_bytes ? (int)std::min<size_t>(8, (160u * 1024u) / lds_bytes) : 8;
  if (per_cu < 1) { set_error("interpreter LDS budget exceeded"); return FDG_E_INTERNAL; }
  const long grid = std::min<long>(nblk, (long)g->n_cu * per_cu);
  const size_t per_block = (size_t)(p.mem_slots + (staged ? p.L : 0u) + (mode ? R : 0u)) * 256u;
  const size_t ws_doubles = (size_t)grid * per_block + (mode ? (size_t)grid * R : 0u);
  rc = ensure_ws(g, std::max<size_t>(ws_doubles, 1) * sizeof(double));
  if (rc) return rc;
  double *ws = (double *)g->d_ws;
  double *partial = ws + (size_t)grid * per_block;
  const uint32_t *code = (const uint32_t *)g->d_code;
#define FDG_LAUNCH(M, S)                                                                              \
  do {                                                                                                \
    if (lds_bytes > 64 * 1024)                                                                        \
      HIP_TRY(hipFuncSetAttribute((const void *)fdg_interp<M, S>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  (int)lds_bytes));                                                   \
    hipLaunchKernelGGL((fdg_interp<M, S>), dim3((unsigned)grid), dim3(256), lds_bytes, a.st, code, a.d_leaf, \
                       (long)a.ss, (long)a.ls, a.d_root, (long)a.rs, (long)a.rk, a.d_weight, partial, (long)a.B, ws,  \
                       p.lds_slots, p.mem_slots, p.L, R);                                              \
  } while (0)
  if (mode == 0) { if (staged) FDG_LAUNCH(0, true); else FDG_LAUNCH(0, false); }
  else { if (staged) FDG_LAUNCH(1, true); else FDG_LAUNCH(1, false); }
#undef FDG_LAUNCH
  HIP_TRY(hipGetLastError());
  if (mode == 1) {
    hipLaunchKernelGGL(fdg_reduce_partials, dim3(std::min<uint32_t>(R, 64u)), dim3(256), 0, a.st, partial,
                       (uint32_t)grid, R, a.d_acc);
    HIP_TRY(hipGetLastError());
  }
  return FDG_OK;
}

// ---- the kernels of the optimizing back end (per-graph gfx950 assembly) ---------------------------------------------------------
// One call's launch plan: grids and workspace offsets of the variants the handle carries, then one method per variant.
struct IsaRun {
  fdg_graph *g;
  const RunArgs &a;
  const Lowered &p;
  const uint32_t R;
  const bool tiled;
  long ntiles = 0, grid = 0, grid2 = 0, grid3 = 0, grid4 = 0, grid5 = 0;
  size_t panel_all = 0;
  bool pool_ok = false, fused_acc = false, wide_ss = false, named = false;
  double *roots = nullptr;          // where mode 1 without fused accumulation puts the roots: the column-major scratch (roots[k * a_rk + b])
  long a_rs = 0, a_rk = 0;

  IsaRun(fdg_graph *g_, const RunArgs &a_) : g(g_), a(a_), p(g_->prog), R(g_->prog.R), tiled(a_.lts != 0 || a_.rts != 0) {}

  // resident waves: one wave per workgroup; bounded by VGPRs, LDS and 32 waves/CU
  uint32_t waves_per_cu(uint32_t vgpr, uint32_t lds_bytes) const {
    const uint32_t valloc = std::max<uint32_t>(8, (vgpr + 7) & ~7u);
    uint32_t per_cu = std::min<uint32_t>(8, 512 / valloc) * 4;
    if (lds_bytes) per_cu = std::min<uint32_t>(per_cu, (160u * 1024u) / lds_bytes);
    per_cu = std::max<uint32_t>(1, std::min<uint32_t>(per_cu, 32));
    if (g->cfg.waves_per_cu > 0) per_cu = (uint32_t)g->cfg.waves_per_cu;
    return per_cu;
  }
  // A persistent wave walks tiles w, w + n, ...  With several times as many workgroups as are resident at once the later ones
  // start as the first finish, which evens out waves that progress at different speeds (oversubscription x8: +2-3 % on the
  // graphs that run against the power budget, +1 % on the headline; profiles/r03_log_oversubscription.txt).  Only while every
  // workgroup keeps a few tiles and the spill panels / partial sums that are sized by the grid stay small.
  long oversub(long resident, size_t bytes_per_wg) const {
    if (g->cfg.waves_per_cu > 0) return resident;
    long f = g->cfg.oversub > 0 ? g->cfg.oversub : 8;
    while (f > 1 && (ntiles < resident * f * 4 || bytes_per_wg * (size_t)(resident * f) > ((size_t)32 << 20))) f >>= 1;
    return resident * f;
  }
  // Graphs bound by memory (fewer than 2.5 executed fold steps per algorithmic byte) stream faster from FEWER resident waves: four per CU
  // (one per SIMD) keep 4 x L x 512 bytes in flight per CU -- enough for the latency-bandwidth product -- and the memory system sees a
  // quarter of the concurrent streams (round 4, profiles/r04_log_waves.txt: headline +3-5 %, the 2-loop graph +6-15 %); the graphs at or
  // above the ridge want every wave they can get.  Options FDG_ISA_MEM_WAVES / FDG_ISA_MEM_OVERSUB / FDG_ISA_MEM_RATIO override (0 waves = off).
  long shape(uint64_t valu, uint64_t bytes, uint32_t vgpr, uint32_t lds, size_t bytes_per_wg, bool accumulating = false) const {
    const long full = (long)waves_per_cu(vgpr, lds);
    // (a wave of a tiny graph keeps little in flight; without root stores more of them help: the 2-loop graph accumulates at 0.84 instead of
    //  0.73, profiles/r04_log_tiny_acc_waves.txt)
    const long mem_waves = g->cfg.mem_waves >= 0 ? g->cfg.mem_waves : (p.L >= 24 ? 4 : (accumulating ? 8 : 5));
    if (mem_waves > 0 && g->cfg.waves_per_cu <= 0 && g->cfg.oversub <= 0 && bytes && (double)valu < g->cfg.mem_ratio * (double)bytes && full > mem_waves) {
      long f = g->cfg.mem_oversub;
      while (f > 1 && ntiles < (long)g->n_cu * mem_waves * f * 4) f >>= 1;
      return (long)g->n_cu * mem_waves * f;
    }
    return oversub((long)g->n_cu * full, bytes_per_wg);
  }
  // a matrix whose 64-sample tiles are whole 128-byte lines (samples of a column contiguous, column stride a multiple of 16
  // doubles, base on a line): the streaming variants may be used (non-temporal accesses would fetch a shared line twice)
  bool line_aligned(const void *base, long sample_stride, long col_stride) const {
    return sample_stride == 1 && (col_stride & 15) == 0 && ((uintptr_t)base & 127) == 0 && !g->cfg.no_streaming;
  }

  int plan() {
    ntiles = (long)((a.B + 63) / 64);
    grid = std::min<long>(ntiles, shape(g->st_valu[0], 8ull * (p.L + R), g->isa_vgpr, g->isa_lds_bytes, (size_t)g->isa_mem_slots * 512u));
    grid2 = g->has_w2 ? (long)g->n_cu * waves_per_cu(g->isa2_vgpr, g->isa2_lds_bytes) : 0;
    const size_t panel = std::max((size_t)std::max<uint32_t>(g->isa_mem_slots, 1) * 512u * (size_t)grid,
                                  (size_t)std::max<uint32_t>(g->isa2_mem_slots, 1) * 1024u * (size_t)grid2);
    // pooled cooperative variant: full tiles of batches whose samples of a leaf are contiguous and whose leaves lie within 2 GB of the tile's first
    pool_ok = g->has_pool && g->fn_isa_pool && a.ss == 1 && a.ls > 0 && (g->pool_unit == 1 || a.ls == 64) &&
              (uint64_t)a.ls * 8u * (uint64_t)std::max<uint32_t>(p.L, 1) < (1ull << 31) && a.B >= 64 && !g->cfg.no_pool;
    // (a graph that has the pooled variant accumulates through it and the root scratch: its fused-accumulation program, with R + 2 fewer value
    //  registers and no pool, runs the 4-loop GV vertex function at 0.87e8 samples/s where the pooled evaluation + the weighted sum do 1.3e8)
    fused_acc = a.mode == 1 && g->has_acc && !g->cfg.no_fused_acc && !(pool_ok && !g->cfg.pool_no_acc);
    grid3 = g->has_acc ? shape(g->st_valu[1], 8ull * p.L, g->isa3_vgpr, g->isa3_lds_bytes, ((size_t)g->isa3_mem_slots + R) * 512u, true) : 0;
    const size_t panel3 = (size_t)std::max<uint32_t>(g->isa3_mem_slots, 1) * 512u * (size_t)grid3;
    grid4 = g->has_rm ? (long)g->n_cu * waves_per_cu(g->isa4_vgpr, g->isa4_lds_bytes) : 0;
    const size_t panel4 = (size_t)std::max<uint32_t>(g->isa4_mem_slots, 1) * 512u * (size_t)grid4;
    grid5 = g->has_rm_acc ? (long)g->n_cu * waves_per_cu(g->isa5_vgpr, g->isa5_lds_bytes) : 0;
    const size_t panel5 = (size_t)std::max<uint32_t>(g->isa5_mem_slots, 1) * 512u * (size_t)grid5;
    panel_all = (std::max(std::max(panel, panel3), std::max(panel4, panel5)) + 4095) & ~(size_t)4095;
    int rc = ensure_ws(g, panel_all + (size_t)std::max(grid3, grid5) * R * 512u + 4096);
    if (rc) return rc;
    roots = a.d_root; a_rs = (long)a.rs; a_rk = (long)a.rk;
    if (a.mode == 1 && !fused_acc) {
      rc = ensure_root_scratch(g, (size_t)((a.B + 15) & ~(int64_t)15) * std::max<uint32_t>(R, 1) * sizeof(double) + (size_t)2048 * R * sizeof(double));
      if (rc) return rc;
      roots = (double *)g->d_ws2; a_rs = 1; a_rk = (long)((a.B + 15) & ~(int64_t)15);      // column-major scratch: root k of sample b at roots[k * ld + b]
    }
    // the kernel forms a lane's offset (lane * stride * 8) in 32 bits: strides that large are brought into the
    // leaf-major workspace first (leaves) or refused (roots); neither occurs with the layouts of DESIGN.md 2
    wide_ss = a.ss < 0 || a.ss >= (1ll << 23);           // (the offset is unsigned: negative strides too)
    return FDG_OK;
  }

  // fused accumulation: acc[k] += sum_b w_b root_k(b) with per-lane accumulators inside the evaluator (fdg_isa_eval_acc[_nt])
  int launch_acc(const double *lf, long lss, long lls, const double *wt, long n, long tls = 0) {
    void *a_wsp = g->d_ws;
    double *part = (double *)((char *)g->d_ws + panel_all);
    long nwg = std::min<long>((n + 63) / 64, grid3), zero = 0;
    if (!tls) tls = 64 * lss;
    void *args[] = {(void *)&lf, &lss, &lls, (void *)&part, &zero, &zero, &a_wsp, &n, &nwg, (void *)&wt, &tls, &zero};
    void *fn = g->fn_isa_acc_nt && line_aligned(lf, lss, lls) && (tls & 15) == 0 ? g->fn_isa_acc_nt : g->fn_isa_acc;
    if (!named) g->last_kernel = fn == g->fn_isa_acc ? "fdg_isa_eval_acc" : "fdg_isa_eval_acc_nt";
    HIP_TRY(hipModuleLaunchKernel((hipFunction_t)fn, (unsigned)nwg, 1, 1, 64, 1, 1, 0, a.st, args, nullptr));
    hipLaunchKernelGGL(fdg_reduce_lane_partials, dim3(std::min<uint32_t>(R, 64u)), dim3(256), 0, a.st, part, (uint32_t)nwg, R, a.d_acc);
    HIP_TRY(hipGetLastError());
    return FDG_OK;
  }
  // one batch with sample stride `lss` (fdg_isa_eval[_nt]): full 128-sample tiles through the two-samples-per-lane kernel
  // when there is one and the samples of a leaf are contiguous, the rest through the W = 1 kernel
  int launch_isa(const double *lf, long lss, long lls, double *rt, long rrs, long rrk, long n, long tls = 0, long trs = 0) {
    void *a_wsp = g->d_ws;
    long done = 0;
    if (g->has_w2 && lss == 1 && n >= 128 && !tls && !trs && !g->cfg.no_w2) {
      long n2 = n & ~127l;
      long nwg = std::min<long>(n2 / 128, grid2);
      const double *nowt = nullptr;
      long tls2 = 128 * lss, trs2 = 128 * rrs;
      void *args[] = {(void *)&lf, &lss, &lls, (void *)&rt, &rrs, &rrk, &a_wsp, &n2, &nwg, (void *)&nowt, &tls2, &trs2};
      HIP_TRY(hipModuleLaunchKernel((hipFunction_t)g->fn_isa_w2, (unsigned)nwg, 1, 1, 64, 1, 1, 0, a.st, args, nullptr));
      g->last_kernel = "fdg_isa_eval_w2";
      done = n2;
    }
    if (done < n) {
      const double *lf1 = lf + done * lss;
      double *rt1 = rt + done * rrs;
      long n1 = n - done;
      long nwg = std::min<long>((n1 + 63) / 64, grid);
      const double *nowt = nullptr;
      if (!tls) tls = 64 * lss;
      if (!trs) trs = 64 * rrs;
      void *args[] = {(void *)&lf1, &lss, &lls, (void *)&rt1, &rrs, &rrk, &a_wsp, &n1, &nwg, (void *)&nowt, &tls, &trs};
      void *fn = g->fn_isa_nt && line_aligned(lf1, lss, lls) && line_aligned(rt1, rrs, rrk) && ((tls | trs) & 15) == 0 ? g->fn_isa_nt : g->fn_isa;
      if (!named && done == 0) g->last_kernel = fn == g->fn_isa ? "fdg_isa_eval" : "fdg_isa_eval_nt";
      HIP_TRY(hipModuleLaunchKernel((hipFunction_t)fn, (unsigned)nwg, 1, 1, 64, 1, 1, 0, a.st, args, nullptr));
    }
    return FDG_OK;
  }
  // mode 1 without fused accumulation: the roots went to the column-major scratch; now the weighted sum
  int finish_scratch_acc() {
    double *partial = roots + (size_t)a_rk * R;
    const uint32_t pb = weighted_segments((long)a.B, R);
    const uint8_t *live = nullptr;
    const int rcl = root_live_mask(g, &live);
    if (rcl) return rcl;
    hipLaunchKernelGGL(fdg_weighted_partials, dim3(pb * R), dim3(256), 0, a.st, roots, a_rk, a.d_weight, (long)a.B, R, pb, partial, live);
    hipLaunchKernelGGL(fdg_reduce_partials, dim3(std::min<uint32_t>(R, 64u)), dim3(256), 0, a.st, partial, pb, R, a.d_acc);
    HIP_TRY(hipGetLastError());
    return FDG_OK;
  }

  // Pooled cooperative variant (fdg_isa_eval_pool): one workgroup per CU walks full 64-sample tiles; a leaf's 64 samples must be contiguous (the
  // pool fetch reads 16 bytes per lane) and every leaf within 2^31 bytes of the tile's first -- tile-major batches, or small leaf-major matrices.
  // The last B % 64 samples go through the one-wave kernel.
  bool wants_pool() const { return pool_ok && ((a.mode == 0 && root_stride_ok(a)) || (a.mode == 1 && !fused_acc)); }
  int run_pool() {
    const int mode = a.mode;
    const long n4 = (long)(a.B & ~(int64_t)63), tail = (long)a.B - n4;
    double *rt0 = mode == 0 ? a.d_root : roots;
    const double *lf = a.d_leaf;
    long nwg = std::min<long>(n4 / 64, (long)g->n_cu), lss = a.ss, lls = a.ls, rrs = mode == 0 ? a.rs : a_rs, rrk = mode == 0 ? a.rk : a_rk;
    int rc = ensure_ws(g, std::max(panel_all + (size_t)grid3 * R * 512u + 4096, (size_t)g->pool_panel_wg * (size_t)nwg + 4096));
    if (rc) return rc;
    void *a_wsp = g->d_ws;
    const double *nowt = nullptr;
    long tls = a.lts ? (long)a.lts : 64 * lss, trs = (mode == 0 && a.rts) ? (long)a.rts : 64 * rrs, nn = n4;
    void *args[] = {(void *)&lf, &lss, &lls, (void *)&rt0, &rrs, &rrk, &a_wsp, &nn, &nwg, (void *)&nowt, &tls, &trs};
    HIP_TRY(hipModuleLaunchKernel((hipFunction_t)g->fn_isa_pool, (unsigned)nwg, 1, 1, g->pool_threads, 1, 1, 0, a.st, args, nullptr));
    g->last_kernel = "fdg_isa_eval_pool";
    named = true;
    if (tail) { rc = launch_isa(a.d_leaf + (size_t)(n4 / 64) * (size_t)tls, lss, lls, rt0 + (size_t)(n4 / 64) * (size_t)trs, rrs, rrk, tail, a.lts ? tls : 0, (mode == 0 && a.rts) ? trs : 0); if (rc) return rc; }
    return mode == 1 ? finish_scratch_acc() : FDG_OK;
  }

  // Cooperative variant (fdg_isa_eval_coop): one workgroup of four waves per CU, every workgroup walks tiles of 64 samples; leaf-major input.
  bool wants_coop() const {
    return a.mode == 0 && g->has_coop && g->coop_enabled && g->fn_isa_coop && !(a.ls == 1 && a.ss != 1 && p.L > 1) && !wide_ss && root_stride_ok(a) && !g->cfg.no_coop;
  }
  int run_coop() {
    const double *lf = a.d_leaf; double *rt = a.d_root;
    long nwg = std::min<long>((long)((a.B + 63) / 64), (long)g->n_cu), lss = a.ss, lls = a.ls, rrs = a.rs, rrk = a.rk, n = (long)a.B;
    int rc = ensure_ws(g, std::max(panel_all + (size_t)grid3 * R * 512u + 4096, (size_t)g->coop_panel_wg * (size_t)nwg + 4096));
    if (rc) return rc;
    void *a_wsp = g->d_ws;
    const double *nowt = nullptr;
    long tls = a.lts ? (long)a.lts : 64 * lss, trs = a.rts ? (long)a.rts : 64 * rrs;
    void *args[] = {(void *)&lf, &lss, &lls, (void *)&rt, &rrs, &rrk, &a_wsp, &n, &nwg, (void *)&nowt, &tls, &trs};
    HIP_TRY(hipModuleLaunchKernel((hipFunction_t)g->fn_isa_coop, (unsigned)nwg, 1, 1, g->coop_threads, 1, 1, 0, a.st, args, nullptr));
    g->last_kernel = "fdg_isa_eval_coop";
    return FDG_OK;
  }

  // Linear row-major variant with fused accumulation (fdg_isa_eval_rl_acc): contiguous rows ([B, L], sample stride == L, 16-byte aligned base);
  // full tiles stream the tile's block into an LDS image, the last B % 64 rows go through launch_acc with the caller's strides
  int run_rl_acc() {
    const long n4 = (long)(a.B & ~(int64_t)63), tail = (long)a.B - n4;
    const long grid7 = (long)g->n_cu * waves_per_cu(g->isa7_vgpr, g->isa7_lds_bytes);
    const size_t panel7 = ((size_t)std::max<uint32_t>(g->isa7_mem_slots, 1) * 512u * (size_t)grid7 + 4095) & ~(size_t)4095;
    // (the partial sums of this launch and of the tail's launch_acc live behind the larger of the two panels)
    int rc = ensure_ws(g, std::max(panel_all, panel7) + (size_t)std::max(std::max(grid3, grid5), grid7) * R * 512u + 4096);
    if (rc) return rc;
    void *a_wsp = g->d_ws;
    double *part = (double *)((char *)g->d_ws + std::max(panel_all, panel7));
    const double *lf = a.d_leaf, *wt = a.d_weight;
    long nwg = std::min<long>(n4 / 64, grid7), lss = a.ss, lls = a.ls, zero = 0, nn = n4, tls = 64 * lss;
    void *args[] = {(void *)&lf, (void *)&lss, (void *)&lls, (void *)&part, &zero, &zero, &a_wsp, &nn, &nwg, (void *)&wt, &tls, &zero};
    HIP_TRY(hipModuleLaunchKernel((hipFunction_t)g->fn_isa_rl_acc, (unsigned)nwg, 1, 1, 64, 1, 1, 0, a.st, args, nullptr));
    hipLaunchKernelGGL(fdg_reduce_lane_partials, dim3(std::min<uint32_t>(R, 64u)), dim3(256), 0, a.st, part, (uint32_t)nwg, R, a.d_acc);
    HIP_TRY(hipGetLastError());
    g->last_kernel = "fdg_isa_eval_rl_acc";
    named = true;
    if (tail) { rc = launch_acc(a.d_leaf + (size_t)n4 * (size_t)a.ss, lss, lls, a.d_weight ? a.d_weight + n4 : nullptr, tail); if (rc) return rc; }
    return FDG_OK;
  }
  // ... and the evaluation (fdg_isa_eval_rl)
  int run_rl() {
    const long n4 = (long)(a.B & ~(int64_t)63), tail = (long)a.B - n4;
    const long grid6 = (long)g->n_cu * waves_per_cu(g->isa6_vgpr, g->isa6_lds_bytes);
    int rc = ensure_ws(g, std::max(panel_all + (size_t)std::max(grid3, grid5) * R * 512u + 4096, (size_t)std::max<uint32_t>(g->isa6_mem_slots, 1) * 512u * (size_t)grid6 + 4096));
    if (rc) return rc;
    void *a_wsp = g->d_ws;
    const double *lf = a.d_leaf; double *rt = a.d_root;
    long nwg = std::min<long>(n4 / 64, grid6), lss = a.ss, lls = a.ls, rrs = a.rs, rrk = a.rk, nn = n4, tls = 64 * lss, trs = 64 * rrs;
    const double *nowt = nullptr;
    void *args[] = {(void *)&lf, (void *)&lss, (void *)&lls, (void *)&rt, &rrs, &rrk, &a_wsp, &nn, &nwg, (void *)&nowt, &tls, &trs};
    HIP_TRY(hipModuleLaunchKernel((hipFunction_t)g->fn_isa_rl, (unsigned)nwg, 1, 1, 64, 1, 1, 0, a.st, args, nullptr));
    g->last_kernel = "fdg_isa_eval_rl";
    named = true;
    if (tail) { rc = launch_isa(a.d_leaf + (size_t)n4 * (size_t)a.ss, lss, lls, a.d_root + (size_t)n4 * (size_t)a.rs, rrs, rrk, tail); if (rc) return rc; }
    return FDG_OK;
  }

  // Row-major leaves ([B, L], leaf stride 1, any row pitch): full 64-row tiles go through the variant that stages chunks of rows in LDS
  // itself (fdg_isa_eval_rm / _rm_acc) -- the matrix is read once, in place -- for evaluation and for fused accumulation alike; the last
  // B % 64 rows go through the plain kernel with the caller's strides (its lanes gather their own rows: fine for under a tile).
  bool wants_rm() const {
    const bool rm_shape = a.ls == 1 && a.ss >= (int64_t)p.L && !wide_ss && p.L >= 16 && a.B >= 64 && !tiled && !g->cfg.no_rm;
    return rm_shape && ((a.mode == 0 && g->has_rm && g->fn_isa_rm && root_stride_ok(a)) || (fused_acc && g->has_rm_acc && g->fn_isa_rm_acc));
  }
  int run_rm() {
    const long n4 = (long)(a.B & ~(int64_t)63), lss = a.ss, lls = a.ls, tail = (long)a.B - n4;
    void *a_wsp = g->d_ws;
    const double *lf = a.d_leaf;
    int rc;
    if (a.mode == 0) {
      double *rt = a.d_root;
      long nwg = std::min<long>(n4 / 64, grid4), rrs = a.rs, rrk = a.rk, nn = n4, tls = 64 * lss, trs = 64 * rrs;
      const double *nowt = nullptr;
      void *args[] = {(void *)&lf, (void *)&lss, (void *)&lls, (void *)&rt, &rrs, &rrk, &a_wsp, &nn, &nwg, (void *)&nowt, &tls, &trs};
      HIP_TRY(hipModuleLaunchKernel((hipFunction_t)g->fn_isa_rm, (unsigned)nwg, 1, 1, 64, 1, 1, 0, a.st, args, nullptr));
      g->last_kernel = "fdg_isa_eval_rm";
      named = true;                    // (the last B % 64 rows below do not rename the call)
      if (tail) { rc = launch_isa(a.d_leaf + (size_t)n4 * (size_t)a.ss, lss, lls, a.d_root + (size_t)n4 * (size_t)a.rs, rrs, rrk, tail); if (rc) return rc; }
    } else {
      double *part = (double *)((char *)g->d_ws + panel_all);
      const double *wt = a.d_weight;
      long nwg = std::min<long>(n4 / 64, grid5), zero = 0, nn = n4, tls = 64 * lss;
      void *args[] = {(void *)&lf, (void *)&lss, (void *)&lls, (void *)&part, &zero, &zero, &a_wsp, &nn, &nwg, (void *)&wt, &tls, &zero};
      HIP_TRY(hipModuleLaunchKernel((hipFunction_t)g->fn_isa_rm_acc, (unsigned)nwg, 1, 1, 64, 1, 1, 0, a.st, args, nullptr));
      hipLaunchKernelGGL(fdg_reduce_lane_partials, dim3(std::min<uint32_t>(R, 64u)), dim3(256), 0, a.st, part, (uint32_t)nwg, R, a.d_acc);
      HIP_TRY(hipGetLastError());
      g->last_kernel = "fdg_isa_eval_rm_acc";
      named = true;
      if (tail) { rc = launch_acc(a.d_leaf + (size_t)n4 * (size_t)a.ss, lss, lls, a.d_weight ? a.d_weight + n4 : nullptr, tail); if (rc) return rc; }
    }
    return FDG_OK;
  }

  // Sample-major input (compile_Python's [B, L]) of a program without a row-major variant, or a sample stride too wide for the kernel's 32-bit
  // lane offset: the ISA kernel wants a wave's 64 samples of one leaf contiguous, so chunks of the batch are transposed to leaf-major first
  // (2 extra HBM passes over the leaves).  Chunks are double-buffered: the transposition of chunk c+1 (HBM-bound) runs on an internal
  // stream while the evaluator works on chunk c (fp64-bound for all but tiny graphs).
  bool wants_transposition() const { return !tiled && ((a.ls == 1 && a.ss != 1 && p.L > 1) || (wide_ss && p.L > 0)); }
  int run_transposed() {
    const int64_t B = a.B, ss = a.ss, ls = a.ls;
    const hipStream_t st = a.st;
    long Bc = std::max<long>((long)g->n_cu * 8 * 64 * 2, (long)(g->cfg.sm_chunk_bytes / (8ull * p.L)));
    Bc = std::min<long>((Bc + 63) & ~63l, (B + 63) & ~63l);
    const size_t one = (size_t)Bc * p.L * sizeof(double);
    const size_t need3 = 2 * one;
    if (g->ws3_bytes < need3) {
      if (g->d_ws3) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(g->d_ws3)); g->d_ws3 = nullptr; g->ws3_bytes = 0; }
      if (hipMalloc(&g->d_ws3, need3) != hipSuccess) { set_error("hipMalloc(transposed leaves) failed"); return FDG_E_NOMEM; }
      g->ws3_bytes = need3;
    }
    if (!g->s2) {
      hipStream_t s; HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); g->s2 = s;
      for (int i = 0; i < 2; ++i) {
        hipEvent_t e; HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); g->ev_t[i] = e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); g->ev_k[i] = e;
      }
      hipEvent_t e; HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); g->ev_in = e;
    }
    hipStream_t s2 = (hipStream_t)g->s2;
    HIP_TRY(hipEventRecord((hipEvent_t)g->ev_in, st));              // the caller's leaves are ready on st
    HIP_TRY(hipStreamWaitEvent(s2, (hipEvent_t)g->ev_in, 0));
    auto transpose = [&](long c0, int buf) {
      const long n = std::min<long>(Bc, B - c0);
      if (ls == 1 && (ss & 1) == 0 && ((uintptr_t)a.d_leaf & 15) == 0 && p.L >= 32 && !g->cfg.transpose_narrow) {
        const long ntile = ((n + 63) / 64) * ((p.L + 63) / 64);
        hipLaunchKernelGGL(fdg_transpose_rows_wide, dim3((unsigned)std::min<long>(ntile, (long)g->n_cu * 8)), dim3(256), 0, s2,
                           a.d_leaf + c0 * ss, (long)ss, (double *)((char *)g->d_ws3 + (size_t)buf * one), Bc, n, p.L);
      } else {
        const long ntile = ((n + 63) / 64) * ((p.L + 31) / 32);
        hipLaunchKernelGGL(fdg_transpose_to_leaf_major, dim3((unsigned)std::min<long>(ntile, (long)g->n_cu * 16)), dim3(256), 0, s2,
                           a.d_leaf + c0 * ss, (long)ss, (long)ls, (double *)((char *)g->d_ws3 + (size_t)buf * one), Bc, n, p.L);
      }
      return hipEventRecord((hipEvent_t)g->ev_t[buf], s2);
    };
    HIP_TRY(transpose(0, 0));
    int c = 0, rc;
    for (long c0 = 0; c0 < B; c0 += Bc, ++c) {
      const int buf = c & 1;
      const long n = std::min<long>(Bc, B - c0);
      if (c0 + Bc < B) {
        // buffer buf^1 was last read by the evaluator of chunk c-1
        if (c >= 1) HIP_TRY(hipStreamWaitEvent(s2, (hipEvent_t)g->ev_k[buf ^ 1], 0));
        HIP_TRY(transpose(c0 + Bc, buf ^ 1));
      }
      HIP_TRY(hipStreamWaitEvent(st, (hipEvent_t)g->ev_t[buf], 0));
      const double *c_leaf = (const double *)((char *)g->d_ws3 + (size_t)buf * one);
      rc = fused_acc ? launch_acc(c_leaf, 1, Bc, a.d_weight ? a.d_weight + c0 : nullptr, n)
                     : launch_isa(c_leaf, 1, Bc, roots + c0 * a_rs, a_rs, a_rk, n);
      if (rc) return rc;
      HIP_TRY(hipEventRecord((hipEvent_t)g->ev_k[buf], st));
    }
    return FDG_OK;
  }

  // the caller's batch as it is: the plain / streaming kernels (a strided matrix or a tile-major batch), fused accumulation or roots
  int run_direct() {
    // (accumulation through the root scratch writes it column-major: an ordinary strided array, whatever the leaves are)
    return fused_acc ? launch_acc(a.d_leaf, (long)a.ss, (long)a.ls, a.d_weight, (long)a.B, (long)a.lts)
                     : launch_isa(a.d_leaf, (long)a.ss, (long)a.ls, roots, a_rs, a_rk, (long)a.B, (long)a.lts, a.mode == 0 ? (long)a.rts : 0);
  }
};

int run_isa(fdg_graph *g, const RunArgs &a, bool rl_shape) {
  int rc = ensure_module(g);
  if (rc) return rc;
  IsaRun r(g, a);
  rc = r.plan();
  if (rc) return rc;
  if (r.wants_pool()) return r.run_pool();
  if (r.wants_coop()) return r.run_coop();
  if (rl_shape && a.mode == 1 && g->fn_isa_rl_acc && !r.tiled) return r.run_rl_acc();
  if (rl_shape && a.mode == 0 && g->fn_isa_rl && !r.tiled) return r.run_rl();
  if (r.wants_rm()) return r.run_rm();
  if (a.mode == 0 && !root_stride_ok(a)) { set_error("root sample stride negative or of 2^23 elements or more is not supported by the ISA kernel"); return FDG_E_UNSUPPORTED; }
  rc = r.wants_transposition() ? r.run_transposed() : r.run_direct();
  if (rc) return rc;
  if (a.mode == 1 && !r.fused_acc) return r.finish_scratch_acc();
  return FDG_OK;
}
}  // namespace

int fdg_run_locked(fdg_graph *g, int mode, const double *d_leaf, int64_t ss, int64_t ls, double *d_root,
                   int64_t rs, int64_t rk, const double *d_weight, double *d_acc, int64_t B, hipStream_t st,
                   int64_t lts, int64_t rts) {
  int rc = ensure_device(g);
  if (rc) return rc;
  const Lowered &p = g->prog;
  const uint32_t R = p.R;
  const RunArgs a{mode, d_leaf, ss, ls, d_root, rs, rk, d_weight, d_acc, B, st, lts, rts};
  const bool isa = !g->code_object.empty() && g->isa;
  // Tile-major batches (fdg_eval_device_tiled): tile t holds samples 64 t .. 64 t + 63 at base + t * tile stride.  Only the
  // kernels of the optimizing back end take a tile stride; a plain strided matrix is the case tile stride = 64 * sample stride.
  const bool tiled = lts != 0 || rts != 0;
  if (tiled && !isa) { set_error("tile-major batches need a handle specialised with FDG_SPEC_ISA"); return FDG_E_UNSUPPORTED; }
  if (tiled && (lts < 0 || rts < 0 || ss < 0 || ss >= (1ll << 23) || (mode == 0 && !root_stride_ok(a)))) {
    set_error("tile-major batch: negative strides or sample strides of 2^23 elements or more are not supported"); return FDG_E_UNSUPPORTED;
  }
  // contiguous rows ([B, L] with sample stride L): the linear row-major variant
  const bool rl_rows = g->has_rl && ls == 1 && ss == (int64_t)p.L && p.L >= 2 && B >= 64 && ((uintptr_t)d_leaf & 15) == 0 && !g->cfg.no_rl;
  const bool rl_shape = rl_rows && ((mode == 0 && root_stride_ok(a)) || (mode == 1 && g->has_rl_acc && g->has_acc && !g->cfg.no_fused_acc));
  if (mode == 0 && isa && g->cfg.root_scratch_min && R >= g->cfg.root_scratch_min && rk == 1 && rs >= (int64_t)R && (rts == 0 || rts == 64 * rs) && B >= 256 &&
      !(ls == 1 && ss != 1 && (g->alt_code.size() || g->has_rm || rl_rows)))     // (the row-major variants write a tile's rows together: left alone)
    return run_through_root_scratch(g, a);
  if (isa && !g->alt_code.empty() && ls == 1 && ss != 1 && p.L > 1 && !tiled && !rl_shape) return run_companion(g, a);
  if (isa) return run_isa(g, a, rl_shape);
  if (!g->code_object.empty()) {      // the compiler-scheduled per-graph kernels (HIP source through hiprtc)
    rc = ensure_module(g);
    if (rc) return rc;
    g->last_kernel = ls == 1 ? "fdg_spec_sm" : "fdg_spec_gen";
    return launch_hip_source(g, (hipFunction_t)((ls == 1) ? g->fn_eval_sm : g->fn_eval_gen), mode, d_leaf, ss, ls, d_root, rs, rk, d_weight, d_acc, B, st);
  }
  return run_interpreter(g, a);
}

// ---------------------------------------------------------------------------
// JIT
// ---------------------------------------------------------------------------
uint64_t fnv1a(const std::string &s, uint64_t h) {
  for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
  return h;
}

// Cache artefacts are code that will run on the GPU and parameters that size register files: only files owned by the
// calling user or by root and not writable by everybody are read back.  (The group bit is not tested: the assembler and
// the linker create their outputs under the caller's umask, 0664/0775 under umask 002, and the directory they sit in is
// vetted by fdg_cache_dir.  FDG_CACHE_TRUST=1 lifts the ownership test.)
static bool cache_trust() { return fdg::knob("FDG_CACHE_TRUST") != nullptr; }
// `deny`: permission bits that disqualify the file -- 002 inside the caller's own vetted cache directory (the assembler and the linker
// create their outputs under the caller's umask: 0664 under umask 002), 022 for files found through $FDG_CACHE_RO_DIR (nothing legitimate is
// written there by this process, so a group-writable artefact is not taken).
static bool read_file_vetted(const std::string &path, std::vector<char> &out, mode_t deny) {
  const int fd = ::open(path.c_str(), O_RDONLY | O_CLOEXEC);
  if (fd < 0) return false;
  struct stat sb;
  if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || (!cache_trust() && ((sb.st_uid != geteuid() && sb.st_uid != 0) || (sb.st_mode & deny)))) { ::close(fd); return false; }
  out.resize((size_t)sb.st_size);
  size_t got = 0;
  while (got < out.size()) {
    const ssize_t r = ::read(fd, out.data() + got, out.size() - got);
    if (r <= 0) break;
    got += (size_t)r;
  }
  ::close(fd);
  out.resize(got);
  return !out.empty();
}
bool read_file(const std::string &path, std::vector<char> &out) { return read_file_vetted(path, out, 002); }

// Lookup of a cached artefact: the (writable, vetted) cache directory first, then the read-only directories named by
// $FDG_CACHE_RO_DIR (colon-separated; e.g. the kernel_cache a package ships, which may belong to root or sit in a
// read-only checkout).  A read-only directory and the file in it must belong to the caller or to root and must be writable by
// their owner only (no group or world write bit); nothing is ever written there.
bool read_cached(const std::string &dir, const std::string &fname, std::vector<char> &out) {
  if (!dir.empty() && read_file(dir + "/" + fname, out)) return true;
  const char *ro = fdg::knob("FDG_CACHE_RO_DIR");
  if (!ro || !*ro) return false;
  std::string all = ro;
  size_t pos = 0;
  while (pos <= all.size()) {
    const size_t e = all.find(':', pos);
    const std::string d = all.substr(pos, e == std::string::npos ? std::string::npos : e - pos);
    pos = e == std::string::npos ? all.size() + 1 : e + 1;
    if (d.empty() || d == dir) continue;
    struct stat sb;
    if (stat(d.c_str(), &sb) != 0 || !S_ISDIR(sb.st_mode)) continue;
    if (!cache_trust() && ((sb.st_uid != geteuid() && sb.st_uid != 0) || (sb.st_mode & 022))) continue;
    if (read_file_vetted(d + "/" + fname, out, 022)) return true;
  }
  return false;
}

// suffix of intermediate files: unique per process AND per call (two threads of one process may specialise two handles of
// the same graph at the same time)
static std::string tmp_suffix() {
  static std::atomic<unsigned long> counter{0};
  return ".tmp." + std::to_string((long)getpid()) + "." + std::to_string(counter.fetch_add(1));
}

// written under a unique name, then renamed: readers (other ranks JIT-ing the same graph) never see half a file
bool write_file(const std::string &path, const char *data, size_t n) {
  const std::string tmp = path + tmp_suffix();
  const int fd = ::open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
  if (fd < 0) return false;
  size_t put = 0;
  while (put < n) {
    const ssize_t r = ::write(fd, data + put, n - put);
    if (r <= 0) { ::close(fd); std::remove(tmp.c_str()); return false; }
    put += (size_t)r;
  }
  ::close(fd);
  if (std::rename(tmp.c_str(), path.c_str()) != 0) { std::remove(tmp.c_str()); return false; }
  return true;
}

// The directory JIT-ed code objects are cached in: the argument, else $FDG_CACHE_DIR, else a per-user directory
// ($XDG_CACHE_HOME/fdg, $HOME/.cache/fdg, /tmp/fdg-cache-<uid>), created 0700.  It must belong to the calling user and
// must not be writable by group or others -- another local user could otherwise plant a code object under a predictable
// name.  Paths with a quote or a newline are refused outright.
int fdg_cache_dir(const char *arg, std::string &dir) {
  if (arg && *arg) dir = arg;
  else if (const char *e = fdg::knob("FDG_CACHE_DIR")) dir = e;
  else if (const char *x = std::getenv("XDG_CACHE_HOME")) { dir = std::string(x); mkdir(dir.c_str(), 0700); dir += "/fdg"; }
  else if (const char *h = std::getenv("HOME")) { dir = std::string(h) + "/.cache"; mkdir(dir.c_str(), 0700); dir += "/fdg"; }
  else dir = "/tmp/fdg-cache-" + std::to_string((long)geteuid());
  if (dir.find_first_of("'\"\n") != std::string::npos) { set_error("cache directory path contains a quote or a newline: " + dir); return FDG_E_INVALID; }
  mkdir(dir.c_str(), 0700);
  struct stat sb;
  const bool exists = stat(dir.c_str(), &sb) == 0 && S_ISDIR(sb.st_mode);
  const bool usable = exists && (cache_trust() || (sb.st_uid == geteuid() && !(sb.st_mode & 022)));
  if (usable) return FDG_OK;
  if (!(arg && *arg)) {
    // the default location cannot be used (foreign owner, group-writable home, read-only file system): compile into a
    // private directory of this process instead of failing -- nothing is cached across runs then
    static std::mutex mu;
    static std::string priv;
    std::lock_guard<std::mutex> lk(mu);
    if (priv.empty()) {
      char tmpl[] = "/tmp/fdg-XXXXXX";
      if (mkdtemp(tmpl)) priv = tmpl;
    }
    if (!priv.empty()) { dir = priv; return FDG_OK; }
  }
  if (!exists) { set_error("cannot create the cache directory " + dir); return FDG_E_JIT; }
  set_error("cache directory " + dir + " must be owned by the calling user and not be writable by group or others (FDG_CACHE_TRUST=1 overrides)");
  return FDG_E_JIT;
}

// runs argv (no shell), stdout + stderr appended to `log_path`; returns the exit status, -1 when it could not be run
static int run_cmd(const std::vector<std::string> &argv, const std::string &log_path) {
  std::vector<char *> av;
  for (const std::string &a : argv) av.push_back(const_cast<char *>(a.c_str()));
  av.push_back(nullptr);
  const pid_t pid = fork();
  if (pid < 0) return -1;
  if (pid == 0) {
    const int fd = ::open(log_path.c_str(), O_WRONLY | O_CREAT | O_APPEND, 0600);
    if (fd >= 0) { dup2(fd, 1); dup2(fd, 2); ::close(fd); }
    execvp(av[0], av.data());      // PATH search: FDG_HIPCC=hipcc or a relative FDG_LLVM_BIN work
    _exit(127);
  }
  int status = 0;
  while (waitpid(pid, &status, 0) < 0) if (errno != EINTR) return -1;
  return WIFEXITED(status) ? WEXITSTATUS(status) : -1;
}
static std::string slurp_and_remove(const std::string &path) {
  std::string log;
  { std::ifstream f(path, std::ios::binary); if (f) log.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>()); }
  std::remove(path.c_str());
  return log;
}

int compile_hiprtc(const std::string &src, bool fast, std::vector<char> &co, std::string &log) {
  hiprtcProgram prog;
  if (hiprtcCreateProgram(&prog, src.c_str(), "fdg_spec.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
    log = "hiprtcCreateProgram failed"; return -1;
  }
  const char *opts[] = {"--offload-arch=gfx950", "-O3", fast ? "-ffp-contract=fast" : "-ffp-contract=off"};
  hiprtcResult r = hiprtcCompileProgram(prog, 3, opts);
  size_t ls = 0;
  hiprtcGetProgramLogSize(prog, &ls);
  if (ls > 1) { log.resize(ls); hiprtcGetProgramLog(prog, &log[0]); }
  if (r != HIPRTC_SUCCESS) { hiprtcDestroyProgram(&prog); if (log.empty()) log = hiprtcGetErrorString(r); return -1; }
  size_t cs = 0;
  hiprtcGetCodeSize(prog, &cs);
  co.resize(cs);
  hiprtcGetCode(prog, co.data());
  hiprtcDestroyProgram(&prog);
  return 0;
}

int compile_hipcc(const std::string &src_path, const std::string &out_path, bool fast, std::string &log) {
  const char *hipcc = fdg::knob("FDG_HIPCC");
  const std::string tmp = out_path + tmp_suffix();
  const int rc = run_cmd({hipcc ? hipcc : "/opt/rocm/bin/hipcc", "--genco", "--offload-arch=gfx950", "-O3",
                          fast ? "-ffp-contract=fast" : "-ffp-contract=off", "-o", tmp, src_path}, tmp + ".log");
  log = slurp_and_remove(tmp + ".log");
  if (rc == 0) ::chmod(tmp.c_str(), 0644);      // whatever the umask: the artefact must pass read_file's vetting
  if (rc != 0 || std::rename(tmp.c_str(), out_path.c_str()) != 0) { std::remove(tmp.c_str()); return -1; }
  return 0;
}

// LDS budget of the interpreter: enough slots for the whole live set when it
// is small (8 blocks/CU), otherwise 40 slots = 80 KiB per block (2 blocks/CU)
static uint32_t interp_lds_budget() {
  uint32_t budget = 40;
  const char *env = fdg::knob("FDG_LDS_SLOTS");
  if (env) budget = (uint32_t)std::max(1, std::atoi(env));
  return std::min(budget, 79u);
}

extern "C" {

const char *fdg_last_error(void) { return fdg::last_error_cstr(); }
int fdg_version(void) { return FDG_VERSION; }
double fdg_powi(double x, int32_t n) { return fdg_powi_impl(x, n); }
void fdg_free(void *p) { std::free(p); }

int fdg_graph_create(const fdg_graph_desc *d, fdg_graph **out) {
  if (!out) { set_error("null out pointer"); return FDG_E_INVALID; }
  *out = nullptr;
  std::string err;
  int rc = validate_desc(d, err);
  if (rc) { set_error(err); return rc; }
  fdg_graph *g = new (std::nothrow) fdg_graph();
  if (!g) { set_error("out of memory"); return FDG_E_NOMEM; }
  Lowered &p = g->prog;
  p.L = d->n_leaf; p.N = d->n_node; p.R = d->n_root; p.E = d->n_edge;
  p.op.assign(d->op, d->op + p.N);
  p.power.assign(d->power, d->power + p.N);
  if (p.N) p.off.assign(d->child_off, d->child_off + p.N + 1); else p.off.assign(1, 0);
  p.idx.assign(d->child_idx, d->child_idx + p.E);
  p.fac.assign(d->child_fac, d->child_fac + p.E);
  p.root_slot.assign(d->root_slot, d->root_slot + p.R);
  analyse(p);
  build_interpreter_program(p, interp_lds_budget());
  g->knobs = fdg::env_snapshot();
  parse_launch_cfg(g);
  *out = g;
  return FDG_OK;
}

// A handle's options (include/fdg.h).  value == NULL removes the option.
// the handle's options as they are now, for entry points that take the handle as const and therefore do not hold its mutex while they work
static fdg::KnobMap knobs_snapshot(const fdg_graph *g) {
  if (!g) return fdg::KnobMap();
  std::lock_guard<std::mutex> lk(const_cast<fdg_graph *>(g)->mu);
  return g->knobs;
}

int fdg_graph_set_option(fdg_graph *g, const char *name, const char *value) {
  if (!g || !name || !*name) { set_error("null handle or option name"); return FDG_E_INVALID; }
  if (std::strncmp(name, "FDG_", 4) != 0) { set_error("option names start with FDG_"); return FDG_E_INVALID; }
  std::lock_guard<std::mutex> lk(g->mu);
  fdg::KnobScope knob_scope(&g->knobs);
  if (value) g->knobs[name] = value; else g->knobs.erase(name);
  parse_launch_cfg(g);
  if (g->cx_twin) {       // (the twin's own launches read its options under ITS mutex)
    std::lock_guard<std::mutex> lk2(g->cx_twin->mu);
    if (value) g->cx_twin->knobs[name] = value; else g->cx_twin->knobs.erase(name);
    parse_launch_cfg(g->cx_twin);
  }
  return FDG_OK;
}
// Process defaults: what handles created from now on start with, and what the entry points without a handle see (fdg_leaf_eval_device).
int fdg_set_default_option(const char *name, const char *value) {
  if (!name || std::strncmp(name, "FDG_", 4) != 0) { set_error("option names start with FDG_"); return FDG_E_INVALID; }
  fdg::set_default_knob(name, value);
  return FDG_OK;
}
const char *fdg_get_default_option(const char *name) { return name ? fdg::knob(name) : nullptr; }   // (the calling thread is inside no handle's entry point: the defaults)
const char *fdg_graph_get_option(const fdg_graph *g, const char *name) {
  if (!g || !name) return nullptr;
  // (copied out under the mutex into one of four slots of the calling thread: the pointer stays valid across a concurrent fdg_graph_set_option)
  thread_local std::string hold[4];
  thread_local unsigned slot = 0;
  std::lock_guard<std::mutex> lk(const_cast<fdg_graph *>(g)->mu);
  auto it = g->knobs.find(name);
  if (it == g->knobs.end()) return nullptr;
  std::string &h = hold[slot++ & 3u];
  h = it->second;
  return h.c_str();
}

// Which of the reference's two evaluators the handle reproduces bit for bit (include/fdg.h).  Before any specialisation.
int fdg_graph_set_association(fdg_graph *g, int assoc) {
  if (!g) { set_error("null handle"); return FDG_E_INVALID; }
  if (assoc != FDG_ASSOC_STATIC && assoc != FDG_ASSOC_INTERP) { set_error("unknown association"); return FDG_E_INVALID; }
  std::lock_guard<std::mutex> lk(g->mu);
  fdg::KnobScope knob_scope(&g->knobs);
  const bool want = assoc == FDG_ASSOC_INTERP;
  if (g->prog.assoc_interp == want) return FDG_OK;
  if (!g->code_object.empty() || !g->alt_code.empty() || !g->fused_code.empty() || !g->mc_code.empty() || g->cx_twin || g->mc_route ||
      !g->typed_code[1].empty() || !g->typed_code[2].empty() || !g->typed_code[3].empty()) {
    set_error("fdg_graph_set_association must be called before the handle is specialised"); return FDG_E_INVALID;
  }
  g->prog.assoc_interp = want;
  build_interpreter_program(g->prog, interp_lds_budget());
  if (g->d_code) { hipDeviceSynchronize(); hipFree(g->d_code); g->d_code = nullptr; }      // (the stream is uploaded again on the next use)
  return FDG_OK;
}

int fdg_graph_release_device(fdg_graph *g) {
  if (!g) return FDG_OK;
  std::lock_guard<std::mutex> lk(g->mu);
  fdg::KnobScope knob_scope(&g->knobs);
  if (g->d_code) { hipFree(g->d_code); g->d_code = nullptr; }
  if (g->d_root_live) { hipFree(g->d_root_live); g->d_root_live = nullptr; }
  {
    fdg_ws_set cur;
    park_current_ws(g, cur);
    free_ws_set(cur);
    load_ws(g, fdg_ws_set());
    for (fdg_ws_set &w : g->ws_pool) free_ws_set(w);
    g->ws_pool.clear();
    g->ws_bound = true;
    g->ws_key = nullptr;
  }
  if (g->module) { hipModuleUnload((hipModule_t)g->module); g->module = nullptr; g->fn_eval_sm = g->fn_eval_gen = nullptr; g->fn_isa = nullptr; g->fn_isa_nt = g->fn_isa_acc_nt = nullptr; }
  if (g->fused_module) { hipModuleUnload((hipModule_t)g->fused_module); g->fused_module = nullptr; g->fn_fused = nullptr; }
  if (g->alt_module) { hipModuleUnload((hipModule_t)g->alt_module); g->alt_module = nullptr; g->fn_alt_sm = g->fn_alt_gen = nullptr; }
  if (g->mc_module) { hipModuleUnload((hipModule_t)g->mc_module); g->mc_module = nullptr; g->fn_mc = g->fn_mc_acc = nullptr; }
  if (g->cx_twin) fdg_graph_release_device(g->cx_twin);
  for (int t = 0; t < 4; ++t) if (g->typed_module[t]) { hipModuleUnload((hipModule_t)g->typed_module[t]); g->typed_module[t] = nullptr; g->fn_typed[t] = nullptr; }
  return FDG_OK;
}

int fdg_graph_destroy(fdg_graph *g) {
  if (!g) return FDG_OK;
  if (g->cx_twin) { fdg_graph_destroy(g->cx_twin); g->cx_twin = nullptr; }
  fdg_graph_release_device(g);
  delete g;
  return FDG_OK;
}

int fdg_graph_query(const fdg_graph *g, fdg_graph_info *o) {
  if (!g || !o) { set_error("null argument"); return FDG_E_INVALID; }
  const Lowered &p = g->prog;
  std::memset(o, 0, sizeof *o);
  o->n_leaf = p.L; o->n_node = p.N; o->n_root = p.R; o->n_edge = p.E;
  o->n_live_node = (uint32_t)p.order.size(); o->n_live_leaf = p.n_live_leaf;
  o->flops_alg = p.flops_alg; o->bytes_alg = 8ull * ((uint64_t)p.L + p.R);
  o->max_live = p.max_live; o->n_slot_lds = p.lds_slots; o->n_slot_mem = p.mem_slots; o->n_ops = p.n_ops;
  o->specialized = g->code_object.empty() ? 0 : 1;
  o->spec_vgpr = g->spec_vgpr; o->spec_lds_bytes = g->spec_lds; o->spec_scratch_bytes = g->spec_scratch;
  return FDG_OK;
}

int fdg_graph_kernel_info(fdg_graph *g, fdg_kernel_info *o) {
  if (!g || !o) { set_error("null argument"); return FDG_E_INVALID; }
  std::lock_guard<std::mutex> lk(g->mu);
  fdg::KnobScope knob_scope(&g->knobs);
  std::memset(o, 0, sizeof *o);
  std::snprintf(o->last_kernel, sizeof o->last_kernel, "%s", g->last_kernel ? g->last_kernel : "");
  if (!(g->isa && !g->code_object.empty())) return FDG_OK;
  auto wpc = [](uint32_t vgpr, uint32_t lds_bytes) {      // the rule of fdg_run_locked (without its environment override)
    const uint32_t valloc = std::max<uint32_t>(8, (vgpr + 7) & ~7u);
    uint32_t per_cu = std::min<uint32_t>(8, 512 / valloc) * 4;
    if (lds_bytes) per_cu = std::min<uint32_t>(per_cu, (160u * 1024u) / lds_bytes);
    return std::max<uint32_t>(1, std::min<uint32_t>(per_cu, 32));
  };
  for (int i = 0; i < 3; ++i) { o->n_valu[i] = g->st_valu[i]; o->n_ld_leaf[i] = g->st_ld_leaf[i]; o->n_panel[i] = g->st_panel[i]; o->n_lds[i] = g->st_lds[i]; }
  o->waves_per_cu[0] = wpc(g->isa_vgpr, g->isa_lds_bytes);
  if (g->has_acc) o->waves_per_cu[1] = wpc(g->isa3_vgpr, g->isa3_lds_bytes);
  if (g->has_rm) o->waves_per_cu[2] = wpc(g->isa4_vgpr, g->isa4_lds_bytes);
  o->has_acc = g->has_acc; o->has_rm = g->has_rm; o->has_coop = g->has_coop && g->coop_enabled; o->rm_bufs = g->rm_bufs;
  o->has_pool = g->has_pool; o->pool_fetch = g->pool_fetch; o->pool_valu = g->pool_valu;
  o->has_rl = g->has_rl; o->rl_reserved = 0; o->rl_valu = g->rl_valu;
  return FDG_OK;
}

int fdg_graph_emit_source(const fdg_graph *g, unsigned flags, char **source) {
  const fdg::KnobMap knobs_now = knobs_snapshot(g);      // (a copy taken under the handle's mutex: fdg_graph_set_option may run on another thread)
  fdg::KnobScope knob_scope(g ? &knobs_now : nullptr);
  if (!g || !source) { set_error("null argument"); return FDG_E_INVALID; }
  std::string s = emit_hip_source(g->prog, flags);
  char *m = (char *)std::malloc(s.size() + 1);
  if (!m) { set_error("out of memory"); return FDG_E_NOMEM; }
  std::memcpy(m, s.c_str(), s.size() + 1);
  *source = m;
  return FDG_OK;
}

static void apply_land_env(fdg::OptParams &q);
static fdg::OptParams to_params(const fdg_opt_params *q) {
  fdg::OptParams prm;
  if (q) {
    if (q->n_reg) prm.n_reg = std::min<uint32_t>(q->n_reg, 125);
    if (q->n_lds) prm.n_lds = std::min<uint32_t>(q->n_lds, 127);
    if (q->lookahead_lds) prm.lookahead_lds = q->lookahead_lds;
    if (q->lookahead_mem) prm.lookahead_mem = q->lookahead_mem;
    if (q->lookahead_leaf) prm.lookahead_leaf = q->lookahead_leaf;
    if (q->n_acc) prm.n_acc = std::min<uint32_t>(q->n_acc, 124);
    if (q->vn_window) prm.vn_window = q->vn_window;
    prm.fma = q->fma != 0;
    prm.remat_window = q->remat_window;
    if (q->remat_cost) prm.remat_cost = std::min<uint32_t>(q->remat_cost, 64);
  }
  return prm;
}

int fdg_graph_set_schedule_groups(fdg_graph *g, const uint32_t *group, uint32_t n_node) {
  if (!g) { set_error("null handle"); return FDG_E_INVALID; }
  std::lock_guard<std::mutex> lk(g->mu);
  fdg::KnobScope knob_scope(&g->knobs);
  if (!group) { g->prog.sched_group.clear(); return FDG_OK; }
  if (n_node != g->prog.N) { set_error("schedule groups: length differs from n_node"); return FDG_E_INVALID; }
  g->prog.sched_group.assign(group, group + n_node);
  return FDG_OK;
}

int fdg_graph_set_opt_params(fdg_graph *g, const fdg_opt_params *prm) {
  if (!g || !prm) { set_error("null argument"); return FDG_E_INVALID; }
  std::lock_guard<std::mutex> lk(g->mu);
  fdg::KnobScope knob_scope(&g->knobs);
  g->opt = *prm;
  g->has_opt = true;
  return FDG_OK;
}

static fdg_opt_params get_opt_params(const fdg_graph *g) { return g->opt; }

int fdg_graph_opt_program(const fdg_graph *g, const fdg_opt_params *q, fdg_mop **ops, uint64_t *n_ops,
                          uint32_t *n_reg_used, uint32_t *n_lds_used, uint32_t *n_mem_used, uint32_t *n_acc_used) {
  const fdg::KnobMap knobs_now = knobs_snapshot(g);      // (a copy taken under the handle's mutex: fdg_graph_set_option may run on another thread)
  fdg::KnobScope knob_scope(g ? &knobs_now : nullptr);
  if (!g || !ops || !n_ops) { set_error("null argument"); return FDG_E_INVALID; }
  fdg::OptProgram prog;
  fdg::OptParams prm = to_params(q);
  apply_land_env(prm);                         // (the experiment switch FDG_LAND reaches the exported program too: replayed in tests)
  fdg::build_opt_program(g->prog, prm, prog);
  if (!prog.supported) { set_error("optimizing back end does not cover this graph: " + prog.why); return FDG_E_UNSUPPORTED; }
  fdg_mop *m = (fdg_mop *)std::malloc(std::max<size_t>(1, prog.ops.size()) * sizeof(fdg_mop));
  if (!m) { set_error("out of memory"); return FDG_E_NOMEM; }
  for (size_t i = 0; i < prog.ops.size(); ++i) {
    const fdg::MOp &o = prog.ops[i];
    m[i] = fdg_mop{o.kind, o.nega, o.negb, o.negc, o.d, o.a, o.b, o.imm, o.c, o.param};
  }
  *ops = m; *n_ops = prog.ops.size();
  if (n_reg_used) *n_reg_used = prog.n_reg_used;
  if (n_lds_used) *n_lds_used = prog.n_lds_used;
  if (n_mem_used) *n_mem_used = prog.n_mem_used;
  if (n_acc_used) *n_acc_used = prog.n_acc_used;
  return FDG_OK;
}

// waves of a cooperative workgroup: 8 (two per SIMD, 256 registers each: the second wave covers memory latency) unless asked otherwise
static uint32_t coop_waves() { const char *e = fdg::knob("FDG_COOP_WAVES"); const int n = e ? std::atoi(e) : 8; return n == 4 ? 4u : (n == 16 ? 16u : 8u); }

// parameters of the pooled cooperative variant: four waves, one per SIMD, with the AGPR level (eight waves of 256 registers each -- FDG_POOL_WAVES=8 --
// spill to the panel and compute a quarter of the fold steps twice on the 4-loop vertex functions)
static uint32_t pool_waves() { const char *e = fdg::knob("FDG_POOL_WAVES"); const int n = e ? std::atoi(e) : 4; return n == 8 ? 8u : 4u; }
static fdg::OptParams pool_params(fdg::OptParams q, uint32_t nw) {
  q.n_reg = std::min<uint32_t>(q.n_reg ? q.n_reg : 120, 120);
  q.n_acc = nw >= 8 ? 0 : 124;
  q.n_land = 0;
  q.lookahead_lds = 32;
  if (const char *e = fdg::knob("FDG_POOL_LA_LDS")) q.lookahead_lds = (uint32_t)std::max(1, std::atoi(e));       // experiments
  return q;
}

static int coop_program_out(const fdg::CoopProgram &cp, uint32_t wave, fdg_mop **ops, uint64_t *n_ops, uint32_t *info);
// (shorter epochs read fewer leaves at a time: tried when the pool cannot hold what a window of epochs reads)
static void build_pool_auto(const fdg::Lowered &p, const fdg::OptParams &q, fdg::CoopProgram &cp, uint32_t nw) {
  for (uint32_t ep : {128u, 64u, 32u, 16u}) {
    fdg::build_pool_program(p, q, cp, nw, ep, 0);
    if (cp.supported || cp.why != "leaf pool exhausted") break;
  }
}

int fdg_graph_pool_program(const fdg_graph *g, const fdg_opt_params *q, uint32_t wave, fdg_mop **ops, uint64_t *n_ops, uint32_t *info) {
  const fdg::KnobMap knobs_now = knobs_snapshot(g);      // (a copy taken under the handle's mutex: fdg_graph_set_option may run on another thread)
  fdg::KnobScope knob_scope(g ? &knobs_now : nullptr);
  if (!g || !ops || !n_ops || wave >= fdg::CoopProgram::MAXW) { set_error("null argument or wave out of range"); return FDG_E_INVALID; }
  fdg::CoopProgram cp;
  const uint32_t nw = pool_waves();
  build_pool_auto(g->prog, pool_params(to_params(q), nw), cp, nw);
  if (cp.supported && wave >= cp.n_wave) { set_error("wave out of range"); return FDG_E_INVALID; }
  if (!cp.supported) { set_error("the pooled cooperative variant does not cover this graph: " + cp.why); return FDG_E_UNSUPPORTED; }
  return coop_program_out(cp, wave, ops, n_ops, info);
}

int fdg_graph_coop_program(const fdg_graph *g, const fdg_opt_params *q, uint32_t wave, fdg_mop **ops, uint64_t *n_ops, uint32_t *info) {
  const fdg::KnobMap knobs_now = knobs_snapshot(g);      // (a copy taken under the handle's mutex: fdg_graph_set_option may run on another thread)
  fdg::KnobScope knob_scope(g ? &knobs_now : nullptr);
  if (!g || !ops || !n_ops || wave >= fdg::CoopProgram::MAXW) { set_error("null argument or wave out of range"); return FDG_E_INVALID; }
  fdg::OptParams prm = to_params(q);
  if (!q || !q->n_acc) prm.n_acc = 124;
  fdg::CoopProgram cp;
  const uint32_t nw = coop_waves();
  if (nw >= 8) prm.n_acc = 0;
  if (nw == 16) prm.n_reg = std::min<uint32_t>(prm.n_reg, 58);
  fdg::build_coop_program(g->prog, prm, cp, nw);
  if (cp.supported && wave >= cp.n_wave) { set_error("wave out of range"); return FDG_E_INVALID; }
  if (!cp.supported) { set_error("the cooperative variant does not cover this graph: " + cp.why); return FDG_E_UNSUPPORTED; }
  return coop_program_out(cp, wave, ops, n_ops, info);
}

static int coop_program_out(const fdg::CoopProgram &cp, uint32_t wave, fdg_mop **ops, uint64_t *n_ops, uint32_t *info) {
  const fdg::OptProgram &prog = cp.wave[wave];
  fdg_mop *m = (fdg_mop *)std::malloc(std::max<size_t>(1, prog.ops.size()) * sizeof(fdg_mop));
  if (!m) { set_error("out of memory"); return FDG_E_NOMEM; }
  for (size_t i = 0; i < prog.ops.size(); ++i) {
    const fdg::MOp &o = prog.ops[i];
    m[i] = fdg_mop{o.kind, o.nega, o.negb, o.negc, o.d, o.a, o.b, o.imm, o.c, o.param};
  }
  *ops = m; *n_ops = prog.ops.size();
  if (info) {
    info[0] = prog.n_reg_used; info[1] = prog.n_lds_used; info[2] = prog.n_mem_used; info[3] = prog.n_acc_used;
    info[4] = cp.n_shared; info[5] = cp.n_epoch; info[6] = (uint32_t)cp.n_transfer; info[7] = (uint32_t)cp.n_duplicate;
  }
  return FDG_OK;
}

int fdg_graph_mc_program(const fdg_graph *g, const fdg_leaf_tables *tab, const fdg_opt_params *q, fdg_mop **ops, uint64_t *n_ops,
                         uint32_t *n_reg_used, uint32_t *n_lds_used, uint32_t *n_mem_used, uint32_t *n_acc_used) {
  const fdg::KnobMap knobs_now = knobs_snapshot(g);      // (a copy taken under the handle's mutex: fdg_graph_set_option may run on another thread)
  fdg::KnobScope knob_scope(g ? &knobs_now : nullptr);
  if (!g || !tab || !ops || !n_ops) { set_error("null argument"); return FDG_E_INVALID; }
  if (tab->n_leaf != g->prog.L) { set_error("leaf tables: n_leaf differs from the graph's"); return FDG_E_INVALID; }
  fdg::LeafSpec ls; ls.tab = tab; ls.kF = tab->kF; ls.beta = tab->beta; ls.lambda = tab->lambda;
  fdg::OptProgram prog;
  fdg::build_mc_program(g->prog, ls, to_params(q), prog);
  if (!prog.supported) { set_error("the fused ISA step does not cover this graph / these leaves: " + prog.why); return FDG_E_UNSUPPORTED; }
  fdg_mop *m = (fdg_mop *)std::malloc(std::max<size_t>(1, prog.ops.size()) * sizeof(fdg_mop));
  if (!m) { set_error("out of memory"); return FDG_E_NOMEM; }
  for (size_t i = 0; i < prog.ops.size(); ++i) {
    const fdg::MOp &o = prog.ops[i];
    m[i] = fdg_mop{o.kind, o.nega, o.negb, o.negc, o.d, o.a, o.b, o.imm, o.c, o.param};
  }
  *ops = m; *n_ops = prog.ops.size();
  if (n_reg_used) *n_reg_used = prog.n_reg_used;
  if (n_lds_used) *n_lds_used = prog.n_lds_used;
  if (n_mem_used) *n_mem_used = prog.n_mem_used;
  if (n_acc_used) *n_acc_used = prog.n_acc_used;
  return FDG_OK;
}

static bool has_opt_params(const fdg_graph *g) { return g->has_opt; }

static int assemble_isa(const fdg_graph *g, const fdg::OptProgram &prog, const std::string &dir, unsigned flags,
                        std::vector<char> &co, std::string &hash, const fdg::OptProgram *prog2 = nullptr,
                        const fdg::OptProgram *prog_acc = nullptr, const char *kname = "fdg_isa_eval",
                        const fdg::OptProgram *prog_rm = nullptr, uint32_t rm_bufs = 0, const fdg::CoopProgram *coop = nullptr,
                        const fdg::OptProgram *prog_rm_acc = nullptr, const fdg::CoopProgram *pool = nullptr,
                        const fdg::OptProgram *prog_rl = nullptr, const fdg::OptProgram *prog_rl_acc = nullptr) {
  const std::string src = fdg::emit_isa(g->prog, prog, kname, prog2, prog_acc, prog_rm, rm_bufs, coop, prog_rm_acc, pool, prog_rl, prog_rl_acc);
  char hbuf[40];
  std::snprintf(hbuf, sizeof hbuf, "%016llx", (unsigned long long)fnv1a(src, fnv1a("isa")));
  hash = hbuf;
  const std::string base = dir + "/fdg_isa_" + hbuf;
  if (!read_cached(dir, std::string("fdg_isa_") + hbuf + ".hsaco", co)) {
    // every intermediate under a process-unique name (ranks of one job assemble the same graph at the same time);
    // the code object appears under its final name by rename
    const std::string tmp = base + tmp_suffix();
    if (!write_file(tmp + ".s", src.c_str(), src.size())) { set_error("cannot write " + tmp + ".s"); return FDG_E_JIT; }
    const char *llvm = fdg::knob("FDG_LLVM_BIN");
    const std::string bin = llvm ? llvm : "/opt/rocm/lib/llvm/bin";
    int rc = run_cmd({bin + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", tmp + ".s", "-o", tmp + ".o"}, tmp + ".log");
    if (rc == 0) rc = run_cmd({bin + "/ld.lld", "-shared", tmp + ".o", "-o", tmp + ".hsaco"}, tmp + ".log");
    const std::string log = slurp_and_remove(tmp + ".log");
    std::remove((tmp + ".o").c_str());
    if (flags & FDG_SPEC_KEEP_SOURCE) std::rename((tmp + ".s").c_str(), (base + ".s").c_str());
    else std::remove((tmp + ".s").c_str());
    if (rc == 0) ::chmod((tmp + ".hsaco").c_str(), 0644);     // ld.lld creates it 0775/0777 & ~umask
    if (rc != 0 || std::rename((tmp + ".hsaco").c_str(), (base + ".hsaco").c_str()) != 0 || !read_file(base + ".hsaco", co)) {
      std::remove((tmp + ".hsaco").c_str());
      set_error("assembling the ISA kernel failed:\n" + log.substr(0, 2000));
      return FDG_E_JIT;
    }
  } else if (flags & FDG_SPEC_KEEP_SOURCE) {
    write_file(base + ".s", src.c_str(), src.size());
  }
  return FDG_OK;
}

static void install_isa(fdg_graph *g, const fdg::OptProgram &prog, std::vector<char> &co, const std::string &hash, unsigned flags,
                        const fdg::OptProgram *prog2 = nullptr, const fdg::OptProgram *prog_acc = nullptr,
                        const fdg::OptProgram *prog_rm = nullptr, uint32_t rm_bufs = 0, const fdg::CoopProgram *coop = nullptr,
                        const fdg::OptProgram *prog_rm_acc = nullptr, const fdg::CoopProgram *pool = nullptr,
                        const fdg::OptProgram *prog_rl = nullptr, const fdg::OptProgram *prog_rl_acc = nullptr) {
  if (g->module) { hipModuleUnload((hipModule_t)g->module); g->module = nullptr; }
  g->has_rl_acc = prog_rl != nullptr && prog_rl_acc != nullptr;
  g->fn_isa_rl_acc = nullptr;
  if (g->has_rl_acc) {
    uint32_t t7 = 0;
    for (const fdg::MOp &o : prog_rl_acc->ops) t7 = std::max(t7, fdg::mop_tmp_pairs(o.kind));
    g->isa7_vgpr = ((6 + 2 * std::max<uint32_t>(prog_rl_acc->n_reg_used, 1) + 2 * (g->prog.R + 2) + 2 * t7 + 2 + 3) & ~3u) + 2 * prog_rl_acc->n_acc_used;
    g->isa7_lds_bytes = ((prog_rl_acc->n_lds_used * 512u + 1023u) & ~1023u) + ((512u * g->prog.L + 1023u) & ~1023u);
    g->isa7_mem_slots = prog_rl_acc->n_mem_used;
  }
  g->has_rl = prog_rl != nullptr;
  g->fn_isa_rl = nullptr;
  if (g->has_rl) {
    uint32_t t = 0;
    for (const fdg::MOp &o : prog_rl->ops) t = std::max(t, fdg::mop_tmp_pairs(o.kind));
    g->isa6_vgpr = ((6 + 2 * std::max<uint32_t>(prog_rl->n_reg_used, 1) + 2 * t + 2 + 3) & ~3u) + 2 * prog_rl->n_acc_used;
    g->isa6_lds_bytes = ((prog_rl->n_lds_used * 512u + 1023u) & ~1023u) + ((512u * g->prog.L + 1023u) & ~1023u);
    g->isa6_mem_slots = prog_rl->n_mem_used;
    g->rl_valu = prog_rl->n_valu;
  }
  g->has_pool = pool && pool->supported;
  g->fn_isa_pool = nullptr;
  if (g->has_pool) {
    g->pool_panel_wg = 0;
    for (uint32_t w = 0; w < pool->n_wave; ++w) g->pool_panel_wg += std::max<uint32_t>(pool->wave[w].n_mem_used, 1) * 512u;
    g->pool_threads = 64 * pool->n_wave;
    g->pool_fetch = (uint32_t)pool->n_transfer;      // leaves brought from memory per tile
    g->pool_unit = pool->pool_unit;
    g->pool_valu = 0;
    for (uint32_t w = 0; w < pool->n_wave; ++w) g->pool_valu += pool->wave[w].n_valu;
  }
  g->has_coop = coop && coop->supported;
  g->coop_enabled = g->has_coop;
  g->fn_isa_coop = nullptr;
  if (g->has_coop) {
    g->coop_panel_wg = 0;
    for (uint32_t w = 0; w < coop->n_wave; ++w) g->coop_panel_wg += std::max<uint32_t>(coop->wave[w].n_mem_used, 1) * 512u;
    g->coop_lds_bytes = (coop->n_shared + coop->n_wave * coop->n_priv_lds) * 512u;
    g->coop_threads = 64 * coop->n_wave;
  }
  g->code_object.swap(co);
  g->isa = true;
  g->fn_isa = nullptr;
  g->fn_isa_nt = g->fn_isa_acc_nt = nullptr;
  g->fn_isa_w2 = nullptr;
  g->fn_isa_acc = nullptr;
  g->has_acc = prog_acc != nullptr;
  auto tmp_vgprs = [](const fdg::OptProgram &q) { uint32_t t = 0; for (const fdg::MOp &o : q.ops) t = std::max(t, fdg::mop_tmp_pairs(o.kind)); return 2 * t; };
  if (prog_acc) {
    g->isa3_vgpr = prog_acc->params.acc_in_agpr
                       ? ((6 + 2 * std::max<uint32_t>(prog_acc->n_reg_used, 1) + 2 * 3 + tmp_vgprs(*prog_acc) + 3) & ~3u) + 2 * (prog_acc->n_acc_used + g->prog.R)
                       : ((6 + 2 * std::max<uint32_t>(prog_acc->n_reg_used, 1) + 2 * (g->prog.R + 2) + tmp_vgprs(*prog_acc) + 3) & ~3u) + 2 * prog_acc->n_acc_used;
    g->isa3_lds_bytes = prog_acc->n_lds_used * 512u;
    g->isa3_mem_slots = prog_acc->n_mem_used;
  }
  g->has_rm = prog_rm != nullptr && rm_bufs > 0;
  g->fn_isa_rm = nullptr;
  if (g->has_rm) {
    g->isa4_vgpr = ((6 + 2 * std::max<uint32_t>(prog_rm->n_reg_used, 1) + tmp_vgprs(*prog_rm) + 10 + 3) & ~3u) + 2 * prog_rm->n_acc_used;
    g->isa4_lds_bytes = ((prog_rm->n_lds_used * 512u + 1023u) & ~1023u) + rm_bufs * 8192u;
    g->isa4_mem_slots = prog_rm->n_mem_used;
  }
  g->has_rm_acc = g->has_rm && prog_rm_acc != nullptr;
  g->fn_isa_rm_acc = nullptr;
  if (g->has_rm_acc) {
    g->isa5_vgpr = ((6 + 2 * std::max<uint32_t>(prog_rm_acc->n_reg_used, 1) + 2 * (g->prog.R + 2) + tmp_vgprs(*prog_rm_acc) + 10 + 3) & ~3u) + 2 * prog_rm_acc->n_acc_used;
    g->isa5_lds_bytes = ((prog_rm_acc->n_lds_used * 512u + 1023u) & ~1023u) + rm_bufs * 8192u;
    g->isa5_mem_slots = prog_rm_acc->n_mem_used;
  }
  g->has_w2 = prog2 != nullptr;
  if (prog2) {
    g->isa2_vgpr = ((6 + 4 * std::max<uint32_t>(prog2->n_reg_used, 1) + 3) & ~3u) + 4 * prog2->n_acc_used;
    g->isa2_lds_bytes = prog2->n_lds_used * 1024u;
    g->isa2_mem_slots = prog2->n_mem_used;
  }
  g->isa_vgpr = ((6 + 2 * std::max<uint32_t>(prog.n_reg_used, 1) + tmp_vgprs(prog) + 3) & ~3u) + 2 * prog.n_acc_used;
  g->isa_lds_bytes = prog.n_lds_used * 512u;
  g->isa_mem_slots = prog.n_mem_used;
  g->spec_vgpr = g->isa_vgpr; g->spec_lds = g->isa_lds_bytes; g->spec_scratch = 0;
  {
    const fdg::OptProgram *pp[3] = {&prog, prog_acc, g->has_rm ? prog_rm : nullptr};
    for (int i = 0; i < 3; ++i) {
      g->st_valu[i] = pp[i] ? pp[i]->n_valu : 0;
      g->st_ld_leaf[i] = pp[i] ? (uint32_t)pp[i]->n_ld_leaf : 0;
      g->st_panel[i] = pp[i] ? (uint32_t)(pp[i]->n_ld_mem + pp[i]->n_st_mem) : 0;
      g->st_lds[i] = pp[i] ? (uint32_t)(pp[i]->n_ld_lds + pp[i]->n_st_lds) : 0;
    }
    g->rm_bufs = g->has_rm ? rm_bufs : 0;
    g->last_kernel = "";
  }
  g->spec_source_hash = hash;
  g->spec_flags = flags;
}

// every program of a handle specialised with FDG_SPEC_FAST_MATH fuses products into sums (v_fma_f64)
static void build_prog(const fdg_graph *g, fdg::OptParams prm, fdg::OptProgram &out) {
  prm.fma = prm.fma || g->isa_fma;
  if (const char *e = fdg::knob("FDG_LA_LDS")) prm.lookahead_lds = (uint32_t)std::max(1, std::atoi(e));       // experiments
  if (const char *e = fdg::knob("FDG_LA_LDS_B")) { if (prm.n_acc) prm.lookahead_lds = (uint32_t)std::max(1, std::atoi(e)); }   // ... one wave per SIMD only
  if (fdg::knob("FDG_ROOTS_LAST") && !g->isa_fma) prm.roots_last = true;   // experiment: every program's root stores back to back at the tile's end (one block of a tile-major root batch)
  fdg::build_opt_program(g->prog, prm, out);
}

// the automatic configurations: S tiny graphs, A two waves/SIMD, B one wave/SIMD + AGPR level
static fdg::OptParams cfg_S() { fdg::OptParams S; S.n_reg = 28; S.n_lds = 1; S.n_acc = 0; S.lookahead_leaf = 300; S.vn_window = 200; return S; }
static fdg::OptParams cfg_A() { fdg::OptParams A; A.n_reg = 120; A.n_lds = 40; A.n_acc = 0; A.lookahead_leaf = 300; A.vn_window = 200; return A; }
static fdg::OptParams cfg_B() {
  fdg::OptParams B; B.n_reg = 120; B.n_lds = 80; B.n_acc = 124; B.lookahead_leaf = 100; B.lookahead_mem = 64; B.vn_window = 1000; return B;
}

// experiment switch: FDG_LAND=<n> AGPR landing slots for the leaf loads of the one-wave configuration (0 = none)
static void apply_land_env(fdg::OptParams &q) {
  if (const char *e = fdg::knob("FDG_LAND")) q.n_land = q.n_acc ? std::min<uint32_t>((uint32_t)std::max(0, std::atoi(e)), q.n_acc / 2) : 0;
  if (const char *e = fdg::knob("FDG_LAND_LA")) q.lookahead_land = (uint32_t)std::max(1, std::atoi(e));
}

static fdg::OptParams auto_program(const fdg_graph *g, fdg::OptProgram &prog) {
  fdg::OptProgram ps;
  build_prog(g, cfg_S(), ps);
  const bool small_ok = ps.supported && ps.n_ld_leaf <= g->prog.n_live_leaf && ps.n_ld_lds + ps.n_st_lds + ps.n_ld_mem + ps.n_st_mem == 0;
  if (small_ok) { prog = std::move(ps); return cfg_S(); }
  build_prog(g, cfg_A(), prog);
  if (prog.supported && (prog.n_ld_mem + prog.n_st_mem) * 100 > prog.n_valu) {   // > 1 % of the ops touch the HBM panel
    fdg::OptProgram pb;
    fdg::OptParams qb = cfg_B();
    apply_land_env(qb);
    build_prog(g, qb, pb);
    if (pb.supported) { prog = std::move(pb); return qb; }
  }
  return cfg_A();
}

// On-device selection among a handful of configurations: each candidate is assembled, run on a
// synthetic batch that fills the chip twice, and the fastest is kept; the choice is remembered in
// the cache directory (keyed by the table), so tuning happens once per graph.
static std::string tuned_path(const fdg_graph *g, const std::string &dir) {
  const fdg::Lowered &p = g->prog;
  uint64_t th = fnv1a("tuned-v2");
  auto mix = [&](const void *d, size_t n) { th = fnv1a(std::string((const char *)d, n), th); };
  mix(&p.L, 4); mix(&p.N, 4); mix(&p.R, 4);
  if (p.N) { mix(p.op.data(), p.op.size()); mix(p.power.data(), p.power.size() * 4); mix(p.off.data(), p.off.size() * 4); }
  if (p.E) { mix(p.idx.data(), p.idx.size() * 4); mix(p.fac.data(), p.fac.size() * 8); }
  if (p.R) mix(p.root_slot.data(), p.root_slot.size() * 4);
  if (!p.sched_group.empty()) mix(p.sched_group.data(), p.sched_group.size() * 4);
  if (p.assoc_interp) mix("eval!", 5);
  char hb[40];
  std::snprintf(hb, sizeof hb, "%016llx", (unsigned long long)th);
  return dir + "/fdg_tuned_" + hb + ".txt";
}

// two samples per lane: a value is four VGPRs, so 60 of them fill the 2-waves-per-SIMD budget; T = tiny graphs
static fdg::OptParams cfg_W2() { fdg::OptParams q; q.n_reg = 60; q.n_lds = 20; q.n_acc = 0; q.lookahead_leaf = 300; q.vn_window = 200; return q; }
static fdg::OptParams cfg_W2T() { fdg::OptParams q; q.n_reg = 14; q.n_lds = 1; q.n_acc = 0; q.lookahead_leaf = 300; q.vn_window = 200; return q; }

// The wide variant pays when the graph is HBM-bound: 16-byte accesses stream faster than 8-byte ones.
// Static estimate: 8(L+R) bytes at 5 TB/s against the executed fold steps at 15 T op/s.
static bool auto_program_w2(const fdg_graph *g, fdg::OptProgram &p2) {
  // Measured on MI355X: the wide variant is SLOWER (sigma2 58 % vs 64 % of HBM peak, gv_sigma4 63 % vs
  // 66 %), so it is opt-in (FDG_ISA_W2=1) and kept only as an experiment.
  if (!fdg::knob("FDG_ISA_W2")) return false;
  const fdg::Lowered &p = g->prog;
  build_prog(g, cfg_W2T(), p2);
  bool ok = p2.supported && p2.n_ld_leaf <= p.n_live_leaf && p2.n_ld_lds + p2.n_st_lds + p2.n_ld_mem + p2.n_st_mem == 0;
  if (!ok) {
    build_prog(g, cfg_W2(), p2);
    ok = p2.supported && p2.n_ld_mem + p2.n_st_mem == 0;
  }
  if (!ok) return false;
  for (const fdg::MOp &o : p2.ops)      // the wide kernel prints plain fold steps only
    if (fdg::mop_is_macro(o.kind) || o.kind == fdg::M_FMAK || o.kind == fdg::M_CONST || o.kind == fdg::M_FMA || o.kind == fdg::M_FMAC) return false;
  const double t_hbm = 8.0 * ((double)p.L + p.R) / 5e12, t_valu = (double)p2.n_valu / 15e12;
  return t_hbm > 1.2 * t_valu;
}

// The fused-accumulate kernel keeps R accumulators, the weight and a temporary in VGPR pairs above
// the values, so its program is allocated with that many fewer registers (same configuration otherwise).
static bool build_acc_program(const fdg_graph *g, const fdg::OptParams &chosen, fdg::OptProgram &pa) {
  const uint32_t extra = g->prog.R + 2;
  // (up to 40 roots: the 26 rows of example/benchmark_GV.jl's vertex function keep their sums in registers; beyond that the
  //  accumulators would take more than a third of the value registers and the roots go through the column-major scratch)
  // (few roots: eight value registers next to the accumulators are enough -- the tiny-graph configuration of the 2-loop
  //  self-energies keeps 28; many roots must leave the values a working set worth having)
  if (g->prog.R > 40 && g->prog.R <= 124 && chosen.n_reg >= 100 && !fdg::knob("FDG_ISA_NO_FUSED_ACC") && !fdg::knob("FDG_ISA_NO_AGPR_ACC")) {
    // 41 ... 124 roots (round 4): the accumulators in AGPR pairs -- the file a kernel launched with one wave per SIMD has to itself (the graphs
    // bound by memory are launched that way whatever their registers allow; the one-wave configuration uses it for spills and keeps what
    // the accumulators leave) -- instead of the detour through the root scratch, which cost the 84-root 3-loop vertex function a third.
    fdg::OptParams q = chosen;
    q.acc_in_agpr = true;
    q.reserve_pairs = 3;                                             // the weight and two temporaries
    q.n_reg = std::min<uint32_t>(chosen.n_reg, (256 - 6 - 2 * 3 - 8) / 2);
    if (chosen.n_acc) q.n_acc = std::min<uint32_t>(chosen.n_acc, 124 - g->prog.R);
    build_prog(g, q, pa);
    return pa.supported && pa.n_acc_used + g->prog.R <= 124;
  }
  if (g->prog.R == 0 || g->prog.R > 40 || chosen.n_reg < extra + 8 || (g->prog.R > 16 && chosen.n_reg < extra + 64) || fdg::knob("FDG_ISA_NO_FUSED_ACC")) return false;
  fdg::OptParams q = chosen;
  // stay inside the occupancy step of the eval kernel (VGPRs per wave: 64 -> 8 waves/SIMD ... 256 -> 2, 512 -> 1)
  static const uint32_t steps[] = {64, 72, 80, 96, 128, 168, 256, 512};
  const uint32_t v0 = ((6 + 2 * chosen.n_reg + 3) & ~3u) + 2 * chosen.n_acc;
  uint32_t budget = 512;
  for (uint32_t st : steps) if (st >= v0) { budget = st; break; }
  const uint32_t arch = std::min<uint32_t>(budget - 2 * chosen.n_acc, 256);   // architectural VGPRs end at v255
  if (arch < 6 + 2 * extra + 16) return false;
  q.n_reg = std::min<uint32_t>(chosen.n_reg, (arch - 6 - 2 * extra) / 2);
  q.reserve_pairs = extra;
  build_prog(g, q, pa);
  return pa.supported;
}

// The row-major variant (compile_Python's [B, L] input read in place, csrc/fdg_isa.cpp): the same configuration with
// part of the LDS budget turned into staging buffers of 8 KB, nine VGPRs of addresses, and leaf loads that come from
// LDS (short prefetch distance).  Not for the tiny-graph configuration (its waves have 5 KB of LDS each; such graphs
// take the HIP-source companion) nor for graphs of fewer than 16 leaves.
static uint32_t build_rm_program(const fdg_graph *g, const fdg::OptParams &chosen, fdg::OptProgram &pr, fdg::OptParams *qsel = nullptr) {
  if (g->prog.L < 16 || fdg::knob("FDG_ISA_NO_RM")) return 0;
  // (graphs in the tiny-graph configuration too, round 4: their contiguous rows take the linear variant, but rows with padding between them
  //  had only the transposition pass left -- 0.2 of the HBM roof on the 3-loop self-energy; FDG_ISA_RM_TINY=0 restores that)
  if (chosen.n_reg < 100 && fdg::knob("FDG_ISA_RM_TINY") && fdg::knob("FDG_ISA_RM_TINY")[0] == '0') return 0;
  // One wave per SIMD whatever the leaf-major kernel runs with: 40 KB of LDS per wave hold up to four staging buffers -- the
  // stream of first uses plus the few chunks a schedule keeps coming back to -- and the AGPR level makes up for the LDS
  // slots given away.  (Graphs that stream leaves are bound by latency, not by occupancy: DESIGN.md 6.)  The fewest
  // buffers that keep re-fetching within a quarter of the chunk count are taken: a small graph then leaves room for
  // more waves per CU.  A program that would move more than 2.5x the matrix (leaves re-read all over a huge graph)
  // gets no such variant: the chunked transposition in front of the leaf-major kernel is cheaper there.
  const uint32_t n_chunk = (g->prog.L + 15) / 16;
  const char *e = fdg::knob("FDG_ISA_RM_BUFS");
  // Two waves per SIMD when the program is small enough: the variant is bound by memory latency (one wave per SIMD spends
  // ~46 % of its cycles waiting for its chunks), and what hides latency is a second wave with its own two
  // buffers in flight.  Budget per wave: 256 registers (120 values + the nine address registers, no AGPR level) and
  // 20 KB of LDS = two staging buffers + eight slots.  Taken when nothing then spills to the HBM panel and two buffers
  // keep re-fetching within a quarter of the chunk count.
  // (Up to eight chunks -- 128 leaves: with longer rows the two buffers of a wave are too short a window; measured round 4,
  //  profiles/r04_log_rm_bufs.txt: 111 leaves +20 % over one wave per SIMD, 175 leaves -16 %.)
  const char *ew = fdg::knob("FDG_ISA_RM_WAVES");
  if (!(ew && std::atoi(ew) == 1) && !e && (n_chunk <= 8 || (ew && std::atoi(ew) == 2))) {
    // (both root orders are tried: in the reference's order the first uses of the leaves walk the row monotonically; round 6: also with
    //  the leaves loaded once and with shorter value-numbering windows, as in the one-wave search below)
    const char *lo_env2 = fdg::knob("FDG_RM_LEAVES_ONCE"), *vn_env2 = fdg::knob("FDG_RM_VN");
    std::vector<uint32_t> windows2 = {chosen.vn_window};
    if (vn_env2) windows2 = {(uint32_t)std::atoi(vn_env2)};
    else if (g->prog.N <= 60000) for (uint32_t w : {1000u, 400u, 200u, 100u}) if (chosen.vn_window == 0 || w < chosen.vn_window) windows2.push_back(w);
    const char *pp = fdg::knob("FDG_ISA_RM_PANEL_PCT");
    const uint64_t panel_pct = pp ? (uint64_t)std::atoi(pp) : 0;
    double best2 = 1e300;
    fdg::OptProgram cand2;
    fdg::OptParams qbest;
    for (uint32_t vw : windows2)
    for (int variant = 0; variant < 4; ++variant) {
      const int keep = variant & 1, once = variant >> 1;
      if (lo_env2 && (lo_env2[0] == '1') != (once != 0)) continue;
      fdg::OptParams q = cfg_A();
      q.vn_window = vw;
      q.n_lds = 8;
      q.reserve_pairs = 5;
      q.lookahead_leaf = 48;
      if (const char *la = fdg::knob("FDG_ISA_RM_LA")) q.lookahead_leaf = (uint32_t)std::max(1, std::atoi(la));
      q.leaves_once = once != 0;
      q.keep_root_order = keep != 0;
      q.roots_last = true;             // the R stores of a row back to back: they share cache lines when the roots are row-major too (+7-11 %)
      build_prog(g, q, cand2);
      if (!cand2.supported || (cand2.n_ld_mem + cand2.n_st_mem) * 100 > cand2.n_valu * panel_pct) continue;
      uint64_t fetches = 0, gathers = 0;
      fdg::rm_plan_stats(g->prog, cand2, 2, fetches, gathers);
      if (fdg::knob("FDG_RM_DEBUG")) std::fprintf(stderr, "[rm] two waves per SIMD (root order %d, leaves once %d, vn %u): %u chunks, %llu fetches, %llu gathers with 2 buffers, %llu fold steps, %llu panel; lds slots %u\n", keep, once, vw, n_chunk,
                                                  (unsigned long long)fetches, (unsigned long long)gathers, (unsigned long long)cand2.n_valu, (unsigned long long)(cand2.n_ld_mem + cand2.n_st_mem), cand2.n_lds_used);
      if (!(fetches * 4 <= (uint64_t)n_chunk * 5 + 4 && (fetches * 8192 + gathers * 2048) * 2 <= (uint64_t)g->prog.L * 512 * 5)) continue;
      const double cost = (double)cand2.n_valu * 4.5 + (double)fetches * 800.0 + (double)gathers * 400.0 + (double)(cand2.n_ld_mem + cand2.n_st_mem) * 60.0 + (double)(cand2.n_ld_lds + cand2.n_st_lds) * 8.0;
      if (cost < best2) { best2 = cost; pr = cand2; qbest = q; }
    }
    if (best2 < 1e300) { if (qsel) *qsel = qbest; return 2; }
  }
  // Four buffers first -- the deepest prefetch: +5-18 % on the graphs whose values then still fit the registers and AGPRs (no panel access
  // with the 14 LDS slots that remain) -- then the fewest buffers that keep re-fetching low (graphs that need their LDS slots:
  // four buffers -9 % on the 5-loop Parquet self-energy; profiles/r04_log_rm_bufs.txt).
  const uint32_t first = e ? (uint32_t)std::max(1, std::min(4, std::atoi(e))) : 2u;
  // Round 6: per buffer count the candidates are {root order} x {leaves re-loadable | loaded once (fdg_opt.h: leaves_once)} x {value-numbering
  // window: the tile-major kernel's, and shorter ones -- a long window keeps values alive that the staging buffers' share of the LDS no longer
  // has room for}, compared by an estimate of what a tile costs its wave in cycles: a fold step 4.5, a chunk fetch 800 (eight LDS-direct loads),
  // a gathered leaf 400 (64 lines for 64 doubles), a panel access 60, an LDS move 8, an AGPR move 12.  (parquet_sigma5: 33 fetches + 13 gathers
  // + 231 panel accesses with re-loadable leaves and the tile-major window -> 17 fetches and a few dozen panel accesses; 0.22 -> 0.35 of the
  // HBM roof with leaves_once alone, profiles/r06_log_sweep_d.txt.)  FDG_RM_LEAVES_ONCE=0 / 1, FDG_RM_VN=<window> force one form.
  const char *lo_env = fdg::knob("FDG_RM_LEAVES_ONCE"), *vn_env = fdg::knob("FDG_RM_VN");
  std::vector<uint32_t> windows = {chosen.vn_window};
  if (vn_env) windows = {(uint32_t)std::atoi(vn_env)};
  else if (g->prog.N <= 60000) for (uint32_t w : {2000u, 1000u, 400u, 200u}) if (chosen.vn_window == 0 || w < chosen.vn_window) windows.push_back(w);
  auto est_cycles = [](const fdg::OptProgram &c, uint64_t fetches, uint64_t gathers) {
    return (double)c.n_valu * 4.5 + (double)fetches * 800.0 + (double)gathers * 400.0 + (double)(c.n_ld_mem + c.n_st_mem) * 60.0 +
           (double)(c.n_ld_lds + c.n_st_lds) * 8.0 + (double)(c.n_ld_acc + c.n_st_acc) * 12.0;
  };
  for (uint32_t pass = e ? 1u : 0u; pass < 2; ++pass)
  for (uint32_t bufs = pass == 0 ? 4u : first; bufs <= 4; ++bufs) {
    double best_cost = 1e300; uint64_t best_fetches = 0, best_gathers = 0, best_panel = 0;
    bool any = false;
    fdg::OptProgram cand;
    for (uint32_t vw : windows)
    for (int variant = 0; variant < 4; ++variant) {
      const int keep = variant & 1, once = variant >> 1;
      if (lo_env && (lo_env[0] == '1') != (once != 0)) continue;
      fdg::OptParams q = cfg_B();
      q.vn_window = vw;
      q.n_lds = 80u - bufs * 16u - 2u;                             // (two slots lost to the 1 KB alignment of the buffers)
      q.reserve_pairs = 5;
      q.lookahead_leaf = 48;
      if (const char *la = fdg::knob("FDG_ISA_RM_LA")) q.lookahead_leaf = (uint32_t)std::max(1, std::atoi(la));
      q.leaves_once = once != 0;
      q.keep_root_order = keep != 0;
      q.roots_last = true;             // the R stores of a row back to back: they share cache lines when the roots are row-major too (+7-11 %)
      q.rm_pair = bufs == 4 && fdg::knob("FDG_RM_PAIR") && fdg::knob("FDG_RM_PAIR")[0] == '1';
      build_prog(g, q, cand);
      if (!cand.supported) { if (!any && variant == 0 && vw == windows[0] && pass == 1) return 0; continue; }
      any = true;
      uint64_t fetches = 0, gathers = 0;
      fdg::rm_plan_stats(g->prog, cand, bufs, fetches, gathers);
      const double cost = est_cycles(cand, fetches, gathers);
      if (fdg::knob("FDG_RM_DEBUG")) std::fprintf(stderr, "[rm] one wave per SIMD (root order %d, leaves once %d, vn %u): %u chunks, %llu fetches, %llu gathers with %u buffers, %llu fold steps, %llu panel accesses, %llu LDS and %llu AGPR moves: %.0f cycles\n", keep, once, vw, n_chunk,
                                                  (unsigned long long)fetches, (unsigned long long)gathers, bufs, (unsigned long long)cand.n_valu, (unsigned long long)(cand.n_ld_mem + cand.n_st_mem),
                                                  (unsigned long long)(cand.n_ld_lds + cand.n_st_lds), (unsigned long long)(cand.n_ld_acc + cand.n_st_acc), cost);
      if (cost < best_cost) { best_cost = cost; best_fetches = fetches; best_gathers = gathers; best_panel = cand.n_ld_mem + cand.n_st_mem; pr = std::move(cand); if (qsel) *qsel = q; }
    }
    const bool cheap = best_fetches * 4 <= (uint64_t)n_chunk * 5 + 4;
    if (pass == 0) { if (best_cost < 1e300 && cheap && best_panel == 0) return 4; break; }
    if (!cheap && bufs < 4 && !e) continue;
    {   // (FDG_ISA_RM_MAX_TRAFFIC=<tenths>: the bound on what the variant may move, in tenths of the matrix; default 25)
      const uint64_t tenths = fdg::knob("FDG_ISA_RM_MAX_TRAFFIC") ? (uint64_t)std::max(10, std::atoi(fdg::knob("FDG_ISA_RM_MAX_TRAFFIC"))) : 25;
      if ((best_fetches * 8192 + best_gathers * 2048) * 10 > (uint64_t)g->prog.L * 512 * tenths) return 0;
    }
    return bufs;
  }
  return 0;
}

struct IsaVariants {
  fdg::OptProgram p2, pa, pr, pra, prl, prla;
  bool rl = false, rl_acc = false;
  fdg::CoopProgram coop, pool;
  bool w2 = false, acc = false, rm_acc = false;
  uint32_t rm_bufs = 0;
  int coop_verdict = -1;       // a remembered measurement: 0 = the cooperative variant loses, 4 / 8 = it wins with that many waves; -1: none
};
// The cooperative variant (four waves of a CU on one tile) is assembled for programs whose one-wave form
// spills to the HBM panel in earnest (more than one panel access per 20 fold steps); FDG_ISA_COOP=1 / 0 forces / forbids.
static void build_coop(const fdg_graph *g, const fdg::OptProgram &prog, IsaVariants &V) {
  const char *e = fdg::knob("FDG_ISA_COOP");
  if (e && e[0] == '0') return;
  (void)prog;
  fdg::OptParams q = cfg_B();
  q.vn_window = 200;
  if (const char *x = fdg::knob("FDG_COOP_LA_LEAF")) q.lookahead_leaf = (uint32_t)std::atoi(x);
  if (const char *x = fdg::knob("FDG_COOP_LA_MEM")) q.lookahead_mem = (uint32_t)std::atoi(x);
  if (const char *x = fdg::knob("FDG_COOP_LA_LDS")) q.lookahead_lds = (uint32_t)std::atoi(x);
  if (!(e && e[0] == '1')) {
    fdg::OptProgram ref;                      // the one-wave program without recomputation decides
    build_prog(g, q, ref);
    if (!ref.supported || (ref.n_ld_mem + ref.n_st_mem) * 20 <= ref.n_valu) return;
  }
  for (uint32_t nw : {V.coop_verdict > 0 ? (uint32_t)V.coop_verdict : coop_waves(), 4u}) {    // (four waves when eight run out of shared slots)
    fdg::OptParams qw = q;
    if (nw >= 8) qw.n_acc = 0;                // two waves per SIMD: 256 registers each
    if (nw == 16) qw.n_reg = 58;              // four per SIMD: 128 registers each
    fdg::build_coop_program(g->prog, qw, V.coop, nw);
    if (V.coop.supported || V.coop_verdict > 0) break;
  }
}
static void build_variants(const fdg_graph *g, const fdg::OptParams &chosen, bool allow_w2, IsaVariants &V) {
  V.w2 = allow_w2 && !fdg::knob("FDG_ISA_NO_W2") && auto_program_w2(g, V.p2);
  V.acc = build_acc_program(g, chosen, V.pa);
  fdg::OptParams qrm;
  V.rm_bufs = build_rm_program(g, chosen, V.pr, &qrm);
  // fused accumulation of row-major input: the row-major program once more with R + 2 fewer value registers
  V.rm_acc = false;
  if (V.rm_bufs && V.acc && g->prog.R >= 1 && g->prog.R <= 40 && !fdg::knob("FDG_ISA_NO_RM_ACC")) {
    qrm.reserve_pairs += g->prog.R + 2;
    qrm.roots_last = false;           // (accumulation: a root is consumed where it is finished)
    build_prog(g, qrm, V.pra);
    uint64_t fetches = 0, gathers = 0;
    if (V.pra.supported) fdg::rm_plan_stats(g->prog, V.pra, V.rm_bufs, fetches, gathers);
    V.rm_acc = V.pra.supported && (fetches * 8192 + gathers * 2048) * 2 <= (uint64_t)g->prog.L * 512 * 5 &&
               ((V.pra.n_lds_used * 512u + 1023u) & ~1023u) + V.rm_bufs * 8192u <= 160u * 1024u;
  }
}
// The pooled cooperative variant (fdg_opt.h: build_pool_program) is assembled for graphs with at least two roots per wave whose one-wave
// program goes back to memory for what it had already: leaf loads + panel accesses above twice the live leaves (the two vertex functions
// of the reference's benchmark programs: 2.9 x and 1.1-1.4 x).  FDG_ISA_POOL=1 / 0 forces / forbids.
static void build_pool(const fdg_graph *g, const fdg::OptProgram &prog, IsaVariants &V) {
  V.pool = fdg::CoopProgram();
  const char *e = fdg::knob("FDG_ISA_POOL");
  if (e && e[0] == '0') return;
  if (g->isa_fma || prog.mc_n_k || prog.mc_n_t) return;
  // (twice the live leaves: the GV vertex function's one-wave program makes 2.9 x, the Parquet one 1.1-1.4 x depending on the configuration
  //  the tuner picked -- and runs 3.4 against 2.6e8 evaluations/s pooled)
  if (!(e && e[0] == '1') && (prog.n_ld_leaf + prog.n_ld_mem + prog.n_st_mem) <= (uint64_t)g->prog.n_live_leaf * 2) return;
  const uint32_t nw = pool_waves();
  fdg::OptParams q = pool_params(cfg_B(), nw);
  q.vn_window = prog.params.vn_window;
  if (const char *v = fdg::knob("FDG_POOL_VN")) q.vn_window = (uint32_t)std::max(0, std::atoi(v));      // (experiment: the pooled waves' own value-numbering window)
  build_pool_auto(g->prog, q, V.pool, nw);
  if (fdg::knob("FDG_POOL_DEBUG")) std::fprintf(stderr, "[pool] %s: %u waves, %u epochs, %llu fetches for %u leaves, %llu duplicated fold steps (%s)\n", V.pool.supported ? "built" : "not built",
                                                  V.pool.n_wave, V.pool.n_epoch, (unsigned long long)V.pool.n_fetch, g->prog.n_live_leaf, (unsigned long long)V.pool.n_duplicate, V.pool.why.c_str());
}
// The linear row-major variant (csrc/fdg_isa.cpp: `rl`): contiguous rows (sample stride == L) of graphs whose 64-row tile, 512 L bytes, leaves
// room for two waves per CU or more.  One wave per SIMD at most, so the AGPR level is there; a leaf comes back from the image by an LDS read.
static void build_rl(const fdg_graph *g, const fdg::OptParams &chosen, IsaVariants &V) {
  V.rl = false;
  const char *e = fdg::knob("FDG_ISA_RL");
  // Three waves per CU or more (image + eight LDS slots three times into 160 KB: up to 98 leaves): with two, a wave that waits for its
  // tile has one partner to cover it and the variant loses to the chunked one -- measured at 116 and 155 leaves: 2.26 vs 2.75e9 and
  // 2.34 vs 2.72e9 evaluations/s; equal at 111 (profiles/r04_log_rl_big.txt).  FDG_ISA_RL_MAX_KB overrides the bound on the image.
  const uint32_t rl_max_bytes = fdg::knob("FDG_ISA_RL_MAX_KB") ? (uint32_t)std::atoi(fdg::knob("FDG_ISA_RL_MAX_KB")) * 1024u : (160u * 1024u / 3u - 8u * 512u);
  if ((e && e[0] == '0') || g->prog.L < 2 || ((512u * g->prog.L + 1023u) & ~1023u) > rl_max_bytes || g->isa_fma) return;
  fdg::OptParams q = cfg_B();
  q.vn_window = chosen.vn_window;
  q.n_lds = 512u * g->prog.L > 72u * 1024u ? 0 : 8;
  q.reserve_pairs = 1;
  q.lookahead_leaf = 32;
  q.pool_leaves = true;          // (an evicted leaf is read again from the image: cheap, and it is not parked anywhere)
  q.roots_last = true;
  if (const char *la = fdg::knob("FDG_ISA_RL_LA")) q.lookahead_leaf = (uint32_t)std::max(1, std::atoi(la));
  build_prog(g, q, V.prl);
  V.rl = V.prl.supported && V.prl.n_ld_mem + V.prl.n_st_mem == 0;
  // fused accumulation over the same rows: R accumulators, the weight and a temporary above the values
  V.rl_acc = false;
  if (V.rl && V.acc && g->prog.R >= 1 && g->prog.R <= 40 && !fdg::knob("FDG_ISA_NO_RL_ACC")) {
    q.reserve_pairs += g->prog.R + 2;
    q.n_reg = std::min<uint32_t>(q.n_reg, (256u - 6u - 2u * (g->prog.R + 2) - 2u - 8u) / 2u);
    q.roots_last = false;            // (a root is consumed where it is finished)
    build_prog(g, q, V.prla);
    V.rl_acc = V.prla.supported && V.prla.n_ld_mem + V.prla.n_st_mem == 0 &&
               ((V.prla.n_lds_used * 512u + 1023u) & ~1023u) + ((512u * g->prog.L + 1023u) & ~1023u) <= 160u * 1024u / 3u;
  }
}
static int assemble_and_install(fdg_graph *g, const fdg::OptProgram &prog, const std::string &dir, unsigned flags, IsaVariants &V) {
  std::vector<char> co; std::string hash;
  if (V.coop_verdict != 0) build_coop(g, prog, V);
  build_pool(g, prog, V);
  build_rl(g, prog.params, V);
  const fdg::CoopProgram *coop = V.coop.supported ? &V.coop : nullptr;
  const fdg::CoopProgram *pool = V.pool.supported ? &V.pool : nullptr;
  const int rc = assemble_isa(g, prog, dir, flags, co, hash, V.w2 ? &V.p2 : nullptr, V.acc ? &V.pa : nullptr, "fdg_isa_eval",
                              V.rm_bufs ? &V.pr : nullptr, V.rm_bufs, coop, V.rm_acc ? &V.pra : nullptr, pool, V.rl ? &V.prl : nullptr,
                              V.rl && V.rl_acc ? &V.prla : nullptr);
  if (rc) return rc;
  install_isa(g, prog, co, hash, flags, V.w2 ? &V.p2 : nullptr, V.acc ? &V.pa : nullptr, V.rm_bufs ? &V.pr : nullptr, V.rm_bufs, coop,
              V.rm_acc ? &V.pra : nullptr, pool, V.rl ? &V.prl : nullptr, V.rl && V.rl_acc ? &V.prla : nullptr);
  return FDG_OK;
}

// returns 1 when a remembered choice was installed, 0 when there is none, < 0 on error
static int use_tuned(fdg_graph *g, const std::string &dir, unsigned flags) {
  const std::string tuned = tuned_path(g, dir);
  std::vector<char> buf;
  if (!read_cached(dir, tuned.substr(dir.size() + 1), buf)) return 0;
  fdg::OptParams q;
  int tuned_coop = -1;
  buf.push_back(0);
  {
    fdg_opt_params r = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // seven numbers, or nine: ... recompute window and cost (files written before round 2 have seven)
    int coop_flag = -1;          // tenth number: the tuner's verdict on the cooperative variant (absent or -1: the static criterion decides)
    unsigned n_land = 0;         // eleventh: AGPR landing slots of the leaf loads (one-wave configuration)
    if (std::sscanf(buf.data(), "%u %u %u %u %u %u %u %u %u %d %u", &r.n_reg, &r.n_lds, &r.n_acc, &r.lookahead_lds, &r.lookahead_mem, &r.lookahead_leaf, &r.vn_window,
                    &r.remat_window, &r.remat_cost, &coop_flag, &n_land) < 7) return 0;
    tuned_coop = coop_flag == 1 ? 4 : coop_flag;       // (files of the first cooperative version wrote 1 for four waves)
    if (r.n_reg < 4) return 0;
    q = to_params(&r);          // the same clamps as parameters handed over through the ABI
    if (!r.n_acc) q.n_acc = 0;
    if (!r.n_lds) q.n_lds = 0;
    q.vn_window = r.vn_window;     // the tuner's own encoding: 0 = value numbering without a window (through the ABI 0 asks for the default)
    q.n_land = q.n_acc ? std::min<uint32_t>(n_land, q.n_acc / 2) : 0;
  }
  apply_land_env(q);
  fdg::OptProgram prog;
  build_prog(g, q, prog);
  if (!prog.supported) return 0;
  IsaVariants V;
  build_variants(g, q, true, V);
  V.coop_verdict = tuned_coop;
  const int rc = assemble_and_install(g, prog, dir, flags, V);
  return rc ? rc : 1;
}

static int autotune_isa(fdg_graph *g, const std::string &dir, unsigned flags) {
  int rc = ensure_device(g);
  if (rc) return rc;
  rc = fdg_bind_stream_ws(g, nullptr);
  if (rc) return rc;
  const fdg::Lowered &p = g->prog;
  const std::string tuned = tuned_path(g, dir);
  auto to_line = [](const fdg::OptParams &q) {
    char b[200];
    std::snprintf(b, sizeof b, "%u %u %u %u %u %u %u %u %u", q.n_reg, q.n_lds, q.n_acc, q.lookahead_lds, q.lookahead_mem, q.lookahead_leaf, q.vn_window,
                  q.remat_window, q.remat_cost);
    return std::string(b);
  };
  size_t n_first_stage = 0;
  std::vector<fdg::OptParams> cand;
  cand.push_back(cfg_S());
  cand.push_back(cfg_A());
  { fdg::OptParams q = cfg_A(); q.vn_window = 1000; cand.push_back(q); }
  { fdg::OptParams q = cfg_A(); q.vn_window = 60; cand.push_back(q); }
  cand.push_back(cfg_B());
  { fdg::OptParams q = cfg_B(); q.vn_window = 0; cand.push_back(q); }
  { fdg::OptParams q = cfg_B(); q.vn_window = 200; cand.push_back(q); }
  { fdg::OptParams q = cfg_B(); q.vn_window = 400; q.lookahead_leaf = 300; q.lookahead_mem = 128; cand.push_back(q); }
  { fdg::OptParams q = cfg_A(); q.vn_window = 300; cand.push_back(q); }
  { fdg::OptParams q = cfg_A(); q.vn_window = 2000; cand.push_back(q); }
  for (uint32_t w : {2000u, 4000u}) { fdg::OptParams q = cfg_B(); q.vn_window = w; cand.push_back(q); }
  for (uint32_t w : {0u, 2000u}) { fdg::OptParams q = cfg_B(); q.vn_window = w; q.lookahead_leaf = 300; q.lookahead_mem = 128; cand.push_back(q); }
  { fdg::OptParams q = cfg_A(); q.n_reg = 80; q.n_lds = 26; cand.push_back(q); }     // three waves per SIMD
  { fdg::OptParams q = cfg_A(); q.n_reg = 56; q.n_lds = 20; cand.push_back(q); }     // four waves per SIMD
  // programs that spill to the HBM panel: forget-and-recompute of cheap nodes (arithmetic instead of panel traffic)
  {
    fdg::OptProgram pb;
    build_prog(g, cfg_B(), pb);
    if (pb.supported && (pb.n_ld_mem + pb.n_st_mem) * 20 > pb.n_valu) {
      for (uint32_t w : {500u, 1000u, 2000u}) for (uint32_t c : {3u, 6u}) { fdg::OptParams q = cfg_B(); q.vn_window = 200; q.remat_window = w; q.remat_cost = c; cand.push_back(q); }
      { fdg::OptParams q = cfg_B(); q.vn_window = 60; cand.push_back(q); }
    }
  }
  // batch: at least two tiles per resident wave, and enough bytes (about 0.4 GB of leaves) that a run
  // is not dominated by launch overhead on tiny graphs
  // (round 3: ten times the bytes -- the ridge graphs run against the chip's power budget, and a launch of 0.1 ms is over
  // before the clocks have followed the load)
  long Bt = std::max<long>((long)g->n_cu * 8 * 64 * 16, (long)(4e9 / (8.0 * std::max<uint32_t>(p.L + p.R, 1))));
  Bt = std::min<long>((Bt + 63) & ~63l, 1l << 24);
  double *d_leaf = nullptr, *d_root = nullptr;
  HIP_TRY(hipMalloc(&d_leaf, std::max<size_t>(1, (size_t)Bt * p.L) * 8));
  if (hipMalloc(&d_root, std::max<size_t>(1, (size_t)Bt * p.R) * 8) != hipSuccess) { hipFree(d_leaf); set_error("hipMalloc failed"); return FDG_E_NOMEM; }
  // the batch the candidates are timed on is TILE-MAJOR (round 4: the layout a driver that owns its batch should allocate, bench.py's default;
  // the larger graphs gain 12-14 % over leaf-major on it and do not always prefer the same configuration); FDG_TUNE_LEAF_MAJOR=1: as before
  const bool tune_tiled = fdg::knob("FDG_TUNE_LEAF_MAJOR") == nullptr;
  if (p.L) {
    if (tune_tiled) hipLaunchKernelGGL(fdg_fill_uniform_tiled, dim3(4096), dim3(256), 0, 0, d_leaf, Bt, p.L, 1L, 64L, 64L * (long)p.L, (uint64_t)1234, (uint64_t)0);
    else hipLaunchKernelGGL(fdg_fill_uniform, dim3(4096), dim3(256), 0, 0, d_leaf, Bt, p.L, 1L, Bt, (uint64_t)1234, (uint64_t)0, 0);
  }
  auto tune_run = [&]() -> int {
    return tune_tiled ? fdg_run_locked(g, 0, d_leaf, 1, 64, d_root, 1, 64, nullptr, nullptr, Bt, nullptr, 64L * (long)p.L, 64L * (long)p.R)
                      : fdg_run_locked(g, 0, d_leaf, 1, Bt, d_root, p.R, 1, nullptr, nullptr, Bt, nullptr);
  };
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  double best_ms = 1e300;
  int best = -1;
  std::string seen;
  n_first_stage = cand.size();
  for (size_t c = 0; ; ++c) {
    if (c == cand.size()) {
      // second stage (FDG_TUNE_LAND=1 only: measured neutral to -5 % on every one-wave kernel, profiles/r03_log_agpr_landing.txt):
      // the best one-wave configuration once more with AGPR landing slots for its leaf loads
      if (!fdg::knob("FDG_TUNE_LAND") || cand.size() != n_first_stage || best < 0 || cand[best].n_acc < 64) break;
      for (uint32_t nl : {16u, 32u, 48u}) { fdg::OptParams q = cand[best]; q.n_land = nl; cand.push_back(q); }
    }
    fdg::OptProgram prog;
    build_prog(g, cand[c], prog);
    if (!prog.supported) continue;
    if (c == 0 && !(prog.n_ld_leaf <= p.n_live_leaf && prog.n_ld_lds + prog.n_st_lds + prog.n_ld_mem + prog.n_st_mem == 0)) continue;
    std::vector<char> co; std::string hash;
    if (assemble_isa(g, prog, dir, flags, co, hash) != FDG_OK) continue;
    if (seen.find(hash) != std::string::npos) continue;
    seen += hash + ";";
    install_isa(g, prog, co, hash, flags);
    // warm-up: the first launches after a change of load run at transient clocks (power management settles
    // within a few tens of milliseconds); candidates are compared in the settled state
    bool ok_run = true;
    for (int w = 0; w < 12 && ok_run; ++w) ok_run = tune_run() == FDG_OK;
    if (!ok_run) continue;
    // the sustained rate: eight launches back to back under one pair of events, twice, the better of the two
    float ms_min = 1e30f;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0, nullptr);
      for (int k = 0; k < 8 && ok_run; ++k) ok_run = tune_run() == FDG_OK;
      hipEventRecord(e1, nullptr);
      hipEventSynchronize(e1);
      if (!ok_run) { ms_min = 1e30f; break; }
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      ms_min = std::min(ms_min, ms);
    }
    if (fdg::knob("FDG_TUNE_VERBOSE")) std::fprintf(stderr, "[tune] %s land %u: %.3f ms per 8 launches of %ld (%llu ops, %llu leaf loads, %llu panel)\n", to_line(cand[c]).c_str(), cand[c].n_land, ms_min, Bt,
                                                      (unsigned long long)prog.n_valu, (unsigned long long)prog.n_ld_leaf, (unsigned long long)(prog.n_ld_mem + prog.n_st_mem));
    if (ms_min < best_ms) { best_ms = ms_min; best = (int)c; }
  }
  if (best < 0) { hipEventDestroy(e0); hipEventDestroy(e1); hipFree(d_leaf); hipFree(d_root); set_error("autotune: no candidate configuration ran"); return FDG_E_JIT; }
  fdg::OptProgram prog;
  build_prog(g, cand[best], prog);
  IsaVariants V;
  build_variants(g, cand[best], true, V);
  rc = assemble_and_install(g, prog, dir, flags, V);
  // the cooperative variant, when the graph gets one, with eight and with four waves against the best one-wave kernel
  int coop_best = 0;
  const bool had_coop = rc == FDG_OK && g->has_coop;
  if (had_coop) {
    auto time_it = [&]() {
      float tm = 1e30f;
      bool ok_run = true;
      for (int w = 0; w < 6 && ok_run; ++w) ok_run = tune_run() == FDG_OK;
      for (int rep = 0; rep < 5 && ok_run; ++rep) {
        hipEventRecord(e0, nullptr);
        ok_run = tune_run() == FDG_OK;
        hipEventRecord(e1, nullptr);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (ok_run) tm = std::min(tm, ms);
      }
      return tm;
    };
    g->coop_enabled = false;
    float t_best = time_it();
    for (int nw : {8, 4}) {
      IsaVariants Vw;
      build_variants(g, cand[best], true, Vw);
      Vw.coop_verdict = nw;
      if (assemble_and_install(g, prog, dir, flags, Vw) != FDG_OK || !g->has_coop) continue;
      g->coop_enabled = true;
      const float tw = time_it();
      if (tw < t_best) { t_best = tw; coop_best = nw; }
    }
    IsaVariants Vf;
    build_variants(g, cand[best], true, Vf);
    Vf.coop_verdict = coop_best;
    rc = assemble_and_install(g, prog, dir, flags, Vf);
    g->coop_enabled = g->has_coop;
  }
  hipEventDestroy(e0); hipEventDestroy(e1);
  hipFree(d_leaf); hipFree(d_root);
  if (rc) return rc;
  const std::string line = to_line(cand[best]) + " " + std::to_string(had_coop ? coop_best : -1) + " " + std::to_string(cand[best].n_land) + "\n";
  write_file(tuned, line.c_str(), line.size());
  return FDG_OK;
}

static int specialize_isa(fdg_graph *g, const std::string &dir, unsigned flags) {
  if (!has_opt_params(g)) {
    if (!fdg::knob("FDG_IGNORE_TUNED")) {          // (with the switch and FDG_SPEC_AUTOTUNE: tune again, overwriting the remembered choice)
      const int t = use_tuned(g, dir, flags);        // a remembered on-device choice wins (no device needed to use it)
      if (t != 0) return t < 0 ? t : FDG_OK;
    }
    if (flags & FDG_SPEC_AUTOTUNE) return autotune_isa(g, dir, flags);
  }
  fdg::OptProgram prog;
  fdg::OptParams chosen;
  if (has_opt_params(g)) {
    const fdg_opt_params q = get_opt_params(g);
    chosen = to_params(&q);
    apply_land_env(chosen);
    build_prog(g, chosen, prog);
  } else {
    chosen = auto_program(g, prog);
  }
  if (!prog.supported) { set_error("optimizing back end does not cover this graph: " + prog.why); return FDG_E_UNSUPPORTED; }
  IsaVariants V;
  build_variants(g, chosen, !has_opt_params(g), V);
  return assemble_and_install(g, prog, dir, flags, V);
}

}  // extern "C"

// ---------------------------------------------------------------------------
// Monte-Carlo step as one kernel of the optimizing back end (route 3 of fdg_graph_specialize_fused): the
// program of fdg::build_mc_program, whose inputs are the n_loop*dim momentum components and n_tau times of a
// sample (168 bytes for the 4-loop self-energy) instead of its L leaf values (1240 bytes).  kF, beta, lambda
// reach the formulas as kernel arguments (-kF^2, beta, -beta, lambda in SGPR pairs): one code object per (graph, tables).
// ---------------------------------------------------------------------------
static fdg_leaf_tables handle_tables(const fdg_graph *g, double kF, double beta, double lambda) {
  fdg_leaf_tables tab;
  tab.n_leaf = g->lt_hdr[0]; tab.n_basis = g->lt_hdr[1]; tab.n_loop = g->lt_hdr[2]; tab.dim = g->lt_hdr[3]; tab.n_tau = g->lt_hdr[4];
  tab.leaf_type = g->lt_i32[0].data(); tab.leaf_order = g->lt_i32[1].data(); tab.tau_in = g->lt_i32[2].data();
  tab.tau_out = g->lt_i32[3].data(); tab.loop_index = g->lt_i32[4].data(); tab.basis = g->lt_basis.data();
  tab.kF = kF; tab.beta = beta; tab.lambda = lambda;
  return tab;
}

static fdg::OptParams mc_params(const fdg_graph *g) {
  fdg::OptParams q = cfg_A();
  if (has_opt_params(g)) { const fdg_opt_params o = get_opt_params(g); q = to_params(&o); }
  q.n_reg = std::min<uint32_t>(q.n_reg, 123);      // two temporaries (4 VGPRs) above the values
  q.fma = q.fma || g->isa_fma;
  return q;
}

bool fdg_mc_isa_supported(fdg_graph *g, const fdg_leaf_tables *tab, std::string &why, bool *recommended) {
  fdg::LeafSpec ls; ls.tab = tab; ls.kF = 1.0; ls.beta = 1.0; ls.lambda = 1.0;
  fdg::OptProgram prog;
  fdg::build_mc_program(g->prog, ls, mc_params(g), prog);
  why = prog.why;
  // Very large graphs keep leaf kernel + evaluator: a leaf that is a computed value must be spilled where a leaf that is
  // input is simply read again, and beyond the on-chip levels that traffic outweighs the leaf matrix it saves
  // (measured: 5-loop self-energy, 13 000 ops, one kernel 1.67x faster; its Taylor expansion, 479 leaves and 79 000 ops, 0.9x;
  // the 4-loop vertex function of example/benchmark.jl, 984 leaves and 65 000 ops, 1.68x).  The other route writes and reads
  // 16 bytes per leaf and sample, which is worth some tens of fold steps: the budget grows with the number of leaves.
  if (fdg::knob("FDG_MC_DEBUG"))
    std::fprintf(stderr, "[mc] one-kernel program: valu %llu ld_leaf %llu ld_mem %llu st_mem %llu ld_lds %llu st_lds %llu\n", (unsigned long long)prog.n_valu,
                 (unsigned long long)prog.n_ld_leaf, (unsigned long long)prog.n_ld_mem, (unsigned long long)prog.n_st_mem, (unsigned long long)prog.n_ld_lds, (unsigned long long)prog.n_st_lds);
  if (recommended) *recommended = prog.supported && prog.n_valu <= 40000 + 30ull * g->prog.L;
  return prog.supported;
}

static uint32_t isa_waves_per_cu(uint32_t vgpr, uint32_t lds_bytes) {
  const uint32_t valloc = std::max<uint32_t>(8, (vgpr + 7) & ~7u);
  uint32_t per_cu = std::min<uint32_t>(8, 512 / valloc) * 4;
  if (lds_bytes) per_cu = std::min<uint32_t>(per_cu, (160u * 1024u) / lds_bytes);
  return std::max<uint32_t>(1, std::min<uint32_t>(per_cu, 32));
}

int fdg_mc_isa_build(fdg_graph *g) {
  if (g->mc_built) return FDG_OK;
  // (the parameter values only fill the ops' imm fields; the kernel reads them from its arguments)
  const fdg_leaf_tables tab = handle_tables(g, 1.0, 1.0, 1.0);
  fdg::LeafSpec ls; ls.tab = &tab; ls.kF = 1.0; ls.beta = 1.0; ls.lambda = 1.0;
  fdg::OptParams q = mc_params(g);
  fdg::OptProgram pe, pa;
  fdg::build_mc_program(g->prog, ls, q, pe);
  if (pe.supported && !has_opt_params(g) && (pe.n_ld_mem + pe.n_st_mem) * 100 > pe.n_valu) {
    // values computed from (K, T) cannot be re-read from the input like leaves: what does not fit on chip goes through
    // the HBM panel (16 bytes per spill).  One wave per SIMD with the AGPR level keeps it all on chip (measured on the
    // Taylor-expanded 4-loop self-energy: 2.1e9 -> 2.8e9 samples/s)
    fdg::OptParams qb = cfg_B();
    qb.n_reg = std::min<uint32_t>(qb.n_reg, 123);
    qb.fma = q.fma;
    // ... and when even that overflows, a shorter value-numbering window trades re-computed fold steps for live values:
    // the candidate with the least (instructions + 40 per panel access) wins -- 8 bytes per sample at the panel's
    // ~4.7 TB/s cost what ~40 instructions do at 27e12 lane-op/s (5-loop self-energy: window 1000 -> 400, 7.8e8 -> 9.6e8)
    fdg::OptProgram best;
    uint64_t best_cost = ~0ull;
    for (uint32_t vn : {1000u, 400u, 200u}) {
      qb.vn_window = vn;
      fdg::OptProgram pb;
      fdg::build_mc_program(g->prog, ls, qb, pb);
      if (!pb.supported) break;
      const uint64_t cost = pb.n_valu + 40 * (pb.n_ld_mem + pb.n_st_mem);
      if (cost < best_cost) { best_cost = cost; best = std::move(pb); q = qb; }
      if (best.n_ld_mem + best.n_st_mem == 0) break;
    }
    if (best_cost != ~0ull) pe = std::move(best);
  }
  if (!pe.supported) { set_error("the fused ISA step does not cover this graph / these leaves: " + pe.why); return FDG_E_UNSUPPORTED; }
  const uint32_t R = g->prog.R;
  bool has_acc = R >= 1 && R <= 40 && q.n_reg >= R + 2 + 8 && (R <= 16 || q.n_reg >= R + 2 + 64);
  if (has_acc) {
    fdg::OptParams qa = q;
    qa.reserve_pairs = R + 2;        // (build_mc_program takes the macro ops' temporaries off the value budget itself)
    fdg::build_mc_program(g->prog, ls, qa, pa);
    has_acc = pa.supported;
  }
  std::vector<char> co;
  std::string hash;
  const int rc = assemble_isa(g, pe, g->mc_dir, g->mc_flags, co, hash, nullptr, has_acc ? &pa : nullptr, "fdg_isa_mc");
  if (rc) return rc;
  if (g->mc_module) { HIP_TRY(hipDeviceSynchronize()); hipModuleUnload((hipModule_t)g->mc_module); g->mc_module = nullptr; }
  g->fn_mc = g->fn_mc_acc = nullptr;
  g->mc_code.swap(co);
  g->mc_has_acc = has_acc;
  auto tmp_vgprs = [](const fdg::OptProgram &q) { uint32_t t = 0; for (const fdg::MOp &o : q.ops) t = std::max(t, fdg::mop_tmp_pairs(o.kind)); return 2 * t; };
  g->mc_vgpr[0] = ((6 + 2 * std::max<uint32_t>(pe.n_reg_used, 1) + tmp_vgprs(pe) + 3) & ~3u) + 2 * pe.n_acc_used;
  g->mc_lds[0] = pe.n_lds_used * 512u; g->mc_mem[0] = pe.n_mem_used;
  if (has_acc) {
    g->mc_vgpr[1] = ((6 + 2 * std::max<uint32_t>(pa.n_reg_used, 1) + 2 * (R + 2) + tmp_vgprs(pa) + 3) & ~3u) + 2 * pa.n_acc_used;
    g->mc_lds[1] = pa.n_lds_used * 512u; g->mc_mem[1] = pa.n_mem_used;
  }
  g->mc_built = true;
  return FDG_OK;
}

int fdg_mc_isa_run(fdg_graph *g, int mode, const double *d_K, int64_t ks, int64_t kc, const double *d_T, int64_t ts, int64_t tc,
                   double kF, double beta, double lambda, double *d_root, int64_t rs, int64_t rk, const double *d_weight,
                   double *d_acc, int64_t B, hipStream_t st) {
  int rc = fdg_mc_isa_build(g);
  if (rc) return rc;
  if (!g->mc_module) {
    hipModule_t m; hipFunction_t f;
    hipError_t e = hipModuleLoadData(&m, g->mc_code.data());
    if (e != hipSuccess) { set_error("hipModuleLoadData failed: " + std::string(hipGetErrorString(e))); return FDG_E_JIT; }
    HIP_TRY(hipModuleGetFunction(&f, m, "fdg_isa_mc"));
    g->mc_module = m; g->fn_mc = f;
    if (g->mc_has_acc) { HIP_TRY(hipModuleGetFunction(&f, m, "fdg_isa_mc_acc")); g->fn_mc_acc = f; }
  }
  const uint32_t R = g->prog.R;
  const uint32_t n_k = g->lt_hdr[2] * g->lt_hdr[3], n_tau = g->lt_hdr[4], n_in = n_k + n_tau;
  const bool use_acc = mode == 1 && g->mc_has_acc;
  if (mode == 1 && !use_acc) {
    // more than 16 roots: no room for the accumulators next to the values -- roots to a scratch matrix, then the
    // deterministic weighted reduction the other back ends use
    if (R == 0) return FDG_OK;
    const int64_t ld = (B + 15) & ~(int64_t)15;                  // column-major scratch (see fdg_weighted_partials)
    const size_t need = (size_t)ld * R * sizeof(double) + (size_t)2048 * R * sizeof(double);
    if (g->ws2_bytes < need) {
      if (g->d_ws2) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(g->d_ws2)); g->d_ws2 = nullptr; g->ws2_bytes = 0; }
      if (hipMalloc(&g->d_ws2, need) != hipSuccess) { set_error("hipMalloc(root scratch) failed"); return FDG_E_NOMEM; }
      g->ws2_bytes = need;
    }
    double *roots = (double *)g->d_ws2, *partial = roots + (size_t)ld * R;
    rc = fdg_mc_isa_run(g, 0, d_K, ks, kc, d_T, ts, tc, kF, beta, lambda, roots, 1, ld, nullptr, nullptr, B, st);
    if (rc) return rc;
    const uint32_t pb = weighted_segments((long)B, R);
    const uint8_t *live = nullptr;
    rc = root_live_mask(g, &live);
    if (rc) return rc;
    hipLaunchKernelGGL(fdg_weighted_partials, dim3(pb * R), dim3(256), 0, st, roots, (long)ld, d_weight, (long)B, R, pb, partial, live);
    hipLaunchKernelGGL(fdg_reduce_partials, dim3(std::min<uint32_t>(R, 64u)), dim3(256), 0, st, partial, pb, R, d_acc);
    HIP_TRY(hipGetLastError());
    return FDG_OK;
  }
  if (mode == 0 && (rs < 0 || rs >= (1ll << 23))) { set_error("root sample stride negative or of 2^23 elements or more is not supported by the ISA kernel"); return FDG_E_UNSUPPORTED; }
  const int v = use_acc ? 1 : 0;
  const long grid = (long)g->n_cu * isa_waves_per_cu(g->mc_vgpr[v], g->mc_lds[v]);
  const size_t panel = (((size_t)std::max<uint32_t>(g->mc_mem[v], 1) * 512u * (size_t)grid) + 4095) & ~(size_t)4095;
  rc = ensure_ws(g, panel + (size_t)grid * std::max<uint32_t>(R, 1) * 512u + 4096);
  if (rc) return rc;
  // The kernel reads its inputs as columns: momentum component c at K[b*ss + c*kc], time i at T[b*ss + i*tc] (two
  // bases, two column strides, one sample stride).  Component-major K and T (sample stride 1: a wave's 64 samples of a
  // column are one 512-byte access) are read in place; anything else is packed into such a pair owned by the handle
  // first (8 (n_k + n_tau) bytes per sample each way).
  const bool in_place = ks == 1 && ts == 1;
  int64_t Bc = std::min<int64_t>((B + 63) & ~63ll, 1ll << 22);
  if (g->cfg.mc_chunk >= 64) Bc = std::min<int64_t>((g->cfg.mc_chunk + 63) & ~63ll, (B + 63) & ~63ll);
  if (!in_place) {
    const size_t need = (size_t)Bc * n_in * sizeof(double);
    if (g->ws4_bytes < need) {
      if (g->d_ws4) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(g->d_ws4)); g->d_ws4 = nullptr; g->ws4_bytes = 0; }
      if (hipMalloc(&g->d_ws4, need) != hipSuccess) { set_error("hipMalloc(packed inputs) failed"); return FDG_E_NOMEM; }
      g->ws4_bytes = need;
    }
  }
  auto pack = [&](const double *src, int64_t ss, int64_t cs, uint32_t ncol, double *dst, int64_t n) -> int {
    if (ss == 1 && cs >= n) {
      HIP_TRY(hipMemcpy2DAsync(dst, (size_t)Bc * 8, src, (size_t)cs * 8, (size_t)n * 8, ncol, hipMemcpyDeviceToDevice, st));
    } else {
      const long ntile = ((n + 63) / 64) * ((ncol + 31) / 32);
      hipLaunchKernelGGL(fdg_transpose_to_leaf_major, dim3((unsigned)std::min<long>(ntile, (long)g->n_cu * 16)), dim3(256), 0, st,
                         src, (long)ss, (long)cs, dst, (long)Bc, (long)n, ncol);
      HIP_TRY(hipGetLastError());
    }
    return FDG_OK;
  };
  for (int64_t c0 = 0; c0 < B; c0 += (in_place ? B : Bc)) {
    long n = (long)(in_place ? B : std::min<int64_t>(Bc, B - c0));
    const double *x = d_K, *x2 = d_T;
    long xls = (long)kc, xls2 = (long)tc, xss = (long)ks;
    if (!in_place) {
      double *X = (double *)g->d_ws4;
      rc = pack(d_K + c0 * ks, ks, kc, n_k, X, n);
      if (rc) return rc;
      rc = pack(d_T + c0 * ts, ts, tc, n_tau, X + (size_t)n_k * Bc, n);
      if (rc) return rc;
      x = X; x2 = X + (size_t)n_k * Bc; xls = xls2 = (long)Bc; xss = 1;
    }
    void *a_wsp = g->d_ws;
    long nwg = std::min<long>((n + 63) / 64, grid), zero = 0, tls = 64 * xss;
    double p_nkf2 = -(kF * kF), p_beta = beta, p_nbeta = -beta, p_lambda = lambda;     // MOp::param 1..4
    if (use_acc) {
      double *part = (double *)((char *)g->d_ws + panel);
      const double *wt = d_weight ? d_weight + c0 : nullptr;
      void *args[] = {(void *)&x, &xss, &xls, (void *)&part, &zero, &zero, &a_wsp, &n, &nwg, (void *)&wt, (void *)&x2, &xls2, &p_nkf2, &p_beta, &p_nbeta, &p_lambda, &tls, &zero};
      HIP_TRY(hipModuleLaunchKernel((hipFunction_t)g->fn_mc_acc, (unsigned)nwg, 1, 1, 64, 1, 1, 0, st, args, nullptr));
      hipLaunchKernelGGL(fdg_reduce_lane_partials, dim3(std::min<uint32_t>(R, 64u)), dim3(256), 0, st, part, (uint32_t)nwg, R, d_acc);
      HIP_TRY(hipGetLastError());
    } else {
      double *rt = d_root + c0 * rs;
      long a_rs = (long)rs, a_rk = (long)rk, trs = 64 * a_rs;
      const double *nowt = nullptr;
      void *args[] = {(void *)&x, &xss, &xls, (void *)&rt, &a_rs, &a_rk, &a_wsp, &n, &nwg, (void *)&nowt, (void *)&x2, &xls2, &p_nkf2, &p_beta, &p_nbeta, &p_lambda, &tls, &trs};
      HIP_TRY(hipModuleLaunchKernel((hipFunction_t)g->fn_mc, (unsigned)nwg, 1, 1, 64, 1, 1, 0, st, args, nullptr));
    }
  }
  return FDG_OK;
}

extern "C" {

int fdg_graph_specialize(fdg_graph *g, const char *cache_dir, unsigned flags) {
  if (!g) { set_error("null handle"); return FDG_E_INVALID; }
  std::lock_guard<std::mutex> lk(g->mu);
  fdg::KnobScope knob_scope(&g->knobs);
  if (flags & FDG_SPEC_ISA) {
    g->isa_fma = (flags & FDG_SPEC_FAST_MATH) != 0;
    std::string dir0;
    { const int rcd = fdg_cache_dir(cache_dir, dir0); if (rcd) return rcd; }
    return specialize_isa(g, dir0, flags);
  }
  const bool companion = (flags & FDG_SPEC_ROW_MAJOR_COMPANION) != 0;
  if (companion && !(g->isa && !g->code_object.empty())) { set_error("FDG_SPEC_ROW_MAJOR_COMPANION needs a handle already specialised with FDG_SPEC_ISA"); return FDG_E_INVALID; }
  const bool fast = (flags & FDG_SPEC_FAST_MATH) != 0;
  const std::string src = emit_hip_source(g->prog, flags);
  char hbuf[40];
  std::snprintf(hbuf, sizeof hbuf, "%016llx", (unsigned long long)fnv1a(src, fnv1a(fast ? "fast" : "strict")));
  std::string dir;
  { const int rcd = fdg_cache_dir(cache_dir, dir); if (rcd) return rcd; }
  const std::string base = dir + "/fdg_" + hbuf;
  std::vector<char> co;
  if (!read_cached(dir, std::string("fdg_") + hbuf + ".hsaco", co)) {
    std::string log;
    const char *force = fdg::knob("FDG_JIT");  // "hipcc" forces the subprocess path
    int rc = -1;
    if (!(force && std::strcmp(force, "hipcc") == 0)) rc = compile_hiprtc(src, fast, co, log);
    if (rc != 0) {
      std::string log2;
      if (!write_file(base + ".hip", src.c_str(), src.size())) { set_error("cannot write " + base + ".hip"); return FDG_E_JIT; }
      rc = compile_hipcc(base + ".hip", base + ".hsaco", fast, log2);
      if (!(flags & FDG_SPEC_KEEP_SOURCE)) std::remove((base + ".hip").c_str());
      if (rc != 0 || !read_file(base + ".hsaco", co)) {
        set_error("kernel specialization failed.\nhiprtc: " + log + "\nhipcc: " + log2);
        return FDG_E_JIT;
      }
    } else {
      write_file(base + ".hsaco", co.data(), co.size());
    }
  }
  if ((flags & FDG_SPEC_KEEP_SOURCE)) write_file(base + ".hip", src.c_str(), src.size());
  if (companion) {
    if (g->alt_module) { hipModuleUnload((hipModule_t)g->alt_module); g->alt_module = nullptr; }
    g->alt_code.swap(co);
    return FDG_OK;
  }
  // the handle changes back end only now that the new code object exists (a failed JIT leaves it as it was)
  g->alt_code.clear();
  if (g->alt_module) { hipModuleUnload((hipModule_t)g->alt_module); g->alt_module = nullptr; }
  if (g->module) { hipModuleUnload((hipModule_t)g->module); g->module = nullptr; }
  g->isa = false;
  g->fn_isa = g->fn_isa_w2 = g->fn_isa_acc = g->fn_isa_nt = g->fn_isa_acc_nt = nullptr;
  g->fn_eval_sm = g->fn_eval_gen = nullptr;
  g->has_w2 = g->has_acc = false;
  g->code_object.swap(co);
  g->spec_source_hash = hbuf;
  g->spec_flags = flags;
  return FDG_OK;
}

// (the body: fdg_graph_specialize_typed calls it with the handle's mutex held and the handle's options in scope)
static int create_complex_view_scoped(const fdg_graph *g, fdg_graph **out);
int fdg_graph_create_complex_view(const fdg_graph *g, fdg_graph **out) {
  const fdg::KnobMap knobs_now = knobs_snapshot(g);      // (a copy taken under the handle's mutex: fdg_graph_set_option may run on another thread)
  fdg::KnobScope knob_scope(g ? &knobs_now : nullptr);
  return create_complex_view_scoped(g, out);
}
static int create_complex_view_scoped(const fdg_graph *g, fdg_graph **out) {
  if (!g || !out) { set_error("null argument"); return FDG_E_INVALID; }
  *out = nullptr;
  fdg::RealTwinTable t;
  std::string why;
  if (!fdg::complex_to_real_table(g->prog, t, why)) { set_error(why); return FDG_E_UNSUPPORTED; }
  fdg_graph_desc d;
  d.n_leaf = t.n_leaf; d.n_node = (uint32_t)t.op.size(); d.n_root = (uint32_t)t.root_slot.size(); d.n_edge = (uint32_t)t.idx.size();
  static const uint8_t no_op = 0; static const int32_t no_pw = 0; static const uint32_t no_u = 0; static const double no_f = 0;
  d.op = t.op.empty() ? &no_op : t.op.data(); d.power = t.power.empty() ? &no_pw : t.power.data();
  d.child_off = t.off.data(); d.child_idx = t.idx.empty() ? &no_u : t.idx.data(); d.child_fac = t.fac.empty() ? &no_f : t.fac.data();
  d.root_slot = t.root_slot.empty() ? &no_u : t.root_slot.data();
  const int rc_twin = fdg_graph_create(&d, out);
  if (!rc_twin && *out) { (*out)->knobs = g->knobs; parse_launch_cfg(*out); }     // the view inherits the handle's options
  return rc_twin;
}

// Element types other than Float64: one HIP-source kernel per (graph, type), JIT-compiled like the Float64 HIP-source kernels
// (hiprtc, the hipcc subprocess as the second route), cached under the same rules.
int fdg_graph_specialize_typed(fdg_graph *g, int dtype, const char *cache_dir, unsigned flags) {
  if (!g) { set_error("null handle"); return FDG_E_INVALID; }
  if (dtype == FDG_DT_F64) return FDG_OK;                       // the handle's ordinary kernels
  if (dtype < 0 || dtype > FDG_DT_C32) { set_error("unknown element type"); return FDG_E_INVALID; }
  std::lock_guard<std::mutex> lk(g->mu);
  fdg::KnobScope knob_scope(&g->knobs);
  if (dtype == FDG_DT_C64 && (flags & FDG_SPEC_ISA) && !g->cx_twin_tried) {
    // ComplexF64 rows (a row of a row-major [B, L] matrix is 2 L doubles re, im, ...): the graph spelled out on real and imaginary
    // parts is an ordinary Float64 graph; when the optimizing back end gives it the in-place row-major variant, such rows take that
    // route in fdg_eval_device_typed (parquet_sigma4: 2.2 -> 2.7e9 evals/s, GV 4-loop: 1.6 -> 2.6e9).  Any failure here just
    // leaves the per-type kernel below in charge.
    g->cx_twin_tried = true;
    fdg_graph *tw = nullptr;
    if (create_complex_view_scoped(g, &tw) == FDG_OK && tw) {
      fdg_kernel_info ki;
      if (fdg_graph_specialize(tw, cache_dir, FDG_SPEC_ISA) == FDG_OK && fdg_graph_kernel_info(tw, &ki) == FDG_OK && ki.has_rm) g->cx_twin = tw;
      else fdg_graph_destroy(tw);
    }
  }
  flags &= ~(unsigned)FDG_SPEC_ISA;
  if (!g->typed_code[dtype].empty() && !(flags & FDG_SPEC_KEEP_SOURCE)) return FDG_OK;      // already there (the graph of a handle never changes)
  bool ok = true; std::string why;
  const std::string src = emit_hip_source_typed(g->prog, dtype, ok, why);
  if (!ok) { set_error(why); return FDG_E_UNSUPPORTED; }
  char hbuf[40];
  std::snprintf(hbuf, sizeof hbuf, "%016llx", (unsigned long long)fnv1a(src, fnv1a("typed-strict")));
  std::string dir;
  { const int rcd = fdg_cache_dir(cache_dir, dir); if (rcd) return rcd; }
  const std::string base = dir + "/fdg_" + hbuf;
  std::vector<char> co;
  if (!read_cached(dir, std::string("fdg_") + hbuf + ".hsaco", co)) {
    std::string log;
    const char *force = fdg::knob("FDG_JIT");
    int rc = -1;
    if (!(force && std::strcmp(force, "hipcc") == 0)) rc = compile_hiprtc(src, false, co, log);
    if (rc != 0) {
      std::string log2;
      if (!write_file(base + ".hip", src.c_str(), src.size())) { set_error("cannot write " + base + ".hip"); return FDG_E_JIT; }
      rc = compile_hipcc(base + ".hip", base + ".hsaco", false, log2);
      if (!(flags & FDG_SPEC_KEEP_SOURCE)) std::remove((base + ".hip").c_str());
      if (rc != 0 || !read_file(base + ".hsaco", co)) { set_error("kernel specialization failed.\nhiprtc: " + log + "\nhipcc: " + log2); return FDG_E_JIT; }
    } else {
      write_file(base + ".hsaco", co.data(), co.size());
    }
  }
  if (flags & FDG_SPEC_KEEP_SOURCE) write_file(base + ".hip", src.c_str(), src.size());
  if (g->typed_module[dtype]) { hipModuleUnload((hipModule_t)g->typed_module[dtype]); g->typed_module[dtype] = nullptr; g->fn_typed[dtype] = nullptr; }
  g->typed_code[dtype].swap(co);
  return FDG_OK;
}

int fdg_eval_device_typed(fdg_graph *g, int dtype, const void *d_leaf, int64_t ss, int64_t ls, void *d_root, int64_t rs, int64_t rk,
                          int64_t B, void *stream) {
  if (!g) { set_error("null handle"); return FDG_E_INVALID; }
  if (dtype == FDG_DT_F64) return fdg_eval_device(g, (const double *)d_leaf, ss, ls, (double *)d_root, rs, rk, B, stream);
  if (dtype < 0 || dtype > FDG_DT_C32) { set_error("unknown element type"); return FDG_E_INVALID; }
  if (B < 0) { set_error("n_sample < 0"); return FDG_E_INVALID; }
  std::lock_guard<std::mutex> lk(g->mu);
  fdg::KnobScope knob_scope(&g->knobs);
  if (g->typed_code[dtype].empty()) { set_error("fdg_eval_device_typed: call fdg_graph_specialize_typed for this element type first"); return FDG_E_INVALID; }
  if (B == 0 || g->prog.R == 0) return FDG_OK;
  if ((g->prog.L && !d_leaf) || !d_root) { set_error("null device buffer"); return FDG_E_INVALID; }
  // (rows whose doubled strides the assembly kernels cannot address go on to the per-type kernel, which takes any stride)
  if (dtype == FDG_DT_C64 && g->cx_twin && ls == 1 && rk == 1 && ss >= (int64_t)g->prog.L && rs >= (int64_t)g->prog.R && B >= 64 &&
      2 * ss < (1ll << 23) && 2 * rs < (1ll << 23)) {
    const int rct = fdg_eval_device(g->cx_twin, (const double *)d_leaf, 2 * ss, 1, (double *)d_root, 2 * rs, 1, B, stream);
    if (rct == FDG_OK) {
      g->last_kernel = std::strcmp(g->cx_twin->last_kernel, "fdg_isa_eval_rl") == 0 ? "fdg_isa_eval_rl (ComplexF64 rows)" : "fdg_isa_eval_rm (ComplexF64 rows)";
      return rct;
    }
    if (rct != FDG_E_UNSUPPORTED) return rct;
  }
  int rc = ensure_device(g);
  if (rc) return rc;
  if (!g->typed_module[dtype]) {
    hipModule_t m; hipFunction_t f;
    hipError_t e = hipModuleLoadData(&m, g->typed_code[dtype].data());
    if (e != hipSuccess) { set_error("hipModuleLoadData failed: " + std::string(hipGetErrorString(e))); return FDG_E_JIT; }
    HIP_TRY(hipModuleGetFunction(&f, m, "fdg_spec_typed"));
    g->typed_module[dtype] = m; g->fn_typed[dtype] = f;
  }
  const long nblk = (long)((B + 255) / 256);
  const long grid = std::min<long>(nblk, (long)g->n_cu * 8);
  long a_ss = ss, a_ls = ls, a_rs = rs, a_rk = rk, a_B = B;
  void *args[] = {(void *)&d_leaf, &a_ss, &a_ls, (void *)&d_root, &a_rs, &a_rk, &a_B};
  HIP_TRY(hipModuleLaunchKernel((hipFunction_t)g->fn_typed[dtype], (unsigned)grid, 1, 1, 256, 1, 1, 0, (hipStream_t)stream, args, nullptr));
  static const char *names[] = {"", "fdg_spec_typed<Float32>", "fdg_spec_typed<ComplexF64>", "fdg_spec_typed<ComplexF32>"};
  g->last_kernel = names[dtype];
  return FDG_OK;
}

int fdg_eval_device(fdg_graph *g, const double *d_leaf, int64_t ss, int64_t ls, double *d_root, int64_t rs,
                    int64_t rk, int64_t B, void *stream) {
  return run(g, 0, d_leaf, ss, ls, d_root, rs, rk, nullptr, nullptr, B, (hipStream_t)stream);
}

int fdg_accumulate_device(fdg_graph *g, const double *d_leaf, int64_t ss, int64_t ls, const double *d_weight,
                          double *d_acc, int64_t B, void *stream) {
  return run(g, 1, d_leaf, ss, ls, nullptr, 0, 0, d_weight, d_acc, B, (hipStream_t)stream);
}

// Tile-major batches (include/fdg.h): a tile stride of 0 stands for 64 sample strides, i.e. the plain strided matrix.
int fdg_eval_device_tiled(fdg_graph *g, const double *d_leaf, int64_t ss, int64_t ls, int64_t lts, double *d_root, int64_t rs,
                          int64_t rk, int64_t rts, int64_t B, void *stream) {
  if (lts < 0 || rts < 0) { set_error("negative tile stride"); return FDG_E_INVALID; }
  if ((lts == 0 || lts == 64 * ss) && (rts == 0 || rts == 64 * rs)) return run(g, 0, d_leaf, ss, ls, d_root, rs, rk, nullptr, nullptr, B, (hipStream_t)stream);
  return run(g, 0, d_leaf, ss, ls, d_root, rs, rk, nullptr, nullptr, B, (hipStream_t)stream, lts ? lts : 64 * ss, rts ? rts : 64 * rs);
}

int fdg_accumulate_device_tiled(fdg_graph *g, const double *d_leaf, int64_t ss, int64_t ls, int64_t lts, const double *d_weight,
                                double *d_acc, int64_t B, void *stream) {
  if (lts < 0) { set_error("negative tile stride"); return FDG_E_INVALID; }
  if (lts == 0 || lts == 64 * ss) return run(g, 1, d_leaf, ss, ls, nullptr, 0, 0, d_weight, d_acc, B, (hipStream_t)stream);
  return run(g, 1, d_leaf, ss, ls, nullptr, 0, 0, d_weight, d_acc, B, (hipStream_t)stream, lts, 0);
}

// host <-> device copy of an [n x m] block of a strided host matrix (element (b, i) at h[b*hs + i*hi]) to / from the
// device block d[b*ds + i*di]; one of (hs, hi) and the matching one of (ds, di) is 1
static hipError_t copy_block(double *d, int64_t ds, int64_t di, const double *h, int64_t hs, int64_t hi, int64_t n, int64_t m, bool to_device) {
  if (n == 0 || m == 0) return hipSuccess;
  const hipMemcpyKind kind = to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost;
  // rows = the slow index, width = the contiguous run
  int64_t rows, width, hp, dp;
  if (hi == 1 && di == 1) { rows = n; width = m; hp = hs; dp = ds; }          // sample-major: a sample's values are contiguous
  else if (hs == 1 && ds == 1) { rows = m; width = n; hp = hi; dp = di; }     // leaf-major (a Julia column-major matrix)
  else return hipErrorInvalidValue;
  if (to_device) return hipMemcpy2D(d, (size_t)dp * 8, h, (size_t)hp * 8, (size_t)width * 8, (size_t)rows, kind);
  return hipMemcpy2D(const_cast<double *>(h), (size_t)hp * 8, d, (size_t)dp * 8, (size_t)width * 8, (size_t)rows, kind);
}

int fdg_eval_strided(fdg_graph *g, const double *leaf, int64_t ss, int64_t ls, double *root, int64_t rs, int64_t rk, int64_t B) {
  if (!g) { set_error("null handle"); return FDG_E_INVALID; }
  if (B < 0) { set_error("n_sample < 0"); return FDG_E_INVALID; }
  if (B == 0) return FDG_OK;
  const int64_t L = g->prog.L, R = g->prog.R;
  if ((L && !leaf) || (R && !root)) { set_error("null host buffer"); return FDG_E_INVALID; }
  const bool leaf_rows = ls == 1 && ss >= L, leaf_cols = ss == 1 && ls >= B;
  const bool root_rows = rk == 1 && rs >= R, root_cols = rs == 1 && rk >= B;
  if ((L && !leaf_rows && !leaf_cols) || (R && !root_rows && !root_cols)) {
    set_error("host matrices must be row-major (value stride 1, sample stride >= row length) or column-major (sample stride 1, value stride >= n_sample)");
    return FDG_E_INVALID;
  }
  { std::lock_guard<std::mutex> lk(g->mu);
  fdg::KnobScope knob_scope(&g->knobs); int rc = ensure_device(g); if (rc) return rc; }
  // host buffers of any size: the batch goes through the device in chunks of at most ~2 GiB (FDG_EVAL_CHUNK
  // samples overrides, for tests), so device memory bounds nothing.  The device copy keeps the host's layout: a
  // column-major (Julia) matrix arrives leaf-major, the layout the ISA kernel is built around -- no transposition anywhere.
  int64_t chunk = std::max<int64_t>(65536, (int64_t)((2ull << 30) / (8ull * (uint64_t)std::max<int64_t>(L + R, 1))));
  if (g->cfg.eval_chunk) chunk = std::max<int64_t>(1, (int64_t)g->cfg.eval_chunk);
  chunk = std::min<int64_t>(chunk, B);
  double *dl = nullptr, *dr = nullptr;
  HIP_TRY(hipMalloc(&dl, (size_t)std::max<int64_t>(1, chunk * L) * 8));
  if (hipMalloc(&dr, (size_t)std::max<int64_t>(1, chunk * R) * 8) != hipSuccess) { hipFree(dl); set_error("hipMalloc failed"); return FDG_E_NOMEM; }
  int rc = FDG_OK;
  for (int64_t c0 = 0; c0 < B && rc == FDG_OK; c0 += chunk) {
    const int64_t n = std::min<int64_t>(chunk, B - c0);
    const int64_t dss = leaf_rows ? L : 1, dls = leaf_rows ? 1 : n;     // device block: compact, same orientation
    const int64_t drs = root_rows ? R : 1, drk = root_rows ? 1 : n;
    if (L && copy_block(dl, dss, dls, leaf + c0 * ss, ss, ls, n, L, true) != hipSuccess) { set_error("H2D copy failed"); rc = FDG_E_NO_DEVICE; break; }
    // eval_graph! leaves root entries it does not assign untouched: start from the caller's values
    if (R && copy_block(dr, drs, drk, root + c0 * rs, rs, rk, n, R, true) != hipSuccess) { set_error("H2D copy failed"); rc = FDG_E_NO_DEVICE; break; }
    rc = run(g, 0, dl, dss, dls, dr, drs, drk, nullptr, nullptr, n, nullptr);
    if (rc) break;
    if (hipDeviceSynchronize() != hipSuccess) { set_error("kernel execution failed"); rc = FDG_E_NO_DEVICE; break; }
    if (R && copy_block(dr, drs, drk, root + c0 * rs, rs, rk, n, R, false) != hipSuccess) { set_error("D2H copy failed"); rc = FDG_E_NO_DEVICE; break; }
  }
  hipFree(dl); hipFree(dr);
  return rc;
}

int fdg_eval(fdg_graph *g, const double *leaf, double *root, int64_t B) {
  return fdg_eval_strided(g, leaf, g ? (int64_t)g->prog.L : 0, 1, root, g ? (int64_t)g->prog.R : 0, 1, B);
}

int fdg_isa_check_hazards(const char *asm_text, char **report) {
  if (!asm_text) { set_error("null argument"); return FDG_E_INVALID; }
  std::string rep;
  const int n = fdg::check_isa_hazards(asm_text, rep);
  if (report) {
    const std::string all = fdg::isa_hazard_table() + rep;
    char *m = (char *)std::malloc(all.size() + 1);
    if (!m) { set_error("out of memory"); return FDG_E_NOMEM; }
    std::memcpy(m, all.c_str(), all.size() + 1);
    *report = m;
  }
  return n;
}

int fdg_copy_device(double *d_dst, const double *d_src, int64_t n, void *stream) {
  if (n < 0 || (n & 1)) { set_error("fdg_copy_device: n must be even and >= 0"); return FDG_E_INVALID; }
  if (n == 0) return FDG_OK;
  if (!d_dst || !d_src || (((uintptr_t)d_dst | (uintptr_t)d_src) & 15)) { set_error("fdg_copy_device: null or not 16-byte aligned"); return FDG_E_INVALID; }
  const long n16 = (long)(n / 2);
  const long grid = std::max<long>(1, std::min<long>(n16 / 1024, 65536L));
  if (!fdg::knob("FDG_COPY_PLAIN"))     // non-temporal accesses: 5.96 TB/s against 5.53 (the ceiling a stream is measured against)
    hipLaunchKernelGGL(fdg_copy16_nt, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const fdg_v2d *)d_src, (fdg_v2d *)d_dst, n16);
  else
    hipLaunchKernelGGL(fdg_copy16, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const fdg_v2d *)d_src, (fdg_v2d *)d_dst, n16);
  HIP_TRY(hipGetLastError());
  return FDG_OK;
}

static int repack_launch(bool to_tiled, double *mat, int64_t ss, int64_t cs, double *tiled, int64_t n, uint32_t C, void *stream, const char *who) {
  if (n < 0 || ss < 0 || cs < 0) { set_error(std::string(who) + ": negative size or stride"); return FDG_E_INVALID; }
  if (n == 0 || C == 0) return FDG_OK;
  if (!mat || !tiled) { set_error(std::string(who) + ": null device buffer"); return FDG_E_INVALID; }
  if (((uintptr_t)tiled) & 15) { set_error(std::string(who) + ": the tile-major array must be 16-byte aligned"); return FDG_E_INVALID; }
  int dev = 0, n_cu = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
  const long ntile = (long)((n + 63) / 64);
  if (ss == 1) {
    const long nrun = ntile * (long)C;
    const unsigned grid = (unsigned)std::max<long>(1, std::min<long>((nrun + 7) / 8, (long)n_cu * 32));
    if (to_tiled) hipLaunchKernelGGL(fdg_repack_runs<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, mat, (long)cs, tiled, (long)n, C);
    else hipLaunchKernelGGL(fdg_repack_runs<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, mat, (long)cs, tiled, (long)n, C);
  } else {
    const long nt = ntile * (((long)C + 63) / 64);
    const unsigned grid = (unsigned)std::max<long>(1, std::min<long>(nt, (long)n_cu * 16));
    if (to_tiled) hipLaunchKernelGGL(fdg_repack_transpose<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, mat, (long)ss, (long)cs, tiled, (long)n, C);
    else hipLaunchKernelGGL(fdg_repack_transpose<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, mat, (long)ss, (long)cs, tiled, (long)n, C);
  }
  HIP_TRY(hipGetLastError());
  return FDG_OK;
}

int fdg_repack_tile_major(const double *d_src, int64_t sample_stride, int64_t col_stride, double *d_tiled, int64_t n_sample, uint32_t n_col, void *stream) {
  return repack_launch(true, const_cast<double *>(d_src), sample_stride, col_stride, d_tiled, n_sample, n_col, stream, "fdg_repack_tile_major");
}

int fdg_unpack_tile_major(const double *d_tiled, double *d_dst, int64_t sample_stride, int64_t col_stride, int64_t n_sample, uint32_t n_col, void *stream) {
  return repack_launch(false, d_dst, sample_stride, col_stride, const_cast<double *>(d_tiled), n_sample, n_col, stream, "fdg_unpack_tile_major");
}

int fdg_read_device(const double *d_src, int64_t n, double *d_sink, void *stream) {
  if (n < 0) { set_error("fdg_read_device: n must be >= 0"); return FDG_E_INVALID; }
  if (n == 0) return FDG_OK;
  if (!d_src || !d_sink) { set_error("fdg_read_device: null pointer"); return FDG_E_INVALID; }
  hipLaunchKernelGGL(fdg_read8_nt, dim3(256 * 16), dim3(256), 0, (hipStream_t)stream, d_src, d_sink, (long)n);
  HIP_TRY(hipGetLastError());
  return FDG_OK;
}

// Measurement aid: one wave that sleeps on a SIMD for `seconds` of wall time and reports how many shader-clock ticks
// (s_memtime) and 100 MHz ticks (s_memrealtime) went by -- the clock the chip sustained under whatever ran next to it.
// Eight VGPRs and no LDS: it fits beside two 248-register waves of an evaluator kernel on the same SIMD.
__global__ void __launch_bounds__(64) fdg_clock_probe_kernel(long long *out, long long wall_ticks) {
  if (threadIdx.x) return;
  const long long w0 = wall_clock64(), c0 = clock64();
  long long w, c;
  do {
    for (int i = 0; i < 16; ++i) __builtin_amdgcn_s_sleep(127);
    w = wall_clock64(); c = clock64();
  } while (w - w0 < wall_ticks);
  out[0] = c - c0;
  out[1] = w - w0;
}

int fdg_clock_probe_device(double seconds, int64_t *d_ticks, void *stream) {
  if (!d_ticks) { set_error("fdg_clock_probe_device: null buffer"); return FDG_E_INVALID; }
  if (!(seconds > 0) || seconds > 30.0) { set_error("fdg_clock_probe_device: seconds must lie in (0, 30]"); return FDG_E_INVALID; }
  hipFuncAttributes at;
  HIP_TRY(hipFuncGetAttributes(&at, reinterpret_cast<const void *>(fdg_clock_probe_kernel)));
  if (at.numRegs > 16 || at.sharedSizeBytes != 0) { set_error("fdg_clock_probe_device: the probe no longer fits beside a full evaluator wave pair"); return FDG_E_UNSUPPORTED; }
  hipLaunchKernelGGL(fdg_clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long *)d_ticks, (long long)(seconds * 1e8));
  HIP_TRY(hipGetLastError());
  return FDG_OK;
}

int fdg_fill_uniform_device(double *d_leaf, int64_t B, uint32_t L, int64_t ss, int64_t ls, uint64_t seed,
                            uint64_t off, void *stream) {
  if (B < 0) { set_error("n_sample < 0"); return FDG_E_INVALID; }
  if (B == 0 || L == 0) return FDG_OK;
  if (!d_leaf) { set_error("null device buffer"); return FDG_E_INVALID; }
  const long total = (long)B * L;
  const long grid = std::min<long>((total + 255) / 256, 256L * 16);
  hipLaunchKernelGGL(fdg_fill_uniform, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, d_leaf, (long)B, L,
                     (long)ss, (long)ls, seed, off, (ls <= ss) ? 1 : 0);
  HIP_TRY(hipGetLastError());
  return FDG_OK;
}

int fdg_fill_uniform_device_tiled(double *d_leaf, int64_t B, uint32_t L, int64_t ss, int64_t ls, int64_t lts, uint64_t seed,
                                  uint64_t off, void *stream) {
  if (B < 0 || lts < 0) { set_error("n_sample < 0 or negative tile stride"); return FDG_E_INVALID; }
  if (lts == 0 || lts == 64 * ss) return fdg_fill_uniform_device(d_leaf, B, L, ss, ls, seed, off, stream);
  if (B == 0 || L == 0) return FDG_OK;
  if (!d_leaf) { set_error("null device buffer"); return FDG_E_INVALID; }
  const long total = (long)((B + 63) / 64) * 64L * L;
  const long grid = std::min<long>((total + 255) / 256, 256L * 16);
  hipLaunchKernelGGL(fdg_fill_uniform_tiled, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, d_leaf, (long)B, L,
                     (long)ss, (long)ls, (long)lts, seed, off);
  HIP_TRY(hipGetLastError());
  return FDG_OK;
}

}  // extern "C"
