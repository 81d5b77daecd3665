// Why do a few stores in a read stream cost 3-4 times their bytes?  Hypothesis: vmcnt counts loads and stores
// together and a wait for a load that was issued AFTER a store is also a wait for that store's acknowledgement from L2 --
// which takes longer than a load under a saturated read stream -- so every tile start stalls on the previous tile's root stores.
// The evaluator's pattern (84 columns x 512 B per tile, 12 fp64 ops per load, 4 root stores per tile, two waves per SIMD) in
// three orders of issue:
//   MODE 0: a tile's stores at its end, then the next tile's first loads (what the evaluator does)
//   MODE 1: the stores of tile t are issued after the first quarter of tile t + 1's loads (those loads are older than the stores:
//           waiting for them does not wait for the stores); the rest of the tile's loads are younger
//   MODE 2: every load is issued one whole tile ahead: no load that is waited for within a tile is younger than the previous
//           tile's stores (168 registers of landing space)
// (dev tool; hipcc --offload-arch=gfx950 -O3 store_ack.hip -o /tmp/store_ack)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
static int g_waves_per_cu = 8;      // resident waves per CU of the persistent grid (argv: --waves N)
#define OPS4(n) for (int i = 0; i < (n); i += 4) asm volatile("v_mul_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));
template <int NCOL, int MODE, int TILED>
__global__ void __launch_bounds__(64, 2) k(const double *__restrict__ src, double *__restrict__ dst, long ntile, long col_stride, int ops, int nstore) {
  const long wave = blockIdx.x, nw = gridDim.x;
  double a0 = threadIdx.x * 1e-9 + 1.0, a1 = a0 + 1e-9, a2 = a0 + 2e-9, a3 = a0 + 3e-9;
  const double m = 1.0000001;
  // leaf-major: column c of tile t at src[c * col_stride + 64 t + lane]; tile-major: src[(t * NCOL + c) * 64 + lane]
  auto at = [&](long t, int c) { return TILED ? src + (t * NCOL + c) * 64 + threadIdx.x : src + c * col_stride + t * 64 + threadIdx.x; };
  auto out = [&](long t, int r) { return TILED ? dst + (t * 8 + r) * 64 + threadIdx.x : dst + r * col_stride + t * 64 + threadIdx.x; };
  auto ld = [&](const double *p) { return __builtin_nontemporal_load(p); };
  auto st = [&](double v, double *p) { __builtin_nontemporal_store(v, p); };
  if (MODE == 2) {
    double v[NCOL], s = 0.0;
#pragma unroll
    for (int c = 0; c < NCOL; ++c) v[c] = ld(at(wave, c));
    for (long t = wave; t < ntile; t += nw) {
      const long tn = t + nw < ntile ? t + nw : t;
      s = 0.0;
#pragma unroll
      for (int c = 0; c < NCOL; ++c) {
        s += v[c];
        v[c] = ld(at(tn, c));
        OPS4(ops)
      }
      for (int r = 0; r < nstore; ++r) st(s + a0 + r, out(t, r));
    }
  } else {
    constexpr int Q = NCOL / 4;            // look-ahead: a quarter of the tile's columns
    double s_prev = 0.0;
    long t_prev = -1;
    for (long t = wave; t < ntile; t += nw) {
      double v[Q], s = 0.0;
#pragma unroll
      for (int c = 0; c < Q; ++c) v[c] = ld(at(t, c));
      if (MODE == 1 && t_prev >= 0) for (int r = 0; r < nstore; ++r) st(s_prev + a0 + r, out(t_prev, r));
#pragma unroll
      for (int c = 0; c < NCOL; ++c) {
        s += v[c % Q];
        if (c + Q < NCOL) v[c % Q] = ld(at(t, c + Q));
        OPS4(ops)
      }
      if (MODE == 0) for (int r = 0; r < nstore; ++r) st(s + a0 + r, out(t, r));
      s_prev = s; t_prev = t;
    }
    if (MODE == 1 && t_prev >= 0) for (int r = 0; r < nstore; ++r) st(s_prev + a0 + r, out(t_prev, r));
  }
  if (a0 + a1 + a2 + a3 == 12345.678) dst[0] = a1;
}
// MODE 3: true software pipelining across tiles: column g + Q of the (tile, column) stream is requested when column g is consumed, so
// every load that is waited for during the Q columns after a tile's stores is OLDER than those stores; NS stores per tile (static)
template <int NCOL, int Q, int NS, int TILED>
__global__ void __launch_bounds__(64, 2) kp(const double *__restrict__ src, double *__restrict__ dst, long ntile, long col_stride, int ops) {
  static_assert(NCOL % Q == 0, "Q must divide the column count");
  const long wave = blockIdx.x, nw = gridDim.x;
  double a0 = threadIdx.x * 1e-9 + 1.0, a1 = a0 + 1e-9, a2 = a0 + 2e-9, a3 = a0 + 3e-9;
  const double m = 1.0000001;
  auto at = [&](long t, int c) { return TILED ? src + (t * NCOL + c) * 64 + threadIdx.x : src + c * col_stride + t * 64 + threadIdx.x; };
  auto out = [&](long t, int r) { return TILED ? dst + (t * 8 + r) * 64 + threadIdx.x : dst + r * col_stride + t * 64 + threadIdx.x; };
  double v[Q];
#pragma unroll
  for (int c = 0; c < Q; ++c) v[c] = __builtin_nontemporal_load(at(wave, c));
  for (long t = wave; t < ntile; t += nw) {
    const long tn = t + nw < ntile ? t + nw : t;
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < NCOL; ++c) {
      s += v[c % Q];
      v[c % Q] = __builtin_nontemporal_load(c + Q < NCOL ? at(t, c + Q) : at(tn, c + Q - NCOL));
      OPS4(ops)
    }
#pragma unroll
    for (int r = 0; r < NS; ++r) __builtin_nontemporal_store(s + a0 + r, out(t, r));
  }
  if (a0 + a1 + a2 + a3 == 12345.678) dst[0] = a1;
}
template <int NCOL, int Q, int NS, int TILED> void runp(const double *src, double *dst, long total_bytes, int ops) {
  const long ntile = total_bytes / (NCOL * 512L);
  const long cs = ntile * 64;
  const int grid = 256 * 4 * 2;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((kp<NCOL, Q, NS, TILED>), dim3(grid), dim3(64), 0, 0, src, dst, ntile, cs, ops);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((kp<NCOL, Q, NS, TILED>), dim3(grid), dim3(64), 0, 0, src, dst, ntile, cs, ops);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("%s pipelined Q=%2d columns=%3d ops/load=%2d stores=%d  %.3f ms  reads %.2f TB/s  reads+writes %.2f TB/s\n", TILED ? "tile-major" : "leaf-major", Q, NCOL, ops, NS, ms,
         (double)ntile * NCOL * 512 / ms / 1e9, (double)ntile * (NCOL + NS) * 512 / ms / 1e9);
}
template <int NCOL, int MODE, int TILED> void run(const double *src, double *dst, long total_bytes, int ops, int nstore) {
  const long ntile = total_bytes / (NCOL * 512L);
  const long cs = ntile * 64;
  const int grid = 256 * g_waves_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<NCOL, MODE, TILED>), dim3(grid), dim3(64), 0, 0, src, dst, ntile, cs, ops, nstore);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<NCOL, MODE, TILED>), dim3(grid), dim3(64), 0, 0, src, dst, ntile, cs, ops, nstore);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("%s mode=%d columns=%3d ops/load=%2d stores=%d  %.3f ms  reads %.2f TB/s  reads+writes %.2f TB/s\n", TILED ? "tile-major" : "leaf-major", MODE, NCOL, ops, nstore, ms,
         (double)ntile * NCOL * 512 / ms / 1e9, (double)ntile * (NCOL + nstore) * 512 / ms / 1e9);
}
// backing: chunk_mb == 0: hipMalloc; else physical chunks of chunk_mb MB mapped in address order (what fdg_batch_alloc does)
static void *backed(size_t bytes, size_t chunk_mb) {
  void *p = nullptr;
  if (!chunk_mb) { hipMalloc(&p, bytes); return p; }
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  const size_t chunk = chunk_mb << 20, total = (bytes + chunk - 1) / chunk * chunk;
  if (hipMemAddressReserve(&p, total, chunk > ((size_t)1 << 30) ? (size_t)1 << 30 : chunk, nullptr, 0) != hipSuccess) return nullptr;
  for (size_t off = 0; off < total; off += chunk) {
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess || hipMemMap((char *)p + off, chunk, 0, h, 0) != hipSuccess) return nullptr;
  }
  hipMemAccessDesc acc = {};
  acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
  if (hipMemSetAccess(p, total, &acc, 1) != hipSuccess) return nullptr;
  return p;
}
int main(int argc, char **argv) {
  if (argc > 2 && std::string(argv[argc - 2]) == "--waves") { g_waves_per_cu = atoi(argv[argc - 1]); argc -= 2; }
  const long gb = argc > 1 ? atol(argv[1]) : 32;
  const long total = gb << 30;
  for (int a = 2; a < (argc > 2 ? argc : 3); ++a) {
    const size_t chunk_mb = argc > 2 ? (size_t)atol(argv[a]) : 0;
    double *src = (double *)backed(total + (1 << 20), chunk_mb), *dst = (double *)backed((total / 84) * 8 + (1 << 20), chunk_mb);
    if (!src || !dst) { printf("allocation failed (chunk %zu MB)\n", chunk_mb); return 1; }
    hipMemset(src, 0, total);
    printf("-- %ld GB, backing: %s %zu MB, %d waves per CU\n", gb, chunk_mb ? "chunks of" : "hipMalloc", chunk_mb, g_waves_per_cu);
    for (int nstore : {0, 4}) {
      run<84, 0, 0>(src, dst, total, 12, nstore);
      run<84, 0, 1>(src, dst, total, 12, nstore);
    }
    runp<84, 21, 0, 1>(src, dst, total, 12); runp<84, 21, 4, 1>(src, dst, total, 12);
    runp<84, 28, 0, 1>(src, dst, total, 12); runp<84, 28, 4, 1>(src, dst, total, 12);
    runp<84, 42, 0, 1>(src, dst, total, 12); runp<84, 42, 4, 1>(src, dst, total, 12);
    runp<84, 42, 0, 0>(src, dst, total, 12); runp<84, 42, 4, 0>(src, dst, total, 12);
    runp<84, 42, 8, 1>(src, dst, total, 12);
  }
  printf("last error: %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
