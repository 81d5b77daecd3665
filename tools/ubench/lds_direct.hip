// Checks the semantics the ISA back end relies on for LDS-direct leaf prefetch on gfx950:
//   s_mov_b32 m0, <lds byte offset>;  global_load_lds_dword voff, s[base:base+1] offset:{0,256}
// writes lane i's dword to LDS[m0 + inst_offset + 4 i]; two of them move a 512-byte tile (64 doubles)
// verbatim, and ds_read_b64 at lane*8 then yields the lane's double.  (dev tool)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(64) k(const double *src, double *dst, int nslot) {
  extern __shared__ double lds[];
  const unsigned lane4 = threadIdx.x * 4, lane8 = threadIdx.x * 8;
  for (int s = 0; s < nslot; ++s) {
    const double *p = src + (size_t)(blockIdx.x * nslot + s) * 64;
    unsigned m = (unsigned)(s % 8) * 512;
    asm volatile("s_mov_b32 m0, %0\n s_nop 0\n global_load_lds_dword %1, %2\n global_load_lds_dword %1, %2 offset:256\n s_waitcnt vmcnt(0)"
                 :: "s"(m), "v"(lane4), "s"(p) : "memory", "m0");
    double v;
    asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lane8 + m) : "memory");
    dst[(size_t)(blockIdx.x * nslot + s) * 64 + threadIdx.x] = v;
  }
  if (src == nullptr) dst[0] = lds[threadIdx.x];   // keep the allocation
}
int main() {
  const int nb = 512, nslot = 16; const size_t n = (size_t)nb * nslot * 64;
  std::vector<double> h(n), o(n);
  for (size_t i = 0; i < n; ++i) h[i] = 1.0 + i * 1e-3;
  double *d, *e; hipMalloc(&d, n * 8); hipMalloc(&e, n * 8);
  hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice); hipMemset(e, 0, n * 8);
  hipLaunchKernelGGL(k, dim3(nb), dim3(64), 4096, 0, d, e, nslot);
  hipMemcpy(o.data(), e, n * 8, hipMemcpyDeviceToHost);
  size_t bad = 0; for (size_t i = 0; i < n; ++i) bad += o[i] != h[i];
  for (int i = 0; i < 6; ++i) printf("o[%d]=%.6f h=%.6f  o[64+%d]=%.6f\n", i, o[i], h[i], i, o[64+i]);
  printf("lds-direct tile copy: %zu mismatches of %zu (%s)\n", bad, n, hipGetErrorString(hipGetLastError()));
  return bad != 0;
}
