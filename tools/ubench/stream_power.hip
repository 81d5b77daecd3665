// Round 5: what does a streamed byte cost in watts as a function of the width of the load?  Read-only streams over 8 GiB (the evaluator's traffic is 95 % reads),
// 4 / 8 / 16 bytes per lane, plain and non-temporal, each variant running back to back for argv[1] seconds (rocm-smi is sampled next to it by
// tools/gpu_stream_power.sh).    hipcc --offload-arch=gfx950 -O2 -o stream_power.bin stream_power.hip && ./stream_power.bin 6 [variant]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
template <typename T, int NT, int UNROLL>
__global__ void __launch_bounds__(256) rd(const T *__restrict__ s, float *out, long n) {
  const long stride = (long)gridDim.x * 256L;
  float acc = 0.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
    T r[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) r[u] = NT ? __builtin_nontemporal_load(s + i + u * stride) : s[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += ((const float *)&r[u])[0];
  }
  if (acc == 123.456f) out[0] = acc;
}
template <typename T, int NT> double run(const char *name, const void *buf, float *out, long bytes, double secs) {
  const long n = bytes / (long)sizeof(T);
  auto launch = [&]() { hipLaunchKernelGGL((rd<T, NT, 8>), dim3(256 * 16), dim3(256), 0, 0, (const T *)buf, out, n); };
  for (int w = 0; w < 5; ++w) launch();
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const auto t0 = std::chrono::steady_clock::now();
  long k = 0;
  (void)hipEventRecord(e0);
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) { for (int r = 0; r < 20; ++r) launch(); (void)hipDeviceSynchronize(); k += 20; }
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double tbs = (double)k * bytes / (ms * 1e-3) / 1e12;
  printf("%-28s %.3f TB/s read\n", name, tbs); fflush(stdout);
  return tbs;
}
int main(int argc, char **argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 5.0;
  const int which = argc > 2 ? atoi(argv[2]) : -1;
  const long bytes = 8L << 30;
  void *a; float *out; (void)hipMalloc(&a, bytes); (void)hipMalloc(&out, 64); (void)hipMemset(a, 0, bytes);
  if (which < 0 || which == 0) run<float, 0>("4 B per lane", a, out, bytes, secs);
  if (which < 0 || which == 1) run<v2f, 0>("8 B per lane", a, out, bytes, secs);
  if (which < 0 || which == 2) run<v4f, 0>("16 B per lane", a, out, bytes, secs);
  if (which < 0 || which == 3) run<v2f, 1>("8 B per lane, non-temporal", a, out, bytes, secs);
  if (which < 0 || which == 4) run<v4f, 1>("16 B per lane, non-temporal", a, out, bytes, secs);
  return 0;
}
