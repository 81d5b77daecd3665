// Sample batches owned by the library (include/fdg.h: fdg_batch_alloc / fdg_batch_free).
//
// A Monte-Carlo driver allocates its sample batch once and streams it thousands of times, and on MI355X the rate of that
// stream depends on how the batch is backed.  hipMalloc leaves the backing to the driver's state at the time of
// the call; this allocator builds the batch explicitly through the virtual-memory-management API -- one reserved address range,
// physical chunks of a stated size created and mapped in address order -- so that the backing is a property of the request and
// not of the process's allocation history.  The reference has no counterpart (its leaf vector is a Julia Vector).
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "fdg_internal.h"

namespace {
struct Batch {
  size_t bytes = 0;        // reserved and mapped (a multiple of the chunk size)
  size_t chunk = 0;
  int device = 0;
  std::vector<hipMemGenericAllocationHandle_t> handles;
};
std::mutex g_mu;
std::map<void *, Batch> g_batches;

int fail(const char *what, hipError_t e) {
  fdg::set_error(std::string(what) + ": " + hipGetErrorString(e));
  return e == hipErrorOutOfMemory ? FDG_E_NOMEM : FDG_E_NO_DEVICE;
}
}  // namespace

extern "C" {

int fdg_batch_alloc(size_t bytes, size_t chunk_bytes, void **d_ptr) {
  if (!d_ptr) { fdg::set_error("null out pointer"); return FDG_E_INVALID; }
  *d_ptr = nullptr;
  if (bytes == 0) { fdg::set_error("empty batch"); return FDG_E_INVALID; }
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return fail("hipGetDevice", e);
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran_min = 0, gran_rec = 0;
  e = hipMemGetAllocationGranularity(&gran_min, &prop, hipMemAllocationGranularityMinimum);
  if (e != hipSuccess) return fail("hipMemGetAllocationGranularity", e);
  e = hipMemGetAllocationGranularity(&gran_rec, &prop, hipMemAllocationGranularityRecommended);
  if (e != hipSuccess || gran_rec < gran_min) gran_rec = gran_min;
  if (gran_min == 0) gran_min = 1 << 21;
  // default: one physical allocation for the whole batch (the driver's buddy allocator then hands out its largest blocks)
  size_t chunk = chunk_bytes ? chunk_bytes : bytes;
  chunk = (chunk + gran_rec - 1) / gran_rec * gran_rec;
  const size_t total = (bytes + chunk - 1) / chunk * chunk;
  void *base = nullptr;
  e = hipMemAddressReserve(&base, total, (size_t)1 << 30 > chunk ? chunk : (size_t)1 << 30, nullptr, 0);
  if (e != hipSuccess) return fail("hipMemAddressReserve", e);
  Batch b;
  b.bytes = total; b.chunk = chunk; b.device = dev;
  auto undo = [&]() {
    for (size_t i = 0; i < b.handles.size(); ++i) { (void)hipMemUnmap((char *)base + i * chunk, chunk); (void)hipMemRelease(b.handles[i]); }
    (void)hipMemAddressFree(base, total);
  };
  for (size_t off = 0; off < total; off += chunk) {
    hipMemGenericAllocationHandle_t h;
    e = hipMemCreate(&h, chunk, &prop, 0);
    if (e != hipSuccess) { undo(); return fail("hipMemCreate", e); }
    e = hipMemMap((char *)base + off, chunk, 0, h, 0);
    if (e != hipSuccess) { (void)hipMemRelease(h); undo(); return fail("hipMemMap", e); }
    b.handles.push_back(h);
  }
  hipMemAccessDesc acc = {};
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = dev;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  e = hipMemSetAccess(base, total, &acc, 1);
  if (e != hipSuccess) { undo(); return fail("hipMemSetAccess", e); }
  {
    std::lock_guard<std::mutex> lk(g_mu);
    g_batches[base] = std::move(b);
  }
  *d_ptr = base;
  return FDG_OK;
}

int fdg_batch_free(void *d_ptr) {
  if (!d_ptr) return FDG_OK;
  Batch b;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_batches.find(d_ptr);
    if (it == g_batches.end()) { fdg::set_error("not a batch of fdg_batch_alloc"); return FDG_E_INVALID; }
    b = std::move(it->second);
    g_batches.erase(it);
  }
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) return fail("hipDeviceSynchronize", e);
  for (size_t i = 0; i < b.handles.size(); ++i) {
    (void)hipMemUnmap((char *)d_ptr + i * b.chunk, b.chunk);
    (void)hipMemRelease(b.handles[i]);
  }
  e = hipMemAddressFree(d_ptr, b.bytes);
  if (e != hipSuccess) return fail("hipMemAddressFree", e);
  return FDG_OK;
}

}  // extern "C"
