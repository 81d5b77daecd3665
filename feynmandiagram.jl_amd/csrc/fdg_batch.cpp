// Sample batches owned by the library (include/fdg.h: fdg_batch_alloc / fdg_batch_free).
//
// A Monte-Carlo driver allocates its sample batch once and streams it thousands of times, and on MI355X the rate of that
// stream depends on how the batch is backed.  hipMalloc leaves the backing to the driver's state at the time of
// the call; this allocator builds the batch explicitly through the virtual-memory-management API -- one reserved address range,
// physical chunks of a stated size created and mapped in address order -- so that the backing is a property of the request and
// not of the process's allocation history.  The reference has no counterpart (its leaf vector is a Julia Vector).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
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
  // alignment: the largest power of two that divides the chunk size (at least the granularity, at most 1 GiB) -- a chunk of 6 MB is not a
  // valid alignment itself
  size_t align = chunk & (~chunk + 1);
  if (align > ((size_t)1 << 30)) align = (size_t)1 << 30;
  e = hipMemAddressReserve(&base, total, align, nullptr, 0);
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
  if (b.chunk == 0) {            // the leaves of fdg_batch_alloc_pair: a plain allocation
    e = hipFree(d_ptr);
    return e == hipSuccess ? FDG_OK : fail("hipFree", e);
  }
  for (size_t i = 0; i < b.handles.size(); ++i) {
    (void)hipMemUnmap((char *)d_ptr + i * b.chunk, b.chunk);
    (void)hipMemRelease(b.handles[i]);
  }
  e = hipMemAddressFree(d_ptr, b.bytes);
  if (e != hipSuccess) return fail("hipMemAddressFree", e);
  return FDG_OK;
}


// ---------------------------------------------------------------------------------------------------------------------------------
// fdg_batch_alloc_pair: the tile-major leaf and root arrays of one handle, backed chunk by chunk so that every chunk of leaves streams
// next to ITS chunk of roots at the fast rate (round 5).
//
// What round 5 measured (profiles/r05_log_chunk_probe.txt, r05_log_pair_probe.txt, r05_log_pair_matrix.txt): the rate of the evaluation
// over a piece of the batch is a stable, LOCAL property of the physical pages under that piece -- and a PAIRWISE one: device memory falls
// into regions of two kinds (gigabytes each), and reading leaves of one kind while writing roots of the same kind runs at 0.75-0.77 of
// 8 TB/s where the other combination runs at 0.85-0.86 (fused accumulation, which writes nothing: 0.89 everywhere).  hipMalloc hands out
// whatever regions it has, so a 70 GB batch is a patchwork of matched and mismatched pieces: the "allocation lottery" of rounds 3-4.
// Physical addresses are not visible to a process, but the rate is: this allocator maps the leaves in chunks, draws more root chunks
// than it needs, TIMES the handle's own evaluator on (leaf chunk, root chunk) pairs, and maps behind every leaf chunk a root chunk that
// gives the fast rate.
namespace {
struct PairCtx {
  fdg_graph *g = nullptr;
  uint32_t L = 0, R = 0;
  size_t chunk_tiles = 0, leaf_chunk = 0, root_chunk = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  uint32_t n_probe = 0;
  bool row_major = false;      // FDG_BATCH_PAIR_ROW_MAJOR: compile_Python's [B, L] / [B, R] instead of the tile-major arrays (64 rows = one "tile")
  int64_t ld = 0;              // FDG_BATCH_PAIR_LEAF_MAJOR: > 0 = column stride (samples) of a Julia column-major B x L / B x R pair; one window: the whole batch
  int eval(const void *leaf, void *root, int64_t n) {
    if (ld) return fdg_eval_device(g, (const double *)leaf, 1, ld, (double *)root, 1, ld, n, nullptr);
    return row_major ? fdg_eval_device(g, (const double *)leaf, (int64_t)L, 1, (double *)root, (int64_t)R, 1, n, nullptr)
                     : fdg_eval_device_tiled(g, (const double *)leaf, 1, 64, 64 * (int64_t)L, (double *)root, 1, 64, 64 * (int64_t)R, n, nullptr);
  }
  int accumulate(const void *leaf, double *d_acc, int64_t n) {
    if (ld) return fdg_accumulate_device(g, (const double *)leaf, 1, ld, nullptr, d_acc, n, nullptr);
    return row_major ? fdg_accumulate_device(g, (const double *)leaf, (int64_t)L, 1, nullptr, d_acc, n, nullptr)
                     : fdg_accumulate_device_tiled(g, (const double *)leaf, 1, 64, 64 * (int64_t)L, nullptr, d_acc, n, nullptr);
  }
  int fill(void *leaf, int64_t n) {
    if (ld) return fdg_fill_uniform_device((double *)leaf, n, L, 1, ld, 20240612u, 0, nullptr);
    return row_major ? fdg_fill_uniform_device((double *)leaf, n, L, (int64_t)L, 1, 20240612u, 0, nullptr)
                     : fdg_fill_uniform_device_tiled((double *)leaf, n, L, 1, 64, 64 * (int64_t)L, 20240612u, 0, nullptr);
  }
  // algorithmic GB/s of the handle's evaluation over one chunk: leaves at `leaf`, roots at `root` (min of `reps` launches after one warm-up)
  int probe(const void *leaf, void *root, double &gbs, int reps = 4) {
    const int64_t n = (int64_t)chunk_tiles * 64;
    float best = 0.f;
    for (int r = 0; r <= reps; ++r) {
      if (hipEventRecord(ev0, nullptr) != hipSuccess) return FDG_E_NO_DEVICE;
      const int rc = eval(leaf, root, n);
      if (rc) return rc;
      if (hipEventRecord(ev1, nullptr) != hipSuccess || hipEventSynchronize(ev1) != hipSuccess) return FDG_E_NO_DEVICE;
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, ev0, ev1) != hipSuccess) return FDG_E_NO_DEVICE;
      if (r > 0 && (best == 0.f || ms < best)) best = ms;
    }
    ++n_probe;
    gbs = best > 0.f ? 8.0 * (double)(L + R) * (double)n / ((double)best * 1e6) : 0.0;
    return FDG_OK;
  }
  // the fused accumulation (nothing written) over one chunk, expressed in the evaluation's bytes: what a pair could reach if its writes were free
  int probe_readonly(const void *leaf, double &gbs, int reps = 4) {
    const int64_t n = (int64_t)chunk_tiles * 64;
    double *d_acc = nullptr;
    if (hipMalloc((void **)&d_acc, sizeof(double) * std::max<uint32_t>(R, 1)) != hipSuccess) return FDG_E_NOMEM;
    (void)hipMemset(d_acc, 0, sizeof(double) * R);
    float best = 0.f;
    int rc = FDG_OK;
    for (int r = 0; r <= reps && rc == FDG_OK; ++r) {
      if (hipEventRecord(ev0, nullptr) != hipSuccess) { rc = FDG_E_NO_DEVICE; break; }
      rc = accumulate(leaf, d_acc, n);
      float ms = 0.f;
      if (rc == FDG_OK && (hipEventRecord(ev1, nullptr) != hipSuccess || hipEventSynchronize(ev1) != hipSuccess || hipEventElapsedTime(&ms, ev0, ev1) != hipSuccess)) rc = FDG_E_NO_DEVICE;
      if (r > 0 && (best == 0.f || ms < best)) best = ms;
    }
    (void)hipFree(d_acc);
    gbs = best > 0.f ? 8.0 * (double)(L + R) * (double)n / ((double)best * 1e6) : 0.0;
    return rc;
  }
  // Wait until the device is quiet.  The driver wipes released memory in the background at 18-28 GB/s, and that write stream depresses every
  // rate by 2-7 % while it lasts (profiles/r05_log_pair_alloc_settle.txt: 1-4.5 s after 30-90 GB were released, then the batch runs at its
  // own rate).  First the time the wipe of `released` bytes takes at 16 GB/s, counted from `since`; then the fused accumulation over `n`
  // samples of the leaves (read-only, ~10 ms for the headline batch), launch after launch, until eight samples in a row agree to 1.5 % (at most 6 s more).
  int settle(const void *leaf, int64_t n, size_t released, std::chrono::steady_clock::time_point since, double *waited) {
    const auto t0 = std::chrono::steady_clock::now();
    const double need = (double)released / 16e9 - std::chrono::duration<double>(t0 - since).count();
    if (need > 0) std::this_thread::sleep_for(std::chrono::duration<double>(need));
    double *d_acc = nullptr;
    if (hipMalloc((void **)&d_acc, sizeof(double) * std::max<uint32_t>(R, 1)) != hipSuccess) return FDG_E_NOMEM;
    (void)hipMemset(d_acc, 0, sizeof(double) * R);
    const auto t1 = std::chrono::steady_clock::now();
    // One sample = four launches back to back, the time of the last one (no pauses between samples either: a device left idle for tens of
    // milliseconds answers the next launch at other clocks, and the samples would never agree).
    double hist[8] = {0};
    int rc = FDG_OK;
    for (int k = 0; rc == FDG_OK; ++k) {
      float ms = 0.f;
      for (int l = 0; l < 4 && rc == FDG_OK; ++l) {
        if (l == 3 && hipEventRecord(ev0, nullptr) != hipSuccess) { rc = FDG_E_NO_DEVICE; break; }
        rc = accumulate(leaf, d_acc, n);
      }
      if (rc) break;
      if (hipEventRecord(ev1, nullptr) != hipSuccess || hipEventSynchronize(ev1) != hipSuccess || hipEventElapsedTime(&ms, ev0, ev1) != hipSuccess) { rc = FDG_E_NO_DEVICE; break; }
      hist[k % 8] = ms > 0.f ? (double)n / (double)ms : 0.0;
      if (k >= 7) {
        double mx = 0, mn = 1e300;
        for (double x : hist) { mx = std::max(mx, x); mn = std::min(mn, x); }
        if (mn > 0.985 * mx) break;
      }
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count() > 6.0) break;
    }
    if (waited) *waited += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    (void)hipFree(d_acc);
    return rc;
  }
};
size_t gcd_sz(size_t a, size_t b) { while (b) { const size_t t = a % b; a = b; b = t; } return a; }
}  // namespace

// The search for a fast root chunk behind every leaf window, without a line of HIP: what it needs of the device are three hooks (PairAllocator
// below gives them; SimulatedPairs, at the end of this file, gives them from a model of the memory, so that the search is tested on the CPU).
//
// Nothing is assumed about how many kinds of memory there are or how they interact (measured: the rate of a pair takes three levels, as if a
// region carried two bits and every bit in which leaves and roots DIFFER bought 5 %; attempts to infer classes from a few reference probes
// were at the mercy of the probes' noise: profiles/r05_log_pair_alloc_v[2-5].txt).  PAIRS are measured.
namespace {
struct PairSearch {
  static constexpr double kFastLevel = 0.965;      // "at the fast level": within 3.5 % of the best pair seen so far
  std::vector<int> pick;                 // [window] the candidate chosen for it (-1: none yet)
  std::vector<double> got, pred;         // [window] the pair's rate when it was chosen; [candidate] its rate in its most recent probe
  std::vector<char> used_;               // [candidate]
  double fast = 0, slow = 0;             // best / worst pair seen
  size_t n_full_scan = 0;
  virtual ~PairSearch() {}
  virtual size_t n_cand() const = 0;
  virtual int probe(size_t window, size_t candidate, double &rate) = 0;
  virtual bool draw_more() = 0;          // more candidates, from other regions of the memory; false: the budget is spent
  bool is_used(size_t j) const { return j < used_.size() && used_[j]; }
  void set_used(size_t j, bool u) { if (used_.size() <= j) used_.resize(j + 1, 0); used_[j] = u; }

  int probe_pair(size_t i, size_t j, double &r) {
    const int rc = probe(i, j, r);
    if (rc) return rc;
    if (pred.size() < n_cand()) pred.resize(n_cand(), 0.0);
    pred[j] = r;
    fast = std::max(fast, r);
    slow = slow == 0 ? r : std::min(slow, r);
    return FDG_OK;
  }
  // every unused candidate from `from` on behind window i: the best one
  int full_scan(size_t i, size_t from, int &bj, double &br) {
    for (size_t j = from; j < n_cand(); ++j) {
      if (is_used(j)) continue;
      double r; const int rc = probe_pair(i, j, r); if (rc) return rc;
      if (r > br) { br = r; bj = (int)j; }
    }
    return FDG_OK;
  }
  // every unused candidate behind window i; while even the best of them is below the fast level, more candidates are drawn
  int scan_and_draw(size_t i, int &bj, double &br) {
    ++n_full_scan;
    int rc = full_scan(i, 0, bj, br); if (rc) return rc;
    while (br < kFastLevel * fast) {
      const size_t from = n_cand();
      if (!draw_more()) break;
      pred.resize(n_cand(), 0.0);
      rc = full_scan(i, from, bj, br); if (rc) return rc;
    }
    return FDG_OK;
  }
  // A window first tries the unused candidates that ran fastest behind the previous window (neighbouring windows mostly lie in one region);
  // when three of them disappoint it times every unused candidate, and draws more when even the best of those is below the fast level.
  int place(size_t i) {
    pred.resize(n_cand(), 0.0);
    std::vector<size_t> order;
    for (size_t j = 0; j < n_cand(); ++j) if (!is_used(j) && pred[j] > 0) order.push_back(j);
    std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return pred[x] > pred[y]; });
    int bj = -1; double br = 0;
    for (size_t q = 0; q < std::min<size_t>(order.size(), 3); ++q) {
      double r; const int rc = probe_pair(i, order[q], r); if (rc) return rc;
      if (r > br) { br = r; bj = (int)order[q]; }
      if (r >= kFastLevel * fast) break;
    }
    if (!(bj >= 0 && br >= kFastLevel * fast)) { const int rc = scan_and_draw(i, bj, br); if (rc) return rc; }
    if (bj >= 0) { pick[i] = bj; set_used((size_t)bj, true); got[i] = br; }
    return FDG_OK;
  }
  // The fast level is "the best pair seen" -- but what if no pair of the best kind is among the candidates at all (the leaves in regions of two
  // neighbouring kinds, the candidates sprinkled over the same regions: every pair at the middle level at best)?  `explore_above` (0: off) is
  // the rate a pair of the best kind should reach, estimated by the caller from the read-only rate of a window; while candidates DO differ
  // (there is something to choose) and the best pair seen stays below it, the first window keeps drawing -- at most `explore_draws` times.
  double explore_above = 0;
  int explore_draws = 24;
  int run_search(size_t n_window) {
    pick.assign(n_window, -1); got.assign(n_window, 0.0);
    for (size_t i = 0; i < n_window; ++i) {
      int rc = place(i); if (rc) return rc;
      if (i == 0 && explore_above > 0) {
        int bj = pick[0]; double br = got[0];
        for (int d = 0; d < explore_draws && fast > 1.03 * slow && fast < explore_above; ++d) {
          const size_t from = n_cand();
          if (!draw_more()) break;
          pred.resize(n_cand(), 0.0);
          rc = full_scan(0, from, bj, br); if (rc) return rc;
        }
        if (bj >= 0 && bj != pick[0] && br > 1.01 * got[0]) { set_used((size_t)pick[0], false); pick[0] = bj; set_used((size_t)bj, true); got[0] = br; }
      }
    }
    // second pass: the level rose while the search went on -- windows that were content with less look again, and draw more candidates if
    // need be (a window's kind of partner may not have been among the candidates when it was placed: one bench process of twenty-odd ended
    // with half its windows at the middle level that way, profiles/r05_k2_bench_line_second_pass_bug.json)
    for (size_t i = 0; i < n_window; ++i) {
      if (pick[i] < 0 || got[i] >= kFastLevel * fast) continue;
      int bj = -1; double br = got[i];
      const int rc = scan_and_draw(i, bj, br); if (rc) return rc;
      if (bj >= 0 && br > 1.01 * got[i]) { set_used((size_t)pick[i], false); pick[i] = bj; set_used((size_t)bj, true); got[i] = br; }
    }
    return FDG_OK;
  }
};
}  // namespace

// The allocation as an object: geometry, the address ranges, the candidates; one method per phase, abort() undoes whatever has been done.
namespace {
struct PairAllocator : PairSearch {
  struct Phys { hipMemGenericAllocationHandle_t h; bool mapped = false; };
  // request
  fdg_graph *g; unsigned flags; int dev = 0;
  // geometry
  size_t gran = 0, chunk_tiles = 0, leaf_chunk = 0, root_chunk = 0, n_chunk = 0, max_cand = 0, filler_bytes = 0;
  bool calibrate = false;
  // state
  hipMemAllocationProp prop = {};
  hipMemAccessDesc acc = {};
  char *leaf_va = nullptr, *root_va = nullptr, *cand_va = nullptr;
  std::vector<Phys> cand, filler;
  std::vector<char> root_mapped;
  PairCtx cx;
  // what the search saw
  double settle_s = 0;
  std::vector<double> before;             // [window] pair in draw order
  size_t n_filler_total = 0;
  bool span_cut = false;                  // the sprinkle ended before its 80 GB: free memory (less what the batch itself still needs) ran out
  bool leaves_retried = false;            // hipMalloc(leaves) failed with the candidates of the sprinkle held: they were given back and the batch mapped uncalibrated
  // what the batch itself still has to allocate: never promised to a filler or to a surplus candidate
  size_t still_needed() const {
    size_t need = leaf_va ? 0 : n_chunk * leaf_chunk;
    if (cand.size() < n_chunk) need += (n_chunk - cand.size()) * root_chunk;
    return need;
  }
  static size_t free_bytes() { size_t f = 0, t = 0; return hipMemGetInfo(&f, &t) == hipSuccess ? f : 0; }
  std::chrono::steady_clock::time_point t_released = std::chrono::steady_clock::now();
  bool verbose() const { return (flags & FDG_BATCH_PAIR_VERBOSE) != 0; }
  int64_t n_all() const { return (int64_t)(n_chunk * chunk_tiles) * 64; }
  char *leaf_at(size_t i) const { return leaf_va + i * leaf_chunk; }
  char *cand_at(size_t j) const { return cand_va + j * root_chunk; }
  char *root_at(size_t i) const { return root_va + i * root_chunk; }

  int hip_fail(const char *what, hipError_t e) { abort(); return fail(what, e); }
  void abort() {
    (void)hipDeviceSynchronize();
    for (size_t i = 0; i < root_mapped.size(); ++i) if (root_mapped[i]) (void)hipMemUnmap(root_at(i), root_chunk);
    for (size_t j = 0; j < cand.size(); ++j) { if (cand[j].mapped) (void)hipMemUnmap(cand_at(j), root_chunk); (void)hipMemRelease(cand[j].h); }
    for (Phys &f : filler) (void)hipMemRelease(f.h);
    cand.clear(); filler.clear(); root_mapped.assign(root_mapped.size(), 0);
    if (leaf_va) (void)hipFree(leaf_va);
    if (root_va) (void)hipMemAddressFree(root_va, n_chunk * root_chunk);
    if (cand_va) (void)hipMemAddressFree(cand_va, max_cand * root_chunk);
    leaf_va = root_va = cand_va = nullptr;
    if (cx.ev0) (void)hipEventDestroy(cx.ev0);
    if (cx.ev1) (void)hipEventDestroy(cx.ev1);
    cx.ev0 = cx.ev1 = nullptr;
  }

  // The LEAVES are one plain allocation (large physically contiguous pieces: what streams fastest, and what the driver gives a hipMalloc of
  // tens of GB; 1 GB physical chunks mapped one by one lose 5 % of the pure read rate, profiles/r05_log_pair_alloc_v1_vmm_leaves.txt); a
  // "chunk" of leaves is a window of it.  The ROOTS are mapped chunk by chunk: a chunk holds the roots of the tiles of one leaf window and
  // is a whole number of mapping granules.
  int geometry(int64_t n_sample, size_t chunk_bytes_hint) {
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail("hipGetDevice", e);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = dev;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
    if (e != hipSuccess || gran == 0) { e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum); if (e != hipSuccess) return fail("hipMemGetAllocationGranularity", e); }
    if (gran == 0) gran = (size_t)1 << 21;
    const uint32_t L = g->prog.L, R = g->prog.R;
    const size_t lt = 512u * (size_t)L, rt = 512u * (size_t)R;                 // bytes of one tile
    const size_t unit = gran / gcd_sz(gran, rt);                                // tiles per root granule
    const size_t T = (size_t)((n_sample + 63) / 64);
    const size_t hint = chunk_bytes_hint ? chunk_bytes_hint : ((size_t)2 << 30);
    size_t k = std::max<size_t>(1, (hint + unit * lt / 2) / (unit * lt));
    k = std::min(k, std::max<size_t>(1, (T + unit - 1) / unit));               // a small batch: one chunk
    chunk_tiles = unit * k; leaf_chunk = chunk_tiles * lt; root_chunk = chunk_tiles * rt;
    n_chunk = (T + chunk_tiles - 1) / chunk_tiles;
    if (flags & FDG_BATCH_PAIR_LEAF_MAJOR) {
      // a Julia column-major pair: the leaves of any window of samples are L pieces spread over the whole matrix, so there is ONE window -- the
      // batch -- and the candidates are whole root matrices.  Pays while the leaf matrix lies in one or two regions of the memory (a few tens of
      // GB); a 70 GB matrix spans every kind and no root matrix suits all of it.
      chunk_tiles = (T + unit - 1) / unit * unit; n_chunk = 1;
      leaf_chunk = chunk_tiles * lt; root_chunk = chunk_tiles * rt;
      cx.ld = (int64_t)chunk_tiles * 64;
    }
    max_cand = 8 * n_chunk + 64;
    filler_bytes = ((((size_t)2 << 30) + gran - 1) / gran) * gran;
    // (a window of less than ~0.5 GB of leaves is evaluated in too short a launch for the levels to separate: nothing to calibrate on)
    calibrate = (flags & FDG_BATCH_PAIR_CALIBRATE) && leaf_chunk >= ((size_t)400 << 20);
    if ((flags & FDG_BATCH_PAIR_LEAF_MAJOR) && root_chunk > ((size_t)1 << 30)) calibrate = false;      // (see above: nothing to choose for a matrix that large)
    pick.assign(n_chunk, -1); root_mapped.assign(n_chunk, 0); before.assign(n_chunk, 0.0);
    cx.row_major = (flags & FDG_BATCH_PAIR_ROW_MAJOR) != 0;
    cx.g = g; cx.L = L; cx.R = R; cx.chunk_tiles = chunk_tiles; cx.leaf_chunk = leaf_chunk; cx.root_chunk = root_chunk;
    return FDG_OK;
  }

  // `surplus`: a candidate beyond the n_chunk the batch needs -- only while the device keeps what the batch still has to allocate plus 4 GB free
  hipError_t new_cand(bool surplus = false) {
    if (cand.size() >= max_cand) return hipErrorOutOfMemory;
    if (surplus && free_bytes() < root_chunk + still_needed() + ((size_t)4 << 30)) return hipErrorOutOfMemory;
    Phys c;
    hipError_t e = hipMemCreate(&c.h, root_chunk, &prop, 0);
    if (e != hipSuccess) return e;
    const size_t j = cand.size();
    e = hipMemMap(cand_at(j), root_chunk, 0, c.h, 0);
    if (e != hipSuccess) { (void)hipMemRelease(c.h); return e; }
    e = hipMemSetAccess(cand_at(j), root_chunk, &acc, 1);
    if (e != hipSuccess) { (void)hipMemUnmap(cand_at(j), root_chunk); (void)hipMemRelease(c.h); return e; }
    c.mapped = true;
    cand.push_back(c);
    return hipSuccess;
  }
  // a 2 GB filler: moves the driver's allocator on to other regions of the memory (`reserve`: what must stay free besides)
  bool new_filler(size_t reserve) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < filler_bytes + reserve + still_needed() + ((size_t)8 << 30)) return false;
    Phys f;
    if (hipMemCreate(&f.h, filler_bytes, &prop, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
    filler.push_back(f);
    return true;
  }

  // Address ranges; root candidates from a wide span of the memory -- a candidate pair after every 2 GB filler over 80 GB: the kinds of memory
  // alternate in runs of 16-32 GB (profiles/r05_log_chunk_probe.txt), so every kind is among them --; the fillers go back to the driver BEFORE
  // anything is timed (its background wipe of released memory disturbs the evaluation exactly as the root writes do:
  // profiles/r05_log_pair_alloc_release.txt); then the leaves, into the space the fillers held (allocated before the sprinkle instead, the
  // headline ran 0.814-0.818 against 0.816-0.828 in eight alternating processes: profiles/r05_b_*).
  int draw() {
    hipError_t e;
    if ((e = hipMemAddressReserve((void **)&root_va, n_chunk * root_chunk, gran, nullptr, 0)) != hipSuccess) return hip_fail("hipMemAddressReserve(roots)", e);
    if ((e = hipMemAddressReserve((void **)&cand_va, max_cand * root_chunk, gran, nullptr, 0)) != hipSuccess) return hip_fail("hipMemAddressReserve(root candidates)", e);
    if ((e = hipEventCreate(&cx.ev0)) != hipSuccess || (e = hipEventCreate(&cx.ev1)) != hipSuccess) return hip_fail("hipEventCreate", e);
    if (calibrate) {
      const size_t span = (size_t)80 << 30;
      // (the sprinkle takes only what the device can spare -- a filler or a surplus candidate that does not fit ends it, it is not
      //  an error: a batch next to other allocations gets a shorter span and the best pairs found within it)
      for (size_t q = 0; q * filler_bytes < span && cand.size() + 2 <= max_cand / 2; ++q) {
        if (!new_filler((size_t)8 << 30)) { span_cut = true; break; }
        bool both = true;
        for (int c = 0; c < 2 && both; ++c) if (new_cand(cand.size() >= n_chunk) != hipSuccess) { (void)hipGetLastError(); both = false; }
        if (!both) { span_cut = true; break; }
      }
      n_filler_total = filler.size();
      for (Phys &f : filler) (void)hipMemRelease(f.h);
      filler.clear();
      t_released = std::chrono::steady_clock::now();
    }
    if ((e = hipMalloc((void **)&leaf_va, n_chunk * leaf_chunk)) != hipSuccess) {
      // the leaves do not fit next to what the sprinkle holds (fragmentation, or another process took memory meanwhile): everything drawn
      // goes back, the leaves are allocated first and the roots mapped in draw order -- a plain batch instead of no batch
      (void)hipGetLastError();
      leaf_va = nullptr;
      for (size_t j = 0; j < cand.size(); ++j) { if (cand[j].mapped) (void)hipMemUnmap(cand_at(j), root_chunk); (void)hipMemRelease(cand[j].h); }
      cand.clear();
      leaves_retried = true; calibrate = false;
      (void)hipDeviceSynchronize();
      if ((e = hipMalloc((void **)&leaf_va, n_chunk * leaf_chunk)) != hipSuccess) { leaf_va = nullptr; return hip_fail("hipMalloc(leaves)", e); }
    }
    while (cand.size() < n_chunk) if ((e = new_cand()) != hipSuccess) return hip_fail("hipMemCreate(root candidate)", e);
    return FDG_OK;
  }

  // ---- PairSearch's view of the device -------------------------------------------------------------------------------------------
  size_t n_cand() const override { return cand.size(); }
  int probe(size_t i, size_t j, double &r) override { return cx.probe(leaf_at(i), cand_at(j), r); }
  // two more candidates behind a filler that moves the driver on to other regions of the memory; false: the budget is spent
  bool draw_more() override {
    const size_t filler_budget = (size_t)144 << 30;
    if (!(filler.size() * filler_bytes < filler_budget && cand.size() + 2 <= max_cand)) return false;
    if (!new_filler((size_t)8 << 30)) { span_cut = true; return false; }
    for (int c = 0; c < 2; ++c) if (new_cand(true) != hipSuccess) { (void)hipGetLastError(); span_cut = true; return c > 0; }      // (one more candidate is still one more)
    return true;
  }

  // fill, wait for the device to be quiet, then PairSearch::run_search
  int search() {
    // the probes must see what the workload will see: uniform random leaves (a window of zeros or of stale data runs at another clock
    // and another rate than its neighbours: profiles/r05_log_pair_alloc_v3.txt, rounds 1-2)
    int rc = cx.fill(leaf_va, n_all());
    if (rc) return rc;
    // the fillers of the sprinkle were released a moment ago: wait until their wipe is over before anything is timed
    rc = cx.settle(leaf_va, n_all(), n_filler_total * filler_bytes, t_released, &settle_s);
    if (rc) return rc;
    // the pairs an uncalibrated mapping would make (chunk i with the i-th candidate drawn)
    for (size_t i = 0; i < n_chunk; ++i) { rc = cx.probe(leaf_at(i), cand_at(i), before[i]); if (rc) return rc; }
    // what a pair of the best kind should reach: the read-only rate of a window, with a written byte costing 2.7 read bytes (measured on the
    // headline: a written byte costs 1.9 / 3.5 / 4.6 read bytes at the three levels: 0.855 / 0.80 / 0.765 against 0.89 read-only)
    {
      double ro = 0;
      rc = cx.probe_readonly(leaf_at(0), ro); if (rc) return rc;
      explore_above = ro > 0 ? ro / (1.0 + 2.7 * (double)cx.R / (double)cx.L) : 0.0;
    }
    rc = run_search(n_chunk); if (rc) return rc;
    if (verbose()) {
      std::fprintf(stderr, "[fdg_batch_alloc_pair] pairs when chosen, GB/s:");
      for (size_t i = 0; i < n_chunk; ++i) std::fprintf(stderr, " %.0f", got[i]);
      std::fprintf(stderr, "\n[fdg_batch_alloc_pair] chosen candidate of each window:");
      for (size_t i = 0; i < n_chunk; ++i) std::fprintf(stderr, " %d", pick[i]);
      std::fprintf(stderr, "\n[fdg_batch_alloc_pair] best / worst pair seen %.0f / %.0f GB/s; %zu windows, %zu candidates, %zu + %zu fillers, %zu full scans, %u probes\n",
                   fast, slow, n_chunk, cand.size(), n_filler_total, filler.size(), n_full_scan, cx.n_probe);
    }
    return FDG_OK;
  }

  // the chosen candidates behind their leaf windows (windows without one -- calibration off, candidates ran out -- take what is left)
  int map_chosen() {
    hipError_t e;
    size_t j = 0;
    for (size_t i = 0; i < n_chunk; ++i) {
      if (pick[i] >= 0) continue;
      while (j < cand.size() && is_used(j)) ++j;
      if (j >= cand.size() && (e = new_cand()) != hipSuccess) return hip_fail("hipMemCreate(root chunk)", e);
      pick[i] = (int)j; set_used(j, true);
    }
    if ((e = hipDeviceSynchronize()) != hipSuccess) return hip_fail("hipDeviceSynchronize", e);
    for (size_t c = 0; c < cand.size(); ++c) if (cand[c].mapped) { (void)hipMemUnmap(cand_at(c), root_chunk); cand[c].mapped = false; }
    for (size_t i = 0; i < n_chunk; ++i) {
      if ((e = hipMemMap(root_at(i), root_chunk, 0, cand[(size_t)pick[i]].h, 0)) != hipSuccess) return hip_fail("hipMemMap(roots)", e);
      root_mapped[i] = 1;
      if ((e = hipMemSetAccess(root_at(i), root_chunk, &acc, 1)) != hipSuccess) return hip_fail("hipMemSetAccess(roots)", e);     // (chunk by chunk, as the candidates were when they were timed)
    }
    return FDG_OK;
  }

  // the mapped pairs once more (BEFORE anything else is released: the wipe of what is released would disturb it), then everything that was not
  // used goes back to the driver, and the batch is handed over when the device is quiet again
  int verify_and_release(fdg_batch_pair_info *info) {
    double after_mean = 0, after_min = 0, before_mean = 0, before_min = 0;
    uint32_t n_matched = 0;
    int rc = FDG_OK;
    if (calibrate) {
      for (size_t i = 0; i < n_chunk && !rc; ++i) {
        double r = 0;
        rc = cx.probe(leaf_at(i), root_at(i), r);
        after_mean += r / (double)n_chunk; after_min = (i == 0 || r < after_min) ? r : after_min;
        before_mean += before[i] / (double)n_chunk; before_min = (i == 0 || before[i] < before_min) ? before[i] : before_min;
        if (r >= 0.95 * fast) ++n_matched;
        if (verbose()) std::fprintf(stderr, "%s%.0f", i ? " " : "[fdg_batch_alloc_pair] pairs as mapped, GB/s: ", r);
      }
      if (verbose()) std::fputc('\n', stderr);
    }
    size_t released = filler.size() * filler_bytes;
    const size_t n_filler = filler.size() + n_filler_total, n_cand = cand.size();
    for (size_t c = 0; c < cand.size(); ++c) if (!is_used(c)) { (void)hipMemRelease(cand[c].h); released += root_chunk; }
    for (Phys &f : filler) (void)hipMemRelease(f.h);
    filler.clear();
    t_released = std::chrono::steady_clock::now();
    (void)hipMemAddressFree(cand_va, max_cand * root_chunk);
    cand_va = nullptr;
    if (calibrate && !rc) rc = cx.settle(leaf_va, n_all(), released, t_released, &settle_s);
    if (rc) {      // undo the final mapping (the unused candidates are gone already)
      std::vector<Phys> kept;
      for (size_t i = 0; i < n_chunk; ++i) kept.push_back(cand[(size_t)pick[i]]);
      cand.swap(kept);
      for (Phys &c : cand) c.mapped = false;
      abort();
      return rc;
    }
    (void)hipEventDestroy(cx.ev0); (void)hipEventDestroy(cx.ev1);
    cx.ev0 = cx.ev1 = nullptr;
    if (info) {
      info->leaf_bytes = n_chunk * leaf_chunk; info->root_bytes = n_chunk * root_chunk; info->chunk_tiles = chunk_tiles;
      info->n_chunk = (uint32_t)n_chunk; info->n_candidate = (uint32_t)n_cand; info->n_filler = (uint32_t)n_filler; info->n_probe = cx.n_probe;
      info->n_matched = n_matched; info->calibrated = fast > 1.05 * slow ? 1u : 0u;
      info->gbs_fast = fast; info->gbs_slow = slow;
      info->gbs_before_mean = before_mean; info->gbs_before_min = before_min; info->gbs_after_mean = after_mean; info->gbs_after_min = after_min;
      info->seconds_settling = settle_s;
      info->span_gb = (uint32_t)((n_filler * filler_bytes) >> 30);
      // 0: no calibration asked for (or the window too small to time).  1: asked for, but the batch was mapped in draw order: the leaves only fitted
      // once the candidates were given back.  2: calibrated over a span cut short by the free memory (the best pairs found within it).  3: the full search.
      info->level_reached = !(flags & FDG_BATCH_PAIR_CALIBRATE) ? 0u : (leaves_retried ? 1u : (!calibrate ? 0u : (span_cut ? 2u : 3u)));
    }
    return FDG_OK;
  }
};
}  // namespace

int fdg_batch_alloc_pair(fdg_graph *g, int64_t n_sample, size_t chunk_bytes_hint, unsigned flags, void **d_leaf, void **d_root,
                         fdg_batch_pair_info *info) {
  if (!g || !d_leaf || !d_root) { fdg::set_error("null argument"); return FDG_E_INVALID; }
  *d_leaf = *d_root = nullptr;
  if (info) std::memset(info, 0, sizeof *info);
  if (n_sample <= 0) { fdg::set_error("empty batch"); return FDG_E_INVALID; }
  if (g->prog.L == 0 || g->prog.R == 0) { fdg::set_error("fdg_batch_alloc_pair: the graph has no leaves or no roots"); return FDG_E_INVALID; }
  if (!(g->isa && !g->code_object.empty())) { fdg::set_error("fdg_batch_alloc_pair: the pairs are timed with the handle's own kernels: it must be specialised with FDG_SPEC_ISA first"); return FDG_E_UNSUPPORTED; }
  const auto t_start = std::chrono::steady_clock::now();
  PairAllocator A;
  A.g = g; A.flags = flags;
  int rc = A.geometry(n_sample, chunk_bytes_hint);
  if (rc) return rc;
  rc = A.draw();
  if (rc) return rc;
  if (A.calibrate) { rc = A.search(); if (rc) { A.abort(); return rc; } }
  rc = A.map_chosen();
  if (rc) return rc;
  rc = A.verify_and_release(info);
  if (rc) return rc;
  {
    Batch bl, br;
    bl.bytes = A.n_chunk * A.leaf_chunk; bl.chunk = 0; bl.device = A.dev;          // chunk 0: a plain allocation (hipFree)
    br.bytes = A.n_chunk * A.root_chunk; br.chunk = A.root_chunk; br.device = A.dev;
    for (size_t i = 0; i < A.n_chunk; ++i) br.handles.push_back(A.cand[(size_t)A.pick[i]].h);
    std::lock_guard<std::mutex> lk(g_mu);
    g_batches[A.leaf_va] = std::move(bl);
    g_batches[A.root_va] = std::move(br);
  }
  *d_leaf = A.leaf_va; *d_root = A.root_va;
  if (info) info->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
  return FDG_OK;
}


// ---- the search on a MODEL of the memory (CPU test of PairSearch; tests/test_batch_alloc.py) ------------------------------------------------
// Regions carry two bits; a (window, candidate) pair runs at 0.855 / 0.80 / 0.765 of 8 TB/s when the two kinds differ in two / one / no bit,
// with multiplicative noise.  Windows: runs of one kind.  Candidates come in draw order from a walk through regions; draw_more() continues the walk.
namespace {
struct SimulatedPairs : PairSearch {
  std::vector<int> window_kind, cand_kind, walk;       // walk: kinds of the regions further draws come from, 8 draws per region
  size_t walk_pos = 0, n_probe = 0, max_draws = 64, n_draws = 0;
  double noise = 0.004;
  uint64_t rng;
  double unit() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (double)(rng >> 11) / 9007199254740992.0; }
  size_t n_cand() const override { return cand_kind.size(); }
  int probe(size_t i, size_t j, double &r) override {
    const int d = __builtin_popcount((unsigned)(window_kind[i] ^ cand_kind[j]) & 3u);
    const double level = d == 2 ? 0.855 : (d == 1 ? 0.80 : 0.765);
    r = 8000.0 * level * (1.0 + noise * (2.0 * unit() - 1.0));
    ++n_probe;
    return FDG_OK;
  }
  bool draw_more() override {
    if (n_draws >= max_draws) return false;
    ++n_draws;
    for (int c = 0; c < 2; ++c) { cand_kind.push_back(walk[(walk_pos / 8) % walk.size()]); ++walk_pos; }
    return true;
  }
};
}  // namespace

// scenario 0: every kind among the first candidates.  1: the windows' complements only behind further draws.  2: the FIRST windows' complement
// absent at first, the later windows' present (the case that left a bench process at 0.749 before the second pass could draw).  3: as 0 with
// 1.5 % noise.  Returns the number of windows whose chosen candidate is of the complementary kind; *n_probe: pairs "timed".
int fdg_selftest_pair_search(uint64_t seed, uint32_t scenario, uint32_t n_window, uint32_t *n_probe) {
  SimulatedPairs S;
  S.rng = seed * 0x9E3779B97F4A7C15ull + 0x1234567ull;
  const int kA = (int)(seed & 3), kB = kA ^ 1;                          // the two kinds the windows are made of
  for (uint32_t i = 0; i < n_window; ++i) S.window_kind.push_back(i < n_window / 2 ? kA : kB);
  auto push = [&](int k, size_t n) { for (size_t q = 0; q < n; ++q) S.cand_kind.push_back(k); };
  switch (scenario) {
    case 0: case 3: for (int rep = 0; rep < 5; ++rep) for (int k = 0; k < 4; ++k) push(k, 4); break;
    case 1: for (int rep = 0; rep < 5; ++rep) { push(kA, 8); push(kB, 8); } break;
    default: for (int rep = 0; rep < 5; ++rep) { push(kA, 6); push(kB, 6); push(kB ^ 3, 4); } break;      // kB's complement present, kA's not
  }
  if (scenario == 3) S.noise = 0.015;
  S.explore_above = 8000.0 * 0.83;                                      // between the middle and the top level (the allocator estimates it from the read-only rate)
  S.walk = {kA, kB, kA ^ 3, kB ^ 3};                                     // further draws: first more of the same, then the complements
  const int rc = S.run_search(n_window);
  if (rc) return -1;
  if (n_probe) *n_probe = (uint32_t)S.n_probe;
  int top = 0;
  for (uint32_t i = 0; i < n_window; ++i) if (S.pick[i] >= 0 && ((S.window_kind[i] ^ S.cand_kind[(size_t)S.pick[i]]) & 3) == 3) ++top;
  return top;
}

}  // extern "C"
