// Internal structures shared by the host-side lowering (fdg_lower.cpp), the
// source emitter (fdg_emit.cpp) and the device runtime (fdg_runtime.hip).
#pragma once
#include <cstdint>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/fdg.h"
#include "fdg_knobs.h"

namespace fdg {

// ---- operand / destination location words of the interpreter stream --------
// bits 31..30 : space   0 = LDS slot, 1 = workspace (HBM panel) slot, 2 = leaf
// bit  29     : operand carries a factor (two more words: lo, hi of the f64)
// bits 28..0  : index
constexpr uint32_t SP_LDS = 0u, SP_MEM = 1u, SP_LEAF = 2u;
constexpr uint32_t LOC_FAC = 1u << 29;
constexpr uint32_t LOC_IDX_MASK = (1u << 29) - 1u;
inline uint32_t mkloc(uint32_t space, uint32_t idx) { return (space << 30) | idx; }

// ---- micro-op header word ---------------------------------------------------
// bits 3..0  : opcode;  bits 31..4 : operand count (SUM/PROD), exponent (POW,
//              signed, stored biased by 2^27), root index (ROOT)
constexpr uint32_t UOP_SUM = 0, UOP_PROD = 1, UOP_POW = 2, UOP_ROOT = 3, UOP_LEAF = 4, UOP_END = 5, UOP_PRODI = 6;   // PRODI: eval!'s product, operands scaled first
// layouts (words):
//   SUM/PROD/PRODI : hdr, dst, then per operand: loc [, fac_lo, fac_hi]
//   POW      : hdr(exponent), dst, loc [, fac_lo, fac_hi]
//   ROOT     : hdr(k), loc                      -- root[k] = value at loc
//   LEAF     : hdr(0), dst, leaf_index          -- copy a leaf into an LDS slot
//   END      : hdr

struct Lowered {
  // copy of the table
  uint32_t L = 0, N = 0, R = 0, E = 0;
  std::vector<uint8_t> op;
  std::vector<int32_t> power;
  std::vector<uint32_t> off, idx;
  std::vector<double> fac;
  std::vector<uint32_t> root_slot;

  std::vector<uint32_t> sched_group;   // optional [N]: producer's grouping hint for the scheduler
  bool assoc_interp = false;           // FDG_ASSOC_INTERP: the association of eval! (src/computational_graph/eval.jl:1-3) instead of the
                                       // generated code's: a Prod folds the already scaled operands, (w1 f1) * (w2 f2) * ...

  // analysis
  std::vector<uint8_t> live;       // [L+N] reachable from a root
  std::vector<uint32_t> order;     // live internal nodes (index into 0..N-1), evaluation order
  uint32_t n_live_leaf = 0;
  uint64_t flops_alg = 0;
  uint32_t max_live = 0;

  // interpreter program
  uint32_t lds_slots = 0;          // slots per sample in LDS
  uint32_t mem_slots = 0;          // slots per sample in the workspace panel
  std::vector<uint32_t> code;      // micro-op stream
  uint32_t n_ops = 0;
  uint64_t opnd_lds = 0, opnd_mem = 0, opnd_leaf = 0;  // operand reads by space (statistics)
};

// lowering entry points (fdg_lower.cpp)
int validate_desc(const fdg_graph_desc *d, std::string &err);
struct RealTwinTable { uint32_t n_leaf = 0; std::vector<uint8_t> op; std::vector<int32_t> power; std::vector<uint32_t> off, idx, root_slot; std::vector<double> fac; };
bool complex_to_real_table(const Lowered &p, RealTwinTable &out, std::string &why);
void analyse(Lowered &p);
void build_interpreter_program(Lowered &p, uint32_t lds_slot_budget);

// source emitter (fdg_emit.cpp)
std::string emit_hip_source(const Lowered &p, unsigned flags);
std::string emit_hip_source_typed(const Lowered &p, int dtype, bool &ok, std::string &why);
std::string emit_fused_source(const Lowered &p, const std::string &leaf_stmts, const std::string &device_functions);

// thread-local error
void set_error(const std::string &s);

double powi(double x, int32_t n);

}  // namespace fdg

// Device scratch of one caller stream.  The members of the same name in fdg_graph are the set bound to the stream of the
// call in progress (fdg_bind_stream_ws, under fdg_graph::mu); the sets of other streams are parked in fdg_graph::ws_pool.
// Kernels enqueued on different streams therefore never share a spill panel, partial sums or a staging buffer.
struct fdg_ws_set {
  void *key = nullptr;
  void *d_ws = nullptr, *d_ws2 = nullptr, *d_ws3 = nullptr, *d_ws4 = nullptr;
  size_t ws_bytes = 0, ws2_bytes = 0, ws3_bytes = 0, ws4_bytes = 0;
  void *s2 = nullptr, *ev_t[2] = {nullptr, nullptr}, *ev_k[2] = {nullptr, nullptr}, *ev_in = nullptr;
  uint64_t last_use = 0;
};

// What the launch path needs of a handle's options, parsed when the handle is created and whenever fdg_graph_set_option changes one: between an
// evaluation entry point and hipModuleLaunchKernel nothing is looked up by name.
struct fdg_launch_cfg {
  bool no_rl = false, no_fused_acc = false, no_pool = false, pool_no_acc = false, no_streaming = false, no_w2 = false, no_coop = false, no_rm = false;
  bool transpose_narrow = false;
  uint32_t root_scratch_min = 16;          // roots from which row-major root matrices go through the column-major scratch (0: never)
  uint64_t root_scratch_mb = 256;
  int waves_per_cu = 0;                    // > 0: resident waves per CU forced (no oversubscription)
  int oversub = 0;                         // > 0: oversubscription factor forced
  long mem_waves = -1;                     // >= 0: resident waves per CU of memory-bound graphs (0: rule off); -1: the library's rule
  long mem_oversub = 1;
  double mem_ratio = 2.5;
  uint64_t sm_chunk_bytes = 1ull << 29;    // row-major input without an in-place variant: bytes of leaves transposed per chunk
  long long eval_chunk = 0, mc_chunk = 0;  // host-buffer / Monte-Carlo chunk sizes in samples (0: the library's)
};

struct fdg_graph {
  fdg::Lowered prog;
  fdg::KnobMap knobs;              // this handle's options (starts as a copy of fdg::env_snapshot(); fdg_graph_set_option)
  fdg_launch_cfg cfg;
  // tuning knobs set by fdg_graph_set_opt_params (has_opt: the next FDG_SPEC_ISA specialisation uses them as they are)
  fdg_opt_params opt = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  bool has_opt = false;
  std::vector<fdg_ws_set> ws_pool;
  void *ws_key = nullptr;
  bool ws_bound = true;            // the members below start out as the set of the null stream
  uint64_t ws_clock = 0;
  // specialization
  std::vector<char> code_object;   // gfx950 code object of the specialized kernels
  std::string spec_source_hash;
  unsigned spec_flags = 0;
  uint32_t spec_vgpr = 0, spec_lds = 0, spec_scratch = 0;
  // device state (guarded by mu)
  std::mutex mu;
  void *d_code = nullptr;          // interpreter stream
  void *d_root_live = nullptr;     // uint8[R]: 0 where root_slot[k] == FDG_NO_ROOT (only allocated for such tables; root scratch paths skip those k)
  void *d_ws = nullptr;            // workspace panels + block partials
  size_t ws_bytes = 0;
  void *module = nullptr;          // hipModule_t
  void *fn_eval_sm = nullptr, *fn_eval_gen = nullptr;  // hipFunction_t
  // ISA specialization (fdg_isa.cpp): one wave = 64 samples, persistent grid
  bool isa = false;
  void *fn_isa = nullptr;
  void *fn_isa_nt = nullptr, *fn_isa_acc_nt = nullptr;   // streaming variants (non-temporal leaf loads / root stores) for line-aligned batches
  uint32_t isa_vgpr = 0, isa_lds_bytes = 0, isa_mem_slots = 0;
  // optional two-samples-per-lane variant in the same code object (sample stride 1, full 128-sample tiles)
  bool has_w2 = false;
  void *fn_isa_w2 = nullptr;
  uint32_t isa2_vgpr = 0, isa2_lds_bytes = 0, isa2_mem_slots = 0;
  // fused accumulate variant (per-lane accumulators in VGPRs, roots never written)
  bool has_acc = false;
  bool isa_fma = false;            // FDG_SPEC_FAST_MATH with FDG_SPEC_ISA: fused multiply-adds (not parity-exact)
  void *fn_isa_acc = nullptr;
  uint32_t isa3_vgpr = 0, isa3_lds_bytes = 0, isa3_mem_slots = 0;
  // row-major variant (leaf stride 1: compile_Python's [B, L]): chunks of rows staged through LDS inside the evaluator
  bool has_rm = false;
  void *fn_isa_rm = nullptr;
  uint32_t isa4_vgpr = 0, isa4_lds_bytes = 0, isa4_mem_slots = 0;
  // ... and its fused-accumulate form (row-major leaves, roots never written)
  bool has_rm_acc = false;
  void *fn_isa_rm_acc = nullptr;
  uint32_t isa5_vgpr = 0, isa5_lds_bytes = 0, isa5_mem_slots = 0;
  // what the installed programs execute per evaluation (fdg_graph_kernel_info): [0] eval, [1] accumulate, [2] row-major
  uint64_t st_valu[3] = {0, 0, 0};
  uint32_t st_ld_leaf[3] = {0, 0, 0}, st_panel[3] = {0, 0, 0}, st_lds[3] = {0, 0, 0};
  uint32_t rm_bufs = 0;
  const char *last_kernel = "";    // evaluator kernel of the last device call (guarded by mu)
  // cooperative variant: the four waves of a CU evaluate one tile together (graphs whose live set overflows one lane)
  bool has_coop = false, coop_enabled = false;
  void *fn_isa_coop = nullptr;
  uint32_t coop_panel_wg = 0, coop_lds_bytes = 0, coop_threads = 256;
  // linear row-major variant: contiguous rows, the tile's block streamed into an LDS image
  bool has_rl = false;
  void *fn_isa_rl = nullptr;
  uint32_t isa6_vgpr = 0, isa6_lds_bytes = 0, isa6_mem_slots = 0;
  bool has_rl_acc = false;           // the linear row-major variant with fused accumulation
  void *fn_isa_rl_acc = nullptr;
  uint32_t isa7_vgpr = 0, isa7_lds_bytes = 0, isa7_mem_slots = 0;
  uint64_t rl_valu = 0;
  // pooled cooperative variant: the waves of a CU evaluate one tile, whole roots each, leaves through a shared LDS pool (full tiles, sample stride 1)
  bool has_pool = false;
  void *fn_isa_pool = nullptr;
  uint32_t pool_panel_wg = 0, pool_threads = 256, pool_fetch = 0, pool_unit = 1;
  uint64_t pool_valu = 0;
  // companion HIP-source kernels of an ISA-specialised handle, used for sample-major input (FDG_SPEC_ROW_MAJOR_COMPANION)
  std::vector<char> alt_code;
  void *alt_module = nullptr, *fn_alt_sm = nullptr, *fn_alt_gen = nullptr;
  // fused Monte-Carlo step: leaves computed in registers from (K, T), then the graph (HIP-source JIT)
  std::vector<char> fused_code;
  void *fused_module = nullptr, *fn_fused = nullptr;
  // element types other than Float64 (fdg_graph_specialize_typed): one HIP-source kernel per type, [FDG_DT_*]
  std::vector<char> typed_code[4];
  void *typed_module[4] = {nullptr, nullptr, nullptr, nullptr}, *fn_typed[4] = {nullptr, nullptr, nullptr, nullptr};
  // ComplexF64 rows through the Float64 assembly back end: the graph spelled out on real and imaginary parts (owned; may be null)
  fdg_graph *cx_twin = nullptr;
  bool cx_twin_tried = false;
  // ... or, for graphs too large for a compiler-scheduled kernel, leaf kernel -> chunk of leaves -> this handle's evaluator
  int mc_route = 0;                // 0 none, 1 fused HIP kernel, 2 leaf kernel + evaluator, 3 fused ISA kernel
  std::vector<int32_t> lt_i32[5];  // copy of the leafstates tables (type, order, tau_in, tau_out, loop_index)
  std::vector<double> lt_basis;
  uint32_t lt_hdr[5] = {0, 0, 0, 0, 0};   // n_leaf, n_basis, n_loop, dim, n_tau
  // ... or (route 3) ONE kernel of the optimizing back end whose leaves are computed in registers from (K, T)
  std::vector<char> mc_code;
  void *mc_module = nullptr, *fn_mc = nullptr, *fn_mc_acc = nullptr;
  bool mc_has_acc = false, mc_built = false;
  uint32_t mc_vgpr[2] = {0, 0}, mc_lds[2] = {0, 0}, mc_mem[2] = {0, 0};   // [0] eval kernel, [1] accumulate kernel
  std::string mc_dir;
  unsigned mc_flags = 0;
  void *d_ws4 = nullptr;           // leaf-major chunk of leaves for route 2 / packed (K, T) columns for route 3
  size_t ws4_bytes = 0;
  void *d_ws2 = nullptr;           // roots scratch for accumulate through the ISA kernel
  size_t ws2_bytes = 0;
  void *d_ws3 = nullptr;           // leaf-major copy of a sample-major chunk for the ISA kernel
  size_t ws3_bytes = 0;
  void *s2 = nullptr;              // internal stream: transposition of the next chunk overlaps the evaluator
  void *ev_t[2] = {nullptr, nullptr}, *ev_k[2] = {nullptr, nullptr}, *ev_in = nullptr;
  int device = -1;
  int n_cu = 0;
};

// ---- shared by the runtime translation units (fdg_runtime.hip, fdg_leaf.hip) -------------------------
#ifdef FDG_RUNTIME_TU
#include <hip/hip_runtime.h>
#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) {                                                             \
      fdg::set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                     \
      return FDG_E_NO_DEVICE;                                                           \
    }                                                                                   \
  } while (0)

void parse_launch_cfg(fdg_graph *g);             // g->knobs -> g->cfg
int ensure_device(fdg_graph *g);                 // binds the handle to the current gfx950 device
int fdg_bind_stream_ws(fdg_graph *g, void *stream);   // makes the scratch set of `stream` the current one (caller holds g->mu)
int ensure_ws(fdg_graph *g, size_t bytes);       // grows the handle's device workspace
int fdg_run_locked(fdg_graph *g, int mode, const double *d_leaf, int64_t ss, int64_t ls, double *d_root, int64_t rs, int64_t rk,
                   const double *d_weight, double *d_acc, int64_t B, hipStream_t st,
                   int64_t lts = 0, int64_t rts = 0);   // caller holds g->mu; lts / rts != 0: tile strides of a tile-major batch (fdg.h)
int launch_reduce_partials(const double *partial, uint32_t nblk, uint32_t R, double *acc, hipStream_t st);
// Monte-Carlo step through one ISA kernel (fdg_runtime.hip); callers hold g->mu
bool fdg_mc_isa_supported(fdg_graph *g, const fdg_leaf_tables *tab, std::string &why, bool *recommended);
int fdg_mc_isa_build(fdg_graph *g);   // host-only (assembler); no-op when built
int fdg_mc_isa_run(fdg_graph *g, int mode, const double *d_K, int64_t ks, int64_t kc, const double *d_T, int64_t ts, int64_t tc,
                   double kF, double beta, double lambda, double *d_root, int64_t rs, int64_t rk, const double *d_weight,
                   double *d_acc, int64_t B, hipStream_t st);
#endif
namespace fdg { const char *last_error_cstr(); }
uint64_t fnv1a(const std::string &s, uint64_t h = 1469598103934665603ull);
int fdg_cache_dir(const char *arg, std::string &dir);   // resolves, creates (0700) and vets the JIT cache directory
bool read_file(const std::string &path, std::vector<char> &out);
bool read_cached(const std::string &dir, const std::string &fname, std::vector<char> &out);   // dir, then $FDG_CACHE_RO_DIR
bool write_file(const std::string &path, const char *data, size_t n);
int compile_hiprtc(const std::string &src, bool fast, std::vector<char> &co, std::string &log);
int compile_hipcc(const std::string &src_path, const std::string &out_path, bool fast, std::string &log);
