/*
 * fdg.h -- C ABI of the MI355X-native evaluator back end for
 * FeynmanDiagram.jl's static computational graphs.
 *
 * This is the drop-in boundary.  The reference has no
 * FFI for this path; each entry point below names the reference interface it
 * stands in for (paths relative to the reference checkout):
 *
 *   fdg_graph_create      <- Compilers.compile / to_julia_str
 *                            (src/backend/static.jl:98-133, 221-227): the host
 *                            (Julia shim or the Python mirror) walks the graphs
 *                            in the reference's order and hands over the flat
 *                            node table; this call plays the role of
 *                            Meta.parse + @RuntimeGeneratedFunction.
 *   fdg_graph_specialize  <- the JIT step of compile (static.jl:225-226): emits
 *                            a straight-line CDNA4 kernel for this one graph.
 *   fdg_eval              <- the generated eval_graph!(root, leafVal)
 *                            (static.jl:100,131) for B samples at once, in the
 *                            batched layout of compile_Python
 *                            (src/backend/compiler_python.jl:23,28,45-47).
 *   fdg_eval_device       <- same, buffers already resident in HBM.
 *   fdg_accumulate_device <- the user integrand's "sum weight*root over
 *                            samples" step around eval_graph!
 *                            (example/benchmark.jl:58-87), fused so roots never
 *                            travel to HBM.
 *   fdg_fill_uniform_device <- test/bench harness: counter-based leaf values on
 *                            device (the examples draw them from MCIntegration).
 *   fdg_graph_kernel_info <- no counterpart (the reference's evaluator is one Julia function): which hand-written
 *                            kernel ran and what it executes per evaluation, for the roofline in bench.py.
 *   fdg_graph_destroy, fdg_graph_query, fdg_last_error: lifetime / errors
 *                            (Julia exceptions in the reference, static.jl:6-11).
 *
 * Conventions: every function returns 0 on success and a negative FDG_E_* code
 * on failure; fdg_last_error() returns a thread-local message.  All buffers are
 * owned by the caller.  The program of a handle is immutable after create/specialize, and its
 * device scratch (spill panels, partial sums, staging buffers) is kept per caller stream, so the
 * device entry points may be called on one handle from several threads and on several streams at
 * once: calls on one stream run in stream order, calls on different streams may overlap on the
 * device.  (Enqueueing is serialised by a mutex inside the handle; up to 8 streams keep their
 * scratch, a ninth releases the least recently used set after a device synchronisation.  The first
 * call on a stream allocates; later ones only launch kernels, so they can be captured in a hipGraph.)
 * fdg_graph_specialize* and fdg_graph_release_device must not run concurrently with evaluations on
 * the same handle.  There is no CPU
 * fallback anywhere behind this ABI: device entry points fail with
 * FDG_E_NO_DEVICE when no gfx950 device is usable.
 *
 * Node table (value index space): leaves 0..n_leaf-1 in leafVal order, then
 * internal nodes n_leaf..n_leaf+n_node-1 in statement order; every child index
 * is smaller than its node's index.
 *
 * Arithmetic contract (what "identical results" means): fp64, no FMA
 * contraction, n-ary Sum/Prod evaluated as the left folds the reference's
 * generated code performs (static.jl:13-31 + Julia's left-associative n-ary
 * + and *):  Sum  = (((c1[*f1]) + (c2[*f2])) + ...),
 *            Prod = ((((c1[*f1]) * c2)[*f2]) * ...),
 *            Power{2} = c*c, Power{3} = c*c*c, other N = fdg_powi(c, N),
 * a factor is applied only when it differs from 1 (static.jl:15,18,25,28).
 */
#ifndef FDG_H
#define FDG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FDG_VERSION 102 /* 0.1.1: tile-major batches, interpreter association */

#define FDG_OP_SUM 0u
#define FDG_OP_PROD 1u
#define FDG_OP_POWER 2u
#define FDG_NO_ROOT 0xFFFFFFFFu /* root id not present in any graph: root[k] left untouched */

enum {
  FDG_OK = 0,
  FDG_E_INVALID = -1,    /* malformed table / bad argument */
  FDG_E_UNSUPPORTED = -2,/* unknown operator (static.jl:6-11) */
  FDG_E_NO_DEVICE = -3,  /* no usable gfx950 device / HIP runtime failure */
  FDG_E_NOMEM = -4,
  FDG_E_JIT = -5,        /* kernel specialization failed */
  FDG_E_INTERNAL = -6
};

typedef struct fdg_graph_desc {
  uint32_t n_leaf;            /* L */
  uint32_t n_node;            /* N internal nodes */
  uint32_t n_root;            /* R */
  uint32_t n_edge;            /* E = child_off[n_node] */
  const uint8_t *op;          /* [N] FDG_OP_* */
  const int32_t *power;       /* [N] exponent for FDG_OP_POWER, ignored otherwise */
  const uint32_t *child_off;  /* [N+1] */
  const uint32_t *child_idx;  /* [E] value index of each operand */
  const double *child_fac;    /* [E] subgraph_factors */
  const uint32_t *root_slot;  /* [R] value index written to root[k], or FDG_NO_ROOT */
} fdg_graph_desc;

typedef struct fdg_graph fdg_graph; /* opaque */

/* What the lowering did; all counts are per graph, not per sample. */
typedef struct fdg_graph_info {
  uint32_t n_leaf, n_node, n_root, n_edge;
  uint32_t n_live_node;     /* internal nodes reachable from a root */
  uint32_t n_live_leaf;     /* leaves reachable from a root */
  uint64_t flops_alg;       /* adds + mults + factor mults + power mults, reachable part */
  uint64_t bytes_alg;       /* 8*(L+R): algorithmic HBM bytes per evaluation */
  uint32_t max_live;        /* peak number of simultaneously live values (leaves on demand) */
  uint32_t n_slot_lds;      /* per-sample fp64 slots kept in LDS by the interpreter */
  uint32_t n_slot_mem;      /* per-sample fp64 slots kept in the HBM workspace panel */
  uint32_t n_ops;           /* micro-ops in the interpreter stream */
  int32_t specialized;      /* 1 when a straight-line kernel is loaded */
  uint32_t spec_vgpr, spec_lds_bytes, spec_scratch_bytes; /* of the specialized kernel */
} fdg_graph_info;

/* flags for fdg_graph_specialize */
#define FDG_SPEC_DEFAULT 0u
#define FDG_SPEC_KEEP_SOURCE 1u   /* leave the generated source next to the code object */
#define FDG_SPEC_FAST_MATH 2u     /* allow FMA contraction (compiler flag for HIP source; with FDG_SPEC_ISA a product used
                                   * once by a sum becomes v_fma_f64): NOT bit-exact -- within 1e-12 of the sums' term
                                   * scale -- and reported separately */
#define FDG_SPEC_AUTOTUNE 8u      /* with FDG_SPEC_ISA: pick the configuration by timing a few candidates on
                                     the device (needs one); the choice is remembered in the cache directory */
#define FDG_SPEC_ROW_MAJOR_COMPANION 16u /* keep the handle's current (FDG_SPEC_ISA) kernels and add the HIP-source ones next to
                                   * them; sample-major input (compile_Python's [B,L]: leaf stride 1) is then evaluated by the
                                   * companion, whose lanes read their own rows, instead of being transposed for the ISA
                                   * kernel.  Worth it below 16 leaves, where the ISA back end has no row-major variant of its own; same bits either way. */
#define FDG_SPEC_ISA 4u           /* optimizing back end: own scheduler + register allocator, gfx950
                                     assembly printed directly (one VALU instruction per fold step) */

const char *fdg_last_error(void);
int fdg_version(void);

/* Host-only: validates and lowers the table (dead-code elimination, slot
 * allocation, interpreter stream).  Needs no device. */
int fdg_graph_create(const fdg_graph_desc *desc, fdg_graph **out);
int fdg_graph_destroy(fdg_graph *g);
int fdg_graph_query(const fdg_graph *g, fdg_graph_info *info);

/* The reference has TWO evaluators of a graph, and they round differently:
 *   FDG_ASSOC_STATIC (default): the generated eval_graph! of Compilers.compile (src/backend/static.jl:13-46), the contract above;
 *   FDG_ASSOC_INTERP: the interpreter eval! (src/computational_graph/eval.jl:1-3,15-39), which the reference's examples and tests
 *     call (example/benchmark.jl:84-86):  Sum = sum(w_i * f_i), Prod = prod(w_i * f_i) = (((w1*f1) * (w2*f2)) * (w3*f3)) ...,
 *     Power{N} = w^N * f -- every operand is scaled by its factor BEFORE it enters the left fold (sum / prod over a generator
 *     are left folds).  Multiplying by a factor 1.0 leaves the bits alone, so the two differ exactly where a Prod has a factor
 *     other than +-1 on its second or a later operand: ((acc * w) * f) against (acc * (w * f)).
 * A handle evaluates with ONE of them, on every back end (interpreter, HIP source, ISA, cooperative, Monte-Carlo step); choose
 * before the first fdg_graph_specialize* call (FDG_E_INVALID afterwards).  Host-only. */
#define FDG_ASSOC_STATIC 0
#define FDG_ASSOC_INTERP 1
int fdg_graph_set_association(fdg_graph *g, int assoc);

/* Options of a handle.  The library never reads the process environment while it specialises or launches: the FDG_* variables an
 * installation may set (FDG_CACHE_DIR, FDG_CACHE_RO_DIR, FDG_CACHE_TRUST, FDG_LLVM_BIN, FDG_HIPCC, FDG_JIT, FDG_MC_ROUTE,
 * FDG_EVAL_CHUNK, FDG_MC_CHUNK, FDG_SM_CHUNK_MB, FDG_IGNORE_TUNED, FDG_LEAF_GENERIC, FDG_TUNE_VERBOSE, FDG_ISA_[NO_]POOL, FDG_ISA_[NO_]RL)
 * are copied ONCE per process, every handle starts with a copy, and a handle's behaviour is a function of its own options from then on.
 * fdg_graph_set_option changes one (value NULL: removes it) -- the same names, plus the variant selectors and tuning knobs the tests and dev
 * tools use (FDG_ISA_W2, FDG_ISA_COOP, FDG_COOP_WAVES, FDG_ISA_NO_FUSED_ACC, FDG_ROOT_SCRATCH_MB, ...).  Options that shape
 * a kernel must be set before the fdg_graph_specialize* call that builds it; options of the launch path (FDG_ISA_NO_*, FDG_*_CHUNK*,
 * FDG_ROOT_SCRATCH_*, FDG_ISA_WAVES_PER_CU, FDG_ISA_OVERSUB, FDG_ISA_MEM_*) take effect with the next call: they are parsed into the
 * handle here, and between an evaluation entry point and hipModuleLaunchKernel nothing is looked up by name.  Thread-safe against
 * concurrent launches of the same handle (taken under the handle's mutex).  Names must start with "FDG_".  fdg_graph_get_option returns
 * the value (owned by the handle, valid until the option changes) or NULL.  No counterpart in the reference. */
int fdg_graph_set_option(fdg_graph *g, const char *name, const char *value);
const char *fdg_graph_get_option(const fdg_graph *g, const char *name);
/* The process defaults: what handles created AFTER the call start with, and what the entry points that take no handle see
 * (fdg_leaf_eval_device*: FDG_LEAF_GENERIC).  Initialised from the environment at first use (supported names only). */
int fdg_set_default_option(const char *name, const char *value);
const char *fdg_get_default_option(const char *name);   /* valid until the calling thread's fourth next call of this function */

/* What the specialised kernels of a handle execute per evaluation and which of them the last device call launched --
 * the figures a roofline needs (bench.py: executed fold steps against the fp64 issue peak, bytes against HBM) without
 * guessing on the host side which variant the library picked.  Counts are per sample (one lane); slot 0 = the evaluator
 * (fdg_isa_eval[_nt]), 1 = fused accumulation (fdg_isa_eval_acc[_nt]), 2 = the row-major variant (fdg_isa_eval_rm).
 * All zero for handles that are not specialised with FDG_SPEC_ISA.  No counterpart in the reference. */
typedef struct fdg_kernel_info {
  char last_kernel[48];       /* name of the evaluator kernel the last device call on this handle launched ("" = none yet) */
  uint64_t n_valu[3];         /* vector-ALU fold steps executed per evaluation (after value numbering / recomputation) */
  uint32_t n_ld_leaf[3];      /* leaf loads from the input matrix (> n_live_leaf: leaves evicted and read again) */
  uint32_t n_panel[3];        /* loads + stores of the HBM workspace panel */
  uint32_t n_lds[3];          /* loads + stores of per-lane LDS slots */
  uint32_t waves_per_cu[3];   /* resident waves per CU the launch uses */
  uint32_t has_acc, has_rm, has_coop, rm_bufs;
  uint32_t has_pool;          /* the pooled cooperative variant (fdg_isa_eval_pool) is installed: full tiles of batches whose samples of a leaf are contiguous */
  uint32_t pool_fetch;        /* ... leaf fetches from memory per evaluation (>= the live leaves) */
  uint64_t pool_valu;         /* ... fold steps executed per evaluation, all waves together */
  uint32_t has_rl;            /* the linear row-major variant (fdg_isa_eval_rl) is installed: contiguous rows, full tiles */
  uint32_t rl_reserved;
  uint64_t rl_valu;           /* ... fold steps it executes per evaluation */
} fdg_kernel_info;
int fdg_graph_kernel_info(fdg_graph *g, fdg_kernel_info *info);

/* Emits HIP source for a straight-line kernel of this graph (one lane = one
 * sample, values in VGPRs, compiler-managed overflow) and returns it as a
 * malloc'ed NUL-terminated string the caller frees with fdg_free.  Host-only. */
int fdg_graph_emit_source(const fdg_graph *g, unsigned flags, char **source);
void fdg_free(void *p);

/* JIT: emit + compile for gfx950 (hiprtc, else `hipcc --genco`) + cache the code
 * object in cache_dir (NULL: $FDG_CACHE_DIR or /tmp/fdg-cache).  The module is
 * loaded lazily on first device use, so this works without a device present
 * (cross-compile at build time, run on the GPU box). */
int fdg_graph_specialize(fdg_graph *g, const char *cache_dir, unsigned flags);

/* Tuning knobs of the FDG_SPEC_ISA back end; zero fields take the default. */
typedef struct fdg_opt_params {
  uint32_t n_reg;          /* fp64 values kept in VGPR pairs (<= 125) */
  uint32_t n_lds;          /* fp64 LDS slots per lane (<= 127) */
  uint32_t lookahead_lds;  /* prefetch distance, in ops, of LDS loads */
  uint32_t lookahead_mem;  /* prefetch distance, in ops, of workspace-panel (L2/HBM) loads */
  uint32_t lookahead_leaf; /* prefetch distance, in ops, of first-use leaf loads (HBM) */
  uint32_t n_acc;          /* AGPR pairs per lane used as a spill level (<= 124; 0 with two waves per SIMD) */
  uint32_t vn_window;      /* value numbering of identical fold steps: 0 default, 1 off, n > 1 window in ops */
  uint32_t fma;            /* fdg_graph_opt_program only: 1 = fuse products into sums like FDG_SPEC_FAST_MATH does */
  uint32_t remat_window;   /* > 0: the value of a cheap node that has not been read for this many ops is forgotten and computed
                            * again by its next consumer (same operations, same bits): arithmetic instead of spill traffic */
  uint32_t remat_cost;     /* ... "cheap" = at most this many fold steps of its own (default 4) */
} fdg_opt_params;

/* One op of the register-allocated program (for inspection and for host-side
 * checkers that replay it): kind 0 LD_LEAF r[d]=leaf[a], 1 LD_LDS r[d]=lds[a],
 * 2 LD_MEM r[d]=ws[a], 3 ST_LDS lds[d]=r[a], 4 ST_MEM ws[d]=r[a],
 * 5 MUL r[d]=(+-r[a])*(+-r[b]), 6 ADD, 7 MULC r[d]=(+-r[a])*imm, 8 ROOT root[d]=+-r[a],
 * 10 LD_ACC r[d]=acc[a], 11 ST_ACC acc[d]=r[a].  Programs of fdg_graph_mc_program also contain the leaf formulas'
 * 16 ADDC r[d]=(+-r[a])+imm, 17 EXP r[d]=exp(+-r[a]), 18 RCP r[d]=1/(+-r[a]),
 * 19 SEL r[d]= cond(+-r[c]) ? +-r[a] : +-r[b] (cond: x>0 if imm==0, x>=0 otherwise), 20 FIXZ r[d]= r[a]==0 ? imm : r[a],
 * 21 SELC r[d]= cond(+-r[a]) ? imm : -imm (x>0 if negb==0, x>=0 otherwise); their LD_LEAF reads input column a
 * (momentum components first, then times). */
typedef struct fdg_mop {
  uint8_t kind, nega, negb, negc;
  uint32_t d, a, b;
  double imm;
  uint32_t c, param;   /* param (programs of fdg_graph_mc_program): 0, or imm is a physical parameter the kernel takes as an argument
                        * -- 1: -kF^2, 2: beta, 3: -beta, 4: lambda -- and holds the value the program was built with;
                        * c: third source of kinds 14 FMA r[d]=(+-r[a])*(+-r[b])+(+-r[c]) and 15 FMAC r[d]=(+-r[a])*imm+(+-r[c]),
                        * which only FDG_SPEC_FAST_MATH programs contain */
} fdg_mop;

/* Optional scheduling hint for FDG_SPEC_ISA: group[n] (n < n_node) tags internal nodes that belong
 * together -- e.g. the Taylor coefficients that taylorexpansion! (src/utility.jl:105-135) derives from
 * one original node share that node's id.  Members of a group are evaluated together.  Only the order
 * of evaluation changes, never a value.  Pass NULL to clear. */
int fdg_graph_set_schedule_groups(fdg_graph *g, const uint32_t *group, uint32_t n_node);

/* Sets the parameters used by the next fdg_graph_specialize(..., FDG_SPEC_ISA). */
int fdg_graph_set_opt_params(fdg_graph *g, const fdg_opt_params *prm);
/* Runs scheduler + allocator and returns the op list (malloc'ed; fdg_free).  Host-only. */
int fdg_graph_opt_program(const fdg_graph *g, const fdg_opt_params *prm, fdg_mop **ops, uint64_t *n_ops,
                          uint32_t *n_reg_used, uint32_t *n_lds_used, uint32_t *n_mem_used, uint32_t *n_acc_used);

/* The program of wave `wave` (0..3) of the cooperative variant -- the four waves of a CU evaluate one 64-sample tile
 * together, each on its share of the graph, values crossing between waves through shared LDS slots (kinds 25 SEND
 * shared[d]=r[a], 26 RECV r[d]=shared[a], 27 BARRIER) -- for inspection and host-side replay.  info[8] (optional):
 * registers, private LDS slots, panel slots, AGPR pairs of this wave; shared slots, barriers per tile, hand-overs per
 * tile, fold steps computed by more than one wave.  Host-only.  FDG_E_UNSUPPORTED when the graph has no wide root sum. */
int fdg_graph_coop_program(const fdg_graph *g, const fdg_opt_params *prm, uint32_t wave, fdg_mop **ops, uint64_t *n_ops,
                           uint32_t *info);

/* The programs of the POOLED cooperative variant (fdg_isa_eval_pool): whole roots are dealt to the eight waves of a CU, and no wave
 * loads a leaf from memory into a register -- the tile's leaves are fetched into a shared LDS pool (kind 29 POOL_FETCH shared[d] =
 * leaf[a], readable from epoch (uint32)imm on; LDS-direct loads issued epochs ahead by the waves in turn) and read from there (kind 26).
 * For graphs whose one-wave kernels re-read their leaves (the vertex functions of example/benchmark.jl and example/benchmark_GV.jl).
 * info as for fdg_graph_coop_program, with info[6] = leaf fetches per tile.  Host-only.  FDG_E_UNSUPPORTED when the graph has fewer
 * than two roots per wave or the pool cannot hold what an epoch reads. */
int fdg_graph_pool_program(const fdg_graph *g, const fdg_opt_params *prm, uint32_t wave, fdg_mop **ops, uint64_t *n_ops,
                           uint32_t *info);

/* ---- element types other than Float64 -----------------------------------------------------------------------------
 * The function Compilers.compile returns is generic in eltype(leafVal) (the generated text has no type in it,
 * src/backend/static.jl:98-133); to_Cstr / compile_C map the weight types they know (static.jl:135-153: Float32, ComplexF32,
 * ComplexF64, ...).  fdg_graph_specialize_typed compiles a per-graph HIP-source kernel for one such type,
 * fdg_eval_device_typed evaluates with leaf and root buffers of that type (strides in elements; a complex element is the pair
 * (re, im), as in Julia and C).  What it computes is what the Julia function computes on Vector{T} arguments: the factors are
 * Float64 literals in the text, so a factor != 1 promotes a Float32 value to Float64 for the rest of its expression (Julia's
 * promotion rules are C++'s here), products and sums of values of the type stay in the type, Complex * Complex is
 * (ar br - ai bi, ar bi + ai br) without contraction, Complex * Real scales both components (base/complex.jl), and a root is
 * converted to the element type when stored.  Covered: Sum, Prod, Power{2}, Power{3} (other literal powers take type-specific
 * paths through Base.power_by_squaring: FDG_E_UNSUPPORTED).  No fused accumulation, no ISA back end for these types: they go
 * through the compiler-scheduled kernel (every BASELINE configuration is Float64).  FDG_DT_F64 forwards to the ordinary entry
 * points. */
#define FDG_DT_F64 0
#define FDG_DT_F32 1
#define FDG_DT_C64 2
#define FDG_DT_C32 3
int fdg_graph_specialize_typed(fdg_graph *g, int dtype, const char *cache_dir, unsigned flags);
/* ComplexF64 rows.  fdg_graph_create_complex_view returns a NEW handle (the caller's to destroy) for the Float64 graph that is g's
 * graph on Complex{Float64} values spelled out on real and imaginary parts: leaves re_0, im_0, re_1, im_1, ... -- a row of a
 * row-major ComplexF64 [B, L] matrix read as 2 L doubles --, roots (re, im) of g's roots, every operation the one
 * base/complex.jl performs (z w = (zr wr - zi wi, zr wi + zi wr), z f = (re f, im f), + componentwise, z^2 = z z, z^3 = (z z) z)
 * in the association of the evaluator's own folds, so the ordinary Float64 entry points give the generic function's bits on
 * ComplexF64 arguments.  FDG_E_UNSUPPORTED for other powers.  fdg_graph_specialize_typed(g, FDG_DT_C64, dir, FDG_SPEC_ISA) builds such
 * a view inside g and specialises it with the optimizing back end; fdg_eval_device_typed then sends row-major batches
 * (leaf_leaf_stride == 1, root_root_stride == 1) through its in-place row-major kernel when it has one, everything else through
 * the per-type kernel. */
int fdg_graph_create_complex_view(const fdg_graph *g, fdg_graph **out);
int fdg_eval_device_typed(fdg_graph *g, int dtype, const void *d_leaf, int64_t leaf_sample_stride, int64_t leaf_leaf_stride,
                          void *d_root, int64_t root_sample_stride, int64_t root_root_stride, int64_t n_sample, void *stream);

/* Evaluate B samples, buffers in device memory.
 *   leaf value i of sample b : d_leaf[b*leaf_sample_stride + i*leaf_leaf_stride]
 *   root value k of sample b : d_root[b*root_sample_stride + k*root_root_stride]
 * (strides in elements).  compile_Python's row-major [B,L] / [B,R] is
 * (L,1)/(R,1); a Julia column-major B x L matrix is (1,B)/(1,B).
 * stream: hipStream_t (NULL = default stream).  Asynchronous.
 * Any strides and bases are accepted and give the same values.  Column-major matrices whose column stride is a
 * multiple of 16 elements and whose bases lie on 128-byte lines (hipMalloc'ed buffers, B a multiple of 16) take the
 * streaming form of the kernel (non-temporal accesses), 3-5 % faster on graphs bound by memory. */
int fdg_eval_device(fdg_graph *g, const double *d_leaf, int64_t leaf_sample_stride,
                    int64_t leaf_leaf_stride, double *d_root, int64_t root_sample_stride,
                    int64_t root_root_stride, int64_t n_sample, void *stream);

/* Host-buffer convenience: row-major leaf[B,L] -> root[B,R]; H2D, eval, D2H,
 * synchronous.  Same in-place semantics as eval_graph!(root, leafVal). */
int fdg_eval(fdg_graph *g, const double *leaf, double *root, int64_t n_sample);

/* Same with strided host matrices (element strides as in fdg_eval_device): row-major [B, L] / [B, R]
 * (value stride 1) or column-major B x L / B x R -- what a Julia Matrix is: sample stride 1, value
 * stride >= n_sample.  The device copy keeps the host's orientation, so a Julia matrix reaches the
 * evaluator leaf-major without any transposition pass.  Leaves and roots may differ in orientation. */
int fdg_eval_strided(fdg_graph *g, const double *leaf, int64_t leaf_sample_stride, int64_t leaf_leaf_stride,
                     double *root, int64_t root_sample_stride, int64_t root_root_stride, int64_t n_sample);

/* d_acc[k] += sum_b weight[b] * root_k(b)   (d_weight may be NULL: weight 1).
 * Per-lane accumulators inside the evaluator (ISA back end) or fixed-shape block trees, then one partial per
 * wave / block and root summed in fixed order by a second kernel: deterministic for a given launch shape, no atomics.
 * d_acc must hold n_root doubles and be zeroed by the caller. */
int fdg_accumulate_device(fdg_graph *g, const double *d_leaf, int64_t leaf_sample_stride,
                          int64_t leaf_leaf_stride, const double *d_weight, double *d_acc,
                          int64_t n_sample, void *stream);

/* ---- tile-major batches ------------------------------------------------------------------------------------------------
 * The layout the evaluator streams best, and the one a Monte-Carlo driver that owns its sample batch should allocate: samples
 * are grouped in tiles of FDG_TILE_SAMPLES = 64 (one wave), and a tile's block of the array is contiguous,
 *   leaf value i of sample b : d_leaf[(b / 64) * leaf_tile_stride + (b % 64) * leaf_sample_stride + i * leaf_leaf_stride]
 *   root value k of sample b : d_root[(b / 64) * root_tile_stride + (b % 64) * root_sample_stride + k * root_root_stride]
 * -- a Julia Array{Float64,3}(undef, 64, L, cld(B, 64)) is (sample, leaf, tile) strides (1, 64, 64 L); its roots
 * Array{Float64,3}(undef, 64, R, cld(B, 64)) are (1, 64, 64 R).  A wave then reads ONE contiguous block of 512 L bytes front to
 * back instead of 64 samples of each of L columns that lie B * 8 bytes apart: L concurrent address streams per wave become one,
 * and every page of the batch is touched by one wave, once (the leaf-major matrix of the headline runs 0.66-0.77
 * of the HBM roofline depending on where its pages landed; the tile-major batch does not show the two modes).  The buffers hold
 * cld(n_sample, 64) whole tiles; lanes of a last partial tile are neither read nor written.  A tile stride of 0 stands for
 * 64 * sample stride: the call is then exactly fdg_eval_device / fdg_accumulate_device on a strided matrix.  Same values,
 * bit for bit, as every other layout.  Needs a handle specialised with FDG_SPEC_ISA (FDG_E_UNSUPPORTED otherwise).
 * Stands in for: eval_graph!(root, leafVal) once per sample (static.jl:100,131), as fdg_eval_device does. */
#define FDG_TILE_SAMPLES 64
int fdg_eval_device_tiled(fdg_graph *g, const double *d_leaf, int64_t leaf_sample_stride, int64_t leaf_leaf_stride,
                          int64_t leaf_tile_stride, double *d_root, int64_t root_sample_stride, int64_t root_root_stride,
                          int64_t root_tile_stride, int64_t n_sample, void *stream);
int fdg_accumulate_device_tiled(fdg_graph *g, const double *d_leaf, int64_t leaf_sample_stride, int64_t leaf_leaf_stride,
                                int64_t leaf_tile_stride, const double *d_weight, double *d_acc, int64_t n_sample, void *stream);
/* harness: fdg_fill_uniform_device's values (same counters: sample_offset + b, i) written into a tile-major batch */
int fdg_fill_uniform_device_tiled(double *d_leaf, int64_t n_sample, uint32_t n_leaf, int64_t leaf_sample_stride,
                                  int64_t leaf_leaf_stride, int64_t leaf_tile_stride, uint64_t seed, uint64_t sample_offset,
                                  void *stream);

/* A matrix in one of the reference's layouts -> the tile-major batch, and back (round 6).  d_tiled is (64, n_col, cld(n_sample, 64)) as
 * fdg_eval_device_tiled takes it; the matrix is m[b * sample_stride + c * col_stride]: a Julia column-major B x C Matrix{Float64} is
 * (1, B) -- its 512-byte runs are copied as they are --, compile_Python's row-major [B, C] is (C, 1) -- 64 x 64 tiles through LDS.  One pass
 * at copy speed (read + written 5-6 TB/s): 2.3 x the time of ONE evaluation of the headline graph over the same batch, so it pays for a
 * batch that is evaluated several times, or when the producer of the leaves cannot write tile-major itself (fdg_leaf_eval_device_tiled
 * and the fused Monte-Carlo step do); DESIGN.md 6f has the measured cost.  Lanes past n_sample of the last tile are not written.
 * Stands in for nothing in the reference (its batched layout is compile_Python's [B, L], compiler_python.jl:23,28,45-47); the Julia shim
 * exposes them as tile_major!(dst, src) / from_tile_major!(dst, src). */
int fdg_repack_tile_major(const double *d_src, int64_t sample_stride, int64_t col_stride, double *d_tiled, int64_t n_sample,
                          uint32_t n_col, void *stream);
int fdg_unpack_tile_major(const double *d_tiled, double *d_dst, int64_t sample_stride, int64_t col_stride, int64_t n_sample,
                          uint32_t n_col, void *stream);

/* Device memory for a sample batch, backed explicitly (HIP virtual-memory management: one reserved address range, physical
 * chunks of chunk_bytes created and mapped in address order; chunk_bytes 0 = one physical allocation for the whole batch) instead
 * of by whatever state the driver's allocator is in when hipMalloc is called -- how a batch of tens of GB is backed decides
 * which of two rates its stream runs at.  The pointer is an ordinary device pointer for every entry point above and
 * for the caller's own kernels; release it with fdg_batch_free (synchronises the device).  The current device is used.
 * No counterpart in the reference (its leaf vector is a Julia Vector on the host). */
int fdg_batch_alloc(size_t bytes, size_t chunk_bytes, void **d_ptr);
int fdg_batch_free(void *d_ptr);

/* The two arrays of a TILE-MAJOR batch of one handle -- leaves (64, L, T) and roots (64, R, T), T = cld(n_sample, 64), strides as in
 * fdg_eval_device_tiled -- backed so that the evaluation streams at its fast rate over every part of the batch, whatever state the
 * driver's allocator is in (round 5).  On MI355X the rate at which a piece of leaves is evaluated depends on the physical
 * pages under that piece AND under the roots it writes: device memory falls into regions of two kinds, and reading one kind while writing
 * the same kind runs 10-12 % slower than the mixed combination (fused accumulation, which writes no roots, does not care).  Physical
 * addresses are invisible to a process; the rate is not.  With FDG_BATCH_PAIR_CALIBRATE the allocator maps the leaves in chunks of about
 * chunk_bytes_hint (0: 2 GB; rounded so that a chunk holds whole tiles of both arrays in whole mapping granules), draws more root chunks
 * than it needs -- from an 80 GB span of the device memory, or as much of it as is free beyond what the batch itself needs --, times the
 * handle's own evaluator on (leaf chunk, root chunk) pairs -- about a millisecond per pair -- and maps behind every leaf chunk the fastest
 * unused root chunk, drawing more while even the best is more than 3.5 % below the best pair seen (three levels were measured: 0.855 /
 * 0.80 / 0.765 of 8 TB/s on the headline graph; nothing is assumed about their number); what it drew and did not use goes back to the
 * driver.  When memory is short the search shrinks instead of failing (info->level_reached).  Monte-Carlo callers that only ACCUMULATE
 * (fdg_accumulate_device_tiled: no root is written) gain nothing from the pairing -- 0.89 of 8 TB/s on any allocation -- and should skip the
 * 7-9 s: allocate without the flag.  Without the flag the chunks are mapped in the order they were drawn (the A/B case).  Both arrays hold whole chunks (>= the T tiles asked for); release each with
 * fdg_batch_free.  `info` (may be NULL) reports what was found.  Needs a handle specialised with FDG_SPEC_ISA and the current device.
 * No counterpart in the reference (its leaf vector and root vector are Julia Vectors on the host, static.jl:100,131). */
#define FDG_BATCH_PAIR_CALIBRATE 1u
#define FDG_BATCH_PAIR_VERBOSE 2u   /* what the search saw, a few lines on stderr */
#define FDG_BATCH_PAIR_LEAF_MAJOR 16u /* the arrays are a Julia column-major pair B' x L and B' x R (strides (1, B'), B' = the mapped sample count: info->chunk_tiles * 64):
                                       * one window -- the whole batch --, the candidates are whole root matrices; pays while the leaf matrix lies in one or
                                       * two regions of the memory (up to a few tens of GB) */
#define FDG_BATCH_PAIR_ROW_MAJOR 8u /* the arrays are compile_Python's row-major [B, L] and [B, R] (compiler_python.jl:23,28,45-47) instead of the
                                     * tile-major ones: 64 consecutive rows take the place of a tile; everything else is the same */
typedef struct fdg_batch_pair_info {
  uint64_t leaf_bytes, root_bytes;   /* mapped bytes of the two arrays */
  uint64_t chunk_tiles;              /* 64-sample tiles per chunk */
  uint32_t n_chunk;                  /* chunks per array */
  uint32_t n_candidate;              /* root chunks drawn */
  uint32_t n_filler;                 /* 2 GB allocations held for a while to make the driver hand out other regions (released) */
  uint32_t n_probe;                  /* timed (leaf chunk, root chunk) pairs */
  uint32_t n_matched;                /* chunk pairs within 5 % of the best pair timed, as mapped */
  uint32_t calibrated;               /* 1: the candidates differed by more than 5 % (there was something to choose); 0: calibration off or no contrast */
  double gbs_fast, gbs_slow;         /* the best and the worst pair timed (algorithmic GB/s of one chunk's launch); "fast level" = within 3.5 % of gbs_fast */
  double gbs_before_mean, gbs_before_min;   /* chunk pairs as an uncalibrated mapping would have made them */
  double gbs_after_mean, gbs_after_min;     /* chunk pairs as mapped */
  double seconds;                    /* wall time of the call */
  double seconds_settling;           /* ... of which: waiting for the driver's background wipe of released memory to end */
  uint32_t level_reached;            /* how far the search got (round 6: it degrades, it does not fail, when memory is short).  0: no calibration (not asked
                                      * for, or windows too small to time); 1: asked for, but the leaves only fitted once every candidate had been given
                                      * back: mapped in draw order; 2: calibrated over a span of the memory cut short by what was free (the best pairs
                                      * found within it); 3: the full search */
  uint32_t span_gb;                  /* GB of fillers the candidates were drawn behind (80 at first, up to 144 more while the best pair seen is below par) */
} fdg_batch_pair_info;
int fdg_batch_alloc_pair(fdg_graph *g, int64_t n_sample, size_t chunk_bytes_hint, unsigned flags, void **d_leaf, void **d_root,
                         fdg_batch_pair_info *info);
/* test hook (host only): the allocator's search on a MODEL of the memory -- regions of four kinds, three pair levels, noise, further candidates
 * only behind further draws (scenario 0 ... 3, csrc/fdg_batch.cpp); returns how many of n_window windows ended with a candidate of the
 * complementary kind, *n_probe = pairs "timed".  So that the search is tested where there is no device. */
int fdg_selftest_pair_search(uint64_t seed, uint32_t scenario, uint32_t n_window, uint32_t *n_probe);

/* d_leaf[b*ss + i*ls] = U[0,1) from Philox4x32-10, key = seed, counter =
 * (sample_offset + b, i): independent of launch geometry and of how samples
 * are sharded over GPUs. */
int fdg_fill_uniform_device(double *d_leaf, int64_t n_sample, uint32_t n_leaf,
                            int64_t leaf_sample_stride, int64_t leaf_leaf_stride, uint64_t seed,
                            uint64_t sample_offset, void *stream);

/* Checks a gfx950 assembly listing (e.g. the `.s` a FDG_SPEC_ISA | FDG_SPEC_KEEP_SOURCE specialisation leaves in the
 * cache directory) against the wait-state table the ISA emitter itself uses (csrc/fdg_isa.cpp: trans result -> VALU,
 * VALU write of SGPR/VCC -> VALU read / v_div_fmas / VMEM, VALU -> v_readlane, wide store data -> VALU overwrite).
 * Returns the number of violations (0 = clean), < 0 on error; *report (optional, fdg_free) lists the rules, the
 * violations and what was checked.  Host-only; no counterpart in the reference (its code generator emits Julia). */
int fdg_isa_check_hazards(const char *asm_text, char **report);

/* Harness only (bench.py's `roofline.measured_copy_gbs`): d_dst[0..n) = d_src[0..n), 16 bytes per lane,
 * n even, both pointers 16-byte aligned.  The box's own streaming ceiling next to the 8 TB/s spec
 * (the reference has no counterpart). */
int fdg_copy_device(double *d_dst, const double *d_src, int64_t n, void *stream);

/* Harness only (bench.py's `roofline.measured_read_gbs`): streams d_src[0..n) through the chip with non-temporal loads and writes nothing
 * (d_sink: one double, written only if the data sums to a value it cannot have).  The memory system's ceiling for a read stream --
 * the evaluator's traffic is 95 % reads -- which is above what a copy reaches.  No counterpart in the reference. */
int fdg_read_device(const double *d_src, int64_t n, double *d_sink, void *stream);

/* Harness only (bench.py's `roofline.clock_ghz`): enqueues ONE wave on `stream` that sleeps for `seconds` of wall time
 * (0 < seconds <= 30) and then writes d_ticks[0] = shader-clock ticks, d_ticks[1] = 100 MHz ticks that went by: launched on
 * a side stream next to the evaluator it reports the clock the chip sustained under that load (the graphs at the
 * compute/memory ridge run against the power budget: 1.8-1.9 GHz instead of 2.4).  8 VGPRs, no LDS: it shares a SIMD with
 * two 248-register evaluator waves.  No counterpart in the reference. */
int fdg_clock_probe_device(double seconds, int64_t *d_ticks, void *stream);

/* ---- leaf values on device (the caller's side of the path) ----------------
 * The per-sample leaf loop of the reference's example integrand (example/benchmark.jl:58-81):
 *   loops = K[:, 1:n_loop] * basis                       (FrontEnds.update, src/frontend/pool.jl:69-76)
 *   type 1 (fermionic G): tau = T[tau_out] - T[tau_in];  eps = |loops[:, loop_index]|^2 - kF^2;
 *                         leaf = green(tau, eps, beta)   (example/benchmark.jl:113-127) for order 0;
 *                         orders 1..5: green_derive (benchmark.jl:93-111) = (-1)^n/n! d^n/d eps^n of the
 *                         fermionic kernel.  The reference calls Lehmann.jl's kernelFermiT_dw* for these;
 *                         Lehmann.jl is not part of the reference checkout, so the definition is restated
 *                         (overflow-safe closed form) and pinned by high-precision vectors, not by
 *                         Lehmann.jl output.  Other orders: "not implemented!" like benchmark.jl:108
 *   type 2 (bosonic V):   invK = 1/(|q|^2 + lambda);     leaf = 8*pi/invK * (lambda*invK)^order
 *   type 0:               leaf left untouched
 * with the tables FrontEnds.leafstates returns (src/frontend/frontends.jl:178-232; indices 1-based as
 * in the reference).  Writes the leaf matrix the evaluator reads, so the Monte-Carlo loop
 * (K, T) -> leaves -> graph -> accumulate never leaves the device.  Transcendentals differ from the
 * host libm in the last ulp: parity for this entry point is 1e-13 relative, not bit-exact. */
typedef struct fdg_leaf_tables {
  uint32_t n_leaf, n_basis, n_loop, dim, n_tau;
  const int32_t *leaf_type;    /* [n_leaf] 0 / 1 / 2 */
  const int32_t *leaf_order;   /* [n_leaf] derivative order of that leaf's own kind */
  const int32_t *tau_in;       /* [n_leaf] 1-based index into T */
  const int32_t *tau_out;      /* [n_leaf] */
  const int32_t *loop_index;   /* [n_leaf] 1-based index into the loop basis */
  const double *basis;         /* [n_basis][n_loop] */
  double kF, beta, lambda;
} fdg_leaf_tables;

/* K element (sample b, loop j, component d): d_K[b*k_sample_stride + (j*dim + d)*k_comp_stride];
 * T element (b, i): d_T[b*t_sample_stride + i*t_comp_stride]; leaves as in fdg_eval_device. */
int fdg_leaf_eval_device(const fdg_leaf_tables *tab, const double *d_K, int64_t k_sample_stride,
                         int64_t k_comp_stride, const double *d_T, int64_t t_sample_stride,
                         int64_t t_comp_stride, double *d_leaf, int64_t leaf_sample_stride,
                         int64_t leaf_leaf_stride, int64_t n_sample, void *stream);

/* The same with TILE-MAJOR leaves (fdg_eval_device_tiled): sample b of leaf i is written to
 * d_leaf[(b / 64) * leaf_tile_stride + (b % 64) * leaf_sample_stride + i * leaf_leaf_stride], so that the Monte-Carlo loop
 * (K, T) -> leaves -> fdg_accumulate_device_tiled runs on the layout the evaluator streams fastest. */
int fdg_leaf_eval_device_tiled(const fdg_leaf_tables *tab, const double *d_K, int64_t k_sample_stride,
                               int64_t k_comp_stride, const double *d_T, int64_t t_sample_stride,
                               int64_t t_comp_stride, double *d_leaf, int64_t leaf_sample_stride,
                               int64_t leaf_leaf_stride, int64_t leaf_tile_stride, int64_t n_sample, void *stream);

/* Fused Monte-Carlo step (the integrand of example/benchmark.jl:58-87 in one kernel):
 * the leaves are worked out in registers from the sample's loop momenta K and times T with the formulas
 * of fdg_leaf_eval_device and fed straight into the graph, so the 8*L bytes per evaluation of the leaf
 * matrix never exist.  fdg_graph_specialize_fused JIT-compiles the kernel for (graph, tables) -- tables as
 * for fdg_leaf_eval_device, n_leaf equal to the graph's, leaves without a formula (type 0) are 1.0 like
 * leafstates' initial leafValue; needs no device -- then
 *   fdg_mc_eval_device       root[b][k]           (strides as in fdg_eval_device)
 *   fdg_mc_accumulate_device acc[k] += sum_b weight[b] * root_k(b)   (weight NULL = 1)
 * with K, T laid out as in fdg_leaf_eval_device.  The single kernel is compiler-scheduled (HIP source through
 * hiprtc) and right for graphs of up to a few thousand operations.  For larger graphs the same calls run the
 * specialised leaf kernel into a chunk of leaves owned by the handle and then the handle's own evaluator
 * (specialise it with FDG_SPEC_ISA first): fdg_graph_specialize_fused picks the route by graph size
 * (FDG_MC_ROUTE=fused|split overrides); the roots are the same bits on either route. */
int fdg_graph_specialize_fused(fdg_graph *g, const fdg_leaf_tables *tab, const char *cache_dir, unsigned flags);
/* The third route, taken for large graphs on a handle specialised with FDG_SPEC_ISA: ONE kernel of the optimizing
 * back end whose inputs are the sample's n_loop*dim momentum components and n_tau times; every leaf is a value
 * computed in registers from them at its first use (ops 16..21 of fdg_mop: add-constant, exp, reciprocal, selects),
 * scheduled and register-allocated together with the graph.  kF, beta, lambda are arguments of that kernel (four scalar
 * registers: -kF^2, beta, -beta, lambda), so one code object per (graph, tables) serves every parameter set.  K and T
 * are read in place when both are component-major (sample stride 1; any two column strides); sample-major input is
 * packed into such arrays first.  Leaves agree with fdg_leaf_eval_device within its stated
 * tolerance (own exp: range reduction + degree-11 polynomial), not bit for bit.  FDG_MC_ROUTE=isa|split overrides.
 * fdg_graph_mc_program returns that program for inspection / host-side replay (tab->kF, beta, lambda are used). */
int fdg_graph_mc_program(const fdg_graph *g, const fdg_leaf_tables *tab, const fdg_opt_params *prm, fdg_mop **ops, uint64_t *n_ops,
                         uint32_t *n_reg_used, uint32_t *n_lds_used, uint32_t *n_mem_used, uint32_t *n_acc_used);
int fdg_mc_eval_device(fdg_graph *g, const double *d_K, int64_t k_sample_stride, int64_t k_comp_stride, const double *d_T,
                       int64_t t_sample_stride, int64_t t_comp_stride, double kF, double beta, double lambda,
                       double *d_root, int64_t root_sample_stride, int64_t root_root_stride, int64_t n_sample, void *stream);
int fdg_mc_accumulate_device(fdg_graph *g, const double *d_K, int64_t k_sample_stride, int64_t k_comp_stride, const double *d_T,
                             int64_t t_sample_stride, int64_t t_comp_stride, double kF, double beta, double lambda,
                             const double *d_weight, double *d_acc, int64_t n_sample, void *stream);

/* Device workspace control: the interpreter keeps per-sample overflow slots in
 * an HBM panel owned by the handle; it is sized on first use for the number of
 * resident waves.  This releases it (and any loaded module). */
int fdg_graph_release_device(fdg_graph *g);

/* ---- multi-GPU -------------------------------------------------
 * Samples are sharded over one process per GPU (rank r of G evaluates the r-th
 * contiguous range, Philox counters = global sample index); each rank runs
 * fdg_accumulate_device into its own acc[R]; ONE collective of R doubles -- RCCL
 * over xGMI -- adds the ranks' partial sums.  No data-path collective exists.
 * The reference has no counterpart (example/benchmark*.jl are single-process).
 * RCCL is bound at run time; without it these calls return FDG_E_NO_DEVICE and
 * everything else keeps working.
 *   rank 0: fdg_comm_unique_id(id) -> ship the 128 bytes to the other ranks by any
 *   means (MPI, a file, torch.distributed) -> every rank, with its device current:
 *   fdg_comm_create(id, rank, world, &c) -> ... fdg_reduce_device(c, d_acc, R, -1, stream). */
#define FDG_COMM_ID_BYTES 128
typedef struct fdg_comm fdg_comm;
int fdg_comm_unique_id(void *id, size_t bytes);
int fdg_comm_create(const void *id, int rank, int world, fdg_comm **out);
int fdg_comm_destroy(fdg_comm *c);
/* d_acc[0..n) <- sum over ranks, in place, on `stream`.  root < 0: every rank gets
 * the sum (all-reduce); else only rank `root` (reduce). */
int fdg_reduce_device(fdg_comm *c, double *d_acc, uint32_t n, int root, void *stream);

/* Integer power used for Power{N}, |N| >= 4 (and N < 0): exposed so host-side
 * checkers can call the very same routine.  Pure host function. */
double fdg_powi(double x, int32_t n);

#ifdef __cplusplus
}
#endif
#endif /* FDG_H */
