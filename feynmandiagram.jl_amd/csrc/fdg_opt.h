// Optimizing back end, host side: the node table is expanded into binary
// micro-ops (keeping the reference's left-fold association), scheduled
// depth-first so that every fold step runs as soon as its operand exists, and
// register-allocated with Belady's rule over a three-level per-lane store:
//   REG  (VGPR pairs)      -- operands of every VALU op
//   LDS  (lds[slot][lane]) -- first overflow level, on chip
//   MEM  (ws[slot][lane])  -- second overflow level, HBM/L2 workspace panel
// Leaves are re-loadable from their source (the input matrix when it is
// leaf-major, the staged panel otherwise), so evicting a leaf never stores.
// The result is a linear list of machine ops that fdg_isa.cpp prints as gfx950
// assembly, one instruction per op.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "fdg_internal.h"

namespace fdg {

enum MKind : uint8_t {
  M_LD_LEAF = 0,  // r[d] = leaf[a]
  M_LD_LDS,       // r[d] = lds[a]
  M_LD_MEM,       // r[d] = ws[a]
  M_ST_LDS,       // lds[d] = r[a]
  M_ST_MEM,       // ws[d] = r[a]
  M_MUL,          // r[d] = (+-r[a]) * (+-r[b])
  M_ADD,          // r[d] = (+-r[a]) + (+-r[b])
  M_MULC,         // r[d] = (+-r[a]) * imm
  M_ROOT,         // root[d] = +-r[a]
  M_MOV,          // r[d] = +-r[a]        (only for a root that aliases a negated value)
  M_LD_ACC,       // r[d] = acc[a]        (AGPR pair -> VGPR pair, two v_accvgpr_read_b32)
  M_ST_ACC,       // acc[d] = r[a]
  M_FMA = 14,     // r[d] = (+-r[a]) * (+-r[b]) + (+-r[c])   one rounding: only with OptParams::fma (not parity-exact)
  M_FMAC = 15,    // r[d] = (+-r[a]) * imm + (+-r[c])
  // ---- leaf formulas inside the kernel (build_mc_program; results within the leaf kernels' tolerance, not bit-pinned) ----
  M_ADDC = 16,    // r[d] = (+-r[a]) + imm
  M_EXP = 17,     // r[d] = exp(+-r[a])                       range reduction + degree-11 polynomial + v_ldexp_f64
  M_RCP = 18,     // r[d] = 1 / (+-r[a])                      v_rcp_f64 + two Newton steps
  M_SEL = 19,     // r[d] = cond(+-r[c]) ? +-r[a] : +-r[b]    cond: x > 0 (imm = 0) or x >= 0 (imm = 1)
  M_FIXZ = 20,    // r[d] = r[a] == 0 ? imm : r[a]
  M_SELC = 21,    // r[d] = cond(+-r[a]) ? imm : -imm         cond as M_SEL, selected by negb
  // ---- Power{N} with N outside {2, 3} and counter-terms above order 3: Julia's pow_body spelled out (fdg_powi.h); exact ----
  M_FMAK = 22,    // r[d] = (+-r[a]) * (+-r[b]) + imm        one rounding, as the reference's fma / muladd
  M_DIV1 = 23,    // r[d] = 1.0 / (+-r[a])                    correctly rounded (v_div_scale / v_div_fmas / v_div_fixup)
  M_CONST = 24,   // r[d] = imm                               (operand a is only a scheduling anchor)
  // M_SEL with imm == 2: cond(x) = isfinite(x)
  // ---- cooperative variant: the four waves of a CU work on one tile and hand values over through shared LDS slots ----
  M_SEND = 25,    // shared[d] = r[a]                          visible to the other waves after the next M_BARRIER
  M_RECV = 26,    // r[d] = shared[a]
  M_BARRIER = 27, // s_barrier (every wave's program has the same number of them per tile)
  // ---- one-wave configuration: leaf loads land in AGPR pairs long before their use (no VGPR is tied up by a load in flight) ----
  M_LD_LEAF_ACC = 28, // acc[d] = leaf[a]                     (global_load into the AGPR pair; M_LD_ACC moves it to a VGPR pair at its use)
  // ---- pooled cooperative variant: the tile's leaves go from memory into a shared LDS pool once (LDS-direct loads, no register in between,
  //      issued epochs ahead) and every wave reads them from there (M_RECV) ----
  M_POOL_FETCH = 29,  // shared[d] = leaf[a]                  readable from epoch (uint32_t)imm on; the issuing wave waits for it before the barrier in front
};
inline bool mop_has_a(uint8_t k) { return k != M_RECV && k != M_BARRIER && k != M_POOL_FETCH; }
inline bool mop_has_b(uint8_t k) { return k == M_MUL || k == M_ADD || k == M_FMA || k == M_SEL || k == M_FMAK; }
inline bool mop_has_c(uint8_t k) { return k == M_FMA || k == M_FMAC || k == M_SEL; }
inline bool mop_is_macro(uint8_t k) { return (k >= M_EXP && k <= M_SELC) || k == M_DIV1; }   // needs the emitter's temporaries
inline uint32_t mop_tmp_pairs(uint8_t k) { return k == M_DIV1 ? 4u : (mop_is_macro(k) ? 2u : 0u); }
inline bool mop_exact_fma(uint8_t k) { return k == M_FMAK; }   // (M_FMA / M_FMAC are exact too when pow_body asks for them; fast-math uses them as a contraction)

struct MOp {
  uint8_t kind;
  uint8_t nega, negb;
  uint32_t d, a, b;
  double imm;
  uint8_t negc = 0;     // M_FMA / M_FMAC only
  uint32_t c = 0;
  uint8_t param = 0;    // Monte-Carlo programs: imm is a physical parameter handed to the kernel as an argument (imm holds the value
                        // the program was built with, for replay): 1 = -kF^2, 2 = beta, 3 = -beta, 4 = lambda
};
enum { MC_P_NEG_KF2 = 1, MC_P_BETA = 2, MC_P_NEG_BETA = 3, MC_P_LAMBDA = 4, MC_N_PARAM = 4 };

struct OptParams {
  uint32_t n_reg = 120;     // fp64 registers available to values
  uint32_t n_lds = 80;      // LDS slots per lane
  uint32_t n_acc = 0;       // AGPR pairs per lane used as a spill level (needs 1 wave/SIMD)
  uint32_t lookahead_lds = 32;    // micro-ops of prefetch distance for LDS loads
  uint32_t lookahead_mem = 128;   // ... for loads from the workspace panel (L2 / HBM)
  uint32_t vn_window = 200;       // value numbering: 1 = off, 0 = reuse any earlier identical op, n > 1 = only results at most n ops old
  uint32_t lookahead_leaf = 300;  // ... for first-use loads of leaves (HBM)
  bool fma = false;               // FDG_SPEC_FAST_MATH: a product used once, by a sum, is fused into it (v_fma_f64)
  uint32_t remat_window = 0;      // > 0: the value of a cheap node not read for this many ops is forgotten and computed again by its
  uint32_t remat_cost = 4;        //      next consumer (nodes whose own fold has at most remat_cost steps); exact, trades arithmetic for spills
  bool roots_last = false;        // all root stores at the end of the tile, back to back (row-major roots: the R stores of a row then hit their
                                  // shared cache lines together instead of hundreds of ops apart)
  bool keep_root_order = false;   // roots in the reference's statement order instead of the cone-overlap order: leaves are numbered by first
                                  // visit in that order, so their first uses then walk the leaf index monotonically (row-major variant)
  uint32_t n_land = 0;            // experiment (FDG_LAND): of the n_acc AGPR pairs, this many are landing slots for leaf loads issued up to
  uint32_t lookahead_land = 1500; // lookahead_land ops ahead -- no VGPR pair is tied up by a load in flight, 2 KB more in flight per slot and CU.
                                  // Measured neutral to -5 % (16-48 slots) on every one-wave kernel: their waits are not for leaves that were
                                  // asked for too late (one wave per SIMD only; not for the row-major or Monte-Carlo programs)
  bool keep_minus_one = false;    // schedules for single-precision element types: `g * -1.0` stays a multiplication (it promotes the value to
                                  // Float64 in the reference's generic function; as a sign on the operand it would not)
  bool mul_keeps_signs = false;   // schedules for ComplexF64 values: a product consumes a negated operand as it is instead of pulling the sign
                                  // out -- ((-z) w) and -(z w) differ in the sign of a real part that cancels exactly
  bool acc_in_agpr = false;       // fused-accumulation kernels of graphs with 41 ... 124 roots: the per-lane accumulators live in AGPR pairs (above the
                                  // program's own), three VGPR pairs (weight, two temporaries) above the values; one wave per SIMD
  bool pool_leaves = false;       // pooled cooperative programs: leaves are re-read from the shared LDS pool (cheap to evict, never parked)
  bool leaves_once = false;       // row-major programs (round 6): a leaf is loaded ONCE, while its 16-leaf chunk of the rows sits in a staging buffer, and is
                                  // from then on a value like any other -- evicted to an LDS slot, an AGPR pair or the panel (512 bytes, coalesced) instead of
                                  // being dropped and "re-loaded", which for a row-major matrix means fetching its whole chunk (8 KB) again or gathering
                                  // 64 cache lines for 64 doubles (parquet_sigma5 row-major: 33 chunk fetches + 13 gathers for 18 chunks)
  bool rm_pair = false;           // row-major programs with four staging buffers: chunks are fetched in pairs (csrc/fdg_isa.cpp: rm_plan)
  uint32_t reserve_pairs = 0;     // VGPR pairs the kernel variant keeps above the values (accumulators, weight): the value budget shrinks
                                  // by this and by the temporaries the program's macro ops need, so that everything stays below v256
};

struct OptProgram {
  OptParams params;
  std::vector<MOp> ops;
  uint32_t n_reg_used = 0, n_lds_used = 0, n_mem_used = 0, n_acc_used = 0;
  // statistics
  uint64_t n_valu = 0, n_ld_leaf = 0, n_ld_lds = 0, n_ld_mem = 0, n_st_lds = 0, n_st_mem = 0, n_ld_acc = 0, n_st_acc = 0;
  uint64_t n_send = 0, n_recv = 0, n_barrier = 0;   // cooperative programs
  uint64_t n_ld_land = 0;                            // leaf loads that went through a landing slot (counted in n_ld_leaf too)
  uint32_t max_live = 0;
  uint32_t mc_n_k = 0, mc_n_t = 0;   // build_mc_program: input columns 0..mc_n_k-1 are momentum components, the next mc_n_t are times
  bool supported = true;    // false: graph uses something the ISA path does not cover
  std::string why;
};

void build_opt_program(const Lowered &p, const OptParams &prm, OptProgram &out);

// The Monte-Carlo step as ONE program: the kernel's inputs are the sample's momentum components and times
// (input column c < n_k: K component c; n_k + i: time i+1), and every leaf of the graph is a value computed from
// them by the formulas of example/benchmark.jl:58-127 (green / green_derive / the Yukawa interaction and its
// counter-terms) at its first use.  kF, beta, lambda enter through four tagged constants (MOp::param) that the
// kernel takes as arguments; the values in LeafSpec only fill MOp::imm for host-side replay.
struct LeafSpec {
  const fdg_leaf_tables *tab = nullptr;
  double kF = 0, beta = 0, lambda = 0;
};
void build_mc_program(const Lowered &p, const LeafSpec &ls, const OptParams &prm, OptProgram &out);

// The scheduled, value-numbered fold steps before register allocation, for back ends that leave registers to a
// compiler: values are numbered 0..n_value-1 (leaves first, in leaf order); an operand reference is
// (value << 1) | negate.  kind: M_MUL d = a*b, M_ADD d = a+b, M_MULC d = a*imm, M_ROOT root[d] = a.
struct SchedOp { uint8_t kind; uint32_t d, a, b; double imm; };
bool build_schedule(const Lowered &p, const OptParams &prm, std::vector<SchedOp> &ops, uint32_t &n_value, std::string &why);
struct CoopProgram;
std::string emit_isa(const Lowered &p, const OptProgram &prog, const std::string &kname, const OptProgram *prog2 = nullptr,
                     const OptProgram *prog_acc = nullptr, const OptProgram *prog_rm = nullptr, uint32_t rm_bufs = 0,
                     const CoopProgram *coop = nullptr, const OptProgram *prog_rm_acc = nullptr, const CoopProgram *pool = nullptr,
                     const OptProgram *prog_rl = nullptr, const OptProgram *prog_rl_acc = nullptr);

// Cooperative variant: the four waves of a CU (one per SIMD) evaluate ONE 64-sample tile together.  Each wave runs its own
// straight-line program on its share of the graph with its own registers, AGPRs, private LDS slots and panel; a value
// another wave needs is published into a shared LDS slot (M_SEND) and becomes readable after the next M_BARRIER (M_RECV).
// Per-sample on-chip state is four times that of the one-wave kernels -- the design point for graphs whose live set
// overflows one lane.  Every wave's program contains the same number of barriers.
struct CoopProgram {
  static constexpr uint32_t MAXW = 16;
  uint32_t n_wave = 4;         // 4: one wave per SIMD, AGPRs as a spill level; 8: two per SIMD (256 registers each, no AGPR level)
  OptProgram wave[MAXW];
  uint32_t n_shared = 0;       // shared LDS slots (of 512 bytes) in front of the waves' private ones
  uint32_t n_priv_lds = 0;     // private LDS slots of each wave
  uint32_t n_epoch = 0;        // barriers per tile
  uint64_t n_transfer = 0;     // values handed over per tile
  uint64_t n_duplicate = 0;    // fold steps computed by more than one wave
  bool pooled = false;         // build_pool_program: leaves come through the shared pool (M_POOL_FETCH / M_RECV); needs sample stride 1, leaf
                               // offsets below 2^31 bytes from the tile's base (tile-major batches) and full 64-sample tiles
  uint32_t pool_unit = 1;      // pooled: leaves per fetch (in: set before build_pool_program; 2 = pairs of adjacent leaves, needs leaf stride 64: tile-major batches)
  uint32_t slack = 0;          // pooled, round 6 (FDG_POOL_SYNC=flags): > 0 = the waves do not meet at s_barrier between the epochs of a tile; every wave publishes
                               // the number of sync points it has reached in an LDS word of its own and passes sync point b when every other wave has reached
                               // b (the first slack + 1 sync points of a tile: strict) or max(slack + 1, b - slack) (M_BARRIER with a = that number, b = the
                               // sync's index; a = 0: a real s_barrier -- the last of a tile).  The
                               // planner keeps a leaf `slack` epochs longer on both sides: fetched so that it has landed `slack` epochs before its first read,
                               // its slot given away no earlier than `slack` epochs after its last
  uint64_t n_fetch = 0;        // pooled: leaf fetches from memory per tile (>= the live leaves; what exceeds them was evicted from the pool and came again)
  bool supported = false;
  std::string why;
};
void build_coop_program(const Lowered &p, const OptParams &prm, CoopProgram &out, uint32_t n_wave = 4);
// Pooled cooperative variant: whole roots are dealt to the waves of a CU (balanced by the fold steps each adds), every wave runs the ordinary
// depth-first schedule on its roots, and NO wave loads a leaf from memory into a register: leaves are fetched into a shared LDS pool by
// LDS-direct loads `ahead` epochs before their first reader needs them and stay there while any wave still reads them (offline Belady over
// the merged read sequence of all waves at epoch granularity).  The vertex functions of the reference's two benchmark programs re-read their
// leaves 2-3 times in the one-wave kernels because 324 on-chip values per sample cannot hold their 1000-1500 leaves; a CU's eight waves and the
// pool together hold 1200.
void build_pool_program(const Lowered &p, const OptParams &prm, CoopProgram &out, uint32_t n_wave = 8, uint32_t epoch_ops = 0, uint32_t ahead = 0);

void rm_plan_stats(const Lowered &p, const OptProgram &prog, uint32_t bufs, uint64_t &fetches, uint64_t &gathers);
// gfx950 wait-state table of the emitter (fdg_isa.cpp): check of a finished listing, and the table as text
int check_isa_hazards(const std::string &text, std::string &report);
std::string isa_hazard_table();

}  // namespace fdg
