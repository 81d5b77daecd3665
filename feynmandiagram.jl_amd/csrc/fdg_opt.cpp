// See fdg_opt.h.  Everything here is host-side graph compilation; it decides
// *where* values live and *when* they move, never what is computed: each
// micro-op is one IEEE fp64 add or multiply of the reference's left fold
// (src/backend/static.jl:13-31), and a factor of -1 is carried as a sign on the
// operand (x * -1.0 == -x exactly).
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <deque>
#include <limits>
#include <map>
#include <memory>
#include <unordered_map>

#include "fdg_opt.h"

namespace fdg {
namespace {

constexpr uint32_t NONE = 0xFFFFFFFFu;

struct UOp {      // micro-op over virtual values; refs are (vid << 1) | neg
  uint8_t kind;   // M_MUL / M_ADD / M_MULC / M_ROOT
  uint32_t d;     // destination vid (root index for M_ROOT)
  uint32_t a, b;  // operand refs (b unused for MULC / ROOT)
  double imm;
  uint32_t c = 0; // addend ref of M_FMA / M_FMAC
  uint8_t param = 0;   // MOp::param
};

struct Builder {
  const Lowered &p;
  std::vector<UOp> u;
  uint32_t next_vid;
  std::vector<uint32_t> ref_of;   // table value -> ref, NONE when not computed yet
  bool ok = true;
  std::string why;

  // Monte-Carlo program (build_mc_program): values in_base .. in_base+n_in-1 are the kernel's input columns
  // (momentum components, then times); the graph's leaves are computed values, built at their first use
  const LeafSpec *mc = nullptr;
  uint32_t in_base = 0, n_in = 0, n_k = 0;

  explicit Builder(const Lowered &p_, const LeafSpec *mc_ = nullptr) : p(p_), next_vid(p_.L), ref_of((size_t)p_.L + p_.N, NONE), mc(mc_) {
    if (mc) {
      in_base = p.L;
      n_k = mc->tab->n_loop * mc->tab->dim;
      n_in = n_k + mc->tab->n_tau;
      next_vid = p.L + n_in;
      born.assign(next_vid, 0);
      return;
    }
    for (uint32_t i = 0; i < p.L; ++i) ref_of[i] = i << 1;
    born.assign(p.L, 0);
  }
  // Value numbering: the reference's straight-line code repeats identical fold steps (products that
  // share a prefix, `g * -1.0` of the same g, ...).  An op with the same kind and the same operands
  // yields the same bits, so it is computed once.  Exact identities used, nothing else:
  //   x*y == y*x, x+y == y+x (IEEE, any zero/inf), (-x)*y == -(x*y), (-x)*c == -(x*c).
  // Sums are NOT canonicalised across negation ((-x)+y vs -(x-y) differ in the sign of an exact zero).
  std::unordered_map<uint64_t, uint32_t> vn_mul, vn_add;
  std::unordered_map<uint64_t, std::vector<std::pair<double, uint32_t>>> vn_mulc;
  bool value_numbering = true;
  uint64_t vn_window = 0;        // reuse only results computed at most this many ops ago (0 = no limit)
  std::vector<uint32_t> born;    // vid -> index of the op that produced it or (vn_touch) last read it
  bool vn_touch = true;          // a value stays reusable while it keeps being read (it is live anyway)
  bool fresh_enough(uint32_t ref) const {
    if (!vn_window) return true;
    const uint32_t v = ref >> 1;
    return v < born.size() && (uint64_t)u.size() - born[v] <= vn_window;
  }
  void touch(uint32_t ref) { if (vn_touch) { const uint32_t v = ref >> 1; if (v < born.size()) born[v] = (uint32_t)u.size(); } }
  uint32_t fresh() { born.resize(next_vid + 1, 0); born[next_vid] = (uint32_t)u.size(); return next_vid++; }
  // Rematerialisation: the value of a cheap node (own fold of at most remat_cost steps) that has not been read for
  // remat_window ops is forgotten; its next consumer computes it again (the same IEEE operations on the same operands:
  // the same bits).  A widely shared value whose uses are thousands of ops apart would otherwise sit in a spill slot and
  // come back through the HBM panel for every use; its operands -- lower-order sub-diagrams that many nodes keep reading --
  // usually are still on chip.  Roots are exempt.
  bool keep_root_order = false;
  const std::vector<uint32_t> *root_rank = nullptr;  // pool programs (experiment FDG_POOL_SEED): [N] position of a root node in this wave's order of evaluation
  const std::vector<uint8_t> *root_mask = nullptr;   // pool programs: [R] the roots this wave computes and stores (others are not its business)
  uint32_t term_window = 1, term_recent = 400;    // out-of-order evaluation of a wide node's terms (see build_uops)
  uint64_t remat_window = 0;
  uint32_t remat_cost = 8;
  std::vector<uint8_t> remat_ok;   // [N]
  uint64_t n_remat = 0;
  uint64_t remat_limit = 0;        // recomputation can cascade (a forgotten node whose operands are forgotten too ...): once the
                                   // program has grown to this many ops nothing is forgotten any more
  bool available(uint32_t c) const {
    if (ref_of[c] == NONE) return false;
    if (!remat_window || c < p.L || !remat_ok[c - p.L] || u.size() > remat_limit) return true;
    const uint32_t v = ref_of[c] >> 1;
    return v >= born.size() || (uint64_t)u.size() - born[v] <= remat_window;
  }
  uint32_t op2(uint8_t k, uint32_t a, uint32_t b) {
    if (!value_numbering) { uint32_t d = fresh(); u.push_back(UOp{k, d, a, b, 0.0}); return d << 1; }
    if (k == M_MUL && mul_keeps_signs) {
      uint32_t x = a, y = b;
      if (x > y) std::swap(x, y);
      const uint64_t key = ((uint64_t)x << 32) | y;
      auto it = vn_mul.find(key);
      if (it != vn_mul.end() && fresh_enough(it->second)) { touch(it->second); return it->second; }
      touch(x); touch(y);
      uint32_t d = fresh();
      u.push_back(UOp{M_MUL, d, x, y, 0.0});
      vn_mul[key] = d << 1;
      return d << 1;
    }
    if (k == M_MUL) {
      const uint32_t sign = (a ^ b) & 1u;
      uint32_t x = a & ~1u, y = b & ~1u;
      if (x > y) std::swap(x, y);
      const uint64_t key = ((uint64_t)x << 32) | y;
      auto it = vn_mul.find(key);
      if (it != vn_mul.end() && fresh_enough(it->second)) { touch(it->second); return it->second | sign; }
      touch(x); touch(y);
      uint32_t d = fresh();
      u.push_back(UOp{M_MUL, d, x, y, 0.0});
      vn_mul[key] = d << 1;
      return (d << 1) | sign;
    }
    uint32_t x = a, y = b;
    if (x > y) std::swap(x, y);
    const uint64_t key = ((uint64_t)x << 32) | y;
    auto it = vn_add.find(key);
    if (it != vn_add.end() && fresh_enough(it->second)) { touch(it->second); return it->second; }
    touch(x); touch(y);
    uint32_t d = fresh();
    u.push_back(UOp{M_ADD, d, x, y, 0.0});
    vn_add[key] = d << 1;
    return d << 1;
  }
  bool keep_minus_one = false;     // OptParams::keep_minus_one
  bool mul_keeps_signs = false;    // OptParams::mul_keeps_signs
  uint32_t mulc(uint32_t a, double f) {
    if (f == 1.0) return a;
    if (f == -1.0 && !keep_minus_one) return a ^ 1u;
    if (!value_numbering) { uint32_t d = fresh(); u.push_back(UOp{M_MULC, d, a, 0, f}); return d << 1; }
    const uint32_t sign = a & 1u, x = a & ~1u;
    auto &lst = vn_mulc[x];
    touch(x);
    for (auto &e : lst) if (std::memcmp(&e.first, &f, 8) == 0) {
      if (fresh_enough(e.second)) { touch(e.second); return e.second | sign; }
      uint32_t d = fresh();
      u.push_back(UOp{M_MULC, d, x, 0, f});
      e.second = d << 1;
      return (d << 1) | sign;
    }
    uint32_t d = fresh();
    u.push_back(UOp{M_MULC, d, x, 0, f});
    lst.push_back({f, d << 1});
    return (d << 1) | sign;
  }

  // ---- leaf formulas (Monte-Carlo program) ---------------------------------------------------------------
  // Any other op kind, computed once per distinct operand tuple (these values are few and costly: no window).
  std::map<std::array<uint64_t, 3>, uint32_t> vn_x;
  uint32_t opx(uint8_t k, uint32_t a, uint32_t b, uint32_t c, double imm, uint8_t param = 0) {
    uint64_t ib; std::memcpy(&ib, &imm, 8);
    if (param) ib = 0x7ff8000000000000ull | param;      // a parameter is identified by its tag, not by its present value
    const std::array<uint64_t, 3> key = {((uint64_t)k << 32) | a, ((uint64_t)b << 32) | c, ib};
    auto it = vn_x.find(key);
    if (it != vn_x.end()) { touch(it->second); return it->second; }
    touch(a); if (mop_has_b(k)) touch(b); if (mop_has_c(k)) touch(c);
    const uint32_t d = fresh();
    UOp o{k, d, a, b, imm}; o.c = c; o.param = param;
    u.push_back(o);
    vn_x[key] = d << 1;
    return d << 1;
  }
  uint32_t addc(uint32_t a, double f, uint8_t param = 0) { return opx(M_ADDC, a, 0, 0, f, param); }
  uint32_t mulp(uint32_t a, double f, uint8_t param) { return opx(M_MULC, a & ~1u, 0, 0, f, param) | (a & 1u); }   // times a parameter
  // 1/(-x) == -(1/x).  A correctly rounded division (the oracle's and the leaf kernels' `1.0 / x`), not v_rcp_f64 + Newton
  // steps: five more instructions per quotient, and the quotients are then the reference's bits (FDG_MC_RCP_NEWTON=1: the
  // round-1 form, within an ulp)
  uint32_t rcp(uint32_t a) {
    const bool newton = fdg::knob("FDG_MC_RCP_NEWTON") != nullptr;
    return opx(newton ? M_RCP : M_DIV1, a & ~1u, 0, 0, 0.0) | (a & 1u);
  }
  // cond(c) ? a : b, cond = c > 0 (ge false) or c >= 0
  uint32_t sel(uint32_t c, uint32_t a, uint32_t b, bool ge) {
    if (a == b) return a;
    if ((a & 1u) && (b & 1u)) return opx(M_SEL, a ^ 1u, b ^ 1u, c, ge ? 1.0 : 0.0) ^ 1u;
    return opx(M_SEL, a, b, c, ge ? 1.0 : 0.0);
  }
  uint32_t mul(uint32_t a, uint32_t b) { return op2(M_MUL, a, b); }
  uint32_t add(uint32_t a, uint32_t b) { return op2(M_ADD, a, b); }
  uint32_t in_k(uint32_t c) const { return (in_base + c) << 1; }
  uint32_t in_t(int32_t i) const { return (in_base + n_k + (uint32_t)(i - 1)) << 1; }

  // ---- integer powers beyond x*x and x*x*x: Base.Math.pow_body spelled out (csrc/fdg_powi.h is the scalar form) ----
  // A quantity is a value of the program or a constant known here (y = 1, the low words = 0 until the loop fills them);
  // operations on constants alone are done here, in the same IEEE arithmetic the device uses.
  struct PV { bool k; double c; uint32_t r; };
  static PV pv(uint32_t r) { return PV{false, 0.0, r}; }
  static PV pk(double c) { return PV{true, c, 0}; }
  static PV pneg(PV a) { return a.k ? pk(-a.c) : pv(a.r ^ 1u); }
  PV pmul(PV a, PV b) {
    if (a.k && b.k) return pk(a.c * b.c);
    if (a.k) std::swap(a, b);
    if (b.k) return pv(mulc(a.r, b.c));                 // x * 1.0 == x, x * -1.0 == -x exactly: mulc's shortcuts are safe
    return pv(mul(a.r, b.r));
  }
  PV padd(PV a, PV b) {
    if (a.k && b.k) return pk(a.c + b.c);
    if (a.k) std::swap(a, b);
    if (b.k) return pv(opx(M_ADDC, a.r, 0, 0, b.c));    // never elided: (-0) + 0 is +0
    return pv(add(a.r, b.r));
  }
  PV pfma(PV a, PV b, PV c) {                           // a * b + c, one rounding
    if (a.k && b.k) {
      const double prod = a.c * b.c;
      if (std::fma(a.c, b.c, -prod) != 0.0 || !std::isfinite(prod)) { ok = false; why = "pow_body: inexact constant product"; return c; }
      return padd(c, pk(prod));                          // an exact product: fma(a, b, c) == (a b) + c
    }
    if (a.k) std::swap(a, b);
    if (b.k) {
      if (c.k) {
        if (b.c == 1.0) return padd(a, c);               // fma(x, 1, c) == x + c
        ok = false; why = "pow_body: x * const + const"; return a;
      }
      return pv(opx(M_FMAC, a.r, 0, c.r, b.c));
    }
    if (c.k) return pv(opx(M_FMAK, a.r, b.r, 0, c.c));
    return pv(opx(M_FMA, a.r, b.r, c.r, 0.0));
  }
  PV pdiv1(PV x) { return pv(opx(M_DIV1, x.r & ~1u, 0, 0, 0.0) | (x.r & 1u)); }       // 1 / (-x) == -(1 / x)
  PV psel_finite(PV test, PV a, PV b, uint32_t anchor) {   // isfinite(test) ? a : b
    if (a.k) a = pv(opx(M_CONST, anchor & ~1u, 0, 0, a.c));
    if (b.k) b = pv(opx(M_CONST, anchor & ~1u, 0, 0, b.c));
    if (a.r == b.r) return a;
    return pv(opx(M_SEL, a.r, b.r, test.r & ~1u, 2.0));
  }
  // value of x^n for a literal n as the reference's generated code computes it (static.jl:45: `(g)^N`; Base.literal_pow, then
  // pow_body for every n it has no shortcut for)
  uint32_t powi(uint32_t xr, int32_t n) {
    if (n == 2) return mul(xr, xr);
    if (n == 3) return mul(mul(xr, xr), xr);
    PV x = pv(xr), y = pk(1.0), xnlo = pk(0.0), ynlo = pk(0.0);
    if (n == -1) return pdiv1(x).r;
    if (n == -2) { const PV r = pdiv1(x); return pmul(r, r).r; }
    int64_t m = n;
    if (m < 0) {
      const PV rx = pdiv1(x);
      const PV lo = pmul(pneg(pfma(x, rx, pk(-1.0))), rx);
      xnlo = psel_finite(x, lo, pk(0.0), xr);             // if (isfinite(x)) xnlo = -fma(x, rx, -1.0) * rx
      x = rx;
      m = -m;
    }
    while (m > 1) {
      if (m & 1) {
        const PV err = pfma(y, xnlo, pmul(x, ynlo));
        const PV yh = pmul(x, y), yl = pfma(x, y, pneg(yh));
        y = yh;
        ynlo = padd(yl, err);
      }
      const PV err = pmul(pmul(x, pk(2.0)), xnlo);
      const PV xh = pmul(x, x), xl = pfma(x, x, pneg(xh));
      x = xh;
      xnlo = padd(xl, err);
      m >>= 1;
    }
    const PV err = pfma(y, xnlo, pmul(x, ynlo));
    const PV r1 = pfma(x, y, err), r2 = pmul(x, y);
    // (isfinite(x) && isfinite(err)) ? fma(x, y, err) : x * y
    const PV inner = err.k ? (std::isfinite(err.c) ? r1 : r2) : psel_finite(err, r1, r2, xr);
    return psel_finite(x, inner, r2, xr).r;
  }

  struct Mom { uint32_t q2 = NONE, w = NONE, g = NONE, bsel = NONE; };
  struct Tau { uint32_t tf = NONE, u = NONE, v = NONE; };
  std::map<int32_t, Mom> moms;
  std::map<std::pair<int32_t, int32_t>, Tau> taus;

  Mom &momentum(int32_t li) {
    Mom &m = moms[li];
    if (m.q2 != NONE) return m;
    const fdg_leaf_tables *t = mc->tab;
    const double *bv = t->basis + (size_t)(li - 1) * t->n_loop;
    uint32_t q2 = NONE;
    for (uint32_t d = 0; d < t->dim; ++d) {                       // q2 = sum_d (sum_j k[j][d] * basis[j])^2   (benchmark.jl:113-117)
      uint32_t q = NONE;
      for (uint32_t j = 0; j < t->n_loop; ++j) {
        if (bv[j] == 0.0) continue;
        const uint32_t term = mulc(in_k(j * t->dim + d), bv[j]);
        q = q == NONE ? term : add(q, term);
      }
      if (q == NONE) { ok = false; why = "a leaf with zero momentum"; m.q2 = in_k(0); return m; }
      const uint32_t sq = mul(q, q);
      q2 = q2 == NONE ? sq : add(q2, sq);
    }
    m.q2 = q2;
    return m;
  }
  // dispersion and the Fermi factor of a momentum: w = q2 - kF^2, g = 1 / (1 + exp(-|w| beta))
  Mom &fermi(int32_t li) {
    Mom &m = momentum(li);
    if (m.w != NONE || !ok) return m;
    m.w = addc(m.q2, -(mc->kF * mc->kF), MC_P_NEG_KF2);
    const uint32_t wb = mulp(m.w, mc->beta, MC_P_BETA);
    const uint32_t e = opx(M_EXP, sel(m.w, wb ^ 1u, wb, false), 0, 0, 0.0);
    m.g = rcp(addc(e, 1.0));
    return m;
  }
  Tau &tau_pair(int32_t tin, int32_t tout) {
    Tau &t = taus[{tin, tout}];
    if (t.tf != NONE) return t;
    const uint32_t tau = add(in_t(tout), in_t(tin) ^ 1u);
    t.tf = opx(M_FIXZ, tau, 0, 0, -1e-10);                           // benchmark.jl:98 / green(): tau == 0 -> -1e-10
    t.u = sel(t.tf, t.tf, addc(t.tf, mc->beta, MC_P_BETA), false);             // w >= 0 branch: a = -(tau > 0 ? tau : tau + beta)
    t.v = sel(t.tf, addc(t.tf, -mc->beta, MC_P_NEG_BETA), t.tf, false);            // w <  0 branch: a = -(tau > 0 ? tau - beta : tau)
    return t;
  }
  // value of table leaf i (the expressions of fdg_leaf.hip, every exponential with a non-positive argument)
  uint32_t leaf_formula(uint32_t i) {
    const fdg_leaf_tables *t = mc->tab;
    const int32_t ty = t->leaf_type[i], n = t->leaf_order[i], li = t->loop_index[i];
    if (ty == 2) {                                                  // 8 pi / invK * (lambda invK)^n, invK = 1 / (q2 + lambda)
      if (n < 0) { ok = false; why = "interaction counter-term of negative order"; return in_k(0); }
      Mom &m = momentum(li);
      if (!ok) return in_k(0);
      const uint32_t s = addc(m.q2, mc->lambda, MC_P_LAMBDA);
      uint32_t v = mulc(s, 8.0 * 3.141592653589793);
      if (n) {
        const uint32_t x = mulp(rcp(s), mc->lambda, MC_P_LAMBDA);
        v = mul(v, n == 1 ? x : powi(x, n));                        // (lambda invK)^n: literal_pow for n <= 3, pow_body above (as fdg_leaf.hip)
      }
      return v;
    }
    if (ty != 1) return addc(mulc(in_k(0), 0.0), 1.0);             // no formula: leafstates' initial leafValue 1.0  ((+-0) + 1.0 == 1.0)
    if (n < 0 || n > 5) { ok = false; why = "green_derive order above 5"; return in_k(0); }
    Mom &m = fermi(li);
    if (!ok) return in_k(0);
    Tau &tp = tau_pair(t->tau_in[i], t->tau_out[i]);
    const uint32_t a = sel(m.w, tp.u, tp.v, n != 0) ^ 1u;           // green() tests w > 0, green_derive's twin w >= 0
    const uint32_t A = opx(M_EXP, mul(m.w, a), 0, 0, 0.0);
    if (const char *dbg = fdg::knob("FDG_MC_DEBUG_STAGE")) {   // development only: a leaf's intermediate instead of its value
      const std::string st = dbg;
      const std::pair<const char *, uint32_t> stages[] = {{"w", m.w}, {"g", m.g}, {"tau", tp.tf}, {"a", a}, {"A", A}, {"u", tp.u}, {"v", tp.v}};
      for (const auto &kv : stages) if (st == kv.first) return kv.second;
      if (st == "wa") return mul(m.w, a);
    }
    uint32_t x;
    if (n == 0) {
      x = mul(A, m.g);
    } else {
      // fdg_fermi_dn: sum_k C(n,k) Q_k(g) a^(n-k) b^k, b = +-beta, Q_k by Horner
      static const double QC[6][7] = {{0, 1, 0, 0, 0, 0, 0}, {0, 1, -1, 0, 0, 0, 0}, {0, 1, -3, 2, 0, 0, 0}, {0, 1, -7, 12, -6, 0, 0},
                                      {0, 1, -15, 50, -60, 24, 0}, {0, 1, -31, 180, -390, 360, -120}};
      static const double BC[6][6] = {{1, 0, 0, 0, 0, 0}, {1, 1, 0, 0, 0, 0}, {1, 2, 1, 0, 0, 0}, {1, 3, 3, 1, 0, 0}, {1, 4, 6, 4, 1, 0}, {1, 5, 10, 10, 5, 1}};
      static const double NF[6] = {1.0, -1.0, 0.5, -1.0 / 6.0, 1.0 / 24.0, -1.0 / 120.0};
      if (m.bsel == NONE) m.bsel = opx(M_SELC, m.w, 1, 0, mc->beta, MC_P_BETA);     // b = w >= 0 ? beta : -beta  (operand b = 1: the >= form)
      uint32_t total = NONE;
      for (int k = 0; k <= n; ++k) {
        uint32_t q = mulc(m.g, QC[k][k + 1]);
        if (QC[k][k] != 0.0) q = addc(q, QC[k][k]);
        for (int c = k - 1; c >= 0; --c) { q = mul(q, m.g); if (QC[k][c] != 0.0) q = addc(q, QC[k][c]); }
        uint32_t term = mulc(q, BC[n][k]);
        for (int j = 0; j < n - k; ++j) term = mul(term, a);
        for (int j = 0; j < k; ++j) term = mul(term, m.bsel);
        total = total == NONE ? term : add(total, term);
      }
      x = mulc(mul(A, total), NF[n]);
    }
    return mul(x, opx(M_SELC, tp.tf, 0, 0, 1.0));                   // antiperiodicity: times the sign of tau (+-1.0, exact)
  }
  // operand `c` of the table is about to be read
  void need(uint32_t c) {
    if (!mc || c >= p.L || ref_of[c] != NONE) return;
    ref_of[c] = leaf_formula(c);
  }
};

struct Frame { uint32_t n, i, acc; };

// Order in which the roots are evaluated (values do not depend on it): start with the root that
// needs the most nodes, then always continue with the root whose cone is already computed to the
// largest extent -- graphs that share most of their sub-expressions (Taylor coefficients of one
// diagram, instant/dynamic parts, the rows of a vertex function) are finished while the shared values are still on chip.
// (Bit sets: 180 roots x 45 000 nodes -- the 4-loop vertex function of example/benchmark.jl -- take 0.1 s.  With
// FDG_ROOT_RECENT=w the overlap counts only the cones of the last w roots: what is likely to be on chip still.)
static void order_roots(const Lowered &p, std::vector<uint32_t> &tops) {
  const uint32_t L = p.L;
  const size_t max_roots = fdg::knob("FDG_ROOT_ORDER_MAX") ? (size_t)std::atoi(fdg::knob("FDG_ROOT_ORDER_MAX")) : 1024;
  if (tops.size() > 1 && tops.size() <= max_roots) {
    const size_t R = tops.size(), W = (p.N + 63) / 64;
    std::vector<std::vector<uint64_t>> cone(R, std::vector<uint64_t>(W, 0));
    std::vector<uint64_t> csize(R, 0);
    std::vector<uint32_t> stk;
    for (size_t r = 0; r < R; ++r) {
      stk.assign(1, tops[r]);
      cone[r][tops[r] >> 6] |= 1ull << (tops[r] & 63);
      while (!stk.empty()) {
        const uint32_t n = stk.back(); stk.pop_back();
        csize[r]++;
        for (uint32_t e = p.off[n]; e < p.off[n + 1]; ++e) {
          const uint32_t c = p.idx[e];
          if (c >= L && !((cone[r][(c - L) >> 6] >> ((c - L) & 63)) & 1)) { cone[r][(c - L) >> 6] |= 1ull << ((c - L) & 63); stk.push_back(c - L); }
        }
      }
    }
    const size_t recent = fdg::knob("FDG_ROOT_RECENT") ? (size_t)std::atoi(fdg::knob("FDG_ROOT_RECENT")) : 0;
    std::vector<uint64_t> done(W, 0);
    std::vector<uint8_t> used(R, 0);
    std::vector<uint32_t> ordered;
    std::vector<size_t> order_idx;
    for (size_t k = 0; k < R; ++k) {
      if (recent && k) {                       // only what the last `recent` roots touched
        std::fill(done.begin(), done.end(), 0);
        for (size_t q = order_idx.size() > recent ? order_idx.size() - recent : 0; q < order_idx.size(); ++q)
          for (size_t w = 0; w < W; ++w) done[w] |= cone[order_idx[q]][w];
      }
      size_t best = R; double best_score = -1.0;
      for (size_t r = 0; r < R; ++r) {
        if (used[r]) continue;
        uint64_t ov = 0;
        if (k) for (size_t w = 0; w < W; ++w) ov += (uint64_t)__builtin_popcountll(cone[r][w] & done[w]);
        const double score = k ? (double)ov / (double)csize[r] + 1e-9 * (double)csize[r] / (double)p.N : (double)csize[r];
        if (score > best_score) { best_score = score; best = r; }
      }
      used[best] = 1;
      ordered.push_back(tops[best]);
      order_idx.push_back(best);
      if (!recent) for (size_t w = 0; w < W; ++w) done[w] |= cone[best][w];
    }
    tops.swap(ordered);
}

}

void build_uops(Builder &B) {
  const Lowered &p = B.p;
  const uint32_t L = p.L;
  std::vector<std::vector<uint32_t>> roots_of((size_t)0);
  std::vector<std::pair<uint32_t, uint32_t>> rootlist;
  for (uint32_t k = 0; k < p.R; ++k)
    if (p.root_slot[k] != FDG_NO_ROOT && (!B.root_mask || (*B.root_mask)[k])) rootlist.push_back({p.root_slot[k], k});
  std::sort(rootlist.begin(), rootlist.end());
  auto emit_roots = [&](uint32_t v) {
    auto it = std::lower_bound(rootlist.begin(), rootlist.end(), std::make_pair(v, 0u));
    for (; it != rootlist.end() && it->first == v; ++it) B.u.push_back(UOp{M_ROOT, it->second, B.ref_of[v], 0, 0.0});
  };
  for (auto &rk : rootlist)
    if (rk.first < L) { B.need(rk.first); B.u.push_back(UOp{M_ROOT, rk.second, B.ref_of[rk.first], 0, 0.0}); }

  std::vector<Frame> st;
  // roots in statement order of the reference (increasing node index)
  std::vector<uint32_t> tops;
  for (auto &rk : rootlist) if (rk.first >= L) tops.push_back(rk.first - L);
  tops.erase(std::unique(tops.begin(), tops.end()), tops.end());
  if (B.root_rank) std::stable_sort(tops.begin(), tops.end(), [&](uint32_t x, uint32_t y) { return (*B.root_rank)[x] < (*B.root_rank)[y]; });
  else if (!B.keep_root_order) order_roots(p, tops);
  // One fold step of frame f (operand already computed).  Returns true when the node is finished.
  auto step = [&](Frame &f) -> bool {
    const uint32_t n = f.n, a = p.off[n], k = p.off[n + 1] - a;
    const uint32_t c = p.idx[a + f.i];
    const uint32_t cr = B.ref_of[c];
    const double fc = p.fac[a + f.i];
    uint32_t acc = f.acc;
    B.touch(cr);
    if (p.op[n] == FDG_OP_SUM) {
      const uint32_t t = B.mulc(cr, fc);                       // c_i * f_i   (static.jl:18)
      acc = (f.i == 0) ? t : B.op2(M_ADD, acc, t);
    } else if (p.op[n] == FDG_OP_PROD && p.assoc_interp) {
      const uint32_t t = B.mulc(cr, fc);                       // acc * (c_i * f_i)    (eval.jl:2: prod(w * f)); f = 1 and f = -1
      acc = (f.i == 0) ? t : B.op2(M_MUL, acc, t);             // leave the bits of w * f as they are (x * 1.0 == x, x * -1.0 == -x)
    } else if (p.op[n] == FDG_OP_PROD) {
      acc = (f.i == 0) ? cr : B.op2(M_MUL, acc, cr);           // ((acc * c_i) * f_i)  (static.jl:28)
      acc = B.mulc(acc, fc);
    } else {  // Power: exactly one child
      acc = B.powi(cr, p.power[n]);                            // (c)^N   (static.jl:34-46)
      acc = B.mulc(acc, fc);
    }
    f.acc = acc;
    f.i++;
    if (f.i < k) return false;
    B.ref_of[L + n] = f.acc;
    emit_roots(L + n);
    return true;
  };

  // plain depth-first evaluation of one node; also how a forgotten value is computed again (see Builder::available)
  auto dfs = [&](uint32_t top) {
    const size_t base = st.size();
    st.push_back(Frame{top, 0, NONE});
    while (st.size() > base) {
      Frame &f = st.back();
      const uint32_t c = p.idx[p.off[f.n] + f.i];
      B.need(c);
      if (!B.available(c)) {
        if (B.ref_of[c] != NONE) B.n_remat++;
        // Terms of a wide Sum / Prod may be COMPUTED out of order (their values wait for their turn in the left fold, which
        // stays in order): among the next `term_window` terms the one that shares most operands with what was touched
        // recently goes first, so that diagrams built from the same propagators are evaluated while those are on chip.
        uint32_t pick = c;
        const uint32_t k = p.off[f.n + 1] - p.off[f.n];
        if (B.term_window > 1 && k >= 16 && B.ref_of[c] == NONE) {
          double best = -1.0;
          for (uint32_t j = f.i; j < k && j < f.i + B.term_window; ++j) {
            const uint32_t cj = p.idx[p.off[f.n] + j];
            if (cj < L || B.ref_of[cj] != NONE) continue;
            uint32_t tot = 0, rec = 0;
            auto look = [&](uint32_t v) {
              tot++;
              const uint32_t r = B.ref_of[v];
              if (r == NONE) return;
              const uint32_t vid = r >> 1;
              if (vid < B.born.size() && B.born[vid] > 0 && (uint64_t)B.u.size() - B.born[vid] < B.term_recent) rec++;
            };
            const uint32_t nj = cj - L;
            for (uint32_t e = p.off[nj]; e < p.off[nj + 1]; ++e) {
              const uint32_t v = p.idx[e];
              look(v);
              if (v >= L && B.ref_of[v] == NONE) for (uint32_t e2 = p.off[v - L]; e2 < p.off[v - L + 1]; ++e2) look(p.idx[e2]);
            }
            const double score = tot ? (double)rec / tot : 0.0;
            if (score > best + 1e-12) { best = score; pick = cj; }
          }
        }
        st.push_back(Frame{pick - L, 0, NONE});
        continue;
      }
      if (step(st.back())) st.pop_back();
    }
  };
  if (B.remat_window) {
    B.remat_limit = 4 * (p.flops_alg + p.N) + 1000;
    B.remat_ok.assign(p.N, 0);
    const char *ce = fdg::knob("FDG_REMAT_COST");
    const uint32_t max_cost = ce ? (uint32_t)std::atoi(ce) : B.remat_cost;
    for (uint32_t n = 0; n < p.N; ++n) {
      uint32_t cost = p.off[n + 1] - p.off[n] - 1;
      for (uint32_t e = p.off[n]; e < p.off[n + 1]; ++e) if (p.fac[e] != 1.0 && p.fac[e] != -1.0) cost++;
      if (p.op[n] == FDG_OP_POWER) cost += 2;
      B.remat_ok[n] = cost <= max_cost;
    }
    for (auto &rk : rootlist) if (rk.first >= L) B.remat_ok[rk.first - L] = 0;
  }

  if (p.sched_group.size() == p.N) {
    // Grouped lock-step schedule.  The producer may tag nodes that belong together (the Taylor
    // coefficients of one original node): all members of a group are evaluated together, advancing
    // one fold step each in turn, and a missing operand pulls in its whole group first.  This is a
    // depth-first walk of the *original* graph with a small vector per node, so the operands shared
    // by the members are consumed while they are still in registers.  Values do not depend on it.
    std::vector<std::vector<uint32_t>> members;
    {
      std::vector<std::pair<uint32_t, uint32_t>> gs;
      for (uint32_t n = 0; n < p.N; ++n) if (p.live[L + n]) gs.push_back({p.sched_group[n], n});
      std::sort(gs.begin(), gs.end());
      std::vector<uint32_t> dense(p.N, 0);
      for (size_t i = 0; i < gs.size(); ++i) {
        if (i == 0 || gs[i].first != gs[i - 1].first) members.emplace_back();
        members.back().push_back(gs[i].second);
        dense[gs[i].second] = (uint32_t)members.size() - 1;
      }
      struct GFrame { std::vector<Frame> fr; size_t cur; };
      std::vector<GFrame> gst;
      std::vector<uint8_t> started(p.N, 0);
      auto push_group = [&](uint32_t node, bool whole) {
        GFrame g; g.cur = 0;
        if (whole) {
          for (uint32_t m : members[dense[node]]) if (B.ref_of[L + m] == NONE && !started[m]) { g.fr.push_back(Frame{m, 0, NONE}); started[m] = 1; }
        }
        if (g.fr.empty() && B.ref_of[L + node] == NONE && !started[node]) { g.fr.push_back(Frame{node, 0, NONE}); started[node] = 1; }
        gst.push_back(std::move(g));
      };
      for (uint32_t top : tops) {
        if (B.ref_of[L + top] != NONE) continue;
        push_group(top, true);
        while (!gst.empty()) {
          GFrame &g = gst.back();
          // drop finished members
          bool any = false;
          for (auto &f : g.fr) if (B.ref_of[L + f.n] == NONE) { any = true; break; }
          if (!any) { gst.pop_back(); continue; }
          if (g.cur >= g.fr.size()) g.cur = 0;
          Frame &f = g.fr[g.cur];
          if (B.ref_of[L + f.n] != NONE) { g.cur++; continue; }
          const uint32_t c = p.idx[p.off[f.n] + f.i];
          B.need(c);
          if (B.ref_of[c] == NONE) {
            const uint32_t cn = c - L;
            if (started[cn]) {
              // operand is a member in progress further down the stack: cannot happen in a DAG unless
              // the group hint ties a node to its own descendant; finish it alone
              B.ok = false; B.why = "inconsistent schedule groups"; return;
            }
            push_group(cn, true);
            continue;
          }
          if (!B.available(c)) {         // forgotten: computed again on its own, then this member goes on
            B.n_remat++;
            dfs(c - L);
            continue;
          }
          step(f);
          g.cur++;
        }
      }
    }
    return;
  }

  for (uint32_t top : tops) {
    if (B.ref_of[L + top] != NONE) continue;
    dfs(top);
  }
}

// FDG_SPEC_FAST_MATH: a product (MUL / MULC) whose only use is as a term of an ADD is folded into that ADD as
// a fused multiply-add at the ADD's position.  One rounding instead of two: results differ from the reference's
// in the last bits (within 1e-12 of the sum's term scale), so this is off unless asked for.
void fuse_fma(std::vector<UOp> &u, uint32_t n_value) {
  std::vector<uint32_t> n_use(n_value, 0), prod(n_value, NONE);
  for (uint32_t j = 0; j < u.size(); ++j) {
    const UOp &o = u[j];
    n_use[o.a >> 1]++;
    if (mop_has_b(o.kind)) n_use[o.b >> 1]++;
    if (mop_has_c(o.kind)) n_use[o.c >> 1]++;
    if (o.kind != M_ROOT) prod[o.d] = j;
  }
  std::vector<uint8_t> dead(u.size(), 0);
  for (uint32_t j = 0; j < u.size(); ++j) {
    UOp &o = u[j];
    if (o.kind != M_ADD) continue;
    auto fusable = [&](uint32_t ref) -> int64_t {
      const uint32_t v = ref >> 1;
      if (n_use[v] != 1 || prod[v] == NONE || dead[prod[v]]) return -1;
      const uint8_t k = u[prod[v]].kind;
      // only a product computed just before: fusing moves the multiplication to the sum's position, and the
      // factors of an older product would have to stay in registers until then
      const uint32_t reach = fdg::knob("FDG_FMA_REACH") ? (uint32_t)std::atoi(fdg::knob("FDG_FMA_REACH")) : 6u;
      if (j - prod[v] > reach) return -1;
      return (k == M_MUL || k == M_MULC) ? (int64_t)prod[v] : -1;
    };
    const int64_t pa = fusable(o.a), pb = (o.b >> 1) != (o.a >> 1) ? fusable(o.b) : -1;
    if (pa < 0 && pb < 0) continue;
    const bool take_b = pb > pa;                    // the product computed last (nearest): its operands are still around
    const UOp m = u[take_b ? pb : pa];
    const uint32_t term = take_b ? o.b : o.a, other = take_b ? o.a : o.b;
    dead[take_b ? pb : pa] = 1;
    UOp f;
    f.kind = m.kind == M_MUL ? M_FMA : M_FMAC;
    f.d = o.d;
    f.a = m.a ^ (term & 1u);                        // -(x*y) == (-x)*y
    f.b = m.kind == M_MUL ? m.b : 0;
    f.imm = m.imm;
    f.param = m.param;
    f.c = other;
    o = f;
  }
  std::vector<UOp> r;
  r.reserve(u.size());
  for (uint32_t j = 0; j < u.size(); ++j) if (!dead[j]) r.push_back(u[j]);
  u.swap(r);
}

// ---------------------------------------------------------------------------
struct Alloc {
  const Lowered &p;
  const OptParams &prm;
  const std::vector<UOp> &u;
  uint32_t nv;
  std::vector<std::vector<uint32_t>> uses;   // positions per vid
  std::vector<uint32_t> up;                  // cursor into uses
  std::vector<uint32_t> reg_of;              // vid -> reg or NONE
  std::vector<uint8_t> home_kind;            // 0 none, 1 lds, 2 mem, 3 leaf source
  std::vector<uint32_t> home_slot;
  std::vector<uint32_t> owner;               // reg -> vid or NONE
  std::vector<uint32_t> lock;                // reg -> op position it is pinned for
  std::deque<uint32_t> free_regs;
  std::vector<uint32_t> free_lds, free_mem, free_acc;
  uint32_t lds_next = 0, mem_next = 0, acc_next = 0, reg_hw = 0;
  // landing slots (home_kind 6): the last n_land AGPR pairs; a leaf sits in one from its M_LD_LEAF_ACC to its move into a register
  std::vector<uint32_t> free_land;
  uint32_t n_land = 0, n_acc_spill = 0;
  const int evict_cost = fdg::knob("FDG_EVICT_COST") ? std::atoi(fdg::knob("FDG_EVICT_COST")) : 2;    // read per program (see take_reg)
  // pooled programs: a leaf comes back from the shared pool by an LDS read, so it is the cheapest thing to evict and is not parked anywhere
  const double pool_leaf_cost = fdg::knob("FDG_POOL_LEAF_COST") ? std::atof(fdg::knob("FDG_POOL_LEAF_COST")) : 0.4;
  const bool pool_nopark = fdg::knob("FDG_POOL_NOPARK") != nullptr;
  std::vector<MOp> out;
  OptProgram &prog;

  uint32_t leaf_lo = 0, leaf_n = 0;          // values leaf_lo .. leaf_lo+leaf_n-1 are input columns (re-loadable)
  // cooperative programs: while a value sits in a shared slot (from its M_SEND to the end of the publication interval)
  // that slot is its home for the owner too -- no private register, AGPR or LDS slot is tied up by it
  std::vector<std::vector<uint32_t>> expire;  // [barrier number] values whose shared slot is released after that barrier
  uint32_t n_barrier_seen = 0;

  Alloc(const Lowered &p_, const OptParams &prm_, const std::vector<UOp> &u_, uint32_t nv_, OptProgram &pr)
      : p(p_), prm(prm_), u(u_), nv(nv_), prog(pr) {
    leaf_n = p.L;
    n_land = prm.n_land < prm.n_acc ? prm.n_land : 0;
    n_acc_spill = prm.n_acc - n_land;
    for (uint32_t i = n_land; i-- > 0;) free_land.push_back(n_acc_spill + i);
  }

  uint32_t next_use(uint32_t v) const {
    return up[v] < uses[v].size() ? uses[v][up[v]] : std::numeric_limits<uint32_t>::max();
  }
  bool get_lds(uint32_t &s) {
    if (!free_lds.empty()) { s = free_lds.back(); free_lds.pop_back(); return true; }
    if (lds_next < prm.n_lds) { s = lds_next++; return true; }
    return false;
  }
  bool get_acc(uint32_t &s) {
    if (!free_acc.empty()) { s = free_acc.back(); free_acc.pop_back(); return true; }
    if (acc_next < n_acc_spill) { s = acc_next++; return true; }
    return false;
  }
  uint32_t get_mem() {
    if (!free_mem.empty()) { uint32_t s = free_mem.back(); free_mem.pop_back(); return s; }
    return mem_next++;
  }
  void release_home(uint32_t v) {
    if (home_kind[v] == 6) { free_land.push_back(home_slot[v]); home_kind[v] = 3; return; }
    if (home_kind[v] == 1) free_lds.push_back(home_slot[v]);
    else if (home_kind[v] == 2) free_mem.push_back(home_slot[v]);
    else if (home_kind[v] == 4) free_acc.push_back(home_slot[v]);
    if (home_kind[v] != 3) home_kind[v] = 0;
  }
  void kill(uint32_t v) {   // no further use
    if (reg_of[v] != NONE) { owner[reg_of[v]] = NONE; free_regs.push_back(reg_of[v]); reg_of[v] = NONE; }
    release_home(v);
  }
  // take a register for use at op position `pos`.  Results of VALU ops take the
  // most recently freed register (usually an operand that just died), loads the
  // one that has been idle longest: that keeps a pool of long-idle registers, so
  // hoist_loads() can issue loads hundreds of ops ahead of their consumer.
  uint32_t take_reg(uint32_t pos, bool for_load = false, uint32_t not_before = NONE) {
    if (!free_regs.empty()) {
      uint32_t r;
      if (for_load) { r = free_regs.front(); free_regs.pop_front(); }
      else { r = free_regs.back(); free_regs.pop_back(); }
      reg_hw = std::max(reg_hw, r + 1);
      return r;
    }
    // Belady: evict the resident value whose next use is farthest away
    uint32_t best = NONE, best_nu = 0;
    // When the on-chip overflow levels are full the candidates do not cost the same: a value without a home takes a panel store
    // and a panel load (on the graphs with a thousand leaves both miss L2), a leaf or a value that already has a panel home one
    // load, a value parked in LDS / AGPRs next to nothing -- so the distances are compared per access (round 3: panel + leaf
    // accesses 3.33 -> 3.05 x L on the GV 4-loop vertex function, +4-6 % there, +7 % on the synthetic stand-in, neutral on the
    // graphs that do not spill; profiles/r03_log_evict_cost.txt).  FDG_EVICT_COST=0: the plain farthest-next-use rule.
    if (evict_cost && free_lds.empty() && lds_next >= prm.n_lds && free_acc.empty() && acc_next >= n_acc_spill) {
      double best_score = -1.0;
      for (uint32_t r = 0; r < prm.n_reg; ++r) {
        if (owner[r] == NONE || lock[r] == pos) continue;
        const uint32_t v = owner[r];
        const uint32_t nu = next_use(v);
        double score;
        if (nu == std::numeric_limits<uint32_t>::max()) score = 1e30;
        else {
          const double c = home_kind[v] == 0 ? (double)evict_cost : (home_kind[v] == 1 || home_kind[v] == 4 ? 0.5 : (home_kind[v] == 3 && prm.pool_leaves ? pool_leaf_cost : 1.0));
          score = (double)(nu - pos) / c;
        }
        if (score > best_score) { best_score = score; best = r; best_nu = nu; }
      }
    } else
    for (uint32_t r = 0; r < prm.n_reg; ++r) {
      if (owner[r] == NONE || lock[r] == pos) continue;
      const uint32_t nu = next_use(owner[r]);
      if (best == NONE || nu > best_nu) { best = r; best_nu = nu; }
    }
    if (not_before != NONE && best_nu <= not_before) return NONE;   // prefetch would displace something needed sooner
    const uint32_t v = owner[best];
    if (home_kind[v] == 0) {            // only copy is in the register: spill it
      uint32_t s;
      if (prm.pool_leaves && free_lds.empty() && lds_next >= prm.n_lds && free_acc.empty() && acc_next >= n_acc_spill) {
        // pooled programs: a leaf parked in an LDS slot or an AGPR pair gives its place to a value that would otherwise go to the HBM panel
        // (the leaf comes back from the shared pool); the parked leaf that is needed again last goes
        uint32_t drop = NONE, drop_nu = 0;
        for (uint32_t x = leaf_lo; x < leaf_lo + leaf_n; ++x) {
          if ((home_kind[x] != 1 && home_kind[x] != 4) || reg_of[x] != NONE) continue;
          const uint32_t nu = next_use(x);
          if (drop == NONE || nu > drop_nu) { drop = x; drop_nu = nu; }
        }
        if (drop != NONE) { (home_kind[drop] == 1 ? free_lds : free_acc).push_back(home_slot[drop]); home_kind[drop] = 3; }
      }
      if (get_lds(s)) { out.push_back(MOp{M_ST_LDS, 0, 0, s, best, 0, 0.0}); home_kind[v] = 1; home_slot[v] = s; prog.n_st_lds++; }
      else if (get_acc(s)) { out.push_back(MOp{M_ST_ACC, 0, 0, s, best, 0, 0.0}); home_kind[v] = 4; home_slot[v] = s; prog.n_st_acc++; }
      else { s = get_mem(); out.push_back(MOp{M_ST_MEM, 0, 0, s, best, 0, 0.0}); home_kind[v] = 2; home_slot[v] = s; prog.n_st_mem++; }
    } else if (home_kind[v] == 3) {
      // a leaf: re-loadable from its source; park it in LDS when there is room so
      // the next use does not go back to HBM
      uint32_t s;
      if (next_use(v) != std::numeric_limits<uint32_t>::max() && !(prm.pool_leaves && pool_nopark)) {
        if (get_lds(s)) { out.push_back(MOp{M_ST_LDS, 0, 0, s, best, 0, 0.0}); home_kind[v] = 1; home_slot[v] = s; prog.n_st_lds++; }
        else if (get_acc(s)) { out.push_back(MOp{M_ST_ACC, 0, 0, s, best, 0, 0.0}); home_kind[v] = 4; home_slot[v] = s; prog.n_st_acc++; }
      }
    }
    reg_of[v] = NONE;
    owner[best] = NONE;
    return best;
  }
  uint32_t ensure_in_reg(uint32_t v, uint32_t pos) {
    if (reg_of[v] != NONE) { lock[reg_of[v]] = pos; return reg_of[v]; }
    const uint32_t r = take_reg(pos, true);
    switch (home_kind[v]) {
      case 6:      // landed: out of its slot (which is free again from here on), the input matrix stays its home
        out.push_back(MOp{M_LD_ACC, 0, 0, r, home_slot[v], 0, 0.0}); prog.n_ld_acc++;
        free_land.push_back(home_slot[v]); home_kind[v] = 3;
        break;
      case 1: out.push_back(MOp{M_LD_LDS, 0, 0, r, home_slot[v], 0, 0.0}); prog.n_ld_lds++; break;
      case 2: out.push_back(MOp{M_LD_MEM, 0, 0, r, home_slot[v], 0, 0.0}); prog.n_ld_mem++; break;
      case 4: out.push_back(MOp{M_LD_ACC, 0, 0, r, home_slot[v], 0, 0.0}); prog.n_ld_acc++; break;
      case 5: out.push_back(MOp{M_RECV, 0, 0, r, home_slot[v], 0, 0.0}); prog.n_recv++; break;          // still in its shared slot
      default: out.push_back(MOp{M_LD_LEAF, 0, 0, r, v - leaf_lo, 0, 0.0}); prog.n_ld_leaf++; if (prm.leaves_once) home_kind[v] = 0; break;   // leaves only
    }
    reg_of[v] = r; owner[r] = v; lock[r] = pos;
    return r;
  }
  // Issue the load of value v now (op position j) for its use at op q > j.
  void prefetch(uint32_t v, uint32_t j, uint32_t q) {
    if (reg_of[v] != NONE || home_kind[v] == 0 || home_kind[v] == 4 || home_kind[v] == 6) return;
    const uint32_t r = take_reg(j, true, q);
    if (r == NONE) return;
    switch (home_kind[v]) {
      case 1: out.push_back(MOp{M_LD_LDS, 0, 0, r, home_slot[v], 0, 0.0}); prog.n_ld_lds++; break;
      case 2: out.push_back(MOp{M_LD_MEM, 0, 0, r, home_slot[v], 0, 0.0}); prog.n_ld_mem++; break;
      case 5: out.push_back(MOp{M_RECV, 0, 0, r, home_slot[v], 0, 0.0}); prog.n_recv++; break;
      default: out.push_back(MOp{M_LD_LEAF, 0, 0, r, v - leaf_lo, 0, 0.0}); prog.n_ld_leaf++; if (prm.leaves_once) home_kind[v] = 0; break;
    }
    reg_of[v] = r; owner[r] = v;
  }
  // scan pointer `pf` up to j + dist and prefetch operands whose home is of `kind`
  void prefetch_window(uint32_t &pf, uint32_t j, uint32_t dist, uint8_t kind) {
    const uint64_t lim = std::min<uint64_t>((uint64_t)u.size(), (uint64_t)j + dist + 1);
    if (pf <= j) pf = j + 1;
    for (; pf < lim; ++pf) {
      const UOp &o = u[pf];
      if (!mop_has_a(o.kind)) continue;
      const uint32_t va = o.a >> 1;
      if (home_kind[va] == kind) prefetch(va, j, pf);
      if (mop_has_b(o.kind)) {
        const uint32_t vb = o.b >> 1;
        if (home_kind[vb] == kind) prefetch(vb, j, pf);
      }
      if (mop_has_c(o.kind)) {
        const uint32_t vc = o.c >> 1;
        if (home_kind[vc] == kind) prefetch(vc, j, pf);
      }
    }
  }
  // scan pointer `pf` up to j + lookahead_land: a leaf that is neither in a register nor landed gets a landing slot while there is one
  void landing_window(uint32_t &pf, uint32_t j) {
    if (!n_land) return;
    const uint64_t lim = std::min<uint64_t>((uint64_t)u.size(), (uint64_t)j + prm.lookahead_land + 1);
    if (pf <= j) pf = j + 1;
    auto land = [&](uint32_t v) {
      if (free_land.empty() || home_kind[v] != 3 || reg_of[v] != NONE) return;
      const uint32_t s = free_land.back(); free_land.pop_back();
      out.push_back(MOp{M_LD_LEAF_ACC, 0, 0, s, v - leaf_lo, 0, 0.0});
      prog.n_ld_leaf++; prog.n_ld_land++;
      home_kind[v] = 6; home_slot[v] = s;
    };
    for (; pf < lim; ++pf) {
      if (free_land.empty()) return;          // (the pointer stays: what it has not seen is looked at again when a slot is free)
      const UOp &o = u[pf];
      if (!mop_has_a(o.kind)) continue;
      land(o.a >> 1);
      if (mop_has_b(o.kind)) land(o.b >> 1);
      if (mop_has_c(o.kind)) land(o.c >> 1);
    }
  }
  void run() {
    uses.assign(nv, {});
    for (uint32_t j = 0; j < u.size(); ++j) {
      const UOp &o = u[j];
      if (!mop_has_a(o.kind)) continue;
      uses[o.a >> 1].push_back(j);
      if (mop_has_b(o.kind)) uses[o.b >> 1].push_back(j);
      if (mop_has_c(o.kind)) uses[o.c >> 1].push_back(j);
    }
    up.assign(nv, 0);
    reg_of.assign(nv, NONE);
    home_kind.assign(nv, 0);
    home_slot.assign(nv, 0);
    for (uint32_t i = 0; i < leaf_n; ++i) home_kind[leaf_lo + i] = 3;
    owner.assign(prm.n_reg, NONE);
    lock.assign(prm.n_reg, NONE);
    for (uint32_t r = 0; r < prm.n_reg; ++r) free_regs.push_back(r);
    uint32_t live = 0;
    uint32_t pf_leaf = 0, pf_lds = 0, pf_mem = 0, pf_land = 0;
    for (uint32_t j = 0; j < u.size(); ++j) {
      const UOp &o = u[j];
      if (prm.lookahead_leaf) prefetch_window(pf_leaf, j, prm.lookahead_leaf, 3);
      landing_window(pf_land, j);
      if (prm.lookahead_mem) prefetch_window(pf_mem, j, prm.lookahead_mem, 2);
      if (prm.lookahead_lds) prefetch_window(pf_lds, j, prm.lookahead_lds, 1);
      if (o.kind == M_BARRIER) {
        if (n_barrier_seen < expire.size())
          for (uint32_t v : expire[n_barrier_seen]) {
            if (home_kind[v] != 5) continue;
            if (next_use(v) != std::numeric_limits<uint32_t>::max() && reg_of[v] == NONE) {      // still needed: back into a register
              const uint32_t r = take_reg(j, true);
              out.push_back(MOp{M_RECV, 0, 0, r, home_slot[v], 0, 0.0}); prog.n_recv++;
              reg_of[v] = r; owner[r] = v; lock[r] = j;
            }
            home_kind[v] = (v >= leaf_lo && v < leaf_lo + leaf_n) ? 3 : 0;
          }
        n_barrier_seen++;
        out.push_back(MOp{M_BARRIER, 0, 0, 0, 0, 0, 0.0}); prog.n_barrier++;
        continue;
      }
      if (o.kind == M_RECV) {             // a value another wave published: it stays re-loadable from its shared slot for its whole life
        const uint32_t rd = take_reg(j, true);
        reg_of[o.d] = rd; owner[rd] = o.d; lock[rd] = j;
        home_kind[o.d] = 5; home_slot[o.d] = o.a;
        out.push_back(MOp{M_RECV, 0, 0, rd, o.a, 0, 0.0});
        prog.n_recv++;
        if (uses[o.d].empty()) kill(o.d);
        continue;
      }
      const bool two = mop_has_b(o.kind);
      const bool three = mop_has_c(o.kind);
      const uint32_t va = o.a >> 1, vb = two ? (o.b >> 1) : NONE, vc = three ? (o.c >> 1) : NONE;
      const uint32_t ra = ensure_in_reg(va, j);
      const uint32_t rb = two ? ensure_in_reg(vb, j) : 0;
      const uint32_t rc = three ? ensure_in_reg(vc, j) : 0;
      // advance use cursors, free dead operands (so the destination may reuse a register)
      up[va]++;
      if (two) up[vb]++;     // (a == b: two entries at position j)
      if (three) up[vc]++;
      if (next_use(va) == std::numeric_limits<uint32_t>::max()) kill(va);
      if (two && vb != va && next_use(vb) == std::numeric_limits<uint32_t>::max()) kill(vb);
      if (three && vc != va && vc != vb && next_use(vc) == std::numeric_limits<uint32_t>::max()) kill(vc);
      if (o.kind == M_ROOT) {
        out.push_back(MOp{M_ROOT, (uint8_t)(o.a & 1), 0, o.d, ra, 0, 0.0});
        continue;
      }
      if (o.kind == M_SEND) {
        out.push_back(MOp{M_SEND, 0, 0, o.d, ra, 0, 0.0});
        prog.n_send++;
        if (reg_of[va] != NONE || next_use(va) != std::numeric_limits<uint32_t>::max()) {
          if (home_kind[va] != 3) release_home(va);
          if (next_use(va) != std::numeric_limits<uint32_t>::max()) {
            home_kind[va] = 5; home_slot[va] = o.d;
            const uint32_t e_end = (uint32_t)o.imm;
            if (expire.size() <= e_end) expire.resize(e_end + 1);
            expire[e_end].push_back(va);
          }
        }
        continue;
      }
      const uint32_t rd = take_reg(j);
      reg_of[o.d] = rd; owner[rd] = o.d; lock[rd] = j;
      if (o.kind == M_MULC) out.push_back(MOp{M_MULC, (uint8_t)(o.a & 1), 0, rd, ra, 0, o.imm});
      else if (o.kind == M_SELC) out.push_back(MOp{M_SELC, (uint8_t)(o.a & 1), (uint8_t)(o.b ? 1 : 0), rd, ra, 0, o.imm});
      else if (o.kind == M_ADDC || o.kind == M_EXP || o.kind == M_RCP || o.kind == M_FIXZ || o.kind == M_DIV1 || o.kind == M_CONST) out.push_back(MOp{o.kind, (uint8_t)(o.a & 1), 0, rd, ra, 0, o.imm});
      else if (o.kind == M_FMAK) out.push_back(MOp{M_FMAK, (uint8_t)(o.a & 1), (uint8_t)(o.b & 1), rd, ra, rb, o.imm});
      else if (three) out.push_back(MOp{o.kind, (uint8_t)(o.a & 1), (uint8_t)(two ? (o.b & 1) : 0), rd, ra, rb, o.imm, (uint8_t)(o.c & 1), rc});
      else out.push_back(MOp{o.kind, (uint8_t)(o.a & 1), (uint8_t)(o.b & 1), rd, ra, rb, 0.0});
      out.back().param = o.param;
      prog.n_valu++;
      if (uses[o.d].empty()) kill(o.d);   // cannot happen for reachable values; keeps the state sane
      (void)live;
    }
    prog.n_reg_used = reg_hw;
    prog.n_lds_used = lds_next;
    prog.n_mem_used = mem_next;
    prog.n_acc_used = n_land ? prm.n_acc : acc_next;     // (landing slots are the top of the range)
  }
};

// Move every load as far up as its destination register and its source slot
// allow (bounded by the prefetch distances): the allocator places loads right
// before their consumer, which would expose the full LDS / HBM latency with one
// wave per SIMD.
void hoist_loads(std::vector<MOp> &ops, const OptParams &prm) {
  const size_t n = ops.size();
  std::vector<int64_t> last_reg(prm.n_reg, -1), last_st_lds, last_st_mem;
  int64_t last_barrier = -1;
  std::vector<std::pair<double, uint32_t>> key(n);
  for (size_t q = 0; q < n; ++q) {
    MOp &o = ops[q];
    double k = (double)q;
    auto touch = [&](uint32_t r) { last_reg[r] = (int64_t)q; };
    switch (o.kind) {
      case M_LD_LEAF: case M_LD_LDS: case M_LD_MEM: {
        int64_t lo = last_reg[o.d] + 1;
        const uint32_t dist = (o.kind == M_LD_LDS) ? prm.lookahead_lds : (o.kind == M_LD_MEM ? prm.lookahead_mem : prm.lookahead_leaf);
        if (o.kind == M_LD_LDS && o.a < last_st_lds.size()) lo = std::max(lo, last_st_lds[o.a] + 1);
        if (o.kind == M_LD_MEM && o.a < last_st_mem.size()) lo = std::max(lo, last_st_mem[o.a] + 1);
        lo = std::max<int64_t>(lo, (int64_t)q - (int64_t)dist);
        if (lo < (int64_t)q) k = (double)lo - 0.5;
        touch(o.d);
        break;
      }
      case M_RECV: {       // never above the barrier that makes the slot's content visible
        int64_t lo = std::max(last_reg[o.d] + 1, last_barrier + 1);
        lo = std::max<int64_t>(lo, (int64_t)q - (int64_t)prm.lookahead_lds);
        if (lo < (int64_t)q) k = (double)lo - 0.5;
        touch(o.d);
        break;
      }
      case M_SEND: touch(o.a); break;
      case M_BARRIER: last_barrier = (int64_t)q; break;
      case M_ST_LDS:
        if (last_st_lds.size() <= o.d) last_st_lds.resize(o.d + 1, -1);
        last_st_lds[o.d] = (int64_t)q; touch(o.a); break;
      case M_ST_MEM:
        if (last_st_mem.size() <= o.d) last_st_mem.resize(o.d + 1, -1);
        last_st_mem[o.d] = (int64_t)q; touch(o.a); break;
      case M_MUL: case M_ADD: touch(o.a); touch(o.b); touch(o.d); break;
      case M_FMA: touch(o.a); touch(o.b); touch(o.c); touch(o.d); break;
      case M_FMAC: touch(o.a); touch(o.c); touch(o.d); break;
      case M_MULC: case M_MOV: case M_ADDC: case M_EXP: case M_RCP: case M_FIXZ: case M_SELC: case M_DIV1: case M_CONST: touch(o.a); touch(o.d); break;
      case M_FMAK: touch(o.a); touch(o.b); touch(o.d); break;
      case M_SEL: touch(o.a); touch(o.b); touch(o.c); touch(o.d); break;
      case M_ROOT: touch(o.a); break;
      case M_LD_ACC: touch(o.d); break;
      case M_ST_ACC: touch(o.a); break;
    }
    key[q] = {k, (uint32_t)q};
  }
  std::stable_sort(key.begin(), key.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
  std::vector<MOp> r;
  r.reserve(n);
  for (auto &kq : key) r.push_back(ops[kq.second]);
  ops.swap(r);
}

// Runs of back-to-back leaf loads (the burst at the start of a tile, groups hoisted to the same point) are
// independent of each other: issue them in ascending leaf order, so that the kernel's column pointer mostly
// advances by one stride (two scalar ops) instead of being rebuilt (six).
void sort_load_runs(std::vector<MOp> &ops) {
  size_t i = 0;
  while (i < ops.size()) {
    if (ops[i].kind != M_LD_LEAF) { ++i; continue; }
    size_t j = i;
    while (j < ops.size() && ops[j].kind == M_LD_LEAF) ++j;
    if (j - i >= 2) {
      bool distinct = true;
      for (size_t a = i; a < j && distinct; ++a)
        for (size_t b = a + 1; b < j; ++b) if (ops[a].d == ops[b].d) { distinct = false; break; }
      // (experiment FDG_LOAD_RUN_CHUNK=c: only within chunks of c loads, so that what the first fold steps need is issued first;
      // loads return in order, and behind a sorted burst of 70 the first fold step waits for whichever of them comes last)
      const size_t chunk = fdg::knob("FDG_LOAD_RUN_CHUNK") ? (size_t)std::max(0, std::atoi(fdg::knob("FDG_LOAD_RUN_CHUNK"))) : 0;
      if (distinct)
        for (size_t s = i; s < j; s += (chunk ? chunk : j - i))
          std::stable_sort(ops.begin() + s, ops.begin() + std::min(j, s + (chunk ? chunk : j - i)), [](const MOp &x, const MOp &y) { return x.a < y.a; });
    }
    i = j;
  }
}

}  // namespace

// architectural VGPRs end at v255: 6 fixed, the values, what the kernel variant reserves, the macro ops' temporaries
static bool fit_registers(const std::vector<UOp> &u, const OptParams &prm, OptProgram &out) {
  uint32_t tmp_pairs = 0;
  for (const UOp &o : u) tmp_pairs = std::max(tmp_pairs, mop_tmp_pairs(o.kind));
  if (tmp_pairs + prm.reserve_pairs + 8 > 125) { out.supported = false; out.why = "too few registers"; return false; }
  out.params = prm;
  out.params.n_reg = std::min<uint32_t>(prm.n_reg, 125 - tmp_pairs - prm.reserve_pairs);
  return true;
}

void build_opt_program(const Lowered &p, const OptParams &prm, OptProgram &out) {
  out = OptProgram();
  out.params = prm;
  Builder B0(p);
  B0.value_numbering = prm.vn_window != 1;     // 1 = off, 0 = unlimited, else window in ops
  B0.vn_window = prm.vn_window > 1 ? prm.vn_window : 0;
  B0.vn_touch = fdg::knob("FDG_VN_BIRTH_WINDOW") == nullptr;   // default: the window counts from the last read
  B0.remat_window = prm.remat_window;
  B0.remat_cost = prm.remat_cost;
  B0.keep_root_order = prm.keep_root_order || fdg::knob("FDG_KEEP_ROOT_ORDER") != nullptr;     // (the environment switch is for experiments)
  if (const char *rw = fdg::knob("FDG_REMAT_WINDOW")) B0.remat_window = (uint64_t)std::atoll(rw);     // experiments
  if (const char *tw = fdg::knob("FDG_TERM_WINDOW")) B0.term_window = (uint32_t)std::max(1, std::atoi(tw));
  if (const char *tr = fdg::knob("FDG_TERM_RECENT")) B0.term_recent = (uint32_t)std::max(1, std::atoi(tr));
  build_uops(B0);
  Lowered plain;
  const bool retry = !B0.ok && B0.why == "inconsistent schedule groups";
  if (retry) { plain = p; plain.sched_group.clear(); }
  Builder B1(retry ? plain : p);
  B1.value_numbering = B0.value_numbering; B1.vn_window = B0.vn_window; B1.vn_touch = B0.vn_touch;
  B1.remat_window = B0.remat_window; B1.remat_cost = B0.remat_cost; B1.keep_root_order = B0.keep_root_order;
  if (retry) build_uops(B1);
  Builder &B = retry ? B1 : B0;
  out.supported = B.ok;
  out.why = B.why;
  if (!B.ok) return;
  if (prm.n_reg < 4) { out.supported = false; out.why = "too few registers"; return; }
  if (prm.fma) fuse_fma(B.u, B.next_vid);
  if (prm.roots_last) {             // ... in root order, so that neighbours in memory are neighbours in the program
    auto tail = std::stable_partition(B.u.begin(), B.u.end(), [](const UOp &o) { return o.kind != M_ROOT; });
    std::stable_sort(tail, B.u.end(), [](const UOp &x, const UOp &y) { return x.d < y.d; });
  }
  if (!fit_registers(B.u, prm, out)) return;
  if (fdg::knob("FDG_LEAVES_ONCE")) out.params.leaves_once = fdg::knob("FDG_LEAVES_ONCE")[0] == '1';     // (experiments through fdg_graph_opt_program)
  Alloc A(p, out.params, B.u, B.next_vid, out);
  A.run();
  out.ops.swap(A.out);
  hoist_loads(out.ops, out.params);
  sort_load_runs(out.ops);
}

void build_mc_program(const Lowered &p, const LeafSpec &ls, const OptParams &prm, OptProgram &out) {
  out = OptProgram();
  out.params = prm;
  Lowered plain;
  bool retry = false;
  for (int pass = 0; pass < 2; ++pass) {
    Builder B(retry ? plain : p, &ls);
    B.value_numbering = true;
    B.vn_window = prm.vn_window > 1 ? prm.vn_window : 0;
    build_uops(B);
    if (!B.ok && !retry && B.why == "inconsistent schedule groups") { plain = p; plain.sched_group.clear(); retry = true; continue; }
    out.supported = B.ok;
    out.why = B.why;
    if (!B.ok) return;
    if (prm.n_reg < 4) { out.supported = false; out.why = "too few registers"; return; }
    if (prm.fma) fuse_fma(B.u, B.next_vid);
    if (!fit_registers(B.u, prm, out)) return;
    Alloc A(p, out.params, B.u, B.next_vid, out);
    A.leaf_lo = B.in_base; A.leaf_n = B.n_in;
    out.mc_n_k = B.n_k; out.mc_n_t = B.n_in - B.n_k;
    A.run();
    out.ops.swap(A.out);
    hoist_loads(out.ops, out.params);
    sort_load_runs(out.ops);
    return;
  }
}

bool build_schedule(const Lowered &p, const OptParams &prm, std::vector<SchedOp> &ops, uint32_t &n_value, std::string &why) {
  Builder B0(p);
  B0.value_numbering = prm.vn_window != 1;
  B0.vn_window = prm.vn_window > 1 ? prm.vn_window : 0;
  B0.vn_touch = fdg::knob("FDG_VN_BIRTH_WINDOW") == nullptr;
  B0.keep_minus_one = prm.keep_minus_one;
  B0.mul_keeps_signs = prm.mul_keeps_signs;
  build_uops(B0);
  Lowered plain;
  const bool retry = !B0.ok && B0.why == "inconsistent schedule groups";
  if (retry) { plain = p; plain.sched_group.clear(); }
  Builder B1(retry ? plain : p);
  B1.value_numbering = B0.value_numbering; B1.vn_window = B0.vn_window; B1.vn_touch = B0.vn_touch; B1.keep_minus_one = B0.keep_minus_one; B1.mul_keeps_signs = B0.mul_keeps_signs;
  if (retry) build_uops(B1);
  Builder &B = retry ? B1 : B0;
  why = B.why;
  if (!B.ok) return false;
  ops.clear();
  ops.reserve(B.u.size());
  for (const UOp &o : B.u) ops.push_back(SchedOp{o.kind, o.d, o.a, o.b, o.imm});
  n_value = B.next_vid;
  return true;
}

}  // namespace fdg

// =====================================================================================================================
// Cooperative variant (fdg_opt.h: CoopProgram).  Host-side simulation of four waves working through the graph in epochs:
// the terms of the wide root sums are dealt to whichever wave is free, a wave that needs a shared sub-expression another
// wave computed in an EARLIER epoch receives it through a shared LDS slot, the root's own left fold runs on its home wave as
// the terms arrive (the association is untouched: only who computes a term changes).  Afterwards the hand-overs are grouped
// into publication intervals, slots are assigned, the owners' programs get their M_SEND ops, and each wave's list goes
// through the ordinary allocator.
// =====================================================================================================================
namespace fdg {
namespace {

struct CoopBuild {
  const Lowered &p;
  const uint32_t L;
  std::vector<std::unique_ptr<Builder>> B;
  // global knowledge about nodes
  std::vector<int8_t> owner;          // [N] wave whose value is the published one, -1 = not computed yet
  std::vector<uint32_t> prod_epoch;   // [N]
  std::vector<uint32_t> pub_ref;      // [N] the owner's ref of the node's value (vid << 1 | neg)
  std::vector<int8_t> in_progress;    // [N] wave that has the node on its stack, -1 = none
  std::vector<uint8_t> cheap;         // [N] may be computed again by a wave that cannot wait for it
  std::vector<uint8_t> is_wide_root;  // [N]
  // copies: a value received by a wave (one M_RECV op each)
  struct Copy { uint32_t node; uint8_t wave; uint32_t e_first, e_last; uint32_t uop; uint32_t interval = NONE; };
  std::vector<Copy> copies;
  std::vector<std::vector<uint32_t>> copy_of;   // [wave][N] index of the live copy, NONE
  struct Wide { uint32_t node, home, next, acc; bool done; };
  std::vector<Wide> wides;
  std::vector<uint32_t> tasks;        // value indices (>= L) to compute, in order
  size_t next_task = 0;
  std::vector<std::vector<Frame>> st;
  std::vector<std::pair<uint32_t, uint32_t>> rootlist;
  uint32_t cur_epoch = 0;
  uint64_t n_dup = 0;
  uint32_t copy_window = 600;         // a received value not read for this many ops of its wave is received again when needed
  bool ok = true;
  std::string why;

  explicit CoopBuild(const Lowered &p_) : p(p_), L(p_.L) {}

  uint32_t fold_cost(uint32_t n) const {
    uint32_t c = p.off[n + 1] - p.off[n] - 1;
    for (uint32_t e = p.off[n]; e < p.off[n + 1]; ++e) if (p.fac[e] != 1.0 && p.fac[e] != -1.0) c++;
    if (p.op[n] == FDG_OP_POWER) c += 4;
    return c;
  }
  void emit_roots(uint32_t w, uint32_t v) {
    auto it = std::lower_bound(rootlist.begin(), rootlist.end(), std::make_pair(v, 0u));
    for (; it != rootlist.end() && it->first == v; ++it) B[w]->u.push_back(UOp{M_ROOT, it->second, B[w]->ref_of[v], 0, 0.0});
  }
  // one fold step of node n with operand ref cr (the reference's left fold, static.jl:13-46)
  uint32_t fold(Builder &Bw, uint32_t n, uint32_t i, uint32_t acc, uint32_t cr) {
    const uint32_t a = p.off[n];
    const double fc = p.fac[a + i];
    Bw.touch(cr);
    if (p.op[n] == FDG_OP_SUM) {
      const uint32_t t = Bw.mulc(cr, fc);
      return i == 0 ? t : Bw.op2(M_ADD, acc, t);
    }
    if (p.op[n] == FDG_OP_PROD && p.assoc_interp) {
      const uint32_t t = Bw.mulc(cr, fc);
      return i == 0 ? t : Bw.op2(M_MUL, acc, t);
    }
    if (p.op[n] == FDG_OP_PROD) {
      acc = i == 0 ? cr : Bw.op2(M_MUL, acc, cr);
      return Bw.mulc(acc, fc);
    }
    return Bw.mulc(Bw.powi(cr, p.power[n]), fc);
  }
  enum Avail { HAVE, COMPUTE, BLOCKED };
  // operand c (value index) as seen by wave w; on HAVE `ref` is set (a receive op may have been emitted)
  Avail operand(uint32_t w, uint32_t c, uint32_t &ref) {
    Builder &Bw = *B[w];
    if (c < L) { ref = Bw.ref_of[c]; return HAVE; }
    const uint32_t n = c - L;
    if (Bw.ref_of[c] != NONE) {
      const uint32_t ci = copy_of[w][n];
      if (ci == NONE) { ref = Bw.ref_of[c]; return HAVE; }                      // computed here
      const uint32_t v = Bw.ref_of[c] >> 1;
      if ((uint64_t)Bw.u.size() - Bw.born[v] <= copy_window) {                   // a copy still around
        copies[ci].e_last = cur_epoch;
        ref = Bw.ref_of[c];
        return HAVE;
      }
      Bw.ref_of[c] = NONE; copy_of[w][n] = NONE;                                 // forgotten: received again below
    }
    if (owner[n] >= 0 && (uint32_t)owner[n] != w) {
      if (prod_epoch[n] < cur_epoch) {
        const uint32_t d = Bw.fresh();
        Bw.u.push_back(UOp{M_RECV, d, 0, 0, 0.0});
        copies.push_back(Copy{n, (uint8_t)w, cur_epoch, cur_epoch, (uint32_t)Bw.u.size() - 1});
        copy_of[w][n] = (uint32_t)copies.size() - 1;
        Bw.ref_of[c] = (d << 1) | (pub_ref[n] & 1u);
        ref = Bw.ref_of[c];
        return HAVE;
      }
      return (cheap[n] && !is_wide_root[n]) ? COMPUTE : BLOCKED;                 // computed by another wave in this very epoch
    }
    if (is_wide_root[n]) return BLOCKED;                                         // folded by its home wave as the terms arrive
    if (in_progress[n] >= 0 && (uint32_t)in_progress[n] != w) return cheap[n] ? COMPUTE : BLOCKED;
    return COMPUTE;
  }
  void finished(uint32_t w, uint32_t n, uint32_t acc) {
    Builder &Bw = *B[w];
    Bw.ref_of[L + n] = acc;
    copy_of[w][n] = NONE;
    if (in_progress[n] == (int8_t)w) in_progress[n] = -1;
    if (owner[n] < 0) { owner[n] = (int8_t)w; prod_epoch[n] = cur_epoch; pub_ref[n] = acc; emit_roots(w, L + n); }
    else n_dup += fold_cost(n);
  }
  // lets wave w do one unit of work; returns the number of ops it added, 0 = idle or blocked (blocked: `blk` set)
  size_t advance(uint32_t w, bool &blk) {
    Builder &Bw = *B[w];
    const size_t before = Bw.u.size();
    blk = false;
    // 1. the fold of a wide root this wave is home of
    for (Wide &W : wides) {
      if (W.done || W.home != w) continue;
      const uint32_t n = W.node, k = p.off[n + 1] - p.off[n];
      const uint32_t c = p.idx[p.off[n] + W.next];
      const bool ready = c < L || Bw.ref_of[c] != NONE || (owner[c - L] >= 0 && ((uint32_t)owner[c - L] == w || prod_epoch[c - L] < cur_epoch));
      if (!ready) continue;
      uint32_t cr;
      const Avail a = operand(w, c, cr);
      if (a != HAVE) continue;
      W.acc = fold(Bw, n, W.next, W.acc, cr);
      if (++W.next == k) { W.done = true; finished(w, n, W.acc); }
      return std::max<size_t>(Bw.u.size() - before, 1);
    }
    // 2. a task
    if (st[w].empty()) {
      while (next_task < tasks.size() && (owner[tasks[next_task] - L] >= 0 || in_progress[tasks[next_task] - L] >= 0)) next_task++;
      if (next_task >= tasks.size()) return 0;
      const uint32_t n = tasks[next_task++] - L;
      in_progress[n] = (int8_t)w;
      st[w].push_back(Frame{n, 0, NONE});
    }
    Frame &f = st[w].back();
    const uint32_t n = f.n, k = p.off[n + 1] - p.off[n];
    if (owner[n] >= 0 && (uint32_t)owner[n] != w && prod_epoch[n] < cur_epoch && f.i == 0) {   // someone else finished it meanwhile
      if (in_progress[n] == (int8_t)w) in_progress[n] = -1;
      st[w].pop_back();
      return 1;
    }
    const uint32_t c = p.idx[p.off[n] + f.i];
    uint32_t cr;
    const Avail a = operand(w, c, cr);
    if (a == BLOCKED) { blk = true; return 0; }
    if (a == COMPUTE) {
      const uint32_t cn = c - L;
      if (in_progress[cn] < 0) in_progress[cn] = (int8_t)w;
      st[w].push_back(Frame{cn, 0, NONE});
      return 1;
    }
    f.acc = fold(Bw, n, f.i, f.acc, cr);
    if (++f.i == k) { const uint32_t acc = f.acc; st[w].pop_back(); finished(w, n, acc); }
    return std::max<size_t>(Bw.u.size() - before, 1);
  }
};

}  // namespace

static void build_coop_once(const Lowered &p, const OptParams &prm, size_t budget, uint32_t gap, CoopProgram &out, uint32_t NW);

// Shorter epochs keep fewer hand-overs in flight: when the shared slots run out the schedule is rebuilt with a smaller budget.
void build_coop_program(const Lowered &p, const OptParams &prm, CoopProgram &out, uint32_t n_wave) {
  const char *te = fdg::knob("FDG_COOP_EPOCH_OPS");
  if (te) { const char *ge = fdg::knob("FDG_COOP_GAP"); build_coop_once(p, prm, (size_t)std::max(8, std::atoi(te)), ge ? (uint32_t)std::atoi(ge) : 2u, out, n_wave); return; }
  for (size_t budget : {192, 128, 96, 64, 48}) {
    build_coop_once(p, prm, budget, budget > 64 ? 2u : 1u, out, n_wave);
    if (out.supported || out.why != "shared LDS slots exhausted") return;
  }
}

static void build_coop_once(const Lowered &p, const OptParams &prm, size_t budget, uint32_t gap, CoopProgram &out, uint32_t NW) {
  out = CoopProgram();
  out.n_wave = NW;
  const uint32_t L = p.L;
  CoopBuild C(p);
  if (const char *e = fdg::knob("FDG_COOP_COPY_WINDOW")) C.copy_window = (uint32_t)std::max(1, std::atoi(e));
  for (uint32_t k = 0; k < p.R; ++k) if (p.root_slot[k] != FDG_NO_ROOT) C.rootlist.push_back({p.root_slot[k], k});
  std::sort(C.rootlist.begin(), C.rootlist.end());
  for (uint32_t w = 0; w < NW; ++w) {
    C.B.emplace_back(new Builder(p));
    C.B[w]->value_numbering = prm.vn_window != 1;
    C.B[w]->vn_window = prm.vn_window > 1 ? prm.vn_window : 0;
  }
  C.owner.assign(p.N, -1); C.prod_epoch.assign(p.N, 0); C.pub_ref.assign(p.N, NONE); C.in_progress.assign(p.N, -1);
  C.cheap.assign(p.N, 0); C.is_wide_root.assign(p.N, 0);
  for (uint32_t n = 0; n < p.N; ++n) C.cheap[n] = C.fold_cost(n) <= 12;
  C.copy_of.assign(NW, std::vector<uint32_t>(p.N, NONE));
  C.st.assign(NW, {});
  // roots: leaves are written by wave 0; a wide Sum / Prod root is folded by a home wave while every wave computes terms
  std::vector<uint32_t> root_nodes;
  for (auto &rk : C.rootlist) {
    if (rk.first < L) { C.B[0]->u.push_back(UOp{M_ROOT, rk.second, C.B[0]->ref_of[rk.first], 0, 0.0}); continue; }
    if (root_nodes.empty() || root_nodes.back() != rk.first - L) root_nodes.push_back(rk.first - L);
  }
  // Many roots (the rows of a vertex function): whole roots are dealt to the waves, in the order that keeps what they share
  // on chip (order_roots); with a few roots the terms of each are dealt, a home wave folding them as they arrive.
  size_t n_wide = 0;
  for (uint32_t rn : root_nodes) n_wide += (p.off[rn + 1] - p.off[rn] >= 8 && p.op[rn] != FDG_OP_POWER);
  const char *rte = fdg::knob("FDG_COOP_ROOT_TASKS");
  const bool root_tasks = rte ? rte[0] == '1' : root_nodes.size() > 4 * (size_t)NW;
  if (root_tasks) order_roots(p, root_nodes);
  std::vector<std::vector<uint32_t>> term_lists;
  for (uint32_t rn : root_nodes) {
    const uint32_t k = p.off[rn + 1] - p.off[rn];
    if (!root_tasks && k >= 8 && p.op[rn] != FDG_OP_POWER) {
      C.is_wide_root[rn] = 1;
      C.wides.push_back(CoopBuild::Wide{rn, (uint32_t)(C.wides.size() % NW), 0, NONE, false});
      std::vector<uint32_t> t;
      for (uint32_t e = p.off[rn]; e < p.off[rn + 1]; ++e) if (p.idx[e] >= L) t.push_back(p.idx[e]);
      term_lists.push_back(t);
    } else {
      C.tasks.push_back(L + rn);
    }
  }
  if (C.wides.empty() && !root_tasks) { out.why = "no wide root to distribute"; return; }
  (void)n_wide;
  for (size_t i = 0;; ++i) {                           // the wide roots' terms, interleaved
    bool any = false;
    for (auto &t : term_lists) if (i < t.size()) { C.tasks.push_back(t[i]); any = true; }
    if (!any) break;
  }
  // ---- epochs ---------------------------------------------------------------------------------------------------------
  std::vector<std::vector<uint32_t>> bar_pos(NW);     // [wave][epoch] index of the barrier op ending that epoch
  uint32_t stalled = 0;
  for (;;) {
    std::vector<size_t> used(NW, 0);
    std::vector<uint8_t> stop(NW, 0);
    size_t progress = 0;
    for (;;) {
      int w = -1;
      for (uint32_t k = 0; k < NW; ++k) if (!stop[k] && used[k] < budget && (w < 0 || used[k] < used[(size_t)w])) w = (int)k;
      if (w < 0) break;
      bool blk;
      const size_t did = C.advance((uint32_t)w, blk);
      if (did == 0) stop[(size_t)w] = 1;
      used[(size_t)w] += did;
      progress += did;
    }
    for (uint32_t w = 0; w < NW; ++w) { bar_pos[w].push_back((uint32_t)C.B[w]->u.size()); C.B[w]->u.push_back(UOp{M_BARRIER, 0, 0, 0, 0.0}); }
    C.cur_epoch++;
    bool all_done = C.next_task >= C.tasks.size();
    for (uint32_t w = 0; w < NW; ++w) all_done = all_done && C.st[w].empty();
    for (auto &W : C.wides) all_done = all_done && W.done;
    if (all_done) break;
    stalled = progress ? 0 : stalled + 1;
    if (stalled > 2 || C.cur_epoch > 200000) { out.why = "cooperative schedule does not make progress"; return; }
    for (uint32_t w = 0; w < NW; ++w) if (!C.B[w]->ok) { out.why = C.B[w]->why; return; }
  }
  for (uint32_t w = 0; w < NW; ++w) if (!C.B[w]->ok) { out.why = C.B[w]->why; return; }
  out.n_epoch = C.cur_epoch;
  out.n_duplicate = C.n_dup;
  // a copy is readable from its shared slot until the epoch of its LAST use in its wave's list (a product whose first
  // operand is a received value keeps it as its accumulator across however many epochs its other operands take)
  for (uint32_t w = 0; w < NW; ++w) {
    std::unordered_map<uint32_t, uint32_t> copy_vid;
    for (uint32_t i = 0; i < C.copies.size(); ++i) if (C.copies[i].wave == w) copy_vid[C.B[w]->u[C.copies[i].uop].d] = i;
    uint32_t e = 0;
    auto seen = [&](uint32_t ref) { auto it = copy_vid.find(ref >> 1); if (it != copy_vid.end()) C.copies[it->second].e_last = std::max(C.copies[it->second].e_last, e); };
    for (const UOp &o : C.B[w]->u) {
      if (o.kind == M_BARRIER) { e++; continue; }
      if (!mop_has_a(o.kind)) continue;
      seen(o.a);
      if (mop_has_b(o.kind)) seen(o.b);
      if (mop_has_c(o.kind)) seen(o.c);
    }
  }
  // ---- publication intervals and shared slots ---------------------------------------------------------------------------
  out.n_priv_lds = NW == 4 ? 16 : (NW == 8 ? 8 : 4);
  if (const char *e = fdg::knob("FDG_COOP_PRIV_LDS")) out.n_priv_lds = (uint32_t)std::max(1, std::min(70, std::atoi(e)));
  out.n_shared = std::min<uint32_t>(256, 312 - NW * out.n_priv_lds);   // (the emitter addresses shared slots as two banks of 128)
  struct Interval { uint32_t node, start, end, slot; };
  std::vector<Interval> ivs;
  {
    std::vector<std::vector<uint32_t>> by_node(p.N);
    for (uint32_t i = 0; i < C.copies.size(); ++i) by_node[C.copies[i].node].push_back(i);
    for (uint32_t n = 0; n < p.N; ++n)
      std::sort(by_node[n].begin(), by_node[n].end(), [&](uint32_t a, uint32_t b) { return C.copies[a].e_first < C.copies[b].e_first; });
    // the value a node publishes may itself be a received copy (a one-child Sum of a remote node): its M_SEND is a use of that
    // copy, at the end of the publishing interval's first epoch -- the copy's own interval must reach that far, which may in turn
    // move that interval's end; repeated until nothing moves
    std::vector<std::unordered_map<uint32_t, uint32_t>> copy_vid(NW);
    for (uint32_t i = 0; i < C.copies.size(); ++i) copy_vid[C.copies[i].wave][C.B[C.copies[i].wave]->u[C.copies[i].uop].d] = i;
    for (int pass = 0; pass < 64; ++pass) {
      ivs.clear();
      for (uint32_t n = 0; n < p.N; ++n)
        for (uint32_t ci : by_node[n]) {
          CoopBuild::Copy &c = C.copies[ci];
          if (!ivs.empty() && ivs.back().node == n && c.e_first - 1 <= ivs.back().end + gap) ivs.back().end = std::max(ivs.back().end, c.e_last);
          else ivs.push_back(Interval{n, c.e_first - 1, c.e_last, NONE});
          c.interval = (uint32_t)ivs.size() - 1;
        }
      bool moved = false;
      for (const Interval &iv : ivs) {
        const uint32_t w = (uint32_t)C.owner[iv.node];
        auto it = copy_vid[w].find(C.pub_ref[iv.node] >> 1);
        if (it != copy_vid[w].end() && C.copies[it->second].e_last < iv.start) { C.copies[it->second].e_last = iv.start; moved = true; }
      }
      if (!moved) break;
    }
    std::vector<uint32_t> order(ivs.size());
    for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return ivs[a].start < ivs[b].start || (ivs[a].start == ivs[b].start && a < b); });
    std::vector<uint32_t> free_at(out.n_shared, 0);   // first epoch in which the slot may be written again
    for (uint32_t i : order) {
      uint32_t best = NONE;
      for (uint32_t s = 0; s < out.n_shared; ++s) if (free_at[s] <= ivs[i].start && (best == NONE || free_at[s] > free_at[best])) best = s;
      if (best == NONE) { out.why = "shared LDS slots exhausted"; return; }
      ivs[i].slot = best;
      free_at[best] = ivs[i].end + 1;
    }
  }
  if (fdg::knob("FDG_COOP_DEBUG")) {
    std::vector<uint32_t> occ(C.cur_epoch + 2, 0);
    for (const Interval &iv : ivs) for (uint32_t e = iv.start; e <= iv.end && e < occ.size(); ++e) occ[e]++;
    uint64_t sum = 0; uint32_t mx = 0;
    for (uint32_t e = 0; e < C.cur_epoch; ++e) { sum += occ[e]; mx = std::max(mx, occ[e]); }
    std::fprintf(stderr, "[coop] epochs %u intervals %zu copies %zu slot occupancy avg %.1f max %u of %u\n", C.cur_epoch, ivs.size(), C.copies.size(),
                 (double)sum / std::max(1u, C.cur_epoch), mx, out.n_shared);
    std::vector<uint32_t> len_hist(8, 0);
    for (const Interval &iv : ivs) len_hist[std::min<uint32_t>(7, (iv.end - iv.start) / 4)]++;
    for (uint32_t k = 0; k < 8; ++k) std::fprintf(stderr, "  interval length %u..%u epochs: %u\n", 4 * k, 4 * k + 3, len_hist[k]);
  }
  out.n_transfer = C.copies.size();
  // ---- the waves' lists with their M_SEND ops, then the ordinary allocator ----------------------------------------------
  std::vector<std::vector<std::vector<UOp>>> sends(NW, std::vector<std::vector<UOp>>(C.cur_epoch));
  for (const Interval &iv : ivs) {
    const uint32_t w = (uint32_t)C.owner[iv.node];
    sends[w][iv.start].push_back(UOp{M_SEND, iv.slot, C.pub_ref[iv.node] & ~1u, 0, (double)iv.end});   // imm: last epoch the slot holds it
  }
  for (const CoopBuild::Copy &c : C.copies) C.B[c.wave]->u[c.uop].a = ivs[c.interval].slot;
  for (uint32_t w = 0; w < NW; ++w) {
    std::vector<UOp> u;
    u.reserve(C.B[w]->u.size() + 64);
    uint32_t e = 0;
    for (uint32_t j = 0; j < C.B[w]->u.size(); ++j) {
      if (e < C.cur_epoch && j == bar_pos[w][e]) { for (const UOp &s : sends[w][e]) u.push_back(s); e++; }
      u.push_back(C.B[w]->u[j]);
    }
    OptParams q = prm;
    q.n_lds = out.n_priv_lds;
    q.reserve_pairs = std::max<uint32_t>(q.reserve_pairs, 2);      // the cooperative section's own registers above the temporaries (v[rm0 .. rm0 + 3], fdg_isa.cpp): with
                                                                   // pow_body's four temporary pairs they ran past v255 (round 5 fuzz with Power{N}, |N| >= 4)
    if (!fit_registers(u, q, out.wave[w])) { out.why = out.wave[w].why; return; }
    Alloc A(p, out.wave[w].params, u, C.B[w]->next_vid, out.wave[w]);
    A.run();
    out.wave[w].ops.swap(A.out);
    hoist_loads(out.wave[w].ops, out.wave[w].params);
    sort_load_runs(out.wave[w].ops);
  }
  out.supported = true;
}


// =====================================================================================================================
// Pooled cooperative variant (fdg_opt.h: build_pool_program).
// =====================================================================================================================
void build_pool_program(const Lowered &p, const OptParams &prm, CoopProgram &out, uint32_t NW, uint32_t epoch_ops, uint32_t ahead) {
  const uint32_t unit_in = out.pool_unit;
  out = CoopProgram();
  out.n_wave = NW;
  out.pooled = true;
  out.pool_unit = unit_in == 2 ? 2 : 1;
  if (const char *e = fdg::knob("FDG_POOL_UNIT")) out.pool_unit = std::atoi(e) == 2 ? 2 : 1;
  const uint32_t L = p.L;
  if (NW < 2 || NW > CoopProgram::MAXW) { out.why = "bad wave count"; return; }
  if (p.sched_group.size() == p.N && p.N) { out.why = "schedule groups are not part of the pooled variant"; return; }
  if (const char *e = fdg::knob("FDG_POOL_EPOCH_OPS")) epoch_ops = (uint32_t)std::max(16, std::atoi(e));
  if (const char *e = fdg::knob("FDG_POOL_AHEAD")) ahead = (uint32_t)std::max(1, std::atoi(e));
  if (!epoch_ops) epoch_ops = 128;
  if (!ahead) ahead = 8;
  // synchronisation between the epochs of a tile: s_barrier (slack 0), or progress words with `slack` epochs of freedom (fdg_opt.h: CoopProgram::slack)
  uint32_t S = 0;
  if (const char *e = fdg::knob("FDG_POOL_SYNC")) if (std::string(e) == "flags") S = 2;
  if (const char *e = fdg::knob("FDG_POOL_SLACK")) if (S) S = (uint32_t)std::max(1, std::min(8, std::atoi(e)));
  out.slack = S;
  // ---- roots to waves: largest cone first, to the wave where it adds the least to the heaviest load ----------------------------
  std::vector<std::pair<uint32_t, uint32_t>> rootlist;
  for (uint32_t k = 0; k < p.R; ++k) if (p.root_slot[k] != FDG_NO_ROOT) rootlist.push_back({p.root_slot[k], k});
  std::vector<uint32_t> root_nodes;             // distinct internal nodes that are roots
  for (auto &rk : rootlist) if (rk.first >= L) root_nodes.push_back(rk.first - L);
  std::sort(root_nodes.begin(), root_nodes.end());
  root_nodes.erase(std::unique(root_nodes.begin(), root_nodes.end()), root_nodes.end());
  if (root_nodes.size() < 2 * (size_t)NW) { out.why = "fewer than two roots per wave: nothing to deal"; return; }
  const size_t W64 = ((size_t)p.N + 63) / 64;
  auto fold_cost = [&](uint32_t n) {
    uint32_t c = p.off[n + 1] - p.off[n] - 1;
    for (uint32_t e = p.off[n]; e < p.off[n + 1]; ++e) if (p.fac[e] != 1.0 && p.fac[e] != -1.0) c++;
    if (p.op[n] == FDG_OP_POWER) c += 2;
    return c;
  };
  std::vector<std::vector<uint64_t>> cone(root_nodes.size(), std::vector<uint64_t>(W64, 0));
  std::vector<uint64_t> csize(root_nodes.size(), 0);
  {
    std::vector<uint32_t> stk;
    for (size_t r = 0; r < root_nodes.size(); ++r) {
      stk.assign(1, root_nodes[r]);
      cone[r][root_nodes[r] >> 6] |= 1ull << (root_nodes[r] & 63);
      while (!stk.empty()) {
        const uint32_t n = stk.back(); stk.pop_back();
        csize[r] += fold_cost(n);
        for (uint32_t e = p.off[n]; e < p.off[n + 1]; ++e) {
          const uint32_t c = p.idx[e];
          if (c >= L && !((cone[r][(c - L) >> 6] >> ((c - L) & 63)) & 1)) { cone[r][(c - L) >> 6] |= 1ull << ((c - L) & 63); stk.push_back(c - L); }
        }
      }
    }
  }
  // Largest cone first, each to the wave where load + affinity * (fold steps it adds there) is smallest: roots that share sub-expressions
  // gather on one wave (what two waves both need is computed twice), and the waves finish together.  (FDG_POOL_DEAL=overlap deals in
  // order_roots' order instead -- neighbours of that order on different waves at the same time: measured on the 4-loop GV vertex function
  // that doubles the duplicated fold steps, 16 617 -> 67 062, and the pool traffic.)
  std::vector<size_t> deal(root_nodes.size());
  for (size_t i = 0; i < deal.size(); ++i) deal[i] = i;
  if (!(fdg::knob("FDG_POOL_DEAL") && std::string(fdg::knob("FDG_POOL_DEAL")) == "overlap")) {
    std::sort(deal.begin(), deal.end(), [&](size_t a, size_t b) { return csize[a] > csize[b] || (csize[a] == csize[b] && a < b); });
  } else {
    std::vector<uint32_t> ordered = root_nodes;
    order_roots(p, ordered);
    std::vector<size_t> pos(p.N, 0);
    for (size_t i = 0; i < root_nodes.size(); ++i) pos[root_nodes[i]] = i;
    for (size_t i = 0; i < ordered.size(); ++i) deal[i] = pos[ordered[i]];
  }
  // (experiment FDG_POOL_SEED=n > 0: the dealing order and every wave's root order shuffled -- how much do the pool's fetches depend on which roots
  //  run next to which?)
  uint64_t pool_seed = fdg::knob("FDG_POOL_SEED") ? (uint64_t)std::atoll(fdg::knob("FDG_POOL_SEED")) : 0;
  auto rnd = [&]() { pool_seed ^= pool_seed << 13; pool_seed ^= pool_seed >> 7; pool_seed ^= pool_seed << 17; return pool_seed; };
  std::vector<uint32_t> root_rank;
  if (pool_seed) {
    pool_seed = pool_seed * 0x9E3779B97F4A7C15ull + 1;
    for (size_t i = deal.size(); i > 1; --i) std::swap(deal[i - 1], deal[rnd() % i]);
    root_rank.assign(p.N, 0);
    for (size_t i = 0; i < root_nodes.size(); ++i) root_rank[root_nodes[i]] = (uint32_t)(rnd() & 0xffffff);
  }
  std::vector<std::vector<uint64_t>> have(NW, std::vector<uint64_t>(W64, 0));
  std::vector<uint64_t> load(NW, 0);
  std::vector<uint32_t> wave_of_node(p.N, NONE);
  const double affinity = fdg::knob("FDG_POOL_AFFINITY") ? std::atof(fdg::knob("FDG_POOL_AFFINITY")) : 2.0;
  for (size_t r : deal) {
    uint32_t best = 0; uint64_t best_load = ~0ull;
    std::vector<uint64_t> add(NW, 0);
    for (uint32_t w = 0; w < NW; ++w) {
      uint64_t a = 0;
      for (size_t i = 0; i < W64; ++i) {
        uint64_t m = cone[r][i] & ~have[w][i];
        while (m) { const uint32_t n = (uint32_t)(i * 64 + (size_t)__builtin_ctzll(m)); a += fold_cost(n); m &= m - 1; }
      }
      add[w] = a;
      const uint64_t score = load[w] + (uint64_t)(affinity * (double)a);
      if (score < best_load) { best_load = score; best = w; }
    }
    load[best] += add[best];
    for (size_t i = 0; i < W64; ++i) have[best][i] |= cone[r][i];
    wave_of_node[root_nodes[r]] = best;
  }
  uint64_t total_cost = 0, single_cost = 0;
  for (uint32_t w = 0; w < NW; ++w) total_cost += load[w];
  for (uint32_t n = 0; n < p.N; ++n) if (p.live[L + n]) single_cost += fold_cost(n);
  out.n_duplicate = total_cost > single_cost ? total_cost - single_cost : 0;
  // ---- every wave's schedule: the ordinary depth-first fold order over its own roots (own value numbering) ----------------------
  std::vector<std::vector<uint8_t>> mask(NW, std::vector<uint8_t>(p.R, 0));
  for (auto &rk : rootlist) mask[rk.first < L ? 0 : wave_of_node[rk.first - L]][rk.second] = 1;      // (roots that are leaves: wave 0 stores them)
  std::vector<std::unique_ptr<Builder>> B;
  for (uint32_t w = 0; w < NW; ++w) {
    B.emplace_back(new Builder(p));
    B[w]->value_numbering = prm.vn_window != 1;
    B[w]->vn_window = prm.vn_window > 1 ? prm.vn_window : 0;
    B[w]->root_mask = &mask[w];
    if (!root_rank.empty()) B[w]->root_rank = &root_rank;
    build_uops(*B[w]);                 // (order_roots inside orders the wave's own roots by what they share)
    if (!B[w]->ok) { out.why = B[w]->why; return; }
  }
  // ---- registers per wave: leaves are re-loadable (from the pool: a short LDS read) ---------------------------------------------------------
  out.n_priv_lds = NW <= 4 ? 16 : (NW == 8 ? 8 : 4);
  if (const char *e = fdg::knob("FDG_COOP_PRIV_LDS")) out.n_priv_lds = (uint32_t)std::max(1, std::min(70, std::atoi(e)));
  out.n_shared = std::min<uint32_t>(256, 312 - NW * out.n_priv_lds);
  if (const char *e = fdg::knob("FDG_POOL_SLOTS")) out.n_shared = (uint32_t)std::max(8, std::min<int>((int)out.n_shared, std::atoi(e)));
  for (uint32_t w = 0; w < NW; ++w) {
    OptParams q = prm;
    q.n_lds = out.n_priv_lds;
    q.n_land = 0;
    q.pool_leaves = true;
    // a pool read is an LDS read -- of a pool all four waves and the fetches hammer: issued 96 fold steps ahead of its use (32, the
    // distance of a wave's private LDS slots: -10 %; 200: the landing registers are missed elsewhere, -4 %; profiles/r04_log_la_lds.txt)
    q.lookahead_leaf = 96;
    if (const char *e = fdg::knob("FDG_POOL_READ_AHEAD")) q.lookahead_leaf = (uint32_t)std::max(1, std::atoi(e));
    q.reserve_pairs = std::max<uint32_t>(q.reserve_pairs, S ? 4u : 2u);      // (as in build_coop_program; three more registers for the progress words)
    if (!fit_registers(B[w]->u, q, out.wave[w])) { out.why = out.wave[w].why; return; }
    Alloc A(p, out.wave[w].params, B[w]->u, B[w]->next_vid, out.wave[w]);
    A.run();
    out.wave[w].ops.swap(A.out);
    hoist_loads(out.wave[w].ops, out.wave[w].params);
  }
  // ---- epochs.  The waves meet at a barrier every `epoch_ops` fold steps' worth of ESTIMATED TIME (in units of one fp64 op = four cycles):
  //      the moves to and from AGPRs, the LDS accesses and the share of the pool fetches a wave issues are counted too, so that the waves
  //      arrive together (by op count alone they arrive up to a quarter of an epoch apart, and every barrier then costs that).  Epoch 0, in
  //      front of the opening barrier, only fetches.
  auto op_time = [](const MOp &o) -> uint32_t {
    switch (o.kind) {
      case M_LD_ACC: case M_ST_ACC: return 2;
      case M_LD_LEAF: case M_LD_LDS: case M_ST_LDS: case M_LD_MEM: case M_ST_MEM: case M_ROOT: return 1;
      case M_BARRIER: return 0;
      default: return 1 + mop_tmp_pairs(o.kind) * 4;
    }
  };
  uint32_t n_epoch = 1;
  {
    uint64_t longest = 0;
    for (uint32_t w = 0; w < NW; ++w) { uint64_t t = 0; for (const MOp &o : out.wave[w].ops) t += op_time(o); longest = std::max(longest, t); }
    n_epoch = 1 + (uint32_t)((longest + epoch_ops - 1) / epoch_ops);
  }
  for (uint32_t w = 0; w < NW; ++w) {
    std::vector<MOp> r;
    r.reserve(out.wave[w].ops.size() + n_epoch);
    r.push_back(MOp{M_BARRIER, 0, 0, 0, 0, 0, 0.0});
    uint64_t t = 0; uint32_t nb = 1;
    for (const MOp &o : out.wave[w].ops) {
      r.push_back(o);
      t += op_time(o);
      while (nb < n_epoch && t >= (uint64_t)nb * epoch_ops) { r.push_back(MOp{M_BARRIER, 0, 0, 0, 0, 0, 0.0}); nb++; }
    }
    for (; nb < n_epoch; ++nb) r.push_back(MOp{M_BARRIER, 0, 0, 0, 0, 0, 0.0});
    if (r.back().kind != M_BARRIER) { out.why = "internal: a pooled program does not end at a barrier"; return; }
    out.wave[w].ops.swap(r);
    out.wave[w].n_barrier = n_epoch;
  }
  out.n_epoch = n_epoch;
  // ---- the pool: which leaf sits in which slot during which epochs (offline Belady at epoch granularity over all waves' reads) -----------
  // Unit of residency: one leaf (512 bytes of the tile), or -- `pool_unit` 2, tile-major batches only -- a PAIR of leaves 2u, 2u + 1, adjacent
  // in the tile, brought by ONE 64-lane LDS-direct load into two adjacent slots (a fetch instruction costs the wave that issues it about a
  // hundred cycles whatever it brings: tools/ubench/vmem_issue.hip; leaf numbers follow the first visit, so a leaf's neighbour is used near it).
  const uint32_t U = out.pool_unit == 2 ? 2 : 1;
  const uint32_t P = out.n_shared / U, LU = (L + U - 1) / U;
  std::vector<std::vector<uint32_t>> read_ep(LU);           // [unit] epochs in which some wave reads it (ascending, unique)
  for (uint32_t w = 0; w < NW; ++w) {
    uint32_t e = 0;
    for (const MOp &o : out.wave[w].ops) {
      if (o.kind == M_BARRIER) { e++; continue; }
      if (o.kind == M_LD_LEAF) read_ep[o.a / U].push_back(e);
      if (o.kind == M_LD_LEAF_ACC) { out.why = "landing slots are not part of the pooled variant"; return; }
    }
  }
  std::vector<std::vector<uint32_t>> need(n_epoch + 1);     // [epoch] units read in it
  for (uint32_t l = 0; l < LU; ++l) {
    auto &v = read_ep[l];
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
    for (uint32_t e : v) { if (e == 0 || e > n_epoch) { out.why = "internal: a leaf is read outside the compute epochs"; return; } need[e].push_back(l); }
  }
  struct Fetch { uint32_t unit, slot, issue, ready, unit2; };       // unit2 != NONE: a second, arbitrary leaf by the same instruction into slot + 1
  std::vector<Fetch> fetches;
  std::vector<uint32_t> slot_of(LU, NONE);                  // resident unit -> unit slot
  std::vector<uint32_t> in_slot(P, NONE), last_read(P, 0);  // slot -> unit; last epoch in which the slot's content is read (as planned so far)
  std::vector<std::vector<std::pair<uint32_t, uint32_t>>> resid(LU);  // [unit] (first epoch readable, slot) per residency, ascending
  auto next_read = [&](uint32_t l, uint32_t from) -> uint32_t {     // first read epoch >= from
    auto &v = read_ep[l];
    auto it = std::lower_bound(v.begin(), v.end(), from);
    return it == v.end() ? std::numeric_limits<uint32_t>::max() : *it;
  };
  // One LDS-direct instruction has 64 lanes of 16 bytes: with single leaves half of them idle.  PAIRED fetches (round 4, second half): the
  // upper 32 lanes bring ANOTHER leaf -- any leaf: their addresses are their own -- into the slot after the first one's, so two leaves that
  // miss in the same epoch share one instruction when an aligned pair of slots (2k, 2k + 1) can be given to them at the same issue epoch
  // (an instruction costs the issuing wave about a hundred cycles whatever it brings).  MEASURED SLOWER, so off unless FDG_POOL_PAIR=1: on the
  // 4-loop GV vertex function 1 245 instead of 2 195 fetch instructions (+9 % leaves) run 4.16 ms against 3.70; with pairs only into slots whose
  // content is dead or 64 epochs from its next read 1 694 instructions, no extra leaf, 3.96 ms (profiles/r04_log_pool_anypair.txt).
  const bool pairing = U == 1 && fdg::knob("FDG_POOL_PAIR") && fdg::knob("FDG_POOL_PAIR")[0] == '1';
  const uint32_t far = fdg::knob("FDG_POOL_FAR") ? (uint32_t)std::atoi(fdg::knob("FDG_POOL_FAR")) : 64;
  const uint32_t INF = std::numeric_limits<uint32_t>::max();
  const uint32_t pair_far = fdg::knob("FDG_POOL_PAIR_FAR") ? (uint32_t)std::atoi(fdg::knob("FDG_POOL_PAIR_FAR")) : 32;
  // may slot s2 be given away at `issue` to content first read in epoch e?  key: how late its present content is needed again (0: no)
  auto victim_key = [&](uint32_t s2, uint32_t issue, uint32_t e) -> uint64_t {
    if (in_slot[s2] == NONE) return (uint64_t)INF + 2;
    if (last_read[s2] + S >= issue) return 0;                          // (still being read when the fetch would be issued -- by a wave up to S sync points behind)
    const uint32_t nr = next_read(in_slot[s2], issue > S ? issue - S : 0);
    if (nr <= e) return 0;                                             // needed again before (or when) the new content is: keep it
    // an EARLY fetch only takes a slot whose content is dead or far from its next read: being early must not cost a re-fetch
    if (issue + 2 + S < e && nr != INF && nr <= e + far) return 0;      // ("early": more than two epochs before the leaf has to have landed, which is S epochs before e)
    return (uint64_t)nr + 1;
  };
  auto install = [&](uint32_t l, uint32_t slot, uint32_t e) {
    if (in_slot[slot] != NONE) slot_of[in_slot[slot]] = NONE;
    in_slot[slot] = l; slot_of[l] = slot; last_read[slot] = e;
    resid[l].push_back({e, slot});
  };
  uint64_t n_paired = 0;
  for (uint32_t e = 1; e <= n_epoch; ++e) {                 // epoch whose reads must be resident
    // (flag synchronisation: a reader in epoch e only knows that the others have reached sync point max(1, e - S), so the fetch must have been
    //  confirmed by then: `rdy` takes the place of e as the epoch by which the leaf has landed)
    //  -- except in the first S + 1 epochs of a tile, whose sync points are strict: the pool starts empty, and everything the first epochs read
    //  cannot be fetched in front of the opening barrier)
    const uint32_t rdy = (S && e > S + 1) ? e - S : e;
    const uint32_t first_issue = rdy > ahead ? rdy - ahead : 0;
    std::vector<uint32_t> missing;
    for (uint32_t l : need[e]) {
      if (slot_of[l] != NONE) last_read[slot_of[l]] = std::max(last_read[slot_of[l]], e);
      else missing.push_back(l);
    }
    for (size_t i = 0; i < missing.size();) {
      // The fetch is issued `ahead` epochs before the read when a slot is free by then, else as early after that as one becomes free
      // (memory latency is a few epochs; the pool is small: the prefetch distance adapts to how much of it the epochs around need).
      // Victim at a given issue epoch: an empty slot, else the slot nobody reads from that epoch on whose content is needed again farthest
      // in the future (and not before the new content is).
      if (pairing && i + 1 < missing.size()) {
        uint32_t best = NONE, issue = first_issue;
        for (; issue < rdy && best == NONE; ++issue) {
          uint64_t best_key = 0;
          for (uint32_t k = 0; k + 1 < P; k += 2) {
            const uint64_t k0 = victim_key(k, issue, e), k1 = k0 ? victim_key(k + 1, issue, e) : 0;
            uint64_t key = std::min(k0, k1);
            // the pair of slots is a worse victim than the two best single slots would be: only contents that are dead, or not read again
            // for `pair_far` epochs, make way for a paired fetch (else the instructions saved come back as re-fetches)
            if (key && key <= (uint64_t)INF && key <= (uint64_t)e + pair_far) key = 0;
            if (key > best_key) { best_key = key; best = k; }
          }
          if (best != NONE) break;
        }
        if (best != NONE) {
          // (a pair is not taken when a single slot would have been free earlier by more than an epoch: latency first)
          // (the smaller leaf index first: the upper lanes' offsets from the first leaf's address are unsigned)
          const uint32_t la = std::min(missing[i], missing[i + 1]), lb = std::max(missing[i], missing[i + 1]);
          install(la, best, e);
          install(lb, best + 1, e);
          fetches.push_back(Fetch{la, best, issue, rdy, lb});
          n_paired++;
          i += 2;
          continue;
        }
      }
      const uint32_t l = missing[i];
      uint32_t best = NONE, issue = first_issue;
      for (; issue < rdy && best == NONE; ++issue) {
        uint64_t best_key = 0;
        for (uint32_t s2 = 0; s2 < P; ++s2) {
          const uint64_t key = victim_key(s2, issue, e);
          if (key > best_key) { best_key = key; best = s2; if (in_slot[s2] == NONE) break; }
        }
        if (best != NONE) break;
      }
      if (best == NONE) { out.why = "leaf pool exhausted"; return; }
      install(l, best, e);
      fetches.push_back(Fetch{l, best, issue, rdy, NONE});
      ++i;
    }
  }
  if (S)
    for (uint32_t w = 0; w < NW; ++w) {
      uint32_t b = 0;
      for (MOp &o : out.wave[w].ops) if (o.kind == M_BARRIER) { ++b; o.b = b; o.a = b == n_epoch ? 0u : (b <= S + 1 ? b : std::max<uint32_t>(S + 1, b - S)); }
    }
  out.n_fetch = fetches.size();
  out.n_transfer = 0;                                       // leaves brought from memory per tile
  for (const Fetch &f : fetches) out.n_transfer += f.unit2 != NONE ? 2 : std::min<uint32_t>(U, L - f.unit * U);
  if (fdg::knob("FDG_POOL_DEBUG")) {
    std::vector<uint32_t> hist(ahead + 2, 0);
    uint64_t sum = 0;
    for (const Fetch &f : fetches) { hist[std::min<uint32_t>(f.ready - f.issue, ahead + 1)]++; sum += f.ready - f.issue; }
    std::fprintf(stderr, "[pool] %llu of the fetches bring two leaves; ", (unsigned long long)n_paired);
    std::fprintf(stderr, "[pool] %zu fetches of %u leaves (%llu leaves in all), issued %.2f epochs of %u ops ahead on average;", fetches.size(), U, (unsigned long long)out.n_transfer,
                 fetches.empty() ? 0.0 : (double)sum / (double)fetches.size(), epoch_ops);
    for (uint32_t k = 1; k < hist.size(); ++k) std::fprintf(stderr, " %u:%u", k, hist[k]);
    std::fprintf(stderr, "\n");
  }
  // ---- fetch ops into the waves' lists (round robin, right after the barrier that opens the issue epoch); leaf reads become pool reads -----
  std::vector<std::vector<std::vector<Fetch>>> at(NW, std::vector<std::vector<Fetch>>(n_epoch + 1));
  {   // dealt epoch by epoch (an issue costs the wave about a hundred cycles: the waves of an epoch should issue equally many), the
      // surplus of an epoch going to the waves after those that got the last one
    std::vector<std::vector<size_t>> by_issue(n_epoch + 1);
    for (size_t i = 0; i < fetches.size(); ++i) by_issue[fetches[i].issue].push_back(i);
    uint32_t turn = 0;
    const bool global_rr = fdg::knob("FDG_POOL_DEAL_GLOBAL") != nullptr;      // (experiment: the round-3 dealing, by global index)
    // (experiment FDG_POOL_FETCH_WAVES=n: only the first n waves issue fetches -- is what a fetch costs its wave a property of the wave or of the CU?)
    const uint32_t NF = fdg::knob("FDG_POOL_FETCH_WAVES") ? (uint32_t)std::max(1, std::min<int>((int)NW, std::atoi(fdg::knob("FDG_POOL_FETCH_WAVES")))) : NW;
    for (uint32_t ep = 0; ep <= n_epoch; ++ep)
      for (size_t i : by_issue[ep]) { at[global_rr ? i % NF : turn % NF][ep].push_back(fetches[i]); turn++; }
  }
  for (uint32_t w = 0; w < NW; ++w) {
    std::vector<MOp> r;
    r.reserve(out.wave[w].ops.size() + fetches.size() / NW + 8);
    uint32_t e = 0;
    auto put_fetches = [&](uint32_t ep) {      // shared[d .. d + b - 1] = leaf[a .. a + b - 1]
      for (const Fetch &f : at[w][ep]) {
        MOp m{M_POOL_FETCH, 0, 0, f.slot * U, f.unit * U, std::min<uint32_t>(U, L - f.unit * U), (double)f.ready};
        if (f.unit2 != NONE) { m.b = 2; m.c = f.unit2; m.negc = 1; }         // negc: the second leaf is c (any leaf), not a + 1
        r.push_back(m);
      }
    };
    put_fetches(0);       // epoch 0 is what precedes the program's opening barrier
    for (const MOp &o : out.wave[w].ops) {
      if (o.kind == M_BARRIER) { r.push_back(o); e++; if (e <= n_epoch) put_fetches(e); continue; }
      if (o.kind == M_LD_LEAF) {
        uint32_t slot = NONE;
        for (const auto &pr : resid[o.a / U]) if (pr.first <= e) slot = pr.second;      // the residency that covers epoch e: the last one that started by then
        if (slot == NONE) { out.why = "internal: leaf read without a residency"; return; }
        MOp m{M_RECV, 0, 0, o.d, slot * U + o.a % U, 0, 0.0};
        r.push_back(m);
        out.wave[w].n_recv++;
        continue;
      }
      r.push_back(o);
    }
    out.wave[w].n_ld_leaf = 0;
    out.wave[w].ops.swap(r);
  }
  out.supported = true;
}

}  // namespace fdg
