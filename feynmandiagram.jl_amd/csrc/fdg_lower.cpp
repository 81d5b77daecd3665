// Host-side lowering of the node table: validation, dead-code elimination,
// liveness, and the slot-allocated micro-op stream the interpreter kernel walks.
//
// Order of evaluation = the table's order = the statement order of the
// reference's generated code (src/backend/static.jl:98-133); only statements
// that cannot influence a root are dropped.  Each node's arithmetic is untouched.
#include <algorithm>
#include <cstring>
#include <initializer_list>
#include <utility>
#include <limits>

#include "fdg_internal.h"
#include "fdg_powi.h"

namespace fdg {

static thread_local std::string g_err;
void set_error(const std::string &s) { g_err = s; }
const char *last_error_cstr() { return g_err.c_str(); }

double powi(double x, int32_t n) { return fdg_powi_impl(x, n); }

// The graph on Complex{Float64} values spelled out on their real and imaginary parts: a Float64 table whose leaves are
// re_0, im_0, re_1, im_1, ... (a row of a row-major ComplexF64 [B, L] matrix read as 2 L doubles) and whose roots are (re, im)
// of the original roots.  Every operation is the one Julia performs (base/complex.jl), in the association the evaluator's own folds
// keep: z * f = (re f, im f) as one-child nodes "(g * f)"; z + w componentwise -- a Sum becomes two Sums with the same children
// order and factors; z * w = (zr wr - zi wi, zr wi + zi wr) as four products, P1 + P2 * -1.0 (x * -1.0 is -x exactly) and
// P3 + P4, an n-ary Prod being the left fold of these with the factors applied where the text applies them; z^2 = z z,
// z^3 = (z z) z (Base.literal_pow).  The host mirror states the same construction in Python (nodetable.complex_to_real).
bool complex_to_real_table(const Lowered &p, RealTwinTable &o, std::string &why) {
  const uint32_t L = p.L;
  o = RealTwinTable();
  if (p.assoc_interp) { why = "the real-imaginary view spells out the generated code's association only"; return false; }
  o.n_leaf = 2 * L;
  o.off.assign(1, 0);
  std::vector<std::pair<uint32_t, uint32_t>> val((size_t)L + p.N);
  for (uint32_t l = 0; l < L; ++l) val[l] = {2 * l, 2 * l + 1};
  auto add = [&](uint8_t op, std::initializer_list<std::pair<uint32_t, double>> ch) {
    o.op.push_back(op); o.power.push_back(0);
    for (auto &c : ch) { o.idx.push_back(c.first); o.fac.push_back(c.second); }
    o.off.push_back((uint32_t)o.idx.size());
    return 2 * L + (uint32_t)o.op.size() - 1;
  };
  using Z = std::pair<uint32_t, uint32_t>;
  auto scale = [&](Z z, double f) -> Z { if (f == 1.0) return z; const uint32_t r = add(FDG_OP_SUM, {{z.first, f}}); const uint32_t i = add(FDG_OP_SUM, {{z.second, f}}); return {r, i}; };
  auto cmul = [&](Z z, Z w) -> Z {
    const uint32_t p1 = add(FDG_OP_PROD, {{z.first, 1.0}, {w.first, 1.0}}), p2 = add(FDG_OP_PROD, {{z.second, 1.0}, {w.second, 1.0}});
    const uint32_t re = add(FDG_OP_SUM, {{p1, 1.0}, {p2, -1.0}});
    const uint32_t p3 = add(FDG_OP_PROD, {{z.first, 1.0}, {w.second, 1.0}}), p4 = add(FDG_OP_PROD, {{z.second, 1.0}, {w.first, 1.0}});
    return {re, add(FDG_OP_SUM, {{p3, 1.0}, {p4, 1.0}})};
  };
  for (uint32_t n = 0; n < p.N; ++n) {
    const uint32_t a = p.off[n], b = p.off[n + 1];
    Z z;
    if (p.op[n] == FDG_OP_SUM) {
      for (int part = 0; part < 2; ++part) {
        o.op.push_back(FDG_OP_SUM); o.power.push_back(0);
        for (uint32_t e = a; e < b; ++e) { o.idx.push_back(part ? val[p.idx[e]].second : val[p.idx[e]].first); o.fac.push_back(p.fac[e]); }
        o.off.push_back((uint32_t)o.idx.size());
        (part ? z.second : z.first) = 2 * L + (uint32_t)o.op.size() - 1;
      }
    } else if (p.op[n] == FDG_OP_PROD) {
      z = scale(val[p.idx[a]], p.fac[a]);
      for (uint32_t e = a + 1; e < b; ++e) z = scale(cmul(z, val[p.idx[e]]), p.fac[e]);
    } else if (p.op[n] == FDG_OP_POWER && (p.power[n] == 2 || p.power[n] == 3)) {
      const Z x = val[p.idx[a]];
      z = cmul(x, x);
      if (p.power[n] == 3) z = cmul(z, x);
      z = scale(z, p.fac[a]);
    } else {
      why = "the spelled-out form of a complex graph covers Sum, Prod, Power{2}, Power{3}";
      return false;
    }
    val[L + n] = z;
  }
  for (uint32_t k = 0; k < p.R; ++k) {
    const uint32_t s = p.root_slot[k];
    if (s == FDG_NO_ROOT) { o.root_slot.push_back(FDG_NO_ROOT); o.root_slot.push_back(FDG_NO_ROOT); }
    else { o.root_slot.push_back(val[s].first); o.root_slot.push_back(val[s].second); }
  }
  return true;
}

int validate_desc(const fdg_graph_desc *d, std::string &err) {
  if (!d) { err = "null descriptor"; return FDG_E_INVALID; }
  const uint64_t L = d->n_leaf, N = d->n_node, E = d->n_edge, R = d->n_root;
  if (L + N >= LOC_IDX_MASK) { err = "graph too large (more than 2^29 values)"; return FDG_E_INVALID; }
  if (N && (!d->op || !d->power || !d->child_off || !d->child_idx || !d->child_fac)) {
    err = "null table array"; return FDG_E_INVALID;
  }
  if (R && !d->root_slot) { err = "null root_slot"; return FDG_E_INVALID; }
  if (N) {
    if (d->child_off[0] != 0) { err = "child_off[0] != 0"; return FDG_E_INVALID; }
    if (d->child_off[N] != E) { err = "child_off[n_node] != n_edge"; return FDG_E_INVALID; }
  } else if (E) { err = "edges without nodes"; return FDG_E_INVALID; }
  for (uint64_t n = 0; n < N; ++n) {
    uint32_t a = d->child_off[n], b = d->child_off[n + 1];
    if (b <= a) { err = "internal node " + std::to_string(n) + " has no children"; return FDG_E_INVALID; }
    if (d->op[n] > FDG_OP_POWER) {
      // static.jl:6-11
      err = "Static representation for computational graph nodes with operator code " +
            std::to_string((int)d->op[n]) + " not yet implemented!";
      return FDG_E_UNSUPPORTED;
    }
    if (d->op[n] == FDG_OP_POWER) {
      if (b - a != 1) { err = "Power node must have one and only one subgraph"; return FDG_E_INVALID; }
      if (d->power[n] == 0 || d->power[n] == 1) { err = "Power{0}/Power{1} makes no sense"; return FDG_E_INVALID; }
      // the interpreter stream stores the exponent biased by 2^27 in 28 bits
      if (d->power[n] >= (1 << 27) || d->power[n] <= -(1 << 27)) { err = "Power{N} with |N| >= 2^27 is not supported"; return FDG_E_UNSUPPORTED; }
    }
    for (uint32_t e = a; e < b; ++e) {
      if (d->child_idx[e] >= L + n) { err = "child index not smaller than its node: table is not topologically sorted"; return FDG_E_INVALID; }
      if (!std::isfinite(d->child_fac[e])) { err = "non-finite subgraph factor"; return FDG_E_INVALID; }
    }
  }
  if (R >= (1ull << 28)) { err = "2^28 roots or more are not supported"; return FDG_E_UNSUPPORTED; }   // 28-bit root index in the interpreter stream
  for (uint64_t k = 0; k < R; ++k)
    if (d->root_slot[k] != FDG_NO_ROOT && d->root_slot[k] >= L + N) { err = "root_slot out of range"; return FDG_E_INVALID; }
  return FDG_OK;
}

void analyse(Lowered &p) {
  const uint32_t L = p.L, N = p.N;
  p.live.assign((size_t)L + N, 0);
  for (uint32_t k = 0; k < p.R; ++k)
    if (p.root_slot[k] != FDG_NO_ROOT) p.live[p.root_slot[k]] = 1;
  for (int64_t n = (int64_t)N - 1; n >= 0; --n) {
    if (!p.live[L + n]) continue;
    for (uint32_t e = p.off[n]; e < p.off[n + 1]; ++e) p.live[p.idx[e]] = 1;
  }
  p.order.clear();
  p.flops_alg = 0;
  for (uint32_t n = 0; n < N; ++n) {
    if (!p.live[L + n]) continue;
    p.order.push_back(n);
    uint32_t k = p.off[n + 1] - p.off[n];
    uint32_t nonunit = 0;
    for (uint32_t e = p.off[n]; e < p.off[n + 1]; ++e) nonunit += (p.fac[e] != 1.0);
    if (p.op[n] == FDG_OP_POWER) {
      int64_t a = std::llabs((long long)p.power[n]);
      uint64_t m = a <= 3 ? (uint64_t)(a - 1) : 2 * (uint64_t)std::ceil(std::log2((double)a));
      p.flops_alg += m + nonunit;
    } else {
      p.flops_alg += (k - 1) + nonunit;   // tree_properties.jl:165-185 + factor multiplies
    }
  }
  p.n_live_leaf = 0;
  for (uint32_t i = 0; i < L; ++i) p.n_live_leaf += p.live[i];

  // peak live set with leaves fetched on demand: intermediates only
  std::vector<uint32_t> last((size_t)L + N, 0);
  for (uint32_t pos = 0; pos < p.order.size(); ++pos) {
    uint32_t n = p.order[pos];
    for (uint32_t e = p.off[n]; e < p.off[n + 1]; ++e) last[p.idx[e]] = pos;
  }
  std::vector<int32_t> delta(p.order.size() + 2, 0);
  for (uint32_t pos = 0; pos < p.order.size(); ++pos) {
    uint32_t v = L + p.order[pos];
    uint32_t lu = std::max(last[v], pos);
    delta[pos] += 1;
    delta[lu + 1] -= 1;
  }
  int32_t cur = 0, peak = 0;
  for (size_t i = 0; i < delta.size(); ++i) { cur += delta[i]; peak = std::max(peak, cur); }
  p.max_live = (uint32_t)peak;
}

namespace {
struct FreeList {
  std::vector<uint32_t> free_;
  uint32_t next = 0, cap;
  explicit FreeList(uint32_t c) : cap(c) {}
  bool get(uint32_t &s) {
    if (!free_.empty()) { s = free_.back(); free_.pop_back(); return true; }
    if (next < cap) { s = next++; return true; }
    return false;
  }
  void put(uint32_t s) { free_.push_back(s); }
};
inline void push_f64(std::vector<uint32_t> &c, double f) {
  uint64_t u; std::memcpy(&u, &f, 8);
  c.push_back((uint32_t)u); c.push_back((uint32_t)(u >> 32));
}
}  // namespace

// Slot allocation for the interpreter: every intermediate gets an LDS slot when
// one is free at its definition, else a workspace (HBM panel) slot; slots are
// recycled after the last use.  Leaves used more than once are copied into a
// free LDS slot at first use (UOP_LEAF) and stay there while live; otherwise an
// operand reads the leaf in place.
void build_interpreter_program(Lowered &p, uint32_t lds_budget) {
  const uint32_t L = p.L;
  const size_t V = (size_t)L + p.N;
  std::vector<uint32_t> last(V, 0), nuse(V, 0);
  for (uint32_t pos = 0; pos < p.order.size(); ++pos) {
    uint32_t n = p.order[pos];
    for (uint32_t e = p.off[n]; e < p.off[n + 1]; ++e) { last[p.idx[e]] = pos; nuse[p.idx[e]]++; }
  }
  // roots per value
  std::vector<std::vector<uint32_t>> roots_of_val;  // sparse: (value, k)
  std::vector<std::pair<uint32_t, uint32_t>> rootlist;
  for (uint32_t k = 0; k < p.R; ++k)
    if (p.root_slot[k] != FDG_NO_ROOT) rootlist.push_back({p.root_slot[k], k});
  std::sort(rootlist.begin(), rootlist.end());

  FreeList lds(lds_budget), mem(std::numeric_limits<uint32_t>::max());
  std::vector<uint32_t> loc(V, 0xFFFFFFFFu);
  std::vector<uint32_t> &c = p.code;
  c.clear();
  p.n_ops = 0; p.opnd_lds = p.opnd_mem = p.opnd_leaf = 0;
  uint32_t lds_hw = 0;

  auto emit_roots = [&](uint32_t v, uint32_t where) {
    auto it = std::lower_bound(rootlist.begin(), rootlist.end(), std::make_pair(v, 0u));
    for (; it != rootlist.end() && it->first == v; ++it) {
      c.push_back(UOP_ROOT | (it->second << 4));
      c.push_back(where);
      p.n_ops++;
    }
  };
  // leaves that are roots themselves (static.jl:115-128 applies to leaves too)
  for (auto &rk : rootlist)
    if (rk.first < L) { c.push_back(UOP_ROOT | (rk.second << 4)); c.push_back(mkloc(SP_LEAF, rk.first)); p.n_ops++; }

  for (uint32_t pos = 0; pos < p.order.size(); ++pos) {
    const uint32_t n = p.order[pos];
    const uint32_t a = p.off[n], b = p.off[n + 1];
    // stage multiply-used leaves into LDS at first touch
    for (uint32_t e = a; e < b; ++e) {
      uint32_t v = p.idx[e];
      if (v < L && loc[v] == 0xFFFFFFFFu && nuse[v] >= 2) {
        uint32_t s;
        if (lds.get(s)) {
          loc[v] = mkloc(SP_LDS, s);
          lds_hw = std::max(lds_hw, s + 1);
          c.push_back(UOP_LEAF); c.push_back(loc[v]); c.push_back(v);
          p.n_ops++;
        } else {
          loc[v] = mkloc(SP_LEAF, v);   // no room: read in place from now on
        }
      }
    }
    const size_t hdr_at = c.size();
    if (p.op[n] == FDG_OP_POWER) c.push_back(UOP_POW | ((uint32_t)(p.power[n] + (1 << 27)) << 4));
    else c.push_back((p.op[n] == FDG_OP_SUM ? UOP_SUM : (p.assoc_interp ? UOP_PRODI : UOP_PROD)) | ((b - a) << 4));
    c.push_back(0);  // dst, patched below
    for (uint32_t e = a; e < b; ++e) {
      uint32_t v = p.idx[e];
      uint32_t w = (v < L && loc[v] == 0xFFFFFFFFu) ? mkloc(SP_LEAF, v) : loc[v];
      switch (w >> 30) { case SP_LDS: p.opnd_lds++; break; case SP_MEM: p.opnd_mem++; break; default: p.opnd_leaf++; }
      if (p.fac[e] != 1.0) { c.push_back(w | LOC_FAC); push_f64(c, p.fac[e]); }
      else c.push_back(w);
    }
    // release operands whose last use is this node (each value once)
    for (uint32_t e = a; e < b; ++e) {
      uint32_t v = p.idx[e];
      if (last[v] == pos && loc[v] != 0xFFFFFFFFu && loc[v] != 0xFFFFFFFEu) {
        uint32_t w = loc[v];
        if ((w >> 30) == SP_LDS) lds.put(w & LOC_IDX_MASK);
        else if ((w >> 30) == SP_MEM) mem.put(w & LOC_IDX_MASK);
        loc[v] = 0xFFFFFFFEu;  // released
      }
    }
    // destination: reads complete before the write in the interpreter, so a
    // slot released just above may be re-used immediately
    const uint32_t v = L + n;
    uint32_t s, w;
    if (lds.get(s)) { w = mkloc(SP_LDS, s); lds_hw = std::max(lds_hw, s + 1); }
    else { mem.get(s); w = mkloc(SP_MEM, s); }
    loc[v] = w;
    c[hdr_at + 1] = w;
    p.n_ops++;
    emit_roots(v, w);
    if (nuse[v] == 0) {  // root-only value: free right away
      if ((w >> 30) == SP_LDS) lds.put(w & LOC_IDX_MASK); else mem.put(w & LOC_IDX_MASK);
      loc[v] = 0xFFFFFFFEu;
    }
  }
  c.push_back(UOP_END);
  p.lds_slots = lds_hw;
  p.mem_slots = mem.next;
}

}  // namespace fdg
