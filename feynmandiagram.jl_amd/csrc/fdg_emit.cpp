// Straight-line HIP source for one graph: the GPU analogue of the reference's
// to_Cstr (src/backend/static.jl:155-197).  One lane evaluates one sample; the
// statement order and each statement's association are the reference's
// (static.jl:13-46: n-ary + and * are left folds, factors equal to 1 are not
// applied); the compiler is told not to contract (-ffp-contract=off) unless
// FDG_SPEC_FAST_MATH is requested.
//
// Two kernels are emitted per graph:
//   fdg_spec_sm  : sample-major leaves (leaf stride 1): a lane reads its own
//                  row with 16-byte loads; roots written as one row per lane.
//   fdg_spec_gen : arbitrary strides (covers the leaf-major / Julia layout,
//                  where consecutive lanes read consecutive addresses).
// mode 0 writes roots, mode 1 accumulates weight*root per lane and leaves one
// partial sum per block and root in `partial` (reduced by fdg_reduce_partials).
#include <cinttypes>
#include <cstdio>
#include <cstring>
#include <functional>
#include <sstream>

#include "fdg_internal.h"
#include "fdg_opt.h"

namespace fdg {

static void put_double(std::ostringstream &os, double f) {
  // hex float: exact, locale independent
  char buf[64];
  std::snprintf(buf, sizeof buf, "%a", f);
  os << buf;
}

static const char *kPrelude = R"SRC(
typedef double fdg_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double fdg_powi_dev(double x, int n) {
  if (n == 2) return x * x;
  if (n == 3) return x * x * x;
  if (n == -1) return 1.0 / x;
  if (n == -2) { double r = 1.0 / x; return r * r; }
  double y = 1.0, xnlo = 0.0, ynlo = 0.0;
  long m = n;
  if (m < 0) {
    double rx = 1.0 / x;
    if (__builtin_isfinite(x)) xnlo = -__builtin_fma(x, rx, -1.0) * rx;
    x = rx; m = -m;
  }
  while (m > 1) {
    if (m & 1) {
      double err = __builtin_fma(y, xnlo, x * ynlo);
      double yh = x * y; double yl = __builtin_fma(x, y, -yh);
      y = yh; ynlo = yl + err;
    }
    double err = x * 2 * xnlo;
    double xh = x * x; double xl = __builtin_fma(x, x, -xh);
    x = xh; xnlo = xl + err;
    m >>= 1;
  }
  double err = __builtin_fma(y, xnlo, x * ynlo);
  return (__builtin_isfinite(x) && __builtin_isfinite(err)) ? __builtin_fma(x, y, err) : x * y;
}
__device__ __forceinline__ double fdg_block_sum(double v, double *sh) {
  // fixed-shape pairwise tree over the 256 lanes of the block: deterministic
  const int t = threadIdx.x;
  sh[t] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s) sh[t] = sh[t] + sh[t + s];
    __syncthreads();
  }
  double r = sh[0];
  __syncthreads();
  return r;
}
)SRC";

// A node's expression as explicit left folds.  Sum: terms (c*f or c) folded
// with +.  Prod: the interleaved sequence c1, f1?, c2, f2?, ... folded with *,
// i.e. (((c1*f1)*c2)*f2) -- exactly how Julia parses "(g1 * f1 * g2 * f2)".
static void emit_node_expr(std::ostringstream &os, const Lowered &p, uint32_t n) {
  const uint32_t L = p.L;
  (void)L;
  const uint32_t a = p.off[n], b = p.off[n + 1];
  if (p.op[n] == FDG_OP_POWER) {
    os << "fdg_powi_dev(g" << p.idx[a] << ", " << p.power[n] << ")";
    if (p.fac[a] != 1.0) { os << " * "; put_double(os, p.fac[a]); }
    return;
  }
  // sequence of fold steps
  std::vector<std::string> terms;
  if (p.op[n] == FDG_OP_SUM) {
    for (uint32_t e = a; e < b; ++e) {
      std::ostringstream t;
      if (p.fac[e] != 1.0) { t << "(g" << p.idx[e] << " * "; put_double(t, p.fac[e]); t << ")"; }
      else t << "g" << p.idx[e];
      terms.push_back(t.str());
    }
  } else if (p.assoc_interp) {      // eval!'s product (eval.jl:2): the operands are scaled first, then folded
    for (uint32_t e = a; e < b; ++e) {
      std::ostringstream t;
      if (p.fac[e] != 1.0) { t << "(g" << p.idx[e] << " * "; put_double(t, p.fac[e]); t << ")"; }
      else t << "g" << p.idx[e];
      terms.push_back(t.str());
    }
  } else {
    for (uint32_t e = a; e < b; ++e) {
      terms.push_back("g" + std::to_string(p.idx[e]));
      if (p.fac[e] != 1.0) { std::ostringstream t; put_double(t, p.fac[e]); terms.push_back(t.str()); }
    }
  }
  const char *sep = p.op[n] == FDG_OP_SUM ? " + " : " * ";
  for (size_t i = 1; i < terms.size(); ++i) os << "(";
  os << terms[0];
  for (size_t i = 1; i < terms.size(); ++i) os << sep << terms[i] << ")";
}

// The graph body in the order of the optimizing back end's schedule (fdg_opt.cpp: depth-first streaming folds,
// grouped Taylor coefficients, value numbering) instead of the reference's statement order: same fold steps, same
// bits, but values die soon after they are born, so the compiler keeps them in registers instead of scratch.
// Leaves are g<i>, computed values v<id>; roots are collected in r<k>.  False: the schedule does not cover the
// graph (Power{N}, N not 2 or 3) and the caller falls back to statement order.
static bool emit_nodes_scheduled(std::ostringstream &os, const Lowered &p, const std::function<void(std::ostringstream &, uint32_t)> &load_leaf,
                                 const char *decl = "const double", bool keep_minus_one = false, bool mul_keeps_signs = false) {
  if (fdg::knob("FDG_HIP_TABLE_ORDER")) return false;
  std::vector<SchedOp> ops;
  uint32_t nv = 0;
  std::string why;
  OptParams prm;
  prm.vn_window = 200;
  prm.keep_minus_one = keep_minus_one;
  prm.mul_keeps_signs = mul_keeps_signs;
  if (!build_schedule(p, prm, ops, nv, why)) return false;
  auto ref = [&](uint32_t r) {
    const uint32_t v = r >> 1;
    std::string n = (v < p.L ? "g" : "v") + std::to_string(v);
    return (r & 1u) ? "(-" + n + ")" : n;
  };
  // leaf loads are written where the schedule first needs the leaf (the compiler may still move them, but it
  // no longer starts from "everything live at the top"); load_leaf emits nothing for a leaf it has done
  auto need = [&](uint32_t r) { if ((r >> 1) < p.L && load_leaf) load_leaf(os, r >> 1); };
  for (const SchedOp &o : ops) {
    need(o.a);
    if (o.kind == M_MUL || o.kind == M_ADD) need(o.b);
    switch (o.kind) {
      case M_MUL: os << "    " << decl << " v" << o.d << " = " << ref(o.a) << " * " << ref(o.b) << ";\n"; break;
      case M_ADD: os << "    " << decl << " v" << o.d << " = " << ref(o.a) << " + " << ref(o.b) << ";\n"; break;
      case M_MULC: os << "    " << decl << " v" << o.d << " = " << ref(o.a) << " * "; put_double(os, o.imm); os << ";\n"; break;
      case M_ROOT: os << "    " << decl << " r" << o.d << " = " << ref(o.a) << ";\n"; break;   // (need(o.a) above covers leaf roots)
      default: return false;
    }
  }
  return true;
}

// load_all: statements that define every live leaf g<i> up front (statement order needs them; the fused kernel
// computes them anyway); load_leaf: on-demand definition of one leaf (may be empty when load_all is used).
static void emit_nodes(std::ostringstream &os, const Lowered &p, const std::string &load_all,
                       const std::function<void(std::ostringstream &, uint32_t)> &load_leaf) {
  {
    std::ostringstream body;
    if (!load_leaf) body << load_all;
    if (emit_nodes_scheduled(body, p, load_leaf)) { os << body.str(); return; }
  }
  os << load_all;
  for (uint32_t n : p.order) {
    os << "    const double g" << (p.L + n) << " = ";
    emit_node_expr(os, p, n);
    os << ";\n";
  }
  for (uint32_t k = 0; k < p.R; ++k)
    if (p.root_slot[k] != FDG_NO_ROOT) os << "    const double r" << k << " = g" << p.root_slot[k] << ";\n";
}

// mode 0: roots of the block's samples; mode 1: w * root accumulated per lane, block sums at the end of the kernel
static void emit_outputs(std::ostringstream &os, const Lowered &p, bool sample_major) {
  os << "    if (mode == 0) {\n      if (valid) {\n";
  os << "        double *rp = root + b * rs;\n";
  bool pair_ok = sample_major;
  for (uint32_t k = 0; k < p.R; ++k) {
    if (p.root_slot[k] == FDG_NO_ROOT) continue;
    if (pair_ok && k % 2 == 0 && k + 1 < p.R && p.root_slot[k + 1] != FDG_NO_ROOT) {
      os << "        if (rk == 1 && ((rs & 1l) == 0l) && ((((unsigned long)root) & 15ul) == 0ul)) { fdg_d2 t; t.x = r"
         << k << "; t.y = r" << (k + 1) << "; __builtin_nontemporal_store(t, (fdg_d2 *)(rp + " << k
         << ")); } else { rp[" << k << "l * rk] = r" << k << "; rp[" << (k + 1) << "l * rk] = r"
         << (k + 1) << "; }\n";
      ++k;
    } else {
      os << "        rp[" << k << "l * rk] = r" << k << ";\n";
    }
  }
  os << "      }\n    } else {\n";
  os << "      const double w = valid ? (weight ? weight[b] : 1.0) : 0.0;\n";
  for (uint32_t k = 0; k < p.R; ++k)
    if (p.root_slot[k] != FDG_NO_ROOT) os << "      acc" << k << " = acc" << k << " + w * r" << k << ";\n";
  os << "    }\n";
  os << "  }\n";
  os << "  if (mode != 0) {\n";
  for (uint32_t k = 0; k < p.R; ++k) {
    os << "    { double s = fdg_block_sum(acc" << k << ", fdg_sh); if (threadIdx.x == 0) partial[(long)blockIdx.x * "
       << p.R << " + " << k << "] = s; }\n";
  }
}

static void emit_kernel(std::ostringstream &os, const Lowered &p, bool sample_major) {
  const uint32_t L = p.L;
  os << "extern \"C\" __global__ void __launch_bounds__(256) " << (sample_major ? "fdg_spec_sm" : "fdg_spec_gen")
     << "(const double *__restrict__ leaf, long ss, long ls, double *__restrict__ root, long rs, long rk,\n"
        "    const double *__restrict__ weight, double *__restrict__ partial, long B, int mode) {\n";
  os << "  __shared__ double fdg_sh[256];\n";
  for (uint32_t k = 0; k < p.R; ++k) os << "  double acc" << k << " = 0.0;\n";
  os << "  const long nblk = (B + 255) / 256;\n";
  os << "  _Pragma(\"unroll 1\")\n";
  os << "  for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {\n";
  os << "    const long b0 = blk * 256 + threadIdx.x;\n";
  os << "    const bool valid = b0 < B;\n";
  os << "    const long b = valid ? b0 : (B - 1);\n";
  // body
  os << "    const double *lp = leaf + b * ss;\n";
  if (sample_major) os << "    const bool al = ((((unsigned long)leaf) & 15ul) == 0ul) && ((ss & 1l) == 0l);\n";
  std::vector<uint8_t> done(L, 0);
  // one leaf (or, sample-major, the aligned pair it belongs to: one 16-byte load)
  auto load_leaf = [&](std::ostringstream &o, uint32_t i) {
    if (done[i] || !p.live[i]) return;
    if (sample_major) {
      const uint32_t e = i & ~1u;
      if (e + 1 < L && p.live[e] && p.live[e + 1]) {
        o << "    double g" << e << ", g" << (e + 1) << ";\n";
        o << "    if (al) { fdg_d2 t = __builtin_nontemporal_load((const fdg_d2 *)(lp + " << e << ")); g" << e
          << " = t.x; g" << (e + 1) << " = t.y; } else { g" << e << " = __builtin_nontemporal_load(lp + " << e
          << "); g" << (e + 1) << " = __builtin_nontemporal_load(lp + " << (e + 1) << "); }\n";
        done[e] = done[e + 1] = 1;
        return;
      }
      o << "    const double g" << i << " = __builtin_nontemporal_load(lp + " << i << ");\n";
    } else {
      o << "    const double g" << i << " = __builtin_nontemporal_load(lp + " << i << "l * ls);\n";
    }
    done[i] = 1;
  };
  std::ostringstream all;
  {
    std::vector<uint8_t> keep = done;
    for (uint32_t i = 0; i < L; ++i) load_leaf(all, i);
    done = keep;                                   // (the string is only used when the lazy form is not)
  }
  emit_nodes(os, p, all.str(), load_leaf);
  // outputs
  emit_outputs(os, p, sample_major);
  os << "  }\n}\n\n";
}

std::string emit_hip_source(const Lowered &p, unsigned flags) {
  (void)flags;
  std::ostringstream os;
  os << "// generated by fdg_graph_emit_source: L=" << p.L << " N=" << p.N << " (live " << p.order.size() << ") R=" << p.R
     << " flops_alg=" << p.flops_alg << "\n";
  os << "#include <hip/hip_runtime.h>\n";
  os << kPrelude;
  emit_kernel(os, p, true);
  emit_kernel(os, p, false);
  return os.str();
}

// Element types other than Float64.  The function Compilers.compile returns is generic in eltype(leafVal) (static.jl:98-133: the
// text has no type in it), and to_Cstr maps the weight types it knows (static.jl:135-153: Float32, ComplexF32, ComplexF64, ...).
// What the generic function computes follows from Julia's promotion rules, which for these operations are C++'s: the factors are
// printed into the text as Float64 literals (string interpolation of an F = Float64 factor), so `g * 1.5` of a Float32 g is a
// Float64, a product of two Float32 values stays Float32, Complex{T} * Real scales both components, Complex * Complex is
// (ar br - ai bi, ar bi + ai br) without contraction, and `root[k] = g` converts to eltype(root).  The kernel is the scheduled
// body of fdg_spec_gen with `auto` values over a small complex type whose operators follow base/complex.jl.
static const char *kTypedPrelude = R"SRC(
template <class T> struct fdg_cx { T re, im; };
template <class A, class B> __device__ __forceinline__ auto operator*(fdg_cx<A> z, fdg_cx<B> w) -> fdg_cx<decltype(z.re * w.re)> {
  return {z.re * w.re - z.im * w.im, z.re * w.im + z.im * w.re};                  // base/complex.jl: *(z::Complex, w::Complex)
}
template <class A> __device__ __forceinline__ auto operator*(fdg_cx<A> z, double x) -> fdg_cx<decltype(z.re * x)> { return {z.re * x, z.im * x}; }
template <class A> __device__ __forceinline__ auto operator*(fdg_cx<A> z, float x) -> fdg_cx<decltype(z.re * x)> { return {z.re * x, z.im * x}; }
template <class A, class B> __device__ __forceinline__ auto operator+(fdg_cx<A> z, fdg_cx<B> w) -> fdg_cx<decltype(z.re + w.re)> {
  return {z.re + w.re, z.im + w.im};
}
template <class A> __device__ __forceinline__ fdg_cx<A> operator-(fdg_cx<A> z) { return {-z.re, -z.im}; }
template <class T, class S> __device__ __forceinline__ T fdg_conv(S x) { return (T)x; }
template <class T, class S> __device__ __forceinline__ T fdg_conv_cx(fdg_cx<S> z) { T r; r.re = z.re; r.im = z.im; return r; }
)SRC";

std::string emit_hip_source_typed(const Lowered &p, int dtype, bool &ok, std::string &why) {
  ok = true;
  const bool cx = dtype == FDG_DT_C64 || dtype == FDG_DT_C32;
  const char *T = dtype == FDG_DT_F32 ? "float" : dtype == FDG_DT_C64 ? "fdg_cx<double>" : dtype == FDG_DT_C32 ? "fdg_cx<float>" : "double";
  std::ostringstream os;
  os << "// generated by fdg_graph_specialize_typed: element type " << T << ", L=" << p.L << " N=" << p.N << " R=" << p.R << "\n";
  os << "#include <hip/hip_runtime.h>\n" << kTypedPrelude;
  os << "extern \"C\" __global__ void __launch_bounds__(256) fdg_spec_typed(const " << T << " *__restrict__ leaf, long ss, long ls, "
     << T << " *__restrict__ root, long rs, long rk, long B) {\n";
  os << "  const long nblk = (B + 255) / 256;\n  _Pragma(\"unroll 1\")\n  for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {\n";
  os << "    const long b = blk * 256 + threadIdx.x;\n    if (b >= B) continue;\n";
  os << "    const " << T << " *lp = leaf + b * ss;\n";
  std::vector<uint8_t> done(p.L, 0);
  auto load_leaf = [&](std::ostringstream &o, uint32_t i) {
    if (done[i] || !p.live[i]) return;
    o << "    const " << T << " g" << i << " = lp[" << i << "l * ls];\n";
    done[i] = 1;
  };
  std::ostringstream body;
  // Single precision: `g * -1.0` stays a multiplication (it widens the value: Julia's promotion).  ComplexF64: -1 still rides as a sign
  // (z * -1.0 is -z exactly, zeros included), but a product consumes a negated operand as it is -- ((-z) w) and -(z w) differ in the
  // sign of a real part that cancels exactly: (-P1) + P2 = +0 where -(P1 - P2) = -0
  const bool single = dtype == FDG_DT_F32 || dtype == FDG_DT_C32;
  if (!emit_nodes_scheduled(body, p, load_leaf, "const auto", single, !single)) {
    ok = false;
    why = "element types other than Float64 cover Sum, Prod and Power{2}, Power{3} (other literal powers go through Base.power_by_squaring / pow_body per type)";
    return std::string();
  }
  os << body.str();
  os << "    " << T << " *rp = root + b * rs;\n";
  for (uint32_t k = 0; k < p.R; ++k)
    if (p.root_slot[k] != FDG_NO_ROOT)
      os << "    rp[" << k << "l * rk] = " << (cx ? "fdg_conv_cx<" : "fdg_conv<") << T << ">(r" << k << ");\n";      // setindex!: convert(eltype(root), g)
  os << "  }\n}\n";
  return os.str();
}

// Fused Monte-Carlo step: the leaves are not read from memory but worked out in
// registers from the sample's momenta and times (`leaf_stmts`, produced by the runtime from the leafstates
// tables: it declares and assigns g0 .. g{L-1}); the graph body and the outputs are those of fdg_spec_gen.
std::string emit_fused_source(const Lowered &p, const std::string &leaf_stmts, const std::string &device_functions) {
  std::ostringstream os;
  os << "// generated: fused leaves + graph, L=" << p.L << " N=" << p.N << " R=" << p.R << "\n";
  os << "#include <hip/hip_runtime.h>\n";
  os << kPrelude;
  os << device_functions << "\n";
  os << "extern \"C\" __global__ void __launch_bounds__(256) fdg_spec_fused(const double *__restrict__ K, long ks, long kc,\n"
        "    const double *__restrict__ T, long ts, long tc, double kF, double beta, double lambda,\n"
        "    double *__restrict__ root, long rs, long rk, const double *__restrict__ weight, double *__restrict__ partial, long B, int mode) {\n";
  os << "  __shared__ double fdg_sh[256];\n";
  for (uint32_t k = 0; k < p.R; ++k) os << "  double acc" << k << " = 0.0;\n";
  os << "  const long nblk = (B + 255) / 256;\n";
  os << "  _Pragma(\"unroll 1\")\n";
  os << "  for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {\n";
  os << "    const long b0 = blk * 256 + threadIdx.x;\n";
  os << "    const bool valid = b0 < B;\n";
  os << "    const long b = valid ? b0 : (B - 1);\n";
  emit_nodes(os, p, leaf_stmts, nullptr);
  emit_outputs(os, p, false);
  os << "  }\n}\n\n";
  return os.str();
}

}  // namespace fdg
