// Leaves of the integrand on the device: fdg_leaf_eval_device -- the leaf loop of
// example/benchmark.jl:58-81,113-127 over the FrontEnds.leafstates tables, table-driven or JIT-specialised to
// the tables -- and the fused Monte-Carlo step (fdg_graph_specialize_fused, fdg_mc_*_device), where the same
// leaf code runs in registers in front of the graph body.  gfx950 only; no CPU path.
#include <hip/hip_runtime.h>
#include <sys/stat.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <sstream>

#define FDG_RUNTIME_TU 1   // fdg_internal.h then also declares the helpers that need the HIP runtime types
#include "fdg_internal.h"
#include "fdg_powi.h"

using namespace fdg;

// green_derive of example/benchmark.jl:93-111 for orders 1..5: (-1)^n / n! d^n/dw^n of the fermionic kernel
// K(tau, w) = exp(-w tau) / (1 + exp(-w beta)), antiperiodic in tau.  The derivative lives in Lehmann.jl
// (Spectral.kernelFermiT_dw*), which is not part of the reference checkout; this is the published definition in
// an overflow-safe form, the twin of oracle.green_derive (pinned by 60-digit mpmath vectors, tests/golden/
// green_derive.npz): on each of green()'s four branches K = sgn A g with A = exp(w a), g = 1/(1+exp(-|w| beta)),
// d^j A = a^j A, d^k g = b^k Q_k(g) (b = +-beta, Q_0 = g, Q_{k+1} = Q_k' g (1-g)), summed by Leibniz.
// One definition, compiled here for the table-driven kernel and pasted as text into the JIT sources.
#define FDG_STRINGIFY(...) #__VA_ARGS__
#define FDG_SHARED_DEVICE_CODE(...) __VA_ARGS__ static const char *kFermiDnSource = FDG_STRINGIFY(__VA_ARGS__);
FDG_SHARED_DEVICE_CODE(
__device__ __forceinline__ double fdg_fermi_dn(double tau, double w, double beta, int n) {
  const double QC[6][7] = {{0, 1, 0, 0, 0, 0, 0}, {0, 1, -1, 0, 0, 0, 0}, {0, 1, -3, 2, 0, 0, 0}, {0, 1, -7, 12, -6, 0, 0},
                           {0, 1, -15, 50, -60, 24, 0}, {0, 1, -31, 180, -390, 360, -120}};
  const double BC[6][6] = {{1, 0, 0, 0, 0, 0}, {1, 1, 0, 0, 0, 0}, {1, 2, 1, 0, 0, 0}, {1, 3, 3, 1, 0, 0}, {1, 4, 6, 4, 1, 0}, {1, 5, 10, 10, 5, 1}};
  const double NF[6] = {1.0, -1.0, 0.5, -1.0 / 6.0, 1.0 / 24.0, -1.0 / 120.0};
  if (tau == 0.0) tau = -1e-10;
  const bool neg = tau < 0.0;
  const bool pos = w >= 0.0;
  const double a = pos ? (neg ? -(tau + beta) : -tau) : (neg ? -tau : beta - tau);
  const double A = exp(w * a);
  const double g = 1.0 / (1.0 + exp(-fabs(w) * beta));
  const double b = pos ? beta : -beta;
  double total = 0.0;
  for (int k = 0; k <= n; ++k) {
    double q = 0.0;
    for (int c = k + 1; c >= 0; --c) q = q * g + QC[k][c];
    double term = BC[n][k] * q;
    for (int j = 0; j < n - k; ++j) term = term * a;
    for (int j = 0; j < k; ++j) term = term * b;
    total = total + term;
  }
  return (neg ? -1.0 : 1.0) * A * total * NF[n];
}
)

// Leaf values from (K, T): one lane = one sample; the tables are wave-uniform (scalar loads); the
// sample's momenta and times are staged once in LDS columns.  The host hands the leaves over sorted by
// (type, loop-basis index): many propagators carry the same momentum (GV 4-loop self-energy: 89 fermionic
// leaves, 23 distinct momenta), so q^2, the dispersion and the Fermi denominator are computed once per
// distinct momentum and only the tau-dependent exponential per leaf.  Same expressions, same order of
// operations as example/benchmark.jl:113-127 -- evaluated once instead of once per leaf.
__global__ void __launch_bounds__(64)
fdg_leaf_kernel(const int32_t *__restrict__ ltype, const int32_t *__restrict__ lorder, const int32_t *__restrict__ tin,
                const int32_t *__restrict__ tout, const int32_t *__restrict__ lidx, const int32_t *__restrict__ oidx,
                const double *__restrict__ basis,
                uint32_t L, uint32_t n_loop, uint32_t dim, uint32_t n_tau, double kF, double beta, double lambda,
                const double *__restrict__ K, long ks, long kc, const double *__restrict__ T, long ts, long tc,
                double *__restrict__ leaf, long ss, long ls, long lts, long B) {
  extern __shared__ double sh[];                 // [(n_loop*dim + n_tau)][64]
  const int t = threadIdx.x;
  double *kk = sh + t;
  double *tt = sh + (size_t)n_loop * dim * 64 + t;
  const long ntile = (B + 63) / 64;
  for (long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const long b0 = tile * 64 + t;
    const bool valid = b0 < B;
    const long b = valid ? b0 : B - 1;
    const long bl = tile * lts + (long)t * ss;          // this lane's sample in the leaf batch (tile stride: plain matrices pass 64 ss)
    for (uint32_t c = 0; c < n_loop * dim; ++c) kk[(size_t)c * 64] = K[b * ks + (long)c * kc];
    for (uint32_t i = 0; i < n_tau; ++i) tt[(size_t)i * 64] = T[b * ts + (long)i * tc];
    int32_t cur = -1;
    double q2 = 0.0, w = 0.0, den = 1.0;
    for (uint32_t i = 0; i < L; ++i) {
      const int32_t ty = ltype[i];
      if (ty == 0) continue;
      if (lidx[i] != cur) {                      // wave-uniform: a new momentum
        cur = lidx[i];
        const double *bv = basis + (size_t)(cur - 1) * n_loop;
        q2 = 0.0;
        for (uint32_t d = 0; d < dim; ++d) {
          double q = 0.0;
          for (uint32_t j = 0; j < n_loop; ++j) q += kk[(size_t)(j * dim + d) * 64] * bv[j];
          q2 += q * q;
        }
        w = q2 - kF * kF;
        den = 1.0 + exp(-fabs(w) * beta);        // 1 + exp(-w beta) for w > 0, 1 + exp(w beta) otherwise
      }
      double v;
      if (ty == 1) {
        double tau = tt[(size_t)(tout[i] - 1) * 64] - tt[(size_t)(tin[i] - 1) * 64];
        if (lorder[i] != 0) {                    // wave-uniform
          v = fdg_fermi_dn(tau, w, beta, lorder[i]);
          if (valid) leaf[bl + (long)oidx[i] * ls] = v;
          continue;
        }
        if (tau == 0.0) tau = -1e-10;
        // one exponential per lane: the four cases of green() differ in the argument and the sign only
        // (lanes of a wave fall into different cases; branching would evaluate exp() once per case)
        const double a_pos = w > 0.0 ? -w * tau : w * (beta - tau);
        const double a_neg = w > 0.0 ? -w * (tau + beta) : -w * tau;
        const double e = exp(tau > 0.0 ? a_pos : a_neg);
        v = (tau > 0.0 ? e : -e) / den;
      } else {
        const double invK = 1.0 / (q2 + lambda);
        v = 8.0 * 3.141592653589793 / invK * fdg_powi_impl(lambda * invK, lorder[i] == 0 ? 0 : lorder[i]);
      }
      if (valid) leaf[bl + (long)oidx[i] * ls] = v;
    }
  }
}

// ---------------------------------------------------------------------------
// Leaf kernel specialised to one set of leafstates tables (the same JIT route as the graph kernels):
// indices, the loop basis (almost all entries 0 / +-1) and the leaf order are compile-time constants, so
// the sample's momenta and times stay in registers, zero coefficients vanish and the compiler schedules the
// whole straight-line body.  Same expressions in the same order as fdg_leaf_kernel, hence the same bits:
// skipping q += k * 0.0 and writing k for k * 1.0 cannot change q (only the sign of a zero that is squared).
// ---------------------------------------------------------------------------
// `fused` = false: statements store each leaf to leaf[b*ss + i*ls]; true: they assign g<i> (declared here, 1.0 for
// leaves without a formula: leafstates' initial leafValue) for the graph body that follows.
static std::string emit_leaf_statements(const fdg_leaf_tables *tab, const std::vector<int32_t> &perm, bool fused) {
  std::ostringstream os;
  auto dbl = [&](double f) { char b[64]; std::snprintf(b, sizeof b, "%a", f); return std::string(b); };
  const uint32_t L = tab->n_leaf, nl = tab->n_loop, dim = tab->dim;
  std::vector<uint8_t> k_used(nl * dim, 0), t_used(tab->n_tau + 1, 0);
  for (uint32_t i = 0; i < L; ++i) {
    if (tab->leaf_type[i] == 0) continue;
    for (uint32_t j = 0; j < nl; ++j)
      if (tab->basis[(size_t)(tab->loop_index[i] - 1) * nl + j] != 0.0) for (uint32_t d = 0; d < dim; ++d) k_used[j * dim + d] = 1;
    if (tab->leaf_type[i] == 1) { t_used[tab->tau_in[i]] = 1; t_used[tab->tau_out[i]] = 1; }
  }
  for (uint32_t c = 0; c < nl * dim; ++c) if (k_used[c]) os << "    const double k" << c << " = K[b * ks + " << c << "L * kc];\n";
  for (uint32_t i = 1; i <= tab->n_tau; ++i) if (t_used[i]) os << "    const double t" << i << " = T[b * ts + " << (i - 1) << "L * tc];\n";
  os << "    double q, q2, w, den, tau, ap, an, e, invK, x, v;\n";
  if (fused) for (uint32_t i = 0; i < L; ++i) os << "    double g" << i << " = 1.0;\n";
  int32_t cur = -1;
  for (uint32_t s = 0; s < L; ++s) {
    const int32_t i = perm[s];
    const int32_t ty = tab->leaf_type[i];
    if (ty == 0) continue;
    if (tab->loop_index[i] != cur) {
      cur = tab->loop_index[i];
      const double *bv = tab->basis + (size_t)(cur - 1) * nl;
      os << "    q2 = 0.0;\n";
      for (uint32_t d = 0; d < dim; ++d) {
        os << "    q = 0.0;";
        for (uint32_t j = 0; j < nl; ++j) {
          if (bv[j] == 0.0) continue;
          if (bv[j] == 1.0) os << " q += k" << (j * dim + d) << ";";
          else os << " q += k" << (j * dim + d) << " * " << dbl(bv[j]) << ";";
        }
        os << " q2 += q * q;\n";
      }
      os << "    w = q2 - kF * kF; den = 1.0 + exp(-fabs(w) * beta);\n";
    }
    if (ty == 1 && tab->leaf_order[i] != 0) {
      os << "    v = fdg_fermi_dn(t" << tab->tau_out[i] << " - t" << tab->tau_in[i] << ", w, beta, " << tab->leaf_order[i] << ");\n";
    } else if (ty == 1) {
      os << "    tau = t" << tab->tau_out[i] << " - t" << tab->tau_in[i] << "; if (tau == 0.0) tau = -1e-10;\n"
            "    ap = w > 0.0 ? -w * tau : w * (beta - tau); an = w > 0.0 ? -w * (tau + beta) : -w * tau;\n"
            "    e = exp(tau > 0.0 ? ap : an); v = (tau > 0.0 ? e : -e) / den;\n";
    } else {
      os << "    invK = 1.0 / (q2 + lambda); x = lambda * invK; v = 8.0 * 3.141592653589793 / invK";
      const int32_t n = tab->leaf_order[i];
      if (n == 0) os << " * 1.0";
      else if (n == 1) os << " * x";
      else if (n == 2) os << " * (x * x)";
      else os << " * (x * x * x)";
      os << ";\n";
    }
    if (fused) os << "    g" << i << " = v;\n";
    else os << "    if (valid) leaf[bl + " << i << "L * ls] = v;\n";
  }
  return os.str();
}

static bool needs_fermi_dn(const fdg_leaf_tables *tab) {
  for (uint32_t i = 0; i < tab->n_leaf; ++i) if (tab->leaf_type[i] == 1 && tab->leaf_order[i] != 0) return true;
  return false;
}

static std::string emit_leaf_source(const fdg_leaf_tables *tab, const std::vector<int32_t> &perm) {
  std::ostringstream os;
  os << "#include <hip/hip_runtime.h>\n";
  if (needs_fermi_dn(tab)) os << kFermiDnSource << "\n";
  os << "extern \"C\" __global__ void __launch_bounds__(64) fdg_leaf_spec(const double *__restrict__ K, long ks, long kc,\n"
        "    const double *__restrict__ T, long ts, long tc, double *__restrict__ leaf, long ss, long ls, long B,\n"
        "    double kF, double beta, double lambda, long lts) {\n"
        "  const long ntile = (B + 63) / 64;\n"
        "  for (long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {\n"
        "    const long b0 = tile * 64 + threadIdx.x;\n    const bool valid = b0 < B;\n    const long b = valid ? b0 : B - 1;\n"
        "    const long bl = tile * lts + (long)threadIdx.x * ss;\n";
  os << emit_leaf_statements(tab, perm, false);
  os << "  }\n}\n";
  return os.str();
}

static std::vector<int32_t> leaf_order_by_momentum(const fdg_leaf_tables *tab) {
  std::vector<int32_t> perm(tab->n_leaf);
  for (uint32_t i = 0; i < tab->n_leaf; ++i) perm[i] = (int32_t)i;
  std::stable_sort(perm.begin(), perm.end(), [&](int32_t a, int32_t b) {
    if (tab->leaf_type[a] != tab->leaf_type[b]) return tab->leaf_type[a] < tab->leaf_type[b];
    return tab->loop_index[a] < tab->loop_index[b];
  });
  return perm;
}

static int check_leaf_tables(const fdg_leaf_tables *tab) {
  if (!tab || !tab->leaf_type || !tab->leaf_order || !tab->tau_in || !tab->tau_out || !tab->loop_index || !tab->basis) {
    set_error("null leaf table"); return FDG_E_INVALID;
  }
  for (uint32_t i = 0; i < tab->n_leaf; ++i) {
    const int32_t ty = tab->leaf_type[i];
    if (ty < 0 || ty > 2) { set_error("this leaftype " + std::to_string(ty) + " not implemented!"); return FDG_E_UNSUPPORTED; }  // benchmark.jl:79
    if (ty == 0) continue;
    if (tab->loop_index[i] < 1 || (uint32_t)tab->loop_index[i] > tab->n_basis) { set_error("loop_index out of range"); return FDG_E_INVALID; }
    if (ty == 1) {
      if (tab->leaf_order[i] < 0 || tab->leaf_order[i] > 5) { set_error("not implemented!"); return FDG_E_UNSUPPORTED; }   // green_derive, benchmark.jl:108
      if (tab->tau_in[i] < 1 || tab->tau_out[i] < 1 || (uint32_t)tab->tau_in[i] > tab->n_tau || (uint32_t)tab->tau_out[i] > tab->n_tau) { set_error("tau index out of range"); return FDG_E_INVALID; }
    } else if (tab->leaf_order[i] < 0) { set_error("negative derivative order"); return FDG_E_INVALID; }
  }
  return FDG_OK;
}

struct LeafModule { int dev; std::string key; hipModule_t mod; hipFunction_t fn; };

// returns the specialised kernel for these tables on the current device, or nullptr (caller uses the generic one)
static hipFunction_t leaf_spec_function(const fdg_leaf_tables *tab, const std::vector<int32_t> &perm, int dev) {
  if (fdg::knob("FDG_LEAF_GENERIC")) return nullptr;
  for (uint32_t i = 0; i < tab->n_leaf; ++i)
    if (tab->leaf_type[i] == 2 && (tab->leaf_order[i] < 0 || tab->leaf_order[i] > 3)) return nullptr;   // pow_body lives in the generic kernel
  const std::string src = emit_leaf_source(tab, perm);
  char hbuf[40];
  std::snprintf(hbuf, sizeof hbuf, "%016llx", (unsigned long long)fnv1a(src, fnv1a("leaf-v1")));
  static std::mutex mu;
  static std::vector<LeafModule> cache;
  std::lock_guard<std::mutex> lk(mu);
  for (auto &m : cache) if (m.dev == dev && m.key == hbuf) return m.fn;
  std::string dir;
  const bool have_dir = fdg_cache_dir(nullptr, dir) == FDG_OK;     // no usable cache directory: compile, do not cache
  const std::string base = dir + "/fdg_leaf_" + hbuf;
  std::vector<char> co;
  if (!read_cached(have_dir ? dir : std::string(), std::string("fdg_leaf_") + hbuf + ".hsaco", co)) {
    std::string log;
    if (compile_hiprtc(src, false, co, log) != 0) {
      cache.push_back(LeafModule{dev, hbuf, nullptr, nullptr});   // do not retry on every call
      return nullptr;
    }
    if (have_dir) write_file(base + ".hsaco", co.data(), co.size());
  }
  hipModule_t mod = nullptr;
  hipFunction_t fn = nullptr;
  if (hipModuleLoadData(&mod, co.data()) != hipSuccess || hipModuleGetFunction(&fn, mod, "fdg_leaf_spec") != hipSuccess) {
    (void)hipGetLastError();
    cache.push_back(LeafModule{dev, hbuf, nullptr, nullptr});
    return nullptr;
  }
  cache.push_back(LeafModule{dev, hbuf, mod, fn});
  return fn;
}

extern "C" {

// Everything that depends only on the tables, built once per (device, table contents): the specialised
// kernel (or, for the table-driven one, the sorted tables on the device).  A Monte-Carlo loop calls with
// the same few partitions millions of times; per call nothing is generated, hashed, allocated or copied.
struct LeafPlan {
  int dev = 0, n_cu = 0;
  std::vector<char> key;
  hipFunction_t fn = nullptr;      // specialised kernel, or nullptr: table-driven kernel with d_tab
  char *d_tab = nullptr;
  size_t ib = 0, boff = 0, lds = 0;
};

static int leaf_plan(const fdg_leaf_tables *tab, const LeafPlan **out) {
  int n = 0, dev = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { set_error("no HIP device available (no CPU fallback)"); return FDG_E_NO_DEVICE; }
  HIP_TRY(hipGetDevice(&dev));
  const uint32_t L = tab->n_leaf;
  const size_t ib = (size_t)L * sizeof(int32_t), bb = (size_t)tab->n_basis * tab->n_loop * sizeof(double);
  std::vector<char> key(5 * sizeof(uint32_t) + 5 * ib + bb + 1);
  {
    char *w = key.data();
    const uint32_t hd[5] = {tab->n_leaf, tab->n_basis, tab->n_loop, tab->dim, tab->n_tau};
    std::memcpy(w, hd, sizeof hd); w += sizeof hd;
    const int32_t *src[5] = {tab->leaf_type, tab->leaf_order, tab->tau_in, tab->tau_out, tab->loop_index};
    for (int k = 0; k < 5; ++k) { std::memcpy(w, src[k], ib); w += ib; }
    std::memcpy(w, tab->basis, bb); w += bb;
    *w = fdg::knob("FDG_LEAF_GENERIC") ? 1 : 0;
  }
  static std::mutex mu;
  static std::vector<LeafPlan *> cache;
  std::lock_guard<std::mutex> lk(mu);
  for (LeafPlan *p : cache) if (p->dev == dev && p->key == key) { *out = p; return FDG_OK; }
  LeafPlan *p = new LeafPlan;
  p->dev = dev; p->key.swap(key); p->ib = ib;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, dev));
  p->n_cu = prop.multiProcessorCount;
  // leaves in (type, loop-basis index) order, so that a momentum shared by several leaves is worked out once
  const std::vector<int32_t> perm = leaf_order_by_momentum(tab);
  p->fn = leaf_spec_function(tab, perm, dev);
  if (!p->fn) {
    p->lds = ((size_t)tab->n_loop * tab->dim + tab->n_tau) * 64 * sizeof(double);
    if (p->lds > 160 * 1024) { delete p; set_error("too many momentum/time components for the LDS staging"); return FDG_E_INVALID; }
    p->boff = (6 * ib + 7) & ~(size_t)7;
    std::vector<char> h_tab(p->boff + bb);
    const int32_t *src[5] = {tab->leaf_type, tab->leaf_order, tab->tau_in, tab->tau_out, tab->loop_index};
    for (int k = 0; k < 5; ++k)
      for (uint32_t i = 0; i < L; ++i) ((int32_t *)(h_tab.data() + k * ib))[i] = src[k][perm[i]];
    std::memcpy(h_tab.data() + 5 * ib, perm.data(), ib);
    std::memcpy(h_tab.data() + p->boff, tab->basis, bb);
    if (hipMalloc((void **)&p->d_tab, h_tab.size() + 64) != hipSuccess ||
        hipMemcpy(p->d_tab, h_tab.data(), h_tab.size(), hipMemcpyHostToDevice) != hipSuccess) {
      delete p; set_error("leaf tables: device allocation / copy failed"); return FDG_E_NOMEM;
    }
    if (p->lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)fdg_leaf_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds));
  }
  cache.push_back(p);
  *out = p;
  return FDG_OK;
}

static int leaf_launch(const LeafPlan *p, const fdg_leaf_tables *tab, const double *d_K, int64_t ks, int64_t kc, const double *d_T,
                       int64_t ts, int64_t tc, double *d_leaf, int64_t ss, int64_t ls, int64_t B, hipStream_t st, int64_t lts = 0) {
  const long ntile = (long)((B + 63) / 64);
  if (!lts) lts = 64 * ss;                  // a plain strided matrix
  if (p->fn) {
    long a_ks = ks, a_kc = kc, a_ts = ts, a_tc = tc, a_ss = ss, a_ls = ls, a_B = B, a_lts = lts;
    double a_kF = tab->kF, a_beta = tab->beta, a_lambda = tab->lambda;
    void *args[] = {(void *)&d_K, &a_ks, &a_kc, (void *)&d_T, &a_ts, &a_tc, (void *)&d_leaf, &a_ss, &a_ls, &a_B, &a_kF, &a_beta, &a_lambda, &a_lts};
    const long grid = std::min<long>(ntile, (long)p->n_cu * 32);
    HIP_TRY(hipModuleLaunchKernel(p->fn, (unsigned)grid, 1, 1, 64, 1, 1, 0, st, args, nullptr));
    return FDG_OK;
  }
  const size_t ib = p->ib;
  const char *d_tab = p->d_tab;
  const long per_cu = std::max<long>(1, std::min<long>(32, (160 * 1024) / std::max<size_t>(p->lds, 1)));
  const long grid = std::min<long>(ntile, (long)p->n_cu * per_cu);
  hipLaunchKernelGGL(fdg_leaf_kernel, dim3((unsigned)grid), dim3(64), p->lds, st, (const int32_t *)d_tab, (const int32_t *)(d_tab + ib),
                     (const int32_t *)(d_tab + 2 * ib), (const int32_t *)(d_tab + 3 * ib), (const int32_t *)(d_tab + 4 * ib),
                     (const int32_t *)(d_tab + 5 * ib),
                     (const double *)(d_tab + p->boff), tab->n_leaf, tab->n_loop, tab->dim, tab->n_tau, tab->kF, tab->beta, tab->lambda, d_K,
                     (long)ks, (long)kc, d_T, (long)ts, (long)tc, d_leaf, (long)ss, (long)ls, (long)lts, (long)B);
  HIP_TRY(hipGetLastError());
  return FDG_OK;
}

int fdg_leaf_eval_device(const fdg_leaf_tables *tab, const double *d_K, int64_t ks, int64_t kc, const double *d_T,
                         int64_t ts, int64_t tc, double *d_leaf, int64_t ss, int64_t ls, int64_t B, void *stream) {
  { const int rc0 = check_leaf_tables(tab); if (rc0) return rc0; }
  if (B < 0) { set_error("n_sample < 0"); return FDG_E_INVALID; }
  if (B == 0 || tab->n_leaf == 0) return FDG_OK;
  if (!d_K || !d_T || !d_leaf) { set_error("null device buffer"); return FDG_E_INVALID; }
  const LeafPlan *p = nullptr;
  { const int rc = leaf_plan(tab, &p); if (rc) return rc; }
  return leaf_launch(p, tab, d_K, ks, kc, d_T, ts, tc, d_leaf, ss, ls, B, (hipStream_t)stream);
}

// Tile-major leaves (fdg.h: fdg_eval_device_tiled): sample b of leaf i at d_leaf[(b / 64) lts + (b % 64) ss + i ls].
int fdg_leaf_eval_device_tiled(const fdg_leaf_tables *tab, const double *d_K, int64_t ks, int64_t kc, const double *d_T,
                               int64_t ts, int64_t tc, double *d_leaf, int64_t ss, int64_t ls, int64_t lts, int64_t B, void *stream) {
  { const int rc0 = check_leaf_tables(tab); if (rc0) return rc0; }
  if (B < 0) { set_error("n_sample < 0"); return FDG_E_INVALID; }
  if (lts == 0) { set_error("leaf tile stride 0 (a plain matrix goes through fdg_leaf_eval_device)"); return FDG_E_INVALID; }
  if (B == 0 || tab->n_leaf == 0) return FDG_OK;
  if (!d_K || !d_T || !d_leaf) { set_error("null device buffer"); return FDG_E_INVALID; }
  const LeafPlan *p = nullptr;
  { const int rc = leaf_plan(tab, &p); if (rc) return rc; }
  return leaf_launch(p, tab, d_K, ks, kc, d_T, ts, tc, d_leaf, ss, ls, B, (hipStream_t)stream, lts);
}

// ---- fused Monte-Carlo step --------------------------------------------------------------------
int fdg_graph_specialize_fused(fdg_graph *g, const fdg_leaf_tables *tab, const char *cache_dir, unsigned flags) {
  if (!g) { set_error("null handle"); return FDG_E_INVALID; }
  { const int rc0 = check_leaf_tables(tab); if (rc0) return rc0; }
  if (tab->n_leaf != g->prog.L) { set_error("leaf tables describe " + std::to_string(tab->n_leaf) + " leaves, the graph has " + std::to_string(g->prog.L)); return FDG_E_INVALID; }
  // interaction counter-terms above order 3 need pow_body: the table-driven leaf kernel and the one-kernel route of the
  // optimizing back end have it, the compiler-scheduled fused kernel (route 1) does not
  bool high_order = false;
  for (uint32_t i = 0; i < tab->n_leaf; ++i) high_order = high_order || (tab->leaf_type[i] == 2 && tab->leaf_order[i] > 3);
  std::lock_guard<std::mutex> lk(g->mu);
  fdg::KnobScope knob_scope(&g->knobs);
  // Route.  1: one compiler-scheduled kernel (leaves in registers, HIP source through hiprtc) -- graphs of up to a few
  // thousand operations.  2: the specialised leaf kernel fills a chunk of leaves that this handle's own evaluator
  // consumes -- larger graphs.  3: a handle specialised with FDG_SPEC_ISA whose leaves the optimizing back end's
  // formulas cover: ONE kernel of that back end with the leaves computed in registers (fdg_runtime.hip) -- measured
  // faster than either, taken for everything but tiny graphs.  FDG_MC_ROUTE=fused|split|isa overrides.
  {
    const char *env = fdg::knob("FDG_MC_ROUTE");
    const bool env_fused = env && std::strcmp(env, "fused") == 0, env_split = env && std::strcmp(env, "split") == 0,
               env_isa = env && std::strcmp(env, "isa") == 0;
    if (high_order && env_fused) { set_error("interaction order > 3 is not covered by the fused HIP kernel (FDG_MC_ROUTE=isa, split or unset)"); return FDG_E_UNSUPPORTED; }
    const bool big = (g->prog.flops_alg > 6000 || env_split || high_order) && !env_fused;
    const bool try_isa = env_isa || (g->isa && g->prog.flops_alg > 300 && !env_fused && !env_split);
    const int32_t *src5[5] = {tab->leaf_type, tab->leaf_order, tab->tau_in, tab->tau_out, tab->loop_index};
    for (int k = 0; k < 5; ++k) g->lt_i32[k].assign(src5[k], src5[k] + tab->n_leaf);
    g->lt_basis.assign(tab->basis, tab->basis + (size_t)tab->n_basis * tab->n_loop);
    const uint32_t hd[5] = {tab->n_leaf, tab->n_basis, tab->n_loop, tab->dim, tab->n_tau};
    std::memcpy(g->lt_hdr, hd, sizeof hd);
    g->mc_built = false;
    { const int rcd = fdg_cache_dir(cache_dir, g->mc_dir); if (rcd) return rcd; }
    g->mc_flags = flags;
    if (try_isa) {
      std::string why;
      bool recommended = false;
      if (fdg_mc_isa_supported(g, tab, why, &recommended) && (recommended || env_isa)) {
        g->mc_route = 3; g->fused_code.clear();
        { const int rb = fdg_mc_isa_build(g); if (rb) return rb; }      // assembled here, host-only; parameters are kernel arguments
        return FDG_OK;
      }
      if (env_isa) { set_error("fused ISA step does not cover this graph / these leaves: " + why); return FDG_E_UNSUPPORTED; }
    }
    if (big) { g->mc_route = 2; g->fused_code.clear(); return FDG_OK; }
  }
  const std::string src = emit_fused_source(g->prog, emit_leaf_statements(tab, leaf_order_by_momentum(tab), true), needs_fermi_dn(tab) ? kFermiDnSource : "");
  char hbuf[40];
  std::snprintf(hbuf, sizeof hbuf, "%016llx", (unsigned long long)fnv1a(src, fnv1a("fused-v1")));
  std::string dir;
  { const int rcd = fdg_cache_dir(cache_dir, dir); if (rcd) return rcd; }
  const std::string base = dir + "/fdg_fused_" + hbuf;
  std::vector<char> co;
  if (!read_cached(dir, std::string("fdg_fused_") + hbuf + ".hsaco", co)) {
    std::string log;
    if (compile_hiprtc(src, false, co, log) != 0) {
      std::string log2;
      if (!write_file(base + ".hip", src.c_str(), src.size())) { set_error("cannot write " + base + ".hip"); return FDG_E_JIT; }
      const int rc = compile_hipcc(base + ".hip", base + ".hsaco", false, log2);
      if (!(flags & FDG_SPEC_KEEP_SOURCE)) std::remove((base + ".hip").c_str());
      if (rc != 0 || !read_file(base + ".hsaco", co)) { set_error("fused kernel specialization failed.\nhiprtc: " + log + "\nhipcc: " + log2); return FDG_E_JIT; }
    } else {
      write_file(base + ".hsaco", co.data(), co.size());
    }
  }
  if (flags & FDG_SPEC_KEEP_SOURCE) write_file(base + ".hip", src.c_str(), src.size());
  if (g->fused_module) { hipModuleUnload((hipModule_t)g->fused_module); g->fused_module = nullptr; g->fn_fused = nullptr; }
  g->fused_code.swap(co);
  g->mc_route = 1;
  return FDG_OK;
}

static int run_fused(fdg_graph *g, int mode, const double *d_K, int64_t ks, int64_t kc, const double *d_T, int64_t ts, int64_t tc,
                     double kF, double beta, double lambda, double *d_root, int64_t rs, int64_t rk, const double *d_weight,
                     double *d_acc, int64_t B, void *stream) {
  if (!g) { set_error("null handle"); return FDG_E_INVALID; }
  if (B < 0) { set_error("n_sample < 0"); return FDG_E_INVALID; }
  if (B == 0) return FDG_OK;
  if (!d_K || !d_T || (mode == 0 && !d_root) || (mode == 1 && !d_acc)) { set_error("null device buffer"); return FDG_E_INVALID; }
  std::lock_guard<std::mutex> lk(g->mu);
  fdg::KnobScope knob_scope(&g->knobs);
  if (g->mc_route == 0) { set_error("fdg_graph_specialize_fused has not been called on this handle"); return FDG_E_INVALID; }
  int rc = ensure_device(g);
  if (rc) return rc;
  rc = fdg_bind_stream_ws(g, stream);
  if (rc) return rc;
  if (g->mc_route == 3)
    return fdg_mc_isa_run(g, mode, d_K, ks, kc, d_T, ts, tc, kF, beta, lambda, d_root, rs, rk, d_weight, d_acc, B, (hipStream_t)stream);
  if (g->mc_route == 2) {
    fdg_leaf_tables tab;
    tab.n_leaf = g->lt_hdr[0]; tab.n_basis = g->lt_hdr[1]; tab.n_loop = g->lt_hdr[2]; tab.dim = g->lt_hdr[3]; tab.n_tau = g->lt_hdr[4];
    tab.leaf_type = g->lt_i32[0].data(); tab.leaf_order = g->lt_i32[1].data(); tab.tau_in = g->lt_i32[2].data();
    tab.tau_out = g->lt_i32[3].data(); tab.loop_index = g->lt_i32[4].data(); tab.basis = g->lt_basis.data();
    tab.kF = kF; tab.beta = beta; tab.lambda = lambda;
    const LeafPlan *plan = nullptr;
    rc = leaf_plan(&tab, &plan);
    if (rc) return rc;
    const size_t L = std::max<uint32_t>(g->prog.L, 1);
    int64_t Bc = std::max<int64_t>(1 << 16, (int64_t)((4ull << 30) / (8ull * L)));     // about 4 GiB of leaves per chunk
    if (g->cfg.mc_chunk >= 64) Bc = g->cfg.mc_chunk;   // samples per chunk (tuning)
    Bc = std::min<int64_t>((Bc + 63) & ~63ll, (B + 63) & ~63ll);
    const size_t need = (size_t)Bc * L * sizeof(double);
    if (g->ws4_bytes < need) {
      if (g->d_ws4) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(g->d_ws4)); g->d_ws4 = nullptr; g->ws4_bytes = 0; }
      if (hipMalloc(&g->d_ws4, need) != hipSuccess) { set_error("hipMalloc(leaf chunk) failed"); return FDG_E_NOMEM; }
      g->ws4_bytes = need;
      // leaves without a formula keep leafstates' initial value 1.0
      std::vector<double> ones((size_t)Bc, 1.0);
      for (uint32_t i = 0; i < g->prog.L; ++i)
        if (tab.leaf_type[i] == 0) HIP_TRY(hipMemcpy((double *)g->d_ws4 + (size_t)i * Bc, ones.data(), (size_t)Bc * 8, hipMemcpyHostToDevice));
    }
    hipStream_t st = (hipStream_t)stream;
    double *d_leaf = (double *)g->d_ws4;
    for (int64_t c0 = 0; c0 < B; c0 += Bc) {
      const int64_t n = std::min<int64_t>(Bc, B - c0);
      rc = leaf_launch(plan, &tab, d_K + c0 * ks, ks, kc, d_T + c0 * ts, ts, tc, d_leaf, 1, Bc, n, st);
      if (rc) return rc;
      rc = fdg_run_locked(g, mode, d_leaf, 1, Bc, mode == 0 ? d_root + c0 * rs : nullptr, rs, rk, d_weight ? d_weight + c0 : nullptr, d_acc, n, st);
      if (rc) return rc;
    }
    return FDG_OK;
  }
  if (!g->fn_fused) {
    hipModule_t m; hipFunction_t f;
    hipError_t e = hipModuleLoadData(&m, g->fused_code.data());
    if (e != hipSuccess) { set_error("hipModuleLoadData failed: " + std::string(hipGetErrorString(e))); return FDG_E_JIT; }
    HIP_TRY(hipModuleGetFunction(&f, m, "fdg_spec_fused"));
    g->fused_module = m; g->fn_fused = f;
  }
  const uint32_t R = g->prog.R;
  const long nblk = (long)((B + 255) / 256);
  const long grid = std::min<long>(nblk, (long)g->n_cu * 8);
  double *partial = nullptr;
  if (mode == 1) {
    rc = ensure_ws(g, (size_t)grid * std::max<uint32_t>(R, 1) * sizeof(double));
    if (rc) return rc;
    partial = (double *)g->d_ws;
  }
  long a_ks = ks, a_kc = kc, a_ts = ts, a_tc = tc, a_rs = rs, a_rk = rk, a_B = B;
  int a_mode = mode;
  void *args[] = {(void *)&d_K, &a_ks, &a_kc, (void *)&d_T, &a_ts, &a_tc, &kF, &beta, &lambda, (void *)&d_root, &a_rs, &a_rk,
                  (void *)&d_weight, (void *)&partial, &a_B, &a_mode};
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(hipModuleLaunchKernel((hipFunction_t)g->fn_fused, (unsigned)grid, 1, 1, 256, 1, 1, 0, st, args, nullptr));
  if (mode == 1 && R) {
    { const int rr = launch_reduce_partials(partial, (uint32_t)grid, R, d_acc, st); if (rr) return rr; }
  }
  return FDG_OK;
}

int fdg_mc_eval_device(fdg_graph *g, const double *d_K, int64_t ks, int64_t kc, const double *d_T, int64_t ts, int64_t tc,
                       double kF, double beta, double lambda, double *d_root, int64_t rs, int64_t rk, int64_t B, void *stream) {
  return run_fused(g, 0, d_K, ks, kc, d_T, ts, tc, kF, beta, lambda, d_root, rs, rk, nullptr, nullptr, B, stream);
}

int fdg_mc_accumulate_device(fdg_graph *g, const double *d_K, int64_t ks, int64_t kc, const double *d_T, int64_t ts, int64_t tc,
                             double kF, double beta, double lambda, const double *d_weight, double *d_acc, int64_t B, void *stream) {
  return run_fused(g, 1, d_K, ks, kc, d_T, ts, tc, kF, beta, lambda, nullptr, 0, 0, d_weight, d_acc, B, stream);
}

}  // extern "C"
