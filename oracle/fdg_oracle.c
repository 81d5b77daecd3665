/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.
 *
 * CPU restatement, in plain C, of the reference's two evaluators over the flat
 * node table (value indices: leaves 0..L-1, then internal nodes in statement
 * order):
 *
 *  oracle_eval_static  -- the arithmetic of the code Compilers.compile emits
 *      (reference src/backend/static.jl:13-46 expression forms, :98-133
 *      statement order).  Julia parses n-ary `a + b + c` / `a * b * c` into one
 *      call that Base evaluates as a left fold, emits no FMA and does not
 *      reassociate; a factor equal to 1 is not applied (static.jl:15,18,25,28):
 *        Sum   : fold(+) over terms  t_i = c_i * f_i   (or c_i when f_i == 1)
 *        Prod  : fold(*) over the interleaved sequence c_1, f_1?, c_2, f_2?, ...
 *        Power : (c)^N [* f];  N = 2 -> c*c, N = 3 -> c*c*c (Base.literal_pow),
 *                N = -1 -> 1/c, N = -2 -> (1/c)^2, otherwise pow_body (below).
 *      A one-child Sum/Prod is "(c [* f])" (static.jl:14-16,24-26).
 *
 *  oracle_eval_interp  -- the tree interpreter ComputationalGraphs.eval!
 *      (reference src/computational_graph/eval.jl:1-3,15-39): different
 *      association -- every edge multiplies by its factor, even 1:
 *        Sum  = foldl(+, w_i * f_i),  Prod = foldl(*, w_i * f_i),
 *        Power = w^N * f.
 *
 * Compile with -O2 -ffp-contract=off (see Makefile) so the C compiler does not
 * fuse what Julia would not.
 *
 * Parity status: the reference is Julia, which is not installed here, so this
 * file cannot be checked against outputs of the reference itself.  It is pinned
 * by the reference's own known-answer tests that reach this path
 * (tests/test_oracle_kat.py: test/compiler.jl:4-15, test/computational_graph.jl:
 * 874-887, test/taylor.jl:115-161,202; README/assets sigma_o2 structure).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define OP_SUM 0
#define OP_PROD 1
#define OP_POWER 2
#define NO_ROOT 0xFFFFFFFFu

/* Julia Base.Math.pow_body(x::Float64, n::Integer), Julia >= 1.8 (base/math.jl;
 * Julia Base is a dependency of the reference, not vendored in /root/reference).
 * muladd is restated as fma. */
static double pow_body(double x, int64_t n) {
  double y = 1.0, xnlo = 0.0, ynlo = 0.0;
  if (n == 3) return x * x * x;
  if (n < 0) {
    double rx = 1.0 / x;
    if (n == -2) return rx * rx;
    if (isfinite(x)) xnlo = -fma(x, rx, -1.0) * rx;
    x = rx;
    n = -n;
  }
  while (n > 1) {
    if (n & 1) {
      double err = fma(y, xnlo, x * ynlo);
      double yh = x * y, yl = fma(x, y, -yh);
      y = yh;
      ynlo = yl + err;
    }
    double err = x * 2 * xnlo;
    double xh = x * x, xl = fma(x, x, -xh);
    x = xh;
    xnlo = xl + err;
    n >>= 1;
  }
  double err = fma(y, xnlo, x * ynlo);
  return (isfinite(x) && isfinite(err)) ? fma(x, y, err) : x * y;
}

/* Base.literal_pow for a literal exponent, then ^(::Float64, ::Int) */
static double julia_pow_literal(double x, int32_t n) {
  switch (n) {
    case 0: return 1.0;
    case 1: return x;
    case 2: return x * x;
    case 3: return x * x * x;
    case -1: return 1.0 / x;
    case -2: { double i = 1.0 / x; return i * i; }
    default: return pow_body(x, n);
  }
}

double oracle_powi(double x, int32_t n) { return julia_pow_literal(x, n); }

/* elementwise IEEE fused multiply-add, for the tests' replay of programs that contain pow_body's fma steps */
void oracle_fma(const double *a, const double *b, const double *c, double *out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) out[i] = fma(a[i], b[i], c[i]);
}

/* val: scratch of L+N doubles. Returns 0, or -1 on a malformed table. */
static int eval_one_static(uint32_t L, uint32_t N, const uint8_t *op, const int32_t *power,
                           const uint32_t *off, const uint32_t *idx, const double *fac,
                           const double *leaf, double *val) {
  memcpy(val, leaf, (size_t)L * sizeof(double));
  for (uint32_t n = 0; n < N; ++n) {
    uint32_t a = off[n], b = off[n + 1];
    double acc;
    if (op[n] == OP_POWER) {
      acc = julia_pow_literal(val[idx[a]], power[n]);
      if (fac[a] != 1.0) acc = acc * fac[a];
    } else if (op[n] == OP_SUM) {
      acc = val[idx[a]];
      if (fac[a] != 1.0) acc = acc * fac[a];
      for (uint32_t e = a + 1; e < b; ++e) {
        double t = val[idx[e]];
        if (fac[e] != 1.0) t = t * fac[e];
        acc = acc + t;
      }
    } else if (op[n] == OP_PROD) {
      acc = val[idx[a]];
      if (fac[a] != 1.0) acc = acc * fac[a];
      for (uint32_t e = a + 1; e < b; ++e) {
        acc = acc * val[idx[e]];
        if (fac[e] != 1.0) acc = acc * fac[e];
      }
    } else {
      return -1; /* static.jl:6-11 */
    }
    val[L + n] = acc;
  }
  return 0;
}

static int eval_one_interp(uint32_t L, uint32_t N, const uint8_t *op, const int32_t *power,
                           const uint32_t *off, const uint32_t *idx, const double *fac,
                           const double *leaf, double *val) {
  memcpy(val, leaf, (size_t)L * sizeof(double));
  for (uint32_t n = 0; n < N; ++n) {
    uint32_t a = off[n], b = off[n + 1];
    double acc;
    if (op[n] == OP_POWER) {
      acc = julia_pow_literal(val[idx[a]], power[n]) * fac[a]; /* eval.jl:3 */
    } else if (op[n] == OP_SUM) {
      acc = val[idx[a]] * fac[a];
      for (uint32_t e = a + 1; e < b; ++e) acc = acc + val[idx[e]] * fac[e]; /* eval.jl:1 */
    } else if (op[n] == OP_PROD) {
      acc = val[idx[a]] * fac[a];
      for (uint32_t e = a + 1; e < b; ++e) acc = acc * (val[idx[e]] * fac[e]); /* eval.jl:2 */
    } else {
      return -1;
    }
    val[L + n] = acc;
  }
  return 0;
}

static int eval_batch(int which, uint32_t L, uint32_t N, uint32_t R, const uint8_t *op,
                      const int32_t *power, const uint32_t *off, const uint32_t *idx,
                      const double *fac, const uint32_t *root_slot, const double *leaf,
                      int64_t ls_sample, int64_t ls_leaf, double *root, int64_t rs_sample,
                      int64_t rs_root, int64_t B) {
  double *val = (double *)malloc(((size_t)L + N + 1) * sizeof(double));
  double *lv = (double *)malloc(((size_t)L + 1) * sizeof(double));
  if (!val || !lv) { free(val); free(lv); return -2; }
  int rc = 0;
  for (int64_t b = 0; b < B && rc == 0; ++b) {
    for (uint32_t i = 0; i < L; ++i) lv[i] = leaf[b * ls_sample + (int64_t)i * ls_leaf];
    rc = which == 0 ? eval_one_static(L, N, op, power, off, idx, fac, lv, val)
                    : eval_one_interp(L, N, op, power, off, idx, fac, lv, val);
    for (uint32_t k = 0; k < R; ++k)
      if (root_slot[k] != NO_ROOT) root[b * rs_sample + (int64_t)k * rs_root] = val[root_slot[k]];
  }
  free(val);
  free(lv);
  return rc;
}

int oracle_eval_static(uint32_t L, uint32_t N, uint32_t R, const uint8_t *op, const int32_t *power,
                       const uint32_t *off, const uint32_t *idx, const double *fac,
                       const uint32_t *root_slot, const double *leaf, int64_t ls_sample,
                       int64_t ls_leaf, double *root, int64_t rs_sample, int64_t rs_root, int64_t B) {
  return eval_batch(0, L, N, R, op, power, off, idx, fac, root_slot, leaf, ls_sample, ls_leaf, root,
                    rs_sample, rs_root, B);
}

int oracle_eval_interp(uint32_t L, uint32_t N, uint32_t R, const uint8_t *op, const int32_t *power,
                       const uint32_t *off, const uint32_t *idx, const double *fac,
                       const uint32_t *root_slot, const double *leaf, int64_t ls_sample,
                       int64_t ls_leaf, double *root, int64_t rs_sample, int64_t rs_root, int64_t B) {
  return eval_batch(1, L, N, R, op, power, off, idx, fac, root_slot, leaf, ls_sample, ls_leaf, root,
                    rs_sample, rs_root, B);
}

/* Sum over the absolute values of the terms of each root's own Sum node: the
 * scale S_k(b) of the comparison |d| <= tol * max(1, S_k). */
int oracle_root_scale(uint32_t L, uint32_t N, uint32_t R, const uint8_t *op, const int32_t *power,
                      const uint32_t *off, const uint32_t *idx, const double *fac,
                      const uint32_t *root_slot, const double *leaf, int64_t ls_sample,
                      int64_t ls_leaf, double *scale, int64_t B) {
  double *val = (double *)malloc(((size_t)L + N + 1) * sizeof(double));
  double *lv = (double *)malloc(((size_t)L + 1) * sizeof(double));
  if (!val || !lv) { free(val); free(lv); return -2; }
  int rc = 0;
  for (int64_t b = 0; b < B && rc == 0; ++b) {
    for (uint32_t i = 0; i < L; ++i) lv[i] = leaf[b * ls_sample + (int64_t)i * ls_leaf];
    rc = eval_one_static(L, N, op, power, off, idx, fac, lv, val);
    for (uint32_t k = 0; k < R; ++k) {
      double s = 0.0;
      uint32_t v = root_slot[k];
      if (v == NO_ROOT) { scale[b * R + k] = 0.0; continue; }
      if (v < L || op[v - L] != OP_SUM) s = fabs(val[v]);
      else for (uint32_t e = off[v - L]; e < off[v - L + 1]; ++e) s += fabs(val[idx[e]] * fac[e]);
      scale[b * R + k] = s;
    }
  }
  free(val);
  free(lv);
  return rc;
}
