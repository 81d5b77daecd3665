"""ORACLE -- TEST INFRASTRUCTURE ONLY (not shipped, never on the product path).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  It holds

* ``eval_static`` / ``eval_interp``: ctypes front end of ``fdg_oracle.c``, the C
  restatement of the reference's compiled evaluator (src/backend/static.jl:13-46,
  98-133) and of its tree interpreter (src/computational_graph/eval.jl:1-39);
* ``eval_static_numpy``: an independent numpy twin of the same arithmetic
  (vectorised over samples; IEEE elementwise ops, so bit-identical to the C);
* ``philox_uniform``: numpy twin of the device leaf generator (Philox4x32-10);
* ``CBaseline``: compiles C text of the reference's ``to_Cstr`` shape
  (static.jl:155-197) with gcc and times it -- the CPU baseline of bench.py.

Parity status: pinned by the reference's own known-answer tests for this path
(tests/test_oracle_kat.py); the reference (Julia) cannot run here, so there is
no oracle/_ref build.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import tempfile
import time
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libfdg_oracle.so")
OP_SUM, OP_PROD, OP_POWER = 0, 1, 2
NO_ROOT = 0xFFFFFFFF


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "fdg_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libfdg_oracle.so"])
    return _LIB


_lib = None


def _load():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        u32, i64, p = C.c_uint32, C.c_int64, C.c_void_p
        sig = [u32, u32, u32, p, p, p, p, p, p, p, i64, i64, p, i64, i64, i64]
        L.oracle_eval_static.argtypes = sig
        L.oracle_eval_interp.argtypes = sig
        L.oracle_root_scale.argtypes = [u32, u32, u32, p, p, p, p, p, p, p, i64, i64, p, i64]
        L.oracle_powi.argtypes = [C.c_double, C.c_int32]
        L.oracle_powi.restype = C.c_double
        L.oracle_fma.argtypes = [p, p, p, p, i64]
        _lib = L
    return _lib


def _arrs(t):
    return (np.ascontiguousarray(t.op, np.uint8), np.ascontiguousarray(t.power, np.int32),
            np.ascontiguousarray(t.child_off, np.uint32), np.ascontiguousarray(t.child_idx, np.uint32),
            np.ascontiguousarray(t.child_fac, np.float64), np.ascontiguousarray(t.root_slot, np.uint32))


def _eval(fn_name: str, t, leaf: np.ndarray, root: Optional[np.ndarray]) -> np.ndarray:
    leaf = np.ascontiguousarray(leaf, dtype=np.float64)
    if leaf.ndim == 1:
        leaf = leaf[None, :]
    B = leaf.shape[0]
    L, N, R = int(t.n_leaf), int(t.op.shape[0]), int(t.root_slot.shape[0])
    if leaf.shape[1] < L:
        raise IndexError("leafVal too short")
    if root is None:
        root = np.zeros((B, R), dtype=np.float64)
    op, pw, off, idx, fac, rs = _arrs(t)
    rc = getattr(_load(), fn_name)(L, N, R, op.ctypes.data, pw.ctypes.data, off.ctypes.data, idx.ctypes.data,
                                   fac.ctypes.data, rs.ctypes.data, leaf.ctypes.data, leaf.shape[1], 1,
                                   root.ctypes.data, R, 1, B)
    if rc != 0:
        raise RuntimeError(f"oracle failed: {rc}")
    return root


def eval_static(table, leaf, root=None) -> np.ndarray:
    """Compiled-evaluator arithmetic (static.jl).  leaf [B, >=L] -> root [B, R]."""
    return _eval("oracle_eval_static", table, leaf, root)


def eval_interp(table, leaf, root=None) -> np.ndarray:
    """Interpreter arithmetic (eval.jl)."""
    return _eval("oracle_eval_interp", table, leaf, root)


def root_scale(table, leaf) -> np.ndarray:
    """S_k(b): sum of |terms| of each root's Sum node (comparison scale)."""
    leaf = np.ascontiguousarray(leaf, dtype=np.float64)
    B = leaf.shape[0]
    L, N, R = int(table.n_leaf), int(table.op.shape[0]), int(table.root_slot.shape[0])
    out = np.zeros((B, R), dtype=np.float64)
    op, pw, off, idx, fac, rs = _arrs(table)
    rc = _load().oracle_root_scale(L, N, R, op.ctypes.data, pw.ctypes.data, off.ctypes.data, idx.ctypes.data,
                                   fac.ctypes.data, rs.ctypes.data, leaf.ctypes.data, leaf.shape[1], 1,
                                   out.ctypes.data, B)
    if rc != 0:
        raise RuntimeError(f"oracle failed: {rc}")
    return out


def abs_graph_scale(table, leaf) -> np.ndarray:
    """A_k(b): the graph evaluated on |leaf| with every factor replaced by its absolute value -- the sum of the absolute
    values of all monomials of root k.  First-order error bound used where leaves themselves carry a rounding difference
    (leaf formulas with another exp): a relative perturbation d of every leaf moves root k by at most deg * d * A_k."""
    import copy
    t = copy.copy(table)
    t.child_fac = np.abs(np.asarray(table.child_fac, dtype=np.float64))
    return eval_static(t, np.abs(np.asarray(leaf, dtype=np.float64)))


def powi(x: float, n: int) -> float:
    return float(_load().oracle_powi(float(x), int(n)))


def fma(a, b, c) -> np.ndarray:
    """Elementwise IEEE fused multiply-add (C's fma), broadcasting."""
    a, b, c = np.broadcast_arrays(np.asarray(a, np.float64), np.asarray(b, np.float64), np.asarray(c, np.float64))
    a, b, c = np.ascontiguousarray(a), np.ascontiguousarray(b), np.ascontiguousarray(c)
    out = np.empty_like(a)
    _load().oracle_fma(a.ctypes.data, b.ctypes.data, c.ctypes.data, out.ctypes.data, a.size)
    return out


def _powi_numpy(x: np.ndarray, n: int) -> np.ndarray:
    if n == 2:
        return x * x
    if n == 3:
        return x * x * x
    return np.array([powi(v, n) for v in x.ravel()], dtype=np.float64).reshape(x.shape)


def eval_static_numpy(table, leaf: np.ndarray) -> np.ndarray:
    """Independent twin of ``eval_static`` in numpy (python loop over nodes)."""
    leaf = np.asarray(leaf, dtype=np.float64)
    if leaf.ndim == 1:
        leaf = leaf[None, :]
    L = int(table.n_leaf)
    N = int(table.op.shape[0])
    vals = [leaf[:, i] for i in range(L)] + [None] * N
    off, idx, fac = table.child_off, table.child_idx, table.child_fac
    for n in range(N):
        a, b = int(off[n]), int(off[n + 1])
        o = int(table.op[n])
        if o == OP_POWER:
            acc = _powi_numpy(vals[int(idx[a])], int(table.power[n]))
            if fac[a] != 1.0:
                acc = acc * fac[a]
        elif o == OP_SUM:
            acc = vals[int(idx[a])]
            if fac[a] != 1.0:
                acc = acc * fac[a]
            for e in range(a + 1, b):
                t = vals[int(idx[e])]
                if fac[e] != 1.0:
                    t = t * fac[e]
                acc = acc + t
        elif o == OP_PROD:
            acc = vals[int(idx[a])]
            if fac[a] != 1.0:
                acc = acc * fac[a]
            for e in range(a + 1, b):
                acc = acc * vals[int(idx[e])]
                if fac[e] != 1.0:
                    acc = acc * fac[e]
        else:
            raise NotImplementedError("operator")
        vals[L + n] = acc
    R = int(table.root_slot.shape[0])
    out = np.zeros((leaf.shape[0], R), dtype=np.float64)
    for k in range(R):
        s = int(table.root_slot[k])
        if s != NO_ROOT:
            out[:, k] = vals[s]
    return out


# --------------------------------------------------------------------------- #
# Element types other than Float64.  The function Compilers.compile returns is generic (src/backend/static.jl:98-133: the
# generated text carries no type), so on Vector{T} arguments it computes what Julia's promotion rules make of that text:
#   * a factor is printed as a Float64 literal, so `g * f` of a Float32 (ComplexF32) g is a Float64 (ComplexF64);
#   * `*` / `+` of two values promote to the wider of their types (conversion Float32 -> Float64 is exact);
#   * Complex * Complex = (ar br - ai bi, ar bi + ai br), Complex * Real scales both parts, Complex + Complex is
#     componentwise (base/complex.jl); n-ary `+` and `*` are left folds (base/operators.jl);
#   * `x^2`, `x^3` are x*x and x*x*x (Base.literal_pow); other exponents are out of this twin's scope;
#   * `root[k] = g` converts to eltype(root) (round to nearest).
# A value is a pair (re, im) of real numpy arrays, im None for real types; every real operation is one numpy ufunc call in the
# operands' common precision (float32 ufuncs round once to single), so nothing is contracted.  PARITY UNPINNED beyond the
# reference's all-ones known answers (which are exact in every type): Julia cannot run here.
_REAL = {"Float32": np.float32, "Float64": np.float64, "ComplexF32": np.float32, "ComplexF64": np.float64}


def _prom(a, b):
    if a.dtype == b.dtype:
        return a, b
    return a.astype(np.float64), b.astype(np.float64)


def _t_mul(x, y):
    (ar, ai), (br, bi) = x, y
    if ai is None and bi is None:
        a, b = _prom(ar, br)
        return (a * b, None)
    if bi is None:                                   # Complex * Real
        r1, b1 = _prom(ar, br)
        i1, b2 = _prom(ai, br)
        return (r1 * b1, i1 * b2)
    if ai is None:                                   # Real * Complex
        a1, r1 = _prom(ar, br)
        a2, i1 = _prom(ar, bi)
        return (a1 * r1, a2 * i1)
    ar2, br2 = _prom(ar, br)
    ai2, bi2 = _prom(ai, bi)
    return (ar2 * br2 - ai2 * bi2, ar2 * bi2 + ai2 * br2)


def _t_mulc(x, f):
    f64 = np.float64(f)
    return tuple(None if c is None else c.astype(np.float64) * f64 for c in x)


def _t_add(x, y):
    (ar, ai), (br, bi) = x, y
    a, b = _prom(ar, br)
    if ai is None and bi is None:
        return (a + b, None)
    if ai is None or bi is None:
        raise NotImplementedError("Real + Complex does not occur: all leaves have one element type")
    c, d = _prom(ai, bi)
    return (a + b, c + d)


def eval_static_typed(table, leaf: np.ndarray, dtype: str) -> np.ndarray:
    """``leaf``: [B, L] of the numpy type of ``dtype`` ("Float32", "ComplexF64", "ComplexF32", "Float64"); returns [B, R] of it."""
    real = _REAL[dtype]
    cx = dtype.startswith("Complex")
    leaf = np.asarray(leaf)
    if leaf.ndim == 1:
        leaf = leaf[None, :]
    L, N = int(table.n_leaf), int(table.op.shape[0])
    if cx:
        vals = [(np.ascontiguousarray(leaf[:, i].real).astype(real), np.ascontiguousarray(leaf[:, i].imag).astype(real)) for i in range(L)]
    else:
        vals = [(np.ascontiguousarray(leaf[:, i]).astype(real), None) for i in range(L)]
    vals += [None] * N
    off, idx, fac = table.child_off, table.child_idx, table.child_fac
    for n in range(N):
        a, b = int(off[n]), int(off[n + 1])
        o = int(table.op[n])
        term = lambda e: vals[int(idx[e])] if fac[e] == 1.0 else _t_mulc(vals[int(idx[e])], fac[e])
        if o == OP_POWER:
            x, pw = vals[int(idx[a])], int(table.power[n])
            if pw == 2:
                acc = _t_mul(x, x)
            elif pw == 3:
                acc = _t_mul(_t_mul(x, x), x)
            else:
                raise NotImplementedError("typed twin covers Power{2}, Power{3}")
            if fac[a] != 1.0:
                acc = _t_mulc(acc, fac[a])
        elif o == OP_SUM:
            acc = term(a)
            for e in range(a + 1, b):
                acc = _t_add(acc, term(e))
        elif o == OP_PROD:
            acc = term(a)
            for e in range(a + 1, b):
                acc = _t_mul(acc, vals[int(idx[e])])
                if fac[e] != 1.0:
                    acc = _t_mulc(acc, fac[e])
        else:
            raise NotImplementedError("operator")
        vals[L + n] = acc
    R = int(table.root_slot.shape[0])
    out = np.zeros((leaf.shape[0], R), dtype=leaf.dtype)
    for k in range(R):
        s = int(table.root_slot[k])
        if s != NO_ROOT:
            re, im = vals[s]
            if cx:                                   # (componentwise: `re + 1j * im` would turn an infinite part into NaN)
                out[:, k].real = re.astype(real)
                out[:, k].imag = im.astype(real)
            else:
                out[:, k] = re.astype(real)
    return out


# --------------------------------------------------------------------------- #
# Philox4x32-10 twin of the device generator (fdg_fill_uniform_device)
# --------------------------------------------------------------------------- #
def philox_uniform(B: int, L: int, seed: int, sample_offset: int = 0) -> np.ndarray:
    b = (np.arange(B, dtype=np.uint64) + np.uint64(sample_offset))[:, None]
    i = np.arange(L, dtype=np.uint64)[None, :]
    c0 = np.broadcast_to(b & np.uint64(0xFFFFFFFF), (B, L)).copy()
    c1 = np.broadcast_to(b >> np.uint64(32), (B, L)).copy()
    c2 = np.broadcast_to(i, (B, L)).copy()
    c3 = np.zeros((B, L), dtype=np.uint64)
    k0 = np.uint64(seed & 0xFFFFFFFF)
    k1 = np.uint64((seed >> 32) & 0xFFFFFFFF)
    M0, M1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ k0
        n1 = p1 & MASK
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ k1
        n3 = p0 & MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(0x9E3779B9)) & MASK
        k1 = (k1 + np.uint64(0xBB67AE85)) & MASK
    m = ((c0 >> np.uint64(5)) << np.uint64(26)) | (c1 >> np.uint64(6))
    return m.astype(np.float64) * 2.0 ** -53


# --------------------------------------------------------------------------- #
# leaf values from (K, T): numpy restatement of example/benchmark.jl:58-81,113-127
# (FrontEnds.update = one matrix product, src/frontend/pool.jl:69-76)
# --------------------------------------------------------------------------- #
def leaf_values(leaf_type, leaf_order, tau_in, tau_out, loop_index, basis, K, T, kF, beta, lam):
    """K: [B, n_loop, dim], T: [B, n_tau]; tables 1-based like FrontEnds.leafstates.  Returns [B, L]
    (entries of type-0 leaves are NaN: the reference leaves them untouched)."""
    K = np.asarray(K, dtype=np.float64)
    T = np.asarray(T, dtype=np.float64)
    basis = np.asarray(basis, dtype=np.float64)                   # [n_basis, n_loop]
    loops = np.einsum("bjd,nj->bnd", K, basis)                    # loops[:, n] = K[:, 1:n_loop] * basis[:, n]
    q2 = (loops * loops).sum(axis=2)                              # [B, n_basis]
    B, L = K.shape[0], len(leaf_type)
    out = np.full((B, L), np.nan)
    for i in range(L):
        ty = int(leaf_type[i])
        if ty == 0:
            continue
        qq = q2[:, int(loop_index[i]) - 1]
        if ty == 1:
            tau = T[:, int(tau_out[i]) - 1] - T[:, int(tau_in[i]) - 1]
            if int(leaf_order[i]) != 0:
                out[:, i] = green_derive(tau, qq - kF * kF, beta, int(leaf_order[i]))
                continue
            tau = np.where(tau == 0.0, -1e-10, tau)
            w = qq - kF * kF
            with np.errstate(over="ignore"):
                pos = np.where(w > 0, np.exp(-w * tau) / (1 + np.exp(-w * beta)), np.exp(w * (beta - tau)) / (1 + np.exp(w * beta)))
                neg = np.where(w > 0, -np.exp(-w * (tau + beta)) / (1 + np.exp(-w * beta)), -np.exp(-w * tau) / (1 + np.exp(w * beta)))
            out[:, i] = np.where(tau > 0, pos, neg)
        elif ty == 2:
            invK = 1.0 / (qq + lam)
            out[:, i] = 8 * np.pi / invK * (lam * invK) ** int(leaf_order[i])
        else:
            raise NotImplementedError(f"this leaftype {ty} not implemented!")
    return out


# Polynomials Q_k with d^k/dw^k [1/(1+exp(-b w))] = b^k Q_k(g), g = 1/(1+exp(-b w)):  Q_0 = g, Q_{k+1} = Q_k'(g) g (1-g)
_Q = [[0, 1], [0, 1, -1], [0, 1, -3, 2], [0, 1, -7, 12, -6], [0, 1, -15, 50, -60, 24], [0, 1, -31, 180, -390, 360, -120]]
_BINOM = [[1], [1, 1], [1, 2, 1], [1, 3, 3, 1], [1, 4, 6, 4, 1], [1, 5, 10, 10, 5, 1]]
_FACT = [1.0, 1.0, 2.0, 6.0, 24.0, 120.0]


def green_derive(tau, w, beta, order):
    """``green_derive`` of example/benchmark.jl:93-111 for order 1..5: (-1)^n / n! * d^n/dw^n of the fermionic
    kernel K(tau, w) = exp(-w tau) / (1 + exp(-w beta)) on 0 < tau <= beta, antiperiodic in tau (tau == 0 is taken
    as 0^- like ``green``, benchmark.jl:115-117).  The derivative itself lives in Lehmann.jl
    (``Spectral.kernelFermiT_dw*``), a dependency that is not part of the reference checkout: this restates the
    published definition in an overflow-safe form -- K = sgn A g with A = exp(w a), g = 1/(1+exp(-|w| beta)), where
    a = -tau, beta - tau, -(tau + beta), -tau on green()'s four branches; d^j A = a^j A, d^k g = b^k Q_k(g), b = +-beta
    -- and is pinned
    by 50-digit mpmath derivatives (tests/golden/green_derive.npz), not by Lehmann.jl output."""
    n = int(order)
    if not 1 <= n <= 5:
        raise NotImplementedError("not implemented!")        # benchmark.jl:108
    tau = np.asarray(tau, dtype=np.float64)
    w = np.asarray(w, dtype=np.float64)
    tau = np.where(tau == 0.0, -1e-10, tau)
    neg = tau < 0.0
    sgn = np.where(neg, -1.0, 1.0)
    pos = w >= 0.0
    # the four branches of green() (benchmark.jl:113-127), each in its overflow-safe form K = sgn * A * g
    a = np.where(pos, np.where(neg, -(tau + beta), -tau), np.where(neg, -tau, beta - tau))
    with np.errstate(over="ignore"):
        A = np.exp(w * a)
        g = 1.0 / (1.0 + np.exp(-np.abs(w) * beta))
    b = np.where(pos, beta, -beta)
    total = np.zeros_like(A)
    for k in range(n + 1):
        q = np.zeros_like(g)
        for c in reversed(_Q[k]):                 # Horner in g
            q = q * g + c
        term = _BINOM[n][k] * q
        for _ in range(n - k):
            term = term * a
        for _ in range(k):
            term = term * b
        total = total + term
    return sgn * A * total * ((-1.0) ** n / _FACT[n])


def green_derive_scale(tau, w, beta, order):
    """Magnitude of the largest Leibniz term of green_derive: the scale against which its rounding error (and
    the cancellation between terms) is to be judged."""
    n = int(order)
    tau = np.asarray(tau, dtype=np.float64)
    w = np.asarray(w, dtype=np.float64)
    tau = np.where(tau == 0.0, -1e-10, tau)
    neg, pos = tau < 0.0, w >= 0.0
    a = np.where(pos, np.where(neg, -(tau + beta), -tau), np.where(neg, -tau, beta - tau))
    with np.errstate(over="ignore"):
        return np.exp(w * a) * (np.abs(a) + beta) ** n / _FACT[n]


# --------------------------------------------------------------------------- #
# CPU baseline: the reference's C back-end text compiled by gcc
# --------------------------------------------------------------------------- #
_DRIVER = r"""
#include <pthread.h>
#include <stdint.h>
typedef struct { const double *leaf; double *root; int64_t b0, b1, L, R; } job_t;
static void *worker(void *p) {
  job_t *j = (job_t *)p;
  for (int64_t b = j->b0; b < j->b1; ++b)
    eval_graph(j->root + b * j->R, (double *)(j->leaf + b * j->L));   /* one sample per call (static.jl:100) */
  return 0;
}
int run_batch(const double *leaf, double *root, int64_t B, int64_t L, int64_t R, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  pthread_t th[256]; job_t jobs[256];
  for (int t = 0; t < nthreads; ++t) {
    jobs[t].leaf = leaf; jobs[t].root = root; jobs[t].L = L; jobs[t].R = R;
    jobs[t].b0 = B * t / nthreads; jobs[t].b1 = B * (t + 1) / nthreads;
    if (t) pthread_create(&th[t], 0, worker, &jobs[t]);
  }
  worker(&jobs[0]);
  for (int t = 1; t < nthreads; ++t) pthread_join(th[t], 0);
  return 0;
}
"""


class CBaseline:
    """gcc -O2 -ffp-contract=off build of ``void eval_graph(double *root, double
    *leafVal)`` text in the reference's to_Cstr shape, called once per sample."""

    def __init__(self, c_text: str, n_leaf: int, n_root: int, opt: str = "-O2"):
        self.n_leaf, self.n_root = n_leaf, n_root
        self._dir = tempfile.mkdtemp(prefix="fdg_cbase_")
        src = os.path.join(self._dir, "eval_graph.c")
        with open(src, "w") as f:
            f.write("#include <math.h>\n")       # compile_C header (static.jl:272-276)
            f.write(c_text)
            f.write("\n")
            f.write(_DRIVER)
        so = os.path.join(self._dir, "eval_graph.so")
        t0 = time.time()
        subprocess.check_call(["gcc", opt, "-ffp-contract=off", "-fPIC", "-shared", "-pthread", src, "-o", so, "-lm"])
        self.compile_seconds = time.time() - t0
        self._lib = C.CDLL(so)
        self._lib.run_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int]

    def __call__(self, leaf: np.ndarray, nthreads: int = 1) -> np.ndarray:
        leaf = np.ascontiguousarray(leaf, dtype=np.float64)
        B = leaf.shape[0]
        assert leaf.shape[1] == self.n_leaf
        root = np.zeros((B, self.n_root), dtype=np.float64)
        self._lib.run_batch(leaf.ctypes.data, root.ctypes.data, B, self.n_leaf, self.n_root, nthreads)
        return root
