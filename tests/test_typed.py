"""Element types other than Float64 (src/backend/static.jl:135-153 names them for to_Cstr; the function Compilers.compile
returns is generic in eltype(leafVal), static.jl:98-133).

CPU part: the typed twin of the oracle (oracle.eval_static_typed) against the reference's known answers (exact in every type),
against the Float64 oracle where the two must coincide, against the gcc-compiled ``to_Cstr(...; datatype=ComplexF64)`` text, and
on a hand-made case that separates Julia's promotion (a Float64 factor literal widens a Float32 value) from single-precision
evaluation.  GPU part: the per-type kernels of fdg_graph_specialize_typed through the C ABI, bit for bit against that twin.
Parity of the typed twin is UNPINNED beyond those answers (Julia cannot run here)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import capi, fixtures, workloads
from feynmandiagram_jl_amd.lowering import lower, table_to_Cstr
from feynmandiagram_jl_amd.nodetable import NodeTable, OP_POWER, OP_PROD, OP_SUM, from_program

NP = {"Float32": np.float32, "ComplexF64": np.complex128, "ComplexF32": np.complex64, "Float64": np.float64}


def rand_leaves(B, L, dtype, seed):
    rng = np.random.default_rng(seed)
    x = rng.random((B, L)) - 0.3
    if dtype.startswith("Complex"):
        x = x + 1j * (rng.random((B, L)) - 0.6)
    return x.astype(NP[dtype])


def same_bits(a, b):
    return a.dtype == b.dtype and a.shape == b.shape and np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))


@pytest.mark.parametrize("dtype", ["Float32", "ComplexF64", "ComplexF32"])
def test_typed_twin_meets_the_reference_known_answers(dtype):
    # test/compiler.jl:4-15: (1 + 2) * 1.5 == 4.5 in every type
    g, leaf, expect = fixtures.kat_compiler_jl()
    t, _, _ = lower([g])
    got = oracle.eval_static_typed(t, np.asarray([leaf], dtype=NP[dtype]), dtype)
    assert got.dtype == NP[dtype] and got[0, 0] == expect
    # test/computational_graph.jl:874-887: 26, 27, 702 on all-ones leaves
    graphs, exp = fixtures.kat_evaluation()
    t, _, _ = lower(list(graphs))
    got = oracle.eval_static_typed(t, np.ones((1, t.n_leaf), dtype=NP[dtype]), dtype)
    assert got[0].tolist() == [NP[dtype](e) for e in exp]


@pytest.mark.parametrize("name", ["sigma2", "parquet_sigma3", "gv_sigma4", "gv_sigma4_taylor2"])
def test_typed_twin_coincides_with_the_float64_oracle(name):
    t = workloads.get(name)
    x = rand_leaves(64, t.n_leaf, "Float64", 3)
    want = oracle.eval_static(t, x)
    assert same_bits(oracle.eval_static_typed(t, x, "Float64"), want)
    z = oracle.eval_static_typed(t, x + 0j, "ComplexF64")           # purely real complex leaves: real parts are the Float64 results
    assert np.array_equal(z.real, want) and not z.imag.any()


def test_float32_values_are_widened_by_a_factor_literal():
    """`g1 * 2.5 + g2 * g3` on Float32 leaves: the literal is a Float64 in the generated text, so the first term and the sum are
    Float64 (Julia's promotion), the product of two leaves is a Float32; the root is rounded once, at the store."""
    # leaves 0, 1, 2; value 3 = g1 * g2 (Prod); value 4 = g0 * 2.5 + g3 (Sum); root = value 4
    t = from_program(3, [(OP_PROD, 0, [(1, 1.0), (2, 1.0)]), (OP_SUM, 0, [(0, 2.5), (3, 1.0)])], [4], name="promo").normalized()
    a = np.float32(0.1) + np.arange(1000, dtype=np.float32) * np.float32(1e-3)
    b = np.float32(1.7) - np.arange(1000, dtype=np.float32) * np.float32(3e-4)
    c = np.float32(0.3) + np.arange(1000, dtype=np.float32) * np.float32(7e-4)
    leaf = np.stack([a, b, c], axis=1)
    got = oracle.eval_static_typed(t, leaf, "Float32")[:, 0]
    julia = (a.astype(np.float64) * 2.5 + (b * c).astype(np.float64)).astype(np.float32)      # Float64(a) * 2.5 + Float64(b * c), then convert
    single = a * np.float32(2.5) + b * c                                                         # what all-single arithmetic would give
    assert same_bits(got, julia)
    assert (julia != single).any()                  # the two readings differ on this input: the test separates them


_CX_DRIVER = r"""
void run_batch(double complex *leaf, double complex *root, long B, long L, long R) {
  for (long b = 0; b < B; ++b) eval_graph(root + b * R, leaf + b * L);
}
"""


@pytest.mark.parametrize("name", ["sigma2", "parquet_sigma3", "gv_sigma4"])
def test_complex_twin_against_the_compiled_to_Cstr_text(name, tmp_path):
    """The reference's own C emitter with datatype = ComplexF64 (static.jl:146-147: `complex double`), compiled by gcc without
    contraction: C's complex * complex, complex * real literal and + are Julia's formulas on finite values."""
    t = workloads.get(name)
    assert not (np.asarray(t.op) == OP_POWER).any()        # (to_static's `pow(g, N)` is not complex C; none of these graphs has a Power node)
    src = tmp_path / "cx.c"
    src.write_text("#include <math.h>\n#include <complex.h>\n" + table_to_Cstr(t, ctype="double complex ") + "\n" + _CX_DRIVER)
    so = tmp_path / "cx.so"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", str(src), "-o", str(so), "-lm"])
    lib = C.CDLL(str(so))
    lib.run_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_long]
    leaf = np.ascontiguousarray(rand_leaves(200, t.n_leaf, "ComplexF64", 11))
    root = np.zeros((200, t.n_root), dtype=np.complex128)
    lib.run_batch(leaf.ctypes.data, root.ctypes.data, 200, t.n_leaf, t.n_root)
    assert same_bits(oracle.eval_static_typed(t, leaf, "ComplexF64"), root)


@pytest.mark.parametrize("name", ["sigma2", "parquet_sigma3", "gv_sigma4", "gv_sigma4_taylor2"])
def test_complex_graph_spelled_out_on_real_parts(name):
    """nodetable.complex_to_real: the Float64 graph over (re, im) pairs evaluates, through the ordinary Float64 oracle in C, to the
    bits of the typed twin in numpy -- two independent statements of Julia's Complex{Float64} arithmetic."""
    from feynmandiagram_jl_amd.nodetable import complex_to_real
    t = workloads.get(name)
    r = complex_to_real(t)
    assert r.n_leaf == 2 * t.n_leaf and r.n_root == 2 * t.n_root
    z = np.ascontiguousarray(rand_leaves(40, t.n_leaf, "ComplexF64", 21))
    got = oracle.eval_static(r, z.view(np.float64).reshape(40, -1))
    assert same_bits(np.ascontiguousarray(got).view(np.complex128).reshape(40, -1), oracle.eval_static_typed(t, z, "ComplexF64"))
    # infinite, huge, zero and NaN parts: the same parts become NaN / inf / -0.0 in both statements
    zs = z[:8].copy()
    zs[0, 0] = complex(np.inf, 1.0); zs[1, -1] = complex(2.0, -np.inf); zs[2, :] = 1e200 + 1e200j; zs[3, 0] = complex(np.nan, 0.0)
    zs[4, :] = 0.0; zs[5, ::2] = complex(-0.0, 0.0); zs[6, :] = 1e-320 + 3e-310j
    with np.errstate(all="ignore"):
        a = np.ascontiguousarray(oracle.eval_static(r, zs.view(np.float64).reshape(8, -1)))
        b = np.ascontiguousarray(oracle.eval_static_typed(t, zs, "ComplexF64")).view(np.float64).reshape(a.shape)
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(a)])
    assert np.array_equal(np.signbit(a[~np.isnan(a)]), np.signbit(b[~np.isnan(a)]))
    with pytest.raises(NotImplementedError):
        complex_to_real(from_program(1, [(OP_POWER, 5, [(0, 1.0)])], [1]).normalized())
    # the library's own construction (fdg_graph_create_complex_view, C++): same size, and its allocated program replays to the same bits
    from test_next_rows import replay
    v = capi.GraphHandle(t).complex_view()
    q = v.info()
    assert (q["n_leaf"], q["n_node"], q["n_root"], q["n_edge"]) == (r.n_leaf, r.n_node, r.n_root, r.n_edge)
    ops, nr, nl, nm = v.opt_program(n_reg=120, n_lds=80, n_acc=124)
    got = replay(ops, nr, nl, nm, v.last_n_acc, z.view(np.float64).reshape(40, -1), r.n_root)
    assert same_bits(np.ascontiguousarray(got).view(np.complex128).reshape(40, -1), oracle.eval_static_typed(t, z, "ComplexF64"))
    with pytest.raises(capi.FdgError):
        capi.GraphHandle(from_program(1, [(OP_POWER, 5, [(0, 1.0)])], [1]).normalized()).complex_view()


def cancelling_product_case():
    """(g * -1.0) * h with g_r h_r == g_i h_i: Julia's real part is (-g_r) h_r - (-g_i) h_i = +0.0; pulling the sign out, -(g_r h_r - g_i h_i),
    would give -0.0.  Second root: h * (g * -1.0), third: the Sum g * -1.0 + g (imaginary and real parts cancel to +0.0)."""
    t = from_program(2, [(OP_PROD, 0, [(0, -1.0), (1, 1.0)]), (OP_PROD, 0, [(1, 1.0), (0, -1.0)]), (OP_SUM, 0, [(0, -1.0), (0, 1.0)])], [2, 3, 4], name="cancel").normalized()
    z = np.array([[1 + 1j, 1 + 1j], [2 + 3j, 3 + 2j], [0.5 - 0.25j, -1 - 2j], [1.5 + 0j, 0 + 2j]], dtype=np.complex128)
    return t, z


def test_sign_of_a_cancelling_real_part_in_the_twin():
    t, z = cancelling_product_case()
    want = oracle.eval_static_typed(t, z, "ComplexF64")
    assert want[0, 0].real == 0.0 and not np.signbit(want[0, 0].real) and want[0, 0].imag == -2.0
    assert not np.signbit(want[1, 0].real) and want[1, 0].real == 0.0
    from feynmandiagram_jl_amd.nodetable import complex_to_real
    got = oracle.eval_static(complex_to_real(t), z.view(np.float64).reshape(4, -1))
    assert same_bits(np.ascontiguousarray(got).view(np.complex128).reshape(4, -1), want)


def test_typed_kernels_compile_for_gfx950_and_refuse_what_they_do_not_cover(libfdg, tmp_path):
    # hiprtc cross-compiles without a device; no evaluation here
    t = workloads.get("parquet_sigma3")
    g = capi.GraphHandle(t)
    for dt in (capi.FDG_DT_F32, capi.FDG_DT_C64, capi.FDG_DT_C32):
        g.specialize_typed(dt, str(tmp_path), capi.FDG_SPEC_KEEP_SOURCE)
    texts = [open(os.path.join(tmp_path, f)).read() for f in os.listdir(tmp_path) if f.endswith(".hip")]
    assert len(texts) == 3 and all("fdg_spec_typed" in x and "const auto v" in x for x in texts)
    assert sum("fdg_cx<float> *__restrict__ leaf" in x for x in texts) == 1 and sum("const float *__restrict__ leaf" in x for x in texts) == 1
    # a single-precision schedule keeps `* -1.0` as a multiplication (it widens the value); a ComplexF64 one carries it as a sign, which a
    # later product consumes on the operand ("(-g3) * v17") instead of pulling it out
    f32 = next(x for x in texts if "const float *__restrict__ leaf" in x)
    c64 = next(x for x in texts if "fdg_cx<double> *__restrict__ leaf" in x)
    assert "* -0x1p+0;" in f32 and "* -0x1p+0;" not in c64 and " = (-" in c64
    # Power{5}: other literal powers take type-specific paths in Julia; the typed kernels say so instead of guessing
    tp = from_program(1, [(OP_POWER, 5, [(0, 1.0)])], [1], name="p5").normalized()
    with pytest.raises(capi.FdgError) as e:
        capi.GraphHandle(tp).specialize_typed(capi.FDG_DT_F32, str(tmp_path))
    assert e.value.code == capi.FDG_E_UNSUPPORTED
    with pytest.raises(capi.FdgError):
        g.specialize_typed(7, str(tmp_path))


GOLD = os.path.join(os.path.dirname(__file__), "golden", "typed_vectors.npz")


@pytest.mark.parametrize("dtype", ["Float32", "ComplexF64", "ComplexF32"])
@pytest.mark.parametrize("name", ["sigma2", "parquet_sigma3", "gv_sigma4"])
def test_typed_twin_reproduces_the_committed_vectors(name, dtype):
    z = np.load(GOLD)
    leaf, want = z[f"{name}:{dtype}:leaf"], z[f"{name}:{dtype}:root"]
    assert leaf.dtype == NP[dtype] and same_bits(oracle.eval_static_typed(workloads.get(name), leaf, dtype), want)


# ---------------------------------------------------------------------------------------------------------- GPU


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["Float32", "ComplexF64", "ComplexF32"])
@pytest.mark.parametrize("name", ["sigma2", "parquet_sigma3", "gv_sigma4"])
def test_typed_committed_vectors_on_device(libfdg, cuda, name, dtype):
    import torch
    z = np.load(GOLD)
    leaf, want = z[f"{name}:{dtype}:leaf"], z[f"{name}:{dtype}:root"]
    f = fd.compile_table(workloads.get(name), specialize="isa")
    for dev_leaf in (torch.from_numpy(leaf).to(cuda), dev_typed(cuda, leaf, "leaf_major")):      # rows (ComplexF64: the spelled-out graph) and columns
        got = f(None, dev_leaf)
        torch.cuda.synchronize()
        assert same_bits(got.cpu().numpy(), want)



def dev_typed(cuda, x, layout):
    import torch
    h = torch.from_numpy(np.ascontiguousarray(x))
    if layout == "leaf_major":           # a Julia column-major B x L matrix
        d = torch.empty((x.shape[1], x.shape[0]), dtype=h.dtype, device=cuda).t()
    elif layout == "padded":
        d = torch.empty((x.shape[0], x.shape[1] + 3), dtype=h.dtype, device=cuda)[:, :x.shape[1]]
    else:
        d = torch.empty(x.shape, dtype=h.dtype, device=cuda)
    d.copy_(h)
    return d


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["sample_major", "leaf_major", "padded"])
@pytest.mark.parametrize("dtype", ["Float32", "ComplexF64", "ComplexF32"])
@pytest.mark.parametrize("name", ["sigma2", "parquet_sigma3", "gv_sigma4", "synthetic_small", "parquet_sigma4", "gv_sigma4_taylor2"])
def test_typed_kernels_match_the_typed_twin_bitwise(libfdg, cuda, name, dtype, layout):
    import torch
    t = workloads.get(name)
    f = fd.compile_table(t, specialize="isa")            # the handle's Float64 kernels stay what they are; the typed one is added on first use
    x = rand_leaves(4099, t.n_leaf, dtype, 5)
    leaf = dev_typed(cuda, x, layout)
    root = f(None, leaf)
    torch.cuda.synchronize()
    if dtype == "ComplexF64" and layout != "leaf_major" and name in ("gv_sigma4", "parquet_sigma4"):
        # rows of re, im pairs: the graph spelled out on real and imaginary parts, the Float64 row-major assembly kernel (graphs whose
        # spelled-out form gets no such variant -- tiny ones, the Taylor graph -- stay with the per-type kernel)
        assert f.last_typed_kernel.startswith("fdg_isa_eval_rm") and "ComplexF64 rows" in f.last_typed_kernel
    elif dtype != "ComplexF64" or layout == "leaf_major":
        assert f.last_typed_kernel.startswith("fdg_spec_typed<" + dtype)
    got = root.cpu().numpy()
    want = oracle.eval_static_typed(t, x, dtype)
    assert same_bits(got, want), np.abs(got - want).max()
    # the Float64 path of the same handle is untouched
    x64 = rand_leaves(513, t.n_leaf, "Float64", 6)
    r64 = f(None, torch.from_numpy(x64).to(cuda))
    torch.cuda.synchronize()
    assert same_bits(r64.cpu().numpy(), oracle.eval_static(t, x64))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["ComplexF64", "ComplexF32"])
def test_sign_of_a_cancelling_real_part_on_device(libfdg, cuda, dtype):
    import torch
    t, z = cancelling_product_case()
    z = z.astype(NP[dtype])
    want = oracle.eval_static_typed(t, z, dtype)
    f = fd.compile_table(t, specialize="isa")
    for leaf in (torch.from_numpy(z).to(cuda), dev_typed(cuda, z, "leaf_major")):
        got = f(None, leaf)
        torch.cuda.synchronize()
        assert same_bits(got.cpu().numpy(), want)          # bit for bit: the signs of the zeros included


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["Float32", "ComplexF64", "ComplexF32"])
def test_typed_known_answers_through_the_device(libfdg, cuda, dtype):
    import torch
    tdt = {"Float32": torch.float32, "ComplexF64": torch.complex128, "ComplexF32": torch.complex64}[dtype]
    g, leaf, expect = fixtures.kat_compiler_jl()
    f, _ = fd.Compilers.compile([g], specialize="isa")
    root = torch.zeros(1, dtype=tdt, device=cuda)
    f(root, torch.tensor(leaf, dtype=tdt, device=cuda))
    torch.cuda.synchronize()
    assert root.cpu().numpy()[0] == expect
    graphs, exp = fixtures.kat_evaluation()
    f, lm = fd.Compilers.compile(list(graphs), specialize="isa")
    root = torch.zeros(3, dtype=tdt, device=cuda)
    f(root, torch.ones(len(lm), dtype=tdt, device=cuda))
    torch.cuda.synchronize()
    assert root.cpu().numpy().tolist() == [NP[dtype](e) for e in exp]
    with pytest.raises(IndexError):
        f(root, torch.ones(1, dtype=tdt, device=cuda))
    with pytest.raises(TypeError):
        f(None, torch.ones(len(lm), dtype=torch.int64, device=cuda))
    # host vectors of the type: the generated function's call convention (root mutated in place, last root returned)
    g, leaf, expect = fixtures.kat_compiler_jl()
    f, _ = fd.Compilers.compile([g], specialize="isa")
    hroot = np.zeros(1, dtype=NP[dtype])
    ret = f(hroot, np.asarray(leaf, dtype=NP[dtype]))
    assert hroot[0] == expect and ret == expect and hroot.dtype == NP[dtype]
    graphs, exp = fixtures.kat_evaluation()            # three roots: the host vector path fills every one of them
    f, lm = fd.Compilers.compile(list(graphs), specialize="isa")
    hroot = np.zeros(3, dtype=NP[dtype])
    ret = f(hroot, np.ones(len(lm), dtype=NP[dtype]))
    assert hroot.tolist() == [NP[dtype](e) for e in exp] and ret == exp[-1]
