"""Host-side mirror of the reference interface (graph model, lowering order,
text emitters, Compilers API) and the C-ABI library surface.  CPU only: no
compute call is made through the library here."""
import ctypes
import os
import re

import numpy as np
import pytest

import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import Compilers, capi, fixtures, workloads
from feynmandiagram_jl_amd.graph import Graph, Power, Prod, Sum, Unitary, PostOrderDFS, reset_uid
from feynmandiagram_jl_amd.lowering import lower
from feynmandiagram_jl_amd.nodetable import OP_POWER, OP_PROD, OP_SUM, from_program

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- graph model ---------------------------------------------------------- #
def test_factor_becomes_wrapping_prod():
    # graph.jl:69-73
    g = Graph.new([], factor=2)
    assert isinstance(g.operator, Prod) and len(g.subgraphs) == 1 and g.subgraph_factors == [2.0]
    assert Graph.new([], factor=1.0).subgraphs == []
    with pytest.raises(ValueError):
        Graph([], factor=2.0)


def test_scalar_mul_merges_trivial_unary():
    # graph.jl:136-144
    g1 = Graph([])
    g2 = Graph.new([], factor=2)
    h = 5 * g2
    assert h.subgraphs[0] is g2.subgraphs[0] and h.subgraph_factors == [10.0]
    h2 = (3 * g1) * 4
    assert h2.subgraphs[0] is g1 and h2.subgraph_factors == [12.0]


def test_linear_combination_rules():
    # graph.jl:178-207, 228-262
    g1, g2 = Graph([]), Graph([])
    s = g1 + g1
    assert len(s.subgraphs) == 1 and s.subgraph_factors == [2.0] and isinstance(s.operator, Sum)
    d = g1 - g2
    assert d.subgraph_factors == [1.0, -1.0]
    lc = fd.linear_combination([g1, g2, g1], [1, 2, 3])
    assert [x.id for x in lc.subgraphs] == [g1.id, g2.id] and lc.subgraph_factors == [4.0, 2.0]


def test_multi_product_rules():
    # graph.jl:304-331, 350-401
    g1, g2 = Graph([]), Graph([])
    p = g1 * g2
    assert isinstance(p.operator, Prod) and p.subgraph_factors == [1.0, 1.0]
    sq = fd.multi_product(g1, g1, 2.0, 3.0)
    assert isinstance(sq.operator, Power) and sq.operator.N == 2 and sq.subgraph_factors == [6.0]
    mp = fd.multi_product([g1, g2, g1], [1, 2, 3])
    assert isinstance(mp.operator, Prod) and len(mp.subgraphs) == 2
    assert isinstance(mp.subgraphs[0].operator, Power) and mp.subgraph_factors == [3.0, 2.0]


def test_power_and_unitary_assertions():
    with pytest.raises(AssertionError):
        Power(1)
    with pytest.raises(AssertionError):
        Power(0)
    with pytest.raises(AssertionError):
        Graph([Graph([]), Graph([])], operator=Power(2))
    with pytest.raises(AssertionError):
        Graph([Graph([])], operator=Unitary())


def test_postorder_is_tree_expansion():
    a, b = Graph([]), Graph([])
    s = Graph([a, b], operator=Sum())
    r = Graph([s, s, a], operator=Prod())
    assert [g.id for g in PostOrderDFS(r)] == [a.id, b.id, s.id, a.id, b.id, s.id, a.id, r.id]


# ---- lowering / emitters --------------------------------------------------- #
def test_sigma2_text_equals_reference_program():
    gs, _ = fixtures.sigma2_graphs()
    s, leafmap = Compilers.to_julia_str(gs)
    lines = s.split("\n")
    assert lines[1] == "function eval_graph!(root::AbstractVector, leafVal::AbstractVector)" and lines[-1] == "end"
    assert "\n".join(lines[2:-1]) + "\n" == fixtures.SIGMA2_JULIA_BODY
    assert [leafmap[i].id for i in range(1, 9)] == [18636, 18643, 18637, 18650, 18630, 18676, 18708, 18709]
    assert [leafmap[i].name for i in range(1, 9)] == list("GGVGVVGG")     # G1 G2 V3 G4 V5 V6 G7 G8 (to_dot labels)


def test_leaf_numbering_and_statement_interleaving():
    a, b, c = Graph([], _id=101), Graph([], _id=102), Graph([], _id=103)
    n0 = Graph([b, c], operator=Sum(), _id=201)
    r = Graph([a, n0, a], operator=Prod(), subgraph_factors=[1, 2, 1], _id=301)
    s, lm = Compilers.to_julia_str([r])
    assert s == ("\nfunction eval_graph!(root::AbstractVector, leafVal::AbstractVector)\n"
                 "    g101 = leafVal[1]\n    g102 = leafVal[2]\n    g103 = leafVal[3]\n"
                 "    g201 = (g102 + g103)\n    g301 = (g101 * g201 * 2.0 * g101)\n    root[1] = g301\nend")
    cs, lm2 = Compilers.to_Cstr([r])
    assert cs == ("\nvoid eval_graph(double *root, double *leafVal)\n{\n"
                  "    double  g101, g102, g103, g201, g301;\n"
                  "    g101 = leafVal[0];\n    g102 = leafVal[1];\n    g103 = leafVal[2];\n"
                  "    g201 = (g102 + g103);\n    g301 = (g101 * g201 * 2.0 * g101);\n    root[0] = g301;\n}")
    ps, _ = Compilers.to_python_str([r])
    assert ps == ("import torch\ndef eval_graph(leafVal):\n"
                  "    root = torch.empty(leafVal.shape[0], 1, dtype=leafVal.dtype, device=leafVal.device)\n"
                  "    g101 = leafVal[:, 0]\n    g102 = leafVal[:, 1]\n    g103 = leafVal[:, 2]\n"
                  "    g201 = (g102 + g103)\n    g301 = (g101 * g201 * 2.0 * g101)\n    root[:, 0] = g301\n"
                  "    return root\n\n")
    assert sorted(lm) == [1, 2, 3] and lm[1] is a and lm2[3] is c


def test_power_text_forms():
    a = Graph([], _id=1)
    p = Graph([a], operator=Power(3), subgraph_factors=[2.0], _id=2)
    assert "g2 = ((g1)^3 * 2.0)" in Compilers.to_julia_str([p])[0]          # static.jl:45
    assert "g2 = pow(g1, 3) * 2.0;" in Compilers.to_Cstr([p])[0]            # static.jl:38-39
    assert "g2 = ((g1)**3 * 2.0)" in Compilers.to_python_str([p])[0]


def test_unknown_operator_is_rejected_at_lowering():
    class Weird(fd.ComputationalGraphs.AbstractOperator):
        pass
    a = Graph([])
    g = Graph([a], operator=Weird())
    with pytest.raises(NotImplementedError, match="not yet implemented"):     # static.jl:6-11
        lower([g])
    # a Unitary node has no children, so it is a leaf for the compiler (graph.jl:118-125)
    c = fd.constant_graph()
    t, lm, _ = lower([Graph([c, a], operator=Sum())])
    assert t.n_leaf == 2 and t.n_node == 1


def test_shared_subgraph_emitted_once_and_dag_order():
    a, b = Graph([]), Graph([])
    s = a + b
    p1 = s * a
    p2 = s * b
    t, lm, ids = lower([p1, p2])
    assert t.n_leaf == 2 and t.n_node == 3
    assert ids[s.id] == 2 and ids[p1.id] == 3 and ids[p2.id] == 4
    assert t.root_slot.tolist() == [3, 4]


def test_compile_file_emitters_append(tmp_path):
    g, _, _ = fixtures.kat_compiler_jl()
    fn = tmp_path / "out.c"
    lm = Compilers.compile_C([g], str(fn))
    Compilers.compile_C([g], str(fn), func_name="second")
    txt = fn.read_text()
    assert txt.startswith("#include <math.h>\n") and txt.count("#include") == 1       # static.jl:272-276
    assert "void eval_graph(double *root, double *leafVal)" in txt and "void second(" in txt
    assert sorted(lm) == [1, 2]
    fj = tmp_path / "out.jl"
    Compilers.compile_Julia([g], str(fj))
    assert "function eval_graph!(root::AbstractVector, leafVal::AbstractVector)" in fj.read_text()
    fp = tmp_path / "out.py"
    Compilers.compile_Python([g], str(fp))
    ns = {}
    exec(fp.read_text(), ns)                          # the emitted torch function runs as is
    import torch
    out = ns["eval_graph"](torch.tensor([[1.0, 2.0], [3.0, 4.0]], dtype=torch.float64))
    assert out[:, 0].tolist() == [4.5, 10.5]


# ---- C ABI surface ----------------------------------------------------------- #
def test_library_exports_every_declared_symbol(libfdg):
    hdr = open(os.path.join(ROOT, "include", "fdg.h")).read()
    declared = sorted(set(re.findall(r"\b(fdg_[a-z_0-9]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(libfdg, name), f"libfdg.so does not export {name}"
    assert sorted(capi.EXPORTS) == declared
    assert libfdg.fdg_version() == 102


def test_graph_create_validation_and_info(libfdg):
    t = workloads.get("sigma2")
    h = capi.GraphHandle(t)
    info = h.info()
    assert (info["n_leaf"], info["n_node"], info["n_root"], info["n_edge"]) == (8, 18, 2, 37)
    assert info["flops_alg"] == 32 and info["bytes_alg"] == 80 and info["n_live_node"] == 18
    assert info["specialized"] == 0
    # not topologically sorted
    bad = from_program(2, [(OP_SUM, 0, [(0, 1.0), (1, 1.0)])], [2])
    bad.child_idx = np.array([0, 5], dtype=np.uint32)
    with pytest.raises(ValueError):
        capi.GraphHandle(bad)
    # bypass the python-side validate: the C side must reject too
    d = capi.GraphDesc()
    tn = bad
    d.n_leaf, d.n_node, d.n_root, d.n_edge = 2, 1, 1, 2
    arrs = [np.array([7], np.uint8), np.array([0], np.int32), np.array([0, 2], np.uint32),
            np.array([0, 1], np.uint32), np.array([1.0, 1.0]), np.array([2], np.uint32)]
    d.op = arrs[0].ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))
    d.power = arrs[1].ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    d.child_off = arrs[2].ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
    d.child_idx = arrs[3].ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
    d.child_fac = arrs[4].ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    d.root_slot = arrs[5].ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
    hp = ctypes.c_void_p()
    rc = libfdg.fdg_graph_create(ctypes.byref(d), ctypes.byref(hp))
    assert rc == capi.FDG_E_UNSUPPORTED and b"not yet implemented" in libfdg.fdg_last_error()   # static.jl:6-11
    arrs[0][0] = OP_POWER
    arrs[1][0] = 2
    rc = libfdg.fdg_graph_create(ctypes.byref(d), ctypes.byref(hp))
    assert rc == capi.FDG_E_INVALID and b"one and only one" in libfdg.fdg_last_error()            # graph.jl:61-62


def test_dead_code_is_dropped_but_roots_kept(libfdg):
    nodes = [(OP_SUM, 0, [(0, 1.0), (1, 2.0)]),      # live
             (OP_PROD, 0, [(1, 1.0), (2, 1.0)]),     # dead
             (OP_PROD, 0, [(3, 1.0), (0, -1.0)])]    # root
    t = from_program(3, nodes, [5])
    info = capi.GraphHandle(t).info()
    assert info["n_live_node"] == 2 and info["n_live_leaf"] == 2 and info["flops_alg"] == 4


def test_emit_source_and_jit_without_device(libfdg, tmp_path, monkeypatch, fdgopt):
    fdgopt.set("FDG_CACHE_RO_DIR", "")           # (without the shipped kernel_cache: the point is to compile)
    t = workloads.get("sigma2")
    h = capi.GraphHandle(t)
    src = h.emit_source()
    assert 'extern "C" __global__' in src and "fdg_spec_sm" in src and "fdg_spec_gen" in src
    assert "fma" not in src.split("fdg_block_sum")[1].split("fdg_spec_gen")[0].replace("fdg_powi", "")
    h.specialize(str(tmp_path))                          # hiprtc cross-compiles for gfx950 without a GPU
    assert h.info()["specialized"] == 1
    assert any(f.endswith(".hsaco") for f in os.listdir(tmp_path))


def test_no_cpu_fallback_without_gpu(libfdg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    f = fd.compile_table(workloads.get("sigma2"))
    with pytest.raises(capi.FdgError) as e:
        f(np.zeros(2), np.ones(8))
    assert e.value.code == capi.FDG_E_NO_DEVICE and "no CPU fallback" in str(e.value)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(capi.FdgLibraryMissing, match="no CPU fallback"):
        capi.lib()


def test_fused_step_specializes_without_device(libfdg, tmp_path):
    """fdg_graph_specialize_fused is host-only (hiprtc cross-compiles): the generated source carries the leaf
    formulas of example/benchmark.jl and the graph body; table/graph mismatches are rejected; the run entry
    points refuse a handle that was not specialised."""
    import numpy as np
    from feynmandiagram_jl_amd import workloads
    gold = os.path.join(os.path.dirname(__file__), "golden")
    z = np.load(os.path.join(gold, "gv_sigma4_leafstates.npz"))
    t = workloads.get("gv_sigma4")
    h = capi.GraphHandle(t)
    tab, keep = capi.make_leaf_tables(z["leaf_type"], z["leaf_order"], z["tau_in"], z["tau_out"], z["loop_index"], z["basis"], 3, int(z["n_tau"]))
    h.specialize_fused(tab, str(tmp_path), capi.FDG_SPEC_KEEP_SOURCE)
    files = os.listdir(tmp_path)
    assert any(f.startswith("fdg_fused_") and f.endswith(".hsaco") for f in files)
    src = open(os.path.join(tmp_path, [f for f in files if f.endswith(".hip")][0])).read()
    assert "fdg_spec_fused" in src and src.count("exp(") >= 89 and "den = 1.0 + exp(-fabs(w) * beta)" in src
    assert "const double r0 = " in src and "const double v" in src      # graph body (scheduled order) follows the leaves
    small = capi.GraphHandle(workloads.get("sigma2"))
    with pytest.raises(capi.FdgError) as e:
        small.specialize_fused(tab, str(tmp_path))
    assert e.value.code == capi.FDG_E_INVALID
    bad = z["leaf_type"].copy()
    bad[0] = 7
    tab2, keep2 = capi.make_leaf_tables(bad, z["leaf_order"], z["tau_in"], z["tau_out"], z["loop_index"], z["basis"], 3, int(z["n_tau"]))
    with pytest.raises(capi.FdgError) as e:
        h.specialize_fused(tab2, str(tmp_path))
    assert e.value.code == capi.FDG_E_UNSUPPORTED                     # "this leaftype ... not implemented!" (benchmark.jl:79)


def test_one_kernel_isa_step_assembles_without_device(libfdg, tmp_path, monkeypatch, no_shipped_cache, fdgopt):
    """Route 3 of fdg_graph_specialize_fused (handle specialised with FDG_SPEC_ISA): the kernels -- eval and accumulate --
    are assembled right away, host-only, and do not depend on kF, beta, lambda (kernel arguments); the assembly carries
    the exp / reciprocal sequences (v_rndne_f64, v_ldexp_f64, v_rcp_f64 followed by the wait state gfx950 needs) and
    loads only the n_loop*dim + n_tau input columns."""
    import numpy as np
    import re
    import feynmandiagram_jl_amd as fd
    from feynmandiagram_jl_amd import workloads
    gold = os.path.join(os.path.dirname(__file__), "golden")
    z = np.load(os.path.join(gold, "gv_sigma4_leafstates.npz"))
    t = workloads.get("gv_sigma4")
    f = fd.compile_table(t, specialize="isa")
    tab, keep = capi.make_leaf_tables(z["leaf_type"], z["leaf_order"], z["tau_in"], z["tau_out"], z["loop_index"], z["basis"], 3, int(z["n_tau"]),
                                      1.919, 3.0, 1.2)
    f.handle.specialize_fused(tab, str(tmp_path), capi.FDG_SPEC_KEEP_SOURCE)
    asm = [x for x in os.listdir(tmp_path) if x.startswith("fdg_isa_") and x.endswith(".s")]
    assert len(asm) == 1 and os.path.exists(os.path.join(tmp_path, asm[0][:-2] + ".hsaco"))
    tab_b, keep_b = capi.make_leaf_tables(z["leaf_type"], z["leaf_order"], z["tau_in"], z["tau_out"], z["loop_index"], z["basis"], 3, int(z["n_tau"]),
                                          1.0, 40.0, 0.3)
    fd.compile_table(t, specialize="isa").handle.specialize_fused(tab_b, str(tmp_path), capi.FDG_SPEC_KEEP_SOURCE)
    assert [x for x in os.listdir(tmp_path) if x.startswith("fdg_isa_") and x.endswith(".s")] == asm      # other parameters, same code object
    src = open(os.path.join(tmp_path, asm[0])).read()
    assert "fdg_isa_mc:" in src and "fdg_isa_mc_acc:" in src
    body = src.split("fdg_isa_mc_acc:")[0]
    n_exp = body.count("v_rndne_f64")
    assert n_exp >= 89 and body.count("v_ldexp_f64") >= n_exp          # one exponential per fermionic leaf + one per momentum (v_ldexp_f64 also carries the factors +-2^k)
    assert body.count("v_div_fixup_f64") == body.count("v_rcp_f64") > 0                # quotients: correctly rounded divisions
    assert capi.isa_check_hazards(src)[0] == 0
    assert body.count("global_load_dwordx2") <= 3 * (int(z["basis"].shape[1]) * 3 + int(z["n_tau"]))   # inputs (a few re-loads), no leaf matrix
    # FDG_MC_ROUTE=isa insists on this route and says why it cannot be taken
    order6 = z["leaf_order"].copy()
    order6[np.nonzero(z["leaf_type"] == 1)[0][0]] = 6         # green_derive beyond order 5: example/benchmark.jl:108 "not implemented!"
    tab0, keep0 = capi.make_leaf_tables(z["leaf_type"], order6, z["tau_in"], z["tau_out"], z["loop_index"], z["basis"], 3, int(z["n_tau"]))
    fdgopt.set("FDG_MC_ROUTE", "isa")
    with pytest.raises(capi.FdgError, match="not implemented"):          # the tables are refused as the reference's green_derive refuses them
        fd.compile_table(t, specialize="isa").handle.specialize_fused(tab0, str(tmp_path))
    # interaction counter-terms of any order are covered (pow_body above x^3), except by the compiler-scheduled fused kernel
    order4 = z["leaf_order"].copy()
    order4[np.nonzero(z["leaf_type"] == 2)[0][0]] = 4
    tab1, keep1 = capi.make_leaf_tables(z["leaf_type"], order4, z["tau_in"], z["tau_out"], z["loop_index"], z["basis"], 3, int(z["n_tau"]))
    fd.compile_table(t, specialize="isa").handle.specialize_fused(tab1, str(tmp_path))
    fdgopt.set("FDG_MC_ROUTE", "fused")
    with pytest.raises(capi.FdgError, match="order > 3"):
        fd.compile_table(t, specialize="isa").handle.specialize_fused(tab1, str(tmp_path))
    fdgopt.unset("FDG_MC_ROUTE")
    fd.compile_table(t, specialize="isa").handle.specialize_fused(tab1, str(tmp_path))


def test_argument_checks_of_the_newer_entry_points(libfdg):
    """Bad arguments are rejected before any device or RCCL work (so this runs without a GPU)."""
    import ctypes as C
    L = capi.lib()
    out = C.c_void_p()
    buf = C.create_string_buffer(capi.COMM_ID_BYTES)
    assert L.fdg_comm_create(None, 0, 1, C.byref(out)) == capi.FDG_E_INVALID
    assert L.fdg_comm_create(buf, 2, 2, C.byref(out)) == capi.FDG_E_INVALID          # rank out of range
    assert L.fdg_comm_create(buf, 0, 0, C.byref(out)) == capi.FDG_E_INVALID
    assert L.fdg_comm_unique_id(buf, 16) == capi.FDG_E_INVALID                       # buffer too small
    assert L.fdg_comm_destroy(None) == 0
    h = capi.GraphHandle(workloads.get("sigma2"))
    with pytest.raises(capi.FdgError) as e:                                          # companion without an ISA kernel
        h.specialize(None, capi.FDG_SPEC_ROW_MAJOR_COMPANION)
    assert e.value.code == capi.FDG_E_INVALID
    with pytest.raises(capi.FdgError) as e:                                          # fused step never specialised
        h.mc_eval_device(8, 1, 1, 8, 1, 1, 1.0, 1.0, 1.0, 8, 1, 1, 4)
    assert e.value.code == capi.FDG_E_INVALID
    with pytest.raises(capi.FdgError):
        h.mc_accumulate_device(8, 1, 1, 8, 1, 1, 1.0, 1.0, 1.0, 0, 0, 4)            # null accumulator


def test_package_tables_equal_golden_fixtures():
    """The node tables the product ships (feynmandiagram.jl_amd/data/, workloads.get / workloads.leafstates) are the
    tables of the golden fixtures, without their vectors: tests/golden/make_package_data.py keeps them in step."""
    from feynmandiagram_jl_amd import workloads
    from feynmandiagram_jl_amd.nodetable import NodeTable
    gold = os.path.join(os.path.dirname(__file__), "golden")
    for name in ("gv_sigma4", "gv_sigma5", "gv_sigma6", "gv_sigma4_taylor2", "gv_sigma5_taylor2"):
        a, b = workloads.get(name), NodeTable.load(os.path.join(gold, name + ".npz"))
        for k in ("op", "power", "child_off", "child_idx", "child_fac", "root_slot"):
            assert np.array_equal(getattr(a, k), getattr(b, k)), (name, k)
        assert a.n_leaf == b.n_leaf and (a.sched_group is None) == (b.sched_group is None)
        if a.sched_group is not None:
            assert np.array_equal(a.sched_group, b.sched_group)
    for name in ("gv_sigma4", "gv_sigma5"):
        z, g = workloads.leafstates(name), np.load(os.path.join(gold, name + "_leafstates.npz"))
        for k in g.files:
            assert np.array_equal(z[k], g[k]), (name, k)
    assert not os.path.commonpath([workloads.DATA, gold]) == gold


def test_cache_directory_is_vetted(tmp_path, monkeypatch, fdgopt):
    """JIT-ed code objects are read back by predictable name, so the cache directory must belong to the caller and must
    not be writable by anybody else; a world-writable artefact inside it is ignored and rebuilt; no shell sees the path."""
    import stat
    from feynmandiagram_jl_amd import capi, workloads
    capi.lib()
    fdgopt.set("FDG_CACHE_RO_DIR", "")
    t = workloads.get("sigma2")
    good = tmp_path / "cache"
    good.mkdir(mode=0o700)
    h = capi.GraphHandle(t)
    h.specialize(str(good), capi.FDG_SPEC_ISA)
    objs = [p for p in good.iterdir() if p.suffix == ".hsaco"]
    assert len(objs) == 1 and not [p for p in good.iterdir() if ".tmp." in p.name]
    blob = objs[0].read_bytes()
    # a planted (world-writable) file of the same name is not trusted: the kernel is assembled again
    objs[0].write_bytes(b"not a code object")
    os.chmod(objs[0], 0o666)
    capi.GraphHandle(t).specialize(str(good), capi.FDG_SPEC_ISA)
    assert objs[0].read_bytes() == blob and not (objs[0].stat().st_mode & stat.S_IWOTH)
    bad = tmp_path / "open"
    bad.mkdir()
    os.chmod(bad, 0o777)
    with pytest.raises(capi.FdgError) as e:
        capi.GraphHandle(t).specialize(str(bad), capi.FDG_SPEC_ISA)
    assert e.value.code == capi.FDG_E_JIT and "writable" in str(e.value)
    quoted = tmp_path / "it's"
    quoted.mkdir(mode=0o700)
    with pytest.raises(capi.FdgError) as e:
        capi.GraphHandle(t).specialize(str(quoted), capi.FDG_SPEC_ISA)
    assert e.value.code == capi.FDG_E_INVALID


def test_artefacts_pass_the_vetting_under_any_umask_and_shipped_cache_is_read_only(tmp_path, monkeypatch, fdgopt):
    """ The linker creates its output under the caller's umask -- 0775 under umask 002 --, which the vetting of
    the cache used to refuse, so every specialisation failed after a successful assembly.  The artefact is chmod'ed before
    it is renamed into place, and reads only insist on "owned by the caller or root, not world-writable".  The kernel_cache
    shipped inside the package is a read-only secondary lookup ($FDG_CACHE_RO_DIR): a hit there writes nothing anywhere;
    the default (per-user) directory is used when no directory is passed, so a root-owned installation works."""
    import stat
    from feynmandiagram_jl_amd import capi, workloads
    capi.lib()
    fdgopt.set("FDG_CACHE_RO_DIR", "")
    t = workloads.get("sigma2")
    old = os.umask(0o002)
    try:
        a = tmp_path / "a"
        a.mkdir(mode=0o700)
        capi.GraphHandle(t).specialize(str(a), capi.FDG_SPEC_ISA)
        objs = [p for p in a.iterdir() if p.suffix == ".hsaco"]
        assert len(objs) == 1 and stat.S_IMODE(objs[0].stat().st_mode) == 0o644
        capi.GraphHandle(t).specialize(str(a), capi.FDG_SPEC_ISA)            # read back, not an error
        os.chmod(objs[0], 0o664)                                             # group-writable artefact in a vetted directory: accepted
        m0 = objs[0].stat().st_mtime_ns
        capi.GraphHandle(t).specialize(str(a), capi.FDG_SPEC_ISA)
        assert objs[0].stat().st_mtime_ns == m0
    finally:
        os.umask(old)
    # the shipped cache as a read-only secondary: nothing is written into the primary on a hit, nothing ever into the secondary.
    #  What is found THROUGH $FDG_CACHE_RO_DIR must be writable by its owner only -- nothing legitimate is written there under
    # the caller's umask --: the group-writable artefact is not taken (the kernel is assembled again into the primary), the 0644 one is.
    fdgopt.set("FDG_CACHE_RO_DIR", "/nonexistent:" + str(a))
    os.chmod(a, 0o555)
    c = tmp_path / "c"
    c.mkdir(mode=0o700)
    capi.GraphHandle(t).specialize(str(c), capi.FDG_SPEC_ISA)
    assert [p for p in c.iterdir() if p.suffix == ".hsaco"]
    os.chmod(objs[0], 0o644)
    b = tmp_path / "b"
    b.mkdir(mode=0o700)
    h = capi.GraphHandle(t)
    h.specialize(str(b), capi.FDG_SPEC_ISA)
    assert h.info()["specialized"] == 1 and not list(b.iterdir())
    os.chmod(a, 0o775)                                                        # a group-writable read-only directory is not consulted at all
    d = tmp_path / "d"
    d.mkdir(mode=0o700)
    capi.GraphHandle(t).specialize(str(d), capi.FDG_SPEC_ISA)
    assert [p for p in d.iterdir() if p.suffix == ".hsaco"]
    os.chmod(a, 0o700)
    # no directory given: the library's per-user default, not the package directory
    monkeypatch.setenv("XDG_CACHE_HOME", str(tmp_path / "xdg"))
    (tmp_path / "xdg").mkdir(mode=0o700)
    fdgopt.set("FDG_CACHE_RO_DIR", "")
    fdgopt.unset("FDG_CACHE_DIR")
    capi.GraphHandle(t).specialize(None, capi.FDG_SPEC_ISA)
    assert [p for p in (tmp_path / "xdg" / "fdg").iterdir() if p.suffix == ".hsaco"]
    # an unusable default location (here: group-writable) does not fail the call: a private directory of the process is used
    os.chmod(tmp_path / "xdg" / "fdg", 0o770)
    for p in (tmp_path / "xdg" / "fdg").iterdir():
        p.unlink()
    capi.GraphHandle(t).specialize(None, capi.FDG_SPEC_ISA)
    assert not list((tmp_path / "xdg" / "fdg").iterdir())


def test_kernel_info_names_the_variants(libfdg):
    """fdg_graph_kernel_info: what the installed ISA programs execute per evaluation; bench.py takes the kernel name and the
    executed fold steps from here instead of guessing which variant the library launched."""
    from feynmandiagram_jl_amd import capi, workloads
    t = workloads.get("parquet_sigma4")
    f = fd.compile_table(t, specialize="isa")
    ki = f.kernel_info()
    assert ki["last_kernel"] == "" and ki["has_acc"] == 1 and ki["has_rm"] == 1 and ki["rm_bufs"] >= 2
    assert 0 < ki["n_valu"][0] <= t.stats()["flops_alg"] and ki["n_ld_leaf"][0] == t.n_leaf and ki["n_panel"] == [0, 0, 0]
    assert ki["waves_per_cu"][0] == 8 and ki["waves_per_cu"][2] >= 4
    assert fd.compile_table(t).kernel_info()["n_valu"] == [0, 0, 0]          # the interpreter: nothing installed
    small = fd.compile_table(workloads.get("sigma2"), specialize="auto")     # < 16 leaves: no row-major variant -> companion
    assert small.kernel_info()["has_rm"] == 0
    # the tiny-graph configuration (28 value registers) keeps its fused accumulation (a bound of R + 2 + 64 registers had dropped it)
    for name in ("sigma2", "parquet_sigma2", "parquet_sigma3"):
        assert fd.compile_table(workloads.get(name), specialize="isa").kernel_info()["has_acc"] == 1, name


def test_exponent_and_root_count_bounds():
    """The interpreter stream packs Power's exponent (biased) and the root index into 28 bits: larger ones are refused
    at fdg_graph_create instead of being truncated."""
    from feynmandiagram_jl_amd import capi
    from feynmandiagram_jl_amd.nodetable import OP_POWER, from_program
    ok = from_program(1, [(OP_POWER, (1 << 27) - 1, [(0, 1.0)])], [1])
    capi.GraphHandle(ok)
    for n in (1 << 27, -(1 << 27), 2**31 - 1):
        with pytest.raises(capi.FdgError) as e:
            capi.GraphHandle(from_program(1, [(OP_POWER, n, [(0, 1.0)])], [1]))
        assert e.value.code == capi.FDG_E_UNSUPPORTED


def test_call_value_is_the_last_root_statement(libfdg):
    """The generated function returns its last statement (static.jl:126-128: `root[k] = g` follows g's own statement).
    With graphs = [S, c], S = a + b and c a leaf visited after S, the text ends with `root[2] = gc`: the call's value
    is root[2], although c is a leaf and S an internal node."""
    reset_uid()
    a, b, c = Graph([]), Graph([]), Graph([])
    S = Graph([a, b], operator=Sum())
    text, _ = Compilers.to_julia_str([S, c])
    assert text.rstrip().splitlines()[-2].strip() == f"root[2] = g{c.id}"
    f, leafmap = Compilers.compile([S, c], specialize=False)
    assert f._last_root_value([3.0, 7.0]) == 7.0
    # the other way round: the leaf's statement precedes the node's
    f2, _ = Compilers.compile([c, S], specialize=False)
    assert f2._last_root_value([7.0, 3.0]) == 3.0


def test_c_text_datatypes(tmp_path):
    """to_Cstr / compile_C's `datatype` keyword (static.jl:134-153, 155-158): every type the reference maps, spelled the
    reference's way; the float text compiles and follows the double one to single precision; the complex text compiles."""
    import subprocess
    from feynmandiagram_jl_amd.lowering import julia_to_C_typestr
    assert [julia_to_C_typestr(x) for x in ("Float64", "Float32", "Int64", "Int32", "ComplexF32", "ComplexF64")] == \
        ["double ", "float ", "long long ", "int ", "complex float ", "complex double "]
    assert julia_to_C_typestr("Vector{Float32}") == "float *" and julia_to_C_typestr(np.float32) == "float "
    with pytest.raises(ValueError, match="Unsupported type"):
        julia_to_C_typestr("BigFloat")
    graphs, _ = fixtures.sigma2_graphs()
    text64, _ = Compilers.to_Cstr(graphs)
    text32, _ = Compilers.to_Cstr(graphs, datatype="Float32", name="eval32")
    textc, _ = Compilers.to_Cstr(graphs, datatype="ComplexF64", name="evalc")
    assert text64.startswith("\nvoid eval_graph(double *root, double *leafVal)") and "void eval32(float *root, float *leafVal)" in text32
    assert "    complex double  g" in textc                 # (the reference's declaration line: type string, then " g<id>,")
    src = tmp_path / "t.c"
    src.write_text("#include <math.h>\n#include <complex.h>\n#include <stdio.h>\n" + text64 + text32 + textc + """
int main(void) {
  double l[8], r[2]; float lf[8], rf[2]; complex double lc[8], rc[2];
  for (int i = 0; i < 8; ++i) { l[i] = 0.3 + 0.07 * i; lf[i] = (float)l[i]; lc[i] = l[i]; }
  eval_graph(r, l); eval32(rf, lf); evalc(rc, lc);
  printf("%.17g %.17g %.9g %.9g %.17g %.17g\\n", r[0], r[1], rf[0], rf[1], creal(rc[0]), creal(rc[1]));
  return 0;
}
""")
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-O1", "-ffp-contract=off", str(src), "-o", str(exe), "-lm"])
    v = [float(x) for x in subprocess.check_output([str(exe)]).split()]
    assert abs(v[2] - v[0]) <= 1e-5 * max(1.0, abs(v[0])) and abs(v[3] - v[1]) <= 1e-5 * max(1.0, abs(v[1]))
    assert v[4] == v[0] and v[5] == v[1]
    leafmap = Compilers.compile_C(graphs, str(tmp_path / "out.c"), datatype="Float32")
    assert "float *root" in (tmp_path / "out.c").read_text() and len(leafmap) == 8


def test_handle_options_replace_the_environment(libfdg, monkeypatch):
    """a handle's behaviour is a function of its own options.  The FDG_* environment is read once per process (the
    supported names only); after that neither a launch nor a specialisation looks at it -- options travel through fdg_graph_set_option /
    fdg_set_default_option -- and no source of the library calls getenv on an FDG_* name."""
    import glob
    import re
    t = workloads.get("sigma2")
    h = capi.GraphHandle(t)
    assert h.get_option("FDG_ISA_W2") is None
    monkeypatch.setenv("FDG_ISA_W2", "1")                        # the environment after the library's first use: not seen by anything
    monkeypatch.setenv("FDG_MC_ROUTE", "isa")
    assert capi.GraphHandle(t).get_option("FDG_ISA_W2") is None and capi.get_default_option("FDG_MC_ROUTE") is None
    h.set_option("FDG_ISA_W2", "1")
    assert h.get_option("FDG_ISA_W2") == "1" and capi.GraphHandle(t).get_option("FDG_ISA_W2") is None      # per handle
    h.set_option("FDG_ISA_W2", None)
    assert h.get_option("FDG_ISA_W2") is None
    capi.set_default_option("FDG_ISA_OVERSUB", "3")               # process default: handles created from now on
    try:
        assert capi.GraphHandle(t).get_option("FDG_ISA_OVERSUB") == "3" and h.get_option("FDG_ISA_OVERSUB") is None
    finally:
        capi.set_default_option("FDG_ISA_OVERSUB", None)
    assert capi.GraphHandle(t).get_option("FDG_ISA_OVERSUB") is None
    with pytest.raises(capi.FdgError):
        h.set_option("PATH", "x")                                 # names start with FDG_
    # the shipped kernel cache reaches the library as a default option, not through os.environ
    assert capi.KERNEL_CACHE in (capi.get_default_option("FDG_CACHE_RO_DIR") or "").split(":")
    src = os.path.join(os.path.dirname(capi.__file__), "csrc")
    for f in glob.glob(os.path.join(src, "*")):
        if f.endswith(("fdg_knobs.cpp", "fdg_knobs.h", "Makefile")):
            continue
        text = open(f, errors="replace").read()
        assert not re.search(r'getenv\("FDG_', text), f
    # the launch path (fdg_run_locked ... the launches) does not even look an option up by name
    rt = open(os.path.join(src, "fdg_runtime.hip")).read()
    body = rt[rt.index("// The launch path."):rt.index("// JIT\n")]
    assert "knob(" not in body and "getenv" not in body
