"""Batches owned by the library (include/fdg.h: fdg_batch_alloc, fdg_batch_alloc_pair, fdg_batch_free).

fdg_batch_alloc_pair is how bench.py's headline batch is allocated since round 5: the leaves one plain allocation, the roots mapped
chunk by chunk behind leaf windows, each chunk chosen by TIMING the handle's own kernel on (leaf window, root chunk) pairs
(DESIGN.md 6a).  Whatever it maps, the values must be the bits every other batch gives."""
import os
import numpy as np
import pytest

import oracle
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import capi, workloads

from test_tile_major import from_tiles


def test_pair_allocator_refuses_what_it_cannot_serve(libfdg):
    t = workloads.get("sigma2")
    h = fd.compile_table(t, specialize=False).handle                 # the table interpreter: no tile-major entry point
    with pytest.raises(capi.FdgError) as e:
        capi.batch_alloc_pair(h, 1000)
    assert e.value.code == capi.FDG_E_UNSUPPORTED
    h = fd.compile_table(t, specialize="isa").handle
    with pytest.raises(capi.FdgError) as e:
        capi.batch_alloc_pair(h, 0)
    assert e.value.code == capi.FDG_E_INVALID
    with pytest.raises(capi.FdgError) as e:
        capi.batch_free(0xdead000)                                    # not one of ours
    assert e.value.code == capi.FDG_E_INVALID


@pytest.mark.gpu
@pytest.mark.parametrize("nbytes,chunk", [(6 << 20, 0), (6 << 20, 2 << 20), (1000003, 0), ((64 << 20) + 4096, 6 << 20)])
def test_batch_alloc_small_and_odd_sizes(libfdg, cuda, nbytes, chunk):
    """no test called fdg_batch_alloc; chunk sizes that are not powers of two were given to hipMemAddressReserve as the alignment."""
    import torch
    t = workloads.get("sigma2")
    L, R = t.n_leaf, t.n_root
    f = fd.compile_table(t, specialize="isa")
    B = min(4099, nbytes // (8 * L * 64) * 64 - 3)
    T = (B + 63) // 64
    st = torch.cuda.current_stream().cuda_stream
    lp = capi.batch_alloc(nbytes, chunk)
    rp = capi.batch_alloc(T * R * 512, chunk)
    try:
        capi.fill_uniform_device_tiled(lp, B, L, 1, 64, 64 * L, 99, 0, st)
        f.handle.eval_device_tiled(lp, 1, 64, 64 * L, rp, 1, 64, 64 * R, B, st)
        torch.cuda.synchronize()
        got = torch.empty((T, R, 64), dtype=torch.float64, device=cuda)
        capi.copy_device(got.data_ptr(), rp, T * R * 64, st)
        torch.cuda.synchronize()
        want = oracle.eval_static(t, oracle.philox_uniform(B, L, 99))
        assert np.array_equal(from_tiles(got.cpu().numpy(), B, R), want)
    finally:
        capi.batch_free(lp)
        capi.batch_free(rp)


@pytest.mark.gpu
@pytest.mark.parametrize("name,B,chunk_mb,calibrate", [("sigma2", 4099, 0, True), ("parquet_sigma4", 70001, 0, False), ("parquet_sigma4", 2_600_000, 512, True),
                                                       ("gv_sigma5", 300_007, 0, True)])
def test_paired_batch_gives_the_bits_of_a_plain_batch(libfdg, cuda, name, B, chunk_mb, calibrate):
    """`a -m gpu test that fdg_batch_alloc-backed tile-major batches are bit-identical to torch.empty ones`.
    The third case is large enough (four windows of 512 MB) for the calibration to run: candidates drawn, pairs timed, chunks re-mapped."""
    import torch
    t = workloads.get(name)
    L, R = t.n_leaf, t.n_root
    f = fd.compile_table(t, specialize="isa")
    st = torch.cuda.current_stream().cuda_stream
    T = (B + 63) // 64
    pb = f.tile_major_pair(B, cuda, calibrate=calibrate, chunk_bytes=chunk_mb << 20)
    try:
        info = pb.info
        assert info["leaf_bytes"] >= T * L * 512 and info["root_bytes"] >= T * R * 512 and info["n_chunk"] >= 1
        assert info["leaf_bytes"] % (info["chunk_tiles"] * 512 * L) == 0 and info["root_bytes"] == info["n_chunk"] * info["chunk_tiles"] * 512 * R
        if chunk_mb and calibrate:
            assert info["n_chunk"] >= 3 and info["n_probe"] >= info["n_chunk"] and info["n_candidate"] >= info["n_chunk"]
            assert info["gbs_after_min"] > 0 and info["gbs_fast"] >= info["gbs_after_min"]
        assert pb.leaf.shape == (T, L, 64) and pb.root.shape == (T, R, 64) and pb.leaf.is_contiguous() and pb.root.is_contiguous()
        pb.root.fill_(7.0)
        capi.fill_uniform_device_tiled(pb.leaf.data_ptr(), B, L, 1, 64, 64 * L, 4321, 11, st)
        f.eval_tiled(pb.root, pb.leaf, B)
        w = torch.rand(B, dtype=torch.float64, device=cuda)
        acc = f.accumulate_tiled(pb.leaf, w, None, B)
        # the same batch on plain allocations
        leaf = torch.empty((T, L, 64), dtype=torch.float64, device=cuda)
        root = torch.full((T, R, 64), 7.0, dtype=torch.float64, device=cuda)
        capi.fill_uniform_device_tiled(leaf.data_ptr(), B, L, 1, 64, 64 * L, 4321, 11, st)
        f.eval_tiled(root, leaf, B)
        acc2 = f.accumulate_tiled(leaf, w, None, B)
        torch.cuda.synchronize()
        assert torch.equal(pb.leaf, leaf) or B % 64          # (lanes past B of the last tile are whatever the allocation held)
        assert torch.equal(pb.root, root)                    # every written lane the same bits, every lane past B still 7.0
        assert torch.equal(acc, acc2)
        n = min(B, 3000)
        lo = B - n                                           # the last samples: ragged last tile, the last window
        h_leaf = oracle.philox_uniform(B, L, 4321, 11)[lo:]
        got = from_tiles(pb.root.cpu().numpy(), B, R)[lo:]
        assert np.array_equal(got, oracle.eval_static(t, h_leaf))
    finally:
        pb.free()
    pb.free()                                                # idempotent


def many_roots_table(seed, R, missing):
    """L = 12 leaves, 60 random nodes, R roots of which `missing` positions carry no graph (FDG_NO_ROOT: root[k] must stay untouched)."""
    from feynmandiagram_jl_amd.nodetable import FDG_NO_ROOT, OP_PROD, OP_SUM, from_program
    rng = np.random.default_rng(seed)
    L, N = 12, 60
    nodes = []
    for n in range(N):
        k = int(rng.integers(2, 4))
        ch = [(int(rng.integers(0, L + n)), float(rng.choice([1.0, -1.0, 0.5, 2.0]))) for _ in range(k)]
        nodes.append((OP_SUM if rng.random() < 0.5 else OP_PROD, 0, ch))
    roots = [int(rng.integers(L, L + N)) for _ in range(R)]
    for k in missing:
        roots[k] = FDG_NO_ROOT
    return from_program(L, nodes, roots, f"many_roots_{seed}_{R}")


@pytest.mark.gpu
@pytest.mark.parametrize("R,missing", [(20, (0, 7, 19)), (45, (3,)), (16, (15,))])
def test_missing_roots_stay_untouched_through_the_root_scratch(libfdg, cuda, R, missing):
    """with 16 roots or more and row-major roots the evaluation goes through a column-major scratch and a transposition,
    and with more than 40 roots the accumulation goes through the scratch and a weighted sum -- the scratch's columns of FDG_NO_ROOT
    positions are never written by the kernel and must neither be copied into root[k] nor added to acc[k] (fdg.h: `left untouched`)."""
    import torch
    from feynmandiagram_jl_amd.nodetable import FDG_NO_ROOT
    t = many_roots_table(R, R, missing)
    live = t.root_slot != FDG_NO_ROOT
    assert (~live).sum() == len(missing)
    L = t.n_leaf
    f = fd.compile_table(t, specialize="isa")
    B = 64 * 9 + 17                                   # >= 256: the scratch path
    h_leaf = oracle.philox_uniform(B, L, 5) * 2 - 0.5
    want = oracle.eval_static(t, h_leaf, np.full((B, R), 9.0))
    assert (want[:, ~live] == 9.0).all()
    w = np.random.default_rng(1).uniform(0.5, 1.5, B)
    dw = torch.from_numpy(w).to(cuda)
    # poison the handle's scratch first: a larger call leaves NaNs where the next call's unwritten columns will be
    big = torch.full((4096, L), float("nan"), dtype=torch.float64, device=cuda)
    f(torch.empty((4096, R), dtype=torch.float64, device=cuda), big)
    f.accumulate(big, None, torch.zeros(R, dtype=torch.float64, device=cuda))
    for layout in ("row_major", "leaf_major", "tile_major"):
        if layout == "row_major":
            leaf = torch.from_numpy(h_leaf).to(cuda)
            root = torch.full((B, R), 9.0, dtype=torch.float64, device=cuda)
            f(root, leaf)
            got = root.cpu().numpy()
            acc = f.accumulate(leaf, dw, torch.full((R,), 3.0, dtype=torch.float64, device=cuda)).cpu().numpy()
        elif layout == "leaf_major":
            leaf = torch.from_numpy(np.ascontiguousarray(h_leaf.T)).to(cuda).t()
            root = torch.full((B, R), 9.0, dtype=torch.float64, device=cuda)      # row-major roots of leaf-major leaves: the scratch + transposition
            f(root, leaf)
            got = root.cpu().numpy()
            acc = f.accumulate(leaf, dw, torch.full((R,), 3.0, dtype=torch.float64, device=cuda)).cpu().numpy()
        else:
            from test_tile_major import to_tiles
            leaf = torch.from_numpy(to_tiles(h_leaf)).to(cuda)
            root = torch.full(((B + 63) // 64, R, 64), 9.0, dtype=torch.float64, device=cuda)
            f.eval_tiled(root, leaf, B)
            got = from_tiles(root.cpu().numpy(), B, R)
            acc = f.accumulate_tiled(leaf, dw, torch.full((R,), 3.0, dtype=torch.float64, device=cuda), B).cpu().numpy()
        assert np.array_equal(got, want), (layout, f.kernel_info()["last_kernel"])
        terms = want[:, live] * w[:, None]
        assert np.all(np.abs(acc[live] - 3.0 - terms.sum(0)) <= 1e-12 * np.maximum(1.0, np.abs(terms).sum(0))), layout
        assert (acc[~live] == 3.0).all(), layout      # acc[k] of a missing root: exactly what it was


@pytest.mark.gpu
@pytest.mark.parametrize("B,chunk_mb,calibrate", [(4099, 0, True), (2_600_000, 512, True)])
def test_paired_row_major_batch_gives_the_bits_of_a_plain_batch(libfdg, cuda, B, chunk_mb, calibrate):
    """FDG_BATCH_PAIR_ROW_MAJOR: the same allocator for compile_Python's [B, L] / [B, R] (compiler_python.jl:23,28,45-47)."""
    import torch
    t = workloads.get("parquet_sigma4")
    L, R = t.n_leaf, t.n_root
    f = fd.compile_table(t, specialize="isa")
    st = torch.cuda.current_stream().cuda_stream
    pb = f.row_major_pair(B, cuda, calibrate=calibrate, chunk_bytes=chunk_mb << 20)
    try:
        assert pb.leaf.shape == (B, L) and pb.root.shape == (B, R) and pb.leaf.is_contiguous() and pb.root.is_contiguous()
        if chunk_mb:
            assert pb.info["n_chunk"] >= 3 and pb.info["n_probe"] >= pb.info["n_chunk"]
        pb.root.fill_(7.0)
        capi.fill_uniform_device(pb.leaf.data_ptr(), B, L, L, 1, 4321, 11, st)
        f(pb.root, pb.leaf)
        leaf = torch.empty((B, L), dtype=torch.float64, device=cuda)
        capi.fill_uniform_device(leaf.data_ptr(), B, L, L, 1, 4321, 11, st)
        root = f(None, leaf)
        torch.cuda.synchronize()
        assert f.kernel_info()["last_kernel"] in ("fdg_isa_eval_rl", "fdg_isa_eval_rm", "fdg_isa_eval", "fdg_isa_eval_nt")
        assert torch.equal(pb.leaf, leaf) and torch.equal(pb.root, root)
        n = min(B, 2000)
        assert np.array_equal(pb.root[B - n:].cpu().numpy(), oracle.eval_static(t, oracle.philox_uniform(B, L, 4321, 11)[B - n:] if B < 10000 else leaf[B - n:].cpu().numpy()))
    finally:
        pb.free()


@pytest.mark.gpu
@pytest.mark.parametrize("B,calibrate", [(4099, True), (1_000_037, True)])
def test_paired_leaf_major_batch_gives_the_bits_of_a_plain_batch(libfdg, cuda, B, calibrate):
    """FDG_BATCH_PAIR_LEAF_MAJOR: a Julia column-major pair B' x L / B' x R (strides (1, B')); one window, whole root matrices as
    candidates (the second case, 670 MB of leaves, runs the search)."""
    import torch
    t = workloads.get("parquet_sigma4")
    L, R = t.n_leaf, t.n_root
    f = fd.compile_table(t, specialize="isa")
    st = torch.cuda.current_stream().cuda_stream
    pb = f.leaf_major_pair(B, cuda, calibrate=calibrate)
    try:
        Bp = pb.info["chunk_tiles"] * 64
        assert pb.info["n_chunk"] == 1 and Bp >= B
        assert pb.leaf.shape == (B, L) and pb.leaf.stride() == (1, Bp) and pb.root.shape == (B, R) and pb.root.stride() == (1, Bp)
        if B > 1_000_000:
            assert pb.info["n_probe"] >= 10 and pb.info["n_candidate"] >= 10
        pb.root.fill_(7.0)
        capi.fill_uniform_device(pb.leaf.data_ptr(), B, L, 1, Bp, 4321, 11, st)
        f(pb.root, pb.leaf)
        leaf = torch.empty((L, B), dtype=torch.float64, device=cuda).t()
        capi.fill_uniform_device(leaf.data_ptr(), B, L, 1, B, 4321, 11, st)
        root = torch.empty((R, B), dtype=torch.float64, device=cuda).t()
        f(root, leaf)
        torch.cuda.synchronize()
        assert torch.equal(pb.leaf, leaf) and torch.equal(pb.root, root)
        n = min(B, 2000)
        assert np.array_equal(pb.root[B - n:].cpu().numpy(), oracle.eval_static(t, leaf[B - n:].cpu().numpy()))
    finally:
        pb.free()


@pytest.mark.parametrize("scenario", [0, 1, 2, 3])
def test_the_allocators_search_on_a_model_of_the_memory(libfdg, scenario):
    """fdg_selftest_pair_search: PairSearch (csrc/fdg_batch.cpp) -- the part of fdg_batch_alloc_pair that chooses -- on a model: regions of four
    kinds, pair levels 0.855 / 0.80 / 0.765 by the number of bits in which window and candidate differ, noise.  Every window must end with a
    candidate of the complementary kind:
      0  every kind among the first candidates;
      1  NO complementary kind among them (every pair at the middle level at best): the exploration rule must keep drawing;
      2  the first windows' complement absent, the later windows' present: the second pass must draw (a bench process ended at 0.749 without it);
      3  as 0 with 1.5 % noise."""
    import ctypes as C
    for seed in range(12):
        n = C.c_uint32()
        top = libfdg.fdg_selftest_pair_search(seed, scenario, 32, C.byref(n))
        assert top == 32, (scenario, seed, top)
        assert 32 <= n.value <= (400 if scenario in (0, 3) else 4000)       # a few probes per window when the candidates are there


def test_pair_info_reports_how_far_the_search_got(libfdg):
    """fdg_batch_pair_info (round 6): level_reached / span_gb appended; the Julia shim's 128-byte buffer still holds it."""
    import ctypes as C
    names = [n for n, _ in capi.BatchPairInfo._fields_]
    assert names[-2:] == ["level_reached", "span_gb"] and C.sizeof(capi.BatchPairInfo) == 120 <= 128
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "fdg.h")).read()
    body = hdr[hdr.index("typedef struct fdg_batch_pair_info"):hdr.index("} fdg_batch_pair_info;")]
    assert body.index("seconds_settling") < body.index("level_reached") < body.index("span_gb")


@pytest.mark.gpu
def test_paired_allocator_degrades_when_memory_is_short(libfdg, cuda):
    """VERDICT r5 item 7 / ADVICE r5 (medium): with only the batch + ~10 GB free, the calibrated allocation must still return a batch -- a shorter
    span of candidates, or, when even that does not fit, the batch mapped in draw order -- and say so in level_reached; results are unchanged."""
    import torch
    t = workloads.get("parquet_sigma4")
    L, R = t.n_leaf, t.n_root
    f = fd.compile_table(t, specialize="isa")
    B = 6_000_000                                              # 4 GB of leaves: two windows, calibration on
    need = 8 * B * (L + R)
    torch.cuda.empty_cache()
    free_b, _total = torch.cuda.mem_get_info(cuda)
    keep = need + (10 << 30)
    if free_b < keep + (4 << 30):
        pytest.skip("not enough free memory to set the scene")
    ballast = [torch.empty((free_b - keep) // 4 // 8, dtype=torch.float64, device=cuda) for _ in range(4)]      # everything but batch + 10 GB
    try:
        pb = f.tile_major_pair(B, cuda, calibrate=True)
        try:
            assert pb.info["level_reached"] in (1, 2, 3) and pb.info["n_chunk"] >= 2
            assert pb.info["span_gb"] <= 12                    # nowhere near the 80 GB a free device would give
            st = torch.cuda.current_stream().cuda_stream
            capi.fill_uniform_device_tiled(pb.leaf.data_ptr(), B, L, 1, 64, 64 * L, 77, 0, st)
            f.eval_tiled(pb.root, pb.leaf, B)
            n = 2048
            got = from_tiles(pb.root[:n // 64].cpu().numpy(), n, R)
            want = oracle.eval_static(t, from_tiles(pb.leaf[:n // 64].cpu().numpy(), n, L))
            assert np.array_equal(got, want)
        finally:
            pb.free()
    finally:
        del ballast
        torch.cuda.empty_cache()
