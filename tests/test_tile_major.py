"""Tile-major batches (include/fdg.h: fdg_eval_device_tiled, fdg_accumulate_device_tiled, fdg_fill_uniform_device_tiled):
sample b = 64 t + l of leaf i at leaf[t * tile_stride + l * sample_stride + i * leaf_stride] -- a Julia
Array{Float64,3}(64, L, cld(B, 64)).  Same bits as every other layout; ragged batches; padded strides; the error behaviour."""
import os

import numpy as np
import pytest

import oracle
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import capi, workloads


def to_tiles(x, pad_cols=0, pad_tile=0):
    """host [B, C] -> [T, C + pad_cols, 64] (+ pad_tile unused doubles per tile through a wider allocation); lanes past B hold NaN"""
    B, C = x.shape
    T = (B + 63) // 64
    out = np.full((T, C + pad_cols, 64), np.nan)
    full = np.full((T * 64, C), np.nan)
    full[:B] = x
    out[:, :C, :] = full.reshape(T, 64, C).transpose(0, 2, 1)
    return out


def from_tiles(r, B, C):
    T = r.shape[0]
    return r[:, :C, :].transpose(0, 2, 1).reshape(T * 64, C)[:B]


def test_host_mirror_checks_shapes(libfdg):
    t = workloads.get("sigma2")
    f = fd.compile_table(t, specialize="isa")
    with pytest.raises(ValueError):
        f.eval_tiled(None, np.zeros((2, t.n_leaf, 64)))         # a host array: tile-major batches live on the device


@pytest.mark.gpu
def test_fill_uniform_tiled_is_the_same_philox_stream(libfdg, cuda):
    import torch
    B, L = 1000, 7
    x = torch.full(((B + 63) // 64, L, 64), -1.0, dtype=torch.float64, device=cuda)
    capi.fill_uniform_device_tiled(x.data_ptr(), B, L, 1, 64, 64 * L, 1234, 5, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    h = x.cpu().numpy()
    assert np.array_equal(from_tiles(h, B, L), oracle.philox_uniform(B, L, 1234, 5))
    assert (h.transpose(0, 2, 1).reshape(-1, L)[B:] == -1.0).all()          # lanes past the batch are not written


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["sigma2", "parquet_sigma4", "gv_sigma5", "gv_sigma4_taylor2", "parquet_sigma4_insdyn"])
@pytest.mark.parametrize("B", [1, 63, 64, 4099, 70001])
def test_tile_major_evaluation_is_bit_exact(libfdg, cuda, name, B):
    import torch
    t = workloads.get(name)
    L, R = t.n_leaf, t.n_root
    f = fd.compile_table(t, specialize="isa")
    h_leaf = oracle.philox_uniform(B, L, 77)
    want = oracle.eval_static(t, h_leaf)
    for pad_l, pad_r in ((0, 0), (3, 2)):
        leaf = torch.from_numpy(to_tiles(h_leaf, pad_l)).to(cuda)
        root = torch.full(((B + 63) // 64, R + pad_r, 64), 9.0, dtype=torch.float64, device=cuda)
        f.eval_tiled(root, leaf, B)
        torch.cuda.synchronize()
        r = root.cpu().numpy()
        assert np.array_equal(from_tiles(r, B, R), want), (name, B, pad_l, f.kernel_info()["last_kernel"])
        tail = r.transpose(0, 2, 1).reshape(-1, R + pad_r)
        assert (tail[B:] == 9.0).all() and (tail[:, R:] == 9.0).all()      # nothing outside the batch is written
    # the unpadded, line-aligned batch takes the streaming form of the kernel
    assert f.kernel_info()["last_kernel"] in ("fdg_isa_eval", "fdg_isa_eval_nt")
    # fused accumulation over the same batch
    w = oracle.philox_uniform(B, 1, 5)[:, 0]
    acc = f.accumulate_tiled(torch.from_numpy(to_tiles(h_leaf)).to(cuda), torch.from_numpy(w).to(cuda), None, B).cpu().numpy()
    ref = (want * w[:, None]).sum(axis=0)
    scale = (np.abs(want) * w[:, None]).sum(axis=0) + 1e-300
    assert np.all(np.abs(acc - ref) <= 1e-12 * scale), (name, B)


@pytest.mark.gpu
def test_tile_stride_zero_is_the_plain_matrix_and_other_back_ends_refuse(libfdg, cuda):
    import torch
    t = workloads.get("parquet_sigma4")
    L, R, B = t.n_leaf, t.n_root, 777
    h_leaf = oracle.philox_uniform(B, L, 9)
    want = oracle.eval_static(t, h_leaf)
    leaf = torch.from_numpy(np.ascontiguousarray(h_leaf.T)).to(cuda)          # leaf-major [L, B]
    root = torch.zeros((R, B), dtype=torch.float64, device=cuda)
    st = torch.cuda.current_stream().cuda_stream
    f = fd.compile_table(t, specialize="isa")
    f.handle.eval_device_tiled(leaf.data_ptr(), 1, B, 0, root.data_ptr(), 1, B, 0, B, st)
    torch.cuda.synchronize()
    assert np.array_equal(root.cpu().numpy().T, want)
    tl = torch.from_numpy(to_tiles(h_leaf)).to(cuda)
    tr = torch.zeros(((B + 63) // 64, R, 64), dtype=torch.float64, device=cuda)
    for spec in (True, False):
        g = fd.compile_table(t, specialize=spec)
        with pytest.raises(capi.FdgError) as e:
            g.eval_tiled(tr, tl, B)
        assert e.value.code == capi.FDG_E_UNSUPPORTED
    with pytest.raises(capi.FdgError) as e:
        f.handle.eval_device_tiled(tl.data_ptr(), 1, 64, -64, tr.data_ptr(), 1, 64, 64 * R, B, st)
    assert e.value.code == capi.FDG_E_INVALID


@pytest.mark.gpu
def test_tile_major_batch_of_config_2_size(libfdg, cuda):
    """BASELINE config 2's 10^7 samples of the 2-loop self-energy, tile-major, against the leaf-major evaluation of the same
    Philox stream (the oracle checks a prefix)."""
    import torch
    t = workloads.get("sigma2")
    L, R, B = t.n_leaf, t.n_root, 10_000_000 + 37
    st = torch.cuda.current_stream().cuda_stream
    f = fd.compile_table(t, specialize="isa")
    leaf_t = f.tile_major_empty(B, L, cuda)
    capi.fill_uniform_device_tiled(leaf_t.data_ptr(), B, L, 1, 64, 64 * L, 1234, 0, st)
    root_t = f.eval_tiled(None, leaf_t, B)
    leaf_c = torch.empty((L, B), dtype=torch.float64, device=cuda)
    capi.fill_uniform_device(leaf_c.data_ptr(), B, L, 1, B, 1234, 0, st)
    root_c = f(None, leaf_c.t())
    torch.cuda.synchronize()
    flat = root_t.permute(0, 2, 1).reshape(-1, R)[:B]
    assert torch.equal(flat.view(torch.int64), root_c.view(torch.int64))
    n = 5000
    assert np.array_equal(flat[:n].cpu().numpy(), oracle.eval_static(t, oracle.philox_uniform(n, L, 1234)))


@pytest.mark.gpu
@pytest.mark.parametrize("name,force", [("parquet_ver4_3", True), ("parquet_ver4_4", True), ("gv_ver4_4", False)])
def test_pooled_cooperative_variant_on_device(libfdg, cuda, monkeypatch, tmp_path, name, force, fdgopt):
    """fdg_isa_eval_pool: the four waves of a CU evaluate one tile, whole roots each, the tile's leaves fetched once into
    a shared LDS pool.  Taken for full tiles of tile-major batches (and of leaf-major matrices whose leaves lie within 2 GB); the last
    B % 64 samples go through the one-wave kernel.  Bit-exact; the graphs of example/benchmark.jl and example/benchmark_GV.jl get it by
    the library's own criterion (example/benchmark_GV.jl's: its one-wave kernel moves 3.0 x the algorithmic bytes) or when forced
    (example/benchmark.jl's moves 1.3 x and is faster left alone: profiles/r04_log_pool_pick.txt)."""
    import torch
    if force:
        fdgopt.set("FDG_ISA_POOL", "1")
    t = workloads.get(name)
    L, R = t.n_leaf, t.n_root
    f = fd.compile_table(t, specialize="isa", cache_dir=str(tmp_path) if force else None)
    ki = f.kernel_info()
    assert ki["has_pool"] == 1 and ki["pool_fetch"] >= f.info()["n_live_leaf"]
    for B in (64, 4099, 70016):
        h_leaf = oracle.philox_uniform(B, L, 91)
        want = oracle.eval_static(t, h_leaf)
        leaf = torch.from_numpy(to_tiles(h_leaf)).to(cuda)
        root = torch.full(((B + 63) // 64, R, 64), 9.0, dtype=torch.float64, device=cuda)
        f.eval_tiled(root, leaf, B)
        torch.cuda.synchronize()
        assert f.kernel_info()["last_kernel"] == "fdg_isa_eval_pool"
        r = root.cpu().numpy()
        assert np.array_equal(from_tiles(r, B, R), want), (name, B)
        assert (r.transpose(0, 2, 1).reshape(-1, R)[B:] == 9.0).all()
    # a leaf-major matrix small enough for 32-bit leaf offsets takes the same kernel; a wide one the one-wave kernel
    B = 6400
    h_leaf = oracle.philox_uniform(B, L, 92)
    leaf = torch.from_numpy(np.ascontiguousarray(h_leaf.T)).to(cuda).t()
    got = f(None, leaf)
    torch.cuda.synchronize()
    assert f.kernel_info()["last_kernel"] == "fdg_isa_eval_pool"
    assert np.array_equal(got.cpu().numpy(), oracle.eval_static(t, h_leaf))
    f.handle.set_option("FDG_ISA_NO_POOL", "1")
    got = f(None, leaf)
    torch.cuda.synchronize()
    assert f.kernel_info()["last_kernel"].startswith("fdg_isa_eval") and f.kernel_info()["last_kernel"] != "fdg_isa_eval_pool"
    assert np.array_equal(got.cpu().numpy(), oracle.eval_static(t, h_leaf))


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("name,slack", [("parquet_ver4_3", 1), ("parquet_ver4_3", 2), ("gv_ver4_4", 1)])
def test_pooled_variant_with_flag_synchronisation_on_device(libfdg, cuda, tmp_path, name, slack):
    """Option FDG_POOL_SYNC=flags (round 6): the pooled kernel without s_barrier between the epochs of a tile -- every wave
    publishes the sync points it has reached in an LDS word of its own and polls the others' only when the copy it read one sync point earlier is
    not enough.  Bit-exact on the device, many tiles per workgroup (the counters run on across tiles), ragged last tile through the one-wave
    kernel.  Measured 3-7 % SLOWER than the barrier form (profiles/r06_log_sweep_k.txt), so it stays an option; the poll gives up after 2^14
    rounds, so a protocol bug would show here as wrong bits, not as a hung device."""
    import torch
    t = workloads.get(name)
    L, R = t.n_leaf, t.n_root
    f = fd.compile_table(t, specialize="isa", cache_dir=str(tmp_path), options={"FDG_ISA_POOL": "1", "FDG_POOL_SYNC": "flags", "FDG_POOL_SLACK": str(slack)})
    assert f.kernel_info()["has_pool"] == 1
    for B in (64, 64 * 2100 + 17):           # 2100 tiles on 256 CUs: eight or nine tiles per workgroup
        h_leaf = oracle.philox_uniform(B, L, 93)
        leaf = torch.from_numpy(to_tiles(h_leaf)).to(cuda)
        root = torch.full(((B + 63) // 64, R, 64), 9.0, dtype=torch.float64, device=cuda)
        for _ in range(3):
            f.eval_tiled(root, leaf, B)
        torch.cuda.synchronize()
        assert f.kernel_info()["last_kernel"] in ("fdg_isa_eval_pool", "fdg_isa_eval", "fdg_isa_eval_nt")
        assert np.array_equal(from_tiles(root.cpu().numpy(), B, R), oracle.eval_static(t, h_leaf)), (name, slack, B)


@pytest.mark.gpu
def test_pooled_graph_accumulates_through_the_pool(libfdg, cuda):
    """A graph with the pooled variant (example/benchmark_GV.jl's vertex function) accumulates through it: pooled evaluation into the
    column-major root scratch, then the weighted sum -- its fused-accumulation program (26 accumulators taken from the value registers,
    no pool) is a third slower.  Full tiles and the last B % 64 samples; tile-major and leaf-major batches; within 1e-12 of the terms' scale."""
    import torch
    t = workloads.get("gv_ver4_4")
    L, R = t.n_leaf, t.n_root
    f = fd.compile_table(t, specialize="isa")
    assert f.kernel_info()["has_pool"] == 1
    for B in (64 * 40, 4099):
        h_leaf = oracle.philox_uniform(B, L, 95)
        want = oracle.eval_static(t, h_leaf)
        w = np.random.default_rng(B).uniform(0.5, 1.5, B)
        terms = want * w[:, None]
        tol = 1e-12 * np.maximum(1.0, np.abs(terms).sum(0))
        leaf = torch.from_numpy(to_tiles(h_leaf)).to(cuda)
        acc = torch.zeros(R, dtype=torch.float64, device=cuda)
        f.accumulate_tiled(leaf, torch.from_numpy(w).to(cuda), acc, B)
        torch.cuda.synchronize()
        assert f.kernel_info()["last_kernel"] == "fdg_isa_eval_pool"
        assert np.all(np.abs(acc.cpu().numpy() - terms.sum(0)) <= tol), B
        lm = torch.from_numpy(np.ascontiguousarray(h_leaf.T)).to(cuda).t()
        acc.zero_()
        f.accumulate(lm, torch.from_numpy(w).to(cuda), acc)
        torch.cuda.synchronize()
        assert f.kernel_info()["last_kernel"] == "fdg_isa_eval_pool"
        assert np.all(np.abs(acc.cpu().numpy() - terms.sum(0)) <= tol), B


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["parquet_sigma4", "sigma2", "parquet_sigma3", "parquet_sigma2"])
def test_linear_row_major_variant_on_device(libfdg, cuda, name):
    """fdg_isa_eval_rl: compile_Python's row-major [B, L] with contiguous rows (src/backend/compiler_python.jl:23,28,45-47) -- a tile's 64 rows
    are one block of 512 L bytes, streamed into an LDS image once, every cache line requested exactly once.  Bit-exact; the last B % 64 rows and
    matrices with padded rows take the other kernels."""
    import torch
    t = workloads.get(name)
    L, R = t.n_leaf, t.n_root
    f = fd.compile_table(t, specialize="isa")
    assert f.kernel_info()["has_rl"] == 1
    for B in (64, 4099, 200_000):
        h_leaf = oracle.philox_uniform(B, L, 93)
        want = oracle.eval_static(t, h_leaf)
        leaf = torch.from_numpy(h_leaf).to(cuda)
        root = torch.full((B, R), 9.0, dtype=torch.float64, device=cuda)
        f(root, leaf)
        torch.cuda.synchronize()
        assert f.kernel_info()["last_kernel"] == "fdg_isa_eval_rl", f.kernel_info()["last_kernel"]
        assert np.array_equal(root.cpu().numpy(), want), (name, B)
    # rows with padding between them are not one block: the chunked row-major variant (or the transposition) takes them
    wide = torch.zeros((300, L + 3), dtype=torch.float64, device=cuda)
    wide[:, :L] = torch.from_numpy(h_leaf[:300]).to(cuda)
    got = f(None, wide[:, :L])
    torch.cuda.synchronize()
    assert f.kernel_info()["last_kernel"] != "fdg_isa_eval_rl"
    assert np.array_equal(got.cpu().numpy(), want[:300])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["parquet_sigma4", "sigma2", "parquet_sigma3"])
def test_linear_row_major_variant_accumulates_on_device(libfdg, cuda, name):
    """fdg_isa_eval_rl_acc: acc[k] += sum_b w_b root_k(b) over compile_Python's row-major [B, L] with contiguous rows -- the linear variant's image
    and the fused accumulators in one kernel (the roots are never written; without it the 2- and 3-loop graphs went through a transposition
    pass: 0.13-0.21 of the HBM roof).  Agrees with the weighted sum of the oracle's roots to 1e-12 of the terms' scale; with and without
    weights; the last B % 64 rows through the plain accumulating kernel."""
    import torch
    t = workloads.get(name)
    L, R = t.n_leaf, t.n_root
    f = fd.compile_table(t, specialize="isa")
    for B, weighted in ((64 * 50, True), (200_003, True), (4099, False)):
        h_leaf = oracle.philox_uniform(B, L, 97)
        want = oracle.eval_static(t, h_leaf)
        w = np.random.default_rng(B).uniform(0.5, 1.5, B) if weighted else np.ones(B)
        leaf = torch.from_numpy(h_leaf).to(cuda)
        acc = torch.zeros(R, dtype=torch.float64, device=cuda)
        f.accumulate(leaf, torch.from_numpy(w).to(cuda) if weighted else None, acc)
        torch.cuda.synchronize()
        assert f.kernel_info()["last_kernel"] == "fdg_isa_eval_rl_acc", f.kernel_info()["last_kernel"]
        terms = want * w[:, None]
        assert np.all(np.abs(acc.cpu().numpy() - terms.sum(0)) <= 1e-12 * np.maximum(1.0, np.abs(terms).sum(0))), (name, B)
        # a second call adds to the accumulators
        f.accumulate(leaf, torch.from_numpy(w).to(cuda) if weighted else None, acc)
        torch.cuda.synchronize()
        assert np.all(np.abs(acc.cpu().numpy() - 2 * terms.sum(0)) <= 2e-12 * np.maximum(1.0, np.abs(terms).sum(0))), (name, B)


@pytest.mark.gpu
@pytest.mark.parametrize("generic", [False, True])
@pytest.mark.parametrize("B", [64 * 37, 70_001])
def test_leaves_from_K_T_into_a_tile_major_batch(libfdg, cuda, monkeypatch, B, generic, fdgopt):
    """fdg_leaf_eval_device_tiled (SURVEY.md 8f row 3 on the layout of 8d): the leaf loop of example/benchmark.jl:58-81 writes a tile-major
    batch -- the same bits as fdg_leaf_eval_device writes into a plain matrix -- and the tiled evaluator takes it from there: the
    Monte-Carlo chain (K, T) -> leaves -> roots on the layout that streams fastest, roots equal bit for bit to the plain chain's."""
    import torch
    if generic: fdgopt.set("FDG_LEAF_GENERIC", "1")          # the table-driven kernel instead of the one specialised to the tables
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "gv_sigma4_leafstates.npz"))
    t = workloads.get("gv_sigma4")
    L, R = t.n_leaf, t.n_root
    dim, n_loop, n_tau = 3, int(z["basis"].shape[1]), int(z["n_tau"])
    kF, beta, lam = 1.919, 3.0, 1.2
    rng = np.random.default_rng(11)
    K = rng.uniform(-2.0, 2.0, size=(B, n_loop, dim))
    T = rng.uniform(0.0, beta, size=(B, n_tau)); T[:, 0] = 0.0
    dK = torch.from_numpy(np.ascontiguousarray(K.reshape(B, n_loop * dim).T)).to(cuda)
    dT = torch.from_numpy(np.ascontiguousarray(T.T)).to(cuda)
    st = torch.cuda.current_stream().cuda_stream
    args = (z["leaf_type"], z["leaf_order"], z["tau_in"], z["tau_out"], z["loop_index"], z["basis"], dim, n_tau)
    plain = torch.ones((L, B), dtype=torch.float64, device=cuda).t()
    capi.leaf_eval_device(*args, kF, beta, lam, dK.data_ptr(), 1, B, dT.data_ptr(), 1, B, plain.data_ptr(), plain.stride(0), plain.stride(1), B, st)
    n_tile = (B + 63) // 64
    tiled = torch.ones((n_tile, L, 64), dtype=torch.float64, device=cuda)          # type-0 leaves stay 1.0 in both
    capi.leaf_eval_device_tiled(*args, kF, beta, lam, dK.data_ptr(), 1, B, dT.data_ptr(), 1, B, tiled.data_ptr(), 1, 64, 64 * L, B, st)
    torch.cuda.synchronize()
    got = tiled.permute(0, 2, 1).reshape(-1, L)[:B]
    assert torch.equal(got, plain)
    if B % 64:
        assert bool((tiled.permute(0, 2, 1).reshape(-1, L)[B:] == 1.0).all())         # nothing written behind the last sample
    f = fd.compile_table(t, specialize="isa")
    want = f(None, plain)
    root = torch.empty((n_tile, R, 64), dtype=torch.float64, device=cuda)
    f.eval_tiled(root, tiled, B)
    torch.cuda.synchronize()
    assert torch.equal(root.permute(0, 2, 1).reshape(-1, R)[:B], want)
    with pytest.raises(capi.FdgError):
        capi.leaf_eval_device_tiled(*args, kF, beta, lam, dK.data_ptr(), 1, B, dT.data_ptr(), 1, B, tiled.data_ptr(), 1, 64, 0, B, st)


@pytest.mark.gpu
@pytest.mark.parametrize("B,C", [(64, 4), (4099, 84), (1000, 1), (130, 67), (20011, 16)])
def test_repack_tile_major_round_trip(libfdg, cuda, B, C):
    """fdg_repack_tile_major / fdg_unpack_tile_major (round 6; the Julia shim's tile_major! / from_tile_major!): a Julia column-major
    B x C matrix (512-byte runs copied as they are), compile_Python's row-major [B, C] (64 x 64 tiles through LDS) and a row-major
    matrix with padded rows all give the tile-major array of to_tiles(); lanes past B are left alone; the way back restores the matrix."""
    import torch
    rng = np.random.default_rng(B * 131 + C)
    x = rng.standard_normal((B, C))
    want = to_tiles(x)
    T = (B + 63) // 64
    srcs = {"row_major": torch.from_numpy(x).to(cuda),
            "col_major": torch.from_numpy(np.ascontiguousarray(x.T)).to(cuda).t(),
            "padded_rows": torch.from_numpy(np.concatenate([x, np.full((B, 3), 7.0)], axis=1)).to(cuda)[:, :C],
            "odd_column_stride": torch.from_numpy(np.concatenate([x.T, np.full((C, 1), 7.0)], axis=1).copy()).to(cuda)[:, :B].t()}
    for name, src in srcs.items():
        assert tuple(src.shape) == (B, C)
        dst = torch.full((T, C, 64), float("nan"), dtype=torch.float64, device=cuda)
        fd.GraphFunc.tile_major_(dst, src)
        got = dst.cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(want)), name
        assert np.array_equal(got[~np.isnan(want)], want[~np.isnan(want)]), name
        back = torch.empty_strided(src.size(), src.stride(), dtype=torch.float64, device=cuda).fill_(5.0)       # the same layout as the source
        fd.GraphFunc.from_tile_major_(back, dst)
        assert np.array_equal(back.cpu().numpy(), x), name


@pytest.mark.gpu
def test_repacked_batch_evaluates_to_the_same_bits(libfdg, cuda):
    """A leaf-major and a row-major matrix repacked with tile_major_ and evaluated tile-major give the bits of the in-place evaluation,
    and the roots unpacked with from_tile_major_ those of the reference's [B, R]."""
    import torch
    t = workloads.get("parquet_sigma4")
    f = fd.compile_table(t, specialize="isa")
    B, L, R = 70_001, t.n_leaf, t.n_root
    leaf = torch.empty((B, L), dtype=torch.float64, device=cuda)
    capi.fill_uniform_device(leaf.data_ptr(), B, L, L, 1, 99, 0, torch.cuda.current_stream().cuda_stream)
    want = f(None, leaf)
    for src in (leaf, leaf.t().contiguous().t()):
        tl = fd.GraphFunc.tile_major_(None, src)
        tr = f.eval_tiled(None, tl, B)
        root = torch.empty((B, R), dtype=torch.float64, device=cuda)
        fd.GraphFunc.from_tile_major_(root, tr)
        assert torch.equal(root, want)
    n = 3000
    assert np.array_equal(want[:n].cpu().numpy(), oracle.eval_static(t, leaf[:n].cpu().numpy()))
