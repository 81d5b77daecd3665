"""BASELINE.json's full sizes in the layout and through the allocator bench.py times.

(i)  config 3: 10^8 + 37 samples of the 4-loop Parquet self-energy as a TILE-MAJOR batch from fdg_batch_alloc_pair -- what the headline
     launches -- checked against the oracle on 4 000 scattered samples (tile indices beyond 2^20 among them), on the ragged last tile, and,
     for the fused accumulation at that size, on a 10^6-sample slice near the end of the batch (weights zero elsewhere).
(ii) config 5: the 10^9-sample job is 500 steps of 2 x 10^6 samples whose Philox counters are the global sample index; three scattered steps
     are laid out as bench.py lays them out, and their partial sums are compared with the oracle's on a strided 40 000-sample slice each
     (the whole step on the CPU would take a minute per step) within 1e-12 of the sum of the absolute terms."""
import numpy as np
import pytest

import oracle
import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import capi, workloads

pytestmark = pytest.mark.gpu


def gather(x, idx, cuda):
    """rows of samples `idx` out of a tile-major [T, C, 64] tensor, as a host [n, C] array"""
    import torch
    i = torch.from_numpy(np.asarray(idx, dtype=np.int64)).to(cuda)
    return x[i // 64, :, i % 64].cpu().numpy()


def test_config3_tile_major_paired_batch_at_full_size(libfdg, cuda):
    import torch
    t = workloads.get("parquet_sigma4")
    L, R = t.n_leaf, t.n_root
    B = 100_000_000 + 37
    free_b, _ = torch.cuda.mem_get_info(cuda)
    if 8 * B * (L + R) + (110 << 30) > free_b:
        pytest.skip("needs 75 GB of device memory for the batch and 100 GB more for the allocator's search")
    f = fd.compile_table(t, specialize="isa")
    st = torch.cuda.current_stream().cuda_stream
    pb = f.tile_major_pair(B, cuda, calibrate=True)
    try:
        T = (B + 63) // 64
        assert T > (1 << 20) and pb.leaf.shape == (T, L, 64) and pb.info["n_chunk"] >= 30 and pb.info["n_probe"] > pb.info["n_chunk"]
        pb.root.fill_(7.0)
        capi.fill_uniform_device_tiled(pb.leaf.data_ptr(), B, L, 1, 64, 64 * L, 1234, 0, st)
        f.eval_tiled(pb.root, pb.leaf, B)
        torch.cuda.synchronize()
        assert f.kernel_info()["last_kernel"] == "fdg_isa_eval_nt"
        rng = np.random.default_rng(1)
        idx = np.sort(rng.choice(B, 4000, replace=False))
        idx[-37:] = np.arange(B - 37, B)                              # the ragged last tile
        idx[:64] = np.arange(64 * (1 << 20), 64 * (1 << 20) + 64)     # tile 2^20 whole
        assert (idx // 64 >= (1 << 20)).sum() > 1000
        sub = gather(pb.leaf, idx, cuda)
        assert np.array_equal(gather(pb.root, idx, cuda), oracle.eval_static(t, sub))
        # lanes past the batch in the last tile: never written
        assert (pb.root[T - 1, :, 37:] == 7.0).all()
        # the leaves are what the counter-based generator defines for these samples
        for b in (0, 64 * (1 << 20) + 5, B - 1):
            assert np.array_equal(gather(pb.leaf, [b], cuda)[0], oracle.philox_uniform(1, L, 1234, int(b))[0])
        # fused accumulation over the whole batch, weights zero outside a 10^6-sample slice near the end: the oracle's weighted sum of the slice
        lo, n = 93_000_017, 1_000_000
        w = torch.zeros(B, dtype=torch.float64, device=cuda)
        w[lo:lo + n] = torch.rand(n, dtype=torch.float64, device=cuda) + 0.5
        acc = f.accumulate_tiled(pb.leaf, w, None, B)
        torch.cuda.synchronize()
        assert f.kernel_info()["last_kernel"] == "fdg_isa_eval_acc_nt"
        sl = np.arange(lo, lo + n)
        h_leaf = np.concatenate([gather(pb.leaf, sl[k:k + 100_000], cuda) for k in range(0, n, 100_000)])
        want = oracle.eval_static(t, h_leaf)
        terms = want * w[lo:lo + n].cpu().numpy()[:, None]
        assert np.all(np.abs(acc.cpu().numpy() - terms.sum(0)) <= 1e-12 * np.maximum(1.0, np.abs(terms).sum(0)))
        # ... and the roots the evaluation wrote for that slice are the oracle's, bit for bit
        assert np.array_equal(np.concatenate([gather(pb.root, sl[k:k + 100_000], cuda) for k in range(0, n, 100_000)]), want)
    finally:
        pb.free()


def test_config5_partial_sums_of_scattered_steps(libfdg, cuda):
    import torch
    t = workloads.get("gv_sigma5")
    L, R = t.n_leaf, t.n_root
    per_step, n_step = 2_000_000, 500                     # bench.py: DEFAULT_B["gv_sigma5"], config5_steps(1)
    assert per_step * n_step == 1_000_000_000
    f = fd.compile_table(t, specialize="isa")
    st = torch.cuda.current_stream().cuda_stream
    T = (per_step + 63) // 64
    leaf = torch.empty((T, L, 64), dtype=torch.float64, device=cuda)
    rng = np.random.default_rng(5)
    sl = np.arange(17, per_step, 50)                      # 40 000 samples of the step
    for step in (0, 249, 499):
        off = step * per_step                             # the global index of the step's first sample: the Philox counter
        capi.fill_uniform_device_tiled(leaf.data_ptr(), per_step, L, 1, 64, 64 * L, 1234, off, st)
        w_h = rng.uniform(0.5, 1.5, per_step)
        w = torch.from_numpy(w_h).to(cuda)
        full = f.accumulate_tiled(leaf, w, None, per_step)
        w_sl = torch.zeros_like(w)
        w_sl[torch.from_numpy(sl).to(cuda)] = w[torch.from_numpy(sl).to(cuda)]
        part = f.accumulate_tiled(leaf, w_sl, None, per_step)
        rest = f.accumulate_tiled(leaf, w - w_sl, None, per_step)
        torch.cuda.synchronize()
        assert "acc" in f.kernel_info()["last_kernel"]
        h_leaf = gather(leaf, sl, cuda)
        assert np.array_equal(h_leaf[0], oracle.philox_uniform(1, L, 1234, off + int(sl[0]))[0])
        terms = oracle.eval_static(t, h_leaf) * w_h[sl][:, None]
        scale = np.maximum(1.0, np.abs(terms).sum(0))
        assert np.all(np.abs(part.cpu().numpy() - terms.sum(0)) <= 1e-12 * scale), step
        # the step's whole partial sum is the slice's plus the rest's (another summation order: the same tolerance, at the step's scale)
        root = f.eval_tiled(None, leaf, per_step)
        s_abs = (root.abs() * 1.5).sum(dim=2).sum(dim=0).cpu().numpy()
        assert np.all(np.abs(full.cpu().numpy() - (part + rest).cpu().numpy()) <= 1e-12 * np.maximum(1.0, s_abs)), step
