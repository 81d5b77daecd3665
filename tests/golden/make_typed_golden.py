"""Generates tests/golden/typed_vectors.npz: seeded leaf inputs of element types Float32 / ComplexF64 / ComplexF32 for three graphs and
the roots the typed twin of the oracle (oracle.eval_static_typed) produces for them.  SELF-GENERATED, NOT PRODUCED BY JULIA (Julia is
not installed here): the vectors guard the twin against regressions and give the device tests fixed inputs that travel to the GPU box;
what pins the twin is listed in tests/test_typed.py.  Run from the repository root:  python tests/golden/make_typed_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle  # noqa: E402
from feynmandiagram_jl_amd import workloads  # noqa: E402

NP = {"Float32": np.float32, "ComplexF64": np.complex128, "ComplexF32": np.complex64}


def main():
    out = {}
    for name in ("sigma2", "parquet_sigma3", "gv_sigma4"):
        t = workloads.get(name)
        for dtype, npdt in NP.items():
            rng = np.random.default_rng(sum(map(ord, name + dtype)))
            x = rng.random((96, t.n_leaf)) * 2 - 0.7
            if dtype.startswith("Complex"):
                x = x + 1j * (rng.random(x.shape) * 2 - 1.1)
            x = x.astype(npdt)
            out[f"{name}:{dtype}:leaf"] = x
            out[f"{name}:{dtype}:root"] = oracle.eval_static_typed(t, x, dtype)
    np.savez_compressed(os.path.join(HERE, "typed_vectors.npz"), **out)
    print("wrote typed_vectors.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
