"""``Compilers`` -- the back-end API surface of the reference, with the HIP
evaluator as the product path.

Mirrors src/backend/compiler.jl:16-18 + static.jl + compiler_python.jl:

* ``compile(graphs; root) -> (GraphFunc, leafmap)``  (static.jl:221-227).  The
  returned callable has the generated function's semantics:
  ``f(root, leafVal)`` mutates ``root`` in place and returns the last assigned
  root value (test/compiler.jl:15,28); it also accepts a ``[B, L]`` leaf matrix
  with a ``[B, R]`` root matrix (the batched layout of compile_Python,
  compiler_python.jl:23,28,45-47) and torch CUDA tensors (zero copy).
* ``compile_Julia / compile_C / compile_Python(graphs, filename; root,
  func_name)`` append source text to a file and return ``leafmap``
  (static.jl:244-251, 269-279; compiler_python.jl:53-60).
* ``to_julia_str / to_Cstr / to_python_str`` (re-exported from lowering).

``GraphFunc`` is the name BASELINE.json's north_star uses for the callable; the
reference itself returns a RuntimeGeneratedFunction.
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

from . import capi
from .graph import Graph
from .lowering import lower, to_Cstr, to_julia_str, to_python_str
from .nodetable import FDG_NO_ROOT, NodeTable

__all__ = ["compile", "compile_hip", "compile_table", "GraphFunc", "compile_Julia", "compile_C",
           "compile_Python", "to_julia_str", "to_Cstr", "to_python_str"]


def _is_torch(x) -> bool:
    return type(x).__module__.split(".")[0] == "torch"


class GraphFunc:
    """Callable evaluator bound to one lowered graph set (one ``fdg_graph``)."""

    def __init__(self, table: NodeTable, specialize=False, cache_dir: Optional[str] = None,
                 flags: int = 0, opt: Optional[dict] = None, association: str = "static", options: Optional[dict] = None):
        """``specialize``: "auto" (ISA, else HIP source), "isa" / "isa-autotune" (optimizing back end,
        gfx950 assembly), True / "hip" (straight-line HIP source through hiprtc), False (table
        interpreter, no JIT).
        ``options``: handle options set before anything is specialised (``fdg_graph_set_option``; what used to be FDG_* environment
        switches: ``{"FDG_ISA_W2": "1"}``).
        ``association``: which of the reference's two evaluators the results equal bit for bit -- "static", the function
        ``Compilers.compile`` generates (static.jl:13-46), or "eval", the interpreter ``eval!`` the reference's examples and
        tests call (eval.jl:1-3,15-39; example/benchmark.jl:84-86), whose products fold the already scaled operands."""
        if association not in ("static", "eval"):
            raise ValueError('association must be "static" or "eval"')
        self.table = table.normalized()
        self._specialize, self._cache_dir, self._flags = specialize, cache_dir, flags
        self.association = association
        self.handle = capi.GraphHandle(self.table)
        self.handle.set_options(options)
        if association == "eval":
            self.handle.set_association(capi.FDG_ASSOC_INTERP)
        self.n_leaf, self.n_root = self.table.n_leaf, self.table.n_root
        if specialize == "auto":
            # best available: gfx950 assembly (every Power{N}: pow_body spelled out); a graph it refuses for another reason (too few
            # registers for its roots, ...) goes through the HIP-source JIT.  Both are JIT back ends of the same ABI, not fallbacks to a CPU.
            try:
                self.handle.specialize(cache_dir, flags | capi.FDG_SPEC_ISA)
                # compile_Python's row-major [B, L] is the reference's batched layout.  Most graphs read it in place through
                # the ISA back end's row-major variant (LDS staging: 4-loop self-energies 5.7e9 evals/s against 1.7e9 for
                # the HIP-source kernels).  Handles without that variant (fewer than 16 leaves, the tiny-graph
                # configuration, plans that would re-fetch too much) would transpose chunks in front of the ISA kernel, so
                # for small graphs the HIP-source kernels, whose lanes read their own rows, ride along as a companion
                # (2-loop self-energy: 4.3e10 vs 1.5e10 evals/s).  The handle says which case it is.
                if not self.handle.kernel_info()["has_rm"] and self.n_leaf > 1 and self.table.n_node <= 4000:
                    self.handle.specialize(cache_dir, flags | capi.FDG_SPEC_ROW_MAJOR_COMPANION)
            except capi.FdgError as e:
                if e.code != capi.FDG_E_UNSUPPORTED:
                    raise
                self.handle.specialize(cache_dir, flags)
        elif specialize in ("isa", "isa-autotune"):
            if opt:
                self.handle.set_opt_params(**opt)
            if specialize == "isa-autotune":
                flags |= capi.FDG_SPEC_AUTOTUNE
            self.handle.specialize(cache_dir, flags | capi.FDG_SPEC_ISA)
        elif specialize:
            self.handle.specialize(cache_dir, flags)

    # -- introspection ------------------------------------------------------- #
    def info(self) -> dict:
        return self.handle.info()

    def kernel_info(self) -> dict:
        return self.handle.kernel_info()

    def specialize(self, cache_dir: Optional[str] = None, flags: int = 0) -> "GraphFunc":
        self.handle.specialize(cache_dir, flags)
        return self

    def _last_root_value(self, root_row):
        slots = [k for k in range(self.n_root) if int(self.table.root_slot[k]) != FDG_NO_ROOT]
        if not slots:
            return None
        # the generated function returns its last assignment: the root whose
        # statement comes last in emission order (static.jl:126-128)
        k = max(slots, key=lambda k: self._emission_rank(int(self.table.root_slot[k])))
        return float(root_row[k])

    def _emission_rank(self, v: int):
        """Position of value v's statement in the emitted text: internal node n is the (n+1)-th internal statement; a
        leaf's load comes after ``leaf_pos`` internal statements (and after the loads of lower-numbered leaves there)."""
        L = self.table.n_leaf
        if v >= L:
            return (v - L + 1, 0, 0)
        return (int(self.table.leaf_positions()[v]), 1, v) if L else (0, 1, v)

    # -- the call ------------------------------------------------------------- #
    def __call__(self, root, leafVal):
        if _is_torch(leafVal):
            return self._call_torch(root, leafVal)
        if isinstance(leafVal, np.ndarray) and leafVal.dtype in (np.float32, np.complex128, np.complex64):
            return self._call_numpy_typed(root, leafVal)
        leaf = np.asarray(leafVal, dtype=np.float64)
        if leaf.ndim == 1:
            if leaf.shape[0] < self.n_leaf:
                raise IndexError(f"BoundsError: attempt to access {leaf.shape[0]}-element leafVal at index [{self.n_leaf}]")
            if len(root) < self.n_root:
                raise IndexError(f"BoundsError: attempt to access {len(root)}-element root at index [{self.n_root}]")
            r = np.array([float(root[k]) for k in range(self.n_root)], dtype=np.float64).reshape(1, -1)
            self.handle.eval_host(leaf[None, :self.n_leaf], r)
            for k in range(self.n_root):
                root[k] = r[0, k]
            return self._last_root_value(r[0])
        if leaf.ndim != 2:
            raise ValueError("leafVal must be a vector or a [B, L] matrix")
        B = leaf.shape[0]
        if root is None:
            root = np.zeros((B, self.n_root), dtype=np.float64)
        if not (isinstance(root, np.ndarray) and root.dtype == np.float64 and (root.flags.c_contiguous or root.flags.f_contiguous)
                and root.shape == (B, self.n_root)):
            raise ValueError("root must be a contiguous float64 array of shape [B, R]")
        self.handle.eval_host(leaf, root)
        return root

    def _call_torch(self, root, leaf):
        import torch
        if not leaf.is_cuda:
            raise RuntimeError("torch leafVal must live on the GPU (no CPU fallback); pass a numpy array for host data")
        if leaf.dtype != torch.float64:
            return self._call_torch_typed(root, leaf)
        squeeze = leaf.dim() == 1
        if squeeze:
            leaf = leaf[None, :]
        B, Lc = leaf.shape
        if Lc < self.n_leaf:
            raise IndexError("BoundsError: leafVal has fewer columns than the graph has leaves")
        if root is None:
            root = torch.zeros((self.n_root,) if squeeze else (B, self.n_root), dtype=torch.float64, device=leaf.device)
        r2 = root[None, :] if squeeze else root
        if r2.dim() != 2 or r2.dtype != torch.float64 or r2.shape[0] != B or r2.shape[1] < self.n_root:
            raise ValueError("root must be a float64 [B, R] tensor on the same device")
        st = torch.cuda.current_stream(leaf.device).cuda_stream
        with torch.cuda.device(leaf.device):
            self.handle.eval_device(leaf.data_ptr(), leaf.stride(0), leaf.stride(1), r2.data_ptr(),
                                    r2.stride(0), r2.stride(1), B, st)
        return root

    # -- tile-major batches (fdg_eval_device_tiled): the layout the evaluator streams best ------------------------------ #
    @staticmethod
    def tile_major_empty(n_sample: int, n_col: int, device, dtype=None):
        """An uninitialised tile-major batch for ``n_sample`` samples of ``n_col`` values: a ``[cld(B, 64), n_col, 64]``
        tensor (tile, value, sample-in-tile) -- memory order of a Julia ``Array{Float64,3}(undef, 64, n_col, cld(B, 64))``."""
        import torch
        return torch.empty(((n_sample + 63) // 64, n_col, 64), dtype=dtype or torch.float64, device=device)

    def row_major_pair(self, n_sample: int, device, calibrate: bool = True, chunk_bytes: int = 0, verbose: bool = False) -> "PairedBatch":
        """The same for ``compile_Python``'s row-major layout: ``.leaf`` is ``[B, L]``, ``.root`` ``[B, R]`` (rows contiguous)."""
        return PairedBatch(self, n_sample, device, calibrate, chunk_bytes, verbose, capi.BATCH_PAIR_ROW_MAJOR)

    def leaf_major_pair(self, n_sample: int, device, calibrate: bool = True, verbose: bool = False) -> "PairedBatch":
        """The same for a Julia column-major pair: ``.leaf`` is ``[B, L]`` with strides ``(1, B')``, ``.root`` ``[B, R]`` with ``(1, B')``
        (``B'``: the mapped sample count).  One window -- the whole batch --, whole root matrices as candidates: pays for batches of up to a
        few tens of GB."""
        return PairedBatch(self, n_sample, device, calibrate, 0, verbose, capi.BATCH_PAIR_LEAF_MAJOR)

    def tile_major_pair(self, n_sample: int, device, calibrate: bool = True, chunk_bytes: int = 0, verbose: bool = False, extra_flags: int = 0) -> "PairedBatch":
        """The leaf and root arrays of a tile-major batch of this function, allocated by the library so that every part of the leaves
        streams next to its part of the roots at the fast rate (``fdg_batch_alloc_pair``: the root chunks are chosen by timing this
        function's own kernel on (leaf window, root chunk) pairs).  ``.leaf`` / ``.root`` are ``[cld(B, 64), L | R, 64]`` tensors viewing
        library-owned memory; call ``.free()`` (or drop the object) when done."""
        return PairedBatch(self, n_sample, device, calibrate, chunk_bytes, verbose, extra_flags)

    @staticmethod
    def tile_major_(dst, src, n_sample: Optional[int] = None):
        """``tile_major!(dst, src)``: a matrix in one of the reference's layouts -- ``src[b, c]`` with any strides: a Julia column-major
        ``B x C`` matrix (torch: ``x.t()`` of a ``[C, B]`` tensor) or ``compile_Python``'s row-major ``[B, C]`` -- into the tile-major
        ``dst[t, c, l]`` (:meth:`tile_major_empty`).  One pass at copy speed (``fdg_repack_tile_major``); returns ``dst``."""
        import torch
        if not (_is_torch(src) and src.is_cuda and src.dtype == torch.float64 and src.dim() == 2):
            raise ValueError("src must be a 2-d float64 CUDA tensor [samples, columns]")
        B = src.shape[0] if n_sample is None else int(n_sample)
        C_ = src.shape[1]
        if dst is None:
            dst = GraphFunc.tile_major_empty(B, C_, src.device)
        if not (_is_torch(dst) and dst.is_cuda and dst.dtype == torch.float64 and dst.dim() == 3 and dst.shape[2] == 64 and dst.shape[1] == C_
                and dst.is_contiguous() and dst.shape[0] >= (B + 63) // 64 and B <= src.shape[0]):
            raise ValueError("dst must be a contiguous float64 CUDA tensor [cld(B, 64), columns, 64]")
        with torch.cuda.device(src.device):
            capi.repack_tile_major(src.data_ptr(), src.stride(0), src.stride(1), dst.data_ptr(), B, C_, torch.cuda.current_stream(src.device).cuda_stream)
        return dst

    @staticmethod
    def from_tile_major_(dst, src, n_sample: Optional[int] = None):
        """``from_tile_major!(dst, src)``: the inverse of :meth:`tile_major_` -- tile-major ``src[t, c, l]`` into the matrix ``dst[b, c]`` (any strides)."""
        import torch
        if not (_is_torch(dst) and dst.is_cuda and dst.dtype == torch.float64 and dst.dim() == 2):
            raise ValueError("dst must be a 2-d float64 CUDA tensor [samples, columns]")
        B = dst.shape[0] if n_sample is None else int(n_sample)
        C_ = dst.shape[1]
        if not (_is_torch(src) and src.is_cuda and src.dtype == torch.float64 and src.dim() == 3 and src.shape[2] == 64 and src.shape[1] == C_
                and src.is_contiguous() and src.shape[0] >= (B + 63) // 64 and B <= dst.shape[0]):
            raise ValueError("src must be a contiguous float64 CUDA tensor [cld(B, 64), columns, 64]")
        with torch.cuda.device(dst.device):
            capi.unpack_tile_major(src.data_ptr(), dst.data_ptr(), dst.stride(0), dst.stride(1), B, C_, torch.cuda.current_stream(dst.device).cuda_stream)
        return dst

    def _check_tiled(self, x, n_col, what):
        import torch
        if not (_is_torch(x) and x.is_cuda and x.dtype == torch.float64 and x.dim() == 3 and x.shape[2] == 64 and x.shape[1] >= n_col):
            raise ValueError(f"{what} must be a float64 CUDA tensor of shape [tiles, >= {n_col}, 64] (tile, value, sample in tile)")

    def eval_tiled(self, root, leaf, n_sample: Optional[int] = None):
        """``root[t, k, l] = root_k(sample 64 t + l)`` from ``leaf[t, i, l]``: both tile-major (see :meth:`tile_major_empty`),
        any strides.  ``n_sample`` defaults to all ``64 * tiles`` samples; lanes past it are neither read nor written."""
        import torch
        self._check_tiled(leaf, self.n_leaf, "leaf")
        T = leaf.shape[0]
        B = 64 * T if n_sample is None else int(n_sample)
        if not (0 <= B <= 64 * T):
            raise ValueError("n_sample exceeds the batch")
        if root is None:
            root = self.tile_major_empty(64 * T, self.n_root, leaf.device)
        self._check_tiled(root, self.n_root, "root")
        if root.shape[0] < (B + 63) // 64 or root.device != leaf.device:
            raise ValueError("root holds fewer tiles than the samples need, or lives on another device")
        st = torch.cuda.current_stream(leaf.device).cuda_stream
        with torch.cuda.device(leaf.device):
            self.handle.eval_device_tiled(leaf.data_ptr(), leaf.stride(2), leaf.stride(1), leaf.stride(0), root.data_ptr(),
                                          root.stride(2), root.stride(1), root.stride(0), B, st)
        return root

    def accumulate_tiled(self, leaf, weight=None, acc=None, n_sample: Optional[int] = None):
        """``acc[k] += sum_b weight[b] * root_k(b)`` over a tile-major leaf batch (``weight``: plain vector indexed by sample)."""
        import torch
        self._check_tiled(leaf, self.n_leaf, "leaf")
        B = 64 * leaf.shape[0] if n_sample is None else int(n_sample)
        if not (0 <= B <= 64 * leaf.shape[0]):
            raise ValueError("n_sample exceeds the batch")
        if acc is None:
            acc = torch.zeros(self.n_root, dtype=torch.float64, device=leaf.device)
        if (not acc.is_cuda or acc.device != leaf.device or acc.dtype != torch.float64 or not acc.is_contiguous()
                or acc.numel() < self.n_root):
            raise ValueError("acc must be a contiguous float64 tensor of at least n_root elements on the leaves' device")
        w = 0
        if weight is not None:
            if (not weight.is_cuda or weight.device != leaf.device or weight.dtype != torch.float64 or weight.dim() != 1
                    or weight.shape[0] < B):
                raise ValueError("weight must be a float64 vector of at least n_sample elements on the leaves' device")
            weight = weight.contiguous()
            w = weight.data_ptr()
        st = torch.cuda.current_stream(leaf.device).cuda_stream
        with torch.cuda.device(leaf.device):
            self.handle.accumulate_device_tiled(leaf.data_ptr(), leaf.stride(2), leaf.stride(1), leaf.stride(0), w, acc.data_ptr(), B, st)
        return acc

    def _call_numpy_typed(self, root, leaf):
        """Host arrays of an element type other than Float64: staged through the device (there is no CPU evaluator behind the ABI)."""
        import torch
        if not torch.cuda.is_available():
            raise capi.FdgError(capi.FDG_E_NO_DEVICE, "no gfx950 device: the evaluator has no CPU fallback")
        vec = leaf.ndim == 1
        d_leaf = torch.from_numpy(np.ascontiguousarray(leaf)).cuda()
        live = [k for k in range(self.n_root) if int(self.table.root_slot[k]) != FDG_NO_ROOT]
        if vec:
            if leaf.shape[0] < self.n_leaf:
                raise IndexError(f"BoundsError: attempt to access {leaf.shape[0]}-element leafVal at index [{self.n_leaf}]")
            if root is None:
                root = np.zeros(self.n_root, dtype=leaf.dtype)
            if len(root) < self.n_root:
                raise IndexError(f"BoundsError: attempt to access {len(root)}-element root at index [{self.n_root}]")
        d_root = self._call_torch_typed(None, d_leaf)
        torch.cuda.synchronize()
        h = d_root.cpu().numpy()
        if vec:
            h = h.reshape(-1)
            for k in live:                   # (root[k] of an id that is in no graph keeps the caller's value, as in the generated function)
                root[k] = h[k]
            return h[max(live, key=lambda k: self._emission_rank(int(self.table.root_slot[k])))] if live else None
        if root is None:
            return h
        root[:, live] = h[:, live]
        return root

    def _call_torch_typed(self, root, leaf):
        """Element types other than Float64: the generated function is generic in ``eltype(leafVal)`` (static.jl:98-133; the
        types ``julia_to_C_typestr`` names, static.jl:135-153).  Float32, ComplexF64 and ComplexF32 tensors go through the
        per-type kernel of ``fdg_graph_specialize_typed`` (compiled on first use); ``root`` has the leaves' type."""
        import torch
        dt = {torch.float32: capi.FDG_DT_F32, torch.complex128: capi.FDG_DT_C64, torch.complex64: capi.FDG_DT_C32}.get(leaf.dtype)
        if dt is None:
            raise TypeError(f"leafVal of element type {leaf.dtype} is not supported (Float64, Float32, ComplexF64, ComplexF32)")
        if not hasattr(self, "_typed_ready"):
            self._typed_ready = set()
        if dt not in self._typed_ready:
            # (FDG_SPEC_ISA: ComplexF64 rows -- compile_Python's row-major [B, L] -- additionally get the graph spelled out on real and
            # imaginary parts through the Float64 assembly back end; the library decides per call which kernel a layout takes)
            isa = self._specialize in ("isa", "isa-autotune", "auto")
            self.handle.specialize_typed(dt, self._cache_dir, capi.FDG_SPEC_ISA if isa else 0)
            self._typed_ready.add(dt)
        squeeze = leaf.dim() == 1
        if squeeze:
            leaf = leaf[None, :]
        B, Lc = leaf.shape
        if Lc < self.n_leaf:
            raise IndexError("BoundsError: leafVal has fewer columns than the graph has leaves")
        if root is None:
            root = torch.zeros((self.n_root,) if squeeze else (B, self.n_root), dtype=leaf.dtype, device=leaf.device)
        r2 = root[None, :] if squeeze else root
        if r2.dim() != 2 or r2.dtype != leaf.dtype or not r2.is_cuda or r2.shape[0] != B or r2.shape[1] < self.n_root:
            raise ValueError("root must be a [B, R] tensor of the leaves' element type on the same device")
        st = torch.cuda.current_stream(leaf.device).cuda_stream
        with torch.cuda.device(leaf.device):
            self.handle.eval_device_typed(dt, leaf.data_ptr(), leaf.stride(0), leaf.stride(1), r2.data_ptr(), r2.stride(0), r2.stride(1), B, st)
        self.last_typed_kernel = self.handle.kernel_info()["last_kernel"]
        return root

    def accumulate(self, leaf, weight=None, acc=None):
        """``acc[k] += sum_b weight[b] * root_k(b)`` on device (torch tensors)."""
        import torch
        if not leaf.is_cuda or leaf.dtype != torch.float64 or leaf.dim() != 2:
            raise TypeError("leaf must be a float64 [B, L] CUDA tensor")
        if leaf.shape[1] < self.n_leaf:
            raise IndexError("BoundsError: leafVal has fewer columns than the graph has leaves")
        if acc is None:
            acc = torch.zeros(self.n_root, dtype=torch.float64, device=leaf.device)
        if (not acc.is_cuda or acc.device != leaf.device or acc.dtype != torch.float64 or not acc.is_contiguous()
                or acc.numel() < self.n_root):
            raise ValueError("acc must be a contiguous float64 tensor of at least n_root elements on the leaves' device")
        w = 0
        if weight is not None:
            if (not weight.is_cuda or weight.device != leaf.device or weight.dtype != torch.float64 or weight.dim() != 1
                    or weight.shape[0] < leaf.shape[0]):
                raise ValueError("weight must be a float64 vector of at least B elements on the leaves' device")
            weight = weight.contiguous()
            w = weight.data_ptr()
        st = torch.cuda.current_stream(leaf.device).cuda_stream
        with torch.cuda.device(leaf.device):
            self.handle.accumulate_device(leaf.data_ptr(), leaf.stride(0), leaf.stride(1), w, acc.data_ptr(),
                                          leaf.shape[0], st)
        return acc


def compile_table(table: NodeTable, specialize: bool = False, **kw) -> GraphFunc:
    return GraphFunc(table, specialize=specialize, **kw)


def compile(graphs: Sequence[Graph], root: Optional[Sequence[int]] = None, specialize="auto",
            **kw) -> Tuple[GraphFunc, Dict[int, Graph]]:
    """``Compilers.compile`` (static.jl:221-227): returns ``(f, leafmap)``."""
    table, leafmap, _ = lower(graphs, root)
    return GraphFunc(table, specialize=specialize, **kw), leafmap


compile_hip = compile


def _append(filename: str, text: str, header: str = "") -> None:
    with open(filename, "a") as f:
        if header and f.tell() == 0:
            f.write(header)
        f.write(text)


def compile_Julia(graphs, filename: str, root=None, func_name: str = "eval_graph!"):
    s, leafmap = to_julia_str(graphs, root=root, name=func_name)
    _append(filename, s)
    return leafmap


def compile_C(graphs, filename: str, datatype="Float64", root=None, func_name: str = "eval_graph"):
    s, leafmap = to_Cstr(graphs, root=root, datatype=datatype, name=func_name)
    _append(filename, s, "#include <math.h>\n")    # static.jl:272-276
    return leafmap


def compile_Python(graphs, filename: str, root=None, func_name: str = "eval_graph"):
    s, leafmap = to_python_str(graphs, root=root, name=func_name)
    _append(filename, s)
    return leafmap


class _DeviceView:
    """A device address as something ``torch.as_tensor`` can wrap without copying (CUDA array interface v2)."""

    def __init__(self, ptr: int, shape):
        self.__cuda_array_interface__ = {"shape": tuple(int(x) for x in shape), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


class PairedBatch:
    """``GraphFunc.tile_major_pair``: tile-major leaves and roots backed by ``fdg_batch_alloc_pair`` (include/fdg.h)."""

    def __init__(self, func, n_sample, device, calibrate=True, chunk_bytes=0, verbose=False, extra_flags=0):
        import torch
        device = torch.device(device)
        if func.handle is None or device.type != "cuda":
            raise capi.FdgError(capi.FDG_E_UNSUPPORTED, "tile_major_pair needs a device handle specialised with the ISA back end and a CUDA device")
        L, R = func.n_leaf, func.n_root
        self.n_sample = int(n_sample)
        with torch.cuda.device(device):
            self._lp, self._rp, self.info = capi.batch_alloc_pair(func.handle, self.n_sample, chunk_bytes, calibrate, verbose, extra_flags)
            T = (self.n_sample + 63) // 64
            if extra_flags & capi.BATCH_PAIR_LEAF_MAJOR:
                Bp = int(self.info["chunk_tiles"]) * 64
                self.leaf = torch.as_tensor(_DeviceView(self._lp, (L, Bp)), device=device).t()[:self.n_sample]
                self.root = torch.as_tensor(_DeviceView(self._rp, (R, Bp)), device=device).t()[:self.n_sample]
            elif extra_flags & capi.BATCH_PAIR_ROW_MAJOR:
                self.leaf = torch.as_tensor(_DeviceView(self._lp, (self.info["leaf_bytes"] // (8 * L), L)), device=device)[:self.n_sample]
                self.root = torch.as_tensor(_DeviceView(self._rp, (self.info["root_bytes"] // (8 * R), R)), device=device)[:self.n_sample]
            else:
                self.leaf = torch.as_tensor(_DeviceView(self._lp, (self.info["leaf_bytes"] // (512 * L), L, 64)), device=device)[:T]
                self.root = torch.as_tensor(_DeviceView(self._rp, (self.info["root_bytes"] // (512 * R), R, 64)), device=device)[:T]
        if self.leaf.data_ptr() != self._lp or self.root.data_ptr() != self._rp:
            self.free()
            raise capi.FdgError(capi.FDG_E_INTERNAL, "torch copied the library's batch instead of viewing it")
        self._device = device

    def free(self):
        import torch
        lp, rp = getattr(self, "_lp", 0), getattr(self, "_rp", 0)
        self._lp = self._rp = 0
        self.leaf = self.root = None
        if lp or rp:
            with torch.cuda.device(self._device) if hasattr(self, "_device") else _nullcontext():
                if lp:
                    capi.batch_free(lp)
                if rp:
                    capi.batch_free(rp)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _nullcontext:
    def __enter__(self): return self
    def __exit__(self, *a): return False
