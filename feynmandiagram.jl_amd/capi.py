"""ctypes binding of the C ABI declared in include/fdg.h (libfdg.so).

This is the reference-side binding a maintainer would add, written for the
Python host (the Julia ``ccall`` twin is julia/hip_compiler.jl).  It is a thin
1:1 wrapper: no arithmetic happens on this side, and there is no fallback --
when the shared library (the HIP extension) is missing, importing any
evaluation entry point raises ``FdgLibraryMissing``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from .nodetable import NodeTable

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FDG_LIBRARY") or os.path.join(_HERE, "lib", "libfdg.so")      # (FDG_LIBRARY: the dev tools point at lib/libfdg_dev.so, `make -C csrc dev`)
KERNEL_CACHE = os.path.join(_HERE, "kernel_cache")

FDG_OK = 0
FDG_E_INVALID, FDG_E_UNSUPPORTED, FDG_E_NO_DEVICE, FDG_E_NOMEM, FDG_E_JIT, FDG_E_INTERNAL = -1, -2, -3, -4, -5, -6
FDG_SPEC_DEFAULT, FDG_SPEC_KEEP_SOURCE, FDG_SPEC_FAST_MATH, FDG_SPEC_ISA, FDG_SPEC_AUTOTUNE = 0, 1, 2, 4, 8
# element types of leafVal / root (include/fdg.h FDG_DT_*); names as the reference's julia_to_C_typestr takes them (static.jl:135-153)
FDG_DT_F64, FDG_DT_F32, FDG_DT_C64, FDG_DT_C32 = 0, 1, 2, 3
DTYPES = {"Float64": FDG_DT_F64, "Float32": FDG_DT_F32, "ComplexF64": FDG_DT_C64, "ComplexF32": FDG_DT_C32}
FDG_SPEC_ROW_MAJOR_COMPANION = 16
FDG_ASSOC_STATIC, FDG_ASSOC_INTERP = 0, 1     # fdg_graph_set_association: static.jl's generated code / eval.jl's interpreter
FDG_TILE_SAMPLES = 64

EXPORTS = [
    "fdg_last_error", "fdg_version", "fdg_graph_create", "fdg_graph_destroy", "fdg_graph_query",
    "fdg_graph_emit_source", "fdg_free", "fdg_graph_specialize", "fdg_eval_device", "fdg_eval",
    "fdg_accumulate_device", "fdg_fill_uniform_device", "fdg_copy_device", "fdg_read_device", "fdg_clock_probe_device", "fdg_graph_specialize_typed", "fdg_eval_device_typed", "fdg_graph_create_complex_view", "fdg_isa_check_hazards", "fdg_graph_release_device", "fdg_powi",
    "fdg_eval_strided", "fdg_graph_coop_program", "fdg_graph_set_opt_params", "fdg_graph_opt_program", "fdg_graph_set_schedule_groups", "fdg_leaf_eval_device", "fdg_leaf_eval_device_tiled",
    "fdg_comm_unique_id", "fdg_comm_create", "fdg_comm_destroy", "fdg_reduce_device",
    "fdg_graph_specialize_fused", "fdg_mc_eval_device", "fdg_mc_accumulate_device", "fdg_graph_mc_program",
    "fdg_graph_kernel_info",
    "fdg_eval_device_tiled", "fdg_accumulate_device_tiled", "fdg_fill_uniform_device_tiled", "fdg_graph_set_association",
    "fdg_batch_alloc", "fdg_batch_free", "fdg_graph_pool_program", "fdg_batch_alloc_pair", "fdg_graph_set_option", "fdg_graph_get_option", "fdg_set_default_option", "fdg_get_default_option", "fdg_selftest_pair_search",
    "fdg_repack_tile_major", "fdg_unpack_tile_major",
]
COMM_ID_BYTES = 128


class FdgLibraryMissing(ImportError):
    pass


class FdgError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"fdg error {code}: {msg}")
        self.code = code


class GraphDesc(C.Structure):
    _fields_ = [("n_leaf", C.c_uint32), ("n_node", C.c_uint32), ("n_root", C.c_uint32), ("n_edge", C.c_uint32),
                ("op", C.POINTER(C.c_uint8)), ("power", C.POINTER(C.c_int32)),
                ("child_off", C.POINTER(C.c_uint32)), ("child_idx", C.POINTER(C.c_uint32)),
                ("child_fac", C.POINTER(C.c_double)), ("root_slot", C.POINTER(C.c_uint32))]


class GraphInfo(C.Structure):
    _fields_ = [("n_leaf", C.c_uint32), ("n_node", C.c_uint32), ("n_root", C.c_uint32), ("n_edge", C.c_uint32),
                ("n_live_node", C.c_uint32), ("n_live_leaf", C.c_uint32),
                ("flops_alg", C.c_uint64), ("bytes_alg", C.c_uint64),
                ("max_live", C.c_uint32), ("n_slot_lds", C.c_uint32), ("n_slot_mem", C.c_uint32),
                ("n_ops", C.c_uint32), ("specialized", C.c_int32),
                ("spec_vgpr", C.c_uint32), ("spec_lds_bytes", C.c_uint32), ("spec_scratch_bytes", C.c_uint32)]

    def asdict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class KernelInfo(C.Structure):
    _fields_ = [("last_kernel", C.c_char * 48), ("n_valu", C.c_uint64 * 3), ("n_ld_leaf", C.c_uint32 * 3),
                ("n_panel", C.c_uint32 * 3), ("n_lds", C.c_uint32 * 3), ("waves_per_cu", C.c_uint32 * 3),
                ("has_acc", C.c_uint32), ("has_rm", C.c_uint32), ("has_coop", C.c_uint32), ("rm_bufs", C.c_uint32),
                ("has_pool", C.c_uint32), ("pool_fetch", C.c_uint32), ("pool_valu", C.c_uint64),
                ("has_rl", C.c_uint32), ("rl_reserved", C.c_uint32), ("rl_valu", C.c_uint64)]


class LeafTables(C.Structure):
    _fields_ = [("n_leaf", C.c_uint32), ("n_basis", C.c_uint32), ("n_loop", C.c_uint32), ("dim", C.c_uint32),
                ("n_tau", C.c_uint32), ("leaf_type", C.c_void_p), ("leaf_order", C.c_void_p), ("tau_in", C.c_void_p),
                ("tau_out", C.c_void_p), ("loop_index", C.c_void_p), ("basis", C.c_void_p),
                ("kF", C.c_double), ("beta", C.c_double), ("lambda_", C.c_double)]


class OptParams(C.Structure):
    _fields_ = [("n_reg", C.c_uint32), ("n_lds", C.c_uint32), ("lookahead_lds", C.c_uint32),
                ("lookahead_mem", C.c_uint32), ("lookahead_leaf", C.c_uint32), ("n_acc", C.c_uint32),
                ("vn_window", C.c_uint32), ("fma", C.c_uint32), ("remat_window", C.c_uint32), ("remat_cost", C.c_uint32)]


class MOp(C.Structure):
    _fields_ = [("kind", C.c_uint8), ("nega", C.c_uint8), ("negb", C.c_uint8), ("negc", C.c_uint8),
                ("d", C.c_uint32), ("a", C.c_uint32), ("b", C.c_uint32), ("imm", C.c_double),
                ("c", C.c_uint32), ("param", C.c_uint32)]


MOP_DTYPE = np.dtype([("kind", "u1"), ("nega", "u1"), ("negb", "u1"), ("negc", "u1"),
                      ("d", "<u4"), ("a", "<u4"), ("b", "<u4"), ("imm", "<f8"), ("c", "<u4"), ("param", "<u4")])

_lib = None


class BatchPairInfo(C.Structure):
    """fdg_batch_pair_info (include/fdg.h)"""
    _fields_ = [("leaf_bytes", C.c_uint64), ("root_bytes", C.c_uint64), ("chunk_tiles", C.c_uint64), ("n_chunk", C.c_uint32),
                ("n_candidate", C.c_uint32), ("n_filler", C.c_uint32), ("n_probe", C.c_uint32), ("n_matched", C.c_uint32),
                ("calibrated", C.c_uint32), ("gbs_fast", C.c_double), ("gbs_slow", C.c_double), ("gbs_before_mean", C.c_double),
                ("gbs_before_min", C.c_double), ("gbs_after_mean", C.c_double), ("gbs_after_min", C.c_double), ("seconds", C.c_double),
                ("seconds_settling", C.c_double), ("level_reached", C.c_uint32), ("span_gb", C.c_uint32)]


def lib():
    """Load libfdg.so once; fail loudly when the HIP extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FdgLibraryMissing(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C feynmandiagram.jl_amd/csrc). "
            "There is no CPU fallback for the evaluator.")
    # PyTorch wheels bundle their own libamdhip64/libhsa-runtime64.  One process
    # must use ONE HIP runtime, or device pointers and streams cannot be shared:
    # when torch is installed, load it first so libfdg.so binds to the runtime
    # torch already mapped (same SONAME) instead of a second copy from /opt/rocm.
    if not os.environ.get("FDG_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    # the code objects and tuned parameters shipped with the package: a read-only secondary lookup (fdg_runtime.hip: read_cached)
    L = C.CDLL(LIB_PATH)
    vp, i64, u64, u32, dp = C.c_void_p, C.c_int64, C.c_uint64, C.c_uint32, C.c_void_p
    L.fdg_last_error.restype = C.c_char_p
    L.fdg_version.restype = C.c_int
    L.fdg_graph_create.argtypes = [C.POINTER(GraphDesc), C.POINTER(vp)]
    L.fdg_graph_destroy.argtypes = [vp]
    L.fdg_graph_query.argtypes = [vp, C.POINTER(GraphInfo)]
    L.fdg_graph_kernel_info.argtypes = [vp, C.POINTER(KernelInfo)]
    L.fdg_graph_emit_source.argtypes = [vp, C.c_uint, C.POINTER(C.c_char_p)]
    L.fdg_free.argtypes = [vp]
    L.fdg_free.restype = None
    L.fdg_graph_specialize.argtypes = [vp, C.c_char_p, C.c_uint]
    L.fdg_eval_device.argtypes = [vp, dp, i64, i64, dp, i64, i64, i64, vp]
    L.fdg_eval.argtypes = [vp, dp, dp, i64]
    L.fdg_graph_set_association.argtypes = [vp, C.c_int]
    L.fdg_batch_alloc.argtypes = [C.c_size_t, C.c_size_t, C.POINTER(vp)]
    L.fdg_batch_free.argtypes = [vp]
    L.fdg_graph_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.fdg_graph_get_option.argtypes = [vp, C.c_char_p]
    L.fdg_graph_get_option.restype = C.c_char_p
    L.fdg_set_default_option.argtypes = [C.c_char_p, C.c_char_p]
    L.fdg_get_default_option.argtypes = [C.c_char_p]
    L.fdg_get_default_option.restype = C.c_char_p
    L.fdg_selftest_pair_search.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    L.fdg_batch_alloc_pair.argtypes = [vp, C.c_int64, C.c_size_t, C.c_uint, C.POINTER(vp), C.POINTER(vp), C.POINTER(BatchPairInfo)]
    L.fdg_eval_device_tiled.argtypes = [vp, dp, i64, i64, i64, dp, i64, i64, i64, i64, vp]
    L.fdg_accumulate_device_tiled.argtypes = [vp, dp, i64, i64, i64, dp, dp, i64, vp]
    L.fdg_fill_uniform_device_tiled.argtypes = [dp, i64, u32, i64, i64, i64, u64, u64, vp]
    L.fdg_eval_strided.argtypes = [vp, dp, i64, i64, dp, i64, i64, i64]
    L.fdg_accumulate_device.argtypes = [vp, dp, i64, i64, dp, dp, i64, vp]
    L.fdg_fill_uniform_device.argtypes = [dp, i64, u32, i64, i64, u64, u64, vp]
    L.fdg_copy_device.argtypes = [dp, dp, i64, vp]
    L.fdg_read_device.argtypes = [dp, i64, dp, vp]
    L.fdg_repack_tile_major.argtypes = [dp, i64, i64, dp, i64, C.c_uint32, vp]
    L.fdg_unpack_tile_major.argtypes = [dp, dp, i64, i64, i64, C.c_uint32, vp]
    L.fdg_clock_probe_device.argtypes = [C.c_double, vp, vp]
    L.fdg_graph_specialize_typed.argtypes = [vp, C.c_int, C.c_char_p, C.c_uint]
    L.fdg_eval_device_typed.argtypes = [vp, C.c_int, vp, i64, i64, vp, i64, i64, i64, vp]
    L.fdg_graph_create_complex_view.argtypes = [vp, C.POINTER(vp)]
    L.fdg_isa_check_hazards.argtypes = [C.c_char_p, C.POINTER(C.c_char_p)]
    L.fdg_graph_release_device.argtypes = [vp]
    L.fdg_leaf_eval_device.argtypes = [C.POINTER(LeafTables), dp, i64, i64, dp, i64, i64, dp, i64, i64, i64, vp]
    L.fdg_leaf_eval_device_tiled.argtypes = [C.POINTER(LeafTables), dp, i64, i64, dp, i64, i64, dp, i64, i64, i64, i64, vp]
    L.fdg_graph_set_schedule_groups.argtypes = [vp, C.c_void_p, u32]
    L.fdg_graph_set_opt_params.argtypes = [vp, C.POINTER(OptParams)]
    L.fdg_graph_opt_program.argtypes = [vp, C.POINTER(OptParams), C.POINTER(C.POINTER(MOp)), C.POINTER(C.c_uint64),
                                        C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
    L.fdg_graph_mc_program.argtypes = [vp, C.POINTER(LeafTables), C.POINTER(OptParams), C.POINTER(C.POINTER(MOp)), C.POINTER(C.c_uint64),
                                       C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
    L.fdg_graph_coop_program.argtypes = [vp, C.POINTER(OptParams), u32, C.POINTER(C.POINTER(MOp)), C.POINTER(C.c_uint64), C.POINTER(u32)]
    L.fdg_graph_pool_program.argtypes = [vp, C.POINTER(OptParams), u32, C.POINTER(C.POINTER(MOp)), C.POINTER(C.c_uint64), C.POINTER(u32)]
    L.fdg_graph_specialize_fused.argtypes = [vp, C.POINTER(LeafTables), C.c_char_p, C.c_uint]
    L.fdg_mc_eval_device.argtypes = [vp, dp, i64, i64, dp, i64, i64, C.c_double, C.c_double, C.c_double, dp, i64, i64, i64, vp]
    L.fdg_mc_accumulate_device.argtypes = [vp, dp, i64, i64, dp, i64, i64, C.c_double, C.c_double, C.c_double, dp, dp, i64, vp]
    L.fdg_comm_unique_id.argtypes = [C.c_void_p, C.c_size_t]
    L.fdg_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(vp)]
    L.fdg_comm_destroy.argtypes = [vp]
    L.fdg_reduce_device.argtypes = [vp, dp, u32, C.c_int, vp]
    L.fdg_powi.argtypes = [C.c_double, C.c_int32]
    L.fdg_powi.restype = C.c_double
    _lib = L
    # the code objects and tuned parameters shipped with the package: a read-only secondary lookup (fdg_runtime.hip: read_cached), appended
    # to what the environment names -- as a process default of the library, not by editing os.environ
    ro = (L.fdg_get_default_option(b"FDG_CACHE_RO_DIR") or b"").decode()
    if KERNEL_CACHE not in ro.split(":"):
        L.fdg_set_default_option(b"FDG_CACHE_RO_DIR", ((ro + ":" if ro else "") + KERNEL_CACHE).encode())
    return L


def check(rc: int):
    if rc != 0:
        raise FdgError(rc, lib().fdg_last_error().decode("utf-8", "replace"))


class GraphHandle:
    """Owns one ``fdg_graph*``."""

    def __init__(self, table: NodeTable):
        t = table.normalized()
        t.validate()
        self.table = t
        d = GraphDesc()
        d.n_leaf, d.n_node, d.n_root, d.n_edge = t.n_leaf, t.n_node, t.n_root, t.n_edge
        d.op = t.op.ctypes.data_as(C.POINTER(C.c_uint8))
        d.power = t.power.ctypes.data_as(C.POINTER(C.c_int32))
        d.child_off = t.child_off.ctypes.data_as(C.POINTER(C.c_uint32))
        d.child_idx = t.child_idx.ctypes.data_as(C.POINTER(C.c_uint32))
        d.child_fac = t.child_fac.ctypes.data_as(C.POINTER(C.c_double))
        d.root_slot = t.root_slot.ctypes.data_as(C.POINTER(C.c_uint32))
        h = C.c_void_p()
        check(lib().fdg_graph_create(C.byref(d), C.byref(h)))
        self._h = h
        if getattr(t, "sched_group", None) is not None:
            self.set_schedule_groups(t.sched_group)

    def complex_view(self) -> "GraphHandle":
        """A new handle for this graph on Complex{Float64} values spelled out on real and imaginary parts (fdg_graph_create_complex_view):
        a Float64 graph with 2 L leaves and 2 R roots.  ``table`` of the result is the host mirror's statement of the same construction."""
        from .nodetable import complex_to_real
        h = C.c_void_p()
        check(lib().fdg_graph_create_complex_view(self._h, C.byref(h)))
        v = GraphHandle.__new__(GraphHandle)
        v.table = complex_to_real(self.table)
        v._h = h
        return v

    def set_option(self, name: str, value=None):
        """fdg_graph_set_option: one option of this handle (``None`` removes it).  Options replace the FDG_* environment switches: the
        library reads the environment once per process, for the supported names only."""
        check(lib().fdg_graph_set_option(self._h, name.encode(), None if value is None else str(value).encode()))

    def set_options(self, options):
        for k, v in (options or {}).items():
            self.set_option(k, v)

    def get_option(self, name: str):
        v = lib().fdg_graph_get_option(self._h, name.encode())
        return None if v is None else v.decode()

    def set_association(self, assoc: int):
        """FDG_ASSOC_STATIC (the generated code, static.jl) or FDG_ASSOC_INTERP (eval!, eval.jl); before any specialisation."""
        check(lib().fdg_graph_set_association(self._h, assoc))

    def set_schedule_groups(self, group):
        if group is None:
            check(lib().fdg_graph_set_schedule_groups(self._h, None, 0))
            return
        g = np.ascontiguousarray(group, dtype=np.uint32)
        check(lib().fdg_graph_set_schedule_groups(self._h, g.ctypes.data, g.shape[0]))

    def close(self):
        if getattr(self, "_h", None):
            lib().fdg_graph_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def ptr(self):
        return self._h

    def info(self) -> dict:
        gi = GraphInfo()
        check(lib().fdg_graph_query(self._h, C.byref(gi)))
        return gi.asdict()

    def kernel_info(self) -> dict:
        """What the ISA kernels of this handle execute per evaluation (slot 0 evaluator, 1 accumulate, 2 row-major) and the
        name of the kernel the last device call launched."""
        ki = KernelInfo()
        check(lib().fdg_graph_kernel_info(self._h, C.byref(ki)))
        out = {"last_kernel": ki.last_kernel.decode()}
        for k in ("n_valu", "n_ld_leaf", "n_panel", "n_lds", "waves_per_cu"):
            out[k] = [int(x) for x in getattr(ki, k)]
        for k in ("has_acc", "has_rm", "has_coop", "rm_bufs", "has_pool", "pool_fetch", "pool_valu", "has_rl", "rl_valu"):
            out[k] = int(getattr(ki, k))
        return out

    def emit_source(self, flags: int = 0) -> str:
        s = C.c_char_p()
        check(lib().fdg_graph_emit_source(self._h, flags, C.byref(s)))
        try:
            return s.value.decode()
        finally:
            lib().fdg_free(s)

    def set_opt_params(self, n_reg=0, n_lds=0, lookahead_lds=0, lookahead_mem=0, lookahead_leaf=0, n_acc=0, vn_window=0, fma=0, remat_window=0, remat_cost=0):
        q = OptParams(n_reg, n_lds, lookahead_lds, lookahead_mem, lookahead_leaf, n_acc, vn_window, fma, remat_window, remat_cost)
        check(lib().fdg_graph_set_opt_params(self._h, C.byref(q)))

    def opt_program(self, n_reg=0, n_lds=0, lookahead_lds=0, lookahead_mem=0, lookahead_leaf=0, n_acc=0, vn_window=0, fma=0, remat_window=0, remat_cost=0):
        """Returns ``(ops, n_reg_used, n_lds_used, n_mem_used)``; ops is a numpy record array (MOP_DTYPE)."""
        q = OptParams(n_reg, n_lds, lookahead_lds, lookahead_mem, lookahead_leaf, n_acc, vn_window, fma, remat_window, remat_cost)
        ops = C.POINTER(MOp)()
        n = C.c_uint64()
        nr, nl, nm, na = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        check(lib().fdg_graph_opt_program(self._h, C.byref(q), C.byref(ops), C.byref(n), C.byref(nr), C.byref(nl),
                                          C.byref(nm), C.byref(na)))
        try:
            buf = C.string_at(ops, n.value * C.sizeof(MOp))
            arr = np.frombuffer(buf, dtype=MOP_DTYPE).copy()
        finally:
            lib().fdg_free(ops)
        self.last_n_acc = na.value
        return arr, nr.value, nl.value, nm.value

    def mc_program(self, tables, n_reg=0, n_lds=0, lookahead_lds=0, lookahead_mem=0, lookahead_leaf=0, n_acc=0, vn_window=0, fma=0, remat_window=0, remat_cost=0):
        """The program of the fused ISA step (leaves computed from the input columns K components, then times;
        ``tables`` from make_leaf_tables with kF, beta, lam set).  Returns like :meth:`opt_program`."""
        q = OptParams(n_reg, n_lds, lookahead_lds, lookahead_mem, lookahead_leaf, n_acc, vn_window, fma, remat_window, remat_cost)
        ops = C.POINTER(MOp)()
        n = C.c_uint64()
        nr, nl, nm, na = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        check(lib().fdg_graph_mc_program(self._h, C.byref(tables), C.byref(q), C.byref(ops), C.byref(n), C.byref(nr),
                                         C.byref(nl), C.byref(nm), C.byref(na)))
        try:
            arr = np.frombuffer(C.string_at(ops, n.value * C.sizeof(MOp)), dtype=MOP_DTYPE).copy()
        finally:
            lib().fdg_free(ops)
        return arr, nr.value, nl.value, nm.value

    def pool_program(self, **kw):
        """The per-wave programs of the pooled cooperative variant (fdg_graph_pool_program): ``([ops_w0, ..], info)`` like
        :meth:`coop_program`; ``info["n_transfer"]`` is the number of leaf fetches per tile."""
        return self.coop_program(_pooled=True, **kw)

    def coop_program(self, _pooled=False, **kw):
        """The four per-wave programs of the cooperative variant: ``([ops_w0, .., ops_w3], info)`` with
        ``info = dict(n_reg, n_lds, n_mem, n_acc per wave; n_shared, n_epoch, n_transfer, n_duplicate)``."""
        q = OptParams(kw.get("n_reg", 0), kw.get("n_lds", 0), kw.get("lookahead_lds", 0), kw.get("lookahead_mem", 0),
                      kw.get("lookahead_leaf", 0), kw.get("n_acc", 0), kw.get("vn_window", 0), 0, 0, 0)
        progs, per_wave = [], []
        info = None
        for w in range(16):         # four waves, or 8 / 16 with FDG_COOP_WAVES
            ops = C.POINTER(MOp)()
            n = C.c_uint64()
            inf = (C.c_uint32 * 8)()
            rc = (lib().fdg_graph_pool_program if _pooled else lib().fdg_graph_coop_program)(self._h, C.byref(q), w, C.byref(ops), C.byref(n), inf)
            if rc != 0 and w >= 4:
                break
            check(rc)
            try:
                progs.append(np.frombuffer(C.string_at(ops, n.value * C.sizeof(MOp)), dtype=MOP_DTYPE).copy())
            finally:
                lib().fdg_free(ops)
            per_wave.append(dict(n_reg=inf[0], n_lds=inf[1], n_mem=inf[2], n_acc=inf[3]))
            info = dict(n_shared=inf[4], n_epoch=inf[5], n_transfer=inf[6], n_duplicate=inf[7], waves=per_wave)
        return progs, info

    def specialize(self, cache_dir: Optional[str] = None, flags: int = 0):
        """``cache_dir`` None: the library's per-user cache ($FDG_CACHE_DIR, $XDG_CACHE_HOME/fdg, ~/.cache/fdg); the
        kernel_cache directory shipped inside the package is only *read* (``FDG_CACHE_RO_DIR``, set in :func:`lib`), so a
        root-owned or read-only installation works.  ``__graft_entry__.build()`` passes ``KERNEL_CACHE`` to fill it."""
        check(lib().fdg_graph_specialize(self._h, cache_dir.encode() if cache_dir else None, flags))

    def specialize_typed(self, dtype: int, cache_dir: Optional[str] = None, flags: int = 0):
        """The per-graph kernel for an element type other than Float64 (FDG_DT_*)."""
        check(lib().fdg_graph_specialize_typed(self._h, dtype, cache_dir.encode() if cache_dir else None, flags))

    def eval_device_typed(self, dtype: int, d_leaf: int, ss: int, ls: int, d_root: int, rs: int, rk: int, B: int, stream: int = 0):
        check(lib().fdg_eval_device_typed(self._h, dtype, d_leaf, ss, ls, d_root, rs, rk, B, stream))

    # raw-pointer device entry points (ints are device addresses) ----------- #
    def eval_device(self, d_leaf: int, ss: int, ls: int, d_root: int, rs: int, rk: int, B: int, stream: int = 0):
        check(lib().fdg_eval_device(self._h, d_leaf, ss, ls, d_root, rs, rk, B, stream))

    def accumulate_device(self, d_leaf: int, ss: int, ls: int, d_weight: int, d_acc: int, B: int, stream: int = 0):
        check(lib().fdg_accumulate_device(self._h, d_leaf, ss, ls, d_weight or None, d_acc, B, stream))

    # tile-major batches: sample b at (b // 64) * tile stride + (b % 64) * sample stride (fdg.h) ------------------- #
    def eval_device_tiled(self, d_leaf: int, ss: int, ls: int, lts: int, d_root: int, rs: int, rk: int, rts: int, B: int, stream: int = 0):
        check(lib().fdg_eval_device_tiled(self._h, d_leaf, ss, ls, lts, d_root, rs, rk, rts, B, stream))

    def accumulate_device_tiled(self, d_leaf: int, ss: int, ls: int, lts: int, d_weight: int, d_acc: int, B: int, stream: int = 0):
        check(lib().fdg_accumulate_device_tiled(self._h, d_leaf, ss, ls, lts, d_weight or None, d_acc, B, stream))

    # fused Monte-Carlo step: leaves from (K, T) in registers, then the graph --------------------- #
    def specialize_fused(self, tables, cache_dir: Optional[str] = None, flags: int = 0):
        """``tables`` = the struct returned by make_leaf_tables."""
        check(lib().fdg_graph_specialize_fused(self._h, C.byref(tables), cache_dir.encode() if cache_dir else None, flags))

    def mc_eval_device(self, d_K, ks, kc, d_T, ts, tc, kF, beta, lam, d_root, rs, rk, B, stream=0):
        check(lib().fdg_mc_eval_device(self._h, d_K, ks, kc, d_T, ts, tc, kF, beta, lam, d_root, rs, rk, B, stream))

    def mc_accumulate_device(self, d_K, ks, kc, d_T, ts, tc, kF, beta, lam, d_weight, d_acc, B, stream=0):
        check(lib().fdg_mc_accumulate_device(self._h, d_K, ks, kc, d_T, ts, tc, kF, beta, lam, d_weight or None, d_acc, B, stream))

    def eval_host(self, leaf: np.ndarray, root: Optional[np.ndarray] = None) -> np.ndarray:
        """Host matrices ``leaf [B, >= L]`` -> ``root [B, R]``, each C-ordered (compile_Python's row-major layout) or
        Fortran-ordered (what a Julia ``Matrix`` is); no transposition copy is made for either (fdg_eval_strided)."""
        leaf = np.asarray(leaf, dtype=np.float64)
        if leaf.ndim != 2 or leaf.shape[1] < self.table.n_leaf:
            raise IndexError("BoundsError: leafVal has fewer columns than the graph has leaves")
        if not (leaf.flags.c_contiguous or leaf.flags.f_contiguous):
            leaf = np.ascontiguousarray(leaf)
        B = leaf.shape[0]
        if root is None:
            root = np.zeros((B, self.table.n_root), dtype=np.float64, order="C" if leaf.flags.c_contiguous else "F")
        if root.dtype != np.float64 or not (root.flags.c_contiguous or root.flags.f_contiguous) or root.shape != (B, self.table.n_root):
            raise ValueError("root must be a contiguous (C- or Fortran-ordered) float64 [B, R] array")
        ss, ls = (leaf.shape[1], 1) if leaf.flags.c_contiguous else (1, B)
        rs, rk = (self.table.n_root, 1) if root.flags.c_contiguous else (1, B)
        check(lib().fdg_eval_strided(self._h, leaf.ctypes.data, ss, ls, root.ctypes.data, rs, rk, B))
        return root

    def release_device(self):
        check(lib().fdg_graph_release_device(self._h))


def fill_uniform_device(d_leaf: int, B: int, L: int, ss: int, ls: int, seed: int, sample_offset: int = 0,
                        stream: int = 0):
    check(lib().fdg_fill_uniform_device(d_leaf, B, L, ss, ls, seed, sample_offset, stream))


def fill_uniform_device_tiled(d_leaf: int, B: int, L: int, ss: int, ls: int, lts: int, seed: int, sample_offset: int = 0,
                               stream: int = 0):
    check(lib().fdg_fill_uniform_device_tiled(d_leaf, B, L, ss, ls, lts, seed, sample_offset, stream))


def set_default_option(name: str, value=None):
    """fdg_set_default_option: the option every handle created from now on starts with (``None`` removes it); also what the entry points
    without a handle see.  The library does not look at ``os.environ`` after its first use."""
    check(lib().fdg_set_default_option(name.encode(), None if value is None else str(value).encode()))


def get_default_option(name: str):
    v = lib().fdg_get_default_option(name.encode())
    return None if v is None else v.decode()


def batch_alloc(n_bytes: int, chunk_bytes: int = 0) -> int:
    """Device address of a batch backed by physical chunks of ``chunk_bytes`` (0: one allocation); :func:`batch_free` releases it."""
    p = C.c_void_p()
    check(lib().fdg_batch_alloc(n_bytes, chunk_bytes, C.byref(p)))
    return int(p.value)


def batch_free(ptr: int):
    check(lib().fdg_batch_free(ptr))


BATCH_PAIR_CALIBRATE = 1
BATCH_PAIR_ROW_MAJOR = 8
BATCH_PAIR_LEAF_MAJOR = 16


def batch_alloc_pair(handle: "GraphHandle", n_sample: int, chunk_bytes: int = 0, calibrate: bool = True, verbose: bool = False, extra_flags: int = 0):
    """fdg_batch_alloc_pair: device addresses (leaf, root) of a tile-major batch of ``handle`` whose root chunks were chosen by timing the
    handle's evaluator on (leaf chunk, root chunk) pairs, and the report as a dict.  Release both with :func:`batch_free`."""
    pl, pr, info = C.c_void_p(), C.c_void_p(), BatchPairInfo()
    check(lib().fdg_batch_alloc_pair(handle.ptr, n_sample, chunk_bytes, (BATCH_PAIR_CALIBRATE if calibrate else 0) | (2 if verbose else 0) | extra_flags, C.byref(pl), C.byref(pr), C.byref(info)))
    return int(pl.value), int(pr.value), {k: getattr(info, k) for k, _ in BatchPairInfo._fields_}


def isa_check_hazards(asm_text: str):
    """(number of violations, report) of a gfx950 listing against the emitter's wait-state table."""
    rep = C.c_char_p()
    n = lib().fdg_isa_check_hazards(asm_text.encode(), C.byref(rep))
    try:
        text = rep.value.decode() if rep.value else ""
    finally:
        lib().fdg_free(rep)
    if n < 0:
        check(n)
    return n, text


def copy_device(d_dst: int, d_src: int, n: int, stream: int = 0):
    check(lib().fdg_copy_device(d_dst, d_src, n, stream))


def repack_tile_major(d_src: int, ss: int, cs: int, d_tiled: int, n_sample: int, n_col: int, stream: int = 0):
    """fdg_repack_tile_major: matrix m[b*ss + c*cs] -> tile-major (64, n_col, cld(n_sample, 64))."""
    check(lib().fdg_repack_tile_major(d_src, ss, cs, d_tiled, n_sample, n_col, stream))


def unpack_tile_major(d_tiled: int, d_dst: int, ss: int, cs: int, n_sample: int, n_col: int, stream: int = 0):
    """fdg_unpack_tile_major: tile-major (64, n_col, cld(n_sample, 64)) -> matrix m[b*ss + c*cs]."""
    check(lib().fdg_unpack_tile_major(d_tiled, d_dst, ss, cs, n_sample, n_col, stream))


def read_device(d_src: int, n: int, d_sink: int, stream: int = 0):
    """Harness: a non-temporal read-only stream over n doubles (the memory system's ceiling for reads)."""
    check(lib().fdg_read_device(d_src, n, d_sink, stream))


def clock_probe_device(seconds: float, d_ticks: int, stream: int):
    """One sleeping wave on ``stream`` for ``seconds``; afterwards ``d_ticks[0] / d_ticks[1] * 0.1`` is the shader clock in GHz
    the chip sustained meanwhile (launch it on a side stream next to the kernels of interest)."""
    check(lib().fdg_clock_probe_device(seconds, d_ticks, stream))


def powi(x: float, n: int) -> float:
    return float(lib().fdg_powi(x, n))


class Comm:
    """One RCCL communicator per process/GPU for the single reduction of the observable (fdg.h, multi-GPU).
    ``Comm.unique_id()`` on rank 0, ship the 128 bytes to the other ranks, ``Comm(id, rank, world)`` everywhere
    (with the rank's device current), then ``reduce(d_acc_ptr, n)``."""

    def __init__(self, unique_id: bytes, rank: int, world: int):
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError("unique id must be %d bytes" % COMM_ID_BYTES)
        h = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), COMM_ID_BYTES)
        check(lib().fdg_comm_create(buf, rank, world, C.byref(h)))
        self._h, self.rank, self.world = h, rank, world

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(COMM_ID_BYTES)
        check(lib().fdg_comm_unique_id(buf, COMM_ID_BYTES))
        return buf.raw

    def reduce(self, d_acc: int, n: int, root: int = -1, stream: int = 0):
        check(lib().fdg_reduce_device(self._h, d_acc, n, root, stream))

    def close(self):
        if self._h:
            lib().fdg_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_leaf_tables(leaf_type, leaf_order, tau_in, tau_out, loop_index, basis, dim, n_tau, kF=0.0, beta=0.0, lam=0.0):
    """``fdg_leaf_tables`` from the vectors of ``FrontEnds.leafstates`` (1-based indices); returns the
    struct and the arrays it points into (keep them alive)."""
    a = [np.ascontiguousarray(x, dtype=np.int32) for x in (leaf_type, leaf_order, tau_in, tau_out, loop_index)]
    bs = np.ascontiguousarray(basis, dtype=np.float64)
    t = LeafTables()
    t.n_leaf, t.n_basis, t.n_loop, t.dim, t.n_tau = a[0].shape[0], bs.shape[0], bs.shape[1], dim, n_tau
    t.leaf_type, t.leaf_order, t.tau_in, t.tau_out, t.loop_index = [x.ctypes.data for x in a]
    t.basis = bs.ctypes.data
    t.kF, t.beta, t.lambda_ = kF, beta, lam
    return t, (a, bs)


def leaf_eval_device(leaf_type, leaf_order, tau_in, tau_out, loop_index, basis, dim, n_tau, kF, beta, lam,
                     d_K: int, ks: int, kc: int, d_T: int, ts: int, tc: int, d_leaf: int, ss: int, ls: int, B: int,
                     stream: int = 0):
    """fdg_leaf_eval_device with the tables of ``FrontEnds.leafstates`` (1-based indices)."""
    t, _keep = make_leaf_tables(leaf_type, leaf_order, tau_in, tau_out, loop_index, basis, dim, n_tau, kF, beta, lam)
    check(lib().fdg_leaf_eval_device(C.byref(t), d_K, ks, kc, d_T, ts, tc, d_leaf, ss, ls, B, stream))


def leaf_eval_device_tiled(leaf_type, leaf_order, tau_in, tau_out, loop_index, basis, dim, n_tau, kF, beta, lam,
                           d_K: int, ks: int, kc: int, d_T: int, ts: int, tc: int, d_leaf: int, ss: int, ls: int, lts: int, B: int,
                           stream: int = 0):
    """fdg_leaf_eval_device_tiled: the leaves of a tile-major batch (sample b of leaf i at ``(b // 64) * lts + (b % 64) * ss + i * ls``)."""
    t, _keep = make_leaf_tables(leaf_type, leaf_order, tau_in, tau_out, loop_index, basis, dim, n_tau, kF, beta, lam)
    check(lib().fdg_leaf_eval_device_tiled(C.byref(t), d_K, ks, kc, d_T, ts, tc, d_leaf, ss, ls, lts, B, stream))
