"""The Julia shim (feynmandiagram.jl_amd/julia/hip_compiler.jl) cannot be executed here -- there is no Julia in the image -- so what CAN be
checked is checked statically: every `ccall` names a symbol that include/fdg.h declares and libfdg.so exports, passes as many arguments as its
type tuple lists and as the C prototype takes, with argument classes that agree (Int64 for int64_t, Ptr/Ref for pointers, Cstring for
const char *, ...); the byte offsets the shim reads out of fdg_kernel_info are the ctypes mirror's; blocks are balanced.  A signature that
drifts on the C side (round 4 added a tile stride to three entry points) fails here instead of in a user's session."""
import ctypes as C
import os
import re

import pytest

from feynmandiagram_jl_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = os.path.join(ROOT, "feynmandiagram.jl_amd", "julia", "hip_compiler.jl")
HDR = os.path.join(ROOT, "include", "fdg.h")


def strip_c_comments(text):
    return re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", text, flags=re.S))


def c_prototypes():
    text = strip_c_comments(open(HDR).read())
    protos = {}
    for m in re.finditer(r"\b(int|void|const\s+char\s*\*)\s*(fdg_\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        params = [p.strip() for p in m.group(3).replace("\n", " ").split(",")]
        if params == ["void"] or params == [""]:
            params = []
        protos[m.group(2)] = (m.group(1).strip(), params)
    return protos


def c_class(param):
    p = re.sub(r"\s+", " ", param)
    if "*" in p:
        return "cstring" if re.match(r"const char \*\s*\w*$", p) else "pointer"
    base = re.sub(r"\b(const|volatile)\b", "", p).split()
    ty = " ".join(base[:-1]) if len(base) > 1 else base[0]
    return {"int64_t": "i64", "uint64_t": "u64", "uint32_t": "u32", "unsigned": "u32", "unsigned int": "u32", "int": "i32", "double": "f64",
            "size_t": "u64", "int32_t": "i32"}.get(ty, "?" + ty)


JL_CLASS = {"Int64": "i64", "UInt64": "u64", "Csize_t": "u64", "UInt32": "u32", "Cuint": "u32", "Cint": "i32", "Int32": "i32", "Cdouble": "f64",
            "Float64": "f64", "Cstring": "cstring"}


def jl_class(ty):
    ty = ty.strip()
    if ty.startswith("Ptr{") or ty.startswith("Ref{"):
        return "pointer"
    return JL_CLASS.get(ty, "?" + ty)


def split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def jl_ccalls():
    text = re.sub(r"#[^\n]*", "", open(JL).read())           # (no string in the shim contains '#')
    calls = []
    for m in re.finditer(r"ccall\(", text):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(text[i], 0)
            i += 1
        parts = split_top(text[m.end():i - 1])
        sym = re.match(r"\(\s*:(\w+)\s*,\s*_libfdg\s*\)", parts[0])
        assert sym, parts[0]
        types = split_top(parts[2].strip()[1:-1]) if parts[2].strip() not in ("()",) else []
        calls.append((sym.group(1), parts[1].strip(), [t for t in types if t], parts[3:]))
    return calls


def test_every_ccall_matches_its_c_prototype(libfdg):
    protos = c_prototypes()
    calls = jl_ccalls()
    assert len(calls) >= 25 and len(protos) >= 40
    for sym, ret, types, args in calls:
        assert sym in protos, f"{sym}: not declared in include/fdg.h"
        assert hasattr(libfdg, sym), f"{sym}: not exported by libfdg.so"
        c_ret, c_params = protos[sym]
        assert len(types) == len(c_params), f"{sym}: the shim lists {len(types)} argument types, the header {len(c_params)} parameters"
        assert len(args) == len(types), f"{sym}: {len(args)} arguments for {len(types)} types"
        assert (ret == "Cstring") == (c_ret.replace(" ", "") == "constchar*"), (sym, ret, c_ret)
        for k, (jt, cp) in enumerate(zip(types, c_params)):
            jc, cc = jl_class(jt), c_class(cp)
            assert not jc.startswith("?") and not cc.startswith("?"), (sym, k, jt, cp)
            ok = jc == cc or (cc == "cstring" and jc == "pointer") or (cc == "pointer" and jc == "cstring")
            assert ok, f"{sym}: argument {k + 1} is `{cp}` in the header and `{jt}` in the shim"


def test_the_entry_points_of_round_4_are_bound():
    bound = {c[0] for c in jl_ccalls()}
    for sym in ("fdg_eval_device_tiled", "fdg_accumulate_device_tiled", "fdg_graph_set_association", "fdg_batch_alloc", "fdg_batch_free",
                "fdg_leaf_eval_device_tiled", "fdg_leaf_eval_device", "fdg_eval_device", "fdg_accumulate_device", "fdg_graph_specialize"):
        assert sym in bound, sym


def test_the_entry_points_of_round_5_are_bound():
    bound = {c[0] for c in jl_ccalls()}
    for sym in ("fdg_batch_alloc_pair", "fdg_graph_set_option"):
        assert sym in bound, sym
    m = re.search(r"info = zeros\(UInt8, (\d+)\)", open(JL).read())
    assert m and int(m.group(1)) >= C.sizeof(capi.BatchPairInfo)          # the report buffer the shim hands to fdg_batch_alloc_pair


def test_kernel_info_offsets_read_by_the_shim():
    """hip_compiler.jl reads has_rm out of the raw fdg_kernel_info bytes (`ki[125:128]`: 1-based, offset 124) into a buffer of 256 bytes."""
    text = open(JL).read()
    m = re.search(r"reinterpret\(UInt32,\s*ki\[(\d+):(\d+)\]\)", text)
    assert m, "the shim no longer reads has_rm this way: update this test"
    lo, hi = int(m.group(1)), int(m.group(2))
    assert hi - lo == 3 and lo - 1 == capi.KernelInfo.has_rm.offset
    size = re.search(r"ki\s*=\s*zeros\(UInt8,\s*(\d+)\)", text) or re.search(r"Vector\{UInt8\}\(undef,\s*(\d+)\)", text)
    assert size and int(size.group(1)) >= C.sizeof(capi.KernelInfo)


def test_blocks_are_balanced():
    text = re.sub(r'"""(.|\n)*?"""', "", open(JL).read())
    text = re.sub(r"#[^\n]*", "", text)
    text = re.sub(r'"(\\.|[^"\\])*"', '""', text)
    # block openers: a keyword that starts a statement, or a trailing `begin` / `do` (comprehensions' `for` / `if` open nothing)
    openers = len(re.findall(r"(?m)^\s*(?:function|if|for|while|let|begin|struct|mutable struct|module|try|macro|quote)\b|\b(?:begin|do)\s*$", text))
    n_end = len(re.findall(r"(?m)^\s*end\b", text))
    assert openers == n_end and openers > 30, (openers, n_end)
    for a, b in ("()", "[]", "{}"):
        assert text.count(a) == text.count(b), (a, text.count(a), text.count(b))
