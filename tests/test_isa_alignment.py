"""Round 5: no 8-byte instruction of a printed kernel straddles a 64-byte line of the instruction stream (a lone wave pays about two issue
slots for each: tools/ubench/valu_align.hip, profiles/r05_ubench_valu_align.txt).  The printer counts encoded bytes itself
(csrc/fdg_isa.cpp: isa_size); this test holds that count against the assembler: the listing is assembled and disassembled, and every
instruction's real address and size are checked.  Without the pads (option FDG_ISA_ALIGN=0) the same listings do straddle, and the two
listings differ by `s_nop 0` lines only."""
import glob
import os
import re
import subprocess

import pytest

import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import capi, workloads

LLVM = "/opt/rocm/lib/llvm/bin"
pytestmark = pytest.mark.skipif(not os.path.exists(LLVM + "/llvm-objdump"), reason="no llvm-objdump")


def listing(name, tmp, align):
    d = os.path.join(str(tmp), f"{name}_{align}")
    os.makedirs(d)
    fd.compile_table(workloads.get(name), specialize="isa", cache_dir=d, flags=capi.FDG_SPEC_KEEP_SOURCE, options={"FDG_ISA_ALIGN": align})
    (src,) = glob.glob(d + "/fdg_isa_*.s")
    return src


def layout(src):
    """{kernel: [(mnemonic, address, size)]} from the assembler's own view of the listing."""
    obj = src[:-2] + ".o"
    subprocess.run([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", src, "-o", obj], check=True)
    out = subprocess.run([LLVM + "/llvm-objdump", "-d", obj], capture_output=True, text=True, check=True).stdout
    kernels, cur = {}, None
    for ln in out.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\w+)>:", ln)
        if m:
            cur = kernels.setdefault(m.group(1), [])
            continue
        m = re.match(r"^\s+(\S+).*//\s*([0-9A-F]+):\s*((?:[0-9A-F]{8}\s*)+)$", ln)
        if m and cur is not None:
            cur.append((m.group(1), int(m.group(2), 16), 4 * len(m.group(3).split())))
    return kernels


def straddlers(insts):
    return [(mn, a) for mn, a, sz in insts if sz >= 8 and a // 64 != (a + sz - 1) // 64]


@pytest.mark.parametrize("name", ["sigma2", "parquet_sigma4", "gv_sigma4", "parquet_sigma4_dyn"])
def test_no_instruction_straddles_a_line(name, tmp_path):
    padded, plain = layout(listing(name, tmp_path, "1")), layout(listing(name, tmp_path, "0"))
    assert set(padded) == set(plain) and any(k.endswith("_nt") for k in padded)
    n_plain = 0
    for k, insts in padded.items():
        if k.endswith("_coop") or k.endswith("_pool"):
            continue                                     # (their own test below: the pooled kernels are padded in front of vector instructions only since round 6, the value-passing ones not at all)
        assert len(insts) > 20 and all(sz in (4, 8) for _, _, sz in insts), k
        assert straddlers(insts) == [], (name, k)
        n_plain += len(straddlers(plain[k]))
        # the pads are the only difference (the assembler's own fill after the last s_endpgm aside)
        strip = lambda L: [mn for mn, _, _ in L if mn != "s_nop"]
        assert strip(insts) == strip(plain[k]), (name, k)
    if name != "sigma2":
        assert n_plain > 10, "the unpadded listings were expected to straddle: is the check blind?"


def test_the_pads_count_as_wait_states_and_the_listing_passes_the_hazard_table(tmp_path):
    text = open(listing("parquet_sigma4", tmp_path, "1")).read()
    rc, report = capi.isa_check_hazards(text)
    assert rc == 0, report[:2000]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["parquet_sigma4", "parquet_sigma5"])
def test_padded_and_unpadded_kernels_give_the_oracles_bits(libfdg, cuda, name):
    """The pads are wait states, nothing else: both listings against the oracle on the device (evaluation and fused accumulation)."""
    import numpy as np
    import torch
    import oracle
    t = workloads.get(name)
    L, R, B = t.n_leaf, t.n_root, 4099
    h_leaf = oracle.philox_uniform(B, L, 91)
    want = oracle.eval_static(t, h_leaf)
    w = oracle.philox_uniform(B, 1, 92)[:, 0]
    leaf = torch.from_numpy(np.ascontiguousarray(h_leaf.T)).to(cuda).t()
    for align in ("1", "0"):
        f = fd.compile_table(t, specialize="isa", options={"FDG_ISA_ALIGN": align})
        root = torch.zeros((R, B), dtype=torch.float64, device=cuda).t()
        f(root, leaf)
        torch.cuda.synchronize()
        assert f.kernel_info()["last_kernel"].startswith("fdg_isa_eval"), align
        assert np.array_equal(root.cpu().numpy(), want), (name, align)
        acc = f.accumulate(leaf, torch.from_numpy(w).to(cuda)).cpu().numpy()
        assert f.kernel_info()["last_kernel"].startswith("fdg_isa_eval_acc"), align
        terms = want * w[:, None]
        assert np.all(np.abs(acc - terms.sum(0)) <= 1e-12 * np.abs(terms).sum(0)), (name, align)


@pytest.mark.parametrize("name,opts", [("parquet_ver4_3", {"FDG_ISA_POOL": "1", "FDG_COOP_ALIGN": "1"}), ("sigma4_standin", {"FDG_COOP_ALIGN": "1"}),
                                       ("gv_sigma4_taylor2", {}), ("parquet_sigma3", {})])
def test_the_assembler_agrees_with_the_printers_offsets(name, opts, tmp_path):
    """Round 6: the printer's running offset (Emit::off) held against the ASSEMBLER's location counter every 32 instructions, in every
    kernel family -- plain / streaming / accumulating, row-major (rm, rl), cooperative and pooled (option FDG_ISA_CHECK_OFF: an `.if (. - kernel) !=
    off / .error` in the listing).  The pooled kernels' prologue printed `s_mov_b32 s, 0xffffffff`, which the assembler encodes as the inline
    constant -1 and isa_size counted as a literal: every pad behind it sat four bytes off, which is why the pads cost those kernels 12 % in
    round 5 instead of helping."""
    d = os.path.join(str(tmp_path), name)
    os.makedirs(d)
    fd.compile_table(workloads.get(name), specialize="isa", cache_dir=d, flags=capi.FDG_SPEC_KEEP_SOURCE, options=dict(opts, FDG_ISA_CHECK_OFF="1"))      # assembling fails on a mismatch
    (src,) = glob.glob(d + "/fdg_isa_*.s")
    text = open(src).read()
    assert text.count("Emit::off is wrong") > 20
    if opts.get("FDG_COOP_ALIGN"):
        kernels = layout(src)
        k = [x for x in kernels if x.endswith("_pool") or x.endswith("_coop")]
        assert k and all(straddlers(kernels[x]) == [] for x in k), name
