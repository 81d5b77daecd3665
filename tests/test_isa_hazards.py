"""The ISA back end emits hand-written gfx950 assembly, so nothing but its own table (csrc/fdg_isa.cpp, kHazards) keeps
dependent instructions the required number of wait states apart.  Here every prebuilt kernel is assembled with
FDG_SPEC_KEEP_SOURCE and its listing is re-parsed against that table by fdg_isa_check_hazards; hand-written listings
that break each rule prove the checker sees them.  CPU only (the assembler runs without a GPU)."""
import glob
import os

import pytest

import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import capi, workloads


def _listings(d):
    return sorted(glob.glob(os.path.join(str(d), "*.s")))


@pytest.mark.parametrize("name", ["sigma2", "synthetic_small", "gv_sigma4", "gv_sigma5", "gv_sigma4_taylor2", "sigma4_standin"])
def test_prebuilt_kernels_have_no_hazard(name, tmp_path):
    t = workloads.get(name)
    h = capi.GraphHandle(t)
    if t.sched_group is not None:
        h.set_schedule_groups(t.sched_group)
    h.specialize(str(tmp_path), capi.FDG_SPEC_ISA | capi.FDG_SPEC_KEEP_SOURCE)
    files = _listings(tmp_path)
    assert files, "FDG_SPEC_KEEP_SOURCE left no listing"
    for f in files:
        n, rep = capi.isa_check_hazards(open(f).read())
        assert n == 0, rep
        assert "checked" in rep and " 0 violation(s)" in rep


def test_monte_carlo_kernels_have_no_hazard(tmp_path):
    """The one-kernel Monte-Carlo step is where the hazards live: v_rcp_f64 feeding Newton steps, v_cmp feeding v_cndmask."""
    for name in ("gv_sigma4", "gv_sigma4_taylor2"):
        t, z = workloads.get(name), workloads.leafstates(name)
        tab, _keep = capi.make_leaf_tables(z["leaf_type"], z["leaf_order"], z["tau_in"], z["tau_out"], z["loop_index"], z["basis"], 3, int(z["n_tau"]))
        d = tmp_path / name
        d.mkdir(mode=0o700)
        h = capi.GraphHandle(t)
        if t.sched_group is not None:
            h.set_schedule_groups(t.sched_group)
        h.specialize(str(d), capi.FDG_SPEC_ISA | capi.FDG_SPEC_KEEP_SOURCE)
        h.specialize_fused(tab, str(d), capi.FDG_SPEC_KEEP_SOURCE)
        seen_trans = False
        for f in _listings(d):
            text = open(f).read()
            n, rep = capi.isa_check_hazards(text)
            assert n == 0, rep
            seen_trans = seen_trans or "v_rcp_f64" in text
        assert seen_trans, "no listing of the Monte-Carlo kernel was kept"


def test_two_samples_per_lane_variant_has_no_hazard(tmp_path, monkeypatch, fdgopt):
    """FDG_ISA_W2: every access is 16 bytes per lane (wide panel / LDS stores whose data registers the next VALU op may overwrite)."""
    fdgopt.set("FDG_ISA_W2", "1")
    h = capi.GraphHandle(workloads.get("sigma2"))
    h.specialize(str(tmp_path), capi.FDG_SPEC_ISA | capi.FDG_SPEC_KEEP_SOURCE)
    texts = [open(f).read() for f in _listings(tmp_path)]
    assert any("global_load_dwordx4" in x for x in texts)      # the wide kernel is in the code object
    for x in texts:
        n, rep = capi.isa_check_hazards(x)
        assert n == 0, rep


BAD = {
    "trans -> VALU": ("v_rcp_f64_e64 v[2:3], v[4:5]\nv_fma_f64 v[6:7], -v[4:5], v[2:3], 1.0\n", 1),
    "trans -> VALU, one wait state is not enough": ("v_rcp_f64_e64 v[2:3], v[4:5]\ns_nop 0\nv_mul_f64 v[6:7], v[2:3], v[2:3]\n", 1),
    "vcc -> cndmask": ("v_cmp_gt_f64_e64 vcc, v[2:3], 0\nv_cndmask_b32_e32 v6, v8, v9, vcc\n", 1),
    "sgpr mask -> cndmask": ("v_cmp_gt_f64_e64 s[20:21], v[2:3], 0\nv_mov_b32_e32 v7, v1\nv_cndmask_b32_e64 v6, v8, v9, s[20:21]\n", 1),
    "vcc -> div_fmas": ("v_div_scale_f64 v[10:11], vcc, 1.0, v[2:3], 1.0\nv_mul_f64 v[8:9], v[10:11], v[6:7]\nv_div_fmas_f64 v[4:5], v[4:5], v[6:7], v[8:9]\n", 1),
    "VALU sgpr -> VMEM": ("v_readfirstlane_b32 s20, v2\ns_nop 3\nglobal_load_dwordx2 v[4:5], v1, s[20:21]\n", 1),
    "VALU -> readfirstlane": ("v_add_u32_e32 v2, v3, v4\nv_readfirstlane_b32 s20, v2\n", 1),
    "wide store -> overwrite": ("global_store_dwordx4 v1, v[4:7], s[20:21]\nv_mul_f64 v[6:7], v[8:9], v[8:9]\n", 1),
    "wide LDS store -> overwrite": ("ds_write_b128 v1, v[4:7] offset:512\nv_mov_b32_e32 v4, 0\n", 1),
}
GOOD = {
    "trans -> VALU, spaced": "v_rcp_f64_e64 v[2:3], v[4:5]\ns_nop 1\nv_fma_f64 v[6:7], -v[4:5], v[2:3], 1.0\n",
    "trans -> unrelated VALU": "v_rcp_f64_e64 v[2:3], v[4:5]\nv_mul_f64 v[6:7], v[8:9], v[8:9]\nv_mul_f64 v[10:11], v[8:9], v[8:9]\nv_mul_f64 v[6:7], v[2:3], v[2:3]\n",
    "vcc -> cndmask, two VALU ops between": "v_cmp_eq_f64_e64 vcc, v[2:3], 0\nv_mov_b32_e32 v10, s4\nv_mov_b32_e32 v11, s5\nv_cndmask_b32_e32 v6, v8, v10, vcc\n",
    "SALU sgpr -> VMEM": "s_add_u32 s48, s48, s30\ns_addc_u32 s49, s49, s31\nglobal_load_dwordx2 v[4:5], v2, s[48:49]\n",
    "narrow store -> overwrite": "global_store_dwordx2 v1, v[4:5], s[20:21]\nv_mul_f64 v[4:5], v[8:9], v[8:9]\n",
    "div_fmas after four": "v_div_scale_f64 v[10:11], vcc, 1.0, v[2:3], 1.0\ns_nop 3\nv_div_fmas_f64 v[4:5], v[4:5], v[6:7], v[8:9]\n",
}


@pytest.mark.parametrize("case", sorted(BAD))
def test_checker_flags_each_rule(case):
    text, want = BAD[case]
    n, rep = capi.isa_check_hazards(text)
    assert n == want, rep


@pytest.mark.parametrize("case", sorted(GOOD))
def test_checker_accepts_spaced_code(case):
    n, rep = capi.isa_check_hazards(GOOD[case])
    assert n == 0, rep
