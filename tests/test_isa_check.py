"""CPU (hipcc cross-compiles): the ISA property behind the one miscompute this code base has met (tools/check_isa_hazards.py,
DESIGN.md 4) - no packed-f32 op with op_sel reads a register that a load wrote over its own address registers, in the GEMM kernel
family whose RoPE epilogue once failed that way."""
import importlib.util
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "check_isa_hazards.py")


def _tool():
    spec = importlib.util.spec_from_file_location("check_isa_hazards", TOOL)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_scanner_flags_the_recorded_pattern_and_nothing_else():
    m = _tool()
    bad = """
_ZN1kE:
	global_load_dwordx2 v[6:7], v[6:7], off
	s_waitcnt vmcnt(0)
	v_pk_fma_f32 v[2:3], v[0:1], v[6:7], v[2:3] op_sel_hi:[0,1,1]
	s_endpgm
"""
    good = """
_ZN1kE:
	global_load_dwordx2 v[6:7], v[6:7], off
	s_waitcnt vmcnt(0)
	v_mul_f32_e32 v4, v1, v6
	v_pk_mul_f32 v[8:9], v[8:9], s[6:7] op_sel_hi:[1,0]
	global_load_dwordx2 v[10:11], v[12:13], off
	v_pk_fma_f32 v[2:3], v[0:1], v[10:11], v[2:3] op_sel_hi:[0,1,1]
	v_mov_b32_e32 v6, 0
	v_mov_b32_e32 v7, 0
	v_pk_fma_f32 v[2:3], v[0:1], v[6:7], v[2:3] op_sel_hi:[0,1,1]
	s_endpgm
"""
    assert len(m.scan(bad)) == 1 and m.scan(bad)[0][0] == "_ZN1kE"
    assert m.scan(good) == []


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_gemm_family_has_no_packed_op_behind_a_self_addressed_load():
    r = subprocess.run([sys.executable, TOOL, os.path.join(ROOT, "sam2_opt_amd", "csrc", "gemm2.hip")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "gemm2.hip" in r.stdout and " 0 behind a load" in r.stdout, r.stdout


def _isa_scan():
    spec = importlib.util.spec_from_file_location("isa_scan", os.path.join(ROOT, "tools", "isa_scan.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_hot_loops_keep_their_fragment_prefetch_and_do_not_spill():
    """What the ISA pass of round 3 found and fixed must stay fixed (tools/isa_scan.py): the K loop of the dominant GEMM tile
    requests its fragments ahead of the MFMAs (left alone the scheduler sinks every ds_read to its MFMA: 1 MFMA per LDS wait), the window
    attention v1 and the stage-1 fused MLP fit two workgroups per CU, and no production MFMA kernel spills inside a loop."""
    m = _isa_scan()
    rows = {r["name"]: r for r in m.scan(["gemm2.hip", "attn_hiera.hip", "mlp_fused.hip", "gemm_xs.hip"])}

    def row(prefix):
        hits = [r for n, r in rows.items() if n.startswith(prefix)]
        assert len(hits) == 1, (prefix, [n for n in rows if n.startswith(prefix[:20])])
        return hits[0]
    g = row("gemm_v2_kernel<128, 192, 4, 2, 2, 0, false>")
    assert g["mfma"] == 12 and g["lgkm_waits"] <= 3 and g["vgpr"] <= 128, g          # 12 MFMAs per K tile behind at most 3 LDS waits, 2 workgroups per CU
    for name in ("hiera_attn_kernel<false, false, true>", "hiera_attn_kernel<false, true, true>", "hiera_attn_kernel<false, false, false>"):
        a = row(name)
        assert a["vgpr"] <= 256 and a["waves_per_simd"] >= 2, a
    assert row("mlp_fused_kernel<144, 1, 3>")["vgpr"] <= 256
    production = [r for n, r in rows.items() if not n.startswith("gemm_v2_kernel<256")]
    spilling = [(r["name"], r["scratch_in_loops"]) for r in production if r["scratch_in_loops"] > 0]
    assert not spilling, spilling
