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
