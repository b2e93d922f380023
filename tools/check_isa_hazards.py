#!/usr/bin/env python3
"""Build-time ISA check (CPU-only: hipcc cross-compiles) for the one miscompute this code base has met:
a build of the RoPE epilogue of gemm_v2_kernel formed `v_pk_mul_f32` / `v_pk_fma_f32 ... op_sel` right behind the `s_waitcnt` of a
table load WHOSE DESTINATION REGISTERS WERE ALSO ITS ADDRESS REGISTERS (`global_load_dwordx2 v[a:a+1], v[a:a+1], off`), and on a few
rows per call lanes 48-63 read the stale address half instead of the loaded cosine.  The source now forces scalar FMAs there; this
script pins the property at the ISA level for every kernel of the library:

    no packed-f32 VALU op (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32) with an op_sel / op_sel_hi modifier may read a VGPR that a
    global / flat load wrote OVER ITS OWN ADDRESS REGISTERS, until that VGPR has been written again.

What round 3 found with this script: the pattern by itself is NOT sufficient - convs.hip holds 618 such sites (conv3x3s2_ln_gelu,
dwconv7), elementwise.hip 2, attn_small.hip 1, and those kernels are bit-stable over repeated calls
(tests/test_plugs_gpu.py::test_memory_encoder_is_deterministic / test_mask_decoder_is_deterministic).  So the recorded hypothesis
("forwarding hazard of dst == address loads into packed ops") is not confirmed; whatever made the round-2 build fail needed more
than this pattern, and its ISA was not kept.  The check therefore FAILS only for the kernel family where the miscompute was seen
(gemm2.hip: must stay at zero sites), and REPORTS the count for the others so that a change shows up in review.

    python tools/check_isa_hazards.py [file.hip ...]        (default: every .hip of the product build)
Writes profiles/<tag>_rope_epilogue.s (the RoPE section of the K-projection instantiation, the good build) when --dump <tag> is given.
Exit code 1 on a finding."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sam2_opt_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def scan(asm_text):
    """-> [(kernel, line number, load line, consumer line)]"""
    findings, kernel, tainted = [], None, {}
    for ln, line in enumerate(asm_text.splitlines(), 1):
        s = line.split(";")[0].strip()
        if s.endswith(":") and not s.startswith("."):
            if s.startswith("_Z"):
                kernel, tainted = s[:-1], {}
            continue                                         # local labels: the taint survives (conservative)
        if not s or s.startswith("."):
            continue
        parts = s.split(None, 1)
        op, args = parts[0], (parts[1] if len(parts) > 1 else "")
        ops = [a.strip() for a in args.split(",")]
        if op.startswith(("global_load_dword", "flat_load_dword")) and len(ops) >= 2:
            dst, addr = regs(ops[0]), regs(ops[1])
            for r in dst:
                tainted.pop(r, None)
            for r in dst & addr:
                tainted[r] = (ln, s)
            continue
        if op.startswith("v_pk_") and op.endswith("_f32") and "op_sel" in s and len(ops) >= 2:
            srcs = set()
            for a in ops[1:]:
                srcs |= regs(a.split(" op_sel")[0])
            hit = srcs & set(tainted)
            if hit:
                r = min(hit)
                findings.append((kernel, ln, tainted[r][1], s))
        # any other instruction that writes VGPRs clears their taint (first operand = destination for VALU / loads / DS reads)
        if ops and not op.startswith(("s_", "global_store", "flat_store", "ds_write", "buffer_store", "v_cmp", "global_atomic")):
            for r in regs(ops[0]):
                tainted.pop(r, None)
    return findings


def main():
    args = sys.argv[1:]
    dump = None
    if "--dump" in args:
        i = args.index("--dump")
        dump = args[i + 1]
        del args[i:i + 2]
    args = [os.path.abspath(a) for a in args]
    files = args or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    experimental = {"gemm3.hip", "gemm4.hip", "gemm_ks.hip"}
    bad = 0
    for f in files:
        if os.path.basename(f) in experimental and not args:
            continue
        with tempfile.NamedTemporaryFile(suffix=".s") as t:
            r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", f, "-o", t.name],
                               capture_output=True, text=True, cwd=CSRC)
            if r.returncode != 0:
                print(f"[isa] {os.path.basename(f)}: compile failed\n{r.stderr[-2000:]}")
                return 2
            text = open(t.name).read()
        found = scan(text)
        npk = len(re.findall(r"v_pk_\w+_f32", text))
        print(f"[isa] {os.path.basename(f)}: {npk} packed-f32 ops, {len(found)} behind a load that overwrote its own address", flush=True)
        for k, ln, ld, use in found[:3]:
            print(f"       {k} line {ln}: `{use}`  after  `{ld}`")
        strict = os.path.basename(f) in ("gemm2.hip", "gemm.hip")
        bad += len(found) if strict else 0
        if dump and os.path.basename(f) == "gemm2.hip":
            # the K projection of the memory bank: 128x64 tile, plain f16 - the RoPE section = the code around the rope_cos / rope_sin loads
            lines = text.splitlines()
            start = next((i for i, l in enumerate(lines) if l.startswith("_ZN") and "gemm_v2_kernelILi128ELi64E" in l and "Li0ELb0E" in l), None)
            if start is not None:
                end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
                body = lines[start:end + 1]
                idx = [i for i, l in enumerate(body) if "global_load_dwordx2" in l]
                out = os.path.join(ROOT, "profiles", f"{dump}_rope_epilogue.s")
                with open(out, "w") as fo:
                    fo.write(f"; {lines[start]}  ({len(body)} lines; windows of +-14 lines around each 8-byte table load = the RoPE rotation)\n")
                    last = -1
                    for i in idx:
                        lo, hi = max(i - 14, last + 1), min(i + 15, len(body))
                        if lo >= hi:
                            continue
                        fo.write(f"; ---- lines {lo}..{hi}\n" + "\n".join(body[lo:hi]) + "\n")
                        last = hi - 1
                print(f"[isa] wrote {out}")
    print("[isa] OK (gemm2.hip: 0 sites)" if not bad else f"[isa] {bad} finding(s) in the strict set")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
