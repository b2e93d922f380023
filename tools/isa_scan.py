"""ISA pass over the MFMA kernels (no GPU needed): compiles every .hip source of sam2_opt_amd/csrc to gfx950 assembly and prints, per
kernel, registers / scratch / static LDS, the waves per SIMD the register count allows, and for the hottest loop block the number of
MFMAs, LDS reads and s_waitcnt lgkmcnt instructions - "MFMAs per wait" near 1 means the scheduler has sunk every ds_read to the MFMA that
uses it (what round 3 found in gemm_v2_kernel and hiera_attn_kernel), scratch > 0 inside a loop means spills in the hot path.

    python tools/isa_scan.py [file.hip ...]        # default: the MFMA kernel sources"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sam2_opt_amd", "csrc")
DEFAULT = ["gemm2.hip", "gemm_xs.hip", "mlp_fused.hip", "attn_hiera.hip", "attn_flash256.hip", "gemm_projln.hip", "gemm_rowln.hip", "attn_precise.hip"]


def demangle(n):
    d = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    return re.sub(r"\(anonymous namespace\)::|^void ", "", d)


def scan(files, hipcc=None):
    """-> list of dicts, one per kernel that issues MFMAs: name, vgpr, waves_per_simd, scratch, lds, and for the hottest loop block
    (None when fully unrolled) mfma / ds_read / lgkm_waits; scratch_in_loops = spill instructions inside any loop block."""
    hipcc = hipcc or os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    rows = []
    for f in files:
        src = f if os.path.isabs(f) else os.path.join(ROOT, f)
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "k.s")
            r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, src], capture_output=True, text=True)
            if not os.path.exists(out):
                raise RuntimeError(f"{f}: compile failed\n{r.stderr[-400:]}")
            txt = open(out).read()
        meta = {}
        for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
            b = m.group(2)
            g = lambda k: int(re.search(k + r"\s+(\d+)", b).group(1))
            meta[m.group(1)] = (g("amdhsa_next_free_vgpr"), g("amdhsa_private_segment_fixed_size"), g("amdhsa_group_segment_fixed_size"))
        for part in re.split(r"\n(?=_Z[A-Za-z0-9_]+:\s*(?:;.*)?\n)", txt):
            if not part.startswith("_Z"):
                continue
            name = part.split(":", 1)[0]
            if name not in meta:
                continue
            body = part.split(".Lfunc_end")[0]
            if "v_mfma" not in body:
                continue
            vg, sc, lds = meta[name]
            alloc = (vg + 7) // 8 * 8
            blocks = re.split(r"\n(?=\.LBB\d+_\d+:)", body)
            loops = [b for b in blocks if "Loop" in b.split("\n")[0] and "v_mfma" in b]
            hot = max(loops, key=lambda b: b.count("v_mfma")) if loops else None
            row = dict(file=os.path.basename(src), name=demangle(name), vgpr=vg, waves_per_simd=min(8, 512 // alloc), scratch=sc, lds=lds,
                       scratch_in_loops=sum(b.count("scratch_") for b in blocks if "Loop" in b.split("\n")[0]), mfma=None, ds_read=None, lgkm_waits=None)
            if hot:
                row.update(mfma=hot.count("v_mfma"), ds_read=hot.count("ds_read"), lgkm_waits=len(re.findall(r"s_waitcnt[^\n]*lgkmcnt", hot)))
            rows.append(row)
    return rows


if __name__ == "__main__":
    print(f"{'kernel':78s} {'vgpr':>4s} {'w/SIMD':>6s} {'scratch':>7s} {'lds':>6s} | hottest loop block: mfma ds_read lgkm-waits mfma/wait  scratch-ops-in-loops")
    for r in scan(sys.argv[1:] or DEFAULT):
        if r["mfma"] is not None:
            tail = f"{r['mfma']:4d} {r['ds_read']:7d} {r['lgkm_waits']:10d} {r['mfma'] / max(r['lgkm_waits'], 1):9.1f}  {r['scratch_in_loops']:4d}"
        else:
            tail = "   (no MFMA loop block: unrolled)"
        print(f"{r['name'][:78]:78s} {r['vgpr']:4d} {r['waves_per_simd']:6d} {r['scratch']:7d} {r['lds']:6d} | {tail}")
