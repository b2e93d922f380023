"""Summarise the SQ / GRBM rocprofv3 --pmc passes of tools/profile_round.sh into the per-kernel utilisation table north_star asks for
("rocprof ... MFMA utilisation against gfx950 peak"): markdown on stdout.

    python tools/pmc_util_summary.py <passA counter_collection.csv> <passB counter_collection.csv> [top_n]

Pass A: SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY
        SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE
Pass B: SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA
        SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
Units (/opt/skills/guides/MI355X_MICROARCH.md, rocprofv3 PMC slots): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles
summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD (32 per 32x32x16 f16 MFMA); GRBM_GUI_ACTIVE = shader-clock cycles
the dispatch kept the chip busy, SUMMED OVER THE 8 XCDs (checked on gemm_v2_kernel<128, 192, ...>: 2.33 M per dispatch of 131 us =
291 k cycles per XCD = 2.22 GHz; SQ_VALU_MFMA_BUSY_CYCLES = 83.6 M = 32 x the 2.61 M MFMA instructions of that launch).  Derived columns:
  mfma_util   = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)    - the fraction of the MFMA pipes' time in use AT THE
                CLOCK THE KERNEL RAN AT (the roofline fractions of bench.py are against the 2.4-GHz peak, so they sit ~8 % lower)
  wait / stall / issue = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES   (disjoint, sum ~ 1): a wave parked
                on s_waitcnt / a barrier, stalled at issue (MFMA RAW, busy pipe), issuing
  lds_stall   = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES (sub-bucket of stall);  bank_conf = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  valu / lds / vmem = SQ_ACTIVE_INST_{VALU,LDS,VMEM} / SQ_WAVE_CYCLES
"""
import collections
import csv
import re
import sys


def load(path):
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k].add(r["Dispatch_Id"])
    return tot, {k: len(v) for k, v in n.items()}


def short(k):
    k = re.sub(r"\(anonymous namespace\)::", "", k)
    k = re.sub(r"^void ", "", k)
    k = re.sub(r"\(.*\)$", "", k)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)(I.*)?E", k)
    return (m.group(1) if m else k)[:64]


def main():
    a, na = load(sys.argv[1])
    b, _ = load(sys.argv[2])
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 14
    rows = sorted(a.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0.0))[:top]
    print("| kernel | launches | share of GPU-active cycles | mfma_util | wait | stall | issue | lds_stall | valu | lds | vmem | bank_conf |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    total_active = sum(v.get("GRBM_GUI_ACTIVE", 0.0) for v in a.values())
    for k, c in rows:
        wc = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        cb = b.get(k, {})
        wcb = max(cb.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        f = lambda x: f"{x:.3f}"
        print(f"| `{short(k)}` | {na[k]} | {c.get('GRBM_GUI_ACTIVE', 0.0) / max(total_active, 1.0):.3f} | "
              f"{f(c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / max(c.get('GRBM_GUI_ACTIVE', 0.0) / 8.0 * 1024.0, 1.0))} | {f(c.get('SQ_WAIT_ANY', 0.0) / wc)} | "
              f"{f(c.get('SQ_WAIT_INST_ANY', 0.0) / wc)} | {f(c.get('SQ_ACTIVE_INST_ANY', 0.0) / wc)} | {f(c.get('SQ_WAIT_INST_LDS', 0.0) / wc)} | "
              f"{f(cb.get('SQ_ACTIVE_INST_VALU', 0.0) / wcb)} | {f(cb.get('SQ_ACTIVE_INST_LDS', 0.0) / wcb)} | {f(cb.get('SQ_ACTIVE_INST_VMEM', 0.0) / wcb)} | "
              f"{f(cb.get('SQ_LDS_BANK_CONFLICT', 0.0) / max(cb.get('SQ_LDS_IDX_ACTIVE', 0.0), 1.0))} |")


if __name__ == "__main__":
    main()
