"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counter_collection CSVs) into a per-kernel HBM-traffic
table (markdown on stdout) and profiles/pmc_traffic.json (what bench.py reports as roofline.traffic).

    python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <label> [json_out]

Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in
KiB-like units of 1 KB; wide coalesced reads are counted at 64 B per 128-B request, so fetch is doubled ("fetch x2").
"""
import collections
import csv
import json
import sys


def load(path, counter):
    tot, n = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        tot[k] += float(r["Counter_Value"])
        n[k].add(r["Dispatch_Id"])
    return tot, {k: len(v) for k, v in n.items()}


def main():
    fetch, nf = load(sys.argv[1], "FETCH_SIZE")
    write, _ = load(sys.argv[2], "WRITE_SIZE")
    label = sys.argv[3]
    rows = []
    for k in fetch:
        f_gb, w_gb = fetch[k] * 1024 / 1e9, write.get(k, 0.0) * 1024 / 1e9
        rows.append((2 * f_gb + w_gb, k, nf[k], f_gb, w_gb))
    rows.sort(reverse=True)
    print(f"| kernel | launches | FETCH_SIZE raw (GB) | fetch x2 (GB) | WRITE_SIZE (GB) | (fetch x2 + write) per launch (MB) |")
    print("|---|---|---|---|---|---|")
    for tot, k, n, f, w in rows[:16]:
        print(f"| `{k[:70]}` | {n} | {f:.2f} | {2 * f:.2f} | {w:.2f} | {tot / n * 1e3:.1f} |")
    out = {}
    for fam in ("gemm_v2_kernel", "gemm_xs_kernel", "mlp_fused_kernel"):
        gem = [(tot, k, n, f, w) for tot, k, n, f, w in rows if fam in k]
        if not gem:
            continue
        tot = sum(g[0] for g in gem)
        n = sum(g[2] for g in gem)
        out[fam] = {"bytes_per_launch": round(tot * 1e9 / n, -5), "launches": n,
                    "fetch_x2_gb": round(sum(2 * g[3] for g in gem), 2), "write_gb": round(sum(g[4] for g in gem), 2),
                    "source": label + f" (all {fam} instantiations)"}
        print(f"\n{fam} (all instantiations): {n} launches, {tot:.1f} GB = {tot / n * 1e3:.1f} MB per launch")
    import re
    inst = {}
    for tot, k, n, f, w in rows:
        m = re.search(r"(gemm_v2_kernel|gemm_xs_kernel|mlp_fused_kernel|gemm_ks_kernel|flash256_v3_kernel|hiera_attn_v2_kernel|hiera_attn_kernel|gemm_projln_kernel)<[^>]*>", k)
        if m:
            inst[m.group(0)] = {"bytes_per_launch": round(tot * 1e9 / n, -5), "launches": n}
    out["by_instantiation"] = inst
    if out and len(sys.argv) > 4:
        json.dump(out, open(sys.argv[4], "w"), indent=1)


if __name__ == "__main__":
    main()
