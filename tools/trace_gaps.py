"""Idle time between consecutive kernels of a single-stream run, from a rocprofv3 --kernel-trace CSV.

    python tools/trace_gaps.py <...kernel_trace.csv> [frames]

Prints the total kernel time, the total gap time (next start - previous end, clipped at 0; gaps > 2 ms are step / warm-up
boundaries and are left out) and the 12 kernels that are followed by the most idle time.  Used with `bench.py --no-overlap`
to see how much of the tracking path's wall time is dispatch latency rather than kernel execution."""
import collections
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 0
busy = sum(e - s for s, e, _ in rows)
gap_after = collections.defaultdict(lambda: [0, 0])
gaps = 0
prev_end, prev_name = rows[0][1], rows[0][2]
for s, e, n in rows[1:]:
    g = s - prev_end
    if 0 < g < 2_000_000:
        gaps += g
        gap_after[prev_name][0] += g
        gap_after[prev_name][1] += 1
    prev_end, prev_name = max(prev_end, e), n
print(f"kernels {len(rows)}  busy {busy / 1e6:.1f} ms  gaps {gaps / 1e6:.1f} ms ({100.0 * gaps / (busy + gaps):.1f} % of busy + gaps)")
if frames:
    print(f"per frame: busy {busy / 1e6 / frames:.3f} ms, gaps {gaps / 1e6 / frames:.3f} ms, {len(rows) / frames:.0f} launches")
for name, (g, c) in sorted(gap_after.items(), key=lambda kv: -kv[1][0])[:12]:
    print(f"  {g / 1e6:8.2f} ms after {c:6d} x {name[:90]}  ({g / c / 1e3:.1f} us each)")
