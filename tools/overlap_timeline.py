"""Timeline of the two-stream run from a rocprofv3 --kernel-trace CSV of `bench.py` (encoder prefetch stream on): which queue is
busy when.  Kernels are assigned to the ENCODER queue or the TRACKING queue by name (the encoder's kernel set is disjoint from the
tracking path's except for the shared GEMM / LayerNorm instantiations, which are told apart by their queue id).

    python tools/overlap_timeline.py <...kernel_trace.csv> [frames_in_timed_region]

Prints, for the last `frames` frames' worth of wall time: wall, union busy, per-queue busy, time both queues run, time only one runs,
time the chip is idle, and a 1-ms-bucket strip of one encoder-batch period (E = encoder only, T = tracking only, B = both, . = idle).

CAUTION (found at the end of round 3): under rocprofv3 --kernel-trace a launch costs the host ~60 us instead of ~3 us, the 341 launches of
an encoder pass take ~20 ms to enqueue, and the launching thread - not the GPU - paces the run: the tracking queue runs dry while the pass
is being enqueued (the "21-ms tracked frame" and the idle stretches of the encoder queue in this tool's output).  The schedule of the
un-profiled run is what tools/event_timeline.py measures with in-stream HIP events: encoder passes back to back, every tracked frame
beside one.  This tool is kept for the per-kernel "wait-before" table only."""
import collections
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]))
rows.sort()
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 200
# the timed region: everything after the last long (> 20 ms) idle gap of the whole trace (weight upload / warm-up boundary excluded by
# taking the last 60 % of the kernels)
rows = rows[int(len(rows) * 0.4):]
byq = collections.defaultdict(list)
for s, e, q, n in rows:
    byq[q].append((s, e, n))
qs = sorted(byq, key=lambda q: -sum(e - s for s, e, _ in byq[q]))[:2]
enc_q = max(qs, key=lambda q: sum(1 for _, _, n in byq[q] if "hiera_attn" in n or "mlp_fused" in n))
trk_q = [q for q in qs if q != enc_q][0] if len(qs) > 1 else enc_q
t0, t1 = rows[0][0], rows[-1][1]


def merged(iv):
    out = []
    for s, e in sorted(iv):
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


E_ = merged([(s, e) for s, e, _ in byq[enc_q]])
T_ = merged([(s, e) for s, e, _ in byq[trk_q]])


def total(iv):
    return sum(e - s for s, e in iv)


def inter(a, b):
    i = j = 0
    tot = 0
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if s < e:
            tot += e - s
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return tot


wall = t1 - t0
both = inter(E_, T_)
eb, tb = total(E_), total(T_)
union = eb + tb - both
ms = 1e-6
print(f"region {wall * ms:.1f} ms ({len(rows)} kernels); encoder queue busy {eb * ms:.1f} ms ({100 * eb / wall:.1f} %), tracking queue busy "
      f"{tb * ms:.1f} ms ({100 * tb / wall:.1f} %)")
print(f"both {both * ms:.1f} ms ({100 * both / wall:.1f} %), encoder only {(eb - both) * ms:.1f} ms ({100 * (eb - both) / wall:.1f} %), "
      f"tracking only {(tb - both) * ms:.1f} ms ({100 * (tb - both) / wall:.1f} %), idle {(wall - union) * ms:.1f} ms ({100 * (wall - union) / wall:.1f} %)")
# strip: 0.5-ms buckets over ~2 encoder batches in the middle of the region
mid = t0 + wall // 2
B = 500_000
line = ""
for k in range(200):
    s, e = mid + k * B, mid + (k + 1) * B
    be = inter(E_, [[s, e]]) / B
    bt = inter(T_, [[s, e]]) / B
    line += "B" if (be > 0.5 and bt > 0.5) else "E" if be > 0.5 else "T" if bt > 0.5 else "."
print("0.5-ms buckets from the middle of the region:")
for i in range(0, len(line), 100):
    print("  " + line[i:i + 100])
# encoder-queue idle stretches (> 1 ms): what the tracking queue ran meanwhile
idle = []
for (s0, e0), (s1, e1) in zip(E_, E_[1:]):
    if s1 - e0 > 1_000_000:
        idle.append((e0, s1))
print(f"encoder queue idle stretches > 1 ms: {len(idle)}, total {sum(b - a for a, b in idle) * ms:.1f} ms, mean {sum(b - a for a, b in idle) * ms / max(len(idle), 1):.2f} ms")
# ---- event list: encoder passes (first / last kernel) and tracked-frame starts inside ~120 ms from the middle
print("events (ms from the middle of the region): E< encoder pass starts, E> ends, t tracked frame starts (mem_assemble_kernel)")
ev = []
enc = sorted(byq[enc_q])
passes, cur = [], [enc[0][0], enc[0][1]]
for s, e, n in enc[1:]:
    if "im2col_patch" in n and s - cur[1] > 0:
        passes.append(cur)
        cur = [s, e]
    cur[1] = max(cur[1], e)
passes.append(cur)
for s, e in passes:
    ev.append((s, "E<"))
    ev.append((e, "E>"))
for s, e, n in byq[trk_q]:
    if "mem_assemble" in n:
        ev.append((s, "t"))
line = []
for t, k in sorted(ev):
    if mid <= t <= mid + 120_000_000:
        line.append(f"{(t - mid) * ms:.1f}{k}")
print("  " + " ".join(line))
# ---- one tracked frame while the encoder runs beside it: where does its time go?
starts = [s for s, e, n in byq[trk_q] if "mem_assemble" in n and s >= mid]
best = None
for a, b in zip(starts, starts[1:]):
    if b - a > 6_000_000 and inter(E_, [[a, b]]) > 0.9 * (b - a):
        best = (a, b)
        break
if best:
    a, b = best
    ks = [(s, e, n) for s, e, n in sorted(byq[trk_q]) if a <= s < b]
    busy = sum(e - s for s, e, n in ks)
    print(f"one co-running tracked frame: {len(ks)} kernels, wall {(b - a) * ms:.2f} ms, kernel time {busy * ms:.2f} ms, gaps {(b - a - busy) * ms:.2f} ms")
    agg = collections.defaultdict(lambda: [0, 0, 0])
    prev = a
    for s, e, n in ks:
        k = n.replace("void ", "").replace("(anonymous namespace)::", "")
        k = (k[:k.index("(")] if "(" in k and not k.startswith("_Z") else k)[:64]
        agg[k][0] += e - s
        agg[k][1] += max(s - prev, 0)
        agg[k][2] += 1
        prev = e
    for k, (d, g, c) in sorted(agg.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))[:22]:
        print(f"   {c:3d} x {k:62s} run {d * ms:6.2f} ms  wait-before {g * ms:6.2f} ms")
