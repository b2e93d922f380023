"""Context of the longest idle gaps of the tracking queue in a rocprofv3 --kernel-trace CSV of the two-stream bench run: for each of the
N longest gaps (inside the last 60 % of the trace) the kernels of that queue around it and what the other queues ran at its two ends.

    python tools/gap_context.py <...kernel_trace.csv> [N]

(What it showed in round 3: the long gaps of the tracking queue under the profiler open when an encoder pass is being ENQUEUED - the
first kernel behind the gap is the one the host launches right after the pass - i.e. they are the profiler's launch overhead, see
tools/overlap_timeline.py.)"""
import collections, csv, re, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"])[:60],
                 r.get("Stream_Id", "?"), r.get("Thread_Id", "?")))
rows.sort()
rows = rows[int(len(rows) * 0.4):]
byq = collections.defaultdict(list)
for x in rows:
    byq[x[2]].append(x)
qs = sorted(byq, key=lambda q: -sum(e - s for s, e, *_ in byq[q]))
print("queues by busy time:", [(q, len(byq[q]), round(sum(e - s for s, e, *_ in byq[q]) / 1e6, 1)) for q in qs[:6]])
enc_q = max(qs[:2], key=lambda q: sum(1 for x in byq[q] if "hiera_attn" in x[3] or "mlp_fused" in x[3]))
trk_q = [q for q in qs[:2] if q != enc_q][0]
T = byq[trk_q]
gaps = sorted(((T[i + 1][0] - T[i][1], i) for i in range(len(T) - 1)), reverse=True)[: int(sys.argv[2]) if len(sys.argv) > 2 else 4]
E_ = byq[enc_q]
for g, i in gaps:
    print(f"\ngap {g / 1e6:.2f} ms in the tracking queue {trk_q}:")
    for x in T[max(0, i - 3): i + 1]:
        print(f"   before  {x[3]:60s} stream {x[4]} thread {x[5]} ran {(x[1] - x[0]) / 1e3:8.1f} us")
    for x in T[i + 1: i + 4]:
        print(f"   after   {x[3]:60s} stream {x[4]} thread {x[5]} ran {(x[1] - x[0]) / 1e3:8.1f} us")
    a, b = T[i][1], T[i + 1][0]
    inside = [x for x in E_ if x[1] > a and x[0] < b]
    print(f"   encoder queue {enc_q} meanwhile: {len(inside)} kernels", end="")
    if inside:
        print(f", first {inside[0][3]} (starts {(inside[0][0] - a) / 1e6:+.2f} ms after the gap opens), last {inside[-1][3]} (ends {(inside[-1][1] - b) / 1e6:+.2f} ms relative to the gap's end)")
        first_patch = [x for x in inside if "im2col_patch" in x[3]]
        print(f"   encoder passes starting inside the gap: {len(first_patch)}")
    else:
        print()
    others = [q for q in qs if q not in (enc_q, trk_q)]
    for q in others[:3]:
        ins = [x for x in byq[q] if x[1] > a - 2_000_000 and x[0] < b + 2_000_000]
        if ins:
            print(f"   queue {q}: {len(ins)} kernels around the gap, e.g. {ins[0][3]} at {(ins[0][0] - a) / 1e6:+.2f} ms")
