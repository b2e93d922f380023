"""The kernel sequence of ONE steady-state tracking frame, from a rocprofv3 --kernel-trace CSV of `bench.py --no-overlap`.

    python tools/track_frame_trace.py <...kernel_trace.csv> [which]

A tracking frame starts at a `mem_assemble_kernel` launch and ends before the next one (or before the next image-encoder
batch, recognised by `im2col_patch_kernel`).  Prints every launch of the `which`-th frame from the end that is a full frame
(default 3) with its duration and the idle time before it, then the totals - the budget table for fusing the tracking path."""
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
which = int(sys.argv[2]) if len(sys.argv) > 2 else 3
if len(sys.argv) > 3 and sys.argv[3] == "enc":
    # one image-encoder batch instead: from an im2col_patch launch to the next mem_assemble launch, consecutive equal launches folded
    starts = [i for i, r in enumerate(rows) if "im2col_patch" in r[2]]
    a = starts[-which]
    b = next((i for i in range(a + 1, len(rows)) if "mem_assemble_kernel" in rows[i][2] or "im2col_patch" in rows[i][2]), len(rows))
    seg, out, busy = rows[a:b], [], 0
    for s, e, n in seg:
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"^void ", "", n)[:100]
        busy += e - s
        if out and out[-1][0] == n and abs(out[-1][1] / out[-1][2] - (e - s)) < 0.3 * (e - s):
            out[-1][1] += e - s
            out[-1][2] += 1
        else:
            out.append([n, e - s, 1])
    for n, t, c in out:
        print(f"{t / c / 1e3:8.1f} us x{c:3d}  {n}")
    print(f"launches {len(seg)}  busy {busy / 1e3:.1f} us  wall {(seg[-1][1] - seg[0][0]) / 1e3:.1f} us")
    sys.exit(0)
starts = [i for i, r in enumerate(rows) if "mem_assemble_kernel" in r[2]]
frames = []
for a, b in zip(starts, starts[1:]):
    seg = rows[a:b]
    if any("im2col_patch" in n for _, _, n in seg):
        continue
    frames.append(seg)
seg = frames[-which]
busy = gaps = 0
prev = None
for s, e, n in seg:
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    g = (s - prev) if prev is not None else 0
    busy += e - s
    gaps += max(g, 0)
    print(f"{(e - s) / 1e3:8.1f} us  gap {g / 1e3:6.1f}  {n[:110]}")
    prev = e
print(f"launches {len(seg)}  busy {busy / 1e3:.1f} us  gaps {gaps / 1e3:.1f} us  wall {(seg[-1][1] - seg[0][0]) / 1e3:.1f} us")
