"""The two-stream schedule WITHOUT a profiler attached (rocprofv3's launch overhead makes the host the pacer: an encoder pass of 341 launches
takes ~20 ms to enqueue under --kernel-trace, 1.1 ms without).  HIP events are recorded in-stream around every tracked frame (tracking
stream) and every encoder pass (encoder stream) of one propagate pass over the benchmark clip; after the final synchronise the elapsed
times between them give the real timeline: when each encoder pass ran, how long each tracked frame took and whether it ran beside a pass."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam2_opt_amd.config import get_config
from sam2_opt_amd.synthetic import normalize_frames, synthetic_frames_u8
from sam2_opt_amd.video_predictor import SAM2VideoPredictor
from sam2_opt_amd.weights import synthetic_state_dict

cfg = get_config("large")
pred = SAM2VideoPredictor("large", state_dict=synthetic_state_dict(cfg, seed=0), encode_batch=8, overlap_encode=True,
                          prefetch_depth=int(os.environ.get("DEPTH", "2")))
frames = normalize_frames(synthetic_frames_u8(seed=2, num_frames=100), cfg).cuda()
state = pred.init_state(frames=frames, video_height=1024, video_width=1024)
pred.add_new_points_or_box(state, 0, 1, points=np.array([[512.0, 512.0]], np.float32), labels=np.array([1], np.int32))
marks = []          # (kind, event)


def wrap(name, kind):
    f = getattr(pred.engine, name)

    def g(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = f(*a, **k)
        e1.record()
        marks.append((kind, e0, e1, time.perf_counter()))
        return r
    setattr(pred.engine, name, g)


wrap("video_track", "t")
wrap("video_encode", "E")
for rep in range(3):
    marks.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    base = torch.cuda.Event(enable_timing=True)
    base.record()
    n = sum(1 for _ in pred.propagate_in_video(state))
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
print(f"{n} frames, wall {wall * 1e3:.1f} ms ({n / wall:.1f} frames/s)")
ev = [(k, base.elapsed_time(a), base.elapsed_time(b), (h - t0) * 1e3) for k, a, b, h in marks]
enc = [(a, b) for k, a, b, _ in ev if k == "E"]
print("encoder passes (start, end, length ms):", " ".join(f"[{a:.1f} {b:.1f} {b - a:.1f}]" for a, b in enc))
trk = [(a, b, h) for k, a, b, h in ev if k == "t"]
line = []
for i, (a, b, h) in enumerate(trk):
    beside = sum(max(0.0, min(b, eb) - max(a, ea)) for ea, eb in enc)
    line.append(f"{b - a:.1f}{'*' if beside > 0.5 * (b - a) else ''}")
print("tracked frame lengths in ms, GPU time from its first to its last kernel (* = more than half of it beside an encoder pass):")
for i in range(0, len(line), 16):
    print("   ", " ".join(line[i:i + 16]))
gaps = [trk[i + 1][0] - trk[i][1] for i in range(len(trk) - 1)]
print("idle between tracked frames > 0.5 ms:", " ".join(f"{i}:{g:.1f}" for i, g in enumerate(gaps) if g > 0.5))
lead = [h - b for a, b, h in trk]
print(f"host lead over the GPU at the end of each frame's enqueue (ms, negative = host ahead): min {min(lead):.1f} median {sorted(lead)[len(lead) // 2]:.1f} max {max(lead):.1f}")
tot_t = sum(b - a for a, b, _ in trk)
tot_e = sum(b - a for a, b in enc)
both = sum(sum(max(0.0, min(b, eb) - max(a, ea)) for ea, eb in enc) for a, b, _ in trk)
print(f"sum of tracked-frame spans {tot_t:.1f} ms, encoder spans {tot_e:.1f} ms, overlap of the two {both:.1f} ms, wall {wall * 1e3:.1f} ms")
