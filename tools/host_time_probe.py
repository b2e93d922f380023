"""Where does the HOST spend a frame?  Wraps the engine entry points the video predictor calls with perf_counter and prints, for
one 100-frame propagate pass of the benchmark clip, the host time per call type (enqueue only - nothing here synchronises) next
to the wall time.  If the sum approaches the wall time, the launch path, not the GPU, paces the run."""
import collections, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam2_opt_amd.config import get_config
from sam2_opt_amd.synthetic import normalize_frames, synthetic_frames_u8
from sam2_opt_amd.video_predictor import SAM2VideoPredictor
from sam2_opt_amd.weights import synthetic_state_dict

overlap = os.environ.get("OVERLAP", "1") == "1"
cfg = get_config("large")
sd = synthetic_state_dict(cfg, seed=0)
pred = SAM2VideoPredictor("large", state_dict=sd, encode_batch=8, overlap_encode=overlap)
frames = normalize_frames(synthetic_frames_u8(seed=2, num_frames=100), cfg)
state = pred.init_state(frames=frames, video_height=1024, video_width=1024)
pred.add_new_points_or_box(state, 0, 1, points=np.array([[512.0, 512.0]], np.float32), labels=np.array([1], np.int32))
acc = collections.defaultdict(lambda: [0.0, 0])


def wrap(obj, name):
    f = getattr(obj, name)

    def g(*a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        e = acc[name]
        e[0] += time.perf_counter() - t
        e[1] += 1
        return r
    setattr(obj, name, g)


for n in ("video_track", "video_encode", "video_encode_u8", "video_encode_memory", "new"):
    if hasattr(pred.engine, n):
        wrap(pred.engine, n)
_orig_wait_event = torch.cuda.Stream.wait_event
def _timed_wait_event(self, ev):
    t = time.perf_counter()
    r = _orig_wait_event(self, ev)
    e = acc["Stream.wait_event"]
    e[0] += time.perf_counter() - t
    e[1] += 1
    return r
torch.cuda.Stream.wait_event = _timed_wait_event
_orig_wait_stream = torch.cuda.Stream.wait_stream
def _timed_wait_stream(self, so):
    t = time.perf_counter()
    r = _orig_wait_stream(self, so)
    e = acc["Stream.wait_stream"]
    e[0] += time.perf_counter() - t
    e[1] += 1
    return r
torch.cuda.Stream.wait_stream = _timed_wait_stream
for n in ("_ensure_features", "_encode_batch", "_select_memory", "_video_res", "_alloc_bank", "_release_stale"):
    if hasattr(pred, n):
        wrap(pred, n)
for step in range(3):
    acc.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = sum(1 for _ in pred.propagate_in_video(state))
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print(f"pass {step}: {n} frames, host loop {t_host * 1e3:.1f} ms, wall {wall * 1e3:.1f} ms ({n / wall:.1f} fps), overlap={overlap}")
    for k, (t, c) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
        print(f"    {k:22s} {t * 1e3:8.1f} ms  {c:5d} calls  {t / max(c, 1) * 1e6:8.1f} us each")
