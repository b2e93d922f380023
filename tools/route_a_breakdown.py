"""Per-plug time of the drop-in route (route A, sam2_opt_amd/route_a.py) on the benchmark clip: HIP-event time around every plug call
and around the torch glue between them, averaged over steady-state frames (memory bank full)."""
import collections, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam2_opt_amd.config import get_config
from sam2_opt_amd.route_a import PlugLevelTracker
from sam2_opt_amd.synthetic import normalize_frames, synthetic_frames_u8
from sam2_opt_amd.weights import synthetic_state_dict

cfg = get_config("large")
sd = synthetic_state_dict(cfg, seed=0)
trk = PlugLevelTracker("large", state_dict=sd, precision=os.environ.get("PRECISION", "f16s"), lookahead=int(os.environ.get("LOOKAHEAD", "8")))
frames = normalize_frames(synthetic_frames_u8(seed=2, num_frames=40), cfg).cuda()
acc = collections.defaultdict(list)


def wrap(name):
    f = getattr(trk.engine, name)

    def g(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = f(*a, **k)
        e1.record()
        acc[name].append((e0, e1))
        return r
    setattr(trk.engine, name, g)


for n in ("image_encoder", "memory_attention", "prompt_encoder_full", "mask_decoder", "memory_encoder"):
    wrap(n)
trk.start(frames, (512.0, 512.0))
for rep in range(3):
    acc.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = sum(1 for _ in trk.propagate())
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    tot = 0.0
    print(f"pass {rep}: {n} frames, {wall * 1e3 / n:.2f} ms per frame ({n / wall:.1f} fps)")
    for k, evs in acc.items():
        ms = [a.elapsed_time(b) for a, b in (evs[10:] if len(evs) > 12 else evs)]
        tot += sum(ms) / len(ms)
        print(f"    {k:22s} {sum(ms) / len(ms):7.3f} ms per call ({len(evs)} calls)")
    print(f"    plugs together {tot:.3f} ms; the rest is torch glue, layout transposes and launch gaps")
    if hasattr(trk.image_plug, "stats"):
        print("    look-ahead plug:", trk.image_plug.stats)
