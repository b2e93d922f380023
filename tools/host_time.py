import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from sam2_opt_amd.config import get_config
from sam2_opt_amd.synthetic import synthetic_frames_u8
from sam2_opt_amd.video_predictor import SAM2VideoPredictor
from sam2_opt_amd.weights import synthetic_state_dict
pred = SAM2VideoPredictor("large", state_dict=synthetic_state_dict(get_config("large"), seed=0), encode_batch=8)
st = pred.init_state(frames_u8=synthetic_frames_u8(seed=2, num_frames=100), video_height=1024, video_width=1024)
pred.add_new_points_or_box(st, 0, 1, points=np.array([[512.0, 512.0]], np.float32), labels=np.array([1], np.int32))
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = sum(1 for _ in pred.propagate_in_video(st))
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"host enqueue {1e3*(t1-t0)/n:.2f} ms/frame, total {1e3*(t2-t0)/n:.2f} ms/frame", flush=True)
# pure host cost of one tracked frame: GPU idle before the call, time until the call returns (everything is enqueued)
pred.overlap_encode = False
ts = []
it = pred.propagate_in_video(st)
next(it)
for k in range(30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    next(it)
    ts.append(time.perf_counter() - t0)
torch.cuda.synchronize()
ts = np.array(ts) * 1e3
print("per-frame host time with an idle GPU (ms): median %.2f  min %.2f  max %.2f (max = frames that also enqueue an encoder batch)" % (np.median(ts), ts.min(), ts.max()), flush=True)
