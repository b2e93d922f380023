"""Multi-object clip timing (tuning aid, not the bench line): K objects on a 60-frame 1024^2 clip, batched tracking pass vs the
reference-style per-object loop."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sam2_opt_amd.config import get_config
from sam2_opt_amd.synthetic import synthetic_frames_u8
from sam2_opt_amd.video_predictor import SAM2VideoPredictor
from sam2_opt_amd.weights import synthetic_state_dict

T = 60
pred = SAM2VideoPredictor("large", state_dict=synthetic_state_dict(get_config("large"), seed=0), encode_batch=8)
u8 = synthetic_frames_u8(seed=2, num_frames=T)
rs = np.random.RandomState(0)
for K in (1, 2, 4, 8):
    for ob in (1, 8):
        pred.object_batch = ob
        st = pred.init_state(frames_u8=u8, video_height=1024, video_width=1024)
        for k in range(K):
            pred.add_new_points_or_box(st, 0, k + 1, points=(rs.rand(1, 2) * 800 + 100).astype(np.float32), labels=np.array([1], np.int32))
        for it in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = sum(1 for _ in pred.propagate_in_video(st))
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{K} objects, object_batch={ob}: {n / dt:7.1f} frames/s  ({1e3 * dt / n:.2f} ms/frame, {1e3 * dt / n / K:.2f} ms per object-frame)", flush=True)
        pred.reset_state(st)
pred.release()
