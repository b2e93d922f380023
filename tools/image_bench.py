"""BASELINE.json configs[4] timing (tuning aid, not the bench line): hiera-large image predictor, a batch of 1024^2 images with
8 independent single-point prompts each.  Prints images/s and the encoder / decoder split."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sam2_opt_amd.config import get_config
from sam2_opt_amd.image_predictor import SAM2ImagePredictor
from sam2_opt_amd.weights import synthetic_state_dict

B = int(os.environ.get("B", "16"))
cfg = get_config("large")
pred = SAM2ImagePredictor("large", state_dict=synthetic_state_dict(cfg, seed=0), max_batch=B)
imgs = [np.random.RandomState(10 + i).randint(0, 256, (1024, 1024, 3)).astype(np.uint8) for i in range(B)]
pts = [(np.random.RandomState(100 + i).rand(8, 1, 2) * 1024).astype(np.float32) for i in range(B)]
lab = np.ones((8, 1), np.int32)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pred.set_image_batch(imgs)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for i in range(B):
        pred._predict(pts[i], lab, None, None, True, True, img_idx=i)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"iter {it}: set_image_batch {1e3*(t1-t0):.1f} ms, 8 prompts x {B} images {1e3*(t2-t1):.1f} ms -> {B/(t2-t0):.1f} images/s", flush=True)
pred.release()
