"""How many torch CPU threads should bench.py's cpu_baseline leg use on this box?  Times the oracle's image encoder (the
dominant part of a frame) and one memory-attention call (L = 7, P = 64) at several thread counts."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import sam2_ref as R
from sam2_opt_amd.config import get_config
from sam2_opt_amd.synthetic import randn, synthetic_image_normed
from sam2_opt_amd.weights import synthetic_state_dict

cfg = get_config("large")
sd = synthetic_state_dict(cfg, seed=0)
img = synthetic_image_normed(seed=1)
ma = (randn(51, 4096, 1, 256), randn(52, 7, 4096, 1, 64), randn(53, 4096, 1, 256), randn(54, 7, 4096, 1, 64), randn(55, 64, 1, 64), randn(56, 64, 1, 64))
print("cpu_count", os.cpu_count(), flush=True)
for n in [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64, 128]:
    torch.set_num_threads(n)
    with torch.inference_mode():
        R.image_encoder(img, sd, cfg)
        t0 = time.perf_counter()
        R.image_encoder(img, sd, cfg)
        t1 = time.perf_counter()
        R.memory_attention(*ma, sd, cfg)
        t2 = time.perf_counter()
    print(f"threads {n}: encoder {t1 - t0:.2f} s, memory attention {t2 - t1:.2f} s", flush=True)
