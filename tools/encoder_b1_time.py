"""BASELINE.json configs[1]: wall time of the image-encoder plug at batch 1 (20 calls after 3 warm-up calls)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam2_opt_amd.native import Engine
from sam2_opt_amd.config import get_config
from sam2_opt_amd.weights import synthetic_state_dict
cfg = get_config("large")
eng = Engine(cfg, state_dict=synthetic_state_dict(cfg, seed=0), max_batch=int(os.environ.get("B", "1")), precision=os.environ.get("PRECISION", "f16s"))
B = int(os.environ.get("B", "1"))
x = torch.randn(B, 3, 1024, 1024, device="cuda")
for _ in range(3):
    eng.image_encoder(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    eng.image_encoder(x)
torch.cuda.synchronize()
print(f"image encoder, batch {B}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per call  (SAM2MI_XS_MINM={os.environ.get('SAM2MI_XS_MINM', 'default')})")
