"""GPU box: which block of the trunk first leaves its tolerance in the outlier-channel scenario (tests/test_outliers_gpu.py)?
One MultiScaleBlock at a time on the oracle's input for that block, in the given precision mode(s)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import sam2_ref as R
from oracle.gen_golden import OUTLIER_GAIN
from sam2_opt_amd.config import get_config
from sam2_opt_amd.native import Engine
from sam2_opt_amd.synthetic import synthetic_image_normed
from sam2_opt_amd.weights import synthetic_state_dict

cfg = get_config("large")
sd = synthetic_state_dict(cfg, seed=0, undamped=True, outlier_gain=float(os.environ.get("GAIN", OUTLIER_GAIN)))
img = synthetic_image_normed(seed=1)
idxs = (-1, 0, 1, 2, 3, 7, 8, 9, 10, 22, 23, 24, 43, 44, 45, 47)
blocks = {i: None for i in idxs}
torch.set_num_threads(16)
with torch.inference_mode():
    outs = R.image_encoder(img, sd, cfg, blocks)
for mode in sys.argv[1:] or ["f16"]:
    e = Engine("large", state_dict=sd, max_batch=2, precision=mode)
    for i in idxs[1:]:
        if blocks.get(i - 1) is None:
            continue
        x, ref = blocks[i - 1].cuda(), blocks[i]
        out = e.debug_hiera_block(i, x, ref.shape).float().cpu()
        d = out - ref
        print(f"[{mode}] block {i}: max_rel {float(d.abs().max() / ref.abs().max()):.3e} l2 {float(d.norm() / ref.norm()):.3e} "
              f"max|in| {float(x.abs().max()):.3g} max|ref| {float(ref.abs().max()):.3g} finite {bool(torch.isfinite(out).all())}", flush=True)
    e.close()
# end to end: the encoder plug in every mode against the oracle with the same weights
for mode in ("f16", "f16s", "f16x3"):
    e = Engine("large", state_dict=sd, max_batch=2, precision=mode)
    got = e.image_encoder(img.cuda())
    line = f"[{mode}] encoder e2e:"
    for k, n in ((0, "vis"), (4, "fpn0"), (5, "fpn1")):
        d = got[k].float().cpu() - outs[k]
        line += f"  {n}: max {float(d.abs().max() / outs[k].abs().max()):.2e} l2 {float(d.norm() / outs[k].norm()):.2e}"
    print(line, flush=True)
    e.close()
