"""CPU experiment: mask-logit error of an N-frame propagation when selected op classes use f16 operands.
See tools/precision_sim.py.  attention modes per head_dim class (72 = Hiera, 256 = memory attention):
exact | f16 (q,k,v rounded) | qk16 (v exact) | v16 (q,k exact).

    python tools/precision_sim_video.py [frames]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

from oracle import sam2_ref as R
from sam2_opt_amd.config import get_config
from sam2_opt_amd.synthetic import normalize_frames, synthetic_frames_u8
from sam2_opt_amd.weights import synthetic_state_dict


def h(x):
    return x.half().float()


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    torch.set_num_threads(8)
    cfg = get_config("large")
    sd = synthetic_state_dict(cfg, seed=0)
    frames = normalize_frames(synthetic_frames_u8(seed=2, num_frames=T), cfg)
    click = (np.array([[512.0, 512.0]], np.float32), np.array([1], np.int32))
    orig_lin, orig_sdpa, orig_conv = R._lin, F.scaled_dot_product_attention, F.conv2d

    def run(lin16, a72, a256):
        def lin(x, sd_, p):
            w, b = sd_[p + ".weight"], sd_[p + ".bias"]
            return F.linear(h(x), h(w), b) if lin16 else F.linear(x, w, b)

        def conv(x, w, b=None, *a, **kw):
            return orig_conv(h(x), h(w), b, *a, **kw) if lin16 else orig_conv(x, w, b, *a, **kw)

        def sdpa(q, k, v, *a, **kw):
            mode = {72: a72, 256: a256}.get(q.shape[-1], "exact")
            if mode in ("f16", "qk16"):
                q, k = h(q), h(k)
            if mode in ("f16", "v16"):
                v = h(v)
            return orig_sdpa(q, k, v, *a, **kw)
        R._lin, F.scaled_dot_product_attention, F.conv2d = lin, sdpa, conv
        try:
            with torch.inference_mode():
                vo = R.VideoOracle(sd, cfg, frames)
                vo.add_new_points(0, *click)
                return {t: m.clone() for t, m in vo.propagate()}
        finally:
            R._lin, F.scaled_dot_product_attention, F.conv2d = orig_lin, orig_sdpa, orig_conv

    ref = run(False, "exact", "exact")
    for cfgm in ((True, "f16", "f16"), (False, "f16", "f16"), (False, "f16", "exact"), (False, "exact", "f16"), (False, "qk16", "qk16"), (False, "v16", "v16")):
        got = run(*cfgm)
        worst = [0, 0, 0]
        for t in ref:
            d = got[t] - ref[t]
            worst[0] = max(worst[0], float(d.abs().max() / ref[t].abs().max()))
            worst[1] = max(worst[1], float(d.norm() / ref[t].norm()))
            worst[2] = max(worst[2], float(((got[t] > 0) != (ref[t] > 0)).float().mean()))
        print(f"lin16={int(cfgm[0])} attn72={cfgm[1]:5s} attn256={cfgm[2]:5s}: max {worst[0]:.2e} L2 {worst[1]:.2e} pix {worst[2]:.2e}", flush=True)


if __name__ == "__main__":
    main()
