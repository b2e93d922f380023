"""CPU experiment (build container, no GPU): where does the f16-operand error of the encoder come from?

Runs the oracle's image encoder on one synthetic frame with the operands of selected op classes rounded the way the
HIP kernels round them, and prints the error of `vision_features` against the exact fp32 run:
  lin16     x and W of every nn.Linear rounded to f16 (what the default MFMA path does)
  lin_w2    W as a 2-term f16 split (hi + lo), x f16
  lin_x2    x as a 2-term split, W f16
  lin_split both 2-term (3 products hi*hi + hi*lo + lo*hi; the lo*lo term is dropped)
  attn16    q, k, v of every SDPA rounded to f16 (probabilities stay f32 here)
This decides which kernels need the split-f16 precision mode.

    python tools/precision_sim.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

from oracle import sam2_ref as R
from sam2_opt_amd.config import get_config
from sam2_opt_amd.weights import synthetic_state_dict


def h(x):
    return x.half().float()


def split2(x):
    hi = h(x)
    return hi, h(x - hi)


def main():
    torch.set_num_threads(8)
    cfg = get_config("large")
    sd = synthetic_state_dict(cfg, seed=0)
    img = torch.from_numpy(np.random.RandomState(1).standard_normal((1, 3, 1024, 1024)).astype(np.float32))
    orig_lin, orig_sdpa, orig_conv = R._lin, F.scaled_dot_product_attention, F.conv2d

    def run(lin_mode, attn16):
        def lin(x, sd_, p):
            w, b = sd_[p + ".weight"], sd_[p + ".bias"]
            if lin_mode == "exact":
                return F.linear(x, w, b)
            if lin_mode == "lin16":
                return F.linear(h(x), h(w), b)
            xh, xl = split2(x)
            wh, wl = split2(w)
            if lin_mode == "lin_w2":
                return F.linear(xh, wh, b) + F.linear(xh, wl)
            if lin_mode == "lin_x2":
                return F.linear(xh, wh, b) + F.linear(xl, wh)
            return F.linear(xh, wh, b) + F.linear(xh, wl) + F.linear(xl, wh)

        def sdpa(q, k, v, *a, **kw):
            if attn16:
                q, k, v = h(q), h(k), h(v)
            return orig_sdpa(q, k, v, *a, **kw)
        def conv(x, w, b=None, *a, **kw):          # patch embed / neck / conv_s0,s1 (dense convs are GEMMs on the device)
            if lin_mode == "exact":
                return orig_conv(x, w, b, *a, **kw)
            if lin_mode == "lin16":
                return orig_conv(h(x), h(w), b, *a, **kw)
            xh, xl = split2(x)
            wh, wl = split2(w)
            y = orig_conv(xh, wh, b, *a, **kw)
            if lin_mode in ("lin_w2", "lin_split"):
                y = y + orig_conv(xh, wl, None, *a, **kw)
            if lin_mode in ("lin_x2", "lin_split"):
                y = y + orig_conv(xl, wh, None, *a, **kw)
            return y
        R._lin = lin
        F.scaled_dot_product_attention = sdpa
        F.conv2d = conv
        try:
            with torch.inference_mode():
                out = R.image_encoder(img, sd, cfg)
        finally:
            R._lin = orig_lin
            F.scaled_dot_product_attention = orig_sdpa
            F.conv2d = orig_conv
        return out

    ref = run("exact", False)
    names = [0, 4, 5]
    for mode, a16 in (("lin16", True), ("lin16", False), ("exact", True), ("lin_w2", False), ("lin_x2", False), ("lin_split", False), ("lin_split", True)):
        got = run(mode, a16)
        line = f"{mode:10s} attn16={int(a16)}:"
        for k in names:
            r, g = ref[k], got[k]
            if not torch.is_tensor(r) or r.shape != g.shape or r.dim() < 3:
                continue
            d = (g - r)
            if float(d.abs().max()) == 0.0:
                continue
            line += f"  {k}: L2 {float(d.norm() / r.norm()):.2e} max {float(d.abs().max() / r.abs().max()):.2e}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
