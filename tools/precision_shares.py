"""CPU experiment (build container, no GPU): which rounding points carry the f16-mode error of the image encoder,
and what the cheaper split schemes leave.  Extends tools/precision_sim.py with per-stage / per-linear / per-operand
selection and with the operand formats a gfx950 kernel can actually feed to an MFMA:

  f16      operand rounded to f16 (the default path)
  exact    operand kept in f32 (what the 3-product f16 split reaches to 2^-22)
  f16+bf8  hi = f16(v), lo = bf8(v - hi): the cross terms hi*lo + lo*hi on the block-scaled fp8 MFMA at twice the f16
           rate, the other operand of a cross term being bf8(hi)
Selection is by a predicate on (block index, kind in {qkv, proj, fc1, fc2, short}, operand in {x, w}).

    python tools/precision_shares.py [quick]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

from oracle import sam2_ref as R
from sam2_opt_amd.config import get_config, hiera_block_specs
from sam2_opt_amd.weights import synthetic_state_dict


def h(x):
    return x.half().float()


def b8(x):
    return x.to(torch.float8_e5m2).float()


def b8_trunc(x):                       # bf8 = the top byte of the f16 pattern (what a v_perm_b32 gives)
    u = x.half().view(torch.int16) & -256
    return u.view(torch.float16).float()


def kind_of(p):
    if p.endswith("attn.qkv"):
        return "qkv"
    if p.endswith("attn.proj"):
        return "proj"
    if p.endswith("mlp.layers.0"):
        return "fc1"
    if p.endswith("mlp.layers.1"):
        return "fc2"
    return "short"


def main():
    quick = len(sys.argv) > 1
    torch.set_num_threads(8)
    cfg = get_config("large")
    sd = synthetic_state_dict(cfg, seed=0)
    specs = hiera_block_specs(cfg)
    stage_of = {}
    st = 1
    for s in specs:
        stage_of[s["idx"]] = st
        if s["stage_end"]:
            st += 1
    img = torch.from_numpy(np.random.RandomState(1).standard_normal((1, 3, 1024, 1024)).astype(np.float32))
    orig_lin, orig_sdpa = R._lin, F.scaled_dot_product_attention

    def run(fmt, attn=None):
        """fmt(blk, kind, operand) -> 'exact' | 'f16' | 'x8' (hi f16 + bf8 lo, rounded) | 'x8t' (bf8 by truncation)"""
        def lin(x, sd_, p):
            w, b = sd_[p + ".weight"], sd_[p + ".bias"]
            if not p.startswith("image_encoder.trunk.blocks."):
                return F.linear(x, w, b)
            blk = int(p.split(".")[3])
            kd = kind_of(p)
            fx, fw = fmt(blk, kd, "x"), fmt(blk, kd, "w")
            xh = x if fx == "exact" else h(x)
            wh = w if fw == "exact" else h(w)
            y = F.linear(xh, wh, b)
            for f, full, hi, other, is_x in ((fx, x, xh, wh, True), (fw, w, wh, xh, False)):
                if f in ("x8", "x8t"):
                    q = b8_trunc if f == "x8t" else b8
                    lo = q((full - hi) * 2048.0) / 2048.0
                    o8 = q(other)
                    y = y + (F.linear(lo, o8) if is_x else F.linear(o8, lo))
            return y

        def sdpa(q, k, v, *a, **kw):
            if attn and q.shape[-1] == 72:
                if "qk" in attn:
                    q, k = h(q), h(k)
                if "v" in attn:
                    v = h(v)
                if "p" in attn:
                    s = (q @ k.transpose(-1, -2)) * (q.shape[-1] ** -0.5)
                    m = s.amax(-1, keepdim=True)
                    e = torch.exp(s - m)
                    return (h(e) @ v) / e.sum(-1, keepdim=True)
            return orig_sdpa(q, k, v, *a, **kw)
        R._lin, F.scaled_dot_product_attention = lin, sdpa
        try:
            with torch.inference_mode():
                out = R.image_encoder(img, sd, cfg)
        finally:
            R._lin, F.scaled_dot_product_attention = orig_lin, orig_sdpa
        return out

    ref = run(lambda *a: "exact")

    def report(name, fmt, attn=None):
        got = run(fmt, attn)
        line = f"{name:44s}"
        for k, nm in ((0, "vis"), (4, "fpn0"), (5, "fpn1")):
            d = got[k] - ref[k]
            line += f"  {nm}: L2 {float(d.norm() / ref[k].norm()):.2e} max {float(d.abs().max() / ref[k].abs().max()):.2e}"
        print(line, flush=True)

    report("all linears f16 (x and w)", lambda b, k, o: "f16")
    report("x only f16", lambda b, k, o: "f16" if o == "x" else "exact")
    report("w only f16", lambda b, k, o: "f16" if o == "w" else "exact")
    for s in (1, 2, 3, 4):
        report(f"stage {s} only f16", lambda b, k, o, s=s: "f16" if stage_of[b] == s else "exact")
    if not quick:
        for kd in ("qkv", "proj", "fc1", "fc2"):
            for op in ("x", "w"):
                report(f"stage 3 {kd} {op} only f16", lambda b, k, o, kd=kd, op=op: "f16" if (stage_of[b] == 3 and k == kd and o == op) else "exact")
    report("x: f16+bf8 lo, w: f16+bf8 lo (rounded)", lambda b, k, o: "x8")
    report("x: f16+bf8 lo, w: f16+bf8 lo (truncated)", lambda b, k, o: "x8t")
    report("x: f16+bf8 lo, w: f16", lambda b, k, o: "x8" if o == "x" else "f16")
    report("stage 3 split (x8), rest f16", lambda b, k, o: "x8" if stage_of[b] == 3 else "f16")
    report("stages 3+4 split (x8), rest f16", lambda b, k, o: "x8" if stage_of[b] >= 3 else "f16")
    report("attention qk f16, linears exact", lambda *a: "exact", "qk")
    report("attention v f16", lambda *a: "exact", "v")
    report("attention p,v f16", lambda *a: "exact", "pv")
    report("attention q,k,p,v f16", lambda *a: "exact", "qkpv")


if __name__ == "__main__":
    main()
