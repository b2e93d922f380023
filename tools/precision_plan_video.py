"""CPU experiment: mask-logit error of a T-frame propagation under SELECTIVE operand precision plans (see
tools/precision_shares.py for the per-operand shares on the encoder).  A plan maps (parameter path, operand in {x, w}) to
'exact' | 'f16'; attention operands per head_dim class (72 = Hiera, 256 = memory attention, others = decoder) to a subset
of 'qkpv' that is rounded to f16.  Prints the three metrics of tests/test_video_gpu.py (max-abs / max|ref|, rel L2,
binarised disagreement), worst frame.

    python tools/precision_plan_video.py [frames] [plan ...]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

from oracle import sam2_ref as R
from sam2_opt_amd.config import get_config, hiera_block_specs
from sam2_opt_amd.synthetic import normalize_frames, synthetic_frames_u8
from sam2_opt_amd.weights import synthetic_state_dict


def h(x):
    return x.half().float()


def kind_of(p):
    for suf, k in (("attn.qkv", "qkv"), ("attn.proj", "proj"), ("mlp.layers.0", "fc1"), ("mlp.layers.1", "fc2")):
        if p.endswith(suf):
            return k
    return "short"


def make_plans(stage_of):
    def trunk(p):
        return p.startswith("image_encoder.trunk.blocks.")

    def blk(p):
        return int(p.split(".")[3])

    def enc(p):
        return p.startswith("image_encoder.") or p.startswith("sam_mask_decoder.conv_s")

    def plan_f16(p, o):
        return "f16"

    def plan_trk16(p, o):                      # encoder exact, everything else f16
        return "exact" if enc(p) else "f16"

    def plan_a(p, o, s12="exact", s3_mlp_w="f16", s4="f16", rest="exact"):
        if not trunk(p):
            return "exact" if enc(p) else rest
        s, kd = stage_of[blk(p)], kind_of(p)
        if s <= 2:
            if s12 == "exact":
                return "exact"
            if s12 == "w":                      # weights exact, x f16 except the projection input
                return "exact" if (o == "w" or kd == "proj") else "f16"
            return "f16"
        if s == 3:
            if kd == "qkv":
                return "exact" if o == "w" else "f16"
            if kd in ("proj", "short"):
                return "exact"
            return s3_mlp_w if o == "w" else "f16"
        return s4
    plans = {
        "all_f16": (plan_f16, {72: "qkpv", 256: "qkpv"}),
        "enc_exact_trk_f16": (plan_trk16, {256: "qkpv"}),
        "A": (plan_a, {72: "pv"}),
        "A_trk16": (lambda p, o: plan_a(p, o, rest="f16"), {72: "pv", 256: "qkpv"}),
        "A_flash16": (plan_a, {72: "pv", 256: "qkpv"}),
        "B_s12w": (lambda p, o: plan_a(p, o, s12="w"), {72: "pv", 256: "qkpv"}),
        "C_mlpw": (lambda p, o: plan_a(p, o, s3_mlp_w="exact"), {72: "pv", 256: "qkpv"}),
        "D_s12f16": (lambda p, o: plan_a(p, o, s12="f16"), {72: "pv", 256: "qkpv"}),
        "E_attn16": (plan_a, {72: "qkpv", 256: "qkpv"}),
        "W_all": (lambda p, o: "exact" if o == "w" else "f16", {72: "pv", 256: "qkpv"}),
    }
    return plans


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    which = sys.argv[2:]
    torch.set_num_threads(8)
    cfg = get_config("large")
    sd = synthetic_state_dict(cfg, seed=0)
    stage_of, st = {}, 1
    for s in hiera_block_specs(cfg):
        stage_of[s["idx"]] = st
        st += s["stage_end"]
    frames = normalize_frames(synthetic_frames_u8(seed=2, num_frames=T), cfg)
    click = (np.array([[512.0, 512.0]], np.float32), np.array([1], np.int32))
    orig_lin, orig_sdpa, orig_conv = R._lin, F.scaled_dot_product_attention, F.conv2d
    name_of = {id(v): k[: -len(".weight")] for k, v in sd.items() if k.endswith(".weight")}

    def run(plan, attn):
        def rnd(v, f):
            return v if f == "exact" else h(v)

        def lin(x, sd_, p):
            return F.linear(rnd(x, plan(p, "x")), rnd(sd_[p + ".weight"], plan(p, "w")), sd_[p + ".bias"])

        def conv(x, w, b=None, *a, **kw):
            p = name_of.get(id(w), "?")
            return orig_conv(rnd(x, plan(p, "x")), rnd(w, plan(p, "w")), b, *a, **kw)

        def sdpa(q, k, v, *a, **kw):
            m = attn.get(q.shape[-1], "")
            if not m:
                return orig_sdpa(q, k, v, *a, **kw)
            if "q" in m:
                q, k = h(q), h(k)
            if "v" in m:
                v = h(v)
            if "p" in m and not a and not kw:
                s = (q @ k.transpose(-1, -2)) * (q.shape[-1] ** -0.5)
                e = torch.exp(s - s.amax(-1, keepdim=True))
                return (h(e) @ v) / e.sum(-1, keepdim=True)
            return orig_sdpa(q, k, v, *a, **kw)
        R._lin, F.scaled_dot_product_attention, F.conv2d = lin, sdpa, conv
        try:
            with torch.inference_mode():
                vo = R.VideoOracle(sd, cfg, frames)
                vo.add_new_points(0, *click)
                return {t: m.clone() for t, m in vo.propagate()}
        finally:
            R._lin, F.scaled_dot_product_attention, F.conv2d = orig_lin, orig_sdpa, orig_conv

    ref = run(lambda p, o: "exact", {})
    plans = make_plans(stage_of)
    for name in (which or list(plans)):
        plan, attn = plans[name]
        got = run(plan, attn)
        worst = [0.0, 0.0, 0.0]
        for t in ref:
            d = got[t] - ref[t]
            worst[0] = max(worst[0], float(d.abs().max() / ref[t].abs().max()))
            worst[1] = max(worst[1], float(d.norm() / ref[t].norm()))
            worst[2] = max(worst[2], float(((got[t] > 0) != (ref[t] > 0)).float().mean()))
        print(f"{name:20s}: max {worst[0]:.2e} L2 {worst[1]:.2e} pix {worst[2]:.2e}", flush=True)


if __name__ == "__main__":
    main()
