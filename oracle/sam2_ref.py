"""CPU oracle for the SAM 2.1 video/image hot path  --  TEST INFRASTRUCTURE ONLY.

A from-scratch fp32 PyTorch restatement of the reference's `backend="torch"`
arithmetic for every plug point on the hot path, written as pure
functions of a checkpoint `state_dict`.  Only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may import it; the product package
(`sam2_opt_amd/`) never does and fails loudly when its HIP library is missing.

Parity pin: `oracle/gen_golden.py` runs the real reference (imported from
/root/reference through oracle/ref_import.py) on seeded inputs and commits its
outputs under `tests/golden/`; `tests/test_oracle_vs_golden.py` checks this file
against those vectors (<=1e-4 abs on O(1..10) activations).  The reference has no
tests of its own for this path, so those generated vectors are the pin.

Every function cites the reference file:line it restates (paths relative to
/root/reference/sam2/sam2/).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from sam2_opt_amd.config import hiera_block_specs

NO_OBJ_SCORE = -1024.0  # modeling/sam2_base_official.py:21


# ----------------------------------------------------------------------------- helpers
def _lin(x, sd, p):
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"])


def _ln(x, sd, p, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def _ln2d(x, sd, p, eps=1e-6):
    """LayerNorm2d over the channel dim of NCHW (modeling/sam2_utils.py:141-153)."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return sd[p + ".weight"][:, None, None] * x + sd[p + ".bias"][:, None, None]


def _mlp(x, sd, p, n, act=F.relu, sigmoid=False):
    """sam2_utils.MLP (modeling/sam2_utils.py:112-136)."""
    for i in range(n):
        x = _lin(x, sd, f"{p}.layers.{i}")
        if i < n - 1:
            x = act(x)
    return torch.sigmoid(x) if sigmoid else x


def sine_pe_2d(H, W, num_pos_feats, temperature=10000.0):
    """PositionEmbeddingSine._pe (modeling/position_encoding.py:90-125), normalize=True,
    scale=2*pi.  Returns (num_pos_feats, H, W)."""
    n = num_pos_feats // 2
    y = torch.arange(1, H + 1, dtype=torch.float32)
    x = torch.arange(1, W + 1, dtype=torch.float32)
    y = y / (y[-1] + 1e-6) * (2 * math.pi)
    x = x / (x[-1] + 1e-6) * (2 * math.pi)
    dim_t = torch.arange(n, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / n)
    px = x[:, None] / dim_t          # (W, n)
    py = y[:, None] / dim_t          # (H, n)
    px = torch.stack((px[:, 0::2].sin(), px[:, 1::2].cos()), dim=2).flatten(1)
    py = torch.stack((py[:, 0::2].sin(), py[:, 1::2].cos()), dim=2).flatten(1)
    pos = torch.cat((py[:, None, :].expand(H, W, n), px[None, :, :].expand(H, W, n)), dim=2)
    return pos.permute(2, 0, 1).contiguous()


def sine_pe_1d(pos_inds, dim, temperature=10000.0):
    """get_1d_sine_pe (modeling/sam2_utils.py:64-74)."""
    pe_dim = dim // 2
    dim_t = torch.arange(pe_dim, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / pe_dim)
    e = pos_inds.unsqueeze(-1) / dim_t
    return torch.cat([e.sin(), e.cos()], dim=-1)


# ----------------------------------------------------------------------------- image encoder (a1-a7)
def hiera_pos_embed(sd, H, W):
    """Hiera._get_pos_embed (modeling/backbones/hieradet.py:273-281) -> (1,H,W,C)."""
    t = "image_encoder.trunk."
    pe = F.interpolate(sd[t + "pos_embed"], size=(H, W), mode="bicubic")
    win = sd[t + "pos_embed_window"]
    pe = pe + win.tile([1, 1, H // win.shape[2], W // win.shape[3]])
    return pe.permute(0, 2, 3, 1)


def _window_partition(x, w):
    """modeling/backbones/utils.py:16-36 (with zero padding when w does not divide)."""
    B, H, W, C = x.shape
    ph, pw = (w - H % w) % w, (w - W % w) % w
    x = F.pad(x, (0, 0, 0, pw, 0, ph))
    Hp, Wp = H + ph, W + pw
    x = x.view(B, Hp // w, w, Wp // w, w, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, w, w, C)
    return x, (Hp, Wp)


def _window_unpartition(xw, w, pad_hw, hw):
    """modeling/backbones/utils.py:39-60."""
    Hp, Wp = pad_hw
    H, W = hw
    B = xw.shape[0] // (Hp * Wp // w // w)
    x = xw.reshape(B, Hp // w, Wp // w, w, w, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, -1)
    return x[:, :H, :W, :]


def _pool2(x):
    """do_pool with MaxPool2d(2,2) on NHWC (hieradet.py:20-36)."""
    return F.max_pool2d(x.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)


def hiera_block(x, sd, spec):
    """MultiScaleBlock.forward + MultiScaleAttention.forward (hieradet.py:56-81,:134-166)."""
    p = f"image_encoder.trunk.blocks.{spec['idx']}."
    heads, w = spec["heads"], spec["window"]
    shortcut = x
    xn = _ln(x, sd, p + "norm1", 1e-6)
    if spec["dim"] != spec["dim_out"]:
        shortcut = _pool2(_lin(xn, sd, p + "proj"))
    H, W = xn.shape[1:3]
    pad_hw = (H, W)
    if w > 0:
        xn, pad_hw = _window_partition(xn, w)
    Bw, Hw, Ww, _ = xn.shape
    qkv = _lin(xn, sd, p + "attn.qkv").reshape(Bw, Hw * Ww, 3, heads, -1)
    q, k, v = torch.unbind(qkv, 2)
    if spec["q_pool"]:
        q = _pool2(q.reshape(Bw, Hw, Ww, -1))
        Hw, Ww = q.shape[1:3]
        q = q.reshape(Bw, Hw * Ww, heads, -1)
    o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2))
    o = o.transpose(1, 2).reshape(Bw, Hw, Ww, -1)
    o = _lin(o, sd, p + "attn.proj")
    if spec["q_pool"]:
        w = w // 2
        H, W = shortcut.shape[1:3]
        pad_hw = (H + (w - H % w) % w, W + (w - W % w) % w) if w > 0 else (H, W)
    if spec["window"] > 0:
        o = _window_unpartition(o, w, pad_hw, (H, W))
    x = shortcut + o
    h = _ln(x, sd, p + "norm2", 1e-6)
    h = _lin(F.gelu(_lin(h, sd, p + "mlp.layers.0")), sd, p + "mlp.layers.1")
    return x + h


def hiera_trunk(img, sd, cfg, return_blocks=None):
    """Hiera.forward (hieradet.py:283-299).  img (B,3,S,S) normalised.  Returns the
    per-stage NCHW feature maps.  `return_blocks`: optional dict filled with selected
    intermediate block outputs (NHWC) for block-level parity tests."""
    t = "image_encoder.trunk."
    x = F.conv2d(img, sd[t + "patch_embed.proj.weight"], sd[t + "patch_embed.proj.bias"], stride=4, padding=3)
    x = x.permute(0, 2, 3, 1)
    x = x + hiera_pos_embed(sd, x.shape[1], x.shape[2])
    if return_blocks is not None and -1 in return_blocks:
        return_blocks[-1] = x.clone()
    outs = []
    for spec in hiera_block_specs(cfg):
        x = hiera_block(x, sd, spec)
        if return_blocks is not None and spec["idx"] in return_blocks:
            return_blocks[spec["idx"]] = x.clone()
        if spec["stage_end"]:
            outs.append(x.permute(0, 3, 1, 2))
    return outs


def image_encoder(img, sd, cfg, return_blocks=None):
    """SAM2Base.inference_image_torch (modeling/sam2_base_official.py:566-582) =
    ImageEncoder.forward (backbones/image_encoder.py:29-42) + FpnNeck.forward (:102-134)
    + conv_s0/conv_s1.  Returns the reference's 7-tuple."""
    xs = hiera_trunk(img, sd, cfg, return_blocks)
    n = len(xs) - 1
    out = [None] * len(xs)
    prev = None
    for i in range(n, -1, -1):
        pfx = f"image_encoder.neck.convs.{n - i}.conv"
        lat = F.conv2d(xs[i], sd[pfx + ".weight"], sd[pfx + ".bias"])
        if i in cfg["fpn_top_down_levels"] and prev is not None:
            prev = lat + F.interpolate(prev, scale_factor=2.0, mode="nearest")
        else:
            prev = lat
        out[i] = prev
    out = out[: len(out) - cfg["scalp"]]
    B = img.shape[0]
    pos = [sine_pe_2d(o.shape[-2], o.shape[-1], cfg["d_model"])[None].repeat(B, 1, 1, 1) for o in out]
    d = "sam_mask_decoder."
    f0 = F.conv2d(out[0], sd[d + "conv_s0.weight"], sd[d + "conv_s0.bias"])
    f1 = F.conv2d(out[1], sd[d + "conv_s1.weight"], sd[d + "conv_s1.bias"])
    return (out[2], pos[0], pos[1], pos[2], f0, f1, out[2])


def set_image_e2e(img01, sd, cfg):
    """SAM2ImagePredictor.set_image_e2e_torch (sam2_image_predictor.py:252-266): input in
    [0,1] un-normalised; Normalize -> forward_image -> + no_mem_embed on the 64x64 level."""
    mean = torch.tensor(cfg["img_mean"]).view(1, 3, 1, 1)
    std = torch.tensor(cfg["img_std"]).view(1, 3, 1, 1)
    o = image_encoder((img01 - mean) / std, sd, cfg)
    f2 = o[6] + sd["no_mem_embed"].view(1, -1, 1, 1)
    return o[4], o[5], f2


# ----------------------------------------------------------------------------- memory attention (a9-a11)
def rope_tables(end_x, end_y, dim, theta=10000.0):
    """compute_axial_rope_cos_sin + the `[..., ::2]` selection of apply_rotary_emb
    (modeling/position_encoding_fix.py:172-205).  Returns cos, sin of shape
    (end_x*end_y, dim//2): pair p<dim/4 rotates by x*theta^(-4p/dim), else by y*..."""
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2).float() / dim))
    t = torch.arange(end_x * end_y, dtype=torch.float32)
    tx = (t % end_x).float()
    ty = torch.div(t, end_x, rounding_mode="floor").float()
    fr = torch.cat([torch.outer(tx, freqs), torch.outer(ty, freqs)], dim=-1)  # (S, dim)
    return fr.cos()[:, ::2].contiguous(), fr.sin()[:, ::2].contiguous()


def apply_rope(x, cos, sin):
    x1, x2 = x[..., ::2], x[..., 1::2]
    return torch.stack([x1 * cos - x2 * sin, x1 * sin + x2 * cos], dim=-1).flatten(-2)


def _rope_attention(q_in, k_in, v_in, sd, p, n_exclude, cos, sin):
    """RoPEAttention.forward (modeling/sam/transformer.py:345-424); 1 head, batch-first."""
    q = _lin(q_in, sd, p + ".q_proj")
    k = _lin(k_in, sd, p + ".k_proj")
    v = _lin(v_in, sd, p + ".v_proj")
    q = apply_rope(q, cos, sin)
    Nk, S = k.shape[-2], q.shape[-2]
    n_rope = Nk - n_exclude
    rep = Nk // S
    ck, sk = cos.repeat(rep, 1), sin.repeat(rep, 1)
    if n_exclude > 0:
        k = torch.cat([apply_rope(k[:, :n_rope], ck[:n_rope], sk[:n_rope]), k[:, n_rope:]], dim=1)
    else:
        k = apply_rope(k, ck, sk)
    o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
    return _lin(o, sd, p + ".out_proj")


def memory_attention(curr, memory, curr_pos, memory_pos, memory_exclude, memory_pos_exclude, sd, cfg):
    """MemoryAttention.inference_memory_attention_torch (modeling/memory_attention.py:299-349)
    with MemoryAttentionLayer.forward (:93-109).  Shapes as the plug: curr (S,N,256),
    memory (L,S,N,64), memory_exclude (P,N,64); returns (S,N,256)."""
    mem = memory.flatten(0, 1)
    mpos = memory_pos.flatten(0, 1)
    P = memory_exclude.shape[0]
    if P > 0:
        mem = torch.cat([mem, memory_exclude], dim=0)
        mpos = torch.cat([mpos, memory_pos_exclude], dim=0)
    x = (curr + 0.1 * curr_pos).transpose(0, 1)
    mem, mpos = mem.transpose(0, 1), mpos.transpose(0, 1)
    fs = cfg["rope_feat_size"]
    cos, sin = rope_tables(fs, fs, cfg["d_model"], cfg["rope_theta"])
    for l in range(cfg["memattn_layers"]):
        p = f"memory_attention.layers.{l}"
        h = _ln(x, sd, p + ".norm1", 1e-5)
        x = x + _rope_attention(h, h, h, sd, p + ".self_attn", 0, cos, sin)
        h = _ln(x, sd, p + ".norm2", 1e-5)
        x = x + _rope_attention(h, mem + mpos, mem, sd, p + ".cross_attn_image", P, cos, sin)
        h = _ln(x, sd, p + ".norm3", 1e-5)
        x = x + _lin(F.relu(_lin(h, sd, p + ".linear1")), sd, p + ".linear2")
    return _ln(x, sd, "memory_attention.norm", 1e-5).transpose(0, 1)


# ----------------------------------------------------------------------------- prompt encoder (a15)
def _pe_random(coords01, sd):
    """PositionEmbeddingRandom._pe_encoding (modeling/position_encoding.py:148-155)."""
    g = sd["sam_prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"]
    c = (2 * coords01 - 1) @ g
    c = 2 * np.pi * c
    return torch.cat([torch.sin(c), torch.cos(c)], dim=-1)


def dense_pe(sd, cfg):
    """PromptEncoder.get_dense_pe (modeling/sam/prompt_encoder.py:113-122) -> (1,256,64,64)."""
    h = w = cfg["image_size"] // cfg["backbone_stride"]
    grid = torch.ones((h, w), dtype=torch.float32)
    y = (grid.cumsum(dim=0) - 0.5) / h
    x = (grid.cumsum(dim=1) - 0.5) / w
    return _pe_random(torch.stack([x, y], dim=-1), sd).permute(2, 0, 1)[None]


def prompt_encoder(points, labels, sd, cfg, mask_input=None):
    """PromptEncoder.inference_prompt_torch (modeling/sam/prompt_encoder.py:215-231) for
    points (+pad point) and optional dense mask prompt.  points (B,Np,2) px, labels (B,Np)."""
    pe = "sam_prompt_encoder."
    B = points.shape[0]
    pts = points + 0.5
    pts = torch.cat([pts, torch.zeros(B, 1, 2)], dim=1)
    lab = torch.cat([labels.to(torch.float32), -torch.ones(B, 1)], dim=1)
    emb = _pe_random(pts / cfg["image_size"], sd)
    emb = torch.where((lab == -1)[..., None], torch.zeros_like(emb) + sd[pe + "not_a_point_embed.weight"], emb)
    for i in range(4):
        emb = torch.where((lab == i)[..., None], emb + sd[pe + f"point_embeddings.{i}.weight"], emb)
    s = cfg["image_size"] // cfg["backbone_stride"]
    if mask_input is None:
        dense = sd[pe + "no_mask_embed.weight"].reshape(1, -1, 1, 1).expand(B, -1, s, s)
    else:
        m = F.conv2d(mask_input, sd[pe + "mask_downscaling.0.weight"], sd[pe + "mask_downscaling.0.bias"], stride=2)
        m = F.gelu(_ln2d(m, sd, pe + "mask_downscaling.1"))
        m = F.conv2d(m, sd[pe + "mask_downscaling.3.weight"], sd[pe + "mask_downscaling.3.bias"], stride=2)
        m = F.gelu(_ln2d(m, sd, pe + "mask_downscaling.4"))
        dense = F.conv2d(m, sd[pe + "mask_downscaling.6.weight"], sd[pe + "mask_downscaling.6.bias"])
    return emb, dense


# ----------------------------------------------------------------------------- mask decoder (a13, a14)
def _attn(q, k, v, sd, p, heads):
    """sam.transformer.Attention.forward (modeling/sam/transformer.py:264-294)."""
    q, k, v = _lin(q, sd, p + ".q_proj"), _lin(k, sd, p + ".k_proj"), _lin(v, sd, p + ".v_proj")

    def split(x):
        b, n, c = x.shape
        return x.reshape(b, n, heads, c // heads).transpose(1, 2)

    o = F.scaled_dot_product_attention(split(q), split(k), split(v))
    b, h, n, c = o.shape
    return _lin(o.transpose(1, 2).reshape(b, n, h * c), sd, p + ".out_proj")


def two_way_transformer(src, pos_src, tokens, sd, cfg):
    """TwoWayTransformer.forward + TwoWayAttentionBlock.forward (transformer.py:98-141,:185-219)."""
    t = "sam_mask_decoder.transformer."
    H = cfg["dec_heads"]
    keys = src.flatten(2).permute(0, 2, 1)
    kpe = pos_src.flatten(2).permute(0, 2, 1)
    q, qpe = tokens, tokens
    for l in range(cfg["dec_depth"]):
        p = f"{t}layers.{l}."
        if l == 0:
            q = _attn(q, q, q, sd, p + "self_attn", H)
        else:
            qq = q + qpe
            q = q + _attn(qq, qq, q, sd, p + "self_attn", H)
        q = _ln(q, sd, p + "norm1", 1e-5)
        q = _ln(q + _attn(q + qpe, keys + kpe, keys, sd, p + "cross_attn_token_to_image", H), sd, p + "norm2", 1e-5)
        q = _ln(q + _mlp(q, sd, p + "mlp", 2), sd, p + "norm3", 1e-5)
        keys = _ln(keys + _attn(keys + kpe, q + qpe, q, sd, p + "cross_attn_image_to_token", H), sd, p + "norm4", 1e-5)
    q = q + _attn(q + qpe, keys + kpe, keys, sd, t + "final_attn_token_to_image", H)
    return _ln(q, sd, t + "norm_final_attn", 1e-5), keys


def predict_masks(src, tokens, pos_src, hr0, hr1, sd, cfg):
    """MaskDecoder.inference_predict_masks_torch (modeling/sam/mask_decoder.py:262-316)."""
    d = "sam_mask_decoder."
    b, c, h, w = src.shape
    hs, keys = two_way_transformer(src, pos_src, tokens, sd, cfg)
    iou_tok = hs[:, 1]
    mask_toks = hs[:, 2:6]
    x = keys.transpose(1, 2).view(b, c, h, w)
    up = F.conv_transpose2d(x, sd[d + "output_upscaling.0.weight"], sd[d + "output_upscaling.0.bias"], stride=2)
    up = F.gelu(_ln2d(up + hr1, sd, d + "output_upscaling.1"))
    up = F.conv_transpose2d(up, sd[d + "output_upscaling.3.weight"], sd[d + "output_upscaling.3.bias"], stride=2)
    up = F.gelu(up + hr0)
    hyper = torch.stack([_mlp(mask_toks[:, i], sd, f"{d}output_hypernetworks_mlps.{i}", 3) for i in range(4)], dim=1)
    b2, c2, h2, w2 = up.shape
    masks = (hyper @ up.view(b2, c2, h2 * w2)).view(b2, -1, h2, w2)
    iou = _mlp(iou_tok, sd, d + "iou_prediction_head", 3, sigmoid=True)
    obj = _mlp(hs[:, 0], sd, d + "pred_obj_score_head", 3)
    return masks, iou, mask_toks, obj


def decoder_tokens(sparse, sd):
    """Token assembly of MaskDecoder.predict_masks (mask_decoder.py:186-202)."""
    d = "sam_mask_decoder."
    out = torch.cat([sd[d + "obj_score_token.weight"], sd[d + "iou_token.weight"], sd[d + "mask_tokens.weight"]], dim=0)
    return torch.cat((out[None].expand(sparse.shape[0], -1, -1), sparse), dim=1)


def _stability_select(masks, iou, cfg):
    """MaskDecoder._dynamic_multimask_via_stability (mask_decoder.py:346-382)."""
    multi, miou = masks[:, 1:], iou[:, 1:]
    best = torch.argmax(miou, dim=-1)
    bi = torch.arange(miou.shape[0])
    best_m, best_i = multi[bi, best].unsqueeze(1), miou[bi, best].unsqueeze(1)
    single, siou = masks[:, 0:1], iou[:, 0:1]
    flat = single.flatten(-2)
    dl = cfg["dynamic_multimask_stability_delta"]
    ai = torch.sum(flat > dl, dim=-1).float()
    au = torch.sum(flat > -dl, dim=-1).float()
    stab = torch.where(au > 0, ai / au, torch.ones_like(au))
    ok = stab >= cfg["dynamic_multimask_stability_thresh"]
    return (torch.where(ok[..., None, None].expand_as(single), single, best_m),
            torch.where(ok.expand_as(siou), siou, best_i))


def mask_decoder(image_embeddings, sparse, dense, hr0, hr1, multimask_output, repeat_image, sd, cfg):
    """MaskDecoder.forward + predict_masks (mask_decoder.py:116-224)."""
    tokens = decoder_tokens(sparse, sd)
    src = image_embeddings
    if repeat_image:
        src = torch.repeat_interleave(src, tokens.shape[0] // src.shape[0], dim=0)
    src = src + dense
    pos_src = torch.repeat_interleave(dense_pe(sd, cfg), tokens.shape[0], dim=0)
    masks, iou, mask_toks, obj = predict_masks(src, tokens, pos_src, hr0, hr1, sd, cfg)
    if multimask_output:
        m, i, tok = masks[:, 1:], iou[:, 1:], mask_toks[:, 1:]
    else:
        m, i = _stability_select(masks, iou, cfg)
        tok = mask_toks[:, 0:1]
    return m, i, tok, obj


def sam_heads(pix_feat, hr0, hr1, sd, cfg, points=None, labels=None, mask_input=None, multimask_output=False):
    """SAM2Base._forward_sam_heads (modeling/sam2_base_official.py:338-494)."""
    B = pix_feat.shape[0]
    if points is None:
        points = torch.zeros(B, 1, 2)
        labels = -torch.ones(B, 1, dtype=torch.int32)
    sparse, dense = prompt_encoder(points, labels, sd, cfg, mask_input)
    low_multi, ious, toks, obj = mask_decoder(pix_feat, sparse, dense, hr0, hr1, multimask_output, False, sd, cfg)
    appearing = obj > 0
    low_multi = torch.where(appearing[:, None, None], low_multi, torch.full_like(low_multi, NO_OBJ_SCORE))
    S = cfg["image_size"]
    high_multi = F.interpolate(low_multi, size=(S, S), mode="bilinear", align_corners=False)
    tok = toks[:, 0]
    if multimask_output:
        best = torch.argmax(ious, dim=-1)
        bi = torch.arange(B)
        low, high = low_multi[bi, best].unsqueeze(1), high_multi[bi, best].unsqueeze(1)
        if toks.shape[1] > 1:
            tok = toks[bi, best]
    else:
        low, high = low_multi, high_multi
    ptr = _mlp(tok, sd, "obj_ptr_proj", 3)
    lam = appearing.float()
    ptr = lam * ptr + (1 - lam) * sd["no_obj_ptr"]
    return dict(low_res_multimasks=low_multi, high_res_multimasks=high_multi, ious=ious, low_res_masks=low,
                high_res_masks=high, obj_ptr=ptr, object_score_logits=obj)


def use_mask_as_output(pix_feat_raw, hr0, hr1, mask_inputs, sd, cfg):
    """SAM2Base._use_mask_as_output (modeling/sam2_base_official.py:496-546): a binary mask input (B,1,S,S) becomes the
    output as it is; the SAM heads (on the raw frame features, with mask_downsample(mask) as dense prompt) only supply
    the object pointer."""
    m = mask_inputs.float()
    high = m * 20.0 - 10.0
    low = F.interpolate(high, size=(high.shape[-2] // 4, high.shape[-1] // 4), align_corners=False, mode="bilinear", antialias=True)
    md = F.conv2d(m, sd["mask_downsample.weight"], sd["mask_downsample.bias"], stride=4)
    ptr = sam_heads(pix_feat_raw, hr0, hr1, sd, cfg, None, None, md, False)["obj_ptr"]
    lam = torch.any(m.flatten(1) > 0.0, dim=1)[..., None].float()
    obj = 20.0 * lam - 10.0
    ptr = lam * ptr + (1 - lam) * sd["no_obj_ptr"]
    return dict(low_res_multimasks=low, high_res_multimasks=high, ious=torch.ones(m.shape[0], 1), low_res_masks=low,
                high_res_masks=high, obj_ptr=ptr, object_score_logits=obj)


# ----------------------------------------------------------------------------- memory encoder (a16)
def memory_encoder(pix_feat, masks, sd, cfg):
    """MemoryEncoder.inference_memory_torch (modeling/memory_encoder.py:233-241) with
    MaskDownSampler (:19-60), CXBlock (:64-119).  masks already sigmoid-scaled."""
    p = "memory_encoder."
    m = masks
    for i in range(4):
        pre = f"{p}mask_downsampler.encoder.{3 * i}"
        m = F.conv2d(m, sd[pre + ".weight"], sd[pre + ".bias"], stride=2, padding=1)
        m = F.gelu(_ln2d(m, sd, f"{p}mask_downsampler.encoder.{3 * i + 1}"))
    m = F.conv2d(m, sd[p + "mask_downsampler.encoder.12.weight"], sd[p + "mask_downsampler.encoder.12.bias"])
    x = F.conv2d(pix_feat, sd[p + "pix_feat_proj.weight"], sd[p + "pix_feat_proj.bias"]) + m
    for l in range(2):
        b = f"{p}fuser.layers.{l}."
        h = F.conv2d(x, sd[b + "dwconv.weight"], sd[b + "dwconv.bias"], padding=3, groups=x.shape[1])
        h = _ln2d(h, sd, b + "norm").permute(0, 2, 3, 1)
        h = _lin(F.gelu(_lin(h, sd, b + "pwconv1")), sd, b + "pwconv2")
        x = x + (sd[b + "gamma"] * h).permute(0, 3, 1, 2)
    x = F.conv2d(x, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])
    pos = sine_pe_2d(x.shape[-2], x.shape[-1], cfg["mem_dim"])[None].repeat(x.shape[0], 1, 1, 1)
    return x, pos


def encode_new_memory(pix_feat, high_res_masks, obj_logits, is_mask_from_pts, sd, cfg):
    """SAM2Base._encode_new_memory (modeling/sam2_base_official.py:978-1026)."""
    if cfg["binarize_mask_from_pts_for_mem_enc"] and is_mask_from_pts:
        m = (high_res_masks > 0).float()
    else:
        m = torch.sigmoid(high_res_masks)
    m = m * cfg["sigmoid_scale_for_mem_enc"] + cfg["sigmoid_bias_for_mem_enc"]
    feats, pos = memory_encoder(pix_feat, m, sd, cfg)
    appearing = (obj_logits > 0).float()
    feats = feats + (1 - appearing[..., None, None]) * sd["no_obj_embed_spatial"][..., None, None].expand(*feats.shape)
    return feats, pos


# ----------------------------------------------------------------------------- memory bank assembly (a8)
def closest_cond_frames(frame_idx, cond_outputs, max_num):
    """select_closest_cond_frames (modeling/sam2_utils.py:19-61)."""
    if max_num == -1 or len(cond_outputs) <= max_num:
        return cond_outputs, {}
    assert max_num >= 2
    chosen = {}
    earlier = [t for t in cond_outputs if t < frame_idx]
    later = [t for t in cond_outputs if t >= frame_idx]
    if earlier:
        chosen[max(earlier)] = cond_outputs[max(earlier)]
    if later:
        chosen[min(later)] = cond_outputs[min(later)]
    rest = sorted([t for t in cond_outputs if t not in chosen], key=lambda t: abs(t - frame_idx))
    for t in rest[: max_num - len(chosen)]:
        chosen[t] = cond_outputs[t]
    return chosen, {t: o for t, o in cond_outputs.items() if t not in chosen}


def assemble_memory(frame_idx, cond_outputs, non_cond_outputs, num_frames, sd, cfg, reverse=False, max_cond_frames_in_attn=-1, stride=1):
    """SAM2Base._prepare_memory_conditioned_features step 1 (sam2_base_official.py:823-946),
    forward or reverse tracking, with max_cond_frames_in_attn (:828-832) and memory_temporal_stride_for_eval (:838-868).
    Outputs dicts hold maskmem_features (1,64,64,64) [bf16-rounded], maskmem_pos_enc (1,64,64,64), obj_ptr (1,256).
    Returns the six plug inputs (without curr/curr_pos): memory (L,4096,1,64), memory_pos,
    memory_exclude (P,1,64), memory_pos_exclude."""
    nm = cfg["num_maskmem"]
    C, M = cfg["d_model"], cfg["mem_dim"]
    all_cond = cond_outputs
    cond_outputs, unselected = closest_cond_frames(frame_idx, all_cond, max_cond_frames_in_attn)
    t_pos_and_prevs = [(0, out) for out in cond_outputs.values()]
    for t_pos in range(1, nm):
        t_rel = nm - t_pos
        if t_rel == 1:
            prev_idx = frame_idx + 1 if reverse else frame_idx - 1
        elif reverse:
            prev_idx = -(-(frame_idx + 2) // stride) * stride + (t_rel - 2) * stride
        else:
            prev_idx = ((frame_idx - 2) // stride) * stride - (t_rel - 2) * stride
        t_pos_and_prevs.append((t_pos, non_cond_outputs.get(prev_idx, unselected.get(prev_idx, None))))
    mems, mposs = [], []
    for t_pos, prev in t_pos_and_prevs:
        if prev is None:
            continue
        mems.append(prev["maskmem_features"].float().flatten(2).permute(2, 0, 1))
        enc = prev["maskmem_pos_enc"].flatten(2).permute(2, 0, 1)
        mposs.append(enc + sd["maskmem_tpos_enc"][nm - t_pos - 1])
    max_ptrs = min(num_frames, cfg["max_obj_ptrs_in_encoder"])
    sign = -1 if reverse else 1                  # use_signed_tpos_enc_to_obj_ptrs, only_obj_ptrs_in_the_past_for_eval (:891-905)
    pos_and_ptrs = [((frame_idx - t) * sign, out["obj_ptr"]) for t, out in cond_outputs.items()
                    if (t >= frame_idx if reverse else t <= frame_idx)]
    for t_diff in range(1, max_ptrs):
        t = frame_idx + t_diff if reverse else frame_idx - t_diff
        if t < 0 or t >= num_frames:
            break
        out = non_cond_outputs.get(t, unselected.get(t, None))
        if out is not None:
            pos_and_ptrs.append((t_diff, out["obj_ptr"]))
    memory = torch.stack(mems, dim=0)            # (L,4096,1,64)
    memory_pos = torch.stack(mposs, dim=0)
    if pos_and_ptrs:
        pos_list, ptrs = zip(*pos_and_ptrs)
        obj_ptrs = torch.stack(ptrs, dim=0)      # (n,1,256)
        B = obj_ptrs.shape[1]
        t_diff_max = max_ptrs - 1
        obj_pos = torch.tensor(pos_list, dtype=torch.float32)
        obj_pos = _lin(sine_pe_1d(obj_pos / t_diff_max, C), sd, "obj_ptr_tpos_proj")
        obj_pos = obj_pos.unsqueeze(1).expand(-1, B, M)
        obj_ptrs = obj_ptrs.reshape(-1, B, C // M, M).permute(0, 2, 1, 3).flatten(0, 1)
        obj_pos = obj_pos.repeat_interleave(C // M, dim=0)
    else:
        obj_ptrs = torch.zeros(0, memory.shape[2], M)
        obj_pos = torch.zeros(0, memory.shape[2], M)
    return memory, memory_pos, obj_ptrs, obj_pos


# ----------------------------------------------------------------------------- video predictor (a17)
class VideoOracle:
    """Single-object propagation exactly as SAM2VideoPredictor does it
    (sam2_video_predictor_official.py: init_state :147-205, add_new_points_or_box :266-399, add_new_mask :403-489,
    propagate_in_video_preflight :585-649, propagate_in_video :651-736,
    _run_single_frame_inference :843-909) with fill_hole_area=0 (what the reference does on
    a box without its CUDA extension, utils/misc.py:321-336)."""

    def __init__(self, sd, cfg, frames, video_hw=None):
        self.sd, self.cfg = sd, cfg
        self.frames = frames                      # (T,3,1024,1024) normalised f32
        self.num_frames = frames.shape[0]
        self.video_hw = video_hw or (cfg["image_size"], cfg["image_size"])
        self.cond, self.non_cond = OrderedDict(), OrderedDict()
        self.temp = {"cond": OrderedDict(), "non_cond": OrderedDict()}      # interacted frames not yet consolidated
        self.tracked = {}                         # frame -> reverse flag
        self.points = {}                          # frame -> (pts, lab) accumulated clicks
        self._feat_cache = {}
        self.trace = {}                           # per-frame debug tensors for parity tests

    def _features(self, t):
        if t not in self._feat_cache:
            self._feat_cache = {t: image_encoder(self.frames[t:t + 1], self.sd, self.cfg)}
        return self._feat_cache[t]

    def _video_res(self, low):
        H, W = self.video_hw
        if low.shape[-2:] == (H, W):
            return low
        return F.interpolate(low, size=(H, W), mode="bilinear", align_corners=False)

    def _memory_conditioned(self, t, reverse=False):
        cfg, sd = self.cfg, self.sd
        f = self._features(t)
        S = f[6].shape[-1]
        curr = f[6].flatten(2).permute(2, 0, 1)
        curr_pos = f[3].flatten(2).permute(2, 0, 1)
        mem, mpos, ex, expos = assemble_memory(t, self.cond, self.non_cond, self.num_frames, sd, cfg, reverse)
        pix = memory_attention(curr, mem, curr_pos, mpos, ex, expos, sd, cfg)
        return pix.permute(1, 2, 0).view(1, cfg["d_model"], S, S), (curr, mem, curr_pos, mpos, ex, expos)

    def add_new_points(self, frame_idx, points, labels, normalize_coords=True, clear_old_points=True):
        cfg, sd = self.cfg, self.sd
        pts = torch.as_tensor(points, dtype=torch.float32).reshape(1, -1, 2)
        lab = torch.as_tensor(labels, dtype=torch.int32).reshape(1, -1)
        if normalize_coords:
            pts = pts / torch.tensor([self.video_hw[1], self.video_hw[0]], dtype=torch.float32)
        pts = pts * cfg["image_size"]
        if not clear_old_points and frame_idx in self.points:
            pts = torch.cat([self.points[frame_idx][0], pts], dim=1)
            lab = torch.cat([self.points[frame_idx][1], lab], dim=1)
        self.points[frame_idx] = (pts, lab)
        is_init = frame_idx not in self.tracked
        key = "cond" if is_init else "non_cond"
        prev = self.temp[key].get(frame_idx) or self.cond.get(frame_idx) or self.non_cond.get(frame_idx)
        prev_logits = torch.clamp(prev["pred_masks"], -32.0, 32.0) if prev is not None else None      # :352-366
        f = self._features(frame_idx)
        if is_init:
            pix = f[6] + sd["no_mem_embed"].view(1, -1, 1, 1)      # directly_add_no_mem_embed (:953-957)
        else:
            pix, _ = self._memory_conditioned(frame_idx, self.tracked[frame_idx])      # correction clicks on a tracked frame
        n = lab.shape[1]
        multimask = cfg["multimask_min_pt_num"] <= n <= cfg["multimask_max_pt_num"]
        out = sam_heads(pix, f[4], f[5], sd, cfg, pts, lab, prev_logits, multimask)
        self.temp[key][frame_idx] = dict(pred_masks=out["low_res_masks"], obj_ptr=out["obj_ptr"],
                                         object_score_logits=out["object_score_logits"], maskmem_features=None,
                                         maskmem_pos_enc=None, is_pts=True)
        self.trace[("click", frame_idx)] = out
        return self._video_res(out["low_res_masks"])

    def add_new_mask(self, frame_idx, mask):
        """mask: bool/float (H, W); resized like the reference (:421-434) when it is not image_size^2."""
        cfg, sd = self.cfg, self.sd
        m = torch.as_tensor(mask).float()[None, None]
        S = cfg["image_size"]
        if m.shape[-2:] != (S, S):
            m = (F.interpolate(m, size=(S, S), align_corners=False, mode="bilinear", antialias=True) >= 0.5).float()
        self.points.pop(frame_idx, None)
        key = "cond" if frame_idx not in self.tracked else "non_cond"
        f = self._features(frame_idx)
        out = use_mask_as_output(f[6], f[4], f[5], m, sd, cfg)        # raw features: no memory, no no_mem_embed (:1120-1131)
        self.temp[key][frame_idx] = dict(pred_masks=out["low_res_masks"], obj_ptr=out["obj_ptr"],
                                         object_score_logits=out["object_score_logits"], maskmem_features=None,
                                         maskmem_pos_enc=None, is_pts=False)
        self.trace[("mask", frame_idx)] = out
        return self._video_res(out["low_res_masks"])

    def _preflight(self):
        """propagate_in_video_preflight: interacted frames get their memory (binarised mask) and move to the output dicts."""
        for key, dst in (("non_cond", self.non_cond), ("cond", self.cond)):
            for t, out in self.temp[key].items():
                hi = F.interpolate(out["pred_masks"], size=(self.cfg["image_size"],) * 2, mode="bilinear", align_corners=False)
                f = self._features(t)
                feats, pos = encode_new_memory(f[6], hi, out["object_score_logits"], True, self.sd, self.cfg)
                out["maskmem_features"] = feats.to(torch.bfloat16)
                out["maskmem_pos_enc"] = pos
                dst[t] = out
            self.temp[key].clear()
        for t in self.cond:
            self.non_cond.pop(t, None)

    def track_frame(self, t, reverse=False):
        cfg, sd = self.cfg, self.sd
        f = self._features(t)
        pix, memattn_in = self._memory_conditioned(t, reverse)
        out = sam_heads(pix, f[4], f[5], sd, cfg, None, None, None, True)   # multimask for tracking
        feats, pos = encode_new_memory(f[6], out["high_res_masks"], out["object_score_logits"], False, sd, cfg)
        self.non_cond[t] = dict(pred_masks=out["low_res_masks"], obj_ptr=out["obj_ptr"],
                                object_score_logits=out["object_score_logits"],
                                maskmem_features=feats.to(torch.bfloat16), maskmem_pos_enc=pos)
        self.trace[("track", t)] = dict(out, memattn_in=memattn_in, pix_feat=pix, maskmem_features=feats)
        return out["low_res_masks"]

    def propagate(self, max_frames=None, start_frame_idx=None, reverse=False):
        self._preflight()
        start = min(self.cond) if start_frame_idx is None else start_frame_idx
        if max_frames is None:
            max_frames = self.num_frames
        if reverse:                                  # processing order of propagate_in_video (:675-686)
            order = range(start, max(start - max_frames, 0) - 1, -1) if start > 0 else []
        else:
            order = range(start, min(start + max_frames, self.num_frames - 1) + 1)
        for t in order:
            if t in self.cond:
                low = self.cond[t]["pred_masks"]
            else:
                low = self.track_frame(t, reverse)
            self.tracked[t] = reverse
            yield t, self._video_res(low)


# ----------------------------------------------------------------------------- image predictor (a18)
def image_predict(feats, points, labels, multimask_output, orig_hw, sd, cfg, return_logits=True, mask_threshold=0.0, mask_input=None):
    """SAM2ImagePredictor._predict (sam2_image_predictor.py:487-589) on `feats` = set_image_e2e output for ONE image.
    points (B,Np,2) already in 1024-pixel units, labels (B,Np) - a box arrives as its two corners with labels 2 / 3 in front of
    the user's points (:509-522); mask_input (B,1,256,256) low-res logits as dense prompt."""
    f0, f1, f2 = feats
    B = points.shape[0]
    sparse, dense = prompt_encoder(points, labels, sd, cfg, mask_input)
    low, iou, _, _ = mask_decoder(f2, sparse, dense, f0, f1, multimask_output, B > 1, sd, cfg)
    masks = F.interpolate(low.float(), orig_hw, mode="bilinear", align_corners=False)
    low = torch.clamp(low, -32.0, 32.0)
    if not return_logits:
        masks = masks > mask_threshold
    return masks, iou, low
