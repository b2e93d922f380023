"""Route A - the drop-in route - driven WITHOUT the reference: a single-object tracker whose per-frame host loop does in
PyTorch exactly the glue the reference's SAM2Base.track_step does around its plug points
(/root/reference/sam2/sam2/modeling/sam2_base_official.py: forward_image :548-564, _prepare_memory_conditioned_features
:797-976, _forward_sam_heads :338-494, _encode_new_memory :978-1026, track_step :1114-1179) and calls libsam2mi.so only
through the five plug-level entry points, with the reference's tensor layouts (NCHW / sequence-first fp32):

    sam2mi_image_encoder      <- SAM2Base.inference_image
    sam2mi_memory_attention   <- MemoryAttention.inference_memory_attention_{none,exclude}
    sam2mi_prompt_encoder_ex  <- PromptEncoder.inference_prompt
    sam2mi_mask_decoder       <- MaskDecoder.inference_predict_masks
    sam2mi_memory_encoder     <- MemoryEncoder.inference_memory

This is what `sam2_opt_amd.plugin.speedup_hip(reference_predictor)` costs per frame: every plug call pays the layout
transposes and fp32 I/O of the boundary, the memory bank lives in torch tensors and is re-assembled (torch.cat) every
frame.  `bench.py` times it next to the fused route D (sam2_opt_amd.video_predictor) so that the price of the boundary is a
measured number; tests/test_video_gpu.py holds its masks to the same reference goldens.  Forward propagation from one
click on frame 0, one object - the benchmark scenario - is all it implements.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from .config import get_config
from .memory_select import select_memory
from .native import Engine

NO_OBJ_SCORE = -1024.0


class PlugLevelTracker:
    def __init__(self, model: str = "large", state_dict=None, device=None, precision: str = "f16", lookahead: int = 8):
        from .plugin import LookaheadImagePlug
        self.cfg = get_config(model)
        self.engine = Engine(self.cfg, state_dict=state_dict, max_batch=max(1, lookahead), device=device, precision=precision)
        # the image plug exactly as speedup_hip installs it on the reference (plugin.py): look-ahead over the clip tensor
        self.image_plug = LookaheadImagePlug(self.engine, lookahead) if lookahead > 1 else self.engine.image_encoder
        self.device = self.engine.device
        keys = ["no_mem_embed", "no_obj_ptr", "maskmem_tpos_enc", "no_obj_embed_spatial", "obj_ptr_tpos_proj.weight", "obj_ptr_tpos_proj.bias",
                "sam_mask_decoder.obj_score_token.weight", "sam_mask_decoder.iou_token.weight", "sam_mask_decoder.mask_tokens.weight"]
        keys += [f"obj_ptr_proj.layers.{i}.{w}" for i in range(3) for w in ("weight", "bias")]
        self.w = {k: torch.as_tensor(state_dict[k]).float().to(self.device) for k in keys}
        self.dense_pe = self.engine.dense_pe()
        self.out_tokens = torch.cat([self.w["sam_mask_decoder.obj_score_token.weight"], self.w["sam_mask_decoder.iou_token.weight"],
                                     self.w["sam_mask_decoder.mask_tokens.weight"]], dim=0)

    def release(self):
        self.engine.close()

    # ---- glue, as the reference does it in torch -------------------------------------------------------------------
    def _sam_heads(self, pix_feat, hr0, hr1, points, labels, multimask):
        """_forward_sam_heads (:338-494) around the prompt-encoder and mask-decoder plugs."""
        B = pix_feat.shape[0]
        if points is None:
            points = torch.zeros(B, 1, 2, device=self.device)
            labels = -torch.ones(B, 1, dtype=torch.int32, device=self.device)
        sparse, dense = self.engine.prompt_encoder_full((points, labels), None, None)
        tokens = torch.cat([self.out_tokens.unsqueeze(0).expand(B, -1, -1), sparse], dim=1).contiguous()       # mask_decoder.py:186-202
        src = (pix_feat + dense).contiguous()
        masks, iou, mask_toks, obj = self.engine.mask_decoder(src, tokens, self.dense_pe.expand(B, -1, -1, -1).contiguous(), hr0, hr1)
        if multimask:
            low_multi, ious, toks = masks[:, 1:], iou[:, 1:], mask_toks[:, 1:]
        else:
            # MaskDecoder._dynamic_multimask_via_stability (mask_decoder.py:346-382): mask 0 unless it is unstable, then the best of 1-3
            d, thr = self.cfg["dynamic_multimask_stability_delta"], self.cfg["dynamic_multimask_stability_thresh"]
            m0 = masks[:, 0:1]
            inter, union = (m0 > d).sum((-1, -2)).float(), (m0 > -d).sum((-1, -2)).float()
            stable = torch.where(union > 0, inter / union, torch.ones_like(union)) >= thr                 # (B, 1)
            bi = torch.arange(B, device=self.device)
            best = torch.argmax(iou[:, 1:], dim=-1)
            low_multi = torch.where(stable[..., None, None], m0, masks[:, 1:][bi, best].unsqueeze(1))
            ious = torch.where(stable, iou[:, 0:1], iou[:, 1:][bi, best].unsqueeze(1))
            toks = mask_toks[:, 0:1]
        appearing = obj > 0
        low_multi = torch.where(appearing[:, None, None], low_multi, torch.full_like(low_multi, NO_OBJ_SCORE))
        S = self.cfg["image_size"]
        high_multi = F.interpolate(low_multi, size=(S, S), mode="bilinear", align_corners=False)
        tok = toks[:, 0]
        if multimask:
            best = torch.argmax(ious, dim=-1)
            bi = torch.arange(B, device=self.device)
            low, high, tok = low_multi[bi, best].unsqueeze(1), high_multi[bi, best].unsqueeze(1), toks[bi, best]
        else:
            low, high = low_multi, high_multi
        ptr = tok
        for i in range(3):                                                                   # obj_ptr_proj: MLP(256, 256, 256, 3)
            ptr = F.linear(ptr, self.w[f"obj_ptr_proj.layers.{i}.weight"], self.w[f"obj_ptr_proj.layers.{i}.bias"])
            if i < 2:
                ptr = F.relu(ptr)
        lam = appearing.float()
        ptr = lam * ptr + (1 - lam) * self.w["no_obj_ptr"]
        return dict(pred_masks=low, high_res_masks=high, obj_ptr=ptr, object_score_logits=obj)

    def _encode_memory(self, feat2, out, is_mask_from_pts):
        """_encode_new_memory (:978-1026) around the memory-encoder plug."""
        high = out["high_res_masks"]
        m = (high > 0).float() if is_mask_from_pts else torch.sigmoid(high)
        m = m * self.cfg["sigmoid_scale_for_mem_enc"] + self.cfg["sigmoid_bias_for_mem_enc"]
        feats, pos = self.engine.memory_encoder(feat2, m.contiguous())
        appearing = (out["object_score_logits"] > 0).float()
        feats = feats + (1 - appearing[..., None, None]) * self.w["no_obj_embed_spatial"][..., None, None].expand(*feats.shape)
        out["maskmem_features"] = feats.to(torch.bfloat16)                                   # sam2_video_predictor_official.py:887
        out["maskmem_pos_enc"] = pos

    def _memory_conditioned(self, t, feats, num_frames):
        """_prepare_memory_conditioned_features (:797-976) around the memory-attention plug."""
        C, M, nm = self.cfg["d_model"], self.cfg["mem_dim"], self.cfg["num_maskmem"]
        mems, ptrs, max_ptrs = select_memory(self.cond, self.non_cond, t, num_frames, False, nm, self.cfg["max_obj_ptrs_in_encoder"])
        mem = torch.stack([o["maskmem_features"].float().flatten(2).permute(2, 0, 1) for _, o in mems], dim=0)
        mpos = torch.stack([o["maskmem_pos_enc"].flatten(2).permute(2, 0, 1) + self.w["maskmem_tpos_enc"][nm - tp - 1] for tp, o in mems], dim=0)
        pos_list = torch.tensor([float(d) for d, _ in ptrs], device=self.device)
        obj_ptrs = torch.stack([o["obj_ptr"] for _, o in ptrs], dim=0)                      # (n, 1, 256)
        pe_dim = C // 2
        dim_t = 10000.0 ** (2 * (torch.arange(pe_dim, device=self.device, dtype=torch.float32) // 2) / pe_dim)
        e = (pos_list / (max_ptrs - 1)).unsqueeze(-1) / dim_t                               # get_1d_sine_pe (sam2_utils.py:64-74)
        obj_pos = F.linear(torch.cat([e.sin(), e.cos()], dim=-1), self.w["obj_ptr_tpos_proj.weight"], self.w["obj_ptr_tpos_proj.bias"])
        obj_pos = obj_pos.unsqueeze(1).expand(-1, 1, M)
        obj_ptrs = obj_ptrs.reshape(-1, 1, C // M, M).permute(0, 2, 1, 3).flatten(0, 1)
        obj_pos = obj_pos.repeat_interleave(C // M, dim=0)
        curr = feats[6].flatten(2).permute(2, 0, 1).contiguous()
        curr_pos = feats[3].flatten(2).permute(2, 0, 1).contiguous()
        pix = self.engine.memory_attention(curr, mem.contiguous(), curr_pos, mpos.contiguous(), obj_ptrs.contiguous(), obj_pos.contiguous())
        return pix.permute(1, 2, 0).reshape(1, C, 64, 64)

    # ---- the benchmark scenario ------------------------------------------------------------------------------------
    @torch.inference_mode()
    def start(self, frames: torch.Tensor, click_xy):
        """frames: (T,3,1024,1024) normalised f32 on the device; one positive click on frame 0."""
        # the reference keeps the clip as one contiguous (T,3,S,S) tensor (utils/misc.py:213-277: torch.zeros(num_frames, 3, S, S))
        self.frames, self.cond, self.non_cond = frames.contiguous(), OrderedDict(), OrderedDict()
        if hasattr(self.image_plug, "clear"):
            self.image_plug.clear()
        f = self.image_plug(self.frames[0].unsqueeze(0))       # the view the reference passes: inference_state["images"][t].unsqueeze(0)
        pix = f[6] + self.w["no_mem_embed"].view(1, -1, 1, 1)
        pts = torch.tensor([[list(click_xy)]], dtype=torch.float32, device=self.device)
        out = self._sam_heads(pix, f[4], f[5], pts, torch.ones(1, 1, dtype=torch.int32, device=self.device), True)
        self._encode_memory(f[6], out, True)
        self.cond[0] = out
        return out["pred_masks"]

    @torch.inference_mode()
    def propagate(self):
        """Yields (frame index, (1,1,256,256) low-res logits) for every frame, like propagate_in_video (frame 0: the stored
        conditioning output)."""
        T = self.frames.shape[0]
        for t in range(T):
            if t in self.cond:
                yield t, self.cond[t]["pred_masks"]
                continue
            f = self.image_plug(self.frames[t].unsqueeze(0))
            pix = self._memory_conditioned(t, f, T)
            out = self._sam_heads(pix, f[4], f[5], None, None, True)
            self._encode_memory(f[6], out, False)
            self.non_cond[t] = out
            self.non_cond.pop(t - self.cfg["max_obj_ptrs_in_encoder"] - 2, None)             # nothing later attends to it (keeps memory flat)
            yield t, out["pred_masks"]
