"""Generate tests/golden/*.npz by running the REAL reference (backend="torch", fp32, CPU)
from /root/reference on seeded synthetic inputs and weights.  Container-only; the
outputs (data, not code) are committed and travel to the GPU box.

    python -m oracle.gen_golden [plugs] [video] [tiny] [interact] [reverse] [multi] [box] [long] [opts] [ingest] [sizes] [outliers] [video2]

Weights: sam2_opt_amd.weights.synthetic_state_dict(cfg, seed=0)   (regenerated anywhere)
Inputs : sam2_opt_amd.synthetic.*  with the seeds named below.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle.golden_io import pack  # noqa: E402
from oracle.ref_import import build_reference_model  # noqa: E402
from sam2_opt_amd.config import get_config  # noqa: E402
from sam2_opt_amd.synthetic import normalize_frames, randn, synthetic_frames_u8, synthetic_image_normed  # noqa: E402
from sam2_opt_amd.weights import synthetic_state_dict  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
BLOCKS = (0, 1, 2, 3, 7, 8, 9, 23, 43, 44, 47)
VIDEO_FRAMES = 24
FULL_FRAMES = (7, 15, 23)       # video-res masks stored in full (f16) in large_video24_full.npz
CLICK = (512.0, 512.0)


def plug_inputs(cfg):
    """Seeded plug-level inputs shared with the tests (tests/plug_inputs mirrors by import)."""
    C, M = cfg["d_model"], cfg["mem_dim"]
    d = {}
    for tag, L, P, seed in (("memattn_L1P4", 1, 4, 11), ("memattn_L3P12", 3, 12, 12), ("memattn_L1P0", 1, 0, 13)):
        d[tag] = (randn(seed, 4096, 1, C), randn(seed + 100, L, 4096, 1, M), randn(seed + 200, 4096, 1, C),
                  randn(seed + 300, L, 4096, 1, M), randn(seed + 400, P, 1, M), randn(seed + 500, P, 1, M))
    for tag, N, T, seed in (("maskdec_N1T8", 1, 8, 21), ("maskdec_N2T15", 2, 15, 22)):
        d[tag] = (randn(seed, N, C, 64, 64), randn(seed + 100, N, T, C), randn(seed + 200, N, C, 64, 64),
                  randn(seed + 300, N, C // 8, 256, 256), randn(seed + 400, N, C // 4, 128, 128))
    d["memenc"] = (randn(31, 1, C, 64, 64), randn(131, 1, 1, 1024, 1024, scale=4.0))
    d["prompt"] = (torch.tensor([[[100.0, 200.0], [512.0, 512.0], [1000.5, 3.25]]]), torch.tensor([[1, 0, 1]], dtype=torch.int32))
    d["samheads"] = (randn(41, 1, C, 64, 64), randn(141, 1, C // 8, 256, 256), randn(241, 1, C // 4, 128, 128))
    return d


@torch.inference_mode()
def gen_plugs():
    cfg = get_config("large")
    sd = synthetic_state_dict(cfg, seed=0)
    model = build_reference_model(cfg, "video", sd)
    store = {}
    t0 = time.time()
    # ---- image encoder (config 2) + selected block outputs
    img = synthetic_image_normed(seed=1)
    blk = {}
    hooks = [model.image_encoder.trunk.blocks[i].register_forward_hook(
        lambda m, a, o, i=i: blk.__setitem__(i, o.detach().clone())) for i in BLOCKS]
    outs = model.inference_image_torch(img)
    for h in hooks:
        h.remove()
    names = ["vision_features", "vision_pos_enc0", "vision_pos_enc1", "vision_pos_enc2",
             "backbone_fpn0", "backbone_fpn1", "backbone_fpn2"]
    for n, o in zip(names, outs):
        pack(store, "enc/" + n, o, 65536)
    for i, o in blk.items():
        pack(store, f"enc/block{i}", o, 32768)
    print("encoder", time.time() - t0)
    pin = plug_inputs(cfg)
    for tag in ("memattn_L1P4", "memattn_L3P12", "memattn_L1P0"):
        o = model.memory_attention.inference_memory_attention_torch(*pin[tag])
        pack(store, tag, o, 65536)
    for tag in ("maskdec_N1T8", "maskdec_N2T15"):
        o = model.sam_mask_decoder.inference_predict_masks_torch(*pin[tag])
        for n, t in zip(("masks", "iou", "tokens", "obj"), o):
            pack(store, f"{tag}/{n}", t, 32768)
    x, pos = model.memory_encoder.inference_memory_torch(*pin["memenc"])
    pack(store, "memenc/x", x, 65536)
    pack(store, "memenc/pos", pos, 16384)
    sp, de = model.sam_prompt_encoder.inference_prompt_torch(pin["prompt"], None, None)
    pack(store, "prompt/sparse", sp)
    pack(store, "prompt/dense", de, 4096)
    pack(store, "prompt/dense_pe", model.sam_prompt_encoder.get_dense_pe(), 65536)
    for mm in (True, False):
        o = model._forward_sam_heads(pin["samheads"][0], None, None, [pin["samheads"][1], pin["samheads"][2]], mm)
        for n, t in zip(("low_multi", "high_multi", "ious", "low", "high", "obj_ptr", "obj_score"), o):
            pack(store, f"samheads_mm{int(mm)}/{n}", t, 16384)
    np.savez_compressed(os.path.join(GOLD, "large_plugs.npz"), **store)
    print("plugs done", time.time() - t0)


OUTLIER_GAIN = 64.0          # the f16-RANGE scenario: operand channels ~80x the median (223 vs 2.7 at block 20)
OUTLIER_GAIN_MILD = 8.0      # the well-conditioned outlier scenario (~10x): end-to-end tolerances are meaningful here


@torch.inference_mode()
def gen_outliers():
    """Outlier-channel scenarios (tests/golden/large_outliers.npz): synthetic weights with the LayerNorm gain of three channels x g
    in every norm of the trunk and of the memory attention, recurrent damping off - the image encoder on the seeded image and one
    memory-attention call (L = 3, P = 12) of the REAL reference, for g = 64 (prefix g64/: attention logits in the thousands, the
    encoder as a whole is ill-conditioned - even 22-bit operands end at 2e-1 - so this one pins RANGE: finite outputs, per-block
    parity) and g = 8 (prefix g8/: well-conditioned, end-to-end tolerances)."""
    cfg = get_config("large")
    store = {}
    for gain, tag in ((OUTLIER_GAIN, "g64/"), (OUTLIER_GAIN_MILD, "g8/")):
        sd = synthetic_state_dict(cfg, seed=0, undamped=True, outlier_gain=gain)
        model = build_reference_model(cfg, "video", sd)
        img = synthetic_image_normed(seed=1)
        stats = {}
        blk = model.image_encoder.trunk.blocks[20]
        h = blk.norm1.register_forward_hook(lambda m, a, o: stats.__setitem__("ln", o.detach().abs().amax(dim=(0, 1, 2))))
        outs = model.inference_image_torch(img)
        h.remove()
        for n, o in zip(["vision_features", "vision_pos_enc0", "vision_pos_enc1", "vision_pos_enc2", "backbone_fpn0", "backbone_fpn1", "backbone_fpn2"], outs):
            if not n.startswith("vision_pos"):
                pack(store, tag + "enc/" + n, o, 65536)
        ln = stats["ln"]
        store[tag + "enc/block20_norm1_absmax_outlier_vs_median"] = np.array([float(ln[list((3, 41, 77))].max()), float(ln.median())], np.float64)
        pin = plug_inputs(cfg)
        o = model.memory_attention.inference_memory_attention_torch(*pin["memattn_L3P12"])
        pack(store, tag + "memattn_L3P12", o, 65536)
    np.savez_compressed(os.path.join(GOLD, "large_outliers.npz"), **store)
    print("outliers done", {k: v for k, v in store.items() if "absmax" in k})


@torch.inference_mode()
def gen_video():
    cfg = get_config("large")
    sd = synthetic_state_dict(cfg, seed=0)
    model = build_reference_model(cfg, "video", sd, fill_hole_area=0)
    frames = normalize_frames(synthetic_frames_u8(seed=2, num_frames=VIDEO_FRAMES), cfg)
    import sam2.sam2_video_predictor_official as vp
    vp.load_video_frames = lambda **kw: (frames, 1024, 1024)
    store = {}
    full = {}                   # second file: EVERY pixel - low-res logits of all frames (f32), three video-res frames (f16)
    rec = {"t": None}
    ma = model.memory_attention
    orig_ex = ma.inference_memory_attention_exclude

    def rec_memattn(*inputs):
        out = orig_ex(*inputs)
        t = rec["t"]
        store[f"f{t}/LP"] = np.array([inputs[1].shape[0], inputs[4].shape[0]], dtype=np.int64)
        for n, x in zip(("curr", "memory", "curr_pos", "memory_pos", "mem_ex", "mem_pos_ex"), inputs):
            pack(store, f"f{t}/memattn_in/{n}", x, 2048)
        pack(store, f"f{t}/memattn_out", out, 8192)
        return out

    ma.inference_memory_attention_exclude = rec_memattn
    orig_heads = model._forward_sam_heads

    def rec_heads(*a, **k):
        o = orig_heads(*a, **k)
        t = rec["t"]
        for n, x in zip(("low_multi", "high_multi", "ious", "low", "high", "obj_ptr", "obj_score"), o):
            if n != "high_multi":
                pack(store, f"f{t}/heads/{n}", x, 4096)
        return o

    model._forward_sam_heads = rec_heads
    t0 = time.time()
    state = model.init_state(video_path="synthetic")
    rec["t"] = "click"
    _, _, vm = model.add_new_points_or_box(state, frame_idx=0, obj_id=1, points=np.array([CLICK], np.float32),
                                           labels=np.array([1], np.int32))
    pack(store, "click/video_res_mask", vm, 16384)
    rec["t"] = "pre"
    gen = model.propagate_in_video(state)
    n = 0
    while True:
        rec["t"] = n          # frames are yielded in order 0..T-1
        try:
            fi, ids, vm = next(gen)
        except StopIteration:
            break
        assert fi == n
        pack(store, f"f{fi}/video_res_mask", vm, 8192)
        out = state["output_dict_per_obj"][0]
        cur = out["cond_frame_outputs"].get(fi) or out["non_cond_frame_outputs"][fi]
        pack(store, f"f{fi}/maskmem_features", cur["maskmem_features"].float(), 8192)
        full[f"f{fi}/pred_masks"] = cur["pred_masks"].float().cpu().numpy()
        if fi in FULL_FRAMES:
            full[f"f{fi}/video_res_mask_f16"] = vm.float().cpu().numpy().astype(np.float16)
        print("frame", fi, time.time() - t0, flush=True)
        n += 1
    store["num_frames"] = np.array([n], dtype=np.int64)
    np.savez_compressed(os.path.join(GOLD, "large_video24.npz"), **store)
    full["num_frames"] = np.array([n], dtype=np.int64)
    np.savez_compressed(os.path.join(GOLD, "large_video24_full.npz"), **full)
    print("video done", time.time() - t0)


@torch.inference_mode()
def gen_video2():
    """A second propagation golden, independent of the one the f16s plan was tuned on: clip seed 7, click at (300, 640), WEIGHT seed 1,
    16 frames - every pixel of the low-res logits of every frame (tests/golden/large_video16_w1_full.npz)."""
    cfg = get_config("large")
    sd = synthetic_state_dict(cfg, seed=1)
    model = build_reference_model(cfg, "video", sd, fill_hole_area=0)
    frames = normalize_frames(synthetic_frames_u8(seed=7, num_frames=16), cfg)
    import sam2.sam2_video_predictor_official as vp
    vp.load_video_frames = lambda **kw: (frames, 1024, 1024)
    full = {}
    t0 = time.time()
    state = model.init_state(video_path="synthetic")
    model.add_new_points_or_box(state, frame_idx=0, obj_id=1, points=np.array([(300.0, 640.0)], np.float32), labels=np.array([1], np.int32))
    n = 0
    for fi, ids, vm in model.propagate_in_video(state):
        out = state["output_dict_per_obj"][0]
        cur = out["cond_frame_outputs"].get(fi) or out["non_cond_frame_outputs"][fi]
        full[f"f{fi}/pred_masks"] = cur["pred_masks"].float().cpu().numpy()
        print("frame", fi, time.time() - t0, float((cur["pred_masks"] > 0).float().mean()), flush=True)
        n += 1
    full["num_frames"] = np.array([n], dtype=np.int64)
    np.savez_compressed(os.path.join(GOLD, "large_video16_w1_full.npz"), **full)
    print("video2 done", time.time() - t0)


INTERACT_FRAMES = 6


def interact_mask():
    """The synthetic mask prompt of the interactive scenario: a disc + a bar, at video resolution (1024^2)."""
    yy, xx = np.mgrid[0:1024, 0:1024]
    return ((yy - 470) ** 2 + (xx - 540) ** 2 < 180 ** 2) | ((abs(yy - 800) < 40) & (abs(xx - 300) < 160))


@torch.inference_mode()
def gen_interact():
    """Mask prompt + correction click (SAM2VideoPredictor.add_new_mask, add_new_points_or_box on a tracked frame):
    mask on frame 0 -> propagate 6 frames -> negative click on tracked frame 3 -> propagate frames 4.. again."""
    cfg = get_config("large")
    sd = synthetic_state_dict(cfg, seed=0)
    model = build_reference_model(cfg, "video", sd, fill_hole_area=0)
    frames = normalize_frames(synthetic_frames_u8(seed=4, num_frames=INTERACT_FRAMES), cfg)
    import sam2.sam2_video_predictor_official as vp
    vp.load_video_frames = lambda **kw: (frames, 1024, 1024)
    store = {}
    t0 = time.time()
    state = model.init_state(video_path="synthetic")
    _, _, vm = model.add_new_mask(state, frame_idx=0, obj_id=1, mask=interact_mask())
    pack(store, "mask0/video_res_mask", vm, 16384)
    cur = state["temp_output_dict_per_obj"][0]["cond_frame_outputs"][0]
    pack(store, "mask0/pred_masks", cur["pred_masks"], 16384)
    pack(store, "mask0/obj_ptr", cur["obj_ptr"], 256)
    pack(store, "mask0/object_score_logits", cur["object_score_logits"], 1)
    for fi, ids, vm in model.propagate_in_video(state):
        pack(store, f"p1/f{fi}/video_res_mask", vm, 8192)
        print("pass 1 frame", fi, time.time() - t0, flush=True)
    _, _, vm = model.add_new_points_or_box(state, frame_idx=3, obj_id=1, points=np.array([[600.0, 400.0]], np.float32),
                                           labels=np.array([0], np.int32))
    pack(store, "fix3/video_res_mask", vm, 16384)
    cur = state["temp_output_dict_per_obj"][0]["non_cond_frame_outputs"][3]
    pack(store, "fix3/obj_ptr", cur["obj_ptr"], 256)
    pack(store, "fix3/object_score_logits", cur["object_score_logits"], 1)
    # a corrected non-conditioning frame is re-tracked (and its correction lost) when propagation passes over it
    # (add_all_frames_to_correct_as_cond=False): continue from the frame after it, which sees the corrected memory
    for fi, ids, vm in model.propagate_in_video(state, start_frame_idx=4):
        pack(store, f"p2/f{fi}/video_res_mask", vm, 8192)
        print("pass 2 frame", fi, time.time() - t0, flush=True)
    np.savez_compressed(os.path.join(GOLD, "large_interact6.npz"), **store)
    print("interact done", time.time() - t0)


@torch.inference_mode()
def gen_reverse():
    """Reverse tracking: one click on the LAST frame of a 6-frame clip, propagate_in_video(reverse=True)."""
    cfg = get_config("large")
    sd = synthetic_state_dict(cfg, seed=0)
    model = build_reference_model(cfg, "video", sd, fill_hole_area=0)
    frames = normalize_frames(synthetic_frames_u8(seed=6, num_frames=INTERACT_FRAMES), cfg)
    import sam2.sam2_video_predictor_official as vp
    vp.load_video_frames = lambda **kw: (frames, 1024, 1024)
    store = {}
    state = model.init_state(video_path="synthetic")
    _, _, vm = model.add_new_points_or_box(state, frame_idx=INTERACT_FRAMES - 1, obj_id=1, points=np.array([CLICK], np.float32),
                                           labels=np.array([1], np.int32))
    pack(store, "click/video_res_mask", vm, 16384)
    order = []
    for fi, ids, vm in model.propagate_in_video(state, reverse=True):
        pack(store, f"f{fi}/video_res_mask", vm, 8192)
        order.append(fi)
    store["order"] = np.array(order, dtype=np.int64)
    np.savez_compressed(os.path.join(GOLD, "large_reverse6.npz"), **store)
    print("reverse done", order)


MULTI_FRAMES = 8
MULTI_CLICKS = {1: (0, (512.0, 512.0)), 2: (0, (300.0, 700.0)), 3: (2, (760.0, 260.0))}      # obj_id -> (frame, click)


def _video_model(seed, num_frames, **kw):
    cfg = get_config("large")
    sd = synthetic_state_dict(cfg, seed=0)
    model = build_reference_model(cfg, "video", sd, fill_hole_area=0, **kw)
    frames = normalize_frames(synthetic_frames_u8(seed=seed, num_frames=num_frames), cfg)
    import sam2.sam2_video_predictor_official as vp
    vp.load_video_frames = lambda **k: (frames, 1024, 1024)
    return model


@torch.inference_mode()
def gen_multi():
    """Multi-object tracking through the reference's per-object loop (sam2_video_predictor_official.py:691-725) and
    _consolidate_temp_output_across_obj / non-overlap logic: scenario A = objects 1 and 2 clicked on frame 0; scenario B adds
    object 3 with its click on frame 2 (staggered prompts) and propagates forward from frame 2, then in reverse from frame 2
    back to 0 (on frames 1 and 0 object 3 is tracked from its conditioning frame 2; objects 1-2 pass their conditioning
    frame 0); scenario C = scenario A with non_overlap_masks=True.  Per frame: the (num_obj,1,H,W) video-res masks (sampled)
    and every object's low-res logits of frames 3 and 7 in full.
    Not pinned here: propagating scenario B forward from frame 0.  Object 3 then has only a FUTURE conditioning frame, so
    its memory holds no object pointers (P = 0) and the reference's fp32 torch path fails in RoPEAttention.v_proj with
    "mat1 and mat2 must have the same dtype" (the bf16 memory is only promoted to f32 by the concatenation with the f32
    pointers, sam2_base_official.py:964-965) - it needs the autocast the reference runs under on a GPU."""
    store = {}
    t0 = time.time()
    for tag, objs, kw in (("A", (1, 2), {}), ("B", (1, 2, 3), {}), ("C", (1, 2), {"non_overlap_masks": True})):
        model = _video_model(8, MULTI_FRAMES, **kw)
        state = model.init_state(video_path="synthetic")
        for oid in objs:
            fr, pt = MULTI_CLICKS[oid]
            _, ids, vm = model.add_new_points_or_box(state, frame_idx=fr, obj_id=oid, points=np.array([pt], np.float32),
                                                     labels=np.array([1], np.int32))
            pack(store, f"{tag}/click{oid}/video_res_mask", vm, 16384)
            store[f"{tag}/click{oid}/obj_ids"] = np.array(ids, dtype=np.int64)
        start = 2 if tag == "B" else None
        for fi, ids, vm in model.propagate_in_video(state, start_frame_idx=start):
            pack(store, f"{tag}/f{fi}/video_res_mask", vm, 16384)
            store[f"{tag}/f{fi}/obj_ids"] = np.array(ids, dtype=np.int64)
            if fi in (3, 7):
                for k in range(len(objs)):
                    od = state["output_dict_per_obj"][k]
                    cur = od["cond_frame_outputs"].get(fi) or od["non_cond_frame_outputs"][fi]
                    store[f"{tag}/f{fi}/obj{k}/pred_masks"] = cur["pred_masks"].float().cpu().numpy()
            print("multi", tag, "frame", fi, time.time() - t0, flush=True)
        if tag == "B":
            for fi, ids, vm in model.propagate_in_video(state, start_frame_idx=2, reverse=True):
                pack(store, f"{tag}/rev/f{fi}/video_res_mask", vm, 16384)
                print("multi", tag, "reverse frame", fi, time.time() - t0, flush=True)
    np.savez_compressed(os.path.join(GOLD, "large_multi8.npz"), **store)
    print("multi done", time.time() - t0)


BOX = (300.0, 380.0, 720.0, 800.0)       # x0, y0, x1, y1


@torch.inference_mode()
def gen_box():
    """Box prompts (labels 2 / 3, sam2_video_predictor_official.py:300-316; prompt_encoder.py:_embed_points with the corner
    embeddings): a box on frame 0; a box + one positive point in the same call on frame 0 of a second run; 4 frames each.
    Also the prompt-encoder plug on (points, boxes, None) directly (PromptEncoder.inference_prompt_torch, :215-231)."""
    store = {}
    t0 = time.time()
    for tag, pts in (("box", None), ("boxpt", np.array([[500.0, 600.0]], np.float32))):
        model = _video_model(12, 4)
        state = model.init_state(video_path="synthetic")
        _, ids, vm = model.add_new_points_or_box(state, frame_idx=0, obj_id=1, box=np.array(BOX, np.float32), points=pts,
                                                 labels=None if pts is None else np.array([1], np.int32))
        pack(store, f"{tag}/click/video_res_mask", vm, 16384)
        cur = state["temp_output_dict_per_obj"][0]["cond_frame_outputs"][0]
        store[f"{tag}/click/pred_masks"] = cur["pred_masks"].float().cpu().numpy()
        pack(store, f"{tag}/click/obj_ptr", cur["obj_ptr"], 256)
        for fi, ids, vm in model.propagate_in_video(state):
            pack(store, f"{tag}/f{fi}/video_res_mask", vm, 16384)
            print("box", tag, "frame", fi, time.time() - t0, flush=True)
    pe = model.sam_prompt_encoder
    pts = (torch.tensor([[[100.0, 200.0], [512.0, 512.0]], [[7.5, 900.0], [640.0, 32.0]]]), torch.tensor([[1, 0], [1, 1]], dtype=torch.int32))
    boxes = torch.tensor([[10.0, 20.0, 300.0, 400.0], [512.0, 512.0, 1000.0, 900.0]])
    sp, de = pe.inference_prompt_torch(pts, boxes, None)
    pack(store, "plug/points_boxes/sparse", sp)
    sp, de = pe.inference_prompt_torch(None, boxes, None)
    pack(store, "plug/boxes/sparse", sp)
    mask_in = randn(77, 2, 1, 256, 256, scale=3.0)
    sp, de = pe.inference_prompt_torch(pts, None, mask_in)
    pack(store, "plug/points_mask/sparse", sp)
    pack(store, "plug/points_mask/dense", de, 65536)
    np.savez_compressed(os.path.join(GOLD, "large_box4.npz"), **store)
    print("box done", time.time() - t0)


LONG_FRAMES = 32
LONG_CLICK_FRAME = 16


@torch.inference_mode()
def gen_long():
    """Clip longer than the pointer horizon (16) + the memory horizon (7): click in the MIDDLE (frame 16), propagate forward to
    the end, then propagate_in_video(reverse=True) from the click back to frame 0 (frame 15 attends to the forward pass's
    memories of frames 17..22, sam2_base_official.py:843-864), then a correction click on frame 5 - more than 17 frames
    behind the forward head - and its re-tracking of frames 6..9.  ADVICE r01: slots of old frames must survive."""
    store = {}
    t0 = time.time()
    model = _video_model(14, LONG_FRAMES)
    state = model.init_state(video_path="synthetic")
    _, _, vm = model.add_new_points_or_box(state, frame_idx=LONG_CLICK_FRAME, obj_id=1, points=np.array([CLICK], np.float32),
                                           labels=np.array([1], np.int32))
    pack(store, "click/video_res_mask", vm, 16384)
    for fi, ids, vm in model.propagate_in_video(state):
        pack(store, f"fwd/f{fi}/video_res_mask", vm, 8192)
        print("long fwd", fi, time.time() - t0, flush=True)
    for fi, ids, vm in model.propagate_in_video(state, reverse=True):
        pack(store, f"rev/f{fi}/video_res_mask", vm, 8192)
        print("long rev", fi, time.time() - t0, flush=True)
    _, _, vm = model.add_new_points_or_box(state, frame_idx=5, obj_id=1, points=np.array([[420.0, 640.0]], np.float32),
                                           labels=np.array([0], np.int32))
    pack(store, "fix5/video_res_mask", vm, 16384)
    for fi, ids, vm in model.propagate_in_video(state, start_frame_idx=6, max_frame_num_to_track=3):
        pack(store, f"fix/f{fi}/video_res_mask", vm, 8192)
        print("long fix", fi, time.time() - t0, flush=True)
    np.savez_compressed(os.path.join(GOLD, "large_long32.npz"), **store)
    print("long done", time.time() - t0)


OPTS_FRAMES = 12
OPTS = dict(max_cond_frames_in_attn=2, memory_temporal_stride_for_eval=2, add_all_frames_to_correct_as_cond=True)
OPTS_CLICKS = ((0, (512.0, 512.0)), (5, (530.0, 500.0)), (10, (500.0, 540.0)))


@torch.inference_mode()
def gen_opts():
    """Non-default predictor / model options (sam2_video_predictor_official.py:24-40, sam2_base_official.py:39-41,:63):
    max_cond_frames_in_attn=2 with three, later four, conditioning frames (select_closest_cond_frames; an unselected one is
    attended to as a non-conditioning frame), memory_temporal_stride_for_eval=2, add_all_frames_to_correct_as_cond=True (the
    correction click on tracked frame 7 turns it into a conditioning frame).  `clear_non_cond_mem_around_input` cannot be
    recorded: the reference calls self._clear_obj_non_cond_mem_around_input (:632,:704), which it does not define."""
    store = {}
    t0 = time.time()
    model = _video_model(16, OPTS_FRAMES, **OPTS)
    state = model.init_state(video_path="synthetic")
    for fr, pt in OPTS_CLICKS:
        _, _, vm = model.add_new_points_or_box(state, frame_idx=fr, obj_id=1, points=np.array([pt], np.float32), labels=np.array([1], np.int32))
        pack(store, f"click{fr}/video_res_mask", vm, 16384)
    for fi, ids, vm in model.propagate_in_video(state):
        pack(store, f"p1/f{fi}/video_res_mask", vm, 8192)
        print("opts pass 1", fi, time.time() - t0, flush=True)
    _, _, vm = model.add_new_points_or_box(state, frame_idx=7, obj_id=1, points=np.array([[600.0, 420.0]], np.float32), labels=np.array([0], np.int32))
    pack(store, "fix7/video_res_mask", vm, 16384)
    for fi, ids, vm in model.propagate_in_video(state, start_frame_idx=6):
        pack(store, f"p2/f{fi}/video_res_mask", vm, 8192)
        print("opts pass 2", fi, time.time() - t0, flush=True)
    od = state["output_dict_per_obj"][0]
    store["cond_frames"] = np.array(sorted(od["cond_frame_outputs"]), dtype=np.int64)
    np.savez_compressed(os.path.join(GOLD, "large_opts12.npz"), **store)
    print("opts done", time.time() - t0, sorted(od["cond_frame_outputs"]))


@torch.inference_mode()
def gen_sizes():
    """The padded-window model sizes (hiera small / base+; tiny has its own end-to-end golden): SAM2Base.inference_image_torch of the
    REAL reference on the seeded image, synthetic weights of that size."""
    store = {}
    img = synthetic_image_normed(seed=1)
    for name in ("small", "base_plus"):
        cfg = get_config(name)
        model = build_reference_model(cfg, "base", synthetic_state_dict(cfg, seed=0))
        outs = model.inference_image_torch(img)
        for n, o in zip(("vision_features", "vision_pos_enc0", "vision_pos_enc1", "vision_pos_enc2", "backbone_fpn0", "backbone_fpn1", "backbone_fpn2"), outs):
            if "pos_enc" not in n:
                pack(store, f"{name}/{n}", o, 65536)
        print("sizes", name, "done", flush=True)
    np.savez_compressed(os.path.join(GOLD, "sizes_encoder.npz"), **store)


def gen_ingest():
    """Frame ingest: three synthetic 180 x 320 JPEGs written to a temporary folder and loaded by the REFERENCE's
    load_video_frames_from_jpg_images (utils/misc.py:213-277: PIL decode + Image.resize((1024, 1024)) + /255 + mean / std).
    The fixture keeps the JPEG bytes themselves (decoders may differ between boxes; the bytes do not) and the sampled result."""
    import io
    import tempfile
    from PIL import Image
    from oracle.ref_import import import_reference
    import_reference()
    from sam2.utils.misc import load_video_frames_from_jpg_images
    store = {}
    with tempfile.TemporaryDirectory() as d:
        for i in range(3):
            rs = np.random.RandomState(40 + i)
            yy, xx = np.mgrid[0:180, 0:320]
            img = np.stack([(np.sin(yy / 9.0 + i) * 0.5 + 0.5) * 255, xx * 255.0 / 319, ((yy + 2 * xx + 7 * i) % 64) * 4.0], -1)
            img = np.clip(img + rs.randint(-25, 25, img.shape), 0, 255).astype(np.uint8)
            buf = io.BytesIO()
            Image.fromarray(img).save(buf, format="JPEG", quality=92)
            store[f"jpeg{i}"] = np.frombuffer(buf.getvalue(), dtype=np.uint8)
            with open(os.path.join(d, f"{i:05d}.jpg"), "wb") as f:
                f.write(buf.getvalue())
        images, H, W = load_video_frames_from_jpg_images(video_path=d, image_size=1024, offload_video_to_cpu=True,
                                                         compute_device=torch.device("cpu"))
    pack(store, "images", images, 200000)
    store["video_hw"] = np.array([H, W], dtype=np.int64)
    store["num_frames"] = np.array([3], dtype=np.int64)
    np.savez_compressed(os.path.join(GOLD, "ingest_jpeg.npz"), **store)
    print("ingest done", images.shape, H, W)


@torch.inference_mode()
def gen_tiny():
    """BASELINE.json configs[0]: SAM2.1-hiera-tiny image predictor, one 1024^2 frame, torch backend on the CPU - the
    reference's own CPU-runnable case.  Synthetic tiny weights, image RandomState(0), one positive click at (512, 512)."""
    cfg = get_config("tiny")
    sd = synthetic_state_dict(cfg, seed=0)
    model = build_reference_model(cfg, "base", sd)            # also puts /root/reference/sam2 on sys.path (with the shims)
    from sam2.sam2_image_predictor import SAM2ImagePredictor
    pred = SAM2ImagePredictor(model)
    img = np.random.RandomState(0).randint(0, 256, (1024, 1024, 3)).astype(np.uint8)
    t0 = time.time()
    pred.set_image(img)
    masks, ious, low = pred.predict(point_coords=np.array([CLICK], np.float32), point_labels=np.array([1], np.int32),
                                    multimask_output=True, return_logits=True)
    store = {}
    pack(store, "tiny/masks_logits", torch.from_numpy(np.asarray(masks)), 40000)
    pack(store, "tiny/ious", torch.from_numpy(np.asarray(ious)), 16)
    pack(store, "tiny/low_res", torch.from_numpy(np.asarray(low)), 40000)
    pack(store, "tiny/image_embed", pred._features["image_embed"], 20000)
    np.savez_compressed(os.path.join(GOLD, "tiny_image.npz"), **store)
    print("tiny done", time.time() - t0, "ious", np.asarray(ious))


if __name__ == "__main__":
    which = sys.argv[1:] or ["plugs", "video"]
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    if "plugs" in which:
        gen_plugs()
    if "video" in which:
        gen_video()
    if "tiny" in which:
        gen_tiny()
    if "interact" in which:
        gen_interact()
    if "reverse" in which:
        gen_reverse()
    if "multi" in which:
        gen_multi()
    if "box" in which:
        gen_box()
    if "long" in which:
        gen_long()
    if "opts" in which:
        gen_opts()
    if "ingest" in which:
        gen_ingest()
    if "sizes" in which:
        gen_sizes()
    if "outliers" in which:
        gen_outliers()
    if "video2" in which:
        gen_video2()
