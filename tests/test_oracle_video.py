"""Oracle video loop (memory-bank assembly, SAM-head selection, memory encoder, bf16 bank
rounding) vs golden vectors from the real reference's propagate_in_video on the first frames
of the synthetic config-3 clip.  CPU only; frames beyond ORACLE_VIDEO_FRAMES (default 9: L grows
1..7, P 4..32) are covered when the env var is raised to 24."""
import os

import numpy as np
import torch

from oracle import sam2_ref as R
from oracle.gen_golden import CLICK, VIDEO_FRAMES
from oracle.golden_io import compare
from sam2_opt_amd.synthetic import normalize_frames, synthetic_frames_u8

NF = int(os.environ.get("ORACLE_VIDEO_FRAMES", "9"))


def test_video_oracle_matches_reference(sd_large, cfg_large, golden_video):
    g = golden_video
    assert int(g["num_frames"][0]) == VIDEO_FRAMES
    frames = normalize_frames(synthetic_frames_u8(seed=2, num_frames=VIDEO_FRAMES), cfg_large)
    vo = R.VideoOracle(sd_large, cfg_large, frames)
    with torch.inference_mode():
        vm = vo.add_new_points(0, np.array([CLICK], np.float32), np.array([1], np.int32))
        ok, msg = compare(g, "click/video_res_mask", vm, atol=2e-3, rtol=1e-3)
        assert ok, msg
        for t, vm in vo.propagate(max_frames=NF - 1):
            ok, msg = compare(g, f"f{t}/video_res_mask", vm, atol=5e-3, rtol=1e-3)
            assert ok, msg
            if t == 0:
                ok, msg = compare(g, "f0/maskmem_features", vo.cond[0]["maskmem_features"].float(), atol=2e-3, rtol=8e-3, outlier_frac=2e-3)  # 1 bf16 ulp; a mask pixel
                # with logit ~0 may binarise differently (sam2_base_official.py:1000-1004) -> rare local outliers
                assert ok, msg
                continue
            tr = vo.trace[("track", t)]
            L, P = (int(v) for v in g[f"f{t}/LP"])
            assert tr["memattn_in"][1].shape[0] == L and tr["memattn_in"][4].shape[0] == P
            for n, x in zip(("curr", "memory", "curr_pos", "memory_pos", "mem_ex", "mem_pos_ex"), tr["memattn_in"]):
                ok, msg = compare(g, f"f{t}/memattn_in/{n}", x, atol=2e-3, rtol=8e-3 if n in ("memory", "mem_ex") else 1e-3,
                                      outlier_frac=2e-3 if n == "memory" else 0.0)
                assert ok, msg
            ok, msg = compare(g, f"f{t}/memattn_out", tr["pix_feat"].flatten(2).permute(2, 0, 1), atol=5e-3, rtol=1e-3)
            assert ok, msg
            for gk, k in (("ious", "ious"), ("obj_ptr", "obj_ptr"), ("obj_score", "object_score_logits"), ("low", "low_res_masks")):
                ok, msg = compare(g, f"f{t}/heads/{gk}", tr[k], atol=5e-3, rtol=1e-3)
                assert ok, msg


def test_oracle_mask_prompt_and_correction_click_match_reference():
    """add_new_mask (mask as output, pointer from the SAM heads), propagation from it, a negative correction click on a
    tracked frame (memory-conditioned features + previous logits as mask prompt) and propagation from the frame after it -
    the oracle against the REAL reference (tests/golden/large_interact6.npz, oracle/gen_golden.py interact)."""
    import os
    import numpy as np
    import torch
    from oracle import sam2_ref as R
    from oracle.gen_golden import INTERACT_FRAMES, interact_mask
    from oracle.golden_io import compare
    from sam2_opt_amd.config import get_config
    from sam2_opt_amd.synthetic import normalize_frames, synthetic_frames_u8
    from sam2_opt_amd.weights import synthetic_state_dict
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "large_interact6.npz"))
    cfg = get_config("large")
    sd = synthetic_state_dict(cfg, seed=0)
    frames = normalize_frames(synthetic_frames_u8(seed=4, num_frames=INTERACT_FRAMES), cfg)

    def chk(name, t, atol=2e-3, frac=2e-3):
        ok, msg = compare(gold, name, t, atol=atol, rtol=1e-3, outlier_frac=frac)
        assert ok, msg
    with torch.inference_mode():
        vo = R.VideoOracle(sd, cfg, frames)
        vm = vo.add_new_mask(0, interact_mask())
        chk("mask0/video_res_mask", vm)
        cur = vo.temp["cond"][0]
        chk("mask0/pred_masks", cur["pred_masks"])
        chk("mask0/obj_ptr", cur["obj_ptr"], atol=5e-4)
        chk("mask0/object_score_logits", cur["object_score_logits"], atol=1e-6)
        for t, vm in vo.propagate():
            chk(f"p1/f{t}/video_res_mask", vm)
        vm = vo.add_new_points(3, np.array([[600.0, 400.0]], np.float32), np.array([0], np.int32))
        chk("fix3/video_res_mask", vm)
        cur = vo.temp["non_cond"][3]
        chk("fix3/obj_ptr", cur["obj_ptr"], atol=5e-4)
        chk("fix3/object_score_logits", cur["object_score_logits"], atol=5e-4)
        for t, vm in vo.propagate(start_frame_idx=4):
            chk(f"p2/f{t}/video_res_mask", vm)


def test_oracle_reverse_tracking_matches_reference():
    """propagate_in_video(reverse=True) from a click on the last frame (signed pointer offsets, memories taken from later
    frames) - the oracle against the REAL reference (tests/golden/large_reverse6.npz)."""
    import os
    import numpy as np
    import torch
    from oracle import sam2_ref as R
    from oracle.gen_golden import CLICK, INTERACT_FRAMES
    from oracle.golden_io import compare
    from sam2_opt_amd.config import get_config
    from sam2_opt_amd.synthetic import normalize_frames, synthetic_frames_u8
    from sam2_opt_amd.weights import synthetic_state_dict
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "large_reverse6.npz"))
    cfg = get_config("large")
    sd = synthetic_state_dict(cfg, seed=0)
    frames = normalize_frames(synthetic_frames_u8(seed=6, num_frames=INTERACT_FRAMES), cfg)
    with torch.inference_mode():
        vo = R.VideoOracle(sd, cfg, frames)
        vm = vo.add_new_points(INTERACT_FRAMES - 1, np.array([CLICK], np.float32), np.array([1], np.int32))
        ok, msg = compare(gold, "click/video_res_mask", vm, atol=2e-3, rtol=1e-3, outlier_frac=2e-3)
        assert ok, msg
        order = []
        for t, vm in vo.propagate(reverse=True):
            ok, msg = compare(gold, f"f{t}/video_res_mask", vm, atol=2e-3, rtol=1e-3, outlier_frac=2e-3)
            assert ok, msg
            order.append(t)
    assert order == list(gold["order"])
