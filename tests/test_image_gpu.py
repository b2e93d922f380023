"""GPU: image predictor (BASELINE configs[4] shape: a batch of 1024^2 images, 8 independent single-point prompts each,
plus one multi-point prompt) of the HIP backend vs the CPU oracle.  Reduced to 2 images to keep the CPU oracle short."""
import numpy as np
import pytest
import torch

from gpu_util import check

pytestmark = pytest.mark.gpu


def test_image_predictor_matches_oracle(sd_large, cfg_large):
    from oracle import sam2_ref as R
    from sam2_opt_amd.image_predictor import SAM2ImagePredictor
    imgs = [np.random.RandomState(10 + i).randint(0, 256, (1024, 1024, 3)).astype(np.uint8) for i in range(2)]
    pred = SAM2ImagePredictor("large", state_dict=sd_large, max_batch=2)
    try:
        pred.set_image_batch(imgs)
        for i, im in enumerate(imgs):
            pts = (np.random.RandomState(100 + i).rand(8, 1, 2) * 1024).astype(np.float32)
            lab = np.ones((8, 1), np.int32)
            img01 = torch.from_numpy(im).permute(2, 0, 1)[None].float() / 255.0
            with torch.inference_mode():
                feats = R.set_image_e2e(img01, sd_large, cfg_large)
                rm, ri, rl = R.image_predict(feats, torch.from_numpy(pts), torch.from_numpy(lab), True, (1024, 1024), sd_large, cfg_large)
            m, s, l = pred._predict(pts, lab, None, None, True, True, img_idx=i)      # normalize by (1024,1024), then x1024
            check(f"image {i} masks (8 prompts x 3)", m, rm, 1e-2, 5e-3)
            check(f"image {i} ious", s, ri, 5e-3, 5e-3)
            check(f"image {i} low_res", l, rl, 1e-2, 5e-3)
        # one 8-point prompt, single-mask output with the stability fallback (T = 15 tokens)
        pts = (np.random.RandomState(7).rand(1, 8, 2) * 1024).astype(np.float32)
        lab = np.array([[1, 0, 1, 1, 0, 1, 1, 0]], np.int32)
        with torch.inference_mode():
            rm, ri, rl = R.image_predict(feats, torch.from_numpy(pts), torch.from_numpy(lab), False, (1024, 1024), sd_large, cfg_large)
        m, s, l = pred._predict(pts, lab, None, None, False, True, img_idx=1)
        check("image 1 single-mask (8-point prompt)", m, rm, 1e-2, 5e-3)
        check("image 1 single-mask iou", s, ri, 5e-3, 5e-3)
        # public API returns numpy, thresholded
        mb, sb, lb = pred.predict(pts[0], lab[0], multimask_output=True, normalize_coords=True)
        assert mb.shape == (3, 1024, 1024) and set(np.unique(mb)) <= {0.0, 1.0} and sb.shape == (3,) and lb.shape == (3, 256, 256)
    finally:
        pred.release()


def test_batched_prompts_equal_single_prompt_calls(sd_large):
    """The decoder batches prompts (chunks of 16, SURVEY 8 f-4: the automatic mask generator's prompt batches): 20 prompts on
    one image in one call give exactly what 20 single-prompt calls give."""
    from sam2_opt_amd.image_predictor import SAM2ImagePredictor
    img = np.random.RandomState(3).randint(0, 256, (1024, 1024, 3)).astype(np.uint8)
    pred = SAM2ImagePredictor("large", state_dict=sd_large, max_batch=1)
    try:
        pred.set_image(img)
        pts = (np.random.RandomState(5).rand(20, 1, 2) * 1024).astype(np.float32)
        lab = np.ones((20, 1), np.int32)
        m, s, l = pred._predict(pts, lab, None, None, True, True, img_idx=0)
        assert m.shape == (20, 3, 1024, 1024) and s.shape == (20, 3)
        for i in (0, 7, 15, 16, 19):
            mi, si, li = pred._predict(pts[i:i + 1], lab[i:i + 1], None, None, True, True, img_idx=0)
            assert torch.equal(mi[0], m[i]) and torch.equal(si[0], s[i]) and torch.equal(li[0], l[i])
        # single-mask output (stability fallback) for a batch as well
        m1, s1, _ = pred._predict(pts[:18], lab[:18], None, None, False, True, img_idx=0)
        assert m1.shape == (18, 1, 1024, 1024) and s1.shape == (18, 1)
        m2, s2, _ = pred._predict(pts[17:18], lab[17:18], None, None, False, True, img_idx=0)
        assert torch.equal(m2[0], m1[17]) and torch.equal(s2[0], s1[17])
    finally:
        pred.release()


def test_config5_batch16_eight_prompts_each(sd_large, cfg_large):
    """BASELINE.json configs[4] at its stated size: ONE encoder call on 16 images (M = 65,536 tokens in stage 3: the kernel
    selection differs from small batches), 8 independent single-point prompts per image.  Images 3 and 12 are held to the
    CPU oracle, all 16 to the batch-1 HIP result (the batch-1 path is the one the oracle tests above pin)."""
    import time
    from oracle import sam2_ref as R
    from sam2_opt_amd.image_predictor import SAM2ImagePredictor
    B = 16
    imgs = [np.random.RandomState(10 + i).randint(0, 256, (1024, 1024, 3)).astype(np.uint8) for i in range(B)]
    pts = [(np.random.RandomState(100 + i).rand(8, 1, 2) * 1024).astype(np.float32) for i in range(B)]
    lab = np.ones((8, 1), np.int32)
    p16 = SAM2ImagePredictor("large", state_dict=sd_large, max_batch=B)
    p1 = SAM2ImagePredictor("large", state_dict=sd_large, max_batch=1)
    try:
        p16.set_image_batch(imgs)                                  # warm-up, then a timed pass for the images/s figure
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        p16.set_image_batch(imgs)
        got = [p16._predict(pts[i], lab, None, None, True, True, img_idx=i) for i in range(B)]
        torch.cuda.synchronize()
        print(f"[config5] 16 images x 8 prompts: {B / (time.perf_counter() - t0):.1f} images/s (one encoder call of 16)", flush=True)
        for i in range(B):
            p1.set_image(imgs[i])
            m1, s1, l1 = p1._predict(pts[i], lab, None, None, True, True, img_idx=0)
            m, s, l = got[i]
            assert m.shape == (8, 3, 1024, 1024) and s.shape == (8, 3)
            check(f"config5 image {i} masks: batch-16 vs batch-1", m, m1, 4e-3, 2e-3)
            check(f"config5 image {i} ious: batch-16 vs batch-1", s, s1, 2e-3, 2e-3)
        for i in (3, 12):
            img01 = torch.from_numpy(imgs[i]).permute(2, 0, 1)[None].float() / 255.0
            with torch.inference_mode():
                feats = R.set_image_e2e(img01, sd_large, cfg_large)
                rm, ri, rl = R.image_predict(feats, torch.from_numpy(pts[i]), torch.from_numpy(lab), True, (1024, 1024), sd_large, cfg_large)
            m, s, l = got[i]
            check(f"config5 image {i} masks vs oracle", m, rm, 1e-2, 5e-3)
            check(f"config5 image {i} ious vs oracle", s, ri, 5e-3, 5e-3)
            check(f"config5 image {i} low_res vs oracle", l, rl, 1e-2, 5e-3)
    finally:
        p16.release()
        p1.release()


def test_amg_style_point_grid_batch_and_refinement_prompts(sd_large, cfg_large):
    """SURVEY 8 f-4, the SAM2AutomaticMaskGenerator caller (automatic_mask_generator.py:294-385): `_process_batch` hands
    `predictor._predict` 64 grid points as 64 single-point prompts (in_points[:, None, :], labels of ones, multimask, logits) on a
    non-square image - here through the reference's own call shape (tensors in the model frame via `_transforms.transform_coords`),
    four prompts held to the oracle and all 64 to single-prompt calls.  Then the refinement prompts of `_predict` (:487-589): a box
    with a point, and a mask_input (the low-res logits of the first pass) with a point, vs the oracle."""
    from oracle import sam2_ref as R
    from sam2_opt_amd.image_predictor import SAM2ImagePredictor
    H, W = 720, 1280
    img = np.random.RandomState(31).randint(0, 256, (H, W, 3)).astype(np.uint8)
    pred = SAM2ImagePredictor("large", state_dict=sd_large, max_batch=1)
    try:
        pred.set_image(img)
        gy, gx = np.meshgrid((np.arange(8) + 0.5) / 8 * H, (np.arange(8) + 0.5) / 8 * W, indexing="ij")
        points = torch.as_tensor(np.stack([gx.ravel(), gy.ravel()], -1), dtype=torch.float32, device=pred.device)      # (64, 2) in pixels
        in_points = pred._transforms.transform_coords(points, normalize=True, orig_hw=(H, W))
        in_labels = torch.ones(in_points.shape[0], dtype=torch.int, device=in_points.device)
        masks, iou, low = pred._predict(in_points[:, None, :], in_labels[:, None], multimask_output=True, return_logits=True)
        assert masks.shape == (64, 3, H, W) and iou.shape == (64, 3) and low.shape == (64, 3, 256, 256)
        for i in (0, 17, 40, 63):
            mi, si, li = pred._predict(in_points[i:i + 1, None, :], in_labels[i:i + 1, None], multimask_output=True, return_logits=True)
            assert torch.equal(mi[0], masks[i]) and torch.equal(si[0], iou[i])
        t = torch.from_numpy(img).permute(2, 0, 1)[None].float().div(255)
        img01 = torch.nn.functional.interpolate(t, size=(1024, 1024), mode="bilinear", align_corners=False, antialias=True)
        with torch.inference_mode():
            feats = R.set_image_e2e(img01, sd_large, cfg_large)
            sel = [0, 17, 40, 63]
            rm, ri, rl = R.image_predict(feats, in_points[sel][:, None, :].cpu(), in_labels[sel][:, None].cpu(), True, (H, W), sd_large, cfg_large)
        check("AMG batch masks (4 of 64) vs oracle", masks[sel], rm, 1e-2, 5e-3)
        check("AMG batch ious vs oracle", iou[sel], ri, 5e-3, 5e-3)
        # refinement: box + positive point, then the same point + the previous low-res logits as mask_input
        box = np.array([300.0, 150.0, 900.0, 600.0], np.float32)
        pt, lb = np.array([[640.0, 360.0]], np.float32), np.array([1], np.int32)
        mb, sb, lowb = pred.predict(pt, lb, box=box, multimask_output=False, return_logits=True)
        mask_in = lowb[0:1]                                                   # (1, 256, 256) like the reference's predict() returns
        mm, sm, lowm = pred.predict(pt, lb, mask_input=mask_in, multimask_output=True, return_logits=True)
        assert mb.shape == (1, H, W) and mm.shape == (3, H, W)
        sx, sy = 1024.0 / W, 1024.0 / H
        bpts = torch.tensor([[[box[0] * sx, box[1] * sy], [box[2] * sx, box[3] * sy], [pt[0, 0] * sx, pt[0, 1] * sy]]])
        with torch.inference_mode():
            rb = R.image_predict(feats, bpts, torch.tensor([[2, 3, 1]], dtype=torch.int32), False, (H, W), sd_large, cfg_large)
            rmk = R.image_predict(feats, bpts[:, 2:3], torch.tensor([[1]], dtype=torch.int32), True, (H, W), sd_large, cfg_large,
                                  mask_input=rb[2][:, 0:1])
        check("box + point (single mask, stability fallback) vs oracle", torch.from_numpy(mb)[None], rb[0], 1e-2, 5e-3)
        check("point + mask_input vs oracle", torch.from_numpy(mm)[None], rmk[0], 2e-2, 1e-2)
        check("point + mask_input ious vs oracle", torch.from_numpy(sm)[None], rmk[1], 1e-2, 1e-2)
    finally:
        pred.release()
