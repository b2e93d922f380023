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
            m, s, l = pred._predict(pts, lab, None, None, True, True, True, i)      # normalize by (1024,1024), then x1024
            check(f"image {i} masks (8 prompts x 3)", m, rm, 1e-2, 5e-3)
            check(f"image {i} ious", s, ri, 5e-3, 5e-3)
            check(f"image {i} low_res", l, rl, 1e-2, 5e-3)
        # one 8-point prompt, single-mask output with the stability fallback (T = 15 tokens)
        pts = (np.random.RandomState(7).rand(1, 8, 2) * 1024).astype(np.float32)
        lab = np.array([[1, 0, 1, 1, 0, 1, 1, 0]], np.int32)
        with torch.inference_mode():
            rm, ri, rl = R.image_predict(feats, torch.from_numpy(pts), torch.from_numpy(lab), False, (1024, 1024), sd_large, cfg_large)
        m, s, l = pred._predict(pts, lab, None, None, False, True, True, 1)
        check("image 1 single-mask (8-point prompt)", m, rm, 1e-2, 5e-3)
        check("image 1 single-mask iou", s, ri, 5e-3, 5e-3)
        # public API returns numpy, thresholded
        mb, sb, lb = pred.predict(pts[0], lab[0], multimask_output=True, normalize_coords=True)
        assert mb.shape == (3, 1024, 1024) and set(np.unique(mb)) <= {0.0, 1.0} and sb.shape == (3,) and lb.shape == (3, 256, 256)
    finally:
        pred.release()
