"""GPU: the padded-window model sizes (SAM 2.1 hiera tiny / small / base+: window spec 8/4/14/7, head_dim 96 / 56) on the HIP path
(csrc/hiera_generic.hip: zero-padded windows, valid-key masks, q pooling inside padded windows - backbones/utils.py:16-60).
 * every Hiera block of hiera-tiny vs the oracle fed with the oracle's block input;
 * the image encoder of all three sizes vs the oracle and vs golden vectors of the REAL reference (tests/golden/sizes_encoder.npz,
   tiny: tests/golden/tiny_image.npz);
 * BASELINE.json configs[0] end to end on the device: hiera-tiny image predictor, one click, vs the REAL reference's masks / IoUs;
 * a 4-frame hiera-tiny video propagation vs the oracle."""
import os

import numpy as np
import pytest
import torch

from gpu_util import check

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _weights(name):
    from sam2_opt_amd.config import get_config
    from sam2_opt_amd.weights import synthetic_state_dict
    cfg = get_config(name)
    return cfg, synthetic_state_dict(cfg, seed=0)


def _sampled(g, name, t, tol):
    stride, size = (int(v) for v in g[name + "/meta"])
    a = t.detach().float().cpu().numpy().reshape(-1)
    assert a.size == size, (name, a.size, size)
    got, ref = a[::stride], g[name + "/sample"]
    m = float(np.abs(got - ref).max() / np.abs(ref).max())
    l2 = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
    print(f"[parity] {name} vs reference golden: max_rel={m:.3e} l2={l2:.3e}", flush=True)
    assert m <= tol[0] and l2 <= tol[1], (name, m, l2)


def test_tiny_blocks_match_oracle():
    from oracle import sam2_ref as R
    from sam2_opt_amd.config import hiera_block_specs
    from sam2_opt_amd.native import Engine
    from sam2_opt_amd.synthetic import synthetic_image_normed
    cfg, sd = _weights("tiny")
    n = len(hiera_block_specs(cfg))
    blocks = {i: None for i in range(-1, n)}
    with torch.inference_mode():
        R.image_encoder(synthetic_image_normed(seed=1), sd, cfg, blocks)
    e = Engine("tiny", state_dict=sd, max_batch=1)
    try:
        for i in range(n):            # windows 8 / 4 / 14 (padded) / 7 (padded), the three pooled transitions, the global blocks
            out = e.debug_hiera_block(i, blocks[i - 1].cuda().contiguous(), blocks[i].shape)
            check(f"tiny hiera block {i}", out, blocks[i], 5e-3, 2e-3)
    finally:
        e.close()


@pytest.mark.parametrize("name", ["tiny", "small", "base_plus"])
def test_encoder_of_every_size(name):
    from oracle import sam2_ref as R
    from sam2_opt_amd.native import Engine
    from sam2_opt_amd.synthetic import synthetic_image_normed
    cfg, sd = _weights(name)
    img = synthetic_image_normed(seed=1)
    with torch.inference_mode():
        ref = R.image_encoder(img, sd, cfg)
    e = Engine(name, state_dict=sd, max_batch=2)
    try:
        got = e.image_encoder(img.cuda())
        for k, n in ((0, "vision_features"), (4, "backbone_fpn0"), (5, "backbone_fpn1")):
            check(f"{name} encoder/{n}", got[k], ref[k], 5e-3, 3e-3)
        if name != "tiny":
            g = np.load(os.path.join(ROOT, "tests", "golden", "sizes_encoder.npz"))
            for k, n in ((0, "vision_features"), (4, "backbone_fpn0"), (5, "backbone_fpn1"), (6, "backbone_fpn2")):
                _sampled(g, f"{name}/{n}", got[k], (5e-3, 3e-3))
        img2 = torch.cat([synthetic_image_normed(seed=7), img], 0).cuda()            # batch 2: same result for the second image
        got2 = e.image_encoder(img2)
        check(f"{name} encoder batch 2 vs batch 1", got2[0][1:2], got[0], 2e-3, 1e-3)
    finally:
        e.close()


def test_config0_tiny_image_predictor_matches_reference_golden():
    """BASELINE.json configs[0] (hiera-tiny image predictor, one 1024^2 frame, one click) - the reference's CPU-runnable case - on
    the device, vs the REAL reference's outputs (tests/golden/tiny_image.npz, oracle/gen_golden.py::gen_tiny)."""
    from sam2_opt_amd.image_predictor import SAM2ImagePredictor
    cfg, sd = _weights("tiny")
    g = np.load(os.path.join(ROOT, "tests", "golden", "tiny_image.npz"))
    img = np.random.RandomState(0).randint(0, 256, (1024, 1024, 3)).astype(np.uint8)
    pred = SAM2ImagePredictor("tiny", state_dict=sd, max_batch=1)
    try:
        pred.set_image(img)
        masks, ious, low = pred.predict(point_coords=np.array([[512.0, 512.0]], np.float32), point_labels=np.array([1], np.int32),
                                        multimask_output=True, return_logits=True)
        _sampled(g, "tiny/masks_logits", torch.from_numpy(masks), (5e-3, 5e-3))
        _sampled(g, "tiny/ious", torch.from_numpy(ious), (5e-3, 5e-3))
        _sampled(g, "tiny/low_res", torch.from_numpy(low), (5e-3, 5e-3))
    finally:
        pred.release()


def test_tiny_video_matches_oracle():
    from oracle import sam2_ref as R
    from sam2_opt_amd.synthetic import normalize_frames, synthetic_frames_u8
    from sam2_opt_amd.video_predictor import SAM2VideoPredictor
    cfg, sd = _weights("tiny")
    frames = normalize_frames(synthetic_frames_u8(seed=2, num_frames=4), cfg)
    click = (np.array([[512.0, 512.0]], np.float32), np.array([1], np.int32))
    pred = SAM2VideoPredictor("tiny", state_dict=sd, encode_batch=2)
    try:
        st = pred.init_state(frames=frames, video_height=1024, video_width=1024)
        pred.add_new_points_or_box(st, 0, 1, points=click[0], labels=click[1])
        got = {t: m.float().cpu() for t, _, m in pred.propagate_in_video(st)}
        vo = R.VideoOracle(sd, cfg, frames)
        with torch.inference_mode():
            vo.add_new_points(0, *click)
            ref = {t: m for t, m in vo.propagate()}
        for t in sorted(ref):
            check(f"tiny video frame {t}", got[t], ref[t], 6e-3, 6e-3)
    finally:
        pred.release()
