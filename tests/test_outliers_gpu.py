"""GPU: outlier channels.  Trained ViT residual streams carry a few outlier channels; f16 has 5 exponent bits where the reference's
reduced-precision mode (bf16 autocast) has 8, and every parity figure of this repo is otherwise measured on tame synthetic weights.
Two scenarios with golden vectors from the REAL reference (oracle/gen_golden.py outliers -> tests/golden/large_outliers.npz), weights =
the synthetic ones with the LayerNorm gain of three channels x g in every norm of the Hiera trunk and of the memory attention,
recurrent damping off:

 * g = 64, the f16-RANGE scenario (GEMM operand channels ~80x the median: 223 vs 2.7 at block 20; residual stream up to 330;
   attention logits in the thousands).  The encoder as a whole is ill-conditioned there - with 22-bit operands (f16x3, 1e-5 per
   block) it still ends 2e-1 away from the reference, any two fp32 evaluations differ by 1e-2 - so the end-to-end check is
   FINITENESS in every mode, and parity is taken block by block on the oracle's inputs (the oracle is pinned to the reference on
   this scenario in tests/test_oracle_vs_golden.py): f16 <= 1.5e-1 max-abs / 6e-3 rel L2 per block (measured 1.0e-1 / 4.2e-3 at
   block 0: a softmax over logits of ~1e3 turns the 2^-11 operand rounding into O(1) changes of single probabilities),
   f16x3 <= 2e-4 / 1e-5.  No operand, hidden tensor, V^T or score overflows; nothing had to move to bf16.
 * g = 8 (~10x outliers, well-conditioned: f16x3 ends at 2e-5): end to end against the reference golden, per mode:
   f16 2e-2 / 6e-3, f16s 1e-2 / 3e-3 (measured 1.8e-2 / 4.6e-3 and 7.5e-3 / 2.2e-3 on the worst output, fpn1), f16x3 1e-4 / 2e-5.
   So the 1e-3 class of f16s is a statement about the tame weights: with 10x outlier channels its error is 2-4x
   that of the tame case, half of plain f16's; f16x3 keeps the ENCODER at 1e-5 regardless.
 * Memory attention is the exception in every mode: the d = 256 flash kernel keeps q / k / p / v in f16 in all three modes
   (on the video its share of the mask error is 4e-6), and with outlier gains on the layer norms that feed its q / k
   projections the logits reach the hundreds, where an 11-bit q / k moves single probabilities by O(1): x8 1.0e-2 / 3.6e-3 (f16,
   f16s) and 1.9e-2 / 2.1e-3 (f16x3) against the reference, x64 2e-2 .. 3e-2 rel L2.  Held to 3e-2 / 6e-3 (x8) and 5e-2 rel L2
   (x64) in all modes - a stated limit, not a parity claim: a split-q/k flash256 is what would lift it (not built)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL_E2E = {"f16": (2e-2, 6e-3), "f16s": (1e-2, 3e-3), "f16x3": (1e-4, 2e-5)}
TOL_MEMATTN = (3e-2, 6e-3)        # every mode: the f16 flash kernel is the floor (see above)
TOL_BLOCK = {"f16": (1.5e-1, 6e-3), "f16s": (1.5e-1, 6e-3), "f16x3": (2e-4, 1e-5)}


def _sampled(t, store, name):
    stride, size = (int(v) for v in store[name + "/meta"])
    a = t.detach().float().cpu().numpy().reshape(-1)
    assert a.size == size, (name, a.size, size)
    return a[::stride], store[name + "/sample"]


def _engine(cfg, gain, mode):
    from sam2_opt_amd.native import Engine
    from sam2_opt_amd.weights import synthetic_state_dict
    sd = synthetic_state_dict(cfg, seed=0, undamped=True, outlier_gain=gain)
    return sd, Engine("large", state_dict=sd, max_batch=2, precision=mode)


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "large_outliers.npz"))


@pytest.mark.parametrize("mode", ["f16", "f16s", "f16x3"])
def test_mild_outliers_end_to_end_vs_reference(cfg_large, golden, mode):
    from oracle.gen_golden import OUTLIER_GAIN_MILD, plug_inputs
    from sam2_opt_amd.synthetic import synthetic_image_normed
    _, eng = _engine(cfg_large, OUTLIER_GAIN_MILD, mode)
    try:
        outs = eng.image_encoder(synthetic_image_normed(seed=1).cuda())
        ma = eng.memory_attention(*[t.cuda() for t in plug_inputs(cfg_large)["memattn_L3P12"]])
        fails = []
        for name, t in (("enc/vision_features", outs[0]), ("enc/backbone_fpn0", outs[4]), ("enc/backbone_fpn1", outs[5]), ("enc/backbone_fpn2", outs[6]),
                        ("memattn_L3P12", ma)):
            got, ref = _sampled(t, golden, "g8/" + name)
            assert np.isfinite(got).all(), (mode, name)
            m = float(np.abs(got - ref).max() / np.abs(ref).max())
            l2 = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
            tol = TOL_MEMATTN if name.startswith("memattn") else TOL_E2E[mode]
            print(f"[parity] outliers x8 {mode} {name}: max_rel={m:.3e} l2_rel={l2:.3e} (tol {tol[0]:.0e}/{tol[1]:.0e})", flush=True)
            if m > tol[0] or l2 > tol[1]:
                fails.append((name, m, l2))
        assert not fails, (mode, fails)
    finally:
        eng.close()


@pytest.mark.parametrize("mode", ["f16", "f16s", "f16x3"])
def test_range_scenario_stays_finite_and_blocks_match(cfg_large, golden, mode):
    from oracle import sam2_ref as R
    from oracle.gen_golden import OUTLIER_GAIN, plug_inputs
    from sam2_opt_amd.synthetic import synthetic_image_normed
    sd, eng = _engine(cfg_large, OUTLIER_GAIN, mode)
    try:
        img = synthetic_image_normed(seed=1)
        for o in eng.image_encoder(img.cuda()):
            assert torch.isfinite(o).all(), mode
        ma = eng.memory_attention(*[t.cuda() for t in plug_inputs(cfg_large)["memattn_L3P12"]])
        assert torch.isfinite(ma).all(), mode
        got, ref = _sampled(ma, golden, "g64/memattn_L3P12")
        l2 = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
        print(f"[parity] outliers x64 {mode} memattn_L3P12: l2_rel={l2:.3e} (logits of ~1e3: single rows flip)", flush=True)
        assert l2 <= 5e-2, (mode, l2)
        idxs = (-1, 0, 1, 2, 3, 7, 8, 9, 22, 23, 43, 44, 45)
        blocks = {i: None for i in idxs}
        with torch.inference_mode():
            R.image_encoder(img, sd, cfg_large, blocks)
        fails = []
        for i in (0, 1, 2, 3, 8, 9, 23, 44, 45):
            x, refb = blocks[i - 1].cuda(), blocks[i]
            out = eng.debug_hiera_block(i, x, refb.shape).float().cpu()
            assert torch.isfinite(out).all(), (mode, i)
            d = out - refb
            m, l2 = float(d.abs().max() / refb.abs().max()), float(d.norm() / refb.norm())
            print(f"[parity] outliers x64 {mode} block {i}: max_rel={m:.3e} l2_rel={l2:.3e} max|x|={float(x.abs().max()):.3g}", flush=True)
            if m > TOL_BLOCK[mode][0] or l2 > TOL_BLOCK[mode][1]:
                fails.append((i, m, l2))
        assert not fails, (mode, fails)
    finally:
        eng.close()
