"""GPU: the drop-in adapter (sam2_opt_amd/plugin.py) installs the reference's plug attributes on a predictor-shaped
object (same attribute names / executor duck type as FasterProcess/sam2_opt; the reference itself cannot travel to the
GPU box) and the installed callables return reference-layout tensors that match the oracle."""
import pytest
import torch

from gpu_util import check

pytestmark = pytest.mark.gpu


class _Sub:
    def __init__(self):
        self.backend_contexts = []
        self.restored = 0

    def set_runtime_backend(self, backend="torch", args=None):
        assert backend == "torch"
        self.restored += 1


class _FakeSam2Base(_Sub):
    """Attribute surface of SAM2Base / SAM2VideoPredictor that speedup() touches (sam2_base_official.py:224-276)."""

    def __init__(self, sd, device):
        super().__init__()
        self._sd = {k: v.to(device) for k, v in sd.items()}
        self.memory_attention, self.sam_mask_decoder, self.memory_encoder, self.sam_prompt_encoder = _Sub(), _Sub(), _Sub(), _Sub()

    def state_dict(self):
        return self._sd

    def parameters(self):
        return iter(self._sd.values())


def test_speedup_hip_installs_reference_plugs(sd_large, cfg_large):
    from oracle import sam2_ref as R
    from oracle.gen_golden import plug_inputs
    from sam2_opt_amd.plugin import release_hip, speedup_hip
    from sam2_opt_amd.synthetic import randn
    dev = torch.device("cuda", 0)
    model = _FakeSam2Base(sd_large, dev)
    speedup_hip(model)
    try:
        # executor duck type kept in backend_contexts
        ex = model.memory_attention.backend_contexts[0]
        assert all(hasattr(ex, n) for n in ("Inference", "warmup", "Release", "GetModelInputDesc"))
        assert len(ex.GetModelInputDesc()) == 6
        inp = plug_inputs(cfg_large)["memattn_L1P4"]
        got = model.memory_attention.inference_memory_attention_exclude(*[t.to(dev) for t in inp])
        with torch.inference_mode():
            ref = R.memory_attention(*inp, sd_large, cfg_large)
        assert got.shape == ref.shape and got.device.type == "cuda"
        check("plugin memattn", got, ref, 5e-3, 2e-3)
        pix, masks = plug_inputs(cfg_large)["memenc"]
        x, pos = model.memory_encoder.inference_memory(pix.to(dev), masks.to(dev))
        with torch.inference_mode():
            rx, rpos = R.memory_encoder(pix, masks, sd_large, cfg_large)
        check("plugin memenc", x, rx, 5e-3, 2e-3)
        outs = model.inference_image(randn(3, 1, 3, 1024, 1024).to(dev))
        assert len(outs) == 7 and tuple(outs[4].shape) == (1, 32, 256, 256)
        pts, lab = plug_inputs(cfg_large)["prompt"]
        sp, de = model.sam_prompt_encoder.inference_prompt((pts.to(dev), lab.to(dev)), None, None)
        with torch.inference_mode():
            rsp, rde = R.prompt_encoder(pts, lab, sd_large, cfg_large)
        check("plugin prompt/sparse", sp, rsp, 1e-4, 1e-4)
        check("plugin prompt/dense", de, rde.contiguous(), 1e-6, 1e-6)
    finally:
        release_hip(model)
    assert model.memory_attention.restored == 1 and model.restored == 1 and model.sam_prompt_encoder.restored == 1
