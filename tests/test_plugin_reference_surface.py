"""CPU, build container only: the drop-in adapter (sam2_opt_amd/plugin.py) against the REAL reference classes
(/root/reference, imported through oracle/ref_import.py) with a stub engine - no GPU involved.

 * speedup_hip only touches attributes that exist on SAM2VideoPredictor / MemoryAttention / MaskDecoder / MemoryEncoder /
   PromptEncoder / SAM2ImagePredictor and that the reference's own set_runtime_backend methods re-point;
 * the reference's modules then really call the installed plugs (forward_image, memory attention with and without pointers,
   the mask decoder, the memory encoder, the prompt encoder with points / boxes / masks), in the reference's own tensor
   layouts;
 * release_hip / predictor.speedup("torch") restore every inference_*_torch method.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/sam2/sam2"), reason="needs /root/reference (build container only)")


class StubEngine:
    """Records calls; returns tensors of the plug's output shapes (ytools-executor style: fresh tensors, caller's device)."""

    def __init__(self):
        self.calls = []
        self.device = torch.device("cpu")
        self.closed = False

    def image_encoder(self, img):
        self.calls.append(("image_encoder", tuple(img.shape)))
        B = img.shape[0]
        return (torch.zeros(B, 256, 64, 64), torch.zeros(B, 256, 256, 256), torch.zeros(B, 256, 128, 128), torch.zeros(B, 256, 64, 64),
                torch.zeros(B, 32, 256, 256), torch.zeros(B, 64, 128, 128), torch.zeros(B, 256, 64, 64))

    def set_image_e2e(self, img01):
        self.calls.append(("set_image_e2e", tuple(img01.shape)))
        B = img01.shape[0]
        return torch.zeros(B, 32, 256, 256), torch.zeros(B, 64, 128, 128), torch.zeros(B, 256, 64, 64)

    def memory_attention(self, curr, memory, curr_pos, memory_pos, mem_ex, mem_pos_ex):
        self.calls.append(("memory_attention", tuple(memory.shape), tuple(mem_ex.shape)))
        return torch.zeros_like(curr)

    def mask_decoder(self, src, tokens, pos_src, hr0, hr1):
        self.calls.append(("mask_decoder", tuple(src.shape), tuple(tokens.shape)))
        N = src.shape[0]
        return torch.zeros(N, 4, 256, 256), torch.zeros(N, 4), torch.zeros(N, 4, 256), torch.full((N, 1), 5.0)

    def memory_encoder(self, pix, masks):
        self.calls.append(("memory_encoder", tuple(pix.shape), tuple(masks.shape)))
        return torch.zeros(pix.shape[0], 64, 64, 64), torch.zeros(pix.shape[0], 64, 64, 64)

    def prompt_encoder_full(self, points, boxes, masks):
        self.calls.append(("prompt_encoder", None if points is None else tuple(points[0].shape), None if boxes is None else tuple(boxes.shape),
                           None if masks is None else tuple(masks.shape)))
        B = points[0].shape[0] if points is not None else (boxes.shape[0] if boxes is not None else masks.shape[0])
        S = (points[0].shape[1] if points is not None else 0) + (2 if boxes is not None else 1 if points is not None else 0)
        return torch.zeros(B, S, 256), torch.zeros(B, 256, 64, 64)

    def close(self):
        self.closed = True


@pytest.fixture(scope="module")
def ref_predictor():
    from oracle.ref_import import build_reference_model
    from sam2_opt_amd.config import get_config
    torch.manual_seed(0)
    return build_reference_model(get_config("tiny"), "video", None)      # class surface is independent of the trunk size


PLUG_ATTRS = [("", "inference_image"), ("memory_attention", "inference_memory_attention_exclude"), ("memory_attention", "inference_memory_attention_none"),
              ("sam_mask_decoder", "inference_predict_masks"), ("memory_encoder", "inference_memory"), ("sam_prompt_encoder", "inference_prompt")]


def _owner(model, path):
    return getattr(model, path) if path else model


def test_speedup_hip_touches_only_existing_reference_attributes(ref_predictor):
    from sam2_opt_amd.plugin import release_hip, speedup_hip
    model = ref_predictor
    for path, attr in PLUG_ATTRS:                   # every plug attribute exists on the reference class and starts on *_torch
        obj = _owner(model, path)
        assert hasattr(obj, attr) and hasattr(obj, "set_runtime_backend") and isinstance(obj.backend_contexts, list), (path, attr)
        assert getattr(obj, attr).__name__.endswith("_torch")
    before = {path: set(vars(_owner(model, path))) for path, _ in PLUG_ATTRS}
    eng = StubEngine()
    speedup_hip(model, engine=eng)
    for path, attr in PLUG_ATTRS:
        obj = _owner(model, path)
        new = set(vars(obj)) - before[path]
        assert new <= {"_sam2mi_engine"}, f"adapter created attributes the reference does not have on {path or 'SAM2Base'}: {new}"
        assert not getattr(obj, attr).__name__.endswith("_torch")
        for ex in obj.backend_contexts:             # executor duck type
            assert all(callable(getattr(ex, n)) for n in ("Inference", "warmup", "Release", "GetModelInputDesc"))
    assert len(model.memory_attention.backend_contexts) == 2       # [none, exclude] like the reference's ORT / TRT backends

    # ---- the reference's own code paths call the plugs, in its layouts
    with torch.inference_mode():
        out = model.forward_image(torch.zeros(1, 3, 1024, 1024))
        assert ("image_encoder", (1, 3, 1024, 1024)) in eng.calls and set(out) >= {"vision_features", "vision_pos_enc", "backbone_fpn"}
        curr = [torch.zeros(4096, 1, 256)]
        model.memory_attention(curr=curr, curr_pos=curr, memory=torch.zeros(2 * 4096 + 8, 1, 64), memory_pos=torch.zeros(2 * 4096 + 8, 1, 64),
                               num_obj_ptr_tokens=8)
        assert eng.calls[-1] == ("memory_attention", (2, 4096, 1, 64), (8, 1, 64))
        model.memory_attention(curr=curr, curr_pos=curr, memory=torch.zeros(4096, 1, 64), memory_pos=torch.zeros(4096, 1, 64), num_obj_ptr_tokens=0)
        assert eng.calls[-1][0] == "memory_attention" and eng.calls[-1][2][0] == 0
        pe = model.sam_prompt_encoder
        pts = (torch.zeros(1, 1, 2), torch.ones(1, 1, dtype=torch.int32))
        sp, de = pe(points=pts, boxes=None, masks=None)
        assert eng.calls[-1] == ("prompt_encoder", (1, 1, 2), None, None) and tuple(sp.shape) == (1, 2, 256)
        pe(points=None, boxes=torch.zeros(1, 4), masks=torch.zeros(1, 1, 256, 256))
        assert eng.calls[-1] == ("prompt_encoder", None, (1, 4), (1, 1, 256, 256))
        n = len(eng.calls)
        model._forward_sam_heads(backbone_features=torch.zeros(1, 256, 64, 64), point_inputs=None, mask_inputs=None,
                                 high_res_features=[torch.zeros(1, 32, 256, 256), torch.zeros(1, 64, 128, 128)], multimask_output=True)
        kinds = [c[0] for c in eng.calls[n:]]
        assert kinds == ["prompt_encoder", "mask_decoder"] and eng.calls[-1][2] == (1, 8, 256)      # 6 output tokens + pad point + pad
        model.memory_encoder(torch.zeros(1, 256, 64, 64), torch.zeros(1, 1, 1024, 1024), skip_mask_sigmoid=True)
        assert eng.calls[-1] == ("memory_encoder", (1, 256, 64, 64), (1, 1, 1024, 1024))

    release_hip(model)
    assert eng.closed and model._sam2mi_engine is None
    for path, attr in PLUG_ATTRS:
        obj = _owner(model, path)
        assert getattr(obj, attr).__name__.endswith("_torch") and obj.backend_contexts == [], (path, attr)


def test_reference_speedup_torch_also_restores(ref_predictor):
    """`predictor.speedup("torch")` (sam2_video_predictor_official.py:58-60) is the reference's own way back."""
    from sam2_opt_amd.plugin import speedup_hip
    model = ref_predictor
    speedup_hip(model, plugs=("image", "memory_attention"), engine=StubEngine())
    model.speedup("torch")
    assert model.inference_image.__name__ == "inference_image_torch"
    assert model.memory_attention.inference_memory_attention_none.__name__ == "inference_memory_attention_torch"
    model._sam2mi_engine = None


def test_image_predictor_surface():
    from oracle.ref_import import build_reference_model
    from sam2.sam2_image_predictor import SAM2ImagePredictor
    from sam2_opt_amd.config import get_config
    from sam2_opt_amd.plugin import release_hip, speedup_hip
    pred = SAM2ImagePredictor(build_reference_model(get_config("tiny"), "base", None))
    assert pred.set_image_e2e.__name__.endswith("_torch")
    eng = StubEngine()
    speedup_hip(pred, engine=eng)
    assert not pred.set_image_e2e.__name__.endswith("_torch") and len(pred.backend_contexts) == 1
    release_hip(pred)
    assert pred.set_image_e2e.__name__.endswith("_torch") and eng.closed


def test_model_size_detection():
    from sam2_opt_amd.config import get_config
    from sam2_opt_amd.plugin import _detect_model
    from sam2_opt_amd.weights import state_dict_spec
    for name in ("tiny", "small", "base_plus", "large"):
        sd = {k: torch.zeros(1).expand(*shape) if len(shape) else torch.zeros(()) for k, shape in state_dict_spec(get_config(name)).items()}
        assert _detect_model(sd) == name
    sd = dict(sd)
    del sd["image_encoder.trunk.blocks.47.norm1.weight"], sd["image_encoder.trunk.blocks.47.norm1.bias"]
    for k in [k for k in sd if k.startswith("image_encoder.trunk.blocks.47.")]:
        del sd[k]
    with pytest.raises(RuntimeError, match="not a SAM 2.1"):
        _detect_model(sd)


def test_reference_video_loop_reaches_the_plug_with_clip_views_and_gets_lookahead(ref_predictor):
    """The reference's own `_get_image_feature` (sam2_video_predictor_official.py:810-841) on a device-resident clip tensor: the
    frame it hands to `inference_image` is a view of the clip, so the look-ahead plug encodes 8 frames per engine call, and a clip
    kept on another device (copied per frame, like `offload_video_to_cpu`) falls back to one frame per call."""
    from sam2_opt_amd.plugin import release_hip, speedup_hip
    model = ref_predictor
    eng = StubEngine()
    eng.max_batch = 8
    speedup_hip(model, plugs=("image",), engine=eng, lookahead=8)
    try:
        T = 11
        state = {"images": torch.zeros(T, 3, 1024, 1024), "device": torch.device("cpu"), "cached_features": {}}
        with torch.inference_mode():
            for t in range(T):
                feats = model._get_image_feature(state, t, 1)
                assert feats[0].shape == (1, 3, 1024, 1024)
        enc = [c[1][0] for c in eng.calls if c[0] == "image_encoder"]
        assert enc == [8, 3], enc
        # a half-precision clip: `.float()` makes a per-frame copy -> no view, one frame per call
        eng.calls.clear()
        state = {"images": torch.zeros(3, 3, 1024, 1024, dtype=torch.float16), "device": torch.device("cpu"), "cached_features": {}}
        with torch.inference_mode():
            for t in range(3):
                model._get_image_feature(state, t, 1)
        assert [c[1][0] for c in eng.calls if c[0] == "image_encoder"] == [1, 1, 1]
    finally:
        release_hip(model)
