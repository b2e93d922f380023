"""Drop-in adapter: install the MI355X backend on the REFERENCE's own predictor objects.

The reference switches backends by re-pointing function attributes
(`SAM2Base.set_runtime_backend`, modeling/sam2_base_official.py:230-276; `MemoryAttention.set_runtime_backend`,
modeling/memory_attention.py:134-261; `MaskDecoder`, `MemoryEncoder`, `SAM2ImagePredictor.set_runtime_backend`)
and keeps executor objects with `Inference / warmup / Release / GetModelInputDesc` in `backend_contexts`
(duck type of `ytools.executor.ModelExectuor`).  `speedup_hip(predictor)` does exactly that with
`HipExecutor` objects that call libsam2mi.so through the C ABI (include/sam2mi.h); `predictor.speedup("torch")`
restores the PyTorch methods as usual because it only re-points the same attributes.

Usage with the reference:

    predictor = build_sam2_video_predictor(cfg, ckpt, device="cuda")      # reference, on ROCm PyTorch
    from sam2_opt_amd.plugin import speedup_hip
    speedup_hip(predictor)                                                 # instead of predictor.speedup("tensorrt")
"""
from __future__ import annotations

from typing import List

import torch

from .native import Engine

BACKEND_NAMES = ("hip", "mi355x", "sam2mi")


class HipExecutor:
    """Executor duck type the reference stores in `backend_contexts` (sam2_image_predictor.py:19,:71,:205-207,:269)."""

    def __init__(self, engine: Engine, fn, n_inputs: int):
        self.engine, self.fn, self.n_inputs = engine, fn, n_inputs

    def Inference(self, inputs: List[torch.Tensor], output_type: str = "torch"):
        outs = self.fn(*[t.to(self.engine.device, torch.float32).contiguous() for t in inputs])
        return list(outs) if isinstance(outs, (tuple, list)) else [outs]

    def warmup(self, inputs: List[torch.Tensor]):
        self.Inference(inputs)

    def GetModelInputDesc(self):
        return [None] * self.n_inputs

    def Release(self):
        pass                                   # the engine is shared; released by release_hip()


def _detect_model(sd) -> str:
    """Which SAM 2.1 size is this state_dict?  (embed_dim, number of Hiera blocks) identify it; anything else must fail here, at
    attach time, not as out-of-bounds device reads."""
    from .config import MODEL_CONFIGS
    w = sd.get("image_encoder.trunk.patch_embed.proj.weight")
    nblk = len({k.split(".")[3] for k in sd if k.startswith("image_encoder.trunk.blocks.")})
    for name, cfg in MODEL_CONFIGS.items():
        if w is not None and int(w.shape[0]) == cfg["embed_dim"] and nblk == sum(cfg["stages"]):
            return name
    got = None if w is None else int(w.shape[0])
    raise RuntimeError(f"speedup('hip'): not a SAM 2.1 hiera tiny / small / base+ / large model (embed_dim {got}, {nblk} blocks)")


def _engine_for(model, precision: str = "f16", engine=None) -> Engine:
    eng = getattr(model, "_sam2mi_engine", None)
    if eng is None:
        if engine is not None:                   # injected (tests drive the adapter without a GPU)
            eng = engine
        else:
            sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
            size = _detect_model(sd)
            dev = next(model.parameters()).device
            if dev.type != "cuda":
                raise RuntimeError("speedup('hip') needs the model on a ROCm GPU (model.to('cuda'))")
            eng = Engine(size, state_dict=sd, max_batch=1, device=dev, precision=precision)
        model._sam2mi_engine = eng
    return eng


PLUGS = ("image", "memory_attention", "mask_decoder", "memory_encoder", "prompt_encoder")


def speedup_hip(predictor, plugs=PLUGS, precision: str = "f16", engine=None):
    """Install the HIP backend on a reference SAM2VideoPredictor / SAM2Base / SAM2ImagePredictor: re-points exactly the
    attributes the reference's own `set_runtime_backend` methods re-point and keeps executor objects in
    `backend_contexts`, so `predictor.speedup("torch")` / `release()` restore the PyTorch path as for any other backend."""
    model = getattr(predictor, "model", predictor)        # SAM2ImagePredictor wraps the SAM2Base in .model
    unknown = set(plugs) - set(PLUGS)
    if unknown:
        raise ValueError(f"unknown plug(s) {sorted(unknown)}; choose from {PLUGS}")
    eng = _engine_for(model, precision, engine)
    if "image" in plugs:
        ex = HipExecutor(eng, eng.image_encoder, 1)
        model.backend_contexts = [ex]
        model.inference_image = lambda img, _ex=ex: tuple(_ex.Inference([img]))
        if hasattr(predictor, "set_image_e2e"):            # SAM2ImagePredictor
            ex2 = HipExecutor(eng, eng.set_image_e2e, 1)
            predictor.backend_contexts = [ex2]
            predictor.set_image_e2e = lambda img, _ex=ex2: tuple(_ex.Inference([img]))
    ma = getattr(model, "memory_attention", None)
    if ma is not None and "memory_attention" in plugs:
        ex = HipExecutor(eng, eng.memory_attention, 6)
        ma.backend_contexts = [ex, ex]
        fn = lambda *a, _ex=ex: _ex.Inference(list(a))[0]           # noqa: E731
        ma.inference_memory_attention_exclude = fn
        ma.inference_memory_attention_none = fn
    md = getattr(model, "sam_mask_decoder", None)
    if md is not None and "mask_decoder" in plugs:
        ex = HipExecutor(eng, eng.mask_decoder, 5)
        md.backend_contexts = [ex]
        md.inference_predict_masks = lambda *a, _ex=ex: tuple(_ex.Inference(list(a)))
    me = getattr(model, "memory_encoder", None)
    if me is not None and "memory_encoder" in plugs:
        ex = HipExecutor(eng, eng.memory_encoder, 2)
        me.backend_contexts = [ex]
        me.inference_memory = lambda pix, m, _ex=ex: tuple(_ex.Inference([pix, m]))
    pe = getattr(model, "sam_prompt_encoder", None)
    if pe is not None and "prompt_encoder" in plugs:
        # PromptEncoder.inference_prompt(points, boxes, masks) (prompt_encoder.py:211,:215-231): arguments are optional /
        # a tuple, so this plug does not go through the list-of-tensors executor call; the executor object is still kept in
        # backend_contexts for Release()
        ex = HipExecutor(eng, eng.prompt_encoder_full, 3)
        pe.backend_contexts = [ex]
        pe.inference_prompt = lambda points, boxes, masks, _e=eng: _e.prompt_encoder_full(points, boxes, masks)
    return predictor


def release_hip(predictor):
    """Back to the PyTorch methods (what `predictor.speedup("torch")` / `release()` do) and free the engine."""
    model = getattr(predictor, "model", predictor)
    seen = set()
    for mod in (predictor, model, getattr(model, "memory_attention", None), getattr(model, "sam_mask_decoder", None),
                getattr(model, "memory_encoder", None), getattr(model, "sam_prompt_encoder", None)):
        if mod is None or id(mod) in seen:
            continue
        seen.add(id(mod))
        if hasattr(mod, "set_runtime_backend"):
            mod.set_runtime_backend("torch")        # releases the executors in backend_contexts, then re-points the attributes
    eng = getattr(model, "_sam2mi_engine", None)
    if eng is not None:
        eng.close()
        model._sam2mi_engine = None
