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


class LookaheadImagePlug:
    """The image plug with look-ahead.  The reference encodes ONE frame per call and caches one frame
    (`_get_image_feature`, /root/reference/sam2/sam2/sam2_video_predictor_official.py:810-841: `inference_state["images"][frame_idx]
    ... .unsqueeze(0)` -> `forward_image` -> `inference_image`), which pins the encoder at batch 1 (5.99 ms / frame against 3.5 ms in
    batches of 8).  But the `img` the plug receives is a VIEW of frame t of the clip tensor: when its storage holds more frames
    behind (or before) it, this plug encodes a batch of `depth` consecutive frames in one engine call, serves the following calls
    from that batch, and - on the device - runs the NEXT batch on a side stream beside the tracking plugs of the current one (the
    engine's encoder and tracking workspaces are separate domains, include/sam2mi.h).  Outputs are the plug's usual 7-tuple for
    one frame (views of the batch outputs).

    Safety: a cached frame is identified by (storage address, element offset) of the view, and the cache holds a reference to
    that storage, so the address cannot be recycled for another clip while entries exist; a call on a different storage drops
    every entry.  A frame that is not a view of a larger contiguous clip (offloaded video, an image predictor, a copy) takes the
    plain batch-1 path.  Frames are assumed immutable while cached (the reference never writes `inference_state["images"]`)."""

    def __init__(self, engine, depth: int = 8, side_stream: bool = True):
        self.engine, self.depth = engine, max(1, int(depth))
        self.cache = {}                   # element offset -> (7 per-frame views, ready event or None)
        self.storage = None               # keeps the clip's storage alive while entries exist
        self.storage_ptr = None
        self.last_off = None
        self.step = 0                     # +1 / -1 frames per call, learnt from the calls
        self.stream = None
        self.use_side = side_stream
        self.stats = dict(calls=0, hits=0, batches=0, frames_encoded=0)
        import threading
        self._lock = threading.RLock()    # one predictor may be driven from several host threads (video_multi_thread.py): calls are serialised

    def clear(self):
        self.cache.clear()
        self.storage = self.storage_ptr = self.last_off = None
        self.step = 0

    def _frames_view(self, img, first_off, n):
        fe = img[0].numel()
        return torch.as_strided(img, (n,) + tuple(img.shape[1:]), (fe,) + tuple(img.stride()[1:]), first_off)

    def _encode(self, img, first_off, n, stream=None):
        """Encode frames at element offsets first_off + k * frame (k < n) in one call; returns {offset: (views, event)}."""
        fe = img[0].numel()
        batch = self._frames_view(img, first_off, n)
        ev = None
        if stream is not None:
            cur = torch.cuda.current_stream(img.device)
            stream.wait_stream(cur)
            with torch.cuda.stream(stream):
                outs = self.engine.image_encoder(batch)
                ev = torch.cuda.Event()
                ev.record(stream)
        else:
            outs = self.engine.image_encoder(batch)
        self.stats["batches"] += 1
        self.stats["frames_encoded"] += n
        return {first_off + k * fe: (tuple(o[k:k + 1] for o in outs), ev) for k in range(n)}

    def __call__(self, img: torch.Tensor):
        with self._lock:
            return self._call(img)

    def _call(self, img: torch.Tensor):
        self.stats["calls"] += 1
        eng = self.engine
        img = img.to(eng.device, torch.float32)
        st = img.untyped_storage()
        fe = img[0].numel() if img.dim() == 4 and img.shape[0] == 1 else 0
        total = st.nbytes() // 4
        lookable = (self.depth > 1 and fe > 0 and img.is_contiguous() and total >= 2 * fe and total % fe == 0 and
                    img.storage_offset() % fe == 0 and getattr(eng, "max_batch", 1) >= 2)
        if not lookable:
            return tuple(eng.image_encoder(img.contiguous()))
        off = img.storage_offset()
        if self.storage_ptr != st.data_ptr():
            self.clear()
            self.storage, self.storage_ptr = st, st.data_ptr()
        if self.last_off is not None and off != self.last_off:
            d = (off - self.last_off) // fe
            if d in (1, -1):
                self.step = d
        self.last_off = off
        hit = self.cache.pop(off, None)
        depth = min(self.depth, getattr(eng, "max_batch", 1))
        fwd = self.step >= 0
        if hit is None:
            # frames available in the direction of travel, starting at this one (a batch is always ascending in memory)
            n = min(depth, (total - off) // fe) if fwd else min(depth, off // fe + 1)
            first = off if fwd else off - (n - 1) * fe
            got = self._encode(img, first, n)
            hit = got.pop(off)
            self.cache.update(got)
        else:
            self.stats["hits"] += 1
        views, ev = hit
        if ev is not None:
            torch.cuda.current_stream(img.device).wait_event(ev)
            for v in views:
                v.record_stream(torch.cuda.current_stream(img.device))
        # keep the side stream one batch ahead: when the frames cached in the direction of travel run out after this batch
        if self.use_side and img.is_cuda and self.step != 0:
            ahead = [o for o in self.cache if (o > off) == fwd]
            if len(ahead) < depth // 2:
                edge = (max(ahead) if fwd else min(ahead)) if ahead else off
                nxt = edge + fe if fwd else edge - fe
                n = min(depth, (total - nxt) // fe) if fwd else min(depth, nxt // fe + 1)
                if n > 0 and 0 <= nxt < total:
                    if self.stream is None:
                        self.stream = torch.cuda.Stream(device=img.device)
                    first = nxt if fwd else nxt - (n - 1) * fe
                    self.cache.update(self._encode(img, first, n, self.stream))
        # bound the cache: frames behind the direction of travel are never asked for again
        if len(self.cache) > 3 * depth:
            for o in sorted(self.cache, key=lambda o: (o - off) * (1 if fwd else -1))[: len(self.cache) - 3 * depth]:
                del self.cache[o]
        return views


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


def _engine_for(model, precision: str = "f16", engine=None, max_batch: int = 1) -> Engine:
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
            if size != "large" and precision != "f16":
                import warnings
                warnings.warn(f"speedup('hip'): the split-operand precision modes exist for hiera-large only; hiera-{size} runs in 'f16'")
                precision = "f16"
            eng = Engine(size, state_dict=sd, max_batch=max_batch, device=dev, precision=precision)
        model._sam2mi_engine = eng
    return eng


PLUGS = ("image", "memory_attention", "mask_decoder", "memory_encoder", "prompt_encoder")


def speedup_hip(predictor, plugs=PLUGS, precision: str = "f16s", engine=None, lookahead: int = 8):
    """Install the HIP backend on a reference SAM2VideoPredictor / SAM2Base / SAM2ImagePredictor: re-points exactly the
    attributes the reference's own `set_runtime_backend` methods re-point and keeps executor objects in
    `backend_contexts`, so `predictor.speedup("torch")` / `release()` restore the PyTorch path as for any other backend.
    `precision`: "f16s" (default: masks within 1e-3 of the PyTorch path), "f16" (faster, ~2e-3), "f16x3".
    `lookahead`: frames the image plug of a VIDEO predictor encodes per engine call when the frame it is given is a view of the
    clip tensor (LookaheadImagePlug; 1 = off, the reference's one-frame-per-call behaviour)."""
    model = getattr(predictor, "model", predictor)        # SAM2ImagePredictor wraps the SAM2Base in .model
    unknown = set(plugs) - set(PLUGS)
    if unknown:
        raise ValueError(f"unknown plug(s) {sorted(unknown)}; choose from {PLUGS}")
    is_video = getattr(model, "memory_attention", None) is not None and not hasattr(predictor, "set_image_e2e")
    depth = max(1, int(lookahead)) if is_video else 1
    eng = _engine_for(model, precision, engine, max_batch=depth)
    if "image" in plugs:
        fn = LookaheadImagePlug(eng, depth) if depth > 1 else eng.image_encoder
        ex = HipExecutor(eng, fn, 1)
        if depth > 1:
            ex.Inference = lambda inputs, output_type="torch", _fn=fn: list(_fn(inputs[0]))      # the view must reach the plug as it is
            eng._lookahead_plug = fn                        # kept on the engine: the reference's objects get no attribute they do not have
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
    if eng is not None and getattr(eng, "_lookahead_plug", None) is not None:
        eng._lookahead_plug.clear()
        eng._lookahead_plug = None
    if eng is not None:
        eng.close()
        model._sam2mi_engine = None
