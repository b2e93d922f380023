"""SAM2VideoPredictor on the MI355X-native backend.

Host-side mirror of the reference predictor's interface and state machine
(/root/reference/sam2/sam2/sam2_video_predictor_official.py: init_state :147-205,
add_new_points_or_box :266-399, propagate_in_video_preflight :585-649, propagate_in_video :651-736,
_run_single_frame_inference :843-909) and of SAM2Base._prepare_memory_conditioned_features
(modeling/sam2_base_official.py:797-976, the memory / object-pointer selection).  All tensor work is
done by libsam2mi.so through sam2_opt_amd.native.Engine: frame features and the memory bank stay
resident in HBM, the host only passes slot indices - there is no per-frame host<->device sync.

Differences from the reference, all deliberate:
  * the image encoder runs on batches of upcoming frames (`encode_batch`), since it does not depend on
    the tracking state; the reference encodes one frame at a time;
  * `fill_hole_area` defaults to 0 (the goldens were recorded from the reference without its CUDA extension, where the
    step is silently skipped, utils/misc.py:321-336); pass 8 (build_sam.py:129) to fill holes on the device;
  * `clear_non_cond_mem_around_input`, `add_all_frames_to_correct_as_cond`, `max_cond_frames_in_attn` and
    `memory_temporal_stride_for_eval` are constructor options with the reference's defaults (False, False, -1, 1).
"""
from __future__ import annotations

import os

import functools
import threading
from collections import OrderedDict
from typing import Optional

import numpy as np
import torch

from .config import get_config
from .memory_select import select_memory
from .native import Engine, MemSelect, default_precision
from .synthetic import normalize_frames


def _locked(fn):
    """Serialise the host side of a predictor call: one predictor may be driven from several threads, each under its own
    torch.cuda.Stream and with its own inference state (/root/reference/video_multi_thread.py:36-87)."""
    @functools.wraps(fn)
    def wrapper(self, *a, **k):
        with self._lock:
            return fn(self, *a, **k)
    return wrapper


# A/B switch (measurements only): queue the encoder prefetch BEFORE the tracking of the frame that triggers it (the round-1 order)
_PREFETCH_EARLY = os.environ.get("SAM2MI_PREFETCH_EARLY") is not None

class SAM2VideoPredictor:
    def __init__(self, model: str = "large", state_dict=None, ckpt_path: Optional[str] = None, device=None,
                 encode_batch: int = 8, bank_slots: int = 384, fill_hole_area: int = 0, non_overlap_masks: bool = False,
                 overlap_encode: bool = True, precision: Optional[str] = None, clear_non_cond_mem_around_input: bool = False,
                 add_all_frames_to_correct_as_cond: bool = False, max_cond_frames_in_attn: int = -1,
                 memory_temporal_stride_for_eval: int = 1, prefetch_depth: int = 2):
        self.cfg = get_config(model)
        if state_dict is None and ckpt_path is not None:
            # same contract as build_sam._load_checkpoint (build_sam.py:164-174)
            state_dict = torch.load(ckpt_path, map_location="cpu", weights_only=True)["model"]
        if state_dict is None:
            raise ValueError("state_dict or ckpt_path is required")
        self.encode_batch = int(encode_batch)
        self.prefetch_depth = max(1, int(prefetch_depth))         # batches the encoder stream may be ahead of the tracking (feature cache = depth + 1 batches)
        self.engine = Engine(self.cfg, state_dict=state_dict, max_batch=self.encode_batch, bank_slots=bank_slots,
                             feat_slots=max((1 + self.prefetch_depth) * self.encode_batch, 4), device=device, precision=precision or default_precision(model))
        self.device = self.engine.device
        self.image_size = self.cfg["image_size"]
        self.num_maskmem = self.cfg["num_maskmem"]
        self.max_obj_ptrs_in_encoder = self.cfg["max_obj_ptrs_in_encoder"]
        self.fill_hole_area = fill_hole_area
        self.non_overlap_masks = non_overlap_masks
        # predictor / model options of the reference (sam2_video_predictor_official.py:24-40, sam2_base_official.py:39-41,:63)
        self.clear_non_cond_mem_around_input = clear_non_cond_mem_around_input
        self.add_all_frames_to_correct_as_cond = add_all_frames_to_correct_as_cond
        self.max_cond_frames_in_attn = max_cond_frames_in_attn
        self.memory_temporal_stride_for_eval = memory_temporal_stride_for_eval
        self.engine.set_fill_hole_area(fill_hole_area)       # 0 = off; > 0: holes filled on the device (postproc.hip)
        # The image encoder of the NEXT batch of frames runs on its own HIP stream beside the tracking of the current
        # batch: the tracking path is a chain of small latency-bound kernels (M = 4096 GEMMs, 8-token decoder ops) that
        # leaves most CUs idle, the encoder fills them.  The two paths share no workspace inside the engine.
        self.object_batch = 8                       # objects per batched tracking pass (1: loop objects like the reference)
        self.overlap_encode = bool(overlap_encode)
        import os as _os
        _prio = int(_os.environ.get("SAM2MI_ENC_PRIORITY", "0"))      # tuning: -1 = high-priority encoder stream
        _reserve = int(_os.environ.get("SAM2MI_ENC_RESERVE_CUS", "0"))   # CUs the encoder stream leaves to the tracking path
        if not self.overlap_encode:
            self._enc_stream = None
        elif _reserve > 0:
            self._enc_stream = self.engine.create_reserved_stream(_reserve)
        else:
            self._enc_stream = torch.cuda.Stream(device=self.device, priority=_prio)
        self.backend = "hip"
        self.debug_trace = None      # set to {} to record per-frame intermediates (parity tests)
        # Feature-cache and memory-bank slots belong to the engine, so the allocators live here and inference states
        # borrow from them: several states may be alive on one predictor (as in the reference, which keeps everything
        # in `inference_state`) without aliasing each other's slots.  `reset_state` / `release_state` return them.
        self._free_feat_slots = list(range(self.engine.feat_slots))
        self._free_bank_slots = list(range(self.engine.bank_slots))
        # frame features are recomputable, so the feature cache is ONE least-recently-encoded list over all live states:
        # (id(state), frame) -> state; a state that needs slots evicts the oldest entry, whichever state it belongs to
        self._feat_lru = OrderedDict()
        self._lock = threading.RLock()

    # ------------------------------------------------------------------ backend switch (reference: speedup :45-145)
    def speedup(self, backend: str = "hip", use_cache: bool = True, model_root_path=None):
        if backend not in ("hip", "mi355x", "sam2mi"):
            raise RuntimeError(f"Unknown backend={backend}: this predictor only runs the MI355X HIP backend "
                               "(attach sam2_opt_amd.plugin to the reference predictor to switch per plug)")
        self.backend = "hip"

    def release(self):
        self.engine.close()

    # ------------------------------------------------------------------ state
    @torch.inference_mode()
    @_locked
    def init_state(self, video_path=None, offload_video_to_cpu: bool = False, offload_state_to_cpu: bool = False,
                   async_loading_frames: bool = False, frames: Optional[torch.Tensor] = None, video_height: Optional[int] = None,
                   video_width: Optional[int] = None, frames_u8=None):
        """`video_path`, `offload_video_to_cpu`, `offload_state_to_cpu`, `async_loading_frames`: the reference's arguments, in its
        order (sam2_video_predictor_official.py:148-154).  `video_path`: a folder of JPEGs (needs PIL; decoded on the host like
        utils/misc.py:92-101, resized on the device, then the uint8 path).  Keyword-only extras of this build: `frames`: float32
        (T,3,1024,1024) already /255 and mean/std normalised (what load_video_frames returns), on CPU or GPU; or `frames_u8`:
        decoded uint8 (T,1024,1024,3) frames (numpy or torch) - normalised on the device inside the patch-embedding gather,
        bit-identical to the float path at a quarter of the bytes.
        `offload_state_to_cpu`: the per-frame outputs the state keeps (low-res mask logits, object scores) live in host memory,
        like the reference's `storage_device` (:180-183); the memory bank itself is a fixed pool of `bank_slots` entries in HBM
        that recycles on its own, so it is what bounds the device footprint of a long clip, not this option.
        `async_loading_frames`: JPEGs are decoded by a background thread (utils/misc.py:104-169); frame 0 is decoded before
        the call returns, a frame that tracking reaches before the thread is decoded on demand."""
        loader = None
        if async_loading_frames:
            if frames is not None or frames_u8 is not None:
                raise ValueError("async_loading_frames needs video_path (frames passed as tensors are loaded already)")
            loader = _AsyncJpegFrames(video_path, self, offload_video_to_cpu)
            frames = loader                                       # len() / shape[0]; frames reach the device batch by batch
            video_height, video_width = video_height or loader.video_height, video_width or loader.video_width
        elif frames is None and frames_u8 is None:
            frames_u8 = _load_jpeg_folder(video_path)             # decoded on the host (PIL), resized on the device below
        if frames is None:
            if isinstance(frames_u8, (list, tuple)):               # frames of any size, decoded (H,W,3) uint8 arrays
                video_height, video_width = video_height or frames_u8[0].shape[0], video_width or frames_u8[0].shape[1]
                frames_u8 = self._resize_frames(frames_u8)
            frames = torch.from_numpy(frames_u8) if isinstance(frames_u8, np.ndarray) else frames_u8
            if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] != 3:
                raise ValueError("frames_u8 must be uint8 (T,H,W,3)")
            if tuple(frames.shape[1:3]) != (self.image_size, self.image_size):
                video_height, video_width = video_height or frames.shape[1], video_width or frames.shape[2]
                frames = self._resize_frames(list(frames))
        if loader is None:
            frames = frames.to("cpu" if offload_video_to_cpu else self.device)
        st = {
            "images": frames, "num_frames": len(frames), "loader": loader, "offload_state_to_cpu": bool(offload_state_to_cpu),
            "storage_device": torch.device("cpu") if offload_state_to_cpu else self.device,
            "video_height": video_height or self.image_size, "video_width": video_width or self.image_size,
            "device": self.device, "offload_video_to_cpu": offload_video_to_cpu,
            "point_inputs_per_obj": {}, "mask_inputs_per_obj": {}, "obj_id_to_idx": OrderedDict(), "obj_idx_to_id": OrderedDict(), "obj_ids": [],
            "output_dict_per_obj": {}, "temp_output_dict_per_obj": {}, "frames_tracked_per_obj": {},
            "feat_slot_of_frame": OrderedDict(),
            "feat_events": {},          # frame -> event recorded on the encoder stream (features still in flight)
            "bank_slots_held": set(),   # memory-bank slots borrowed from the predictor
            "stream": torch.cuda.current_stream(self.device),       # the stream this state is driven on (updated by every call)
            "stale_outputs": OrderedDict(),   # id(out) -> non-cond output no later frame attends to: recycled when the bank is full
        }
        self._sync_encoder_stream()
        self._ensure_features(st, 0, forward=True)       # warm up the backbone like the reference (:204)
        return st

    def _stored(self, st, t):
        """A per-frame output as the state keeps it: on the host with offload_state_to_cpu (the reference's `storage_device`,
        sam2_video_predictor_official.py:180-183,:880-897), else where it is."""
        return t.to("cpu") if st["offload_state_to_cpu"] else t

    def _resize_frames(self, frames):
        """load_video_frames_from_jpg_images' PIL resize (utils/misc.py:92-101) on the device: bit-exact bicubic, frame by frame."""
        S = self.image_size
        out = torch.empty(len(frames), S, S, 3, dtype=torch.uint8, device=self.device)
        for i, f in enumerate(frames):
            f = torch.as_tensor(np.ascontiguousarray(f) if isinstance(f, np.ndarray) else f).to(self.device).contiguous()
            out[i] = self.engine.resize_u8_pil_bicubic(f, S)
        return out

    def _sync_encoder_stream(self):
        """All encoder launches share one workspace: work of the caller's stream must not start an encoder pass (or reuse
        feature slots) while a prefetch is still running on the encoder stream."""
        if self._enc_stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self._enc_stream)

    @_locked
    def reset_state(self, st):
        self._sync_encoder_stream()
        for d in list(st["output_dict_per_obj"].values()) + list(st["temp_output_dict_per_obj"].values()):
            for k in ("cond_frame_outputs", "non_cond_frame_outputs"):
                for out in d[k].values():
                    self._free_bank(st, out)
                d[k].clear()
        st["stale_outputs"].clear()
        for k in ("point_inputs_per_obj", "mask_inputs_per_obj", "obj_id_to_idx", "obj_idx_to_id", "output_dict_per_obj",
                  "temp_output_dict_per_obj", "frames_tracked_per_obj"):
            st[k].clear()
        st["obj_ids"] = []

    @_locked
    def release_state(self, st):
        """Return every slot the state borrowed (feature cache too); the state must not be used afterwards."""
        self.reset_state(st)
        self._free_bank_slots.extend(st["bank_slots_held"])       # anything not reachable through the output dicts
        st["bank_slots_held"].clear()
        for t, sl in st["feat_slot_of_frame"].items():
            self._feat_lru.pop((id(st), t), None)
            self._free_feat_slots.append(sl)
        st["feat_slot_of_frame"].clear()
        st["feat_events"].clear()

    def _obj_id_to_idx(self, st, obj_id):
        idx = st["obj_id_to_idx"].get(obj_id)
        if idx is not None:
            return idx
        idx = len(st["obj_id_to_idx"])
        st["obj_id_to_idx"][obj_id] = idx
        st["obj_idx_to_id"][idx] = obj_id
        st["obj_ids"] = list(st["obj_id_to_idx"])
        st["point_inputs_per_obj"][idx] = {}
        st["mask_inputs_per_obj"][idx] = {}
        st["output_dict_per_obj"][idx] = {"cond_frame_outputs": {}, "non_cond_frame_outputs": {}}
        st["temp_output_dict_per_obj"][idx] = {"cond_frame_outputs": {}, "non_cond_frame_outputs": {}}
        st["frames_tracked_per_obj"][idx] = {}
        return idx

    # ------------------------------------------------------------------ slots
    def _alloc_bank(self, st, protect=()) -> int:
        """A free memory-bank slot.  `protect`: slots the pending request reads (its MemSelect) - never recycled for it."""
        if not self._free_bank_slots:
            # bank full: recycle the oldest output no later frame of the pass that produced it attends to.  It keeps its
            # low-res mask; should a later request (reverse pass, re-interaction far behind the tracking head) select its
            # memory, _select_memory raises instead of tracking without it.
            for key in list(st["stale_outputs"]):
                out = st["stale_outputs"][key]
                if out.get("slot") is None:
                    del st["stale_outputs"][key]
                    continue
                if out["slot"] in protect:
                    continue
                del st["stale_outputs"][key]
                self._free_bank(st, out)
                out["has_mem"] = False
                out["recycled"] = True
                break
        if not self._free_bank_slots:
            raise RuntimeError("memory bank exhausted: raise bank_slots (or release_state() inference states no longer in use)")
        sl = self._free_bank_slots.pop()
        st["bank_slots_held"].add(sl)
        return sl

    def _free_bank(self, st, out):
        if out is not None and out.get("slot") is not None:
            st["bank_slots_held"].discard(out["slot"])
            self._free_bank_slots.append(out["slot"])
            out["slot"] = None

    def _encode_batch(self, st, start: int, forward: bool, side: bool, keep=None):
        """Encode up to `encode_batch` uncached frames from `start` in the tracking direction into feature-cache slots."""
        m = st["feat_slot_of_frame"]
        T = st["num_frames"]
        step = 1 if forward else -1
        idxs = [t for t in range(start, start + step * self.encode_batch, step) if 0 <= t < T and t not in m]
        if not idxs:
            return
        main = torch.cuda.current_stream(self.device)
        st["stream"] = main
        foreign = set()                                         # streams of OTHER states whose cached frames get evicted
        # frames the tracking head still needs: the one in use and the rest of its batch in the tracking direction
        head = start if keep is None else keep
        in_use = {(id(st), t) for t in range(head, head + step * self.encode_batch, step)} | {(id(st), t) for t in idxs}
        while len(self._free_feat_slots) < len(idxs):
            # evict (a) this state's cached frame farthest BEHIND the head in the tracking direction (a pass never returns to it;
            # after a direction change these are the frames of the old pass's far end, not the ones about to be tracked),
            # else (b) the least recently USED frame of any state that is not in use (hits refresh the order, _ensure_features)
            behind = [t for t in m if (t - head) * step < 0 and (id(st), t) not in in_use]
            if behind:
                key_old = (id(st), max(behind, key=lambda t: abs(t - head)))
            else:
                key_old = next((k for k in self._feat_lru if k not in in_use), None)
                if key_old is None:
                    raise RuntimeError("feature cache too small for encode_batch (every cached frame is in use)")
            st_old = self._feat_lru.pop(key_old)
            t_old = key_old[1]
            self._free_feat_slots.append(st_old["feat_slot_of_frame"].pop(t_old))
            st_old["feat_events"].pop(t_old, None)
            if st_old is not st and st_old["stream"] != main:
                foreign.add(st_old["stream"])
        slots = [self._free_feat_slots.pop() for _ in idxs]

        def run():
            # consecutive frames (the usual case) are a view of the clip: no gather, and no host-blocking upload of an index tensor
            # (which, queued behind the previous batch on the encoder stream, stalled the launching thread for a whole batch)
            lo, hi = min(idxs), max(idxs)
            if st.get("loader") is not None:                      # async_loading_frames: decoded (on demand) + resized on this stream
                imgs = st["loader"].get(idxs)
            elif hi - lo + 1 == len(idxs):
                imgs = st["images"][lo:hi + 1] if forward else st["images"][lo:hi + 1].flip(0)
            else:
                imgs = st["images"][idxs]
            if imgs.dtype == torch.uint8:                         # decoded HWC frames: normalised inside the engine
                self.engine.video_encode_u8(imgs.to(self.device).contiguous(), slots)
            else:
                self.engine.video_encode(imgs.to(self.device, dtype=torch.float32).contiguous(), slots)
        if side:
            self._enc_stream.wait_stream(main)                   # the evicted slots were read by work already queued on `main`
            for so in foreign:                                   # ... or on the stream of the state they were taken from
                self._enc_stream.wait_stream(so)
            with torch.cuda.stream(self._enc_stream):
                run()
                ev = torch.cuda.Event()
                ev.record(self._enc_stream)
            for t in idxs:
                st["feat_events"][t] = ev
        else:
            self._sync_encoder_stream()          # a prefetch in flight uses the same encoder workspace
            for so in foreign:
                main.wait_stream(so)
            run()
        for t, sl in zip(idxs, slots):
            m[t] = sl
            self._feat_lru[(id(st), t)] = st

    def _ensure_features(self, st, frame_idx: int, forward: bool = True, prefetch: bool = True) -> int:
        """Feature-cache slot of `frame_idx`; on a miss encode a batch of frames starting there
        (cf. _get_image_feature :810-841, which caches exactly one frame).  With `overlap_encode` the batch AFTER the one
        in use is encoded ahead of time on the encoder stream."""
        m = st["feat_slot_of_frame"]
        if frame_idx not in m:
            self._encode_batch(st, frame_idx, forward, side=False)
        else:
            self._feat_lru.move_to_end((id(st), frame_idx))      # a hit refreshes the frame: eviction is least recently USED
        ev = st["feat_events"].pop(frame_idx, None)
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
        if prefetch:
            self._prefetch_features(st, frame_idx, forward)
        return m[frame_idx]

    def _prefetch_features(self, st, frame_idx: int, forward: bool = True):
        """With `overlap_encode`: queue the encoder pass of the batch after the one in use on the encoder stream.  propagate_in_video
        calls this AFTER it has queued the tracking of `frame_idx`, so that the tracking stream is not left empty while the host
        spends a millisecond enqueuing the ~350 launches of an encoder pass."""
        m = st["feat_slot_of_frame"]
        if self.overlap_encode:
            step = 1 if forward else -1
            t = frame_idx + step
            ahead = self.encode_batch * self.prefetch_depth         # how far the encoder stream may run in front of the tracking
            while t in m and abs(t - frame_idx) <= ahead:
                t += step
            if 0 <= t < st["num_frames"] and abs(t - frame_idx) <= ahead:
                self._encode_batch(st, t, forward, side=True, keep=frame_idx)

    # ------------------------------------------------------------------ prompts
    @torch.inference_mode()
    @_locked
    def add_new_points_or_box(self, inference_state, frame_idx, obj_id, points=None, labels=None, clear_old_points=True,
                              normalize_coords=True, box=None):
        st = inference_state
        obj_idx = self._obj_id_to_idx(st, obj_id)
        if (points is not None) != (labels is not None):
            raise ValueError("points and labels must be provided together")
        if points is None and box is None:
            raise ValueError("at least one of points or box must be provided as input")
        pts = np.zeros((0, 2), np.float32) if points is None else np.asarray(points, np.float32).reshape(-1, 2)
        lab = np.zeros((0,), np.int32) if labels is None else np.asarray(labels, np.int32).reshape(-1)
        if box is not None:
            if not clear_old_points:
                raise ValueError("cannot add box without clearing old points, since box prompt must be provided "
                                 "before any point prompt (please use clear_old_points=True instead)")
            pts = np.concatenate([np.asarray(box, np.float32).reshape(2, 2), pts], axis=0)
            lab = np.concatenate([np.array([2, 3], np.int32), lab], axis=0)
        if normalize_coords:
            pts = pts / np.array([st["video_width"], st["video_height"]], np.float32)
        pts = pts * self.image_size
        per_frame = st["point_inputs_per_obj"][obj_idx]
        if not clear_old_points and frame_idx in per_frame:
            pts = np.concatenate([per_frame[frame_idx][0], pts], axis=0)
            lab = np.concatenate([per_frame[frame_idx][1], lab], axis=0)
        per_frame[frame_idx] = (pts, lab)
        st["mask_inputs_per_obj"][obj_idx].pop(frame_idx, None)
        # an untracked frame is an initial conditioning frame (SAM on the frame alone); a tracked one gets correction clicks on
        # top of its memory-conditioned features (sam2_video_predictor_official.py:333-346)
        tracked = st["frames_tracked_per_obj"][obj_idx].get(frame_idx)
        is_init = tracked is None
        # an output is a conditioning one if the frame is an initial conditioning frame or every corrected frame counts (:348-350)
        key = "cond_frame_outputs" if (is_init or self.add_all_frames_to_correct_as_cond) else "non_cond_frame_outputs"
        od, td = st["output_dict_per_obj"][obj_idx], st["temp_output_dict_per_obj"][obj_idx]
        # the previous prediction on this frame, if any, goes in as a mask prompt (clamped; :352-366)
        prev = td[key].get(frame_idx) or od["cond_frame_outputs"].get(frame_idx) or od["non_cond_frame_outputs"].get(frame_idx)
        prev_logits = torch.clamp(prev["pred_masks"].to(self.device), -32.0, 32.0).contiguous() if prev is not None else None
        feat = self._ensure_features(st, frame_idx)
        sel = None if is_init else self._select_memory(od, frame_idx, st["num_frames"], tracked["reverse"])    # may raise: before any slot is taken
        slot = self._alloc_bank(st, protect=() if sel is None else _sel_slots(sel))
        n = len(lab)
        # _use_multimask (sam2_base_official.py:1181-1189): only on initial conditioning frames (or while tracking), 1 point
        multimask = self.cfg["multimask_min_pt_num"] <= n <= self.cfg["multimask_max_pt_num"]
        low = self.engine.new(1, 1, 256, 256)
        score = self.engine.new(1, 1)
        outs = dict(low_res_masks=low, object_score_logits=score)
        if is_init:
            self.engine.video_click(feat, pts, lab, multimask, slot, outs, mask_logits=prev_logits)
        else:
            self.engine.video_track(feat, sel, slot, False, outs, points=pts, labels=lab, multimask=multimask, mask_logits=prev_logits)
        self._free_bank(st, td[key].pop(frame_idx, None))
        td[key][frame_idx] = dict(slot=slot, pred_masks=self._stored(st, low), object_score_logits=self._stored(st, score), has_mem=False, is_pts=True)
        return frame_idx, st["obj_ids"], self._video_res(st, self._consolidated(st, frame_idx))

    def add_new_points(self, *a, **k):
        return self.add_new_points_or_box(*a, **k)

    @torch.inference_mode()
    @_locked
    def add_new_mask(self, inference_state, frame_idx, obj_id, mask):
        """Mask prompt (sam2_video_predictor_official.py:403-489): the mask becomes the frame's output as it is
        (SAM2Base._use_mask_as_output); the SAM decoder only supplies the object pointer."""
        st = inference_state
        obj_idx = self._obj_id_to_idx(st, obj_id)
        m = torch.as_tensor(np.asarray(mask) if not isinstance(mask, torch.Tensor) else mask)
        if m.dim() != 2:
            raise ValueError("mask must be a 2-D array")
        m = m.to(self.device).float()[None, None]
        S = self.image_size
        if m.shape[-2:] != (S, S):              # host-side plumbing, like the reference: bilinear antialias, then >= 0.5
            m = (torch.nn.functional.interpolate(m, size=(S, S), align_corners=False, mode="bilinear", antialias=True) >= 0.5).float()
        m = m.contiguous()
        st["mask_inputs_per_obj"][obj_idx][frame_idx] = m
        st["point_inputs_per_obj"][obj_idx].pop(frame_idx, None)
        is_init = frame_idx not in st["frames_tracked_per_obj"][obj_idx]
        key = "cond_frame_outputs" if (is_init or self.add_all_frames_to_correct_as_cond) else "non_cond_frame_outputs"
        td = st["temp_output_dict_per_obj"][obj_idx]
        feat = self._ensure_features(st, frame_idx)
        slot = self._alloc_bank(st)
        low = self.engine.new(1, 1, 256, 256)
        score = self.engine.new(1, 1)
        self.engine.video_mask(feat, m, slot, dict(low_res_masks=low, object_score_logits=score))
        self._free_bank(st, td[key].pop(frame_idx, None))
        td[key][frame_idx] = dict(slot=slot, pred_masks=self._stored(st, low), object_score_logits=self._stored(st, score), has_mem=False, is_pts=False)
        return frame_idx, st["obj_ids"], self._video_res(st, self._consolidated(st, frame_idx))

    @torch.inference_mode()
    @_locked
    def clear_all_prompts_in_frame(self, inference_state, frame_idx, obj_id, need_output=True):
        """Remove all input points / mask of an object on a frame (sam2_video_predictor_official.py:739-779): the frame's
        conditioning output, if any, is downgraded to a non-conditioning one."""
        st = inference_state
        obj_idx = self._obj_id_to_idx(st, obj_id)
        st["point_inputs_per_obj"][obj_idx].pop(frame_idx, None)
        st["mask_inputs_per_obj"][obj_idx].pop(frame_idx, None)
        td, od = st["temp_output_dict_per_obj"][obj_idx], st["output_dict_per_obj"][obj_idx]
        for key in ("cond_frame_outputs", "non_cond_frame_outputs"):
            self._free_bank(st, td[key].pop(frame_idx, None))
        out = od["cond_frame_outputs"].pop(frame_idx, None)
        if out is not None:
            self._free_bank(st, od["non_cond_frame_outputs"].get(frame_idx))
            od["non_cond_frame_outputs"][frame_idx] = out
            st["frames_tracked_per_obj"][obj_idx].pop(frame_idx, None)
        if not need_output:
            return None
        return frame_idx, st["obj_ids"], self._video_res(st, self._consolidated(st, frame_idx))

    @torch.inference_mode()
    @_locked
    def remove_object(self, inference_state, obj_id, strict=False, need_output=True):
        """Remove an object id from the tracking state (sam2_video_predictor_official.py:973-1060)."""
        st = inference_state
        old_idx = st["obj_id_to_idx"].get(obj_id)
        updated = []
        if old_idx is None:
            if not strict:
                return st["obj_ids"], updated
            raise RuntimeError(f"Cannot remove object id {obj_id} as it doesn't exist. All existing object ids: {st['obj_ids']}.")
        if len(st["obj_id_to_idx"]) == 1:
            self.reset_state(st)
            return st["obj_ids"], updated
        input_frames = set(st["point_inputs_per_obj"][old_idx]) | set(st["mask_inputs_per_obj"][old_idx])
        for t in input_frames:
            self.clear_all_prompts_in_frame(st, t, obj_id, need_output=False)
        # free the object's memory-bank slots, then re-index the per-object containers
        for d in (st["output_dict_per_obj"][old_idx], st["temp_output_dict_per_obj"][old_idx]):
            for key in ("cond_frame_outputs", "non_cond_frame_outputs"):
                for out in d[key].values():
                    self._free_bank(st, out)
        old_ids = list(st["obj_ids"])
        remain = [i for i in range(len(old_ids)) if i != old_idx]
        new_ids = [old_ids[i] for i in remain]
        remap = {o: n for n, o in enumerate(remain)}
        st["obj_id_to_idx"] = OrderedDict((oid, i) for i, oid in enumerate(new_ids))
        st["obj_idx_to_id"] = OrderedDict((i, oid) for i, oid in enumerate(new_ids))
        st["obj_ids"] = new_ids
        for name in ("point_inputs_per_obj", "mask_inputs_per_obj", "output_dict_per_obj", "temp_output_dict_per_obj", "frames_tracked_per_obj"):
            c = st[name]
            st[name] = type(c)((remap[k], v) for k, v in c.items() if k in remap)
        if need_output:
            for t in sorted(input_frames):
                updated.append((t, self._video_res(st, self._consolidated(st, t))))
        return st["obj_ids"], updated

    def _consolidated(self, st, frame_idx):
        """(num_obj,1,256,256) low-res logits on `frame_idx`; objects without output get NO_OBJ_SCORE (:525-570)."""
        outs = []
        for obj_idx in range(len(st["obj_ids"])):
            out = None
            for d in (st["temp_output_dict_per_obj"][obj_idx], st["output_dict_per_obj"][obj_idx]):
                for k in ("cond_frame_outputs", "non_cond_frame_outputs"):
                    out = out or d[k].get(frame_idx)
            outs.append(out["pred_masks"].to(self.device) if out is not None else torch.full((1, 1, 256, 256), -1024.0, device=self.device))
        return torch.cat(outs, dim=0) if len(outs) > 1 else outs[0]

    def _video_res(self, st, low):
        """_get_orig_video_res_output (sam2_video_predictor_official.py:489-509)."""
        H, W = st["video_height"], st["video_width"]
        vid = low if low.shape[-2:] == (H, W) else self.engine.resize_bilinear(low, (H, W))
        if self.non_overlap_masks and vid.shape[0] > 1:
            # SAM2Base._apply_non_overlapping_constraints (sam2_base_official.py:1191-1209): keep the best object per pixel
            keep = torch.argmax(vid, dim=0, keepdim=True) == torch.arange(vid.shape[0], device=vid.device)[:, None, None, None]
            vid = torch.where(keep, vid, torch.clamp(vid, max=-10.0))
        return vid

    # ------------------------------------------------------------------ propagation
    @torch.inference_mode()
    @_locked
    def propagate_in_video_preflight(self, inference_state):
        st = inference_state
        if len(st["obj_ids"]) == 0:
            raise RuntimeError("No input points or masks are provided for any object; please add inputs first.")
        for obj_idx in range(len(st["obj_ids"])):
            od, td = st["output_dict_per_obj"][obj_idx], st["temp_output_dict_per_obj"][obj_idx]
            for key in ("non_cond_frame_outputs", "cond_frame_outputs"):
                for t, out in td[key].items():
                    if not out["has_mem"]:
                        # memory encoder on the interacted frame, binarised mask (:610-627, sam2_base :1000-1010)
                        self.engine.video_encode_memory(self._ensure_features(st, t), out["slot"], True)
                        out["has_mem"] = True
                    self._free_bank(st, od[key].get(t))
                    od[key][t] = out
                    if self.clear_non_cond_mem_around_input:
                        self._clear_obj_non_cond_mem_around_input(st, t, obj_idx)
                td[key].clear()
            if len(od["cond_frame_outputs"]) == 0:
                raise RuntimeError(f"No input points or masks are provided for object id {st['obj_idx_to_id'][obj_idx]}; "
                                   "please add inputs first.")
            for t in od["cond_frame_outputs"]:
                self._free_bank(st, od["non_cond_frame_outputs"].pop(t, None))

    def _clear_obj_non_cond_mem_around_input(self, st, frame_idx: int, obj_idx: int):
        """Drop one object's non-conditioning memories within memory_temporal_stride * num_maskmem frames of an interacted
        frame.  The reference calls a method of this name (sam2_video_predictor_official.py:632,:704) but only defines the
        all-object `_clear_non_cond_mem_around_input` (:1062-1080), so its own option raises AttributeError; this is that
        method's body for one object (what upstream SAM 2 does)."""
        r = self.memory_temporal_stride_for_eval
        nc = st["output_dict_per_obj"][obj_idx]["non_cond_frame_outputs"]
        for t in range(frame_idx - r * self.num_maskmem, frame_idx + r * self.num_maskmem + 1):
            self._free_bank(st, nc.pop(t, None))

    def _select_memory(self, od, frame_idx: int, num_frames: int, reverse: bool) -> MemSelect:
        """SAM2Base._prepare_memory_conditioned_features step 1 (sam2_base_official.py:823-946): the selection itself is
        sam2_opt_amd.memory_select.select_memory; here its outputs become bank-slot lists for the device."""
        mems, ptrs, max_ptrs = select_memory(od["cond_frame_outputs"], od["non_cond_frame_outputs"], frame_idx, num_frames, reverse,
                                             self.num_maskmem, self.max_obj_ptrs_in_encoder, self.max_cond_frames_in_attn,
                                             self.memory_temporal_stride_for_eval)
        sel = MemSelect()
        n = 0
        for t_pos, out in mems:
            if out.get("recycled"):
                raise RuntimeError("a memory frame this request attends to was recycled because the memory bank was full: "
                                   "construct the predictor with a larger bank_slots")
            if not out["has_mem"]:
                continue
            if n >= 8:
                raise NotImplementedError("more than 8 spatial memories: lower max_cond_frames_in_attn (the reference's TensorRT "
                                          "engines take at most 7, sam2_video_predictor_official.py:116-138)")
            sel.mem_slot[n] = out["slot"]
            sel.mem_tpos[n] = self.num_maskmem - t_pos - 1
            n += 1
        sel.num_mem = n
        if len(ptrs) > 32:
            raise NotImplementedError("more than 32 object pointers")
        for i, (dt, out) in enumerate(ptrs):
            if out.get("recycled") or out["slot"] is None:
                raise RuntimeError("an object pointer this request attends to was recycled because the memory bank was full: "
                                   "construct the predictor with a larger bank_slots")
            sel.ptr_slot[i] = out["slot"]
            sel.ptr_dt[i] = float(dt)
        sel.num_ptr = len(ptrs)
        sel.ptr_tmax = float(max_ptrs - 1)
        return sel

    def _release_stale(self, st, obj_idx: int, od, frame_idx: int, reverse: bool):
        """Non-conditioning outputs that no later frame of THIS pass can attend to become candidates for recycling.  Like the
        reference they keep their memory (a reverse pass or a correction click far behind the tracking head needs it);
        `_alloc_bank` takes their slots back, oldest first, only once the bank is full."""
        # reach of select_memory: object pointers go back max_obj_ptrs_in_encoder frames, spatial memories (num_maskmem - 2) * r + 1
        # frames with memory_temporal_stride_for_eval = r (sam2_base_official.py:851-868: ((t - 2) // r) * r - (t_rel - 2) * r)
        r = max(1, self.memory_temporal_stride_for_eval)
        horizon = max(self.max_obj_ptrs_in_encoder, (self.num_maskmem - 2) * r + 1) + 1
        t = frame_idx + horizon + 1 if reverse else frame_idx - horizon - 1
        out = od["non_cond_frame_outputs"].get(t)
        if out is not None and out.get("slot") is not None:
            st["stale_outputs"][id(out)] = out

    @torch.inference_mode()
    def propagate_in_video(self, inference_state, start_frame_idx=None, max_frame_num_to_track=None, reverse=False):
        st = inference_state
        self.propagate_in_video_preflight(st)
        num_frames = st["num_frames"]
        if start_frame_idx is None:
            start_frame_idx = min(t for od in st["output_dict_per_obj"].values() for t in od["cond_frame_outputs"])
        if max_frame_num_to_track is None:
            max_frame_num_to_track = num_frames
        if reverse:
            end = max(start_frame_idx - max_frame_num_to_track, 0)
            order = range(start_frame_idx, end - 1, -1) if start_frame_idx > 0 else []
        else:
            end = min(start_frame_idx + max_frame_num_to_track, num_frames - 1)
            order = range(start_frame_idx, end + 1)
        for frame_idx in order:
            with self._lock:                                   # host side of one frame; never held across the yield
                st["stream"] = torch.cuda.current_stream(self.device)
                per_obj = [None] * len(st["obj_ids"])
                todo = []                                          # objects to track on this frame (no stored conditioning output)
                for obj_idx in range(len(st["obj_ids"])):
                    od = st["output_dict_per_obj"][obj_idx]
                    if frame_idx in od["cond_frame_outputs"]:
                        per_obj[obj_idx] = od["cond_frame_outputs"][frame_idx]["pred_masks"].to(self.device)
                        if self.clear_non_cond_mem_around_input:
                            self._clear_obj_non_cond_mem_around_input(st, frame_idx, obj_idx)
                    else:
                        todo.append(obj_idx)
                if todo:
                    feat = self._ensure_features(st, frame_idx, forward=not reverse, prefetch=_PREFETCH_EARLY)
                # the reference loops objects with B = 1 (:691-725); here up to 8 objects go through one batched pass
                for c0 in range(0, len(todo), self.object_batch):
                    chunk = todo[c0:c0 + self.object_batch]
                    sels, slots, outs_l = [], [], []
                    for obj_idx in chunk:
                        od = st["output_dict_per_obj"][obj_idx]
                        sels.append(self._select_memory(od, frame_idx, num_frames, reverse))
                        self._free_bank(st, od["non_cond_frame_outputs"].pop(frame_idx, None))
                        slots.append(self._alloc_bank(st, protect={sl for s_ in sels for sl in _sel_slots(s_)}))
                        outs = dict(low_res_masks=self.engine.new(1, 1, 256, 256), object_score_logits=self.engine.new(1, 1))
                        if self.debug_trace is not None:
                            outs.update(pix_feat=self.engine.new(4096, 1, 256), ious=self.engine.new(1, 3), obj_ptr=self.engine.new(1, 256),
                                        low_res_multimasks=self.engine.new(1, 3, 256, 256),
                                        best_idx=self.engine.new(1, dtype=torch.int32))
                            self.debug_trace[(obj_idx, frame_idx)] = dict(outs, L=sels[-1].num_mem, P=4 * sels[-1].num_ptr)
                        outs_l.append(outs)
                    if len(chunk) == 1:
                        self.engine.video_track(feat, sels[0], slots[0], True, outs_l[0])
                    else:
                        self.engine.video_track_batch(feat, sels, slots, True, outs_l)
                    for obj_idx, slot, outs in zip(chunk, slots, outs_l):
                        od = st["output_dict_per_obj"][obj_idx]
                        od["non_cond_frame_outputs"][frame_idx] = dict(slot=slot, pred_masks=self._stored(st, outs["low_res_masks"]),
                                                                       object_score_logits=self._stored(st, outs["object_score_logits"]),
                                                                       has_mem=True, is_pts=False)
                        self._release_stale(st, obj_idx, od, frame_idx, reverse)
                        per_obj[obj_idx] = outs["low_res_masks"]
                if todo and not _PREFETCH_EARLY:
                    self._prefetch_features(st, frame_idx, forward=not reverse)      # after this frame's tracking has been queued
                for obj_idx in range(len(st["obj_ids"])):
                    st["frames_tracked_per_obj"][obj_idx][frame_idx] = {"reverse": reverse}
                low_all = torch.cat(per_obj, dim=0) if len(per_obj) > 1 else per_obj[0]
                result = (frame_idx, st["obj_ids"], self._video_res(st, low_all))
            yield result


def _sel_slots(sel: MemSelect):
    return {sel.mem_slot[i] for i in range(sel.num_mem)} | {sel.ptr_slot[i] for i in range(sel.num_ptr)}


def _jpeg_paths(path):
    """Frame files of a JPEG folder in the reference's order (utils/misc.py:246-255): <frame index>.jpg, sorted by that index."""
    if not isinstance(path, str) or not os.path.isdir(path):
        raise NotImplementedError("Only JPEG frames are supported at this moment (pass a folder of <frame index>.jpg files)")
    names = sorted([p for p in os.listdir(path) if os.path.splitext(p)[-1].lower() in (".jpg", ".jpeg")],
                   key=lambda p: int(os.path.splitext(p)[0]))
    if not names:
        raise RuntimeError(f"no images found in {path}")
    return [os.path.join(path, n) for n in names]


def _decode_jpeg(path):
    from PIL import Image
    return np.array(Image.open(path).convert("RGB"))


def _load_jpeg_folder(path):
    """The decode half of load_video_frames_from_jpg_images (utils/misc.py:213-277): decoded to RGB uint8 by PIL.  Returns a
    list of (H,W,3) arrays at the video's own resolution."""
    return [_decode_jpeg(p) for p in _jpeg_paths(path)]


class _AsyncJpegFrames:
    """`async_loading_frames=True`: the reference's AsyncVideoFrameLoader (utils/misc.py:104-169).  Frame 0 is decoded in the
    constructor (it fixes the video size and is where the first click usually lands), a daemon thread decodes the others in
    order; a frame requested before the thread reached it is decoded by the caller, and an exception of the thread is re-raised
    on the next request.  The thread only DECODES (host work); the bit-exact bicubic resize runs on the device, on the stream of
    the encoder pass that asks for the frames (`get`), so no GPU work is ever issued from the background thread."""

    def __init__(self, video_path, predictor, offload_video_to_cpu):
        self.paths = _jpeg_paths(video_path)
        self.predictor = predictor
        self.offload = bool(offload_video_to_cpu)
        self.host = [None] * len(self.paths)            # decoded (H,W,3) uint8 arrays at the video's resolution
        self.resized = {}                               # frame -> (S,S,3) uint8 tensor (device, or host with offload_video_to_cpu)
        self.exception = None
        first = self._decoded(0)
        self.video_height, self.video_width = first.shape[0], first.shape[1]

        def _load_frames():
            try:
                for n in range(len(self.paths)):
                    self._decoded(n)
            except Exception as e:                     # surfaced by the next get()
                self.exception = e
        self.thread = threading.Thread(target=_load_frames, daemon=True)
        self.thread.start()

    def __len__(self):
        return len(self.paths)

    def _decoded(self, i):
        img = self.host[i]
        if img is None:
            img = self.host[i] = _decode_jpeg(self.paths[i])      # a race with the thread decodes a frame twice: harmless, as in the reference
        return img

    def get(self, idxs):
        """uint8 (n,S,S,3) device tensor of the frames `idxs`, in that order."""
        if self.exception is not None:
            raise RuntimeError("Failure in frame loading thread") from self.exception
        out = []
        for t in idxs:
            r = self.resized.get(t)
            if r is None:
                r = self.predictor._resize_frames([self._decoded(t)])[0]
                self.resized[t] = r.cpu() if self.offload else r
            out.append(r.to(self.predictor.device))
        return torch.stack(out)
