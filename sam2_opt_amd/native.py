"""ctypes binding of libsam2mi.so (include/sam2mi.h).  There is NO fallback: if the HIP library is
missing or no MI355X is visible, construction raises."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np
import torch

from .config import get_config

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsam2mi.so")

EXPORTS = [
    "sam2mi_abi_version", "sam2mi_create", "sam2mi_destroy", "sam2mi_last_error", "sam2mi_load_weight",
    "sam2mi_finalize_weights", "sam2mi_image_encoder", "sam2mi_set_image_e2e", "sam2mi_memory_attention",
    "sam2mi_mask_decoder", "sam2mi_memory_encoder", "sam2mi_prompt_encoder", "sam2mi_prompt_encoder_ex", "sam2mi_dense_pe", "sam2mi_video_encode", "sam2mi_video_encode_u8", "sam2mi_fill_holes", "sam2mi_set_fill_hole_area",
    "sam2mi_video_click", "sam2mi_video_mask", "sam2mi_image_predict", "sam2mi_image_predict_ex", "sam2mi_video_encode_memory", "sam2mi_video_track", "sam2mi_video_track_batch", "sam2mi_resize_bilinear",
    "sam2mi_resize_u8_pil_bicubic", "sam2mi_resize_image_aa_bilinear", "sam2mi_stream_create_reserved", "sam2mi_stream_destroy", "sam2mi_profile_enable", "sam2mi_profile_read", "sam2mi_profile_read_mlp", "sam2mi_profile_read_xs", "sam2mi_profile_read_ks", "sam2mi_profile_read_kernels", "sam2mi_debug_gemm", "sam2mi_debug_hiera_attention",
    "sam2mi_debug_flash256", "sam2mi_debug_rowln", "sam2mi_debug_projln", "sam2mi_debug_hiera_block", "sam2mi_debug_read", "sam2mi_debug_gemm_bench", "sam2mi_debug_flash_bench", "sam2mi_debug_mlp",
]


class Sam2miConfig(C.Structure):
    _fields_ = [("embed_dim", C.c_int), ("num_heads", C.c_int), ("stages", C.c_int * 4),
                ("global_att_blocks", C.c_int * 8), ("window_spec", C.c_int * 4), ("image_size", C.c_int),
                ("max_batch", C.c_int), ("bank_slots", C.c_int), ("feat_slots", C.c_int), ("precision", C.c_int)]


ABI_VERSION = 2
PRECISIONS = {"f16": 0, "f16x3": 1, "f16s": 2}      # SAM2MI_PRECISION_* (include/sam2mi.h)


def default_precision(model) -> str:
    """What the predictors use when no precision is given: the fastest mode inside the north-star bar (masks within 1e-3 of the
    reference's fp32 path) where it exists - "f16s" for hiera-large; the padded-window sizes only have "f16"."""
    name = model if isinstance(model, str) else model.get("name", "")
    return "f16s" if name in ("large", "sam2.1_hiera_large") else "f16"


class MemSelect(C.Structure):
    _fields_ = [("num_mem", C.c_int), ("mem_slot", C.c_int * 16), ("mem_tpos", C.c_int * 16), ("num_ptr", C.c_int),
                ("ptr_slot", C.c_int * 32), ("ptr_dt", C.c_float * 32), ("ptr_tmax", C.c_float)]


class Prompt(C.Structure):
    _fields_ = [("coords", C.c_void_p), ("labels", C.c_void_p), ("num_points", C.c_int32), ("multimask", C.c_int32),
                ("mask_logits", C.c_void_p)]


class FrameOut(C.Structure):
    _fields_ = [("low_res_masks", C.c_void_p), ("low_res_multimasks", C.c_void_p), ("ious", C.c_void_p),
                ("obj_ptr", C.c_void_p), ("object_score_logits", C.c_void_p), ("pix_feat", C.c_void_p),
                ("best_idx", C.c_void_p)]


_lib = None


def load_library() -> C.CDLL:
    """dlopen libsam2mi.so; raises with build instructions when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: build it with `python -m sam2_opt_amd.build` "
                               "(hipcc --offload-arch=gfx950). This backend has no CPU/PyTorch fallback.")
        lib = C.CDLL(LIB_PATH)
        for name in EXPORTS:
            if not hasattr(lib, name):
                raise RuntimeError(f"libsam2mi.so does not export {name}")
        if lib.sam2mi_abi_version() != ABI_VERSION:
            raise RuntimeError(f"libsam2mi.so has ABI version {lib.sam2mi_abi_version()}, this binding needs {ABI_VERSION}: "
                               "rebuild with `python -m sam2_opt_amd.build`")
        lib.sam2mi_last_error.restype = C.c_char_p
        lib.sam2mi_last_error.argtypes = [C.c_void_p]
        lib.sam2mi_create.argtypes = [C.POINTER(Sam2miConfig), C.POINTER(C.c_void_p)]
        lib.sam2mi_destroy.argtypes = [C.c_void_p]
        lib.sam2mi_destroy.restype = None
        _lib = lib
    return _lib


def _ptr(t: Optional[torch.Tensor]):
    if t is None:
        return C.c_void_p(0)
    return C.c_void_p(t.data_ptr())


def _chk_f32(*ts):
    for t in ts:
        if t is not None:
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), (t.device, t.dtype, t.is_contiguous())


class Engine:
    """One sam2mi context (one per host thread / stream)."""

    def __init__(self, model: str = "large", state_dict=None, max_batch: int = 1, bank_slots: int = 64,
                 feat_slots: int = 16, device: Optional[torch.device] = None, precision: str = "f16"):
        """`precision` (include/sam2mi.h): "f16" (plain f16 MFMA operands, f32 accumulation: the bf16-class tier, ~2e-3 of the
        reference's fp32 path), "f16s" (selective 2-term f16 split of the operands whose rounding carries the error: masks within 1e-3 at
        0.87x the f16 rate; what bench.py and plugin.speedup_hip use by default), "f16x3" (every operand split, three MFMAs per product:
        ~1e-5 at 0.36x).  The split modes exist for hiera-large only."""
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(PRECISIONS)}")
        self.precision = precision
        if not torch.cuda.is_available():
            raise RuntimeError("sam2_opt_amd needs a ROCm GPU (MI355X); no CPU fallback exists")
        self.lib = load_library()
        self.cfg = get_config(model) if isinstance(model, str) else model
        dev = torch.device(device if device is not None else "cuda")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        torch.cuda.set_device(self.device)
        c = Sam2miConfig()
        c.embed_dim, c.num_heads, c.image_size = self.cfg["embed_dim"], self.cfg["num_heads"], self.cfg["image_size"]
        for i in range(4):
            c.stages[i] = self.cfg["stages"][i]
            c.window_spec[i] = self.cfg["window_spec"][i]
        for i in range(8):
            c.global_att_blocks[i] = self.cfg["global_att_blocks"][i] if i < len(self.cfg["global_att_blocks"]) else -1
        c.max_batch, c.bank_slots, c.feat_slots = max_batch, bank_slots, max(feat_slots, max_batch)
        c.precision = PRECISIONS[precision]
        self.max_batch, self.bank_slots, self.feat_slots = max_batch, bank_slots, c.feat_slots
        h = C.c_void_p()
        if self.lib.sam2mi_create(C.byref(c), C.byref(h)) != 0:
            raise RuntimeError("sam2mi_create failed: " + self.lib.sam2mi_last_error(None).decode())
        self.h = h
        if state_dict is not None:
            self.load_state_dict(state_dict)

    # ------------------------------------------------------------------ infrastructure
    def _check(self, rc: int, what: str):
        if rc != 0:
            raise RuntimeError(f"{what} failed: {self.lib.sam2mi_last_error(self.h).decode()}")

    @property
    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if getattr(self, "h", None):
            self.lib.sam2mi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_state_dict(self, sd):
        """Strict load of a SAM 2.1 `state_dict` (checkpoint["model"]); cf. build_sam._load_checkpoint: every tensor the
        configured architecture holds must be present with exactly its shape, and nothing else - a base+ / small / tiny or a
        truncated checkpoint fails here, by key name, instead of uploading wrong-sized operands."""
        from .weights import state_dict_spec
        spec = state_dict_spec(self.cfg)
        missing = [k for k in spec if k not in sd]
        unexpected = [k for k in sd if k not in spec]
        wrong = [f"{k}: {tuple(sd[k].shape)} != {tuple(spec[k])}" for k in spec if k in sd and tuple(sd[k].shape) != tuple(spec[k])]
        if missing or unexpected or wrong:
            raise RuntimeError("state_dict does not match the configured SAM 2.1 architecture: "
                               f"{len(missing)} missing (e.g. {missing[:3]}), {len(unexpected)} unexpected (e.g. {unexpected[:3]}), "
                               f"{len(wrong)} shape mismatches (e.g. {wrong[:3]})")
        for k, v in sd.items():
            a = v.detach().to(torch.float32).cpu().contiguous().numpy() if isinstance(v, torch.Tensor) else np.ascontiguousarray(v, np.float32)
            shape = (C.c_int64 * max(a.ndim, 1))(*a.shape)
            rc = self.lib.sam2mi_load_weight(self.h, k.encode(), a.ctypes.data_as(C.POINTER(C.c_float)), shape, a.ndim)
            self._check(rc, f"sam2mi_load_weight({k})")
        self._check(self.lib.sam2mi_finalize_weights(self.h), "sam2mi_finalize_weights")

    def new(self, *shape, dtype=torch.float32):
        return torch.empty(*shape, dtype=dtype, device=self.device)

    # ------------------------------------------------------------------ plug-level API (reference layouts)
    def image_encoder(self, img: torch.Tensor):
        """SAM2Base.inference_image: (B,3,1024,1024) -> 7-tuple."""
        _chk_f32(img)
        B = img.shape[0]
        outs = [self.new(B, 256, 64, 64), self.new(B, 256, 256, 256), self.new(B, 256, 128, 128), self.new(B, 256, 64, 64),
                self.new(B, 32, 256, 256), self.new(B, 64, 128, 128), self.new(B, 256, 64, 64)]
        arr = (C.c_void_p * 7)(*[o.data_ptr() for o in outs])
        self._check(self.lib.sam2mi_image_encoder(self.h, self.stream, _ptr(img), B, arr), "sam2mi_image_encoder")
        return tuple(outs)

    def set_image_e2e(self, img01: torch.Tensor):
        _chk_f32(img01)
        B = img01.shape[0]
        f0, f1, f2 = self.new(B, 32, 256, 256), self.new(B, 64, 128, 128), self.new(B, 256, 64, 64)
        self._check(self.lib.sam2mi_set_image_e2e(self.h, self.stream, _ptr(img01), B, _ptr(f0), _ptr(f1), _ptr(f2)),
                    "sam2mi_set_image_e2e")
        return f0, f1, f2

    def memory_attention(self, curr, memory, curr_pos, memory_pos, memory_exclude, memory_pos_exclude):
        L, P, N = memory.shape[0], memory_exclude.shape[0], curr.shape[1]
        ts = [t.contiguous() for t in (curr, memory, curr_pos, memory_pos, memory_exclude, memory_pos_exclude)]
        _chk_f32(*ts)
        out = self.new(*curr.shape)
        self._check(self.lib.sam2mi_memory_attention(self.h, self.stream, *[_ptr(t) for t in ts], L, P, N, _ptr(out)),
                    "sam2mi_memory_attention")
        return out

    def mask_decoder(self, src, tokens, pos_src, hr0, hr1):
        ts = [t.contiguous() for t in (src, tokens, pos_src, hr0, hr1)]
        _chk_f32(*ts)
        N, T = tokens.shape[0], tokens.shape[1]
        masks, iou, tok, obj = self.new(N, 4, 256, 256), self.new(N, 4), self.new(N, 4, 256), self.new(N, 1)
        self._check(self.lib.sam2mi_mask_decoder(self.h, self.stream, *[_ptr(t) for t in ts], N, T, _ptr(masks), _ptr(iou),
                                                 _ptr(tok), _ptr(obj)), "sam2mi_mask_decoder")
        return masks, iou, tok, obj

    def memory_encoder(self, pix_feat, masks):
        pix_feat, masks = pix_feat.contiguous(), masks.contiguous()
        _chk_f32(pix_feat, masks)
        N = pix_feat.shape[0]
        x, pos = self.new(N, 64, 64, 64), self.new(N, 64, 64, 64)
        self._check(self.lib.sam2mi_memory_encoder(self.h, self.stream, _ptr(pix_feat), _ptr(masks), N, _ptr(x), _ptr(pos)),
                    "sam2mi_memory_encoder")
        return x, pos

    def prompt_encoder(self, coords, labels):
        coords = coords.contiguous().to(torch.float32)
        labels = labels.contiguous().to(torch.int32)
        B, Np = labels.shape
        sparse, dense = self.new(B, Np + 1, 256), self.new(B, 256, 64, 64)
        self._check(self.lib.sam2mi_prompt_encoder(self.h, self.stream, _ptr(coords), _ptr(labels), B, Np, _ptr(sparse), _ptr(dense)),
                    "sam2mi_prompt_encoder")
        return sparse, dense

    def prompt_encoder_full(self, points=None, boxes=None, masks=None):
        """PromptEncoder.inference_prompt(points, boxes, masks) (prompt_encoder.py:215-231): points = (coords (B,Np,2), labels
        (B,Np)) or None, boxes (B,4) or None, masks (B,1,256,256) or None -> (sparse (B,S,256), dense (B,256,64,64))."""
        coords = labels = None
        if points is not None:
            coords, labels = points[0].to(self.device, torch.float32), points[1].to(self.device, torch.int32)
        if boxes is not None:                    # the two corners of a box are points with labels 2 and 3 (_embed_boxes :168-176)
            bc = boxes.to(self.device, torch.float32).reshape(-1, 2, 2)
            bl = torch.tensor([2, 3], dtype=torch.int32, device=self.device).expand(bc.shape[0], 2)
            coords = bc if coords is None else torch.cat([coords, bc], dim=1)
            labels = bl if labels is None else torch.cat([labels, bl], dim=1)
        B = coords.shape[0] if coords is not None else (masks.shape[0] if masks is not None else 1)
        Np = 0 if coords is None else coords.shape[1]
        pad = 1 if (points is not None and boxes is None) else 0
        if coords is not None:
            coords, labels = coords.contiguous(), labels.contiguous()
        if masks is not None:
            masks = masks.to(self.device, torch.float32).contiguous()
            if tuple(masks.shape[1:]) != (1, 256, 256) or masks.shape[0] != B:
                raise ValueError("mask prompts must be (B,1,256,256)")
        sparse, dense = self.new(B, Np + pad, 256), self.new(B, 256, 64, 64)
        self._check(self.lib.sam2mi_prompt_encoder_ex(self.h, self.stream, _ptr(coords), _ptr(labels), B, Np, pad, _ptr(masks), _ptr(sparse),
                                                      _ptr(dense)), "sam2mi_prompt_encoder_ex")
        return sparse, dense

    def resize_u8_pil_bicubic(self, frame_u8: torch.Tensor, size: int) -> torch.Tensor:
        """PIL Image.resize((size, size)) of a decoded RGB frame (H,W,3) uint8 on the device (bit-exact; utils/misc.py:92-101)."""
        assert frame_u8.is_cuda and frame_u8.dtype == torch.uint8 and frame_u8.dim() == 3 and frame_u8.shape[2] == 3 and frame_u8.is_contiguous()
        out = torch.empty(size, size, 3, dtype=torch.uint8, device=self.device)
        self._check(self.lib.sam2mi_resize_u8_pil_bicubic(self.h, self.stream, _ptr(frame_u8), frame_u8.shape[0], frame_u8.shape[1], _ptr(out), size),
                    "sam2mi_resize_u8_pil_bicubic")
        return out

    def resize_image_aa_bilinear(self, img_u8: torch.Tensor, size: int) -> torch.Tensor:
        """ToTensor + torchvision Resize((size, size)) of an RGB image (H,W,3) uint8 -> (3,size,size) f32 in [0,1] (utils/transforms.py:27-41)."""
        assert img_u8.is_cuda and img_u8.dtype == torch.uint8 and img_u8.dim() == 3 and img_u8.shape[2] == 3 and img_u8.is_contiguous()
        out = torch.empty(3, size, size, dtype=torch.float32, device=self.device)
        self._check(self.lib.sam2mi_resize_image_aa_bilinear(self.h, self.stream, _ptr(img_u8), img_u8.shape[0], img_u8.shape[1], _ptr(out), size),
                    "sam2mi_resize_image_aa_bilinear")
        return out

    def dense_pe(self):
        out = self.new(1, 256, 64, 64)
        self._check(self.lib.sam2mi_dense_pe(self.h, self.stream, _ptr(out)), "sam2mi_dense_pe")
        return out

    def resize_bilinear(self, x: torch.Tensor, size):
        x = x.contiguous()
        _chk_f32(x)
        lead = x.shape[:-2]
        Cn = int(np.prod(lead)) if len(lead) else 1
        out = self.new(*lead, size[0], size[1])
        self._check(self.lib.sam2mi_resize_bilinear(self.h, self.stream, _ptr(x), Cn, x.shape[-2], x.shape[-1], _ptr(out),
                                                    size[0], size[1]), "sam2mi_resize_bilinear")
        return out

    # ------------------------------------------------------------------ fused video path
    def video_encode(self, frames: torch.Tensor, feat_slots: Sequence[int]):
        _chk_f32(frames)
        B = frames.shape[0]
        sl = (C.c_int32 * B)(*feat_slots)
        self._check(self.lib.sam2mi_video_encode(self.h, self.stream, _ptr(frames), B, sl), "sam2mi_video_encode")

    def video_encode_u8(self, frames_hwc: torch.Tensor, feat_slots: Sequence[int]):
        """frames_hwc: uint8 (B, S, S, 3) decoded frames on the engine's device; normalised inside the patch-embed gather."""
        if frames_hwc.dtype != torch.uint8 or not frames_hwc.is_contiguous() or frames_hwc.device != self.device:
            raise ValueError("video_encode_u8 wants a contiguous uint8 (B,S,S,3) tensor on the engine's device")
        S = self.cfg["image_size"]
        if tuple(frames_hwc.shape[1:]) != (S, S, 3):
            raise ValueError(f"video_encode_u8 wants (B,{S},{S},3) frames, got {tuple(frames_hwc.shape)}")
        B = frames_hwc.shape[0]
        sl = (C.c_int32 * B)(*feat_slots)
        self._check(self.lib.sam2mi_video_encode_u8(self.h, self.stream, _ptr(frames_hwc), B, sl), "sam2mi_video_encode_u8")

    def set_fill_hole_area(self, max_area: int):
        self._check(self.lib.sam2mi_set_fill_hole_area(self.h, int(max_area)), "sam2mi_set_fill_hole_area")

    def fill_holes(self, masks: torch.Tensor, max_area: int) -> torch.Tensor:
        """masks: float32 (..., H, W) on the engine's device -> new tensor, small background holes set to 0.1."""
        _chk_f32(masks)
        H, W = masks.shape[-2:]
        out = torch.empty_like(masks)
        self._check(self.lib.sam2mi_fill_holes(self.h, self.stream, _ptr(masks), _ptr(out), masks.numel() // (H * W), H, W, int(max_area)),
                    "sam2mi_fill_holes")
        return out

    def _frame_out(self, want: dict) -> FrameOut:
        fo = FrameOut()
        for k, t in want.items():
            setattr(fo, k, t.data_ptr() if t is not None else None)
        return fo

    def video_click(self, feat_slot: int, coords: np.ndarray, labels: np.ndarray, multimask: bool, bank_slot: int, outs: dict,
                    mask_logits: Optional[torch.Tensor] = None):
        coords = np.ascontiguousarray(coords, np.float32).reshape(-1, 2)
        labels = np.ascontiguousarray(labels, np.int32).reshape(-1)
        if mask_logits is not None:
            _chk_f32(mask_logits)
            assert mask_logits.numel() == 256 * 256
        fo = self._frame_out(outs)
        self._check(self.lib.sam2mi_video_click(self.h, self.stream, feat_slot, coords.ctypes.data_as(C.c_void_p),
                                                labels.ctypes.data_as(C.c_void_p), len(labels), _ptr(mask_logits), int(multimask), bank_slot,
                                                C.byref(fo)), "sam2mi_video_click")

    def video_mask(self, feat_slot: int, mask1024: torch.Tensor, bank_slot: int, outs: dict):
        """mask1024: float32 {0,1} (image_size, image_size) on the device: the mask IS the output (_use_mask_as_output)."""
        _chk_f32(mask1024)
        S = self.cfg["image_size"]
        assert mask1024.numel() == S * S
        fo = self._frame_out(outs)
        self._check(self.lib.sam2mi_video_mask(self.h, self.stream, feat_slot, _ptr(mask1024), bank_slot, C.byref(fo)), "sam2mi_video_mask")

    def image_predict(self, feat_slot: int, coords, labels, multimask: bool, mask_inputs: Optional[torch.Tensor] = None, num_prompts: int = 1):
        """coords (N, Np, 2) / labels (N, Np): N independent prompts on one image, one batched decoder pass; `mask_inputs`
        (N,1,256,256) low-res logits as dense prompts; coords None = no sparse prompt (then N = num_prompts or the mask batch).
        -> masks (N, 3 or 1, 256, 256), iou (N, 3 or 1)."""
        if coords is not None:
            coords = np.ascontiguousarray(coords, np.float32)
            labels = np.ascontiguousarray(labels, np.int32)
            if coords.ndim == 2:
                coords, labels = coords[None], labels.reshape(1, -1)
            N, Np = labels.shape
            assert coords.shape == (N, Np, 2), (coords.shape, labels.shape)
            cp, lp = coords.ctypes.data_as(C.c_void_p), labels.ctypes.data_as(C.c_void_p)
        else:
            N, Np, cp, lp = (mask_inputs.shape[0] if mask_inputs is not None else num_prompts), 0, C.c_void_p(0), C.c_void_p(0)
        if mask_inputs is not None:
            mask_inputs = mask_inputs.to(self.device, torch.float32).contiguous()
            if tuple(mask_inputs.shape) != (N, 1, 256, 256):
                raise ValueError(f"mask_input must be ({N},1,256,256) low-res logits")
        c = 3 if multimask else 1
        masks, iou = self.new(N, c, 256, 256), self.new(N, c)
        self._check(self.lib.sam2mi_image_predict_ex(self.h, self.stream, feat_slot, cp, lp, N, Np, _ptr(mask_inputs), int(multimask), _ptr(masks),
                                                     _ptr(iou)), "sam2mi_image_predict_ex")
        return masks, iou

    def video_encode_memory(self, feat_slot: int, bank_slot: int, is_mask_from_pts: bool):
        self._check(self.lib.sam2mi_video_encode_memory(self.h, self.stream, feat_slot, bank_slot, int(is_mask_from_pts)),
                    "sam2mi_video_encode_memory")

    def video_track(self, feat_slot: int, sel: MemSelect, bank_slot: int, run_mem_encoder: bool, outs: dict,
                    points: Optional[np.ndarray] = None, labels: Optional[np.ndarray] = None, multimask: bool = True,
                    mask_logits: Optional[torch.Tensor] = None):
        """Plain propagation when `points` is None; otherwise correction clicks (+ previous mask logits) on a tracked frame."""
        fo = self._frame_out(outs)
        pr = None
        if points is not None:
            coords = np.ascontiguousarray(points, np.float32).reshape(-1, 2)
            lab = np.ascontiguousarray(labels, np.int32).reshape(-1)
            if mask_logits is not None:
                _chk_f32(mask_logits)
            pr = Prompt(coords.ctypes.data_as(C.c_void_p), lab.ctypes.data_as(C.c_void_p), len(lab), int(multimask),
                        C.c_void_p(mask_logits.data_ptr()) if mask_logits is not None else None)
        self._check(self.lib.sam2mi_video_track(self.h, self.stream, feat_slot, C.byref(sel), C.byref(pr) if pr is not None else None,
                                                bank_slot, int(run_mem_encoder), C.byref(fo)), "sam2mi_video_track")

    def video_track_batch(self, feat_slot: int, sels: Sequence[MemSelect], bank_slots: Sequence[int], run_mem_encoder: bool,
                          outs: Sequence[dict]):
        """Plain propagation of N (<= 8) objects on one frame in one pass; same results as N video_track calls."""
        N = len(sels)
        sel_arr = (MemSelect * N)(*sels)
        slot_arr = (C.c_int32 * N)(*bank_slots)
        fo_arr = (FrameOut * N)(*[self._frame_out(o) for o in outs])
        self._check(self.lib.sam2mi_video_track_batch(self.h, self.stream, feat_slot, N, sel_arr, slot_arr, int(run_mem_encoder), fo_arr),
                    "sam2mi_video_track_batch")

    # ------------------------------------------------------------------ profiling
    def profile_enable(self, on: bool):
        self._check(self.lib.sam2mi_profile_enable(self.h, int(on)), "sam2mi_profile_enable")

    def profile_read(self) -> dict:
        v = [C.c_double(), C.c_double(), C.c_int64(), C.c_double(), C.c_double(), C.c_int64()]
        self._check(self.lib.sam2mi_profile_read(self.h, *[C.byref(x) for x in v]), "sam2mi_profile_read")
        m = [C.c_double(), C.c_double(), C.c_int64()]
        self._check(self.lib.sam2mi_profile_read_mlp(self.h, *[C.byref(x) for x in m]), "sam2mi_profile_read_mlp")
        x = [C.c_double(), C.c_double(), C.c_int64()]
        self._check(self.lib.sam2mi_profile_read_xs(self.h, *[C.byref(y) for y in x]), "sam2mi_profile_read_xs")
        return dict(gemm_ms=v[0].value, gemm_flops=v[1].value, gemm_launches=v[2].value, attn_ms=v[3].value,
                    attn_flops=v[4].value, attn_launches=v[5].value, mlp_ms=m[0].value, mlp_flops=m[1].value,
                    mlp_launches=m[2].value, xs_ms=x[0].value, xs_flops=x[1].value, xs_launches=x[2].value, **self._prof3("ks"))

    def create_reserved_stream(self, reserve: int):
        """torch.cuda.ExternalStream over a HIP stream that leaves `reserve` CUs (spread over the XCDs) to other streams."""
        h = C.c_void_p()
        self._check(self.lib.sam2mi_stream_create_reserved(self.h, int(reserve), C.byref(h)), "sam2mi_stream_create_reserved")
        self._ext_streams = getattr(self, "_ext_streams", []) + [h]
        return torch.cuda.ExternalStream(h.value, device=self.device)

    def profile_read_kernels(self) -> dict:
        """{kernel instantiation name: dict(ms, flops, launches, bytes)} of the GEMM-family launches since profile_enable(True);
        bytes = algorithmic HBM bytes (operands read once, outputs written once)."""
        buf = C.create_string_buffer(1 << 16)
        n = self.lib.sam2mi_profile_read_kernels(self.h, buf, len(buf))
        if n < 0:
            raise RuntimeError("sam2mi_profile_read_kernels failed")
        out = {}
        for line in buf.value.decode().splitlines():
            name, ms, fl, cnt, by = line.split("\t")
            out[name] = dict(ms=float(ms), flops=float(fl), launches=int(cnt), bytes=float(by))
        return out

    def _prof3(self, name):
        v = [C.c_double(), C.c_double(), C.c_int64()]
        fn = getattr(self.lib, f"sam2mi_profile_read_{name}")
        self._check(fn(self.h, *[C.byref(y) for y in v]), f"sam2mi_profile_read_{name}")
        return {f"{name}_ms": v[0].value, f"{name}_flops": v[1].value, f"{name}_launches": v[2].value}

    # ------------------------------------------------------------------ single-kernel debug entry points (tests)
    def debug_gemm(self, A, W, bias=None, act=0, residual=None, tile_hint=0, pool_w=0):
        act = act | (tile_hint << 8) | (pool_w << 16)
        M, K = A.shape
        N = W.shape[0]
        out = self.new(M // 4 if pool_w else M, N)
        self._check(self.lib.sam2mi_debug_gemm(self.h, self.stream, _ptr(A.contiguous()), _ptr(W.contiguous()), _ptr(bias), M, N, K,
                                               act, _ptr(residual), _ptr(out)), "sam2mi_debug_gemm")
        return out

    def debug_hiera_attention(self, q, k, v, groups, heads, GQ, GK, wq, wk):
        out = self.new(*q.shape)
        self._check(self.lib.sam2mi_debug_hiera_attention(self.h, self.stream, _ptr(q.contiguous()), _ptr(k.contiguous()),
                                                          _ptr(v.contiguous()), groups, heads, GQ, GK, wq, wk, _ptr(out)),
                    "sam2mi_debug_hiera_attention")
        return out

    def debug_flash256(self, q, k, v):
        out = self.new(*q.shape)
        self._check(self.lib.sam2mi_debug_flash256(self.h, self.stream, _ptr(q.contiguous()), _ptr(k.contiguous()),
                                                   _ptr(v.contiguous()), q.shape[0], k.shape[0], _ptr(out)), "sam2mi_debug_flash256")
        return out

    def debug_rowln(self, a, ml, W, bias, x, ln_w, ln_b):
        """gemm_rowln_kernel: a = flash partials (splits, M, 256) with ml (splits, M, 2), or a plain (M, 256) operand with ml None.
        Returns (x + a' W^T + bias, LayerNorm of that as the f16 values)."""
        splits = a.shape[0] if ml is not None else 0
        M = x.shape[0]
        x = x.clone().contiguous()
        h = self.new(M, 256)
        self._check(self.lib.sam2mi_debug_rowln(self.h, self.stream, _ptr(a.contiguous()), _ptr(ml.contiguous()) if ml is not None else None, splits,
                                                _ptr(W.contiguous()), _ptr(bias.contiguous()), _ptr(x), _ptr(ln_w.contiguous()), _ptr(ln_b.contiguous()),
                                                M, _ptr(h)), "sam2mi_debug_rowln")
        return x, h

    def debug_projln(self, a, W, bias, x, ln_w, ln_b):
        """gemm_projln_kernel: (x + a W^T + bias, LayerNorm of that (eps 1e-6) as the f16 values); a (M, C), W (C, C), C in {144, 288, 576}."""
        M, Cc = a.shape
        x = x.clone().contiguous()
        h = self.new(M, Cc)
        self._check(self.lib.sam2mi_debug_projln(self.h, self.stream, _ptr(a.contiguous()), _ptr(W.contiguous()), _ptr(bias.contiguous()), _ptr(x),
                                                 _ptr(ln_w.contiguous()), _ptr(ln_b.contiguous()), M, Cc, _ptr(h)), "sam2mi_debug_projln")
        return x, h

    def debug_hiera_block(self, idx: int, x_nhwc: torch.Tensor, out_shape):
        out = self.new(*out_shape)
        self._check(self.lib.sam2mi_debug_hiera_block(self.h, self.stream, idx, _ptr(x_nhwc.contiguous()), x_nhwc.shape[0], _ptr(out)),
                    "sam2mi_debug_hiera_block")
        return out

    def debug_read(self, name: str, *shape):
        out = self.new(*shape)
        self._check(self.lib.sam2mi_debug_read(self.h, self.stream, name.encode(), _ptr(out), C.c_int64(out.numel())), "sam2mi_debug_read")
        return out

    def debug_flash_bench(self, Nq, Nk, iters=10) -> float:
        ms = C.c_float()
        self._check(self.lib.sam2mi_debug_flash_bench(self.h, self.stream, Nq, Nk, iters, C.byref(ms)), "sam2mi_debug_flash_bench")
        return ms.value

    def debug_mlp(self, xn, W1, b1, W2, b2, x, fused=True, iters=0):
        """x + fc2(GELU(fc1(xn))) through the fused MLP kernel (or the two-GEMM path); returns (out, ms per launch or None)."""
        M, Cc = xn.shape
        out = x.clone().contiguous()
        ms = C.c_float()
        self._check(self.lib.sam2mi_debug_mlp(self.h, self.stream, _ptr(xn.contiguous()), _ptr(W1.contiguous()), _ptr(b1.contiguous()),
                                              _ptr(W2.contiguous()), _ptr(b2.contiguous()), _ptr(out), M, Cc, int(fused), int(iters),
                                              C.byref(ms)), "sam2mi_debug_mlp")
        return out, (ms.value if iters > 0 else None)

    def debug_gemm_bench(self, M, N, K, iters=20, mode=0) -> float:
        ms = C.c_float()
        self._check(self.lib.sam2mi_debug_gemm_bench(self.h, self.stream, M, N, K, iters, mode, C.byref(ms)), "sam2mi_debug_gemm_bench")
        return ms.value
