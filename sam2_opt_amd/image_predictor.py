"""SAM2ImagePredictor on the MI355X-native backend - host-side mirror of the reference class
(/root/reference/sam2/sam2/sam2_image_predictor.py: set_image :140-171, set_image_batch :279-300,
predict :387-454, _prep_prompts :456-485, _predict :487-589; coordinate transforms utils/transforms.py:50-76,
postprocess_masks :78-120 with max_hole_area = max_sprinkle_area = 0).

Images enter as uint8 HWC arrays like in the reference.  Inputs that are not 1024x1024 are resized on the device by
sam2mi_resize_image_aa_bilinear (csrc/resize.hip): the operator torchvision's tensor `Resize` applies (bilinear, antialias).
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from .config import get_config
from .native import Engine, default_precision


class SAM2ImagePredictor:
    def __init__(self, model: str = "large", state_dict=None, ckpt_path: Optional[str] = None, device=None,
                 mask_threshold: float = 0.0, max_batch: int = 8, precision: Optional[str] = None):
        self.cfg = get_config(model)
        if state_dict is None and ckpt_path is not None:
            state_dict = torch.load(ckpt_path, map_location="cpu", weights_only=True)["model"]
        if state_dict is None:
            raise ValueError("state_dict or ckpt_path is required")
        self.max_batch = max_batch
        self.engine = Engine(self.cfg, state_dict=state_dict, max_batch=max_batch, feat_slots=max(2 * max_batch, 16), device=device, precision=precision or default_precision(model))
        self.device = self.engine.device
        self.mask_threshold = mask_threshold
        self.image_size = self.cfg["image_size"]
        self._transforms = _Transforms(self.image_size)
        self.reset_predictor()

    def speedup(self, backend: str = "hip", use_cache: bool = True, model_root_path=None):
        if backend not in ("hip", "mi355x", "sam2mi"):
            raise RuntimeError(f"Unknown backend={backend}: this predictor only runs the MI355X HIP backend")

    def release(self):
        self.engine.close()

    def reset_predictor(self):
        self._is_image_set = False
        self._orig_hw: List = []
        self._slots: List[int] = []

    # ------------------------------------------------------------------ images
    @torch.no_grad()
    def set_image(self, image):
        if not isinstance(image, np.ndarray):          # PIL image
            image = np.array(image)
        self.set_image_batch([image])

    @torch.no_grad()
    def set_image_batch(self, image_list: List[np.ndarray]):
        self.reset_predictor()
        assert isinstance(image_list, list)
        if len(image_list) > self.engine.feat_slots:
            raise ValueError(f"at most {self.engine.feat_slots} images per batch (feat_slots)")
        S = self.image_size
        mean = torch.tensor(self.cfg["img_mean"], device=self.device).view(1, 3, 1, 1)
        std = torch.tensor(self.cfg["img_std"], device=self.device).view(1, 3, 1, 1)
        tensors = []
        for im in image_list:
            assert isinstance(im, np.ndarray), "Images are expected to be an np.ndarray in RGB format, and of shape  HWC"
            self._orig_hw.append(tuple(im.shape[:2]))
            u8 = torch.from_numpy(np.ascontiguousarray(im[..., :3])).to(self.device)
            if u8.dtype != torch.uint8:
                raise ValueError("images must be uint8 HWC RGB arrays")
            if tuple(u8.shape[:2]) != (S, S):        # ToTensor + torchvision Resize (bilinear, antialias) on the device (resize.hip)
                t = self.engine.resize_image_aa_bilinear(u8.contiguous(), S)[None]
            else:
                t = u8.permute(2, 0, 1)[None].float() / 255.0
            tensors.append((t - mean) / std)
        self._slots = list(range(len(tensors)))
        for i in range(0, len(tensors), self.max_batch):
            batch = torch.cat(tensors[i:i + self.max_batch], dim=0).contiguous()
            self.engine.video_encode(batch, self._slots[i:i + self.max_batch])
        self._is_image_set = True

    # ------------------------------------------------------------------ prompts
    @torch.no_grad()
    def predict(self, point_coords=None, point_labels=None, box=None, mask_input=None, multimask_output=True,
                return_logits=False, normalize_coords=True, img_idx: int = -1):
        """SAM2ImagePredictor.predict (sam2_image_predictor.py:387-454): numpy in, numpy out for ONE prompt set."""
        if not self._is_image_set:
            raise RuntimeError("An image must be set with .set_image(...) before mask prediction.")
        mask_input, unnorm_coords, labels, unnorm_box = self._prep_prompts(point_coords, point_labels, box, mask_input, normalize_coords, img_idx)
        masks, iou, low = self._predict(unnorm_coords, labels, unnorm_box, mask_input, multimask_output, return_logits=return_logits, img_idx=img_idx)
        return masks[0].float().cpu().numpy(), iou[0].float().cpu().numpy(), low[0].float().cpu().numpy()

    @torch.no_grad()
    def predict_batch(self, point_coords_batch=None, point_labels_batch=None, box_batch=None, mask_input_batch=None,
                      multimask_output=True, return_logits=False, normalize_coords=True):
        """Per image: (masks, ious, low_res) for that image's prompts (sam2_image_predictor.py:302-385)."""
        if not self._is_image_set:
            raise RuntimeError("An image must be set with .set_image_batch(...) before mask prediction.")
        outs = []
        for i in range(len(self._slots)):
            pc = None if point_coords_batch is None else point_coords_batch[i]
            pl = None if point_labels_batch is None else point_labels_batch[i]
            bx = None if box_batch is None else box_batch[i]
            mi = None if mask_input_batch is None else mask_input_batch[i]
            mi, uc, lab, ub = self._prep_prompts(pc, pl, bx, mi, normalize_coords, img_idx=i)
            m, s, l = self._predict(uc, lab, ub, mi, multimask_output, return_logits=return_logits, img_idx=i)
            outs.append((m.float().cpu().numpy(), s.float().cpu().numpy(), l.float().cpu().numpy()))
        return outs

    def _prep_prompts(self, point_coords, point_labels, box, mask_logits, normalize_coords, img_idx=-1):
        """sam2_image_predictor.py:456-485: to device tensors in the model's 1024-pixel frame."""
        unnorm_coords, labels, unnorm_box, mask_input = None, None, None, None
        if point_coords is not None:
            assert point_labels is not None, "point_labels must be supplied if point_coords is supplied."
            point_coords = torch.as_tensor(point_coords, dtype=torch.float, device=self.device)
            unnorm_coords = self._transforms.transform_coords(point_coords, normalize=normalize_coords, orig_hw=self._orig_hw[img_idx])
            labels = torch.as_tensor(point_labels, dtype=torch.int, device=self.device)
            if len(unnorm_coords.shape) == 2:
                unnorm_coords, labels = unnorm_coords[None, ...], labels[None, ...]
        if box is not None:
            box = torch.as_tensor(box, dtype=torch.float, device=self.device)
            unnorm_box = self._transforms.transform_boxes(box, normalize=normalize_coords, orig_hw=self._orig_hw[img_idx])      # Bx2x2
        if mask_logits is not None:
            mask_input = torch.as_tensor(mask_logits, dtype=torch.float, device=self.device)
            if len(mask_input.shape) == 3:
                mask_input = mask_input[None, :, :, :]
        return mask_input, unnorm_coords, labels, unnorm_box

    @torch.no_grad()
    def _predict(self, point_coords, point_labels, boxes=None, mask_input=None, multimask_output=True, return_logits=False, img_idx=-1):
        """SAM2ImagePredictor._predict (sam2_image_predictor.py:487-589), the reference's signature: prompts are tensors already in
        the model frame (what `_prep_prompts` / SAM2AutomaticMaskGenerator._process_batch hand over: point_coords (B,N,2), labels
        (B,N), boxes (B,2,2) or (B,4), mask_input (B,1,256,256)).  All B prompts go through one batched decoder pass on the device."""
        if not self._is_image_set:
            raise RuntimeError("An image must be set with .set_image(...) before mask prediction.")
        orig_hw = self._orig_hw[img_idx]
        slot = self._slots[img_idx]
        pts = None if point_coords is None else torch.as_tensor(point_coords, dtype=torch.float32)
        lab = None if point_labels is None else torch.as_tensor(point_labels).to(torch.int32)
        if pts is not None and pts.dim() == 2:
            pts, lab = pts[None], lab[None]
        if boxes is not None:                      # a box = its two corners as points with labels 2 / 3, in front of the user's points (:509-522)
            bc = torch.as_tensor(boxes, dtype=torch.float32).reshape(-1, 2, 2).to(pts.device if pts is not None else self.device)
            bl = torch.tensor([[2, 3]], dtype=torch.int32, device=bc.device).repeat(bc.shape[0], 1)
            pts = bc if pts is None else torch.cat([bc, pts.to(bc.device)], dim=1)
            lab = bl if lab is None else torch.cat([bl, lab.to(bc.device)], dim=1)
        if mask_input is not None:
            mask_input = torch.as_tensor(mask_input, dtype=torch.float32)
            if mask_input.dim() == 3:
                mask_input = mask_input[None]
        low, iou = self.engine.image_predict(slot, None if pts is None else pts.cpu().numpy(), None if lab is None else lab.cpu().numpy(),
                                             multimask_output, mask_inputs=mask_input)       # all prompts in one batched decoder pass
        masks = self.engine.resize_bilinear(low, orig_hw) if tuple(low.shape[-2:]) != tuple(orig_hw) else low
        low = torch.clamp(low, -32.0, 32.0)
        if not return_logits:
            masks = masks > self.mask_threshold
        return masks, iou, low

    def get_image_embedding(self):
        raise NotImplementedError("the frame features stay inside the engine (feature slots); use Engine.image_encoder for the plug-level tensors")


class _Transforms:
    """The coordinate half of SAM2Transforms (utils/transforms.py:50-76); the image half runs on the device (csrc/resize.hip)."""

    def __init__(self, resolution: int):
        self.resolution = resolution

    def transform_coords(self, coords: torch.Tensor, normalize=False, orig_hw=None) -> torch.Tensor:
        if normalize:
            assert orig_hw is not None
            h, w = orig_hw
            coords = coords.clone()
            coords[..., 0] = coords[..., 0] / w
            coords[..., 1] = coords[..., 1] / h
        return coords * self.resolution

    def transform_boxes(self, boxes: torch.Tensor, normalize=False, orig_hw=None) -> torch.Tensor:
        return self.transform_coords(boxes.reshape(-1, 2, 2), normalize, orig_hw)
