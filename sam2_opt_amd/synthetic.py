"""Seeded synthetic inputs (there is no dataset or checkpoint offline).

The same generators feed bench.py, the parity tests and oracle/gen_golden.py, so a
golden vector is identified by (generator, seed) alone and inputs never have to be
committed.  Recipes follow SURVEY.md 8(d).
"""
from __future__ import annotations

import numpy as np
import torch

from .config import get_config


def synthetic_image_normed(seed: int = 1, batch: int = 1, size: int = 1024) -> torch.Tensor:
    """Config 2: an already-normalised encoder input, N(0,1)."""
    rs = np.random.RandomState(seed)
    return torch.from_numpy(rs.standard_normal((batch, 3, size, size)).astype(np.float32))


def synthetic_frames_u8(seed: int = 2, num_frames: int = 100, size: int = 1024) -> np.ndarray:
    """Config 3: uint8 RGB frames (T,H,W,3).  Low-frequency blobs + noise so that the
    tracked mask has structure (pure white noise gives near-constant logits)."""
    rs = np.random.RandomState(seed)
    # noise first, frames after and in order: the first n frames of a longer clip with the
    # same seed are identical to an n-frame clip
    noise = rs.randint(0, 64, (size, size, 3)).astype(np.uint8)
    base = rs.randint(0, 256, (num_frames, size // 32, size // 32, 3)).astype(np.uint8)
    frames = np.repeat(np.repeat(base, 32, axis=1), 32, axis=2)
    return (frames // 4 * 3 + noise[None]).astype(np.uint8)


def normalize_frames(frames_u8: np.ndarray, cfg: dict | None = None) -> torch.Tensor:
    """uint8 (T,H,W,3) -> float32 (T,3,H,W), /255 then mean/std - what
    load_video_frames_from_jpg_images does after decoding
    (/root/reference/sam2/sam2/utils/misc.py:92-101,:270-276); frames are already 1024^2
    so the PIL resize is the identity."""
    cfg = cfg or get_config("large")
    x = torch.from_numpy(frames_u8).permute(0, 3, 1, 2).to(torch.float32) / 255.0
    mean = torch.tensor(cfg["img_mean"], dtype=torch.float32).view(1, 3, 1, 1)
    std = torch.tensor(cfg["img_std"], dtype=torch.float32).view(1, 3, 1, 1)
    return (x - mean) / std


def randn(seed: int, *shape, scale: float = 1.0) -> torch.Tensor:
    rs = np.random.RandomState(seed)
    return torch.from_numpy((rs.standard_normal(shape) * scale).astype(np.float32))
