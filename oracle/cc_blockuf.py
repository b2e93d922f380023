"""TEST INFRASTRUCTURE (oracle): CPU restatement of the reference's connected-components CUDA kernel.

Restates, stage by stage, /root/reference/sam2/sam2/csrc/connected_components.cu (block-based union-find on 2x2 pixel
blocks, 8-connectivity, adapted there from zsef123/Connected_components_PyTorch):

    init_labeling   :62-70    every 2x2 block (origin = its top-left pixel, even row / even col) starts as its own root
    merge           :72-118   a block looks at its four already-visited neighbour blocks (top-left, top, top-right, left)
                              through the 16-bit neighbourhood mask P and unions with those it touches
    compression     :120-127  path compression of every block root
    final_labeling  :129-166  the four pixels of a block get label root + 1 where the image is set, 0 elsewhere
    init_counting / final_counting :168-209  per-pixel area = number of pixels carrying the same label
    get_connected_componnets :212-282  host entry; REJECTS odd H or W (:226-227) - utils/misc.py:321-336 then catches the
                              error, warns and SKIPS hole filling, so on odd sizes the reference returns the mask unchanged

The CUDA kernels run the unions concurrently with atomicMin; union-find with "smaller index wins" gives the same
partition and the same roots (the minimum block-origin index of a component) in any order, so a sequential sweep is an
exact restatement.  The CUDA extension itself cannot be built here (no CUDA toolchain), so this file is the pin that
ties oracle/postproc.py (scipy.ndimage.label, an independent implementation of the definition) and, through it,
csrc/postproc.hip to the reference's algorithm: tests/test_postproc.py checks the two oracles equal on random and
adversarial masks (identical partition, identical per-pixel areas) and the known-answer cases.

Only tests/ may import this module.  Pure Python loops over blocks: meant for masks up to 256x256 (16,384 blocks).
"""
from __future__ import annotations

import numpy as np


def _find(buf, n):
    """find (:29-33)."""
    while buf[n] != n:
        n = buf[n]
    return n


def _union(buf, a, b):
    """union_ (:44-60): the smaller root wins (atomicMin); sequentially one pass suffices."""
    a, b = _find(buf, a), _find(buf, b)
    if a < b:
        buf[b] = a
    elif b < a:
        buf[a] = b


def get_connected_components(img: np.ndarray):
    """img: (H, W) bool / uint8, non-zero = foreground of the labelling.  Returns (labels, counts) int32 (H, W) exactly as
    get_connected_componnets does for one [1, 1, H, W] image: label = root block-origin index + 1, 0 on background;
    counts = area of the pixel's component, 0 on background.  Raises ValueError on odd sizes like the AT_ASSERTMs."""
    img = np.ascontiguousarray(np.asarray(img) != 0)
    if img.ndim != 2:
        raise ValueError("inputs must be [H, W]")
    H, W = img.shape
    if H % 2:
        raise ValueError("height must be an even number")          # :226
    if W % 2:
        raise ValueError("width must be an even number")           # :227
    im = img.reshape(-1)
    label = np.zeros(H * W, dtype=np.int64)
    # ---- init_labeling
    for row in range(0, H, 2):
        for col in range(0, W, 2):
            label[row * W + col] = row * W + col
    # ---- merge: P bit k = "pixel k of the 4x4 neighbourhood whose rows/cols are (row-1 .. row+2, col-1 .. col+2) may matter";
    # bit 4*r + c <-> neighbourhood row r, column c.  0x777 = the 3x3 patch around the block's top-left pixel, shifted by 1 /
    # 4 for the pixel to the right / below (the bottom-right pixel of the block adds nothing the other three do not cover).
    for row in range(0, H, 2):
        for col in range(0, W, 2):
            idx = row * W + col
            P = 0
            if im[idx]:
                P |= 0x777
            if row + 1 < H and im[idx + W]:
                P |= 0x777 << 4
            if col + 1 < W and im[idx + 1]:
                P |= 0x777 << 1
            if col == 0:
                P &= 0xEEEE
            if col + 1 >= W:
                P &= 0x3333
            elif col + 2 >= W:
                P &= 0x7777
            if row == 0:
                P &= 0xFFF0
            if row + 1 >= H:
                P &= 0xFF
            if P > 0:
                if (P >> 0) & 1 and im[idx - W - 1]:
                    _union(label, idx, idx - 2 * W - 2)             # top-left block
                if ((P >> 1) & 1 and im[idx - W]) or ((P >> 2) & 1 and im[idx - W + 1]):
                    _union(label, idx, idx - 2 * W)                 # top block
                if (P >> 3) & 1 and im[idx + 2 - W]:
                    _union(label, idx, idx - 2 * W + 2)             # top-right block
                if ((P >> 4) & 1 and im[idx - 1]) or ((P >> 8) & 1 and im[idx + W - 1]):
                    _union(label, idx, idx - 2)                     # left block
    # ---- compression + final_labeling
    out = np.zeros(H * W, dtype=np.int32)
    for row in range(0, H, 2):
        for col in range(0, W, 2):
            idx = row * W + col
            y = _find(label, idx) + 1
            for d in (0, 1, W, W + 1):
                out[idx + d] = y if im[idx + d] else 0
    # ---- init_counting / final_counting
    counts_init = np.bincount(out[out > 0] - 1, minlength=H * W)
    counts = np.where(out > 0, counts_init[np.maximum(out - 1, 0)], 0).astype(np.int32)
    return out.reshape(H, W), counts.reshape(H, W)


def fill_holes_in_mask_scores(mask: np.ndarray, max_area: int) -> np.ndarray:
    """utils/misc.py:312-338 on top of the restated kernel, including the reference's failure path: when the kernel rejects
    the input (odd H or W) the mask comes back unchanged."""
    assert max_area > 0, "max_area must be positive"
    out = np.array(mask, dtype=np.float32, copy=True)
    flat = out.reshape(-1, *out.shape[-2:])
    for m in flat:
        try:
            labels, areas = get_connected_components(m <= 0)
        except ValueError:
            continue                                                # misc.py:325-336: warn and skip
        m[(labels > 0) & (areas <= max_area)] = 0.1
    return out
