"""TEST INFRASTRUCTURE (oracle): CPU restatement of the reference's small-hole filling.

Follows fill_holes_in_mask_scores (/root/reference/sam2/sam2/utils/misc.py:312-338) and the contract of
get_connected_components (:47-62): 8-connectivity components of the background (score <= 0); components of area
<= max_area are set to 0.1.  The reference computes the components with its CUDA extension
(csrc/connected_components.cu), which cannot be built here (no CUDA).  This file uses scipy.ndimage.label, an
independent implementation of the same definition; it is pinned (a) by hand-made known-answer cases and (b) against
oracle/cc_blockuf.py, the stage-by-stage CPU restatement of that CUDA kernel's block-based union-find: identical
partitions and per-pixel areas on random and adversarial masks (tests/test_postproc.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
import numpy as np
from scipy import ndimage

_EIGHT = np.ones((3, 3), dtype=bool)


def connected_components(mask: np.ndarray):
    """mask: bool (H, W), True = foreground of the labelling -> (labels int32 (0 = background), areas int32 per pixel)."""
    labels, n = ndimage.label(mask, structure=_EIGHT)
    counts = np.bincount(labels.ravel(), minlength=n + 1)
    counts[0] = 0
    return labels.astype(np.int32), counts[labels].astype(np.int32)


def fill_holes_in_mask_scores(mask: np.ndarray, max_area: int) -> np.ndarray:
    """mask: float32 (..., H, W) scores -> copy with background holes of area <= max_area set to 0.1 (misc.py:320-324)."""
    assert max_area > 0, "max_area must be positive"
    out = np.array(mask, dtype=np.float32, copy=True)
    flat = out.reshape(-1, *out.shape[-2:])
    for m in flat:
        labels, areas = connected_components(m <= 0)
        m[(labels > 0) & (areas <= max_area)] = 0.1
    return out
