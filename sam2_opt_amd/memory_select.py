"""Host-side memory-bank selection of the video path: which stored frames a tracked frame cross-attends to.

Pure Python, no tensors: restates step 1 of SAM2Base._prepare_memory_conditioned_features
(/root/reference/sam2/sam2/modeling/sam2_base_official.py:823-920) and select_closest_cond_frames
(modeling/sam2_utils.py:19-61) on dictionaries `frame index -> stored output`, so the same code serves the predictor (outputs =
bank-slot records) and the CPU tests (outputs = anything).  The device side only receives the resulting slot lists.
"""
from __future__ import annotations

from typing import Dict, List, Tuple


def select_closest_cond_frames(frame_idx: int, cond: Dict[int, object], max_cond_frame_num: int):
    """sam2_utils.py:19-61: (selected, unselected) conditioning outputs; -1 = all."""
    if max_cond_frame_num == -1 or len(cond) <= max_cond_frame_num:
        return dict(cond), {}
    assert max_cond_frame_num >= 2, "we should allow using 2+ conditioning frames"
    sel = {}
    before = max((t for t in cond if t < frame_idx), default=None)
    if before is not None:
        sel[before] = cond[before]
    after = min((t for t in cond if t >= frame_idx), default=None)
    if after is not None:
        sel[after] = cond[after]
    remain = sorted((t for t in cond if t not in sel), key=lambda x: abs(x - frame_idx))[:max_cond_frame_num - len(sel)]
    sel.update((t, cond[t]) for t in remain)
    return sel, {t: v for t, v in cond.items() if t not in sel}


def select_memory(cond: Dict[int, object], non_cond: Dict[int, object], frame_idx: int, num_frames: int, reverse: bool,
                  num_maskmem: int = 7, max_obj_ptrs_in_encoder: int = 16, max_cond_frames_in_attn: int = -1,
                  memory_temporal_stride_for_eval: int = 1) -> Tuple[List[Tuple[int, object]], List[Tuple[int, object]], int]:
    """Returns (spatial memories [(t_pos, output)], object pointers [(signed frame distance, output)], max_ptrs).

    Spatial memories (:826-868): the selected conditioning frames first (t_pos 0), then the last num_maskmem - 1 frames before
    (after, when tracking in reverse) the current one - the nearest frame always, the others on the stride grid; an unselected
    conditioning frame standing at one of those positions is attended to like a non-conditioning one.  Missing frames are
    skipped.  Pointers (:887-920): selected conditioning frames in the past of the tracking direction (signed distance), then
    up to max_ptrs - 1 previous frames (non-conditioning outputs, or unselected conditioning ones)."""
    selected, unselected = select_closest_cond_frames(frame_idx, cond, max_cond_frames_in_attn)
    mems: List[Tuple[int, object]] = [(0, out) for out in selected.values()]
    r = memory_temporal_stride_for_eval
    for t_pos in range(1, num_maskmem):
        t_rel = num_maskmem - t_pos
        if t_rel == 1:
            prev = frame_idx + t_rel if reverse else frame_idx - t_rel
        elif not reverse:
            prev = ((frame_idx - 2) // r) * r - (t_rel - 2) * r
        else:
            prev = -(-(frame_idx + 2) // r) * r + (t_rel - 2) * r
        out = non_cond.get(prev)
        if out is None:
            out = unselected.get(prev)
        if out is not None:
            mems.append((t_pos, out))
    max_ptrs = min(num_frames, max_obj_ptrs_in_encoder)
    sign = -1 if reverse else 1
    ptrs: List[Tuple[int, object]] = [((frame_idx - t) * sign, out) for t, out in selected.items()
                                      if (t >= frame_idx if reverse else t <= frame_idx)]
    for t_diff in range(1, max_ptrs):
        t = frame_idx + t_diff if reverse else frame_idx - t_diff
        if t < 0 or t >= num_frames:
            break
        out = non_cond.get(t)
        if out is None:
            out = unselected.get(t)
        if out is not None:
            ptrs.append((t_diff, out))
    return mems, ptrs, max_ptrs
