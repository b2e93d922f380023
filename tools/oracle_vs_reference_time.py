"""Build-container script: is the CPU oracle (oracle/sam2_ref.py, what bench.py times as `cpu_baseline` on the GPU box, where
the reference cannot travel) a fair stand-in for the reference's own torch backend?  Same clip, same click, same thread count:
wall time of the propagate loop of both, and the largest mask difference.

    python tools/oracle_vs_reference_time.py [frames] [threads]        (needs /root/reference; ~6 s per frame and side)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import sam2_ref as R
from oracle.ref_import import build_reference_model
from sam2_opt_amd.config import get_config
from sam2_opt_amd.synthetic import normalize_frames, synthetic_frames_u8
from sam2_opt_amd.weights import synthetic_state_dict


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    torch.set_num_threads(int(sys.argv[2]) if len(sys.argv) > 2 else 8)
    cfg = get_config("large")
    sd = synthetic_state_dict(cfg, seed=0)
    frames = normalize_frames(synthetic_frames_u8(seed=2, num_frames=T), cfg)
    click = (np.array([[512.0, 512.0]], np.float32), np.array([1], np.int32))
    res = {}
    with torch.inference_mode():
        model = build_reference_model(cfg, "video", sd, fill_hole_area=0)
        import sam2.sam2_video_predictor_official as vp
        vp.load_video_frames = lambda **kw: (frames, 1024, 1024)
        state = model.init_state(video_path="synthetic")
        model.add_new_points_or_box(state, frame_idx=0, obj_id=1, points=click[0], labels=click[1])
        t0 = time.perf_counter()
        ref = [vm.clone() for _, _, vm in model.propagate_in_video(state)]
        res["reference"] = time.perf_counter() - t0
        vo = R.VideoOracle(sd, cfg, frames)
        vo.add_new_points(0, *click)
        t0 = time.perf_counter()
        ora = [m.clone() for _, m in vo.propagate()]
        res["oracle"] = time.perf_counter() - t0
    diff = max(float((a - b).abs().max()) for a, b in zip(ref, ora))
    ratio = res["oracle"] / res["reference"]
    print(f"{T} frames, {torch.get_num_threads()} threads: reference {res['reference']:.1f} s ({T / res['reference']:.3f} frames/s), "
          f"oracle {res['oracle']:.1f} s ({T / res['oracle']:.3f} frames/s), oracle / reference = {ratio:.3f}, max |mask diff| = {diff:.2e}")
    assert 0.9 <= ratio <= 1.1, "the oracle is not within +-10 % of the reference's wall time"


if __name__ == "__main__":
    main()
