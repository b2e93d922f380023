#!/usr/bin/env python3
"""Headline benchmark: frames/s of SAM2.1-hiera-large 1024x1024 video propagation (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload = BASELINE.json configs[2]/[3]: every GPU tracks its own synthetic 100-frame 1024^2 clip (seed 2 + rank)
from one positive click on frame 0 with seeded random-init hiera-large weights.  One STEP = one pass of
`propagate_in_video` over the whole clip (the method of the upstream FPS benchmark,
/root/reference/sam2/sam2/benchmark.py:71-92: wall time of the propagate loop, frames already on the device,
fps = frames / time).  W warm-up steps, then K timed steps between barrier + torch.cuda.synchronize();
time = MAX over ranks; value = N * K * frames / time (weak scaling: independent clips, no data-path collective).

The JSON line also carries
  roofline     - the dominant kernel = the instantiation of the three MFMA GEMM kernel families (tiled `gemm_v2_kernel`,
                 X-stationary `gemm_xs_kernel`, fused-MLP `mlp_fused_kernel`) that takes the most time, under its
                 rocprofv3 name: algorithmic FLOPs / launch time, measured with HIP events around every launch on its own
                 stream in a second, un-timed pass of the same K steps (sam2mi_profile_enable), vs the 2.5 PFLOP/s dense
                 f16 MFMA peak; every other instantiation and the family totals beside it;
  cpu_baseline - the CPU oracle (oracle/sam2_ref.py, a port of the reference's torch backend) timed on this
                 box's host cores on the first 16 frames of the same clip; value = STEADY-STATE frames/s over frames 8..15
                 (memory bank full: L = 7), the plan of BASELINE.md 3 (rank 0, N = 1 only);
  parity       - the timed precision mode against the REAL reference's 24-frame golden (tests/golden/large_video24_full.npz: every
                 low-res pixel of every frame), measured by this run on the bench configuration (encode_batch, prefetch stream):
                 max-abs / max|ref|, relative L2, binarised-pixel disagreement, worst frame; `parity_class` = the bar it is held to;
  secondary    - (N = 1) the same clip in the other precision modes - "f16" (plain f16 operands, the bf16-class parity tier) with
                 its own roofline and parity objects, "f16x3" (every operand split) - the drop-in route, and images/s of
                 BASELINE.json configs[4] (16 images x 8 point prompts, one encoder call).

Precision modes (sam2mi_config.precision): the default timed mode is "f16s" - f16 MFMA operands with a SELECTIVE
2-term split of the operands whose rounding carries the error - because it is the fastest mode that meets the north-star bar
"masks within 1e-3 of the reference"; "f16" is faster and sits at 2e-3.

`--backend gloo --stub-predictor` runs the SAME orchestration (rank / world from the environment, process-group init, warm-up,
barrier, timed steps, MAX-over-ranks reduction, result gather, the JSON line) on the CPU with a stand-in predictor: that is how
tests/test_dist_gloo.py covers the N > 1 path, which otherwise only ever executes on the driver's 8-GPU node.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F16_TFLOPS = 2500.0                 # dense f16/bf16 MFMA peak of MI355X (MI355X_MICROARCH.md)
PEAK_HBM_GBPS = 8000.0                   # HBM3E peak (same guide); ridge = 2500e12 / 8e12 = 312 FLOP/B
CLIP_GFLOP_PROPAGATE = 240450.0          # algorithmic GFLOP of one 100-frame propagate loop


def _pmc_traffic(kernel):
    """HBM bytes per launch of the kernel instantiation `kernel` from the committed rocprofv3 --pmc passes (separate
    FETCH_SIZE and WRITE_SIZE runs, gfx950 x2 fetch correction; profiles/*_pmc_traffic.md).  PMC collection cannot run
    inside the timed process, so this is the last committed measurement, or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            d = json.load(f)
        return d.get("by_instantiation", {}).get(kernel, {}).get("bytes_per_launch")
    except Exception:
        return None


def _lib_hash():
    """Hash of the sources libsam2mi.so was built from (sam2_opt_amd/build.py): ties a result to a code state."""
    try:
        from sam2_opt_amd.build import built_hash
        return built_hash()[:16]
    except Exception:
        return None


def timed_steps(D, dist, device, one_step, warmup: int, steps: int):
    """The timing contract: W untimed warm-up steps, barrier + device sync, EXACTLY K timed steps, barrier + device sync, MAX of
    the wall time over ranks, then the trivial gather of (frames, seconds, mask checksum) per rank.
    `one_step() -> (frames processed, last masks or None)`."""
    for _ in range(warmup):
        one_step()
    D.barrier(dist, device)
    t0 = time.perf_counter()
    nframes, last = 0, None
    for _ in range(steps):
        n, last = one_step()
        nframes += n
    D.barrier(dist, device)
    dt = time.perf_counter() - t0
    checksum = float((last > 0).float().mean().item()) if last is not None else 0.0
    dt, total_frames, records = D.reduce_time_and_gather(dist, nframes, dt, checksum, device)
    return dt, total_frames, records, nframes, last


def stub_main(args, rank, world):
    """Orchestration-only run on the CPU (gloo): per-rank seeds, process group, barrier, reduction, gather and the JSON line of
    the real benchmark, with a predictor that just sleeps.  The printed line is marked `"data": "stub"` and is not a result."""
    from sam2_opt_amd import dist as D
    device = torch.device("cpu")
    dist = D.init(args.backend)
    seed = D.clip_seed_for_rank(2, rank)

    def one_step():
        time.sleep(0.002 * args.frames * (1 + rank))              # the higher rank is slower: MAX over ranks must pick it
        return args.frames, torch.full((1, 1, 4, 4), float(seed))
    dt, total_frames, records, nframes, last = timed_steps(D, dist, device, one_step, args.warmup, args.steps)
    devices = D.gather_strings(dist, f"cpu (rank {rank}, pid {os.getpid()})")
    if rank == 0:
        print(json.dumps({"metric": "frames/sec SAM2.1-hiera-large 1024x1024 video propagation", "value": round(total_frames / dt, 3),
                          "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": args.precision, "data": "stub",
                          "config": {"workload": "orchestration test (stub predictor on CPU)", "frames_per_step": args.frames,
                                     "parallelism": f"clips x{world} (one process per rank)", "per_rank": records, "clip_seed_rank0": seed,
                                     "rccl_ranks": dist.get_world_size() if dist is not None else 1, "devices": devices}}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def roofline_pass(args, pred, one_step, steps, dt):
    """Second, un-timed pass of the same `steps` steps with HIP events around every launch (sam2mi_profile_enable) -> the roofline
    object of the JSON line.  `dt` = wall time of the timed steps (for the shares)."""
    # per-launch kernel durations (HIP events on the launch stream): measured with the encoder prefetch stream off, so
    # that every kernel runs alone on the chip as it does under rocprofv3 --kernel-trace (profiles/)
    torch.cuda.synchronize()
    overlap, pred.overlap_encode = pred.overlap_encode, False
    pred.engine.profile_enable(True)
    for _ in range(steps):
        one_step()
    torch.cuda.synchronize()
    pr = pred.engine.profile_read()
    pred.engine.profile_enable(False)
    pred.overlap_encode = overlap
    # The MFMA GEMM work of the path runs in three hand-written kernel families (tiled gemm_v2_kernel, X-stationary
    # gemm_xs_kernel, fused-MLP mlp_fused_kernel).  The roofline object is the kernel INSTANTIATION that takes the most time,
    # under the name rocprofv3 prints for it, so that its average launch duration can be checked against profiles/*_kernel_stats.csv;
    # the other instantiations and the family totals are listed beside it, all measured in the same pass.
    what = {"gemm_v2_kernel": "tiled LDS-DMA GEMM: projections, fc2, stage 4, neck, tracking path",
            "gemm_xs_kernel": "X-stationary short-K GEMM: QKV of stages 1-3, fc1 of stage 3",
            "mlp_fused_kernel": "fused fc1+GELU+fc2+residual of stages 1-2", "gemm_ks_kernel": "accumulator-stationary N=576 GEMM (opt-in)"}
    kern = {}
    for name, v in pred.engine.profile_read_kernels().items():
        if v["launches"] == 0 or v["ms"] <= 0:
            continue
        tf = v["flops"] / (v["ms"] * 1e-3) / 1e12
        gbps = v["bytes"] / (v["ms"] * 1e-3) / 1e9                       # ALGORITHMIC bytes / launch time
        ai = v["flops"] / max(v["bytes"], 1.0)
        # which roof is the lower one for this kernel's arithmetic intensity: min(peak MFMA, AI * peak HBM)
        bound = "hbm" if ai * PEAK_HBM_GBPS * 1e9 < PEAK_F16_TFLOPS * 1e12 else "mfma"
        kern[name] = {"bound": bound, "achieved": round(tf, 2), "frac": round(tf / PEAK_F16_TFLOPS, 4), "hbm_gbps": round(gbps, 1),
                      "hbm_frac": round(gbps / PEAK_HBM_GBPS, 4), "flop_per_byte": round(ai, 1), "launches": v["launches"],
                      "gflop_per_launch": round(v["flops"] / v["launches"] / 1e9, 3), "mb_per_launch": round(v["bytes"] / v["launches"] / 1e6, 1),
                      "avg_launch_us": round(v["ms"] * 1e3 / v["launches"], 2), "share_of_timed_region": round(v["ms"] * 1e-3 / dt, 3)}
    dom = max(kern, key=lambda k: kern[k]["share_of_timed_region"])
    fams = {}
    for fam, key in (("gemm_v2_kernel", "gemm"), ("gemm_xs_kernel", "xs"), ("mlp_fused_kernel", "mlp"), ("gemm_ks_kernel", "ks")):
        ms, fl, n = pr[f"{key}_ms"], pr[f"{key}_flops"], int(pr[f"{key}_launches"])
        if n:
            fams[fam] = {"what": what[fam], "achieved": round(fl / (ms * 1e-3) / 1e12, 2), "launches": n,
                         "share_of_timed_region": round(ms * 1e-3 / dt, 3)}
    tot_ms = sum(pr[f"{k}_ms"] for k in ("gemm", "xs", "mlp", "ks"))
    tot_fl = sum(pr[f"{k}_flops"] for k in ("gemm", "xs", "mlp", "ks"))
    # the dominant kernel against the roof that bounds it: "hbm" when its algorithmic intensity lies under the ridge
    # (312 FLOP/B) - achieved = algorithmic GB/s of 8,000 - else "mfma" - achieved = TFLOP/s of 2,500; the other fraction beside it
    d = kern[dom]
    hbm = d["bound"] == "hbm"
    roofline = {
        "bound": d["bound"], "kernel": dom, "achieved": d["hbm_gbps"] if hbm else d["achieved"], "peak": PEAK_HBM_GBPS if hbm else PEAK_F16_TFLOPS,
        "unit": "GB/s" if hbm else "TFLOP/s", "frac": d["hbm_frac"] if hbm else d["frac"], "traffic": _pmc_traffic(dom),
        "mfma": {"achieved": d["achieved"], "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s", "frac": d["frac"]},
        "hbm": {"achieved": d["hbm_gbps"], "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": d["hbm_frac"], "flop_per_byte": d["flop_per_byte"],
                "algorithmic_mb_per_launch": d["mb_per_launch"]},
        "launches": kern[dom]["launches"], "gflop_per_launch": kern[dom]["gflop_per_launch"], "avg_launch_us": kern[dom]["avg_launch_us"],
        "share_of_timed_region": kern[dom]["share_of_timed_region"],
        "kernels": dict(sorted(kern.items(), key=lambda kv: -kv[1]["share_of_timed_region"])),
        "families": fams,
        "all_gemm_kernels": {"achieved": round(tot_fl / (tot_ms * 1e-3) / 1e12, 2), "frac": round(tot_fl / (tot_ms * 1e-3) / 1e12 / PEAK_F16_TFLOPS, 4),
                             "share_of_timed_region": round(tot_ms * 1e-3 / dt, 3)},
        "attention_kernels": {"achieved": round(pr["attn_flops"] / max(pr["attn_ms"] * 1e-3, 1e-9) / 1e12, 2),
                              "launches": int(pr["attn_launches"]), "ms": round(pr["attn_ms"], 2)},
        "whole_path": {"algorithmic_gflop_per_step": CLIP_GFLOP_PROPAGATE * args.frames / 100.0,
                       "achieved_tflops": round(CLIP_GFLOP_PROPAGATE * args.frames / 100.0 * steps / 1e3 / dt, 2)},
    }
    roofline["whole_path"]["frac_of_mfma_peak"] = round(roofline["whole_path"]["achieved_tflops"] / PEAK_F16_TFLOPS, 4)

    return roofline


def parity_leg(args, sd, cfg, device, precision):
    """The 24-frame clip of tests/golden/large_video24_full.npz (seed 2, one click; recorded from the real reference by
    oracle/gen_golden.py) through the predictor in the bench configuration: worst-frame metrics over every low-res pixel."""
    from sam2_opt_amd.synthetic import normalize_frames, synthetic_frames_u8
    from sam2_opt_amd.video_predictor import SAM2VideoPredictor
    path = os.path.join(ROOT, "tests", "golden", "large_video24_full.npz")
    if not os.path.exists(path):
        return None
    g = np.load(path)
    T = int(g["num_frames"][0])
    frames = normalize_frames(synthetic_frames_u8(seed=2, num_frames=T), cfg)
    pred = SAM2VideoPredictor("large", state_dict=sd, encode_batch=args.encode_batch, device=device, overlap_encode=not args.no_overlap, precision=precision)
    try:
        st = pred.init_state(frames=frames, video_height=1024, video_width=1024)
        pred.add_new_points_or_box(st, 0, 1, points=np.array([[512.0, 512.0]], np.float32), labels=np.array([1], np.int32))
        worst = [0.0, 0.0, 0.0]
        for t, _ids, _vm in pred.propagate_in_video(st):
            od = st["output_dict_per_obj"][0]
            cur = od["cond_frame_outputs"].get(t) or od["non_cond_frame_outputs"][t]
            got, ref = cur["pred_masks"].float().cpu().numpy(), g[f"f{t}/pred_masks"]
            d = got - ref
            worst[0] = max(worst[0], float(np.abs(d).max() / np.abs(ref).max()))
            worst[1] = max(worst[1], float(np.linalg.norm(d) / np.linalg.norm(ref)))
            worst[2] = max(worst[2], float(((got > 0) != (ref > 0)).mean()))
        return {"vs": "real reference (fp32 torch backend), tests/golden/large_video24_full.npz", "frames": T, "pixels_per_frame": 65536,
                "max_abs_over_max_ref": float(f"{worst[0]:.3e}"), "rel_l2": float(f"{worst[1]:.3e}"), "binarised_disagreement": float(f"{worst[2]:.3e}"),
                "within_1e-3": bool(max(worst) <= 1e-3)}
    finally:
        pred.release()


PARITY_CLASS = {"f16s": "masks within 1e-3 of the reference fp32 path (north star)", "f16x3": "masks within 1e-3 of the reference fp32 path (north star)",
                "f16": "bf16-class tier: ~2e-3 of the reference fp32 path (outside the north-star 1e-3)"}


def mode_leg(args, sd, cfg, frames, device, precision, steps, with_roofline):
    """One more precision mode on the same clip and method: 1 warm-up + `steps` timed steps, then the same roofline pass and parity
    leg as the headline mode."""
    from sam2_opt_amd.video_predictor import SAM2VideoPredictor
    p2 = SAM2VideoPredictor("large", state_dict=sd, encode_batch=args.encode_batch, device=device, overlap_encode=not args.no_overlap, precision=precision,
                            prefetch_depth=args.prefetch_depth)
    try:
        st = p2.init_state(frames=frames, video_height=1024, video_width=1024)
        p2.add_new_points_or_box(st, 0, 1, points=np.array([[512.0, 512.0]], np.float32), labels=np.array([1], np.int32))

        def one_step():
            n = 0
            for _ in p2.propagate_in_video(st):
                n += 1
            return n, None
        one_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        for _ in range(steps):
            n += one_step()[0]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out = {"frames_per_s": round(n / dt, 2), "ms_per_frame": round(dt / n * 1e3, 3), "steps": steps, "dtype": "f16", "precision_mode": precision,
               "parity_class": PARITY_CLASS[precision]}
        if with_roofline:
            out["roofline"] = roofline_pass(args, p2, one_step, steps, dt)
        out["parity"] = parity_leg(args, sd, cfg, device, precision)
        return out
    finally:
        p2.release()


def secondary_legs(args, pred, sd, cfg, frames, device):
    """(a) the same clip in the other precision modes, (b) the drop-in route, (c) BASELINE.json configs[4]: 16 images x 8 point prompts."""
    from sam2_opt_amd.image_predictor import SAM2ImagePredictor
    out = {}
    for other in ("f16s", "f16", "f16x3"):
        if other != args.precision:
            out[other] = mode_leg(args, sd, cfg, frames, device, other, 2, with_roofline=(other != "f16x3") and not args.no_roofline)
    # (c) route A: the drop-in route - a torch host loop around the five plug-level entry points in the reference's tensor
    # layouts (sam2_opt_amd/route_a.py), what sam2_opt_amd.plugin.speedup_hip(reference_predictor) pays per frame
    from sam2_opt_amd.route_a import PlugLevelTracker
    nA = min(100, frames.shape[0])
    fa = frames[:nA].to(device)
    for prec in dict.fromkeys((args.precision, "f16")):             # the timed mode, and plain f16 beside it
        trk = PlugLevelTracker("large", state_dict=sd, device=device, precision=prec)
        try:
            for k in range(2):
                trk.start(fa, (512.0, 512.0))
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = sum(1 for _ in trk.propagate())
                torch.cuda.synchronize()
            key = "route_a_frames_per_s" if prec == args.precision else f"route_a_{prec}_frames_per_s"
            out[key] = round(n / (time.perf_counter() - t0), 2)
            if prec == args.precision:
                # BASELINE.json configs[1]: the image-encoder plug alone at batch 1 (what a frame costs where the look-ahead cannot apply)
                for _ in range(3):
                    trk.engine.image_encoder(fa[0:1].contiguous())
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    trk.engine.image_encoder(fa[0:1].contiguous())
                torch.cuda.synchronize()
                out["config1_image_encoder_batch1_ms"] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
        finally:
            trk.release()
    out["route_a_note"] = (f"plug-level route (image encoder, memory attention, prompt encoder, mask decoder, memory encoder called one by one "
                           f"per frame with NCHW / sequence-first fp32 tensors, torch glue and memory bank; the image plug looks 8 frames ahead over the clip tensor, "
                           f"plugin.LookaheadImagePlug): propagate loop over the first {nA} "
                           f"frames, precision={args.precision} (and f16 beside it); the headline `value` is the fused route (frame features and memory bank resident "
                           "in the engine, one C call per tracked frame)")
    B = 16
    ip = SAM2ImagePredictor("large", state_dict=sd, max_batch=B, device=device, precision=args.precision)
    try:
        imgs = [np.random.RandomState(10 + i).randint(0, 256, (1024, 1024, 3)).astype(np.uint8) for i in range(B)]
        pts = [(np.random.RandomState(100 + i).rand(8, 1, 2) * 1024).astype(np.float32) for i in range(B)]
        lab = np.ones((8, 1), np.int32)
        for k in range(3):
            if k == 1:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            ip.set_image_batch(imgs)
            for i in range(B):
                ip._predict(pts[i], lab, None, None, True, True, img_idx=i)
        torch.cuda.synchronize()
        out["config5_images_per_s"] = round(2 * B / (time.perf_counter() - t0), 2)
        out["config5_note"] = (f"BASELINE.json configs[4]: {B} synthetic 1024x1024 uint8 images per set_image_batch (host upload + "
                               f"normalise included), 8 single-point prompts each, multimask output, precision={args.precision}")
    finally:
        ip.release()
    return out


def cpu_baseline_leg(args, sd, cfg, frames):
    """The CPU oracle (a port of the reference's torch backend; the checker, here only timed) on the first frames of the same clip.
    Steady state = frames from index 8 on, where the memory bank holds 7 frames."""
    from oracle import sam2_ref as R
    ncpu = max(2, min(args.cpu_frames, args.frames))
    old_threads = torch.get_num_threads()
    torch.set_num_threads(max(1, min(args.cpu_threads, os.cpu_count() or 1)))
    cores = torch.get_num_threads()
    try:
        vo = R.VideoOracle(sd, cfg, frames[:ncpu].cpu())
        stamps = []
        with torch.inference_mode():
            vo.add_new_points(0, np.array([[512.0, 512.0]], np.float32), np.array([1], np.int32))
            t1 = time.perf_counter()
            for _t, _m in vo.propagate(max_frames=ncpu - 1):
                stamps.append(time.perf_counter())
        total = stamps[-1] - t1
        first_steady = 8 if len(stamps) > 9 else 1                       # stamps[i] = end of frame i
        steady = (len(stamps) - first_steady) / (stamps[-1] - stamps[first_steady - 1])
        return {"value": round(steady, 4), "unit": "frames/s", "cores": cores, "kind": "port",
                "sample": f"oracle propagate loop over frames 0..{len(stamps) - 1} of the same clip ({total:.1f} s, fp32 torch CPU kernels, "
                          f"{cores} threads); value = frames {first_steady}..{len(stamps) - 1} only (memory bank full, L = 7)",
                "whole_sample_frames_per_s": round(len(stamps) / total, 4)}
    finally:
        torch.set_num_threads(old_threads)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--encode-batch", type=int, default=8)
    ap.add_argument("--prefetch-depth", type=int, default=2, help="encoder batches the prefetch stream may run ahead of the tracking")
    ap.add_argument("--no-overlap", action="store_true", help="encode and track on one stream (no encoder prefetch stream)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=16, help="frames of the clip timed on the CPU oracle (steady state = frames 8..)")
    ap.add_argument("--cpu-threads", type=int, default=32, help="torch CPU threads of the baseline leg (reported as `cores`)")
    ap.add_argument("--precision", default="f16s", choices=("f16", "f16x3", "f16s"),
                    help="precision mode of the timed run (default: the fastest one inside the north-star 1e-3 bar)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the parity leg, the other precision modes, the drop-in route and config 5")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"))
    ap.add_argument("--stub-predictor", action="store_true", help="CPU stand-in predictor (orchestration tests only; never a result)")
    args = ap.parse_args()

    from sam2_opt_amd import dist as D
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started without a launcher: become one.  N fresh child processes (one per LOCAL_RANK, RANK / WORLD_SIZE / MASTER_* in
        # their environment, the same command line), started BEFORE anything in this process touches a GPU; rank 0's JSON line
        # passes through on the inherited stdout.
        return D.self_spawn(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:])
    rank, local_rank, world = D.rank_world()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.stub_predictor:
        return stub_main(args, rank, world)
    if args.backend != "nccl":
        raise SystemExit("--backend gloo is only for --stub-predictor runs")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    dist = D.init("nccl", device)          # RCCL over xGMI; None when N == 1

    from sam2_opt_amd.config import get_config
    from sam2_opt_amd.synthetic import normalize_frames, synthetic_frames_u8
    from sam2_opt_amd.video_predictor import SAM2VideoPredictor
    from sam2_opt_amd.weights import synthetic_state_dict

    cfg = get_config("large")
    sd = synthetic_state_dict(cfg, seed=0)
    pred = SAM2VideoPredictor("large", state_dict=sd, encode_batch=args.encode_batch, device=device, overlap_encode=not args.no_overlap,
                              precision=args.precision, prefetch_depth=args.prefetch_depth)
    frames_u8 = synthetic_frames_u8(seed=D.clip_seed_for_rank(2, rank), num_frames=args.frames)
    frames = normalize_frames(frames_u8, cfg)
    state = pred.init_state(frames=frames, video_height=1024, video_width=1024)       # frames resident in HBM from here on
    pred.add_new_points_or_box(state, 0, 1, points=np.array([[512.0, 512.0]], np.float32), labels=np.array([1], np.int32))

    # optional: drive the tracking path from a high-priority stream (the caller's current stream is what the predictor
    # launches on, like the reference); tuning knob, off by default
    if os.environ.get("SAM2MI_TRACK_PRIORITY"):
        hp = torch.cuda.Stream(device=device, priority=int(os.environ["SAM2MI_TRACK_PRIORITY"]))
        hp.wait_stream(torch.cuda.current_stream(device))
        torch.cuda.set_stream(hp)

    def one_step():
        n, chk = 0, None
        for _, _, masks in pred.propagate_in_video(state):
            n += 1
            chk = masks
        return n, chk

    dt, total_frames, records, nframes, last = timed_steps(D, dist, device, one_step, args.warmup, args.steps)
    checksum = records[rank][2]
    value = total_frames / dt
    devices = D.gather_strings(dist, f"{torch.cuda.get_device_name(device)} (cuda:{local_rank})")
    rccl_ranks = dist.get_world_size() if dist is not None else 1

    roofline = None
    if not args.no_roofline and rank == 0:
        roofline = roofline_pass(args, pred, one_step, args.steps, dt)
    parity = parity_leg(args, sd, cfg, device, args.precision) if (rank == 0 and world == 1 and not args.no_secondary) else None

    secondary = None
    if not args.no_secondary and rank == 0 and world == 1:
        secondary = secondary_legs(args, pred, sd, cfg, frames, device)

    cpu_baseline = None
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        cpu_baseline = cpu_baseline_leg(args, sd, cfg, frames)

    if rank == 0:
        out = {
            "metric": "frames/sec SAM2.1-hiera-large 1024x1024 video propagation", "value": round(value, 3), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "precision_mode": args.precision, "parity_class": PARITY_CLASS[args.precision], "data": "synthetic",
            "config": {"workload": f"config3: {args.frames}-frame 1024x1024 clip per GPU, 1 click, 1 object, SAM2.1-hiera-large "
                                   "(random-init weights), propagate_in_video loop", "frames_per_step": args.frames,
                       "encode_batch": args.encode_batch, "overlap_encode_stream": not args.no_overlap, "parallelism": f"clips x{world} (one process per GPU)",
                       "rccl_ranks": rccl_ranks, "devices": devices, "clip_seeds": [D.clip_seed_for_rank(2, r) for r in range(world)],
                       "ms_per_frame": round(dt / max(nframes, 1) * 1e3, 3), "mask_checksum": round(checksum, 6), "per_rank": records},
            "roofline": roofline, "parity": parity, "cpu_baseline": cpu_baseline, "secondary": secondary,
            "library": {"source_hash": _lib_hash()},
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
