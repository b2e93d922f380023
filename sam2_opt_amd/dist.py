"""Clip-sharded multi-GPU execution: one process per GPU, independent clips, no data-path collective.

The video path shards by CLIP (frames inside a clip are strictly sequential), so the only
communication is a barrier, the MAX of the per-rank wall times and one all-gather of a 3-number result record.
Backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU tests.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch


def rank_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend: str, device: Optional[torch.device] = None):
    """Initialise torch.distributed when WORLD_SIZE > 1; returns the module or None."""
    _, _, world = rank_world()
    if world <= 1:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend=backend, **kw)
    return dist


def clip_seed_for_rank(base_seed: int, rank: int) -> int:
    """Rank r tracks the synthetic clip with seed base+r (BASELINE.json configs[3])."""
    return base_seed + rank


def shard_clips(num_clips: int, rank: int, world: int) -> List[int]:
    """Static round-robin assignment of clip ids to ranks (every clip exactly once)."""
    return list(range(rank, num_clips, world))


def barrier(dist, device: Optional[torch.device] = None):
    if dist is not None:
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def reduce_time_and_gather(dist, frames: int, seconds: float, checksum: float, device: torch.device):
    """Returns (max_seconds, total_frames, per_rank_records).  per_rank_records: list of (frames, seconds, checksum)."""
    if dist is None:
        return seconds, frames, [(frames, seconds, checksum)]
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    mine = torch.tensor([float(frames), seconds, checksum], dtype=torch.float64, device=device)
    allr = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(allr, mine)
    recs = [(int(x[0].item()), float(x[1].item()), float(x[2].item())) for x in allr]
    return float(t.item()), sum(r[0] for r in recs), recs


def gather_strings(dist, mine: str) -> List[str]:
    """One string per rank (device names for the result record), on every rank."""
    if dist is None:
        return [mine]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, mine)
    return [str(x) for x in out]


def self_spawn(world: int, argv: List[str]) -> int:
    """Launcher of last resort (bench.py --gpus N started without torch.distributed.run): start `world` child processes running
    `argv`, child r with RANK = LOCAL_RANK = r, WORLD_SIZE and a fresh MASTER_ADDR / MASTER_PORT on 127.0.0.1; wait for all of them;
    return the largest exit code (a child that dies takes the others down instead of leaving them in a rendezvous).
    The calling process must not have touched a GPU: the children are fresh interpreters, nothing is inherited but the environment."""
    import socket
    import subprocess
    import time
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen(argv, env=env))
    code = 0
    alive = list(procs)
    while alive:
        time.sleep(0.2)
        for p in list(alive):
            rc = p.poll()
            if rc is None:
                continue
            alive.remove(p)
            code = max(code, abs(rc))
            if rc != 0:                      # one rank failed: the others would wait in a collective forever
                for q in alive:
                    q.terminate()
    return code
