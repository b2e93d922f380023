"""CPU: the N > 1 path of bench.py (clip sharding + MAX-time reduction + result gather) with world_size 2 on gloo."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from sam2_opt_amd import dist as D
    dist = D.init("gloo")
    assert dist is not None and dist.get_world_size() == world
    clips = D.shard_clips(5, rank, world)
    frames = 100 * len(clips)
    seconds = 1.0 + rank            # rank 1 is the slow one
    D.barrier(dist, torch.device("cpu"))
    tmax, total, recs = D.reduce_time_and_gather(dist, frames, seconds, 0.25 * (rank + 1), torch.device("cpu"))
    q.put((rank, clips, D.clip_seed_for_rank(2, rank), tmax, total, recs))
    dist.destroy_process_group()


def test_two_rank_sharding_and_gather():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, c0, s0, t0, n0, rec0), (r1, c1, s1, t1, n1, rec1) = out
    assert sorted(c0 + c1) == [0, 1, 2, 3, 4] and not set(c0) & set(c1)          # every clip exactly once
    assert (s0, s1) == (2, 3)
    assert t0 == t1 == 2.0                                                         # MAX over ranks
    assert n0 == n1 == 500                                                         # whole-job frame count
    assert rec0 == rec1 == [(300, 1.0, 0.25), (200, 2.0, 0.5)]


def test_single_process_passthrough():
    from sam2_opt_amd import dist as D
    os.environ.pop("WORLD_SIZE", None)
    assert D.init("gloo") is None
    assert D.reduce_time_and_gather(None, 100, 0.5, 0.1, torch.device("cpu")) == (0.5, 100, [(100, 0.5, 0.1)])
