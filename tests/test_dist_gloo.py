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


def test_bench_orchestration_two_ranks_gloo():
    """bench.py's own N > 1 code path - launched exactly as the driver launches it (torch.distributed.run, one process per
    rank, RANK / WORLD_SIZE / MASTER_* from the environment) - with the gloo backend and a stand-in predictor: process-group
    init, W warm-up + K timed steps between barriers, MAX-over-ranks time, the per-rank gather and the single JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--frames", "10", "--backend", "gloo", "--stub-predictor"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                                        # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak" and out["data"] == "stub"
    recs = out["config"]["per_rank"]
    assert [r_[0] for r_ in recs] == [30, 30]                               # K * frames per rank
    slow = max(r_[1] for r_ in recs)
    assert abs(out["ms_per_step"] * 3 / 1e3 - slow) < 0.05                  # the time is the MAX over ranks ...
    assert abs(out["value"] - 60 / slow) / out["value"] < 0.05              # ... and the value the whole-job aggregate over it
    # the timed region ends with a barrier, so BOTH ranks measure the pace of the slow one (rank 1 sleeps 2 x 20 ms per step)
    assert min(r_[1] for r_ in recs) >= 3 * 0.04 * 0.95
    assert out["config"]["clip_seed_rank0"] == 2 and recs[0][2] == recs[1][2] == 1.0


def test_bench_self_spawns_without_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment: bench.py starts the two ranks itself (fresh child
    processes, before anything touches a GPU) and the result is the same single JSON line as under torch.distributed.run."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--frames", "10",
                        "--backend", "gloo", "--stub-predictor"], capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["rccl_ranks"] == 2 and len(out["config"]["devices"]) == 2
    assert [r_[0] for r_ in out["config"]["per_rank"]] == [20, 20]
    pids = {d.split("pid ")[1].rstrip(")") for d in out["config"]["devices"]}
    assert len(pids) == 2                                                   # two processes, not two threads


def test_bench_self_spawn_propagates_a_failing_rank():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    # --backend gloo without --stub-predictor is refused by every rank: the launcher must come back non-zero, promptly
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo"], capture_output=True, text=True, timeout=120,
                       env=env, cwd=root)
    assert r.returncode != 0


def test_bench_refuses_mismatched_world():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["WORLD_SIZE"] = "1"                  # a launcher's environment that disagrees with --gpus
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--stub-predictor", "--backend", "gloo"],
                       capture_output=True, text=True, timeout=120, env=env, cwd=root)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
