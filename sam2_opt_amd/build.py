"""Builds libsam2mi.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m sam2_opt_amd.build [--force]
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libsam2mi.so")
SOURCES = ["gemm.hip", "gemm2.hip", "gemm3.hip", "gemm4.hip", "mlp_fused.hip", "gemm_xs.hip", "gemm_ks.hip", "attn_hiera.hip", "attn_flash256.hip", "attn_small.hip", "elementwise.hip", "convs.hip",
           "heads.hip", "postproc.hip", "engine_core.hip", "engine_encoder.hip", "engine_track.hip", "engine_abi.hip", "engine_debug.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result"]


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force: bool = False, verbose: bool = True) -> str:
    """Build (or reuse) the in-tree library.  Serialised across processes with a file lock: the ranks of a multi-GPU
    launch call this concurrently and must not compile into the same object files at once."""
    import fcntl
    os.makedirs(OBJ, exist_ok=True)
    with open(os.path.join(OBJ, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force: bool, verbose: bool) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "sam2mi.h"))
    hdr_time = max(os.path.getmtime(h) for h in headers)
    src_time = max(os.path.getmtime(os.path.join(CSRC, s)) for s in SOURCES)
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= max(hdr_time, src_time):
        return LIB                      # prebuilt library is current (e.g. on the GPU box)
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _newer(s, o) or hdr_time > os.path.getmtime(o):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        r = subprocess.run([hipcc, *FLAGS, "-c", s, "-o", o], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{r.stderr[-4000:]}")
        return o

    if jobs:
        if verbose:
            print(f"[sam2mi] compiling {len(jobs)} file(s) for gfx950", flush=True)
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or not os.path.exists(LIB):
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
        if verbose:
            print(f"[sam2mi] linked {LIB}", flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
