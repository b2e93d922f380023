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
SOURCES = ["gemm.hip", "gemm2.hip", "gemm_rowln.hip", "gemm_projln.hip", "mlp_fused.hip", "gemm_xs.hip", "attn_hiera.hip", "attn_flash256.hip", "attn_precise.hip", "attn_small.hip",
           "elementwise.hip", "convs.hip", "heads.hip", "postproc.hip", "resize.hip", "hiera_generic.hip", "engine_core.hip", "engine_encoder.hip", "engine_track.hip",
           "engine_abi.hip", "engine_debug.hip"]
# kernels that were measured and lost (DESIGN.md "GEMM tuning log"): only built with SAM2MI_EXPERIMENTAL=1 in the environment,
# which also defines -DSAM2MI_EXPERIMENTAL (extra tile instantiations, tile_hint dispatch, the SAM2MI_KS / SAM2MI_FLASH_V2 switches)
EXPERIMENTAL_SOURCES = ["gemm3.hip", "gemm4.hip", "gemm_ks.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result"]
if os.environ.get("SAM2MI_EXPERIMENTAL"):
    SOURCES = SOURCES + EXPERIMENTAL_SOURCES
    FLAGS = FLAGS + ["-DSAM2MI_EXPERIMENTAL"]
HASHFILE = LIB + ".srchash"


def source_hash() -> str:
    """SHA-256 over every source, header and the compile flags: what the library on disk must have been built from."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    files = [os.path.join(CSRC, s) for s in SOURCES] + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    files.append(os.path.join(os.path.dirname(HERE), "include", "sam2mi.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def built_hash() -> str:
    try:
        with open(HASHFILE) as f:
            return f.read().strip()
    except OSError:
        return ""


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


def _file_hash(path: str) -> str:
    import hashlib
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def _build_locked(force: bool, verbose: bool) -> str:
    """Rebuild decisions never look at mtimes (a copy / checkout need not preserve them): the library is current iff the
    hash stored beside it equals the hash of the sources, and an object file is current iff the hash of its own source +
    all headers + flags, recorded in build/objhash.json, still matches."""
    import json
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    want = source_hash()
    if not force and os.path.exists(LIB) and built_hash() == want:
        return LIB                      # prebuilt library matches the sources byte for byte (e.g. on the GPU box)
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    headers.append(os.path.join(os.path.dirname(HERE), "include", "sam2mi.h"))
    common = " ".join(FLAGS) + "".join(_file_hash(h) for h in headers)
    rec_path = os.path.join(OBJ, "objhash.json")
    try:
        with open(rec_path) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        rec = {}
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        key = _file_hash(s) + common
        if force or not os.path.exists(o) or rec.get(src) != key:
            jobs.append((s, o, src, key))

    def cc(job):
        s, o, _, _ = job
        r = subprocess.run([hipcc, *FLAGS, "-c", s, "-o", o], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{r.stderr[-4000:]}")
        return o

    if jobs:
        if verbose:
            print(f"[sam2mi] compiling {len(jobs)} file(s) for gfx950", flush=True)
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(cc, jobs))
        for _, _, src, key in jobs:
            rec[src] = key
        with open(rec_path, "w") as f:
            json.dump(rec, f)
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    if verbose:
        print(f"[sam2mi] linked {LIB}", flush=True)
    with open(HASHFILE, "w") as f:
        f.write(want + "\n")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
