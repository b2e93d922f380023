"""Compact golden-vector format shared by oracle/gen_golden.py and the tests.

A tensor is pinned by (a) a strided sample of its flattened values (stride = a prime,
so it does not alias with tensor dims) and (b) float64 sum / abs-sum / square-sum over
ALL elements.  TEST INFRASTRUCTURE.
"""
from __future__ import annotations

import numpy as np
import torch


def _prime_at_least(n: int) -> int:
    n = max(2, int(n))
    while True:
        if all(n % d for d in range(2, int(n ** 0.5) + 1)):
            return n
        n += 1


def pack(store: dict, name: str, t, target: int = 16384):
    a = t.detach().to(torch.float32).cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t, np.float32)
    flat = a.reshape(-1)
    stride = 1 if flat.size <= target else _prime_at_least(flat.size / target)
    store[name + "/sample"] = flat[::stride].copy()
    store[name + "/meta"] = np.array([stride, flat.size], dtype=np.int64)
    f64 = flat.astype(np.float64)
    store[name + "/stats"] = np.array([f64.sum(), np.abs(f64).sum(), (f64 * f64).sum()], dtype=np.float64)
    store[name + "/shape"] = np.array(a.shape, dtype=np.int64)


def compare(store, name: str, t, atol: float, rtol: float = 0.0, stat_rtol: float = 1e-4, outlier_frac: float = 0.0):
    """Returns (ok, message).  `t` is the oracle's full tensor."""
    a = t.detach().to(torch.float32).cpu().numpy().reshape(-1)
    stride, size = (int(v) for v in store[name + "/meta"])
    if a.size != size:
        return False, f"{name}: size {a.size} != golden {size}"
    ref = store[name + "/sample"]
    got = a[::stride]
    err = np.abs(got - ref)
    tol = atol + rtol * np.abs(ref)
    bad = int((err > tol).sum())
    if bad > outlier_frac * err.size:
        i = int(np.argmax(err - tol))
        return False, (f"{name}: {bad}/{err.size} samples beyond tolerance; max abs err {err.max():.3e} at sample {i} "
                       f"(ref {ref[i]:.6f} got {got[i]:.6f}), atol {atol}")
    if outlier_frac > 0:
        # discontinuous quantities (binarised masks, bf16 rounding): only the bulk is comparable
        l2 = float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-12))
        return (l2 <= 50 * (rtol + atol)), f"{name}: ok with {bad} outliers (max err {err.max():.2e}, rel L2 {l2:.2e})"
    f64 = a.astype(np.float64)
    stats = np.array([f64.sum(), np.abs(f64).sum(), (f64 * f64).sum()])
    g = store[name + "/stats"]
    # sums of N elements each within atol: allow N*atol drift on the linear sums
    lin_tol = size * atol + stat_rtol * abs(g[1])
    if abs(stats[0] - g[0]) > lin_tol or abs(stats[1] - g[1]) > lin_tol:
        return False, f"{name}: checksum mismatch sum {stats[0]:.6e} vs {g[0]:.6e}, abs {stats[1]:.6e} vs {g[1]:.6e}"
    if abs(stats[2] - g[2]) > stat_rtol * abs(g[2]) + size * atol * atol + 2 * atol * g[1]:
        return False, f"{name}: square-sum mismatch {stats[2]:.6e} vs {g[2]:.6e}"
    return True, f"{name}: ok (max err {err.max():.2e})"
