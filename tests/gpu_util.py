import torch


def err(a: torch.Tensor, b: torch.Tensor):
    """(max-abs error / max|ref|, relative L2) of a vs reference b."""
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    d = (a - b)
    return (d.abs().max() / b.abs().max().clamp_min(1e-12)).item(), (d.norm() / b.norm().clamp_min(1e-12)).item()


def check(name, a, b, max_rel, l2_rel):
    m, l = err(a, b)
    print(f"[parity] {name}: max_rel={m:.3e} l2_rel={l:.3e} (tol {max_rel:.1e}/{l2_rel:.1e})", flush=True)
    assert m <= max_rel and l <= l2_rel, f"{name}: max_rel {m:.3e} (tol {max_rel}) l2_rel {l:.3e} (tol {l2_rel})"
