"""Fused Hiera MLP kernel vs the two-GEMM path on the stage-1/2 shapes (tuning aid): us per launch and TFLOP/s."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sam2_opt_amd.native import Engine
B = int(os.environ.get("B", "8"))
eng = Engine("large", state_dict=None)
for name, M, C in [("s1", B * 65536, 144), ("s2", B * 16384, 288)]:
    g = torch.Generator(device="cpu").manual_seed(1)
    xn = torch.randn(M, C, generator=g).cuda()
    W1 = (torch.randn(4 * C, C, generator=g) / math.sqrt(C)).cuda()
    b1 = torch.randn(4 * C, generator=g).cuda()
    W2 = (torch.randn(C, 4 * C, generator=g) / math.sqrt(4 * C)).cuda()
    b2 = torch.randn(C, generator=g).cuda()
    x = torch.randn(M, C, generator=g).cuda()
    fl = 2.0 * 2.0 * M * C * 4 * C
    row = f"{name} M={M} C={C}:"
    for fused in (1, 0):
        _, ms = eng.debug_mlp(xn, W1, b1, W2, b2, x, fused=bool(fused), iters=10)
        row += f"  {'fused' if fused else '2gemm'} {ms*1e3:8.1f}us {fl/ms/1e9:7.1f}TF"
    print(row, flush=True)
