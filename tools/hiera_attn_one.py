"""Stage-3 windowed and global Hiera attention launches through the debug entry (for rocprofv3 --pmc / --kernel-trace runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sam2_opt_amd.native import Engine
eng = Engine("large", state_dict=None)
g = torch.Generator(device="cpu").manual_seed(0)
for name, groups, GQ in (("windowed", 128, 256), ("global", 8, 4096)):
    C = 8 * 72
    q = (torch.randn(groups * GQ, C, generator=g) * 1.5).cuda()
    k = (torch.randn(groups * GQ, C, generator=g) * 1.5).cuda()
    v = torch.randn(groups * GQ, C, generator=g).cuda()
    for _ in range(3):
        eng.debug_hiera_attention(q, k, v, groups, 8, GQ, GQ, GQ, GQ)
    torch.cuda.synchronize()
    print(name, "done", flush=True)
