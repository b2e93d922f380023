"""d=256 flash attention timing on the GPU (tuning aid): self-attention and cross-attention sizes of the memory attention."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam2_opt_amd.native import Engine
eng = Engine("large", state_dict=None)
for name, Nq, Nk in [("self", 4096, 4096), ("cross L=1", 4096, 4100), ("cross L=4", 4096, 4 * 4096 + 32), ("cross L=7", 4096, 7 * 4096 + 64)]:
    ms = eng.debug_flash_bench(Nq, Nk, 10)
    print(f"{name:10s} Nq={Nq} Nk={Nk:6d}: {ms*1e3:8.1f} us  {4.0*Nq*Nk*256/ms/1e9:7.1f} TF", flush=True)
