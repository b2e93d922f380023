import os, sys
sys.path.insert(0, "/root/repo")
import torch
from sam2_opt_amd.native import Engine
eng = Engine("large", state_dict=None)
def run(tag):
    ms1 = eng.debug_gemm_bench(32768, 2304, 576, 10, 0 | (30 << 4))
    ms2 = eng.debug_gemm_bench(32768, 576, 2304, 10, 1)
    print(f"{tag}: xs fc1 {ms1*1e3:.1f} us, tiled fc2 {ms2*1e3:.1f} us", flush=True)
run("default stream")
for r in (0, 1, 2, 8, 16):
    st = eng.create_reserved_stream(r)
    with torch.cuda.stream(st):
        run(f"masked stream reserve={r}")
