"""Tail-effect probe for the 128x192 GEMM tile (fc2 of stage 3, N = 576, K = 2304): time per launch over row counts that give
510 / 768 / 1020 / 1536 tiles on the 512 workgroup slots of the chip (2 per CU), and the other production tiles on the same shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam2_opt_amd.native import Engine
eng = Engine("large", state_dict=None)
for name, N, K in (("s3 fc2", 576, 2304), ("s4 fc2", 1152, 4608)):
    for M in (128 * 128, 128 * 256, 128 * 512):
        if M * max(N, K) > 8 * 65536 * 576:
            continue
        row = f"{name} M={M:6d} N={N} K={K}:"
        for h in (0, 16, 10, 13, 18):      # 18: the 256x288 tile of SAM2MI_EXPERIMENTAL builds (n/a otherwise)
            try:
                ms = eng.debug_gemm_bench(M, N, K, 10, 1 | (h << 4))
            except RuntimeError:
                row += f"  h{h}: n/a"
                continue
            row += f"  h{h}: {ms * 1e3:7.1f}us {2.0 * M * N * K / ms / 1e9:6.1f}TF"
        print(row, flush=True)
