"""GEMM shape sweep on the GPU: TFLOP/s of the production GEMM per model shape and per forced tile (tuning aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam2_opt_amd.native import Engine
B = int(os.environ.get("B", "8"))
MODE = int(os.environ.get("MODE", "-1"))
HINTS = [int(x) for x in os.environ.get("HINTS", "0").split(",")]
shapes = [
    ("s1 qkv", B * 65536, 432, 144, 0), ("s1 proj", B * 65536, 144, 144, 1), ("s1 fc1", B * 65536, 576, 144, 0), ("s1 fc2", B * 65536, 144, 576, 1),
    ("s2 qkv", B * 16384, 864, 288, 0), ("s2 proj", B * 16384, 288, 288, 1), ("s2 fc1", B * 16384, 1152, 288, 0), ("s2 fc2", B * 16384, 288, 1152, 1),
    ("s3 qkv", B * 4096, 1728, 576, 0), ("s3 proj", B * 4096, 576, 576, 1), ("s3 fc1", B * 4096, 2304, 576, 0), ("s3 fc2", B * 4096, 576, 2304, 1),
    ("s4 qkv", B * 1024, 3456, 1152, 0), ("s4 proj", B * 1024, 1152, 1152, 1), ("s4 fc1", B * 1024, 4608, 1152, 0), ("s4 fc2", B * 1024, 1152, 4608, 1),
    ("patch", B * 65536, 144, 160, 1), ("neck0", B * 65536, 256, 144, 1), ("conv_s0", B * 65536, 32, 256, 1),
    ("ma qkv", 4096, 768, 256, 0), ("ma out", 4096, 256, 256, 1), ("ma ff1", 4096, 2048, 256, 0), ("ma ff2", 4096, 256, 2048, 1),
    ("ma kall", 28736, 1024, 64, 0), ("dec kv", 4096, 128, 256, 1), ("me pw1", 4096, 1024, 256, 0), ("me pw2", 4096, 256, 1024, 1),
]
eng = Engine("large", state_dict=None)
for name, M, N, K, mode in shapes:
    row = f"{name:8s} M={M:7d} N={N:5d} K={K:5d} m{mode}:"
    for h in HINTS:
        try:
            ms = eng.debug_gemm_bench(M, N, K, 10, (MODE if MODE >= 0 else mode) | (h << 4))
        except RuntimeError:
            row += f"  h{h}:      n/a"
            continue
        row += f"  h{h}: {ms*1e3:8.1f}us {2.0*M*N*K/ms/1e9:7.1f}TF"
    print(row, flush=True)
