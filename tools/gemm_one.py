"""Run one GEMM shape/tile repeatedly (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam2_opt_amd.native import Engine
M, N, K, mode = (int(x) for x in sys.argv[1:5])
hints = [int(x) for x in sys.argv[5].split(",")]
eng = Engine("large", state_dict=None)
for h in hints:
    ms = eng.debug_gemm_bench(M, N, K, 5, mode | (h << 4))
    print(f"hint {h}: {ms*1e3:.1f} us {2.0*M*N*K/ms/1e9:.1f} TF", flush=True)
