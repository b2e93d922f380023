"""one cross-attention sized flash256 launch set (for PMC counter runs)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam2_opt_amd.native import Engine
eng = Engine("large", state_dict=None)
ms = eng.debug_flash_bench(4096, 7 * 4096 + 64, 4)
print(ms)
