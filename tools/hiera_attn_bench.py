"""Hiera attention timing through the debug entry (tuning aid): stage-3 windowed and global shapes, B frames."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sam2_opt_amd.native import Engine
eng = Engine("large", state_dict=None)
import inspect
print(inspect.signature(eng.debug_hiera_attention))
