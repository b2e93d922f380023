"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/sam2mi.h declares.
No compute is called (there is no GPU in the build container)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "sam2mi.h")).read()
    return sorted(set(re.findall(r"\b(sam2mi_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_header_symbols():
    from sam2_opt_amd.build import build
    lib_path = build(verbose=False)
    assert os.path.exists(lib_path)
    lib = ctypes.CDLL(lib_path)
    names = _declared()
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in sam2mi.h but not exported: {missing}"
    assert lib.sam2mi_abi_version() == 2


def test_python_binding_lists_every_export():
    from sam2_opt_amd import native
    assert sorted(native.EXPORTS) == _declared()


def test_engine_fails_loudly_without_gpu():
    import pytest
    import torch
    from sam2_opt_amd import native
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        native.Engine("large")
