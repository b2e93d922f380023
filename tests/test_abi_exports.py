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


def test_state_dict_spec_is_strict(monkeypatch):
    """Engine.load_state_dict validates names and shapes against the configured architecture before anything is uploaded
    (ADVICE r01): checked here on the CPU by calling the method on a bare object."""
    import pytest
    import torch
    from sam2_opt_amd import native
    from sam2_opt_amd.config import get_config
    from sam2_opt_amd.weights import state_dict_spec
    eng = object.__new__(native.Engine)
    eng.cfg = get_config("large")
    spec = state_dict_spec(eng.cfg)
    sd = {k: torch.zeros(1).expand(*shape) if len(shape) else torch.zeros(()) for k, shape in spec.items()}
    k0 = "image_encoder.trunk.blocks.9.attn.qkv.weight"
    bad = dict(sd)
    bad[k0] = torch.zeros(1).expand(1728, 288)
    with pytest.raises(RuntimeError, match="shape mismatches"):
        native.Engine.load_state_dict(eng, bad)
    bad = dict(sd)
    del bad[k0]
    bad["something.else"] = torch.zeros(1)
    with pytest.raises(RuntimeError, match="1 missing.*1 unexpected"):
        native.Engine.load_state_dict(eng, bad)
    tiny = {k: torch.zeros(1).expand(*shape) for k, shape in state_dict_spec(get_config("tiny")).items()}
    with pytest.raises(RuntimeError, match="does not match"):
        native.Engine.load_state_dict(eng, tiny)
