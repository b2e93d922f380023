import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test")


def pytest_report_header(config):
    """Ties a test log to a code state: the hash of the sources libsam2mi.so was built from (sam2_opt_amd/build.py) and whether it
    still matches the sources in the tree."""
    try:
        from sam2_opt_amd.build import built_hash, source_hash
        b, s_ = built_hash(), source_hash()
        return f"libsam2mi.so built from source hash {b[:16] or 'n/a'} ({'current' if b == s_ else 'STALE: tree is ' + s_[:16]})"
    except Exception as e:                       # never block a test run on the header
        return f"libsam2mi.so source hash unavailable: {e}"


@pytest.fixture(scope="session")
def cfg_large():
    from sam2_opt_amd.config import get_config
    return get_config("large")


@pytest.fixture(scope="session")
def sd_large(cfg_large):
    """Synthetic hiera-large weights, seed 0 (the seed every golden vector was made with)."""
    from sam2_opt_amd.weights import synthetic_state_dict
    return synthetic_state_dict(cfg_large, seed=0)


@pytest.fixture(scope="session")
def golden_plugs():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "large_plugs.npz"))


@pytest.fixture(scope="session")
def golden_video():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "large_video24.npz"))
