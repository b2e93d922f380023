import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test")


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
