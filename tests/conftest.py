import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than ~30 s on CPU")


def load_golden(name, raw=False):
    import torch

    z = np.load(os.path.join(GOLD, name + ".npz"))
    if raw:  # numpy arrays as stored (string entries hold JSON)
        return {k: z[k] for k in z.files}
    return {k: torch.from_numpy(z[k]) for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden
