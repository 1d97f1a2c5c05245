import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Load a committed fixture as {key: torch tensor / numpy array}."""
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        out = {}
        for k in z.files:
            a = z[k]
            out[k] = torch.from_numpy(a) if a.dtype.kind == "f" and a.ndim > 0 else a
        return out


def rel_l2(a, b):
    """||a-b|| / max(||b||, tiny) in float64."""
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.fixture(scope="session")
def golden():
    return load_golden
