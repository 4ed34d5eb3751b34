import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def rel_l2(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="session")
def golden():
    def load(name):
        path = os.path.join(ROOT, "tests", "golden", f"{name}.npz")
        z = np.load(path, allow_pickle=False)
        case = json.loads(str(z["case"]))
        return case, z
    return load
