import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def rel_l2(ref, x):
    import torch
    ref = torch.as_tensor(ref).detach().double().cpu()
    x = torch.as_tensor(x).detach().double().cpu()
    return float((ref - x).norm() / (ref.norm() + 1e-30))


def maxdiff(a, b):
    import torch
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max())
