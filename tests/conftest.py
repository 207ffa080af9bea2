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
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def golden_state_dict(g):
    """fp32 state_dict from the committed fp16 weights of a fixture."""
    return {k[3:]: torch.from_numpy(v.astype(np.float32)) for k, v in g.items() if k.startswith("sd:")}


@pytest.fixture(scope="session")
def tiny():
    return load_golden("tiny_clip.npz")


@pytest.fixture(scope="session")
def tiny3():
    return load_golden("tiny3_clip.npz")


def has_gpu():
    return torch.cuda.is_available()
