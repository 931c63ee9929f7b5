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


def bits_to_bf16(a):
    """uint16 bit patterns (how the fixtures store bf16) -> torch.bfloat16."""
    return torch.from_numpy(np.asarray(a).astype(np.int16)).view(torch.bfloat16)


def bf16_bits(t):
    return (t.detach().cpu().contiguous().view(torch.int16).to(torch.int32) & 0xFFFF).numpy().astype(np.uint16)


def golden_cases(kind):
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz") and kind in f)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def cuda():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
