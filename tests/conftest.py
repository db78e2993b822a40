import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # torch's CPU kernels regress badly with hundreds of threads (the GPU box has 256): the oracle
    # runs 10x faster with a moderate count
    import torch
    torch.set_num_threads(min(16, os.cpu_count() or 1))


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    return load


def expand_masks(row0, n):
    """[nM, W] uint8 row-0 -> [nM, n, W] int64 one-hot masks (constant down columns)."""
    import torch
    m = torch.from_numpy(row0.astype(np.int64))
    return m[:, None, :].expand(m.shape[0], n, m.shape[1]).contiguous()
