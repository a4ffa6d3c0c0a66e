import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def synth_frames(seed, nz, ny, nx, sparsity, depth=12, dark_lo=80, dark_hi=120):
    """SURVEY §8d synthetic distribution (host form): Bernoulli events above a per-pixel dark level."""
    rng = np.random.default_rng(seed)
    dark = rng.integers(dark_lo, dark_hi + 1, (ny, nx)).astype(np.uint16)
    top = min((1 << depth) - 1 - dark_hi, 2047)
    frames = np.empty((nz, ny, nx), np.uint16)
    for z in range(nz):
        mask = rng.random((ny, nx)) < sparsity
        amp = rng.integers(1, top + 1, (ny, nx)).astype(np.uint16)
        below = np.floor(rng.random((ny, nx)) * (dark.astype(np.float64) + 1)).astype(np.uint16)
        frames[z] = np.where(mask, dark + amp, below)
    return dark, frames
