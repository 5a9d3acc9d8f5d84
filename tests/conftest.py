"""pytest configuration: `gpu` marker, import paths, golden-fixture helpers."""

import hashlib
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "radio-core_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

# Parity tolerance of BASELINE.json's north star: max|delta| <= 1e-4 * max|ref|.
TOL = 1e-4


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def rel_err(a, ref):
    """max|a - ref| / max|ref| (the north-star metric); shapes must match."""
    a = np.asarray(a)
    ref = np.asarray(ref)
    assert a.shape == ref.shape, (a.shape, ref.shape)
    peak = float(np.max(np.abs(ref)))
    return float(np.max(np.abs(a - ref))) / (peak if peak > 0 else 1.0)


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


class Golden:
    """Lazy access to tests/golden/<name>.npz with input-checksum checking."""

    def __init__(self, name):
        self._z = np.load(os.path.join(HERE, "golden", name + ".npz"))

    def __getitem__(self, key):
        return self._z[key]

    def __contains__(self, key):
        return key in self._z.files

    def check_input(self, key, x):
        want = str(self._z[key])
        got = digest(x)
        assert got == want, "synthetic input drifted for %s: %s != %s" % (key, got, want)


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return get


def have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
