"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path.

`-m "not gpu"`: oracle vs fixtures / fp64 restatement / closed forms, host logic, C-ABI export check.
`-m gpu`     : parity tests proper — HIP path (through the C-ABI) vs the oracle on a real MI355X.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def syn():
    import kwave_amd  # noqa: F401  (repo-root shim)
    from kwave_amd import synthetic
    return synthetic


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle


def rel_l2(a, b):
    import numpy as np
    a = np.asarray(a)
    b = np.asarray(b)
    dt = np.complex128 if (np.iscomplexobj(a) or np.iscomplexobj(b)) else np.float64
    a, b = a.astype(dt), b.astype(dt)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
