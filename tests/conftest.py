"""Shared test plumbing.

`-m "not gpu"` tests: the oracle against the committed golden vectors, host logic, C-ABI symbol export.
`-m gpu` tests: parity of the HIP path (through the C-ABI) against the oracle on the same seeded inputs.
Nothing here reads /root/reference at run time (it does not exist on the GPU box).
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "sc-a-loam_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tools", "synth"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun)")


@pytest.fixture(scope="session")
def O():
    import oracle_py
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def S():
    import scaloam
    scaloam.lib()
    return scaloam


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


@pytest.fixture(scope="session")
def worlds():
    import scansynth
    cache = {}

    def get(sensor, seed):
        key = (sensor, seed)
        if key not in cache:
            cache[key] = scansynth.World(sensor, seed)
        return cache[key]
    return get


@pytest.fixture(scope="session")
def hdl64_stream(O, worlds):
    """First scans of the seeded HDL-64 sequence (BASELINE config #2), generated once per session."""
    w = worlds(O.HDL64, 205)
    scans = {}

    def get(k):
        if k not in scans:
            scans[k] = w.scan(k)
        return scans[k]
    return get
