"""pytest configuration: `gpu` marker, deterministic seeds, repo root on sys.path."""

import os
import random
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _seed_everything():
    # same policy as the reference's tests/conftest.py:8-48: fixed seed before every test
    random.seed(42)
    np.random.seed(42)
    torch.manual_seed(42)
    yield


@pytest.fixture(autouse=True)
def _fresh_pattern_cache():
    """Every test starts with an empty pattern cache: the cache recognises index tensors by CONTENT and keeps the last patterns whose
    tensors have died adoptable (`_pattern._RECENT`), so plans built under one test's switches would otherwise be adopted by the next
    test that builds the same matrix."""
    try:
        from torchsparsegradutils_amd import _pattern
    except Exception:  # noqa: BLE001  (tests of the oracle alone do not need the package)
        yield
        return
    _pattern.clear_cache()
    yield
