import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

GOLDEN = os.path.join(ROOT, "tests", "golden")
TINY_PRESETS = ("tiny_rb1", "tiny_rb2_ms", "tiny_dp")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def golden_cases(npz):
    return sorted(set(k.split("/")[0] for k in npz.files))


def case_get(npz, case, key):
    k = f"{case}/{key}"
    return npz[k] if k in npz.files else None
