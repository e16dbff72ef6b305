import os
import sys

# The C oracle (oracle/libvits_oracle.so) is OpenMP code: on the GPU box's 256 hardware threads its default team turns the
# tiny fixtures' loops into barrier traffic (one tiny_rb1 rendering: 34 s there, 0.03 s with 8 threads) - and the GPU suite's
# time was mostly that.  Set before anything loads libgomp; an explicit OMP_NUM_THREADS in the environment wins.
os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(16, os.cpu_count() or 1))))

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

GOLDEN = os.path.join(ROOT, "tests", "golden")
TINY_PRESETS = ("tiny_rb1", "tiny_rb2_ms", "tiny_dp")
# reference-generated voices whose generator channel counts are all multiples of 32: these run on conv_sx_kernel,
# the split-operand matrix-core engine of every full-size voice (vits_hparam "gen_sx" == 1)
SX_PRESETS = ("sx_rb1", "sx_rb2_ms")
ALL_PRESETS = TINY_PRESETS + SX_PRESETS


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


def zero_tails(ref, y_lengths, hop):
    """The reference's padded rendering [B,1,1,S] with every sample behind an utterance's end (y_lengths[b] * hop) set to
    zero: what the engine returns by default (MiSession(tails="zero"): those samples are not rendered; the exported graph
    fills them with its unmasked generator's response to zeros, models.py:348-368 / SURVEY 0.9)."""
    import numpy as np
    out = np.array(ref, copy=True)
    for b in range(out.shape[0]):
        out[b, ..., int(y_lengths[b]) * hop:] = 0
    return out
