"""Pins the torch-CPU restatement that bench.py times as its `cpu_baseline` (oracle/torch_baseline.py, written from
SURVEY App. A) to the reference-generated fixtures: the number bench reports beside the GPU value is the speed of
something that computes the reference's outputs."""
import os

import numpy as np
import pytest

from conftest import ALL_PRESETS, GOLDEN, case_get, golden_cases


@pytest.mark.parametrize("preset", ALL_PRESETS)
def test_torch_baseline_matches_reference_goldens(preset):
    from torch_baseline import TorchVits
    m = TorchVits(os.path.join(GOLDEN, preset + ".onnx"))
    g = np.load(os.path.join(GOLDEN, preset + ".npz"))
    for c in golden_cases(g):
        r = m.infer(case_get(g, c, "ids"), case_get(g, c, "lens"), case_get(g, c, "scales"), case_get(g, c, "sid"),
                    case_get(g, c, "noise_dp"), case_get(g, c, "noise_z"))
        assert np.array_equal(r["w_ceil"], case_get(g, c, "out_w_ceil")), (preset, c)
        assert np.array_equal(r["y_lengths"], case_get(g, c, "out_y_lengths")), (preset, c)
        for k in ("x", "m_p", "logs_p", "logw", "z_p", "z"):
            np.testing.assert_allclose(r[k], case_get(g, c, "out_" + k), atol=1e-4, rtol=0, err_msg=f"{preset}/{c}/{k}")
        np.testing.assert_allclose(r["output"], case_get(g, c, "out_output"), atol=1e-5, rtol=0)
