"""Pins the C oracle (oracle/vits_oracle.c) to fixtures generated from the reference's
own PyTorch graph definition (oracle/gen_golden.py, run in the build container)."""
import os

import numpy as np
import pytest

from conftest import ALL_PRESETS, GOLDEN, case_get, golden_cases
from vits_oracle import VitsOracle, conv1d, conv_transpose1d

STAGES = ("x", "m_p", "logs_p", "logw", "z_p", "z", "output")


@pytest.mark.parametrize("preset", ALL_PRESETS)
def test_oracle_matches_reference_goldens(preset):
    o = VitsOracle(os.path.join(GOLDEN, preset + ".onnx"))
    g = np.load(os.path.join(GOLDEN, preset + ".npz"))
    for c in golden_cases(g):
        r = o.infer(case_get(g, c, "ids"), case_get(g, c, "lens"), case_get(g, c, "scales"),
                    case_get(g, c, "sid"), case_get(g, c, "noise_dp"), case_get(g, c, "noise_z"))
        # durations are integers: bit-exact (SURVEY §7 "hard parts": compare w_ceil first)
        assert np.array_equal(r["w_ceil"], case_get(g, c, "out_w_ceil")), (preset, c)
        assert np.array_equal(r["y_lengths"], case_get(g, c, "out_y_lengths")), (preset, c)
        for k in STAGES:
            ref = case_get(g, c, "out_" + k)
            assert r[k].shape == ref.shape, (preset, c, k, r[k].shape, ref.shape)
            # fp32 tolerance: 1e-4 on intermediates, north_star's 1e-3 on the waveform
            np.testing.assert_allclose(r[k], ref, atol=1e-4 if k != "output" else 1e-5, rtol=0,
                                       err_msg=f"{preset}/{c}/{k}")


def test_output_rank_and_length():
    # known-answer facts: rank 4 [B,1,1,S], S = prod(upsample_rates) * max y_len
    o = VitsOracle(os.path.join(GOLDEN, "tiny_rb1.onnx"))
    g = np.load(os.path.join(GOLDEN, "tiny_rb1.npz"))
    r = o.infer(case_get(g, "b3_noise", "ids"), case_get(g, "b3_noise", "lens"), [0, 1.3, 0])
    assert r["output"].ndim == 4 and r["output"].shape[1:3] == (1, 1)
    assert r["output"].shape[3] == 4 * 4 * 2 * 2 * int(r["y_lengths"].max())
    # determinism at zero noise
    r2 = o.infer(case_get(g, "b3_noise", "ids"), case_get(g, "b3_noise", "lens"), [0, 1.3, 0])
    assert np.array_equal(r["output"], r2["output"])


def test_embedding_lookup_bit_exact():
    # first encoder op: emb[id] * sqrt(H) in fp32, masked (models.py:199): the oracle's "emb" tap must be the
    # NumPy gather bit for bit (north_star: integer phoneme id -> embedding lookup bit-exact)
    o = VitsOracle(os.path.join(GOLDEN, "tiny_dp.onnx"))
    emb = o.tensors["enc_p.emb.weight"]
    assert np.array_equal(emb, o.model.init["enc_p.emb.weight"])   # the resolved table IS the initializer
    V, H = emb.shape
    ids = np.stack([np.arange(16), np.arange(V - 16, V)]).astype(np.int64)
    lens = np.array([16, 9], np.int64)
    r = o.infer(ids, lens, [0, 1, 0])
    want = (emb[ids] * np.float32(np.sqrt(H))).transpose(0, 2, 1) * (np.arange(16)[None, None, :] < lens[:, None, None])
    assert r["emb"].shape == (2, H, 16)
    assert np.array_equal(r["emb"], want.astype(np.float32))


def test_conv_primitives_against_numpy():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 5, 37)).astype(np.float32)
    w = rng.standard_normal((7, 5, 3)).astype(np.float32)
    b = rng.standard_normal(7).astype(np.float32)
    y = conv1d(x, w, b, dil=2, pad_l=2, pad_r=2)
    ref = np.zeros((2, 7, 37))
    xp = np.pad(x, ((0, 0), (0, 0), (2, 2))).astype(np.float64)
    for k in range(3):
        ref += np.einsum("oc,bct->bot", w[:, :, k].astype(np.float64), xp[:, :, 2 * k:2 * k + 37])
    ref += b[None, :, None]
    np.testing.assert_allclose(y, ref, atol=1e-5)
    # transposed conv vs direct definition
    wt = rng.standard_normal((5, 4, 8)).astype(np.float32)
    bt = rng.standard_normal(4).astype(np.float32)
    yt = conv_transpose1d(x, wt, bt, stride=4, pad=2)
    To = (37 - 1) * 4 - 4 + 8
    ref = np.zeros((2, 4, To))
    for i in range(37):
        for k in range(8):
            t = i * 4 - 2 + k
            if 0 <= t < To:
                ref[:, :, t] += np.einsum("bc,co->bo", x[:, :, i].astype(np.float64), wt[:, :, k].astype(np.float64))
    ref += bt[None, :, None]
    np.testing.assert_allclose(yt, ref, atol=1e-5)


def test_bad_inputs_raise():
    o = VitsOracle(os.path.join(GOLDEN, "tiny_rb2_ms.onnx"))
    ids = np.zeros((1, 4), np.int64)
    with pytest.raises(RuntimeError):  # multi-speaker graph without sid (models.py:693)
        o.infer(ids, [4], [0, 1, 0])
    with pytest.raises(RuntimeError):
        o.infer(ids + 10_000, [4], [0, 1, 0], sid=[0])
