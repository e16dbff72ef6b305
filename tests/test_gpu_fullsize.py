"""Full-size parity on the GPU box: synthetic "medium" / "high" / multi-speaker voices
(phoonnx_amd/synth.py writes .onnx files with the exporter's structure), HIP path vs the C
oracle on identical inputs and injected noise, plus size-independent properties at the
BASELINE batch size."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CACHE = os.environ.get("VITSMI_BENCH_CACHE", "/tmp/vitsmi_bench")


def _voice(preset, **over):
    from phoonnx_amd.synth import write_voice
    tag = preset + "".join(f"_{k}{v}" for k, v in sorted(over.items()))
    tag = "".join(ch if ch.isalnum() or ch in "_-" else "" for ch in tag)
    path = os.path.join(CACHE, f"synth_{tag}.onnx")
    if not os.path.exists(path):
        os.makedirs(CACHE, exist_ok=True)
        write_voice(path + ".tmp", preset, seed=1234, **over)
        os.replace(path + ".tmp", path)
    return path


@pytest.mark.parametrize("preset,over,B,T,precision",
                         [("medium", {}, 3, 96, None), ("high", {}, 2, 64, None),
                          ("medium", {"n_speakers": 4}, 2, 80, None),
                          # 64 -> 128 -> 64 -> 32 channels: raw-format z / conv_pre input and stages
                          ("small", {"upsample_initial_channel": 128, "upsample_rates": (8, 4),
                                     "upsample_kernel_sizes": (16, 8)}, 3, 70, None),
                          # the six-product exact arithmetic (the default is f16x3)
                          ("medium", {}, 3, 96, "bf16x6"), ("high", {}, 2, 64, "bf16x6")])
def test_fullsize_pipeline_matches_oracle(monkeypatch, preset, over, B, T, precision):
    from phoonnx_amd import MiSession
    from vits_oracle import VitsOracle
    path = _voice(preset, **over)
    if precision:
        monkeypatch.setenv("VITSMI_GEN_PRECISION", precision)
    s, o = MiSession(path), VitsOracle(path)
    assert s.hparam("gen_nprod") == (6 if precision == "bf16x6" else 2)
    rng = np.random.default_rng(99)
    lens = np.array([T] + [int(x) for x in rng.integers(T // 3, T, B - 1)], np.int64)
    ids = np.zeros((B, T), np.int64)
    for b in range(B):
        ids[b, :lens[b]] = rng.integers(0, 256, lens[b])
    sid = rng.integers(0, 4, B).astype(np.int64) if over.get("n_speakers", 1) > 1 else None
    scales = np.array([0.667, 1.4, 0.8], np.float32)
    ndp = rng.standard_normal((B, 2, T)).astype(np.float32)
    nz = rng.standard_normal((B, s.hparam("inter"), T * 8)).astype(np.float32)
    ref = o.infer(ids, lens, scales, sid, ndp, nz)
    got = s.synthesize_batch(ids, lens, scales, sid, ndp, nz, taps=("x", "m_p", "logs_p", "logw", "w_ceil", "z_p", "z"))
    assert np.array_equal(got["w_ceil"], ref["w_ceil"])          # integer durations: exact
    assert np.array_equal(got["y_lengths"], ref["y_lengths"])
    for k in ("x", "m_p", "logs_p", "logw", "z_p", "z"):
        np.testing.assert_allclose(got[k], ref[k], atol=5e-4, rtol=0, err_msg=k)
    assert got["output"].shape == ref["output"].shape
    err = np.abs(got["output"] - ref["output"]).max()
    print(f"{preset} {precision or 'f16x3'}: waveform max-abs error vs oracle {err:.3g}")
    assert err < 1e-3, err                                        # north_star tolerance
    assert 0.02 < np.abs(ref["output"]).max() < 0.999             # the comparison is not vacuous
    s.close()


def test_generator_engines_agree(monkeypatch):
    """The same voice through the split-operand generator (default f16x3 arithmetic) and (VITSMI_GEN_ENGINE=f32)
    through the f32-MFMA generator: two independent implementations of the same fp32 arithmetic must agree to
    rounding."""
    from phoonnx_amd import MiSession
    path = _voice("medium")
    rng = np.random.default_rng(5)
    B, T = 4, 128
    ids = rng.integers(0, 256, (B, T)).astype(np.int64)
    lens = np.array([T, T - 17, T // 2, T - 1], np.int64)
    sc = np.array([0.667, 1.3, 0.8], np.float32)
    ndp = rng.standard_normal((B, 2, T)).astype(np.float32)
    nz = rng.standard_normal((B, 192, T * 8)).astype(np.float32)
    s1 = MiSession(path)
    assert s1.hparam("gen_sx") == 1
    a = s1.synthesize_batch(ids, lens, sc, None, ndp, nz)
    s1.close()
    monkeypatch.setenv("VITSMI_GEN_ENGINE", "f32")
    s2 = MiSession(path)
    assert s2.hparam("gen_sx") == 0
    b = s2.synthesize_batch(ids, lens, sc, None, ndp, nz)
    s2.close()
    assert np.array_equal(a["y_lengths"], b["y_lengths"])
    assert np.abs(a["output"]).max() > 0.02
    np.testing.assert_allclose(a["output"], b["output"], atol=2e-5, rtol=0)


@pytest.mark.parametrize("preset,B,T", [("medium", 4, 128), ("high", 2, 96)])
def test_generator_f16_mode_agrees_with_exact_mode(monkeypatch, preset, B, T):
    """The default generator arithmetic (f16x3: operands as two fp16 planes, three MFMA products per fp32 product,
    weights scaled per tensor) against VITSMI_GEN_PRECISION=bf16x6 (six bf16 plane products, every product exact):
    the same waveform to fp32 rounding noise."""
    from phoonnx_amd import MiSession
    path = _voice(preset)
    rng = np.random.default_rng(6)
    ids = rng.integers(0, 256, (B, T)).astype(np.int64)
    lens = np.array([T] + [int(v) for v in rng.integers(T // 2, T, B - 1)], np.int64)
    sc = np.array([0.667, 1.3, 0.8], np.float32)
    ndp = rng.standard_normal((B, 2, T)).astype(np.float32)
    nz = rng.standard_normal((B, 192, T * 8)).astype(np.float32)
    s1 = MiSession(path)
    assert s1.hparam("gen_nprod") == 2
    b = s1.synthesize_batch(ids, lens, sc, None, ndp, nz)
    s1.close()
    monkeypatch.setenv("VITSMI_GEN_PRECISION", "bf16x6")
    s2 = MiSession(path)
    assert s2.hparam("gen_nprod") == 6
    a = s2.synthesize_batch(ids, lens, sc, None, ndp, nz)
    s2.close()
    assert np.array_equal(a["y_lengths"], b["y_lengths"])
    assert np.abs(a["output"]).max() > 0.02
    err = float(np.abs(a["output"] - b["output"]).max())
    print(f"{preset}: f16x3 vs exact max-abs {err:.3g}")
    assert err < 2e-5, err


@pytest.mark.parametrize("mode,tol", [("bf16x3", 1e-3), ("bf16", 8e-2)])
def test_config4_multispeaker_mixed_lengths_reduced_precision_vocoder(monkeypatch, mode, tol):
    """BASELINE config 4: multi-speaker voice, mixed-length padded batch, reduced-precision vocoder
    (VITSMI_GEN_PRECISION).  Everything up to z is computed exactly as always; the generator then uses three
    (bf16x3) or one (bf16) bf16 plane product per fp32 product.  Declared waveform tolerances: bf16x3 stays inside
    north_star's 1e-3; plain bf16 is a quality/speed trade-off, 8e-2 max-abs on a unit-scale waveform."""
    from phoonnx_amd import MiSession
    from vits_oracle import VitsOracle
    path = _voice("medium", n_speakers=4)
    monkeypatch.setenv("VITSMI_GEN_PRECISION", mode)
    s, o = MiSession(path), VitsOracle(path)
    assert s.hparam("gen_nprod") == {"bf16x3": 3, "bf16": 1}[mode]
    rng = np.random.default_rng(4)
    B, T = 8, 72
    lens = np.array([T] + [int(v) for v in rng.integers(T // 4, T, B - 1)], np.int64)
    ids = np.zeros((B, T), np.int64)
    for b in range(B):
        ids[b, :lens[b]] = rng.integers(0, 256, lens[b])
    sid = rng.integers(0, 4, B).astype(np.int64)
    scales = np.array([0.667, 1.2, 0.8], np.float32)
    ndp = rng.standard_normal((B, 2, T)).astype(np.float32)
    nz = rng.standard_normal((B, 192, T * 8)).astype(np.float32)
    ref = o.infer(ids, lens, scales, sid, ndp, nz)
    got = s.synthesize_batch(ids, lens, scales, sid, ndp, nz, taps=("z",))
    assert np.array_equal(got["y_lengths"], ref["y_lengths"])
    np.testing.assert_allclose(got["z"], ref["z"], atol=5e-4, rtol=0)      # exact part of the pipeline
    hop = s.hparam("hop")
    err = max(float(np.abs(got["output"][b, 0, 0, :int(ref["y_lengths"][b]) * hop] -
                           ref["output"][b, 0, 0, :int(ref["y_lengths"][b]) * hop]).max()) for b in range(B))
    assert err < tol, (mode, err)
    s.close()


def test_baseline_batch_properties():
    """B=32 x 256 ids (BASELINE config 3) is too slow for the CPU oracle inside a test, so check
    size-independent properties: batch-composition invariance of durations, shape law
    S = hop * max(y_len), finiteness, |audio| <= 1, and agreement of item 0 with its batch-1 run."""
    from phoonnx_amd import MiSession
    path = _voice("medium")
    s = MiSession(path)
    rng = np.random.default_rng(1234)
    B, T = 32, 256
    ids = rng.integers(0, 256, (B, T)).astype(np.int64)
    lens = np.full(B, T, np.int64)
    sc = np.array([0, 1.5, 0], np.float32)
    r = s.synthesize_batch(ids, lens, sc, taps=("w_ceil",))
    hop = s.hparam("hop")
    assert r["output"].shape == (B, 1, 1, hop * int(r["y_lengths"].max()))
    assert np.isfinite(r["output"]).all() and np.abs(r["output"]).max() <= 1.0
    assert np.array_equal(r["w_ceil"].sum(1).astype(np.int64), r["y_lengths"])
    one = s.synthesize_batch(ids[:1], lens[:1], sc, taps=("w_ceil",))
    assert np.array_equal(one["w_ceil"][0], r["w_ceil"][0])
    n = (int(one["y_lengths"][0]) - 64) * hop  # minus the generator's receptive field at the right edge
    np.testing.assert_allclose(r["output"][0, 0, 0, :n], one["output"][0, 0, 0, :n], atol=2e-5)
    s.close()
