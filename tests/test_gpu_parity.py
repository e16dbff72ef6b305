"""GPU parity tests (run with -m gpu on an MI355X).  Everything goes through the C ABI
(libvitsmi.so); the C oracle and the committed reference fixtures are the checkers.

Tolerances: integer/index work (durations, frame counts, output shape) bit-exact; fp32
intermediates 2e-4 max-abs; the waveform 1e-3 max-abs (north_star) — in practice ~1e-5.
"""
import os

import numpy as np
import pytest

from conftest import ALL_PRESETS, GOLDEN, ROOT, SX_PRESETS, TINY_PRESETS, case_get, golden_cases, zero_tails

pytestmark = pytest.mark.gpu

WAVE_TOL = 1e-3   # BASELINE.json north_star: max-abs on the final fp32 waveform
STAGE_TOL = 2e-4


def _session(preset, tails=None):
    from phoonnx_amd import MiSession
    return MiSession(os.path.join(GOLDEN, preset + ".onnx"), tails=tails)


def _ref_wave(g, c, s, tails):
    """The reference fixture's waveform as the session's tails mode returns it: the graph's padded rendering
    ("reference") or that with the samples behind each utterance's end zeroed ("zero", the default)."""
    ref = case_get(g, c, "out_output")
    return ref if tails == "reference" else zero_tails(ref, case_get(g, c, "out_y_lengths"), s.hparam("hop"))


# ------------------------------------------------------------------ kernel level: conv engine

CONV_CASES = [
    # (B, Cin, Cout, T, K, dil)                      tile config exercised
    (2, 32, 32, 300, 3, 1),     # cfg0 32x512
    (1, 4, 4, 77, 7, 3),        # tiny channels, padded M and K
    (2, 29, 16, 65, 1, 1),      # odd Cin
    (1, 16, 1, 130, 7, 1),      # Cout = 1
    (2, 64, 64, 513, 11, 5),    # cfg1 64x256, widest receptive field of the "high" preset
    (1, 96, 192, 40, 1, 1),     # flow pre
    (1, 192, 96, 33, 1, 1),     # flow post (Cout not multiple of 64)
    (2, 192, 384, 200, 5, 1),   # cfg2 128x128, WN in_layer
    (1, 128, 128, 700, 7, 12),  # medium preset's widest dilation
    (1, 192, 576, 19, 1, 1),    # fused qkv
    (3, 48, 200, 257, 3, 2),    # ragged everything
]


@pytest.mark.parametrize("hint", [0, 1, 2])
@pytest.mark.parametrize("B,Cin,Cout,T,K,dil", CONV_CASES)
def test_conv_engine_matches_oracle(B, Cin, Cout, T, K, dil, hint):
    # hint selects the tile family (generator / flow / token domain); T % 4 != 0 cases take the
    # 4-byte LDS-DMA path, T % 4 == 0 the 16-byte one
    from phoonnx_amd.session import test_conv1d
    from vits_oracle import conv1d
    rng = np.random.default_rng(B * 1000 + Cin + Cout + T)
    x = rng.standard_normal((B, Cin, T)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, K)) / np.sqrt(Cin * K)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    pad = dil * (K - 1) // 2
    got = test_conv1d(x, w, b, dil=dil, pad_l=pad, hint=hint)
    ref = conv1d(x, w, b, dil=dil, pad_l=pad, pad_r=dil * (K - 1) - pad)
    np.testing.assert_allclose(got, ref, atol=2e-5, rtol=1e-5)
    # fused leaky-relu prologue + relu epilogue
    got = test_conv1d(x, w, None, dil=dil, pad_l=pad, lrelu_slope=0.1, relu=True, hint=hint)
    xa = np.where(x > 0, x, x * np.float32(0.1)).astype(np.float32)
    ref = np.maximum(conv1d(xa, w, None, dil=dil, pad_l=pad, pad_r=dil * (K - 1) - pad), 0)
    np.testing.assert_allclose(got, ref, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("T", [128, 1000, 1024, 4100])
def test_conv_engine_long_rows_many_chunks(T):
    # several channel chunks (double-buffered LDS-DMA pipeline), wide halo, left pad not a multiple of 4
    from phoonnx_amd.session import test_conv1d
    from vits_oracle import conv1d
    rng = np.random.default_rng(T)
    x = rng.standard_normal((2, 128, T)).astype(np.float32)
    w = (rng.standard_normal((128, 128, 11)) / np.sqrt(128 * 11)).astype(np.float32)
    for dil in (1, 5):
        pad = dil * 5
        got = test_conv1d(x, w, None, dil=dil, pad_l=pad)
        ref = conv1d(x, w, None, dil=dil, pad_l=pad, pad_r=pad)
        np.testing.assert_allclose(got, ref, atol=3e-5, rtol=1e-5)


def test_conv_engine_identity_asymmetric():
    # A = I check with an asymmetric operand (catches transposed C/D maps)
    from phoonnx_amd.session import test_conv1d
    C, T = 64, 96
    w = np.zeros((C, C, 1), np.float32)
    w[np.arange(C), np.arange(C), 0] = 1
    x = (np.arange(C)[:, None] * 1000 + np.arange(T)[None, :]).astype(np.float32)[None]
    got = test_conv1d(x, w)
    assert np.array_equal(got, x)


@pytest.mark.parametrize("B,Cin,Cout,T,K,u", [(2, 32, 16, 50, 16, 8), (1, 64, 32, 37, 8, 4), (2, 16, 8, 129, 4, 2),
                                              (1, 512, 256, 9, 16, 8), (1, 6, 3, 20, 6, 2)])
def test_conv_transpose_matches_oracle(B, Cin, Cout, T, K, u):
    from phoonnx_amd.session import test_conv_transpose1d
    from vits_oracle import conv_transpose1d
    rng = np.random.default_rng(K * 100 + u)
    x = rng.standard_normal((B, Cin, T)).astype(np.float32)
    w = (rng.standard_normal((Cin, Cout, K)) / np.sqrt(Cin * K / u)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    got = test_conv_transpose1d(x, w, b, u)
    ref = conv_transpose1d(x, w, b, u, (K - u) // 2)
    assert got.shape == ref.shape == (B, Cout, T * u)
    np.testing.assert_allclose(got, ref, atol=2e-5, rtol=1e-5)


# ------------------------------------------------------------------ kernel level: split-exact bf16 engine

SX_CASES = [
    # (B, Cin, Cout, T, K, dil)                      sx tile config exercised
    (2, 32, 32, 300, 3, 1),     # 32x256, two chunks
    (1, 64, 64, 513, 11, 5),    # 64x256, widest receptive field of the "high" preset
    (2, 128, 128, 700, 7, 3),   # 128x128
    (1, 192, 512, 45, 7, 1),    # conv_pre of the "high" preset
    (1, 16, 96, 260, 5, 2),     # single chunk, Cout % 64 != 0
    (1, 256, 128, 130, 3, 1),   # sixteen chunks
    (3, 48, 32, 257, 1, 1),     # k = 1, ragged T
    (1, 64, 64, 5, 7, 1),       # sequence shorter than the kernel (raw-input path, everything is halo)
    (2, 128, 128, 31, 3, 1),    # sequence much shorter than the 256-column tile (plane-input path)
    (1, 128, 128, 400, 11, 13), # receptive field wider than any VITS layer: 7 DMA rounds per x tile, the general DMA path
    # round 3, the 16x16x32 main loop (f16x3, plane input, Cin % 32 == 0): its tile shapes and pipeline corners
    (2, 192, 384, 700, 1, 1),   # 1 x 1 conv taken as it is: every step opens a chunk (the flow's res_skip conv)
    (1, 192, 192, 260, 1, 1),   # ... on 64-row tiles
    (2, 256, 256, 530, 11, 5),  # widest halo of the headline voice: x stage rows of 320 cells, 2 x 80 KiB of LDS per CU
    (1, 96, 160, 300, 5, 2),    # three 32-channel chunks, Cout % 64 != 0 (32-row tiles)
]


@pytest.mark.parametrize("B,Cin,Cout,T,K,dil", SX_CASES)
def test_conv_sx_engine_matches_oracle(B, Cin, Cout, T, K, dil):
    from phoonnx_amd.session import test_conv1d_sx
    from vits_oracle import conv1d
    rng = np.random.default_rng(B * 1000 + Cin + Cout + T)
    x = rng.standard_normal((B, Cin, T)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, K)) / np.sqrt(Cin * K)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    pad = dil * (K - 1) // 2
    ref = conv1d(x, w, b, dil=dil, pad_l=pad, pad_r=dil * (K - 1) - pad)
    got = test_conv1d_sx(x, w, b, dil=dil, pad_l=pad)                       # fp32 raw output
    np.testing.assert_allclose(got, ref, atol=2e-5, rtol=1e-5)
    got = test_conv1d_sx(x, w, b, dil=dil, pad_l=pad, planes_slope=0.1)     # bf16 planes of leaky_relu(out)
    np.testing.assert_allclose(got, np.where(ref > 0, ref, ref * np.float32(0.1)), atol=2e-5, rtol=1e-5)
    if Cin == Cout:                                                          # residual epilogue
        got = test_conv1d_sx(x, w, b, dil=dil, pad_l=pad, residual=True)
        np.testing.assert_allclose(got, ref + x, atol=2e-5, rtol=1e-5)
    if Cin <= 64:   # raw-input kernels (fp32 tile split into planes in the kernel): leaky_relu on the way in
        xa = np.where(x > 0, x, x * np.float32(0.1)).astype(np.float32)
        ref = conv1d(xa, w, b, dil=dil, pad_l=pad, pad_r=dil * (K - 1) - pad)
        got = test_conv1d_sx(x, w, b, dil=dil, pad_l=pad, in_slope=0.1, residual=Cin == Cout)
        np.testing.assert_allclose(got, ref + (x if Cin == Cout else 0), atol=2e-5, rtol=1e-5)


def test_conv_sx_engine_is_exact_on_identity():
    # W = I: every product is (1.0 x plane), so the three planes must reassemble each fp32 input bit for bit
    # (exactness of the split, the operand layouts and the C/D map in one check)
    from phoonnx_amd.session import test_conv1d_sx
    C, T = 64, 301
    rng = np.random.default_rng(7)
    w = np.zeros((C, C, 1), np.float32)
    w[np.arange(C), np.arange(C), 0] = 1
    x = (rng.standard_normal((2, C, T)) * np.exp(rng.uniform(-20, 20, (2, C, T)))).astype(np.float32)
    assert np.array_equal(test_conv1d_sx(x, w), x)
    assert np.array_equal(test_conv1d_sx(x, w, planes_slope=1.0), x)


def test_conv_sx_engine_error_is_fp32_grade():
    # against float64: the split-exact engine must be as accurate as the fp32-FMA engine (it drops only
    # terms below 2^-24 relative), not bf16-grade
    from phoonnx_amd.session import test_conv1d, test_conv1d_sx
    rng = np.random.default_rng(11)
    B, C, T, K, dil = 1, 128, 640, 7, 3
    x = rng.standard_normal((B, C, T)).astype(np.float32)
    w = (rng.standard_normal((C, C, K)) / np.sqrt(C * K)).astype(np.float32)
    pad = dil * (K - 1) // 2
    xp = np.pad(x.astype(np.float64), ((0, 0), (0, 0), (pad, pad)))
    ref = sum(np.einsum("oc,bct->bot", w[:, :, k].astype(np.float64), xp[:, :, k * dil:k * dil + T]) for k in range(K))
    e_sx = np.abs(test_conv1d_sx(x, w, dil=dil, pad_l=pad) - ref).max()
    e_f32 = np.abs(test_conv1d(x, w, dil=dil, pad_l=pad) - ref).max()
    assert e_sx <= 2 * e_f32 + 1e-7, (e_sx, e_f32)
    assert e_sx < 5e-6


def test_conv_sx_reduced_precision_mode_has_its_declared_error():
    # the optional vocoder mode (VITSMI_GEN_PRECISION=f16, BASELINE config 4): one fp16 plane per operand (11 significant
    # bits each), one product, fp32 accumulation; measured against float64 on unit-variance outputs
    from phoonnx_amd.session import test_conv1d_sx
    rng = np.random.default_rng(21)
    B, C, T, K, dil = 1, 128, 512, 7, 1
    x = rng.standard_normal((B, C, T)).astype(np.float32)
    w = (rng.standard_normal((C, C, K)) / np.sqrt(C * K)).astype(np.float32)
    xp = np.pad(x.astype(np.float64), ((0, 0), (0, 0), (3, 3)))
    ref = sum(np.einsum("oc,bct->bot", w[:, :, k].astype(np.float64), xp[:, :, k:k + T]) for k in range(K))
    err = {p: float(np.abs(test_conv1d_sx(x, w, pad_l=3, precision=p) - ref).max()) for p in ("f32", "f16x3", "f16")}
    print(err)
    assert err["f32"] < 1e-5, err               # measured 5.5e-6 (fp32 accumulation of 896 terms)
    assert err["f16x3"] < 1e-5, err
    assert 2e-5 < err["f16"] < 5e-3, err        # two operands rounded to 11 bits: ~2^-11 per product, averaged over 896 terms


@pytest.mark.parametrize("B,Cin,Cout,T,K,dil", SX_CASES)
def test_conv_sx_f16_mode_matches_oracle(B, Cin, Cout, T, K, dil):
    # the engine's fp16 two-plane mode (three products per fp32 product, scaled weights): every epilogue / input path
    from phoonnx_amd.session import test_conv1d_sx
    from vits_oracle import conv1d
    rng = np.random.default_rng(B * 1000 + Cin + Cout + T + 1)
    x = rng.standard_normal((B, Cin, T)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, K)) / np.sqrt(Cin * K)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    pad = dil * (K - 1) // 2
    ref = conv1d(x, w, b, dil=dil, pad_l=pad, pad_r=dil * (K - 1) - pad)
    kw = dict(dil=dil, pad_l=pad, precision="f16x3")
    np.testing.assert_allclose(test_conv1d_sx(x, w, b, **kw), ref, atol=2e-5, rtol=1e-5)
    got = test_conv1d_sx(x, w, b, planes_slope=0.1, **kw)                   # read back from the two fp16 planes
    np.testing.assert_allclose(got, np.where(ref > 0, ref, ref * np.float32(0.1)), atol=2e-5, rtol=1e-5)
    if Cin == Cout:
        np.testing.assert_allclose(test_conv1d_sx(x, w, b, residual=True, **kw), ref + x, atol=2e-5, rtol=1e-5)
    if Cin <= 64:
        xa = np.where(x > 0, x, x * np.float32(0.1)).astype(np.float32)
        ref = conv1d(xa, w, b, dil=dil, pad_l=pad, pad_r=dil * (K - 1) - pad)
        got = test_conv1d_sx(x, w, b, in_slope=0.1, residual=Cin == Cout, **kw)
        np.testing.assert_allclose(got, ref + (x if Cin == Cout else 0), atol=2e-5, rtol=1e-5)


def _f16_operands(x, w):
    """What the single-plane arithmetic multiplies: activations rounded to fp16, weights rounded to fp16 after the per-tensor
    power-of-two scale that lifts the largest one into [2^14, 2^15) (model.cpp pack_conv_sx) - as float64."""
    _, e = np.frexp(np.float32(np.abs(w).max()))
    mul = np.float32(2.0) ** (15 - int(e))
    wq = (w * mul).astype(np.float16).astype(np.float64) / float(mul)
    return x.astype(np.float16).astype(np.float64), wq


H1_CASES = [c for c in SX_CASES if c[1] % 32 == 0]


@pytest.mark.parametrize("B,Cin,Cout,T,K,dil", H1_CASES)
def test_conv_sx_single_plane_mode_is_the_fp16_operand_product(B, Cin, Cout, T, K, dil):
    """gen_precision "f16" (BASELINE config 4) at kernel level: one fp16 plane per operand, ONE product, fp32 accumulation.
    Pinned to its definition: the result equals the float64 convolution of the fp16-ROUNDED operands to fp32-accumulation
    error - on every epilogue the generator uses in that mode: fp32 output (the multi-receptive-field sum), fp16 plane
    output carrying the consumer's leaky_relu, and the residual recovered from the activated input plane."""
    from phoonnx_amd.session import test_conv1d_sx
    rng = np.random.default_rng(B * 1000 + Cin + Cout + T + 2)
    x = rng.standard_normal((B, Cin, T)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, K)) / np.sqrt(Cin * K)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    pad = dil * (K - 1) // 2
    sl = np.float32(0.1)

    def conv64(xq, wq):
        xp = np.pad(xq, ((0, 0), (0, 0), (pad, dil * (K - 1) - pad)))
        return sum(np.einsum("oc,bct->bot", wq[:, :, k], xp[:, :, k * dil:k * dil + T]) for k in range(K)) + b[None, :, None].astype(np.float64)

    xq, wq = _f16_operands(x, w)
    ref = conv64(xq, wq)
    kw = dict(dil=dil, pad_l=pad, precision="f16")
    np.testing.assert_allclose(test_conv1d_sx(x, w, b, **kw), ref, atol=3e-5, rtol=1e-5)            # fp32 output
    act = np.where(ref > 0, ref, ref * float(sl))
    for only in (False, True):                                                                      # fp16 plane output
        got = test_conv1d_sx(x, w, b, planes_slope=0.1, planes_only=only, **kw)
        np.testing.assert_allclose(got, act, atol=3e-5, rtol=2.0 ** -10)
    if Cin == Cout:
        np.testing.assert_allclose(test_conv1d_sx(x, w, b, residual=True, **kw), ref + xq, atol=3e-5, rtol=1e-5)
        # the generator's in-block step: the input plane holds leaky_relu(x); the conv consumes it, the residual is x again
        xa = np.where(x > 0, x, x * sl).astype(np.float32)
        xaq, _ = _f16_operands(xa, w)
        xres = np.where(xaq >= 0, xaq, xaq * (np.float32(1.0) / sl).astype(np.float64))           # (1 / 0.1f in fp32, as the engine)
        want = conv64(xaq, wq) + xres
        np.testing.assert_allclose(test_conv1d_sx(x, w, b, in_slope=0.1, residual=True, **kw), want, atol=3e-5, rtol=1e-5)
        got = test_conv1d_sx(x, w, b, in_slope=0.1, planes_slope=0.1, residual=True, planes_only=True, **kw)
        np.testing.assert_allclose(got, np.where(want > 0, want, want * float(sl)), atol=3e-5, rtol=2.0 ** -10)
        # ... and the stored x is within one fp16 rounding of the true x
        assert np.abs(xres - x).max() <= np.abs(x).max() * 2.0 ** -10


@pytest.mark.parametrize("xscale,wscale", [(1.0, 1.0), (0.05, 1.0), (30.0, 1e-3), (1.0, 40.0), (3e-3, 1.0), (1e-4, 1e-2),
                                           (1e-6, 1.0)])
def test_conv_sx_f16_mode_error_is_fp32_grade(xscale, wscale):
    # against float64, next to the f32-MFMA engine on the same data.  Weights carry a per-tensor power-of-two scale,
    # so their magnitude must not matter; activations are stored unscaled, their low plane 2^11 up: full precision
    # down to |x| = 2^-14, an absolute floor of 2^-36 below.
    from phoonnx_amd.session import test_conv1d, test_conv1d_sx
    rng = np.random.default_rng(31)
    B, C, T, K, dil = 1, 128, 640, 7, 3
    x = (rng.standard_normal((B, C, T)) * xscale).astype(np.float32)
    w = (rng.standard_normal((C, C, K)) / np.sqrt(C * K) * wscale).astype(np.float32)
    pad = dil * (K - 1) // 2
    xp = np.pad(x.astype(np.float64), ((0, 0), (0, 0), (pad, pad)))
    ref = sum(np.einsum("oc,bct->bot", w[:, :, k].astype(np.float64), xp[:, :, k * dil:k * dil + T]) for k in range(K))
    rms = float(np.sqrt((ref ** 2).mean()))
    e_h = float(np.abs(test_conv1d_sx(x, w, dil=dil, pad_l=pad, precision="f16x3") - ref).max()) / rms
    e_f32 = float(np.abs(test_conv1d(x, w, dil=dil, pad_l=pad) - ref).max()) / rms
    e_h1 = float(np.abs(test_conv1d_sx(x, w, dil=dil, pad_l=pad, precision="f16") - ref).max()) / rms
    print(f"xscale {xscale} wscale {wscale}: f16x3 {e_h:.3g}  f32 engine {e_f32:.3g}  f16 (one plane) {e_h1:.3g} (max err / output rms)")
    floor = 2.0 ** -36 / xscale * 30          # the activations' absolute resolution, relative to their magnitude
    assert e_h <= 1.5 * e_f32 + floor, (e_h, e_f32)
    if xscale >= 1e-4:
        assert e_h < e_h1 / 20          # (the single-plane mode is the declared reduced-precision one)


def test_conv_sx_f16_mode_saturates_instead_of_overflowing():
    from phoonnx_amd.session import test_conv1d_sx
    C, T = 32, 64
    w = np.zeros((C, C, 1), np.float32)
    w[np.arange(C), np.arange(C), 0] = 1
    x = np.full((1, C, T), 3.0e5, np.float32)
    x[0, :, ::2] = -1.0e6
    x[0, 0, :] = 1234.5678
    got = test_conv1d_sx(x, w, precision="f16x3")
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got[0, 0], x[0, 0], rtol=3e-7)
    assert np.array_equal(got[0, 1:], np.clip(x[0, 1:], -65504, 65504))


PAIR_CASES = [
    # (B, C, T, K, dil1)        one fused launch = a ResBlock1 step on a raw-format stage
    (2, 32, 700, 3, 1), (1, 32, 1000, 7, 3), (2, 32, 246, 11, 5), (1, 32, 247, 11, 1), (1, 32, 5, 7, 5),
    (2, 64, 700, 3, 3), (1, 64, 513, 7, 5), (1, 64, 1000, 11, 5), (2, 64, 31, 11, 3), (1, 64, 256, 3, 1),
]


@pytest.mark.parametrize("B,C,T,K,dil", PAIR_CASES)
def test_conv_pair_sx_equals_two_launches_and_oracle(B, C, T, K, dil):
    """conv_sx_pair_kernel (c1 -> LDS -> c2 + x in one launch) against the two-launch form of the same f16x3
    arithmetic - bit for bit: same operand planes, same products, same accumulation order, only the tile boundaries
    move - and against the oracle's fp32 convolutions."""
    from phoonnx_amd.session import test_conv1d_sx, test_conv_pair_sx
    from vits_oracle import conv1d
    rng = np.random.default_rng(C * 1000 + T + K + dil)
    x = rng.standard_normal((B, C, T)).astype(np.float32)
    w1 = (rng.standard_normal((C, C, K)) / np.sqrt(C * K)).astype(np.float32)
    w2 = (rng.standard_normal((C, C, K)) / np.sqrt(C * K) * 3).astype(np.float32)
    b1 = rng.standard_normal(C).astype(np.float32)
    b2 = rng.standard_normal(C).astype(np.float32)
    got = test_conv_pair_sx(x, w1, b1, w2, b2, dil1=dil, dil2=1, slope=0.1)
    pad1, pad2 = dil * (K - 1) // 2, (K - 1) // 2
    mid = test_conv1d_sx(x, w1, b1, dil=dil, pad_l=pad1, in_slope=0.1, precision="f16x3")
    two = test_conv1d_sx(mid, w2, b2, dil=1, pad_l=pad2, in_slope=0.1, precision="f16x3") + x
    assert np.array_equal(got, two), float(np.abs(got - two).max())
    lr = lambda v: np.where(v > 0, v, v * np.float32(0.1)).astype(np.float32)
    ref = conv1d(lr(conv1d(lr(x), w1, b1, dil=dil, pad_l=pad1, pad_r=pad1)), w2, b2, dil=1, pad_l=pad2, pad_r=pad2) + x
    np.testing.assert_allclose(got, ref, atol=5e-5, rtol=1e-5)


CHAIN_CASES = [
    # (B, C, T, K, dil1, dil2): two ResBlock2 steps per launch (medium preset: k (3,5,7), d ((1,2),(2,6),(3,12)))
    (2, 32, 700, 3, 1, 2), (1, 32, 1000, 5, 2, 6), (2, 32, 300, 7, 3, 12), (1, 32, 9, 5, 2, 6),
    (2, 64, 700, 3, 1, 2), (1, 64, 513, 5, 2, 6), (1, 64, 31, 3, 1, 2),
]


@pytest.mark.parametrize("B,C,T,K,d1,d2", CHAIN_CASES)
def test_conv_chain_sx_equals_two_launches_and_oracle(B, C, T, K, d1, d2):
    """CHAIN mode of conv_sx_pair_kernel: x1 = c1(lrelu(x)) + x; out = c2(lrelu(x1)) + x1 in one launch, against the
    two-launch form bit for bit and against the oracle."""
    from phoonnx_amd.session import test_conv1d_sx, test_conv_pair_sx
    from vits_oracle import conv1d
    rng = np.random.default_rng(C * 999 + T + K + d1)
    x = rng.standard_normal((B, C, T)).astype(np.float32)
    w1 = (rng.standard_normal((C, C, K)) / np.sqrt(C * K)).astype(np.float32)
    w2 = (rng.standard_normal((C, C, K)) / np.sqrt(C * K) * 2).astype(np.float32)
    b1 = rng.standard_normal(C).astype(np.float32)
    b2 = rng.standard_normal(C).astype(np.float32)
    got = test_conv_pair_sx(x, w1, b1, w2, b2, dil1=d1, dil2=d2, chain=True, slope=0.1)
    p1, p2 = d1 * (K - 1) // 2, d2 * (K - 1) // 2
    x1 = test_conv1d_sx(x, w1, b1, dil=d1, pad_l=p1, in_slope=0.1, residual=True, precision="f16x3")
    two = test_conv1d_sx(x1, w2, b2, dil=d2, pad_l=p2, in_slope=0.1, residual=True, precision="f16x3")
    assert np.array_equal(got, two), float(np.abs(got - two).max())
    lr = lambda v: np.where(v > 0, v, v * np.float32(0.1)).astype(np.float32)
    r1 = conv1d(lr(x), w1, b1, dil=d1, pad_l=p1, pad_r=p1) + x
    ref = conv1d(lr(r1), w2, b2, dil=d2, pad_l=p2, pad_r=p2) + r1
    np.testing.assert_allclose(got, ref, atol=5e-5, rtol=1e-5)


def _pair_ref(x, w1, b1, w2, b2, K, d1, d2, chain):
    from vits_oracle import conv1d
    lr = lambda v: np.where(v > 0, v, v * np.float32(0.1)).astype(np.float32)
    p1, p2 = d1 * (K - 1) // 2, d2 * (K - 1) // 2
    if chain:
        r1 = conv1d(lr(x), w1, b1, dil=d1, pad_l=p1, pad_r=p1) + x
        return conv1d(lr(r1), w2, b2, dil=d2, pad_l=p2, pad_r=p2) + r1
    return conv1d(lr(conv1d(lr(x), w1, b1, dil=d1, pad_l=p1, pad_r=p1)), w2, b2, dil=d2, pad_l=p2, pad_r=p2) + x


PAIR16_CASES = [(B, C, T, K, d, 1, False) for (B, C, T, K, d) in PAIR_CASES] + [c + (True,) for c in CHAIN_CASES] + [
    (1, 32, 390, 3, 1, 1, False),     # three tiles of 126 kept columns, the last one ragged
    (2, 64, 129, 5, 1, 1, True),
]


@pytest.mark.parametrize("B,C,T,K,d1,d2,chain", PAIR16_CASES)
def test_conv_pair16_f16x3_matches_two_launches_and_oracle(B, C, T, K, d1, d2, chain):
    """conv_sx_pair16_kernel (the fused ResBlock step on the 16x16x32 loop) in the default arithmetic: same products in the
    same order as two conv_sx_kernel launches; the input arrives as the two operand planes of leaky_relu(x) (by LDS-DMA), the
    residual is x reconstructed from them in LDS (22 bits),
    so the result is within ~2^-22 |x| of the two-launch form (no longer bit-identical) - and within the engine's tolerance
    of the oracle's fp32 convolutions."""
    from phoonnx_amd.session import test_conv1d_sx, test_conv_pair_sx
    rng = np.random.default_rng(C * 1000 + T + K + d1 + 7)
    x = rng.standard_normal((B, C, T)).astype(np.float32)
    w1 = (rng.standard_normal((C, C, K)) / np.sqrt(C * K)).astype(np.float32)
    w2 = (rng.standard_normal((C, C, K)) / np.sqrt(C * K) * 2).astype(np.float32)
    b1 = rng.standard_normal(C).astype(np.float32)
    b2 = rng.standard_normal(C).astype(np.float32)
    got = test_conv_pair_sx(x, w1, b1, w2, b2, dil1=d1, dil2=d2, chain=chain, slope=0.1, kernel="pair16")
    p1, p2 = d1 * (K - 1) // 2, d2 * (K - 1) // 2
    if chain:
        x1 = test_conv1d_sx(x, w1, b1, dil=d1, pad_l=p1, in_slope=0.1, residual=True, precision="f16x3")
        two = test_conv1d_sx(x1, w2, b2, dil=d2, pad_l=p2, in_slope=0.1, residual=True, precision="f16x3")
    else:
        mid = test_conv1d_sx(x, w1, b1, dil=d1, pad_l=p1, in_slope=0.1, precision="f16x3")
        two = test_conv1d_sx(mid, w2, b2, dil=d2, pad_l=p2, in_slope=0.1, precision="f16x3") + x
    assert float(np.abs(got - two).max()) < 6e-6, float(np.abs(got - two).max())
    ref = _pair_ref(x, w1, b1, w2, b2, K, d1, d2, chain)
    np.testing.assert_allclose(got, ref, atol=5e-5, rtol=1e-5)
    # ... and written as the operand planes of leaky_relu(out): 22 bits
    pl = test_conv_pair_sx(x, w1, b1, w2, b2, dil1=d1, dil2=d2, chain=chain, slope=0.1, kernel="pair16", from_plane=True)
    np.testing.assert_allclose(pl, np.where(ref > 0, ref, ref * np.float32(0.1)), atol=5e-5, rtol=1e-5)


@pytest.mark.parametrize("B,C,T,K,d1,d2,chain", PAIR16_CASES)
def test_conv_pair16_single_plane_mode(B, C, T, K, d1, d2, chain):
    """The same kernel in the reduced-precision arithmetic (gen_precision "f16"): fp16 plane in (the consumer's leaky_relu
    already applied), one product, the intermediate as one fp16 plane in LDS, residual = the stored plane with the leaky_relu
    undone, fp32 or fp16-plane out.  Against the float64 evaluation of exactly those roundings (differences: fp32
    accumulation, and the rare intermediate that rounds the other way), and against the unrounded fp32 reference at the
    mode's own resolution."""
    from phoonnx_amd.session import test_conv_pair_sx
    rng = np.random.default_rng(C * 1000 + T + K + d1 + 9)
    x = rng.standard_normal((B, C, T)).astype(np.float32)
    w1 = (rng.standard_normal((C, C, K)) / np.sqrt(C * K)).astype(np.float32)
    w2 = (rng.standard_normal((C, C, K)) / np.sqrt(C * K) * 2).astype(np.float32)
    b1 = rng.standard_normal(C).astype(np.float32)
    b2 = rng.standard_normal(C).astype(np.float32)
    sl = np.float32(0.1)
    lr = lambda v: np.where(v > 0, v, v * float(sl))
    h = lambda v: np.asarray(v, np.float32).astype(np.float16).astype(np.float64)

    def conv64(xq, wq, bb, d):
        p = d * (K - 1) // 2
        xp = np.pad(xq, ((0, 0), (0, 0), (p, p)))
        return sum(np.einsum("oc,bct->bot", wq[:, :, k], xp[:, :, k * d:k * d + T]) for k in range(K)) + bb[None, :, None].astype(np.float64)

    xa = h(lr(x).astype(np.float32))                                  # the stored input plane
    _, w1q = _f16_operands(x, w1)
    _, w2q = _f16_operands(x, w2)
    xres = np.where(xa >= 0, xa, xa * float(np.float32(1.0) / sl))      # the residual the kernel recovers
    c1 = conv64(xa, w1q, b1, d1)
    if chain:
        x1 = c1 + xres
        want = conv64(h(lr(x1)), w2q, b2, d2) + x1
    else:
        want = conv64(h(lr(c1)), w2q, b2, d2) + xres
    got = test_conv_pair_sx(x, w1, b1, w2, b2, dil1=d1, dil2=d2, chain=chain, slope=0.1, kernel="pair16_f16")
    np.testing.assert_allclose(got, want, atol=3e-3, rtol=0)
    assert float(np.sqrt(((got - want) ** 2).mean())) < 1e-4             # ... and nearly everywhere to fp32-accumulation error
    ref = _pair_ref(x, w1, b1, w2, b2, K, d1, d2, chain)
    rms = float(np.sqrt((ref.astype(np.float64) ** 2).mean()))
    assert float(np.sqrt(((got - ref) ** 2).mean())) < 2e-3 * rms        # fp16 operands: ~2^-11 per rounding
    pl = test_conv_pair_sx(x, w1, b1, w2, b2, dil1=d1, dil2=d2, chain=chain, slope=0.1, kernel="pair16_f16", from_plane=True)
    np.testing.assert_allclose(pl, lr(want), atol=3e-3, rtol=2.0 ** -10)


def test_conv_pair_sx_refuses_what_it_cannot_fuse():
    """The fused launch is refused - never silently computed some other way - where the second conv's reach leaves too
    little of a 256-column tile (64 channels keep >= 200 columns: a k = 7, dilation 12 second conv keeps 184), and for
    channel counts other than 32 / 64; the generator then runs those steps as two launches (sx_pair_ok)."""
    from phoonnx_amd.session import SessionError, test_conv_pair_sx
    rng = np.random.default_rng(3)
    for C, K, d1, d2 in ((64, 7, 3, 12), (32, 11, 1, 12), (128, 3, 1, 1)):
        x = rng.standard_normal((1, C, 600)).astype(np.float32)
        w = (rng.standard_normal((C, C, K)) / np.sqrt(C * K)).astype(np.float32)
        with pytest.raises(SessionError):
            test_conv_pair_sx(x, w, None, w, None, dil1=d1, dil2=d2, chain=True, slope=0.1)


def test_conv_pair_sx_is_used_by_the_generator_and_can_be_switched_off(monkeypatch):
    """A ResBlock1 voice with 64- and 32-channel raw-format stages: fused and unfused generators give the same
    waveform (to ~2^-22 of the residual stream: the fused kernel on the 16x16x32 loop reconstructs the residual from its
    operand planes), and the fused one issues fewer launches."""
    from phoonnx_amd import MiSession
    path = os.path.join(GOLDEN, "sx_rb1.onnx")
    g = np.load(os.path.join(GOLDEN, "sx_rb1.npz"))
    args = [case_get(g, "b3_noise", k) for k in ("ids", "lens", "scales", "sid", "noise_dp", "noise_z")]
    s = MiSession(path)
    a = s.synthesize_batch(*args)
    na = s.stats()["sx_launches"]
    s.close()
    monkeypatch.setenv("VITSMI_SX_NO_PAIR", "1")
    import subprocess, sys  # (the switch is read once per process: ask a fresh one)
    code = ("import sys, numpy as np; sys.path.insert(0, sys.argv[1]); from phoonnx_amd import MiSession; "
            "g = np.load(sys.argv[3]); s = MiSession(sys.argv[2]); "
            "r = s.synthesize_batch(*[g['b3_noise/' + k] if 'b3_noise/' + k in g.files else None for k in "
            "('ids', 'lens', 'scales', 'sid', 'noise_dp', 'noise_z')]); "
            "np.save(sys.argv[4], r['output']); print('LAUNCHES', s.stats()['sx_launches'])")
    out = os.path.join(os.environ.get("TMPDIR", "/tmp"), "nopair_out.npy")
    from conftest import ROOT
    r = subprocess.run([sys.executable, "-c", code, ROOT, path, os.path.join(GOLDEN, "sx_rb1.npz"), out],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-800:]
    nb = int(r.stdout.split("LAUNCHES")[1].split()[0])
    assert na < nb, (na, nb)                     # 64- and 32-channel ResBlock1 steps: one launch instead of two
    np.testing.assert_allclose(np.load(out), a["output"], atol=5e-6, rtol=0)


@pytest.mark.parametrize("B,Cin,Cout,T,K,u", [(2, 32, 32, 50, 16, 8), (2, 128, 64, 129, 4, 2)])
def test_conv_transpose_sx_f16_mode_matches_oracle(B, Cin, Cout, T, K, u):
    from phoonnx_amd.session import test_conv_transpose1d
    from vits_oracle import conv_transpose1d
    rng = np.random.default_rng(K * 100 + u + 1)
    x = rng.standard_normal((B, Cin, T)).astype(np.float32)
    w = (rng.standard_normal((Cin, Cout, K)) / np.sqrt(Cin * K / u)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    got = test_conv_transpose1d(x, w, b, u, sx="f16")
    np.testing.assert_allclose(got, conv_transpose1d(x, w, b, u, (K - u) // 2), atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("B,Cin,Cout,T,K,u", [(2, 32, 32, 50, 16, 8), (1, 64, 32, 37, 8, 4), (2, 128, 64, 129, 4, 2),
                                              (1, 512, 256, 9, 16, 8)])
def test_conv_transpose_sx_matches_oracle(B, Cin, Cout, T, K, u):
    from phoonnx_amd.session import test_conv_transpose1d
    from vits_oracle import conv_transpose1d
    rng = np.random.default_rng(K * 100 + u)
    x = rng.standard_normal((B, Cin, T)).astype(np.float32)
    w = (rng.standard_normal((Cin, Cout, K)) / np.sqrt(Cin * K / u)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    got = test_conv_transpose1d(x, w, b, u, sx=True)
    ref = conv_transpose1d(x, w, b, u, (K - u) // 2)
    assert got.shape == ref.shape == (B, Cout, T * u)
    np.testing.assert_allclose(got, ref, atol=2e-5, rtol=1e-5)


# ------------------------------------------------------------------ kernel level: attention

@pytest.mark.parametrize("B,heads,dk,T,lens", [(2, 2, 16, 40, [40, 17]), (1, 2, 96, 256, [256]), (3, 4, 8, 5, [5, 1, 3]),
                                               (2, 2, 96, 70, [33, 70]), (1, 1, 64, 3, [3]), (2, 2, 48, 129, [129, 64])])
def test_attention_matches_oracle(B, heads, dk, T, lens):
    from phoonnx_amd.session import test_attention
    from vits_oracle import attention_core
    rng = np.random.default_rng(T * 7 + dk)
    C = heads * dk
    qkv = rng.standard_normal((B, 3 * C, T)).astype(np.float32)
    rk = (rng.standard_normal((9, dk)) * dk ** -0.5).astype(np.float32)
    rv = (rng.standard_normal((9, dk)) * dk ** -0.5).astype(np.float32)
    lens = np.asarray(lens, np.int64)
    got = test_attention(qkv, heads, rk, rv, lens)
    ref = attention_core(qkv, heads, rk, rv, lens)
    for b in range(B):  # padded query rows never reach valid outputs; compare valid ones
        np.testing.assert_allclose(got[b, :, :lens[b]], ref[b, :, :lens[b]], atol=2e-5, rtol=1e-4)
        assert np.all(got[b, :, lens[b]:] == 0)


def test_attention_sharp_softmax():
    # large logits force the online-softmax rescale branch across key blocks
    from phoonnx_amd.session import test_attention
    from vits_oracle import attention_core
    rng = np.random.default_rng(5)
    B, heads, dk, T = 1, 2, 32, 200
    C = heads * dk
    qkv = rng.standard_normal((B, 3 * C, T)).astype(np.float32)
    qkv[:, :C] *= 6.0
    qkv[:, C:2 * C, 150:] *= 5.0   # the maximum jumps late in the key sweep
    rk = rng.standard_normal((9, dk)).astype(np.float32)
    rv = rng.standard_normal((9, dk)).astype(np.float32)
    lens = np.asarray([T], np.int64)
    got = test_attention(qkv, heads, rk, rv, lens)
    ref = attention_core(qkv, heads, rk, rv, lens)
    np.testing.assert_allclose(got, ref, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("B,Cin,Cout,T,K,dil", [(2, 128, 128, 700, 3, 1), (1, 128, 128, 333, 7, 3), (2, 192, 256, 97, 7, 1),
                                                 (1, 256, 256, 130, 11, 5), (3, 128, 128, 64, 5, 2)])
@pytest.mark.parametrize("mode", ["raw", "planes", "planes_only", "residual"])
def test_short_launch_kernel_generator_epilogue_matches_the_engine(B, Cin, Cout, T, K, dil, mode):
    """The generator's epilogue (raw cells, operand planes with the consumer's leaky_relu, residual) on the short-launch kernel
    (conv_sx_small.hip.hpp: plane-input convs of the > 64-channel stages at batch 1) against the engine's launch of the same
    conv and against float64."""
    from phoonnx_amd.session import test_conv1d_sx
    if mode == "residual" and Cin != Cout:
        pytest.skip("residual needs Cin == Cout")
    rng = np.random.default_rng(Cin + T + K)
    x = rng.standard_normal((B, Cin, T)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, K)) / np.sqrt(Cin * K)).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32)
    pad = dil * (K - 1) // 2
    kw = dict(dil=dil, pad_l=pad, precision="f16x3")
    if mode == "planes":
        kw["planes_slope"] = 0.1
    elif mode == "planes_only":
        kw.update(planes_slope=0.1, planes_only=True)
    elif mode == "residual":
        kw["residual"] = True
    a = test_conv1d_sx(x, w, bias, small=True, **kw)
    e = test_conv1d_sx(x, w, bias, **kw)
    want = _conv_same_f64(x, w, bias, dil)
    if mode == "residual":
        want = want + x
    if mode in ("planes", "planes_only"):
        want = np.where(want > 0, want, 0.1 * want)
    scale = float(np.abs(want).max())
    np.testing.assert_allclose(a, want, atol=3e-6 * scale, rtol=0)
    np.testing.assert_allclose(a, e, atol=3e-6 * scale, rtol=0)


def _planes_value(pl):
    """fp16 operand planes [B, 3, C/8, T, 8] (f16x3 format: h0, h1 * 2^11) -> the fp32 values they stand for [B, C, T]"""
    h = pl.view(np.float16).astype(np.float64)
    v = h[:, 0] + h[:, 1] / 2048.0
    B, CG, T, _ = v.shape
    return v.transpose(0, 1, 3, 2).reshape(B, CG * 8, T)


ATT16_CASES = [(1, 2, 96, 256, [256]), (2, 2, 96, 70, [33, 70]), (1, 1, 64, 3, [3]), (2, 2, 32, 129, [129, 64]),
               (3, 2, 96, 500, [500, 1, 257]), (2, 4, 32, 65, [64, 65]), (1, 2, 96, 33, [32]),
               # (a launch of more than one workgroup per CU, ragged)
               (40, 2, 96, 300, [300 - 7 * i for i in range(40)])]


@pytest.mark.parametrize("B,heads,dk,T,lens", ATT16_CASES)
def test_attention16_matches_oracle(B, heads, dk, T, lens):
    """the f16x3 16x16x32 attention kernel (attention16.hip.hpp) against the oracle's attention core at the fp32 kernel's
    tolerance, and its second output (conv_o's operand planes) against its first"""
    from phoonnx_amd.session import test_attention16
    from vits_oracle import attention_core
    rng = np.random.default_rng(T * 7 + dk)
    C = heads * dk
    qkv = rng.standard_normal((B, 3 * C, T)).astype(np.float32)
    rk = (rng.standard_normal((9, dk)) * dk ** -0.5).astype(np.float32)
    rv = (rng.standard_normal((9, dk)) * dk ** -0.5).astype(np.float32)
    lens = np.asarray(lens, np.int64)
    got, pl = test_attention16(qkv, heads, rk, rv, lens, planes=True)
    ref = attention_core(qkv, heads, rk, rv, lens)
    val = _planes_value(pl)
    for b in range(B):
        np.testing.assert_allclose(got[b, :, :lens[b]], ref[b, :, :lens[b]], atol=2e-5, rtol=1e-4)
        assert np.all(got[b, :, lens[b]:] == 0)
        np.testing.assert_allclose(val[b], got[b].astype(np.float64), atol=1e-9, rtol=3e-7)


def test_attention16_window_2_and_sharp_softmax():
    from phoonnx_amd.session import test_attention16
    from vits_oracle import attention_core
    rng = np.random.default_rng(5)
    B, heads, dk, T = 1, 2, 32, 200
    C = heads * dk
    qkv = rng.standard_normal((B, 3 * C, T)).astype(np.float32)
    qkv[:, :C] *= 6.0
    qkv[:, C:2 * C, 150:] *= 5.0   # the maximum jumps late in the key sweep
    lens = np.asarray([T], np.int64)
    for nrel in (9, 5):
        rk = rng.standard_normal((nrel, dk)).astype(np.float32)
        rv = rng.standard_normal((nrel, dk)).astype(np.float32)
        got = test_attention16(qkv, heads, rk, rv, lens)
        ref = attention_core(qkv, heads, rk, rv, lens)
        np.testing.assert_allclose(got, ref, atol=1e-4, rtol=1e-4)


# ------------------------------------------------------------------ whole path vs reference fixtures

TAPS = ("x", "m_p", "logs_p", "logw", "w_ceil", "z_p", "z")


@pytest.mark.parametrize("tails", ["zero", "reference"])
@pytest.mark.parametrize("preset", TINY_PRESETS)
def test_pipeline_matches_reference_goldens(preset, tails):
    s = _session(preset, tails)
    g = np.load(os.path.join(GOLDEN, preset + ".npz"))
    for c in golden_cases(g):
        r = s.synthesize_batch(case_get(g, c, "ids"), case_get(g, c, "lens"), case_get(g, c, "scales"),
                               case_get(g, c, "sid"), case_get(g, c, "noise_dp"), case_get(g, c, "noise_z"), taps=TAPS)
        lens = case_get(g, c, "lens")
        ylen = case_get(g, c, "out_y_lengths")
        # integer work first: durations and frame counts are exact
        assert np.array_equal(r["w_ceil"], case_get(g, c, "out_w_ceil")), (preset, c)
        assert np.array_equal(r["y_lengths"], ylen), (preset, c)
        for k in ("x", "m_p", "logs_p", "logw", "z_p", "z"):
            ref = case_get(g, c, "out_" + k)
            assert r[k].shape == ref.shape, (preset, c, k, r[k].shape, ref.shape)
            np.testing.assert_allclose(r[k], ref, atol=STAGE_TOL, rtol=0, err_msg=f"{preset}/{c}/{k}")
        # tails="reference": the graph's whole padded output; "zero": its valid samples, exact zeros behind them
        ref = _ref_wave(g, c, s, tails)
        assert r["output"].shape == ref.shape and r["output"].dtype == np.float32
        np.testing.assert_allclose(r["output"], ref, atol=WAVE_TOL, rtol=0, err_msg=f"{preset}/{c}/output")
        assert np.abs(r["output"] - ref).max() < 5e-5, "fp32 path should sit far inside the 1e-3 budget"
    s.close()


def test_reserved_workspaces_do_not_grow_and_change_no_result():
    """vits_reserve (MiSession.reserve): a handle sized up front for the largest request renders the fixtures without growing
    its workspaces (hparam "workspace_bytes" stays put) and bit-identically to a handle that grew on demand; a request
    beyond the reservation still works (grows); a reservation the device cannot hold fails with the engine's error."""
    preset = TINY_PRESETS[0]
    g = np.load(os.path.join(GOLDEN, preset + ".npz"))
    cases = golden_cases(g)
    args = lambda c: (case_get(g, c, "ids"), case_get(g, c, "lens"), case_get(g, c, "scales"), case_get(g, c, "sid"),  # noqa: E731
                      case_get(g, c, "noise_dp"), case_get(g, c, "noise_z"))
    grown, fixed = _session(preset), _session(preset)
    Bm = max(case_get(g, c, "ids").shape[0] for c in cases)
    Tm = max(case_get(g, c, "ids").shape[1] for c in cases)
    Fm = max(int(case_get(g, c, "out_y_lengths").max()) for c in cases)
    before = fixed.hparam("workspace_bytes")
    fixed.reserve(Bm, Tm, Fm)
    cap = fixed.hparam("workspace_bytes")
    assert cap > before
    for c in cases:
        a, b = grown.synthesize_batch(*args(c)), fixed.synthesize_batch(*args(c))
        assert np.array_equal(a["output"], b["output"]) and np.array_equal(a["y_lengths"], b["y_lengths"]), c
        assert fixed.hparam("workspace_bytes") == cap, (c, "a reserved handle allocated")
    fixed.reserve(Bm, Tm, Fm)            # (idempotent)
    fixed.reserve(1, 0, 0)               # (smaller: nothing shrinks)
    assert fixed.hparam("workspace_bytes") == cap
    small = _session(preset)
    small.reserve(1, 4, 4)
    c = cases[-1]
    assert np.array_equal(small.synthesize_batch(*args(c))["output"], grown.synthesize_batch(*args(c))["output"])
    with pytest.raises(Exception, match="hipMalloc"):
        small.reserve(4096, 4096, 1 << 20)
    assert np.array_equal(small.synthesize_batch(*args(c))["output"], grown.synthesize_batch(*args(c))["output"])
    with pytest.raises(Exception):
        small.reserve(0, 1, 1)
    # A reservation that GROWS a workspace frees what the last run left there (vitsmi.h: fetch first, reserve afterwards):
    # the result calls must then refuse ("no completed run") rather than read freed device memory, and the next run is
    # unaffected.  A reservation that grows nothing leaves the last run's results readable.
    r0 = grown.synthesize_batch(*args(c))
    B0, S0 = r0["output"].shape[0], r0["output"].shape[3]
    pcm0 = grown.last_pcm16(True, 1.0, shape=(B0, S0))
    grown.reserve(1, 0, 0)                                   # (grows nothing)
    assert np.array_equal(grown.last_pcm16(True, 1.0, shape=(B0, S0)), pcm0)
    z0 = grown.tap("z")
    cap0 = grown.hparam("workspace_bytes")
    grown.reserve(4 * Bm, 4 * Tm, 8 * Fm)                    # (grows all three)
    assert grown.hparam("workspace_bytes") > cap0
    with pytest.raises(Exception, match="no completed run"):
        grown.last_pcm16(True, 1.0, shape=(B0, S0))
    with pytest.raises(Exception, match="no completed run"):
        grown.tap("z")
    with pytest.raises(Exception, match="no completed run"):
        grown._fetch(np.empty_like(r0["output"]), 0, B0)
    cap1 = grown.hparam("workspace_bytes")
    r1 = grown.synthesize_batch(*args(c))
    assert np.array_equal(r1["output"], r0["output"]) and np.array_equal(grown.tap("z"), z0)
    assert np.array_equal(grown.last_pcm16(True, 1.0, shape=(B0, S0)), pcm0)
    assert grown.hparam("workspace_bytes") == cap1, "last_pcm16 allocated behind a reservation that covers the request"
    for s in (grown, fixed, small):
        s.close()


@pytest.mark.parametrize("tails", ["zero", "reference"])
@pytest.mark.parametrize("precision,nprod", [("f16x3", 2), ("bf16x6", 6)])
@pytest.mark.parametrize("preset", SX_PRESETS)
def test_sx_generator_matches_reference_goldens(monkeypatch, preset, precision, nprod, tails):
    """The headline kernel pinned to the reference: these fixtures come from the reference's own PyTorch graph
    (oracle/gen_golden.py) and their generators (256 -> 128 -> 64 -> 32 channels) run on conv_sx_kernel - plane-format
    ResBlock pairs at 128 channels, raw-format stages below, pixel-shuffled upsamplers, the speaker bias - in BOTH
    arithmetics.  A break in conv_sx_kernel or pack_conv_sx fails here against numbers the reference produced."""
    monkeypatch.setenv("VITSMI_GEN_PRECISION", precision)
    s = _session(preset, tails)
    assert s.hparam("gen_sx") == 1 and s.hparam("gen_nprod") == nprod
    g = np.load(os.path.join(GOLDEN, preset + ".npz"))
    worst = 0.0
    for c in golden_cases(g):
        r = s.synthesize_batch(case_get(g, c, "ids"), case_get(g, c, "lens"), case_get(g, c, "scales"),
                               case_get(g, c, "sid"), case_get(g, c, "noise_dp"), case_get(g, c, "noise_z"), taps=TAPS)
        assert np.array_equal(r["w_ceil"], case_get(g, c, "out_w_ceil")), (preset, c)
        assert np.array_equal(r["y_lengths"], case_get(g, c, "out_y_lengths")), (preset, c)
        for k in ("x", "m_p", "logs_p", "logw", "z_p", "z"):
            np.testing.assert_allclose(r[k], case_get(g, c, "out_" + k), atol=STAGE_TOL, rtol=0, err_msg=f"{preset}/{c}/{k}")
        ref = _ref_wave(g, c, s, tails)
        assert r["output"].shape == ref.shape and r["output"].dtype == np.float32
        assert np.abs(ref).max() > 0.05                       # the comparison is not vacuous
        err = float(np.abs(r["output"] - ref).max())
        worst = max(worst, err)
        assert err < WAVE_TOL, (preset, c, err)
        assert err < 5e-5, "both full-precision arithmetics sit far inside the 1e-3 budget"
    print(f"{preset} {precision}: worst waveform error vs the reference fixture {worst:.3g}")
    assert s.stats()["sx_launches"] > 0                          # ... and it really ran on the sx engine
    s.close()


@pytest.mark.parametrize("res", ["planes", "raw"])
@pytest.mark.parametrize("preset", SX_PRESETS)
def test_residual_stream_of_the_wide_stages_matches_reference_goldens_in_both_forms(monkeypatch, preset, res):
    """The > 64-channel stages of the f16x3 generator keep their residual stream as operand planes only (the default:
    SX_RES_PL, a residual is recovered from the planes of leaky_relu(x); 12 instead of 16 bytes per element on the residual
    convs) or as fp32 raw tensors next to the planes (VITSMI_F16X3_RES=raw: rounds 1-4); the <= 64-channel stages stay on
    the fused raw-format kernels either way.  Same fixtures, same bar."""
    monkeypatch.setenv("VITSMI_F16X3_RES", res)
    s = _session(preset, "reference")
    assert s.hparam("gen_sx") == 1 and s.hparam("gen_nprod") == 2
    g = np.load(os.path.join(GOLDEN, preset + ".npz"))
    worst = 0.0
    for c in golden_cases(g):
        r = s.synthesize_batch(case_get(g, c, "ids"), case_get(g, c, "lens"), case_get(g, c, "scales"),
                               case_get(g, c, "sid"), case_get(g, c, "noise_dp"), case_get(g, c, "noise_z"))
        assert np.array_equal(r["y_lengths"], case_get(g, c, "out_y_lengths")), (preset, c)
        worst = max(worst, float(np.abs(r["output"] - case_get(g, c, "out_output")).max()))
    print(f"{preset} residual stream as {res}: worst waveform error vs the reference fixture {worst:.3g}")
    assert worst < 5e-5
    s.close()


@pytest.mark.parametrize("tails", ["zero", "reference"])
@pytest.mark.parametrize("precision,tol", [("f16x3", 5e-5), ("f16", 1e-2)])
@pytest.mark.parametrize("preset", SX_PRESETS)
def test_plane_stream_generator_matches_reference_goldens(monkeypatch, preset, precision, tol, tails):
    """The PLANE-STREAM generator (vitsmi.hip run_generator_planes: every inter-conv tensor stored once, as the operand
    planes of its consumer's leaky_relu; residuals recovered from them; fused ResBlock steps on conv_sx_pair16_kernel)
    against the reference fixtures - in the single-plane arithmetic, whose only generator it is (declared tolerance 1e-2 /
    35 dB, measured ~1e-3), and in f16x3, where it is selectable (VITSMI_F16X3_STREAM=planes; the default there is the
    raw-stream generator, which measures 3-5 % faster) and must be as accurate as the default."""
    monkeypatch.setenv("VITSMI_GEN_PRECISION", precision)
    monkeypatch.setenv("VITSMI_F16X3_STREAM", "planes")
    s = _session(preset, tails)
    assert s.hparam("gen_sx") == 1 and s.hparam("gen_nprod") == (2 if precision == "f16x3" else 1)
    g = np.load(os.path.join(GOLDEN, preset + ".npz"))
    worst, snr = 0.0, 1e9
    for c in golden_cases(g):
        r = s.synthesize_batch(case_get(g, c, "ids"), case_get(g, c, "lens"), case_get(g, c, "scales"),
                               case_get(g, c, "sid"), case_get(g, c, "noise_dp"), case_get(g, c, "noise_z"), taps=("z",))
        assert np.array_equal(r["y_lengths"], case_get(g, c, "out_y_lengths")), (preset, c)
        np.testing.assert_allclose(r["z"], case_get(g, c, "out_z"), atol=STAGE_TOL, rtol=0)
        ref = _ref_wave(g, c, s, tails)
        worst = max(worst, float(np.abs(r["output"] - ref).max()))
        d = (r["output"] - ref).astype(np.float64)
        snr = min(snr, 10 * np.log10((ref.astype(np.float64) ** 2).sum() / max((d ** 2).sum(), 1e-30)))
    print(f"{preset} {precision} plane stream: worst waveform error vs the reference fixture {worst:.3g}, SNR {snr:.1f} dB")
    assert worst < tol and snr > 35.0
    recs = s.stats()
    assert recs["sx_launches"] > 0
    s.close()


def _conv_same_f64(x, w, bias, dil):
    B, Cin, T = x.shape
    Cout, _, K = w.shape
    pad = dil * (K - 1) // 2
    xp = np.zeros((B, Cin, T + 2 * pad))
    xp[:, :, pad:pad + T] = x
    y = np.zeros((B, Cout, T))
    for k in range(K):
        y += np.einsum("oc,bct->bot", w[:, :, k].astype(np.float64), xp[:, :, k * dil:k * dil + T])
    return y + (0 if bias is None else bias.astype(np.float64)[None, :, None])


PLANAR_CASES = [
    # (B, Cin, Cout, T, K, row_split, pl_rows, options)                      as used by
    (2, 192, 576, 200, 1, None, 0, dict()),                                   # encoder q|k|v: planar out, nothing else
    (2, 192, 192, 333, 1, None, 0, dict(residual=True)),                      # encoder o: x = x_in + o(att)
    (3, 192, 768, 256, 3, None, 768, dict(relu=True, mask=True)),             # FFN conv_1: ReLU, mask, operand planes
    (2, 768, 192, 131, 3, None, 0, dict(mask=True, accumulate=True)),         # FFN conv_2: x += y * mask
    (2, 192, 384, 824, 1, 192, 192, dict(mask=True, accumulate=True)),        # flow res_skip: x / skip update, planes of x
    (2, 192, 384, 300, 1, 192, 0, dict(mask=True, accumulate=True, store2=True)),   # ... first layer: skip stored
    (2, 192, 192, 500, 1, 0, 192, dict(mask=True, accumulate=True, planes_of2=True)),  # ... last layer: planes of skip
    (2, 192, 96, 700, 1, None, 96, dict(mask=True, accumulate=True, coupling=True)),   # coupling post: x1 update + planes
    (1, 96, 192, 97, 1, None, 192, dict(mask=True)),                          # coupling pre (three 32-channel chunks)
    (2, 768, 192, 300, 3, None, 192, dict(mask=True, accumulate=True)),       # FFN conv_2 with planes
    (2, 256, 128, 200, 3, None, 0, dict(relu=True, mask=True)),               # eight chunks, ReLU, no old operand
]


@pytest.mark.parametrize("small", [False, True], ids=["engine", "short-launch"])
@pytest.mark.parametrize("B,Cin,Cout,T,K,row_split,pl_rows,opt", PLANAR_CASES)
def test_conv_sx_planar_epilogue_matches_float64(B, Cin, Cout, T, K, row_split, pl_rows, opt, small):
    """Kernel level: the planar epilogue of the split-operand engine (f16x3 products, 16x16x32 loop) - and of the
    short-launch kernel that takes its place on small grids (conv_sx_small.hip.hpp: reduction split over the waves) -
    against a float64 restatement of `old + act(conv(x) + bias) * mask` (modules.py:200-209, 447-466;
    attentions.py:66-75, 419-427)."""
    from phoonnx_amd.session import test_conv1d_sx_planar
    rng = np.random.default_rng(B * 1000 + Cin + Cout + T)
    x = rng.standard_normal((B, Cin, T)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, K)) / np.sqrt(Cin * K)).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32)
    lens = np.array([T] + [int(v) for v in rng.integers(T // 3, T, B - 1)], np.int64)
    rs = Cout if row_split is None else row_split
    old = None
    if opt.get("accumulate"):
        old = rng.standard_normal((B, Cout, T)).astype(np.float32)
    elif opt.get("residual"):
        old = rng.standard_normal((B, Cout, T)).astype(np.float32)
    got, planes = test_conv1d_sx_planar(x, w, bias, lens=lens, old=old, row_split=row_split, pl_rows=pl_rows, small=small, **opt)
    v = _conv_same_f64(x, w, bias, 1)
    if opt.get("relu"):
        v = np.maximum(v, 0)
    mk = (np.arange(T)[None, :] < lens[:, None]).astype(np.float64)[:, None, :] if opt.get("mask") else 1.0
    o64 = np.zeros_like(v) if old is None else old.astype(np.float64)
    if opt.get("residual"):
        want = o64 + v * mk
    elif opt.get("coupling"):
        want = (o64 - v * mk) * mk
    else:
        want = o64 + v * mk
        if opt.get("store2"):
            want[:, rs:] = (v * mk)[:, rs:]
    scale = float(np.abs(want).max())
    assert scale > 1.0
    np.testing.assert_allclose(got, want, atol=2e-5 * scale, rtol=0)
    if pl_rows:
        ref = want[:, rs:rs + pl_rows] if opt.get("planes_of2") else want[:, :pl_rows]
        # (two fp16 planes carry ~22 bits of a value)
        np.testing.assert_allclose(planes, ref, atol=2e-5 * scale, rtol=0)


@pytest.mark.parametrize("small", [False, True], ids=["engine", "short-launch"])
@pytest.mark.parametrize("planes", [False, True], ids=["planar", "planes"])
@pytest.mark.parametrize("B,Cin,H,T,K", [(2, 192, 192, 300, 5), (1, 192, 192, 97, 5), (3, 96, 64, 50, 3)])
def test_conv_sx_gate_epilogue_matches_float64(B, Cin, H, T, K, planes, small):
    """Kernel level: the WN in-layer + gate (fused_add_tanh_sigmoid_multiply, commons.py:99-106 on modules.py:195-203) on the
    split-operand engine and on the short-launch kernel, against float64: acts = tanh(a + g_a) * sigmoid(b + g_b)."""
    from phoonnx_amd.session import test_conv1d_sx_gate
    rng = np.random.default_rng(B * 100 + T + K)
    x = rng.standard_normal((B, Cin, T)).astype(np.float32)
    w = (rng.standard_normal((2 * H, Cin, K)) / np.sqrt(Cin * K)).astype(np.float32)
    bias = rng.standard_normal(2 * H).astype(np.float32)
    g = rng.standard_normal((B, 2 * H)).astype(np.float32)
    got = test_conv1d_sx_gate(x, w, bias, g, small=small, planes=planes)
    v = _conv_same_f64(x, w, bias, 1) + g.astype(np.float64)[:, :, None]
    want = np.tanh(v[:, :H]) / (1.0 + np.exp(-v[:, H:]))
    np.testing.assert_allclose(got, want, atol=3e-6, rtol=0)


def test_fused_mrf_stage_is_bit_identical_to_separate_chains(monkeypatch):
    """The 32-channel ResBlock2 stage as ONE launch (conv_sx_pair_kernel<.., NCH>: every chain from one resident x tile,
    the multi-receptive-field sum in registers; opt-in, VITSMI_SX_MRF=1) against the default chain-per-launch form: same
    products, same order of additions - the same bits.  sx_rb2_ms: kernels (3, 5, 7), dilations (1, 2) / (2, 6) / (3, 12),
    290 frames = several tiles per utterance, ragged lengths."""
    g = np.load(os.path.join(GOLDEN, "sx_rb2_ms.npz"))
    outs = []
    monkeypatch.setenv("VITSMI_SX_MRF", "1")
    for off in (False, True):
        if off:
            monkeypatch.delenv("VITSMI_SX_MRF")
        s = _session("sx_rb2_ms")
        res = []
        for c in golden_cases(g):
            r = s.synthesize_batch(case_get(g, c, "ids"), case_get(g, c, "lens"), case_get(g, c, "scales"),
                                   case_get(g, c, "sid"), case_get(g, c, "noise_dp"), case_get(g, c, "noise_z"))
            res.append(r["output"])
        outs.append((res, s.stats()["total_launches"]))
        s.close()
    assert outs[0][1] < outs[1][1], "the fused form was not taken"
    for a, b in zip(outs[0][0], outs[1][0]):
        assert np.abs(a).max() > 0.02 and np.array_equal(a, b)


@pytest.mark.parametrize("preset", SX_PRESETS)
def test_sx_vocoder_only_matches_reference_z(preset):
    """BASELINE config 2 (vocoder only) on an sx voice: feed the reference's own z (already masked) to
    vits_run_vocoder and compare with the reference's waveform."""
    s = _session(preset)
    g = np.load(os.path.join(GOLDEN, preset + ".npz"))
    for c in ("b1_zero", "b3_noise"):
        z = case_get(g, c, "out_z")
        ylen = case_get(g, c, "out_y_lengths")
        mask = (np.arange(z.shape[2])[None, :] < ylen[:, None]).astype(np.float32)[:, None, :]
        got = s.vocoder(z * mask, case_get(g, c, "sid"))
        ref = case_get(g, c, "out_output")
        assert got.shape == ref.shape
        np.testing.assert_allclose(got, ref, atol=5e-5, rtol=0)
    s.close()


@pytest.mark.parametrize("preset", ALL_PRESETS)
def test_embedding_lookup_is_bit_exact(preset):
    """north_star: integer phoneme id -> embedding lookup bit-exact.  Tap "emb" is emb[ids] * sqrt(hidden) * mask
    (models.py:199): a row gather and ONE fp32 multiply, so the GPU must reproduce NumPy bit for bit."""
    from vits_oracle import VitsOracle
    s = _session(preset)
    o = VitsOracle(os.path.join(GOLDEN, preset + ".onnx"))
    emb = o.tensors["enc_p.emb.weight"]
    V, H = emb.shape
    rng = np.random.default_rng(77)
    B, T = 3, 50
    ids = rng.integers(0, V, (B, T)).astype(np.int64)
    ids[0, :4] = [0, V - 1, 1, V - 2]                          # table edges
    lens = np.array([T, 31, 1], np.int64)
    sid = np.zeros(B, np.int64) if s.hparam("n_speakers") > 1 else None
    # (length scale 0.02: one frame per token - the tap does not depend on it, and the C oracle's generator, which the last
    # assertion runs as part of its one entry point, is what this test's time goes to)
    sc = np.array([0, 0.02, 0], np.float32)
    r = s.synthesize_batch(ids, lens, sc, sid, taps=("emb",))
    want = (emb[ids] * np.float32(np.sqrt(H))).transpose(0, 2, 1)
    want = want * (np.arange(T)[None, None, :] < lens[:, None, None])
    assert r["emb"].shape == (B, H, T) and r["emb"].dtype == np.float32
    assert np.array_equal(r["emb"], want.astype(np.float32))
    ro = o.infer(ids, lens, sc, sid)
    assert np.array_equal(ro["emb"], r["emb"])                  # the oracle agrees bit for bit as well
    assert np.array_equal(ro["y_lengths"], r["y_lengths"])
    s.close()


def test_session_run_duck_types_onnxruntime():
    # the exact call sequence of phoonnx/voice.py:347-377
    s = _session("tiny_rb1")
    g = np.load(os.path.join(GOLDEN, "tiny_rb1.npz"))
    expected_args = [i.name for i in s.get_inputs()]
    assert expected_args == ["input", "input_lengths", "scales"]
    ids = case_get(g, "b1_zero", "ids")
    args = {"input": ids, "input_lengths": np.array([ids.shape[1]], dtype=np.int64),
            "scales": np.array([0.0, 1.0, 0.0], dtype=np.float32),
            "langid": np.array([0], dtype=np.int64), "sid": np.array([0], dtype=np.int64)}
    args = {k: v for k, v in args.items() if k in expected_args}
    out = s.run(None, args)
    assert isinstance(out, list) and out[0].ndim == 4 and out[0].shape[:3] == (1, 1, 1)
    audio = out[0].squeeze()
    np.testing.assert_allclose(audio, case_get(g, "b1_zero", "out_output").squeeze(), atol=WAVE_TOL)
    from phoonnx_amd import SessionError
    with pytest.raises(SessionError):
        s.run(None, dict(args, bogus=np.zeros(1)))
    with pytest.raises(SessionError):
        s.run(None, {"input": ids})
    with pytest.raises(SessionError):  # id out of the embedding's range
        s.run(None, dict(args, input=ids + 100000))
    s.close()


def test_multispeaker_requires_sid_and_uses_it():
    s = _session("tiny_rb2_ms")
    assert [i.name for i in s.get_inputs()] == ["input", "input_lengths", "scales", "sid"]
    ids = np.arange(1, 13, dtype=np.int64)[None]
    lens = np.array([12], np.int64)
    sc = np.array([0, 1.2, 0], np.float32)
    from phoonnx_amd import SessionError
    with pytest.raises(SessionError):
        s.synthesize_batch(ids, lens, sc)
    a = s.synthesize_batch(ids, lens, sc, sid=np.array([0], np.int64))["output"]
    b = s.synthesize_batch(ids, lens, sc, sid=np.array([3], np.int64))["output"]
    assert a.shape != b.shape or np.abs(a - b).max() > 1e-3
    s.close()


def test_determinism_and_seeded_noise():
    s = _session("tiny_rb1")
    ids = np.arange(3, 33, dtype=np.int64)[None]
    lens = np.array([30], np.int64)
    z0 = s.synthesize_batch(ids, lens, np.array([0, 1, 0], np.float32))["output"]
    z1 = s.synthesize_batch(ids, lens, np.array([0, 1, 0], np.float32))["output"]
    assert np.array_equal(z0, z1)                      # zero noise: bit-identical run to run
    s.set_seed(7)
    n0 = s.synthesize_batch(ids, lens, np.array([0.667, 1, 0.8], np.float32))
    n1 = s.synthesize_batch(ids, lens, np.array([0.667, 1, 0.8], np.float32))
    # like the graph's unseeded RandomNormalLike nodes, successive calls draw fresh noise
    assert n0["output"].shape != n1["output"].shape or not np.array_equal(n0["output"], n1["output"])
    assert np.isfinite(n0["output"]).all() and np.abs(n0["output"]).max() <= 1.0
    s.close()


def test_padded_batch_interior_equals_single(tmp_path):
    # inside a padded batch an item equals its batch-1 rendering except the tail
    s = _session("tiny_rb1")
    rng = np.random.default_rng(3)
    ids = np.zeros((2, 60), np.int64)
    ids[0] = rng.integers(1, 200, 60)
    ids[1, :35] = rng.integers(1, 200, 35)
    sc = np.array([0, 3.0, 0], np.float32)
    both = s.synthesize_batch(ids, np.array([60, 35], np.int64), sc)
    one = s.synthesize_batch(ids[1:2, :35].copy(), np.array([35], np.int64), sc)
    assert both["y_lengths"][1] == one["y_lengths"][0]
    hop = s.hparam("hop")
    frames = int(one["y_lengths"][0])
    rf = 40  # frames: > one-sided receptive field of the tiny generator (conv_pre 3 + MRF stacks ~25)
    assert frames > 2 * rf, frames
    n = (frames - rf) * hop
    np.testing.assert_allclose(both["output"][1, 0, 0, :n], one["output"][0, 0, 0, :n], atol=1e-5)
    # ... while the tail differs because the generator is not masked (models.py:720)
    assert both["output"].shape[3] > one["output"].shape[3]
    s.close()


def test_timing_levels():
    """vits_set_timing: 1 = stage marks + events around every conv launch (per-launch records), 2 = stage marks only,
    0 = nothing recorded; the audio does not depend on it."""
    s = _session("sx_rb2_ms")
    rng = np.random.default_rng(21)
    B, T = 2, 48
    ids = rng.integers(1, 200, (B, T)).astype(np.int64)
    lens = np.array([T, T - 9], np.int64)
    sid = np.array([1, 3], np.int64)
    sc = np.array([0, 1.5, 0], np.float32)
    ref = s.synthesize_batch(ids, lens, sc, sid)["output"]
    s.set_timing(True)
    a = s.synthesize_batch(ids, lens, sc, sid)["output"]
    st = s.stats()
    recs = s.launch_records()
    assert st["total_ms"] > 0 and st["conv_ms"] > 0 and len(recs) == st["conv_launches"] > 0
    assert all(r["ms"] > 0 and r["flops"] > 0 for r in recs)
    s.set_timing(2)
    b = s.synthesize_batch(ids, lens, sc, sid)["output"]
    st2 = s.stats()
    assert st2["total_ms"] > 0 and st2["dec_ms"] > 0 and st2["conv_ms"] == 0 and s.launch_records() == []
    s.set_timing(False)
    assert np.array_equal(ref, a) and np.array_equal(ref, b)
    s.close()


def test_pipelined_session_equals_plain_session():
    # two handles / streams sharing one weight arena, each rendering half of the batch: same utterances, same audio
    # (equal-length rows, so every part pads to the same frame count; zero noise scales make it deterministic)
    from phoonnx_amd import PipelinedSession
    s = _session("tiny_rb2_ms")
    p = PipelinedSession(s, parts=2)
    assert p.parts[1].arena_device() == s.arena_device() and p.parts[1].arena_bytes() == s.arena_bytes()
    rng = np.random.default_rng(8)
    B, T = 5, 40
    ids = rng.integers(1, 200, (B, T)).astype(np.int64)
    lens = np.full(B, T, np.int64)
    sid = rng.integers(0, 4, B).astype(np.int64)
    sc = np.array([0, 2.0, 0], np.float32)
    plain = s.synthesize_batch(ids, lens, sc, sid)
    piped = p.synthesize_batch(ids, lens, sc, sid)
    assert np.array_equal(plain["y_lengths"], piped["y_lengths"])
    hop = s.hparam("hop")
    rf = 40
    for b in range(B):
        n = (int(plain["y_lengths"][b]) - rf) * hop  # the unmasked generator's tail depends on the padding after it
        assert n > 0
        np.testing.assert_allclose(piped["output"][b, 0, 0, :n], plain["output"][b, 0, 0, :n], atol=1e-5)
    p.close()


def test_pipelined_session_free_running_steps():
    # one host thread per part, several passes back to back: every pass reports the frame counts of a plain run
    import ctypes as C
    from phoonnx_amd import PipelinedSession
    s = _session("tiny_rb1")
    p = PipelinedSession(s, parts=2)
    rng = np.random.default_rng(9)
    B, T = 6, 32
    ids = rng.integers(1, 200, (B, T)).astype(np.int64)
    lens = np.full(B, T, np.int64)
    hip = C.CDLL("libamdhip64.so")    # device copies of the inputs (the engine's own runtime; no torch in the tests)
    dptr = []
    for arr in (ids, lens):
        d = C.c_void_p()
        assert hip.hipMalloc(C.byref(d), C.c_size_t(arr.nbytes)) == 0
        assert hip.hipMemcpy(d, C.c_void_p(arr.ctypes.data), C.c_size_t(arr.nbytes), 1) == 0
        dptr.append(d)
    sc = np.array([0, 1.7, 0], np.float32)   # zero noise: durations are deterministic
    ref = s.synthesize_batch(ids, lens, sc)["y_lengths"]
    got = p.run_device_steps(dptr[0].value, dptr[1].value, B, T, sc, steps=4)
    assert got.shape == (4, B)
    for k in range(4):
        assert np.array_equal(got[k], ref)
    # request-level pipelining: whole passes dealt to the handles in turn (pass k on part k mod 2), odd pass count
    got = p.run_device_steps(dptr[0].value, dptr[1].value, B, T, sc, steps=5, alternate=True)
    assert got.shape == (5, B)
    for k in range(5):
        assert np.array_equal(got[k], ref)
    assert all(pp.last_y_lengths().shape == (B,) for pp in p.parts)      # each handle rendered the WHOLE batch
    p.close()
    for d in dptr:
        hip.hipFree(d)


def test_vocoder_only_matches_oracle():
    from vits_oracle import VitsOracle
    for preset in ("tiny_rb1", "tiny_rb2_ms"):
        s = _session(preset)
        o = VitsOracle(os.path.join(GOLDEN, preset + ".onnx"))
        rng = np.random.default_rng(11)
        z = rng.standard_normal((2, s.hparam("inter"), 37)).astype(np.float32)
        sid = np.array([1, 2], np.int64) if s.hparam("n_speakers") > 1 else None
        got = s.vocoder(z, sid)
        ref = o.vocoder(z, sid)
        assert got.shape == ref.shape
        np.testing.assert_allclose(got, ref, atol=1e-4)
        s.close()


# ------------------------------------------------------------------ the user-facing API on top

def test_ttsvoice_load_and_synthesize_wav(tmp_path):
    """TTSVoice.load() -> synthesize_wav() through the engine (config 1 of BASELINE.json, on the GPU):
    the WAV holds exactly the int16 rendering of what the session returns for each sentence."""
    import io
    import json
    import shutil
    import wave
    from phoonnx_amd.config import SynthesisConfig
    from phoonnx_amd.voice import TTSVoice
    model = tmp_path / "voice.onnx"
    shutil.copy(os.path.join(GOLDEN, "tiny_rb1.onnx"), model)
    id_map = {chr(97 + i): i + 4 for i in range(26)}
    id_map.update({"_": 0, "^": 1, "$": 2, " ": 3})
    (tmp_path / "voice.onnx.json").write_text(json.dumps({
        "phoneme_type": "raw", "lang_code": "en", "alphabet": "ipa", "audio": {"sample_rate": 22050},
        "phoneme_id_map": id_map, "pad": "_", "blank": "_", "bos": "^", "eos": "$",
        "inference": {"noise_scale": 0.0, "length_scale": 1.5, "noise_w": 0.0}}))
    voice = TTSVoice.load(str(model))
    voice.dedupe_sentences = True
    text = "hello world. this is a test."
    buf = io.BytesIO()
    with wave.open(buf, "wb") as w:
        voice.synthesize_wav(text, w)
    with wave.open(io.BytesIO(buf.getvalue())) as r:
        assert (r.getframerate(), r.getsampwidth(), r.getnchannels()) == (22050, 2, 1)
        frames = np.frombuffer(r.readframes(r.getnframes()), np.int16)
    chunks = list(voice.synthesize(text))
    assert len(chunks) == 2
    assert np.array_equal(frames, np.concatenate([c.audio_int16_array for c in chunks]))
    # sentence batching (extension f1): same ids in one padded batch; each item trimmed to its own length.
    # Interior samples agree with the sequential rendering; the unmasked generator differs near the end.
    batched = list(voice.synthesize(text, batch_sentences=True))
    assert [len(c.audio_float_array) for c in batched] == [len(c.audio_float_array) for c in chunks]
    longest = int(np.argmax([len(c.audio_float_array) for c in chunks]))
    np.testing.assert_allclose(batched[longest].audio_float_array, chunks[longest].audio_float_array, atol=2e-4)
    # device-side post-processing (extension f2): the same batch leaves the GPU as PCM16 only; the WAV frames
    # are bit-identical to the int16 rendering of the batched float path
    buf2 = io.BytesIO()
    with wave.open(buf2, "wb") as w:
        voice.synthesize_wav(text, w, device_pcm16=True)
    with wave.open(io.BytesIO(buf2.getvalue())) as r:
        assert (r.getframerate(), r.getsampwidth(), r.getnchannels()) == (22050, 2, 1)
        frames2 = np.frombuffer(r.readframes(r.getnframes()), np.int16)
    assert np.array_equal(frames2, np.concatenate([c.audio_int16_array for c in batched]))
    # a reference-style voice object: session.run() returns rank-4, squeeze() gives [S]
    ids = voice.phonemes_to_ids(list("hello"))
    audio = voice.phoneme_ids_to_audio(ids, SynthesisConfig())
    assert audio.ndim == 1 and audio.dtype == np.float32 and len(audio) % voice.session.hparam("hop") == 0
    voice.session.close()


@pytest.mark.parametrize("normalize,volume", [(True, 1.0), (True, 0.5), (False, 1.0), (False, 2.5)])
def test_device_pcm16_matches_numpy_postprocessing(normalize, volume):
    """f2: peak-normalise / volume / clip / int16 on the GPU is bit-identical to what
    TTSVoice.synthesize + AudioChunk.audio_int16_array compute with NumPy (voice.py:271-282, 88-91)."""
    from phoonnx_amd.config import SynthesisConfig
    from phoonnx_amd.voice import AudioChunk, TTSVoice
    s = _session("tiny_rb1")
    rng = np.random.default_rng(12)
    ids = np.zeros((3, 40), np.int64)
    lens = np.array([40, 21, 9], np.int64)
    for b in range(3):
        ids[b, :lens[b]] = rng.integers(1, 200, lens[b])
    r = s.synthesize_batch(ids, lens, np.array([0.3, 1.4, 0.5], np.float32))
    B, S = r["output"].shape[0], r["output"].shape[3]
    pcm = s.last_pcm16(normalize, volume, shape=(B, S))
    hop = s.hparam("hop")
    post = TTSVoice._postprocess  # the NumPy path of the interface mirror (pinned to the reference's bytes)
    syn = SynthesisConfig(normalize_audio=normalize, volume=volume)
    for b in range(B):
        n = int(r["y_lengths"][b]) * hop
        ref = AudioChunk(22050, 2, 1, post(None, r["output"][b, 0, 0, :n], syn)).audio_int16_array
        assert np.array_equal(pcm[b, :n], ref), (b, np.abs(pcm[b, :n].astype(int) - ref.astype(int)).max())
        assert not pcm[b, n:].any()
    s.close()


def test_ttsvoice_loads_from_the_onnx_alone_and_rejects_inconsistent_json(tmp_path):
    """f3 end to end: sx_rb2_ms.onnx carries a real phoneme_id_map in its metadata_props (as export_onnx.py:335-345
    writes it) and no JSON: TTSVoice.load rebuilds the config from the file and synthesizes on the sx engine.  A JSON
    that contradicts the file is rejected with ValueError instead of a warning."""
    import json
    import shutil
    from phoonnx_amd.config import SynthesisConfig
    from phoonnx_amd.voice import TTSVoice
    model = tmp_path / "voice.onnx"
    shutil.copy(os.path.join(GOLDEN, "sx_rb2_ms.onnx"), model)
    voice = TTSVoice.load(str(model))
    assert voice.config.num_speakers == 4 and voice.config.sample_rate == 22050
    assert voice.config.phoneme_id_map["a"] == 4 and voice.config.include_whitespace
    assert voice.session.hparam("gen_sx") == 1
    voice.dedupe_sentences = True
    syn = SynthesisConfig(speaker_id=2, noise_scale=0.0, noise_w_scale=0.0, length_scale=1.5)
    chunks = list(voice.synthesize("hello world. again, hello?", syn))
    # (one chunk per piece between delimiters, base.py:62-86: "hello world" | "again" | "hello"; hop 32: short audio)
    assert len(chunks) == 3 and all(len(c.audio_float_array) > 100 for c in chunks)
    # the same ids by hand through the session: identical audio
    ids = voice.phonemes_to_ids(voice.phonemize("hello world. again, hello?")[0])
    raw = voice.phoneme_ids_to_audio(ids, syn)
    assert np.array_equal(chunks[0].audio_float_array, voice._postprocess(raw, syn))
    voice.session.close()
    for bad in ({"num_speakers": 2}, {"audio": {"sample_rate": 16000}}, {"num_symbols": 99},
                {"phoneme_id_map": {"a": 5000}}):
        cfg = {"phoneme_type": "raw", "lang_code": "en", "alphabet": "ipa", "audio": {"sample_rate": 22050},
               "num_speakers": 4, "num_symbols": 256, "phoneme_id_map": {"a": 4}}
        cfg.update(bad)
        (tmp_path / "voice.onnx.json").write_text(json.dumps(cfg))
        with pytest.raises(ValueError, match="inconsistent voice"):
            TTSVoice.load(str(model))
    TTSVoice.load(str(model), strict=False).session.close()    # opt-out: the reference's behaviour (no checks)


@pytest.mark.gpu
@pytest.mark.parametrize("preset", ["tiny_dp", "sx_rb2_ms"])
def test_renamed_graph_renders_the_same_waveform(tmp_path, preset):
    """Structure-keyed weight resolution end to end: a fixture whose nodes were renamed `<op>_<n>` (Piper-era style,
    tools/rename_nodes.py) renders bit for bit what the file with module-path names renders."""
    import sys
    from phoonnx_amd import MiSession
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from rename_nodes import rename
    g = np.load(os.path.join(GOLDEN, preset + ".npz"))
    case = golden_cases(g)[-1]
    args = [case_get(g, case, k) for k in ("ids", "lens", "scales", "sid", "noise_dp", "noise_z")]
    p2 = tmp_path / "renamed.onnx"
    p2.write_bytes(rename(open(os.path.join(GOLDEN, preset + ".onnx"), "rb").read()))
    a = MiSession(os.path.join(GOLDEN, preset + ".onnx"))
    ra = a.synthesize_batch(*args)
    a.close()
    b = MiSession(str(p2))
    rb = b.synthesize_batch(*args)
    b.close()
    assert np.array_equal(ra["y_lengths"], rb["y_lengths"]) and np.array_equal(ra["output"], rb["output"])


def test_persistent_pair16_form_on_multi_tile_launches():
    """conv_sx_pair16_kernel<.., PERSIST> (opt-in, VITSMI_PAIR16_PERSIST=1: a workgroup walks several tiles and requests tile
    i + 1's x tile under tile i's second conv, with exact vmcnt accounting; measured slower) on launches of more
    tiles than the persistent grid has workgroups - ragged ends, PAIR and CHAIN, both arithmetics - against the 32x32x16 pair
    kernel.  This is the code that exposed hazard 5 (a v_readlane in front of an inline-asm load): wrong addresses, not wrong
    numbers, so the test's first duty is to finish.  In a child process: the switch is read once per process."""
    import subprocess
    import sys
    code = r'''
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from phoonnx_amd.session import test_conv_pair_sx
rng = np.random.default_rng(3)
for kern, C, K, d1, d2, chain, B, T, tol in (("pair16", 32, 3, 1, 1, False, 2, 150001, 2e-5), ("pair16", 64, 7, 3, 1, False, 3, 50003, 2e-5),
                                             ("pair16", 32, 5, 2, 6, True, 2, 140000, 2e-5), ("pair16_f16", 32, 3, 1, 1, False, 2, 150001, 2e-2),
                                             ("pair16_f16", 64, 3, 1, 2, True, 2, 80000, 2e-2), ("pair16", 32, 11, 5, 1, False, 1, 140009, 2e-5)):
    x = rng.standard_normal((B, C, T)).astype(np.float32)
    w1 = (rng.standard_normal((C, C, K)) / np.sqrt(C * K)).astype(np.float32)
    w2 = (rng.standard_normal((C, C, K)) / np.sqrt(C * K)).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)
    ref = test_conv_pair_sx(x, w1, b, w2, b, dil1=d1, dil2=d2, chain=chain, kernel="pair")
    got = test_conv_pair_sx(x, w1, b, w2, b, dil1=d1, dil2=d2, chain=chain, kernel=kern)
    err = float(np.abs(got - ref).max())
    assert err < tol, (kern, C, K, chain, err)
print("PERSIST_OK")
'''
    env = dict(os.environ, VITSMI_PAIR16_PERSIST="1")
    r = subprocess.run([sys.executable, "-c", code, ROOT], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "PERSIST_OK" in r.stdout, (r.returncode, r.stdout[-300:], r.stderr[-1500:])


def test_the_xcd_deal_of_the_time_tiles_changes_no_bit():
    """Round 6: time tiles are dealt to the 8 XCDs in groups of 2^xgs consecutive tiles (sx_xcd_tile; VITSMI_XCD_GROUP, default
    4) so that neighbouring tiles - which share their halo columns - meet in one L2.  That is a permutation of which workgroup
    renders which tile: every group size must give the same bits, on the plane-input engine (several row tiles per time tile),
    the raw-input engine and the fused pair / chain kernels, for tile counts that are not multiples of a round (8 x group) and
    for tensors too short to group at all.  In child processes: the switch is read once per process."""
    import subprocess
    import sys
    import hashlib
    code = r'''
import os, sys, hashlib
import numpy as np
sys.path.insert(0, sys.argv[1])
from phoonnx_amd.session import test_conv1d_sx, test_conv_pair_sx
rng = np.random.default_rng(21)
h = hashlib.sha256()
for Cin, Cout, K, dil, B, T in ((256, 256, 3, 1, 2, 9001), (128, 128, 11, 5, 3, 13999), (64, 64, 7, 3, 2, 30011), (128, 128, 3, 1, 1, 700)):
    x = rng.standard_normal((B, Cin, T)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, K)) / np.sqrt(Cin * K)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    pad = dil * (K - 1) // 2
    h.update(test_conv1d_sx(x, w, b, dil=dil, pad_l=pad, residual=True, precision="f16x3").tobytes())
for C, K, d1, d2, chain, B, T in ((32, 3, 1, 1, False, 2, 70001), (64, 7, 3, 1, False, 2, 30003), (32, 5, 2, 6, True, 3, 50000), (32, 3, 1, 2, True, 1, 900)):
    x = rng.standard_normal((B, C, T)).astype(np.float32)
    w1 = (rng.standard_normal((C, C, K)) / np.sqrt(C * K)).astype(np.float32)
    w2 = (rng.standard_normal((C, C, K)) / np.sqrt(C * K)).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)
    h.update(test_conv_pair_sx(x, w1, b, w2, b, dil1=d1, dil2=d2, chain=chain, kernel="pair").tobytes())
print("XCD_DIGEST", h.hexdigest())
'''
    digests = {}
    for g in ("1", "2", "4", "16"):
        env = dict(os.environ, VITSMI_XCD_GROUP=g)
        r = subprocess.run([sys.executable, "-c", code, ROOT], capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0 and "XCD_DIGEST" in r.stdout, (g, r.returncode, r.stdout[-300:], r.stderr[-1500:])
        digests[g] = r.stdout.split("XCD_DIGEST")[1].split()[0]
    assert len(set(digests.values())) == 1, digests
