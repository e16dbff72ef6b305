#!/usr/bin/env python3
"""Randomised shape sweep of the split-exact conv engine against the oracle's conv (run on the GPU box):
    python tools/sx_fuzz.py [--cases 150] [--seed 0]
Covers every tile config, plane- and raw-input kernels, residual / planes / input-activation epilogues,
transposed convs, sequence lengths from 1 to a few tiles.  Exits non-zero on the first mismatch."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=150)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    from phoonnx_amd.session import test_conv1d_sx, test_conv_transpose1d
    from vits_oracle import conv1d, conv_transpose1d
    rng = np.random.default_rng(a.seed)
    worst = 0.0
    for case in range(a.cases):
        B = int(rng.integers(1, 4))
        Cin = int(rng.choice([16, 32, 48, 64, 80, 96, 128, 192, 256]))
        Cout = int(rng.choice([32, 64, 96, 128, 192, 256, 384]))
        K = int(rng.choice([1, 2, 3, 5, 7, 11]))
        dil = int(rng.choice([1, 1, 2, 3, 5, 12]))
        T = int(rng.choice([1, 2, 7, 31, 64, 255, 256, 257, 300, 513, 700]))
        if (K - 1) * dil > 120:
            dil = 1
        x = rng.standard_normal((B, Cin, T)).astype(np.float32)
        w = (rng.standard_normal((Cout, Cin, K)) / np.sqrt(Cin * K)).astype(np.float32)
        b = rng.standard_normal(Cout).astype(np.float32) if rng.random() < 0.7 else None
        pad = dil * (K - 1) // 2
        ref = conv1d(x, w, b, dil=dil, pad_l=pad, pad_r=dil * (K - 1) - pad)
        variant = int(rng.integers(0, 4))
        kw = {}
        want = ref
        if variant == 1:
            kw["planes_slope"] = 0.1
            want = np.where(ref > 0, ref, ref * np.float32(0.1))
        elif variant == 2 and Cin == Cout:
            kw["residual"] = True
            want = ref + x
        elif variant == 3 and Cin <= 64:
            kw["in_slope"] = 0.1
            xa = np.where(x > 0, x, x * np.float32(0.1)).astype(np.float32)
            want = conv1d(xa, w, b, dil=dil, pad_l=pad, pad_r=dil * (K - 1) - pad)
        got = test_conv1d_sx(x, w, b, dil=dil, pad_l=pad, **kw)
        err = float(np.abs(got - want).max())
        worst = max(worst, err)
        tag = f"case {case}: B={B} Cin={Cin} Cout={Cout} K={K} dil={dil} T={T} {kw}"
        if not np.allclose(got, want, atol=3e-5, rtol=1e-5):
            print("MISMATCH", tag, "max err", err)
            return 1
        if case % 25 == 0:
            print("ok", tag, f"err {err:.1e}", flush=True)
    for case in range(a.cases // 5):
        B = int(rng.integers(1, 3))
        Cin = int(rng.choice([32, 64, 128, 256]))
        Cout = int(rng.choice([32, 64, 128]))
        u = int(rng.choice([2, 4, 8]))
        K = 2 * u
        T = int(rng.choice([1, 5, 33, 129, 300]))
        x = rng.standard_normal((B, Cin, T)).astype(np.float32)
        w = (rng.standard_normal((Cin, Cout, K)) / np.sqrt(Cin * K / u)).astype(np.float32)
        b = rng.standard_normal(Cout).astype(np.float32)
        got = test_conv_transpose1d(x, w, b, u, sx=True)
        want = conv_transpose1d(x, w, b, u, (K - u) // 2)
        err = float(np.abs(got - want).max())
        worst = max(worst, err)
        if not np.allclose(got, want, atol=3e-5, rtol=1e-5):
            print(f"MISMATCH convT case {case}: B={B} Cin={Cin} Cout={Cout} u={u} T={T} max err {err}")
            return 1
    print(f"all {a.cases} conv + {a.cases // 5} transposed-conv cases agree; worst max-abs error {worst:.2e}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
