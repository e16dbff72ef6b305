#!/usr/bin/env python3
"""Micro-benchmark of the fused pair / chain launches (conv_sx_pair_kernel) on the generator's raw-format stage shapes
(run on the GPU box):  python tools/pair_bench.py [--frames 860] [--batch 32]
Prints ms per launch, fp32-equivalent TFLOP/s and the algorithmic bytes/s (x read once + out written once)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=860)
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    from phoonnx_amd.session import test_conv_pair_sx
    rng = np.random.default_rng(0)
    cases = [  # (C, samples per frame at that stage, K, dil1, dil2, chain)
        (64, 128, 3, 1, 1, False), (64, 128, 7, 3, 1, False), (64, 128, 11, 5, 1, False),
        (32, 256, 3, 1, 1, False), (32, 256, 7, 3, 1, False), (32, 256, 11, 5, 1, False),
        (64, 64, 3, 1, 2, True), (64, 64, 5, 2, 6, True), (32, 256, 3, 1, 2, True), (32, 256, 5, 2, 6, True),
        (32, 256, 7, 3, 12, True),
    ]
    for C, up, K, d1, d2, chain in cases:
        T = a.frames * up
        x = rng.standard_normal((a.batch, C, T)).astype(np.float32)
        w1 = (rng.standard_normal((C, C, K)) / np.sqrt(C * K)).astype(np.float32)
        w2 = (rng.standard_normal((C, C, K)) / np.sqrt(C * K)).astype(np.float32)
        b = rng.standard_normal(C).astype(np.float32)
        try:
            _, ms = test_conv_pair_sx(x, w1, b, w2, b, dil1=d1, dil2=d2, chain=chain, timed=True)
        except Exception as e:  # noqa: BLE001
            print(f"C={C} K={K} d=({d1},{d2}) {'chain' if chain else 'pair '}: not fused ({e})")
            continue
        flop = 2.0 * 2 * C * C * K * T * a.batch
        byts = 2.0 * 4 * C * T * a.batch
        print(f"C={C:3d} T={T:7d} K={K:2d} d=({d1},{d2:2d}) {'chain' if chain else 'pair '}: {ms:7.3f} ms  "
              f"{flop / ms / 1e9:7.1f} TFLOP/s  {byts / ms / 1e9:6.2f} TB/s algorithmic")


if __name__ == "__main__":
    main()
