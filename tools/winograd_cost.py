#!/usr/bin/env python3
"""Costing of a Winograd / Toom-Cook F(2,3) form of the generator's 128-row convs from measurements of
the engine as it is - no new kernel.  Run on the GPU box.

F(2,3) on the stride-`dil` subsequence forms 2 outputs from 4 transformed inputs with 4 products instead of 6: per output PAIR a
k = 3 conv costs 4 products, k = 7 costs 10 (two k = 3 blocks + one single tap), k = 11 costs 16 (three blocks + two taps).  Its
main loop is therefore EXACTLY the loop of this engine run over T / 2 columns with K' = 4 / 10 / 16 "taps" (same MFMA count per
column, same weight bytes per column, same x-stage traffic per column - the transformed input has two values per input position),
and its epilogue (output transform, bias, residual, split, stores) handles all T output columns.  So

    t_winograd >= t_engine(K', T / 2) + e(T) / 2          (the second term: the half of the per-column fixed cost that the
                                                           proxy launch at T / 2 does not pay)

with e(T) from a linear fit t_engine(K, T) = e(T) + m(T) K over K = 1 .. 11, and that bound EXCLUDES what the form adds: the input
transform (2 transformed values per input, each split into two fp16 planes: ~2x the producer epilogue's split work, or ~25 % of
the loop's issue slots if done in the consumer), the 2-column halo a tile of transformed inputs needs from its neighbour
(+12.5 % of a 256-column producer tile), and the output transform.  Kill criterion (VERDICT): < 1.15x; keep: >= 1.25x.

    python tools/winograd_cost.py [--batch 32] [--frames 860] [--iters 10] [--json out.json]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=860)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    from phoonnx_amd.session import bench_conv1d_sx
    F16X3 = 128
    rows = []
    for name, C, T in (("256 ch (stage 1)", 256, a.frames * 8), ("128 ch (stage 2)", 128, a.frames * 64)):
        for epi, tag in ((0, "planes epilogue (c1)"), (8, "residual + raw + planes epilogue (c2)")):
            def t(K, TT, dil=1):
                best = 1e9
                for _ in range(3):
                    ms, _cfg = bench_conv1d_sx(a.batch, C, C, TT, K, dil, epi | F16X3, a.iters)
                    best = min(best, ms)
                return best
            ks = [1, 3, 5, 7, 9, 11]
            ts = [t(K, T) for K in ks]
            m, e = np.polyfit(ks, ts, 1)
            print(f"{name}, {tag}: T = {T}, t(K) ms = " + " ".join(f"K{K}:{v:.3f}" for K, v in zip(ks, ts)) +
                  f"  fit: e = {e:.3f} ms fixed + {m:.4f} ms per tap", flush=True)
            for K, Kp in ((3, 4), (7, 10), (11, 16)):
                tw = t(Kp, T // 2) + e / 2
                tk = ts[ks.index(K)]
                rows.append({"channels": C, "epilogue": tag, "K": K, "T": T, "t_engine_ms": tk, "products_per_pair": Kp,
                             "t_proxy_half_T_ms": tw - e / 2, "fixed_ms": e, "per_tap_ms": m,
                             "t_winograd_lower_bound_ms": tw, "speedup_upper_bound": tk / tw})
                print(f"    k = {K:2d}: engine {tk:.3f} ms; F(2,3) loop as K' = {Kp} over T / 2: {tw - e / 2:.3f} + e / 2 = {tw:.3f} ms "
                      f"-> at most x{tk / tw:.3f} before the transforms and the halo", flush=True)
    if a.json:
        json.dump({"batch": a.batch, "frames": a.frames, "rows": rows}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
