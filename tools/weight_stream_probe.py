#!/usr/bin/env python3
"""Item 7 of VERDICT r3 (the per-CU shared weight stream of the 128-row tiles): the evidence for NOT building it.
Times the two dominant conv shapes of the headline voice (256 and 128 channels, k = 3, planes epilogue and residual epilogue) on
the 16x16x32 loop of conv_sx_kernel.  Run three ways:
  1. as shipped                               python tools/weight_stream_probe.py
  2. VITSMI_LIB=<-DSX_NOA=1 build>            weights fetched for the first step only (wrong results, timing only): an UPPER
                                              bound on what any cheaper weight delivery can give
  3. under rocprofv3 --pmc ...                L2 hit rate of the launch, LDS / vector-memory instruction counts, wait cycles
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phoonnx_amd.session import bench_conv1d_sx  # noqa: E402


def main():
    B, F = 32, 860
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    for name, C, T, K, dil in (("256 ch k3", 256, F * 8, 3, 1), ("256 ch k11 d5", 256, F * 8, 11, 5), ("128 ch k3", 128, F * 64, 3, 1),
                               ("128 ch k7 d3", 128, F * 64, 7, 3)):
        for tag, dbg in (("planes", 128), ("residual+planes", 128 | 8)):
            ms, cfg = bench_conv1d_sx(B, C, C, T, K, dil, dbg, iters)
            tf = 2.0 * B * C * C * K * T / (ms * 1e-3) / 1e12
            print(f"{name:14s} {tag:16s} cfg{cfg} {ms:7.3f} ms  {tf:6.1f} TFLOP/s fp32-equivalent ({tf / 838.9:.3f} of the f16x3 roof)", flush=True)


if __name__ == "__main__":
    main()
