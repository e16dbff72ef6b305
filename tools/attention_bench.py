#!/usr/bin/env python3
"""A/B timing of the two attention kernels through the test hook (vits_test_attention16: HIP events around 50 launches):
attention_relpos16_kernel (f16x3 products on v_mfma_f32_16x16x32_f16) against attention_relpos_kernel (fp32 MFMA), 2 heads x 96
channels, window 4.  VITSMI_ATT16_NS / VITSMI_ATT16_QT select the LDS stages / query tiles per wave; a library built with
-DATT16_PROF=1 (python -m phoonnx_amd.build --variant x.so "-DATT16_PROF=1", VITSMI_LIB=x.so) prints per-workgroup cycle stamps.
Run on the GPU box: python tools/attention_bench.py"""
import numpy as np, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phoonnx_amd.session import test_attention16
rng = np.random.default_rng(0)
for B, T in ((32, 500), (1, 500), (32, 128), (64, 300), (32, 256), (1, 256)):
    heads, dk = 2, 96
    C = heads * dk
    qkv = rng.standard_normal((B, 3 * C, T)).astype(np.float32)
    rk = (rng.standard_normal((9, dk)) * dk ** -0.5).astype(np.float32)
    rv = (rng.standard_normal((9, dk)) * dk ** -0.5).astype(np.float32)
    lens = np.full(B, T, np.int64)
    o1, m1 = test_attention16(qkv, heads, rk, rv, lens, kernel=1, reps=50)
    o0, m0 = test_attention16(qkv, heads, rk, rv, lens, kernel=0, reps=50)
    print(f"B={B} T={T}: att16 {m1*1e3:.1f} us  old {m0*1e3:.1f} us  maxdiff {np.abs(o1-o0).max():.2e}", flush=True)
