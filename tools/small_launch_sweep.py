#!/usr/bin/env python3
"""Sweep of conv_sx()'s short-launch limit (VITSMI_SX_SMALL_MAX / vits_test_set_sx_small_max): wall ms of one
MiSession.synthesize_batch call of B x 256 ids per limit, both voices, B = 1 .. 32.  Run on the GPU box."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from phoonnx_amd import MiSession, _ffi
from phoonnx_amd.synth import write_voice
lib = _ffi.load()
default_limit = lib.vits_test_set_sx_small_max(0)   # (returns the previous value: restored after every voice)
lib.vits_test_set_sx_small_max(default_limit)
for preset in ("medium", "high"):
    path = f"/tmp/vitsmi_bench/synth_{preset}.onnx"
    os.makedirs("/tmp/vitsmi_bench", exist_ok=True)
    if not os.path.exists(path):
        write_voice(path, preset, seed=1234)
    s = MiSession(path)
    g = torch.Generator(device="cpu").manual_seed(4321)
    for B in (1, 2, 4, 8, 16, 32):
        ids = torch.randint(0, 256, (B, 256), generator=g, dtype=torch.int64).numpy()
        lens = np.full((B,), 256, np.int64)
        sc = np.array([0.667, 1.95, 0.8], np.float32)
        row = []
        for lim in (0, 384, 768, 1536, 3072, 100000):
            lib.vits_test_set_sx_small_max(lim)
            for _ in range(2):
                s.synthesize_batch(ids, lens, sc)
            per = []
            for _ in range(8):
                t0 = time.perf_counter(); s.synthesize_batch(ids, lens, sc); per.append((time.perf_counter() - t0) * 1e3)
            row.append("%d:%.2f" % (lim, np.median(per)))
        print(preset, "B=%d" % B, " ".join(row), flush=True)
    lib.vits_test_set_sx_small_max(default_limit)
    s.close()
