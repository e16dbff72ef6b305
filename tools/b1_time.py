#!/usr/bin/env python3
"""Wall time of one-utterance calls (host ids in, host waveform out: the reference's call shape), median of --calls, both voices:
    python tools/b1_time.py [--calls 300]            (VITSMI_LIB picks the library: A/B of two builds in separate processes)"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=300)
    ap.add_argument("--presets", default="medium,high")
    a = ap.parse_args()
    import torch
    from phoonnx_amd import MiSession
    from phoonnx_amd.synth import write_voice
    cache = os.environ.get("VITSMI_BENCH_CACHE", "/tmp/vitsmi_bench")
    os.makedirs(cache, exist_ok=True)
    out = {}
    for preset in a.presets.split(","):
        path = os.path.join(cache, f"synth_{preset}.onnx")
        if not os.path.exists(path):
            write_voice(path, preset, seed=1234)
        s = MiSession(path)
        g = torch.Generator().manual_seed(4321)
        ids = torch.randint(0, 256, (1, 256), generator=g, dtype=torch.int64).numpy()
        lens = np.full((1,), 256, np.int64)
        scales = np.array([0.667, 1.95, 0.8], np.float32)
        for _ in range(30):
            s.synthesize_batch(ids, lens, scales)
        ts = []
        for _ in range(a.calls):
            t0 = time.perf_counter()
            s.synthesize_batch(ids, lens, scales)
            ts.append(time.perf_counter() - t0)
        s.close()
        out[preset] = round(1e3 * float(np.median(ts)), 4)
    print(out)


if __name__ == "__main__":
    main()
