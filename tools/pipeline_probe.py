#!/usr/bin/env python3
"""Does splitting a batch over several engine handles (= HIP streams) pay?  The token / frame-domain stages of
one sub-batch use only part of the chip (small grids) and can overlap with another sub-batch's generator.
    python tools/pipeline_probe.py [--preset high] [--batch 32] [--parts 1 2 4] [--steps 10]
Prints samples/s for each number of parts (run on the GPU box)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="high")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--tokens", type=int, default=256)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--parts", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--threads", action="store_true", help="one host thread per part instead of one for all")
    a = ap.parse_args()
    import torch
    from phoonnx_amd import MiSession
    from phoonnx_amd.synth import write_voice
    voice = f"/tmp/vitsmi_bench/synth_{a.preset}.onnx"
    if not os.path.exists(voice):
        os.makedirs(os.path.dirname(voice), exist_ok=True)
        write_voice(voice, a.preset, seed=1234)
    B, T = a.batch, a.tokens
    scales = np.array([0.667, 1.5, 0.8], np.float32)
    g = torch.Generator(device="cpu").manual_seed(1234)
    ids = torch.randint(0, 256, (B, T), generator=g, dtype=torch.int64).cuda()
    lens = torch.full((B,), T, dtype=torch.int64).cuda()
    for parts in a.parts:
        sess = [MiSession(voice) for _ in range(parts)]
        for i, s in enumerate(sess):
            s.set_seed(1234 + i)
        hop = sess[0].hparam("hop")
        bounds = [B * i // parts for i in range(parts + 1)]

        def one(i):
            s = sess[i]
            b0, b1 = bounds[i], bounds[i + 1]
            s.run_device(ids[b0:b1].data_ptr(), lens[b0:b1].data_ptr(), b1 - b0, T, scales)
            return int(s.last_y_lengths().sum()) * hop

        if a.threads and parts > 1:
            from concurrent.futures import ThreadPoolExecutor
            pool = ThreadPoolExecutor(parts)

            def step():
                return sum(pool.map(one, range(parts)))
        else:
            def step():
                return sum(one(i) for i in range(parts))

        for _ in range(3):
            step()
        for s in sess:
            s.sync()
        t0 = time.perf_counter()
        samples = 0
        for _ in range(a.steps):
            samples += step()
        for s in sess:
            s.sync()
        dt = time.perf_counter() - t0
        print(f"{a.preset} B={B} parts={parts}: {samples / dt / 1e6:8.2f} M samples/s  {dt / a.steps * 1e3:7.2f} ms/step", flush=True)
        for s in sess:
            s.close()


if __name__ == "__main__":
    main()
