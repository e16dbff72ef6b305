#!/usr/bin/env python3
"""Per-launch table of one voice's conv-engine launches on the GPU box: kernel instantiation, conv shape, HIP-event
time, algorithmic TFLOP/s and layer-granular TB/s of every launch of one step (vits_launch_records), grouped by
pipeline stage.  One handle, whole batch (events serialise the stream).

    python tools/launch_table.py --preset medium [--batch 32] [--json gpurun_out/x.json]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="medium")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--tokens", type=int, default=256)
    ap.add_argument("--length-scale", type=float, default=1.95)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    import torch
    from phoonnx_amd import MiSession
    from phoonnx_amd.synth import write_voice
    cache = os.environ.get("VITSMI_BENCH_CACHE", "/tmp/vitsmi_bench")
    os.makedirs(cache, exist_ok=True)
    path = os.path.join(cache, f"synth_{a.preset}.onnx")
    if not os.path.exists(path):
        write_voice(path, a.preset, seed=1234)
    s = MiSession(path)
    g = torch.Generator().manual_seed(1234)
    ids = torch.randint(0, 256, (a.batch, a.tokens), generator=g, dtype=torch.int64).cuda()
    lens = torch.full((a.batch,), a.tokens, dtype=torch.int64).cuda()
    scales = np.array([0.667, a.length_scale, 0.8], np.float32)
    s.set_seed(1234)
    s.set_timing(True)
    for _ in range(2):
        s.run_device(ids.data_ptr(), lens.data_ptr(), a.batch, a.tokens, scales)
        s.stats()
    acc = None
    st_acc = {}
    for _ in range(a.reps):
        s.run_device(ids.data_ptr(), lens.data_ptr(), a.batch, a.tokens, scales)
        st = s.stats()
        recs = s.launch_records()
        if acc is None:
            acc = recs
        else:
            for r0, r in zip(acc, recs):
                r0["ms"] += r["ms"]
        for k in ("enc_ms", "dp_ms", "flow_ms", "dec_ms", "total_ms", "conv_ms"):
            st_acc[k] = st_acc.get(k, 0.0) + st[k]
    for r in acc:
        r["ms"] /= a.reps
    st_acc = {k: v / a.reps for k, v in st_acc.items()}
    names = ["enc", "dp", "flow", "dec"]
    print(f"# {a.preset} B={a.batch} T={a.tokens}: stages {json.dumps({k: round(v, 3) for k, v in st_acc.items()})}")
    print(f"{'#':>3} {'stage':5} {'kernel':58} {'Cin':>4} {'Cout':>5} {'K':>2} {'d':>2} {'T':>7} {'us':>8} {'TFLOP/s':>8} {'TB/s':>6}")
    tot = {}
    for i, r in enumerate(acc):
        sec = r["ms"] * 1e-3
        tf = r["flops"] / sec / 1e12 if sec > 0 else 0
        tb = r["bytes"] / sec / 1e12 if sec > 0 else 0
        print(f"{i:3d} {names[r['stage']]:5} {r['kernel'].replace(' ', ''):58} {r['cin']:4d} {r['cout']:5d} {r['k']:2d} {r['dil']:2d} "
              f"{r['t']:7d} {r['ms'] * 1e3:8.1f} {tf:8.1f} {tb:6.2f}")
        tot[names[r["stage"]]] = tot.get(names[r["stage"]], 0.0) + r["ms"]
    print("# conv-engine ms per stage:", json.dumps({k: round(v, 3) for k, v in tot.items()}))
    if a.json:
        json.dump({"preset": a.preset, "batch": a.batch, "stages": st_acc, "launches": acc}, open(a.json, "w"), indent=1)
    s.close()


if __name__ == "__main__":
    main()
