#!/usr/bin/env python3
"""One call's kernel launches in order, from a rocprofv3 kernel trace (CSV): name, grid, duration, gap to the predecessor.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python3 tools/trace_sequence.py --run medium   (one utterance per call)
    python3 tools/trace_sequence.py /tmp/tr --first embed_kernel        # the launches from the last `embed_kernel` on
"""
import argparse
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(preset, batch, calls):
    import numpy as np
    import torch
    from phoonnx_amd import MiSession
    from phoonnx_amd.synth import write_voice
    cache = os.environ.get("VITSMI_BENCH_CACHE", "/tmp/vitsmi_bench")
    os.makedirs(cache, exist_ok=True)
    path = os.path.join(cache, f"synth_{preset}.onnx")
    if not os.path.exists(path):
        write_voice(path, preset, seed=1234)
    s = MiSession(path)
    g = torch.Generator().manual_seed(4321)
    ids = torch.randint(0, 256, (batch, 256), generator=g, dtype=torch.int64).numpy()
    lens = np.full((batch,), 256, np.int64)
    scales = np.array([0.667, 1.95, 0.8], np.float32)
    for _ in range(calls):
        s.synthesize_batch(ids, lens, scales)
    s.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir", nargs="?")
    ap.add_argument("--run", default=None, help="preset: run `--calls` host-in / host-out calls of one batch (to be traced)")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--calls", type=int, default=20)
    ap.add_argument("--first", default="embed_kernel", help="a call starts at the last launch whose name contains this")
    a = ap.parse_args()
    if a.run:
        return run(a.run, a.batch, a.calls)
    rows = []
    for f in glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    name = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("vitsmi::", "").split("(")[0]
    starts = [i for i, r in enumerate(rows) if a.first in name(r)]
    lo = starts[-1]
    t0 = int(rows[lo]["Start_Timestamp"])
    tot = 0.0
    print(f"{len(rows) - lo} launches from the last '{a.first}':")
    for i in range(lo, len(rows)):
        r = rows[i]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        gap = (int(r["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3 if i > lo else 0.0
        tot += d
        print(f"{i - lo:4d} {name(r)[:64]:64s} grid {r.get('Grid_Size_X', '?'):>8s}  {d:8.2f} us  gap {gap:6.2f}  at {(int(r['Start_Timestamp']) - t0) / 1e3:8.1f}")
    print(f"kernel time {tot:.1f} us, span {(int(rows[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
