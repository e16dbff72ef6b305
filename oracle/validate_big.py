#!/usr/bin/env python3
"""In-container only: check the C oracle against the reference PyTorch module on the
full-size presets (medium / high / medium_ms).  Needs `gen_golden.py --big DIR` first.
Prints max-abs error per stage; recorded in DESIGN.md §Oracle."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from vits_oracle import VitsOracle  # noqa: E402

STAGES = ("x", "m_p", "logs_p", "logw", "w_ceil", "y_lengths", "z_p", "z", "output")


def check(d, name):
    o = VitsOracle(os.path.join(d, name + ".onnx"))
    g = np.load(os.path.join(d, name + ".npz"))
    worst = 0.0
    for c in sorted(set(k.split("/")[0] for k in g.files)):
        get = lambda k: g[f"{c}/{k}"] if f"{c}/{k}" in g.files else None
        r = o.infer(get("ids"), get("lens"), get("scales"), get("sid"), get("noise_dp"), get("noise_z"))
        errs = {k: float(np.abs(r[k].astype(np.float64) - get("out_" + k)).max()) for k in STAGES}
        assert errs["w_ceil"] == 0 and errs["y_lengths"] == 0, errs
        worst = max(worst, errs["output"])
        print(name, c, " ".join(f"{k}:{v:.2e}" for k, v in errs.items()))
    return worst


if __name__ == "__main__":
    d = sys.argv[1] if len(sys.argv) > 1 else "/tmp/vits_big"
    for n in ("medium", "high", "medium_ms"):
        w = check(d, n)
        assert w < 1e-4, w
    print("OK")
