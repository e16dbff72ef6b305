#!/usr/bin/env python3
"""A graph written by the REFERENCE's exporter at full size, rendered by the GPU path and compared with the reference's own
outputs.  The .onnx / .npz pair comes from `oracle/gen_golden.py --big <dir>` in the build container
(the reference's SynthesizerTrn, its export call, its outputs with injected noise); it is too large to commit (64-114 MB) and
travels to the GPU box as data.

    python tools/check_exported_graph.py <dir> [preset ...] [--precision f16x3|bf16x6|f16]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("presets", nargs="*", default=["medium"])
    ap.add_argument("--precision", default="f16x3")
    a = ap.parse_args()
    from phoonnx_amd import MiSession
    bad = 0
    for preset in a.presets:
        path = os.path.join(a.dir, preset + ".onnx")
        g = np.load(os.path.join(a.dir, preset + ".npz"))
        s = MiSession(path, gen_precision=a.precision)
        print(f"{preset}: {os.path.getsize(path) / 1e6:.1f} MB, gen_sx {s.hparam('gen_sx')}, gen_nprod {s.hparam('gen_nprod')}, "
              f"hidden {s.hparam('hidden')}, n_layers {s.hparam('n_layers')}, resblock {s.hparam('resblock')}, hop {s.hparam('hop')}")
        cases = sorted({k.split("/")[0] for k in g.files})
        for c in cases:
            get = lambda k: g[f"{c}/{k}"] if f"{c}/{k}" in g.files else None
            r = s.synthesize_batch(get("ids"), get("lens"), get("scales"), get("sid"), get("noise_dp"), get("noise_z"),
                                   taps=("x", "m_p", "logs_p", "logw", "w_ceil", "z_p", "z"))
            ok_int = bool(np.array_equal(r["w_ceil"], get("out_w_ceil")) and np.array_equal(r["y_lengths"], get("out_y_lengths")))
            errs = {k: float(np.abs(r[k] - get("out_" + k)).max()) for k in ("x", "m_p", "logs_p", "logw", "z_p", "z")}
            ref = get("out_output")
            hop = s.hparam("hop")
            werr, snr = 0.0, 1e9
            for b in range(ref.shape[0]):
                n = int(get("out_y_lengths")[b]) * hop
                d = (r["output"][b, 0, 0, :n] - ref[b, 0, 0, :n]).astype(np.float64)
                werr = max(werr, float(np.abs(d).max()))
                snr = min(snr, 10 * np.log10((ref[b, 0, 0, :n].astype(np.float64) ** 2).sum() / max((d ** 2).sum(), 1e-30)))
            tol = 1e-2 if a.precision == "f16" else 1e-3
            good = ok_int and werr < tol and all(v < 5e-4 for v in errs.values())
            bad += 0 if good else 1
            print(f"  {c}: B={ref.shape[0]} frames={get('out_y_lengths').tolist()} durations/frame counts exact: {ok_int}; "
                  f"stage taps max-abs {max(errs.values()):.2e}; waveform max-abs {werr:.2e}, SNR {snr:.1f} dB  {'OK' if good else 'FAIL'}")
        s.close()
    print("RESULT", "all within tolerance" if not bad else f"{bad} case(s) out of tolerance")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
