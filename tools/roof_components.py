#!/usr/bin/env python3
"""Where a step's time is, as an HBM component and a matrix component per kernel instantiation:

    t_hbm  = measured HBM bytes (PMC: 2 x FETCH_SIZE + WRITE_SIZE per launch, profiles/*_pmc.json) / 5 TB/s
    t_mfma = 3 x algorithmic fp32 FLOPs (three fp16 products per fp32 product) / 1 270 TFLOP/s (the chip's f16 matrix rate on
             random data, MI355X_MICROARCH.md 'DVFS give-back' item 1)

next to the measured time of the same launches (HIP events, one handle: tools/launch_table.py --json).  A launch at max(t_hbm,
t_mfma) overlaps the two perfectly, one at their sum not at all.

    python tools/roof_components.py <launch_table.json> <pmc.json> [--hbm 5e12] [--mfma 1270e12]
"""
import argparse
import json


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("launch_table")
    ap.add_argument("pmc")
    ap.add_argument("--hbm", type=float, default=5e12)
    ap.add_argument("--mfma", type=float, default=1270e12)
    a = ap.parse_args()
    lt = json.load(open(a.launch_table))
    pmc = json.load(open(a.pmc))["kernels"]
    key = lambda n: n.replace("vitsmi::", "").replace("void ", "").replace(" ", "").split("(")[0]
    by = {key(k): v for k, v in pmc.items()}
    agg = {}
    for r in lt["launches"]:
        k = key(r["kernel"])
        g = agg.setdefault(k, {"n": 0, "ms": 0.0, "flops": 0.0})
        g["n"] += 1
        g["ms"] += r["ms"]
        g["flops"] += r["flops"]
    print(f"# {lt['preset']} B={lt['batch']}: per step, one handle; t_hbm at {a.hbm / 1e12:g} TB/s of PMC bytes, t_mfma at "
          f"{a.mfma / 1e12:g} TFLOP/s of 3 x fp32 FLOPs")
    print(f"{'kernel':52} {'n':>3} {'measured':>9} {'t_hbm':>8} {'t_mfma':>8} {'sum':>8} {'max':>8}   us")
    tot = [0.0, 0.0, 0.0]
    for k, g in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
        c = by.get(k)
        if not c or "hbm_read_bytes_per_launch" not in c:
            print(f"{k[:52]:52} {g['n']:3d} {g['ms'] * 1e3:9.0f}   (no counters)")
            tot[0] += g["ms"] * 1e3
            continue
        th = (c["hbm_read_bytes_per_launch"] + c["hbm_write_bytes_per_launch"]) * g["n"] / a.hbm * 1e6
        tm = 3.0 * g["flops"] / a.mfma * 1e6
        print(f"{k[:52]:52} {g['n']:3d} {g['ms'] * 1e3:9.0f} {th:8.0f} {tm:8.0f} {th + tm:8.0f} {max(th, tm):8.0f}")
        tot[0] += g["ms"] * 1e3
        tot[1] += th
        tot[2] += tm
    print(f"{'all conv launches':52} {'':3} {tot[0]:9.0f} {tot[1]:8.0f} {tot[2]:8.0f} {tot[1] + tot[2]:8.0f} {max(tot[1], tot[2]):8.0f}")


if __name__ == "__main__":
    main()
