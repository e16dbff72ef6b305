#!/usr/bin/env python3
"""Summarise the rocprofv3 passes of tools/profile_gpu.sh into the two files kept under profiles/:
  <prefix>_kernel_stats.csv   per kernel: calls, total / average duration, share of GPU time
  <prefix>_pmc.json           per kernel: counter sums, MFMA busy %, effective clock, HBM bytes per launch
Reads either output format of rocprofv3 (rocpd SQLite `*_results.db`, or `--output-format csv`).
HBM bytes follow MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KiB, and on gfx950 FETCH_SIZE
reports half of a wide streaming read, so reads = FETCH_SIZE * 1024 * 2."""
import argparse
import csv
import glob
import json
import os
import sqlite3
import sys
from collections import defaultdict


def find(root, pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def dispatches(passdir):
    """-> list of (kernel name, duration ns)"""
    out = []
    for f in find(passdir, "*_results.db"):
        db = sqlite3.connect(f)
        out += [(n, int(d)) for n, d in db.execute("select name, duration from kernels")]
    for f in find(passdir, "*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            out.append((r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return out


def counters(passdir):
    """-> list of (kernel name, dispatch id, counter name, value)"""
    out = []
    for f in find(passdir, "*_results.db"):
        db = sqlite3.connect(f)
        out += [(k, d, c, float(v)) for k, d, c, v in
                db.execute("select kernel_name, dispatch_id, counter_name, value from counters_collection")]
    for f in find(passdir, "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            out.append((r["Kernel_Name"], r["Dispatch_Id"], r["Counter_Name"], float(r["Counter_Value"])))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("outdir")
    ap.add_argument("prefix")
    ap.add_argument("--preset", default="")
    ap.add_argument("--length-scale", type=float, default=None, help="scales[1] of the profiled bench command")
    a = ap.parse_args()

    # ---- pass "stats": kernel trace -> per-kernel time
    dur = defaultdict(list)
    for k, d in dispatches(os.path.join(a.outdir, "stats")):
        dur[k].append(d)
    total = sum(sum(v) for v in dur.values()) or 1
    rows = sorted(((k, len(v), sum(v), sum(v) / len(v), min(v), max(v)) for k, v in dur.items()), key=lambda r: -r[2])
    with open(a.prefix + "_kernel_stats.csv", "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for k, n, tot, avg, mn, mx in rows:
            w.writerow([k, n, tot, f"{avg:.1f}", f"{100.0 * tot / total:.2f}", mn, mx])

    # ---- passes "pmc*": counters summed per kernel over all dispatches, one counter set per pass
    kern = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(int)
    kdur = defaultdict(float)
    for sub in sorted(d for d in os.listdir(a.outdir) if d.startswith("pmc") and os.path.isdir(os.path.join(a.outdir, d))):
        p = os.path.join(a.outdir, sub)
        seen = defaultdict(set)
        for k, did, cname, val in counters(p):
            kern[k][cname] += val
            seen[k].add(did)
        for k, s in seen.items():
            launches[k] = max(launches[k], len(s))
        if sub == "pmc1":
            for k, d in dispatches(p):
                kdur[k] += d
    commit = None
    try:
        info = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "phoonnx_amd",
                                           "_build_info.json")))
        commit = info["commit"] + ("+dirty" if info.get("dirty") else "")
    except Exception:
        pass
    source_sha = None
    try:  # (what identifies the code that ran: a hash over the sources the library is built from; bench.py config.source_sha)
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from phoonnx_amd import build as _b
        source_sha = _b.source_sha()
    except Exception:
        pass
    out = {"preset": a.preset, "commit": commit, "source_sha": source_sha, "length_scale": a.length_scale,
           "source": "rocprofv3 --kernel-trace --pmc <set>, one counter set per run (tools/profile_gpu.sh); sums over all "
                     "dispatches of the run unless a key says per_launch",
           "kernels": {}}
    for k, c in sorted(kern.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
        d = dict(c)
        n = launches.get(k, 0) or 1
        d["launches"] = n
        if c.get("GRBM_GUI_ACTIVE"):
            # matrix-pipe busy share: busy cycles summed over the 1024 SIMDs / (active cycles per XCD x 1024);
            # GRBM_GUI_ACTIVE is the sum over the 8 XCDs
            d["MfmaUtil_pct"] = 100.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
            if kdur.get(k):
                d["avg_duration_ms_under_pmc"] = kdur[k] / n / 1e6
                # (GRBM_GUI_ACTIVE also counts the dispatch's ramp-up and drain: for launches under ~100 us the ratio
                # says nothing about the shader clock - it read 3-10 "GHz" there - so it is only derived for long ones)
                if kdur[k] / n >= 100e3:
                    d["effective_clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / kdur[k]
        if "FETCH_SIZE" in c:
            d["hbm_read_bytes_per_launch"] = c["FETCH_SIZE"] * 1024.0 * 2.0 / n
        if "WRITE_SIZE" in c:
            d["hbm_write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024.0 / n
        out["kernels"][short(k) if short(k) not in out["kernels"] else k] = d
    json.dump(out, open(a.prefix + "_pmc.json", "w"), indent=1)
    for k, n, tot, avg, mn, mx in rows[:8]:
        print(f"{100.0 * tot / total:6.2f}%  {n:5d} x {avg / 1e3:9.1f} us  {short(k)[:100]}")


if __name__ == "__main__":
    sys.exit(main())
