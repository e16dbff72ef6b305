#!/usr/bin/env python3
"""Static check of the conv engines' ISA for the scalar-load hazard: every kernel of the split-operand
engine feeds its MFMA loop through inline-asm ds_reads behind COUNTED `s_waitcnt lgkmcnt(n)`.  hipcc sinks kernel-argument
loads (s_load_dword*) that only the epilogue needs into the block between the unrolled main loop and its tail steps; a
scalar load in flight counts in lgkmcnt and returns out of order, so a counted wait could pass with a ds_read outstanding.
The engines drain the counter once at that boundary; this script compiles each instantiation unit to assembly and verifies
that EVERY s_load between a kernel's first and last MFMA is followed by `s_waitcnt lgkmcnt(0)` before any counted wait.

Second check (the build's own rule, phoonnx_amd.build.spill_hazards): no spill store of any kernel may hit a register that
is the destination of an asynchronous inline-asm load not yet consumed (hazard 1).  Third (sgpr_vmem_hazards): no inline-asm
memory instruction may read a scalar base that a v_readlane / v_readfirstlane wrote fewer than five wait states before
(hazard 5: the compiler's hazard recogniser does not look into inline asm).

    python tools/check_isa_hazards.py        (needs hipcc; ~1 min per unit; exit code 1 on a finding)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phoonnx_amd.build import sgpr_vmem_hazards, spill_hazards  # noqa: E402
CSRC = os.path.join(ROOT, "phoonnx_amd", "csrc")
UNITS = ["tu_sx_h1", "tu_sx_s16p", "tu_sx_s16", "tu_sx_s32", "tu_sx_bf16", "tu_pair16", "tu_pair", "tu_conv_f32", "vitsmi", "g2p"]
# kernels with counted lgkmcnt waits: the conv engines, and the 16x16x32 attention kernel (attention16.hip.hpp, in vitsmi.hip)
KERNELS = r"_ZN6vitsmi[0-9]+(?:conv_sx|attention_relpos16)\w*"


def check(unit):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, unit + ".s")
        r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-result", "-x", "hip",
                            "--cuda-device-only", "-S", os.path.join(CSRC, unit + ".hip"), "-o", out], capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stderr)
            raise SystemExit(f"hipcc failed on {unit}")
        txt = open(out).read()
    bad, n = [], 0
    for name in re.findall(r"\n(" + KERNELS + r"):", txt):
        i = txt.index("\n" + name + ":")
        lines = txt[i:txt.index(".Lfunc_end", i)].split("\n")  # (a kernel with an early exit has several s_endpgm)
        n += 1
        mf = [k for k, l in enumerate(lines) if "v_mfma" in l]
        if not mf:
            continue
        for k, l in enumerate(lines):
            if "s_load_dword" in l and mf[0] < k < mf[-1]:
                nxt = next((m for m in lines[k + 1:] if "lgkmcnt" in m), "")
                if "lgkmcnt(0)" not in nxt:
                    bad.append((name, k, "s_load followed by '" + nxt.strip() + "'"))
                    break
    hz, nspill = spill_hazards(txt)
    for name, f in hz.items():
        bad.append((name, f[0][0], "spill of an asynchronous load's destination: " + f[0][1]))
    for name, f in sgpr_vmem_hazards(txt).items():
        bad.append((name, f[0][0], f"'{f[0][3]}' only {f[0][1]} wait states after '{f[0][2]}' (hazard 5)"))
    return n, bad, nspill


def main():
    total = 0
    for u in UNITS:
        n, bad, nspill = check(u)
        print(f"{u}: {n} conv / attention kernels, {nspill} kernels that spill, unsafe: {len(bad)}")
        for name, k, what in bad:
            print(f"   {name}: line {k}: {what}")
        total += len(bad)
    sys.exit(1 if total else 0)


if __name__ == "__main__":
    main()
