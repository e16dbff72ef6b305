#!/usr/bin/env python3
"""Micro-benchmark of the MFMA conv engine on the generator / flow / encoder conv shapes of the
"high" and "medium" presets (kernel tuning aid; run on the GPU box).

  python tools/conv_bench.py [--batch 32] [--frames 670] [--iters 10] [--sweep]
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phoonnx_amd import _ffi  # noqa: E402

PEAK = 157.3


def bench(lib, B, Cin, Cout, T, K, dil, hint, iters, cfg=-1, ck=-1):
    out = (C.c_float * 4)()
    rc = lib.vits_bench_conv1d(0, B, Cin, Cout, T, K, dil, hint, iters, cfg, ck, out)
    if rc != 0:
        return None
    ms = out[0]
    tf = 2.0 * B * Cin * Cout * K * T / (ms * 1e-3) / 1e12
    return ms, tf, int(out[1]), int(out[2])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=672)
    ap.add_argument("--tokens", type=int, default=256)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--sweep", action="store_true", help="try every tile config / chunk depth that fits")
    ap.add_argument("--ablate", action="store_true", help="time with DMA / epilogue / prologue switched off")
    ap.add_argument("--prof", action="store_true", help="with --sx: per-step cycle breakdown of the 128x128 kernel")
    ap.add_argument("--sx", action="store_true", help="benchmark the split-operand engine instead (f16x3 arithmetic)")
    ap.add_argument("--bf16x6", action="store_true", help="with --sx: the six-product exact arithmetic")
    ap.add_argument("--all-shapes", action="store_true", help="with --sx: the token- and frame-domain shapes too")
    ap.add_argument("--shapes", action="store_true",
                    help="with --sx: A/B of the two MFMA shapes of the main loop (16x16x32 where it applies vs 32x32x16), "
                         "interleaved rounds in one process")
    a = ap.parse_args()
    lib = _ffi.load()
    F, B = a.frames, a.batch
    shapes = [
        # name, Cin, Cout, T, K, dil, hint
        ("dec s1 256 k3", 256, 256, F * 8, 3, 1, 0), ("dec s1 256 k7 d3", 256, 256, F * 8, 7, 3, 0),
        ("dec s1 256 k11 d5", 256, 256, F * 8, 11, 5, 0),
        ("dec s2 128 k3", 128, 128, F * 64, 3, 1, 0), ("dec s2 128 k7 d3", 128, 128, F * 64, 7, 3, 0),
        ("dec s2 128 k11 d5", 128, 128, F * 64, 11, 5, 0),
        ("dec s3 64 k3", 64, 64, F * 128, 3, 1, 0), ("dec s3 64 k7", 64, 64, F * 128, 7, 1, 0),
        ("dec s3 64 k11 d5", 64, 64, F * 128, 11, 5, 0),
        ("dec s4 32 k3", 32, 32, F * 256, 3, 1, 0), ("dec s4 32 k7 d3", 32, 32, F * 256, 7, 3, 0),
        ("dec s4 32 k11 d5", 32, 32, F * 256, 11, 5, 0),
        ("med s2 64 k7 d12", 64, 64, F * 64, 7, 12, 0), ("med s3 32 k5 d6", 32, 32, F * 256, 5, 6, 0),
        ("conv_pre 192->512 k7", 192, 512, F, 7, 1, 0),
        ("flow in 192->384 k5", 192, 384, F, 5, 1, 1), ("flow rs 192->384 k1", 192, 384, F, 1, 1, 1),
        ("flow pre 96->192", 96, 192, F, 1, 1, 1),
        ("enc ffn1 192->768 k3", 192, 768, a.tokens, 3, 1, 2), ("enc ffn2 768->192 k3", 768, 192, a.tokens, 3, 1, 2),
        ("enc qkv 192->576", 192, 576, a.tokens, 1, 1, 2), ("enc 1x1 192->192", 192, 192, a.tokens, 1, 1, 2),
    ]
    if a.sx:  # split-exact bf16 engine: generator shapes only (Cin % 16 == 0, Cout % 32 == 0)
        from phoonnx_amd.session import bench_conv1d_sx
        mode = 0 if a.bf16x6 else 128
        if a.shapes:
            for name, Cin, Cout, T, K, dil, hint in shapes:
                if hint == 2 or Cin % 32 or Cout % 32 or Cin <= 64:
                    continue
                res = {0: [], 256: []}
                for rnd in range(3):
                    for sh in (0, 256):
                        for dbg in (0, 8):
                            ms, cfg = bench_conv1d_sx(B, Cin, Cout, T, K, dil, dbg | mode | sh, a.iters)
                            res[sh].append((dbg, ms))
                def best(sh, dbg):
                    return min(ms for d, ms in res[sh] if d == dbg)
                fl = 2.0 * B * Cin * Cout * K * T / 1e9
                print(f"{name:24s} T={T:7d} cfg{cfg}  planes: 16x16x32 {best(0, 0):7.3f} ms {fl / best(0, 0):5.0f} TF | 32x32x16 "
                      f"{best(256, 0):7.3f} ms {fl / best(256, 0):5.0f} TF | x{best(256, 0) / best(0, 0):.3f}    res+raw+planes: "
                      f"{best(0, 8):7.3f} vs {best(256, 8):7.3f} ms x{best(256, 8) / best(0, 8):.3f}", flush=True)
            return
        for name, Cin, Cout, T, K, dil, hint in shapes:
            if hint != 0 and not a.all_shapes:
                continue
            line = f"{name:24s} T={T:7d}"
            for tag, dbg in (("planes", 0), ("res+raw+planes", 8), ("noDMA", 1), ("noEPI", 2), ("none", 3)):
                ms, cfg = bench_conv1d_sx(B, Cin, Cout, T, K, dil, dbg | mode, a.iters)
                tf = 2.0 * B * Cin * Cout * K * T / (ms * 1e-3) / 1e12
                if dbg == 0:
                    line += f" sx{cfg} {ms:8.3f} ms"
                line += f"  {tag}:{tf:5.0f}"
                if not a.ablate and dbg == 8:
                    break
            print(line + "  TF/s fp32-equivalent", flush=True)
            if a.prof and Cout % 128 == 0:
                for dbg in (16, 16 | 3):
                    ms, cfg, pc = bench_conv1d_sx(B, Cin, Cout, T, K, dil, dbg | mode, a.iters)
                    print(f"      prof dbg={dbg & 15}: s_memtime ticks/step: lgkm {pc[0]:.0f} vmwait {pc[1]:.0f} barrier {pc[2]:.0f} "
                          f"dma-issue {pc[3]:.0f} loads+mfma {pc[4]:.0f}  (sum {sum(pc[:5]):.0f})  clock {pc[5]:.2f} GHz", flush=True)
        return
    for name, Cin, Cout, T, K, dil, hint in shapes:
        r = bench(lib, B, Cin, Cout, T, K, dil, hint, a.iters)
        if r is None:
            print(f"{name:24s} FAILED: {_ffi.last_error(None)}")
            continue
        ms, tf, cfg, ck = r
        line = f"{name:24s} T={T:7d} cfg{cfg} CK{ck:2d}  {ms:8.3f} ms  {tf:6.1f} TF/s ({100 * tf / PEAK:4.1f}%)"
        if a.ablate:
            parts = []
            for tag, dbg in (("noACT", 4), ("noACT+res", 12), ("noACT+noDMA", 5), ("noACT+noEPI", 6), ("none", 7)):
                rr = bench(lib, B, Cin, Cout, T, K, dil, hint | (dbg << 8), a.iters)
                if rr:
                    parts.append(f"{tag}:{rr[1]:.0f}")
            line += "  | ablate TF/s: " + " ".join(parts)
        if a.sweep:
            best = []
            for c in range(6):
                for k in (8, 16, 32, 64, 96, 192):
                    rr = bench(lib, B, Cin, Cout, T, K, dil, hint, a.iters, c, k)
                    if rr:
                        best.append((rr[1], c, k))
            best.sort(reverse=True)
            line += "  | sweep: " + " ".join(f"cfg{c}/CK{k}:{t:.0f}" for t, c, k in best[:5])
        print(line, flush=True)


if __name__ == "__main__":
    main()
