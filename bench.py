#!/usr/bin/env python3
"""bench.py — audio samples/s of the MI355X-native VITS path on synthetic fixed-length
phoneme batches (BASELINE.json: batch-32, 256-phoneme utterances, 22.05 kHz).

  python bench.py --gpus N --steps K --warmup W [--preset high|medium] [--batch 32] [--tokens 256]

N > 1 is launched by torch.distributed.run, one rank per GPU: rank 0 reads and packs the
.onnx, the packed weight arena is broadcast over RCCL/xGMI, then every rank synthesises its
own 32 utterances with no further communication ("scaling": "weak").
A step = one pass of the whole path (encoder + duration predictor + flow + vocoder) over one
batch whose inputs are already resident in HBM.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 MFMA (v_mfma_f32_32x32x2_f32) = fp32 vector peak
# split-exact engine: dense bf16 MFMA peak (256 CUs x 4096 FLOP/clk x 2.4 GHz = 2516.6 TFLOP/s) / 6 plane products
SX_PEAK_TFLOPS = 2516.6 / 6
# ... the same pipe (f16 MFMA has the bf16 rate) / 3 products of the default two-fp16-plane arithmetic
SX_F16_PEAK_TFLOPS = 2516.6 / 3


def pmc_traffic(preset, kernel_prefix):
    """HBM bytes per launch of the dominant kernel family from the committed PMC passes of the SAME command
    (profiles/*_<preset>_b32_pmc.json, written by tools/profile_gpu.sh: FETCH_SIZE x 2 + WRITE_SIZE, separate
    passes, per MI355X_MICROARCH.md).  Launch-weighted mean over the family's instantiations; None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{preset}_b32_pmc.json")))
    if not files:
        return None
    try:
        ks = json.load(open(files[-1]))["kernels"]
    except Exception:
        return None
    tot = n = 0.0
    for name, c in ks.items():
        if kernel_prefix in name and "hbm_read_bytes_per_launch" in c and "hbm_write_bytes_per_launch" in c:
            tot += (c["hbm_read_bytes_per_launch"] + c["hbm_write_bytes_per_launch"]) * c["launches"]
            n += c["launches"]
    return {"bytes_per_launch": tot / n, "source": os.path.basename(files[-1])} if n else None
LENGTH_SCALE = {"high": 1.5, "medium": 1.5, "small": 1.5}  # gives ~3 frames per phoneme id with synth weights


def cpu_baseline(voice_path, preset, tokens, scales, seed):
    """Oracle (C restatement, OpenMP, all host cores) timed on a bounded sample of the workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import vits_oracle
    try:
        o = vits_oracle.VitsOracle(voice_path, native=True)
    except Exception:
        o = vits_oracle.VitsOracle(voice_path, native=False)
    Bs = 2 if preset == "high" else 4
    rng = np.random.default_rng(seed)
    ids = rng.integers(0, 256, size=(Bs, tokens)).astype(np.int64)
    lens = np.full((Bs,), tokens, np.int64)
    ndp = rng.standard_normal((Bs, 2, tokens)).astype(np.float32)
    nz = rng.standard_normal((Bs, o.inter_channels, tokens * 12)).astype(np.float32)
    t0 = time.perf_counter()
    r = o.infer(ids, lens, scales, None, ndp, nz)
    dt = time.perf_counter() - t0
    hop = r["output"].shape[3] // int(r["y_lengths"].max())
    samples = int(r["y_lengths"].sum()) * hop
    return {"value": samples / dt, "unit": "samples/s", "cores": int(o.lib.vo_num_threads()), "kind": "port",
            "sample": f"C/OpenMP restatement (oracle/vits_oracle.c), B={Bs} x {tokens} ids, {samples} samples in {dt:.1f}s",
            "rtf": dt / (samples / 22050.0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--preset", default="high", choices=["high", "medium", "small"])
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--tokens", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--gen-precision", default="f16x3", choices=["f16x3", "bf16x6", "bf16x3", "bf16"],
                    help="arithmetic of the generator's convs (fp32 operands and results in every mode).  f16x3 (default): "
                         "two fp16 planes per operand, three MFMA products per fp32 product, error no larger than the "
                         "f32-MFMA engine's; bf16x6: three bf16 planes, six products, every product exact; bf16x3 / "
                         "bf16: the declared reduced-precision vocoder modes of BASELINE config 4 (reported with their "
                         "dtype, never as the headline number)")
    ap.add_argument("--no-exact-check", action="store_true",
                    help="skip the second, shorter measurement of the same workload with the six-product exact arithmetic "
                         "(bf16x6), which the N=1 line carries next to the headline value")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) and broadcast the weight arena even at world size 1 "
                         "(exercises the N > 1 code path on a one-GPU box; needs the torchrun environment)")
    ap.add_argument("--lockstep", action="store_true",
                    help="enqueue the parts of every step from one host thread and join them per step, instead of one "
                         "free-running host thread per part (two serving workers)")
    ap.add_argument("--parts", type=int, default=2,
                    help="render each batch as this many sub-batches on as many engine handles / HIP streams sharing "
                         "one weight arena (PipelinedSession); 1 = a single handle")
    a = ap.parse_args()
    os.environ["VITSMI_GEN_PRECISION"] = a.gen_precision

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or a.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from phoonnx_amd import PipelinedSession
    from phoonnx_amd.sharding import open_sharded
    from phoonnx_amd.synth import write_voice

    cache = os.environ.get("VITSMI_BENCH_CACHE", "/tmp/vitsmi_bench")
    voice = os.path.join(cache, f"synth_{a.preset}.onnx")
    if rank == 0 and not os.path.exists(voice):
        os.makedirs(cache, exist_ok=True)
        write_voice(voice + ".tmp", a.preset, seed=1234)
        os.replace(voice + ".tmp", voice)
    if dist:
        dist.barrier()

    # weights: rank 0 reads + packs, RCCL broadcast of the arena, every rank opens on its GPU
    sess, arena_keepalive = open_sharded(voice, local_rank, dist)
    hop = sess.hparam("hop")
    sess_gen_sx = bool(sess.hparam("gen_sx"))
    gen_nprod = int(sess.hparam("gen_nprod"))
    sess.set_seed(1234 + rank * 16)
    pipe = PipelinedSession(sess, max(1, a.parts))  # extra handles borrow sess's weight arena
    pipe.set_seed(1234 + rank * 16)

    B, T = a.batch, a.tokens
    scales = np.array([0.667, LENGTH_SCALE[a.preset], 0.8], np.float32)
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    ids = torch.randint(0, 256, (B, T), generator=g, dtype=torch.int64).cuda()
    lens = torch.full((B,), T, dtype=torch.int64).cuda()
    torch.cuda.synchronize()

    def step():
        pipe.run_device(ids.data_ptr(), lens.data_ptr(), B, T, scales)
        return int(pipe.last_y_lengths(B).sum()) * hop

    def step_one():  # the whole batch on the first handle: per-kernel timing for the roofline block
        sess.run_device(ids.data_ptr(), lens.data_ptr(), B, T, scales)

    # K steps = K passes of the whole path over the batch.  Default: every part (half batch, own handle and stream) is
    # driven by its own host thread through its K passes, like two serving workers; --lockstep joins them per step.
    def run_steps(k):
        if a.lockstep or len(pipe.parts) == 1:
            return sum(step() for _ in range(k))
        return int(pipe.run_device_steps(ids.data_ptr(), lens.data_ptr(), B, T, scales, k).sum()) * hop

    if a.warmup > 0:
        run_steps(a.warmup)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    samples = run_steps(a.steps)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    tot = torch.tensor([dt, float(samples)], dtype=torch.float64, device="cuda")
    if dist:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tot.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt_max, samples_all = float(mx[0]), float(sm[1])
    else:
        dt_max, samples_all = dt, float(samples)

    roofline = None
    stage = None
    if rank == 0 and not a.no_roofline:
        # per-kernel timing with HIP events on the engine's own stream (vits_set_timing)
        sess.set_timing(True)
        fl = ms = by = 0.0
        launches = 0
        agg = {}
        n_t = max(3, min(a.steps, 5))
        step_one()  # (untimed: the first timed run creates the handle's HIP events)
        sess.stats()
        for _ in range(n_t):
            step_one()
            st = sess.stats()
            fl += st["conv_flops"]
            by += st["conv_bytes"]
            ms += st["conv_ms"]
            launches += st["conv_launches"]
            for k in ("enc_ms", "dp_ms", "flow_ms", "dec_ms", "total_ms", "dec_flops", "dec_bytes", "flow_flops",
                      "sx_flops", "sx_ms", "sx_launches"):
                agg[k] = agg.get(k, 0.0) + st[k]
        sess.set_timing(False)
        if agg.get("sx_launches", 0) > 0:
            # dominant kernel: the generator's split-exact conv (six bf16 MFMA plane products per fp32 product)
            kfl, kms, kn = agg["sx_flops"], agg["sx_ms"], int(agg["sx_launches"])
            if gen_nprod == 2:
                kname = "conv_sx_kernel (implicit-GEMM Conv1d, fp32 operands as 2 fp16 planes, 3 x v_mfma_f32_32x32x16_f16 per product)"
                peak = SX_F16_PEAK_TFLOPS
            else:
                kname = "conv_sx_kernel (implicit-GEMM Conv1d, fp32-exact via 3 bf16 planes, v_mfma_f32_32x32x16_bf16)"
                peak = SX_PEAK_TFLOPS if gen_nprod == 6 else 2516.6 / gen_nprod  # (reduced modes: fewer plane products)
        else:
            kfl, kms, kn = fl, ms, launches
            kname = "conv_engine_kernel (implicit-GEMM Conv1d, v_mfma_f32_32x32x2_f32)"
            peak = FP32_PEAK_TFLOPS
        ach = kfl / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
        roofline = {"bound": "mfma", "kernel": kname,
                    "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                    "traffic": (pmc_traffic(a.preset, kname.split(" ")[0]) or {}).get("bytes_per_launch")
                    if (B, T) == (32, 256) else None,
                    "traffic_source": (pmc_traffic(a.preset, kname.split(" ")[0]) or {}).get("source"),
                    "launches_per_step": kn // n_t,
                    "avg_launch_ms": kms / max(kn, 1),
                    "algorithmic_gflop_per_launch": kfl / max(kn, 1) / 1e9,
                    "all_conv_launches_per_step": launches // n_t,
                    "all_conv_tflops": fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
                    "algorithmic_gflop_per_step": fl / n_t / 1e9,
                    "algorithmic_gbytes_per_step": by / n_t / 1e9,
                    "hbm_frac_of_8TBs": (by / (ms * 1e-3)) / 8.0e12 if ms > 0 else 0.0}
        stage = {k: v / n_t for k, v in agg.items()}
        if stage.get("dec_ms", 0) > 0:
            stage["dec_tflops"] = stage["dec_flops"] / (stage["dec_ms"] * 1e-3) / 1e12
            stage["dec_hbm_frac"] = stage["dec_bytes"] / (stage["dec_ms"] * 1e-3) / 8.0e12

    # The same workload once more with every fp32 product exact (VITSMI_GEN_PRECISION=bf16x6), so that the line
    # carries both arithmetics of the generator: N=1 only, after (outside) the timed region of the headline value.
    exact = None
    if world == 1 and gen_nprod == 2 and not a.no_exact_check:
        try:
            os.environ["VITSMI_GEN_PRECISION"] = "bf16x6"
            pe = PipelinedSession.open(voice, device_id=local_rank, parts=max(1, a.parts))
            pe.set_seed(1234)
            ke = max(3, a.steps // 2)
            for _ in range(max(2, a.warmup)):
                pe.run_device(ids.data_ptr(), lens.data_ptr(), B, T, scales)
            torch.cuda.synchronize()
            te = time.perf_counter()
            ne = 0
            for _ in range(ke):
                pe.run_device(ids.data_ptr(), lens.data_ptr(), B, T, scales)
                ne += int(pe.last_y_lengths(B).sum()) * hop
            torch.cuda.synchronize()
            te = time.perf_counter() - te
            exact = {"gen_precision": "bf16x6", "gen_nprod": int(pe.hparam("gen_nprod")), "value": ne / te,
                     "unit": "samples/s", "steps": ke, "ms_per_step": te / ke * 1e3,
                     "note": "six bf16 plane products per fp32 product (each exact to 2^-24); same batch and handle count, measured "
                             "right after the headline run (chip already at its power / thermal limit: a standalone "
                             "`bench.py --gen-precision bf16x6` reads ~10 % higher)"}
            pe.close()
        except Exception as e:
            exact = {"gen_precision": "bf16x6", "value": None, "note": f"failed: {e}"}
        finally:
            os.environ["VITSMI_GEN_PRECISION"] = a.gen_precision

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:  # (N = 1 only: the other ranks would sit in the final barrier)
        try:
            cpu = cpu_baseline(voice, a.preset, T, scales, 1234)
        except Exception as e:  # the baseline is a report, never the product
            cpu = {"value": None, "unit": "samples/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}

    if rank == 0:
        value = samples_all / dt_max
        line = {
            "metric": "audio samples/sec (22.05 kHz), batch-32 256-phoneme utterances",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt_max / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if gen_nprod in (6, 2) else f"f32 + {a.gen_precision} vocoder (reduced precision)",
            "data": "synthetic",
            "dtype_note": ("fp32 MFMA (v_mfma_f32_32x32x2_f32) and fp32 VALU" if not sess_gen_sx else
                           "fp32 operands and fp32 accumulation everywhere; the generator's convs evaluate each fp32 product "
                           "as three fp16 MFMA products of its operands' two fp16 planes (h0g0 + h0g1 + h1g0, dropped term "
                           "<= 2^-24 relative; measured error vs float64 below the f32-MFMA engine's)" if gen_nprod == 2 else
                           "fp32 operands and fp32 accumulation everywhere; the generator's convs evaluate each fp32 product "
                           "exactly-to-2^-24 as six bf16 MFMA plane products (three bf16 planes per operand)"),
            "gen_precision": a.gen_precision,
            "rtf": dt_max / (samples_all / world / 22050.0) if samples_all else None,
            "config": {"workload": f"VITS full pipeline (encoder+duration+flow+HiFi-GAN), preset={a.preset}, "
                                   f"batch={B}/GPU x {T} phoneme ids, scales=[0.667,{scales[1]:.2f},0.8], "
                                   f"device Philox noise, seeded synthetic weights",
                       "preset": a.preset, "batch_per_gpu": B, "tokens": T, "hop": hop,
                       "pipeline_parts": len(pipe.parts),
                       "pipeline_host": "lockstep" if (a.lockstep or len(pipe.parts) == 1) else "one free-running host thread per part",
                       "samples_per_step": samples_all / a.steps,
                       "frames_per_id": samples_all / a.steps / hop / (B * world * T),
                       "weights": "RCCL broadcast of packed arena" if world > 1 else "local"},
            "roofline": roofline, "cpu_baseline": cpu, "exact_arithmetic": exact, "stages": stage,
        }
        if cpu and cpu.get("value"):
            line["gpu_over_cpu"] = value / world / cpu["value"]
        print(json.dumps(line), flush=True)
    pipe.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
