#!/usr/bin/env python3
"""bench.py — audio samples/s of the MI355X-native VITS path on synthetic fixed-length
phoneme batches (BASELINE.json: batch-32, 256-phoneme utterances, 22.05 kHz).

  python bench.py --gpus N --steps K --warmup W [--preset high|medium] [--batch 32] [--tokens 256]
                  [--total-batch 256]

One process per GPU.  `--gpus N` with N > 1 started by hand (no RANK in the environment) launches the N ranks itself
(`python -m torch.distributed.run`, fresh child processes) and relays rank 0's line; under torchrun the ranks find
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment.  Rank 0 reads and packs the .onnx, the packed weight
arena is broadcast over RCCL/xGMI (every rank verifies its checksum and compares it with a local pack), then every
rank synthesises its own utterances with no further communication.  Default: 32 utterances per GPU ("scaling":
"weak"); `--total-batch T`: T utterances split over the GPUs ("strong").

A step = one pass of the whole path (encoder + duration predictor + flow + vocoder) over one batch whose inputs are
already resident in HBM.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 MFMA (v_mfma_f32_32x32x2_f32) = fp32 vector peak
# split-operand engine: dense 16-bit MFMA peak (256 CUs x 4096 FLOP/clk x 2.4 GHz = 2516.6 TFLOP/s) / plane products
MFMA16_PEAK_TFLOPS = 2516.6
HBM_PEAK_BPS = 8.0e12      # HBM3E peak (MI355X_MICROARCH.md)
# scales[1]: ~3 frames per phoneme id with the synthetic weight sets (1.5 gave 2.4)
LENGTH_SCALE = {"high": 1.95, "medium": 1.95, "small": 1.95}
DTYPE = {2: "f32 (f16x3 split: fp32 operands as two fp16 planes, three MFMA products, fp32 accumulate)",
         6: "f32 (bf16x6 split: fp32 operands as three bf16 planes, six exact MFMA products, fp32 accumulate)",
         1: "f32 up to z + f16 vocoder (reduced precision: one fp16 plane per operand, one MFMA product, fp32 accumulate, "
            "fp16 activation storage)"}


def git_head():
    """Commit of this tree: from git where there is one (with "+dirty" if tracked files differ from it), else the commit the
    library's build record names (the GPU box receives no .git; "+" = the tree had uncommitted changes when it was built:
    config.source_sha, a hash over the sources themselves, is what identifies the code that ran)."""
    try:
        h = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                           timeout=5).stdout.strip()
        if h:
            d = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--untracked-files=no"], capture_output=True, text=True,
                               timeout=10).stdout.strip()
            return h + ("+dirty" if d else "")
    except Exception:
        pass
    try:
        info = json.load(open(os.path.join(ROOT, "phoonnx_amd", "_build_info.json")))
        return info["commit"] + ("+" if info.get("dirty") else "")
    except Exception:
        return None


def source_sha():
    """phoonnx_amd.build.source_sha(): sha256 over every file libvitsmi.so is compiled from (recomputable from a checkout)"""
    try:
        from phoonnx_amd import build
        return build.source_sha()
    except Exception:
        return None


def pmc_profile(preset):
    """The newest committed PMC summary of the SAME command (profiles/*_<preset>_b32_pmc.json, written by
    tools/profile_gpu.sh: FETCH_SIZE x 2 + WRITE_SIZE, separate passes, per MI355X_MICROARCH.md) -> (doc, file name)."""
    import glob
    import re

    # newest = highest (round, version) in the name rNN_vM_...: a fresh checkout gives every file the same mtime
    def ver(f):
        m = re.match(r"r(\d+)(?:_v(\d+))?_", os.path.basename(f))
        return (int(m.group(1)), int(m.group(2) or 0)) if m else (0, 0)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{preset}_b32_pmc.json")), key=ver)
    if not files:
        return None, None
    try:
        return json.load(open(files[-1])), os.path.basename(files[-1])
    except Exception:
        return None, None


def kname_key(name):
    """'void vitsmi::conv_sx_kernel<2, 4, 2, 2, 5120, false, false, 2>(vitsmi::SxArgs)' -> 'conv_sx_kernel<2,4,2,2,5120,false,false,2>'"""
    return name.split("(")[0].replace("void ", "").replace("vitsmi::", "").replace(" ", "").strip()


def pmc_traffic(preset, family):
    """HBM bytes per launch of a kernel FAMILY (every instantiation whose name contains `family`: "conv_sx" covers
    conv_sx_kernel AND conv_sx_pair_kernel), launch-weighted; None if no committed profile."""
    doc, fname = pmc_profile(preset)
    if not doc:
        return None
    tot = n = 0.0
    for name, c in doc.get("kernels", {}).items():
        if family in name and "hbm_read_bytes_per_launch" in c and "hbm_write_bytes_per_launch" in c:
            tot += (c["hbm_read_bytes_per_launch"] + c["hbm_write_bytes_per_launch"]) * c["launches"]
            n += c["launches"]
    if not n:
        return None
    return {"bytes_per_launch": tot / n, "source": fname, "commit": doc.get("commit"), "source_sha": doc.get("source_sha"),
            "length_scale": doc.get("length_scale")}


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(voice_path, preset, tokens, scales, seed, hop, budget_s=60.0):
    """What the reference's hot call costs on this box's host cores, on a bounded sample of the
    workload.  Preferred: onnxruntime's CPU provider with default session options, as phoonnx/voice.py:167-171 builds
    it - probed, absent on the build and GPU images (and the synthetic bench voice carries only the parameter nodes of
    the graph, so it would need a real export anyway).  Otherwise the faster of two restatements of the graph, both
    pinned to the reference-generated fixtures: PyTorch CPU kernels op by op (oracle/torch_baseline.py, oneDNN/MKL on
    all cores - the closest stand-in for onnxruntime) and the C/OpenMP checker (oracle/vits_oracle.c)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    out = {"unit": "samples/s", "cores": os.cpu_count(), "kind": "port"}
    try:
        import onnxruntime  # noqa: F401
        out["onnxruntime"] = "importable, but the bench voice holds only the graph's parameter nodes (phoonnx_amd/synth.py)"
    except Exception as e:  # noqa: BLE001
        out["onnxruntime"] = f"not available on this box ({type(e).__name__})"
    rng = np.random.default_rng(seed)
    cands = {}

    def sample(B):
        ids = rng.integers(0, 256, size=(B, tokens)).astype(np.int64)
        lens = np.full((B,), tokens, np.int64)
        ndp = rng.standard_normal((B, 2, tokens)).astype(np.float32)
        nz = rng.standard_normal((B, 192, tokens * 12)).astype(np.float32)
        return ids, lens, ndp, nz

    def timed(name, infer, hop, threads, share=0.5):
        # one short warm-up call (thread pool, allocator, code paths), one utterance to learn the rate, then ONE batch
        # sized to fill what is left of this implementation's share of the budget (at most the bench batch, 32)
        ids, lens, ndp, nz = sample(1)
        infer(ids[:, :64], np.full((1,), 64, np.int64), ndp[:, :, :64], nz)
        t0 = time.perf_counter()
        r = infer(ids, lens, ndp, nz)
        t1 = time.perf_counter() - t0
        B = int(max(1, min(32, (budget_s * share - t1) / max(t1, 1e-3) * 0.8)))
        if B > 1:
            ids, lens, ndp, nz = sample(B)
            t0 = time.perf_counter()
            r = infer(ids, lens, ndp, nz)
            t1 = time.perf_counter() - t0
        samples = int(np.asarray(r["y_lengths"]).sum()) * hop
        cands[name] = {"value": samples / t1, "B": B, "samples": samples, "seconds": t1, "threads": threads,
                       "rtf": t1 / (samples / 22050.0)}

    try:
        import torch
        from torch_baseline import TorchVits
        m = TorchVits(voice_path)
        # thread sweep: on a 2 x 64-core host the oneDNN / MKL kernels of these small convs do not scale to every core
        # (round 2: 128 threads measured BELOW the survey's 8-thread figure).  One utterance per setting; the sweep is reported.
        ncpu = os.cpu_count() or 1
        sweep = {}
        ids1, lens1, ndp1, nz1 = sample(1)
        # (not every core: 256 threads measured 1.2 k samples/s on the 2 x 64-core / 256-thread host - one utterance took 160 s)
        for nt in sorted({t for t in (8, 16, 32, 64, 128) if t <= ncpu} or {ncpu}):
            torch.set_num_threads(nt)
            m.infer(ids1[:, :64], np.full((1,), 64, np.int64), scales, None, ndp1[:, :, :64], nz1[:, :m.C])
            t0 = time.perf_counter()
            r1 = m.infer(ids1, lens1, scales, None, ndp1, nz1[:, :m.C])
            sweep[nt] = int(np.asarray(r1["y_lengths"]).sum()) * m.hop / (time.perf_counter() - t0)
        best_nt = max(sweep, key=sweep.get)
        torch.set_num_threads(best_nt)
        out["torch_thread_sweep"] = {"unit": "samples/s at B=1", **{str(k): v for k, v in sweep.items()}}
        # (a) the reference's own call shape: one utterance per call, calls one after the other (voice.py:265-269, 350-351)
        t0 = time.perf_counter()
        n1 = k1 = 0
        while time.perf_counter() - t0 < budget_s * 0.15:
            r1 = m.infer(ids1, lens1, scales, None, ndp1, nz1[:, :m.C])
            n1 += int(np.asarray(r1["y_lengths"]).sum()) * m.hop
            k1 += 1
        t1 = time.perf_counter() - t0
        cands["torch_cpu_b1"] = {"value": n1 / t1, "B": 1, "samples": n1, "seconds": t1, "threads": best_nt, "rtf": t1 / (n1 / 22050.0),
                                 "shape": f"{k1} sequential calls of one utterance"}
        # (b) the bench's batch in one call
        timed("torch_cpu", lambda i, l, a, b: m.infer(i, l, scales, None, a, b[:, :m.C]), m.hop, best_nt, share=0.35)
        cands["torch_cpu"]["shape"] = "one batched call"
        del m
        # (c) what the whole host can do: P processes x best_nt threads, each rendering utterance after utterance (P x best_nt =
        # half the logical CPUs, i.e. the physical cores of an SMT-2 host)
        P = ncpu // (2 * best_nt)
        if P >= 2:
            secs = budget_s * 0.2
            start = time.time() + 20.0   # (every worker has loaded the voice and warmed up by then)
            cmd = [sys.executable, os.path.join(ROOT, "oracle", "torch_baseline.py"), "--worker", voice_path, str(tokens),
                   str(float(scales[1])), str(best_nt), str(secs), str(start)]
            procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(P)]
            tot_n, max_t, ok = 0, 0.0, True
            for pr in procs:
                try:
                    o, _ = pr.communicate(timeout=secs + 120)
                    a_, b_ = o.split()[-2:]
                    tot_n += int(a_)
                    max_t = max(max_t, float(b_))
                except Exception:  # noqa: BLE001
                    ok = False
                    pr.kill()
            if ok and max_t > 0:
                cands["torch_cpu_procs"] = {"value": tot_n / max_t, "B": 1, "samples": tot_n, "seconds": max_t, "threads": P * best_nt,
                                            "rtf": max_t / (tot_n / 22050.0),
                                            "shape": f"{P} processes x {best_nt} threads, each one utterance per call"}
    except Exception as e:  # noqa: BLE001 - the baseline is a report, never the product
        out["torch_cpu_error"] = f"{type(e).__name__}: {e}"
    try:
        import vits_oracle
        try:
            o = vits_oracle.VitsOracle(voice_path, native=True)
        except Exception:  # noqa: BLE001
            o = vits_oracle.VitsOracle(voice_path, native=False)
        timed("c_openmp", lambda i, l, a, b: o.infer(i, l, scales, None, a, b[:, :o.inter_channels]), hop,
              int(o.lib.vo_num_threads()), share=0.2)
        cands["c_openmp"]["shape"] = "one batched call"
    except Exception as e:  # noqa: BLE001
        out["c_openmp_error"] = f"{type(e).__name__}: {e}"
    if not cands:
        out.update(value=None, sample="failed")
        return out
    # `value` = the best the host does in ANY of these configurations (implementation x call shape x threads x processes)
    name = max(cands, key=lambda k: cands[k]["value"])
    b = cands[name]
    impl = {"torch_cpu": "PyTorch CPU kernels op by op (oracle/torch_baseline.py)",
            "torch_cpu_b1": "PyTorch CPU kernels op by op (oracle/torch_baseline.py)",
            "torch_cpu_procs": "PyTorch CPU kernels op by op (oracle/torch_baseline.py)",
            "c_openmp": "C/OpenMP restatement (oracle/vits_oracle.c)"}[name]
    out.update(value=b["value"], cores=b["threads"], rtf=b["rtf"], implementation=name,
               sample=f"{impl}, {b['shape']}, {tokens} ids each, same voice and scales, {b['samples']} samples in "
                      f"{b['seconds']:.1f}s after a warm-up call; best of {sorted(cands)} (a {budget_s:.0f}s budget in all; "
                      f"thread count = best of the sweep)",
               candidates={k: {"value": v["value"], "B": v["B"], "threads": v["threads"], "shape": v["shape"]} for k, v in cands.items()},
               host_cores=os.cpu_count())
    return out


# ------------------------------------------------------------------------------------------------ launcher
def launch_ranks(a, argv, timeout_s=None):
    """`--gpus N` without a torchrun environment: start the N ranks as fresh child processes (nothing in this process
    has touched the GPU) and relay rank 0's JSON line.  The children get `timeout_s` (BENCH_LAUNCH_TIMEOUT_S, default 1800 s)
    to finish; on a timeout, a non-zero exit or a missing result line ONE machine-readable line says what happened:
    {"error": ..., "rank": ..., "stderr_tail": ...} - the first such line a rank printed itself (phoonnx_amd.sharding.
    report_rank_failure), else one made here from the launcher's view."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if timeout_s is None:
        timeout_s = float(os.environ.get("BENCH_LAUNCH_TIMEOUT_S", "1800"))
    # (own process group: on a timeout every rank goes, not just the elastic agent)
    pr = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    timed_out = False
    try:
        out, err = pr.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        timed_out = True
        import signal
        try:
            os.killpg(pr.pid, signal.SIGKILL)
        except Exception:  # noqa: BLE001
            pr.kill()
        out, err = pr.communicate()
    sys.stderr.write(err or "")
    line = rank_err = None
    for ln in (out or "").splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        elif ln.startswith("{") and '"error"' in ln and rank_err is None:
            rank_err = ln
    if line and not timed_out and pr.returncode == 0:
        print(line, flush=True)
        return 0
    if rank_err is None:
        rank_err = json.dumps({"error": f"timed out after {timeout_s:.0f} s" if timed_out else
                               (f"the ranks exited with code {pr.returncode}" if pr.returncode else "no result line from rank 0"),
                               "rank": None, "n_gpus": a.gpus, "stderr_tail": (err or "")[-1500:]})
    print(rank_err, flush=True)
    return pr.returncode if pr.returncode else 1


def pctl(xs):
    xs = np.asarray(xs, np.float64)
    return {"median": float(np.median(xs)), "p10": float(np.percentile(xs, 10)), "p90": float(np.percentile(xs, 90)),
            "n": int(xs.size)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--preset", default="high", choices=["high", "medium", "small"])
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU (weak scaling)")
    ap.add_argument("--total-batch", type=int, default=0,
                    help="strong scaling: this many utterances in total, split evenly over the GPUs (BASELINE config 5: 256)")
    ap.add_argument("--tokens", type=int, default=256)
    ap.add_argument("--speakers", type=int, default=1,
                    help="> 1: a multi-speaker voice (speaker-embedding path, gin 512) with sid uniform in [0, speakers) "
                         "(BASELINE config 4: --preset medium --speakers 4 --batch 64 --mixed-lengths --gen-precision f16)")
    ap.add_argument("--mixed-lengths", action="store_true",
                    help="utterance lengths uniform in [tokens/4, tokens] (seed 1235), zero-padded, instead of all = tokens")
    ap.add_argument("--prewarm-s", type=float, default=2.0,
                    help="untimed passes for this many seconds before the W warm-up steps of a voice's first measurement "
                         "(a fresh process' first passes touch the workspace for the first time; 0 = none)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary measurements of the N=1 line (exact arithmetic, host-in/host-out, the medium "
                         "preset, per-step percentiles)")
    ap.add_argument("--gen-precision", default="f16x3", choices=["f16x3", "bf16x6", "f16"],
                    help="arithmetic of the generator's convs (fp32 operands and results in every mode).  f16x3 (default): "
                         "two fp16 planes per operand, three MFMA products per fp32 product, error no larger than the "
                         "f32-MFMA engine's, range-guarded; bf16x6: three bf16 planes, six products, every product exact; "
                         "f16: the declared reduced-precision vocoder of BASELINE config 4 - fp16 storage, one product - "
                         "(reported with its dtype, never as the headline number)")
    ap.add_argument("--no-exact-check", action="store_true",
                    help="skip the second, shorter measurement of the same workload with the six-product exact arithmetic")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) and broadcast the weight arena even at world size 1 "
                         "(exercises the N > 1 code path on a one-GPU box)")
    ap.add_argument("--lockstep", action="store_true",
                    help="enqueue the parts of every step from one host thread and join them per step, instead of one "
                         "free-running host thread per part (two serving workers)")
    ap.add_argument("--schedule", default="alternate", choices=["split", "alternate"],
                    help="how the K passes meet the engine handles: split = every pass divided over the handles (rows each); "
                         "alternate (default) = whole passes dealt to the handles in turn (request-level pipelining: two serving "
                         "workers, each rendering complete batches - the generator keeps the full batch's grids while the next "
                         "pass's token / frame stages run under it: r05j, same box: default voice 757 -> 799 M samples/s, headline "
                         "135.2 -> 135.9 M)")
    ap.add_argument("--parts", type=int, default=2,
                    help="render each batch as this many sub-batches on as many engine handles / HIP streams sharing "
                         "one weight arena (PipelinedSession); 1 = a single handle")
    ap.add_argument("--also-parts", type=int, default=3,
                    help="handles for the `also` (medium) measurement: that voice's token / frame-domain stages are half of "
                         "its step, and three sub-batches hide more of them than two (573 vs 534 M samples/s; four or "
                         "more handles per process fall off a cliff: 413 M)")
    a = ap.parse_args()

    if a.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(a, sys.argv[1:]))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world_env:
        sys.exit(f"bench.py: --gpus {a.gpus} but the launcher started {world_env} rank(s) (WORLD_SIZE)")
    os.environ["VITSMI_GEN_PRECISION"] = a.gen_precision

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world_env > 1 or a.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    world = dist.get_world_size() if dist else 1

    from phoonnx_amd import MiSession, PipelinedSession
    from phoonnx_amd.sharding import arena_checksum, open_sharded
    from phoonnx_amd.synth import write_voice

    cache = os.environ.get("VITSMI_BENCH_CACHE", "/tmp/vitsmi_bench")

    def voice_path(preset, speakers=1):
        p = os.path.join(cache, f"synth_{preset}{'' if speakers <= 1 else f'_spk{speakers}'}.onnx")
        if rank == 0 and not os.path.exists(p):
            os.makedirs(cache, exist_ok=True)
            write_voice(p + ".tmp", preset, seed=1234, **({"n_speakers": speakers} if speakers > 1 else {}))
            os.replace(p + ".tmp", p)
        return p

    voice = voice_path(a.preset, a.speakers)
    if dist:
        dist.barrier()

    # weights: rank 0 reads + packs, RCCL broadcast of the arena (checksum-verified), every rank opens on its GPU
    t_load = time.perf_counter()
    open_stats = {}
    sess, arena_keepalive = open_sharded(voice, local_rank, dist, force_broadcast=a.force_dist, stats=open_stats)
    t_load = time.perf_counter() - t_load
    weights = {"mode": "local pack", "load_s": t_load}
    if arena_keepalive is not None:
        # every rank: the arena that arrived over RCCL equals what packing the file here would give
        probe = MiSession(voice, host_only=True)
        local_sum = arena_checksum(torch.from_numpy(np.array(probe.arena_host(), copy=True)))
        probe.close()
        got = arena_checksum(arena_keepalive)
        ok = torch.tensor([1 if got == local_sum else 0], device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) != 1:
            sys.exit(f"rank {rank}: broadcast weight arena differs from a local pack ({got:#x} vs {local_sum:#x})")
        weights = {"mode": "RCCL broadcast of the packed arena from rank 0", "bytes": int(arena_keepalive.numel()),
                   "checksum": f"{got & (2**64 - 1):#018x}", "verified_equal_to_local_pack_on_every_rank": True,
                   "load_s": t_load, "pack_s_rank0": open_stats.get("pack_s"), "broadcast_s": open_stats.get("bcast_s"),
                   "layout_open_s": open_stats.get("open_s")}
    hop = sess.hparam("hop")
    gen_nprod = int(sess.hparam("gen_nprod"))

    from phoonnx_amd.sharding import partition
    T = a.tokens
    shard_info = None

    def global_batch(seed, n):
        g = torch.Generator(device="cpu").manual_seed(seed)
        ids = torch.randint(0, 256, (n, T), generator=g, dtype=torch.int64)
        lens = torch.full((n,), T, dtype=torch.int64)
        if a.mixed_lengths:  # ragged lengths, zero-padded ids, padding mask in play
            g2 = torch.Generator(device="cpu").manual_seed(1235 + seed)
            lens = torch.randint(max(1, T // 4), T + 1, (n,), generator=g2, dtype=torch.int64)
            lens[0] = T
            ids = ids * (torch.arange(T)[None, :] < lens[:, None])
        return ids, lens

    if a.total_batch:
        # strong scaling (BASELINE config 5): ONE request of total_batch utterances, the same on every rank (same seed),
        # dealt to the ranks by sharding.partition (length-sorted snake deal: equal work per rank, little padding inside
        # a shard); this rank renders its shard, padded to the shard's own longest utterance
        ids_all, lens_all = global_batch(1234, a.total_batch)
        shards, _inv = partition(lens_all.numpy(), world)
        mine = torch.from_numpy(np.ascontiguousarray(shards[rank]))
        lens_mine = lens_all[mine]
        Tm = int(lens_mine.max()) if len(mine) else 1
        ids_mine = ids_all[mine][:, :Tm].contiguous()
        B, T = int(len(mine)), Tm
        if B == 0:
            sys.exit(f"rank {rank}: --total-batch {a.total_batch} leaves this rank without an utterance")
        shard_info = {"path": "sharding.partition (length-sorted snake deal)", "total_batch": a.total_batch,
                      "rank0_rows": B, "rank0_padded_tokens": T, "rank0_tokens": int(lens_mine.sum()),
                      "tokens_per_rank": [int(lens_all[torch.from_numpy(np.ascontiguousarray(sh))].sum()) for sh in shards],
                      "rows_per_rank": [int(len(sh)) for sh in shards]}
    else:
        B = a.batch

    def make_inputs(seed, Bn=None):
        if a.total_batch:
            return ids_mine, lens_mine
        return global_batch(seed, B if Bn is None else Bn)

    def make_sid(seed, first, Bn=None):
        Bn = B if Bn is None else Bn
        if int(first.hparam("n_speakers")) <= 1:
            return None
        g = torch.Generator(device="cpu").manual_seed(77 + seed)
        return torch.randint(0, int(first.hparam("n_speakers")), (Bn,), generator=g, dtype=torch.int64)

    prewarmed = {}
    reserved = {}

    def measure(first, preset, steps, warmup, parts, lockstep, seed, inputs_h=None, pipe=None, schedule=None):
        """K timed passes of the whole path on `parts` handles sharing `first`'s arena -> (dt, samples, pipe).
        inputs_h: (ids, lens, sid) host tensors of another workload than the command line's (the config-4 block).
        pipe: an existing pipeline of `first` to run on instead of a new one."""
        schedule = schedule or a.schedule
        if pipe is None:
            pipe = PipelinedSession(first, max(1, parts))
        pipe.set_seed(1234 + rank * 16)
        scales = np.array([0.667, LENGTH_SCALE[preset], 0.8], np.float32)
        if inputs_h is not None:
            ids_h, lens_h, sid_h = inputs_h
        else:
            ids_h, lens_h = make_inputs(seed)
            sid_h = make_sid(seed, first)
        B, T = int(ids_h.shape[0]), int(ids_h.shape[1])
        ids, lens = ids_h.cuda(), lens_h.cuda()
        sid = None if sid_h is None else sid_h.cuda()
        sid_ptr = None if sid is None else sid.data_ptr()
        torch.cuda.synchronize()

        def run_steps(k):
            if lockstep or len(pipe.parts) == 1:
                n = 0
                for _ in range(k):
                    pipe.run_device(ids.data_ptr(), lens.data_ptr(), B, T, scales, sid_ptr)
                    n += int(pipe.last_y_lengths(B).sum()) * hop
                return n
            return int(pipe.run_device_steps(ids.data_ptr(), lens.data_ptr(), B, T, scales, k, sid_ptr,
                                             alternate=schedule == "alternate").sum()) * hop

        # a fresh process' first passes are slow (first touch of ~60 GB of workspace, clock ramp: the first bench run on a fresh
        # box measured 567 M samples/s where every later one measured 740-750 M): pre-warm for --prewarm-s seconds (untimed,
        # like the model load), THEN the W warm-up steps and the K timed ones of the contract
        # A batch's frame count depends on the predicted durations, i.e. on each pass' noise: the headline batch needs 13.6-15.4
        # GB of frame-domain workspace per handle from pass to pass, and a pass that needs more than any before it frees and
        # reallocates the slab (device-wide sync + hipMalloc of 15-45 GB: 0.3 ms to 5 s, measured) - inside the timed region
        # that read as 69 instead of 139 M samples/s (r05t) and 88 instead of 1022 M on config 4.  As a serving process would
        # at start-up, size the workspaces once for the largest request admitted: this workload + 25 % more frames than the
        # probe pass rendered (MiSession.reserve -> vits_reserve).
        if not reserved.get(id(pipe)):
            reserved[id(pipe)] = True
            pipe.run_device(ids.data_ptr(), lens.data_ptr(), B, T, scales, sid_ptr)
            f_probe = int(pipe.last_y_lengths(B).max())
            pipe.sync()
            pipe.reserve(B, T, int(f_probe * 1.25) + 64,
                         whole_batch=(schedule == "alternate" or a.schedule == "alternate") and not (lockstep or len(pipe.parts) == 1))
        if a.prewarm_s > 0 and not prewarmed.get(id(first)):
            prewarmed[id(first)] = True
            t_pw = time.perf_counter()
            while time.perf_counter() - t_pw < a.prewarm_s:
                run_steps(2)
        if warmup > 0:
            run_steps(warmup)
        pipe.sync()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        samples = run_steps(steps)
        pipe.sync()  # (also the f16 range verdict of the last pass: a violation raises here)
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if os.environ.get("BENCH_TRACE"):  # (diagnostics: when each timed pass' call returned on its worker, ms from the start)
            st = getattr(pipe, "last_pass_stamps", None) or []
            sys.stderr.write(f"bench trace: {preset} parts={len(pipe.parts)} steps={steps} dt={dt * 1e3:.1f} ms; pass returns at "
                             f"{[round((x - t0) * 1e3, 1) for x in st]}\n")
        return dt, samples, pipe, (ids, lens, ids_h, lens_h, scales, sid, sid_h)

    def roofline_of(s, preset, inputs, n_t):
        """Per-kernel timing with HIP events on the engine's own stream (vits_set_timing), one handle, whole batch."""
        ids, lens, _, _, scales, sid, _ = inputs
        sid_ptr = None if sid is None else sid.data_ptr()
        s.set_timing(True)
        fl = ms = by = 0.0
        launches = 0
        agg = {}
        per_k = {}
        s.run_device(ids.data_ptr(), lens.data_ptr(), B, T, scales, sid_ptr)  # (untimed: creates the handle's HIP events)
        s.stats()
        for _ in range(n_t):
            s.run_device(ids.data_ptr(), lens.data_ptr(), B, T, scales, sid_ptr)
            st = s.stats()
            for r in s.launch_records():
                e = per_k.setdefault(kname_key(r["kernel"]), {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "stage": r["stage"]})
                e["launches"] += 1
                e["ms"] += r["ms"]
                e["flops"] += r["flops"]
                e["bytes"] += r["bytes"]
            fl += st["conv_flops"]
            by += st["conv_bytes"]
            ms += st["conv_ms"]
            launches += st["conv_launches"]
            for k in ("enc_ms", "dp_ms", "flow_ms", "dec_ms", "total_ms", "dec_flops", "dec_bytes", "flow_flops",
                      "sx_flops", "sx_bytes", "sx_ms", "sx_launches", "total_launches"):
                agg[k] = agg.get(k, 0.0) + st[k]
            rng_stats = {"f16_peak_max": st["f16_peak_max"], "f16_peak_min": st["f16_peak_min"],
                         "f16_launches_tracked": st["f16_tracked"], "f16_saturated": st["f16_saturated"]}
        s.set_timing(False)
        nprod = int(s.hparam("gen_nprod"))
        kby = by
        if agg.get("sx_launches", 0) > 0:
            kfl, kms, kn, kby = agg["sx_flops"], agg["sx_ms"], int(agg["sx_launches"]), agg["sx_bytes"]
            kname = ("conv_sx_kernel (implicit-GEMM Conv1d, fp32 operands as 2 fp16 planes, 3 x v_mfma_f32_16x16x32_f16 per product)"
                     if nprod == 2 else
                     "conv_sx_kernel (implicit-GEMM Conv1d, fp16 operands, one v_mfma_f32_16x16x32_f16 per product, fp32 accumulate)"
                     if nprod == 1 else
                     "conv_sx_kernel (implicit-GEMM Conv1d, fp32 operands as 3 bf16 planes, v_mfma_f32_32x32x16_bf16 plane products)")
            peak = MFMA16_PEAK_TFLOPS / (3 if nprod == 2 else nprod)
        else:
            kfl, kms, kn = fl, ms, launches
            kname = "conv_engine_kernel (implicit-GEMM Conv1d, v_mfma_f32_32x32x2_f32)"
            peak = FP32_PEAK_TFLOPS
        ach = kfl / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
        # (the whole family: conv_sx_kernel AND conv_sx_pair_kernel - launches_per_step counts both)
        tr = pmc_traffic(preset, "conv_sx" if agg.get("sx_launches", 0) > 0 else "conv_engine") if (B, T) == (32, 256) else None
        # the six instantiations with the most time: live HIP-event time and algorithmic work from this run, PMC columns
        # (HBM bytes, matrix-pipe busy share, clock) joined by name from the committed profile of the same command
        pdoc, pfile = pmc_profile(preset) if (B, T) == (32, 256) else (None, None)
        pk = {kname_key(k): v for k, v in (pdoc or {}).get("kernels", {}).items()}
        rows = []
        for name, e in sorted(per_k.items(), key=lambda kv: -kv[1]["ms"])[:6]:
            n_l = e["launches"]
            is_sx = name.startswith("conv_sx")
            k_peak = MFMA16_PEAK_TFLOPS / (3 if nprod == 2 else nprod) if is_sx else FP32_PEAK_TFLOPS
            sec = e["ms"] * 1e-3
            c = pk.get(name, {})
            rd, wr = c.get("hbm_read_bytes_per_launch"), c.get("hbm_write_bytes_per_launch")
            rows.append({
                "name": name, "stage": ["enc", "dp", "flow", "dec"][e["stage"]] if 0 <= e["stage"] < 4 else None,
                "launches_per_step": n_l / n_t, "avg_ms": e["ms"] / n_l, "share_of_conv_ms": e["ms"] / ms if ms > 0 else None,
                "algorithmic_gbytes": e["bytes"] / n_l / 1e9, "algorithmic_gflop": e["flops"] / n_l / 1e9,
                "mfma_frac": e["flops"] / sec / 1e12 / k_peak if sec > 0 else None,
                "hbm_frac": e["bytes"] / sec / HBM_PEAK_BPS if sec > 0 else None,
                "pmc_read_gbytes": None if rd is None else rd / 1e9, "pmc_write_gbytes": None if wr is None else wr / 1e9,
                "pmc_over_algorithmic": None if rd is None or wr is None or not e["bytes"] else (rd + wr) / (e["bytes"] / n_l),
                # the bytes the launch REALLY moves (PMC, committed profile) over its live time: a fused launch never writes its
                # intermediate, so its layer-granular hbm_frac overstates what the memory system does
                "hbm_frac_pmc": None if rd is None or wr is None or sec <= 0 else (rd + wr) * n_l / sec / HBM_PEAK_BPS,
                "mfma_util_pct": c.get("MfmaUtil_pct"), "clock_ghz": c.get("effective_clock_GHz")})
            rows[-1]["frac"] = max(rows[-1]["mfma_frac"] or 0.0, rows[-1]["hbm_frac"] or 0.0)
        # Which roof bounds this kernel family on this voice: t_min = max(FLOPs / matrix peak, layer-granular bytes /
        # 8 TB/s).  The LJSpeech-size voice is matrix-bound (152 FLOP/B), phoonnx's default voice is
        # HBM-bound under the split arithmetic (60 FLOP/B against a ridge of 838.9 / 8 = 105).
        t_mfma, t_hbm = kfl / (peak * 1e12), kby / HBM_PEAK_BPS
        hbm_ach = kby / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
        if t_hbm > t_mfma:
            roof = {"bound": "hbm", "kernel": kname, "achieved": hbm_ach, "peak": HBM_PEAK_BPS / 1e9, "unit": "GB/s",
                    "frac": hbm_ach / (HBM_PEAK_BPS / 1e9)}
        else:
            roof = {"bound": "mfma", "kernel": kname, "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak}
        roof.update({
                "mfma_frac": ach / peak, "hbm_frac": hbm_ach / (HBM_PEAK_BPS / 1e9),
                "algorithmic_gbytes_per_launch": kby / max(kn, 1) / 1e9,
                "traffic": (tr or {}).get("bytes_per_launch"),
                "traffic_source": (tr or {}).get("source"), "traffic_commit": (tr or {}).get("commit"),
                "traffic_source_sha": (tr or {}).get("source_sha"),  # (== config.source_sha: the profile is of the code that ran)
                "traffic_note": "launch-weighted mean over every conv_sx_kernel and conv_sx_pair_kernel instantiation",
                "kernels": rows, "kernels_pmc_source": pfile,
                "launches_per_step": kn // n_t, "avg_launch_ms": kms / max(kn, 1),
                "algorithmic_gflop_per_launch": kfl / max(kn, 1) / 1e9,
                "all_conv_launches_per_step": launches // n_t,
                "all_kernel_launches_per_step": int(agg.get("total_launches", 0)) // n_t,
                "all_conv_tflops": fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
                "algorithmic_gflop_per_step": fl / n_t / 1e9,
                "algorithmic_gbytes_per_step": by / n_t / 1e9,
                "hbm_frac_of_8TBs": (by / (ms * 1e-3)) / 8.0e12 if ms > 0 else 0.0,
                "note": "one handle, whole batch, HIP events around every conv launch (serialises the stream): the step "
                        "this describes is stages.total_ms, not ms_per_step of the pipelined headline run"})
        stage = {k: v / n_t for k, v in agg.items()}
        if stage.get("dec_ms", 0) > 0:
            stage["dec_tflops"] = stage["dec_flops"] / (stage["dec_ms"] * 1e-3) / 1e12
            stage["dec_hbm_frac"] = stage["dec_bytes"] / (stage["dec_ms"] * 1e-3) / 8.0e12
        # the same passes once more with the five stage marks only (vits_set_timing(2)): no event records between the conv
        # launches, i.e. the stage times of the unserialised stream
        s.set_timing(2)
        s.run_device(ids.data_ptr(), lens.data_ptr(), B, T, scales, sid_ptr)
        s.stats()
        marks = {}
        for _ in range(n_t):
            s.run_device(ids.data_ptr(), lens.data_ptr(), B, T, scales, sid_ptr)
            st = s.stats()
            for k in ("enc_ms", "dp_ms", "flow_ms", "dec_ms", "total_ms"):
                marks[k] = marks.get(k, 0.0) + st[k] / n_t
        s.set_timing(False)
        if marks.get("dec_ms", 0) > 0 and stage.get("dec_bytes", 0) > 0:
            marks["dec_hbm_frac"] = stage["dec_bytes"] / (marks["dec_ms"] * 1e-3) / 8.0e12
            marks["dec_tflops"] = stage["dec_flops"] / (marks["dec_ms"] * 1e-3) / 1e12
        marks["note"] = "stage marks only (no events between the conv launches): the stages of the unserialised one-handle step"
        stage["stage_marks_only"] = marks
        return roof, stage, rng_stats

    # ---------------------------------------------------------------- the headline measurement
    dt, samples, pipe, inputs = measure(sess, a.preset, a.steps, a.warmup, a.parts, a.lockstep, 1234 + rank)
    tot = torch.tensor([dt, float(samples)], dtype=torch.float64, device="cuda")
    per_rank = None
    if dist:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tot.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt_max, samples_all = float(mx[0]), float(sm[1])
        # every rank's own numbers, for rank 0's line (outside the timed region; a few hundred bytes over the host path)
        props = torch.cuda.get_device_properties(local_rank)
        bus = ":".join(f"{getattr(props, k):02x}" for k in ("pci_domain_id", "pci_bus_id", "pci_device_id") if hasattr(props, k)) or None
        mine_rec = {"rank": rank, "device": local_rank, "pci_bus": bus, "device_uuid": str(getattr(props, "uuid", "")) or None,
                    "rows": B, "padded_tokens": T, "ms_per_step": dt / a.steps * 1e3,
                    "samples_per_step": samples / a.steps, "samples_per_s": samples / dt if dt > 0 else None,
                    "load_s": t_load, "broadcast_s": open_stats.get("bcast_s"),
                    "layout_open_s": open_stats.get("open_s"), "arena_checksum": weights.get("checksum")}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine_rec)
    else:
        dt_max, samples_all = dt, float(samples)

    roofline = stage = f16_range = None
    if rank == 0 and not a.no_roofline:
        roofline, stage, f16_range = roofline_of(sess, a.preset, inputs, max(3, min(a.steps, 5)))

    extras = world == 1 and not a.no_extras
    ids, lens, ids_h, lens_h, scales, sid, sid_h = inputs
    sid_ptr = None if sid is None else sid.data_ptr()

    # per-step wall times (lock-step: every step joined), median / p10 / p90
    step_pct = None
    if extras:
        per = []
        for _ in range(max(a.steps, 10)):
            t0 = time.perf_counter()
            pipe.run_device(ids.data_ptr(), lens.data_ptr(), B, T, scales, sid_ptr)
            pipe.sync()
            per.append((time.perf_counter() - t0) * 1e3)
        step_pct = dict(pctl(per), unit="ms", note="lock-step passes, each one synchronised (no overlap between passes)")

    # the reference's call shape: host arrays in, host fp32 waveform out (session.run, voice.py:374-377) - never `value`
    host_io = None
    if extras:
        ids_n, lens_n = ids_h.numpy(), lens_h.numpy()
        sid_n = None if sid_h is None else sid_h.numpy()
        pipe.synthesize_batch(ids_n, lens_n, scales, sid_n)
        per, n_s = [], 0
        for _ in range(max(3, a.steps // 2)):
            t0 = time.perf_counter()
            r = pipe.synthesize_batch(ids_n, lens_n, scales, sid_n)
            per.append(time.perf_counter() - t0)
            n_s += int(r["y_lengths"].sum()) * hop
        host_io = {"value": n_s / sum(per), "unit": "samples/s", "ms_per_step": 1e3 * sum(per) / len(per),
                   "over_value": (n_s / sum(per)) / (samples / dt) if samples else None,
                   "note": "PipelinedSession.synthesize_batch: host int64 ids in, ONE host fp32 [B,1,1,S] array out (what "
                           "session.run returns): H2D of the ids, the run, D2H by the DMA engine straight into the "
                           "returned (pinned) array, each part's copy-out under the other parts' render"}

    # the schedule the roofline block describes (one handle, whole batch) as a throughput figure of its own, untimed
    # by events: `value` is the pipelined schedule (config.pipeline_parts handles), the kernel rows are per launch and
    # hold for both
    one_handle = None
    if extras and len(pipe.parts) > 1:
        k1 = max(3, a.steps // 2)
        dt1, n1, _p1, _ = measure(sess, a.preset, k1, 1, 1, True, 1234 + rank)  # (shares `sess`: closed with `pipe`)
        _p1.close(close_first=False)
        one_handle = {"value": n1 / dt1, "unit": "samples/s", "steps": k1, "ms_per_step": dt1 / k1 * 1e3,
                      "note": "the same batch on ONE engine handle / stream: the schedule of the `roofline` and `stages` blocks"}

    # The round-4 schedule of the same handles (every pass split over them, rows each), so that rounds can be compared like
    # for like: under it a request's latency is one step; under `alternate` n whole requests are in flight on n handles.
    split_sched = None
    if extras and len(pipe.parts) > 1 and a.schedule == "alternate" and not a.lockstep:
        ks = max(3, a.steps // 2)
        dts, ns, _ps, _ = measure(sess, a.preset, ks, 2, a.parts, a.lockstep, 1234 + rank, pipe=pipe, schedule="split")
        split_sched = {"value": ns / dts, "unit": "samples/s", "steps": ks, "ms_per_step": dts / ks * 1e3,
                       "note": "--schedule split: every pass divided over the handles (the bench default up to round 4)"}

    # The same batch as the exported graph itself renders it (tails="reference"): its generator is not masked, so every
    # utterance is rendered to the longest one's length.  `value` does not render those tails (the default: each generator
    # launch ends an utterance's tensors gen_rf_frames behind its end; every valid sample is bit-identical, tests/
    # test_gpu_fullsize.py::test_ragged_rendering_equals_the_padded_rendering_on_every_valid_sample); both count VALID
    # samples only.
    padded = None
    if extras and B > 1:
        try:
            pipe.set_tails("reference")
            kp = max(3, a.steps // 2)
            dtp, npd, _pp, _ = measure(sess, a.preset, kp, 2, a.parts, a.lockstep, 1234 + rank, pipe=pipe)
            padded = {"value": npd / dtp, "unit": "samples/s", "steps": kp, "ms_per_step": dtp / kp * 1e3,
                      "over_value": (npd / dtp) / (samples / dt) if samples else None,
                      "note": "tails=\"reference\": the graph's own padded rendering of the same batch (every utterance rendered to the "
                              "longest one's length, as onnxruntime would); same valid samples bit for bit, same count of them"}
        except Exception as e:  # noqa: BLE001
            padded = {"value": None, "note": f"failed: {type(e).__name__}: {e}"}
        finally:
            pipe.set_tails("zero")

    # the same workload once more with every fp32 product exact (bf16x6), after (outside) the headline's timed region
    exact = None
    if extras and gen_nprod == 2 and not a.no_exact_check:
        try:
            pe_first = MiSession(voice, device_id=local_rank, gen_precision="bf16x6")
            ke = max(3, a.steps // 2)
            dte, ne, pe, _ = measure(pe_first, a.preset, ke, max(2, a.warmup), a.parts, a.lockstep, 1234 + rank)
            exact = {"gen_precision": "bf16x6", "gen_nprod": int(pe_first.hparam("gen_nprod")), "value": ne / dte,
                     "unit": "samples/s", "steps": ke, "ms_per_step": dte / ke * 1e3,
                     "note": "six bf16 plane products per fp32 product (each exact to 2^-24, fp32 range); same batch and "
                             "handle count, measured after the headline run on the already power-limited chip"}
            pe.close()
        except Exception as e:  # noqa: BLE001
            exact = {"gen_precision": "bf16x6", "value": None, "note": f"failed: {e}"}


    # ---------------------------------------------------------------- the reference's own call shape: ONE utterance per call
    # (voice.py:350-351; sentences one after the other, :265-269) - wall clock per session.run-shaped call (host ids in, host
    # waveform out), and the time to the first chunk of the streaming form
    def b1_of(s_one, preset):
        g = torch.Generator(device="cpu").manual_seed(4321)
        ids1 = torch.randint(0, 256, (1, a.tokens), generator=g, dtype=torch.int64).numpy()
        lens1 = np.full((1,), a.tokens, np.int64)
        sc = np.array([0.667, LENGTH_SCALE[preset], 0.8], np.float32)
        for _ in range(3):
            s_one.synthesize_batch(ids1, lens1, sc)
        per, n_s = [], 0
        for _ in range(20):
            t0 = time.perf_counter()
            r1 = s_one.synthesize_batch(ids1, lens1, sc)
            per.append((time.perf_counter() - t0) * 1e3)
            n_s = int(r1["y_lengths"].sum()) * hop_of(s_one)
        launches = int(s_one.stats()["total_launches"])  # (of the last call: the counters restart with every run)
        first = []
        for _ in range(5):
            t0 = time.perf_counter()
            it = s_one.synthesize_stream(ids1, lens1, sc, chunk_frames=64)
            next(it)
            first.append((time.perf_counter() - t0) * 1e3)
            for _c in it:
                pass
        return {"preset": preset, "ms_per_call": pctl(per), "samples": n_s, "rtf": float(np.median(per)) * 1e-3 / (n_s / 22050.0),
                "first_chunk_ms": pctl(first), "chunk_frames": 64, "launches_per_call": launches,
                "note": "MiSession.synthesize_batch on one 256-id utterance: wall clock of the whole call, host arrays in and out; "
                        "first_chunk_ms = synthesize_stream until its first 64-frame chunk is in host memory"}

    b1 = None
    if extras and a.batch == 32 and not a.total_batch and a.speakers <= 1:
        try:
            b1 = {a.preset: b1_of(sess, a.preset)}
        except Exception as e:  # noqa: BLE001
            b1 = {"error": f"{type(e).__name__}: {e}"}

    # The headline voice's handles are done: free their ~60 GB of workspace before another voice is measured (with them
    # open, the second voice's freshly allocated workspace measured 25-30 % slower on its HBM-bound kernels - 404 vs 523 M
    # samples/s, tools/exp_order.py - and recovered the moment they were closed)
    n_parts = len(pipe.parts)
    gen_sx = bool(sess.hparam("gen_sx"))
    pipe.close()

    # "headline = high; always also report medium"
    also = None
    if extras and a.preset != "medium" and a.speakers <= 1:
        try:
            mfirst = MiSession(voice_path("medium"), device_id=local_rank)
            km = max(5, a.steps)
            dtm, nm, mp_, minputs = measure(mfirst, "medium", km, max(2, a.warmup), a.also_parts, a.lockstep, 1234 + rank)
            mroof, mstage, mrange = (None, None, None) if a.no_roofline else roofline_of(mfirst, "medium", minputs, 3)
            also = {"preset": "medium", "value": nm / dtm, "unit": "samples/s", "steps": km, "ms_per_step": dtm / km * 1e3,
                    "pipeline_parts": len(mp_.parts),
                    "frames_per_id": nm / km / hop_of(mfirst) / (B * T), "roofline": mroof, "stages": mstage,
                    "f16_range": mrange,
                    "note": "measured in the same process after the headline voice's handles were closed"}
            mp_.close(close_first=False)
            if b1 is not None and "error" not in b1:
                try:
                    b1["medium"] = b1_of(mfirst, "medium")
                except Exception as e:  # noqa: BLE001
                    b1["medium"] = {"error": f"{type(e).__name__}: {e}"}
            mfirst.close()
        except Exception as e:  # noqa: BLE001
            also = {"preset": "medium", "value": None, "note": f"failed: {type(e).__name__}: {e}"}

    # ---------------------------------------------------------------- BASELINE config 4: multi-speaker voice (speaker-embedding
    # path), batch 64 of mixed lengths with the padding mask in play, reduced-precision ("f16") vocoder - next to the same
    # batch in the default fp32-grade arithmetic
    config4 = None
    if extras and a.batch == 32 and not a.total_batch and a.speakers <= 1:
        try:
            c4 = {}
            g4 = torch.Generator(device="cpu").manual_seed(1234)
            B4 = 64
            ids4 = torch.randint(0, 256, (B4, a.tokens), generator=g4, dtype=torch.int64)
            lens4 = torch.randint(max(1, a.tokens // 4), a.tokens + 1, (B4,), generator=torch.Generator(device="cpu").manual_seed(2469),
                                  dtype=torch.int64)
            lens4[0] = a.tokens
            ids4 = ids4 * (torch.arange(a.tokens)[None, :] < lens4[:, None])
            sid4 = torch.randint(0, 4, (B4,), generator=torch.Generator(device="cpu").manual_seed(77), dtype=torch.int64)
            for prec in ("f16", "f16x3"):
                f4 = MiSession(voice_path("medium", 4), device_id=local_rank, gen_precision=prec)
                k4 = max(5, a.steps)
                dt4, n4, p4, _ = measure(f4, "medium", k4, max(2, a.warmup), a.also_parts, a.lockstep, 1234, (ids4, lens4, sid4))
                c4[prec] = {"value": n4 / dt4, "unit": "samples/s", "ms_per_step": dt4 / k4 * 1e3, "steps": k4,
                            "gen_nprod": int(f4.hparam("gen_nprod")), "pipeline_parts": len(p4.parts)}
                p4.close()
            config4 = {"workload": f"VITS full pipeline, preset=medium, 4 speakers (sid per utterance), batch=64 x {a.tokens} phoneme ids "
                                   f"(lengths uniform in [T/4, T], zero-padded, padding mask), scales=[0.667,{LENGTH_SCALE['medium']:.2f},0.8]",
                       "dtype": DTYPE[1], "value": c4["f16"]["value"], "unit": "samples/s", "ms_per_step": c4["f16"]["ms_per_step"],
                       "f16x3_same_batch": c4["f16x3"], "speedup_over_f16x3": c4["f16"]["value"] / c4["f16x3"]["value"],
                       "accuracy": "waveform within 1e-2 max-abs and >= 35 dB SNR of the fp32 oracle (tests/test_gpu_fullsize.py: "
                                   "measured 7.8e-4 / 62.6 dB on this voice)",
                       "high_f16": None}
            fh = MiSession(voice, device_id=local_rank, gen_precision="f16")
            kh = max(5, a.steps)
            dth, nh, ph, _ = measure(fh, a.preset, kh, max(2, a.warmup), a.parts, a.lockstep, 1234 + rank)
            config4["high_f16"] = {"preset": a.preset, "value": nh / dth, "unit": "samples/s", "ms_per_step": dth / kh * 1e3,
                                   "speedup_over_headline": (nh / dth) / (samples / dt),
                                   "note": "the headline batch (single speaker, fixed lengths) with the f16 vocoder"}
            ph.close()
        except Exception as e:  # noqa: BLE001
            config4 = {"error": f"{type(e).__name__}: {e}"}

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and a.speakers <= 1:  # (N = 1 only: the other ranks would sit in the final barrier)
        try:
            cpu = cpu_baseline(voice, a.preset, T, scales, 1234, hop)
        except Exception as e:  # noqa: BLE001 - the baseline is a report, never the product
            cpu = {"value": None, "unit": "samples/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}

    def g_ms(d):
        return d.get("ms_per_step") if isinstance(d, dict) else None

    if rank == 0:
        value = samples_all / dt_max
        line = {
            "metric": "audio samples/sec (22.05 kHz), batch-32 256-phoneme utterances",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt_max / a.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if a.total_batch else "weak",
            "vs_baseline": None, "dtype": DTYPE.get(gen_nprod, "f32") if gen_sx else "f32",
            "data": "synthetic", "gen_precision": a.gen_precision,
            "rtf": dt_max / (samples_all / world / 22050.0) if samples_all else None,
            "config": {"workload": f"VITS full pipeline (encoder+duration+flow+HiFi-GAN), preset={a.preset}"
                                   f"{'' if a.speakers <= 1 else f', {a.speakers} speakers (sid per utterance)'}, "
                                   f"batch={f'{a.total_batch} over {world} GPU(s)' if a.total_batch else f'{B}/GPU'} x {a.tokens} phoneme ids"
                                   f"{' (lengths uniform in [T/4, T], zero-padded)' if a.mixed_lengths else ''}, "
                                   f"scales=[0.667,{scales[1]:.2f},0.8], device Philox noise, seeded synthetic weights",
                       "preset": a.preset, "batch_per_gpu": B, "global_batch": a.total_batch or B * world, "tokens": a.tokens, "hop": hop,
                       "pipeline_parts": n_parts, "prewarm_s": a.prewarm_s,
                       "pipeline_host": "lockstep" if (a.lockstep or n_parts == 1) else "one free-running host thread per part",
                       "pipeline_schedule": "lockstep: every pass split over the handles" if (a.lockstep or n_parts == 1) else
                                            ("whole passes dealt to the handles in turn (request-level pipelining)" if a.schedule == "alternate"
                                             else "every pass split over the handles (rows each)"),
                       "samples_per_step": samples_all / a.steps,
                       "frames_per_id": samples_all / a.steps / hop /
                                        (float(lens_all.sum()) if a.total_batch else float(lens_h.sum()) * world),
                       "sharding": shard_info, "ranks": per_rank,
                       "ranks_summary": None if not per_rank else {
                           "nccl_world": world, "nccl_world_equals_n_gpus": world == a.gpus,
                           "samples_per_s_min": min(r["samples_per_s"] for r in per_rank),
                           "samples_per_s_max": max(r["samples_per_s"] for r in per_rank),
                           "samples_per_s_sum": sum(r["samples_per_s"] for r in per_rank),
                           "distinct_devices": len({r["pci_bus"] or r["device"] for r in per_rank}),
                           "broadcast_s_max": max((r["broadcast_s"] or 0.0) for r in per_rank)},
                       "tails": "zero: samples behind an utterance's end are not rendered (MiSession tails=\"zero\", the default); "
                                "padded_rendering = the graph's own padded form of the same batch",
                       "weights": weights, "commit": git_head(), "source_sha": source_sha()},
            "roofline": roofline, "cpu_baseline": cpu, "step_ms": step_pct, "host_io": host_io, "one_handle": one_handle,
            "exact_arithmetic": exact, "padded_rendering": padded, "also": also, "stages": stage, "f16_range": f16_range,
            "b1": b1, "config4": config4, "split_schedule": split_sched,
            "request_latency_ms": {
                "saturated": dt_max / a.steps * 1e3 * (n_parts if (a.schedule == "alternate" and not a.lockstep and n_parts > 1) else 1),
                "unloaded": g_ms(one_handle) if one_handle else dt_max / a.steps * 1e3,
                "note": "ms_per_step is the interval between finished requests.  Under the default schedule (whole requests dealt "
                        "to the handles in turn) n = pipeline_parts requests are in flight at any time, so a request submitted to "
                        "the saturated pipeline takes n x ms_per_step (Little's law); `unloaded` = one request alone on one "
                        "handle (one_handle.ms_per_step)"},
            "value_is": "device-resident ids in, waveform left on the device (the contract's HBM-resident timed region); the "
                        "figure shaped like the reference's session.run (host ids in, one host fp32 [B,1,1,S] array out) is "
                        "host_io.value",
        }
        if cpu and cpu.get("value"):
            line["gpu_over_cpu"] = value / world / cpu["value"]

        # every reported figure once more, short, as the LAST key of the line (the tail of a long line is what survives a
        # bounded stdout capture)
        def g(d, *ks):
            for k in ks:
                d = d.get(k) if isinstance(d, dict) else None
            return d
        line["summary"] = {
            "value": value, "ms_per_step": line["ms_per_step"], "n_gpus": world, "preset": a.preset, "dtype": a.gen_precision,
            "padded_rendering.value": g(padded, "value"), "one_handle.value": g(one_handle, "value"), "host_io.value": g(host_io, "value"),
            "split_schedule.value": g(split_sched, "value"), "request_latency_ms.saturated": g(line, "request_latency_ms", "saturated"),
            "request_latency_ms.unloaded": g(line, "request_latency_ms", "unloaded"),
            "exact_arithmetic.value": g(exact, "value"), "also.preset": g(also, "preset"), "also.value": g(also, "value"),
            "also.ms_per_step": g(also, "ms_per_step"), "also.roofline.bound": g(also, "roofline", "bound"),
            "also.roofline.frac": g(also, "roofline", "frac"), "also.dec_hbm_frac_marks": g(also, "stages", "stage_marks_only", "dec_hbm_frac"),
            "roofline.bound": g(roofline, "bound"), "roofline.frac": g(roofline, "frac"), "roofline.achieved": g(roofline, "achieved"),
            "roofline.peak": g(roofline, "peak"), "roofline.unit": g(roofline, "unit"), "roofline.traffic": g(roofline, "traffic"),
            "stages_marks_ms": {k: g(stage, "stage_marks_only", k) for k in ("enc_ms", "dp_ms", "flow_ms", "dec_ms", "total_ms")},
            "also.stages_marks_ms": {k: g(also, "stages", "stage_marks_only", k) for k in ("enc_ms", "dp_ms", "flow_ms", "dec_ms", "total_ms")},
            "b1_ms": {k: g(b1, k, "ms_per_call", "median") for k in ("high", "medium")},
            "b1_first_chunk_ms": {k: g(b1, k, "first_chunk_ms", "median") for k in ("high", "medium")},
            "config4.value": g(config4, "value"), "config4.f16x3_same_batch": g(config4, "f16x3_same_batch", "value"),
            "config4.high_f16": g(config4, "high_f16", "value"),
            "cpu_baseline.value": g(cpu, "value"), "cpu_baseline.cores": g(cpu, "cores"), "cpu_baseline.kind": g(cpu, "kind"),
            "gpu_over_cpu": line.get("gpu_over_cpu"), "commit": line["config"]["commit"], "source_sha": line["config"].get("source_sha")}
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes a version banner through C stdio: flush it first, so that the JSON line is the last line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(line), flush=True)


def hop_of(s):
    return s.hparam("hop")


def guarded_main():
    """main(), with a rank's failure turned into one {"error", "rank", "stderr_tail"} line on stdout and a non-zero exit (a
    peer that died inside a collective, a device error): the launcher relays it instead of a result."""
    try:
        main()
    except SystemExit:
        raise
    except BaseException as e:  # noqa: BLE001
        from phoonnx_amd.sharding import report_rank_failure
        report_rank_failure(e, int(os.environ.get("RANK", "0")), "bench.py")
        os._exit(1)   # (no destructors: a process group whose peer is gone can block in its own teardown)


if __name__ == "__main__":
    guarded_main()
