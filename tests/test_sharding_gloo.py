"""N > 1 path on CPU: two processes, gloo backend.  Covers what can be checked without a GPU:
the utterance partition / restore-order logic and the weight-arena broadcast (rank 0 parses and
packs the .onnx, every rank ends up with the same bytes a local pack would produce)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

from phoonnx_amd.sharding import pad_batch, partition


def test_partition_covers_everything_and_restores_order():
    rng = np.random.default_rng(0)
    for world in (1, 2, 3, 8):
        for n in (0, 1, 5, 8, 37, 256):
            lengths = rng.integers(1, 300, n)
            shards, inv = partition(lengths, world)
            assert len(shards) == world
            flat = np.concatenate(shards) if n else np.zeros(0, np.int64)
            assert sorted(flat.tolist()) == list(range(n))              # every utterance exactly once
            assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
            for sh in shards:                                            # each shard stays length-sorted: little padding
                assert np.all(np.diff(lengths[sh]) <= 0)
            # balanced work: no rank carries more than the lightest one plus the longest utterance (snake dealing);
            # the contiguous-block deal this replaces gave rank 0 all the longest ones
            if n >= world:
                loads = [int(lengths[sh].sum()) for sh in shards]
                assert max(loads) - min(loads) <= int(lengths.max()), (world, n, loads)
            results = [f"utt{i}" for i in flat]                          # "concatenated per-rank results"
            assert [results[inv[i]] for i in range(n)] == [f"utt{i}" for i in range(n)]


def test_pad_batch_layout():
    ids, lens = pad_batch([[5, 6, 7], [1], [2, 3]])
    assert ids.dtype == np.int64 and lens.dtype == np.int64
    assert ids.tolist() == [[5, 6, 7], [1, 0, 0], [2, 3, 0]] and lens.tolist() == [3, 1, 2]


def _worker(rank, world, port, onnx_path, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from phoonnx_amd import MiSession
    from phoonnx_amd.sharding import broadcast_arena, partition as part
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        arena = broadcast_arena(onnx_path, dist, device=None, src=0)           # (verifies rank 0's checksum itself)
        local = MiSession(onnx_path, host_only=True)
        same = bool(np.array_equal(arena.numpy(), np.asarray(local.arena_host())))
        from phoonnx_amd.sharding import arena_checksum
        import torch
        same = same and arena_checksum(arena) == arena_checksum(torch.from_numpy(np.array(local.arena_host(), copy=True)))
        # what a receiving rank opens with: the layout alone must describe exactly these bytes
        lay = MiSession(onnx_path, layout_only=True)
        same = same and lay.arena_bytes() == arena.numel()
        lay.close()
        # each rank synthesises only its shard; shards are disjoint and cover the request
        shards, _ = part([9, 3, 7, 1, 5], world)
        objs = [None] * world
        dist.all_gather_object(objs, shards[rank].tolist())
        q.put((rank, same, int(arena.numel()), objs))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("voice", ["fixture", "split_exact"])
def test_arena_broadcast_two_ranks_gloo(voice, tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    if voice == "fixture":
        path = os.path.join(GOLDEN, "tiny_rb2_ms.onnx")
    else:  # channel counts that put the generator and the flow on the split-exact engine (bf16-plane weights)
        from phoonnx_amd import MiSession
        from phoonnx_amd.synth import write_voice
        path = str(tmp_path / "sx_voice.onnx")
        write_voice(path, "small", seed=3, upsample_initial_channel=128, upsample_rates=(8, 4),
                    upsample_kernel_sizes=(16, 8))
        probe = MiSession(path, host_only=True)
        assert probe.hparam("gen_sx") == 1
        probe.close()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, path, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort()
    assert [g[0] for g in got] == [0, 1]
    assert all(g[1] for g in got), "broadcast arena differs from a local pack"
    assert got[0][2] == got[1][2] > 0
    assert sorted(got[0][3][0] + got[0][3][1]) == [0, 1, 2, 3, 4] and got[0][3] == got[1][3]


class _StubSession:
    """Stands in for MiSession on a CPU box: a deterministic "waveform" per utterance that depends on its ids only
    (so any rank renders the same thing for the same utterance), frame count = token count."""
    HOP = 4

    def hparam(self, key):
        assert key == "hop"
        return self.HOP

    def synthesize_batch(self, ids, lens, scales, sid=None):
        B, T = ids.shape
        S = int(lens.max()) * self.HOP
        out = np.zeros((B, 1, 1, S), np.float32)
        for b in range(B):
            n = int(lens[b]) * self.HOP
            seed = int(ids[b, :lens[b]].sum()) + (0 if sid is None else 1000 * int(sid[b]))
            out[b, 0, 0, :n] = np.sin(np.arange(n, dtype=np.float32) * 0.01 * (seed % 97 + 1)) * float(scales[1])
        return {"output": out, "y_lengths": lens.astype(np.int64)}

    def close(self):
        pass


def _request(seed=5, n=64):
    rng = np.random.default_rng(seed)
    utts = [rng.integers(1, 50, size=int(k)).tolist() for k in rng.integers(1, 40, size=n)]
    sids = rng.integers(0, 4, size=n).tolist()
    return utts, sids


def _gather_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from phoonnx_amd.sharding import ShardedSynthesizer
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sh = ShardedSynthesizer("unused.onnx", 0, dist, session=_StubSession())
        utts, sids = _request()
        scales = np.array([0.667, 1.5, 0.8], np.float32)
        mine = sh.synthesize(utts, scales, sids)                 # this rank's shard: [(original index, waveform)]
        everything = sh.synthesize(utts, scales, sids, gather=True)
        at_root = sh.synthesize(utts, scales, sids, gather="root", dst=1)
        assert (at_root is None) == (rank != 1)
        if at_root is not None:
            assert all(a.tobytes() == b.tobytes() for a, b in zip(at_root, everything))
        # a shard that FAILS on one rank (a bad id, a device error ...): both ranks raise - the failing one its own error,
        # its peer a RuntimeError - instead of the peer waiting in the gather for a buffer that never comes
        class Failing(_StubSession):
            def synthesize_batch(self, ids, lens, scales, sid=None):
                if rank == 1:
                    raise ValueError("phoneme id 999 is out of range")
                return super().synthesize_batch(ids, lens, scales, sid)
        bad = ShardedSynthesizer("unused.onnx", 0, dist, session=Failing())
        try:
            bad.synthesize(utts, scales, sids, gather=True)
            verdict = "no error"
        except ValueError as e:
            verdict = "own:" + str(e)
        except RuntimeError as e:
            verdict = "peer:" + str(e)
        assert verdict.startswith("own:phoneme id 999" if rank == 1 else "peer:rank 0: a peer's shard"), verdict
        # ... and the group is still usable afterwards
        again = sh.synthesize(utts, scales, sids, gather=True)
        assert all(a.tobytes() == b.tobytes() for a, b in zip(again, everything))
        q.put((rank, [i for i, _ in mine], [w.tobytes() for w in everything]))
    finally:
        dist.destroy_process_group()


def test_sharded_synthesize_gathers_in_the_original_order_two_ranks_gloo():
    """ShardedSynthesizer.synthesize(gather=True) end to end over a real process group (gloo, two
    ranks, stub engine), a 64-utterance mixed-length request: snake partition -> per-rank padded batch -> sample counts by
    all_gather_into_tensor -> flat fp32 buffers by all_gather_into_tensor (no pickle) -> restore order.  Every rank must
    hold every utterance's waveform at its ORIGINAL index, equal to what one process renders."""
    import torch.multiprocessing as mp
    from phoonnx_amd.sharding import ShardedSynthesizer
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    utts, sids = _request()
    solo = ShardedSynthesizer("unused.onnx", 0, None, session=_StubSession())
    want = [w for _, w in sorted(solo.synthesize(utts, np.array([0.667, 1.5, 0.8], np.float32), sids))]
    assert sorted(got[0][1] + got[1][1]) == list(range(len(utts)))       # shards: disjoint, complete
    assert abs(len(got[0][1]) - len(got[1][1])) <= 1
    for rank, _, everything in got:
        assert len(everything) == len(utts)
        for i, w in enumerate(everything):
            assert len(w) == 4 * len(utts[i]) * _StubSession.HOP, (rank, i)
            assert w == want[i].tobytes(), (rank, i)      # byte-equal to the single-rank rendering


def test_arena_checksum_detects_corruption():
    import torch
    from phoonnx_amd import MiSession
    from phoonnx_amd.sharding import arena_checksum
    s = MiSession(os.path.join(GOLDEN, "tiny_dp.onnx"), host_only=True)
    a = torch.from_numpy(np.array(s.arena_host(), copy=True))
    s.close()
    c0 = arena_checksum(a)
    b = a.clone()
    b[12345] ^= 0x40
    assert arena_checksum(b) != c0 and arena_checksum(a.clone()) == c0


def test_bench_gpus_flag_launches_that_many_ranks():
    """`python bench.py --gpus N` used to measure one GPU whatever N was.  Without a GPU the ranks cannot run,
    but the launcher can: --gpus 2 started by hand must spawn two torch.distributed ranks (each of which then refuses to
    run without an MI355X) and exit non-zero; under a launcher whose WORLD_SIZE disagrees with --gpus it must refuse."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    # (the elastic agent terminates the surviving rank as soon as the first one has failed: on a loaded box the second rank can
    # be killed before it reaches its own device check, so the observation is retried - it only has to be made once)
    for attempt in range(4):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                           capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode != 0
        assert '"metric"' not in r.stdout
        # ... and the launcher says so in ONE machine-readable line instead of printing nothing
        errs = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"error"' in ln]
        assert len(errs) == 1 and "stderr_tail" in errs[0], r.stdout[-800:]
        if r.stderr.count("bench.py needs an MI355X") >= 2:
            break
    assert r.stderr.count("bench.py needs an MI355X") >= 2, r.stderr[-1500:]      # both ranks got as far as the device check
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True,
                       timeout=120, env=env2)
    assert r.returncode != 0 and "--gpus 4 but the launcher started 1 rank" in (r.stderr + r.stdout)


_KILL_WORKER = r'''
import os, sys
sys.path.insert(0, {root!r})
sys.path.insert(0, os.path.join({root!r}, "tests"))
import datetime
import numpy as np
import torch.distributed as dist
from phoonnx_amd.sharding import ShardedSynthesizer, report_rank_failure
from test_sharding_gloo import _StubSession, _request
rank = int(os.environ["RANK"])
dist.init_process_group("gloo", rank=rank, world_size=2, timeout=datetime.timedelta(seconds=20))
sh = ShardedSynthesizer("unused.onnx", 0, dist, session=_StubSession())
utts, sids = _request()
scales = np.array([0.667, 1.5, 0.8], np.float32)
sh.synthesize(utts, scales, sids, gather=True)      # one healthy request
if rank == 1:
    os._exit(7)                                     # ... then this rank dies, mid-service, without a word
try:
    sh.synthesize(utts, scales, sids, gather=True)  # the survivor's next gather cannot complete
    print("UNEXPECTED: the gather returned", flush=True)
    os._exit(0)
except BaseException as e:
    report_rank_failure(e, rank, "synthesize(gather=True)")
    os._exit(3)
'''


def test_a_rank_that_dies_mid_request_makes_the_survivor_exit_with_an_error_line(tmp_path):
    """two gloo ranks serve one gathered request, then rank 1 is gone (os._exit) while rank 0 enters the
    next gather.  Rank 0 must neither hang nor return garbage: the collective fails (connection closed, or the process
    group's timeout), phoonnx_amd.sharding.report_rank_failure prints ONE {"error", "rank", "stderr_tail"} line on stdout
    and the process exits non-zero - what bench.py's launcher relays and a serving supervisor acts on."""
    import json
    import subprocess
    import time
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "kill_worker.py"
    script.write_text(_KILL_WORKER.format(root=ROOT))
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    t0 = time.time()
    out0, err0 = procs[0].communicate(timeout=240)
    procs[1].communicate(timeout=60)
    assert procs[1].returncode == 7
    assert procs[0].returncode == 3, (procs[0].returncode, out0[-500:], err0[-1500:])
    assert time.time() - t0 < 200
    lines = [ln for ln in out0.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and "UNEXPECTED" not in out0
    rec = json.loads(lines[0])
    assert rec["rank"] == 0 and rec["error"] and rec["what"] == "synthesize(gather=True)" and "Traceback" in rec["stderr_tail"]
