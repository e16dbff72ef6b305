"""Multi-GPU path: independent utterances sharded across the GPUs of one node.

VITS inference has no exchange step: utterances share nothing but the
weights.  So the only collective is ONE broadcast of the packed weight arena (RCCL over
xGMI when the backend is "nccl") at load time; in steady state every rank runs its own
sub-batch on its own handle/stream with no inter-GPU dependence.

One process per GPU (torch.distributed); torch is used here for device memory and the
process group only.
"""
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .session import MiSession


def partition(lengths: Sequence[int], world: int) -> Tuple[List[np.ndarray], np.ndarray]:
    """Deal utterances to ranks so that every rank gets the same amount of work: utterances sorted by length
    (longest first) are dealt in snake order (ranks 0..N-1, then N-1..0, ...), which keeps each rank's shard
    length-sorted (little padding inside its batch) and its total length within one utterance of every other
    rank's - a gathered request finishes when the slowest rank does.  Returns (per-rank index arrays into the
    original order, inverse permutation that restores the original order from the concatenation of the per-rank
    results)."""
    lengths = np.asarray(lengths)
    n = len(lengths)
    order = np.argsort(-lengths, kind="stable")
    world = max(int(world), 1)
    pos = np.arange(n)
    lap, slot = pos // world, pos % world
    rank_of = np.where(lap % 2 == 0, slot, world - 1 - slot)
    shards = [order[rank_of == r] for r in range(world)]
    inv = np.empty(n, dtype=np.int64)
    inv[np.concatenate(shards) if n else np.zeros(0, np.int64)] = np.arange(n)
    return shards, inv


def pad_batch(utts: Sequence[Sequence[int]], pad_id: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """ids int64 [B, Tmax] zero-padded + lengths int64 [B] (the feed layout of voice.py:350-351, batched)."""
    lens = np.asarray([len(u) for u in utts], np.int64)
    T = int(lens.max()) if len(utts) else 0
    ids = np.full((len(utts), T), pad_id, np.int64)
    for i, u in enumerate(utts):
        ids[i, :len(u)] = np.asarray(u, np.int64)
    return ids, lens


def arena_checksum(arena) -> int:
    """Order-independent 64-bit checksum of a packed arena (torch uint8 tensor, host or device): the wrapping sum of
    its little-endian 32-bit words.  Cheap enough to run on the GPU right after the broadcast."""
    import torch
    n = arena.numel() // 4 * 4
    return int(arena[:n].view(torch.int32).sum(dtype=torch.int64).item())


def broadcast_arena(path: str, dist, device: Optional[int], src: int = 0, verify: bool = True, stats: Optional[dict] = None):
    """Rank `src` parses + packs the .onnx on the host; the packed arena is broadcast to every
    rank (device tensors over RCCL when `device` is not None, host tensors otherwise, e.g. gloo).
    verify: every rank compares the checksum of what it received with rank `src`'s (a second, 8-byte broadcast) and
    raises on a mismatch.  Returns a uint8 torch tensor holding the arena on this rank.
    stats (optional dict) receives pack_s (rank `src`: parse + pack), bcast_s (header + arena broadcast, synchronised),
    bytes, checksum."""
    import time
    import torch
    rank = dist.get_rank()
    t0 = time.perf_counter()
    dev = torch.device("cuda", device) if device is not None else torch.device("cpu")
    hdr = torch.zeros(2, dtype=torch.int64, device=dev)  # [bytes, checksum]
    host = None
    if rank == src:
        host = MiSession(path, host_only=True)
        arena = torch.from_numpy(np.array(host.arena_host(), copy=True))
        hdr[0] = host.arena_bytes()
        hdr[1] = arena_checksum(arena)
        arena = arena.to(dev)
        host.close()
    t1 = time.perf_counter()
    dist.broadcast(hdr, src)
    nbytes, want = int(hdr[0].item()), int(hdr[1].item())
    if rank != src:
        arena = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    dist.broadcast(arena, src)  # ~64-119 MB once; 7 xGMI links x ~153 GB/s -> sub-millisecond class
    got = arena_checksum(arena) if verify else None  # (.item(): also waits for the broadcast on a device tensor)
    if device is not None:
        torch.cuda.synchronize(dev)
    t2 = time.perf_counter()
    if stats is not None:
        stats.update(pack_s=t1 - t0, bcast_s=t2 - t1, bytes=nbytes, checksum=want, verified=bool(verify))
    if verify and got != want:
        raise RuntimeError(f"rank {rank}: weight arena checksum {got:#x} differs from rank {src}'s {want:#x}")
    return arena


def open_sharded(path: str, device_id: int, dist=None, src: int = 0, force_broadcast: bool = False, verify: bool = True,
                 stats: Optional[dict] = None):
    """Open one engine handle per rank.  With a process group, weights arrive by broadcast and the handle adopts the
    device arena (vits_open_with_arena: the file is parsed for the model description and the arena LAYOUT only - no
    rank but `src` packs a weight); without one this is a plain open.  force_broadcast takes the broadcast path even at
    world size 1 (exercises the N > 1 code on a one-GPU box).
    Returns (session, arena_tensor_or_None) - keep the tensor alive as long as the session.  stats: as
    broadcast_arena, plus open_s (layout-only open on the adopted arena)."""
    import time
    if dist is None or (dist.get_world_size() == 1 and not force_broadcast):
        t0 = time.perf_counter()
        sess = MiSession(path, device_id=device_id)
        if stats is not None:
            stats.update(open_s=time.perf_counter() - t0)
        return sess, None
    arena = broadcast_arena(path, dist, device_id, src, verify, stats)
    # (every rank resolves the arithmetic the same way rank `src` did when it packed: VITSMI_GEN_PRECISION / default)
    t0 = time.perf_counter()
    sess = MiSession(path, device_id=device_id, arena_device_ptr=arena.data_ptr(), arena_bytes=arena.numel())
    if stats is not None:
        stats.update(open_s=time.perf_counter() - t0)
    return sess, arena


class _DeviceArray:
    """A device buffer the engine owns, described to torch through the CUDA array interface (zero copy)."""

    def __init__(self, ptr: int, shape, typestr: str = "<f4"):
        self.__cuda_array_interface__ = {"shape": tuple(int(x) for x in shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2}


def report_rank_failure(exc: BaseException, rank: int, what: str = "") -> str:
    """One machine-readable line for a rank that cannot go on (a peer died inside a collective, a device error, ...):
    {"error": ..., "rank": ..., "what": ..., "stderr_tail": traceback}.  Printed to stdout and returned; the caller exits
    non-zero.  A front end (bench.py launch_ranks, a serving supervisor) relays it instead of waiting for a result."""
    import json
    import sys
    import traceback
    tb = "".join(traceback.format_exception(type(exc), exc, exc.__traceback__))
    line = json.dumps({"error": f"{type(exc).__name__}: {exc}", "rank": int(rank), "what": what, "stderr_tail": tb[-1500:]})
    print(line, flush=True)
    sys.stderr.write(tb)
    return line


class ShardedSynthesizer:
    """Batched-utterance front: `synthesize(utterances)` runs this rank's shard and (optionally)
    gathers the waveforms of all ranks on the host in the original order."""

    def __init__(self, path: str, device_id: int, dist=None, force_broadcast: bool = False, session=None):
        """session: an already opened engine session for this rank (anything with synthesize_batch(ids, lens, scales,
        sid) -> {"output", "y_lengths"} and hparam("hop")); default: open_sharded(path, device_id, dist)."""
        self.dist = dist
        self.rank = dist.get_rank() if dist else 0
        self.world = dist.get_world_size() if dist else 1
        self.open_stats = {}
        if session is not None:
            self.session, self._arena = session, None
        else:
            self.session, self._arena = open_sharded(path, device_id, dist, force_broadcast=force_broadcast,
                                                     stats=self.open_stats)
        self.hop = self.session.hparam("hop")

    def synthesize(self, utterances: Sequence[Sequence[int]], scales, sids: Optional[Sequence[int]] = None,
                   gather=False, dst: int = 0):
        """This rank's shard of the request, rendered.
        gather=False: [(original index, waveform)] of this rank's utterances - results stay where they were made.
        gather=True:  every rank returns every utterance's waveform in the original order.
        gather="root": only rank `dst` does (the others return None) - what a front end that answers one client wants.
        The gathers move tensors, not pickles: one small all_gather of the sample counts, then ONE collective over flat
        fp32 buffers (every rank's valid samples back to back, padded to the longest rank) - on the device over RCCL/xGMI
        when the backend is nccl, on the host with gloo."""
        shards, inv = partition([len(u) for u in utterances], self.world)
        mine = shards[self.rank]
        scales = np.asarray(scales, np.float32)
        # the device path speaks MiSession.run_device's result ({"data_ptr", "dims", "y_lengths_ptr"} of ONE handle): an
        # injected session (the documented contract is synthesize_batch + hparam only - PipelinedSession.run_device, for one,
        # returns a list of parts) takes the host path
        use_dev = bool(gather) and self.dist is not None and self.dist.get_backend() == "nccl" and isinstance(self.session, MiSession)
        if use_dev:
            return self._gather_device(utterances, shards, mine, scales, sids, gather, dst)
        local, failure = [], None
        try:
            if len(mine):
                ids, lens = pad_batch([utterances[i] for i in mine])
                sid = None if sids is None else np.asarray([sids[i] for i in mine], np.int64)
                r = self.session.synthesize_batch(ids, lens, scales, sid)
                for b in range(len(mine)):
                    n = int(r["y_lengths"][b]) * self.hop
                    local.append(r["output"][b, 0, 0, :n])
        except Exception as exc:  # noqa: BLE001 - re-raised below, after the peers have been told
            if not gather or self.dist is None:
                raise
            failure = exc
        if not gather or self.dist is None:
            return [(int(i), w.copy()) for i, w in zip(mine, local)]
        import torch
        dist = self.dist
        dev = torch.device("cpu")
        self._agree(failure, dev)
        # 1. sample counts of every utterance of every rank (shards differ by at most one utterance: pad with zeros)
        rows = max(len(sh) for sh in shards)
        cnt = torch.zeros(rows, dtype=torch.int64)
        cnt[:len(local)] = torch.tensor([len(w) for w in local], dtype=torch.int64)
        allcnt = torch.zeros(self.world * rows, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(allcnt, cnt.to(dev))
        allcnt = allcnt.cpu().view(self.world, rows)
        width = int(allcnt.sum(1).max())
        # 2. the waveforms: this rank's valid samples back to back in one flat buffer of the common width
        flat = torch.zeros(max(width, 1), dtype=torch.float32)
        if local:
            flat[:int(cnt.sum())] = torch.from_numpy(np.concatenate(local))
        return self._collect(flat, allcnt, shards, len(utterances), gather, dst)

    def _gather_device(self, utterances, shards, mine, scales, sids, gather, dst):
        """nccl backend: the shard is rendered device to device (vits_run_device), its valid samples are packed back to back
        on the GPU straight out of the engine's output buffer, and that buffer goes into the collective - the waveform crosses
        PCIe once, on its way to whoever asked for it (the host path costs a D2H, an H2D and another D2H per rank)."""
        import torch
        dist = self.dist
        dev = torch.device("cuda", torch.cuda.current_device())
        rows = max(len(sh) for sh in shards)
        cnt = torch.zeros(rows, dtype=torch.int64, device=dev)
        out = ylen = None
        failure = None
        try:
            if len(mine):
                ids, lens = pad_batch([utterances[i] for i in mine])
                sid = None if sids is None else np.asarray([sids[i] for i in mine], np.int64)
                # vits_run_device takes device pointers as they are (vits_run's host-side checks do not see them, and the
                # embedding kernel zeroes an out-of-vocabulary id rather than fault): the request is checked HERE, on the host
                self._validate(ids, lens, sid)
                d_ids, d_lens = torch.from_numpy(ids).to(dev), torch.from_numpy(lens).to(dev)
                d_sid = None if sid is None else torch.from_numpy(sid).to(dev)
                torch.cuda.synchronize(dev)   # (the engine runs on its own stream: the inputs are in place before it starts)
                r = self.session.run_device(d_ids.data_ptr(), d_lens.data_ptr(), len(mine), ids.shape[1], scales,
                                            None if d_sid is None else d_sid.data_ptr())
                self.session.sync()           # ... and its output is complete before torch's stream reads it
                B, S = int(r["dims"][0]), int(r["dims"][3])
                out = torch.as_tensor(_DeviceArray(r["data_ptr"], (B, S)), device=dev)
                ylen = torch.as_tensor(_DeviceArray(r["y_lengths_ptr"], (B,), "<i8"), device=dev)
                cnt[:B] = ylen * self.hop
        except Exception as exc:  # noqa: BLE001 - re-raised by _agree, after the peers have been told
            failure = exc
        self._agree(failure, dev)
        allcnt = torch.zeros(self.world * rows, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(allcnt, cnt)
        allcnt = allcnt.cpu().view(self.world, rows)
        width = int(allcnt.sum(1).max())
        flat = torch.zeros(max(width, 1), dtype=torch.float32, device=dev)
        if out is not None:
            mask = torch.arange(out.shape[1], device=dev)[None, :] < cnt[:out.shape[0], None]
            flat[:int(allcnt[self.rank].sum())] = out[mask]   # row-major: every row's valid prefix, rows in shard order
        return self._collect(flat, allcnt, shards, len(utterances), gather, dst)

    def _validate(self, ids, lens, sid):
        """The checks vits_run makes on host inputs (vitsmi.hip stage_inputs), for the path that hands the engine device
        pointers: ids inside the vocabulary, lengths inside the row, speaker ids inside the table."""
        n_vocab = int(self.session.hparam("n_vocab"))
        if ids.size and (int(ids.min()) < 0 or int(ids.max()) >= n_vocab):
            bad = np.argwhere((ids < 0) | (ids >= n_vocab))[0]
            raise ValueError(f"phoneme id {int(ids[tuple(bad)])} at [{int(bad[0])},{int(bad[1])}] is out of range [0,{n_vocab})")
        if lens.size and (int(lens.min()) < 0 or int(lens.max()) > ids.shape[1]):
            raise ValueError(f"input_lengths outside [0,{ids.shape[1]}]")
        if int(self.session.hparam("gin")):
            if sid is None:
                raise ValueError("Missing speaker id")
            n_spk = int(self.session.hparam("n_speakers"))
            if sid.size and (int(sid.min()) < 0 or int(sid.max()) >= n_spk):
                raise ValueError(f"sid out of range [0,{n_spk})")

    def _agree(self, failure, dev):
        """A gathered request ends in a collective every rank must enter.  A rank whose shard failed (a bad id, a device
        error, a range violation) would leave its peers waiting in it: one 4-byte all_reduce(MIN) of an ok flag first, and
        EVERY rank raises - the failing one its own exception, the others a RuntimeError naming the cause as 'a peer'."""
        import torch
        ok = torch.tensor([0 if failure is not None else 1], dtype=torch.int32, device=dev)
        self.dist.all_reduce(ok, op=torch.distributed.ReduceOp.MIN)
        if failure is not None:
            raise failure
        if int(ok.item()) == 0:
            raise RuntimeError(f"rank {self.rank}: a peer's shard of this request failed; nothing was gathered")

    def _collect(self, flat, allcnt, shards, n_utts, gather, dst):
        """ONE collective over the ranks' flat buffers (device tensors over RCCL, host tensors over gloo), then the rows are
        cut apart on the host and put back into the request's order."""
        import torch
        dist = self.dist
        root_only = gather == "root"
        if root_only:
            parts = [torch.empty_like(flat) for _ in range(self.world)] if self.rank == dst else None
            dist.gather(flat, parts, dst=dst)
            if self.rank != dst:
                return None
            everything = torch.stack(parts).cpu().numpy()
        else:
            out = torch.empty(self.world * flat.numel(), dtype=torch.float32, device=flat.device)
            dist.all_gather_into_tensor(out, flat)
            everything = out.cpu().numpy().reshape(self.world, flat.numel())
        # 3. cut the rows apart and restore the request's order
        res = [None] * n_utts
        for rk, sh in enumerate(shards):
            off = 0
            for j, i in enumerate(sh):
                n = int(allcnt[rk, j])
                res[int(i)] = everything[rk, off:off + n].copy()
                off += n
        return res

    def close(self):
        self.session.close()
