"""MiSession — the object that stands where `onnxruntime.InferenceSession` stands in
phoonnx (`phoonnx/voice.py:107,167-171,347,374-377`).

It duck-types exactly the three calls the reference makes:
    MiSession(path, sess_options=..., providers=...)      voice.py:167-171
    session.get_inputs() -> objects with .name            voice.py:347
    session.run(None, feed)[0] -> float32 [B,1,1,S]       voice.py:374-377
and forwards them over the C ABI (include/vitsmi.h) to the gfx950 engine.  There is no CPU
fallback: without libvitsmi.so + an MI355X this raises.
"""
import ctypes as C
import time
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _ffi


class SessionError(RuntimeError):
    """Raised for every failure of the engine (onnxruntime raises its own
    InvalidArgument/RuntimeException at voice.py:374; callers there do not catch either)."""


class RangeError(SessionError):
    """VITS_E_RANGE: the f16x3 generator arithmetic met an activation beyond the range of its fp16 operand planes
    (|x| > 65504) or a non-finite value.  The engine reports this instead of returning clamped audio; `MiSession`
    (range_fallback=True, the default) then reopens the voice with the bf16x6 arithmetic (fp32 range) and repeats the
    call, so user code only sees this when the fallback is switched off or impossible (borrowed weight arena)."""


class _PinnedPool:
    """Page-locked host blocks handed out as NumPy arrays (the DMA engine writes a result straight into the array the
    caller receives: no staging copy, no second copy into a fresh array).  A block goes back to the pool when the last
    array viewing it has been garbage-collected; at most `keep_bytes` of idle blocks are retained.

    The finalizer takes no lock: it may run inside ANY allocation of ANY thread (a cyclic-GC pass), including one made by
    array() while it holds the pool's lock.  It only appends to a deque (atomic); array() sorts the returned blocks into
    the free lists, under the lock, the next time it runs."""

    def __init__(self, keep_bytes=2 << 30):
        import collections
        import threading
        self._free = {}
        self._idle = 0
        self._keep = keep_bytes
        self._mu = threading.Lock()
        self._returned = collections.deque()   # (ptr, cap) of blocks whose arrays have died; drained by array()
        self._types = {}                       # cap -> ctypes array type (ctypes caches every distinct type forever)

    @staticmethod
    def _cap(nbytes):
        return max(1 << 16, (int(nbytes) + (1 << 20) - 1) >> 20 << 20) if nbytes > (1 << 16) else 1 << 16

    def _give(self, ptr, cap):
        self._returned.append((ptr, cap))  # (lock-free: see the class docstring)

    def _drain(self):
        """(under self._mu) returned blocks -> free lists; what exceeds keep_bytes is released"""
        release = []
        while True:
            try:
                ptr, cap = self._returned.popleft()
            except IndexError:
                break
            if self._idle + cap <= self._keep:
                self._free.setdefault(cap, []).append(ptr)
                self._idle += cap
            else:
                release.append(ptr)
        return release

    def drain(self, keep_bytes=None):
        """Sort the blocks whose arrays have died into the free lists NOW and release what exceeds `keep_bytes` (default: the
        pool's cap; 0 releases every idle block).  array() does this as it goes; a process that rendered one large batch and
        then goes idle, streams, or asks for pageable results calls it through MiSession.close() / _fetch so that the host
        memory does not stay page-locked until the next pooled allocation."""
        with self._mu:
            release = self._drain()
            if keep_bytes is not None:
                for c in sorted(self._free, reverse=True):
                    while self._free[c] and self._idle > keep_bytes:
                        release.append(self._free[c].pop())
                        self._idle -= c
        for q in release:
            _ffi.load().vits_host_free(q)
        return len(release)

    def array(self, shape, dtype=np.float32):
        import weakref
        count = int(np.prod(shape))
        n = count * np.dtype(dtype).itemsize
        cap = self._cap(n)
        ptr = None
        with self._mu:
            release = self._drain()
            # an idle block of this size class, or the smallest one up to twice as large
            for c in sorted([k for k, v in self._free.items() if v and cap <= k <= 2 * cap]):
                ptr, cap = self._free[c].pop(), c
                self._idle -= c
                break
            ctype = self._types.get(cap)
            if ctype is None:
                ctype = self._types[cap] = C.c_char * cap   # one type per size class, not per output length
        for q in release:
            _ffi.load().vits_host_free(q)
        if ptr is None:
            ptr = _ffi.load().vits_host_alloc(cap)
            if not ptr:
                raise SessionError(f"cannot allocate {cap} bytes of pinned host memory")
        buf = ctype.from_address(ptr)
        fin = weakref.finalize(buf, self._give, ptr, cap)
        fin.atexit = False  # (at interpreter exit the driver reclaims it)
        return np.frombuffer(buf, dtype, count=count).reshape(shape)


_POOL = _PinnedPool()


@dataclass
class NodeArg:
    name: str
    type: str
    shape: list


_INPUT_SPECS = {
    "input": ("tensor(int64)", ["batch_size", "phonemes"]),
    "input_lengths": ("tensor(int64)", ["batch_size"]),
    "scales": ("tensor(float)", [3]),
    "sid": ("tensor(int64)", ["batch_size"]),
    "langid": ("tensor(int64)", ["batch_size"]),  # third-party exports (voice.py:369); accepted, validated, unused
}


class ModelMeta:
    def __init__(self, custom):
        self.custom_metadata_map = custom
        self.producer_name = "pytorch"
        self.graph_name = "vits"


class MiSession:
    def __init__(self, path_or_bytes, sess_options=None, providers=None, provider_options=None, device_id: int = 0,
                 arena_device_ptr: Optional[int] = None, arena_bytes: int = 0, host_only: bool = False,
                 gen_precision: Optional[str] = None, range_fallback: bool = True, layout_only: bool = False,
                 pinned_results: bool = True, tails: Optional[str] = None, **kwargs):
        """tails: what a padded batch (B > 1, unequal frame counts) holds behind each utterance's end - None (VITSMI_TAILS or
        the default "zero"), "zero" (those samples are not rendered at all and read 0.0; every valid sample is bit-identical to
        the padded rendering), "reference" (the exported graph's own padded rendering: its generator is not masked,
        models.py:348-368, so the tails are its response to zeros - what onnxruntime returns).  The reference calls the
        model with B = 1 only (voice.py:350-351), where the two are the same thing.
        gen_precision: arithmetic of the generator's convs - None (VITSMI_GEN_PRECISION or the default "f16x3"),
        "f16x3", "bf16x6" (exact products), "f16" (the reduced-precision vocoder of BASELINE config 4: fp16 storage, one
        fp16 product per fp32 product, fp32 accumulation).  range_fallback: on a RangeError of an fp16 arithmetic, reopen
        with "bf16x6" and repeat the call (never silently clamped audio).
        pinned_results: run() / synthesize_batch() return arrays that VIEW page-locked host memory (the DMA engine's
        target; recycled when the array is garbage-collected).  An application that keeps many results alive thereby pins
        that much host RAM: pass False to get ordinary pageable arrays (one extra host copy per call).
        Thread safety: like onnxruntime's session.run, every entry point may be called from several threads; calls on one
        session are serialised by a per-session lock (a run is three C calls - enqueue, frame counts, copy-out - on one
        handle's workspace)."""
        import threading
        self._mu = threading.RLock()
        self.pinned_results = bool(pinned_results)
        if not isinstance(path_or_bytes, (str, bytes)) or isinstance(path_or_bytes, bytes):
            if isinstance(path_or_bytes, bytes):
                raise SessionError("MiSession loads a model from a file path, not from serialized bytes")
            path_or_bytes = str(path_or_bytes)
        self._lib = _ffi.load()
        self._h = C.c_void_p()
        self.path = path_or_bytes
        self.device_id = device_id
        self.host_only = host_only or layout_only
        self.range_fallback = bool(range_fallback) and arena_device_ptr is None
        self._open_args = dict(arena_device_ptr=arena_device_ptr, arena_bytes=arena_bytes, host_only=self.host_only,
                               layout_only=layout_only)
        if tails not in (None, "zero", "reference"):
            raise SessionError(f"tails must be None, 'zero' or 'reference' (got {tails!r})")
        self.tails = tails
        self._seed = 0
        self.range_fallbacks = 0  # times this session reopened itself with bf16x6 after a RangeError (stats())
        self._open(gen_precision)

    def _open(self, gen_precision):
        a = self._open_args
        o = _ffi.VitsOpenOptions()
        o.device_id = self.device_id
        o.gen_precision = gen_precision.encode() if gen_precision else None
        o.arena_dev = a["arena_device_ptr"]
        o.arena_bytes = a["arena_bytes"] if a["arena_device_ptr"] is not None else 0
        o.host_only = 1 if a["host_only"] else 0
        o.layout_only = 1 if a["layout_only"] else 0
        h = C.c_void_p()
        rc = self._lib.vits_open_opts(self.path.encode(), C.byref(o), C.byref(h))
        if rc != 0:
            raise SessionError(f"vits_open({self.path!r}) failed [{rc}]: {_ffi.last_error(None)}")
        self._h = h
        self.gen_precision = gen_precision
        if self.tails is not None:
            self._lib.vits_set_tails(self._h, 1 if self.tails == "reference" else 0)
        n = self._lib.vits_num_inputs(self._h)
        self._input_names = [self._lib.vits_input_name(self._h, i).decode() for i in range(n)]

    def _raise(self, what, rc):
        cls = RangeError if rc == _ffi.VITS_E_RANGE else SessionError
        raise cls(f"{what} failed [{rc}]: {self._err()}")

    # How long a call from ANOTHER thread waits for a chunked run in progress before it gives up (a generator that was
    # neither exhausted nor closed keeps its run - and the session lock - until it is garbage-collected).
    busy_timeout_s = 120.0

    def _locked(self, read_only: bool = False):
        """The per-session lock, aware of chunked runs.  A chunked run holds the lock on its worker thread from the first
        chunk to the last, and the engine holds the handle's own mutex with it.  The CONSUMER of that generator must not
        block on either: its read-only calls (frame counts - valid from the first chunk on - and everything answered from
        host state) pass, anything that needs the handle raises instead of deadlocking.  Other threads wait for the run
        to end, for at most busy_timeout_s."""
        import contextlib
        import threading
        import time

        @contextlib.contextmanager
        def cm():
            owner = getattr(self, "_stream_owner", None)
            if owner is not None and owner == threading.get_ident():
                if read_only:
                    yield
                    return
                raise SessionError("a chunked run is in progress on this session (the synthesize_stream / vocoder_stream "
                                   "generator this thread is consuming): exhaust or close() it before other calls")
            deadline = None
            while not self._mu.acquire(timeout=0.25):
                if getattr(self, "_stream_owner", None) is None:
                    deadline = None      # an ordinary batch call of another thread: wait as long as it takes
                    continue
                now = time.monotonic()
                deadline = deadline or now + float(self.busy_timeout_s)
                if now > deadline:
                    raise SessionError(f"session busy: a chunked run has been in progress for more than "
                                       f"{self.busy_timeout_s:g} s (an unclosed synthesize_stream generator?)")
            try:
                yield
            finally:
                self._mu.release()
        return cm()

    def _fall_back_to_bf16x6(self, exc):
        """After a RangeError: reopen this voice with the exact six-product arithmetic (bf16 planes: fp32 range)."""
        import logging
        if not self.range_fallback or self.gen_precision == "bf16x6":  # (only fp16 planes raise it: encoder, flow, generator)
            raise exc
        logging.getLogger(__name__).warning("%s: %s - reopening with gen_precision='bf16x6' (1.8x slower; "
                                            "stats()['range_fallbacks'] counts these)", self.path, exc)
        self.close()
        self._open("bf16x6")
        self.range_fallbacks += 1

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.vits_close(self._h)
            self._h = C.c_void_p()
            try:
                _POOL.drain()   # dead result arrays of this session: back to the pool, the excess over its cap released
            except Exception:
                pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _err(self):
        return _ffi.last_error(self._h)

    # ------------------------------------------------------------------ onnxruntime surface
    def get_inputs(self) -> List[NodeArg]:
        out = []
        for n in self._input_names:
            t, s = _INPUT_SPECS.get(n, ("tensor(int64)", ["batch_size"]))
            out.append(NodeArg(n, t, list(s)))
        return out

    def get_outputs(self) -> List[NodeArg]:
        return [NodeArg("output", "tensor(float)", ["batch_size", 1, 1, "time"])]

    def get_providers(self):
        return ["MI355XExecutionProvider"]

    def get_modelmeta(self) -> ModelMeta:
        keys = ("model_type", "n_speakers", "n_vocab", "sample_rate", "alphabet", "phoneme_type",
                "phonemizer_model", "phoneme_id_map", "has_espeak", "comment")
        return ModelMeta({k: v for k in keys if (v := self.meta(k)) is not None})

    def run(self, output_names: Optional[Sequence[str]], input_feed: Dict[str, np.ndarray], run_options=None):
        if output_names is not None and list(output_names) != ["output"]:
            raise SessionError(f"unknown output names {output_names!r}; the graph has one output 'output'")
        for k in input_feed:
            if k not in self._input_names:
                raise SessionError(f"Invalid input name: {k}")
        for k in self._input_names:
            if k not in input_feed:
                raise SessionError(f"Required input {k} is missing")
        if "langid" in input_feed:  # declared by the graph, consumed by nothing this engine runs: validate, then ignore
            lg = np.asarray(input_feed["langid"])
            if lg.dtype != np.int64 or lg.shape != (np.asarray(input_feed["input"]).shape[0],):
                raise SessionError("Unexpected input: 'langid' must be int64 of shape [batch_size]")
        scales = input_feed["scales"] if "scales" in self._input_names else np.array([0.667, 1.0, 0.8], np.float32)
        out = self.synthesize_batch(input_feed["input"], input_feed["input_lengths"], scales, input_feed.get("sid"))
        return [out["output"]]

    # ------------------------------------------------------------------ extensions
    def set_seed(self, seed: int):
        """Seed of the device-side Philox stream that replaces the graph's two unseeded
        RandomNormalLike nodes (models.py:111, :718)."""
        self._seed = int(seed)

    def synthesize_batch(self, ids, lens, scales, sid=None, noise_dp=None, noise_z=None, taps=()):
        """One batched run.  Returns {"output": [B,1,1,S] float32, "y_lengths": int64 [B], taps...}.
        noise_dp [B,2,T] / noise_z [B,inter,>=F] inject the graph's noise for parity runs."""
        ids = np.ascontiguousarray(ids)
        lens = np.ascontiguousarray(lens)
        scales = np.ascontiguousarray(scales)
        if ids.dtype != np.int64 or lens.dtype != np.int64:
            raise SessionError("Unexpected input data type: 'input'/'input_lengths' must be tensor(int64)")
        if scales.dtype != np.float32 or scales.shape != (3,):
            raise SessionError("Unexpected input: 'scales' must be float32 of shape [3]")
        if ids.ndim != 2 or lens.ndim != 1 or lens.shape[0] != ids.shape[0]:
            raise SessionError(f"Invalid rank/shape for input: {ids.shape} / input_lengths: {lens.shape}")
        B, T = ids.shape
        if sid is not None:
            sid = np.ascontiguousarray(sid)
            if sid.dtype != np.int64 or sid.shape != (B,):
                raise SessionError("Unexpected input: 'sid' must be int64 of shape [batch_size]")
        noise = _ffi.VitsNoise()
        noise.seed = self._seed
        if noise_dp is not None:
            noise_dp = np.ascontiguousarray(noise_dp, np.float32)
            if noise_dp.shape != (B, 2, T):
                raise SessionError(f"noise_dp must be [B,2,T], got {noise_dp.shape}")
            noise.noise_dp = noise_dp.ctypes.data
        if noise_z is not None:
            noise_z = np.ascontiguousarray(noise_z, np.float32)
            if noise_z.ndim != 3 or noise_z.shape[0] != B or noise_z.shape[1] != self.hparam("inter"):
                raise SessionError(f"noise_z must be [B,inter,F], got {noise_z.shape}")
            noise.noise_z = noise_z.ctypes.data
            noise.noise_z_stride = noise_z.shape[2]
        with self._locked():  # enqueue -> frame counts -> copy-out -> taps all use this handle's one workspace
            try:
                self._begin(ids, lens, scales, sid, noise)
                ylen = self.last_y_lengths()
                S = int(ylen.max()) * self.hparam("hop")
                audio = _POOL.array((B, 1, 1, S)) if self.pinned_results else np.empty((B, 1, 1, S), np.float32)
                self._fetch(audio, 0, B)
            except RangeError as exc:
                self._fall_back_to_bf16x6(exc)
                return self.synthesize_batch(ids, lens, scales, sid, noise_dp, noise_z, taps)
            res = {"output": audio, "y_lengths": ylen}
            for t in taps:
                res[t] = self.tap(t)
            return res

    def _begin(self, ids, lens, scales, sid, noise):
        """vits_run_async: validated host arrays in, the whole path enqueued; frame counts are known on return."""
        B, T = ids.shape
        rc = self._lib.vits_run_async(self._h, _ffi.ptr(ids), _ffi.ptr(lens), B, T, _ffi.ptr(scales), _ffi.ptr(sid),
                                      C.byref(noise))
        if rc != 0:
            self._raise("vits_run", rc)

    def _fetch(self, out, row0, rows):
        """vits_fetch_output: this handle's [rows, S] waveform -> rows [row0, row0 + rows) of `out` [B,1,1,S_out]
        (pinned or pageable host memory), zero-filled past S.  Waits for the run."""
        S_out = out.shape[3]
        dst = out.ctypes.data + row0 * S_out * 4
        rc = self._lib.vits_fetch_output(self._h, C.c_void_p(dst), S_out, rows * S_out)
        if rc != 0:
            self._raise("vits_run", rc)
        if not self.pinned_results:
            _POOL.drain()   # (no pooled allocation will come along to do it)

    def vocoder(self, z, sid=None):
        z = np.ascontiguousarray(z, np.float32)
        B, Cc, F = z.shape
        if Cc != self.hparam("inter"):
            raise SessionError(f"z must have {self.hparam('inter')} channels")
        sid = None if sid is None else np.ascontiguousarray(sid, np.int64)
        out = _ffi.VitsOutput()
        with self._locked():
            rc = self._lib.vits_run_vocoder(self._h, _ffi.ptr(z), B, F, _ffi.ptr(sid), C.byref(out))
            if rc == _ffi.VITS_E_RANGE:
                try:
                    self._raise("vits_run_vocoder", rc)
                except RangeError as exc:
                    self._fall_back_to_bf16x6(exc)
                return self.vocoder(z, sid)
            if rc != 0:
                self._raise("vits_run_vocoder", rc)
            try:
                dims = tuple(out.dims[i] for i in range(4))
                return np.ctypeslib.as_array(out.data, shape=(int(np.prod(dims)),)).reshape(dims).copy()
            finally:
                self._lib.vits_free_output(self._h, C.byref(out))

    # ------------------------------------------------------------------ chunked (streaming) rendering, SURVEY §8 f1
    def _stream(self, start):
        """Run `start(callback)` (a blocking C call) on a worker thread and yield (first_sample, samples [B, n]) as the
        engine hands chunks over: the consumer works on chunk i while chunk i + 1 renders."""
        import queue
        import threading
        if getattr(self, "_stream_owner", None) == threading.get_ident():
            raise SessionError("a chunked run is already in progress on this session in this thread: exhaust or close() "
                               "its generator first")
        q = queue.Queue(maxsize=2)  # the engine renders at most two chunks ahead of the consumer
        stop = threading.Event()

        def put(item):
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.05)
                    return
                except queue.Full:
                    pass

        @_ffi.CHUNK_FN
        def on_chunk(user, samples, B, first, n, total):
            if stop.is_set():
                return 1  # the consumer is gone: end the run (vits_chunk_fn's "stop" return)
            put((int(first), np.ctypeslib.as_array(samples, shape=(B * n,)).reshape(B, n).copy(), int(total)))
            return 1 if stop.is_set() else 0

        def work():
            # the session lock is held for the whole chunked run (taken on THIS thread: an RLock belongs to the thread that
            # acquired it): a batch call from another thread then waits instead of running between this run's chunks on the
            # same workspace - and returning this run's frame counts and audio as its own
            try:
                with self._mu:
                    self._stream_owner = consumer
                    try:
                        rc = start(on_chunk)
                        err = None if rc == 0 or stop.is_set() else (rc, self._err())
                    finally:
                        self._stream_owner = None
                put(err)
            except BaseException as e:  # noqa: BLE001 - re-raised on the consumer's thread
                put(e)

        # (the consumer = the thread that iterates this generator: its frame-count queries pass the lock, see _locked)
        consumer = threading.get_ident()
        t = threading.Thread(target=work, daemon=True)
        t.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                if len(item) == 2:
                    cls = RangeError if item[0] == _ffi.VITS_E_RANGE else SessionError
                    raise cls(f"chunked run failed [{item[0]}]: {item[1]}")
                yield item
        finally:
            # generator closed early (break / GeneratorExit) or failed: tell the engine to stop after the chunk in
            # flight, so that the handle's mutex is released and the next call does not queue behind a dead render
            stop.set()
            t.join()

    def synthesize_stream(self, ids, lens, scales, sid=None, chunk_frames: int = 64, noise_dp=None, noise_z=None):
        """The whole path with the waveform delivered in chunks of `chunk_frames` frames (hop samples each): yields
        (first_sample, float32 [B, n], total_samples).  Concatenated, the chunks are bit-identical to
        synthesize_batch(...)["output"][:, 0, 0, :]; frame counts afterwards from last_y_lengths().  Chunks are handed out
        as they finish, so a range violation of the f16x3 arithmetic cannot be repaired by a silent re-run: it raises
        RangeError at the end (no bf16x6 fallback here; reopen with gen_precision="bf16x6").  Closing the generator
        early stops the engine after the chunk in flight."""
        ids = np.ascontiguousarray(ids, np.int64)
        lens = np.ascontiguousarray(lens, np.int64)
        scales = np.ascontiguousarray(scales, np.float32)
        B, T = ids.shape
        sid = None if sid is None else np.ascontiguousarray(sid, np.int64)
        noise = _ffi.VitsNoise()
        noise.seed = self._seed
        keep = []
        if noise_dp is not None:
            keep.append(np.ascontiguousarray(noise_dp, np.float32))
            noise.noise_dp = keep[-1].ctypes.data
        if noise_z is not None:
            keep.append(np.ascontiguousarray(noise_z, np.float32))
            noise.noise_z = keep[-1].ctypes.data
            noise.noise_z_stride = keep[-1].shape[2]
        return self._stream(lambda cb: self._lib.vits_run_chunked(
            self._h, _ffi.ptr(ids), _ffi.ptr(lens), B, T, _ffi.ptr(scales), _ffi.ptr(sid), C.byref(noise),
            int(chunk_frames), cb, None))

    def vocoder_stream(self, z, sid=None, chunk_frames: int = 64):
        """Vocoder only, chunked: yields (first_sample, float32 [B, n], total_samples)."""
        z = np.ascontiguousarray(z, np.float32)
        B, Cc, F = z.shape
        if Cc != self.hparam("inter"):
            raise SessionError(f"z must have {self.hparam('inter')} channels")
        sid = None if sid is None else np.ascontiguousarray(sid, np.int64)
        return self._stream(lambda cb: self._lib.vits_run_vocoder_chunked(self._h, _ffi.ptr(z), B, F, _ffi.ptr(sid),
                                                                          int(chunk_frames), cb, None))

    def tap(self, name):
        with self._locked():  # (two C calls on the last run's workspace)
            dims = (C.c_int64 * 4)()
            nd = self._lib.vits_tap(self._h, name.encode(), None, 0, dims)
            if nd < 0:
                raise SessionError(f"vits_tap({name}) failed: {self._err()}")
            shape = tuple(dims[i] for i in range(nd))
            buf = np.empty(shape, np.float32)
            nd = self._lib.vits_tap(self._h, name.encode(), _ffi.ptr(buf), buf.size, dims)
            if nd < 0:
                raise SessionError(f"vits_tap({name}) failed: {self._err()}")
            return buf

    def meta(self, key):
        buf = C.create_string_buffer(1 << 16)
        n = self._lib.vits_meta(self._h, key.encode(), buf, len(buf))
        return None if n < 0 else buf.value.decode()

    def hparam(self, key) -> int:
        v = C.c_int64()
        if self._lib.vits_hparam(self._h, key.encode(), C.byref(v)) != 0:
            raise SessionError(self._err())
        return v.value

    def set_tails(self, tails: str):
        """"zero" / "reference": see the constructor (applies to the following runs)."""
        if tails not in ("zero", "reference"):
            raise SessionError(f"tails must be 'zero' or 'reference' (got {tails!r})")
        with self._locked():
            self.tails = tails
            self._lib.vits_set_tails(self._h, 1 if tails == "reference" else 0)

    def reserve(self, batch: int, tokens: int = 0, frames: int = 0):
        """Size the device workspaces now for requests of up to `batch` utterances x `tokens` ids rendering up to `frames`
        frames each (vits_reserve): a serving process calls this once at start-up with the largest request it admits, so that
        no request reallocates tens of GB mid-stream (a device-wide synchronisation measured at up to seconds)."""
        with self._locked():
            if self._lib.vits_reserve(self._h, int(batch), int(tokens), int(frames)) != 0:
                raise SessionError(self._err())

    def set_timing(self, on=True):
        """True / 1: stage marks + events around every conv launch; 2: stage marks only; False / 0: off."""
        self._lib.vits_set_timing(self._h, 2 if on == 2 else (1 if on else 0))

    def stats(self):
        s = _ffi.VitsStats()
        with self._locked():
            self._lib.vits_get_stats(self._h, C.byref(s))
        d = {k: getattr(s, k) for k, _ in _ffi.VitsStats._fields_}
        d["range_fallbacks"] = self.range_fallbacks
        return d

    def launch_records(self):
        """Per conv-engine launch of the last run made with set_timing(True) (call stats() first: it reads the events):
        [{"kernel", "ms", "flops", "bytes", "stage"}] in launch order."""
        n = self._lib.vits_launch_records(self._h, None, 0)
        if n <= 0:
            return []
        buf = (_ffi.VitsLaunchRecord * n)()
        self._lib.vits_launch_records(self._h, buf, n)
        return [{"kernel": r.kernel.decode(), "ms": r.ms, "flops": r.flops, "bytes": r.bytes, "stage": r.stage,
                 "cin": r.cin, "cout": r.cout, "k": r.k, "dil": r.dil, "t": r.t} for r in buf]

    def arena_bytes(self):
        return self._lib.vits_arena_bytes(self._h)

    def arena_host(self) -> np.ndarray:
        n = self.arena_bytes()
        p = self._lib.vits_arena_host(self._h)
        if not p:
            raise SessionError("this handle holds no host copy of the weight arena (opened with a device arena / layout only)")
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n,))

    def arena_device(self) -> int:
        return self._lib.vits_arena_device(self._h) or 0

    def stream(self) -> int:
        return self._lib.vits_stream(self._h) or 0

    # device-resident run for the benchmark / sharded path: arguments are device pointers (ints)
    def run_device(self, ids_ptr, lens_ptr, B, T, scales, sid_ptr=None, noise_dp_ptr=None, noise_z_ptr=None,
                   noise_z_stride=0):
        scales = np.ascontiguousarray(scales, np.float32)
        noise = _ffi.VitsNoise()
        noise.seed = self._seed
        noise.noise_dp = noise_dp_ptr
        noise.noise_z = noise_z_ptr
        noise.noise_z_stride = noise_z_stride
        out = _ffi.VitsOutput()
        rc = self._lib.vits_run_device(self._h, C.c_void_p(ids_ptr), C.c_void_p(lens_ptr), B, T, _ffi.ptr(scales),
                                       C.c_void_p(sid_ptr) if sid_ptr else None, C.byref(noise), C.byref(out))
        if rc != 0:
            self._raise("vits_run_device", rc)
        dims = tuple(out.dims[i] for i in range(4))
        return {"data_ptr": C.cast(out.data, C.c_void_p).value, "dims": dims,
                "y_lengths_ptr": C.cast(out.y_lengths, C.c_void_p).value}

    def last_y_lengths(self) -> np.ndarray:
        with self._locked(read_only=True):  # (size, then contents: both of the same run; host state of the handle)
            n = self._lib.vits_last_y_lengths(self._h, None, 0)
            buf = np.zeros(max(n, 0), np.int64)
            if n > 0:
                self._lib.vits_last_y_lengths(self._h, buf.ctypes.data_as(C.POINTER(C.c_int64)), n)
            return buf

    def last_pcm16(self, normalize: bool = True, volume: float = 1.0, shape=None) -> np.ndarray:
        """int16 PCM of the last run, post-processed on the GPU exactly as TTSVoice.synthesize + AudioChunk do
        (voice.py:271-282, 88-91); [B, S], zeros past each utterance's length."""
        if shape is None:
            raise SessionError("last_pcm16 needs the [B, S] shape of the last output")
        out = np.zeros(shape, np.int16)
        with self._locked():
            rc = self._lib.vits_last_pcm16(self._h, 1 if normalize else 0, float(volume), _ffi.ptr(out), out.size)
        if rc != 0:
            raise SessionError(f"vits_last_pcm16 failed [{rc}]: {self._err()}")
        return out

    def synthesize_batch_pcm16(self, ids, lens, scales, sid=None, normalize: bool = True, volume: float = 1.0):
        """One batched run whose result leaves the GPU as 16-bit PCM only: peak-normalise / volume / clip / int16
        happen on the device (bit-identical to TTSVoice._postprocess + AudioChunk), the fp32 waveform is never
        copied to the host.  Returns (pcm int16 [B, S], y_lengths int64 [B])."""
        ids = np.ascontiguousarray(ids, np.int64)
        lens = np.ascontiguousarray(lens, np.int64)
        scales = np.ascontiguousarray(scales, np.float32)
        if ids.ndim != 2 or lens.shape != (ids.shape[0],) or scales.shape != (3,):
            raise SessionError(f"Invalid rank/shape for input: {ids.shape} / input_lengths: {lens.shape}")
        B, T = ids.shape
        if sid is not None:
            sid = np.ascontiguousarray(sid, np.int64)
        noise = _ffi.VitsNoise()
        noise.seed = self._seed
        with self._locked():
            rc = self._lib.vits_run(self._h, _ffi.ptr(ids), _ffi.ptr(lens), B, T, _ffi.ptr(scales), _ffi.ptr(sid),
                                    C.byref(noise), None)
            if rc != 0:
                raise SessionError(f"vits_run failed [{rc}]: {self._err()}")
            ylen = self.last_y_lengths()
            S = int(ylen.max()) * self.hparam("hop")
            return self.last_pcm16(normalize, volume, shape=(B, S)), ylen

    def sync(self):
        with self._locked():
            rc = self._lib.vits_sync(self._h)
            if rc != 0:
                self._raise("vits_sync", rc)


class PipelinedSession:
    """A batch rendered as `parts` sub-batches on `parts` engine handles (= HIP streams) that share ONE weight
    arena.  The encoder / duration / flow stages of a sub-batch have small grids that leave most of the chip idle;
    on separate streams they overlap with another sub-batch's generator (measured on one MI355X at batch 32 x 256
    ids: +3.5 % on the LJSpeech-size voice, +9 % on the default one; more than 2 parts loses).  Utterances are
    independent, so every sub-batch is an ordinary run and results are those of `MiSession` on the same rows
    (device noise streams differ per part: seed + part index)."""

    def __init__(self, first: MiSession, parts: int = 2):
        if parts < 1:
            raise SessionError("parts must be >= 1")
        self.parts = [first]
        # `first` owns the weight arena the other handles borrow: it must never close and reopen itself under them
        # (MiSession's own fallback would free the arena while the borrowers run on it).  The fallback happens HERE,
        # for all handles together (_fall_back).
        import threading
        self._mu = threading.RLock()   # one batched call at a time per pipeline (its parts' locks are taken inside)
        self.range_fallback = first.range_fallback
        self._first_had_fallback = first.range_fallback   # handed back by close()
        first.range_fallback = False
        self.range_fallbacks = 0
        self._n_parts = parts
        self._borrow()
        self.set_seed(first._seed)

    def _borrow(self):
        """(Re)create the handles that share parts[0]'s arena (laid out for ITS arithmetic)."""
        first = self.parts[0]
        # the arithmetic the owner was opened with, as it was asked for: the layout of every split-operand conv (encoder,
        # flow, generator) follows the request, also where this voice's generator cannot use that engine
        precision = first.gen_precision
        for _ in range(self._n_parts - 1):
            self.parts.append(MiSession(first.path, device_id=first.device_id, arena_device_ptr=first.arena_device(),
                                        arena_bytes=first.arena_bytes(), gen_precision=precision, tails=first.tails))

    def _fall_back(self, exc):
        """After a RangeError on any part: every borrower is synchronised and closed, THEN the owner reopens with the
        six-product arithmetic (bf16 planes: fp32 range), then the borrowers are recreated on the new arena."""
        import logging
        first = self.parts[0]
        if not self.range_fallback or first.gen_precision == "bf16x6":
            raise exc
        logging.getLogger(__name__).warning("%s: %s - reopening all %d handles with gen_precision='bf16x6'", first.path, exc,
                                            len(self.parts))
        for s in reversed(self.parts[1:]):
            try:
                s.sync()
            except SessionError:
                pass
            s.close()
        del self.parts[1:]
        try:
            first.sync()
        except SessionError:
            pass
        seed = first._seed
        first.close()
        first._open("bf16x6")
        first.range_fallbacks += 1
        self.range_fallbacks += 1
        self._borrow()
        self.set_seed(seed)

    @classmethod
    def open(cls, path, device_id: int = 0, parts: int = 2):
        return cls(MiSession(path, device_id=device_id), parts)

    def set_seed(self, seed: int):
        for i, s in enumerate(self.parts):
            s.set_seed(int(seed) + i)

    def reserve(self, batch: int, tokens: int = 0, frames: int = 0, whole_batch: bool = True):
        """MiSession.reserve on every part.  whole_batch=True sizes every part for the WHOLE batch (run_device_steps(...,
        alternate=True) deals complete requests to the parts); False for its rows only (the split schedule)."""
        bnd = self.bounds(batch)
        for i in range(len(self.parts) if whole_batch else len(bnd) - 1):
            self.parts[i].reserve(batch if whole_batch else bnd[i + 1] - bnd[i], tokens, frames)

    def set_tails(self, tails: str):
        """"zero" / "reference" on every handle (MiSession.set_tails)."""
        with self._mu:
            for s in self.parts:
                s.set_tails(tails)

    def hparam(self, key):
        return self.parts[0].hparam(key)

    def bounds(self, B):
        n = min(len(self.parts), B)
        return [B * i // n for i in range(n + 1)]

    def run_device(self, ids_ptr, lens_ptr, B, T, scales, sid_ptr=None):
        """Device pointers in (rows of one [B, T] int64 tensor / [B] tensors); enqueues every sub-batch on its own
        stream and returns [(first_row, rows, MiSession.run_device result)] - the waveform of a part stays in that
        part's workspace.  Call sync() before reading."""
        out = []
        bnd = self.bounds(B)
        for i in range(len(bnd) - 1):
            b0, nb = bnd[i], bnd[i + 1] - bnd[i]
            r = self.parts[i].run_device(ids_ptr + b0 * T * 8, lens_ptr + b0 * 8, nb, T, scales,
                                         sid_ptr + b0 * 8 if sid_ptr else None)
            out.append((b0, nb, r))
        return out

    def run_device_steps(self, ids_ptr, lens_ptr, B, T, scales, steps, sid_ptr=None, alternate=False):
        """`steps` back-to-back passes over the same device-resident batch with one host thread per part, as two
        serving workers would run: a part goes on to its next pass without waiting for the other one, and part i
        starts once part i-1 has handed its first generator to the GPU, so that the small-grid stages of one part
        keep falling under the generator of the other.  Returns the frame count of every utterance of every pass,
        int64 [steps, B]; all work has completed on return.
        alternate=False: every pass is split over the parts (rows [b0, b1) each) - a pass's latency is that of a sub-batch.
        alternate=True: WHOLE passes are dealt to the parts in turn (pass k on part k mod n: request-level pipelining, each
        worker renders complete batches) - the generator keeps the full batch's grids (a sub-batch of 11 leaves the default
        voice's 128-channel stage at 283 workgroups on 512 slots), while pass k + 1's token and frame stages still fall
        under pass k's generator on the other handle."""
        import threading
        bnd = self.bounds(B)
        n = len(bnd) - 1
        out = np.zeros((steps, B), np.int64)
        stamps = self.last_pass_stamps = [0.0] * steps  # (alternate: when pass k's call returned on its worker; diagnostics)
        started = [threading.Event() for _ in range(n)]
        errors = []

        def work(i):
            try:
                if i > 0:
                    started[i - 1].wait()
                if alternate:
                    for k in range(i, steps, n):
                        self.parts[i].run_device(ids_ptr, lens_ptr, B, T, scales, sid_ptr)
                        started[i].set()
                        out[k, :] = self.parts[i].last_y_lengths()
                        stamps[k] = time.perf_counter()
                    self.parts[i].sync()
                    return
                b0, nb = bnd[i], bnd[i + 1] - bnd[i]
                for k in range(steps):
                    self.parts[i].run_device(ids_ptr + b0 * T * 8, lens_ptr + b0 * 8, nb, T, scales,
                                             sid_ptr + b0 * 8 if sid_ptr else None)
                    started[i].set()
                    out[k, b0:b0 + nb] = self.parts[i].last_y_lengths()
                self.parts[i].sync()
            except Exception as e:  # noqa: BLE001 - re-raised on the caller's thread
                errors.append(e)
            finally:
                started[i].set()

        if n == 1:
            work(0)
        else:
            th = [threading.Thread(target=work, args=(i,)) for i in range(n)]
            for t in th:
                t.start()
            for t in th:
                t.join()
        if errors:
            raise errors[0]
        return out

    def last_y_lengths(self, B) -> np.ndarray:
        n = len(self.bounds(B)) - 1
        return np.concatenate([self.parts[i].last_y_lengths() for i in range(n)])

    def synthesize_batch(self, ids, lens, scales, sid=None):
        """Host arrays in, host arrays out, like MiSession.synthesize_batch.  One worker thread per sub-batch (the C
        calls release the GIL): each enqueues its part (vits_run_async), learns its frame counts, meets the others at a
        barrier where the ONE [B,1,1,S_max] result array (pinned host memory) is sized, then waits for its own render
        and lets the DMA engine write its rows straight into that array (vits_fetch_output) - the copy-out of a part
        that finishes early runs under the render of the others, and no byte is copied twice on the host."""
        import threading
        ids = np.ascontiguousarray(ids)
        lens = np.ascontiguousarray(lens)
        scales = np.ascontiguousarray(scales)
        if ids.dtype != np.int64 or lens.dtype != np.int64 or ids.ndim != 2 or lens.shape != (ids.shape[0],):
            raise SessionError("Unexpected input: 'input' int64 [B,T], 'input_lengths' int64 [B]")
        if scales.dtype != np.float32 or scales.shape != (3,):
            raise SessionError("Unexpected input: 'scales' must be float32 of shape [3]")
        B = ids.shape[0]
        if sid is not None:
            sid = np.ascontiguousarray(sid)
            if sid.dtype != np.int64 or sid.shape != (B,):
                raise SessionError("Unexpected input: 'sid' must be int64 of shape [batch_size]")
        with self._mu:
            return self._synthesize_batch_locked(ids, lens, scales, sid, B)

    def _synthesize_batch_locked(self, ids, lens, scales, sid, B):
        import threading
        bnd = self.bounds(B)
        n = len(bnd) - 1
        hop = self.hparam("hop")
        ylen = np.zeros(B, np.int64)
        box = {}
        pinned = self.parts[0].pinned_results

        def size_output():  # (barrier action: runs once, in one thread, when every part knows its frame counts)
            shape = (B, 1, 1, int(ylen.max()) * hop)
            box["out"] = _POOL.array(shape) if pinned else np.empty(shape, np.float32)

        bar = threading.Barrier(n, action=size_output)
        errors = []

        def work(i):
            b0, b1 = bnd[i], bnd[i + 1]
            p = self.parts[i]
            try:
                noise = _ffi.VitsNoise()
                noise.seed = p._seed
                with p._mu:  # (a part may also be used on its own, e.g. bench.py's measure(): same per-session lock)
                    p._begin(ids[b0:b1], lens[b0:b1], scales, None if sid is None else sid[b0:b1], noise)
                    ylen[b0:b1] = p.last_y_lengths()
                    bar.wait()
                    p._fetch(box["out"], b0, b1 - b0)
            except threading.BrokenBarrierError:
                pass  # another part failed; its error is reported
            except Exception as e:  # noqa: BLE001 - re-raised on the caller's thread
                errors.append(e)
                bar.abort()

        if n == 1:
            work(0)
        else:
            th = [threading.Thread(target=work, args=(i,)) for i in range(n)]
            for t in th:
                t.start()
            for t in th:
                t.join()
        if errors:
            rng = [e for e in errors if isinstance(e, RangeError)]
            if rng:
                self._fall_back(rng[0])
                return self.synthesize_batch(ids, lens, scales, sid)
            raise errors[0]
        return {"output": box["out"], "y_lengths": ylen}

    def sync(self):
        for s in self.parts:
            s.sync()

    def close(self, close_first: bool = True):
        """Closes the borrowed handles and (close_first) the arena owner.  close_first=False hands the first session back
        to its caller as it was given: open, with its own range fallback restored."""
        for s in reversed(self.parts[1:]):  # the first handle owns the arena the others borrow
            s.close()
        first = self.parts[0]
        del self.parts[1:]
        first.range_fallback = self._first_had_fallback
        if close_first:
            first.close()


# kernel-level hooks (tests)
def test_conv1d(x, w, bias=None, dil=1, pad_l=0, lrelu_slope=None, relu=False, device_id=0, hint=0):
    """hint: tile-size class as chosen at pack time (0 generator, 1 flow, 2 token domain)."""
    lib = _ffi.load()
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    B, Cin, T = x.shape
    Cout, _, K = w.shape
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    out = np.empty((B, Cout, T), np.float32)
    flags = (1 if lrelu_slope is not None else 0) | (2 if relu else 0) | ((hint & 3) << 8)
    rc = lib.vits_test_conv1d(device_id, _ffi.ptr(x), B, Cin, T, _ffi.ptr(w), _ffi.ptr(b), Cout, K, dil, pad_l, flags,
                              float(lrelu_slope or 0.0), _ffi.ptr(out))
    if rc != 0:
        raise SessionError(_ffi.last_error(None))
    return out


def test_conv1d_sx(x, w, bias=None, dil=1, pad_l=0, planes_slope=None, residual=False, in_slope=None, device_id=0,
                   precision="f32", planes_only=False, small=False):
    """The split-operand engine (Cin % 16 == 0, Cout % 32 == 0).  planes_slope: read the result back from
    the 16-bit output planes, which carry leaky_relu(conv, planes_slope); residual: out = conv(x) + x;
    in_slope: out = conv(leaky_relu(x, in_slope)) [+ x] - Cin <= 64 (the raw-input kernels), or precision "f16", whose
    input plane then holds leaky_relu(x) and whose residual is recovered from that plane.
    precision: "f32" = six exact bf16 plane products; "f16x3" = two fp16 planes, three products (fp32-grade); "f16" = one
    fp16 plane, one product, fp16 storage (the reduced-precision vocoder of BASELINE config 4).
    planes_only: the planes are the launch's only output (the specialised plane epilogues); needs planes_slope.
    small: the short-launch kernel (conv_sx_small.hip.hpp, f16x3 on plane inputs, Cin % 32 == 0) with the same epilogue."""
    lib = _ffi.load()
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    B, Cin, T = x.shape
    Cout, _, K = w.shape
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    out = np.empty((B, Cout, T), np.float32)
    flags = (1 if planes_slope is not None else 0) | (4 if residual else 0) | (8 if in_slope is not None else 0)
    flags |= {"f32": 0, "f16": 2, "f16x3": 3}[precision] << 4
    flags |= 128 if planes_only else 0
    flags |= 256 if small else 0
    if in_slope is not None and planes_slope is not None and in_slope != planes_slope:
        raise ValueError("the hook takes one slope value")
    if in_slope is not None and Cin > 64 and precision != "f16":
        raise ValueError("in_slope needs a raw-input conv (Cin <= 64) or the single-plane arithmetic")
    slope = planes_slope if planes_slope is not None else in_slope
    rc = lib.vits_test_conv1d_sx(device_id, _ffi.ptr(x), B, Cin, T, _ffi.ptr(w), _ffi.ptr(b), Cout, K, dil, pad_l,
                                 flags, float(slope or 0.0), _ffi.ptr(out))
    if rc != 0:
        raise SessionError(_ffi.last_error(None))
    return out


def test_conv1d_sx_planar(x, w, bias=None, dil=1, lens=None, old=None, row_split=None, pl_rows=0, relu=False, mask=False,
                          residual=False, accumulate=False, coupling=False, store2=False, planes_of2=False, device_id=0,
                          small=False):
    """The split-operand engine's planar epilogue (f16x3, "same" padding): o = old + act(conv(x) + bias) * mask, rows
    [0, row_split) in a first tensor, the rest in a second (see include/vitsmi.h).  small=True: through the short-launch
    kernel (conv_sx_small.hip.hpp) instead of the engine's.  Returns (out [B, Cout, T], planes [B, pl_rows, T] or None)."""
    lib = _ffi.load()
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    B, Cin, T = x.shape
    Cout, _, K = w.shape
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    ln = None if lens is None else np.ascontiguousarray(lens, np.int64)
    od = None if old is None else np.ascontiguousarray(old, np.float32)
    out = np.empty((B, Cout, T), np.float32)
    pl = np.empty((B, pl_rows, T), np.float32) if pl_rows else None
    flags = (1 if relu else 0) | (2 if mask else 0) | (4 if residual else 0) | (8 if accumulate else 0) | \
            (16 if coupling else 0) | (32 if store2 else 0) | (64 if planes_of2 else 0) | (128 if small else 0)
    rc = lib.vits_test_conv1d_sx_planar(device_id, _ffi.ptr(x), B, Cin, T, _ffi.ptr(w), _ffi.ptr(b), Cout, K, dil, flags,
                                        _ffi.ptr(ln), _ffi.ptr(od), Cout if row_split is None else row_split, pl_rows,
                                        _ffi.ptr(out), _ffi.ptr(pl))
    if rc != 0:
        raise SessionError(_ffi.last_error(None))
    return out, pl


def test_conv1d_sx_gate(x, w, bias, g, dil=1, small=False, planes=False, device_id=0):
    """The flow's WN in-layer with its gate epilogue: acts = tanh(a + g_a) * sigmoid(b + g_b) where (a | b) = conv(x) + bias
    ("same" padding), w [2H, Cin, K], bias [2H], g [B, 2H] in the module's channel order (modules.py:195-203,
    commons.py:99-106).  The hook wants the rows pair-interleaved the way the packer lays them out; that permutation is
    applied here.  small: the short-launch kernel; planes: the result through the fp16 operand planes.  -> [B, H, T]"""
    lib = _ffi.load()
    x = np.ascontiguousarray(x, np.float32)
    B, Cin, T = x.shape
    C2, _, K = w.shape
    H = C2 // 2
    perm = np.concatenate([np.concatenate([np.arange(p, p + 32), H + np.arange(p, p + 32)]) for p in range(0, H, 32)])
    wp = np.ascontiguousarray(np.asarray(w, np.float32)[perm])
    bp = np.ascontiguousarray(np.asarray(bias, np.float32)[perm])
    gg = np.ascontiguousarray(g, np.float32)
    out = np.empty((B, H, T), np.float32)
    rc = lib.vits_test_conv1d_sx_gate(device_id, _ffi.ptr(x), B, Cin, T, _ffi.ptr(wp), _ffi.ptr(bp), _ffi.ptr(gg), C2, K, dil,
                                      (1 if small else 0) | (2 if planes else 0), _ffi.ptr(out))
    if rc != 0:
        raise SessionError(_ffi.last_error(None))
    return out


def test_conv_pair_sx(x, w1, b1, w2, b2, dil1=1, dil2=1, chain=False, slope=0.1, device_id=0, timed=False, kernel="pair",
                      from_plane=False):
    """Two dependent convs in ONE fused launch (32- / 64-channel stage of the generator):
    chain=False (ResBlock1 step): out = c2(lrelu(c1(lrelu(x)))) + x
    chain=True (two ResBlock2 steps): x1 = c1(lrelu(x)) + x; out = c2(lrelu(x1)) + x1.    timed=True -> (out, ms).
    kernel: "pair" = conv_sx_pair_kernel (32x32x16 loop, f16x3), "pair16" = conv_sx_pair16_kernel (16x16x32 loop) in f16x3,
    "pair16_f16" = the same in the single-plane arithmetic (fp16 plane in; from_plane: the result read back from the output
    plane, leaky_relu(out, slope) as fp16)."""
    lib = _ffi.load()
    x = np.ascontiguousarray(x, np.float32)
    w1 = np.ascontiguousarray(w1, np.float32)
    w2 = np.ascontiguousarray(w2, np.float32)
    B, Cc, T = x.shape
    K = w1.shape[2]
    if w1.shape != (Cc, Cc, K) or w2.shape != (Cc, Cc, K):
        raise ValueError("both convs are C -> C with the same kernel size")
    b1 = None if b1 is None else np.ascontiguousarray(b1, np.float32)
    b2 = None if b2 is None else np.ascontiguousarray(b2, np.float32)
    out = np.empty_like(x)
    ms = C.c_float(0.0)
    rc = lib.vits_test_conv_pair_sx(device_id, _ffi.ptr(x), B, Cc, T, _ffi.ptr(w1), _ffi.ptr(b1), _ffi.ptr(w2),
                                    _ffi.ptr(b2), K, dil1, dil2,
                                    (1 if chain else 0) | ({"pair": 0, "pair16": 1, "pair16_f16": 2}[kernel] << 1) | (8 if from_plane else 0),
                                    float(slope), _ffi.ptr(out),
                                    C.byref(ms) if timed else None)
    if rc != 0:
        raise SessionError(_ffi.last_error(None))
    return (out, float(ms.value)) if timed else out


def bench_conv1d_sx(B, Cin, Cout, T, K, dil=1, dbg=0, iters=20, device_id=0):
    """Average launch time (ms) of one conv shape on the split-exact engine -> (ms, tile config)."""
    lib = _ffi.load()
    res = np.zeros(8, np.float32)
    rc = lib.vits_bench_conv1d_sx(device_id, B, Cin, Cout, T, K, dil, dbg, iters,
                                  res.ctypes.data_as(_ffi.C.POINTER(_ffi.C.c_float)))
    if rc != 0:
        raise SessionError(_ffi.last_error(None))
    if dbg & 16:
        return float(res[0]), int(res[1]), [float(v) for v in res[3:8]] + [float(res[2])]
    return float(res[0]), int(res[1])


def test_conv_transpose1d(x, w, bias, stride, device_id=0, sx=False):
    lib = _ffi.load()
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    B, Cin, T = x.shape
    _, Cout, K = w.shape
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    out = np.empty((B, Cout, T * stride), np.float32)
    fn = lib.vits_test_conv_transpose1d_sx if sx else lib.vits_test_conv_transpose1d
    # sx="f16": the split-exact engine in its fp16 two-plane mode (the hook takes it as a negative stride)
    rc = fn(device_id, _ffi.ptr(x), B, Cin, T, _ffi.ptr(w), _ffi.ptr(b), Cout, K, -stride if sx == "f16" else stride,
            _ffi.ptr(out))
    if rc != 0:
        raise SessionError(_ffi.last_error(None))
    return out


def test_attention16(qkv, n_heads, rel_k, rel_v, lens, device_id=0, kernel=1, planes=False, reps=0):
    """attention16.hip.hpp (kernel=1) or the fp32-MFMA kernel (0) on q|k|v planar fp32 [B, 3C, T]; returns out [B, C, T],
    plus the output's operand planes (uint16 [B, 3, C/8, T, 8]) when planes=True, plus ms per launch when reps > 0."""
    lib = _ffi.load()
    qkv = np.ascontiguousarray(qkv, np.float32)
    B, C3, T = qkv.shape
    Cc = C3 // 3
    rel_k = np.ascontiguousarray(rel_k, np.float32)
    rel_v = np.ascontiguousarray(rel_v, np.float32)
    window = (rel_k.shape[0] - 1) // 2
    lens = np.ascontiguousarray(lens, np.int64)
    out = np.empty((B, Cc, T), np.float32)
    opl = np.empty((B, 3, Cc // 8, T, 8), np.uint16) if planes else None
    ms = (C.c_float * 1)()
    rc = lib.vits_test_attention16(device_id, _ffi.ptr(qkv), B, Cc, T, n_heads, _ffi.ptr(rel_k), _ffi.ptr(rel_v), window,
                                   _ffi.ptr(lens), _ffi.ptr(out), _ffi.ptr(opl) if planes else None, int(kernel), int(reps), ms)
    if rc != 0:
        raise SessionError(_ffi.last_error(None))
    res = (out,)
    if planes:
        res += (opl,)
    if reps > 0:
        res += (float(ms[0]),)
    return res[0] if len(res) == 1 else res


def test_attention(qkv, n_heads, rel_k, rel_v, lens, device_id=0):
    lib = _ffi.load()
    qkv = np.ascontiguousarray(qkv, np.float32)
    B, C3, T = qkv.shape
    Cc = C3 // 3
    rel_k = np.ascontiguousarray(rel_k, np.float32)
    rel_v = np.ascontiguousarray(rel_v, np.float32)
    window = (rel_k.shape[0] - 1) // 2
    lens = np.ascontiguousarray(lens, np.int64)
    out = np.empty((B, Cc, T), np.float32)
    rc = lib.vits_test_attention(device_id, _ffi.ptr(qkv), B, Cc, T, n_heads, _ffi.ptr(rel_k), _ffi.ptr(rel_v), window,
                                 _ffi.ptr(lens), _ffi.ptr(out))
    if rc != 0:
        raise SessionError(_ffi.last_error(None))
    return out
