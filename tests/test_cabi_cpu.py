"""CPU-side checks of the C ABI: the library loads, exports every symbol include/vitsmi.h
declares, and the product's C++ .onnx reader agrees with the oracle's independent Python
walker.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, TINY_PRESETS

from phoonnx_amd import MiSession, SessionError, _ffi


def test_header_symbols_are_exported():
    hdr = open(os.path.join(ROOT, "include", "vitsmi.h")).read()
    declared = set(re.findall(r"\b(vits_[a-z0-9_]+)\s*\(", hdr))
    lib = _ffi.load()
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in vitsmi.h but not exported"
    assert declared == set(_ffi.EXPORTS), declared ^ set(_ffi.EXPORTS)


@pytest.mark.parametrize("preset", TINY_PRESETS)
def test_host_only_open_matches_oracle_reader(preset):
    import json
    from vits_oracle import VitsOracle
    path = os.path.join(GOLDEN, preset + ".onnx")
    s = MiSession(path, host_only=True)
    o = VitsOracle(path)
    hp = json.load(open(os.path.join(GOLDEN, preset + ".hparams.json")))
    assert [i.name for i in s.get_inputs()] == o.input_names
    assert s.hparam("hidden") == hp["hidden_channels"]
    assert s.hparam("inter") == hp["inter_channels"]
    assert s.hparam("filter") == hp["filter_channels"]
    assert s.hparam("n_heads") == hp["n_heads"]
    assert s.hparam("n_layers") == hp["n_layers"]
    assert s.hparam("n_vocab") == hp["n_vocab"]
    assert s.hparam("n_speakers") == hp["n_speakers"]
    assert s.hparam("gin") == hp["gin_channels"]
    assert s.hparam("use_sdp") == int(hp["use_sdp"])
    assert s.hparam("hop") == int(np.prod(hp["upsample_rates"]))
    assert s.hparam("resblock") == int(hp["resblock"])
    assert s.meta("sample_rate") == "22050" and s.meta("model_type") == "vits"
    assert s.meta("nope") is None
    assert s.arena_bytes() > 0 and s.arena_bytes() % 4 == 0
    # every raw parameter the oracle resolved is present bit-for-bit somewhere in the packed arena
    arena = s.arena_host().view(np.float32)
    emb = o.tensors["enc_p.emb.weight"].ravel()
    # the embedding table is stored verbatim (bit-exact lookup requirement)
    idx = np.flatnonzero(arena == emb[0])
    assert any(np.array_equal(arena[i:i + emb.size], emb) for i in idx)
    with pytest.raises(SessionError):
        s.synthesize_batch(np.zeros((1, 4), np.int64), np.array([4], np.int64), np.array([0, 1, 0], np.float32),
                           sid=np.zeros(1, np.int64))  # host-only handles cannot run
    s.close()


def test_open_errors():
    with pytest.raises(SessionError, match="cannot open"):
        MiSession("/nonexistent/voice.onnx", host_only=True)
    bad = os.path.join(ROOT, "tests", "golden", "tiny_dp.hparams.json")
    with pytest.raises(SessionError):
        MiSession(bad, host_only=True)


def test_damaged_files_are_rejected_not_crashed_on(tmp_path):
    """Truncated and byte-flipped .onnx files must either load or raise SessionError: the protobuf walker and the
    packer never trust a length field.  Runs in a child process so that a crash would fail this test only."""
    import subprocess
    import sys
    code = r'''
import random, sys
sys.path.insert(0, sys.argv[1])
from phoonnx_amd import MiSession
from phoonnx_amd.session import SessionError
src = open(sys.argv[2], "rb").read()
rng = random.Random(7)
ok = err = 0
for it in range(60):
    b = bytearray(src)
    if it % 2:
        b = b[:rng.randrange(0, len(b))]
    else:
        for _ in range(rng.randrange(1, 8)):
            b[rng.randrange(len(b))] = rng.randrange(256)
    p = sys.argv[3]
    open(p, "wb").write(bytes(b))
    try:
        MiSession(p, host_only=True).close()
        ok += 1
    except SessionError:
        err += 1
print("survived", ok, err)
'''
    r = subprocess.run([sys.executable, "-c", code, ROOT, os.path.join(GOLDEN, "tiny_rb1.onnx"),
                        str(tmp_path / "damaged.onnx")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "survived" in r.stdout, (r.returncode, r.stdout[-300:], r.stderr[-300:])
    n_ok, n_err = (int(v) for v in r.stdout.split()[-2:])
    assert n_ok + n_err == 60 and n_err >= 30   # every truncation is rejected


def test_work_counts_match_survey_formulas():
    # SURVEY App. C formulas evaluated for the tiny preset: conv MACs per frame of the generator
    import json
    s = MiSession(os.path.join(GOLDEN, "tiny_rb1.onnx"), host_only=True)
    hp = json.load(open(os.path.join(GOLDEN, "tiny_rb1.hparams.json")))
    C, C0 = hp["inter_channels"], hp["upsample_initial_channel"]
    macs, t, ch = C * C0 * 7, 1, C0
    for u, k in zip(hp["upsample_rates"], hp["upsample_kernel_sizes"]):
        macs += t * ch * (ch // 2) * k
        t *= u
        ch //= 2
        for rk, rd in zip(hp["resblock_kernel_sizes"], hp["resblock_dilation_sizes"]):
            macs += t * ch * ch * rk * len(rd) * 2
    macs += t * ch * 7
    assert s.hparam("dec_macs_per_frame") == macs
