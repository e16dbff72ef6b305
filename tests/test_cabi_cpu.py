"""CPU-side checks of the C ABI: the library loads, exports every symbol include/vitsmi.h
declares, and the product's C++ .onnx reader agrees with the oracle's independent Python
walker.  No compute calls (no GPU here)."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

from conftest import ALL_PRESETS, GOLDEN, ROOT, TINY_PRESETS

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


@pytest.mark.parametrize("preset", ALL_PRESETS)
@pytest.mark.parametrize("precision", [None, "bf16x6"])
def test_layout_only_open_equals_full_pack(preset, precision):
    """vits_open_with_arena (a rank that received the weights by broadcast, a second handle on one GPU) lays the arena
    out without packing a weight: same size, same description as the full pack, and no host copy."""
    path = os.path.join(GOLDEN, preset + ".onnx")
    full = MiSession(path, host_only=True, gen_precision=precision)
    lay = MiSession(path, layout_only=True, gen_precision=precision)
    assert lay.arena_bytes() == full.arena_bytes() > 0
    for k in ("hidden", "inter", "filter", "n_heads", "n_layers", "n_vocab", "n_speakers", "gin", "gen_sx", "gen_nprod",
              "use_sdp", "hop", "n_ups", "resblock", "window", "upsample_initial_channel", "dec_macs_per_frame",
              "flow_macs_per_frame", "enc_macs_per_token", "gen_rf_frames"):
        assert lay.hparam(k) == full.hparam(k), k
    assert [i.name for i in lay.get_inputs()] == [i.name for i in full.get_inputs()]
    with pytest.raises(SessionError):
        lay.arena_host()
    assert full.arena_host().size == full.arena_bytes()
    lay.close()
    full.close()


def test_gen_precision_option_and_fixture_runs_on_sx():
    path = os.path.join(GOLDEN, "sx_rb1.onnx")
    for name, nprod in (("f16x3", 2), ("bf16x6", 6), ("f16", 1), (None, 2)):
        s = MiSession(path, host_only=True, gen_precision=name)
        assert s.hparam("gen_sx") == 1 and s.hparam("gen_nprod") == nprod
        s.close()
    with pytest.raises(SessionError, match="VITSMI_GEN_PRECISION|precision"):
        MiSession(path, host_only=True, gen_precision="fp8")
    tiny = MiSession(os.path.join(GOLDEN, "tiny_rb1.onnx"), host_only=True)   # channels < 32: the f32-MFMA engine
    assert tiny.hparam("gen_sx") == 0
    tiny.close()


def test_generator_receptive_field():
    """gen_rf_frames = what chunked rendering discards on either side of a chunk.  LJSpeech-size generator
    (models.py:299-368): conv_pre 3 + per stage (transposed-conv reach / rate_in + widest ResBlock / rate_out) + conv_post."""
    from phoonnx_amd.synth import write_voice
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "v.onnx")
        write_voice(p, "small", seed=1)
        s = MiSession(p, layout_only=True)
        # small: ups (8,4,2), k (16,8,4), ResBlock2 k (3,5,7) d ((1,2),(2,6),(3,12)): widest = 3*3 + 3*12 = 45 samples
        r = 3 + 1 / 1 + 45 / 8 + 1 / 8 + 45 / 32 + 1 / 32 + 45 / 64 + 3 / 64
        assert s.hparam("gen_rf_frames") == int(np.ceil(r)) + 1
        s.close()


def test_langid_input_is_listed_and_unknown_inputs_are_rejected(tmp_path):
    """voice.py:369 offers `langid` to graphs that declare it.  A graph that declares it without consuming a language
    table loads (the name is listed so the caller's feed filter keeps it); unknown inputs are not a VITS graph."""
    from phoonnx_amd.synth import write_voice
    p = str(tmp_path / "lang.onnx")
    write_voice(p, "small", seed=2, extra_inputs=("langid",))
    s = MiSession(p, host_only=True)
    assert [i.name for i in s.get_inputs()] == ["input", "input_lengths", "scales", "langid"]
    s.close()
    q = str(tmp_path / "odd.onnx")
    write_voice(q, "small", seed=2, extra_inputs=("prosody",))
    with pytest.raises(SessionError, match="unsupported graph input"):
        MiSession(q, host_only=True)


def _tensor_spans(buf):
    """(dims byte offsets, name) of every TensorProto whose layout is dims* data_type name raw_data (what the
    exporter and synth.py write): 08 d .. 10 01 42 len name."""
    out = []
    i = 0
    while True:
        i = buf.find(b"\x10\x01\x42", i)
        if i < 0:
            return out
        n = buf[i + 3]
        name = bytes(buf[i + 4:i + 4 + n])
        j, dims = i, []
        while j >= 2 and buf[j - 2] == 0x08 and buf[j - 1] < 0x80:
            dims.append(j - 1)
            j -= 2
        if dims and name.isascii() and b"." in name:
            out.append((dims[::-1], name.decode()))
        i += 3


def test_targeted_shape_and_wire_type_damage_is_rejected(tmp_path):
    """random byte flips rarely hit a key byte or a dims field.  Targeted damage: (a) every parameter's dims
    rewritten (0, 1, swapped, rank dropped), (b) the key byte of name / raw_data / node strings / metadata switched to
    another wire type.  Each file must load or raise SessionError - in a child process, so a crash fails only this."""
    import subprocess
    import sys
    src = bytearray(open(os.path.join(GOLDEN, "tiny_rb2_ms.onnx"), "rb").read())
    spans = _tensor_spans(src)
    assert len(spans) > 40
    cases = []
    rng = np.random.default_rng(5)
    for dims, name in spans:
        for kind in range(4):
            b = bytearray(src)
            if kind == 0:
                b[dims[int(rng.integers(len(dims)))]] = 0
            elif kind == 1:
                b[dims[int(rng.integers(len(dims)))]] = int(rng.integers(1, 120))
            elif kind == 2 and len(dims) > 1:
                b[dims[0]], b[dims[-1]] = b[dims[-1]], b[dims[0]]
            elif kind == 3:
                b[dims[0] - 1] = 0x18   # the first dims key becomes an unknown varint field: rank drops by one
            else:
                continue
            cases.append(bytes(b))
    for key, repl in ((b"\x10\x01\x42", b"\x10\x01\x40"), (b"\x10\x01\x42", b"\x10\x01\x45")):
        pos = [m for m in range(len(src) - 3) if src[m:m + 3] == key][:25]
        for m in pos:
            b = bytearray(src)
            b[m:m + 3] = repl
            cases.append(bytes(b))
    cases.append(bytes([0x3a, 0x04, 0x2a, 0x02, 0x40, 0x01]))     # graph{initializer{key 0x40 (name as varint), 1}}
    cases.append(bytes([0x3a, 0x04, 0x2a, 0x02, 0x48, 0x01]))     # ... raw_data as varint
    cases.append(bytes([0x3a, 0x04, 0x0a, 0x02, 0x18, 0x01]))     # graph{node{name as varint}}
    cases.append(bytes([0x72, 0x02, 0x08, 0x01]))                 # metadata key as varint
    blob = tmp_path / "cases.bin"
    with open(blob, "wb") as f:
        for c in cases:
            f.write(len(c).to_bytes(4, "little") + c)
    code = r'''
import sys
sys.path.insert(0, sys.argv[1])
from phoonnx_amd import MiSession, SessionError
data = open(sys.argv[2], "rb").read()
i = ok = err = 0
while i < len(data):
    n = int.from_bytes(data[i:i + 4], "little"); i += 4
    open(sys.argv[3], "wb").write(data[i:i + n]); i += n
    try:
        MiSession(sys.argv[3], host_only=True).close(); ok += 1
    except SessionError:
        err += 1
print("survived", ok, err)
'''
    r = subprocess.run([sys.executable, "-c", code, ROOT, str(blob), str(tmp_path / "c.onnx")], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "survived" in r.stdout, (r.returncode, r.stdout[-300:], r.stderr[-500:])
    n_ok, n_err = (int(v) for v in r.stdout.split()[-2:])
    assert n_ok + n_err == len(cases) and n_err > len(cases) // 2, (n_ok, n_err)


# ---- structure-keyed weight resolution (SURVEY §7 "Weight lookup", App. B)
def _renamed(tmp_path, preset, strip):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from rename_nodes import rename
    out = tmp_path / f"{preset}{'_strip' if strip else ''}.onnx"
    out.write_bytes(rename(open(os.path.join(GOLDEN, preset + ".onnx"), "rb").read(), strip=strip))
    return str(out)


@pytest.mark.parametrize("preset", ALL_PRESETS)
@pytest.mark.parametrize("strip", [False, True])
def test_graphs_without_module_path_node_names_resolve_by_structure(tmp_path, preset, strip):
    """Older (Piper-era) exports name their nodes `Conv_123`, or not at all.  With every node of a fixture renamed to
    `<op>_<n>` (or its name removed) the loader walks Conv / ConvTranspose / Gather / LayerNorm / Pad nodes in graph order
    and recognises each module by position and (Cout, Cin, kernel, group): the packed weight arena and every derived
    hyper-parameter must equal those of the file with module-path names - which itself is only accepted because the
    structural walk and its names agree (model.cpp resolve: cross-check)."""
    a = MiSession(os.path.join(GOLDEN, preset + ".onnx"), host_only=True)
    b = MiSession(_renamed(tmp_path, preset, strip), host_only=True)
    assert a.arena_bytes() == b.arena_bytes()
    assert np.array_equal(a.arena_host(), b.arena_host())
    for k in ("hidden", "inter", "filter", "n_heads", "n_layers", "n_vocab", "n_speakers", "gin", "use_sdp", "hop", "n_ups",
              "resblock", "gen_sx", "gen_rf_frames", "dec_macs_per_frame", "flow_macs_per_frame"):
        assert a.hparam(k) == b.hparam(k), k
    a.close()
    b.close()


def test_structure_that_is_not_a_vits_export_is_rejected_with_a_reason(tmp_path):
    """No names AND an unexpected node order (here: the graph truncated after the encoder's convs) -> an error that says
    what did not fit, not a crash and not a half-resolved model."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import rename_nodes as rn
    src = open(os.path.join(GOLDEN, "tiny_rb1.onnx"), "rb").read()
    # drop the last Conv node (dec.conv_post) from the renamed graph
    out = bytearray()
    for fn, wt, f0, a, b in rn._fields(src, 0, len(src)):
        if fn != 7 or wt != 2:
            out += src[f0:b]
            continue
        nodes = [(g0, ga, gb) for gfn, gwt, g0, ga, gb in rn._fields(src, a, b) if gfn == 1 and gwt == 2]
        last_conv = max(i for i, (g0, ga, gb) in enumerate(nodes)
                        if any(nfn == 4 and bytes(src[na:nb]) == b"Conv" for nfn, nwt, n0, na, nb in rn._fields(src, ga, gb)))
        g = bytearray()
        k = 0
        for gfn, gwt, g0, ga, gb in rn._fields(src, a, b):
            if gfn == 1 and gwt == 2:
                if k != last_conv:
                    g += src[g0:gb]
                k += 1
            else:
                g += src[g0:gb]
        out += rn._ld(7, bytes(g))
    bad = tmp_path / "cut.onnx"
    bad.write_bytes(rn.rename(bytes(out)))
    with pytest.raises(SessionError, match="structure|conv_post|not found"):
        MiSession(str(bad), host_only=True)


@pytest.mark.parametrize("variant", ["opset11", "opset13", "opset17", "initinputs", "nofold"])
def test_exporter_variants_resolve_or_are_refused_with_a_reason(variant):
    """VERDICT r3 item 6c.  The reference exports one way (opset 15, constant folding on, initializers not listed as inputs:
    phoonnx_train/export_onnx.py:318-327); third-party voices differ.  tests/golden/variants/ holds the `tiny_rb2_ms` model
    of the fixtures written by the SAME exporter call with one knob changed (oracle/gen_golden.py --variants).  The reader
    must produce the identical packed arena and hyper-parameters - opset 11 / 13 (other shape-plumbing nodes), opset 17
    (LayerNorm as ONE LayerNormalization node instead of a ReduceMean .. Mul Add chain), initializers listed as graph inputs
    (not to be mistaken for feeds) - or refuse with the reason: without constant folding the flow's weight-norm chains stay in
    the graph as arithmetic this reader does not evaluate."""
    base = MiSession(os.path.join(GOLDEN, "tiny_rb2_ms.onnx"), host_only=True)
    path = os.path.join(GOLDEN, "variants", f"tiny_rb2_ms.{variant}.onnx")
    if variant == "nofold":
        with pytest.raises(SessionError, match="computed inside the graph.*constant folding"):
            MiSession(path, host_only=True)
        base.close()
        return
    s = MiSession(path, host_only=True)
    assert [i.name for i in s.get_inputs()] == ["input", "input_lengths", "scales", "sid"]
    for k in ("hidden", "inter", "filter", "n_heads", "n_layers", "n_speakers", "gin", "use_sdp", "hop", "n_ups", "resblock"):
        assert s.hparam(k) == base.hparam(k), k
    assert np.array_equal(np.asarray(s.arena_host()), np.asarray(base.arena_host()))
    s.close()
    base.close()



# ------------------------------------------------------------------ the build's ISA rule for spilled registers
_ASM_HEAD = "\n_ZN6vitsmi4demoEv: ; @demo\n"
_ASM_TAIL = "\ts_endpgm\n.Lfunc_end0:\n"


def _kernel(body):
    return _ASM_HEAD + "\n".join("\t" + ln if not ln.startswith((";;#", ".L")) else ln for ln in body) + "\n" + _ASM_TAIL


def test_spill_hazard_check_finds_a_spilled_destination_of_an_asynchronous_load():
    """phoonnx_amd.build.spill_hazards (DESIGN 5.1g hazard 1) on hand-written ISA: a scratch_store of a register that an
    inline-asm load has been issued to, before anything read it or a full drain passed, is a finding; the same spill behind
    the drain, behind the first use, or of a register the compiler itself loaded, is not."""
    from phoonnx_amd.build import spill_hazards
    load = [";;#ASMSTART", "global_load_dwordx4 v[10:13], v2, s[4:5] offset:1024", ";;#ASMEND"]
    spill = ["scratch_store_dwordx4 off, v[10:13], off offset:16 ; 16-byte Folded Spill"]
    # 1. spilled while in flight
    f, n = spill_hazards(_kernel(load + ["v_add_u32_e32 v1, v2, v3"] + spill))
    assert n == 1 and list(f) == ["_ZN6vitsmi4demoEv"] and "scratch_store_dwordx4" in f["_ZN6vitsmi4demoEv"][0][1]
    # 2. one register of the quad is enough; a counted wait (vmcnt(2)) proves nothing
    f, _ = spill_hazards(_kernel(load + [";;#ASMSTART", "s_waitcnt vmcnt(2)", ";;#ASMEND", "scratch_store_dword off, v12, off"]))
    assert f
    # 3. behind a full drain of the counter: fine
    f, n = spill_hazards(_kernel(load + [";;#ASMSTART", "s_waitcnt vmcnt(0)", ";;#ASMEND"] + spill))
    assert n == 1 and not f
    # 4. behind its first use (the hand-written wait sits in front of that use): fine
    f, _ = spill_hazards(_kernel(load + ["v_mfma_f32_32x32x16_f16 v[50:65], v[10:13], v[86:89], v[50:65]"] + spill))
    assert not f
    # 5. a load the COMPILER issued (outside an asm block) is tracked by the compiler itself: fine
    f, _ = spill_hazards(_kernel(["global_load_dwordx4 v[10:13], v[2:3], off"] + spill))
    assert not f
    # 6. LDS reads count under lgkmcnt: vmcnt(0) does not cover them, lgkmcnt(0) does
    lds = [";;#ASMSTART", "ds_read_b128 v[20:23], v5 offset:512", ";;#ASMEND"]
    f, _ = spill_hazards(_kernel(lds + ["s_waitcnt vmcnt(0)", "scratch_store_dwordx4 off, v[20:23], off"]))
    assert f
    f, _ = spill_hazards(_kernel(lds + ["s_waitcnt lgkmcnt(0)", "scratch_store_dwordx4 off, v[20:23], off"]))
    assert not f
    # 7. a loop: the load sits at the bottom of the body, the spill at its top - they meet across the back edge
    loop = [".LBB0_1:", "scratch_store_dwordx4 off, v[10:13], off offset:16"] + load + ["s_cbranch_scc1 .LBB0_1"]
    f, _ = spill_hazards(_kernel(loop))
    assert f
    # 8. kernels that do not spill are not examined
    f, n = spill_hazards(_kernel(load + ["v_add_f32_e32 v1, v10, v11"]))
    assert n == 0 and not f
    # 9. control flow is followed, not the text order: the load is issued, a FORWARD branch jumps over the block that would
    # consume it into the middle of a loop whose body spills the register - a finding; with the consuming block on the only
    # path, none
    body = [".LBB0_3:", "scratch_store_dwordx4 off, v[10:13], off offset:16", "s_cbranch_scc1 .LBB0_3"]
    f, _ = spill_hazards(_kernel(load + ["s_cbranch_vccnz .LBB0_3", "v_add_f32_e32 v1, v10, v11", "v_add_f32_e32 v1, v12, v13"] + body))
    assert f
    f, _ = spill_hazards(_kernel(load + ["v_mfma_f32_32x32x16_f16 v[50:65], v[10:13], v[86:89], v[50:65]", "s_cbranch_vccnz .LBB0_3",
                                         "v_add_f32_e32 v1, v2, v3"] + body))
    assert not f
    # 10. two paths meet: in flight on ONE of them is in flight at the join
    f, _ = spill_hazards(_kernel(["s_cbranch_scc0 .LBB0_5"] + load + [".LBB0_5:"] + spill))
    assert f
    # 11. an unconditional branch ends its block: the spill behind it is only reached from the label's other predecessor
    f, _ = spill_hazards(_kernel(load + ["s_branch .LBB0_7", ".LBB0_6:"] + spill + ["s_endpgm", ".LBB0_7:",
                                         ";;#ASMSTART", "s_waitcnt vmcnt(0)", ";;#ASMEND", "s_branch .LBB0_6"]))
    assert not f


def test_name_and_structure_disagreements_are_all_reported_through_the_abi(tmp_path):
    """A graph whose node NAMES put two convs in each other's module (the names win: the file loads, weights bound by name)
    reports every such node through vits_meta("vitsmi.name_warnings") - a C-ABI caller cannot read stderr - and a clean
    file has no such key."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import rename_nodes as rn
    src = open(os.path.join(GOLDEN, "tiny_rb1.onnx"), "rb").read()
    A, B = b"/enc_p/encoder/ffn_layers.0/conv_1/Conv", b"/enc_p/encoder/ffn_layers.1/conv_1/Conv"
    out, seen = bytearray(), 0
    for fn, wt, f0, a, b in rn._fields(src, 0, len(src)):
        if fn != 7 or wt != 2:
            out += src[f0:b]
            continue
        g = bytearray()
        for gfn, gwt, g0, ga, gb in rn._fields(src, a, b):
            if gfn != 1 or gwt != 2:
                g += src[g0:gb]
                continue
            node = bytearray()
            for nfn, nwt, n0, na, nb in rn._fields(src, ga, gb):
                nm = bytes(src[na:nb])
                if nfn == 3 and nwt == 2 and nm in (A, B):
                    node += rn._ld(3, B if nm == A else A)
                    seen += 1
                else:
                    node += src[n0:nb]
            g += rn._ld(1, bytes(node))
        out += rn._ld(7, bytes(g))
    assert seen == 2
    bad = tmp_path / "swapped.onnx"
    bad.write_bytes(bytes(out))
    s = MiSession(str(bad), host_only=True)
    w = s.meta("vitsmi.name_warnings")
    assert w is not None and len(w.strip().splitlines()) == 2 and A.decode() in w and B.decode() in w
    s.close()
    clean = MiSession(os.path.join(GOLDEN, "tiny_rb1.onnx"), host_only=True)
    assert clean.meta("vitsmi.name_warnings") is None
    clean.close()


def test_a_stream_holds_the_session_lock_against_calls_from_other_threads():
    """A chunked run (synthesize_stream / vocoder_stream) owns the handle's workspace until it ends: a call from another
    thread (here: last_y_lengths, the middle one of a batch call's three C calls) must wait for it instead of running between
    its chunks.  Stub library, no GPU: the worker thread of _stream takes MiSession's RLock around the C call."""
    import ctypes as C
    import threading
    import time

    log = []

    class FakeLib:
        def vits_last_y_lengths(self, h, buf, n):
            log.append("Y")
            return 0

    s = object.__new__(MiSession)
    s._mu = threading.RLock()
    s._lib = FakeLib()
    s._h = None

    def start(cb):  # stands for vits_run_chunked: blocks, hands three chunks to the callback
        log.append("S0")
        buf = (C.c_float * 4)(1, 2, 3, 4)
        for i in range(3):
            time.sleep(0.05)
            cb(None, C.cast(buf, C.POINTER(C.c_float)), 1, 4 * i, 4, 12)
        log.append("S1")
        return 0

    got = []
    t = threading.Thread(target=lambda: got.extend(c for c in s._stream(start)))
    t.start()
    time.sleep(0.06)          # the stream is between its chunks now
    s.last_y_lengths()        # ... and this call has to wait for its end
    t.join()
    assert log == ["S0", "S1", "Y"], log
    assert [c[0] for c in got] == [0, 4, 8] and all(np.array_equal(c[1], [[1, 2, 3, 4]]) for c in got)


def test_the_consumer_of_a_stream_can_ask_for_frame_counts_and_nothing_deadlocks():
    """The consumer of synthesize_stream may call last_y_lengths() between chunks (the counts are valid from the first chunk
    on): the worker thread holds the session lock for the whole run, so that call must pass the lock instead of waiting
    for the run it is itself keeping alive.  Calls that need the handle (the engine holds its mutex for the whole chunked
    run) raise instead of deadlocking; a second stream on the same session from the same thread is refused; another thread
    gives up after busy_timeout_s when the generator is abandoned unclosed.  Stub library, no GPU."""
    import ctypes as C
    import threading
    import time

    class FakeLib:
        def vits_last_y_lengths(self, h, buf, n):
            if buf is not None and n >= 2:
                buf[0], buf[1] = 7, 9
            return 2

        def vits_sync(self, h):
            return 0

    s = object.__new__(MiSession)
    s._mu = threading.RLock()
    s._lib = FakeLib()
    s._h = None
    release = threading.Event()

    def start(cb):  # stands for vits_run_chunked: blocks, hands chunks to the callback until told to stop
        buf = (C.c_float * 4)(1, 2, 3, 4)
        for i in range(6):
            if cb(None, C.cast(buf, C.POINTER(C.c_float)), 1, 4 * i, 4, 24):
                break
        release.wait(5)
        return 0

    done = []

    def consume():
        gen = s._stream(start)
        first = next(gen)
        assert first[0] == 0
        assert np.array_equal(s.last_y_lengths(), [7, 9])       # passes the lock the worker holds
        with pytest.raises(SessionError, match="chunked run is in progress"):
            s.sync()                                             # would wait on the handle's mutex for ever
        with pytest.raises(SessionError, match="already in progress"):
            next(s._stream(start))
        done.append("mid")
        # another thread: waits, then gives up (the generator is neither exhausted nor closed)
        s.busy_timeout_s = 0.4
        err = []
        t0 = time.monotonic()

        def other():
            try:
                s.sync()
                err.append("no error")
            except SessionError as e:
                err.append(str(e))
        t = threading.Thread(target=other)
        t.start()
        t.join(5)
        assert not t.is_alive() and "session busy" in err[0] and time.monotonic() - t0 < 4, err
        release.set()
        gen.close()
        done.append("closed")
        s.sync()                                                 # the lock is free again
        done.append("after")

    t = threading.Thread(target=consume)
    t.start()
    t.join(15)
    assert not t.is_alive(), "deadlock: " + repr(done)
    assert done == ["mid", "closed", "after"], done


def test_pinned_pool_drain_releases_idle_blocks_without_waiting_for_the_next_allocation(monkeypatch):
    """_PinnedPool.drain(): blocks whose arrays have died go back to the free lists at once and the excess over the cap is
    released - MiSession.close() and the pageable-result path call it, so an idle process does not keep a large batch's
    result page-locked.  Host allocator stubbed (no GPU here)."""
    import ctypes as C
    import gc
    from phoonnx_amd import session as ses

    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.free.argtypes = [C.c_void_p]
    freed = []

    class Stub:
        def vits_host_alloc(self, n):
            return libc.malloc(n)

        def vits_host_free(self, p):
            freed.append(p)
            libc.free(p)

    monkeypatch.setattr(ses._ffi, "load", lambda: Stub())
    pool = ses._PinnedPool(keep_bytes=3 << 20)
    a = pool.array((1 << 20,), np.float32)      # 4 MiB block
    b = pool.array((1 << 18,), np.float32)      # 1 MiB block
    a[:] = 1.0
    b[:] = 2.0
    del a, b
    gc.collect()
    assert pool.drain() == 1 and len(freed) == 1            # the 4 MiB block exceeds the cap next to nothing: released
    assert pool._idle == 1 << 20                            # the 1 MiB one is kept for reuse
    c = pool.array((1 << 18,), np.float32)
    assert len(freed) == 1                                  # ... and reused
    del c
    gc.collect()
    assert pool.drain(keep_bytes=0) == 1 and pool._idle == 0 and len(freed) == 2


def test_alternate_schedule_deals_whole_passes_to_the_handles_in_turn():
    """PipelinedSession.run_device_steps(alternate=True), stubbed handles (no GPU): pass k runs on part k mod n with the WHOLE
    batch (request-level pipelining); the split schedule gives every part its rows of every pass."""
    from phoonnx_amd.session import PipelinedSession

    class Part:
        def __init__(self, i):
            self.i, self.calls, self.last = i, [], None

        def run_device(self, ids_ptr, lens_ptr, B, T, scales, sid_ptr=None):
            self.calls.append((ids_ptr, lens_ptr, B))
            self.last = np.full(B, 100 * self.i + len(self.calls), np.int64)

        def last_y_lengths(self):
            return self.last

        def sync(self):
            pass

    import threading
    p = object.__new__(PipelinedSession)
    p.parts = [Part(0), Part(1), Part(2)]
    p._mu = threading.RLock()
    B, T = 8, 16
    out = p.run_device_steps(1000, 2000, B, T, None, 7, alternate=True)
    assert [len(q.calls) for q in p.parts] == [3, 2, 2]
    assert all(c == (1000, 2000, B) for q in p.parts for c in q.calls)                 # whole batch, same pointers
    assert [int(out[k, 0]) // 100 for k in range(7)] == [0, 1, 2, 0, 1, 2, 0]          # pass k on part k mod 3
    for q in p.parts:
        q.calls.clear()
    out = p.run_device_steps(1000, 2000, B, T, None, 2)
    bnd = p.bounds(B)
    assert [len(q.calls) for q in p.parts] == [2, 2, 2]
    assert [q.calls[0] for q in p.parts] == [(1000 + bnd[i] * T * 8, 2000 + bnd[i] * 8, bnd[i + 1] - bnd[i]) for i in range(3)]


def test_f16_request_on_a_voice_whose_generator_cannot_run_it_is_refused_with_the_reason():
    """gen_precision="f16" (the reduced-precision vocoder) exists on the split-operand engine only: a voice whose generator
    runs on the f32 engine (a channel count that is not a multiple of 32: the tiny fixtures) refuses the request at open -
    it used to load and silently render with the exact arithmetic.  The fp32-grade requests still load there."""
    path = os.path.join(GOLDEN, "tiny_rb1.onnx")
    with pytest.raises(SessionError, match="f16"):
        MiSession(path, host_only=True, gen_precision="f16")
    for ok in ("f16x3", "bf16x6"):
        s = MiSession(path, host_only=True, gen_precision=ok)
        assert s.hparam("gen_sx") == 0
        s.close()
    s = MiSession(os.path.join(GOLDEN, "sx_rb1.onnx"), host_only=True, gen_precision="f16")
    assert s.hparam("gen_sx") == 1 and s.hparam("gen_nprod") == 1
    s.close()


def test_sgpr_vmem_hazard_check_on_hand_written_isa():
    """phoonnx_amd.build.sgpr_vmem_hazards (DESIGN 5.1g hazard 5): an inline-asm memory instruction whose scalar base was
    written by v_readlane / v_readfirstlane fewer than five wait states earlier is a finding (the compiler inserts the wait
    states only for memory instructions it emits itself); with the wait states, with a scalar-ALU producer, or outside inline
    asm it is not."""
    from phoonnx_amd.build import sgpr_vmem_hazards
    load = [";;#ASMSTART", "global_load_dwordx4 v[2:5], v34, s[4:5] offset:0", ";;#ASMEND"]
    f = sgpr_vmem_hazards(_kernel(["v_readlane_b32 s4, v236, 6", "v_lshlrev_b32_e32 v34, 4, v182", "v_readlane_b32 s5, v236, 7"] + load))
    assert list(f) == ["_ZN6vitsmi4demoEv"] and f["_ZN6vitsmi4demoEv"][0][1] == 0
    f = sgpr_vmem_hazards(_kernel(["v_readfirstlane_b32 s5, v3", "s_nop 1", "v_mov_b32_e32 v1, v2"] + load))
    assert f and f["_ZN6vitsmi4demoEv"][0][1] == 3
    assert not sgpr_vmem_hazards(_kernel(["v_readlane_b32 s5, v236, 7", "s_nop 4"] + load))
    assert not sgpr_vmem_hazards(_kernel(["v_readlane_b32 s5, v236, 7", ";;#ASMSTART", "s_nop 4",
                                          "global_load_dwordx4 v[2:5], v34, s[4:5] offset:0", ";;#ASMEND"]))
    assert not sgpr_vmem_hazards(_kernel(["s_add_u32 s4, s8, s30", "s_addc_u32 s5, s9, 0"] + load))        # scalar-ALU producer
    assert not sgpr_vmem_hazards(_kernel(["v_readlane_b32 s5, v236, 7", "global_load_dwordx4 v[2:5], v34, s[4:5]"]))  # the compiler's own
    assert not sgpr_vmem_hazards(_kernel(["v_readlane_b32 s9, v236, 7"] + load))                             # another register
