"""GPU parity of the ByT5 G2P engine through the C ABI (include/g2pmi.h): logits against the outputs of the
transformers model the fixture was exported from and against the NumPy oracle on fresh inputs; generated ids (integer
work) exact; the device-side greedy loop against the reference's call-by-call loop."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

LOGIT_TOL = 2e-3   # logits are O(30); fp32 chains through 5 blocks: measured ~3e-5


@pytest.fixture(scope="module")
def sess():
    from phoonnx_amd.g2p import MiG2PSession
    s = MiG2PSession(os.path.join(GOLDEN, "byt5_tiny.onnx"))
    yield s
    s.close()


def test_g2p_logits_match_transformers_goldens(sess):
    G = np.load(os.path.join(GOLDEN, "byt5_tiny.npz"))
    for c in range(4):
        ids, dec = G[f"c{c}/input_ids"], G[f"c{c}/decoder_input_ids"]
        out = sess.run(["logits"], {"input_ids": ids, "attention_mask": np.ones_like(ids), "decoder_input_ids": dec})
        assert isinstance(out, list) and out[0].dtype == np.float32 and out[0].shape == G[f"c{c}/logits"].shape
        err = float(np.abs(out[0] - G[f"c{c}/logits"]).max())
        assert err < LOGIT_TOL, (c, err)
        assert np.array_equal(out[0].argmax(-1), G[f"c{c}/logits"].argmax(-1))


def test_g2p_generate_matches_reference_greedy_loop(sess):
    """Integer work: the ids the greedy loop of mul.py:192-230 produces are exact - by the device-side loop with its KV
    cache, and by the reference's own call sequence through session.run."""
    G = np.load(os.path.join(GOLDEN, "byt5_tiny.npz"))
    for c in range(4):
        ids, want = G[f"c{c}/input_ids"], G[f"c{c}/greedy"].tolist()
        assert sess.generate(ids[0], max_length=len(want)) == want
    # the call-by-call loop (what the reference executes), first 12 tokens of one case
    ids, want = G["c0/input_ids"], G["c0/greedy"].tolist()[:12]
    dec, gen = np.array([[0]], np.int64), []
    for _ in range(len(want)):
        lg = sess.run(None, {"input_ids": ids, "attention_mask": np.ones_like(ids), "decoder_input_ids": dec})[0]
        gen.append(int(np.argmax(lg[0, -1])))
        dec = np.concatenate((dec, np.array([[gen[-1]]], np.int64)), axis=1)
    assert gen == want


def test_persistent_decoder_step_generates_the_same_ids(sess, monkeypatch):
    """The opt-in one-launch decoder step (g2p.hip g2p_decode_step_kernel: the phases' bodies behind grid barriers) is the
    launch-per-phase path's arithmetic: same ids for one sequence and for batches of 2 and 3 (NB 2 and 4)."""
    G = np.load(os.path.join(GOLDEN, "byt5_tiny.npz"))
    monkeypatch.setenv("VITSMI_G2P_PERSIST", "1")
    for c in range(4):
        ids, want = G[f"c{c}/input_ids"], G[f"c{c}/greedy"].tolist()
        assert sess.generate(ids[0], max_length=len(want)) == want
    many = [G[f"c{c}/input_ids"][0] for c in range(3)]
    for nb in (2, 3):
        got = sess.generate_batch(many[:nb], max_length=24)
        monkeypatch.setenv("VITSMI_G2P_PERSIST", "0")
        assert got == sess.generate_batch(many[:nb], max_length=24)
        monkeypatch.setenv("VITSMI_G2P_PERSIST", "1")


def test_g2p_matches_oracle_on_fresh_inputs(sess):
    from t5_oracle import T5Oracle
    o = T5Oracle(os.path.join(GOLDEN, "byt5_tiny.onnx"))
    rng = np.random.default_rng(5)
    for S, T in ((1, 1), (7, 3), (130, 40), (300, 9)):
        ids = rng.integers(3, 259, (1, S)).astype(np.int64)
        dec = np.concatenate(([0], rng.integers(3, 259, T - 1))).astype(np.int64)[None]
        got = sess.run(None, {"input_ids": ids, "decoder_input_ids": dec})[0]
        ref = o.logits(ids[0], dec[0])
        assert float(np.abs(got - ref).max()) < LOGIT_TOL, (S, T)
    ids = rng.integers(3, 259, 57).astype(np.int64)
    assert sess.generate(ids, max_length=40) == o.greedy(ids, max_length=40)


@pytest.fixture(scope="module")
def sess64():
    """Heads of width 64 (every ByT5 / mT5 size): the engine's attention has a body of its own for that width
    (g2p.hip g2p_attention_body_t<64>) which the 16-wide tiny model never reaches.  oracle/gen_g2p_golden.py --name byt5_dk64."""
    from phoonnx_amd.g2p import MiG2PSession
    s = MiG2PSession(os.path.join(GOLDEN, "byt5_dk64.onnx"))
    yield s
    s.close()


def test_g2p_dk64_logits_match_transformers_goldens(sess64):
    G = np.load(os.path.join(GOLDEN, "byt5_dk64.npz"))
    assert sess64.hparam("d_kv") == 64
    for c in range(3):   # (case 2: 321 input bytes - cross attention over more keys than the 256 the kernel prefetches)
        ids, dec = G[f"c{c}/input_ids"], G[f"c{c}/decoder_input_ids"]
        out = sess64.run(["logits"], {"input_ids": ids, "attention_mask": np.ones_like(ids), "decoder_input_ids": dec})[0]
        err = float(np.abs(out - G[f"c{c}/logits"]).max())
        assert err < LOGIT_TOL, (c, err)
        assert sess64.generate(ids[0], max_length=len(G[f"c{c}/greedy"]), eos_id=-1) == G[f"c{c}/greedy"].tolist()


@pytest.mark.parametrize("which", ["tiny", "dk64"])
def test_the_decoder_step_path_computes_the_logits_of_the_whole_graph(which, sess, sess64):
    """The greedy ids of a randomly initialised T5 are nearly constant sequences - equal ids say little about the step
    path's arithmetic.  g2p_test_forced_steps drives the SAME kernels `generate` runs per token (matrix-vector products with
    the norm and the gate folded in, one-query attention over the key / value caches) with given decoder inputs and returns
    every step's logits: they must be the logits of the whole graph for the same prefix - `run` (pinned to the transformers
    model above) and, on fresh inputs, the NumPy oracle.  Covered: both attention bodies (head width 16 and 64), more than
    256 cached keys in self attention (300 steps) and in cross attention (321 / 300 input bytes), 1, 2, 3 and 4 sequences side
    by side (the NB = 1 / 2 / 4 kernels, padded encoder batch with per-sequence key counts), and the one-launch step."""
    from t5_oracle import T5Oracle
    s = sess if which == "tiny" else sess64
    name = "byt5_tiny" if which == "tiny" else "byt5_dk64"
    G = np.load(os.path.join(GOLDEN, name + ".npz"))
    rng = np.random.default_rng(11)
    # 1. the fixture's own cases: the transformers model's logits
    for c in range(3):
        ids, dec = G[f"c{c}/input_ids"], G[f"c{c}/decoder_input_ids"]
        got = s.forced_step_logits([ids[0]], dec)
        assert got.shape == G[f"c{c}/logits"].shape
        assert float(np.abs(got - G[f"c{c}/logits"]).max()) < LOGIT_TOL, (which, c)
    # 2. long random prefixes against run() (the whole graph on the prefill kernels) and the oracle
    o = T5Oracle(os.path.join(GOLDEN, name + ".onnx"))
    for S, T in ((5, 1), (300, 40), (64, 300)):
        ids = rng.integers(3, 259, S).astype(np.int64)
        dec = np.concatenate(([0], rng.integers(3, 259, T - 1))).astype(np.int64)[None]
        got = s.forced_step_logits([ids], dec)[0]
        ref = s.run(None, {"input_ids": ids[None], "decoder_input_ids": dec})[0][0]
        assert float(np.abs(got - ref).max()) < LOGIT_TOL, (which, S, T, float(np.abs(got - ref).max()))
        if T <= 40:
            assert float(np.abs(got - o.logits(ids, dec[0])).max()) < LOGIT_TOL, (which, S, T)
    # 3. several sequences of different lengths side by side = each one alone
    seqs = [rng.integers(3, 259, n).astype(np.int64) for n in (7, 130, 33, 270)]
    decs = np.concatenate((np.zeros((4, 1), np.int64), rng.integers(3, 259, (4, 23))), axis=1)
    alone = [s.forced_step_logits([x], decs[i:i + 1])[0] for i, x in enumerate(seqs)]
    for nb in (2, 3, 4):
        got = s.forced_step_logits(seqs[:nb], decs[:nb])
        for i in range(nb):
            assert float(np.abs(got[i] - alone[i]).max()) < 1e-4, (which, nb, i)


def test_the_one_launch_decoder_step_computes_the_same_logits(sess64, monkeypatch):
    rng = np.random.default_rng(12)
    ids = rng.integers(3, 259, 90).astype(np.int64)
    dec = np.concatenate(([0], rng.integers(3, 259, 30))).astype(np.int64)[None]
    want = sess64.forced_step_logits([ids], dec)
    monkeypatch.setenv("VITSMI_G2P_PERSIST", "1")
    got = sess64.forced_step_logits([ids], dec)
    assert np.array_equal(got, want)   # (the same bodies in the same order: bit for bit)


def test_g2p_bad_inputs_raise(sess):
    from phoonnx_amd.session import SessionError
    ok = {"input_ids": np.array([[10, 11]], np.int64), "decoder_input_ids": np.array([[0]], np.int64)}
    sess.run(None, ok)
    with pytest.raises(SessionError):
        sess.run(None, dict(ok, input_ids=np.array([[10, 5000]], np.int64)))
    # a right-padded mask = the unpadded input; a mask with a hole (or nothing left) is refused
    padded = dict(ok, input_ids=np.array([[10, 11, 0, 0]], np.int64), attention_mask=np.array([[1, 1, 0, 0]], np.int64))
    assert np.array_equal(sess.run(None, padded)[0], sess.run(None, ok)[0])
    with pytest.raises(SessionError):
        sess.run(None, dict(ok, attention_mask=np.array([[0, 1]], np.int64)))
    with pytest.raises(SessionError):
        sess.run(None, dict(ok, attention_mask=np.array([[0, 0]], np.int64)))
    with pytest.raises(SessionError):
        sess.run(None, dict(ok, bogus=np.zeros(1)))
    with pytest.raises(SessionError):
        sess.run(None, {"input_ids": ok["input_ids"]})


def test_byt5_phonemizer_mirror(tmp_path):
    from phoonnx_amd.g2p import ByT5Phonemizer
    F = json.load(open(os.path.join(GOLDEN, "byt5_frontend.json"), encoding="utf-8"))
    cfg = tmp_path / "tokenizer_config.json"
    cfg.write_text(json.dumps(F["tokenizer_config"]))
    p = ByT5Phonemizer(os.path.join(GOLDEN, "byt5_tiny.onnx"), str(cfg))
    a = p._infer("hello world", "en-US", max_length=24)
    p.device_loop = False
    b = p._infer("hello world", "en-US", max_length=24)
    assert a == b and isinstance(a, str)           # device loop == the reference's call-by-call loop
    assert p.phonemize_string("   ", "en-US") == ""
    assert p.get_lang("pt") == "pt-BR" or p.get_lang("pt").startswith("pt")
    p.session.close()


def test_charsiu_phonemizer_goes_word_by_word(tmp_path):
    from phoonnx_amd.g2p import ByT5Phonemizer, CharsiuPhonemizer
    path = os.path.join(GOLDEN, "byt5_tiny.onnx")
    F = json.load(open(os.path.join(GOLDEN, "byt5_frontend.json"), encoding="utf-8"))
    cfg = tmp_path / "tokenizer_config.json"    # (a random-weight model emits special ids: they must be known as such)
    added = dict(F["tokenizer_config"]["added_tokens_decoder"])
    added.update({str(i): {"content": f"<extra_id_{i - 259}>"} for i in range(259, 384)})   # ByT5's 125 sentinel ids
    cfg.write_text(json.dumps({"added_tokens_decoder": added}))
    c = CharsiuPhonemizer(path, str(cfg))
    words = "one two".split()
    c_out = c._infer
    got = c.phonemize_string("one  two", "eng-us")
    assert got == " ".join(c_out(w, "eng-us") for w in words)       # mul.py:284-286
    with pytest.raises(ValueError):
        c.get_lang("xx-YY")
    c.session.close()


def test_ttsvoice_with_the_byt5_phonemizer_end_to_end(tmp_path):
    """f4 behind the boundary: a voice whose JSON says phoneme_type "byt5" + a local phonemizer_model gets its phonemes
    from the G2P engine (config.py:405-406 -> mul.py), then its audio from the VITS engine: text -> wav, both on the GPU."""
    import shutil
    from phoonnx_amd.config import SynthesisConfig
    from phoonnx_amd.g2p import ByT5Phonemizer
    from phoonnx_amd.voice import TTSVoice
    F = json.load(open(os.path.join(GOLDEN, "byt5_frontend.json"), encoding="utf-8"))
    g2p_dir = tmp_path / "g2p"
    g2p_dir.mkdir()
    shutil.copy(os.path.join(GOLDEN, "byt5_tiny.onnx"), g2p_dir / "model.onnx")
    added = dict(F["tokenizer_config"]["added_tokens_decoder"])
    added.update({str(i): {"content": f"<extra_id_{i - 259}>"} for i in range(259, 384)})
    (g2p_dir / "tokenizer_config.json").write_text(json.dumps({"added_tokens_decoder": added}))
    model = tmp_path / "voice.onnx"
    shutil.copy(os.path.join(GOLDEN, "tiny_rb1.onnx"), model)
    # (a random-weight G2P emits arbitrary bytes: map every printable latin-1 character, the rest is skipped with a
    # warning as in the reference, phoneme_ids.py)
    id_map = {"_": 0, "^": 1, "$": 2, " ": 3}
    for c in range(33, 256):
        if chr(c) not in id_map and len(id_map) < 60:
            id_map[chr(c)] = len(id_map)
    (tmp_path / "voice.onnx.json").write_text(json.dumps({
        "phoneme_type": "byt5", "phonemizer_model": str(g2p_dir / "model.onnx"), "lang_code": "en-US", "alphabet": "ipa",
        "audio": {"sample_rate": 22050}, "phoneme_id_map": id_map, "pad": "_", "blank": "_", "bos": "^", "eos": "$",
        "inference": {"noise_scale": 0.0, "length_scale": 1.5, "noise_w": 0.0}}))
    voice = TTSVoice.load(str(model))
    assert isinstance(voice.phonemizer, ByT5Phonemizer) or voice.phonemizer is None
    voice.dedupe_sentences = True
    # (what a random-weight model says is arbitrary: for "testing the engine" this one repeats the byte "9" until
    # max_length, for "again" only special ids, i.e. no phonemes - oracle/t5_oracle.py greedy says the same)
    text = "testing the engine. again, testing the engine"
    sentences = voice.phonemize(text)
    assert isinstance(voice.phonemizer, ByT5Phonemizer)
    assert len(sentences) == 3 and sentences[1] == [] and set(sentences[0]) == {"9"} and sentences[2] == sentences[0]
    # the phonemes are what the G2P engine's greedy loop says for each chunk, character by character
    assert sentences[0] == list(voice.phonemizer.phonemize_string("testing the engine", "en-US"))
    chunks = list(voice.synthesize(text, SynthesisConfig(noise_scale=0.0, noise_w_scale=0.0)))
    assert len(chunks) == 2 and all(c.sample_rate == 22050 for c in chunks)     # (an entry without phonemes is skipped)
    assert np.array_equal(chunks[0].audio_float_array, chunks[1].audio_float_array)
    assert all(np.isfinite(c.audio_float_array).all() and len(c.audio_float_array) > 0 for c in chunks)
    voice.phonemizer.session.close()
    voice.session.close()


def test_generate_batch_equals_one_call_each(sess):
    """g2p_generate_batch: padded batch, masked attention, shared weight stream - each sequence's ids are exactly those of
    a call of its own (the arithmetic per sequence is the same, only the neighbours differ)."""
    rng = np.random.default_rng(11)
    for B in (1, 2, 3, 5, 8, 11):
        inputs = [rng.integers(3, 259, int(n)).astype(np.int64) for n in rng.integers(4, 60, B)]
        singles = [sess.generate(x, max_length=24) for x in inputs]
        assert sess.generate_batch(inputs, max_length=24) == singles, B
    # more inputs than one engine call takes; eos handling: with eos_id = the first token every sequence stops at once
    inputs = [rng.integers(3, 259, int(n)).astype(np.int64) for n in rng.integers(4, 40, 19)]
    assert sess.generate_batch(inputs, max_length=12) == [sess.generate(x, max_length=12) for x in inputs]
    first = sess.generate(inputs[0], max_length=4)[0]
    assert sess.generate_batch(inputs[:3], max_length=8, eos_id=first)[0] == [first]
    from phoonnx_amd.session import SessionError
    with pytest.raises(SessionError):
        sess.generate_batch([np.array([5000], np.int64)], max_length=4)


def test_phonemize_batches_the_chunks(tmp_path):
    from phoonnx_amd.g2p import ByT5Phonemizer, CharsiuPhonemizer
    F = json.load(open(os.path.join(GOLDEN, "byt5_frontend.json"), encoding="utf-8"))
    added = dict(F["tokenizer_config"]["added_tokens_decoder"])
    added.update({str(i): {"content": f"<extra_id_{i - 259}>"} for i in range(259, 384)})
    cfg = tmp_path / "tokenizer_config.json"
    cfg.write_text(json.dumps({"added_tokens_decoder": added}))
    path = os.path.join(GOLDEN, "byt5_tiny.onnx")
    text = "one two, three. four; five six seven"
    for cls, lang in ((ByT5Phonemizer, "en-US"), (CharsiuPhonemizer, "eng-us")):
        p = cls(path, str(cfg))
        batched = p.phonemize(text, lang)
        p.device_loop = False          # the reference's call-by-call loop, chunk after chunk
        assert p.phonemize(text, lang) == batched and len(batched) == 4
        p.session.close()


def test_generate_batch_argument_checks_and_limits(sess):
    """g2p_generate_batch refuses what it cannot do (no clamping, no partial results), and works at its limits."""
    import ctypes as C
    from phoonnx_amd import _ffi
    from phoonnx_amd.session import SessionError
    lib = sess._lib
    ids = np.arange(3, 23, dtype=np.int64)
    out = np.zeros((65, 8), np.int64)
    n = np.zeros(65, np.int32)

    def call(lens, B, max_length=8, start=0):
        lens = np.asarray(lens, np.int32)
        return lib.g2p_generate_batch(sess._h, _ffi.ptr(ids), _ffi.ptr(lens), B, max_length, start, 1, _ffi.ptr(out), _ffi.ptr(n))
    assert call([20], 1) == 0
    assert call([20], 0) < 0 and call([0], 1) < 0 and call([-3], 1) < 0          # no sequences / empty / negative length
    assert call([1] * 65, 65) < 0                                                  # more than G2P_MAX_BATCH
    assert call([20], 1, max_length=0) < 0 and call([20], 1, max_length=1024) < 0  # decoder positions: 1 .. 1023
    assert call([20], 1, start=sess.hparam("vocab")) < 0
    assert b"" != lib.g2p_last_error(sess._h)
    # 64 one-byte inputs side by side, and one input of the longest length the tables cover
    many = [np.array([3 + i], np.int64) for i in range(64)]
    assert sess.generate_batch(many, max_length=3) == [sess.generate(x, max_length=3) for x in many]
    long_in = np.random.default_rng(5).integers(3, 259, 1023).astype(np.int64)
    assert sess.generate_batch([long_in, ids], max_length=4) == [sess.generate(long_in, max_length=4), sess.generate(ids, max_length=4)]
    with pytest.raises(SessionError):
        sess.generate(np.zeros(1024, np.int64) + 5, max_length=2)
