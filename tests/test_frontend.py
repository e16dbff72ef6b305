"""Front-end around the hot call vs outputs of the reference's own code
(tests/golden/frontend.json, produced by oracle/gen_frontend_golden.py in the build container):
phonemes_to_ids, VoiceConfig.from_dict dialects, feed construction, post-processing, WAV framing."""
import base64
import dataclasses
import io
import json
import os
import types
import wave

import numpy as np
import pytest

from conftest import GOLDEN

from phoonnx_amd.config import SynthesisConfig, VoiceConfig
from phoonnx_amd.phoneme_ids import (DEFAULT_IPA_PHONEME_ID_MAP, BlankBetween, load_phoneme_ids, load_phoneme_map,
                                     phonemes_to_ids)
from phoonnx_amd.voice import AudioChunk, TTSVoice


@pytest.fixture(scope="module")
def G():
    with open(os.path.join(GOLDEN, "frontend.json"), encoding="utf-8") as f:
        return json.load(f)


def _enc(o):
    if dataclasses.is_dataclass(o):
        return {f.name: _enc(getattr(o, f.name)) for f in dataclasses.fields(o)}
    if isinstance(o, dict):
        return {str(k): _enc(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_enc(v) for v in o]
    if hasattr(o, "value") and not isinstance(o, (int, float)) and type(o) is not str:
        return o.value
    return o


def test_default_id_map_is_the_reference_table(G):
    assert len(DEFAULT_IPA_PHONEME_ID_MAP) == G["default_map_size"] == 161
    for k, v in G["default_map_probe"].items():
        assert DEFAULT_IPA_PHONEME_ID_MAP[k] == v
    assert sorted(v[0] for v in DEFAULT_IPA_PHONEME_ID_MAP.values()) == list(range(161))
    assert phonemes_to_ids(list("hello world")) == G["default_hello_world"]


def test_phonemes_to_ids_grid(G):
    n = 0
    for case in G["phonemes_to_ids"]:
        id_map = None if case["map"] == "default" else G["id_maps"][case["map"]]
        kw = dict(case["kw"])
        kw["blank_between"] = BlankBetween(kw["blank_between"])
        if case["error"]:
            with pytest.raises(Exception) as ei:
                phonemes_to_ids(list(case["phonemes"]), id_map=id_map, **kw)
            assert type(ei.value).__name__ == case["error"], case
        else:
            assert phonemes_to_ids(list(case["phonemes"]), id_map=id_map, **kw) == case["ids"], case
        n += 1
    assert n == len(G["phonemes_to_ids"]) > 2000


def test_voice_config_dialects(G, tmp_path):
    for case in G["voice_config"]:
        cfg = json.loads(json.dumps(case["config"]))
        ptxt = None
        if "phonemes_txt" in case:
            p = tmp_path / "phonemes.txt"
            p.write_text(case["phonemes_txt"], encoding="utf-8")
            ptxt = str(p)
        if case["error"]:
            with pytest.raises(Exception) as ei:
                VoiceConfig.from_dict(cfg, phonemes_txt=ptxt)
            assert type(ei.value).__name__ == case["error"], case["name"]
        else:
            got = _enc(VoiceConfig.from_dict(cfg, phonemes_txt=ptxt))
            assert got == case["result"], (case["name"], got, case["result"])


def test_load_phoneme_files():
    ids = load_phoneme_ids(io.StringIO("# c\n0 _\n1 ^\n\n7 \na 5\n12\n"))
    assert ids == {"_": 0, "^": 1, " ": 7, "a": 5}
    m = load_phoneme_map(io.StringIO("# c\na b c\nx \nq\n"))
    assert m == {"a": ["b", "c"], "x": [" "]}


class _FakeSession:
    """The reference's session duck type: get_inputs() + run(None, feed) -> [B,1,1,S]."""

    def __init__(self, names, audio):
        self.names, self.audio, self.feeds = names, np.asarray(audio, np.float32), []

    def get_inputs(self):
        return [types.SimpleNamespace(name=n) for n in self.names]

    def run(self, _none, feed):
        self.feeds.append(feed)
        return [self.audio.reshape(1, 1, 1, -1)]


class _Phon:
    def add_diacritics(self, text, lang):
        return text

    def phonemize(self, text, lang):
        return [list(s.strip()) for s in text.split(".") if s.strip()]


def _voice(G, names, audio):
    cfg = next(c for c in G["voice_config"] if c["name"] == "phoonnx_raw")["config"]
    return TTSVoice(session=_FakeSession(names, audio), config=VoiceConfig.from_dict(json.loads(json.dumps(cfg))),
                    phonemizer=_Phon())


def test_feeds_postprocessing_and_wav_match_reference(G):
    for case in G["feeds"]:
        voice = _voice(G, case["input_names"], G["fake_audio"])
        syn = SynthesisConfig(**case["syn"])
        chunks = list(voice.synthesize(case["text"], syn))
        assert len(chunks) == case["n_run_calls_synthesize"]          # includes the reference's sentence doubling
        assert len(voice.session.feeds) == len(case["feeds"])
        for got, want in zip(voice.session.feeds, case["feeds"]):
            assert list(got.keys()) == list(want.keys())
            for k in want:
                assert str(got[k].dtype) == want[k]["dtype"] and list(got[k].shape) == want[k]["shape"], k
                assert got[k].tolist() == want[k]["data"], k
        assert [[c.sample_rate, c.sample_width, c.sample_channels] for c in chunks] == case["chunk_meta"]
        # float post-processing and int16 conversion: bit-exact
        assert np.array_equal(chunks[0].audio_float_array, np.asarray(case["chunk_float"][0], np.float32))
        assert chunks[0].audio_int16_bytes == base64.b64decode(case["chunk_int16_b64"][0])
        buf = io.BytesIO()
        with wave.open(buf, "wb") as w:
            voice.synthesize_wav(case["text"], w, syn)
        assert buf.getvalue() == base64.b64decode(case["wav_b64"])    # WAV file byte-for-byte


def test_silent_audio_and_dedupe_option(G):
    voice = _voice(G, ["input", "input_lengths", "scales"], np.zeros(50, np.float32))
    ch = list(voice.synthesize("ab.", None))
    assert len(ch) == G["postprocess"][0]["n"]
    assert ch[0].audio_int16_bytes == base64.b64decode(G["postprocess"][0]["int16_b64"])
    voice.dedupe_sentences = True
    assert len(list(voice.synthesize("ab. ba.", None))) == 2   # declared deviation switch: no doubling


def test_config_from_onnx_metadata():
    # the keys export_onnx.py:335-345 writes; values are strings in metadata_props
    from phoonnx_amd.voice import config_from_metadata
    meta = {"model_type": "vits", "n_speakers": "4", "n_vocab": "130", "sample_rate": "16000", "alphabet": "ipa",
            "phoneme_type": "raw", "phonemizer_model": "", "phoneme_id_map": json.dumps({"a": 1, " ": 2}),
            "has_espeak": "False"}
    vc = VoiceConfig.from_dict(config_from_metadata(meta))
    assert (vc.sample_rate, vc.num_speakers, vc.num_symbols) == (16000, 4, 130)
    assert vc.phoneme_id_map == {"a": 1, " ": 2} and vc.include_whitespace is True
    assert vc.phoneme_type.value == "raw" and vc.phonemizer_model is None
    with pytest.raises(ValueError):
        config_from_metadata({})
    with pytest.raises(ValueError):
        config_from_metadata({"phoneme_id_map": "{not json"})


def test_audio_chunk_int16():
    c = AudioChunk(22050, 2, 1, np.array([0.0, 1.0, -1.0, 0.5, 2.0, -3.0], np.float32))
    assert c.audio_int16_array.tolist() == [0, 32767, -32767, 16383, 32767, -32767]


def test_builtin_phonemizers_and_missing_id_map():
    from phoonnx_amd.config import PhonemeType
    from phoonnx_amd.phonemizers import get_phonemizer
    # base.py:70 marks EVERY chunk end-of-sentence (`results += [(phoneme_str, punct, True)]`): one entry per chunk
    assert get_phonemizer(PhonemeType.RAW).phonemize("ab, c. de!", "en") == [list("ab"), list("c"), list("de")]
    # like the reference, punctuation is stripped from a chunk before its phonemize_string (base.py:64)
    assert get_phonemizer(PhonemeType.GRAPHEMES).phonemize("Hello  World.", "en") == [list("hello world")]
    assert get_phonemizer(PhonemeType.UNICODE).phonemize("é", "pt") == [["e", "́"]]
    with pytest.raises(ValueError):
        get_phonemizer(PhonemeType.ESPEAK)
    v = TTSVoice(session=_FakeSession(["input"], [0.1]), config=VoiceConfig(
        num_symbols=1, num_speakers=1, num_langs=1, sample_rate=16000, lang_code=None, phoneme_id_map=None,
        phoneme_type=PhonemeType.RAW, alphabet=None, phonemizer_model=None))
    assert v.config.lang_code == "und"
    with pytest.raises(ValueError):
        v.phonemes_to_ids(["a"])


def test_consistency_validation_of_json_metadata_and_graph():
    """f3: JSON, .onnx metadata_props (export_onnx.py:335-350) and the graph must agree; only stated values count."""
    from phoonnx_amd.voice import check_consistency
    meta = {"n_speakers": "4", "n_vocab": "130", "sample_rate": "22050"}
    ok = {"num_speakers": 4, "num_symbols": 130, "audio": {"sample_rate": 22050}, "phoneme_id_map": {"a": 1, "b": [129]}}
    check_consistency(ok, meta, 4, 130)
    check_consistency({"phoneme_id_map": {"a": 1}}, meta, 4, 130)            # a JSON that states nothing: fine
    check_consistency(ok, {}, 4, 130)                                          # no metadata: fine
    check_consistency({"num_speakers": 1}, {"n_speakers": "1"}, 1, 130)       # single-speaker graphs have no table
    for bad, msg in ((dict(ok, num_speakers=2), "num_speakers=2"),
                     (dict(ok, audio={"sample_rate": 16000}), "sample_rate"),
                     (dict(ok, num_symbols=256), "num_symbols=256"),
                     (dict(ok, phoneme_id_map={"a": 130}), "id 130"),
                     (dict(ok, phoneme_id_map={"a": [3, 400]}), "id 400")):
        with pytest.raises(ValueError, match=msg):
            check_consistency(bad, meta, 4, 130)
    with pytest.raises(ValueError, match="speaker table has 2 rows"):
        check_consistency(ok, meta, 2, 130)
