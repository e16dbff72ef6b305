"""SURVEY §8 f4 (ByT5 G2P): the NumPy restatement of the T5 graph (oracle/t5_oracle.py) against outputs of the Hugging Face
transformers model the fixture was exported from (oracle/gen_g2p_golden.py), and the text <-> id mirror against the outputs
of the reference's own `_encode_text` / `_decode_phones` (mul.py:135-170)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(GOLDEN, "byt5_tiny.npz"))


def test_relative_position_buckets_are_exact(G):
    from t5_oracle import relative_position_bucket
    assert np.array_equal(relative_position_bucket(G["bucket/rel"], True), G["bucket/enc"])
    assert np.array_equal(relative_position_bucket(G["bucket/rel"], False), G["bucket/dec"])


def test_t5_oracle_matches_transformers(G):
    from t5_oracle import T5Oracle
    o = T5Oracle(os.path.join(GOLDEN, "byt5_tiny.onnx"))
    hp = json.load(open(os.path.join(GOLDEN, "byt5_tiny.hparams.json")))
    assert (o.d_model, o.heads, o.d_kv, o.n_enc, o.n_dec, o.num_buckets) == (
        hp["d_model"], hp["num_heads"], hp["d_kv"], hp["num_layers"], hp["num_decoder_layers"],
        hp["relative_attention_num_buckets"])
    assert not o.tied
    for c in range(4):
        ids = G[f"c{c}/input_ids"][0]
        enc = o.encode(ids)
        np.testing.assert_allclose(enc, G[f"c{c}/encoder_out"][0], atol=2e-5, rtol=0)
        lg = o.logits(ids, G[f"c{c}/decoder_input_ids"][0])
        np.testing.assert_allclose(lg, G[f"c{c}/logits"], atol=2e-4, rtol=0)
        want = G[f"c{c}/greedy"].tolist()
        assert o.greedy(ids, max_length=len(want)) == want     # integer work: exact (mul.py:192-230)


def test_text_mirror_matches_reference_functions():
    from phoonnx_amd.g2p import decode_phones, encode_text
    F = json.load(open(os.path.join(GOLDEN, "byt5_frontend.json"), encoding="utf-8"))
    for c in F["encode"]:
        got = encode_text(c["text"], c["lang"])
        assert got.dtype == np.int64 and got.shape == (1, len(c["ids"])) and got[0].tolist() == c["ids"]
    tokens = F["tokenizer_config"]["added_tokens_decoder"]
    for c in F["decode"]:
        assert decode_phones(c["ids"], tokens) == c["text"]


def test_g2p_abi_surface_and_model_description(G):
    """CPU side of the product: g2pmi.h symbols exported, the C++ reader derives the T5 geometry from the .onnx, and its
    relative-position buckets (integers) equal the transformers function's over the whole supported range."""
    import re
    from conftest import ROOT
    from phoonnx_amd import _ffi
    from phoonnx_amd.g2p import MiG2PSession
    from phoonnx_amd.session import SessionError
    hdr = open(os.path.join(ROOT, "include", "g2pmi.h")).read()
    declared = set(re.findall(r"\b(g2p_[a-z0-9_]+)\s*\(", hdr))
    lib = _ffi.load()
    assert declared == set(_ffi.G2P_EXPORTS), declared ^ set(_ffi.G2P_EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    s = MiG2PSession(os.path.join(GOLDEN, "byt5_tiny.onnx"), host_only=True)
    hp = json.load(open(os.path.join(GOLDEN, "byt5_tiny.hparams.json")))
    assert s.hparam("vocab") == hp["vocab_size"] and s.hparam("d_model") == hp["d_model"]
    assert s.hparam("heads") == hp["num_heads"] and s.hparam("d_kv") == hp["d_kv"] and s.hparam("d_ff") == hp["d_ff"]
    assert s.hparam("n_enc") == hp["num_layers"] and s.hparam("n_dec") == hp["num_decoder_layers"]
    assert s.hparam("num_buckets") == hp["relative_attention_num_buckets"]
    assert s.hparam("gated") == 1 and s.hparam("act") == 0 and s.hparam("scale_out") == 0    # gated-gelu (tanh form), untied
    assert [o.name for o in s.get_outputs()] == ["logits"]
    rel = G["bucket/rel"]
    assert [s.bucket(int(r)) for r in rel] == G["bucket/enc"].tolist()
    assert [s.bucket(int(r), decoder=True) for r in rel] == G["bucket/dec"].tolist()
    with pytest.raises(SessionError):   # host-only handles cannot run
        s.run(None, {"input_ids": np.array([[5, 6]], np.int64), "decoder_input_ids": np.array([[0]], np.int64)})
    s.close()
    with pytest.raises(SessionError):   # a VITS graph is not a T5 graph
        MiG2PSession(os.path.join(GOLDEN, "tiny_dp.onnx"), host_only=True)


def test_language_tags_resolve_like_the_reference_match_lang():
    """TTSVoice hands BCP-47 tags (config.lang_code) to the phonemizer; the reference resolves them to the
    phonemizer's own list through langcodes (phonemizers/base.py:86-122).  Expected values = langcodes' answers for the
    reference's lists (mul.py:31-33, 248-256)."""
    from phoonnx_amd.g2p import ByT5Phonemizer as B, CharsiuPhonemizer as Ch
    byt5 = {"en-US": "en-US", "en": "en-US", "en-GB": "en-GB", "en-AU": "en-GB", "pt": "pt-BR", "pt-PT": "pt-PT",
            "de": "de-DE", "de-AT": "de-DE", "zh": "zh-CN", "no": "nb-NO", "yue": "yue-CN", "fr-CA": "fr-FR"}
    charsiu = {"en-US": "eng-us", "en": "eng-us", "en-GB": "eng-uk", "pt-BR": "por-bz", "pt": "por-bz", "pt-PT": "por-po",
               "de": "ger", "de-DE": "ger", "nl-NL": "dut", "cs": "cze", "el-GR": "gre", "zh-TW": "zho-t", "zh-CN": "zho-s",
               "es-MX": "spa-me", "es-AR": "spa-latin", "es": "spa", "fr-CA": "fra-qu", "fr": "fra", "eng-us": "eng-us",
               "fa-IR": "fas", "is": "ice", "cy": "wel-nw", "vi": "vie-n"}
    for t, want in byt5.items():
        assert B.get_lang(t) == want, (t, B.get_lang(t), want)
    for t, want in charsiu.items():
        assert Ch.get_lang(t) == want, (t, Ch.get_lang(t), want)
    for bad in ("xx-YY", "tlh", ""):
        for cls in (B, Ch):
            with pytest.raises(ValueError):
                cls.get_lang(bad)
