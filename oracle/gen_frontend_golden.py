#!/usr/bin/env python3
"""Front-end golden generator.  TEST INFRASTRUCTURE — runs only in the build container.

Imports the reference's `phoonnx.phoneme_ids`, `phoonnx.config` and `phoonnx.voice` (the
latter with stub modules for the dependencies this image lacks) and
records, as JSON, what the parts of the front-end that surround the hot path do:
  * phonemes_to_ids for a grid of options                     (phoneme_ids.py:209-310)
  * VoiceConfig.from_dict for the piper / mimic3 / coqui / cotovia / phoonnx dialects
                                                              (config.py:218-358)
  * the feed dict TTSVoice.phoneme_ids_to_audio builds        (voice.py:347-373)
  * synthesize() post-processing + int16 conversion           (voice.py:271-282, 88-91)
  * synthesize_wav() framing                                   (voice.py:307-326)
Only inputs and outputs are stored (tests/golden/frontend.json), never source text.
"""
import base64
import dataclasses
import io
import json
import os
import sys
import types
import wave

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "frontend.json")


def stub_modules():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Sess:
        def __init__(self, *a, **k):
            pass

    mod("onnxruntime", InferenceSession=_Sess, SessionOptions=lambda: None)
    mod("langcodes", closest_match=lambda lang, langs: (langs[0] if langs else "und", 1000),
        tag_distance=lambda a, b: 0 if a == b else 100)
    mod("quebra_frases", sentence_tokenize=lambda t: [s.strip() for s in
                                                     __import__("re").split(r"(?<=[.!?])\s+", t) if s.strip()])
    mod("ovos_date_parser", nice_time=lambda *a, **k: "", nice_date=lambda *a, **k: "")
    mod("ovos_number_parser", pronounce_number=lambda *a, **k: "", is_fractional=lambda *a, **k: False,
        pronounce_fraction=lambda *a, **k: "")
    mod("ovos_number_parser.util", is_numeric=lambda s: False)

    class _Rbnf:
        @staticmethod
        def for_language(lang):
            raise ValueError("no rbnf")

    mod("unicode_rbnf", RbnfEngine=_Rbnf, FormatPurpose=types.SimpleNamespace(CARDINAL=0, ORDINAL=1, YEAR=2))


def enc(o):
    if dataclasses.is_dataclass(o):
        return {f.name: enc(getattr(o, f.name)) for f in dataclasses.fields(o)}
    if isinstance(o, dict):
        return {str(k): enc(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [enc(v) for v in o]
    if hasattr(o, "value") and not isinstance(o, (int, float, str)):
        return o.value
    if isinstance(o, (np.integer,)):
        return int(o)
    if isinstance(o, (np.floating,)):
        return float(o)
    if isinstance(o, str) and type(o) is not str:  # str-Enum
        return str(getattr(o, "value", o))
    return o


def main():
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference")
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    stub_modules()
    from phoonnx import phoneme_ids as P
    from phoonnx import config as Cfg
    import phoonnx.voice as V

    out = {"phonemes_to_ids": [], "voice_config": [], "feeds": [], "postprocess": [], "wav": []}

    # ---- phonemes_to_ids
    id_map_list = {"_": [0], "^": [1], "$": [2], " ": [3], "a": [4], "b": [5], "c": [6], "ab": [7, 8], "ˈ": [9]}
    id_map_int = {"_": 0, "^": 1, "$": 2, " ": 3, "a": 4, "b": 5, "c": 6}
    id_map_nows = {"_": 0, "^": 1, "$": 2, "#": 3, "a": 4, "b": 5, "c": 6}
    phon_cases = [list("abc"), list("a b c"), list("ab c"), list("xaxb"), [], list("a"), list("ˈab  c"), list(" a ")]
    for pm_name, pm in (("list", id_map_list), ("int", id_map_int), ("nows", id_map_nows), ("default", None)):
        for ph in phon_cases:
            for bb in (P.BlankBetween.TOKENS, P.BlankBetween.WORDS, P.BlankBetween.TOKENS_AND_WORDS):
                for inc_ws in (True, False):
                    for toks in ((("_", "^", "$"),), ((None, "^", "$"),), (("_", None, None),), ((None, None, None),)):
                        blank, bos, eos = toks[0]
                        for start, end in ((True, True), (False, True), (True, False)):
                            kw = dict(id_map=pm, blank_token=blank, bos_token=bos, eos_token=eos,
                                      word_sep_token="#" if pm_name == "nows" else " ", include_whitespace=inc_ws,
                                      blank_at_start=start, blank_at_end=end, blank_between=bb)
                            try:
                                res = P.phonemes_to_ids(list(ph), **kw)
                                err = None
                            except Exception as e:  # the reference raises KeyError for some combinations
                                res, err = None, type(e).__name__
                            out["phonemes_to_ids"].append({"map": pm_name, "phonemes": ph,
                                                           "kw": {k: enc(v) for k, v in kw.items() if k != "id_map"},
                                                           "ids": res, "error": err})
    out["id_maps"] = {"list": id_map_list, "int": id_map_int, "nows": id_map_nows}
    out["default_map_size"] = len(P.DEFAULT_IPA_PHONEME_ID_MAP)
    out["default_map_probe"] = {k: P.DEFAULT_IPA_PHONEME_ID_MAP[k] for k in ["_", "^", "$", " ", "a", "ə", "ʷ", "g"]}
    out["default_hello_world"] = P.phonemes_to_ids(list("hello world"))

    # ---- VoiceConfig.from_dict dialects
    cfgs = {
        "piper_espeak": {"piper_version": "1.0.0", "phoneme_type": "espeak", "audio": {"sample_rate": 22050},
                         "espeak": {"voice": "en-us"}, "language": {"code": "en_US"},
                         "inference": {"noise_scale": 0.5, "length_scale": 1.1, "noise_w": 0.7},
                         "phoneme_id_map": {"_": [0], "^": [1], "$": [2], " ": [3], "a": [4]}, "num_symbols": 130,
                         "num_speakers": 2, "speaker_id_map": {"x": 0, "y": 1}},
        "piper_text": {"phoneme_type": "text", "audio": {"sample_rate": 16000}, "espeak": {"voice": "ar"},
                       "phoneme_id_map": {"_": [0], "^": [1], "$": [2], "a": [4]}},
        "phoonnx_raw": {"phoneme_type": "raw", "lang_code": "pt-PT", "alphabet": "ipa", "audio": {"sample_rate": 22050},
                        "phoneme_id_map": {"a": 1, "b": 2, " ": 3}, "num_symbols": 60},
        "phoonnx_tokens": {"phoneme_type": "graphemes", "lang_code": "en", "alphabet": "unicode",
                           "phoneme_id_map": {"a": 1, "b": 2}, "pad": "_", "blank": "_", "bos": "^", "eos": "$",
                           "blank_at_start": False, "blank_word": "#", "phonemizer_model": "m"},
        "coqui": {"characters": {"characters_class": "TTS.tts.models.vits.VitsCharacters", "characters": "abc ",
                                 "punctuations": "!,.", "pad": "<PAD>", "eos": "<EOS>", "bos": "<BOS>",
                                 "blank": "<BLNK>"}, "add_blank": True, "enable_eos_bos_chars": False,
                  "datasets": [{"language": "gl"}], "audio": {"sample_rate": 22050}},
        "coqui_noblank": {"characters": {"characters_class": "TTS.tts.utils.text.characters.Graphemes",
                                         "characters": "xyz", "punctuations": "?", "pad": "_", "eos": "~",
                                         "bos": "^", "blank": None}, "add_blank": False, "lang_code": "es"},
        "cotovia": {"characters": {"characters_class": "TTS.tts.models.vits.VitsCharacters", "characters": "abc",
                                   "punctuations": ".", "pad": "_"}, "phoneme_type": "cotovia", "lang_code": "gl"},
    }
    # Piper JSON as the reference's exporter writes it (export_onnx.py:97-130, convert_to_piper): the function is pulled
    # out of the reference file with `ast` and run here (export_onnx.py itself does not import in this container -
    # pytorch_lightning is absent), on a phoonnx voice config; what it writes is one more dialect VoiceConfig must read
    import ast
    import tempfile
    from pathlib import Path
    src = open(os.path.join(REF, "phoonnx_train", "export_onnx.py"), encoding="utf-8").read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "convert_to_piper"]
    ns = {"Path": Path, "json": json, "Dict": dict, "Any": object}
    exec(compile(ast.Module(body=fn, type_ignores=[]), "convert_to_piper", "exec"), ns)
    with tempfile.TemporaryDirectory() as td:
        for tag, ptype in (("espeak", "espeak"), ("raw", "raw")):
            src_cfg = {"phoneme_type": ptype, "lang_code": "pt-PT", "alphabet": "ipa", "audio": {"sample_rate": 22050},
                       "inference": {"noise_scale": 0.6, "length_scale": 1.2, "noise_w": 0.9}, "num_symbols": 70,
                       "num_speakers": 3, "phoonnx_version": "0.2.3",
                       "phoneme_id_map": {"_": 0, "^": 1, "$": 2, " ": 3, "a": 4, "b": 5}}
            cin, cout = Path(td) / "in.json", Path(td) / "piper.json"
            cin.write_text(json.dumps(src_cfg), encoding="utf-8")
            ns["convert_to_piper"](cin, cout)
            cfgs["piper_from_export_" + tag] = json.loads(cout.read_text(encoding="utf-8"))
    for name, cfg in cfgs.items():
        try:
            vc = Cfg.VoiceConfig.from_dict(json.loads(json.dumps(cfg)))
            out["voice_config"].append({"name": name, "config": cfg, "result": enc(vc), "error": None})
        except Exception as e:
            out["voice_config"].append({"name": name, "config": cfg, "result": None, "error": type(e).__name__})
    # mimic3 needs phonemes.txt
    ptxt = "/tmp/_mimic3_phonemes.txt"
    with open(ptxt, "w", encoding="utf-8") as f:
        f.write("# comment\n0 _\n1 ^\n2 $\n3 #\n4 a\n5 b\n6 \n")
    mimic = {"phonemizer": "gruut", "text_language": "en-us", "audio": {"sample_rate": 22050},
             "phonemes": {"pad": "_", "bos": "^", "eos": "$", "blank": "#", "blank_word": "#",
                          "blank_between": "words", "blank_at_start": True, "blank_at_end": False}}
    vc = Cfg.VoiceConfig.from_dict(json.loads(json.dumps(mimic)), phonemes_txt=ptxt)
    out["voice_config"].append({"name": "mimic3_gruut", "config": mimic, "phonemes_txt": open(ptxt, encoding="utf-8").read(),
                                "result": enc(vc), "error": None})
    try:
        Cfg.VoiceConfig.from_dict(json.loads(json.dumps(mimic)))
        err = None
    except Exception as e:
        err = type(e).__name__
    out["voice_config"].append({"name": "mimic3_missing_txt", "config": mimic, "result": None, "error": err})
    mimic_sym = dict(mimic, phonemizer="symbols")
    vc = Cfg.VoiceConfig.from_dict(json.loads(json.dumps(mimic_sym)), phonemes_txt=ptxt)
    out["voice_config"].append({"name": "mimic3_symbols", "config": mimic_sym,
                                "phonemes_txt": open(ptxt, encoding="utf-8").read(), "result": enc(vc), "error": None})

    # ---- feeds + post-processing + wav framing through the reference TTSVoice with a fake session
    class FakeSession:
        def __init__(self, names, audio):
            self.names, self.audio, self.feeds = names, audio, []

        def get_inputs(self):
            return [types.SimpleNamespace(name=n) for n in self.names]

        def run(self, _none, feed):
            self.feeds.append({k: {"dtype": str(v.dtype), "shape": list(v.shape), "data": v.tolist()} for k, v in feed.items()})
            return [self.audio.reshape(1, 1, 1, -1)]

    class Phon:  # minimal phonemizer object, the shape voice.py expects (base.py:58-66 output)
        def add_diacritics(self, text, lang):
            return text

        def phonemize(self, text, lang):
            return [list(s.strip()) for s in text.split(".") if s.strip()]

    rng = np.random.default_rng(7)
    audio = (rng.standard_normal(300) * 0.3).astype(np.float32)
    audio[17] = 0.9
    for names in (["input", "input_lengths", "scales"], ["input", "input_lengths", "scales", "sid"],
                  ["input", "input_lengths"], ["input", "input_lengths", "scales", "sid", "langid"]):
        for syn_kw in ({}, {"speaker_id": 3, "length_scale": 1.3, "noise_scale": 0.1, "noise_w_scale": 0.2},
                       {"normalize_audio": False, "volume": 0.5}, {"volume": 2.0}):
            sess = FakeSession(names, audio)
            vc = Cfg.VoiceConfig.from_dict(json.loads(json.dumps(cfgs["phoonnx_raw"])))
            voice = V.TTSVoice(session=sess, config=vc, phonemizer=Phon())
            syn = Cfg.SynthesisConfig(**syn_kw)
            text = "ab. ba a."
            chunks = list(voice.synthesize(text, syn))
            buf = io.BytesIO()
            with wave.open(buf, "wb") as w:
                voice.synthesize_wav(text, w, syn)
            out["feeds"].append({
                "input_names": names, "syn": syn_kw, "text": text, "feeds": sess.feeds[:len(chunks)],
                "n_run_calls_synthesize": len(chunks), "n_run_calls_total": len(sess.feeds),
                "chunk_float": [c.audio_float_array.tolist() for c in chunks[:1]],
                "chunk_int16_b64": [base64.b64encode(c.audio_int16_bytes).decode() for c in chunks[:1]],
                "chunk_meta": [[c.sample_rate, c.sample_width, c.sample_channels] for c in chunks],
                "wav_b64": base64.b64encode(buf.getvalue()).decode(),
            })
    out["fake_audio"] = audio.tolist()
    # silent audio branch (max < 1e-8 -> zeros), voice.py:272-275
    sess = FakeSession(["input", "input_lengths", "scales"], np.zeros(50, np.float32))
    voice = V.TTSVoice(session=sess, config=Cfg.VoiceConfig.from_dict(json.loads(json.dumps(cfgs["phoonnx_raw"]))),
                       phonemizer=Phon())
    ch = list(voice.synthesize("ab.", None))
    out["postprocess"].append({"case": "silent", "int16_b64": base64.b64encode(ch[0].audio_int16_bytes).decode(),
                               "n": len(ch)})
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False, indent=0)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(out["phonemes_to_ids"]), "id cases,",
          len(out["voice_config"]), "configs,", len(out["feeds"]), "feed cases")


if __name__ == "__main__":
    main()
