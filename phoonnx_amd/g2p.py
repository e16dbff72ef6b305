"""ByT5 G2P on the MI355X: the session object that stands where `onnxruntime.InferenceSession` stands in
`phoonnx/phonemizers/mul.py:106`, and a mirror of `ByT5Phonemizer` around it.

    MiG2PSession(path)                       mul.py:106
    session.get_outputs() -> [.name]         mul.py:183
    session.run(names, feed) -> [logits]     mul.py:199-211   (float32 [1, T, vocab])
plus `generate(input_ids)`: the whole greedy loop of mul.py:192-230 in one engine call with a key/value cache (the
reference re-runs the entire graph, encoder included, for every generated token).

No CPU fallback: without libvitsmi.so and an MI355X this raises.
"""
import ctypes as C
import json
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _ffi
from .phonemizers import SimplePhonemizer
from .session import NodeArg, SessionError

BYT5_LANGS = ['ca-ES', 'cy-GB', 'da-DK', 'de-DE', 'en-GB', 'en-US', 'es-ES', 'et-EE', 'eu-ES', 'fa-IR', 'fr-FR',
              'ga-IE', 'hr-HR', 'hu-HU', 'id-ID', 'is-IS', 'it-IT', 'ja-JP', 'ko-KR', 'nb-NO', 'nl-NL', 'pl-PL',
              'pt-BR', 'pt-PT', 'qu-PE', 'ro-RO', 'sr-RS', 'sv-SE', 'tr-TR', 'yue-CN', 'zh-CN']  # mul.py:31-33


# ---- language matching.  The reference resolves a BCP-47 tag to a phonemizer's own tag list with langcodes'
# tag_distance (`phonemizers/base.py:86-122` match_lang: exact tag, else the closest supported one within distance 10,
# else ValueError).  langcodes is not a dependency here; its outcome for these two short lists is reproduced by
# (language, region / script) comparison with CLDR's likely regions as the default.
_ISO639_3_TO_1 = {
    "afr": "af", "sqi": "sq", "amh": "am", "ara": "ar", "arg": "an", "arm": "hy", "hye": "hy", "aze": "az", "bak": "ba",
    "eus": "eu", "baq": "eu", "bel": "be", "ben": "bn", "bos": "bs", "bul": "bg", "bur": "my", "mya": "my", "cat": "ca",
    "zho": "zh", "chi": "zh", "cmn": "zh", "cze": "cs", "ces": "cs", "dan": "da", "dut": "nl", "nld": "nl", "eng": "en",
    "epo": "eo", "est": "et", "fin": "fi", "fra": "fr", "fre": "fr", "gla": "gd", "geo": "ka", "kat": "ka", "ger": "de",
    "deu": "de", "gre": "el", "ell": "el", "grn": "gn", "guj": "gu", "hin": "hi", "hun": "hu", "ido": "io", "ind": "id",
    "ina": "ia", "ita": "it", "jpn": "ja", "kaz": "kk", "khm": "km", "kor": "ko", "kur": "ku", "lat": "la", "lit": "lt",
    "ltz": "lb", "mac": "mk", "mkd": "mk", "mlt": "mt", "nob": "nb", "nor": "nb", "ori": "or", "fas": "fa", "per": "fa",
    "pol": "pl", "por": "pt", "ron": "ro", "rum": "ro", "rus": "ru", "san": "sa", "srp": "sr", "hbs": "sh", "snd": "sd",
    "slo": "sk", "slk": "sk", "slv": "sl", "spa": "es", "swa": "sw", "swe": "sv", "tgl": "tl", "tam": "ta", "tat": "tt",
    "tha": "th", "tur": "tr", "tuk": "tk", "ukr": "uk", "vie": "vi", "wel": "cy", "cym": "cy", "ice": "is", "isl": "is",
    "gle": "ga", "glg": "gl", "sme": "se", "hrv": "hr", "que": "qu", "min": "nan",
    "no": "nb", "nn": "nb", "iw": "he", "in": "id", "tl": "tl", "fil": "tl",  # macro-language / legacy two-letter aliases
}
_SUBTAG = {  # tag-list spellings of a region / script / variant -> BCP-47 subtag
    "uk": "GB", "us": "US", "po": "PT", "bz": "BR", "qu": "CA", "latin": "419", "me": "MX", "t": "Hant", "s": "Hans",
    "latn": "Latn", "cyrl": "Cyrl",
}
_DEFAULT_REGION = {"en": "US", "pt": "BR", "es": "ES", "fr": "FR", "zh": "Hans", "sh": "Latn"}  # CLDR likely subtags
_NEAR = {  # an unsupported region -> the supported one CLDR's matching puts closest
    "en": lambda r: "US" if r in ("US", "CA", "PH", "PR", "UM", "VI") else "GB",
    "pt": lambda r: "BR" if r == "BR" else "PT",
    "es": lambda r: r if r in ("ES", "MX") else ("ES" if r in ("EA", "IC", "GQ") else "419"),
    "fr": lambda r: "CA" if r == "CA" else "FR",
    "zh": lambda r: "Hant" if r in ("TW", "HK", "MO", "Hant") else "Hans",
}


def _parse_tag(tag: str):
    parts = tag.replace("_", "-").split("-")
    lang = parts[0].lower()
    lang = _ISO639_3_TO_1.get(lang, lang)
    sub = None
    for p in parts[1:]:
        q = _SUBTAG.get(p.lower())
        if q is None and (len(p) == 2 or (len(p) == 3 and p.isdigit())):
            q = p.upper()
        if q is None and len(p) == 4 and p.isalpha():
            q = p.title()
        if q is not None and sub is None:
            sub = q
    return lang, sub


def match_lang(target_lang: str, valid_langs: Sequence[str]) -> str:
    """`BasePhonemizer.match_lang` (phonemizers/base.py:86-122): the tag itself when supported, else the closest supported
    tag, else ValueError("unsupported language code: ...")."""
    if target_lang in valid_langs:
        return target_lang
    lang, sub = _parse_tag(target_lang)
    cands = [(v,) + _parse_tag(v) for v in valid_langs]
    cands = [(v, s) for v, l, s in cands if l == lang]
    if not cands:
        raise ValueError(f"unsupported language code: {target_lang}")
    if lang == "zh" and sub in ("CN", "SG", "MY"):
        sub = "Hans"
    # A script mismatch is not "close": langcodes' tag_distance (what the reference asks, maximum 10) puts zh-Hant / zh-TW
    # against a list that only has zh-CN, or sr-Cyrl against sr-Latn, far beyond that, and the reference raises.
    def script_of(l, s):
        if l == "zh":
            return "Hant" if s in ("TW", "HK", "MO", "Hant") else "Hans"
        return s if s is not None and len(s) == 4 and s.isalpha() else None
    ws = script_of(lang, sub)
    if ws is not None and not any(script_of(lang, s) in (ws, None) for _, s in cands):
        raise ValueError(f"unsupported language code: {target_lang}")
    want = sub if sub is not None else _DEFAULT_REGION.get(lang)
    for v, s in cands:  # same region / script (the ByT5 list spells Chinese as zh-CN)
        if s == want or (lang == "zh" and {s, want} == {"CN", "Hans"}):
            return v
    if sub is not None and lang in _NEAR:
        near = _NEAR[lang](sub)
        for v, s in cands:
            if s == near or (lang == "zh" and {s, near} == {"CN", "Hans"}):
                return v
    want = _DEFAULT_REGION.get(lang)
    for v, s in cands:
        if s == want or s is None:
            return v
    return cands[0][0]


def encode_text(text: str, lang: str) -> np.ndarray:
    """`ByT5Phonemizer._encode_text` (mul.py:152-170): "<lang>: text" as UTF-8 bytes, shifted by the 3 special ids."""
    data = f"<{lang}>: {text}".encode("utf-8")
    return np.array([[b + 3 for b in data]], dtype=np.int64)


def decode_phones(preds: Sequence[int], added_tokens: Dict[str, object]) -> str:
    """`ByT5Phonemizer._decode_phones` (mul.py:135-150): ids minus 3 are bytes; special / added tokens are dropped."""
    data = b"".join(bytes([t - 3]) for t in preds if str(t) not in added_tokens)
    return data.decode("utf-8", errors="ignore")


class MiG2PSession:
    def __init__(self, path, sess_options=None, providers=None, device_id: int = 0, host_only: bool = False, **kwargs):
        self._lib = _ffi.load()
        self._h = C.c_void_p()
        self.path = str(path)
        rc = self._lib.g2p_open(self.path.encode(), -1 if host_only else device_id, C.byref(self._h))
        if rc != 0:
            self._h = C.c_void_p()
            raise SessionError(f"g2p_open({self.path!r}) failed [{rc}]: {self._lib.g2p_last_error(None).decode()}")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.g2p_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _err(self):
        return self._lib.g2p_last_error(self._h).decode("utf-8", "replace")

    def hparam(self, key) -> int:
        v = C.c_int64()
        if self._lib.g2p_hparam(self._h, key.encode(), C.byref(v)) != 0:
            raise SessionError(self._err())
        return v.value

    def bucket(self, rel: int, decoder: bool = False) -> int:
        return self._lib.g2p_bucket(self._h, 1 if decoder else 0, int(rel))

    # ------------------------------------------------------------------ onnxruntime surface
    def get_inputs(self) -> List[NodeArg]:
        return [NodeArg(n, "tensor(int64)", ["batch", "sequence"]) for n in ("input_ids", "attention_mask", "decoder_input_ids")]

    def get_outputs(self) -> List[NodeArg]:
        n = self._lib.g2p_num_outputs(self._h)
        return [NodeArg(self._lib.g2p_output_name(self._h, i).decode(), "tensor(float)", ["batch", "target", "vocab"])
                for i in range(n)]

    def run(self, output_names: Optional[Sequence[str]], input_feed: Dict[str, np.ndarray], run_options=None):
        names = [o.name for o in self.get_outputs()]
        if output_names is not None and any(n not in names for n in output_names):
            raise SessionError(f"unknown output names {output_names!r}")
        for k in input_feed:
            if k not in ("input_ids", "attention_mask", "decoder_input_ids"):
                raise SessionError(f"Invalid input name: {k}")
        for k in ("input_ids", "decoder_input_ids"):
            if k not in input_feed:
                raise SessionError(f"Required input {k} is missing")
        ids = np.ascontiguousarray(input_feed["input_ids"])
        dec = np.ascontiguousarray(input_feed["decoder_input_ids"])
        mask = input_feed.get("attention_mask")
        if ids.dtype != np.int64 or dec.dtype != np.int64 or ids.ndim != 2 or dec.ndim != 2 or ids.shape[0] != 1 or dec.shape[0] != 1:
            raise SessionError("input_ids / decoder_input_ids must be int64 of shape [1, length] (batch 1, as mul.py feeds them)")
        if mask is not None:
            mask = np.ascontiguousarray(mask)
            if mask.dtype != np.int64 or mask.shape != ids.shape:
                raise SessionError("attention_mask must be int64 with the shape of input_ids")
        T, V = dec.shape[1], self.hparam("vocab")
        out = np.empty((1, T, V), np.float32)
        rc = self._lib.g2p_run(self._h, _ffi.ptr(ids), ids.shape[1], _ffi.ptr(mask), _ffi.ptr(dec), T, _ffi.ptr(out))
        if rc != 0:
            raise SessionError(f"g2p_run failed [{rc}]: {self._err()}")
        return [out]

    # ------------------------------------------------------------------ extension: the whole greedy loop on the device
    def generate(self, input_ids, max_length: int = 512, start_id: int = 0, eos_id: int = 1) -> List[int]:
        ids = np.ascontiguousarray(np.asarray(input_ids, np.int64).reshape(-1))
        out = np.zeros(max_length, np.int64)
        n = C.c_int(0)
        rc = self._lib.g2p_generate(self._h, _ffi.ptr(ids), ids.shape[0], int(max_length), int(start_id), int(eos_id),
                                    _ffi.ptr(out), C.byref(n))
        if rc != 0:
            raise SessionError(f"g2p_generate failed [{rc}]: {self._err()}")
        return out[:n.value].tolist()


    MAX_BATCH = 64  # G2P_MAX_BATCH (include/g2pmi.h)

    def generate_batch(self, inputs: Sequence[Sequence[int]], max_length: int = 512, start_id: int = 0,
                       eos_id: int = 1) -> List[List[int]]:
        """`generate` for several inputs side by side (g2p_generate_batch): same ids as one call each.  Inputs are
        grouped by length, at most MAX_BATCH per engine call; results come back in the order given."""
        seqs = [np.ascontiguousarray(np.asarray(x, np.int64).reshape(-1)) for x in inputs]
        order = sorted(range(len(seqs)), key=lambda i: len(seqs[i]))
        res: List[List[int]] = [[] for _ in seqs]
        for g in range(0, len(order), self.MAX_BATCH):
            idx = order[g:g + self.MAX_BATCH]
            flat = np.ascontiguousarray(np.concatenate([seqs[i] for i in idx]))
            lens = np.array([len(seqs[i]) for i in idx], np.int32)
            out = np.zeros((len(idx), max_length), np.int64)
            n = np.zeros(len(idx), np.int32)
            rc = self._lib.g2p_generate_batch(self._h, _ffi.ptr(flat), _ffi.ptr(lens), len(idx), int(max_length), int(start_id),
                                              int(eos_id), _ffi.ptr(out), _ffi.ptr(n))
            if rc != 0:
                raise SessionError(f"g2p_generate_batch failed [{rc}]: {self._err()}")
            for j, i in enumerate(idx):
                res[i] = out[j, :n[j]].tolist()
        return res


    def forced_step_logits(self, inputs: Sequence[Sequence[int]], decoder_input_ids) -> np.ndarray:
        """Test hook (g2p_test_forced_steps): the decoder-step kernels of `generate` driven with GIVEN decoder inputs
        [B, T] (column 0 = the start token) for up to four inputs side by side -> the logits of every step, float32
        [B, T, vocab]: position t equals what run() returns there for decoder_input_ids[:, :t + 1]."""
        seqs = [np.ascontiguousarray(np.asarray(x, np.int64).reshape(-1)) for x in inputs]
        dec = np.ascontiguousarray(np.asarray(decoder_input_ids, np.int64))
        if dec.ndim != 2 or dec.shape[0] != len(seqs):
            raise SessionError(f"decoder_input_ids must be [{len(seqs)}, T], got {dec.shape}")
        flat = np.ascontiguousarray(np.concatenate(seqs))
        lens = np.array([len(x) for x in seqs], np.int32)
        V = self.hparam("vocab")
        out = np.zeros((len(seqs), dec.shape[1], V), np.float32)
        rc = self._lib.g2p_test_forced_steps(self._h, _ffi.ptr(flat), _ffi.ptr(lens), len(seqs), _ffi.ptr(dec), int(dec.shape[1]),
                                             _ffi.ptr(out))
        if rc != 0:
            raise SessionError(f"g2p_test_forced_steps failed [{rc}]: {self._err()}")
        return out


class ByT5Phonemizer(SimplePhonemizer):
    """Mirror of `phoonnx.phonemizers.mul.ByT5Phonemizer` over `MiG2PSession` (no downloads: the model and tokenizer
    config are files the caller supplies).  `phonemize_string(text, lang)` as mul.py:232-233; `phonemize(text, lang)`
    (chunking, one entry per chunk) from the base class, as in the reference."""

    def __init__(self, model: str, tokenizer_config: Optional[str] = None, device_id: int = 0, device_loop: bool = True):
        self.session = MiG2PSession(model, device_id=device_id)
        self.tokens: Dict[str, object] = {"0": "<pad>", "1": "</s>", "2": "<unk>"}
        if tokenizer_config:
            with open(tokenizer_config, "r") as f:
                self.tokens = json.load(f).get("added_tokens_decoder", {})
        self.device_loop = device_loop

    @staticmethod
    def get_lang(target_lang: str) -> str:
        """mul.py:111-125: the closest tag of BYT5_LANGS ('en' -> 'en-US', 'pt' -> 'pt-BR', 'en-AU' -> 'en-GB')."""
        return match_lang(target_lang, BYT5_LANGS)

    def _prefix_lang(self, lang: str) -> str:
        """The tag written in front of the text, "<tag>: text" (mul.py:152-170)."""
        return self.get_lang(lang)

    def _infer(self, text: str, lang: str, max_length: int = 512) -> str:
        if not text.strip():
            return ""
        ids = encode_text(text, self._prefix_lang(lang))
        if self.device_loop:
            return decode_phones(self.session.generate(ids[0], max_length), self.tokens)
        # the reference's loop, call for call (mul.py:192-230)
        names = [o.name for o in self.session.get_outputs()]
        mask = np.ones_like(ids)
        dec = np.array([[0]], np.int64)
        gen: List[int] = []
        for _ in range(max_length):
            logits = self.session.run(names, {"input_ids": ids, "attention_mask": mask, "decoder_input_ids": dec})[0]
            nxt = int(np.argmax(logits[0, -1, :]))
            gen.append(nxt)
            if nxt == 1:
                break
            dec = np.concatenate((dec, np.array([[nxt]], np.int64)), axis=1)
        return decode_phones(gen, self.tokens)

    def phonemize_string(self, text: str, lang: str) -> str:
        return self._infer(text, lang)

    def phonemize_strings(self, chunks: List[str], lang: str, max_length: int = 512) -> List[str]:
        """Every chunk of a text in one batched greedy loop (the reference runs them one after the other, base.py:66-70;
        the ids of each chunk are the same either way)."""
        if not self.device_loop or len(chunks) < 2:
            return [self.phonemize_string(c, lang) for c in chunks]
        todo = [i for i, c in enumerate(chunks) if c.strip()]
        gen = self.session.generate_batch([encode_text(chunks[i], self._prefix_lang(lang))[0] for i in todo], max_length)
        out = [""] * len(chunks)
        for i, ids in zip(todo, gen):
            out[i] = decode_phones(ids, self.tokens)
        return out


CHARSIU_LANGS = ['ady', 'afr', 'sqi', 'amh', 'ara', 'arg', 'arm-e', 'arm-w', 'aze', 'bak', 'eus', 'bel', 'ben', 'bos',
                 'bul', 'bur', 'cat', 'yue', 'zho-t', 'zho-s', 'min', 'cze', 'dan', 'dut', 'eng-uk', 'eng-us', 'epo',
                 'est', 'fin', 'fra', 'fra-qu', 'gla', 'geo', 'ger', 'gre', 'grc', 'grn', 'guj', 'hin', 'hun', 'ido',
                 'ind', 'ina', 'ita', 'jam', 'jpn', 'kaz', 'khm', 'kor', 'kur', 'lat-clas', 'lat-eccl', 'lit', 'ltz',
                 'mac', 'mlt', 'tts', 'nob', 'ori', 'pap', 'fas', 'pol', 'por-po', 'por-bz', 'ron', 'rus', 'san',
                 'srp', 'hbs-latn', 'hbs-cyrl', 'snd', 'slo', 'slv', 'spa', 'spa-latin', 'spa-me', 'swa', 'swe', 'tgl',
                 'tam', 'tat', 'tha', 'tur', 'tuk', 'ukr', 'vie-n', 'vie-c', 'vie-s', 'wel-nw', 'wel-sw', 'ice', 'ang',
                 'gle', 'enm', 'syc', 'glg', 'sme', 'egy']  # mul.py:248-256


def _charsiu_to_bcp47(tag: str) -> str:
    lang, sub = _parse_tag(tag)
    return lang if sub is None else f"{lang}-{sub}"


class CharsiuPhonemizer(ByT5Phonemizer):
    """Mirror of `phoonnx.phonemizers.mul.CharsiuPhonemizer` (mul.py:239-286): the same engine, Charsiu's language tags,
    and - these models cannot handle whitespace - one G2P call per word.

    reference_prefix: which tag goes in front of each word.  In the reference `_encode_text` is a staticmethod that
    always calls `ByT5Phonemizer.get_lang` (mul.py:159), so a Charsiu model is fed "<en-US>: word" although
    `CharsiuPhonemizer.get_lang` would say "eng-us", the tag those checkpoints were trained with.  True (default)
    reproduces the reference's input ids exactly; False writes the Charsiu tag."""

    def __init__(self, model: str, tokenizer_config: Optional[str] = None, device_id: int = 0, device_loop: bool = True,
                 reference_prefix: bool = True):
        super().__init__(model, tokenizer_config, device_id, device_loop)
        self.reference_prefix = reference_prefix

    @staticmethod
    def get_lang(target_lang: str) -> str:
        """mul.py:269-284: the closest Charsiu tag ('en-US' -> 'eng-us', 'pt-BR' -> 'por-bz', 'de' -> 'ger')."""
        return match_lang(target_lang, CHARSIU_LANGS)

    def _prefix_lang(self, lang: str) -> str:
        if self.reference_prefix:
            try:
                return ByT5Phonemizer.get_lang(lang)          # mul.py:159, as the reference does
            except ValueError:
                return ByT5Phonemizer.get_lang(_charsiu_to_bcp47(self.get_lang(lang)))
        return self.get_lang(lang)

    def phonemize_string(self, text: str, lang: str) -> str:
        return " ".join(self._infer(w, lang) for w in text.split())

    def phonemize_strings(self, chunks: List[str], lang: str, max_length: int = 512) -> List[str]:
        """All words of all chunks in one batched loop, re-joined per chunk (mul.py:284-286 word by word)."""
        if not self.device_loop:
            return [self.phonemize_string(c, lang) for c in chunks]
        words = [c.split() for c in chunks]
        flat = [w for ws in words for w in ws]
        if not flat:
            return ["" for _ in chunks]
        gen = self.session.generate_batch([encode_text(w, self._prefix_lang(lang))[0] for w in flat], max_length)
        it = iter(decode_phones(ids, self.tokens) for ids in gen)
        return [" ".join(next(it) for _ in ws) for ws in words]
