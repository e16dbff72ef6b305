"""Phonemizer plug point.

Text -> phoneme conversion is upstream of the hot path and OUT OF SCOPE here (SURVEY.md §2
rows 7-10: thirty G2P wrappers over external libraries, CPU string processing).  TTSVoice
accepts any object with the reference's phonemizer shape:
    phonemize(text, lang) -> list[list[str]]   (phonemes grouped by sentence)
    add_diacritics(text, lang) -> str
Provided here: the three "no G2P" phonemizers of `phoonnx/phonemizers/base.py:175-222`
(raw phonemes, graphemes, unicode code points) with a plain sentence splitter and without the
reference's number/date text normalisation (`phoonnx/util.py`, also out of scope).  For every
other phoneme type the reference package's phonemizer is used if it is importable.
"""
import os
import re
import string
import unicodedata
from typing import List, Optional

from .config import Alphabet, PhonemeType

_SENTENCE_END = re.compile(r"(?<=[.!?…])\s+")
_CHUNK_DELIMS = re.compile(r"(, |:|;|\.\.\.|\|)")
_PUNCT = re.compile("[" + re.escape(string.punctuation) + "]")
_LANG_FLAG = re.compile(r"\([^)]+\)")
_WS = re.compile(r"\s+")


class SimplePhonemizer:
    """Sentence splitting + per-chunk `phonemize_string`, phonemes = characters of the result."""

    def phonemize_string(self, text: str, lang: str) -> str:
        return text

    def add_diacritics(self, text: str, lang: str) -> str:
        return text  # Hebrew/Arabic diacritisers are separate neural models in the reference (out of scope)

    def phonemize(self, text: str, lang: str) -> List[List[str]]:
        """base.py:62-86: every chunk (a sentence, or a piece of one between `, ` `:` `;` `...` `|`) is phonemized on its
        own and closes an entry of its own (the reference marks each chunk end-of-sentence, base.py:70), the delimiters
        themselves are dropped."""
        chunks: List[str] = []
        for sentence in (s.strip() for s in _SENTENCE_END.split(text or "")):
            if not sentence:
                continue
            chunks.extend(_PUNCT.sub("", chunk).strip() for chunk in _CHUNK_DELIMS.split(sentence)[::2])
        return [list(_LANG_FLAG.sub("", p)) for p in self.phonemize_strings(chunks, lang)]

    def phonemize_strings(self, chunks: List[str], lang: str) -> List[str]:
        """All chunks of a text; a phonemizer that can run them side by side overrides this (g2p.ByT5Phonemizer)."""
        return [self.phonemize_string(c, lang) for c in chunks]


class RawPhonemes(SimplePhonemizer):
    """The text already is phonemes."""


class GraphemePhonemizer(SimplePhonemizer):
    """Characters of the lower-cased, lightly cleaned text (base.py:186-212)."""

    def phonemize_string(self, text: str, lang: str) -> str:
        text = text.lower().replace(";", ",").replace("-", " ").replace(":", ",")
        text = re.sub(r"[<>()\[\]\"]+", "", text)
        return _WS.sub(" ", text).strip()


class UnicodeCodepointPhonemizer(SimplePhonemizer):
    """Unicode code points after normalisation (NFD splits accents off), base.py:215-226."""

    def __init__(self, form: str = "NFD"):
        self.form = form

    def phonemize_string(self, text: str, lang: str) -> str:
        return unicodedata.normalize(self.form, text)


def get_phonemizer(phoneme_type: PhonemeType, alphabet: Optional[Alphabet] = None, model: Optional[str] = None):
    """Counterpart of config.py:392-465 for the phoneme types that need no external G2P."""
    if phoneme_type == PhonemeType.RAW:
        return RawPhonemes()
    if phoneme_type == PhonemeType.GRAPHEMES:
        return GraphemePhonemizer()
    if phoneme_type == PhonemeType.UNICODE:
        return UnicodeCodepointPhonemizer()
    if phoneme_type in (PhonemeType.BYT5, PhonemeType.CHARSIU) and model and os.path.isfile(model):
        # the neural G2P runs on this engine too; `model` is a local .onnx (nothing is downloaded), its
        # tokenizer_config.json is looked for next to it (mul.py:60-63 keeps one per data directory)
        from .g2p import ByT5Phonemizer, CharsiuPhonemizer
        tok = os.path.join(os.path.dirname(os.path.abspath(model)), "tokenizer_config.json")
        cls = ByT5Phonemizer if phoneme_type == PhonemeType.BYT5 else CharsiuPhonemizer
        return cls(model, tok if os.path.isfile(tok) else None)
    try:  # defer to the reference package when it is installed next to us
        from phoonnx.config import PhonemeType as RefType, get_phonemizer as ref_get
        return ref_get(RefType(phoneme_type.value), alphabet.value if alphabet else "ipa", model)
    except ImportError as exc:
        raise ValueError(
            f"phoneme type {phoneme_type.value!r} needs an external G2P phonemizer; install phoonnx's phonemizers or "
            f"pass TTSVoice(..., phonemizer=<object with phonemize()/add_diacritics()>)") from exc
