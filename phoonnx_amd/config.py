"""Voice configuration in front of the engine.

Counterpart of `phoonnx/config.py:20-389` (Alphabet, PhonemeType, VoiceConfig with its
dialect sniffing, SynthesisConfig): same names, fields, defaults and `from_dict` results for
Piper / Mimic3 / Coqui-VITS / Cotovia / phoonnx-native JSON configs.  Results are pinned to
the reference's own outputs in tests/golden/frontend.json.
"""
import json
import logging
from dataclasses import dataclass, field
from enum import Enum
from typing import Any, Dict, Mapping, Optional, Sequence

from .phoneme_ids import (DEFAULT_BLANK_TOKEN, DEFAULT_BLANK_WORD_TOKEN, DEFAULT_BOS_TOKEN, DEFAULT_EOS_TOKEN,
                          DEFAULT_PAD_TOKEN, BlankBetween, load_phoneme_ids)

LOG = logging.getLogger(__name__)

DEFAULT_NOISE_SCALE = 0.667
DEFAULT_LENGTH_SCALE = 1.0
DEFAULT_NOISE_W_SCALE = 0.8

Alphabet = Enum("Alphabet", {n.upper().replace("-", ""): n for n in (
    "unicode", "ipa", "arpa", "sampa", "x-sampa", "hangul", "kana", "hira", "hepburn", "kunrei", "nihon",
    "pinyin", "eraab", "cotovia", "hanzi", "buckwalter")}, type=str)

PhonemeType = Enum("PhonemeType", {
    "RAW": "raw", "UNICODE": "unicode", "GRAPHEMES": "graphemes", "MISAKI": "misaki", "ESPEAK": "espeak",
    "GRUUT": "gruut", "GORUUT": "goruut", "EPITRAN": "epitran", "BYT5": "byt5", "CHARSIU": "charsiu",
    "TRANSPHONE": "transphone", "MIRANDESE": "mwl_phonemizer", "DEEPPHONEMIZER": "deepphonemizer",
    "OPENPHONEMIZER": "openphonemizer", "G2PEN": "g2pen", "G2PFA": "g2pfa", "OPENJTALK": "openjtalk",
    "CUTLET": "cutlet", "PYKAKASI": "pykakasi", "COTOVIA": "cotovia", "PHONIKUD": "phonikud", "MANTOQ": "mantoq",
    "VIPHONEME": "viphoneme", "G2PK": "g2pk", "KOG2PK": "kog2p", "G2PC": "g2pc", "G2PM": "g2pm",
    "PYPINYIN": "pypinyin", "XPINYIN": "xpinyin", "JIEBA": "jieba"}, type=str)

_COQUI_CHARACTER_CLASSES = ("TTS.tts.models.vits.VitsCharacters", "TTS.tts.utils.text.characters.Graphemes")
_PHONEME_TYPE_VALUES = {p.value for p in PhonemeType}


@dataclass
class VoiceConfig:
    """TTS model configuration (field order and defaults as the reference dataclass)."""
    num_symbols: int
    num_speakers: int
    num_langs: int
    sample_rate: int
    lang_code: Optional[str]
    phoneme_id_map: Optional[Mapping[str, Sequence[int]]]
    phoneme_type: PhonemeType
    alphabet: Optional[Alphabet]
    phonemizer_model: Optional[str]
    speaker_id_map: Mapping[str, int] = field(default_factory=dict)
    lang_id_map: Mapping[str, int] = field(default_factory=dict)
    # inference settings
    length_scale: float = DEFAULT_LENGTH_SCALE
    noise_scale: float = DEFAULT_NOISE_SCALE
    noise_w_scale: float = DEFAULT_NOISE_W_SCALE
    # tokenisation settings
    blank_at_start: bool = True
    blank_at_end: bool = True
    include_whitespace: Optional[bool] = True
    pad_token: Optional[str] = DEFAULT_PAD_TOKEN
    blank_token: Optional[str] = DEFAULT_PAD_TOKEN
    bos_token: Optional[str] = DEFAULT_BOS_TOKEN
    eos_token: Optional[str] = DEFAULT_EOS_TOKEN
    word_sep_token: Optional[str] = DEFAULT_BLANK_WORD_TOKEN
    blank_between: BlankBetween = BlankBetween.TOKENS_AND_WORDS

    def __post_init__(self):
        self.lang_code = self.lang_code or "und"

    # ---- dialect sniffing (config.py:131-216)
    @staticmethod
    def is_mimic3(config: Dict[str, Any]) -> bool:
        return (isinstance(config.get("phonemizer"), str) and isinstance(config.get("phonemes"), dict)
                and config["phonemizer"] in ("symbols", "gruut", "espeak", "epitran"))

    @staticmethod
    def is_piper(config: Dict[str, Any]) -> bool:
        if "piper_version" in config:
            return True
        return (isinstance(config.get("phoneme_type"), str) and isinstance(config.get("phoneme_id_map"), dict)
                and config["phoneme_type"] in ("text", "espeak"))

    @staticmethod
    def is_coqui_vits(config: Dict[str, Any]) -> bool:
        chars = config.get("characters")
        return isinstance(chars, dict) and chars.get("characters_class", "") in _COQUI_CHARACTER_CLASSES

    @staticmethod
    def is_phoonnx(config: Dict[str, Any]) -> bool:
        return (isinstance(config.get("phoneme_type"), str) and "lang_code" in config
                and config["phoneme_type"] in _PHONEME_TYPE_VALUES)

    @staticmethod
    def is_cotovia(config: Dict[str, Any]) -> bool:
        return (VoiceConfig.is_coqui_vits(config) and VoiceConfig.is_phoonnx(config)
                and config["phoneme_type"] == PhonemeType.COTOVIA)

    @staticmethod
    def from_dict(config: Dict[str, Any], phonemes_txt: Optional[str] = None, lang_code: Optional[str] = None,
                  phoneme_type_str: Optional[str] = None) -> "VoiceConfig":
        """Build the configuration from a voice JSON (config.py:218-358).  NOTE: like the reference, this writes the
        special-token keys it derives back into `config`.

        Structure: what the caller / the JSON say outright goes into a small resolution record; the dialect adapter that
        recognises the JSON (Piper, Mimic 3, Coqui VITS incl. Cotovia; a phoonnx JSON needs none) overrides what its
        format fixes; one constructor call at the end reads the record and the (possibly updated) JSON."""
        external = _external_id_map(phonemes_txt)
        res = _Resolved(lang_code=lang_code or config.get("lang_code"),
                        phoneme_type=phoneme_type_str or config.get("phoneme_type"),
                        id_map=config.get("phoneme_id_map") if external is None else external,
                        alphabet=config.get("alphabet"), has_phonemes_txt=bool(phonemes_txt))
        for recognises, adapt in ((VoiceConfig.is_piper, _adapt_piper), (VoiceConfig.is_mimic3, _adapt_mimic3),
                                  (VoiceConfig.is_coqui_vits, _adapt_coqui)):
            if recognises(config):
                adapt(config, res)
                break
        phoneme_type = PhonemeType(res.phoneme_type)
        LOG.debug("phonemizer: %s", phoneme_type)
        inference = config.get("inference", {})
        scalar = config.get
        return VoiceConfig(
            num_langs=scalar("num_langs", 1), num_symbols=scalar("num_symbols", 256), num_speakers=scalar("num_speakers", 1),
            sample_rate=config.get("audio", {}).get("sample_rate", 16000),
            noise_scale=inference.get("noise_scale", DEFAULT_NOISE_SCALE),
            length_scale=inference.get("length_scale", DEFAULT_LENGTH_SCALE),
            noise_w_scale=inference.get("noise_w", DEFAULT_NOISE_W_SCALE),
            lang_code=res.lang_code, alphabet=res.alphabet, phonemizer_model=scalar("phonemizer_model"),
            phoneme_id_map=res.id_map, phoneme_type=phoneme_type, speaker_id_map=scalar("speaker_id_map", {}),
            blank_between=res.blank_between,
            include_whitespace=" " in scalar("characters", "") or " " in scalar("phoneme_id_map", {}),
            blank_at_start=scalar("blank_at_start", True), blank_at_end=scalar("blank_at_end", True),
            pad_token=scalar("pad"), blank_token=scalar("blank"), bos_token=scalar("bos"), eos_token=scalar("eos"),
            word_sep_token=scalar("word_sep_token") or scalar("blank_word", " "))


@dataclass
class _Resolved:
    """What VoiceConfig.from_dict has settled so far (caller arguments first, then the JSON, then the dialect adapter)."""
    lang_code: Optional[str]
    phoneme_type: Optional[str]
    id_map: Optional[Dict[str, Any]]
    alphabet: Any
    has_phonemes_txt: bool = False
    blank_between: BlankBetween = BlankBetween.TOKENS_AND_WORDS


def _external_id_map(path: Optional[str]):
    """phonemes.txt (Mimic 3 style) or a JSON id map next to the voice; None when absent or of another kind."""
    if not path:
        return None
    if path.endswith(".txt"):
        with open(path, "r", encoding="utf-8") as fh:
            return load_phoneme_ids(fh)
    if path.endswith(".json"):
        with open(path) as fh:
            return json.load(fh)
    return None


def _adapt_piper(config: Dict[str, Any], res: _Resolved) -> None:
    """Piper voices (config.py:259-275): language from `language.code` / `espeak.voice`, "text" = unicode graphemes,
    everything else IPA; the four special tokens are fixed by the format."""
    res.lang_code = res.lang_code or (config.get("language", {}).get("code") or config.get("espeak", {}).get("voice"))
    kind = config.get("phoneme_type", PhonemeType.ESPEAK.value)
    res.phoneme_type, res.alphabet = ((PhonemeType.UNICODE.value, Alphabet.UNICODE) if kind == "text"
                                      else (kind, Alphabet.IPA))
    config.update(pad=DEFAULT_PAD_TOKEN, blank=DEFAULT_BLANK_TOKEN, bos=DEFAULT_BOS_TOKEN, eos=DEFAULT_EOS_TOKEN)


def _adapt_mimic3(config: Dict[str, Any], res: _Resolved) -> None:
    """Mimic 3 voices (config.py:277-297): the symbol table lives in phonemes.txt, the `phonemes` block carries the
    special tokens and the blank policy; "symbols" = a grapheme model."""
    if not res.has_phonemes_txt:
        raise ValueError("mimic3 models require an external phonemes.txt file in addition to the config")
    res.lang_code = config.get("text_language")
    kind = config.get("phonemizer", PhonemeType.GRUUT.value)
    block = config.get("phonemes", {})
    res.blank_between = BlankBetween(block.get("blank_between", "tokens_and_words"))
    config.update(block)
    res.phoneme_type, res.alphabet = ((PhonemeType.GRAPHEMES.value, Alphabet.UNICODE) if kind == "symbols"
                                      else (kind, Alphabet.IPA))


def _adapt_coqui(config: Dict[str, Any], res: _Resolved) -> None:
    """Coqui VITS voices, Cotovia included (config.py:299-338): graphemes (or Cotovia's alphabet), language from the
    first dataset, vocabulary = [pad] + punctuations + characters + [blank]."""
    res.phoneme_type, res.alphabet = ((PhonemeType.COTOVIA.value, Alphabet.COTOVIA) if VoiceConfig.is_cotovia(config)
                                      else (PhonemeType.GRAPHEMES.value, Alphabet.UNICODE))
    datasets = config.get("datasets", [])
    if datasets and not res.lang_code:
        res.lang_code = datasets[0].get("language")
    chars = config.get("characters", {})
    if config.get("add_blank", True):
        res.blank_between = BlankBetween.TOKENS
        chars["blank"] = chars.get("blank") or "<BLNK>"
    config.update(chars)
    if not config.get("enable_eos_bos_chars", True):
        config["bos"] = config["eos"] = None
    edge = lambda key: [chars[key]] if chars.get(key) is not None else []  # noqa: E731
    vocab = edge("pad") + list(chars.get("punctuations") or "") + list(chars.get("characters") or "") + edge("blank")
    res.id_map = {sym: i for i, sym in enumerate(vocab)}


@dataclass
class SynthesisConfig:
    """Per-call synthesis options (config.py:361-389)."""
    speaker_id: Optional[int] = None
    lang_id: Optional[int] = None
    length_scale: Optional[float] = None
    noise_scale: Optional[float] = None
    noise_w_scale: Optional[float] = None
    normalize_audio: bool = True
    volume: float = 1.0
    enable_phonetic_spellings: bool = True
    add_diacritics: bool = True
