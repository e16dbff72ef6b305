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
        """Build the configuration from a voice JSON (NOTE: like the reference, this writes the
        special-token keys it derives back into `config`)."""
        blank_between = BlankBetween.TOKENS_AND_WORDS
        lang_code = lang_code or config.get("lang_code")
        phoneme_type_str = phoneme_type_str or config.get("phoneme_type")
        id_map = config.get("phoneme_id_map")
        alphabet = config.get("alphabet")

        if phonemes_txt:
            if phonemes_txt.endswith(".txt"):
                with open(phonemes_txt, "r", encoding="utf-8") as fh:
                    id_map = load_phoneme_ids(fh)
            elif phonemes_txt.endswith(".json"):
                with open(phonemes_txt) as fh:
                    id_map = json.load(fh)

        if VoiceConfig.is_piper(config):
            lang_code = lang_code or (config.get("language", {}).get("code") or config.get("espeak", {}).get("voice"))
            phoneme_type_str = config.get("phoneme_type", PhonemeType.ESPEAK.value)
            if phoneme_type_str == "text":
                phoneme_type_str, alphabet = PhonemeType.UNICODE.value, Alphabet.UNICODE
            else:
                alphabet = Alphabet.IPA
            # fixed in piper
            config.update(pad=DEFAULT_PAD_TOKEN, blank=DEFAULT_BLANK_TOKEN, bos=DEFAULT_BOS_TOKEN, eos=DEFAULT_EOS_TOKEN)
        elif VoiceConfig.is_mimic3(config):
            if not phonemes_txt:
                raise ValueError("mimic3 models require an external phonemes.txt file in addition to the config")
            lang_code = config.get("text_language")
            phoneme_type_str = config.get("phonemizer", PhonemeType.GRUUT.value)
            ph = config.get("phonemes", {})
            blank_between = BlankBetween(ph.get("blank_between", "tokens_and_words"))
            config.update(ph)
            if phoneme_type_str == "symbols":  # grapheme model, symbols come from phonemes.txt
                phoneme_type_str, alphabet = PhonemeType.GRAPHEMES.value, Alphabet.UNICODE
            else:
                alphabet = Alphabet.IPA
        elif VoiceConfig.is_coqui_vits(config):  # includes cotovia
            if VoiceConfig.is_cotovia(config):
                phoneme_type_str, alphabet = PhonemeType.COTOVIA.value, Alphabet.COTOVIA
            else:
                phoneme_type_str, alphabet = PhonemeType.GRAPHEMES.value, Alphabet.UNICODE
            datasets = config.get("datasets", [])
            if datasets and not lang_code:
                lang_code = datasets[0].get("language")
            chars = config.get("characters", {})
            if config.get("add_blank", True):
                blank_between = BlankBetween.TOKENS
                chars["blank"] = chars.get("blank") or "<BLNK>"
            config.update(chars)
            if not config.get("enable_eos_bos_chars", True):
                config["bos"] = config["eos"] = None
            # vocabulary order: [pad] + punctuations + characters + [blank]
            vocab = []
            if chars.get("pad") is not None:
                vocab.append(chars["pad"])
            vocab.extend(chars.get("punctuations") or "")
            vocab.extend(chars.get("characters") or "")
            if chars.get("blank") is not None:
                vocab.append(chars["blank"])
            id_map = {sym: i for i, sym in enumerate(vocab)}

        phoneme_type = PhonemeType(phoneme_type_str)
        LOG.debug("phonemizer: %s", phoneme_type)
        inference = config.get("inference", {})
        include_whitespace = " " in config.get("characters", "") or " " in config.get("phoneme_id_map", {})
        return VoiceConfig(
            num_langs=config.get("num_langs", 1),
            num_symbols=config.get("num_symbols", 256),
            num_speakers=config.get("num_speakers", 1),
            sample_rate=config.get("audio", {}).get("sample_rate", 16000),
            noise_scale=inference.get("noise_scale", DEFAULT_NOISE_SCALE),
            length_scale=inference.get("length_scale", DEFAULT_LENGTH_SCALE),
            noise_w_scale=inference.get("noise_w", DEFAULT_NOISE_W_SCALE),
            lang_code=lang_code, alphabet=alphabet, phonemizer_model=config.get("phonemizer_model"),
            phoneme_id_map=id_map, phoneme_type=phoneme_type, speaker_id_map=config.get("speaker_id_map", {}),
            blank_between=blank_between, include_whitespace=include_whitespace,
            blank_at_start=config.get("blank_at_start", True), blank_at_end=config.get("blank_at_end", True),
            pad_token=config.get("pad"), blank_token=config.get("blank"), bos_token=config.get("bos"),
            eos_token=config.get("eos"),
            word_sep_token=config.get("word_sep_token") or config.get("blank_word", " "))


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
