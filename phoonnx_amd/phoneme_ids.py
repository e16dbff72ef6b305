"""Phoneme -> id conversion in front of the engine.

Counterpart of `phoonnx/phoneme_ids.py:209-341` (phonemes_to_ids, load_phoneme_ids,
load_phoneme_map): same names, arguments, defaults and results — the ids produced here are
exactly what the model was trained on, so the behaviour is pinned against outputs of the
reference (tests/golden/frontend.json), quirks included:
  * a *missing* special token resolves to the id `len(id_map)` (phoneme_ids.py:232-240);
  * an integer bos token takes the EOS token's value (phoneme_ids.py:238);
  * unknown phonemes are skipped with a warning, never an error (phoneme_ids.py:275-281).
"""
import logging
from enum import Enum
from typing import Dict, List, Mapping, Optional, Sequence, TextIO, Union

LOG = logging.getLogger(__name__)

# The default IPA table assigns ids 0..160 to these symbols in this order
# (data of phoneme_ids.py:20-182, stored as one ordered string instead of a literal dict).
_DEFAULT_SYMBOLS = (
    "_^$ !'(),-.:;?abcdefhijklmnopqrstuvwxyz"
    "\xe6\xe7\xf0\xf8ħŋœǀǁǂǃ"
    "ɐɑɒɓɔɕɖɗɘəɚɛɜɞɟ"
    "ɠɡɢɣɤɥɦɧɨɪɫɬɭɮɯ"
    "ɰɱɲɳɴɵɶɸɹɺɻɽɾ"
    "ʀʁʂʃʄʈʉʊʋʌʍʎʏ"
    "ʐʑʒʔʕʘʙʛʜʝʟʡʢʲ"
    "ˈˌːˑ˞βθχᵻⱱ"
    "0123456789̧̪̯̩̃ʰˤε↓#\"↑̺̻"
    "gʦX̝̊ɝʷ"
)
DEFAULT_IPA_PHONEME_ID_MAP: Dict[str, List[int]] = {s: [i] for i, s in enumerate(_DEFAULT_SYMBOLS)}

DEFAULT_PAD_TOKEN = DEFAULT_BLANK_TOKEN = "_"
DEFAULT_BOS_TOKEN = "^"
DEFAULT_EOS_TOKEN = "$"
DEFAULT_BLANK_WORD_TOKEN = " "

STRESS = {"ˈ", "ˌ"}
PUNCTUATION_MAP = {";": ",", ":": ",", "?": ".", "!": "."}


class BlankBetween(str, Enum):
    """Where blank tokens are interleaved."""
    TOKENS = "tokens"
    WORDS = "words"
    TOKENS_AND_WORDS = "tokens_and_words"


def _special(token, table, fallback):
    """ids of a special token: ints pass through, unknown / empty tokens map to `fallback`."""
    if isinstance(token, int):
        return token
    if token:
        return table.get(token, fallback)
    return fallback


def phonemes_to_ids(phonemes: Sequence[str],
                    id_map: Optional[Mapping[str, Union[int, Sequence[int]]]] = None,
                    blank_token: Optional[str] = DEFAULT_BLANK_TOKEN,
                    bos_token: Optional[str] = DEFAULT_BOS_TOKEN,
                    eos_token: Optional[str] = DEFAULT_EOS_TOKEN,
                    word_sep_token: Optional[str] = DEFAULT_BLANK_WORD_TOKEN,
                    include_whitespace: Optional[bool] = True,
                    blank_at_start: bool = True,
                    blank_at_end: bool = True,
                    blank_between: BlankBetween = BlankBetween.TOKENS_AND_WORDS) -> List[int]:
    """Flatten a phoneme list into model ids, interleaving blanks the way the voice was trained."""
    if not phonemes:
        return []
    table = {k: (v if isinstance(v, list) else [v]) for k, v in (id_map or DEFAULT_IPA_PHONEME_ID_MAP).items()}
    unknown = [len(table)]
    blank = _special(blank_token, table, unknown)
    eos = _special(eos_token, table, unknown)
    # reference quirk: an int bos takes the eos token's value
    bos = eos_token if isinstance(bos_token, int) else _special(bos_token, table, unknown)

    have_blank = blank_token is not None
    per_token = have_blank and blank_between in (BlankBetween.TOKENS, BlankBetween.TOKENS_AND_WORDS)
    per_word = have_blank and blank_between in (BlankBetween.WORDS, BlankBetween.TOKENS_AND_WORDS)

    out: List[int] = []
    if bos_token is not None:
        out.extend(bos)
    if have_blank and blank_at_start:
        out.extend(blank)

    # multi-character keys (diphthongs with their own id, mimic3 style), longest first
    multi = sorted((k for k in table if len(k) > 1), key=len, reverse=True)
    n = len(phonemes)
    pos = 0
    while pos < n:
        hit = next((m for m in multi if "".join(phonemes[pos:pos + len(m)]) == m), None)
        if hit is not None:
            out.extend(table[hit])
            pos += len(hit)
            if per_token and pos < n:
                out.extend(blank)
            continue
        ph = phonemes[pos]
        pos += 1
        if ph not in table:
            if not (ph == " " and not include_whitespace):
                LOG.warning("Missing phoneme from id map: %s", ph)
            continue
        if ph == " ":
            if include_whitespace:
                out.extend(table[ph])
                if per_token:
                    out.extend(blank)
            elif per_word:
                out.extend(table[word_sep_token])
                if per_token:
                    out.extend(blank)
            continue
        out.extend(table[ph])
        if per_token and pos < n:
            out.extend(blank)

    if have_blank and blank_at_end:
        if not include_whitespace and word_sep_token and per_word:
            if per_token:
                out.extend(blank)
            out.extend(table[word_sep_token])
            if per_token:
                out.extend(blank)
        else:
            out.extend(blank)
    if eos_token is not None:
        out.extend(eos)
    return out


def load_phoneme_ids(phonemes_file: TextIO) -> Dict[str, int]:
    """Parse `ID PHONEME` lines (`#` comments; a bare number is the id of the space)."""
    table: Dict[str, int] = {}
    for raw in phonemes_file:
        line = raw.strip("\r\n")
        if not line or line.startswith("#") or " " not in line:
            continue
        if line.strip().isdigit():
            table[" "] = int(line)
            continue
        left, right = line.split(" ", maxsplit=1)
        if right.isdigit():  # PHONEME ID order
            left, right = right, left
        table[right] = int(left)
    return table


def load_phoneme_map(phoneme_map_file: TextIO) -> Dict[str, List[str]]:
    """Parse `FROM TO [TO ...]` lines; an empty TO side maps to a single space."""
    table: Dict[str, List[str]] = {}
    for raw in phoneme_map_file:
        line = raw.strip("\r\n")
        if not line or line.startswith("#") or " " not in line:
            continue
        src, dst = line.split(" ", maxsplit=1)
        table[src] = dst.split() if dst.strip() else [" "]
    return table
