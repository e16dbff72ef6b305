"""TTSVoice — the user-facing object of phoonnx with the MI355X engine behind it.

Counterpart of `phoonnx/voice.py:61-379`: `TTSVoice.load()`, `synthesize()`,
`synthesize_wav()`, `phoneme_ids_to_audio()`, `AudioChunk` keep the reference's names,
arguments and observable behaviour; the one difference is the object stored in
`self.session`: a `MiSession` (C ABI -> HIP kernels) instead of an onnxruntime session.
Behaviour around the hot call is pinned to the reference's own outputs
(tests/golden/frontend.json): feed construction, post-processing, int16 conversion, WAV
framing — including the reference's sentence-list duplication (voice.py:203-206), which is
reproduced by default and can be switched off with `dedupe_sentences=True`.

Extensions: `synthesize(..., batch_sentences=True)` renders all
sentences of a text in ONE batched engine call.
"""
import json
import logging
import re
import wave
from dataclasses import dataclass
from pathlib import Path
from typing import Any, Iterable, List, Optional, Union

import numpy as np

from .config import PhonemeType, SynthesisConfig, VoiceConfig
from .phoneme_ids import BlankBetween, phonemes_to_ids
from .phonemizers import get_phonemizer

LOG = logging.getLogger(__name__)

_PHONEME_BLOCK = re.compile(r"(\[\[.*?\]\])")
_MAX_WAV_VALUE = 32767.0


def config_from_metadata(meta: dict) -> dict:
    """Voice-config dict from the `.onnx` metadata_props (keys of export_onnx.py:335-345).
    Raises ValueError when the file carries no usable metadata."""
    if not meta or "phoneme_id_map" not in meta:
        raise ValueError("no voice config JSON and the .onnx has no phoonnx metadata_props to rebuild it from")
    try:
        id_map = json.loads(meta["phoneme_id_map"])
    except json.JSONDecodeError as exc:
        raise ValueError("metadata_props['phoneme_id_map'] is not valid JSON") from exc
    return {
        "phoneme_type": meta.get("phoneme_type") or "raw",
        "alphabet": meta.get("alphabet") or None,
        "phonemizer_model": meta.get("phonemizer_model") or None,
        "lang_code": meta.get("lang_code") or "und",
        "audio": {"sample_rate": int(meta.get("sample_rate", 22050))},
        "num_symbols": int(meta.get("n_vocab", 256)),
        "num_speakers": int(meta.get("n_speakers", 1)),
        "phoneme_id_map": id_map,
    }


def check_consistency(config_dict: dict, meta: dict, graph_speakers: int, graph_vocab: int) -> None:
    """Extension: the three places that describe a voice must agree - the voice JSON, the
    metadata_props export_onnx.py:335-350 wrote into the .onnx, and the graph itself (embedding table sizes).  The
    reference never compares them (config.py:335-358 just reads the JSON) and a mismatch surfaces later as wrong-rate
    audio or an out-of-range Gather inside onnxruntime.  Only values a side actually STATES are compared; raises
    ValueError naming both sides."""
    def as_int(v):
        try:
            return int(v)
        except (TypeError, ValueError):
            return None

    problems = []
    pairs = (("num_speakers", config_dict.get("num_speakers"), "n_speakers"),
             ("audio.sample_rate", (config_dict.get("audio") or {}).get("sample_rate"), "sample_rate"),
             ("num_symbols", config_dict.get("num_symbols"), "n_vocab"))
    for cfg_key, cfg_val, meta_key in pairs:
        c, m = as_int(cfg_val), as_int((meta or {}).get(meta_key))
        if c is not None and m is not None and c != m:
            problems.append(f"config {cfg_key}={c} but the .onnx metadata says {meta_key}={m}")
    c = as_int(config_dict.get("num_speakers"))
    if c is not None and c != graph_speakers and not (c <= 1 and graph_speakers <= 1):
        problems.append(f"config num_speakers={c} but the graph's speaker table has {graph_speakers} rows")
    m = as_int((meta or {}).get("n_speakers"))
    if m is not None and m != graph_speakers and not (m <= 1 and graph_speakers <= 1):
        problems.append(f".onnx metadata n_speakers={m} but the graph's speaker table has {graph_speakers} rows")
    id_map = config_dict.get("phoneme_id_map") or {}
    top = -1
    for v in id_map.values():
        for i in (v if isinstance(v, (list, tuple)) else [v]):
            top = max(top, as_int(i) if as_int(i) is not None else -1)
    if top >= graph_vocab:
        problems.append(f"phoneme_id_map uses id {top} but the graph's embedding table has {graph_vocab} rows")
    if problems:
        raise ValueError("inconsistent voice: " + "; ".join(problems))


@dataclass
class AudioChunk:
    """A chunk of raw audio: float samples in [-1, 1] plus their PCM16 rendering."""
    sample_rate: int
    sample_width: int
    sample_channels: int
    audio_float_array: np.ndarray
    _audio_int16_array: Optional[np.ndarray] = None
    _audio_int16_bytes: Optional[bytes] = None
    _MAX_WAV_VALUE: float = _MAX_WAV_VALUE

    @property
    def audio_int16_array(self) -> np.ndarray:
        if self._audio_int16_array is None:
            scaled = self.audio_float_array * self._MAX_WAV_VALUE
            self._audio_int16_array = np.clip(scaled, -self._MAX_WAV_VALUE, self._MAX_WAV_VALUE).astype(np.int16)
        return self._audio_int16_array

    @property
    def audio_int16_bytes(self) -> bytes:
        return self.audio_int16_array.tobytes()


@dataclass
class TTSVoice:
    session: Any  # MiSession (or anything with get_inputs()/run(): the onnxruntime duck type)
    config: VoiceConfig
    phonetic_spellings: Optional[Any] = None
    phonemizer: Optional[Any] = None
    dedupe_sentences: bool = False  # False = reference behaviour (every sentence list is doubled)

    def __post_init__(self):
        if self.phonemizer is None:
            self.phonemizer = get_phonemizer(self.config.phoneme_type, self.config.alphabet,
                                             self.config.phonemizer_model)

    # ------------------------------------------------------------------ construction
    @staticmethod
    def load(model_path: Union[str, Path], config_path: Optional[Union[str, Path]] = None,
             phonemes_txt: Optional[str] = None, phoneme_map: Optional[str] = None, lang_code: Optional[str] = None,
             phoneme_type_str: Optional[str] = None, use_cuda: bool = False, device_id: int = 0,
             phonemizer: Optional[Any] = None, strict: bool = True) -> "TTSVoice":
        """Load a voice: `<model>.onnx` + `<model>.onnx.json` (voice.py:125-172).  `use_cuda` is
        accepted for signature compatibility; the engine always runs on the MI355X `device_id`.
        strict (extension): raise ValueError when JSON, .onnx metadata and graph disagree (check_consistency)."""
        import os
        from .session import MiSession
        if config_path is None:
            config_path = f"{model_path}.json"
            LOG.debug("Guessing voice config path: %s", config_path)
        session = MiSession(str(model_path), sess_options=None, providers=["MI355XExecutionProvider"],
                            device_id=device_id)
        if os.path.exists(config_path):
            with open(config_path, "r", encoding="utf-8") as fh:
                config_dict = json.load(fh)
        else:
            # extension: no JSON next to the model -> rebuild the config from the
            # metadata_props export_onnx.py:335-350 wrote into the .onnx itself
            config_dict = config_from_metadata(session.get_modelmeta().custom_metadata_map)
        if strict:
            try:
                check_consistency(config_dict, session.get_modelmeta().custom_metadata_map,
                                  session.hparam("n_speakers"), session.hparam("n_vocab"))
            except ValueError:
                session.close()
                raise
        config = VoiceConfig.from_dict(config_dict, phonemes_txt=phonemes_txt, lang_code=lang_code,
                                       phoneme_type_str=phoneme_type_str)
        return TTSVoice(session=session, config=config, phonemizer=phonemizer)

    # ------------------------------------------------------------------ text -> phonemes -> ids
    def phonemize(self, text: str) -> List[List[str]]:
        """Text to phonemes grouped by sentence; `[[ ... ]]` blocks carry literal phonemes."""
        sentences: List[List[str]] = []
        parts = _PHONEME_BLOCK.split(text)
        for i, part in enumerate(parts):
            if part.startswith("[["):
                if not sentences:
                    sentences.append([])
                if i > 0 and parts[i - 1].endswith(" "):
                    sentences[-1].append(" ")
                sentences[-1].extend(list(part[2:-2].strip()))
                if i < len(parts) - 1 and parts[i + 1].startswith(" "):
                    sentences[-1].append(" ")
                continue
            result = self.phonemizer.phonemize(part, self.config.lang_code)
            if self.dedupe_sentences:
                sentences.extend(result)
            else:
                # voice.py:203-206 replaces the accumulator by this part's result and extends it with itself
                sentences = result
                sentences.extend(sentences)
        if sentences and not sentences[-1]:
            sentences.pop()
        return sentences

    def phonemes_to_ids(self, phonemes: List[str]) -> List[int]:
        if self.config.phoneme_id_map is None:
            raise ValueError("self.config.phoneme_id_map is None")
        c = self.config
        return phonemes_to_ids(phonemes, c.phoneme_id_map, blank_token=c.blank_token, bos_token=c.bos_token,
                               eos_token=c.eos_token, word_sep_token=c.word_sep_token,
                               include_whitespace=c.include_whitespace, blank_at_start=c.blank_at_start,
                               blank_at_end=c.blank_at_end,
                               blank_between=BlankBetween.TOKENS_AND_WORDS)  # voice.py:231 ignores config.blank_between

    # ------------------------------------------------------------------ synthesis
    def _postprocess(self, audio: np.ndarray, syn_config: SynthesisConfig) -> np.ndarray:
        """voice.py:271-282: peak-normalise, volume, clip, float32."""
        if syn_config.normalize_audio:
            peak = np.max(np.abs(audio))
            audio = np.zeros_like(audio) if peak < 1e-8 else audio / peak
        if syn_config.volume != 1.0:
            audio = audio * syn_config.volume
        return np.clip(audio, -1.0, 1.0).astype(np.float32)

    def synthesize(self, text: str, syn_config: Optional[SynthesisConfig] = None,
                   batch_sentences: bool = False) -> Iterable[AudioChunk]:
        """One AudioChunk per sentence.  `batch_sentences=True` (extension) renders all sentences in a
        single padded batch on the GPU instead of one engine call per sentence."""
        if syn_config is None:
            syn_config = SynthesisConfig()
        LOG.debug("text=%s", text)
        if self.phonetic_spellings and syn_config.enable_phonetic_spellings:
            text = self.phonetic_spellings.apply(text)
        if syn_config.add_diacritics:
            text = self.phonemizer.add_diacritics(text, self.config.lang_code)
        sentence_phonemes = self.phonemize(text)
        LOG.debug("phonemes=%s", sentence_phonemes)
        all_ids = [self.phonemes_to_ids(p) for p in sentence_phonemes if p]
        all_ids = [ids for ids in all_ids if ids]
        if batch_sentences and len(all_ids) > 1 and hasattr(self.session, "synthesize_batch"):
            audios = self.phoneme_ids_batch_to_audio(all_ids, syn_config)
        else:
            audios = (self.phoneme_ids_to_audio(ids, syn_config) for ids in all_ids)
        for audio in audios:
            yield AudioChunk(sample_rate=self.config.sample_rate, sample_width=2, sample_channels=1,
                             audio_float_array=self._postprocess(audio, syn_config))

    def synthesize_wav(self, text: str, wav_file: wave.Wave_write, syn_config: Optional[SynthesisConfig] = None,
                       set_wav_format: bool = True, batch_sentences: bool = False, device_pcm16: bool = False) -> None:
        """Synthesize and write 16-bit PCM frames (voice.py:291-326).  `device_pcm16=True` (extension):
        all sentences in one batch, peak-normalise / volume / clip / int16 on the GPU, only PCM bytes cross PCIe;
        the frames written are bit-identical to the default path's."""
        if device_pcm16 and hasattr(self.session, "synthesize_batch_pcm16"):
            return self._synthesize_wav_device_pcm16(text, wav_file, syn_config or SynthesisConfig(), set_wav_format)
        sentence_silence = 0.0  # seconds of silence after each sentence (fixed in the reference)
        silence = bytes(int(self.config.sample_rate * sentence_silence * 2))
        first = True
        for chunk in self.synthesize(text, syn_config=syn_config, batch_sentences=batch_sentences):
            if first:
                if set_wav_format:
                    wav_file.setframerate(chunk.sample_rate)
                    wav_file.setsampwidth(chunk.sample_width)
                    wav_file.setnchannels(chunk.sample_channels)
                first = False
            # NOTE (mirrors voice.py:317-324): `first` was cleared just above, so the "silence between sentences" is also
            # written before the first one; it is zero bytes long today (sentence_silence = 0.0), which is the only
            # reason this is inaudible.  Kept for byte-identity with the reference (tests/golden/frontend.json).
            if not first:
                wav_file.writeframes(silence)
            wav_file.writeframes(chunk.audio_int16_bytes)

    def _synthesize_wav_device_pcm16(self, text: str, wav_file: wave.Wave_write, syn_config: SynthesisConfig,
                                     set_wav_format: bool) -> None:
        from .sharding import pad_batch
        if self.phonetic_spellings and syn_config.enable_phonetic_spellings:
            text = self.phonetic_spellings.apply(text)
        if syn_config.add_diacritics:
            text = self.phonemizer.add_diacritics(text, self.config.lang_code)
        all_ids = [self.phonemes_to_ids(p) for p in self.phonemize(text) if p]
        all_ids = [ids for ids in all_ids if ids]
        if set_wav_format:
            wav_file.setframerate(self.config.sample_rate)
            wav_file.setsampwidth(2)
            wav_file.setnchannels(1)
        if not all_ids:
            return
        ids, lens = pad_batch(all_ids)
        expected = [i.name for i in self.session.get_inputs()]
        sid = np.full((len(all_ids),), syn_config.speaker_id or 0, np.int64) if "sid" in expected else None
        pcm, ylen = self.session.synthesize_batch_pcm16(ids, lens, self._scales(syn_config), sid,
                                                        normalize=syn_config.normalize_audio, volume=syn_config.volume)
        hop = self.session.hparam("hop")
        for b in range(len(all_ids)):
            wav_file.writeframes(pcm[b, :int(ylen[b]) * hop].tobytes())

    def _scales(self, syn_config: SynthesisConfig) -> np.ndarray:
        c = self.config
        length = c.length_scale if syn_config.length_scale is None else syn_config.length_scale
        noise = c.noise_scale if syn_config.noise_scale is None else syn_config.noise_scale
        noise_w = c.noise_w_scale if syn_config.noise_w_scale is None else syn_config.noise_w_scale
        return np.array([noise, length, noise_w], dtype=np.float32)

    def phoneme_ids_to_audio(self, phoneme_ids: List[int], syn_config: Optional[SynthesisConfig] = None) -> np.ndarray:
        """Raw (un-normalised) audio for one id sequence: the hot call (voice.py:328-379)."""
        if syn_config is None:
            syn_config = SynthesisConfig()
        expected = [i.name for i in self.session.get_inputs()]
        ids = np.expand_dims(np.array(phoneme_ids, dtype=np.int64), 0)
        feed = {"input": ids, "input_lengths": np.array([ids.shape[1]], dtype=np.int64)}
        if "scales" in expected:
            feed["scales"] = self._scales(syn_config)
        feed["langid"] = np.array([syn_config.lang_id or 0], dtype=np.int64)
        feed["sid"] = np.array([syn_config.speaker_id or 0], dtype=np.int64)
        feed = {k: v for k, v in feed.items() if k in expected}  # voices differ in their inputs
        return self.session.run(None, feed)[0].squeeze()

    def phoneme_ids_batch_to_audio(self, batch_ids: List[List[int]],
                                   syn_config: Optional[SynthesisConfig] = None) -> List[np.ndarray]:
        """Extension: all sequences in one padded batch; each waveform is trimmed to its own length
        (y_lengths * hop), since the generator also renders the padding (models.py:720)."""
        if syn_config is None:
            syn_config = SynthesisConfig()
        from .sharding import pad_batch
        ids, lens = pad_batch(batch_ids)
        expected = [i.name for i in self.session.get_inputs()]
        sid = np.full((len(batch_ids),), syn_config.speaker_id or 0, np.int64) if "sid" in expected else None
        out = self.session.synthesize_batch(ids, lens, self._scales(syn_config), sid)
        hop = self.session.hparam("hop")
        return [out["output"][b, 0, 0, :int(out["y_lengths"][b]) * hop].copy() for b in range(len(batch_ids))]
