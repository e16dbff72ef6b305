"""phoonnx_amd — MI355X-native (gfx950) VITS inference behind phoonnx's TTSVoice API.

Only the hot path of the reference lives here: the engine that replaces the onnxruntime
session (`MiSession`, libvitsmi.so) plus the host-side mirror of the reference interface
around it (`TTSVoice`, `VoiceConfig`, `phonemes_to_ids`)."""
from .session import MiSession, PipelinedSession, RangeError, SessionError  # noqa: F401

__all__ = ["MiSession", "PipelinedSession", "RangeError", "SessionError"]
__version__ = "0.1.0"
