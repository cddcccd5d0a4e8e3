"""16-bit WAV output as the reference's CLI writes it (Sources/Qwen3TTSDemo/main.swift:134-165) and a small WAV reader
for reference clips (the reference reads them through AVFoundation, Core/AudioUtils.swift -- Apple-only I/O, so the
reader here is plain Python). The quantisation and the file layout are done by the engine library (q3tts_pcm_to_int16,
q3tts_write_wav)."""
from __future__ import annotations

import ctypes as C
import wave

import numpy as np

from . import _lib as L


def pcm_to_int16(pcm: np.ndarray) -> np.ndarray:
    """Int16(clamp(x, -1, 1) * 32767): truncation toward zero (main.swift:158-162)."""
    x = np.ascontiguousarray(pcm, np.float32).reshape(-1)
    out = np.empty(x.size, np.int16)
    L.lib().q3tts_pcm_to_int16(x.ctypes.data_as(L.f32p), x.size, out.ctypes.data_as(C.POINTER(C.c_int16)))
    return out


def write_wav(path: str, pcm: np.ndarray, sample_rate: int = 24000) -> None:
    x = np.ascontiguousarray(pcm, np.float32).reshape(-1)
    st = L.lib().q3tts_write_wav(str(path).encode(), x.ctypes.data_as(L.f32p), x.size, int(sample_rate))
    if st != 0:
        raise OSError((L.lib().q3tts_last_error(None) or b"q3tts_write_wav failed").decode())


def read_wav(path: str):
    """(sample_rate, float32 mono samples in [-1, 1]) of a 16-bit PCM WAV file; channels are averaged."""
    with wave.open(str(path), "rb") as w:
        if w.getsampwidth() != 2:
            raise ValueError("only 16-bit PCM WAV files are supported")
        raw = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").astype(np.float32) / np.float32(32768.0)
        if w.getnchannels() > 1:
            raw = raw.reshape(-1, w.getnchannels()).mean(axis=1).astype(np.float32)
        return w.getframerate(), raw
