"""Host-side mirror of the reference's public surface on top of the C ABI.

Names and argument meaning follow /root/reference/Sources/Qwen3TTS/Models/Qwen3.swift
(`Qwen3TTSModel.fromPretrained` :1382, `generate` :1291-1301, `supportedSpeakers` :965-971) and
Qwen3+Streaming.swift (`generateStream` :8-18) with the event enum of
Core/GenerationTypes.swift:51-58. Tokenisation stays outside the engine like in the reference
(swift-transformers there): callers pass token ids, or a `tokenizer` callable text -> ids.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Callable, Iterator, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib as L


class Qwen3TTSError(RuntimeError):
    """AudioGenerationError (GenerationTypes.swift:63-84); `.status` is the q3tts_status code."""

    def __init__(self, status: int, message: str):
        super().__init__(message)
        self.status = status


@dataclass
class AudioGenerationInfo:  # GenerationTypes.swift:15-21
    prompt_token_count: int
    generation_token_count: int
    prefill_time: float
    generate_time: float
    tokens_per_second: float
    peak_memory_usage: float

    @property
    def summary(self) -> str:
        """AudioGenerationInfo.summary (GenerationTypes.swift:39-45): the same three lines and number formats."""
        return ("Prompt:     %d tokens, %.2f tokens/s, %.3fs\n"
                "Generation: %d tokens, %.2f tokens/s, %.3fs\n"
                "Peak Memory Usage: %s GB" % (self.prompt_token_count, self.prompt_token_count / max(self.prefill_time, 0.001),
                                              self.prefill_time, self.generation_token_count, self.tokens_per_second,
                                              self.generate_time, repr(float(self.peak_memory_usage))))


@dataclass
class GenerationRequest:
    """One utterance after tokenisation (see q3tts_request in include/q3tts.h)."""
    text_ids: Sequence[int]
    target_token_count: int
    instruct_ids: Optional[Sequence[int]] = None
    speaker: Optional[str] = None
    language: str = "auto"
    max_tokens: int = 2048
    # voice clone (generateVoiceClone, Qwen3.swift:1009-1020): 24 kHz mono float32 + tokens of
    # "<|im_start|>assistant\n{referenceText}<|im_end|>\n"
    ref_audio: Optional[np.ndarray] = None
    ref_text_ids: Optional[Sequence[int]] = None
    # 0: generate() (routed by tts_model_type); 1 / 2: generateVoiceDesign / generateCustomVoice called directly (q3tts.h)
    route: int = 0


@dataclass
class GenerationResult:
    audio: np.ndarray  # float32 [n_samples] @ 24 kHz
    codes: np.ndarray  # int32 [n_frames][16]
    info: AudioGenerationInfo
    status: int = 0


def chat_template_ids(tokenizer: Callable[[str], List[int]], text: str, instruct: Optional[str] = None) -> dict:
    """The three tokenisations the reference performs (Qwen3.swift:274-275, 364-365, 822)."""
    out = {"text_ids": tokenizer(f"<|im_start|>assistant\n{text}<|im_end|>\n<|im_start|>assistant\n"),
           "target_token_count": len(tokenizer(text))}
    if instruct:
        out["instruct_ids"] = tokenizer(f"<|im_start|>user\n{instruct}<|im_end|>\n")
    return out


class _ResultBlock:
    """Owns one q3tts_result array; frees it when the last view over its rows is gone."""

    def __init__(self, lib, res, n):
        self._lib, self._res, self._n = lib, res, n

    def __del__(self):
        try:
            self._lib.q3tts_result_free(self._res, self._n)
        except Exception:  # interpreter shutdown
            pass


class _RowBuf:
    """One row's buffer for numpy (__array_interface__): the array's base is this object, which keeps the block alive."""

    def __init__(self, block, ptr, shape, typestr):
        self._block = block
        self.__array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False), "version": 3}


class Qwen3TTSModel:
    def __init__(self, handle: C.c_void_p, lib):
        self._h = handle
        self._lib = lib
        info = L.ModelInfo()
        self._check(lib.q3tts_model_get_info(self._h, C.byref(info)))
        self.info = info
        self.tokenizer: Optional[Callable[[str], List[int]]] = None
        self.last_info: Optional[AudioGenerationInfo] = None  # .info of the last call's first request

    # -- loading ---------------------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, model_path: str, device: int = 0, max_batch: int = 1, max_frames: int = 2048,
                        max_prompt: int = 512, use_graph: bool = True,
                        weights_from_broadcast: bool = False, n_streams: int = 0,
                        codec_overlap_cus: int = 0, codec_fp32: bool = False) -> "Qwen3TTSModel":
        lib = L.lib()
        o = L.LoadOpts()
        lib.q3tts_default_load_opts(C.byref(o))
        o.device, o.max_batch, o.max_frames, o.max_prompt = device, max_batch, max_frames, max_prompt
        o.use_graph = 1 if use_graph else 0
        o.weights_from_broadcast = 1 if weights_from_broadcast else 0
        o.n_streams = n_streams
        o.codec_overlap_cus = codec_overlap_cus
        o.codec_fp32 = 1 if codec_fp32 else 0
        h = C.c_void_p()
        st = lib.q3tts_model_load(model_path.encode(), C.byref(o), C.byref(h))
        if st != 0:
            raise Qwen3TTSError(st, (lib.q3tts_last_error(None) or b"").decode())
        m = cls(h, lib)
        # AutoTokenizer.from(modelFolder:) in postLoadHook (Qwen3.swift:1456-1459): the engine's own BPE when the files exist
        if any(os.path.exists(os.path.join(model_path, f)) for f in ("tokenizer.json", "vocab.json")):
            m.tokenizer = NativeTokenizer(model_path)
        return m

    def close(self):
        if self._h:
            self._lib.q3tts_model_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st: int):
        if st != 0:
            raise Qwen3TTSError(st, (self._lib.q3tts_last_error(self._h) or b"").decode())

    # -- properties (Qwen3.swift:1262-1271, 965-971, 1210-1214) -----------------------------------
    @property
    def sample_rate(self) -> int:
        return self.info.sample_rate

    @property
    def tts_model_type(self) -> str:
        return self.info.tts_model_type.decode()

    @property
    def supported_speakers(self) -> List[str]:
        n = self._lib.q3tts_model_num_speakers(self._h)
        return [self._lib.q3tts_model_speaker_name(self._h, i).decode() for i in range(n)]

    @property
    def supports_voice_cloning(self) -> bool:
        return bool(self.info.supports_voice_cloning)

    def arena(self) -> Tuple[int, int]:
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self._lib.q3tts_model_arena(self._h, C.byref(p), C.byref(n)))
        return int(p.value), int(n.value)

    @staticmethod
    def comm_unique_id() -> bytes:
        """q3tts_comm_get_unique_id: 128 bytes that rank 0 ships to the other ranks of a sharded job (any channel)."""
        cid = L.CommId()
        st = L.lib().q3tts_comm_get_unique_id(C.byref(cid))
        if st != 0:
            raise Qwen3TTSError(st, (L.lib().q3tts_last_error(None) or b"").decode())
        return C.string_at(C.addressof(cid), 128)  # (c_char arrays stop at the first NUL when read as .bytes)

    def broadcast_weights(self, comm_id: bytes, rank: int, world: int, root: int = 0) -> None:
        """q3tts_model_broadcast: the load-time RCCL broadcast of the weight arena (collective: every rank calls it)."""
        assert len(comm_id) == 128
        cid = L.CommId()
        C.memmove(C.addressof(cid), comm_id, 128)
        self._check(self._lib.q3tts_model_broadcast(self._h, C.byref(cid), rank, world, root))

    def arena_checksum(self) -> int:
        v = C.c_uint64()
        self._check(self._lib.q3tts_model_arena_checksum(self._h, C.byref(v)))
        return int(v.value)

    def last_timing(self) -> L.Timing:
        t = L.Timing()
        self._lib.q3tts_last_timing(self._h, C.byref(t))
        return t

    # -- generation --------------------------------------------------------------------------------
    def _marshal(self, reqs: Sequence[GenerationRequest]):
        arr = (L.Request * len(reqs))()
        keep = []
        for i, r in enumerate(reqs):
            t = np.ascontiguousarray(r.text_ids, np.int32)
            keep.append(t)
            arr[i].text_ids = t.ctypes.data_as(L.i32p)
            arr[i].n_text_ids = t.size
            if r.instruct_ids is not None and len(r.instruct_ids):
                ii = np.ascontiguousarray(r.instruct_ids, np.int32)
                keep.append(ii)
                arr[i].instruct_ids = ii.ctypes.data_as(L.i32p)
                arr[i].n_instruct_ids = ii.size
            arr[i].target_token_count = int(r.target_token_count)
            arr[i].speaker = r.speaker.encode() if r.speaker is not None else None
            arr[i].language = (r.language or "auto").encode()
            arr[i].max_tokens = int(r.max_tokens)
            arr[i].route = int(getattr(r, "route", 0))
            if r.ref_audio is not None:
                ra = np.ascontiguousarray(np.asarray(r.ref_audio, np.float32).reshape(-1))
                rt = np.ascontiguousarray(r.ref_text_ids if r.ref_text_ids is not None else [], np.int32)
                keep += [ra, rt]
                arr[i].ref_audio = ra.ctypes.data_as(L.f32p)
                arr[i].n_ref_samples = ra.size
                arr[i].ref_text_ids = rt.ctypes.data_as(L.i32p)
                arr[i].n_ref_text_ids = rt.size
        return arr, keep

    @staticmethod
    def _sampling(temperature, top_k, top_p, repetition_penalty, seed, force_frames, audio_chunk_frames=0,
                  audio_window_frames=0, audio_lookahead_frames=4, row_base=0) -> L.Sampling:
        s = L.Sampling()
        s.temperature, s.top_k, s.top_p = temperature, top_k, top_p
        s.repetition_penalty, s.seed, s.force_frames = repetition_penalty, seed, force_frames
        s.audio_chunk_frames = audio_chunk_frames
        s.audio_window_frames, s.audio_lookahead_frames = audio_window_frames, audio_lookahead_frames
        s.row_base = row_base
        return s

    def generate_batch(self, reqs: Sequence[GenerationRequest], temperature: float = 0.9, top_k: int = 50,
                       top_p: float = 1.0, repetition_penalty: float = 1.05, seed: int = 0, force_frames: int = 0,
                       on_event: Optional[Callable[[int, str, object], None]] = None,
                       audio_chunk_frames: int = 0, audio_window_frames: int = 0,
                       audio_lookahead_frames: int = 4, row_base: int = 0) -> List[GenerationResult]:
        """n utterances in one call (row-independent). `on_event(request_index, kind, payload)` receives
        ("token", id) / ("info", AudioGenerationInfo) / ("audio", ndarray) in the reference's order; with
        audio_chunk_frames > 0 also ("audio_chunk", (sample_offset, ndarray)) pieces of the final audio, in order,
        between the last token and info (the decoder's causal tail run chunk by chunk; same samples). With
        audio_window_frames > 0 as well the pieces leave while tokens are still being generated (q3tts.h: the
        pre-transformer then sees a sliding window; the waveform is within a stated tolerance of the one-shot decode)."""
        arr, keep = self._marshal(reqs)
        s = self._sampling(temperature, top_k, top_p, repetition_penalty, seed, force_frames, audio_chunk_frames,
                           audio_window_frames, audio_lookahead_frames, row_base)
        cb = self._event_cb(on_event)
        res = (L.Result * len(reqs))()
        st = self._lib.q3tts_generate(self._h, arr, len(reqs), C.byref(s), cb, None, res)
        del keep
        return self._collect(st, res, len(reqs))

    @staticmethod
    def _event_cb(on_event):
        if not on_event:
            return C.cast(None, L.EVENT_CB)

        def _cb(_user, evp):
            ev = evp.contents
            if ev.kind == 3:
                on_event(ev.request_index, "audio_chunk",
                         (int(ev.sample_offset), np.ctypeslib.as_array(ev.pcm, shape=(ev.n_samples,)).copy()))
            elif ev.kind == 0:
                on_event(ev.request_index, "token", int(ev.token))
            elif ev.kind == 1:
                i = ev.info.contents
                on_event(ev.request_index, "info", AudioGenerationInfo(
                    i.prompt_token_count, i.generation_token_count, i.prefill_time, i.generate_time,
                    i.tokens_per_second, i.peak_memory_usage))
            else:
                on_event(ev.request_index, "audio", np.ctypeslib.as_array(ev.pcm, shape=(ev.n_samples,)).copy())

        return L.EVENT_CB(_cb)

    def _collect(self, st, res, n) -> List[GenerationResult]:
        """Results as numpy arrays that VIEW the library's buffers (49 MB of PCM per 32 x 16 s batch: no second copy); the
        buffers go back through q3tts_result_free when the last array over them is collected."""
        block = _ResultBlock(self._lib, res, n)
        self._check(st)
        out = []
        for i in range(n):
            r = res[i]
            inf = r.info
            info = AudioGenerationInfo(inf.prompt_token_count, inf.generation_token_count, inf.prefill_time,
                                       inf.generate_time, inf.tokens_per_second, inf.peak_memory_usage)
            if r.status != 0 or not r.pcm or not r.codes:
                out.append(GenerationResult(np.zeros(0, np.float32), np.zeros((0, 16), np.int32), info, r.status))
                continue
            audio = np.asarray(_RowBuf(block, C.cast(r.pcm, C.c_void_p).value, (int(r.n_samples),), "<f4"))
            codes = np.asarray(_RowBuf(block, C.cast(r.codes, C.c_void_p).value, (int(r.n_frames), 16), "<i4"))
            out.append(GenerationResult(audio, codes, info, 0))
        self.last_info = out[0].info if out else None
        return out

    def generate_batch_begin(self, reqs: Sequence[GenerationRequest], temperature: float = 0.9, top_k: int = 50,
                             top_p: float = 1.0, repetition_penalty: float = 1.05, seed: int = 0, force_frames: int = 0,
                             on_event: Optional[Callable[[int, str, object], None]] = None, audio_chunk_frames: int = 0,
                             more_follows: bool = True, row_base: int = 0):
        """First half of generate_batch (q3tts_generate_begin): returns a job once the AR loop has produced the codes and
        their codec decode is queued. The next batch may be begun before this one is ended: its AR loop then overlaps
        this batch's decode. At most two jobs may be outstanding. more_follows=False (the last batch of a queue) lets the
        decode use the whole chip instead of leaving room for a next batch."""
        arr, keep = self._marshal(reqs)
        s = self._sampling(temperature, top_k, top_p, repetition_penalty, seed, force_frames, audio_chunk_frames, row_base=row_base)
        cb = self._event_cb(on_event)
        job = C.c_void_p()
        self._check(self._lib.q3tts_generate_begin(self._h, arr, len(reqs), C.byref(s), cb, None, 1 if more_follows else 0,
                                                   C.byref(job)))
        del keep  # request memory is only read during begin
        return (job, len(reqs), cb)  # the callback object must outlive the job (INFO / AUDIO fire in end)

    def generate_batch_end(self, job) -> List[GenerationResult]:
        handle, n, _cb = job
        res = (L.Result * n)()
        return self._collect(self._lib.q3tts_generate_end(self._h, handle, res), res, n)

    def _request_from_text(self, text, speaker, instruct, language, max_tokens, text_ids, instruct_ids,
                           target_token_count) -> GenerationRequest:
        if text_ids is None:
            if self.tokenizer is None:
                raise Qwen3TTSError(1, "Model not initialized: Tokenizer not loaded")  # Qwen3.swift:265-267
            t = chat_template_ids(self.tokenizer, text, instruct)
            text_ids, target_token_count = t["text_ids"], t["target_token_count"]
            instruct_ids = t.get("instruct_ids")
        return GenerationRequest(text_ids, int(target_token_count or 0), instruct_ids, speaker, language, max_tokens)

    def generate(self, text: Optional[str] = None, speaker: Optional[str] = None, instruct: Optional[str] = None,
                 language: str = "auto", temperature: float = 0.9, top_k: int = 50, top_p: float = 1.0,
                 repetition_penalty: float = 1.05, max_tokens: int = 2048, *, seed: int = 0,
                 text_ids: Optional[Sequence[int]] = None, instruct_ids: Optional[Sequence[int]] = None,
                 target_token_count: Optional[int] = None) -> np.ndarray:
        """generate(text:speaker:instruct:language:temperature:topK:topP:repetitionPenalty:maxTokens:)
        (Qwen3.swift:1291-1301). Returns float32 samples at 24 kHz."""
        req = self._request_from_text(text, speaker, instruct, language, max_tokens, text_ids, instruct_ids,
                                      target_token_count)
        r = self.generate_batch([req], temperature, top_k, top_p, repetition_penalty, seed)[0]
        if r.status != 0:
            raise Qwen3TTSError(r.status, "Generation failed: No tokens generated")
        return r.audio

    def generate_voice_design(self, text: Optional[str] = None, language: str = "auto", instruct: Optional[str] = None,
                              temperature: float = 0.9, top_k: int = 50, top_p: float = 1.0,
                              repetition_penalty: float = 1.05, max_tokens: int = 2048,
                              on_token: Optional[Callable[[int], None]] = None, *, seed: int = 0, text_ids=None,
                              instruct_ids=None, target_token_count=None) -> np.ndarray:
        """generateVoiceDesign(text:language:instruct:...:onToken:) (Qwen3.swift:587-597); on_token sees every first-codebook
        id as it is generated (:698)."""
        req = self._request_from_text(text, None, instruct, language, max_tokens, text_ids, instruct_ids, target_token_count)
        req.route = 1  # the named prompt builder whatever the checkpoint's type, as the reference's direct call
        return self._one(req, temperature, top_k, top_p, repetition_penalty, seed, on_token)

    def generate_custom_voice(self, text: Optional[str] = None, speaker: str = "", language: str = "auto",
                              instruct: Optional[str] = None, temperature: float = 0.9, top_k: int = 50,
                              top_p: float = 1.0, repetition_penalty: float = 1.05, max_tokens: int = 2048,
                              on_token: Optional[Callable[[int], None]] = None, *, seed: int = 0, text_ids=None,
                              instruct_ids=None, target_token_count=None) -> np.ndarray:
        """generateCustomVoice(text:speaker:language:instruct:...:onToken:) (Qwen3.swift:783-794)."""
        req = self._request_from_text(text, speaker, instruct, language, max_tokens, text_ids, instruct_ids, target_token_count)
        req.route = 2
        return self._one(req, temperature, top_k, top_p, repetition_penalty, seed, on_token)

    def _one(self, req, temperature, top_k, top_p, repetition_penalty, seed, on_token) -> np.ndarray:
        cb = (lambda i, k, p: on_token(p) if k == "token" else None) if on_token else None
        r = self.generate_batch([req], temperature, top_k, top_p, repetition_penalty, seed, on_event=cb)[0]
        if r.status != 0:
            raise Qwen3TTSError(r.status, "Generation failed: No tokens generated")
        return r.audio

    def generate_stream(self, text: Optional[str] = None, speaker: Optional[str] = None,
                        instruct: Optional[str] = None, language: str = "auto", temperature: float = 0.9,
                        top_k: int = 50, top_p: float = 1.0, repetition_penalty: float = 1.05,
                        max_tokens: int = 2048, *, seed: int = 0, text_ids=None, instruct_ids=None,
                        target_token_count=None) -> Iterator[Tuple[str, object]]:
        """generateStream (Qwen3+Streaming.swift:8-125): yields ("token", id)..., ("info", info), ("audio", pcm)."""
        req = self._request_from_text(text, speaker, instruct, language, max_tokens, text_ids, instruct_ids,
                                      target_token_count)
        events: List[Tuple[str, object]] = []
        res = self.generate_batch([req], temperature, top_k, top_p, repetition_penalty, seed,
                                  on_event=lambda i, k, p: events.append((k, p)))
        if res[0].status != 0:
            raise Qwen3TTSError(res[0].status, "Generation failed: No tokens generated")
        yield from events

    def generate_voice_clone(self, text: Optional[str] = None, reference_audio: Optional[np.ndarray] = None,
                             reference_text: Optional[str] = None, language: str = "auto", temperature: float = 0.9,
                             top_k: int = 50, top_p: float = 1.0, repetition_penalty: float = 1.5,
                             max_tokens: int = 2048, *, seed: int = 0, text_ids: Optional[Sequence[int]] = None,
                             ref_text_ids: Optional[Sequence[int]] = None,
                             target_token_count: Optional[int] = None) -> np.ndarray:
        """generateVoiceClone(text:referenceAudio:referenceText:language:...) (Qwen3.swift:1009-1203): repetition
        penalty defaults to 1.5 on this path. Returns the audio of `text` only."""
        if text_ids is None:
            if self.tokenizer is None:
                raise Qwen3TTSError(1, "Model not initialized: Tokenizer not loaded")  # Qwen3.swift:424-426
            t = chat_template_ids(self.tokenizer, text)
            text_ids, target_token_count = t["text_ids"], t["target_token_count"]
            ref_text_ids = self.tokenizer(f"<|im_start|>assistant\n{reference_text}<|im_end|>\n")
        req = GenerationRequest(text_ids, int(target_token_count or 0), None, None, language, max_tokens,
                                ref_audio=reference_audio, ref_text_ids=ref_text_ids)
        r = self.generate_batch([req], temperature, top_k, top_p, repetition_penalty, seed)[0]
        if r.status != 0:
            raise Qwen3TTSError(r.status, "Generation failed: No tokens generated")
        return r.audio

    # -- codec only ------------------------------------------------------------------------------
    def codec_encode(self, audio: np.ndarray) -> np.ndarray:
        """Qwen3TTSSpeechTokenizer.encode (SpeechTokenizer.swift:841-846): waveform [S] -> codes [16][T] int32."""
        a = np.ascontiguousarray(np.asarray(audio, np.float32).reshape(-1))
        cap = max(1, int(self._lib.q3tts_codec_encoded_frames(self._h, a.size)))
        codes = np.zeros((16, cap), np.int32)
        n = C.c_int32(0)
        self._check(self._lib.q3tts_codec_encode(self._h, a.ctypes.data_as(L.f32p), a.size, codes.ctypes.data_as(L.i32p),
                                                 cap, C.byref(n)))
        return codes.reshape(-1)[: 16 * n.value].reshape(16, n.value)

    def extract_speaker_embedding(self, audio: np.ndarray, sample_rate: int = 24000) -> np.ndarray:
        """extractSpeakerEmbedding (Qwen3.swift:222-249) -> float32 [enc_dim]."""
        a = np.ascontiguousarray(np.asarray(audio, np.float32).reshape(-1))
        out = np.zeros(max(1, self.info.speaker_embedding_dim), np.float32)
        self._check(self._lib.q3tts_speaker_embedding(self._h, a.ctypes.data_as(L.f32p), a.size, sample_rate,
                                                      out.ctypes.data_as(L.f32p), out.size))
        return out

    def debug_frontend_stage(self, audio: np.ndarray, stage: str, cap_floats: int = 1 << 26) -> np.ndarray:
        a = np.ascontiguousarray(np.asarray(audio, np.float32).reshape(-1))
        out = np.zeros(cap_floats, np.float32)
        T, Cc = C.c_int32(0), C.c_int32(0)
        self._check(self._lib.q3tts_debug_frontend_stage(self._h, a.ctypes.data_as(L.f32p), a.size, stage.encode(),
                                                         out.ctypes.data_as(L.f32p), out.size, C.byref(T), C.byref(Cc)))
        return out[: T.value * Cc.value].reshape(T.value, Cc.value).copy()

    def codec_decode(self, codes: np.ndarray, n_frames: Optional[Sequence[int]] = None):
        """Qwen3TTSSpeechTokenizer.decode (SpeechTokenizer.swift:823-836). codes [B][F][16] int32."""
        codes = np.ascontiguousarray(codes, np.int32)
        if codes.ndim == 2:
            codes = codes[None]
        B, F, _ = codes.shape
        nf = np.ascontiguousarray(n_frames if n_frames is not None else [F] * B, np.int32)
        up = self.info.samples_per_frame
        pcm = np.zeros((B, F * up), np.float32)
        lens = np.zeros(B, np.int64)
        self._check(self._lib.q3tts_codec_decode(self._h, codes.ctypes.data_as(L.i32p), nf.ctypes.data_as(L.i32p), B, F,
                                                 pcm.ctypes.data_as(L.f32p), lens.ctypes.data_as(C.POINTER(C.c_int64))))
        return pcm, lens

    # -- test hooks -------------------------------------------------------------------------------
    def codec_decode_streamed(self, codes: np.ndarray, chunk_frames: int, window: int, lookahead: int = 4,
                              n_frames: Optional[Sequence[int]] = None) -> np.ndarray:
        """The decode the way a stream produces it (q3tts_codec_decode_streamed): window < 0 = pre-transformer over all
        frames (bit-identical to codec_decode); otherwise a sliding window. Returns pcm [batch][max_frames * 1920]."""
        codes = np.ascontiguousarray(codes, np.int32)
        if codes.ndim == 2:
            codes = codes[None]
        B, F, G = codes.shape
        nf = np.asarray(n_frames if n_frames is not None else [F] * B, np.int32)
        pcm = np.zeros((B, F * self.info.samples_per_frame), np.float32)
        self._check(self._lib.q3tts_codec_decode_streamed(self._h, codes.ctypes.data_as(L.i32p), nf.ctypes.data_as(L.i32p), B, F,
                                                          chunk_frames, window, lookahead, pcm.ctypes.data_as(L.f32p)))
        return pcm

    def debug_prepare_inputs(self, req: GenerationRequest):
        arr, keep = self._marshal([req])
        H = self.info.hidden_size
        cap = 1024
        ie = np.zeros((cap, H), np.uint16)
        tr = np.zeros((cap, H), np.uint16)
        pad = np.zeros((H,), np.uint16)
        n1, n2 = C.c_int32(), C.c_int32()
        self._check(self._lib.q3tts_debug_prepare_inputs(self._h, arr, ie.ctypes.data_as(L.u16p), cap, C.byref(n1),
                                                         tr.ctypes.data_as(L.u16p), cap, C.byref(n2),
                                                         pad.ctypes.data_as(L.u16p)))
        del keep
        return ie[: n1.value].copy(), tr[: n2.value].copy(), pad

    def debug_generate_forced(self, reqs: Sequence[GenerationRequest], forced_codes: np.ndarray, temperature=0.0,
                              top_k=50, top_p=1.0, repetition_penalty=1.05, seed=0):
        arr, keep = self._marshal(reqs)
        n = len(reqs)
        forced = np.ascontiguousarray(forced_codes, np.int32).reshape(n, -1, 16)
        F = forced.shape[1]
        V, Vc, G = self.info.vocab_size, self.info.cp_vocab_size, self.info.num_code_groups
        tl = np.zeros((n, F, V), np.uint16)
        cl = np.zeros((n, F, G - 1, Vc), np.uint16)
        sampled = np.zeros((n, F, 16), np.int32)
        s = self._sampling(temperature, top_k, top_p, repetition_penalty, seed, 0)
        self._check(self._lib.q3tts_debug_generate_forced(
            self._h, arr, n, C.byref(s), forced.ctypes.data_as(L.i32p), F, tl.ctypes.data_as(L.u16p),
            cl.ctypes.data_as(L.u16p), sampled.ctypes.data_as(L.i32p)))
        del keep
        return tl, cl, sampled

    def debug_sample(self, logits: np.ndarray, temperature=0.9, top_k=50, top_p=1.0, repetition_penalty=1.0,
                     seed=0, seen: Optional[np.ndarray] = None, suppress=(0, 0), eos_id=-1, row0=0, draw=0,
                     mask_eos=False):
        logits = np.ascontiguousarray(logits, np.uint16)
        rows, V = logits.shape
        s = self._sampling(temperature, top_k, top_p, repetition_penalty, seed, 1 if mask_eos else 0)
        toks = np.zeros(rows, np.int32)
        sp = np.ascontiguousarray(seen, np.uint8).ctypes.data_as(L.u8p) if seen is not None else None
        self._check(self._lib.q3tts_debug_sample(self._h, logits.ctypes.data_as(L.u16p), rows, V, C.byref(s), sp,
                                                 suppress[0], suppress[1], eos_id, row0, draw,
                                                 toks.ctypes.data_as(L.i32p)))
        return toks

    def debug_linear(self, x: np.ndarray, W: np.ndarray, bias: Optional[np.ndarray] = None) -> np.ndarray:
        x = np.ascontiguousarray(x, np.uint16)
        W = np.ascontiguousarray(W, np.uint16)
        M, K = x.shape
        N = W.shape[0]
        y = np.zeros((M, N), np.uint16)
        bp = np.ascontiguousarray(bias, np.uint16).ctypes.data_as(L.u16p) if bias is not None else None
        self._check(self._lib.q3tts_debug_linear(self._h, x.ctypes.data_as(L.u16p), W.ctypes.data_as(L.u16p), bp, M, K, N,
                                                 y.ctypes.data_as(L.u16p)))
        return y

    def debug_codec_stage(self, codes: np.ndarray, stage: str) -> np.ndarray:
        codes = np.ascontiguousarray(codes, np.int32).reshape(-1, 16)
        F = codes.shape[0]
        cap = F * self.info.samples_per_frame * 128
        out = np.zeros(cap, np.float32)
        T, Cc = C.c_int32(), C.c_int32()
        self._check(self._lib.q3tts_debug_codec_stage(self._h, codes.ctypes.data_as(L.i32p), F, stage.encode(),
                                                      out.ctypes.data_as(L.f32p), cap, C.byref(T), C.byref(Cc)))
        return out[: T.value * Cc.value].reshape(T.value, Cc.value).copy()


class NativeTokenizer:
    """The engine's own Qwen2 byte-level BPE (csrc/tokenizer.cc) behind q3tts_tokenizer_*: a callable text -> ids, usable
    as `Qwen3TTSModel.tokenizer`. `path` is a model directory or a tokenizer.json file."""

    def __init__(self, path: str):
        self._lib = L.lib()
        h = C.c_void_p()
        st = self._lib.q3tts_tokenizer_load(path.encode(), C.byref(h))
        if st != 0:
            raise Qwen3TTSError(st, (self._lib.q3tts_last_error(None) or b"").decode())
        self._h = h

    def __call__(self, text: str) -> List[int]:
        n = C.c_int32(0)
        raw = text.encode("utf-8")
        cap = max(16, len(raw) + 8)  # a token covers at least one byte
        ids = np.zeros(cap, np.int32)
        st = self._lib.q3tts_tokenizer_encode(self._h, raw, ids.ctypes.data_as(L.i32p), cap, C.byref(n))
        if st != 0:
            raise Qwen3TTSError(st, (self._lib.q3tts_last_error(None) or b"").decode())
        return ids[: n.value].tolist()

    def close(self):
        if getattr(self, "_h", None):
            self._lib.q3tts_tokenizer_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
