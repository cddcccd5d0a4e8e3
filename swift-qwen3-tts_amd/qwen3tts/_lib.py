"""ctypes binding of include/q3tts.h (libq3tts_hip.so). The library is the product; this module
only marshals arguments. It fails loudly when the HIP extension is missing -- there is no CPU
fallback and nothing here touches oracle/."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# Q3TTS_LIB: another build of the same library (A/B measurements of a kernel change inside one gpurun call)
LIB_PATH = os.environ.get("Q3TTS_LIB") or os.path.join(_HERE, "libq3tts_hip.so")

u16p = C.POINTER(C.c_uint16)
i32p = C.POINTER(C.c_int32)
f32p = C.POINTER(C.c_float)
u8p = C.POINTER(C.c_uint8)


class LoadOpts(C.Structure):
    _fields_ = [("device", C.c_int32), ("max_batch", C.c_int32), ("max_frames", C.c_int32),
                ("max_prompt", C.c_int32), ("use_graph", C.c_int32), ("weights_from_broadcast", C.c_int32),
                ("n_streams", C.c_int32), ("codec_overlap_cus", C.c_int32), ("codec_fp32", C.c_int32)]


class CommId(C.Structure):  # q3tts_comm_id == ncclUniqueId
    _fields_ = [("bytes", C.c_char * 128)]


class ModelInfo(C.Structure):
    _fields_ = [("tts_model_type", C.c_char * 32), ("sample_rate", C.c_int32),
                ("supports_voice_cloning", C.c_int32), ("has_voice_cloning", C.c_int32),
                ("hidden_size", C.c_int32), ("num_layers", C.c_int32), ("vocab_size", C.c_int32),
                ("text_vocab_size", C.c_int32), ("num_code_groups", C.c_int32),
                ("cp_hidden_size", C.c_int32), ("cp_num_layers", C.c_int32), ("cp_vocab_size", C.c_int32),
                ("codec_eos_token_id", C.c_int32), ("samples_per_frame", C.c_int32), ("max_batch", C.c_int32),
                ("weight_bytes", C.c_int64), ("speaker_embedding_dim", C.c_int32)]


class Request(C.Structure):
    _fields_ = [("text_ids", i32p), ("n_text_ids", C.c_int32), ("instruct_ids", i32p),
                ("n_instruct_ids", C.c_int32), ("target_token_count", C.c_int32), ("speaker", C.c_char_p),
                ("language", C.c_char_p), ("max_tokens", C.c_int32),
                ("ref_audio", f32p), ("n_ref_samples", C.c_int64), ("ref_text_ids", i32p),
                ("n_ref_text_ids", C.c_int32), ("route", C.c_int32)]


class Sampling(C.Structure):
    _fields_ = [("temperature", C.c_float), ("top_k", C.c_int32), ("top_p", C.c_float),
                ("repetition_penalty", C.c_float), ("seed", C.c_uint64), ("force_frames", C.c_int32), ("audio_chunk_frames", C.c_int32),
                ("audio_window_frames", C.c_int32), ("audio_lookahead_frames", C.c_int32), ("row_base", C.c_uint32)]


class GenInfo(C.Structure):
    _fields_ = [("prompt_token_count", C.c_int32), ("generation_token_count", C.c_int32),
                ("prefill_time", C.c_double), ("generate_time", C.c_double), ("tokens_per_second", C.c_double),
                ("peak_memory_usage", C.c_double)]


class Event(C.Structure):
    _fields_ = [("kind", C.c_int), ("request_index", C.c_int32), ("token", C.c_int32),
                ("info", C.POINTER(GenInfo)), ("pcm", f32p), ("n_samples", C.c_int64), ("sample_offset", C.c_int64)]


EVENT_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(Event))


class Result(C.Structure):
    _fields_ = [("status", C.c_int), ("pcm", f32p), ("n_samples", C.c_int64), ("codes", i32p),
                ("n_frames", C.c_int32), ("info", GenInfo)]


class Timing(C.Structure):
    _fields_ = [("prefill_ms", C.c_double), ("decode_ms", C.c_double), ("codec_ms", C.c_double),
                ("frame_steps", C.c_int32), ("rows", C.c_int32), ("kv_bytes_read", C.c_int64),
                ("frontend_ms", C.c_double), ("first_audio_ms", C.c_double), ("launches_per_frame_step", C.c_int32)]


_lib = None


def lib() -> C.CDLL:
    """Loads libq3tts_hip.so; raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build the HIP engine first "
                           "(make -C swift-qwen3-tts_amd); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.q3tts_default_load_opts.argtypes = [C.POINTER(LoadOpts)]
    L.q3tts_default_sampling.argtypes = [C.POINTER(Sampling)]
    L.q3tts_model_load.argtypes = [C.c_char_p, C.POINTER(LoadOpts), C.POINTER(vp)]
    L.q3tts_model_free.argtypes = [vp]
    L.q3tts_model_free.restype = None
    L.q3tts_last_error.argtypes = [vp]
    L.q3tts_last_error.restype = C.c_char_p
    L.q3tts_model_arena.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.q3tts_model_get_info.argtypes = [vp, C.POINTER(ModelInfo)]
    L.q3tts_comm_get_unique_id.argtypes = [C.POINTER(CommId)]
    L.q3tts_model_broadcast.argtypes = [vp, C.POINTER(CommId), C.c_int32, C.c_int32, C.c_int32]
    L.q3tts_model_arena_checksum.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.q3tts_model_num_speakers.argtypes = [vp]
    L.q3tts_model_speaker_name.argtypes = [vp, C.c_int32]
    L.q3tts_model_speaker_name.restype = C.c_char_p
    L.q3tts_generate.argtypes = [vp, C.POINTER(Request), C.c_int32, C.POINTER(Sampling), EVENT_CB, vp,
                                 C.POINTER(Result)]
    L.q3tts_generate_begin.argtypes = [vp, C.POINTER(Request), C.c_int32, C.POINTER(Sampling), EVENT_CB, vp, C.c_int32, C.POINTER(vp)]
    L.q3tts_generate_end.argtypes = [vp, vp, C.POINTER(Result)]
    L.q3tts_pcm_to_int16.argtypes = [f32p, C.c_int64, C.POINTER(C.c_int16)]
    L.q3tts_pcm_to_int16.restype = None
    L.q3tts_write_wav.argtypes = [C.c_char_p, f32p, C.c_int64, C.c_int32]
    L.q3tts_result_free.argtypes = [C.POINTER(Result), C.c_int32]
    L.q3tts_result_free.restype = None
    L.q3tts_codec_decode.argtypes = [vp, i32p, i32p, C.c_int32, C.c_int32, f32p, C.POINTER(C.c_int64)]
    L.q3tts_codec_decode_streamed.argtypes = [vp, i32p, i32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, f32p]
    L.q3tts_debug_set_codec_scratch.argtypes = [C.c_uint64]
    L.q3tts_debug_set_codec_scratch.restype = None
    L.q3tts_debug_reload_env.argtypes = []
    L.q3tts_debug_reload_env.restype = None
    L.q3tts_last_timing.argtypes = [vp, C.POINTER(Timing)]
    L.q3tts_codec_encode.argtypes = [vp, f32p, C.c_int64, i32p, C.c_int32, i32p]
    L.q3tts_codec_encoded_frames.argtypes = [vp, C.c_int64]
    L.q3tts_speaker_embedding.argtypes = [vp, f32p, C.c_int64, C.c_int32, f32p, C.c_int32]
    L.q3tts_debug_frontend_stage.argtypes = [vp, f32p, C.c_int64, C.c_char_p, f32p, C.c_int64, i32p, i32p]
    L.q3tts_tokenizer_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.q3tts_tokenizer_free.argtypes = [vp]
    L.q3tts_tokenizer_free.restype = None
    L.q3tts_tokenizer_encode.argtypes = [vp, C.c_char_p, i32p, C.c_int32, i32p]
    L.q3tts_debug_prepare_inputs.argtypes = [vp, C.POINTER(Request), u16p, C.c_int32, i32p, u16p, C.c_int32,
                                             i32p, u16p]
    L.q3tts_debug_generate_forced.argtypes = [vp, C.POINTER(Request), C.c_int32, C.POINTER(Sampling), i32p,
                                              C.c_int32, u16p, u16p, i32p]
    L.q3tts_debug_sample.argtypes = [vp, u16p, C.c_int32, C.c_int32, C.POINTER(Sampling), u8p, C.c_int32,
                                     C.c_int32, C.c_int32, C.c_uint32, C.c_uint32, i32p]
    L.q3tts_debug_linear.argtypes = [vp, u16p, u16p, u16p, C.c_int32, C.c_int32, C.c_int32, u16p]
    L.q3tts_debug_codec_stage.argtypes = [vp, i32p, C.c_int32, C.c_char_p, f32p, C.c_int64, i32p, i32p]
    _lib = L
    return L


def reload_debug_env() -> None:
    """The launchers read their diagnostic switches (Q3TTS_*) once per model load; after changing one on a live model call this."""
    lib().q3tts_debug_reload_env()
