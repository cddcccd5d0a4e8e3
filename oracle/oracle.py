"""CPU oracle for the Qwen3-TTS hot path -- composition layer (TEST INFRASTRUCTURE, NOT PRODUCT).

Heavy arithmetic lives in q3tts_oracle.c; this file restates the reference's control flow:
loader + sanitisers, prompt assembly, the autoregressive loop and the codec-decoder pipeline.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

PARITY UNPINNED (see q3tts_oracle.c header and DESIGN.md): no reference-run fixture exists.

Citations are relative to /root/reference/Sources/Qwen3TTS/Models/.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import struct
import subprocess
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libq3tts_oracle.so")

u16p = C.POINTER(C.c_uint16)
f32p = C.POINTER(C.c_float)
u8p = C.POINTER(C.c_uint8)
i32p = C.POINTER(C.c_int)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "q3tts_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


class _Stack(C.Structure):
    _fields_ = [("hidden", C.c_int), ("n_layers", C.c_int), ("n_heads", C.c_int), ("n_kv", C.c_int),
                ("head_dim", C.c_int), ("eps", C.c_float), ("rope_base", C.c_float),
                ("inter", i32p)] + [(n, C.POINTER(u16p)) for n in
                                    ("ln1", "ln2", "qw", "kw", "vw", "ow", "qn", "kn", "gw", "uw", "dw")] + \
               [("norm", u16p)]


class _Cache(C.Structure):
    _fields_ = [("k", u16p), ("v", u16p), ("len", C.c_int), ("cap", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        # OpenMP team: the GPU boxes expose every hardware thread but grant a CPU share of about 16 cores; a team as
        # large as the machine then spends its time in throttled spin-waits (a tiny forward pass took > 60 s once).
        # Sleep in barriers and cap the team unless the caller chose a size.
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        # tests/test_sanitizers.py points this at an ASan / UBSan build of the same source
        _lib = C.CDLL(os.environ.get("Q3TTS_ORACLE_LIB") or build())
        if "OMP_NUM_THREADS" not in os.environ:
            try:
                ncpu = len(os.sched_getaffinity(0))
            except AttributeError:
                ncpu = os.cpu_count() or 1
            _lib.o_set_num_threads(C.c_int(max(1, min(16, ncpu))))
        _lib.o_gumbel.restype = C.c_float
        _lib.o_gumbel.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        _lib.o_logf.restype = C.c_float
        _lib.o_logf.argtypes = [C.c_float]
        _lib.o_sample_token.restype = C.c_int
        _lib.o_sample_token.argtypes = [u16p, C.c_int, C.c_float, C.c_int, C.c_float, C.c_float, u8p,
                                        C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint32,
                                        C.c_uint32]
        _lib.o_stack_forward.restype = C.c_int
        _lib.o_num_threads.restype = C.c_int
    return _lib


def _p16(a: np.ndarray):
    assert a.dtype == np.uint16 and a.flags.c_contiguous
    return a.ctypes.data_as(u16p)


def _pf(a: Optional[np.ndarray]):
    if a is None:
        return None
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return a.ctypes.data_as(f32p)


def bf16_to_f32(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(x.shape, np.uint16)
    lib().o_f32_to_bf16(_pf(x), C.c_int64(x.size), _p16(out))
    return out


# ----------------------------------------------------------------------------------------------
# safetensors reader (MLX.loadArrays at Qwen3.swift:1397,1478)
# ----------------------------------------------------------------------------------------------
_NP = {"F32": np.float32, "BF16": np.uint16, "I32": np.int32, "U32": np.uint32, "F16": np.float16,
       "I64": np.int64, "U8": np.uint8}


def load_safetensors_dir(d: str) -> Dict[str, np.ndarray]:
    out: Dict[str, np.ndarray] = {}
    for fn in sorted(os.listdir(d)):
        if not fn.endswith(".safetensors"):
            continue
        with open(os.path.join(d, fn), "rb") as f:
            n = struct.unpack("<Q", f.read(8))[0]
            hdr = json.loads(f.read(n))
            base = 8 + n
            for k, m in hdr.items():
                if k == "__metadata__":
                    continue
                a, b = m["data_offsets"]
                f.seek(base + a)
                arr = np.frombuffer(f.read(b - a), dtype=_NP[m["dtype"]]).reshape(m["shape"]).copy()
                out[k] = arr
    return out


# ----------------------------------------------------------------------------------------------
# config defaults (Config.swift:148-158, 292-328, 388-408, 638-651)
# ----------------------------------------------------------------------------------------------
_CP_DEF = dict(vocab_size=2048, hidden_size=1024, intermediate_size=3072, num_hidden_layers=5,
               num_attention_heads=16, num_key_value_heads=8, head_dim=128, num_code_groups=16,
               rms_norm_eps=1e-6, rope_theta=1e6)
_TALKER_DEF = dict(vocab_size=3072, text_vocab_size=151936, hidden_size=2048, text_hidden_size=2048,
                   intermediate_size=6144, num_hidden_layers=28, num_attention_heads=16,
                   num_key_value_heads=8, head_dim=128, num_code_groups=16, rms_norm_eps=1e-6,
                   rope_theta=1e6, codec_eos_token_id=2150, codec_think_id=2154,
                   codec_nothink_id=2155, codec_think_bos_id=2156, codec_think_eos_id=2157,
                   codec_pad_id=2148, codec_bos_id=2149,
                   codec_language_id={"chinese": 2055, "english": 2050, "german": 2053,
                                      "italian": 2070, "portuguese": 2071, "spanish": 2054,
                                      "japanese": 2058, "korean": 2064, "french": 2061,
                                      "russian": 2069})
_MODEL_DEF = dict(tts_model_type="voice_design", tts_pad_token_id=151671, tts_bos_token_id=151672,
                  tts_eos_token_id=151673, sample_rate=24000)
_DEC_DEF = dict(latent_dim=1024, codebook_dim=512, codebook_size=2048, decoder_dim=1536,
                hidden_size=512, intermediate_size=1024, num_hidden_layers=8, num_attention_heads=16,
                num_key_value_heads=16, head_dim=64, rms_norm_eps=1e-5, num_quantizers=16,
                num_semantic_quantizers=1, semantic_codebook_size=4096, upsample_rates=[8, 5, 4, 3],
                upsampling_ratios=[2, 2], layer_scale_initial_scale=0.01)


# Qwen3TTSTokenizerEncoderConfig (Config.swift:476-504), Qwen3TTSSpeakerEncoderConfig (Config.swift:80-91)
_ENC_DEF = dict(frame_rate=12.5, audio_channels=1, codebook_dim=256, codebook_size=2048, compress=2,
                dilation_growth_rate=2, head_dim=64, hidden_size=512, intermediate_size=2048, kernel_size=7,
                last_kernel_size=3, layer_scale_initial_scale=0.01, max_position_embeddings=8000,
                num_attention_heads=8, num_filters=64, num_hidden_layers=8, num_key_value_heads=8,
                num_quantizers=32, num_residual_layers=1, residual_kernel_size=3, rope_theta=10000.0,
                sampling_rate=24000, sliding_window=250, upsampling_ratios=[8, 6, 5, 4], use_causal_conv=True,
                use_conv_shortcut=False)
_SPK_DEF = dict(mel_dim=128, enc_dim=1024, enc_channels=[512, 512, 512, 512, 1536], enc_kernel_sizes=[5, 3, 3, 3, 1],
                enc_dilations=[1, 2, 3, 4, 1], enc_attention_channels=128, enc_res2net_scale=8, enc_se_channels=128,
                sample_rate=24000)


def _with_defaults(d: Optional[dict], defaults: dict) -> dict:
    out = dict(defaults)
    out.update(d or {})
    return out


# ----------------------------------------------------------------------------------------------
# sanitisers
# ----------------------------------------------------------------------------------------------

def _is_mlx_conv_layout(shape) -> bool:
    """checkArrayShapeQwen3 (Qwen3.swift:1246-1260)."""
    _, d2, d3 = shape
    if d2 == 1:
        return d3 > 64
    if d3 == 1:
        return d2 <= 64
    return d2 < d3


_DEC_IDX = {"decoder.decoder.0": "decoder.decoder.initConv", "decoder.decoder.1": "decoder.decoder.block0",
            "decoder.decoder.2": "decoder.decoder.block1", "decoder.decoder.3": "decoder.decoder.block2",
            "decoder.decoder.4": "decoder.decoder.block3", "decoder.decoder.5": "decoder.decoder.outSnake",
            "decoder.decoder.6": "decoder.decoder.outConv"}


_ENC_SEANET = {"encoder.encoder.layers.0.": "encoder.encoder.init_conv1d.",  # Qwen3.swift:1517-1528
               "encoder.encoder.layers.1.": "encoder.encoder.layers.0.residuals.0.",
               "encoder.encoder.layers.3.": "encoder.encoder.layers.0.downsample.",
               "encoder.encoder.layers.4.": "encoder.encoder.layers.1.residuals.0.",
               "encoder.encoder.layers.6.": "encoder.encoder.layers.1.downsample.",
               "encoder.encoder.layers.7.": "encoder.encoder.layers.2.residuals.0.",
               "encoder.encoder.layers.9.": "encoder.encoder.layers.2.downsample.",
               "encoder.encoder.layers.10.": "encoder.encoder.layers.3.residuals.0.",
               "encoder.encoder.layers.12.": "encoder.encoder.layers.3.downsample.",
               "encoder.encoder.layers.14.": "encoder.encoder.final_conv1d."}


def _sanitize_encoder_key(key: str, value: np.ndarray):
    """Encoder half of sanitizeSpeechTokenizerWeights (Qwen3.swift:1592-1700): returns (new_key, new_value)."""
    nk, nv = key, value
    for a, b in _ENC_SEANET.items():  # :1594-1599
        if nk.startswith(a):
            nk = b + nk[len(a):]
            break
    if ".residuals." in nk:  # :1603-1607
        nk = nk.replace(".block.1.", ".block.0.").replace(".block.3.", ".block.1.")
    is_seanet = nk.startswith("encoder.encoder.") and "encoder_transformer" not in nk and "quantizer" not in nk \
        and (".conv.weight" in nk or ".conv.bias" in nk)  # :1612-1615
    if is_seanet:
        nk = nk.replace(".conv.weight", ".conv.conv.weight").replace(".conv.bias", ".conv.conv.bias")
        if nk.endswith(".weight") and value.ndim == 3:  # forced transpose, :1622-1624
            nv = value.transpose(0, 2, 1)
    if "encoder.encoder_transformer.layers." in nk:  # :1629-1649
        nk = nk.replace("encoder.encoder_transformer.layers.", "encoder.encoder_transformer.transformer.layers.")
        for a, b in ((".input_layernorm.", ".norm1."), (".post_attention_layernorm.", ".norm2."),
                     (".mlp.fc1.", ".gating.linear1."), (".mlp.fc2.", ".gating.linear2."),
                     (".self_attn_layer_scale.", ".layer_scale_1."), (".mlp_layer_scale.", ".layer_scale_2.")):
            nk = nk.replace(a, b)
    if nk.startswith("encoder.downsample.conv.") and "encoder.downsample.conv.conv." not in nk:  # :1652-1659
        is_w = nk.endswith(".weight")
        nk = nk.replace("encoder.downsample.conv.", "encoder.downsample.conv.conv.conv.")
        if is_w and value.ndim == 3:
            nv = value.transpose(0, 2, 1)
    if "encoder.quantizer." in nk:  # :1664-1676
        nk = nk.replace(".semantic_residual_vector_quantizer.", ".rvq_first.") \
               .replace(".acoustic_residual_vector_quantizer.", ".rvq_rest.")
        nk = nk.replace(".rvq_first.layers.", ".rvq_first.vq.layers.").replace(".rvq_rest.layers.", ".rvq_rest.vq.layers.")
    was_seanet_w = nk.startswith("encoder.encoder.") and "encoder_transformer" not in nk and "quantizer" not in nk \
        and nk.endswith(".conv.conv.weight")  # :1682-1685
    is_proj = ("input_proj.weight" in nk or "output_proj.weight" in nk) and "quantizer" in nk
    if is_proj and value.ndim == 3:  # :1688-1692
        nv = value.transpose(0, 2, 1)
    if "conv.weight" in nk and value.ndim == 3 and not is_proj and not was_seanet_w:  # :1696-1700
        if not _is_mlx_conv_layout(value.shape):
            nv = value.transpose(0, 2, 1)
    return nk, np.ascontiguousarray(nv)


def sanitize_speech_tokenizer(weights: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """sanitizeSpeechTokenizerWeights (Qwen3.swift:1498-1750), decoder and encoder halves."""
    out: Dict[str, np.ndarray] = {}
    cb: Dict[str, Dict[str, np.ndarray]] = {}
    ecb: Dict[str, Dict[str, np.ndarray]] = {}
    for key, value in weights.items():
        if "._codebook.cluster_usage" in key or "._codebook.embedding_sum" in key:  # :1532-1543
            base = key.split("._codebook.")[0]
            cb.setdefault(base, {})["cluster_usage" if "cluster_usage" in key else "embedding_sum"] = value
            continue
        if key.startswith("encoder.quantizer.") and ".codebook." in key:  # :1546-1565
            parts = key.split(".codebook.")
            if len(parts) == 2 and parts[1] in ("embed_sum", "cluster_usage"):
                ecb.setdefault(parts[0], {})[parts[1]] = value
                continue
            if ".initialized" in key:
                continue
        if key.startswith("encoder."):
            nk, nv = _sanitize_encoder_key(key, value)
            out[nk] = nv
            continue
        nk = key
        for a, b in _DEC_IDX.items():  # :1573-1578
            if key.startswith(a + "."):
                nk = b + key[len(a):]
                break
        if nk.startswith("decoder."):  # :1581-1588
            for a, b in ((".block.0.", ".snake."), (".block.1.", ".upsample."), (".block.2.", ".res1."),
                         (".block.3.", ".res2."), (".block.4.", ".res3.")):
                nk = nk.replace(a, b)
        nv = value
        is_proj = ("input_proj.weight" in nk or "output_proj.weight" in nk) and "quantizer" in nk
        if is_proj and value.ndim == 3:  # :1688-1692
            nv = value.transpose(0, 2, 1)
        if "conv.weight" in nk and value.ndim == 3 and not is_proj:  # :1696-1700
            if not _is_mlx_conv_layout(value.shape):
                nv = value.transpose(0, 2, 1)
        is_tr = ("upsample" in nk and ".0.conv.weight" in nk) or \
                ("decoder.decoder.block" in nk and "upsample.conv.weight" in nk)  # :1704-1711
        if is_tr and value.ndim == 3 and not _is_mlx_conv_layout(value.shape):
            nv = value.transpose(1, 2, 0)
        out[nk] = np.ascontiguousarray(nv)
    for base, d in cb.items():  # :1716-1724
        if "cluster_usage" in d and "embedding_sum" in d:
            usage = np.clip(d["cluster_usage"].astype(np.float32)[:, None], np.float32(1e-5), None)
            out[base + ".codebook.embed.weight"] = (d["embedding_sum"].astype(np.float32) / usage).astype(np.float32)
    for base, d in ecb.items():  # :1727-1747; EncoderEuclideanCodebook.updateInPlace (SpeechTokenizerEncoder.swift:738-743)
        if "cluster_usage" in d and "embed_sum" in d:
            nb = base.replace(".semantic_residual_vector_quantizer.", ".rvq_first.") \
                     .replace(".acoustic_residual_vector_quantizer.", ".rvq_rest.")
            nb = nb.replace(".rvq_first.layers.", ".rvq_first.vq.layers.", 1).replace(".rvq_rest.layers.", ".rvq_rest.vq.layers.", 1)
            out[nb + ".codebook.embeddingSum"] = d["embed_sum"].astype(np.float32)
            out[nb + ".codebook.clusterUsage"] = d["cluster_usage"].astype(np.float32)
    return out


def sanitize_main(weights: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """Qwen3TTSModel.sanitize (Qwen3.swift:1219-1243): drop position_ids, conv weights to MLX layout
    (only the speaker encoder has 3-d conv weights in the main checkpoint)."""
    out = {}
    for k, v in weights.items():
        if "position_ids" in k:
            continue
        if ("conv" in k or "speaker_encoder.fc" in k) and "weight" in k and v.ndim == 3 and not _is_mlx_conv_layout(v.shape):
            v = np.ascontiguousarray(v.transpose(0, 2, 1))
        out[k] = v
    return out


def dequantize_mlx_affine(w: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """QuantizedLinear / QuantizedEmbedding as installed by quantize(model:...) (Qwen3.swift:1412-1425), MLX affine
    mode: w[n][k] = q*scale + bias, q = k-th 4-bit field (little-endian, 8 per uint32) of row n, one bf16
    scale/bias per 64 inputs; the dequantised weight is an array of the model dtype (bf16). Dequantising once at
    load is arithmetically identical to o_qlinear_bf16 (checked in tests/test_quantized.py)."""
    out = dict(w)
    for key in [k for k in w if k.endswith(".scales")]:
        base = key[: -len(".scales")]
        packed, sc, bi = w[base + ".weight"], bf16_to_f32(w[key]), bf16_to_f32(w[base + ".biases"])
        n, k8 = packed.shape
        q = ((packed[:, :, None] >> (4 * np.arange(8, dtype=np.uint32))[None, None, :]) & 15).reshape(n, k8 * 8)
        prod = (q.astype(np.float32) * np.repeat(sc, 64, axis=1)).astype(np.float32)   # mul and add rounded separately
        out[base + ".weight"] = f32_to_bf16((prod + np.repeat(bi, 64, axis=1)).astype(np.float32))
        del out[key], out[base + ".biases"]
    return out


# ----------------------------------------------------------------------------------------------
# model
# ----------------------------------------------------------------------------------------------

def pcm_to_int16(pcm: np.ndarray) -> np.ndarray:
    """The reference CLI's WAV quantisation (Sources/Qwen3TTSDemo/main.swift:158-162): clamp to [-1, 1], multiply by
    32767 in Float, Int16(_:) truncates toward zero."""
    x = np.clip(np.asarray(pcm, np.float32), np.float32(-1.0), np.float32(1.0))
    return np.trunc((x * np.float32(32767.0)).astype(np.float32)).astype(np.int16)


@dataclass
class Sampling:
    temperature: float = 0.9
    top_k: int = 50
    top_p: float = 1.0
    repetition_penalty: float = 1.05
    seed: int = 0
    force_frames: int = 0  # bench only: EOS masked, exactly this many frames


@dataclass
class Request:
    text_ids: List[int]
    target_token_count: int
    instruct_ids: Optional[List[int]] = None
    speaker: Optional[str] = None
    language: str = "auto"
    ref_audio: Optional[np.ndarray] = None      # voice clone: 24 kHz mono float32 (Qwen3.swift:1009-1013)
    ref_text_ids: Optional[List[int]] = None    # tokens of "<|im_start|>assistant\n{refText}<|im_end|>\n" (:448-449)
    max_tokens: int = 2048
    # tests only: reference codes [16][T] to use instead of encoding ref_audio (the encoder is checked on its own; a
    # near-tie in its RVQ search may legitimately flip between summation orders and would change the whole prompt)
    ref_codes_override: Optional[np.ndarray] = None


@dataclass
class GenTrace:
    codes: np.ndarray                       # [F][16] int32
    talker_logits: List[np.ndarray] = field(default_factory=list)  # per frame, bf16 bits [V]
    cp_logits: List[np.ndarray] = field(default_factory=list)      # per frame [15][Vcp]
    hit_eos: bool = False
    ref_codes: Optional[np.ndarray] = None  # voice clone: [16][T_ref]


class _StackHolder:
    """Owns the pointer tables of an o_stack."""

    def __init__(self, w: Dict[str, np.ndarray], prefix: str, hidden, inter, n_layers, n_heads, n_kv,
                 head_dim, eps, base):
        self.keep = []
        s = _Stack()
        s.hidden, s.n_layers, s.n_heads, s.n_kv, s.head_dim = hidden, n_layers, n_heads, n_kv, head_dim
        s.eps, s.rope_base = eps, base
        self.inter = np.asarray(inter, np.int32)
        s.inter = self.inter.ctypes.data_as(i32p)
        names = dict(ln1="input_layernorm", ln2="post_attention_layernorm", qw="self_attn.q_proj",
                     kw="self_attn.k_proj", vw="self_attn.v_proj", ow="self_attn.o_proj",
                     qn="self_attn.q_norm", kn="self_attn.k_norm", gw="mlp.gate_proj", uw="mlp.up_proj",
                     dw="mlp.down_proj")
        for f, n in names.items():
            arr = (u16p * n_layers)()
            for l in range(n_layers):
                a = w[f"{prefix}.layers.{l}.{n}.weight"]
                arr[l] = _p16(a)
            self.keep.append(arr)
            setattr(s, f, arr)
        s.norm = _p16(w[f"{prefix}.norm.weight"])
        self.s = s
        self.kd = n_kv * head_dim
        self.n_layers = n_layers

    def new_cache(self, cap: int):
        k = np.zeros((self.n_layers, cap, self.kd), np.uint16)
        v = np.zeros((self.n_layers, cap, self.kd), np.uint16)
        c = _Cache(_p16(k), _p16(v), 0, cap)
        return c, (k, v)

    def forward(self, cache, x: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x, np.uint16)
        out = np.empty_like(x)
        rc = lib().o_stack_forward(C.byref(self.s), C.byref(cache[0]), _p16(x), C.c_int(x.shape[0]), _p16(out))
        assert rc == 0, "oracle KV cache overflow"
        return out


class OracleModel:
    """Qwen3TTSModel restated (fromPretrained: Qwen3.swift:1382-1495)."""

    def __init__(self, model_dir: str):
        L = lib()
        with open(os.path.join(model_dir, "config.json")) as f:
            raw = json.load(f)
        self.cfg = _with_defaults(raw, _MODEL_DEF)
        self.t = _with_defaults(raw.get("talker_config"), _TALKER_DEF)
        self.cp = _with_defaults(self.t.get("code_predictor_config"), _CP_DEF)
        w = load_safetensors_dir(model_dir)
        if raw.get("quantization"):
            q = raw["quantization"]
            assert q.get("bits", 4) == 4 and q.get("group_size", 64) == 64
            w = dequantize_mlx_affine(w)
        w = sanitize_main(w)  # Qwen3.swift:1219-1243
        self.token_map = w.pop("talker.model.text_token_map", None)  # :1434-1444
        self.w = w
        t, cp = self.t, self.cp
        self.H, self.V = t["hidden_size"], t["vocab_size"]
        inter = t.get("per_layer_intermediate_sizes") or [t["intermediate_size"]] * t["num_hidden_layers"]
        self.talker = _StackHolder(w, "talker.model", self.H, inter, t["num_hidden_layers"],
                                   t["num_attention_heads"], t["num_key_value_heads"], t["head_dim"],
                                   t["rms_norm_eps"], t["rope_theta"])
        self.cpm = _StackHolder(w, "talker.code_predictor.model", cp["hidden_size"],
                                [cp["intermediate_size"]] * cp["num_hidden_layers"], cp["num_hidden_layers"],
                                cp["num_attention_heads"], cp["num_key_value_heads"], cp["head_dim"],
                                cp["rms_norm_eps"], cp["rope_theta"])
        self.has_proj = "talker.code_predictor.small_to_mtp_projection.weight" in w
        st_dir = os.path.join(model_dir, "speech_tokenizer")
        self.codec = None
        self.ec = None
        if os.path.isdir(st_dir):  # postLoadHook, Qwen3.swift:1462-1494
            with open(os.path.join(st_dir, "config.json")) as f:
                sraw = json.load(f)
            self.dc = _with_defaults(sraw.get("decoder_config"), _DEC_DEF)
            self.ec = _with_defaults(sraw["encoder_config"], _ENC_DEF) if sraw.get("encoder_config") else None
            raw_codec = load_safetensors_dir(st_dir)
            # "lite" checkpoints store the speech tokenizer in float16 (docs/paper.tex:207): the reference then computes the decoder
            # in float16 (MLX promotes nothing: fp16 weights x fp16 activations). codec_decode(..., f16=True) restates that.
            self.codec_f16 = any(k.startswith("decoder.decoder.") and v.dtype == np.float16 for k, v in raw_codec.items())
            self.codec = {k: (v.astype(np.float32) if v.dtype == np.float16 else v)
                          for k, v in sanitize_speech_tokenizer(raw_codec).items()}
        self.sc = _with_defaults(raw["speaker_encoder_config"], _SPK_DEF) if raw.get("speaker_encoder_config") else None
        _ = L

    # -- small helpers ---------------------------------------------------------------------
    def linear(self, x: np.ndarray, name: str, bias: bool = False) -> np.ndarray:
        W = self.w[name + ".weight"]
        b = self.w.get(name + ".bias") if bias else None
        x = np.ascontiguousarray(x, np.uint16)
        M, K = x.shape
        N = W.shape[0]
        assert W.shape[1] == K
        out = np.empty((M, N), np.uint16)
        lib().o_linear_bf16(_p16(x), _p16(W), _p16(b) if b is not None else None, C.c_int(M), C.c_int(K),
                            C.c_int(N), _p16(out))
        return out

    @staticmethod
    def add(a: np.ndarray, b: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(a, np.uint16)
        b = np.ascontiguousarray(np.broadcast_to(b, a.shape), np.uint16)
        out = np.empty_like(a)
        lib().o_add_bf16(_p16(a), _p16(b), C.c_int64(a.size), _p16(out))
        return out

    def embed_text(self, ids) -> np.ndarray:
        """embedText (Talker.swift:627-633)."""
        ids = np.asarray(ids, np.int64)
        if self.token_map is not None:
            ids = self.token_map[ids].astype(np.int64)
        return self.w["talker.model.text_embedding.weight"][ids]

    def text_projection(self, x: np.ndarray) -> np.ndarray:
        """ResizeMLP (Talker.swift:475-487)."""
        h = self.linear(x, "talker.text_projection.linear_fc1", True)
        s = np.empty_like(h)
        lib().o_silu_bf16(_p16(h), C.c_int64(h.size), _p16(s))
        return self.linear(s, "talker.text_projection.linear_fc2", True)

    def codec_embed(self, ids) -> np.ndarray:
        return self.w["talker.model.codec_embedding.weight"][np.asarray(ids, np.int64)]

    def cp_embed(self, i: int, ids) -> np.ndarray:
        return self.w[f"talker.code_predictor.model.codec_embedding.{i}.weight"][np.asarray(ids, np.int64)]

    # -- prompt assembly ---------------------------------------------------------------------
    def resolve_language(self, language: str, speaker: Optional[str]) -> Optional[int]:
        """Qwen3.swift:303-319."""
        t = self.t
        lang = language.lower()
        lid = None
        if lang != "auto":
            lid = t["codec_language_id"].get(lang)
        if lang in ("chinese", "auto") and speaker is not None and t.get("spk_is_dialect"):
            dv = t["spk_is_dialect"].get(speaker.lower())
            if isinstance(dv, str):
                did = t["codec_language_id"].get(dv)
                if did is not None:
                    lid = did
        return lid

    def prepare_generation_inputs(self, req: Request):
        """prepareGenerationInputs (Qwen3.swift:259-409). Returns bf16-bit arrays
        (input_embeds [P][H], trailing_text_hidden [n][H], tts_pad_embed [1][H])."""
        t, cfg = self.t, self.cfg
        text_embed = self.text_projection(self.embed_text(req.text_ids))
        tts = self.text_projection(self.embed_text([cfg["tts_bos_token_id"], cfg["tts_eos_token_id"],
                                                    cfg["tts_pad_token_id"]]))
        tts_bos, tts_eos, tts_pad = tts[0:1], tts[1:2], tts[2:3]
        speaker_embed = None
        if req.speaker is not None and t.get("spk_id") and req.speaker.lower() in t["spk_id"]:
            speaker_embed = self.codec_embed([t["spk_id"][req.speaker.lower()]])
        lid = self.resolve_language(req.language, req.speaker)
        if lid is None:
            prefill = [t["codec_nothink_id"], t["codec_think_bos_id"], t["codec_think_eos_id"]]
        else:
            prefill = [t["codec_think_id"], t["codec_think_bos_id"], lid, t["codec_think_eos_id"]]
        codec_embed = self.codec_embed(prefill)
        suffix = self.codec_embed([t["codec_pad_id"], t["codec_bos_id"]])
        parts = [codec_embed] + ([speaker_embed] if speaker_embed is not None else []) + [suffix]
        codec_embed = np.concatenate(parts, 0)
        instruct_embed = None
        if req.instruct_ids:
            instruct_embed = self.text_projection(self.embed_text(req.instruct_ids))
        role = text_embed[0:3]
        n = codec_embed.shape[0]
        combined = np.concatenate([np.repeat(tts_pad, n - 2, 0), tts_bos], 0)
        combined = self.add(combined, codec_embed[: n - 1])
        parts = ([instruct_embed] if instruct_embed is not None else []) + [role, combined]
        first_text = self.add(text_embed[3:4], codec_embed[n - 1:])
        input_embeds = np.concatenate(parts + [first_text], 0)
        tl = text_embed.shape[0]
        if tl - 5 > 4:
            trailing = np.concatenate([text_embed[4: tl - 5], tts_eos], 0)
        else:
            trailing = tts_eos
        return (np.ascontiguousarray(input_embeds), np.ascontiguousarray(trailing),
                np.ascontiguousarray(tts_pad))

    # -- forward pieces ----------------------------------------------------------------------
    def talker_forward(self, cache, x: np.ndarray):
        """Qwen3TTSTalkerForConditionalGeneration.callAsFunction (Talker.swift:637-646)."""
        hidden = self.talker.forward(cache, x)
        logits = self.linear(hidden, "talker.codec_head")
        return logits, hidden

    def cp_forward(self, cache, x: np.ndarray, step: int) -> np.ndarray:
        """Qwen3TTSCodePredictor.callAsFunction (CodePredictor.swift:320-339)."""
        h = x
        if self.has_proj:
            h = self.linear(h, "talker.code_predictor.small_to_mtp_projection", True)
        h = self.cpm.forward(cache, h)
        return self.linear(h, f"talker.code_predictor.lm_head.{step}")

    def sample(self, logits_row: np.ndarray, s: Sampling, seen: Optional[np.ndarray], suppress_lo: int,
               suppress_hi: int, eos: int, row: int, draw: int, rep_penalty: float) -> int:
        lr = np.ascontiguousarray(logits_row, np.uint16)
        return int(lib().o_sample_token(
            _p16(lr), C.c_int(lr.size), C.c_float(s.temperature), C.c_int(s.top_k), C.c_float(s.top_p),
            C.c_float(rep_penalty), seen.ctypes.data_as(u8p) if seen is not None else None,
            C.c_int(suppress_lo), C.c_int(suppress_hi), C.c_int(eos), C.c_int(1 if s.force_frames else 0),
            C.c_uint64(s.seed), C.c_uint32(row), C.c_uint32(draw)))

    def effective_max_tokens(self, req: Request, s: Sampling) -> int:
        if s.force_frames:
            return s.force_frames
        return min(req.max_tokens, max(75, req.target_token_count * 6))  # Qwen3.swift:822-823

    # -- AR loop -----------------------------------------------------------------------------
    def generate_codes(self, req: Request, s: Sampling, row: int = 0,
                       forced_codes: Optional[np.ndarray] = None, keep_logits: bool = False) -> GenTrace:
        """The AR loop of generateCustomVoice / generateVoiceDesign (Qwen3.swift:847-936,
        640-729). `forced_codes` teacher-forces the sampled tokens (tests only): logits are
        still produced by the oracle, the fed-back tokens come from the array.
        RNG draw index: frame*16 + codebook."""
        t, cp = self.t, self.cp
        ref_codes = None
        if req.ref_audio is not None:
            inp, trailing, tts_pad, ref_codes = self.prepare_icl_generation_inputs(req)
        else:
            inp, trailing, tts_pad = self.prepare_generation_inputs(req)
        max_tok = self.effective_max_tokens(req, s)
        if forced_codes is not None:
            max_tok = forced_codes.shape[0]
        eos = t["codec_eos_token_id"]
        V = self.V
        cache = self.talker.new_cache(inp.shape[0] + max_tok + 1)
        seen = np.zeros(V, np.uint8)
        codes: List[List[int]] = []
        tr = GenTrace(codes=np.zeros((0, 16), np.int32))
        tr.ref_codes = ref_codes
        cur = inp
        trailing_idx = 0
        ncg = t["num_code_groups"]
        for frame in range(max_tok):
            logits, hidden = self.talker_forward(cache, cur)
            if keep_logits:
                tr.talker_logits.append(logits[-1].copy())
            tok = self.sample(logits[-1], s, seen, V - 1024, V, eos, row, frame * 16, s.repetition_penalty)
            if forced_codes is not None:
                tok = int(forced_codes[frame, 0])
            if tok < V:
                seen[tok] = 1
            if tok == eos:
                tr.hit_eos = True
                break
            frame_codes = [tok]
            code_hidden = hidden[-1:]
            cpc = self.cpm.new_cache(ncg + 1)  # fresh cache per frame, Qwen3.swift:879
            cpl = []
            for ci in range(ncg - 1):
                if ci == 0:
                    x = np.concatenate([code_hidden, self.codec_embed([tok])], 0)  # :884-887
                else:
                    x = self.cp_embed(ci - 1, [frame_codes[ci]])  # :889-892
                cl = self.cp_forward(cpc, x, ci)
                if keep_logits:
                    cpl.append(cl[-1].copy())
                c = self.sample(cl[-1], s, None, 0, 0, -1, row, frame * 16 + 1 + ci, 1.0)
                if forced_codes is not None:
                    c = int(forced_codes[frame, 1 + ci])
                frame_codes.append(c)
            if keep_logits:
                tr.cp_logits.append(np.stack(cpl))
            codes.append(frame_codes)
            if trailing_idx < trailing.shape[0]:  # :919-925
                text_e = trailing[trailing_idx: trailing_idx + 1]
                trailing_idx += 1
            else:
                text_e = tts_pad
            ce = self.codec_embed([tok])  # :928-933, left-to-right bf16 adds
            for i, c in enumerate(frame_codes[1:]):
                ce = self.add(ce, self.cp_embed(i, [c]))
            cur = self.add(text_e, ce)  # :935
        tr.codes = np.asarray(codes, np.int32).reshape(-1, ncg)
        return tr

    # -- codec decoder ------------------------------------------------------------------------
    def _conv(self, x, prefix, K, dil=1, groups=1, with_bias=True):
        W, b = self.codec[prefix + ".weight"], (self.codec.get(prefix + ".bias") if with_bias else None)
        T, Cin = x.shape
        Cout = W.shape[0]
        assert W.shape[1] == K and W.shape[2] == Cin // groups, (prefix, W.shape, K, Cin)
        out = np.empty((T, Cout), np.float32)
        lib().o_conv1d_causal(_pf(np.ascontiguousarray(x)), _pf(W), _pf(b), C.c_int(T), C.c_int(Cin),
                              C.c_int(Cout), C.c_int(K), C.c_int(dil), C.c_int(groups), _pf(out))
        return out

    def _convtr(self, x, prefix, K, stride, with_bias=True):
        W, b = self.codec[prefix + ".weight"], (self.codec.get(prefix + ".bias") if with_bias else None)
        T, Cin = x.shape
        Cout = W.shape[0]
        assert W.shape[1] == K and W.shape[2] == Cin, (prefix, W.shape)
        out = np.empty((T * stride, Cout), np.float32)
        lib().o_convtr1d_causal(_pf(np.ascontiguousarray(x)), _pf(W), _pf(b), C.c_int(T), C.c_int(Cin),
                                C.c_int(Cout), C.c_int(K), C.c_int(stride), _pf(out))
        return out

    def _lin(self, x, prefix, bias):
        W = self.codec[prefix + ".weight"]
        b = self.codec.get(prefix + ".bias") if bias else None
        M, K = x.shape
        out = np.empty((M, W.shape[0]), np.float32)
        lib().o_linear_f32(_pf(np.ascontiguousarray(x)), _pf(W), _pf(b), C.c_int(M), C.c_int(K),
                           C.c_int(W.shape[0]), _pf(out))
        return out

    def _snake(self, x, prefix):
        out = np.empty_like(x)
        lib().o_snake(_pf(np.ascontiguousarray(x)), _pf(self.codec[prefix + ".alpha"]),
                      _pf(self.codec[prefix + ".beta"]), C.c_int(x.shape[0]), C.c_int(x.shape[1]), _pf(out))
        return out

    def _rms(self, x, prefix):
        out = np.empty_like(x)
        lib().o_rmsnorm_f32(_pf(np.ascontiguousarray(x)), _pf(self.codec[prefix + ".weight"]),
                            C.c_float(self.dc["rms_norm_eps"]), C.c_int(x.shape[0]), C.c_int(x.shape[1]), _pf(out))
        return out

    def _dec_transformer_layer(self, x, p, nh, hd):
        """DecoderTransformerLayer (SpeechTokenizer.swift:567-602): RMSNorm -> MHA without positions or mask (:512-528)
        -> LayerScale -> + x -> RMSNorm -> SwiGLU -> LayerScale -> + x. x: [F][hidden]."""
        cw = self.codec
        F = x.shape[0]
        xn = self._rms(x, p + ".input_layernorm")
        q = self._lin(xn, p + ".self_attn.q_proj", False)
        k = self._lin(xn, p + ".self_attn.k_proj", False)
        v = self._lin(xn, p + ".self_attn.v_proj", False)
        ao = np.empty_like(q)
        lib().o_attention_full_f32(_pf(q), _pf(k), _pf(v), C.c_int(F), C.c_int(nh), C.c_int(hd), _pf(ao))
        y = self._lin(ao, p + ".self_attn.o_proj", False)
        x = (x + y * cw[p + ".self_attn_layer_scale.scale"]).astype(np.float32)
        xn = self._rms(x, p + ".post_attention_layernorm")
        g = self._lin(xn, p + ".mlp.gate_proj", False)
        u = self._lin(xn, p + ".mlp.up_proj", False)
        a = np.empty_like(g)
        lib().o_silu_mul_f32(_pf(g), _pf(u), C.c_int64(g.size), _pf(a))
        y = self._lin(a, p + ".mlp.down_proj", False)
        return (x + y * cw[p + ".mlp_layer_scale.scale"]).astype(np.float32)

    def _convnext(self, h, p):
        """ConvNeXtBlock (SpeechTokenizer.swift:359-402): depthwise causal k7 -> LayerNorm(1e-6) -> Linear C->4C ->
        GELU(erf) -> Linear 4C->C -> * gamma -> + residual. h: [T][C] channels-last."""
        cw = self.codec
        res = h
        d = self._conv(h, p + ".dwconv.conv", 7, groups=h.shape[1])
        n = np.empty_like(d)
        lib().o_layernorm_f32(_pf(d), _pf(cw[p + ".norm.weight"]), _pf(cw[p + ".norm.bias"]),
                              C.c_float(1e-6), C.c_int(d.shape[0]), C.c_int(d.shape[1]), _pf(n))
        a = self._lin(n, p + ".pwconv1", True)
        ga = np.empty_like(a)
        lib().o_gelu_f32(_pf(a), C.c_int64(a.size), _pf(ga))
        b2 = self._lin(ga, p + ".pwconv2", True)
        return (res + cw[p + ".gamma"] * b2).astype(np.float32)

    def _resunit(self, h, rp, dil):
        """DecoderResidualUnit (SpeechTokenizer.swift:408-438): snake -> conv k7 dilated -> snake -> conv k1 -> + x."""
        y = self._snake(h, rp + ".act1")
        y = self._conv(y, rp + ".conv1.conv", 7, dil=dil)
        y = self._snake(y, rp + ".act2")
        y = self._conv(y, rp + ".conv2.conv", 1)
        return (h + y).astype(np.float32)

    def _decoder_block(self, h, p, rate):
        """DecoderBlock (SpeechTokenizer.swift:444-481): snake -> transposed conv k=2r, stride r (right trim) -> three
        residual units with dilations 1, 3, 9."""
        h = self._snake(h, p + ".snake")
        h = self._convtr(h, p + ".upsample.conv", 2 * rate, rate)
        for j, dil in ((1, 1), (2, 3), (3, 9)):
            h = self._resunit(h, f"{p}.res{j}", dil)
        return h

    def _codec_front(self, codes: np.ndarray, stages: Optional[dict] = None) -> np.ndarray:
        """Steps 1-4 of Qwen3TTSSpeechTokenizerDecoder.callAsFunction (SpeechTokenizer.swift:757-765): split-RVQ
        dequantisation, pre_conv (causal k3), pre_transformer over ALL the frames it is given (no mask, no positions:
        :512-528, :763). codes [F][16] -> [F][latent]."""
        dc, cw = self.dc, self.codec
        codes = np.asarray(codes, np.int64)
        F = codes.shape[0]
        nsem = dc["num_semantic_quantizers"]

        def rvq(name, cols):  # ResidualVectorQuantizer.decode, :161-169 with :81-96
            q = None
            for j, col in enumerate(cols):
                e = cw[f"decoder.quantizer.{name}.vq.layers.{j}.codebook.embed.weight"][codes[:, col]]
                q = e if q is None else (q + e).astype(np.float32)
            W = cw[f"decoder.quantizer.{name}.output_proj.weight"]  # [out][1][in]
            out = np.empty((F, W.shape[0]), np.float32)
            lib().o_conv1d_causal(_pf(np.ascontiguousarray(q, np.float32)), _pf(W), None, C.c_int(F),
                                  C.c_int(W.shape[2]), C.c_int(W.shape[0]), C.c_int(1), C.c_int(1), C.c_int(1), _pf(out))
            return out

        h = rvq("rvq_first", list(range(nsem)))
        if codes.shape[1] > nsem:  # :220-223
            h = (h + rvq("rvq_rest", list(range(nsem, codes.shape[1])))).astype(np.float32)
        if stages is not None:
            stages["quantizer"] = h
        h = self._conv(h, "decoder.pre_conv.conv", 3)
        if stages is not None:
            stages["pre_conv"] = h
        # pre_transformer (:629-643), no mask, no positions
        nh, hd = dc["num_attention_heads"], dc["head_dim"]
        pt = "decoder.pre_transformer"
        x = self._lin(h, pt + ".input_proj", True)
        for l in range(dc["num_hidden_layers"]):
            x = self._dec_transformer_layer(x, f"{pt}.layers.{l}", nh, hd)
        x = self._rms(x, pt + ".norm")
        h = self._lin(x, pt + ".output_proj", True)
        if stages is not None:
            stages["pre_transformer"] = h
        return h

    def _codec_tail(self, h: np.ndarray, stages: Optional[dict] = None) -> np.ndarray:
        """Steps 5-7 (:767-781): upsample stages, MainDecoder, clip. Every layer here is causal (:298-301, :346-351).
        [F][latent] -> pcm [F * upsample]."""
        dc = self.dc
        for i, r in enumerate(dc["upsampling_ratios"]):  # :767-775
            h = self._convtr(h, f"decoder.upsample.{i}.0.conv", r, r)
            h = self._convnext(h, f"decoder.upsample.{i}.1")
            if stages is not None:
                stages[f"upsample{i}"] = h
        h = self._conv(h, "decoder.decoder.initConv.conv", 7)  # MainDecoder :681-690
        if stages is not None:
            stages["init_conv"] = h
        for b, rate in enumerate(dc["upsample_rates"]):
            h = self._decoder_block(h, f"decoder.decoder.block{b}", rate)
            if stages is not None:
                stages[f"block{b}"] = h
        h = self._snake(h, "decoder.decoder.outSnake")
        h = self._conv(h, "decoder.decoder.outConv.conv", 7)
        return np.clip(h[:, 0], -1.0, 1.0).astype(np.float32)  # :781

    # -- the MainDecoder the way the reference computes it on a float16 checkpoint --------------------------------------------
    # MLX evaluates every op in the arrays' dtype: with float16 weights the decoder's tensors are float16 and EVERY op result is
    # rounded to float16 (conv, + bias, x * alpha, sin, s * s, (1 / beta) * q, x + r, residual + h: SpeechTokenizer.swift:246-253,
    # 298-306, 346-352, 430-437). What MLX does INSIDE an op is not visible from the Swift: contractions are taken with fp32
    # accumulation and sin in fp32 before the rounding (documented choice, like the bf16 stacks'). Restated from initConv on
    # (96 % of the decoder's arithmetic): the front end and the two ConvNeXt stages stay in fp32, as in the engine -- wider than
    # the reference there, by less than one float16 rounding of the tensor that enters initConv.
    @staticmethod
    def _r16(a):
        return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)

    def _snake16(self, x, prefix):
        r16 = self._r16
        ea = r16(np.exp(r16(self.codec[prefix + ".alpha"])))
        be = r16(np.exp(r16(self.codec[prefix + ".beta"])))            # (+ eps: Float 1e-9 is 0 in float16)
        ib = r16(np.float32(1.0) / be)
        t = r16(x * ea)
        s = r16(np.sin(t.astype(np.float32)))
        q = r16(s * s)
        return r16(x + r16(ib * q))

    def _conv16(self, x, prefix, K, dil=1):
        o = self._r16(self._conv(x, prefix, K, dil=dil, with_bias=False))
        b = self.codec.get(prefix + ".bias")
        return o if b is None else self._r16(o + self._r16(b))

    def _convtr16(self, x, prefix, K, stride):
        o = self._r16(self._convtr(x, prefix, K, stride, with_bias=False))
        b = self.codec.get(prefix + ".bias")
        return o if b is None else self._r16(o + self._r16(b))

    def _main_decoder16(self, h, stages=None):
        dc, r16 = self.dc, self._r16
        h = self._conv16(r16(h), "decoder.decoder.initConv.conv", 7)
        if stages is not None:
            stages["init_conv"] = h
        for b, rate in enumerate(dc["upsample_rates"]):
            p = f"decoder.decoder.block{b}"
            h = self._snake16(h, p + ".snake")
            h = self._convtr16(h, p + ".upsample.conv", 2 * rate, rate)
            for j, dil in ((1, 1), (2, 3), (3, 9)):
                rp = f"{p}.res{j}"
                y = self._snake16(h, rp + ".act1")
                y = self._conv16(y, rp + ".conv1.conv", 7, dil=dil)
                y = self._snake16(y, rp + ".act2")
                y = self._conv16(y, rp + ".conv2.conv", 1)
                h = r16(h + y)
            if stages is not None:
                stages[f"block{b}"] = h
        h = self._snake16(h, "decoder.decoder.outSnake")
        h = self._conv16(h, "decoder.decoder.outConv.conv", 7)
        return np.clip(h[:, 0], -1.0, 1.0).astype(np.float32)

    def codec_decode(self, codes: np.ndarray, stages: Optional[dict] = None, f16: bool = False):
        """Qwen3TTSSpeechTokenizer.decode for one utterance (SpeechTokenizer.swift:823-836 ->
        754-784). codes [F][16] int. Returns (pcm float32 [1920*F], valid_len). Intermediate
        activations ([T][C] channels-last) are stored into `stages` when given. f16: the MainDecoder in float16 as the reference
        runs a float16 ("lite") speech tokenizer (_main_decoder16)."""
        dc = self.dc
        codes = np.asarray(codes, np.int64)
        if f16:
            h = self._codec_front(codes, stages)
            for i, r in enumerate(dc["upsampling_ratios"]):
                h = self._convtr(h, f"decoder.upsample.{i}.0.conv", r, r)
                h = self._convnext(h, f"decoder.upsample.{i}.1")
                if stages is not None:
                    stages[f"upsample{i}"] = h
            pcm = self._main_decoder16(h, stages)
            up = int(np.prod(dc["upsample_rates"]) * np.prod(dc["upsampling_ratios"]))
            return pcm, int((codes[:, 0] > 0).sum()) * up
        pcm = self._codec_tail(self._codec_front(codes, stages), stages)
        up = int(np.prod(dc["upsample_rates"]) * np.prod(dc["upsampling_ratios"]))
        valid = int((codes[:, 0] > 0).sum()) * up  # :831-833
        return pcm, valid

    def _codec_tail16(self, h: np.ndarray) -> np.ndarray:
        """_codec_tail for a float16 speech tokenizer: the upsample stages as above, the MainDecoder in float16 (codec_decode's
        f16 branch, as one function so that the streamed restatement below can share it)."""
        dc = self.dc
        for i, r in enumerate(dc["upsampling_ratios"]):
            h = self._convtr(h, f"decoder.upsample.{i}.0.conv", r, r)
            h = self._convnext(h, f"decoder.upsample.{i}.1")
        return self._main_decoder16(h)

    def codec_decode_streamed(self, codes: np.ndarray, chunk: int, window: int, lookahead: int, f16: bool = False) -> np.ndarray:
        """What a STREAM can compute of the decode above (row f1 of SURVEY 8f; the reference has no streaming decode -- gap noted
        at README.md:140 -- so this restates the engine's definition, include/q3tts.h `audio_window_frames`, as a composition
        of the reference's own functions). The tail (steps 5-7) is causal, so feeding it frame by frame with carried conv
        state IS the one-shot tail over the concatenated latents. The pre_transformer is bidirectional over whatever it is
        given (:763): for chunk [f0, f1) a stream gives it the frames [max(0, f0 - window), min(F, f1 + lookahead)) -- pre_conv's
        causal padding starts at the window's first frame -- and keeps rows [f0, f1) of the result. window < 0: everything
        (the one-shot decode). codes [F][16] -> pcm [F * upsample]."""
        codes = np.asarray(codes, np.int64)
        F = codes.shape[0]
        tail = self._codec_tail16 if f16 else self._codec_tail   # f16: a float16 ("lite") speech tokenizer, as codec_decode(f16=True)
        if window < 0:
            return tail(self._codec_front(codes))
        lat = []
        for f0 in range(0, F, chunk):
            f1 = min(F, f0 + chunk)
            w0, w1 = max(0, f0 - window), min(F, f1 + lookahead)
            lat.append(self._codec_front(codes[w0:w1])[f0 - w0: f1 - w0])
        return tail(np.ascontiguousarray(np.concatenate(lat, 0)))

    def generate(self, req: Request, s: Sampling, row: int = 0):
        """generateCustomVoice end to end (Qwen3.swift:783-962): codes -> PCM, trimmed (:954-959)."""
        tr = self.generate_codes(req, s, row)
        if tr.codes.shape[0] == 0:
            raise RuntimeError("Generation failed: No tokens generated")
        pcm, valid = self.codec_decode(tr.codes)
        if 0 < valid < pcm.shape[0]:
            pcm = pcm[:valid]
        return pcm, tr

    # -- voice clone front end: codec encoder (V1) ---------------------------------------------
    @property
    def has_encoder(self) -> bool:  # Qwen3TTSSpeechTokenizer.hasEncoder (SpeechTokenizer.swift:816-818)
        return self.codec is not None and self.ec is not None

    @property
    def supports_voice_cloning(self) -> bool:  # Qwen3.swift:1210-1214
        return self.cfg["tts_model_type"] == "base" and self.has_encoder

    @staticmethod
    def _conv_f32(x, W, b, stride, dil, pl, pr, mode):
        x = np.ascontiguousarray(x, np.float32)
        T, Cin = x.shape
        Cout, K, ci = W.shape
        assert ci == Cin, (W.shape, x.shape)
        args = [_pf(x), _pf(W), _pf(b), C.c_int(T), C.c_int(Cin), C.c_int(Cout), C.c_int(K), C.c_int(stride),
                C.c_int(dil), C.c_int(pl), C.c_int(pr), C.c_int(mode)]
        Tout = lib().o_conv1d_f32(*args, None)
        out = np.empty((max(Tout, 0), Cout), np.float32)
        if Tout > 0:
            lib().o_conv1d_f32(*args, _pf(out))
        return out

    def _sconv(self, x, prefix, K, stride=1, dil=1):
        """StreamableConv1d, causal (SpeechTokenizerEncoder.swift:163-186) over NormConv1d/EncoderConv1d
        (:218-282). Padding is always zeros: `padMode` is stored but never read (:184)."""
        W, b = self.codec[prefix + ".conv.conv.weight"], self.codec.get(prefix + ".conv.conv.bias")
        assert W.shape[1] == K, (prefix, W.shape, K)
        T = x.shape[0]
        eff = (K - 1) * dil + 1
        ptotal = eff - stride
        nframes = np.float32(max(T + ptotal - eff, 0)) / np.float32(stride) + np.float32(1.0)  # :115, Float
        ideal = (int(np.ceil(nframes)) - 1) * stride + eff - ptotal
        extra = max(0, ideal - T)
        return self._conv_f32(x, W, b, stride, dil, ptotal, extra, 0)

    @staticmethod
    def _elu(x):  # :1075-1077, alpha 1
        x = x.astype(np.float32)
        return np.where(x > 0, x, (np.exp(np.minimum(x, 0)).astype(np.float32) - np.float32(1.0))).astype(np.float32)

    def _lin_w(self, x, W):
        M, K = x.shape
        out = np.empty((M, W.shape[0]), np.float32)
        lib().o_linear_f32(_pf(np.ascontiguousarray(x, np.float32)), _pf(np.ascontiguousarray(W, np.float32)), None,
                           C.c_int(M), C.c_int(K), C.c_int(W.shape[0]), _pf(out))
        return out

    @staticmethod
    def rope_tables_f32(T: int, dim: int, base: float):
        """MLXNN.RoPE(dimensions: dim, traditional: false, base) at offset 0 (SpeechTokenizerEncoder.swift:494,
        505-508): angle(pos, i) = pos * base^(-i/(dim/2)), halves layout. Tables are evaluated in double and
        rounded to fp32 (the engine builds the same tables on the host)."""
        half = dim // 2
        inv = np.power(float(base), -np.arange(half, dtype=np.float64) / half)
        ang = np.arange(T, dtype=np.float64)[:, None] * inv[None, :]
        return np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32)

    def encoder_codebook(self, name: str, j: int):
        """EncoderEuclideanCodebook.updateInPlace (:738-743): embedding = embed_sum / max(usage, 1e-5),
        c2 = sum(e^2)/2 (squares in fp32, the sum accumulated in double and rounded once: the centre of every
        fp32 summation order MLX could pick)."""
        p = f"encoder.quantizer.{name}.vq.layers.{j}.codebook"
        usage = np.maximum(self.codec[p + ".clusterUsage"], np.float32(1e-5))[:, None]
        emb = (self.codec[p + ".embeddingSum"] / usage).astype(np.float32)
        c2 = ((emb * emb).astype(np.float32).astype(np.float64).sum(-1).astype(np.float32) / np.float32(2)).astype(np.float32)
        return emb, c2

    def codec_encode(self, audio: np.ndarray, stages: Optional[dict] = None, all_layers: bool = False):
        """Qwen3TTSSpeechTokenizerEncoder.encode (SpeechTokenizerEncoder.swift:1031-1056) for one waveform
        [S] -> codes [16][T] int32. `stages` receives channels-last activations and, per RVQ layer, the gap
        between the two smallest distances (tests use it to recognise near-ties)."""
        ec, cw = self.ec, self.codec
        assert ec["num_residual_layers"] == 1 and not ec["use_conv_shortcut"] and ec["use_causal_conv"]
        x = np.ascontiguousarray(np.asarray(audio, np.float32).reshape(-1, 1))
        x = self._sconv(x, "encoder.encoder.init_conv1d", ec["kernel_size"])  # SeanetEncoder :436-443
        if stages is not None:
            stages["init_conv"] = x
        for i, ratio in enumerate(reversed(ec["upsampling_ratios"])):  # :417
            p = f"encoder.encoder.layers.{i}"
            y = self._sconv(self._elu(x), p + ".residuals.0.block.0", ec["residual_kernel_size"])  # :333-347
            y = self._sconv(self._elu(y), p + ".residuals.0.block.1", 1)
            x = (y + x).astype(np.float32)
            x = self._sconv(self._elu(x), p + ".downsample", 2 * ratio, stride=ratio)  # :384-390
            if stages is not None:
                stages[f"layer{i}"] = x
        x = self._sconv(self._elu(x), "encoder.encoder.final_conv1d", ec["last_kernel_size"])
        if stages is not None:
            stages["seanet"] = x
        # EncoderProjectedTransformer (:650-674): no input/output projection (dims equal), causal mask (:1039-1043)
        T = x.shape[0]
        nh, hd = ec["num_attention_heads"], ec["hidden_size"] // ec["num_attention_heads"]
        assert ec["num_key_value_heads"] == nh
        cos, sin = self.rope_tables_f32(T, hd, ec["rope_theta"])
        half = hd // 2

        def rope(a):  # [T][nh*hd]
            a = a.reshape(T, nh, hd)
            x1, x2 = a[:, :, :half], a[:, :, half:]
            c, s_ = cos[:, None, :], sin[:, None, :]
            o1 = ((x1 * c).astype(np.float32) - (x2 * s_).astype(np.float32)).astype(np.float32)
            o2 = ((x1 * s_).astype(np.float32) + (x2 * c).astype(np.float32)).astype(np.float32)
            return np.ascontiguousarray(np.concatenate([o1, o2], -1).reshape(T, nh * hd))

        def ln(a, pfx):  # LayerNorm eps 1e-5 (:559-560)
            out = np.empty_like(a)
            lib().o_layernorm_f32(_pf(np.ascontiguousarray(a)), _pf(cw[pfx + ".weight"]), _pf(cw[pfx + ".bias"]),
                                  C.c_float(1e-5), C.c_int(a.shape[0]), C.c_int(a.shape[1]), _pf(out))
            return out

        for l in range(ec["num_hidden_layers"]):  # EncoderTransformerLayer :571-590
            p = f"encoder.encoder_transformer.transformer.layers.{l}"
            n1 = ln(x, p + ".norm1")
            q = rope(self._lin_w(n1, cw[p + ".self_attn.q_proj.weight"]))
            k = rope(self._lin_w(n1, cw[p + ".self_attn.k_proj.weight"]))
            v = self._lin_w(n1, cw[p + ".self_attn.v_proj.weight"])
            ao = np.empty_like(q)
            lib().o_attention_causal_f32(_pf(q), _pf(k), _pf(v), C.c_int(T), C.c_int(nh), C.c_int(hd), _pf(ao))
            y = self._lin_w(ao, cw[p + ".self_attn.o_proj.weight"])
            x = (x + (y * cw[p + ".layer_scale_1.scale"]).astype(np.float32)).astype(np.float32)
            n2 = ln(x, p + ".norm2")
            h1 = self._lin_w(n2, cw[p + ".gating.linear1.weight"])
            # geluApprox (:1080-1082): x*0.5*(1+tanh(0.7978845608*(x+0.044715*x^3)))
            f = np.float32
            inner = (f(0.7978845608) * (h1 + (f(0.044715) * (h1 * h1 * h1).astype(f)).astype(f)).astype(f)).astype(f)
            g = ((h1 * f(0.5)).astype(f) * (f(1.0) + np.tanh(inner).astype(f)).astype(f)).astype(f)
            y = self._lin_w(g, cw[p + ".gating.linear2.weight"])
            x = (x + (y * cw[p + ".layer_scale_2.scale"]).astype(np.float32)).astype(np.float32)
        if stages is not None:
            stages["transformer"] = x
        enc_rate = np.float32(ec["sampling_rate"]) / np.float32(int(np.prod(ec["upsampling_ratios"])))  # :1008-1009
        ds = int(enc_rate / np.float32(ec["frame_rate"]))
        x = self._sconv(x, "encoder.downsample.conv", 2 * ds, stride=ds)  # EncoderConvDownsample1d, no bias
        if stages is not None:
            stages["downsample"] = x
        Tq = x.shape[0]
        codes = []
        gaps = []
        for name, nq in (("rvq_first", 1), ("rvq_rest", ec["num_quantizers"] - 1)):  # :934-941
            W = cw[f"encoder.quantizer.{name}.input_proj.weight"]  # [dim][1][in]
            r = self._lin_w(x, W[:, 0, :])  # EncoderConv1dProj :897-902
            if stages is not None:
                stages[f"{name}_in"] = r
            n_used = nq if all_layers else min(nq, 16 - len(codes))  # later layers never reach the output (:1055)
            for j in range(n_used):  # EncoderResidualVectorQuantization.encode :816-829
                emb, c2 = self.encoder_codebook(name, j)
                dist = (c2[None, :] - self._lin_w(r, emb)).astype(np.float32)  # :755-756
                idx = np.argmin(dist, axis=-1)
                part = np.partition(dist, 1, axis=-1)
                gaps.append((part[:, 1] - part[:, 0]).astype(np.float32))
                r = (r - emb[idx]).astype(np.float32)  # :823
                codes.append(idx.astype(np.int32))
        if stages is not None:
            stages["gaps"] = np.stack(gaps[:16])
        return np.stack(codes[:16]).reshape(16, Tq)  # :1055

    # -- voice clone front end: speaker encoder (V2) ---------------------------------------------
    @staticmethod
    def mel_spectrogram(audio: np.ndarray, n_fft=1024, num_mels=128, sample_rate=24000, hop=256, win=1024,
                        fmin=0.0, fmax=12000.0) -> np.ndarray:
        """melSpectrogram (SpeakerEncoder.swift:410-456) -> [T][mels] float32. The DFT is evaluated in double and
        rounded to complex64 (the reference runs a single-precision FFT; both are the exact DFT to fp32 rounding)."""
        f = np.float32
        x = np.asarray(audio, f).reshape(-1)
        n = np.arange(win, dtype=f)
        window = (f(0.5) * (f(1.0) - np.cos((f(2.0) * f(np.pi)) * n / f(win - 1)).astype(f))).astype(f)  # :459-462
        padded = np.concatenate([np.zeros(n_fft // 2, f), x, np.zeros(n_fft // 2, f)])  # :430-431
        nfr = (padded.shape[0] - n_fft) // hop + 1  # :469
        frames = np.stack([padded[i * hop: i * hop + n_fft] * window for i in range(nfr)]).astype(f)  # :473-477
        spec = np.fft.rfft(frames.astype(np.float64), axis=-1).astype(np.complex64)  # :484-486, 513 bins
        mag = np.abs(spec).astype(f)
        power = (mag * mag).astype(f)  # :437
        nfreq = n_fft // 2 + 1
        fb = np.empty((nfreq, num_mels), f)
        lib().o_mel_filterbank(C.c_int(n_fft), C.c_int(num_mels), C.c_int(sample_rate), C.c_float(fmin), C.c_float(fmax),
                               _pf(fb))
        mel = np.empty((nfr, num_mels), f)
        lib().o_linear_f32(_pf(np.ascontiguousarray(power)), _pf(np.ascontiguousarray(fb.T)), None, C.c_int(nfr),
                           C.c_int(nfreq), C.c_int(num_mels), _pf(mel))  # :449
        return np.log(np.maximum(mel, f(1e-10))).astype(f)  # :452

    def _tdnn(self, x, prefix, K, dil):
        """TimeDelayNetBlock (SpeakerEncoder.swift:45-70): reflect pad, conv, ReLU. x [T][C]."""
        W, b = self.w_f32(prefix + ".conv.weight"), self.w_f32(prefix + ".conv.bias")
        pad = (K - 1) * dil // 2
        assert W.shape[1] == K and pad < x.shape[0]
        return np.maximum(self._conv_f32(x, W, b, 1, dil, pad, pad, 1), np.float32(0))

    def w_f32(self, name: str) -> np.ndarray:
        """Speaker-encoder parameters are stored in the checkpoint dtype (bf16) and meet fp32 activations:
        MLX promotes to fp32, i.e. an exact upcast of the stored values."""
        a = self.w[name]
        return np.ascontiguousarray(bf16_to_f32(a) if a.dtype == np.uint16 else a.astype(np.float32))

    def speaker_embedding(self, audio: np.ndarray, stages: Optional[dict] = None) -> np.ndarray:
        """extractSpeakerEmbedding (Qwen3.swift:222-249) -> Qwen3TTSSpeakerEncoder (SpeakerEncoder.swift:364-394).
        Returns fp32 [enc_dim]."""
        sc = self.sc
        f = np.float32
        mel = self.mel_spectrogram(audio, 1024, 128, 24000, 256, 1024, 0.0, 12000.0)
        if stages is not None:
            stages["mel"] = mel
        ch, ks, dl = sc["enc_channels"], sc["enc_kernel_sizes"], sc["enc_dilations"]
        scale = sc["enc_res2net_scale"]
        h = self._tdnn(mel, "speaker_encoder.blocks.0", ks[0], dl[0])
        if stages is not None:
            stages["h0"] = h
        hs = []
        for bi in (1, 2, 3):  # SqueezeExcitationRes2NetBlock :204-211
            p = f"speaker_encoder.blocks.{bi}"
            res = h
            o = self._tdnn(h, p + ".tdnn1", 1, 1)
            cs = o.shape[1] // scale
            parts, prev = [], None
            for i in range(scale):  # Res2NetBlock :96-116
                chunk = o[:, i * cs:(i + 1) * cs]
                if i == 0:
                    prev = chunk
                elif i == 1:
                    prev = self._tdnn(chunk, f"{p}.res2net_block.blocks.{i - 1}", ks[bi], dl[bi])
                else:
                    prev = self._tdnn((chunk + prev).astype(f), f"{p}.res2net_block.blocks.{i - 1}", ks[bi], dl[bi])
                parts.append(prev)
            o = np.ascontiguousarray(np.concatenate(parts, 1))
            o = self._tdnn(o, p + ".tdnn2", 1, 1)
            # SqueezeExcitationBlock :143-155
            m = (o.astype(np.float64).mean(0)).astype(f)[None, :]
            s1 = np.maximum(self._lin_w(m, self.w_f32(p + ".se_block.conv1.weight")[:, 0, :]) + self.w_f32(p + ".se_block.conv1.bias"), f(0)).astype(f)
            z = (self._lin_w(s1, self.w_f32(p + ".se_block.conv2.weight")[:, 0, :]) + self.w_f32(p + ".se_block.conv2.bias")).astype(f)
            se = (f(1.0) / (f(1.0) + np.exp(-z).astype(f))).astype(f)
            h = ((o * se).astype(f) + res).astype(f)
            hs.append(h)
            if stages is not None:
                stages[f"h{bi}"] = h
        o = self._tdnn(np.ascontiguousarray(np.concatenate(hs, 1)), "speaker_encoder.mfa", ks[4], dl[4])  # :379-380
        if stages is not None:
            stages["mfa"] = o
        # AttentiveStatisticsPooling :238-272
        T = o.shape[0]
        o64 = o.astype(np.float64)
        mean = o64.mean(0).astype(f)
        var = ((o64 - o64.mean(0)) ** 2).mean(0).astype(f)
        std = np.sqrt(var + f(1e-12)).astype(f)
        att_in = np.ascontiguousarray(np.concatenate([o, np.broadcast_to(mean, o.shape), np.broadcast_to(std, o.shape)], 1), f)
        a = np.tanh(self._tdnn(att_in, "speaker_encoder.asp.tdnn", 1, 1)).astype(f)
        a = (self._lin_w(a, self.w_f32("speaker_encoder.asp.conv.weight")[:, 0, :]) + self.w_f32("speaker_encoder.asp.conv.bias")).astype(f)
        a = a - a.max(0, keepdims=True)
        e = np.exp(a).astype(f)
        att = (e / e.astype(np.float64).sum(0).astype(f)).astype(f)  # softmax over time
        wmean = (att.astype(np.float64) * o64).sum(0).astype(f)
        wvar = (att.astype(np.float64) * (o64 - wmean.astype(np.float64)) ** 2).sum(0).astype(f)
        wstd = np.sqrt(np.maximum(wvar, f(1e-12))).astype(f)
        pooled = np.concatenate([wmean, wstd])[None, :].astype(f)
        if stages is not None:
            stages["pooled"] = pooled[0]
        out = (self._lin_w(pooled, self.w_f32("speaker_encoder.fc.weight")[:, 0, :]) + self.w_f32("speaker_encoder.fc.bias")).astype(f)
        _ = T
        return out[0]

    # -- voice clone: prompt + generation (V3) ---------------------------------------------------
    def prepare_icl_generation_inputs(self, req: Request):
        """prepareICLGenerationInputs (Qwen3.swift:418-582). Returns (input_embeds, trailing, tts_pad, ref_codes
        [16][T]). The speaker x-vector (fp32) is rounded to bf16 when it enters the prompt: the talker's storage
        dtype, as for every other prompt row (MLX would instead promote the concatenated prefix to fp32)."""
        t, cfg = self.t, self.cfg
        if not self.has_encoder:
            raise RuntimeError("Model not initialized: Speech tokenizer encoder not available")
        audio = np.asarray(req.ref_audio, np.float32).reshape(-1)
        ref_codes = self.codec_encode(audio) if req.ref_codes_override is None else np.asarray(req.ref_codes_override)  # :443
        ref_ids, target_ids = list(req.ref_text_ids), list(req.text_ids)
        ref_text_ids = ref_ids[3: len(ref_ids) - 2]  # :451
        text_ids = target_ids[3: len(target_ids) - 5]  # :457
        tts = self.text_projection(self.embed_text([cfg["tts_bos_token_id"], cfg["tts_eos_token_id"],
                                                    cfg["tts_pad_token_id"]]))
        tts_bos, tts_eos, tts_pad = tts[0:1], tts[1:2], tts[2:3]
        text_embed = self.text_projection(self.embed_text(ref_text_ids + text_ids))  # :473-475
        text_embed = np.concatenate([text_embed, tts_eos], 0)
        ce = self.codec_embed(ref_codes[0])  # :485-491
        for i in range(t["num_code_groups"] - 1):
            ce = self.add(ce, self.cp_embed(i, ref_codes[i + 1]))
        codec_icl = np.concatenate([self.codec_embed([t["codec_bos_id"]]), ce], 0)  # :494-496
        text_with_pad = self.add(text_embed, self.codec_embed([t["codec_pad_id"]]))  # :505-506
        codec_with_pad = self.add(codec_icl, tts_pad)  # :509-510
        icl = np.concatenate([text_with_pad, codec_with_pad], 0)
        lang = req.language.lower()
        lid = t["codec_language_id"].get(lang) if lang != "auto" else None  # :515-519 (no dialect override here)
        speaker_embed = None
        if self.sc is not None:  # :522-525
            speaker_embed = f32_to_bf16(self.speaker_embedding(audio))[None, :]
        if lid is None:
            prefill = [t["codec_nothink_id"], t["codec_think_bos_id"], t["codec_think_eos_id"]]
        else:
            prefill = [t["codec_think_id"], t["codec_think_bos_id"], lid, t["codec_think_eos_id"]]
        prefix = self.codec_embed(prefill)
        suffix = self.codec_embed([t["codec_pad_id"], t["codec_bos_id"]])
        prefix = np.concatenate([prefix] + ([speaker_embed] if speaker_embed is not None else []) + [suffix], 0)
        role = self.text_projection(self.embed_text(target_ids[0:3]))  # :564-566
        n = prefix.shape[0]
        combined = np.concatenate([np.repeat(tts_pad, n - 2, 0), tts_bos], 0)
        combined = self.add(combined, prefix[: n - 1])  # :569-573
        input_embeds = np.concatenate([role, combined, icl], 0)  # :576
        return (np.ascontiguousarray(input_embeds), np.ascontiguousarray(tts_pad), np.ascontiguousarray(tts_pad),
                ref_codes)

    def generate_voice_clone(self, req: Request, s: Sampling, row: int = 0):
        """generateVoiceClone (Qwen3.swift:1009-1203): ICL prompt, the same AR loop (EOS is tested before the
        history append, :1098-1102, which cannot change any output), decode [ref ++ generated] and cut the
        reference part proportionally (:1195-1199). Returns (pcm, GenTrace, ref_codes)."""
        tr = self.generate_codes(req, s, row)
        if tr.codes.shape[0] == 0:
            raise RuntimeError("Generation failed: No tokens generated")
        ref_codes = tr.ref_codes
        full = np.concatenate([ref_codes.T.astype(np.int32), tr.codes], 0)  # :1178-1180
        pcm, valid = self.codec_decode(full)
        if 0 < valid < pcm.shape[0]:
            pcm = pcm[:valid]
        ref_len, total = ref_codes.shape[1], full.shape[0]
        cut = int(np.float32(ref_len) / np.float32(max(total, 1)) * np.float32(pcm.shape[0]))  # :1196
        if 0 < cut < pcm.shape[0]:
            pcm = pcm[cut:]
        return pcm, tr, ref_codes
