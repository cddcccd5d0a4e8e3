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
        _lib = C.CDLL(build())
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


def sanitize_speech_tokenizer(weights: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """Decoder half of sanitizeSpeechTokenizerWeights (Qwen3.swift:1498-1750). Encoder keys are
    passed through untouched (voice-clone path, not restated yet)."""
    out: Dict[str, np.ndarray] = {}
    cb: Dict[str, Dict[str, np.ndarray]] = {}
    for key, value in weights.items():
        if "._codebook.cluster_usage" in key or "._codebook.embedding_sum" in key:  # :1532-1543
            base = key.split("._codebook.")[0]
            cb.setdefault(base, {})["cluster_usage" if "cluster_usage" in key else "embedding_sum"] = value
            continue
        if key.startswith("encoder."):
            out[key] = value
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
    max_tokens: int = 2048


@dataclass
class GenTrace:
    codes: np.ndarray                       # [F][16] int32
    talker_logits: List[np.ndarray] = field(default_factory=list)  # per frame, bf16 bits [V]
    cp_logits: List[np.ndarray] = field(default_factory=list)      # per frame [15][Vcp]
    hit_eos: bool = False


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
        w = {k: v for k, v in w.items() if "position_ids" not in k}  # sanitize, Qwen3.swift:1223-1226
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
        if os.path.isdir(st_dir):  # postLoadHook, Qwen3.swift:1462-1494
            with open(os.path.join(st_dir, "config.json")) as f:
                sraw = json.load(f)
            self.dc = _with_defaults(sraw.get("decoder_config"), _DEC_DEF)
            self.codec = sanitize_speech_tokenizer(load_safetensors_dir(st_dir))
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
    def _conv(self, x, prefix, K, dil=1, groups=1):
        W, b = self.codec[prefix + ".weight"], self.codec.get(prefix + ".bias")
        T, Cin = x.shape
        Cout = W.shape[0]
        assert W.shape[1] == K and W.shape[2] == Cin // groups, (prefix, W.shape, K, Cin)
        out = np.empty((T, Cout), np.float32)
        lib().o_conv1d_causal(_pf(np.ascontiguousarray(x)), _pf(W), _pf(b), C.c_int(T), C.c_int(Cin),
                              C.c_int(Cout), C.c_int(K), C.c_int(dil), C.c_int(groups), _pf(out))
        return out

    def _convtr(self, x, prefix, K, stride):
        W, b = self.codec[prefix + ".weight"], self.codec.get(prefix + ".bias")
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

    def codec_decode(self, codes: np.ndarray, stages: Optional[dict] = None):
        """Qwen3TTSSpeechTokenizer.decode for one utterance (SpeechTokenizer.swift:823-836 ->
        754-784). codes [F][16] int. Returns (pcm float32 [1920*F], valid_len). Intermediate
        activations ([T][C] channels-last) are stored into `stages` when given."""
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
            p = f"{pt}.layers.{l}"
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
            x = (x + y * cw[p + ".mlp_layer_scale.scale"]).astype(np.float32)
        x = self._rms(x, pt + ".norm")
        h = self._lin(x, pt + ".output_proj", True)
        if stages is not None:
            stages["pre_transformer"] = h
        for i, r in enumerate(dc["upsampling_ratios"]):  # :767-775
            h = self._convtr(h, f"decoder.upsample.{i}.0.conv", r, r)
            p = f"decoder.upsample.{i}.1"
            res = h
            d = self._conv(h, p + ".dwconv.conv", 7, groups=h.shape[1])
            n = np.empty_like(d)
            lib().o_layernorm_f32(_pf(d), _pf(cw[p + ".norm.weight"]), _pf(cw[p + ".norm.bias"]),
                                  C.c_float(1e-6), C.c_int(d.shape[0]), C.c_int(d.shape[1]), _pf(n))
            a = self._lin(n, p + ".pwconv1", True)
            ga = np.empty_like(a)
            lib().o_gelu_f32(_pf(a), C.c_int64(a.size), _pf(ga))
            b2 = self._lin(ga, p + ".pwconv2", True)
            h = (res + cw[p + ".gamma"] * b2).astype(np.float32)
            if stages is not None:
                stages[f"upsample{i}"] = h
        h = self._conv(h, "decoder.decoder.initConv.conv", 7)  # MainDecoder :681-690
        if stages is not None:
            stages["init_conv"] = h
        for b, rate in enumerate(dc["upsample_rates"]):
            p = f"decoder.decoder.block{b}"
            h = self._snake(h, p + ".snake")
            h = self._convtr(h, p + ".upsample.conv", 2 * rate, rate)
            for j, dil in ((1, 1), (2, 3), (3, 9)):
                rp = f"{p}.res{j}"
                r0 = h
                y = self._snake(h, rp + ".act1")
                y = self._conv(y, rp + ".conv1.conv", 7, dil=dil)
                y = self._snake(y, rp + ".act2")
                y = self._conv(y, rp + ".conv2.conv", 1)
                h = (r0 + y).astype(np.float32)
            if stages is not None:
                stages[f"block{b}"] = h
        h = self._snake(h, "decoder.decoder.outSnake")
        h = self._conv(h, "decoder.decoder.outConv.conv", 7)
        pcm = np.clip(h[:, 0], -1.0, 1.0).astype(np.float32)  # :781
        up = int(np.prod(dc["upsample_rates"]) * np.prod(dc["upsampling_ratios"]))
        valid = int((codes[:, 0] > 0).sum()) * up  # :831-833
        return pcm, valid

    def generate(self, req: Request, s: Sampling, row: int = 0):
        """generateCustomVoice end to end (Qwen3.swift:783-962): codes -> PCM, trimmed (:954-959)."""
        tr = self.generate_codes(req, s, row)
        if tr.codes.shape[0] == 0:
            raise RuntimeError("Generation failed: No tokens generated")
        pcm, valid = self.codec_decode(tr.codes)
        if 0 < valid < pcm.shape[0]:
            pcm = pcm[:valid]
        return pcm, tr
