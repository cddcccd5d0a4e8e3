"""Synthetic Qwen3-TTS checkpoint writer.

There are no real checkpoints in this environment, so tests and bench.py use
random-init checkpoints written in the *reference's* on-disk layout: `config.json` +
`*.safetensors` in the model dir and `speech_tokenizer/{config.json,*.safetensors}` with the
upstream (PyTorch-style) key names and tensor layouts that the reference loader sanitises
(/root/reference/Sources/Qwen3TTS/Models/Qwen3.swift:1382-1495, 1498-1750). The engine's C++
loader and the oracle's Python loader both consume these files, so the key remaps, conv-weight
transposes and the codebook = embedding_sum / clip(cluster_usage) step are exercised for real.

Weight statistics follow BASELINE.md section 3: N(0, 0.02^2) matrices, norm weights 1,
LayerScale 0.01, Snake alpha/beta ~ N(0, 0.1^2), codebooks N(0, 1).
"""
from __future__ import annotations

import json
import os
import struct
from typing import Dict, Tuple

import numpy as np

# --------------------------------------------------------------------------------------------
# minimal safetensors writer (bf16 stored as raw uint16; numpy has no bf16 dtype)
# --------------------------------------------------------------------------------------------

_DT = {"F32": np.float32, "BF16": np.uint16, "I32": np.int32, "U32": np.uint32, "F16": np.float16}


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even fp32 -> bf16 bit patterns (uint16)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return u.astype(np.uint16)


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << 16).view(np.float32)


def save_safetensors(path: str, tensors: Dict[str, Tuple[str, np.ndarray]]) -> None:
    """tensors: name -> (dtype tag, array). BF16 arrays are uint16 bit patterns."""
    header = {}
    off = 0
    order = sorted(tensors)
    for k in order:
        tag, arr = tensors[k]
        n = arr.size * arr.dtype.itemsize
        header[k] = {"dtype": tag, "shape": list(arr.shape), "data_offsets": [off, off + n]}
        off += n
    hj = json.dumps(header, separators=(",", ":")).encode()
    hj += b" " * ((8 - len(hj) % 8) % 8)
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(hj)))
        f.write(hj)
        for k in order:
            f.write(memoryview(np.ascontiguousarray(tensors[k][1])).cast("B"))


# --------------------------------------------------------------------------------------------
# config presets
# --------------------------------------------------------------------------------------------

def _codec_cfg(tiny: bool) -> dict:
    if tiny:
        # channel counts stay > 64 so the reference's conv-layout shape heuristic
        # (Qwen3.swift:1246-1260) classifies every tensor the way it does for real checkpoints
        return dict(latent_dim=128, codebook_dim=128, codebook_size=256, decoder_dim=1280,
                    hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                    num_attention_heads=2, num_key_value_heads=2, head_dim=64, rms_norm_eps=1e-5,
                    num_quantizers=16, num_semantic_quantizers=1, semantic_codebook_size=2048,
                    upsample_rates=[8, 5, 4, 3], upsampling_ratios=[2, 2],
                    layer_scale_initial_scale=0.01)
    return dict(latent_dim=1024, codebook_dim=512, codebook_size=2048, decoder_dim=1536,
                hidden_size=512, intermediate_size=1024, num_hidden_layers=8,
                num_attention_heads=16, num_key_value_heads=16, head_dim=64, rms_norm_eps=1e-5,
                num_quantizers=16, num_semantic_quantizers=1, semantic_codebook_size=4096,
                upsample_rates=[8, 5, 4, 3], upsampling_ratios=[2, 2],
                layer_scale_initial_scale=0.01)


def _encoder_cfg(tiny: bool) -> dict:
    """speech_tokenizer encoder_config (Config.swift:476-504). The tiny variant keeps the 1920-sample hop
    (ratios 8*6*5*4, stride-2 downsample) and head_dim 64."""
    if tiny:
        return dict(frame_rate=12.5, audio_channels=1, codebook_dim=64, codebook_size=128, compress=2,
                    dilation_growth_rate=2, head_dim=64, hidden_size=128, intermediate_size=256, kernel_size=7,
                    last_kernel_size=3, layer_scale_initial_scale=0.01, max_position_embeddings=8000,
                    num_attention_heads=2, num_filters=8, num_hidden_layers=2, num_key_value_heads=2,
                    num_quantizers=32, num_residual_layers=1, residual_kernel_size=3, rope_theta=10000.0,
                    sampling_rate=24000, sliding_window=250, upsampling_ratios=[8, 6, 5, 4], use_causal_conv=True,
                    use_conv_shortcut=False)
    return dict(frame_rate=12.5, audio_channels=1, codebook_dim=256, codebook_size=2048, compress=2,
                dilation_growth_rate=2, head_dim=64, hidden_size=512, intermediate_size=2048, kernel_size=7,
                last_kernel_size=3, layer_scale_initial_scale=0.01, max_position_embeddings=8000,
                num_attention_heads=8, num_filters=64, num_hidden_layers=8, num_key_value_heads=8,
                num_quantizers=32, num_residual_layers=1, residual_kernel_size=3, rope_theta=10000.0,
                sampling_rate=24000, sliding_window=250, upsampling_ratios=[8, 6, 5, 4], use_causal_conv=True,
                use_conv_shortcut=False)


def _speaker_cfg(enc_dim: int, tiny: bool) -> dict:
    """speaker_encoder_config (Config.swift:80-91). mel_dim stays 128: extractSpeakerEmbedding hard-codes it
    (Qwen3.swift:232-241). Tiny: 1x1 conv widths stay > 64 for the layout heuristic (Qwen3.swift:1246-1260)."""
    if tiny:
        return dict(mel_dim=128, enc_dim=enc_dim, enc_channels=[128, 128, 128, 128, 384], enc_kernel_sizes=[5, 3, 3, 3, 1],
                    enc_dilations=[1, 2, 3, 4, 1], enc_attention_channels=96, enc_res2net_scale=8, enc_se_channels=96,
                    sample_rate=24000)
    return dict(mel_dim=128, enc_dim=enc_dim, enc_channels=[512, 512, 512, 512, 1536], enc_kernel_sizes=[5, 3, 3, 3, 1],
                enc_dilations=[1, 2, 3, 4, 1], enc_attention_channels=128, enc_res2net_scale=8, enc_se_channels=128,
                sample_rate=24000)


def preset(name: str) -> dict:
    """Returns {"config": <config.json dict>, "speech_tokenizer": <speech_tokenizer/config.json>}.

    tiny-a : talker hidden == code-predictor hidden (no small_to_mtp_projection, like 0.6B)
    tiny-b : talker hidden != code-predictor hidden (projection present, like 1.7B)
    0.6b / 1.7b : full dimensions (SURVEY.md section 8 header)
    """
    spk = {"aiden": 3010, "vivian": 3011, "eric": 3012}
    if name in ("tiny-a", "tiny-b"):
        H = 256 if name == "tiny-a" else 384
        cp = dict(vocab_size=256, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                  num_attention_heads=4, num_key_value_heads=2, head_dim=128, num_code_groups=16,
                  rms_norm_eps=1e-6, rope_theta=1e6)
        talker = dict(vocab_size=3072, text_vocab_size=1024, hidden_size=H, text_hidden_size=256,
                      intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, head_dim=128, num_code_groups=16, rms_norm_eps=1e-6,
                      rope_theta=1e6, code_predictor_config=cp, spk_id=spk,
                      spk_is_dialect={"aiden": False, "vivian": False, "eric": "sichuan_dialect"},
                      codec_language_id={"chinese": 2055, "english": 2050, "sichuan_dialect": 2062})
        cfg = dict(model_type="qwen3_tts", tts_model_size="tiny", tts_model_type="custom_voice",
                   talker_config=talker, im_start_token_id=1000, im_end_token_id=1001,
                   tts_pad_token_id=1010, tts_bos_token_id=1011, tts_eos_token_id=1012,
                   sample_rate=24000)
        return {"config": cfg, "speech_tokenizer": {"decoder_config": _codec_cfg(True)}}
    if name in ("tiny-base", "tiny-base-fullenc", "0.6b-base", "1.7b-base"):
        # Base checkpoints (voice clone, BASELINE config 4): tts_model_type "base", a speaker encoder in the main
        # file and an encoder half in speech_tokenizer/ (Qwen3.swift:55-57, 1210-1214; SpeechTokenizer.swift:808-812).
        # tiny-base-fullenc: tiny talker behind the full-size codec encoder / speaker encoder (real conv shapes).
        tiny = name == "tiny-base"
        p = preset("tiny-b" if name.startswith("tiny") else name[:-5])
        if name == "tiny-base-fullenc":  # encoder codes must index the talker-side embedding tables
            p["config"]["talker_config"]["code_predictor_config"]["vocab_size"] = 2048
            p["speech_tokenizer"]["decoder_config"]["codebook_size"] = 2048
        p["config"]["tts_model_type"] = "base"
        p["config"]["speaker_encoder_config"] = _speaker_cfg(p["config"]["talker_config"]["hidden_size"], tiny)
        p["config"]["talker_config"].setdefault("spk_id", spk)
        p["speech_tokenizer"]["encoder_config"] = _encoder_cfg(tiny)
        return p
    if name == "tiny-q":
        # like BASELINE config 5 in miniature: MLX affine int4 (group 64) Linears + pruned text vocabulary with a
        # token map (docs/paper.tex:160-178, 232-256); embeddings stay bf16 (no `.scales` keys)
        p = preset("tiny-b")
        p["config"]["quantization"] = {"group_size": 64, "bits": 4}
        p["config"]["talker_config"]["pruned_text_rows"] = 600  # writer-only hint, ignored by the loaders
        return p
    if name == "tiny-qe":
        # tiny-q with QuantizedEmbedding tables as well: the reference quantises an embedding exactly when its `.scales`
        # key exists in the checkpoint (Qwen3.swift:1402-1406, 1419-1422)
        p = preset("tiny-q")
        p["config"]["talker_config"]["quantized_embeddings"] = True  # writer-only hint, ignored by the loaders
        return p
    if name == "0.6b-q4":  # BASELINE config 5: 0.6B, int4-g64 Linears, pruned text vocabulary (47,427 rows)
        p = preset("0.6b")
        p["config"]["quantization"] = {"group_size": 64, "bits": 4}
        p["config"]["talker_config"]["pruned_text_rows"] = 47427
        p["codec_f16"] = True  # writer-only: the "lite" models store the speech tokenizer in float16 (docs/paper.tex:207)
        return p
    if name == "tiny-h":  # tiny-b with a float16 speech tokenizer (the lite checkpoints' storage, docs/paper.tex:207)
        p = preset("tiny-b")
        p["codec_f16"] = True
        return p
    if name in ("0.6b", "1.7b"):
        H, I = (1024, 3072) if name == "0.6b" else (2048, 6144)
        cp = dict(vocab_size=2048, hidden_size=1024, intermediate_size=3072, num_hidden_layers=5,
                  num_attention_heads=16, num_key_value_heads=8, head_dim=128, num_code_groups=16,
                  rms_norm_eps=1e-6, rope_theta=1e6)
        talker = dict(vocab_size=3072, text_vocab_size=151936, hidden_size=H, text_hidden_size=2048,
                      intermediate_size=I, num_hidden_layers=28, num_attention_heads=16,
                      num_key_value_heads=8, head_dim=128, num_code_groups=16, rms_norm_eps=1e-6,
                      rope_theta=1e6, code_predictor_config=cp, spk_id=spk)
        cfg = dict(model_type="qwen3_tts", tts_model_size="0b6" if name == "0.6b" else "1b7",
                   tts_model_type="custom_voice" if name == "0.6b" else "voice_design",
                   talker_config=talker, sample_rate=24000)
        if name == "1.7b":
            talker.pop("spk_id")
        return {"config": cfg, "speech_tokenizer": {"decoder_config": _codec_cfg(False)}}
    raise ValueError(f"unknown preset {name}")


# --------------------------------------------------------------------------------------------
# tensor generation
# --------------------------------------------------------------------------------------------

class _Gen:
    """Deterministic normal generator. numpy PCG64 for small tensors; torch CPU for big ones
    (two orders of magnitude faster for the 1.7B checkpoint)."""

    def __init__(self, seed: int, big: bool):
        self.rng = np.random.Generator(np.random.PCG64(seed))
        self.big = big
        self.tgen = None
        self.dev = "cpu"
        if big:
            import torch
            self.torch = torch
            # the big presets are only used by bench.py; both the engine and the oracle read the
            # files written here, so the generator does not need to be the same on every machine
            self.dev = "cuda" if torch.cuda.is_available() else "cpu"
            self.tgen = torch.Generator(device=self.dev).manual_seed(seed)

    def normal(self, shape, std: float, mean: float = 0.0) -> np.ndarray:
        n = int(np.prod(shape))
        if self.big and n >= (1 << 16):
            t = self.torch.empty(tuple(shape), dtype=self.torch.float32, device=self.dev)
            t.normal_(mean, std, generator=self.tgen)
            return t.cpu().numpy()
        return (self.rng.standard_normal(shape, dtype=np.float32) * np.float32(std)
                + np.float32(mean)).astype(np.float32)

    def normal_bf16(self, shape, std: float, mean: float = 0.0) -> Tuple[str, np.ndarray]:
        """N(mean, std^2) rounded to bf16, returned as a safetensors ("BF16", uint16 bits) pair."""
        n = int(np.prod(shape))
        if self.big and n >= (1 << 16):
            t = self.torch.empty(tuple(shape), dtype=self.torch.float32, device=self.dev)
            t.normal_(mean, std, generator=self.tgen)
            return ("BF16", t.to(self.torch.bfloat16).view(self.torch.int16).cpu().numpy().view(np.uint16))
        return ("BF16", f32_to_bf16_bits(self.normal(shape, std, mean)))

    def uniform(self, shape, lo: float, hi: float) -> np.ndarray:
        return self.rng.uniform(lo, hi, size=shape).astype(np.float32)


def _bf16(x: np.ndarray) -> Tuple[str, np.ndarray]:
    return ("BF16", f32_to_bf16_bits(x))


def _f32(x: np.ndarray) -> Tuple[str, np.ndarray]:
    return ("F32", np.ascontiguousarray(x, dtype=np.float32))


def talker_tensors(cfg: dict, g: _Gen, std: float = 0.02) -> Dict[str, Tuple[str, np.ndarray]]:
    """Main-model tensors, keys as the reference module tree (Talker.swift:165-171,403-405,439-440,
    495-496,589-591; CodePredictor.swift:72-78,144-146,165-169,206,283,288)."""
    t = cfg["talker_config"]
    cp = t["code_predictor_config"]
    H, TH, V, TV = t["hidden_size"], t["text_hidden_size"], t["vocab_size"], t["text_vocab_size"]
    nh, nkv, hd = t["num_attention_heads"], t["num_key_value_heads"], t["head_dim"]
    out: Dict[str, Tuple[str, np.ndarray]] = {}

    quant = cfg.get("quantization")

    def lin(name, n, k, bias=False):
        if quant:  # MLX affine 4-bit: packed uint32 [n][k/8], bf16 scales/biases [n][k/64], w = q*scale + bias
            if g.big and n * k >= (1 << 16):  # every 32-bit pattern is a valid row of eight 4-bit fields
                t = g.torch.randint(-(1 << 31), (1 << 31) - 1, (n, k // 8), dtype=g.torch.int64, device=g.dev, generator=g.tgen)
                packed = t.to(g.torch.int32).cpu().numpy().view(np.uint32)
            else:
                q = g.rng.integers(0, 16, size=(n, k), dtype=np.uint32)
                packed = np.zeros((n, k // 8), np.uint32)
                for j in range(8):
                    packed |= q[:, j::8] << np.uint32(4 * j)
            sc = g.uniform((n, k // 64), 0.002, 0.006)
            out[name + ".weight"] = ("U32", packed)
            out[name + ".scales"] = ("BF16", f32_to_bf16_bits(sc))
            out[name + ".biases"] = ("BF16", f32_to_bf16_bits(-7.5 * sc + g.normal((n, k // 64), 0.0005)))
        else:
            out[name + ".weight"] = g.normal_bf16((n, k), std)
        if bias:
            out[name + ".bias"] = g.normal_bf16((n,), std)

    def emb(name, rows, dim):
        if quant and t.get("quantized_embeddings"):  # QuantizedEmbedding: same packing as a quantised Linear, row = q*scale + bias
            q = g.rng.integers(0, 16, size=(rows, dim), dtype=np.uint32)
            packed = np.zeros((rows, dim // 8), np.uint32)
            for j in range(8):
                packed |= q[:, j::8] << np.uint32(4 * j)
            sc = g.uniform((rows, dim // 64), 0.002, 0.006)
            out[name + ".weight"] = ("U32", packed)
            out[name + ".scales"] = ("BF16", f32_to_bf16_bits(sc))
            out[name + ".biases"] = ("BF16", f32_to_bf16_bits(-7.5 * sc + g.normal((rows, dim // 64), 0.0005)))
        else:
            out[name + ".weight"] = g.normal_bf16((rows, dim), std)

    def norm(name, d):
        # norm weights 1 with a small perturbation so a forgotten weight multiply is caught
        out[name + ".weight"] = g.normal_bf16((d,), 0.05, 1.0)

    def stack(prefix, hidden, inter_list, n_heads, n_kv, head_dim):
        for l, inter in enumerate(inter_list):
            p = f"{prefix}.layers.{l}"
            lin(p + ".self_attn.q_proj", n_heads * head_dim, hidden)
            lin(p + ".self_attn.k_proj", n_kv * head_dim, hidden)
            lin(p + ".self_attn.v_proj", n_kv * head_dim, hidden)
            lin(p + ".self_attn.o_proj", hidden, n_heads * head_dim)
            norm(p + ".self_attn.q_norm", head_dim)
            norm(p + ".self_attn.k_norm", head_dim)
            lin(p + ".mlp.gate_proj", inter, hidden)
            lin(p + ".mlp.up_proj", inter, hidden)
            lin(p + ".mlp.down_proj", hidden, inter)
            norm(p + ".input_layernorm", hidden)
            norm(p + ".post_attention_layernorm", hidden)
        norm(prefix + ".norm", hidden)

    inter = t.get("per_layer_intermediate_sizes") or [t["intermediate_size"]] * t["num_hidden_layers"]
    emb("talker.model.codec_embedding", V, H)
    pruned = t.get("pruned_text_rows")
    if pruned:  # compact table + original-id -> compact-index map (Qwen3.swift:1434-1444, Talker.swift:627-633)
        emb("talker.model.text_embedding", pruned, TH)
        out["talker.model.text_token_map"] = ("I32", g.rng.integers(0, pruned, size=(TV,)).astype(np.int32))
    else:
        emb("talker.model.text_embedding", TV, TH)
    stack("talker.model", H, inter, nh, nkv, hd)
    lin("talker.text_projection.linear_fc1", TH, TH, bias=True)
    lin("talker.text_projection.linear_fc2", H, TH, bias=True)
    lin("talker.codec_head", V, H)
    ch = cp["hidden_size"]
    if ch != H:
        lin("talker.code_predictor.small_to_mtp_projection", ch, H, bias=True)
    for i in range(cp["num_code_groups"] - 1):
        emb(f"talker.code_predictor.model.codec_embedding.{i}", cp["vocab_size"], H)
        lin(f"talker.code_predictor.lm_head.{i}", cp["vocab_size"], ch)
    stack("talker.code_predictor.model", ch, [cp["intermediate_size"]] * cp["num_hidden_layers"],
          cp["num_attention_heads"], cp["num_key_value_heads"], cp["head_dim"])
    return out


FULL_WIDTH_OUT_WSTD = 0.0012


def codec_tensors(dc: dict, g: _Gen, std: float = 0.02, out_wstd: float | None = None) -> Dict[str, Tuple[str, np.ndarray]]:
    """speech_tokenizer tensors with the upstream key names / PyTorch layouts that
    sanitizeSpeechTokenizerWeights (Qwen3.swift:1498-1750) expects as input. `out_wstd`: std of the tail conv's weights;
    at the real layer widths 0.0012 keeps the waveform inside (-1, 1) (the real checkpoint: block3 rms 8.3 -> audio std
    0.17, Tests/Qwen3TTSTests/Qwen3TTSTests.swift:231, :271) so that the final clip does not hide errors."""
    out: Dict[str, Tuple[str, np.ndarray]] = {}
    cd, latent, dd = dc["codebook_dim"], dc["latent_dim"], dc["decoder_dim"]
    inner = cd // 2
    hs, isz = dc["hidden_size"], dc["intermediate_size"]
    nh, hd = dc["num_attention_heads"], dc["head_dim"]

    def codebook(prefix, size):
        emb = g.normal((size, inner), 1.0)
        usage = g.uniform((size,), 0.5, 2.0)
        usage[::7] = 0.0  # exercises the clip(usage, 1e-5) branch (Qwen3.swift:1721)
        out[prefix + "._codebook.cluster_usage"] = _f32(usage)
        out[prefix + "._codebook.embedding_sum"] = _f32(emb * np.maximum(usage, 1e-5)[:, None])

    codebook("decoder.quantizer.rvq_first.vq.layers.0", dc["semantic_codebook_size"])
    for i in range(dc["num_quantizers"] - dc["num_semantic_quantizers"]):
        codebook(f"decoder.quantizer.rvq_rest.vq.layers.{i}", dc["codebook_size"])
    for r in ("rvq_first", "rvq_rest"):
        out[f"decoder.quantizer.{r}.input_proj.weight"] = _f32(g.normal((inner, cd, 1), std))
        out[f"decoder.quantizer.{r}.output_proj.weight"] = _f32(g.normal((cd, inner, 1), 0.06))

    def conv(prefix, cout, cin_g, k, wstd=std):  # torch Conv1d [out, in/groups, k]
        out[prefix + ".weight"] = _f32(g.normal((cout, cin_g, k), wstd))
        out[prefix + ".bias"] = _f32(g.normal((cout,), std))

    def convtr(prefix, cin, cout, k):  # torch ConvTranspose1d [in, out, k]
        out[prefix + ".weight"] = _f32(g.normal((cin, cout, k), std))
        out[prefix + ".bias"] = _f32(g.normal((cout,), std))

    def lin(prefix, n, k, bias):
        out[prefix + ".weight"] = _f32(g.normal((n, k), std))
        if bias:
            out[prefix + ".bias"] = _f32(g.normal((n,), std))

    def snake(prefix, c):
        out[prefix + ".alpha"] = _f32(g.normal((c,), 0.1))
        out[prefix + ".beta"] = _f32(g.normal((c,), 0.1))

    conv("decoder.pre_conv.conv", latent, cd, 3)
    pt = "decoder.pre_transformer"
    lin(pt + ".input_proj", hs, latent, True)
    lin(pt + ".output_proj", latent, hs, True)
    for l in range(dc["num_hidden_layers"]):
        p = f"{pt}.layers.{l}"
        lin(p + ".self_attn.q_proj", nh * hd, hs, False)
        lin(p + ".self_attn.k_proj", dc["num_key_value_heads"] * hd, hs, False)
        lin(p + ".self_attn.v_proj", dc["num_key_value_heads"] * hd, hs, False)
        lin(p + ".self_attn.o_proj", hs, nh * hd, False)
        lin(p + ".mlp.gate_proj", isz, hs, False)
        lin(p + ".mlp.up_proj", isz, hs, False)
        lin(p + ".mlp.down_proj", hs, isz, False)
        out[p + ".input_layernorm.weight"] = _f32(g.normal((hs,), 0.05, 1.0))
        out[p + ".post_attention_layernorm.weight"] = _f32(g.normal((hs,), 0.05, 1.0))
        # LayerScale 0.01 nominal; perturbed so a dropped multiply shows
        out[p + ".self_attn_layer_scale.scale"] = _f32(g.normal((hs,), 0.002, dc["layer_scale_initial_scale"]))
        out[p + ".mlp_layer_scale.scale"] = _f32(g.normal((hs,), 0.002, dc["layer_scale_initial_scale"]))
    out[pt + ".norm.weight"] = _f32(g.normal((hs,), 0.05, 1.0))
    for i, r in enumerate(dc["upsampling_ratios"]):
        convtr(f"decoder.upsample.{i}.0.conv", latent, latent, r)
        p = f"decoder.upsample.{i}.1"
        conv(p + ".dwconv.conv", latent, 1, 7, wstd=0.3)
        out[p + ".norm.weight"] = _f32(g.normal((latent,), 0.05, 1.0))
        out[p + ".norm.bias"] = _f32(g.normal((latent,), 0.02))
        lin(p + ".pwconv1", 4 * latent, latent, True)
        lin(p + ".pwconv2", latent, 4 * latent, True)
        out[p + ".gamma"] = _f32(g.normal((latent,), 0.02, 0.1))
    conv("decoder.decoder.0.conv", dd, latent, 7)
    c = dd
    for b, rate in enumerate(dc["upsample_rates"]):
        p = f"decoder.decoder.{b + 1}.block"
        snake(p + ".0", c)
        convtr(p + ".1.conv", c, c // 2, 2 * rate)
        c //= 2
        for j in (2, 3, 4):
            snake(f"{p}.{j}.act1", c)
            conv(f"{p}.{j}.conv1.conv", c, c, 7)
            snake(f"{p}.{j}.act2", c)
            conv(f"{p}.{j}.conv2.conv", c, c, 1, wstd=0.05)
    snake("decoder.decoder.5", c)
    conv("decoder.decoder.6.conv", 1, c, 7, wstd=0.05 if out_wstd is None else out_wstd)
    return out


def encoder_tensors(ec: dict, g: _Gen) -> Dict[str, Tuple[str, np.ndarray]]:
    """speech_tokenizer encoder half with the upstream (HF Mimi style) key names and PyTorch layouts that the
    sanitiser's encoder branch consumes (Qwen3.swift:1517-1528, 1546-1565, 1592-1700). Fan-in scaled weights so
    the waveform still drives the codes after ~20 layers."""
    out: Dict[str, Tuple[str, np.ndarray]] = {}
    nf, hs = ec["num_filters"], ec["hidden_size"]

    def conv(prefix, cout, cin, k, bias=True):  # torch Conv1d [out, in, k]
        out[prefix + ".weight"] = _f32(g.normal((cout, cin, k), 1.0 / np.sqrt(cin * k)))
        if bias:
            out[prefix + ".bias"] = _f32(g.normal((cout,), 0.02))

    conv("encoder.encoder.layers.0.conv", nf, ec["audio_channels"], ec["kernel_size"])
    mult, idx = 1, 1
    for ratio in reversed(ec["upsampling_ratios"]):  # layer indices 1,3 | 4,6 | 7,9 | 10,12 (ELUs in between)
        c = mult * nf
        conv(f"encoder.encoder.layers.{idx}.block.1.conv", c // ec["compress"], c, ec["residual_kernel_size"])
        conv(f"encoder.encoder.layers.{idx}.block.3.conv", c, c // ec["compress"], 1)
        conv(f"encoder.encoder.layers.{idx + 2}.conv", 2 * c, c, 2 * ratio)
        mult *= 2
        idx += 3
    conv("encoder.encoder.layers.14.conv", hs, mult * nf, ec["last_kernel_size"])
    nh = ec["num_attention_heads"]
    hd = hs // nh
    for l in range(ec["num_hidden_layers"]):
        p = f"encoder.encoder_transformer.layers.{l}"
        for nm, n, k in (("q_proj", nh * hd, hs), ("k_proj", ec["num_key_value_heads"] * hd, hs),
                         ("v_proj", ec["num_key_value_heads"] * hd, hs), ("o_proj", hs, nh * hd)):
            out[f"{p}.self_attn.{nm}.weight"] = _f32(g.normal((n, k), 1.0 / np.sqrt(k)))
        out[p + ".mlp.fc1.weight"] = _f32(g.normal((ec["intermediate_size"], hs), 1.0 / np.sqrt(hs)))
        out[p + ".mlp.fc2.weight"] = _f32(g.normal((hs, ec["intermediate_size"]), 1.0 / np.sqrt(ec["intermediate_size"])))
        for nm in ("input_layernorm", "post_attention_layernorm"):
            out[f"{p}.{nm}.weight"] = _f32(g.normal((hs,), 0.05, 1.0))
            out[f"{p}.{nm}.bias"] = _f32(g.normal((hs,), 0.02))
        out[p + ".self_attn_layer_scale.scale"] = _f32(g.normal((hs,), 0.002, ec["layer_scale_initial_scale"]))
        out[p + ".mlp_layer_scale.scale"] = _f32(g.normal((hs,), 0.002, ec["layer_scale_initial_scale"]))
    enc_rate = ec["sampling_rate"] / int(np.prod(ec["upsampling_ratios"]))
    ds = int(enc_rate / ec["frame_rate"])
    conv("encoder.downsample.conv", hs, hs, 2 * ds, bias=False)
    cd, bins = ec["codebook_dim"], ec["codebook_size"]
    for nm, nq in (("semantic_residual_vector_quantizer", 1), ("acoustic_residual_vector_quantizer", ec["num_quantizers"] - 1)):
        q = f"encoder.quantizer.{nm}"
        out[q + ".input_proj.weight"] = _f32(g.normal((cd, hs, 1), 1.0 / np.sqrt(hs)))
        out[q + ".output_proj.weight"] = _f32(g.normal((hs, cd, 1), 1.0 / np.sqrt(cd)))
        for j in range(nq):
            # residual energy shrinks layer by layer; scale the codebooks with it so later layers stay informative
            emb = g.normal((bins, cd), 1.0) * np.float32(0.2 * 0.8 ** j)
            usage = g.uniform((bins,), 0.5, 2.0)
            usage[::11] = 0.0  # the max(usage, 1e-5) branch (SpeechTokenizerEncoder.swift:739)
            out[f"{q}.layers.{j}.codebook.cluster_usage"] = _f32(usage)
            out[f"{q}.layers.{j}.codebook.embed_sum"] = _f32(emb * np.maximum(usage, 1e-5)[:, None])
            out[f"{q}.layers.{j}.codebook.initialized"] = _f32(np.ones((1,), np.float32))
    return out


def speaker_encoder_tensors(sc: dict, g: _Gen) -> Dict[str, Tuple[str, np.ndarray]]:
    """ECAPA-TDNN parameters in the main checkpoint: bf16, PyTorch Conv1d layout [out, in, k]; Qwen3TTSModel.sanitize
    transposes them (Qwen3.swift:1231-1237). Module tree: SpeakerEncoder.swift:283-297, 163-166, 77, 123-124, 219-220."""
    out: Dict[str, Tuple[str, np.ndarray]] = {}
    ch, ks = sc["enc_channels"], sc["enc_kernel_sizes"]
    scale = sc["enc_res2net_scale"]

    def conv(prefix, cout, cin, k, gain=1.0):
        out[prefix + ".weight"] = _bf16(g.normal((cout, cin, k), gain / np.sqrt(cin * k)))
        out[prefix + ".bias"] = _bf16(g.normal((cout,), 0.05))

    conv("speaker_encoder.blocks.0.conv", ch[0], sc["mel_dim"], ks[0], gain=0.3)  # log-mel inputs are O(5)
    for bi in (1, 2, 3):
        p = f"speaker_encoder.blocks.{bi}"
        conv(p + ".tdnn1.conv", ch[bi], ch[bi - 1], 1)
        for j in range(scale - 1):
            conv(f"{p}.res2net_block.blocks.{j}.conv", ch[bi] // scale, ch[bi] // scale, ks[bi])
        conv(p + ".tdnn2.conv", ch[bi], ch[bi], 1)
        conv(p + ".se_block.conv1", sc["enc_se_channels"], ch[bi], 1)
        conv(p + ".se_block.conv2", ch[bi], sc["enc_se_channels"], 1)
    conv("speaker_encoder.mfa.conv", ch[4], ch[1] + ch[2] + ch[3], ks[4])
    conv("speaker_encoder.asp.tdnn.conv", sc["enc_attention_channels"], 3 * ch[4], 1)
    conv("speaker_encoder.asp.conv", ch[4], sc["enc_attention_channels"], 1)
    conv("speaker_encoder.fc", sc["enc_dim"], 2 * ch[4], 1, gain=0.01)  # x-vector at the scale of an embedding row
    return out


def synthetic_reference_audio(row: int = 0, seconds: float = 3.0, sample_rate: int = 24000) -> np.ndarray:
    """BASELINE config 4: white noise x0.1, seed 99 (+row)."""
    rng = np.random.Generator(np.random.PCG64(99 + row))
    return (rng.standard_normal(int(seconds * sample_rate), dtype=np.float32) * np.float32(0.1)).astype(np.float32)


def write_checkpoint(model_dir: str, name: str = "tiny-a", seed: int = 1234,
                     overrides: dict | None = None) -> dict:
    """Write a synthetic checkpoint for preset `name` into `model_dir`. Returns the preset dict."""
    p = preset(name)
    if overrides:
        for k, v in overrides.items():
            d = p["config"]
            parts = k.split(".")
            for q in parts[:-1]:
                d = d[q]
            d[parts[-1]] = v
    big = name in ("0.6b", "1.7b", "0.6b-q4", "0.6b-base", "1.7b-base")
    g = _Gen(seed, big)
    os.makedirs(os.path.join(model_dir, "speech_tokenizer"), exist_ok=True)
    with open(os.path.join(model_dir, "config.json"), "w") as f:
        json.dump(p["config"], f, indent=1)
    with open(os.path.join(model_dir, "speech_tokenizer", "config.json"), "w") as f:
        json.dump(p["speech_tokenizer"], f, indent=1)
    main = talker_tensors(p["config"], g)
    codec = codec_tensors(p["speech_tokenizer"]["decoder_config"], g, out_wstd=FULL_WIDTH_OUT_WSTD if big else None)
    if p["config"].get("speaker_encoder_config"):
        main.update(speaker_encoder_tensors(p["config"]["speaker_encoder_config"], g))
    if p["speech_tokenizer"].get("encoder_config"):
        codec.update(encoder_tensors(p["speech_tokenizer"]["encoder_config"], g))
    if p.get("codec_f16"):  # Float32 -> float16 conversion of every speech-tokenizer tensor (docs/paper.tex:207)
        codec = {k: (("F16", v.astype(np.float16)) if tag == "F32" else (tag, v)) for k, (tag, v) in codec.items()}
    save_safetensors(os.path.join(model_dir, "model.safetensors"), main)
    save_safetensors(os.path.join(model_dir, "speech_tokenizer", "model.safetensors"), codec)
    return p


def synthetic_prompt(row: int, n_text: int = 32, n_instruct: int = 0, text_vocab: int = 151643,
                     im_start: int = 151644, im_end: int = 151645) -> dict:
    """Fixed-length synthetic token ids shaped like the reference's chat template
    (Qwen3.swift:274-275, 364-365): '<|im_start|>assistant\\n' + text + '<|im_end|>\\n<|im_start|>assistant\\n'.
    BASELINE.md section 3: 32 uniform ids, seed 7 + row."""
    rng = np.random.Generator(np.random.PCG64(7 + row))
    nl, assistant, user = 198 % text_vocab, 77091 % text_vocab, 872 % text_vocab
    body = rng.integers(0, text_vocab, size=n_text).astype(np.int32).tolist()
    text_ids = [im_start, assistant, nl] + body + [im_end, nl, im_start, assistant, nl]
    out = {"text_ids": text_ids, "target_token_count": n_text}

    if n_instruct:
        ib = rng.integers(0, text_vocab, size=n_instruct).astype(np.int32).tolist()
        out["instruct_ids"] = [im_start, user, nl] + ib + [im_end, nl]
    # voice clone: tokens of "<|im_start|>assistant\n{refText}<|im_end|>\n" (Qwen3.swift:448-449); own stream so
    # the ids above do not depend on it
    rng2 = np.random.Generator(np.random.PCG64(1007 + row))
    ref_body = rng2.integers(0, text_vocab, size=max(4, n_text // 2)).astype(np.int32).tolist()
    out["ref_text_ids"] = [im_start, assistant, nl] + ref_body + [im_end, nl]
    return out
