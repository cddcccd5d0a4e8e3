"""Independent cross-checks of the oracle's COMPOSITIONS against Hugging Face reference implementations of the same
blocks: `transformers.models.qwen3_omni_moe` holds the upstream Code2Wav decoder blocks the
reference's codec decoder was ported from (SnakeBeta, causal conv, residual unit, ConvNeXt block, decoder block), and
`transformers.models.mimi` is the upstream of the codec ENCODER (SEANet + causal transformer + stride-2 downsample +
split RVQ): the synthetic checkpoints use Mimi's own key names, so its weights load into `MimiModel` by name, which also
checks the sanitiser's index -> name mapping (Qwen3.swift:1517-1528) against the real module tree.

Divergences respected (reference behaviour on the left, restated by the oracle):
  * transposed convs are trimmed on the right only (SpeechTokenizer.swift:346-351); HF trims K - s on both sides, so
    HF's output is the oracle's minus its first K - s samples;
  * the encoder MLP uses the tanh GELU approximation (SpeechTokenizerEncoder.swift:1080-1082): HF configured with
    hidden_act = gelu_pytorch_tanh;
  * the stride-2 downsample conv zero-pads (`padMode: "edge"` is stored but never read, :184, :697): HF's replicate
    padding is switched to constant for the comparison;
  * no 250-frame sliding window in the encoder transformer (:1039-1043): clips stay below 250 frames.
CPU only; nothing here runs on the GPU box's product path."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def base(tmp_path_factory):
    from oracle import oracle as O
    from qwen3tts import synth
    d = str(tmp_path_factory.mktemp("tiny_base_hf"))
    synth.write_checkpoint(d, "tiny-base", seed=2024)
    return d, O.OracleModel(d)


def t2n(x):
    return x.detach().cpu().numpy().astype(np.float32)


def close(a, b, tol):
    return float(np.abs(a - b).max()) <= tol * max(1.0, float(np.abs(b).max()))


def _torch_conv_weight(w_mlx):  # oracle / MLX [O][K][I/g] -> torch [O][I/g][K]
    return torch.from_numpy(np.ascontiguousarray(np.transpose(w_mlx, (0, 2, 1))))


def _load_causal_conv(mod, cw, prefix):
    mod.conv.weight.data = _torch_conv_weight(cw[prefix + ".weight"])
    mod.conv.bias.data = torch.from_numpy(cw[prefix + ".bias"].copy())


def _load_snake(mod, cw, prefix):
    mod.alpha.data = torch.from_numpy(cw[prefix + ".alpha"].copy())
    mod.beta.data = torch.from_numpy(cw[prefix + ".beta"].copy())


# ---------------------------------------------------------------------------------------------------------
# codec decoder blocks vs transformers.models.qwen3_omni_moe (C3, C6, C7)
# ---------------------------------------------------------------------------------------------------------
def test_convnext_block_matches_hf(base):
    from transformers.models.qwen3_omni_moe import modeling_qwen3_omni_moe as M
    _, om = base
    cw = om.codec
    p = "decoder.upsample.0.1"
    C_ = cw[p + ".gamma"].shape[0]
    blk = M.Qwen3OmniMoeConvNeXtBlock(C_).eval()
    _load_causal_conv(blk.dwconv, cw, p + ".dwconv.conv")
    for nm in ("norm", "pwconv1", "pwconv2"):
        getattr(blk, nm).weight.data = torch.from_numpy(cw[f"{p}.{nm}.weight"].copy())
        getattr(blk, nm).bias.data = torch.from_numpy(cw[f"{p}.{nm}.bias"].copy())
    blk.gamma.data = torch.from_numpy(cw[p + ".gamma"].copy())
    x = np.random.default_rng(1).standard_normal((23, C_)).astype(np.float32)
    with torch.no_grad():
        want = t2n(blk(torch.from_numpy(x.T[None]))[0].T)
    got = om._convnext(x, p)
    assert got.shape == want.shape and close(got, want, 2e-6), float(np.abs(got - want).max())


@pytest.mark.parametrize("j,dil", [(1, 1), (2, 3), (3, 9)])
def test_residual_unit_matches_hf(base, j, dil):
    from transformers.models.qwen3_omni_moe import modeling_qwen3_omni_moe as M
    _, om = base
    cw = om.codec
    rp = f"decoder.decoder.block1.res{j}"
    C_ = cw[rp + ".act1.alpha"].shape[0]
    unit = M.Qwen3OmniMoeCode2WavDecoderResidualUnit(C_, dil).eval()
    _load_snake(unit.act1, cw, rp + ".act1")
    _load_snake(unit.act2, cw, rp + ".act2")
    _load_causal_conv(unit.conv1, cw, rp + ".conv1.conv")
    _load_causal_conv(unit.conv2, cw, rp + ".conv2.conv")
    x = np.random.default_rng(j).standard_normal((61, C_)).astype(np.float32)
    with torch.no_grad():
        want = t2n(unit(torch.from_numpy(x.T[None]))[0].T)
    got = om._resunit(x, rp, dil)
    assert close(got, want, 2e-6), float(np.abs(got - want).max())


def test_decoder_block_matches_hf_behind_the_trim_offset(base):
    """Whole DecoderBlock. HF's transposed conv drops K - s samples on BOTH sides, the reference on the right only, so
    HF's block output equals the oracle's shifted by K - s once the causal receptive field of the three residual units
    (6 * (1 + 3 + 9) = 78 samples) no longer reaches the dropped samples."""
    from transformers.models.qwen3_omni_moe import modeling_qwen3_omni_moe as M
    _, om = base
    cw, dc = om.codec, om.dc
    b = 1
    rate = dc["upsample_rates"][b]
    p = f"decoder.decoder.block{b}"
    cin, cout = cw[p + ".snake.alpha"].shape[0], cw[p + ".res1.act1.alpha"].shape[0]
    sn = M.Qwen3OmniMoeSnakeBeta(cin).eval()
    _load_snake(sn, cw, p + ".snake")
    tc = M.Qwen3OmniMoeCausalTransConvNet(cin, cout, 2 * rate, rate).eval()
    w = cw[p + ".upsample.conv.weight"]  # oracle / MLX ConvTransposed1d [O][K][I] -> torch [I][O][K]
    tc.conv.weight.data = torch.from_numpy(np.ascontiguousarray(np.transpose(w, (2, 0, 1))))
    tc.conv.bias.data = torch.from_numpy(cw[p + ".upsample.conv.bias"].copy())
    units = []
    for j, dil in ((1, 1), (2, 3), (3, 9)):
        u = M.Qwen3OmniMoeCode2WavDecoderResidualUnit(cout, dil).eval()
        rp = f"{p}.res{j}"
        _load_snake(u.act1, cw, rp + ".act1")
        _load_snake(u.act2, cw, rp + ".act2")
        _load_causal_conv(u.conv1, cw, rp + ".conv1.conv")
        _load_causal_conv(u.conv2, cw, rp + ".conv2.conv")
        units.append(u)
    T = 40
    x = np.random.default_rng(5).standard_normal((T, cin)).astype(np.float32)
    with torch.no_grad():
        h = tc(sn(torch.from_numpy(x.T[None])))
        up_hf = t2n(h[0].T)
        for u in units:
            h = u(h)
        want = t2n(h[0].T)
    off = 2 * rate - rate  # K - s
    got_up = om._convtr(om._snake(x, p + ".snake"), p + ".upsample.conv", 2 * rate, rate)
    assert got_up.shape[0] == T * rate and up_hf.shape[0] == T * rate - off
    assert close(got_up[off:], up_hf, 2e-6)  # the transposed conv alone: exact relation
    got = om._decoder_block(x, p, rate)
    rf = 6 * (1 + 3 + 9)
    assert close(got[off + rf:], want[rf:], 5e-6), float(np.abs(got[off + rf:] - want[rf:]).max())


def test_decoder_transformer_layer_matches_hf_without_positions(base):
    """DecoderTransformerLayer (C5). The reference applies neither RoPE nor a mask (SpeechTokenizer.swift:512-528, 763)
    although upstream does: HF's layer is driven with cos = 1, sin = 0 and no mask, which removes exactly those two."""
    from transformers.models.qwen3_omni_moe import modeling_qwen3_omni_moe as M
    from transformers.models.qwen3_omni_moe.configuration_qwen3_omni_moe import Qwen3OmniMoeCode2WavConfig
    _, om = base
    cw, dc = om.codec, om.dc
    p = "decoder.pre_transformer.layers.1"
    nh, hd = dc["num_attention_heads"], dc["head_dim"]
    hidden, inter = cw[p + ".self_attn.q_proj.weight"].shape[1], cw[p + ".mlp.gate_proj.weight"].shape[0]
    cfg = Qwen3OmniMoeCode2WavConfig(hidden_size=hidden, intermediate_size=inter, num_attention_heads=nh,
                                     num_key_value_heads=nh, head_dim=hd, rms_norm_eps=dc["rms_norm_eps"],
                                     num_hidden_layers=2, sliding_window=10000, attention_bias=False, hidden_act="silu")
    cfg._attn_implementation = "eager"
    layer = M.Qwen3OmniMoeCode2WavTransformerLayer(cfg, 0).eval()
    sd = {k[len(p) + 1:]: torch.from_numpy(v.copy()) for k, v in cw.items() if k.startswith(p + ".")}
    res = layer.load_state_dict(sd, strict=True)
    F = 19
    x = np.random.default_rng(11).standard_normal((F, hidden)).astype(np.float32)
    cos, sin = torch.ones(1, F, hd), torch.zeros(1, F, hd)
    with torch.no_grad():
        want = t2n(layer(torch.from_numpy(x[None]), attention_mask=None, position_embeddings=(cos, sin))[0])
    got = om._dec_transformer_layer(x, p, nh, hd)
    assert close(got, want, 3e-6), float(np.abs(got - want).max())


# ---------------------------------------------------------------------------------------------------------
# codec encoder vs transformers.models.mimi (V1)
# ---------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def mimi(base):
    from safetensors.numpy import load_file
    from transformers.models.mimi import modeling_mimi as MM
    from transformers.models.mimi.configuration_mimi import MimiConfig
    d, om = base
    ec = om.ec
    cfg = MimiConfig(sampling_rate=ec["sampling_rate"], audio_channels=ec["audio_channels"], hidden_size=ec["hidden_size"],
                     num_filters=ec["num_filters"], num_residual_layers=ec["num_residual_layers"],
                     upsampling_ratios=list(ec["upsampling_ratios"]), kernel_size=ec["kernel_size"],
                     last_kernel_size=ec["last_kernel_size"], residual_kernel_size=ec["residual_kernel_size"],
                     dilation_growth_rate=ec["dilation_growth_rate"], use_causal_conv=True, pad_mode="constant",
                     compress=ec["compress"], codebook_size=ec["codebook_size"], codebook_dim=ec["codebook_dim"],
                     num_quantizers=ec["num_quantizers"], use_conv_shortcut=False,
                     vector_quantization_hidden_dimension=ec["codebook_dim"], num_semantic_quantizers=1,
                     upsample_groups=ec["hidden_size"], num_hidden_layers=ec["num_hidden_layers"],
                     intermediate_size=ec["intermediate_size"], num_attention_heads=ec["num_attention_heads"],
                     num_key_value_heads=ec["num_key_value_heads"], head_dim=ec["hidden_size"] // ec["num_attention_heads"],
                     hidden_act="gelu_pytorch_tanh", norm_eps=1e-5, sliding_window=ec["sliding_window"],
                     layer_scale_initial_scale=ec["layer_scale_initial_scale"],
                     rope_parameters={"rope_type": "default", "rope_theta": ec["rope_theta"]})
    cfg._attn_implementation = "eager"
    model = MM.MimiModel(cfg).eval()
    raw = load_file(os.path.join(d, "speech_tokenizer", "model.safetensors"))
    sd = {k[len("encoder."):]: torch.from_numpy(v.copy()) for k, v in raw.items() if k.startswith("encoder.")}
    res = model.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys  # every encoder tensor of the checkpoint has a home in Mimi
    enc_missing = [k for k in res.missing_keys
                   if k.startswith(("encoder.", "encoder_transformer.", "downsample.")) or ("quantizer" in k and "codebook" in k)]
    assert not enc_missing, enc_missing
    model.downsample.pad_mode = "constant"  # the reference zero-pads here (module docstring)
    return model


def test_mimi_encoder_stages_match_hf(base, mimi):
    _, om = base
    from qwen3tts import synth
    audio = synth.synthetic_reference_audio(3, 1.3)
    st = {}
    codes = om.codec_encode(audio, st)
    x = torch.from_numpy(audio.reshape(1, 1, -1))
    with torch.no_grad():
        e = mimi.encoder(x)
        assert close(st["seanet"], t2n(e[0].T), 3e-5), float(np.abs(st["seanet"] - t2n(e[0].T)).max())
        tr = mimi.encoder_transformer(e.transpose(1, 2), return_dict=True).last_hidden_state
        assert close(st["transformer"], t2n(tr[0]), 5e-5), float(np.abs(st["transformer"] - t2n(tr[0])).max())
        ds = mimi.downsample(tr.transpose(1, 2))
        assert close(st["downsample"], t2n(ds[0].T), 5e-5)
        assert ds.shape[-1] == codes.shape[1] == om.codec_encode(audio).shape[1]
        # RVQ search on the ORACLE's downsampled frames (so that both sides quantise identical inputs)
        hf_codes = mimi.quantizer.encode(torch.from_numpy(st["downsample"].T[None].copy()), 32)[:, 0].numpy()
    assert hf_codes.shape == (32, ds.shape[-1]) and codes.shape == (16, ds.shape[-1])  # the reference keeps 16 (:1055)
    # layer by layer; a mismatch is only acceptable at a near-tie (HF minimises ||x - e||, the reference ||e||^2/2 - x.e)
    # and hides the later layers of that frame within its quantiser (semantic: layer 0; acoustic: layers 1..31)
    alive = np.ones(codes.shape[1], bool)
    n_diff = 0
    gaps = st["gaps"]
    for layer in range(16):
        if layer == 1:
            alive[:] = True
        diff = (hf_codes[layer] != codes[layer]) & alive
        assert (gaps[layer][diff] < 1e-4).all(), (layer, gaps[layer][diff])
        n_diff += int(diff.sum())
        alive &= ~diff
    assert n_diff <= 2


def test_mimi_full_encode_matches_hf(base, mimi):
    """MimiModel.encode end to end (the whole of V1) against the oracle's codec_encode: first 16 codebooks."""
    _, om = base
    from qwen3tts import synth
    audio = synth.synthetic_reference_audio(1, 0.9)
    st = {}
    want = om.codec_encode(audio, st)
    with torch.no_grad():
        got = mimi.encode(torch.from_numpy(audio.reshape(1, 1, -1)), num_quantizers=32).audio_codes[0, :16].numpy()
    assert got.shape == want.shape
    alive = np.ones(want.shape[1], bool)
    for layer in range(16):
        if layer == 1:
            alive[:] = True
        diff = (got[layer] != want[layer]) & alive
        # upstream activations differ at the 1e-5 level between the two implementations: accept flips at small gaps only
        assert (st["gaps"][layer][diff] < 1e-3).all(), (layer, st["gaps"][layer][diff])
        alive &= ~diff
    assert (got == want).mean() > 0.9


# ---------------------------------------------------------------------------------------------------------
# the code predictor's frame loop vs transformers' Qwen3-Omni talker code predictor (A12)
# ---------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def tiny_a(tmp_path_factory):
    from oracle import oracle as O
    from qwen3tts import synth
    d = str(tmp_path_factory.mktemp("tiny_a_hf"))
    synth.write_checkpoint(d, "tiny-a", seed=2025)   # talker and predictor share the hidden size: no small_to_mtp projection
    return d, O.OracleModel(d)


def test_code_predictor_frame_matches_hf_qwen3_omni_code_predictor(tiny_a):
    """Qwen3-TTS's code predictor is the Qwen3-Omni talker's (same upstream): 5 Qwen3 layers over a per-frame cache, fifteen
    embedding tables and fifteen heads. What this pins against an implementation the oracle was not written from is the
    INDEXING of a frame (CodePredictor.swift:320-339, Qwen3.swift:879-909): pass 0 takes two positions [hidden, embed(c0)]
    and reads lm_head[0] at the last one; pass s >= 1 takes table s - 1 at code s and reads lm_head[s]; positions run
    0, 1, 2, ... over one cache. The checkpoint's tensors load into the HF module by their own names."""
    from transformers.models.qwen3_omni_moe import modeling_qwen3_omni_moe as M
    from transformers.models.qwen3_omni_moe.configuration_qwen3_omni_moe import Qwen3OmniMoeTalkerCodePredictorConfig
    from oracle import oracle as O
    d, om = tiny_a
    assert not om.has_proj          # (HF's module embeds the codes itself: a projection in front of it could not be injected)
    import json
    cpc = json.load(open(os.path.join(d, "config.json")))["talker_config"]["code_predictor_config"]
    cfg = Qwen3OmniMoeTalkerCodePredictorConfig(
        vocab_size=cpc["vocab_size"], hidden_size=cpc["hidden_size"], intermediate_size=cpc["intermediate_size"],
        num_hidden_layers=cpc["num_hidden_layers"], num_attention_heads=cpc["num_attention_heads"],
        num_key_value_heads=cpc["num_key_value_heads"], head_dim=cpc["head_dim"], rms_norm_eps=cpc["rms_norm_eps"],
        rope_theta=cpc["rope_theta"], max_position_embeddings=64, num_code_groups=cpc["num_code_groups"], attention_bias=False,
        use_sliding_window=False)
    cfg._attn_implementation = "eager"
    hf = M.Qwen3OmniMoeTalkerCodePredictorModelForConditionalGeneration(cfg).float().eval()
    pre = "talker.code_predictor."
    sd = {k[len(pre):]: torch.from_numpy(O.bf16_to_f32(v)) for k, v in om.w.items()
          if k.startswith(pre) and "small_to_mtp" not in k}
    missing, unexpected = hf.load_state_dict(sd, strict=False)
    assert not unexpected and all("rotary" in k or "inv_freq" in k for k in missing), (missing, unexpected)

    rng = np.random.default_rng(17)
    ncg, Vc, H = cpc["num_code_groups"], cpc["vocab_size"], cpc["hidden_size"]
    worst, agree, confident = 0.0, 0, 0
    for frame in range(3):
        hidden = O.f32_to_bf16(rng.standard_normal((1, H)).astype(np.float32))
        c0 = int(rng.integers(0, 2048))
        codes = [c0]
        cache = om.cpm.new_cache(ncg + 1)
        past = None
        for s in range(ncg - 1):
            if s == 0:
                x = np.concatenate([hidden, om.codec_embed([c0])], 0)
                with torch.no_grad():
                    out = hf(inputs_embeds=torch.from_numpy(O.bf16_to_f32(x))[None], use_cache=True)
            else:
                x = om.cp_embed(s - 1, [codes[s]])
                with torch.no_grad():
                    out = hf(input_ids=torch.tensor([[codes[s]]]), generation_steps=s, past_key_values=past, use_cache=True)
            past = out.past_key_values
            ref = t2n(out.logits[0, -1])
            got = O.bf16_to_f32(om.cp_forward(cache, x, s)[-1])
            assert got.shape == ref.shape == (Vc,)
            worst = max(worst, float(np.abs(got - ref).max()) / max(1.0, float(np.abs(ref).max())))
            top2 = np.sort(ref)[-2:]
            if top2[1] - top2[0] > 0.05 * max(1.0, float(np.abs(ref).max())):   # a winner no bf16 rounding can flip
                confident += 1
                agree += int(np.argmax(got) == np.argmax(ref))
            codes.append(int(np.argmax(got)))
    # bf16 storage at every op vs fp32 HF over <= 16 positions: the bar of the decoder-layer test (test_oracle_blocks.py)
    assert worst < 0.05, worst
    assert confident >= 10 and agree == confident, (agree, confident)


# ---------------------------------------------------------------------------------------------------------
# the speaker encoder vs transformers' ECAPA_TimeDelayNet (V2)
# ---------------------------------------------------------------------------------------------------------
def test_speaker_encoder_matches_hf_ecapa_tdnn(base):
    """Qwen3TTSSpeakerEncoder (SpeakerEncoder.swift:278-395) is the ECAPA-TDNN of Qwen2.5-Omni's token2wav, which transformers
    carries as `ECAPA_TimeDelayNet` with the same block names -- so the checkpoint's `speaker_encoder.*` tensors load into it BY
    NAME, in the torch conv layout the checkpoint stores (which also checks the oracle's conv-layout sanitiser, Qwen3.swift:
    1246-1260, from the other side: the oracle transposes what HF takes as is). Same log-mel in, every block's output and the
    x-vector out: reflect 'same' padding, the Res2Net chain (chunk i takes chunk + previous output from i = 2 on), squeeze-
    excitation over the time mean, multi-layer aggregation of blocks 1-3, attentive statistics pooling, the final 1x1 conv."""
    import types
    from transformers.models.qwen2_5_omni import modeling_qwen2_5_omni as M
    from oracle import oracle as O
    from qwen3tts import synth
    d, om = base
    raw = O.load_safetensors_dir(d)
    hf = M.ECAPA_TimeDelayNet(types.SimpleNamespace(**om.sc)).float().eval()
    sd = {k[len("speaker_encoder."):]: torch.from_numpy(O.bf16_to_f32(v) if v.dtype == np.uint16 else v.astype(np.float32))
          for k, v in raw.items() if k.startswith("speaker_encoder.")}
    hf.load_state_dict(sd, strict=True)
    for row, seconds in ((0, 0.6), (1, 1.3)):
        audio = synth.synthetic_reference_audio(row, seconds)
        stages = {}
        emb = om.speaker_embedding(audio, stages)
        mel = torch.from_numpy(stages["mel"])[None]                    # [1][T][128]
        with torch.no_grad():
            x = mel.transpose(1, 2)
            outs = []
            for blk in hf.blocks:
                x = blk(x)
                outs.append(x)
            mfa = hf.mfa(torch.cat(outs[1:], dim=1))
            ref = hf(mel)[0]
        for name, t in (("h0", outs[0]), ("h1", outs[1]), ("h2", outs[2]), ("h3", outs[3]), ("mfa", mfa)):
            assert close(stages[name], t2n(t[0]).T, 2e-5), name
        assert close(emb, t2n(ref), 5e-5)


# ---------------------------------------------------------------------------------------------------------
# the decoder's split-RVQ dequantisation vs transformers' Mimi RVQ (C2)
# ---------------------------------------------------------------------------------------------------------
def test_decoder_rvq_dequantisation_matches_hf_mimi(base):
    """SplitResidualVectorQuantizer.decode (SpeechTokenizer.swift:175-227, 124-170, 61-97): codebook = embedding_sum /
    clip(cluster_usage, 1e-5) (the sanitiser's job, Qwen3.swift:1716-1724), the layers' rows summed, one bias-free 1x1
    output projection per half, the halves added. Mimi's `MimiResidualVectorQuantizer.decode` is the upstream of both halves:
    two instances (the reference's semantic table is larger than the acoustic ones, which one MimiConfig cannot say) take the
    checkpoint's raw `decoder.quantizer.rvq_{first,rest}.*` tensors -- sums and usages, not the divided tables -- so the
    division, the layer order and the projection are all HF's."""
    from safetensors.numpy import load_file
    from transformers.models.mimi import modeling_mimi as MM
    from transformers.models.mimi.configuration_mimi import MimiConfig
    d, om = base
    raw = load_file(os.path.join(d, "speech_tokenizer", "model.safetensors"))
    dc = om.dc

    def half(name, n_layers, bins):
        w_out = raw[f"decoder.quantizer.{name}.output_proj.weight"]       # torch layout [out][in][1]
        cfg = MimiConfig(codebook_size=bins, codebook_dim=w_out.shape[1], vector_quantization_hidden_dimension=w_out.shape[1],
                         hidden_size=w_out.shape[0], num_quantizers=max(n_layers, 2), num_semantic_quantizers=1)   # (the config wants > 1)
        q = MM.MimiResidualVectorQuantizer(cfg, n_layers).eval()
        sd = {"output_proj.weight": torch.from_numpy(w_out.copy()),
              "input_proj.weight": torch.from_numpy(raw[f"decoder.quantizer.{name}.input_proj.weight"].copy())}
        for j in range(n_layers):
            p = f"decoder.quantizer.{name}.vq.layers.{j}._codebook."
            sd[f"layers.{j}.codebook.embed_sum"] = torch.from_numpy(raw[p + "embedding_sum"].copy())
            sd[f"layers.{j}.codebook.cluster_usage"] = torch.from_numpy(raw[p + "cluster_usage"].copy())
            sd[f"layers.{j}.codebook.initialized"] = torch.tensor([1.0])
        q.load_state_dict(sd, strict=True)
        return q

    nsem, nq = dc["num_semantic_quantizers"], dc["num_quantizers"]
    first = half("rvq_first", nsem, dc["semantic_codebook_size"])
    rest = half("rvq_rest", nq - nsem, dc["codebook_size"])
    rng = np.random.default_rng(23)
    F = 9
    codes = np.concatenate([rng.integers(0, dc["semantic_codebook_size"], (F, nsem)),
                            rng.integers(0, dc["codebook_size"], (F, nq - nsem))], 1).astype(np.int64)
    stages = {}
    om._codec_front(codes, stages)
    ct = torch.from_numpy(codes.T[None])                                   # [1][K][F]
    with torch.no_grad():
        ref = first.decode(ct[:, :nsem]) + rest.decode(ct[:, nsem:])       # [1][C][F]
    assert close(stages["quantizer"], t2n(ref[0]).T, 1e-5)


def test_text_projection_matches_hf_resize_mlp(tiny_a):
    """ResizeMLP (Talker.swift:475-487) = transformers' Qwen3OmniMoeTalkerResizeMLP: fc2(silu(fc1(x))) with both biases. The HF
    module is built on a stand-in config (three sizes and the activation are all it reads) and takes the checkpoint's
    `talker.text_projection.*` tensors by name."""
    import types
    from transformers.models.qwen3_omni_moe import modeling_qwen3_omni_moe as M
    from oracle import oracle as O
    d, om = tiny_a
    w1 = om.w["talker.text_projection.linear_fc1.weight"]
    w2 = om.w["talker.text_projection.linear_fc2.weight"]
    cfg = types.SimpleNamespace(thinker_hidden_size=w1.shape[1],
                                text_config=types.SimpleNamespace(intermediate_size=w1.shape[0], hidden_size=w2.shape[0], hidden_act="silu"))
    hf = M.Qwen3OmniMoeTalkerResizeMLP(cfg).float().eval()
    sd = {k[len("talker.text_projection."):]: torch.from_numpy(O.bf16_to_f32(v)) for k, v in om.w.items()
          if k.startswith("talker.text_projection.")}
    hf.load_state_dict(sd, strict=True)
    x = O.f32_to_bf16(np.random.default_rng(29).standard_normal((7, w1.shape[1])).astype(np.float32))
    with torch.no_grad():
        ref = t2n(hf(torch.from_numpy(O.bf16_to_f32(x))))
    got = O.bf16_to_f32(om.text_projection(x))
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 0.03 * max(1.0, float(np.abs(ref).max()))   # bf16 storage after each of the three ops
