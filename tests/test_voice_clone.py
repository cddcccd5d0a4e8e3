"""Voice clone: codec encoder, speaker encoder + log-mel, ICL prompt and the
generateVoiceClone loop.

CPU tests pin the oracle's blocks against torch / numpy restatements and check the host logic; GPU tests
compare the HIP front end (through the C ABI) with the oracle on the same synthetic Base checkpoint.

Bars: fp32 activations within 2e-4 of the stage's scale (summation order differs: MFMA tiles vs the oracle's
sequential sums); RVQ codes bit-exact except where the oracle's two smallest distances are closer than 1e-4
(a near-tie that fp32 reordering may legitimately flip; later layers of that frame are then not compared);
prompt rows and sampled tokens under the LM bars of test_gpu_parity.py."""
import numpy as np
import pytest

from conftest import bf16_to_f32

ULP = 2.0 ** -7


@pytest.fixture(scope="module")
def base_dir(tmp_path_factory):
    from qwen3tts import synth
    d = str(tmp_path_factory.mktemp("tiny_base"))
    synth.write_checkpoint(d, "tiny-base", seed=4321)
    return d


@pytest.fixture(scope="module")
def oracle_base(base_dir):
    from oracle import oracle as O
    return O.OracleModel(base_dir)


def ref_audio(row=0, seconds=1.0):
    from qwen3tts import synth
    return synth.synthetic_reference_audio(row, seconds)


def clone_prompt(row=0, n_text=10):
    from qwen3tts import synth
    return synth.synthetic_prompt(row, n_text=n_text, text_vocab=1000, im_start=1000, im_end=1001)


# ---------------------------------------------------------------------------------------------------
# CPU: oracle blocks vs independent restatements
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("T,Cin,Cout,K,stride,dil", [(50, 8, 12, 7, 1, 1), (53, 16, 8, 8, 4, 1), (10, 4, 4, 3, 1, 2),
                                                      (1, 4, 8, 10, 5, 1), (7, 8, 8, 4, 2, 1)])
def test_streamable_conv_matches_torch(T, Cin, Cout, K, stride, dil):
    """StreamableConv1d padding rule (SpeechTokenizerEncoder.swift:114-118, 163-186) + conv vs torch.conv1d."""
    import torch
    from oracle import oracle as O
    rng = np.random.default_rng(T + K)
    x = rng.standard_normal((T, Cin)).astype(np.float32)
    W = rng.standard_normal((Cout, K, Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    eff = (K - 1) * dil + 1
    ptotal = eff - stride
    nframes = max(T + ptotal - eff, 0) / stride + 1
    extra = max(0, (int(np.ceil(nframes)) - 1) * stride + eff - ptotal - T)
    y = O.OracleModel._conv_f32(x, W, b, stride, dil, ptotal, extra, 0)
    xt = torch.nn.functional.pad(torch.from_numpy(x.T)[None], (ptotal, extra))
    yt = torch.nn.functional.conv1d(xt, torch.from_numpy(W.transpose(0, 2, 1).copy()), torch.from_numpy(b), stride=stride,
                                    dilation=dil)[0].T.numpy()
    assert y.shape == yt.shape == (int(np.ceil(T / stride)), Cout)
    np.testing.assert_allclose(y, yt, rtol=1e-4, atol=1e-4)


def test_reflect_conv_matches_torch():
    """TimeDelayNetBlock (SpeakerEncoder.swift:45-70): reflect pad + dilated conv."""
    import torch
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    x = rng.standard_normal((20, 8)).astype(np.float32)
    W = rng.standard_normal((8, 3, 8)).astype(np.float32)
    y = O.OracleModel._conv_f32(x, W, None, 1, 4, 4, 4, 1)
    xt = torch.nn.functional.pad(torch.from_numpy(x.T)[None], (4, 4), mode="reflect")
    yt = torch.nn.functional.conv1d(xt, torch.from_numpy(W.transpose(0, 2, 1).copy()), dilation=4)[0].T.numpy()
    np.testing.assert_allclose(y, yt, rtol=1e-4, atol=1e-4)


def test_mel_matches_torch_stft():
    """melSpectrogram (SpeakerEncoder.swift:410-456) vs torch.stft with the same window / padding and the oracle's
    filterbank; and the filterbank's structural properties (triangles on integer bin edges, 128 columns)."""
    import torch
    from oracle import oracle as O
    a = ref_audio(1, 0.5)
    mel = O.OracleModel.mel_spectrogram(a)
    n = np.arange(1024, dtype=np.float64)
    win = torch.from_numpy(0.5 * (1 - np.cos(2 * np.pi * n / 1023)))
    padded = torch.nn.functional.pad(torch.from_numpy(a.astype(np.float64)), (512, 512))
    st = torch.stft(padded, 1024, 256, 1024, window=win, center=False, return_complex=True)  # [513][T]
    power = (st.abs() ** 2).T.numpy()
    fb = np.empty((513, 128), np.float32)
    O.lib().o_mel_filterbank(1024, 128, 24000, O.C.c_float(0.0), O.C.c_float(12000.0), O._pf(fb))
    ref = np.log(np.maximum(power @ fb.astype(np.float64), 1e-10))
    assert mel.shape == ref.shape == (a.size // 256 + 1, 128)
    np.testing.assert_allclose(mel, ref, rtol=0, atol=2e-3)
    assert (fb >= 0).all() and fb.max() <= 1.0 and (fb.sum(0) > 0).sum() >= 120


def test_encoder_sanitiser_layouts(base_dir, oracle_base):
    """The encoder half of sanitizeSpeechTokenizerWeights (Qwen3.swift:1592-1747): key names and [out][k][in] layouts."""
    cw = oracle_base.codec
    ec = oracle_base.ec
    nf = ec["num_filters"]
    assert cw["encoder.encoder.init_conv1d.conv.conv.weight"].shape == (nf, 7, 1)
    assert cw["encoder.encoder.layers.0.residuals.0.block.0.conv.conv.weight"].shape == (nf // 2, 3, nf)
    assert cw["encoder.encoder.layers.0.residuals.0.block.1.conv.conv.weight"].shape == (nf, 1, nf // 2)
    assert cw["encoder.encoder.layers.0.downsample.conv.conv.weight"].shape == (2 * nf, 8, nf)  # ratios reversed: 4 first
    assert cw["encoder.encoder.layers.3.downsample.conv.conv.weight"].shape == (16 * nf, 16, 8 * nf)
    assert cw["encoder.downsample.conv.conv.conv.weight"].shape == (ec["hidden_size"], 4, ec["hidden_size"])
    assert cw["encoder.quantizer.rvq_first.input_proj.weight"].shape == (ec["codebook_dim"], 1, ec["hidden_size"])
    assert "encoder.encoder_transformer.transformer.layers.0.gating.linear1.weight" in cw
    assert "encoder.encoder_transformer.transformer.layers.1.layer_scale_2.scale" in cw
    assert cw["encoder.quantizer.rvq_rest.vq.layers.30.codebook.embeddingSum"].shape == (ec["codebook_size"], ec["codebook_dim"])
    assert not any(k.endswith(".initialized") for k in cw)
    # speaker encoder convs of the main checkpoint: [out][k][in]
    sc = oracle_base.sc
    assert oracle_base.w["speaker_encoder.blocks.0.conv.weight"].shape == (sc["enc_channels"][0], 5, 128)
    assert oracle_base.w["speaker_encoder.fc.weight"].shape == (sc["enc_dim"], 1, 2 * sc["enc_channels"][4])
    assert oracle_base.supports_voice_cloning and oracle_base.has_encoder


def test_oracle_encode_shapes_and_determinism(oracle_base):
    a = ref_audio(0, 1.0)
    st = {}
    codes = oracle_base.codec_encode(a, st)
    T = int(np.ceil(np.ceil(np.ceil(np.ceil(np.ceil(a.size / 4) / 5) / 6) / 8) / 2))
    assert codes.shape == (16, T) and codes.dtype == np.int32
    assert codes.min() >= 0 and codes.max() < oracle_base.ec["codebook_size"]
    assert (codes == oracle_base.codec_encode(a)).all()
    assert len(np.unique(codes)) > 8  # the waveform drives the codes
    # causality: a later change of the waveform leaves earlier frames untouched
    b = a.copy()
    b[-1920:] += 0.3
    cb = oracle_base.codec_encode(b)
    assert (cb[:, : T - 2] == codes[:, : T - 2]).all()
    # all 32 layers give the same first 16 rows
    assert (oracle_base.codec_encode(a, all_layers=True) == codes).all()


def test_oracle_speaker_embedding_properties(oracle_base):
    a = ref_audio(0, 1.0)
    e = oracle_base.speaker_embedding(a)
    assert e.shape == (oracle_base.sc["enc_dim"],) and np.isfinite(e).all() and np.abs(e).max() > 0
    e2 = oracle_base.speaker_embedding(ref_audio(3, 1.0))
    assert np.abs(e - e2).max() > 1e-6


def test_oracle_icl_prompt_layout(oracle_base):
    """prepareICLGenerationInputs (Qwen3.swift:418-582): row count and a few rows recomputed by hand."""
    from oracle import oracle as O
    pr = clone_prompt(0)
    a = ref_audio(0, 1.0)
    req = O.Request(text_ids=pr["text_ids"], target_token_count=pr["target_token_count"], language="english",
                    ref_audio=a, ref_text_ids=pr["ref_text_ids"])
    inp, trailing, pad, ref_codes = oracle_base.prepare_icl_generation_inputs(req)
    T = ref_codes.shape[1]
    n_ref, n_txt = len(pr["ref_text_ids"]) - 5, len(pr["text_ids"]) - 8
    prefix = 4 + 1 + 2  # think, think_bos, lang, think_eos | x-vector | pad, bos
    assert inp.shape[0] == 3 + (prefix - 1) + (n_ref + n_txt + 1) + (T + 1)
    assert trailing.shape[0] == 1 and (trailing == pad).all()
    t = oracle_base.t
    # the row after the role + prefix block is the first reference-text token overlaid with codec_pad
    first_text = oracle_base.add(oracle_base.text_projection(oracle_base.embed_text([pr["ref_text_ids"][3]])),
                                 oracle_base.codec_embed([t["codec_pad_id"]]))
    assert (inp[3 + prefix - 1] == first_text[0]).all()
    # the x-vector row: tts_pad + bf16(speaker embedding)
    spk = O.f32_to_bf16(oracle_base.speaker_embedding(a))[None]
    assert (inp[3 + 4] == oracle_base.add(pad, spk)[0]).all()
    # last row: tts_pad + sum of the 16 embeddings of the last reference frame
    ce = oracle_base.codec_embed([ref_codes[0, -1]])
    for i in range(15):
        ce = oracle_base.add(ce, oracle_base.cp_embed(i, [ref_codes[i + 1, -1]]))
    assert (inp[-1] == oracle_base.add(ce, pad)[0]).all()


def test_oracle_voice_clone_end_to_end(oracle_base):
    from oracle import oracle as O
    pr = clone_prompt(1, n_text=4)
    a = ref_audio(1, 0.5)
    req = O.Request(text_ids=pr["text_ids"], target_token_count=pr["target_token_count"], ref_audio=a,
                    ref_text_ids=pr["ref_text_ids"])
    s = O.Sampling(temperature=0.9, top_k=50, repetition_penalty=1.5, seed=11, force_frames=5)
    pcm, tr, ref_codes = oracle_base.generate_voice_clone(req, s)
    total = ref_codes.shape[1] + 5
    full = total * 1920
    cut = int(np.float32(ref_codes.shape[1]) / np.float32(total) * np.float32(full))
    assert tr.codes.shape == (5, 16)
    assert pcm.shape[0] in (full - cut, ) or pcm.shape[0] < full  # valid-length trim may shorten it further
    assert np.isfinite(pcm).all() and np.abs(pcm).max() <= 1.0


def test_abi_rejects_clone_without_encoder_symbols():
    """The C ABI exports the voice-clone entry points (no compute without a GPU)."""
    import ctypes as C
    from qwen3tts import _lib
    L = C.CDLL(_lib.LIB_PATH)
    for name in ("q3tts_codec_encode", "q3tts_codec_encoded_frames", "q3tts_speaker_embedding", "q3tts_debug_frontend_stage"):
        assert hasattr(L, name)
    assert C.sizeof(_lib.Request) == 88   # (every struct against the header: tests/test_abi.py)


# ---------------------------------------------------------------------------------------------------
# GPU: HIP front end vs the oracle
# ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def engine_base(base_dir):
    from qwen3tts import Qwen3TTSModel
    m = Qwen3TTSModel.from_pretrained(base_dir, max_batch=4, max_frames=96, max_prompt=160)
    yield m
    m.close()


def close(a, b, tol=2e-4):
    scale = max(1e-6, float(np.abs(b).max()))
    return float(np.abs(a - b).max()) <= tol * scale


@pytest.mark.gpu
def test_gpu_model_reports_voice_cloning(engine_base):
    assert engine_base.supports_voice_cloning
    assert engine_base.info.has_voice_cloning == 1
    assert engine_base.info.speaker_embedding_dim == engine_base.info.hidden_size


@pytest.mark.gpu
@pytest.mark.parametrize("seconds", [1.0, 0.437])
def test_gpu_encoder_stages(engine_base, oracle_base, seconds):
    a = ref_audio(0, seconds)
    st = {}
    oracle_base.codec_encode(a, st)
    for name in ("init_conv", "layer0", "layer1", "layer2", "layer3", "seanet", "transformer", "downsample",
                 "rvq_first_in", "rvq_rest_in"):
        got = engine_base.debug_frontend_stage(a, name)
        assert got.shape == st[name].shape, name
        assert close(got, st[name]), (name, float(np.abs(got - st[name]).max()), float(np.abs(st[name]).max()))


@pytest.mark.gpu
@pytest.mark.parametrize("row,seconds", [(0, 1.0), (1, 3.0), (2, 0.2503), (3, 0.081)])
def test_gpu_codec_encode_codes(engine_base, oracle_base, row, seconds):
    """Codes are bit-exact; a mismatch is only accepted at an oracle near-tie, and masks the later layers of that frame."""
    a = ref_audio(row, seconds)
    st = {}
    want = oracle_base.codec_encode(a, st)
    got = engine_base.codec_encode(a)
    assert got.shape == want.shape
    gaps = st["gaps"]
    alive = np.ones(want.shape[1], bool)
    for layer in range(16):
        if layer == 1:
            alive[:] = True  # the acoustic chain restarts from its own projection (SpeechTokenizerEncoder.swift:934-941)
        diff = (got[layer] != want[layer]) & alive
        assert (gaps[layer][diff] < 1e-4).all(), (layer, gaps[layer][diff])
        alive &= ~diff
    assert (got == want).mean() > 0.98


@pytest.mark.gpu
def test_gpu_speaker_encoder_stages(engine_base, oracle_base):
    a = ref_audio(0, 1.0)
    st = {}
    want = oracle_base.speaker_embedding(a, st)
    for name, tol in (("mel", 2e-4), ("h0", 2e-4), ("h1", 5e-4), ("h2", 5e-4), ("h3", 5e-4), ("mfa", 5e-4)):
        got = engine_base.debug_frontend_stage(a, name)
        assert got.shape == st[name].shape, name
        assert close(got, st[name], tol), (name, float(np.abs(got - st[name]).max()), float(np.abs(st[name]).max()))
    got = engine_base.debug_frontend_stage(a, "pooled")[0]
    assert close(got, st["pooled"], 1e-3)
    emb = engine_base.extract_speaker_embedding(a)
    assert close(emb, want, 1e-3)


@pytest.mark.gpu
def test_gpu_speaker_embedding_rejects_other_rates(engine_base):
    from qwen3tts import Qwen3TTSError
    with pytest.raises(Qwen3TTSError) as e:
        engine_base.extract_speaker_embedding(ref_audio(0, 0.5), sample_rate=16000)
    assert "Only 24kHz audio is supported" in str(e.value)


def _reqs(row, n_text=10, seconds=1.0, language="english"):
    from oracle import oracle as O
    from qwen3tts import GenerationRequest
    pr = clone_prompt(row, n_text)
    a = ref_audio(row, seconds)
    g = GenerationRequest(pr["text_ids"], pr["target_token_count"], None, None, language, ref_audio=a,
                          ref_text_ids=pr["ref_text_ids"])
    o = O.Request(text_ids=pr["text_ids"], target_token_count=pr["target_token_count"], language=language, ref_audio=a,
                  ref_text_ids=pr["ref_text_ids"])
    return g, o


@pytest.mark.gpu
@pytest.mark.parametrize("language", ["english", "auto"])
def test_gpu_icl_prompt(engine_base, oracle_base, language):
    g, o = _reqs(0, language=language)
    want, wtr, wpad, ref_codes = oracle_base.prepare_icl_generation_inputs(o)
    if not (engine_base.codec_encode(o.ref_audio) == ref_codes).all():
        pytest.skip("reference codes differ at a near-tie; covered by test_gpu_codec_encode_codes")
    got, gtr, gpad = engine_base.debug_prepare_inputs(g)
    assert got.shape == want.shape and gtr.shape == wtr.shape
    assert (gpad == wpad[0]).all() and (gtr == wtr).all()
    a, b = bf16_to_f32(got), bf16_to_f32(want)
    # text rows and embedding sums are bit-exact except for 1-ulp GEMM differences; the x-vector row carries the
    # speaker encoder's fp32 tolerance into one bf16 rounding
    assert (np.abs(a - b) <= 2 * ULP * np.maximum(np.abs(b), 2.0 ** -9)).all()
    assert (got != want).mean() < 0.02


@pytest.mark.gpu
def test_gpu_voice_clone_forced_frames(engine_base, oracle_base):
    """generateVoiceClone end to end with a fixed frame count: codes via greedy consistency, PCM vs the oracle decode of
    the engine's own codes, reference part removed proportionally."""
    from oracle import oracle as O
    g, o = _reqs(1, n_text=6, seconds=0.5)
    res = engine_base.generate_batch([g], temperature=0.0, repetition_penalty=1.5, seed=3, force_frames=6)[0]
    assert res.status == 0 and res.codes.shape == (6, 16)
    ref_codes = engine_base.codec_encode(o.ref_audio)
    full = np.concatenate([ref_codes.T, res.codes], 0)
    pcm, valid = oracle_base.codec_decode(full)
    if 0 < valid < pcm.shape[0]:
        pcm = pcm[:valid]
    cut = int(np.float32(ref_codes.shape[1]) / np.float32(full.shape[0]) * np.float32(pcm.shape[0]))
    if 0 < cut < pcm.shape[0]:
        pcm = pcm[cut:]
    assert res.audio.shape == pcm.shape
    assert np.abs(res.audio - pcm).max() < 2e-4
    # teacher-forced oracle pass over the engine's codes: every engine token is the oracle's argmax within the LM margin
    s = O.Sampling(temperature=0.0, repetition_penalty=1.5, seed=3, force_frames=6)
    tr = oracle_base.generate_codes(o, s, forced_codes=res.codes, keep_logits=True)
    if (tr.ref_codes == ref_codes).all():
        seen = np.zeros(oracle_base.V, bool)
        for f in range(6):
            lg = bf16_to_f32(tr.talker_logits[f]).copy()
            lg[oracle_base.V - 1024:] = -np.inf
            pen = np.where(lg < 0, lg * 1.5, lg / 1.5)
            lg = np.where(seen, pen, lg)
            tok = int(res.codes[f, 0])
            assert lg[tok] >= lg.max() - 4 * ULP * max(1.0, abs(lg.max())), (f, tok, int(lg.argmax()))
            seen[tok] = True


@pytest.mark.gpu
def test_gpu_voice_clone_batch_matches_single(engine_base):
    """Rows are independent: a mixed batch (two clone rows with different reference lengths + a preset-speaker row)
    reproduces each row's batch-1 result."""
    from qwen3tts import GenerationRequest
    from conftest import tiny_request
    g0, _ = _reqs(0, n_text=8, seconds=1.0)
    g1, _ = _reqs(2, n_text=5, seconds=0.4)
    r = tiny_request(row=3)
    g2 = GenerationRequest(r["text_ids"], r["target_token_count"], None, r["speaker"], r["language"])
    kw = dict(temperature=0.9, top_k=20, repetition_penalty=1.5, seed=5, force_frames=5)
    batch = engine_base.generate_batch([g0, g1, g2], **kw)
    tm = engine_base.last_timing()
    assert tm.frontend_ms > 0
    for i, g in enumerate((g0, g1, g2)):
        # a row's RNG stream is its global row index: run it alone at the same index by padding with copies
        alone = engine_base.generate_batch([g0, g1, g2][: i] + [g], **kw)[i]
        assert (alone.codes == batch[i].codes).all()
        assert alone.audio.shape == batch[i].audio.shape and np.abs(alone.audio - batch[i].audio).max() < 1e-5


@pytest.mark.gpu
def test_gpu_clone_errors(engine_base, ckpt_dirs):
    from qwen3tts import GenerationRequest, Qwen3TTSError, Qwen3TTSModel
    pr = clone_prompt(0)
    with pytest.raises(Qwen3TTSError) as e:  # missing reference text
        engine_base.generate_batch([GenerationRequest(pr["text_ids"], 10, ref_audio=ref_audio(0, 0.3))], force_frames=2)
    assert e.value.status == 3
    m = Qwen3TTSModel.from_pretrained(ckpt_dirs["tiny-a"], max_batch=1, max_frames=16, max_prompt=64)
    try:
        assert not m.supports_voice_cloning
        with pytest.raises(Qwen3TTSError) as e:  # no encoder in this checkpoint (Qwen3.swift:1033-1038)
            m.generate_batch([GenerationRequest(pr["text_ids"], 10, ref_audio=ref_audio(0, 0.3), ref_text_ids=pr["ref_text_ids"])])
        assert e.value.status == 1 and "speech tokenizer encoder" in str(e.value)
        with pytest.raises(Qwen3TTSError):
            m.codec_encode(ref_audio(0, 0.3))
    finally:
        m.close()


# ---------------------------------------------------------------------------------------------------
# GPU: the real front-end shapes (64..1024-channel SEANet, 8x512 transformer, 2048x256 codebooks, 512/1536-wide ECAPA)
# ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def fullenc(tmp_path_factory):
    from oracle import oracle as O
    from qwen3tts import Qwen3TTSModel, synth
    d = str(tmp_path_factory.mktemp("tiny_base_fullenc"))
    synth.write_checkpoint(d, "tiny-base-fullenc", seed=77)
    m = Qwen3TTSModel.from_pretrained(d, max_batch=2, max_frames=64, max_prompt=192)
    yield m, O.OracleModel(d)
    m.close()


@pytest.mark.gpu
def test_gpu_full_size_front_end(fullenc):
    """BASELINE config 4's reference clip (3.0 s, 72 000 samples) through the full-size encoders."""
    m, o = fullenc
    a = ref_audio(0, 3.0)
    st = {}
    want = o.codec_encode(a, st)
    assert want.shape == (16, 38)
    for name in ("init_conv", "layer0", "layer2", "seanet", "transformer", "downsample", "rvq_rest_in"):
        got = m.debug_frontend_stage(a, name)
        assert got.shape == st[name].shape, name
        assert close(got, st[name], 3e-4), (name, float(np.abs(got - st[name]).max()), float(np.abs(st[name]).max()))
    got = m.codec_encode(a)
    gaps = st["gaps"]
    alive = np.ones(want.shape[1], bool)
    for layer in range(16):
        if layer == 1:
            alive[:] = True
        diff = (got[layer] != want[layer]) & alive
        assert (gaps[layer][diff] < 1e-4).all(), (layer, gaps[layer][diff])
        alive &= ~diff
    assert (got == want).mean() > 0.95
    st = {}
    emb = o.speaker_embedding(a, st)
    for name in ("mel", "h0", "h3", "mfa"):
        g = m.debug_frontend_stage(a, name)
        assert g.shape == st[name].shape and close(g, st[name], 5e-4), name
    assert close(m.extract_speaker_embedding(a), emb, 1e-3)
    assert m.last_timing().frontend_ms > 0


@pytest.mark.gpu
def test_gpu_full_size_clone_generation(fullenc):
    """Clone rows with the full-size front end: batch == singles, reference part removed."""
    from qwen3tts import GenerationRequest
    m, _ = fullenc
    reqs = []
    for row, sec in ((0, 3.0), (1, 1.37)):
        pr = clone_prompt(row, 8)
        reqs.append(GenerationRequest(pr["text_ids"], pr["target_token_count"], None, None, "english",
                                      ref_audio=ref_audio(row, sec), ref_text_ids=pr["ref_text_ids"]))
    kw = dict(temperature=0.9, top_k=50, repetition_penalty=1.5, seed=9, force_frames=4)
    both = m.generate_batch(reqs, **kw)
    first = m.generate_batch(reqs[:1], **kw)[0]
    assert (both[0].codes == first.codes).all() and np.abs(both[0].audio - first.audio).max() < 1e-5
    for r, sec in zip(both, (3.0, 1.37)):
        assert r.status == 0 and r.codes.shape == (4, 16)
        ref_T = int(np.ceil(sec * 24000 / 1920))
        assert abs(r.audio.shape[0] - 4 * 1920) <= 1920 and r.audio.shape[0] <= (ref_T + 4) * 1920


# ---------------------------------------------------------------------------------------------------
# committed golden vectors (tests/golden/tiny_base.npz, written by tests/golden/make_golden.py)
# ---------------------------------------------------------------------------------------------------
def _golden():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny_base.npz"))


def test_oracle_reproduces_voice_clone_golden(oracle_base):
    from oracle import oracle as O
    g = _golden()
    a = ref_audio(0, 1.0)
    st = {}
    assert (oracle_base.codec_encode(a, st) == g["ref_codes"]).all()
    np.testing.assert_allclose(st["transformer"], g["transformer"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(oracle_base.speaker_embedding(a), g["xvec"], rtol=0, atol=1e-6)
    req = O.Request(text_ids=g["text_ids"].tolist(), target_token_count=10, language="english", ref_audio=a,
                    ref_text_ids=g["ref_text_ids"].tolist())
    pcm, tr, _ = oracle_base.generate_voice_clone(req, O.Sampling(temperature=0.0, repetition_penalty=1.5, force_frames=5))
    assert (tr.codes == g["clone_codes"]).all()
    np.testing.assert_allclose(pcm, g["clone_pcm"], rtol=0, atol=1e-6)


@pytest.mark.gpu
def test_gpu_matches_voice_clone_golden(engine_base):
    from qwen3tts import GenerationRequest
    g = _golden()
    a = ref_audio(0, 1.0)
    got = engine_base.codec_encode(a)
    want, gaps = g["ref_codes"], g["rvq_gaps"]
    alive = np.ones(want.shape[1], bool)
    for layer in range(16):
        if layer == 1:
            alive[:] = True
        diff = (got[layer] != want[layer]) & alive
        assert (gaps[layer][diff] < 1e-4).all()
        alive &= ~diff
    for name in ("seanet", "transformer", "downsample", "mel"):
        assert close(engine_base.debug_frontend_stage(a, name), g[name]), name
    assert close(engine_base.extract_speaker_embedding(a), g["xvec"], 1e-3)
    if (got == want).all():
        req = GenerationRequest(g["text_ids"].tolist(), 10, None, None, "english", ref_audio=a,
                                ref_text_ids=g["ref_text_ids"].tolist())
        ie, _, pad = engine_base.debug_prepare_inputs(req)
        assert ie.shape == g["input_embeds"].shape and (pad == g["tts_pad"][0]).all()
        assert (ie != g["input_embeds"]).mean() < 0.02
        res = engine_base.generate_batch([req], temperature=0.0, repetition_penalty=1.5, force_frames=5)[0]
        # greedy codes can only differ where two logits are within the LM margin; on this fixture they do not
        assert (res.codes[:, 0] == g["clone_codes"][:, 0]).mean() >= 0.8
        if (res.codes == g["clone_codes"]).all():
            assert res.audio.shape == g["clone_pcm"].shape and np.abs(res.audio - g["clone_pcm"]).max() < 2e-4


@pytest.mark.gpu
def test_gpu_clone_audio_chunks_skip_the_reference_part(engine_base):
    """Chunked delivery (q3tts_sampling.audio_chunk_frames) on voice-clone rows: the decoder runs over [reference ++
    generated] frames (Qwen3.swift:1176-1186) and the reference's share of the samples is cut proportionally (:1195-1199);
    the chunks must carry exactly the remaining audio, in order, and a mixed batch must work."""
    from qwen3tts import GenerationRequest
    reqs = []
    for row, sec in ((0, 0.9), (1, 0.4)):
        pr = clone_prompt(row, 7)
        reqs.append(GenerationRequest(pr["text_ids"], pr["target_token_count"], None, None, "english",
                                      ref_audio=ref_audio(row, sec), ref_text_ids=pr["ref_text_ids"]))
    kw = dict(temperature=0.9, top_k=50, repetition_penalty=1.5, seed=4, force_frames=14)
    want = engine_base.generate_batch(reqs, **kw)
    pieces = {0: [], 1: []}
    got = engine_base.generate_batch(reqs, audio_chunk_frames=6,
                                     on_event=lambda i, k, p: pieces[i].append(p) if k == "audio_chunk" else None, **kw)
    for i, (a, b) in enumerate(zip(got, want)):
        assert a.status == 0 and (a.codes == b.codes).all() and (a.audio == b.audio).all()
        assert pieces[i][0][0] == 0 and (np.concatenate([p for _, p in pieces[i]]) == b.audio).all()
        assert b.audio.size < (14 + 12) * 1920   # the reference's frames are not part of the result


@pytest.mark.gpu
def test_gpu_reference_audio_with_a_nan_sample_is_rejected(engine_base):
    """One NaN in the clip would spread through both encoders into every logit of the row: an 'Invalid input' instead."""
    from qwen3tts import Qwen3TTSError
    g, _ = _reqs(0, n_text=6, seconds=0.5)
    bad = np.array(g.ref_audio, np.float32, copy=True)
    bad[1234] = np.nan
    g.ref_audio = bad
    with pytest.raises(Qwen3TTSError) as e:
        engine_base.generate_batch([g], temperature=0.0, force_frames=3)
    assert e.value.status == 3 and "non-finite" in str(e.value)
    g.ref_audio[1234] = np.inf
    with pytest.raises(Qwen3TTSError) as e:
        engine_base.generate_batch([g], temperature=0.0, force_frames=3)
    assert e.value.status == 3 and "non-finite" in str(e.value)
