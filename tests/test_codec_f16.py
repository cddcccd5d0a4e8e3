"""Float16 speech tokenizers (the "lite" checkpoints store every speech-tokenizer tensor in float16, docs/paper.tex:207).
The reference then computes the decoder in float16 -- MLX evaluates every op in the arrays' dtype -- and the engine follows it
from initConv on (96 % of the decoder's arithmetic): float16 tensors in HBM, one matrix-core product per block, one float16
rounding per op of the reference (csrc/kernels/codec_conv_h1.hip). Oracle: OracleModel.codec_decode(..., f16=True).

Tolerance. Every tensor here is a chain of float16 roundings (2^-11 relative each). Two implementations that round at the SAME
points still differ where a value sits on a rounding boundary and their fp32 sums (or their sines) differ in the last fp32 bit:
one float16 ulp there. Behind a 4480-term contraction such flips do not die out, they breed: a fraction f of flipped inputs moves
every output by ~0.3 sqrt(f) of ITS ulp, which flips ~0.3 sqrt(f) of the outputs -- a map whose fixed point (f ~ 0.09) does not
depend on where it started. Measured: initConv agrees on 99.7 % of its elements (r.m.s. 0.02 ulp of the stage scale), one block
later the distance is 0.34 ulp r.m.s. and it then sits at 0.4-0.9 ulp whatever the stage -- exactly the distance between the
float16 oracle and the fp32 oracle (printed beside it), i.e. the noise floor of float16 arithmetic itself, which the reference's
own MLX kernels sit at too; at the real layer widths (K up to 7 x 1536) the floor is 3-4 ulp r.m.s. after four blocks. The bar
follows the full-size logit tests: the engine must stay within 1.25x (r.m.s.) / 1.5x (max) of the oracle's OWN floor -- the
distance between its float16 and fp32 readings of the same checkpoint -- with absolute caps (r.m.s. 6 ulp, max 40 ulp of the
stage scale; PCM 2e-2 absolute, 4e-3 r.m.s.) so that a regression in the oracle cannot widen it silently; and the first stage,
where nothing has bred yet, within 2 ulp max / 0.1 r.m.s. (0.02 tiny, 0.05 at the real widths)."""
import os

import numpy as np
import pytest

from conftest import tiny_request

pytestmark = pytest.mark.gpu
H_ULP = 2.0 ** -11


@pytest.fixture(scope="module")
def tiny_h(tmp_path_factory):
    from qwen3tts import synth
    d = str(tmp_path_factory.mktemp("tiny_h"))
    synth.write_checkpoint(d, "tiny-h", seed=1234)
    return d


def _compare(m, om, codes, stages):
    so16, so32 = {}, {}
    want, _ = om.codec_decode(codes, so16, f16=True)
    ref32, _ = om.codec_decode(codes, so32)
    for st in stages:
        got = m.debug_codec_stage(codes, st)
        w = so16[st]
        assert got.shape == w.shape, (st, got.shape, w.shape)
        scale = float(np.abs(w).max())
        err = np.abs(got - w) / scale / H_ULP
        own = np.abs(w - so32[st]) / scale / H_ULP
        print("%-10s scale %8.3g | engine vs float16 oracle max %.2f rms %.3f ulp | float16 oracle vs fp32 oracle max %.1f rms %.2f ulp"
              % (st, scale, err.max(), np.sqrt((err ** 2).mean()), own.max(), np.sqrt((own ** 2).mean())))
        e32 = np.abs(got - so32[st]) / scale / H_ULP
        print("%-10s engine vs fp32 oracle max %.2f rms %.3f ulp" % ("", e32.max(), np.sqrt((e32 ** 2).mean())))
        if st == "init_conv":
            assert err.max() <= 2.0 and np.sqrt((err ** 2).mean()) <= 0.1, (st, float(err.max()))
        rms, own_rms = float(np.sqrt((err ** 2).mean())), float(np.sqrt((own ** 2).mean()))
        assert rms <= 1.25 * own_rms + 0.1 and err.max() <= 1.5 * own.max() + 2.0, (st, rms, own_rms, float(err.max()), float(own.max()))
        assert rms <= 6.0 and err.max() <= 40.0 and own_rms <= 6.0, (st, rms, float(err.max()))
    got, lens = m.codec_decode(codes[None])
    err = np.abs(got[0] - want)
    print("pcm        engine vs float16 oracle max %.2e rms %.2e | float16 oracle vs fp32 oracle max %.2e (signal rms %.3f)"
          % (err.max(), np.sqrt((err ** 2).mean()), np.abs(want - ref32).max(), np.sqrt((want ** 2).mean())))
    assert lens[0] == codes.shape[0] * 1920
    floor = np.abs(want - ref32)
    assert err.max() <= 1.5 * floor.max() + 1e-3 and np.sqrt((err ** 2).mean()) <= 1.25 * np.sqrt((floor ** 2).mean()) + 1e-4
    assert err.max() <= 2e-2 and np.sqrt((err ** 2).mean()) <= 4e-3
    return got[0], want, ref32


def test_float16_main_decoder_matches_the_float16_oracle(tiny_h):
    from oracle import oracle as O
    from qwen3tts import Qwen3TTSModel
    om = O.OracleModel(tiny_h)
    assert om.codec_f16
    codes = np.random.default_rng(0).integers(1, 32, size=(9, 16)).astype(np.int32)
    m = Qwen3TTSModel.from_pretrained(tiny_h, max_batch=4, max_frames=32, max_prompt=64)
    try:
        # what stays fp32 (front end, ConvNeXt stages) keeps the fp32 bar
        s32 = {}
        om.codec_decode(codes, s32)
        for st in ("pre_transformer", "upsample1"):
            got = m.debug_codec_stage(codes, st)
            assert np.abs(got - s32[st]).max() <= 1e-4 * np.abs(s32[st]).max(), st
        _compare(m, om, codes, ("init_conv", "block0", "block1", "block2", "block3"))
        # ragged batch, rows independent, deterministic
        F = [9, 4, 7]
        batch = np.zeros((3, 9, 16), np.int32)
        rng = np.random.default_rng(1)
        for b, f in enumerate(F):
            batch[b, :f] = rng.integers(1, 32, size=(f, 16))
        a, _ = m.codec_decode(batch, n_frames=F)
        b2, _ = m.codec_decode(batch, n_frames=F)
        assert (a == b2).all()
        for i, f in enumerate(F):
            alone, _ = m.codec_decode(batch[i:i + 1, :f])
            assert (alone[0] == a[i, : f * 1920]).all(), i
        # end to end: generate -> the same decode of the generated codes
        r = tiny_request(row=0, n_text=7)
        from qwen3tts import GenerationRequest
        res = m.generate_batch([GenerationRequest(r["text_ids"], r["target_token_count"], None, "aiden", "english")], temperature=0.9,
                               top_k=40, seed=5, force_frames=8)[0]
        dec, _ = m.codec_decode(res.codes[None])
        assert res.status == 0 and (dec[0] == res.audio).all()
    finally:
        m.close()


def test_float16_checkpoint_through_the_upcast_path_keeps_the_fp32_bar(tiny_h, monkeypatch):
    """Q3TTS_CODEC_NO_F16=1 (and the fp32 re-decode of rows that leave the float16 range) take a float16 checkpoint through the
    two-plane fp32-equivalent kernels: PCM within 1e-4 of the fp32 oracle, as for any checkpoint."""
    from oracle import oracle as O
    from qwen3tts import Qwen3TTSModel
    monkeypatch.setenv("Q3TTS_CODEC_NO_F16", "1")
    om = O.OracleModel(tiny_h)
    codes = np.random.default_rng(2).integers(1, 32, size=(6, 16)).astype(np.int32)
    m = Qwen3TTSModel.from_pretrained(tiny_h, max_batch=2, max_frames=32, max_prompt=64)
    try:
        got, _ = m.codec_decode(codes[None])
        want, _ = om.codec_decode(codes)
        assert np.abs(got[0] - want).max() <= 1e-4
    finally:
        m.close()


def test_float16_main_decoder_at_the_real_layer_widths(tmp_path_factory):
    """The shipped decoder geometry (1536 -> 768 -> 384 -> 192 -> 96 channels, x1920) stored in float16."""
    import json
    from oracle import oracle as O
    from qwen3tts import Qwen3TTSModel, synth
    d = str(tmp_path_factory.mktemp("full_codec_f16"))
    p = synth.preset("tiny-a")
    p["speech_tokenizer"]["decoder_config"] = synth._codec_cfg(False)
    p["config"]["talker_config"]["code_predictor_config"]["vocab_size"] = 2048
    os.makedirs(os.path.join(d, "speech_tokenizer"), exist_ok=True)
    g = synth._Gen(1234, False)
    json.dump(p["config"], open(os.path.join(d, "config.json"), "w"))
    json.dump(p["speech_tokenizer"], open(os.path.join(d, "speech_tokenizer", "config.json"), "w"))
    synth.save_safetensors(os.path.join(d, "model.safetensors"), synth.talker_tensors(p["config"], g))
    codec = synth.codec_tensors(p["speech_tokenizer"]["decoder_config"], g, out_wstd=synth.FULL_WIDTH_OUT_WSTD)
    codec = {k: (("F16", v.astype(np.float16)) if tag == "F32" else (tag, v)) for k, (tag, v) in codec.items()}
    synth.save_safetensors(os.path.join(d, "speech_tokenizer", "model.safetensors"), codec)
    om = O.OracleModel(d)
    assert om.codec_f16
    codes = np.random.default_rng(3).integers(1, 2048, size=(5, 16)).astype(np.int32)
    m = Qwen3TTSModel.from_pretrained(d, max_batch=2, max_frames=16, max_prompt=64)
    try:
        got, want, ref32 = _compare(m, om, codes, ("init_conv", "block0", "block2", "block3"))  # blocks 2, 3: the fused residual units
        assert np.abs(want).max() < 0.999 and np.sqrt((want ** 2).mean()) > 1e-3
    finally:
        m.close()


def _full_width_f16_checkpoint(d):
    import json
    from qwen3tts import synth
    p = synth.preset("tiny-a")
    p["speech_tokenizer"]["decoder_config"] = synth._codec_cfg(False)
    p["config"]["talker_config"]["code_predictor_config"]["vocab_size"] = 2048
    os.makedirs(os.path.join(d, "speech_tokenizer"), exist_ok=True)
    g = synth._Gen(1234, False)
    json.dump(p["config"], open(os.path.join(d, "config.json"), "w"))
    json.dump(p["speech_tokenizer"], open(os.path.join(d, "speech_tokenizer", "config.json"), "w"))
    synth.save_safetensors(os.path.join(d, "model.safetensors"), synth.talker_tensors(p["config"], g))
    codec = synth.codec_tensors(p["speech_tokenizer"]["decoder_config"], g, out_wstd=synth.FULL_WIDTH_OUT_WSTD)
    codec = {k: (("F16", v.astype(np.float16)) if tag == "F32" else (tag, v)) for k, (tag, v) in codec.items()}
    synth.save_safetensors(os.path.join(d, "speech_tokenizer", "model.safetensors"), codec)


@pytest.mark.parametrize("width", ["tiny", "real"])
def test_float16_streamed_decode_is_the_float16_decode(tiny_h, tmp_path_factory, width):
    """Audio that leaves while tokens are still being generated, from a float16 speech tokenizer: the stream's tail runs the same
    float16 kernels with the conv state carried in the tensors' history margins (CodecRunner::run_main_h1_stream), so
      * with the pre-transformer over all frames (window < 0) the streamed waveform IS the one-shot float16 decode, bit for bit,
        ragged rows included -- the property the fp32-equivalent stream has had since round 3;
      * with a sliding window it is the oracle's windowed restatement with the float16 tail (codec_decode_streamed(f16=True))
        within the float16 noise floor of test_float16_main_decoder_matches_the_float16_oracle;
      * and a streamed generate call delivers chunks that concatenate to the call's own audio."""
    from oracle import oracle as O
    from qwen3tts import GenerationRequest, Qwen3TTSModel
    if width == "tiny":
        d, hi = tiny_h, 32
    else:
        d, hi = str(tmp_path_factory.mktemp("full_codec_f16_stream")), 2048
        _full_width_f16_checkpoint(d)
    om = O.OracleModel(d)
    assert om.codec_f16
    rng = np.random.default_rng(8)
    F = [13, 7, 10]
    codes = np.zeros((3, 13, 16), np.int32)
    for b, f in enumerate(F):
        codes[b, :f] = rng.integers(1, hi, size=(f, 16))
    m = Qwen3TTSModel.from_pretrained(d, max_batch=4, max_frames=32, max_prompt=64)
    try:
        one_shot, _ = m.codec_decode(codes, n_frames=F)
        for chunk in (4, 5):
            streamed = m.codec_decode_streamed(codes, chunk, -1, n_frames=F)
            for b, f in enumerate(F):
                assert (streamed[b, : f * 1920] == one_shot[b, : f * 1920]).all(), (chunk, b)
        # a sliding window: against the oracle's windowed decode with the float16 tail, bar = the one-shot float16 bar
        chunk, window, look = 4, 4, 2
        got = m.codec_decode_streamed(codes, chunk, window, look, n_frames=F)
        for b, f in enumerate(F):
            want = om.codec_decode_streamed(codes[b, :f], chunk, window, look, f16=True)
            ref32 = om.codec_decode_streamed(codes[b, :f], chunk, window, look)
            err, floor = np.abs(got[b, : f * 1920] - want), np.abs(want - ref32)
            print("row %d windowed: engine vs float16 oracle max %.2e rms %.2e | float16 vs fp32 oracle max %.2e rms %.2e"
                  % (b, err.max(), np.sqrt((err ** 2).mean()), floor.max(), np.sqrt((floor ** 2).mean())))
            assert err.max() <= 1.5 * floor.max() + 1e-3 and np.sqrt((err ** 2).mean()) <= 1.25 * np.sqrt((floor ** 2).mean()) + 1e-4
            assert err.max() <= 2e-2 and np.sqrt((err ** 2).mean()) <= 4e-3
        if width == "tiny":
            # streamed generate: AUDIO_CHUNK events while the frame loop runs, exact mode (window < 0 is not a generate option; the
            # default window) -- the chunks concatenate to what the call returns
            r = tiny_request(row=1, n_text=7)
            chunks = []
            res = m.generate_batch([GenerationRequest(r["text_ids"], r["target_token_count"], None, "aiden", "english")], temperature=0.9,
                                   top_k=40, seed=6, force_frames=12, audio_chunk_frames=4, audio_window_frames=8,
                                   on_event=lambda i, kind, payload: chunks.append(payload) if kind == "audio_chunk" else None)[0]
            assert res.status == 0 and chunks
            cat = np.concatenate([c[1] for c in sorted(chunks, key=lambda c: c[0])])
            assert cat.shape == res.audio.shape and (cat == res.audio).all()
    finally:
        m.close()
