"""The decoder's default convolutions split every fp32 operand into two fp16 planes (csrc/kernels/codec_conv.hip): same
accuracy as the fp32 matrix cores, but an activation beyond 65504 cannot be represented. The reference computes in fp32 and
decodes such a checkpoint, so the engine must too: a row whose waveform comes out non-finite is decoded again on the fp32
matrix cores (Engine::redo_rows_fp32 / Engine::codec_decode) and handed out with status OK -- the same samples the
codec_fp32 load option gives -- in every delivery mode: one-shot, in pieces after the last token, streamed."""
import os

import numpy as np
import pytest

from conftest import tiny_request


def _big_bias_checkpoint(tmp_path):
    from safetensors.numpy import load_file, save_file
    from qwen3tts import synth
    d = str(tmp_path / "m")
    synth.write_checkpoint(d, "tiny-a", seed=1234)
    f = os.path.join(d, "speech_tokenizer", "model.safetensors")
    t = load_file(f)
    key = [k for k in t if k.endswith("decoder.0.conv.bias") or k.endswith("initConv.conv.bias")]
    assert key, sorted(t)[:40]
    t[key[0]] = (t[key[0]].astype(np.float32) + np.float32(3.0e5)).astype(t[key[0]].dtype)   # far beyond fp16's 65504
    save_file(t, f)
    return d


@pytest.mark.gpu
def test_out_of_range_activations_fall_back_to_the_fp32_matrix_cores(tmp_path):
    from qwen3tts import GenerationRequest, Qwen3TTSModel
    d = _big_bias_checkpoint(tmp_path)
    codes = np.random.default_rng(0).integers(1, 32, size=(2, 6, 16)).astype(np.int32)
    r = tiny_request(row=0, n_text=6)
    reqs = [GenerationRequest(r["text_ids"], 1, None, "aiden", "english")] * 2
    kw = dict(temperature=0.0, force_frames=24)
    m = Qwen3TTSModel.from_pretrained(d, max_batch=2, max_frames=32, max_prompt=64, codec_fp32=True)
    try:
        want, lens = m.codec_decode(codes)
        assert np.isfinite(want).all() and (lens == 6 * 1920).all()
        gen_want = m.generate_batch(reqs, **kw)
        assert all(x.status == 0 and np.isfinite(x.audio).all() and np.abs(x.audio).max() > 0 for x in gen_want)
    finally:
        m.close()
    m = Qwen3TTSModel.from_pretrained(d, max_batch=2, max_frames=32, max_prompt=64)   # the default (fp16 two-plane) kernels
    try:
        got, lens = m.codec_decode(codes)                     # q3tts_codec_decode: the whole call again in fp32
        assert (lens == 6 * 1920).all() and (got == want).all()
        for mode in (dict(), dict(audio_chunk_frames=8), dict(audio_chunk_frames=8, audio_window_frames=64, audio_lookahead_frames=64)):
            pieces = {0: [], 1: []}
            kinds = {0: [], 1: []}

            def on_event(i, kind, payload):
                kinds[i].append(kind)
                if kind == "audio_chunk":
                    pieces[i].append(payload)

            out = m.generate_batch(reqs, on_event=on_event, **kw, **mode)
            for i, (a, b) in enumerate(zip(out, gen_want)):
                assert a.status == 0 and (a.codes == b.codes).all(), mode
                assert np.isfinite(a.audio).all() and a.audio.shape == b.audio.shape
                if "audio_window_frames" not in mode:           # exact modes: the fp32 kernels' samples
                    assert (a.audio == b.audio).all(), mode
                else:                                           # (a window over everything: the one-shot arithmetic again)
                    assert np.abs(a.audio - b.audio).max() <= 1e-5, mode
                if mode:
                    offs = [o for o, _ in pieces[i]]
                    assert offs == sorted(offs) and offs[0] == 0 and len(pieces[i]) == 3
                    cat = np.concatenate([p for _, p in pieces[i]])
                    assert np.isfinite(cat).all() and (cat == a.audio).all(), mode   # nothing non-finite ever left in a chunk
                assert kinds[i][-2:] == ["info", "audio"]
    finally:
        m.close()


@pytest.mark.gpu
def test_fp32_fallback_of_voice_clone_rows_cut_and_chunked(tmp_path):
    """The re-decode works in decoded-stream coordinates: a voice-clone row's audio starts `cut` samples into its decode
    (the reference's share, Qwen3.swift:1195-1199), and a chunked delivery's pieces are clipped against that cut. A Base
    checkpoint whose decoder overflows the fp16 range, clone rows with different reference lengths next to a preset-speaker row,
    delivered in pieces: every row equals what the codec_fp32 load gives, and the pieces concatenate to it."""
    from safetensors.numpy import load_file, save_file
    from qwen3tts import GenerationRequest, Qwen3TTSModel, synth
    d = str(tmp_path / "b")
    synth.write_checkpoint(d, "tiny-base", seed=4321)
    f = os.path.join(d, "speech_tokenizer", "model.safetensors")
    t = load_file(f)
    key = [k for k in t if k.endswith("decoder.0.conv.bias") or k.endswith("initConv.conv.bias")]
    assert key
    t[key[0]] = (t[key[0]].astype(np.float32) + np.float32(3.0e5)).astype(t[key[0]].dtype)
    save_file(t, f)

    def reqs():
        out = []
        for row, secs in ((0, 1.0), (2, 0.4)):
            p = synth.synthetic_prompt(row, n_text=6, text_vocab=1000, im_start=1000, im_end=1001)
            out.append(GenerationRequest(p["text_ids"], p["target_token_count"], None, None, "english",
                                         ref_audio=synth.synthetic_reference_audio(row, secs), ref_text_ids=p["ref_text_ids"]))
        r = tiny_request(row=3)
        out.append(GenerationRequest(r["text_ids"], r["target_token_count"], None, r["speaker"], r["language"]))
        return out

    kw = dict(temperature=0.9, top_k=20, repetition_penalty=1.5, seed=5, force_frames=12)
    m = Qwen3TTSModel.from_pretrained(d, max_batch=3, max_frames=32, max_prompt=128, codec_fp32=True)
    try:
        want = m.generate_batch(reqs(), **kw)
        assert all(w.status == 0 and np.isfinite(w.audio).all() for w in want)
    finally:
        m.close()
    m = Qwen3TTSModel.from_pretrained(d, max_batch=3, max_frames=32, max_prompt=128)
    try:
        for mode in (dict(), dict(audio_chunk_frames=5)):
            pieces = {0: [], 1: [], 2: []}
            got = m.generate_batch(reqs(), on_event=lambda i, k, p: pieces[i].append(p) if k == "audio_chunk" else None, **kw, **mode)
            for i, (a, b) in enumerate(zip(got, want)):
                assert a.status == 0 and (a.codes == b.codes).all() and a.audio.shape == b.audio.shape, (mode, i)
                assert (a.audio == b.audio).all(), (mode, i)
                if mode:
                    offs = [o for o, _ in pieces[i]]
                    assert offs == sorted(offs) and offs[0] == 0
                    assert (np.concatenate([p for _, p in pieces[i]]) == a.audio).all(), (mode, i)
    finally:
        m.close()
