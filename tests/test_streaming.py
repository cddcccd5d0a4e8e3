"""Row f1 of SURVEY 8f: the codec decode in pieces. The pre-transformer (bidirectional, SpeechTokenizer.swift:763) runs
once over the generated frames; everything behind it is causal (:298-301, :346-351, :767-781), so the tail is evaluated
chunk by chunk behind a left context of CodecRunner::tail_context_frames() frames, and AUDIO_CHUNK events deliver the
waveform while later chunks are still being computed. The bar is exact equality with the one-shot decode."""
import numpy as np
import pytest

from conftest import tiny_request

pytestmark = pytest.mark.gpu


def _req(**kw):
    from qwen3tts import GenerationRequest
    r = tiny_request(**kw)
    return GenerationRequest(r["text_ids"], r["target_token_count"], r["instruct_ids"], r["speaker"], r["language"])


@pytest.mark.parametrize("chunk", [1, 7, 16, 1000])
def test_chunked_tail_pcm_equals_one_shot_bit_for_bit(ckpt_dirs, chunk):
    from qwen3tts import Qwen3TTSModel
    m = Qwen3TTSModel.from_pretrained(ckpt_dirs["tiny-b"], max_batch=4, max_frames=96, max_prompt=96)
    try:
        reqs = [_req(row=i, n_text=5 + 3 * i) for i in range(3)]
        kw = dict(temperature=0.9, top_k=40, repetition_penalty=1.05, seed=11, force_frames=45)
        want = m.generate_batch(reqs, **kw)
        pieces = {i: [] for i in range(3)}
        order = []

        def on_event(i, kind, payload):
            order.append((i, kind))
            if kind == "audio_chunk":
                pieces[i].append(payload)

        got = m.generate_batch(reqs, on_event=on_event, audio_chunk_frames=chunk, **kw)
        for i, (a, b) in enumerate(zip(got, want)):
            assert (a.codes == b.codes).all() and a.audio.shape == b.audio.shape and (a.audio == b.audio).all()
            offs = [o for o, _ in pieces[i]]
            assert offs == sorted(offs) and offs[0] == 0
            assert (np.concatenate([p for _, p in pieces[i]]) == b.audio).all()          # the pieces ARE the audio
            assert all(o == sum(p.size for _, p in pieces[i][:k]) for k, (o, _) in enumerate(pieces[i]))
            assert len(pieces[i]) == -(-45 // min(chunk, 45))
            kinds = [k for j, k in order if j == i]
            assert kinds == ["token"] * 45 + ["audio_chunk"] * len(pieces[i]) + ["info", "audio"]
    finally:
        m.close()


def test_chunked_tail_ragged_rows_and_eos(ckpt_dirs):
    """Rows that stop at different frames (max_tokens caps): a chunk only carries the samples a row still has."""
    from qwen3tts import GenerationRequest, Qwen3TTSModel
    m = Qwen3TTSModel.from_pretrained(ckpt_dirs["tiny-a"], max_batch=4, max_frames=96, max_prompt=96)
    try:
        reqs = []
        for i, cap in enumerate((9, 33, 20)):
            r = tiny_request(row=i, n_text=6)
            reqs.append(GenerationRequest(r["text_ids"], 1, None, "aiden", "english", max_tokens=cap))
        kw = dict(temperature=0.0)
        want = m.generate_batch(reqs, **kw)
        pieces = {i: [] for i in range(3)}
        got = m.generate_batch(reqs, audio_chunk_frames=8, on_event=lambda i, k, p: pieces[i].append(p) if k == "audio_chunk" else None, **kw)
        for i, (a, b) in enumerate(zip(got, want)):
            assert a.status == b.status == 0 and (a.codes == b.codes).all() and (a.audio == b.audio).all()
            assert (np.concatenate([p for _, p in pieces[i]]) == b.audio).all()
    finally:
        m.close()


def test_chunked_tail_at_the_real_layer_widths(tmp_path_factory):
    """Full-size codec decoder (1024-wide transformer, 1536 -> 96 channel stack, fused residual units in the last block):
    chunked == one-shot bit for bit, and the left context the engine uses is what the layer shapes require."""
    import json
    import os
    from qwen3tts import Qwen3TTSModel, synth
    d = str(tmp_path_factory.mktemp("full_codec_stream"))
    p = synth.preset("tiny-a")
    p["speech_tokenizer"]["decoder_config"] = synth._codec_cfg(False)
    p["config"]["talker_config"]["code_predictor_config"]["vocab_size"] = 2048
    os.makedirs(os.path.join(d, "speech_tokenizer"), exist_ok=True)
    g = synth._Gen(1234, False)
    json.dump(p["config"], open(os.path.join(d, "config.json"), "w"))
    json.dump(p["speech_tokenizer"], open(os.path.join(d, "speech_tokenizer", "config.json"), "w"))
    synth.save_safetensors(os.path.join(d, "model.safetensors"), synth.talker_tensors(p["config"], g))
    synth.save_safetensors(os.path.join(d, "speech_tokenizer", "model.safetensors"),
                           synth.codec_tensors(p["speech_tokenizer"]["decoder_config"], g, out_wstd=synth.FULL_WIDTH_OUT_WSTD))
    m = Qwen3TTSModel.from_pretrained(d, max_batch=2, max_frames=64, max_prompt=64)
    try:
        reqs = [_req(row=i, n_text=5 + i) for i in range(2)]
        kw = dict(temperature=0.9, top_k=50, seed=3, force_frames=40)
        want = m.generate_batch(reqs, **kw)
        got = m.generate_batch(reqs, audio_chunk_frames=12, on_event=lambda *a: None, **kw)
        for a, b in zip(got, want):
            assert (a.codes == b.codes).all() and (a.audio == b.audio).all()
            assert np.abs(b.audio).max() > 1e-3
    finally:
        m.close()
