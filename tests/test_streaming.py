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


# ---------------------------------------------------------------------------------------------------------
# streamed decode: audio while tokens are still being generated (q3tts_sampling.audio_window_frames > 0)
# ---------------------------------------------------------------------------------------------------------
def _full_codec_dir(tmp_path_factory, name):
    import json
    import os
    from qwen3tts import synth
    d = str(tmp_path_factory.mktemp(name))
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
    return d


@pytest.fixture(scope="module")
def full_codec_model(tmp_path_factory):
    from qwen3tts import Qwen3TTSModel
    d = _full_codec_dir(tmp_path_factory, "full_codec_streamed")
    m = Qwen3TTSModel.from_pretrained(d, max_batch=4, max_frames=96, max_prompt=64)
    m.ckpt_dir = d  # (the oracle loads the same files)
    yield m
    m.close()


@pytest.mark.parametrize("chunk", [3, 7, 16, 200])
def test_carried_state_tail_is_bit_identical(ckpt_dirs, full_codec_model, chunk):
    """The causal tail with its conv state carried from chunk to chunk (no left context recomputed): with the pre-transformer
    run once over all frames (window < 0) the streamed decode IS the one-shot decode, bit for bit -- tiny and real layer
    widths (fused residual units in the last block, two-launch units in the others), ragged rows."""
    from qwen3tts import Qwen3TTSModel
    rng = np.random.default_rng(chunk)
    for m, own in ((Qwen3TTSModel.from_pretrained(ckpt_dirs["tiny-b"], max_batch=4, max_frames=96, max_prompt=64), True),
                   (full_codec_model, False)):
        try:
            F = [37, 5, 22]
            vc = m.info.cp_vocab_size
            codes = np.zeros((3, 37, 16), np.int32)
            for b, f in enumerate(F):
                codes[b, :f] = rng.integers(1, min(vc, 2048), size=(f, 16))
            want, lens = m.codec_decode(codes, n_frames=F)
            got = m.codec_decode_streamed(codes, chunk, -1, n_frames=F)
            for b, f in enumerate(F):
                assert lens[b] == f * 1920
                assert (got[b, :f * 1920] == want[b, :f * 1920]).all(), (chunk, b)
        finally:
            if own:
                m.close()


def test_a_failed_stream_open_leaves_the_runner_usable(ckpt_dirs):
    """stream_open sizes its arena in a counting pass during which the conv launcher launches nothing; an allocation that
    fails behind that pass (here: a chunk of fifty million frames) must not leave the runner counting -- every later decode of
    the model would then skip its convolutions and return finite garbage with status OK."""
    from qwen3tts import Qwen3TTSError, Qwen3TTSModel
    m = Qwen3TTSModel.from_pretrained(ckpt_dirs["tiny-b"], max_batch=4, max_frames=96, max_prompt=64)
    try:
        codes = np.random.default_rng(4).integers(1, 32, size=(2, 12, 16)).astype(np.int32)
        want, _ = m.codec_decode(codes)
        with pytest.raises(Qwen3TTSError):
            m.codec_decode_streamed(codes, 50_000_000, 4, 0)
        again, _ = m.codec_decode(codes)
        assert np.abs(want).max() > 1e-4 and (again == want).all()
        assert (m.codec_decode_streamed(codes, 4, -1) == want).all()   # and streams still open
    finally:
        m.close()


def test_windowed_stream_matches_the_oracles_windowed_decode(ckpt_dirs, full_codec_model):
    """The windowed mode against a second implementation: OracleModel.codec_decode_streamed composes the reference's own
    functions the way a stream can run them (pre_transformer over [f0 - window, f1 + lookahead) per chunk, causal tail over the
    concatenated latents). The bar is the one-shot decode's: PCM within 1e-4 absolute on every sample -- tiny and real layer
    widths, ragged rows, chunk sizes that do and do not divide the rows' lengths."""
    from oracle import oracle as O
    from qwen3tts import Qwen3TTSModel
    tiny = Qwen3TTSModel.from_pretrained(ckpt_dirs["tiny-b"], max_batch=4, max_frames=96, max_prompt=64)
    tiny.ckpt_dir = ckpt_dirs["tiny-b"]
    try:
        for m, F, geoms in ((tiny, [37, 5, 22], ((8, 16, 2), (5, 4, 0), (7, 3, 9), (16, 64, 64))),
                            (full_codec_model, [19, 8], ((8, 16, 2), (5, 4, 0)))):
            om = O.OracleModel(m.ckpt_dir)
            rng = np.random.default_rng(len(F))
            codes = np.zeros((len(F), max(F), 16), np.int32)
            for b, f in enumerate(F):
                codes[b, :f] = rng.integers(1, min(m.info.cp_vocab_size, 2048), size=(f, 16))
            for C_, W, L_ in geoms:
                got = m.codec_decode_streamed(codes, C_, W, L_, n_frames=F)
                for b, f in enumerate(F):
                    want = om.codec_decode_streamed(codes[b, :f], C_, W, L_)
                    err = float(np.abs(got[b, :f * 1920] - want).max())
                    assert want.size == f * 1920 and err <= 1e-4, (C_, W, L_, b, err)
    finally:
        tiny.close()


def test_sliding_window_distance_from_the_one_shot_decode(full_codec_model):
    """What a stream gives up: the pre-transformer is bidirectional over the whole utterance (SpeechTokenizer.swift:763); a
    chunk decoded while later tokens do not exist yet sees `window` frames to the left and `lookahead` to the right. The
    distance from the one-shot decode is measured at the real layer widths over a (window, lookahead) grid and bounded.
    (Synthetic weights: LayerScale 0.01 as in the shipped initialisation, so the attention's share of the residual stream --
    and with it this distance -- is what the checkpoint writer makes it; the numbers are a property of this checkpoint.)"""
    m = full_codec_model
    rng = np.random.default_rng(21)
    F = 72
    codes = rng.integers(1, 2048, size=(2, F, 16)).astype(np.int32)
    want, _ = m.codec_decode(codes)
    rms = float(np.sqrt(np.mean(want ** 2)))
    assert np.abs(want).max() < 0.999 and rms > 1e-3
    table = {}
    for W, L in ((4, 0), (16, 0), (16, 4), (32, 4), (64, 8), (F, F)):
        got = m.codec_decode_streamed(codes, 8, W, L)
        err = np.abs(got - want)
        table[(W, L)] = (float(err.max()), float(np.sqrt(np.mean(err ** 2))))
    print("window, lookahead -> max |pcm - one-shot|, rms (signal rms %.3f):" % rms)
    for k, v in table.items():
        print("  W=%3d L=%3d   max %.3e   rms %.3e" % (k + v))
    assert table[(F, F)][0] <= 1e-6                 # a window that covers everything: the one-shot decode again
    # The distances themselves are REPORTED, not promised: they are a property of the checkpoint (include/q3tts.h). What the
    # arithmetic guarantees is pinned against the oracle in test_windowed_stream_matches_the_oracles_windowed_decode.
    assert table[(64, 8)][1] <= table[(4, 0)][1] + 1e-9


def test_audio_leaves_before_the_last_token(ckpt_dirs):
    """generate with audio_window_frames > 0: AUDIO_CHUNK events arrive while TOKEN events are still coming, the pieces
    concatenate to the final audio, and that audio is exactly the streamed decode of the final codes."""
    from qwen3tts import Qwen3TTSModel
    m = Qwen3TTSModel.from_pretrained(ckpt_dirs["tiny-b"], max_batch=4, max_frames=96, max_prompt=96)
    try:
        reqs = [_req(row=i, n_text=5 + 3 * i) for i in range(3)]
        kw = dict(temperature=0.9, top_k=40, repetition_penalty=1.05, seed=11, force_frames=64)
        base = m.generate_batch(reqs, **kw)
        order, pieces = [], {i: [] for i in range(3)}

        def on_event(i, kind, payload):
            order.append((i, kind))
            if kind == "audio_chunk":
                pieces[i].append(payload)

        got = m.generate_batch(reqs, on_event=on_event, audio_chunk_frames=8, audio_window_frames=16, audio_lookahead_frames=2, **kw)
        tm = m.last_timing()
        codes = np.stack([r.codes for r in got])
        ref = m.codec_decode_streamed(codes, 8, 16, 2)
        for i, (a, b) in enumerate(zip(got, base)):
            assert (a.codes == b.codes).all() and a.audio.size == b.audio.size == 64 * 1920
            kinds = [k for j, k in order if j == i]
            first_chunk, last_token = kinds.index("audio_chunk"), len(kinds) - 1 - kinds[::-1].index("token")
            assert first_chunk < last_token, "no audio left before the last token"
            assert kinds[-2:] == ["info", "audio"] and kinds.count("audio_chunk") == 8
            offs = [o for o, _ in pieces[i]]
            assert offs == [k * 8 * 1920 for k in range(8)]
            cat = np.concatenate([p for _, p in pieces[i]])
            assert (cat == a.audio).all() and (a.audio == ref[i]).all()
        assert 0 < tm.first_audio_ms < tm.prefill_ms + tm.decode_ms, (tm.first_audio_ms, tm.decode_ms)
    finally:
        m.close()


def test_row_groups_when_the_scratch_is_short(ckpt_dirs):
    """A batch whose activations exceed the decoder's scratch budget goes through in groups of rows -- one-shot and chunked
    alike -- instead of failing after the AR loop has already produced the codes; the samples do not change."""
    from qwen3tts import Qwen3TTSModel, _lib
    m = Qwen3TTSModel.from_pretrained(ckpt_dirs["tiny-b"], max_batch=4, max_frames=96, max_prompt=96)
    try:
        reqs = [_req(row=i, n_text=5 + i) for i in range(4)]
        kw = dict(temperature=0.9, top_k=40, seed=5, force_frames=30)
        want = m.generate_batch(reqs, **kw)
        codes = np.stack([r.codes for r in want])
        _lib.lib().q3tts_debug_set_codec_scratch(1 << 20)   # far below one row's activations: one row per group
        try:
            got = m.generate_batch(reqs, audio_chunk_frames=8, on_event=lambda *a: None, **kw)
            pcm, _ = m.codec_decode(codes)
        finally:
            _lib.lib().q3tts_debug_set_codec_scratch(0)
        for i in range(4):
            assert (got[i].codes == want[i].codes).all() and (got[i].audio == want[i].audio).all()
            assert (pcm[i] == want[i].audio).all()
    finally:
        m.close()


def test_streamed_rows_are_independent_and_ragged(ckpt_dirs):
    """Rows that stop at different frames while the stream is running: every row's streamed audio equals the streamed decode of
    ITS OWN final codes alone (a chunk's window only ever covers frames of the same row), rows that are already final do not
    hold the others back, and a streamed job can be followed by an ordinary pipelined one on the same handle."""
    from qwen3tts import GenerationRequest, Qwen3TTSModel
    m = Qwen3TTSModel.from_pretrained(ckpt_dirs["tiny-a"], max_batch=4, max_frames=96, max_prompt=96)
    try:
        reqs = []
        for i, cap in enumerate((9, 41, 20)):
            r = tiny_request(row=i, n_text=6)
            reqs.append(GenerationRequest(r["text_ids"], 1, None, "aiden", "english", max_tokens=cap))
        pieces = {i: [] for i in range(3)}
        got = m.generate_batch(reqs, temperature=0.0, audio_chunk_frames=8, audio_window_frames=16, audio_lookahead_frames=2,
                               on_event=lambda i, k, p: pieces[i].append(p) if k == "audio_chunk" else None)
        base = m.generate_batch(reqs, temperature=0.0)
        for i, (a, b) in enumerate(zip(got, base)):
            assert a.status == b.status == 0 and (a.codes == b.codes).all() and a.audio.size == b.audio.size
            F = a.codes.shape[0]
            alone = m.codec_decode_streamed(a.codes[None], 8, 16, 2)[0]
            assert (a.audio == alone[:a.audio.size]).all(), i
            cat = np.concatenate([p for _, p in pieces[i]])
            assert (cat == a.audio).all() and len(pieces[i]) == -(-F // 8)
        # an ordinary two-deep pipeline right behind it
        j1 = m.generate_batch_begin(reqs, temperature=0.0)
        j2 = m.generate_batch_begin(reqs, temperature=0.0, more_follows=False)
        r1, r2 = m.generate_batch_end(j1), m.generate_batch_end(j2)
        for x, y, z in zip(r1, r2, base):
            assert (x.audio == z.audio).all() and (y.audio == z.audio).all()
    finally:
        m.close()
