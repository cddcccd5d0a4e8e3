"""BASELINE.json's full-size configurations on the GPU against the oracle.

Every config of BASELINE.json is loaded at its real dimensions and batch size and teacher-forced through the HIP engine
(q3tts_debug_generate_forced: same kernels and launch geometry as generate(), eager launches) while the oracle runs the
same request at batch 1 on the host (about 2 s per 1.7B frame on 8 cores, seconds in total): talker and code-predictor
logits must agree within 2 bf16 ulps of the row's largest |logit|, and what the engine's sampler picks must be the
oracle's argmax up to that margin. On top of that, properties that hold for any weights: determinism, row independence
across scheduling modes, hipGraph replay == eager launches, the codec decoder's length rule. Synthetic weights (there
are no checkpoints offline): reference Talker.swift:532-574, CodePredictor.swift:320-339 at real dims."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


ULP = 2.0 ** -7


def _ckpt(preset):
    """Synthetic checkpoint of a full-size preset, shared with bench.py (same directory, same completeness marker:
    bench.ensure_checkpoint writes the marker last and checks the tensor files' sizes recorded in it)."""
    import bench
    return bench.ensure_checkpoint(preset, 0, None)


@pytest.fixture(scope="module")
def full_dir():
    return _ckpt("1.7b")


def _reqs(n):
    import bench
    return bench.build_requests("1.7b", 0, n, 32, 16)


def test_full_size_properties(full_dir, monkeypatch):
    from qwen3tts import Qwen3TTSModel
    F = 12
    kw = dict(temperature=0.9, top_k=50, top_p=1.0, repetition_penalty=1.05, seed=1234, force_frames=F)
    m = Qwen3TTSModel.from_pretrained(full_dir, max_batch=32, max_frames=F + 8, max_prompt=128)
    try:
        assert m.info.hidden_size == 2048 and m.info.num_layers == 28 and m.info.weight_bytes > 3.0e9
        reqs = _reqs(32)
        a = m.generate_batch(reqs, **kw)
        b = m.generate_batch(reqs, **kw)
        for x, y in zip(a, b):  # determinism
            assert x.status == 0 and x.codes.shape == (F, 16)
            assert (x.codes == y.codes).all() and (x.audio == y.audio).all()
        assert len({tuple(r.codes[:, 0]) for r in a}) > 16          # rows differ (own prompts, own RNG streams)
        # the frame step as a chain: 28 talker layers x 5 launches, the head (+ riders), sampler, projection, the predictor's pair
        # pass and 14 more passes of 5 layers x 5 launches + head + sampler (548 graph nodes)
        n_launch = m.last_timing().launches_per_frame_step
        print("launches per 1.7B frame step at batch 32:", n_launch)
        assert 540 <= n_launch <= 556
        # prefill chunks (512 rows per launch here) go through the tall GEMM (gemm_prefill.hip): the same bits as the skinny
        # kernel's wave partials at K = 2048 and 6144, eight phases
        from qwen3tts import _lib
        monkeypatch.setenv("Q3TTS_NO_TALL_GEMM", "1")
        _lib.reload_debug_env()  # (the switches are read once per model load)
        a2 = m.generate_batch(reqs, **kw)
        monkeypatch.delenv("Q3TTS_NO_TALL_GEMM")
        _lib.reload_debug_env()
        for x, y in zip(a, a2):
            assert (x.codes == y.codes).all() and (x.audio == y.audio).all()
        small = m.generate_batch(reqs[:3], **kw)                       # row independence across scheduling modes
        for x, y in zip(small, a[:3]):
            assert (x.codes == y.codes).all() and np.abs(x.audio - y.audio).max() < 1e-6
        for r in a[:4]:                                                # length rule + range of the PCM
            assert r.audio.shape[0] == F * 1920 and np.isfinite(r.audio).all() and np.abs(r.audio).max() <= 1.0
        tm = m.last_timing()
        assert tm.frame_steps == F and tm.rows == 3
        # the two-deep pipeline at the real sizes: the second batch's frame loop runs beside the first batch's decode on
        # the CU-masked stream, rows are staged by the background thread -- bit for bit the sequential results
        ja = m.generate_batch_begin(reqs, more_follows=True, **kw)
        jb = m.generate_batch_begin(reqs[:8], more_follows=False, **kw)
        pa, pb = m.generate_batch_end(ja), m.generate_batch_end(jb)
        for x, y in zip(pa, a):
            assert x.status == 0 and (x.codes == y.codes).all() and (x.audio == y.audio).all()
        for x, y in zip(pb, a[:8]):
            assert (x.codes == y.codes).all() and np.abs(x.audio - y.audio).max() < 1e-6
    finally:
        m.close()
    e = Qwen3TTSModel.from_pretrained(full_dir, max_batch=32, max_frames=F + 8, max_prompt=128, use_graph=False)
    try:
        c = e.generate_batch(_reqs(32)[:8], **kw)                      # eager launches == hipGraph replay
        for x, y in zip(c, a[:8]):
            assert (x.codes == y.codes).all() and np.abs(x.audio - y.audio).max() < 1e-6
    finally:
        e.close()


@pytest.fixture(scope="module")
def full_codec_dir(tmp_path_factory):
    """Tiny talker, FULL-SIZE codec decoder (1024-wide transformer, 1536 -> 96 channel conv stack)."""
    import json
    from qwen3tts import synth
    d = str(tmp_path_factory.mktemp("full_codec"))
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


def test_full_size_codec_both_contraction_paths_match_the_oracle(full_codec_dir, monkeypatch):
    """The codec decoder contracts on fp16 matrix cores with every fp32 operand split into two fp16 planes (three
    products per block, csrc/kernels/codec_conv.hip); Q3TTS_CODEC_FP32=1 (q3tts_load_opts.codec_fp32) selects the plain
    fp32 matrix-core kernel. Both must sit at fp32 rounding noise from the oracle's fmaf chains at the real layer widths,
    stage by stage."""
    from oracle import oracle as O
    from qwen3tts import Qwen3TTSModel
    om = O.OracleModel(full_codec_dir)
    codes = np.random.default_rng(3).integers(1, 2048, size=(3, 16)).astype(np.int32)
    st = {}
    pcm_o, _ = om.codec_decode(codes, st)
    stages = ("pre_transformer", "upsample1", "init_conv", "block0", "block1", "block2", "block3")
    worst = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("Q3TTS_CODEC_FP32", mode)
        m = Qwen3TTSModel.from_pretrained(full_codec_dir, max_batch=1, max_frames=8, max_prompt=64)
        try:
            for s in stages:
                a = m.debug_codec_stage(codes, s)
                assert a.shape == st[s].shape
                err = float(np.abs(a - st[s]).max() / max(1.0, np.abs(st[s]).max()))
                worst[mode] = max(worst.get(mode, 0.0), err)
                assert err <= 1e-4, (mode, s, err)      # tolerance of the codec stages (test_gpu_parity.py)
        finally:
            m.close()
    print("worst stage error / stage scale: fp16x2 %.2e, fp32 MFMA %.2e" % (worst["0"], worst["1"]))
    # measured: 4.3e-5 for both at block3 (the SnakeBeta chain amplifies fp32 rounding noise of ANY summation order)
    assert worst["0"] <= 1.5 * worst["1"] + 1e-6         # the split path is no noisier than the fp32 matrix cores


def test_full_size_codec_ragged_rows_match_the_oracle(full_codec_dir):
    """Two rows of different length through the real-width decoder (fused residual units in the 96-channel block, hoisted
    SnakeBeta elsewhere): each row must equal the oracle's decode of that row alone, and nothing may leak past a row's end."""
    from oracle import oracle as O
    from qwen3tts import Qwen3TTSModel
    om = O.OracleModel(full_codec_dir)
    rng = np.random.default_rng(9)
    F = [3, 2]
    codes = np.zeros((2, 3, 16), np.int32)
    for b, f in enumerate(F):
        codes[b, :f] = rng.integers(1, 2048, size=(f, 16))
    m = Qwen3TTSModel.from_pretrained(full_codec_dir, max_batch=2, max_frames=8, max_prompt=64)
    try:
        pcm, lens = m.codec_decode(codes, n_frames=F)
        for b, f in enumerate(F):
            ref, valid = om.codec_decode(codes[b, :f])
            assert lens[b] == valid == f * 1920
            # the synthetic tail conv is scaled so that the waveform stays inside the clip (synth.FULL_WIDTH_OUT_WSTD):
            # the north star's waveform tolerance, 1e-4 absolute, on EVERY sample at the real layer widths
            assert np.abs(ref).max() < 0.999
            assert np.abs(pcm[b, :f * 1920] - ref).max() <= 1e-4
            assert (pcm[b, f * 1920:] == 0).all()
    finally:
        m.close()


# ---------------------------------------------------------------------------------------------------------
# oracle parity at the real dimensions, one test per BASELINE config
# ---------------------------------------------------------------------------------------------------------
def _oracle_request(g, ref_codes=None):
    from oracle import oracle as O
    return O.Request(text_ids=list(g.text_ids), target_token_count=g.target_token_count,
                     instruct_ids=None if g.instruct_ids is None else list(g.instruct_ids), speaker=g.speaker,
                     language=g.language, ref_audio=g.ref_audio,
                     ref_text_ids=None if g.ref_text_ids is None else list(g.ref_text_ids), ref_codes_override=ref_codes)


def _forced_parity(m, om, reqs, rows, F, seed, rep=1.05, ref_codes=None):
    """Teacher-force `F` frames of random codes through the engine for ALL of `reqs` (the batch size decides the kernel
    instantiations) and through the oracle for the rows in `rows`; compare logits and the sampler's picks.

    Tolerance. At the tiny test sizes the logits sit within 2 bf16 ulps of the row's largest |logit|. At 28 layers x
    2048 that bar is no longer a property of the arithmetic but of luck: every Linear rounds an fp32 sum to bf16, two
    valid fp32 summation orders land on different sides of a rounding boundary now and then, and each such flip (0.4 %
    of one element) is carried through the remaining layers. MLX's own order is not visible from the Swift, so the
    oracle is run in TWO orders (index order and reversed, o_set_sum_order): their distance is the floor any faithful
    implementation sits at, and the engine must stay within 1.5x of it (or within the 2-ulp bar where the floor is
    lower), in the mean and at the maximum. A wrong epsilon, scale, position or weight would shift the whole
    distribution by far more than that."""
    from conftest import bf16_to_f32
    from oracle import oracle as O
    n = len(reqs)
    V, Vc = m.info.vocab_size, m.info.cp_vocab_size
    rng = np.random.default_rng(seed)
    # first codes below the suppressed range and away from EOS so that no row finishes early
    forced = np.concatenate([rng.integers(1, V - 1024, size=(n, F, 1)), rng.integers(0, Vc, size=(n, F, 15))], -1).astype(np.int32)
    tl, cl, sampled = m.debug_generate_forced(reqs, forced, temperature=0.0, repetition_penalty=rep)
    sampled_run = m.debug_generate_forced(reqs, forced, temperature=0.9, top_k=50, repetition_penalty=rep, seed=seed)
    assert (sampled_run[0] == tl).all() and (sampled_run[1] == cl).all()   # teacher forcing: the logits do not depend on the draws
    worst = 0.0
    for r in rows:
        oreq = _oracle_request(reqs[r], None if ref_codes is None else ref_codes[r])
        osp = O.Sampling(temperature=0.0, repetition_penalty=rep, force_frames=F)
        tr = om.generate_codes(oreq, osp, forced_codes=forced[r], keep_logits=True)
        O.lib().o_set_sum_order(1)
        try:
            tr2 = om.generate_codes(oreq, osp, forced_codes=forced[r], keep_logits=True)
        finally:
            O.lib().o_set_sum_order(0)
        for got, exp, exp2, what in ((tl[r], np.stack(tr.talker_logits), np.stack(tr2.talker_logits), "talker"),
                                     (cl[r], np.stack(tr.cp_logits), np.stack(tr2.cp_logits), "cp")):
            a, b, b2 = bf16_to_f32(got), bf16_to_f32(exp), bf16_to_f32(exp2)
            scale = np.abs(b).max(axis=-1, keepdims=True)
            err = np.minimum(np.abs(a - b), np.abs(a - b2)) / scale / ULP   # distance to the nearer of the two readings
            floor = np.abs(b - b2) / scale / ULP
            print("%s row %d: engine-oracle mean %.3f max %.2f ulp | oracle order floor mean %.3f max %.2f ulp"
                  % (what, r, err.mean(), err.max(), floor.mean(), floor.max()))
            worst = max(worst, float(err.max()))
            assert err.max() <= max(2.0, 1.5 * floor.max()), (what, r, float(err.max()), float(floor.max()))
            assert err.mean() <= max(0.25, 1.5 * floor.mean()), (what, r, float(err.mean()), float(floor.mean()))
            # ... and absolutely: the floor is the oracle's own property, so a regression THERE must not widen the bar silently
            # (measured over all five configs: engine mean <= 0.36 / max <= 3.7 ulp, floor mean <= 0.57 / max <= 4.2 ulp)
            assert floor.mean() <= 0.8 and floor.max() <= 6.5, ("oracle order floor grew", what, r, float(floor.mean()), float(floor.max()))
            assert err.mean() <= 0.5 and err.max() <= 6.0, (what, r, float(err.mean()), float(err.max()))
            # per element against ITS OWN floor: beyond 2 ulp + 1.5x the distance between the oracle's two readings of that
            # very element only a handful of rounding flips remain
            far = float((err > 2.0 + 1.5 * floor).mean())
            print("   elements beyond 2 ulp + 1.5 x own floor: %.2e" % far)
            assert far <= 1e-3, (what, r, far)   # measured <= 3.3e-4
        # sampled decoding as the bench runs it (T = 0.9, top-k 50): the oracle's sampler on the ENGINE's logits with the same
        # Philox key (seed, global row, frame * 16 + codebook) must draw the engine's token -- every codebook of every frame
        import ctypes as C
        tl_s, cl_s, sampled_s = sampled_run
        seen = np.zeros(V, np.uint8)
        for f in range(F):
            tok = O.lib().o_sample_token(O._p16(tl_s[r, f]), V, C.c_float(0.9), 50, C.c_float(1.0), C.c_float(rep),
                                         seen.ctypes.data_as(O.u8p), V - 1024, V, m.info.codec_eos_token_id, 0, C.c_uint64(seed), r, f * 16)
            assert tok == int(sampled_s[r, f, 0]), ("sampled talker token", r, f, tok, int(sampled_s[r, f, 0]))
            seen[forced[r, f, 0]] = 1   # generatedTokens holds what was fed back (the forced code)
            for i in range(15):
                tok = O.lib().o_sample_token(O._p16(cl_s[r, f, i]), Vc, C.c_float(0.9), 50, C.c_float(1.0), C.c_float(1.0), None, 0, 0,
                                             -1, 0, C.c_uint64(seed), r, f * 16 + 1 + i)
                assert tok == int(sampled_s[r, f, 1 + i]), ("sampled predictor token", r, f, i)
        # the engine's greedy picks from its own logits: the oracle's argmax wherever the oracle's margin exceeds the bar
        for f in range(F):
            b = bf16_to_f32(tr.cp_logits[f])
            for i in range(15):
                tok = int(sampled[r, f, 1 + i])
                assert b[i].max() - b[i][tok] <= 3 * ULP * np.abs(b[i]).max(), ("cp pick", r, f, i)
    return worst


def test_config2_1p7b_batch32_logits_match_the_oracle(full_dir):
    """BASELINE configs[2]: Qwen3-TTS-1.7B-VoiceDesign bf16, batch 32 (H = 2048 norm prologue, two-pair gate/up,
    512-thread attention, grid.y row splits, 2048 -> 1024 small_to_mtp_projection and the projected tables)."""
    from oracle import oracle as O
    from qwen3tts import Qwen3TTSModel
    m = Qwen3TTSModel.from_pretrained(full_dir, max_batch=32, max_frames=16, max_prompt=128)
    try:
        worst = _forced_parity(m, O.OracleModel(full_dir), _reqs(32), rows=(0, 21), F=4, seed=101)
        print("1.7B batch 32: worst logit error %.2f bf16 ulp of the row scale" % worst)
    finally:
        m.close()


def test_config2_1p7b_batch32_long_cache_matches_the_oracle(full_dir):
    """The bench regime of BASELINE configs[2]: the headline run decodes 200 frames behind a 48-position prompt, so its attention
    walks caches of up to 248 tokens = four KV pages through attn_decode_kernel<2, 512, true> (non-temporal, 32 lane groups,
    page lookups through the block table) -- a regime the 4-frame tests above never reach. Here the prompts are 194 positions
    long (180 instruct tokens), so the cache already spans four pages when the first frame is decoded and the chunked prefill
    crosses three page boundaries; one row per 16-row block edge (0, 15, 16, 31) against the oracle, same bars as above
    (Talker.swift:532-574)."""
    import bench
    from oracle import oracle as O
    from qwen3tts import Qwen3TTSModel
    m = Qwen3TTSModel.from_pretrained(full_dir, max_batch=32, max_frames=16, max_prompt=208)
    try:
        reqs = bench.build_requests("1.7b", 0, 32, 32, 180)
        om = O.OracleModel(full_dir)
        inp, _, _ = om.prepare_generation_inputs(_oracle_request(reqs[0]))
        assert inp.shape[0] > 3 * 64, inp.shape    # the prompt alone fills more than three pages of 64 tokens
        worst = _forced_parity(m, om, reqs, rows=(0, 15, 16, 31), F=4, seed=107)
        print("1.7B batch 32, %d-position prompts: worst logit error %.2f bf16 ulp of the row scale" % (inp.shape[0], worst))
    finally:
        m.close()


@pytest.fixture(scope="module")
def small_dir():
    return _ckpt("0.6b")


def test_config1_0p6b_batch8_logits_match_the_oracle(small_dir):
    """BASELINE configs[1]: Qwen3-TTS-0.6B-CustomVoice bf16, batch 8, speaker Aiden (H = 1024, no projection)."""
    import bench
    from oracle import oracle as O
    from qwen3tts import Qwen3TTSModel
    m = Qwen3TTSModel.from_pretrained(small_dir, max_batch=8, max_frames=16, max_prompt=128)
    try:
        assert m.info.hidden_size == 1024 and m.tts_model_type == "custom_voice"
        reqs = bench.build_requests("0.6b", 0, 8, 32, 0)
        worst = _forced_parity(m, O.OracleModel(small_dir), reqs, rows=(0, 5), F=4, seed=102)
        print("0.6B batch 8: worst logit error %.2f bf16 ulp of the row scale" % worst)
    finally:
        m.close()


def test_config0_0p6b_batch1_greedy_is_oracle_consistent(small_dir):
    """BASELINE configs[0] (its MLX-CPU leg cannot exist here): 0.6B-CustomVoice, speaker Aiden, batch 1,
    greedy, free running through the hipGraph path; the oracle is then teacher-forced on the engine's codes and every
    emitted token must be its argmax within the logit bar; PCM vs the oracle's decode of the same codes."""
    import bench
    from conftest import bf16_to_f32
    from oracle import oracle as O
    from qwen3tts import Qwen3TTSModel
    m = Qwen3TTSModel.from_pretrained(small_dir, max_batch=1, max_frames=16, max_prompt=128)
    try:
        om = O.OracleModel(small_dir)
        req = bench.build_requests("0.6b", 0, 1, 32, 0)[0]
        F = 6
        res = m.generate_batch([req], temperature=0.0, repetition_penalty=1.05, force_frames=F)[0]
        assert res.status == 0 and res.codes.shape == (F, 16)
        tr = om.generate_codes(_oracle_request(req), O.Sampling(temperature=0.0, repetition_penalty=1.05, force_frames=F),
                               forced_codes=res.codes, keep_logits=True)
        V = om.V
        seen = np.zeros(V, bool)
        pen = float(bf16_to_f32(O.f32_to_bf16(np.array([1.05], np.float32)))[0])  # the sampler works in bf16
        for f in range(F):
            lg = bf16_to_f32(tr.talker_logits[f]).astype(np.float64)
            lg[V - 1024:] = -np.inf
            lg = np.where(seen, np.where(lg < 0, lg * pen, lg / pen), lg)
            tok = int(res.codes[f, 0])
            fin = np.isfinite(lg)
            assert lg[fin].max() - lg[tok] <= 3 * ULP * np.abs(lg[fin]).max(), (f, tok, int(np.argmax(lg)))
            seen[tok] = True
            cpl = bf16_to_f32(tr.cp_logits[f])
            for i in range(15):
                assert cpl[i].max() - cpl[i][res.codes[f, 1 + i]] <= 2 * ULP * np.abs(cpl[i]).max(), (f, i)
        pcm, valid = om.codec_decode(res.codes)
        if 0 < valid < pcm.size:
            pcm = pcm[:valid]
        assert res.audio.shape == pcm.shape
        assert np.abs(pcm).max() < 0.999 and np.abs(res.audio - pcm).max() <= 1e-4  # waveform tolerance of the north star
    finally:
        m.close()


def test_config4_0p6b_int4_batch64_logits_match_the_oracle():
    """BASELINE configs[4], one GPU's share: 0.6B with int4-g64 Linears and the pruned text vocabulary (token map), 64
    rows per GPU (four row blocks per launch, dequant-in-register GEMM at the real tile shapes)."""
    import bench
    from oracle import oracle as O
    from qwen3tts import Qwen3TTSModel
    d = _ckpt("0.6b-q4")
    m = Qwen3TTSModel.from_pretrained(d, max_batch=64, max_frames=16, max_prompt=128)
    try:
        reqs = bench.build_requests("0.6b-q4", 0, 64, 32, 0)
        worst = _forced_parity(m, O.OracleModel(d), reqs, rows=(3, 50), F=3, seed=104)
        print("0.6B int4 batch 64: worst logit error %.2f bf16 ulp of the row scale" % worst)
        # and the shard end to end: 64 rows sampled, deterministic, rows distinct
        kw = dict(temperature=0.9, top_k=50, repetition_penalty=1.05, seed=1234, force_frames=4)
        a, b = m.generate_batch(reqs, **kw), m.generate_batch(reqs, **kw)
        assert all(x.status == 0 and (x.codes == y.codes).all() for x, y in zip(a, b))
        assert len({tuple(r.codes[:, 0]) for r in a}) > 32
    finally:
        m.close()


def test_config3_1p7b_base_voice_clone_batch16_logits_match_the_oracle():
    """BASELINE configs[3]: 1.7B-Base voice clone, batch 16: 3.0 s reference clip per row through the full-size codec
    encoder and speaker encoder, the ICL prompt (about 130 positions, prefilled four per launch) and 4 teacher-forced
    frames behind it. The oracle takes the ENGINE's reference codes for the compared row (the encoder has its own tests;
    an RVQ near-tie flipped by fp32 summation order would change the whole prompt), its own x-vector."""
    import bench
    from oracle import oracle as O
    from qwen3tts import Qwen3TTSModel
    d = _ckpt("1.7b-base")
    m = Qwen3TTSModel.from_pretrained(d, max_batch=16, max_frames=16, max_prompt=192)
    try:
        assert m.info.hidden_size == 2048 and m.supports_voice_cloning
        reqs = bench.build_requests("1.7b-base", 0, 16, 32, 0)
        row = 9
        codes = {row: m.codec_encode(reqs[row].ref_audio)}
        assert codes[row].shape == (16, 38)
        om = O.OracleModel(d)
        worst = _forced_parity(m, om, reqs, rows=(row,), F=4, seed=103, rep=1.5, ref_codes=codes)
        print("1.7B-Base clone batch 16: worst logit error %.2f bf16 ulp of the row scale" % worst)
        # ... and one row END TO END against the oracle: its own encoder, its own x-vector, its own ICL prompt. Taken from the
        # rows whose reference codes the two encoders agree on exactly (an RVQ near-tie flipped by fp32 summation order would
        # change the whole prompt; the encoder's own tests cover those).
        row2 = None
        for cand in (2, 5, 12, 14):
            if np.array_equal(m.codec_encode(reqs[cand].ref_audio), om.codec_encode(reqs[cand].ref_audio)):
                row2 = cand
                break
        assert row2 is not None, "no row whose reference codes match the oracle's exactly"
        worst2 = _forced_parity(m, om, reqs, rows=(row2,), F=3, seed=113, rep=1.5, ref_codes=None)
        print("1.7B-Base clone batch 16, row %d end to end (oracle's own encoder): worst %.2f ulp" % (row2, worst2))
    finally:
        m.close()
