"""Scheduling switches that only regroup work must leave results bit-identical: projected embedding tables vs run-time
projection, a replica filled through the weight arena, rows per launch / row split / gate-up pairing."""
import os

import numpy as np
import pytest

from conftest import tiny_request

pytestmark = pytest.mark.gpu


def _req(max_tokens=2048, **kw):
    from qwen3tts import GenerationRequest
    r = tiny_request(**kw)
    return GenerationRequest(r["text_ids"], r["target_token_count"], r["instruct_ids"], r["speaker"], r["language"], max_tokens)


def test_projected_embedding_tables_change_nothing(ckpt_dirs, monkeypatch):
    """The 1.7B-like preset projects every code embedding (small_to_mtp_projection) before the next predictor pass; the
    engine takes that projection once per table row at load (Engine::build_cp_proj_tables) instead of once per pass.
    Same GEMM kernel either way, so logits under teacher forcing, sampled codes and PCM must be bit-identical."""
    from qwen3tts import Qwen3TTSModel
    d = ckpt_dirs["tiny-b"]   # talker 384 wide, predictor 256: has the projection
    a = Qwen3TTSModel.from_pretrained(d, max_batch=6, max_frames=64, max_prompt=96)
    monkeypatch.setenv("Q3TTS_NO_PROJ_TABLES", "1")
    b = Qwen3TTSModel.from_pretrained(d, max_batch=6, max_frames=64, max_prompt=96)
    try:
        reqs = [_req(row=i, n_text=6 + 3 * i) for i in range(5)]
        kw = dict(temperature=0.9, top_k=40, top_p=0.95, repetition_penalty=1.05, seed=23, force_frames=24)
        for x, y in zip(a.generate_batch(reqs, **kw), b.generate_batch(reqs, **kw)):
            assert (x.codes == y.codes).all() and (x.audio == y.audio).all()
        rng = np.random.default_rng(4)
        forced = np.concatenate([rng.integers(0, 2048, size=(3, 5, 1)), rng.integers(0, 256, size=(3, 5, 15))], -1).astype(np.int32)
        ta, ca, sa = a.debug_generate_forced(reqs[:3], forced, temperature=0.0)
        tb, cb, sb = b.debug_generate_forced(reqs[:3], forced, temperature=0.0)
        assert (ta == tb).all() and (ca == cb).all() and (sa == sb).all()
    finally:
        a.close()
        b.close()


def test_replica_filled_through_the_arena_generates_the_same(ckpt_dirs):
    """Multi-GPU load path: a replica allocates its weight arena empty (weights_from_broadcast) and the
    caller fills it afterwards -- RCCL broadcast in bench.py, a device copy here. Everything the engine derives from
    the weights (the projected embedding tables) must therefore be cut at first use, not at load."""
    import ctypes
    from qwen3tts import Qwen3TTSModel
    hip = ctypes.CDLL("libamdhip64.so")
    d = ckpt_dirs["tiny-b"]
    a = Qwen3TTSModel.from_pretrained(d, max_batch=4, max_frames=48, max_prompt=96)
    b = Qwen3TTSModel.from_pretrained(d, max_batch=4, max_frames=48, max_prompt=96, weights_from_broadcast=True)
    try:
        (pa, na), (pb, nb) = a.arena(), b.arena()
        assert na == nb and na > 0
        assert hip.hipMemcpy(ctypes.c_void_p(pb), ctypes.c_void_p(pa), ctypes.c_size_t(na), 3) == 0  # device to device
        assert hip.hipDeviceSynchronize() == 0
        reqs = [_req(row=i, n_text=7 + 2 * i) for i in range(3)]
        kw = dict(temperature=0.9, top_k=40, top_p=1.0, repetition_penalty=1.05, seed=5, force_frames=20)
        for x, y in zip(a.generate_batch(reqs, **kw), b.generate_batch(reqs, **kw)):
            assert (x.codes == y.codes).all() and (x.audio == y.audio).all()
    finally:
        a.close()
        b.close()


@pytest.mark.parametrize("name", ["tiny-a", "tiny-b"])
def test_launch_geometry_switches_change_nothing(ckpt_dirs, monkeypatch, name):
    """Rows per launch (256 vs 64: prefill chunk length, one- or two-pass predictor step 0), the row split of narrow GEMMs,
    the gate/up pairing, the tall prefill GEMM (off / its other tile shape) and the chunk attention's query split only regroup work; per-row arithmetic is the
    same, so codes and PCM must be bit-identical."""
    from qwen3tts import Qwen3TTSModel
    d = ckpt_dirs[name]
    reqs = [_req(row=i, n_text=5 + 4 * i) for i in range(6)]
    kw = dict(temperature=0.9, top_k=40, top_p=0.95, repetition_penalty=1.05, seed=31, force_frames=16)
    ref = None
    all_switches = ("Q3TTS_ROWS_64", "Q3TTS_GEMM_NO_ROW_SPLIT", "Q3TTS_GEMM_ONE_PAIR", "Q3TTS_NO_TALL_GEMM", "Q3TTS_TALL_SHAPE",
                    "Q3TTS_CHUNK_QSPLIT", "Q3TTS_PF", "Q3TTS_PF_BUDGET_KB", "Q3TTS_PF_AHEAD")
    for env in ({}, {"Q3TTS_ROWS_64": "1"}, {"Q3TTS_GEMM_NO_ROW_SPLIT": "1", "Q3TTS_GEMM_ONE_PAIR": "1"}, {"Q3TTS_NO_TALL_GEMM": "1"},
                {"Q3TTS_TALL_SHAPE": "2", "Q3TTS_CHUNK_QSPLIT": "1"}, {"Q3TTS_CHUNK_QSPLIT": "4"},
                # next-launch weight touch (kernels/prefetch.h): off, and two extreme plans. The SHIPPED build compiles the touch code out
                # (it lost), so these three only exercise the switches' plumbing there; on a library built by
                # tools/build_pf_variants.sh they are the bit-identity check of the planner (run that way in round 4)
                {"Q3TTS_PF": "0"}, {"Q3TTS_PF_BUDGET_KB": "64", "Q3TTS_PF_AHEAD": "7"}, {"Q3TTS_PF_BUDGET_KB": "100000", "Q3TTS_PF_AHEAD": "1"}):
        for k in all_switches:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        m = Qwen3TTSModel.from_pretrained(d, max_batch=6, max_frames=32, max_prompt=96)
        try:
            out = m.generate_batch(reqs, **kw)
        finally:
            m.close()
        if ref is None:
            ref = out
        else:
            for x, y in zip(ref, out):
                assert (x.codes == y.codes).all() and (x.audio == y.audio).all(), env
    from qwen3tts import _lib
    for k in all_switches:
        monkeypatch.delenv(k, raising=False)
    _lib.reload_debug_env()  # the switches are process-wide and read at model load: leave the defaults behind


def test_pipelined_jobs_equal_sequential_calls(ckpt_dirs):
    """q3tts_generate_begin / _end: the second batch's AR loop runs while the first batch's codec decode is still in
    flight on the codec stream. The decode reads job-owned copies of the codes, so interleaving must not change a bit:
    [begin A, begin B, end A, begin C, end B, end C] == three plain generate calls. A third begin without an end is refused."""
    from qwen3tts import Qwen3TTSError, Qwen3TTSModel
    m = Qwen3TTSModel.from_pretrained(ckpt_dirs["tiny-b"], max_batch=6, max_frames=64, max_prompt=96)
    try:
        batches = [[_req(row=i + 10 * k, n_text=5 + 2 * i + k) for i in range(3 + k)] for k in range(3)]
        kws = [dict(temperature=0.9, top_k=40, repetition_penalty=1.05, seed=50 + k, force_frames=20 + 7 * k) for k in range(3)]
        want = [m.generate_batch(b, **kw) for b, kw in zip(batches, kws)]
        events = []
        ja = m.generate_batch_begin(batches[0], on_event=lambda i, k, p: events.append(("a", i, k)), **kws[0])
        jb = m.generate_batch_begin(batches[1], **kws[1])
        with pytest.raises(Qwen3TTSError):
            m.generate_batch_begin(batches[2], **kws[2])
        ra = m.generate_batch_end(ja)
        jc = m.generate_batch_begin(batches[2], **kws[2])
        rb = m.generate_batch_end(jb)
        rc = m.generate_batch_end(jc)
        for got, exp in zip((ra, rb, rc), want):
            assert len(got) == len(exp)
            for x, y in zip(got, exp):
                assert x.status == 0 and (x.codes == y.codes).all() and (x.audio == y.audio).all()
        kinds = [k for (_, i, k) in events if i == 0]
        assert kinds == ["token"] * 20 + ["info", "audio"]  # TOKEN in begin, INFO / AUDIO in end, reference order
        tm = m.last_timing()
        assert tm.codec_ms > 0 and tm.decode_ms > 0 and tm.rows == 5
        # results are views over buffers the library handed over (no copy in the mirror): they must survive later jobs in the
        # same job slots, and the model itself
        keep = [(x.audio, x.audio.copy(), x.codes, x.codes.copy()) for x in ra]
        assert all(a.base is not None for a, _, _, _ in keep)
        del ra
        jd = m.generate_batch_begin(batches[1], **kws[1])
        je = m.generate_batch_begin(batches[2], more_follows=False, **kws[2])
        rd, re_ = m.generate_batch_end(jd), m.generate_batch_end(je)
        for got, exp in zip((rd, re_), want[1:]):
            for x, y in zip(got, exp):
                assert (x.codes == y.codes).all() and (x.audio == y.audio).all()
    finally:
        m.close()
    for a, a0, c, c0 in keep:
        assert (a == a0).all() and (c == c0).all()


def test_a_failed_begin_leaves_the_outstanding_job_and_both_slots_intact(ckpt_dirs):
    """q3tts_generate_begin that fails -- an unknown speaker (caught while the requests are resolved), a prompt beyond max_prompt
    (caught while the prompts are assembled, after the voice front end would have run) -- must not leak its job slot or disturb
    the job already in flight: that job ends with the undisturbed result, and two further jobs can be begun afterwards."""
    from qwen3tts import GenerationRequest, Qwen3TTSError, Qwen3TTSModel
    m = Qwen3TTSModel.from_pretrained(ckpt_dirs["tiny-b"], max_batch=4, max_frames=48, max_prompt=64)
    try:
        a = [_req(row=i, n_text=5 + i) for i in range(3)]
        b = [_req(row=7 + i, n_text=6 + i) for i in range(2)]
        kw = dict(temperature=0.9, top_k=40, seed=9, force_frames=10)
        want_a, want_b = m.generate_batch(a, **kw), m.generate_batch(b, **kw)
        ja = m.generate_batch_begin(a, **kw)
        bad_speaker = _req(row=1, n_text=5)
        bad_speaker.speaker = "nobody"
        too_long = _req(row=2, n_text=80)
        for bad in ([bad_speaker], [a[0], too_long]):
            with pytest.raises(Qwen3TTSError) as e:
                m.generate_batch_begin(bad, **kw)
            assert e.value.status == 3
        ra = m.generate_batch_end(ja)
        for x, y in zip(ra, want_a):
            assert x.status == 0 and (x.codes == y.codes).all() and (x.audio == y.audio).all()
        j1 = m.generate_batch_begin(b, **kw)           # both slots are free again
        j2 = m.generate_batch_begin(a, more_follows=False, **kw)
        r1, r2 = m.generate_batch_end(j1), m.generate_batch_end(j2)
        for got, exp in ((r1, want_b), (r2, want_a)):
            for x, y in zip(got, exp):
                assert x.status == 0 and (x.codes == y.codes).all() and (x.audio == y.audio).all()
    finally:
        m.close()


def test_model_freed_with_jobs_outstanding():
    """q3tts_model_free with two begun jobs that were never ended (a host that gives up mid-queue): the streams are drained, the
    staging thread is joined, nothing is leaked into the next model of the process -- which then generates what the first one
    did. In a child process: the failure mode of this path is a crash or a hang, not an assertion."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "_free_outstanding_worker.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-1000:] + r.stderr[-3000:]
