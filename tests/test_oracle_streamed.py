"""Row f1's oracle: OracleModel.codec_decode_streamed restates the windowed streaming decode (include/q3tts.h
`audio_window_frames`) as a composition of the functions the one-shot oracle decode is made of. CPU properties that pin the
composition itself; the HIP path is compared with it in tests/test_streaming.py."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def om(ckpt_dirs):
    from oracle import oracle as O
    return O.OracleModel(ckpt_dirs["tiny-b"])


def _codes(F, seed=3):
    return np.random.default_rng(seed).integers(1, 32, size=(F, 16)).astype(np.int32)


def test_a_window_over_everything_is_the_one_shot_decode(om):
    codes = _codes(11)
    want, _ = om.codec_decode(codes)
    for chunk, W, L in ((11, 0, 0), (4, -1, 0), (4, 11, 11), (64, 0, 0)):
        got = om.codec_decode_streamed(codes, chunk, W, L)
        assert got.shape == want.shape and (got == want).all(), (chunk, W, L)


def test_a_chunk_depends_on_its_window_and_nothing_later(om):
    """Chunk [f0, f1) may only see codes below f1 + lookahead: changing later frames must not move a sample of it; changing a
    frame inside the look-ahead must (the pre_transformer attends to it)."""
    F, C, W, L = 12, 4, 3, 2
    a = _codes(F)
    b = a.copy()
    b[C + L:] = _codes(F, seed=9)[C + L:]          # everything beyond chunk 0's look-ahead
    pa, pb = om.codec_decode_streamed(a, C, W, L), om.codec_decode_streamed(b, C, W, L)
    assert (pa[: C * 1920] == pb[: C * 1920]).all()
    c = a.copy()
    c[C + L - 1] = (c[C + L - 1] + 1) % 32            # the last frame chunk 0 can see
    pc = om.codec_decode_streamed(c, C, W, L)
    assert np.abs(pa[: C * 1920] - pc[: C * 1920]).max() > 0


def test_the_window_is_an_approximation_and_shrinks_with_context(om):
    codes = _codes(20, seed=5)
    want, _ = om.codec_decode(codes)
    e_small = np.abs(om.codec_decode_streamed(codes, 4, 1, 0) - want).max()
    e_big = np.abs(om.codec_decode_streamed(codes, 4, 12, 4) - want).max()
    assert e_small > 0 and e_big <= e_small
