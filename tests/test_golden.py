"""Oracle vs the committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py) and
vs the shape facts the reference's own test holds (Tests/Qwen3TTSTests/Qwen3TTSTests.swift:119-120,175-253)."""
import os

import numpy as np
import pytest

from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["tiny-a", "tiny-b"])
def test_oracle_reproduces_golden(ckpt_dirs, name):
    g = np.load(os.path.join(GOLD, name.replace("-", "_") + ".npz"))
    om = O.OracleModel(ckpt_dirs[name])
    req = O.Request(text_ids=g["text_ids"].tolist(), target_token_count=12, speaker="aiden", language="english")
    ie, tr, pad = om.prepare_generation_inputs(req)
    assert np.array_equal(ie, g["input_embeds"]) and np.array_equal(tr, g["trailing"]) and np.array_equal(pad, g["tts_pad"])
    greedy = om.generate_codes(req, O.Sampling(temperature=0.0, repetition_penalty=1.05, force_frames=6), keep_logits=True)
    assert np.array_equal(greedy.codes, g["greedy_codes"])                      # integer work: bit-exact
    assert np.array_equal(np.stack(greedy.talker_logits), g["greedy_talker_logits"])
    sampled = om.generate_codes(req, O.Sampling(temperature=0.9, top_k=50, seed=42, force_frames=6))
    assert np.array_equal(sampled.codes, g["sampled_codes"])                    # Philox + q3_logf: bit-exact
    pcm, valid = om.codec_decode(greedy.codes)
    assert valid == int(g["valid"]) and np.abs(pcm - g["pcm"]).max() <= 1e-6      # fp32, same libm


def test_decoder_stage_lengths_match_the_reference_test(ckpt_dirs):
    """The reference test feeds 5 frames and expects lengths 10 -> 20 after the two 2x stages and
    20 -> 160 -> 800 -> 3200 -> 9600 through the 8/5/4/3 blocks (Qwen3TTSTests.swift:119-120, 175-253):
    L_out = L * stride exactly, because only the right side is trimmed."""
    om = O.OracleModel(ckpt_dirs["tiny-a"])
    codes = np.array([[5 + i] * 16 for i in range(5)], np.int32)
    st = {}
    pcm, valid = om.codec_decode(codes, st)
    assert [st[k].shape[0] for k in ("pre_transformer", "upsample0", "upsample1", "init_conv", "block0", "block1", "block2",
                                     "block3")] == [5, 10, 20, 20, 160, 800, 3200, 9600]
    assert pcm.shape == (9600,) and valid == 9600 and np.abs(pcm).max() <= 1.0


def test_audio_lengths_count_only_positive_first_codes(ckpt_dirs):
    """audioLengths = count(code0 > 0) * 1920 (SpeechTokenizer.swift:831-833): a legitimate id 0 counts as padding."""
    om = O.OracleModel(ckpt_dirs["tiny-a"])
    codes = np.full((4, 16), 9, np.int32)
    codes[1, 0] = 0
    _, valid = om.codec_decode(codes)
    assert valid == 3 * 1920
