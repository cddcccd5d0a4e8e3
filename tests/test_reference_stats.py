"""The reference's only known-answer data: stage statistics of the codec decoder for five fixed frames
(Tests/Qwen3TTSTests/Qwen3TTSTests.swift:37-43 codes, :71-271 values, :274-275 asserts), committed as data in
tests/golden/reference_codec_stats.json.

The values are functions of the real Qwen3-TTS-1.7B-VoiceDesign speech_tokenizer weights. Offline there are no
checkpoints, so:
  * always: the weight-independent part (stage shapes = frames x stride products, sample count) is checked against the
    oracle at the real layer widths with synthetic weights;
  * opt-in, exactly like the reference's own test: when QWEN3_TTS_VOICEDESIGN_MODEL_PATH points at the checkpoint, the
    oracle (CPU) and the HIP stage hooks (GPU) are both held to every recorded statistic."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
STATS = json.load(open(os.path.join(HERE, "golden", "reference_codec_stats.json")))
CODES = np.asarray(STATS["codes"], np.int32)
REAL = os.environ.get("QWEN3_TTS_VOICEDESIGN_MODEL_PATH")  # the reference's variable (TestResources.swift:36)


def _check_stage(name, a, exp):
    assert list(a.shape) == exp["shape_tc"], (name, a.shape)
    tol = lambda v: 6e-4 + 1.5e-3 * abs(v)  # values are printed with four decimals
    if "min" in exp:
        assert abs(float(a.min()) - exp["min"]) <= tol(exp["min"]), (name, "min", float(a.min()))
        assert abs(float(a.max()) - exp["max"]) <= tol(exp["max"]), (name, "max", float(a.max()))
    if "rms" in exp:
        rms = float(np.sqrt(np.mean(a.astype(np.float64) ** 2)))
        assert abs(rms - exp["rms"]) <= tol(exp["rms"]), (name, "rms", rms)
    if "std" in exp:
        assert abs(float(a.astype(np.float64).std()) - exp["std"]) <= tol(exp["std"]), (name, "std", float(a.std()))
    if "first_channel0_values" in exp:
        want = np.asarray(exp["first_channel0_values"], np.float32)
        assert np.abs(a[0, :want.size] - want).max() <= 6e-3, (name, a[0, :want.size])  # two decimals


def test_fixture_is_consistent():
    assert CODES.shape == (5, 16) and CODES.min() >= 0 and CODES.max() < 2048
    up = 1
    for nm, exp in STATS["stages"].items():
        assert exp["shape_tc"][0] % 5 == 0
    assert STATS["audio"]["n_samples"] == 5 * 1920
    a = STATS["asserted_by_reference"]
    q = STATS["stages"]["quantizer"]
    assert q["std"] > a["quantizer_std_gt"] and abs(q["min"] - a["quantizer_min_near"]) < a["quantizer_min_tol"]


@pytest.fixture(scope="module")
def full_codec_dir(tmp_path_factory):
    """Tiny talker, FULL-SIZE codec decoder (the shapes of the reference's test do not depend on the weights)."""
    from qwen3tts import synth
    d = str(tmp_path_factory.mktemp("full_codec_stats"))
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


def test_oracle_stage_shapes_match_the_reference_test(full_codec_dir):
    from oracle import oracle as O
    om = O.OracleModel(full_codec_dir)
    st = {}
    pcm, valid = om.codec_decode(CODES, st)
    for name, exp in STATS["stages"].items():
        assert list(st[name].shape) == exp["shape_tc"], (name, st[name].shape)
    assert pcm.shape == (STATS["audio"]["n_samples"],) and valid == 9600
    assert list(om.codec["decoder.decoder.initConv.conv.weight"].shape) == STATS["weights"]["decoder.decoder.initConv.conv.weight"]["shape"]


@pytest.mark.gpu
def test_hip_stage_shapes_match_the_reference_test(full_codec_dir):
    from qwen3tts import Qwen3TTSModel
    m = Qwen3TTSModel.from_pretrained(full_codec_dir, max_batch=1, max_frames=8, max_prompt=64)
    try:
        for name, exp in STATS["stages"].items():
            assert list(m.debug_codec_stage(CODES, name).shape) == exp["shape_tc"], name
        pcm, lens = m.codec_decode(CODES[None])
        assert pcm.shape[-1] == 9600 and lens[0] == 9600
    finally:
        m.close()


@pytest.mark.skipif(not REAL, reason="QWEN3_TTS_VOICEDESIGN_MODEL_PATH not set (the reference's own test skips the same way)")
def test_oracle_reproduces_the_reference_statistics():
    from oracle import oracle as O
    om = O.OracleModel(REAL)
    st = {}
    pcm, _ = om.codec_decode(CODES, st)
    for name, exp in STATS["stages"].items():
        _check_stage(name, st[name], exp)
    w = STATS["weights"]
    iw = om.codec["decoder.decoder.initConv.conv.weight"]
    assert abs(float(iw.min()) - w["decoder.decoder.initConv.conv.weight"]["min"]) < 2e-6
    assert abs(float(iw.std()) - w["decoder.decoder.initConv.conv.weight"]["std"]) < 2e-6
    ib = om.codec["decoder.decoder.initConv.conv.bias"]
    assert abs(float(ib.mean()) - w["decoder.decoder.initConv.conv.bias"]["mean"]) < 2e-6
    sn = w["decoder.decoder.block0.snake"]
    assert abs(float(np.exp(om.codec["decoder.decoder.block0.snake.alpha"]).mean()) - sn["exp_alpha_mean"]) < 2e-6
    assert abs(float(np.exp(om.codec["decoder.decoder.block0.snake.beta"]).mean()) - sn["exp_beta_mean"]) < 2e-6
    ones = STATS["init_conv_on_ones"]
    y = om._conv(np.ones(ones["input_shape_tc"], np.float32), "decoder.decoder.initConv.conv", 7)
    assert abs(float(y.min()) - ones["min"]) < 6e-4 and abs(float(y.max()) - ones["max"]) < 6e-4
    assert abs(float(y.mean()) - ones["mean"]) < 2e-6 and abs(float(y.std()) - ones["std"]) < 6e-4
    a = STATS["audio"]
    assert pcm.shape == (a["n_samples"],)
    assert abs(float(pcm.min()) - a["min"]) < 6e-4 and abs(float(pcm.max()) - a["max"]) < 6e-4
    assert abs(float(pcm.astype(np.float64).std()) - a["std"]) < 6e-4


@pytest.mark.gpu
@pytest.mark.skipif(not REAL, reason="QWEN3_TTS_VOICEDESIGN_MODEL_PATH not set (the reference's own test skips the same way)")
def test_hip_reproduces_the_reference_statistics():
    from qwen3tts import Qwen3TTSModel
    m = Qwen3TTSModel.from_pretrained(REAL, max_batch=1, max_frames=8, max_prompt=64)
    try:
        for name, exp in STATS["stages"].items():
            _check_stage(name, m.debug_codec_stage(CODES, name), exp)
        pcm, _ = m.codec_decode(CODES[None])
        a = STATS["audio"]
        pcm = pcm[0, :a["n_samples"]]
        assert abs(float(pcm.min()) - a["min"]) < 6e-4 and abs(float(pcm.max()) - a["max"]) < 6e-4
        assert abs(float(pcm.astype(np.float64).std()) - a["std"]) < 6e-4
    finally:
        m.close()
