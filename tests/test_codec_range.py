"""The decoder's default convolutions split every fp32 operand into two fp16 planes (csrc/kernels/codec_conv.hip): same
accuracy as the fp32 matrix cores, but an activation beyond 65504 cannot be represented. Such a decode must not hand out a
waveform with holes in it: the row reports Q3TTS_ERR_AUDIO_DECODING_FAILED and names the load option that has the
reference's range (codec_fp32), under which the same checkpoint decodes."""
import os

import numpy as np
import pytest


@pytest.mark.gpu
def test_out_of_range_activation_is_reported_and_the_fp32_path_decodes(tmp_path):
    from safetensors.numpy import load_file, save_file
    from qwen3tts import Qwen3TTSError, Qwen3TTSModel, synth
    d = str(tmp_path / "m")
    synth.write_checkpoint(d, "tiny-a", seed=1234)
    f = os.path.join(d, "speech_tokenizer", "model.safetensors")
    t = load_file(f)
    key = [k for k in t if k.endswith("decoder.0.conv.bias") or k.endswith("initConv.conv.bias")]
    assert key, sorted(t)[:40]
    t[key[0]] = (t[key[0]].astype(np.float32) + np.float32(3.0e5)).astype(t[key[0]].dtype)   # far beyond fp16's 65504
    save_file(t, f)
    codes = np.random.default_rng(0).integers(1, 32, size=(2, 6, 16)).astype(np.int32)
    m = Qwen3TTSModel.from_pretrained(d, max_batch=2, max_frames=16, max_prompt=64)
    try:
        with pytest.raises(Qwen3TTSError) as e:
            m.codec_decode(codes)
        assert e.value.status == 4 and "codec_fp32" in str(e.value)
    finally:
        m.close()
    m = Qwen3TTSModel.from_pretrained(d, max_batch=2, max_frames=16, max_prompt=64, codec_fp32=True)
    try:
        pcm, lens = m.codec_decode(codes)
        assert np.isfinite(pcm).all() and (lens == 6 * 1920).all()
    finally:
        m.close()
