"""Row f3 end to end on the GPU box: the command-line front end (python -m qwen3tts; flags, prints and WAV output of the
reference demo, Sources/Qwen3TTSDemo/main.swift:34-89, 134-165, 294-313) run as a program would run it -- checkpoint
directory with tokenizer files on disk, text in, WAV file out -- and the file checked against the oracle's decode of the
same codes."""
import os
import re
import shutil
import struct

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.gpu
def test_cli_writes_the_reference_wav(tmp_path, capsys):
    from oracle import oracle as O
    from qwen3tts import GenerationRequest, Qwen3TTSModel, audio, synth
    from qwen3tts.__main__ import main
    from qwen3tts.model import chat_template_ids
    d = str(tmp_path / "model")
    synth.write_checkpoint(d, "tiny-b", seed=1234)
    shutil.copy(os.path.join(GOLD, "tokenizer.json"), os.path.join(d, "tokenizer.json"))
    out = str(tmp_path / "out.wav")
    text = "Hello there, it's a test."
    rc = main(["--model", d, "--text", text, "--speaker", "aiden", "--language", "english", "--temperature", "0",
               "--max-tokens", "9", "--output", out])
    assert rc == 0
    printed = capsys.readouterr().out
    # the demo's accounting lines (main.swift:294-313)
    m = re.search(r"Generated (\d+) samples \(([\d.]+)s audio\)", printed)
    assert m and "Real-time factor:" in printed and "Model loaded in" in printed and "Saved to: " + out in printed
    n = int(m.group(1))
    raw = open(out, "rb").read()
    assert raw[:4] == b"RIFF" and raw[8:16] == b"WAVEfmt " and raw[36:40] == b"data"
    assert struct.unpack("<IHHIIHH", raw[16:36]) == (16, 1, 1, 24000, 48000, 2, 16)        # main.swift:138-156
    assert struct.unpack("<I", raw[40:44])[0] == 2 * n == len(raw) - 44
    wav = np.frombuffer(raw[44:], "<i2")

    # the same request through the library: greedy decoding is deterministic, so these are the codes the CLI decoded
    mdl = Qwen3TTSModel.from_pretrained(d, max_batch=1, max_frames=32, max_prompt=128)
    try:
        ids = chat_template_ids(mdl.tokenizer, text, None)
        req = GenerationRequest(ids["text_ids"], ids["target_token_count"], None, "aiden", "english", 9)
        res = mdl.generate_batch([req], temperature=0.0)[0]
        info = res.info
    finally:
        mdl.close()
    F = res.codes.shape[0]
    assert 0 < F <= 9
    assert n == int((res.codes[:, 0] > 0).sum()) * 1920 == res.audio.size                   # SpeechTokenizer.swift:831-833
    assert (wav == audio.pcm_to_int16(res.audio)).all()                                     # Int16(clamp(x) * 32767), :158-162
    pcm, valid = O.OracleModel(d).codec_decode(res.codes)
    assert valid == n
    assert np.abs(wav.astype(np.int32) - O.pcm_to_int16(pcm[:n]).astype(np.int32)).max() <= 1   # oracle's PCM, within one LSB
    # AudioGenerationInfo.summary (GenerationTypes.swift:39-45)
    lines = info.summary.split("\n")
    assert len(lines) == 3 and lines[0].startswith("Prompt:     %d tokens, " % info.prompt_token_count)
    assert lines[1].startswith("Generation: %d tokens, %.2f tokens/s, " % (F, info.tokens_per_second)) and lines[2].startswith("Peak Memory Usage: ")


@pytest.mark.gpu
def test_int16_and_wav_on_the_gpu_box(tmp_path):
    """The quantiser / WAV writer (host code inside libq3tts_hip.so) again under the gpu marker."""
    from test_host_logic import test_int16_quantisation_and_wav_file_follow_the_reference_cli as t
    t(tmp_path)
