"""Host-side logic that needs no GPU: sharding, prompt templates, synthetic checkpoint layout."""
import json
import os

import numpy as np


def test_shard_rows_partitions_exactly():
    from bench import shard_rows
    for total in (1, 7, 32, 64, 256, 512):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_rows(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_chat_template_ids_follow_the_reference():
    from qwen3tts import chat_template_ids
    seen = []

    def tok(s):
        seen.append(s)
        return list(range(len(s.split())))

    out = chat_template_ids(tok, "hello there", "calm voice")
    # Qwen3.swift:274-275, 364-365, 822
    assert seen[0] == "<|im_start|>assistant\nhello there<|im_end|>\n<|im_start|>assistant\n"
    assert seen[1] == "hello there"
    assert seen[2] == "<|im_start|>user\ncalm voice<|im_end|>\n"
    assert out["target_token_count"] == 2 and "instruct_ids" in out


def test_synthetic_checkpoint_uses_the_reference_layout(ckpt_dirs):
    from oracle import oracle as O
    d = ckpt_dirs["tiny-b"]
    main = O.load_safetensors_dir(d)
    st = O.load_safetensors_dir(os.path.join(d, "speech_tokenizer"))
    cfg = json.load(open(os.path.join(d, "config.json")))
    assert cfg["talker_config"]["code_predictor_config"]["hidden_size"] != cfg["talker_config"]["hidden_size"]
    # module-tree keys (Talker.swift:165-171, CodePredictor.swift:283-288)
    for k in ("talker.model.layers.0.self_attn.q_norm.weight", "talker.codec_head.weight",
              "talker.text_projection.linear_fc1.bias", "talker.code_predictor.small_to_mtp_projection.bias",
              "talker.code_predictor.lm_head.14.weight", "talker.code_predictor.model.codec_embedding.14.weight"):
        assert k in main
    # upstream speech-tokenizer names that the sanitiser must remap (Qwen3.swift:1504-1512, 1581-1588)
    for k in ("decoder.decoder.1.block.1.conv.weight", "decoder.decoder.4.block.4.conv2.conv.weight",
              "decoder.quantizer.rvq_rest.vq.layers.14._codebook.embedding_sum", "decoder.upsample.1.0.conv.weight"):
        assert k in st
    assert st["decoder.decoder.1.block.1.conv.weight"].shape == (1280, 640, 16)  # torch ConvTranspose1d [in,out,k]


def test_sanitiser_restatement(ckpt_dirs):
    from oracle import oracle as O
    d = ckpt_dirs["tiny-a"]
    raw = O.load_safetensors_dir(os.path.join(d, "speech_tokenizer"))
    s = O.sanitize_speech_tokenizer(raw)
    # index -> name remap and MLX layouts [O][K][I]
    assert s["decoder.decoder.block0.upsample.conv.weight"].shape == (640, 16, 1280)
    assert s["decoder.decoder.block3.res3.conv2.conv.weight"].shape == (80, 1, 80)
    assert s["decoder.decoder.outConv.conv.weight"].shape == (1, 7, 80)
    assert s["decoder.upsample.0.1.dwconv.conv.weight"].shape == (128, 7, 1)
    assert s["decoder.quantizer.rvq_first.output_proj.weight"].shape == (128, 1, 64)
    # transposed conv: [in,out,k] -> [out,k,in] (value.transposed(1,2,0), Qwen3.swift:1709)
    w = raw["decoder.decoder.1.block.1.conv.weight"]
    assert np.array_equal(s["decoder.decoder.block0.upsample.conv.weight"], w.transpose(1, 2, 0))
    # codebook = embedding_sum / clip(cluster_usage, 1e-5) (Qwen3.swift:1716-1724), incl. unused clusters
    base = "decoder.quantizer.rvq_rest.vq.layers.3"
    usage = raw[base + "._codebook.cluster_usage"]
    assert (usage == 0).any()
    exp = raw[base + "._codebook.embedding_sum"] / np.clip(usage[:, None], np.float32(1e-5), None)
    assert np.array_equal(s[base + ".codebook.embed.weight"], exp.astype(np.float32))


def test_layout_heuristic_matches_reference_cases():
    from oracle.oracle import _is_mlx_conv_layout as f
    # Qwen3.swift:1246-1260 on the real decoder shapes
    assert not f((1024, 512, 3)) and f((1024, 3, 512))          # pre_conv torch vs MLX
    assert not f((1024, 1, 7)) and not f((96, 96, 1))           # depthwise / k1 conv in torch layout
    assert f((96, 1, 96))                                       # k1 conv already MLX
    assert not f((1536, 768, 16))                               # ConvTranspose1d torch [in,out,k]


def test_int16_quantisation_and_wav_file_follow_the_reference_cli(tmp_path):
    """f3: Int16(clamp(x) * 32767) truncates toward zero (Sources/Qwen3TTSDemo/main.swift:158-162); the WAV file is the
    44-byte PCM header the CLI writes (:138-156) + those samples. Host code behind the C ABI, checked against the oracle's
    restatement and against Python's own WAV reader."""
    import struct
    import wave
    from oracle import oracle as O
    from qwen3tts import audio
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-1.3, 1.3, 5000).astype(np.float32),
                        np.array([0.0, -0.0, 1.0, -1.0, 0.99999, -0.99999, 1.5, -7.0, 3.0518e-05, -3.0518e-05, 0.5, -0.5], np.float32)])
    got = audio.pcm_to_int16(x)
    assert got.dtype == np.int16 and (got == O.pcm_to_int16(x)).all()
    assert got[5002] == 32767 and got[5003] == -32767 and got[5006] == 32767 and got[5007] == -32767  # never -32768
    assert got[5008] == 0 and got[5009] == 0                                                            # toward zero, both signs
    path = tmp_path / "o.wav"
    audio.write_wav(str(path), x, 24000)
    raw = path.read_bytes()
    assert len(raw) == 44 + 2 * x.size and raw[:4] == b"RIFF" and raw[8:16] == b"WAVEfmt " and raw[36:40] == b"data"
    assert struct.unpack("<IHHIIHH", raw[16:36]) == (16, 1, 1, 24000, 48000, 2, 16)
    assert struct.unpack("<I", raw[4:8])[0] == 36 + 2 * x.size and struct.unpack("<I", raw[40:44])[0] == 2 * x.size
    with wave.open(str(path), "rb") as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (1, 2, 24000, x.size)
        assert (np.frombuffer(w.readframes(x.size), "<i2") == got).all()
    sr, back = audio.read_wav(str(path))
    assert sr == 24000 and np.abs(back - np.clip(x, -1, 1)).max() < 1.0 / 16384
