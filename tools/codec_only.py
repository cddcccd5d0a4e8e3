"""Codec-decoder-only workload for profiling (rocprofv3 --pmc / --kernel-trace): B x F random codes -> PCM."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "swift-qwen3-tts_amd"))
from qwen3tts import Qwen3TTSModel, synth  # noqa: E402

B, F, reps = int(sys.argv[1]) if len(sys.argv) > 1 else 16, int(sys.argv[2]) if len(sys.argv) > 2 else 200, 3
f16 = os.environ.get("Q3TTS_CODEC_ONLY_F16") == "1"   # the speech tokenizer stored in float16 ("lite" checkpoints): codec_conv_h1.hip
d = "/tmp/q3tts_codec_only" + ("_f16" if f16 else "")
if not os.path.exists(os.path.join(d, ".complete")):
    p = synth.preset("tiny-a")          # tiny talker, FULL-SIZE codec decoder
    p["speech_tokenizer"]["decoder_config"] = synth._codec_cfg(False)
    p["config"]["talker_config"]["code_predictor_config"]["vocab_size"] = 2048
    os.makedirs(os.path.join(d, "speech_tokenizer"), exist_ok=True)
    import json
    g = synth._Gen(1234, False)
    json.dump(p["config"], open(os.path.join(d, "config.json"), "w"))
    json.dump(p["speech_tokenizer"], open(os.path.join(d, "speech_tokenizer", "config.json"), "w"))
    synth.save_safetensors(os.path.join(d, "model.safetensors"), synth.talker_tensors(p["config"], g))
    ct = synth.codec_tensors(p["speech_tokenizer"]["decoder_config"], g, out_wstd=synth.FULL_WIDTH_OUT_WSTD)
    if f16:
        ct = {k: (("F16", v.astype(np.float16)) if tag == "F32" else (tag, v)) for k, (tag, v) in ct.items()}
    synth.save_safetensors(os.path.join(d, "speech_tokenizer", "model.safetensors"), ct)
    open(os.path.join(d, ".complete"), "w").write("ok")
rng = np.random.default_rng(0)
codes = rng.integers(1, 2048, size=(B, F, 16)).astype(np.int32)
pcm = {}
modes = ("0", "1") if os.environ.get("Q3TTS_CODEC_COMPARE") else (os.environ.get("Q3TTS_CODEC_FP32", "0"),)
for mode in modes:   # "1": fp32 matrix-core path, "0": fp16x2 (default)
    os.environ["Q3TTS_CODEC_FP32"] = mode
    m = Qwen3TTSModel.from_pretrained(d, max_batch=1, max_frames=8, max_prompt=64)
    for i in range(reps):
        t0 = time.time()
        out = m.codec_decode(codes)
        print(f"codec_decode fp32_mfma={mode} B={B} F={F}: wall {1e3 * (time.time() - t0):.1f} ms, "
              f"device {m.last_timing().codec_ms:.1f} ms", flush=True)
    pcm[mode] = np.asarray(out[0] if isinstance(out, (tuple, list)) else out)
    m.close()
if len(pcm) == 2:
    b = pcm["1"]
    for nm, a in (("fp16x2", pcm["0"]),):
        print(f"{nm} vs fp32 MFMA: max |diff| {np.abs(a - b).max():.3e}, rms diff {np.sqrt(np.mean((a - b) ** 2)):.3e}, "
              f"rms signal {np.sqrt(np.mean(b ** 2)):.3e}, clipped {np.mean(np.abs(b) >= 1.0):.3f}, finite {np.isfinite(a).all()}", flush=True)
