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
d = "/tmp/q3tts_codec_only"
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
    synth.save_safetensors(os.path.join(d, "speech_tokenizer", "model.safetensors"),
                           synth.codec_tensors(p["speech_tokenizer"]["decoder_config"], g))
    open(os.path.join(d, ".complete"), "w").write("ok")
m = Qwen3TTSModel.from_pretrained(d, max_batch=1, max_frames=8, max_prompt=64)
rng = np.random.default_rng(0)
codes = rng.integers(1, 2048, size=(B, F, 16)).astype(np.int32)
for i in range(reps):
    t0 = time.time()
    m.codec_decode(codes)
    print(f"codec_decode B={B} F={F}: wall {1e3 * (time.time() - t0):.1f} ms, device {m.last_timing().codec_ms:.1f} ms", flush=True)
m.close()
