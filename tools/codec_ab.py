"""A/B of a codec-decoder switch on one box: the same codes decoded under each environment variant (a fresh model load per
variant: the switches are read at load), device time of the decode and whether the PCM is bit-identical to the first variant's.
  usage: codec_ab.py B F "VAR=1" "OTHER=1" ...     ("-" = no variable set; Q3TTS_CODEC_ONLY_F16=1 selects the float16 checkpoint)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "swift-qwen3-tts_amd"))
from qwen3tts import Qwen3TTSModel  # noqa: E402

B, F = int(sys.argv[1]), int(sys.argv[2])
variants = sys.argv[3:] or ["-"]
f16 = os.environ.get("Q3TTS_CODEC_ONLY_F16") == "1"
d = "/tmp/q3tts_codec_only" + ("_f16" if f16 else "")
if not os.path.exists(os.path.join(d, ".complete")):
    import subprocess
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "codec_only.py"), "1", "2"], stdout=subprocess.DEVNULL)
codes = np.random.default_rng(0).integers(1, 2048, size=(B, F, 16)).astype(np.int32)
ref = None
for v in variants:
    keys = []
    if v != "-":
        for kv in v.split():
            k, val = kv.split("=", 1)
            os.environ[k] = val
            keys.append(k)
    m = Qwen3TTSModel.from_pretrained(d, max_batch=1, max_frames=8, max_prompt=64)
    best = 1e9
    for _ in range(4):
        out, _ = m.codec_decode(codes)
        best = min(best, m.last_timing().codec_ms)
    m.close()
    for k in keys:
        del os.environ[k]
    same = "reference" if ref is None else ("bit-identical" if (out == ref).all() else "max |diff| %.3e" % np.abs(out - ref).max())
    if ref is None:
        ref = out
    print(f"{v:40s} codec decode {B} x {F}: {best:8.2f} ms   PCM {same}", flush=True)
