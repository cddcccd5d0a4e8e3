"""Voice-clone front end only (codec encoder + speaker encoder at the real shapes) for profiling."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "swift-qwen3-tts_amd"))
from qwen3tts import Qwen3TTSModel, synth  # noqa: E402

d = "/tmp/q3tts_frontend_only"
if not os.path.exists(os.path.join(d, ".complete")):
    synth.write_checkpoint(d, "tiny-base-fullenc", seed=1234)
    open(os.path.join(d, ".complete"), "w").write("ok")
m = Qwen3TTSModel.from_pretrained(d, max_batch=1, max_frames=8, max_prompt=64)
clip = synth.synthetic_reference_audio(0, float(sys.argv[1]) if len(sys.argv) > 1 else 3.0)
for i in range(3):
    t0 = time.time()
    m.codec_encode(clip)
    e = m.last_timing().frontend_ms
    m.extract_speaker_embedding(clip)
    s = m.last_timing().frontend_ms
    print(f"encode {e:.2f} ms, speaker {s:.2f} ms, wall {1e3 * (time.time() - t0):.1f} ms", flush=True)
m.close()
