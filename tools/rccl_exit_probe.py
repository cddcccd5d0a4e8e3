"""Does a process that has used the library's RCCL broadcast still exit cleanly after more GPU work? (round 4: a pytest run
aborted at exit with 'double free or corruption' once RCCL had been opened.)  usage: rccl_exit_probe.py <mode>
  modes: none | id | bcast ; then a 0.6B model is loaded, run and freed, optionally the oracle is used (mode suffix +o)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "swift-qwen3-tts_amd"))
import tempfile

import numpy as np

mode = sys.argv[1]
from qwen3tts import GenerationRequest, Qwen3TTSModel, synth

d = tempfile.mkdtemp()
synth.write_checkpoint(d, "tiny-b", seed=1)
m = Qwen3TTSModel.from_pretrained(d, max_batch=2, max_frames=16, max_prompt=64)
if mode.startswith("id") or mode.startswith("bcast"):
    cid = Qwen3TTSModel.comm_unique_id()
    if mode.startswith("bcast"):
        m.broadcast_weights(cid, 0, 1, 0)
p = synth.synthetic_prompt(0, n_text=8, text_vocab=1000, im_start=1000, im_end=1001)
req = GenerationRequest(p["text_ids"], p["target_token_count"], None, "aiden", "english")
r = m.generate_batch([req], temperature=0.0, force_frames=4)
m.close()
if "+big" in mode:
    import bench
    dd = bench.ensure_checkpoint("0.6b", 0, None)
    mm = Qwen3TTSModel.from_pretrained(dd, max_batch=8, max_frames=16, max_prompt=128)
    rr = mm.generate_batch(bench.build_requests("0.6b", 0, 8, 32, 0), temperature=0.9, seed=1, force_frames=4)
    mm.close()
if "+o" in mode:
    from oracle import oracle as O
    om = O.OracleModel(d)
    om.codec_decode(r[0].codes)
print("done", mode, flush=True)
