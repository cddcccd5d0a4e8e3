"""Child process of tests/test_scheduling.py::test_model_freed_with_jobs_outstanding (a crash here must not take the test run with it)."""
import os
import sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "swift-qwen3-tts_amd"))
sys.path.insert(0, HERE)
import tempfile
from qwen3tts import Qwen3TTSModel, GenerationRequest, synth
from conftest import tiny_request
d = tempfile.mkdtemp(); synth.write_checkpoint(d, "tiny-b", seed=1234)
def req(row, n):
    r = tiny_request(row=row, n_text=n)
    return GenerationRequest(r["text_ids"], r["target_token_count"], r["instruct_ids"], r["speaker"], r["language"])
m = Qwen3TTSModel.from_pretrained(d, max_batch=4, max_frames=48, max_prompt=64)
kw = dict(temperature=0.9, top_k=40, seed=9, force_frames=30)
want = m.generate_batch([req(0, 6), req(1, 7)], **kw)
j1 = m.generate_batch_begin([req(0, 6), req(1, 7)], **kw)
j2 = m.generate_batch_begin([req(2, 6)], more_follows=False, **kw)
m.close()          # two jobs outstanding, never ended
print("closed with two jobs outstanding")
m = Qwen3TTSModel.from_pretrained(d, max_batch=4, max_frames=48, max_prompt=64)
got = m.generate_batch([req(0, 6), req(1, 7)], **kw)
assert all((a.codes == b.codes).all() and (a.audio == b.audio).all() for a, b in zip(got, want))
m.close()
print("ok")
