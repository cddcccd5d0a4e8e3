"""Regenerates tests/golden/*.npz with the CPU oracle on the synthetic tiny checkpoints.

The reference cannot run here, so these are the oracle's own outputs: they
pin the oracle against regressions and give the GPU tests fixed vectors that do not need the oracle
at collection time. Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "swift-qwen3-tts_amd"))
from oracle import oracle as O  # noqa: E402
from qwen3tts import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    for name in ("tiny-a", "tiny-b"):
        with tempfile.TemporaryDirectory() as d:
            synth.write_checkpoint(d, name, seed=1234)
            om = O.OracleModel(d)
            p = synth.synthetic_prompt(0, n_text=12, text_vocab=1000, im_start=1000, im_end=1001)
            req = O.Request(text_ids=p["text_ids"], target_token_count=12, speaker="aiden", language="english")
            ie, tr, pad = om.prepare_generation_inputs(req)
            greedy = om.generate_codes(req, O.Sampling(temperature=0.0, repetition_penalty=1.05, force_frames=6), keep_logits=True)
            sampled = om.generate_codes(req, O.Sampling(temperature=0.9, top_k=50, seed=42, force_frames=6))
            stages = {}
            pcm, valid = om.codec_decode(greedy.codes, stages)
            np.savez_compressed(
                os.path.join(OUT, f"{name.replace('-', '_')}.npz"),
                text_ids=np.asarray(p["text_ids"], np.int32), input_embeds=ie, trailing=tr, tts_pad=pad,
                greedy_codes=greedy.codes, greedy_talker_logits=np.stack(greedy.talker_logits),
                greedy_cp_logits=np.stack(greedy.cp_logits), sampled_codes=sampled.codes, pcm=pcm, valid=np.int64(valid),
                stage_names=np.array(list(stages.keys())),
                stage_std=np.array([float(v.std()) for v in stages.values()], np.float64),
                stage_shapes=np.array([v.shape for v in stages.values()], np.int64))
            print(name, "codes", greedy.codes[0, :4], "pcm std", float(pcm.std()))


if __name__ == "__main__":
    main()
