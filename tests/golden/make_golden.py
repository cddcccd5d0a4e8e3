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
    voice_clone()


def voice_clone():
    """tiny_base.npz: voice-clone vectors from the oracle on the tiny Base checkpoint."""
    with tempfile.TemporaryDirectory() as d:
        synth.write_checkpoint(d, "tiny-base", seed=4321)
        om = O.OracleModel(d)
        p = synth.synthetic_prompt(0, n_text=10, text_vocab=1000, im_start=1000, im_end=1001)
        audio = synth.synthetic_reference_audio(0, 1.0)
        st = {}
        codes = om.codec_encode(audio, st)
        sst = {}
        xvec = om.speaker_embedding(audio, sst)
        req = O.Request(text_ids=p["text_ids"], target_token_count=10, language="english", ref_audio=audio,
                        ref_text_ids=p["ref_text_ids"])
        ie, tr, pad, _ = om.prepare_icl_generation_inputs(req)
        s = O.Sampling(temperature=0.0, repetition_penalty=1.5, force_frames=5)
        pcm, gen, _ = om.generate_voice_clone(req, s)
        np.savez_compressed(
            os.path.join(OUT, "tiny_base.npz"),
            text_ids=np.asarray(p["text_ids"], np.int32), ref_text_ids=np.asarray(p["ref_text_ids"], np.int32),
            ref_codes=codes, rvq_gaps=st["gaps"], seanet=st["seanet"], transformer=st["transformer"],
            downsample=st["downsample"], mel=sst["mel"], pooled=sst["pooled"], xvec=xvec, input_embeds=ie, tts_pad=pad,
            clone_codes=gen.codes, clone_pcm=pcm)
        print("tiny-base ref codes", codes[:3, :4].tolist(), "clone codes", gen.codes[0, :4], "pcm", pcm.shape)


if __name__ == "__main__":
    if "--voice-clone-only" in sys.argv:
        voice_clone()
    else:
        main()
