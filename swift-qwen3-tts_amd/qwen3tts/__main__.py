"""Command-line front end with the reference demo's flags and printed accounting
(Sources/Qwen3TTSDemo/main.swift:34-89 flags, :240-313 flow): load time, generation time, real-time factor =
audio seconds / generation seconds (:303), 16-bit WAV output (:134-165), peak memory (:312).

    python -m qwen3tts --model DIR --text "..." [--speaker NAME] [--instruct "..."] [--language auto]
                       [--temperature 0.9] [--top-k 50] [--max-tokens 2048] [--output output.wav]
                       [--reference-audio clip.wav --reference-text "..."]

The model directory must hold the tokenizer files the checkpoints ship (tokenizer.json or vocab.json + merges.txt),
as the reference's postLoadHook requires (Qwen3.swift:1456-1459)."""
from __future__ import annotations

import argparse
import sys
import time


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="qwen3tts", description="Qwen3 TTS Demo - Text to Speech Generation")
    ap.add_argument("--text", "-t", default="Hello, this is a test of the Qwen3 text to speech system.")
    ap.add_argument("--instruct", "-i", default=None)
    ap.add_argument("--speaker", "-s", default=None)
    ap.add_argument("--model", "-m", required=True)
    ap.add_argument("--output", "-o", default="output.wav")
    ap.add_argument("--language", "-l", default="auto")
    ap.add_argument("--temperature", type=float, default=0.9)
    ap.add_argument("--top-k", type=int, default=50)
    ap.add_argument("--max-tokens", type=int, default=2048)
    ap.add_argument("--reference-audio", default=None)
    ap.add_argument("--reference-text", default=None)
    ap.add_argument("--seed", type=int, default=0, help="new: the reference has no seed")
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)

    from . import Qwen3TTSModel
    from .audio import read_wav, write_wav

    print("=== Qwen3 TTS Demo ===")
    print(f'Text: "{a.text}"')
    if a.speaker:
        print(f"Speaker: {a.speaker}")
    if a.instruct:
        print(f'Instruct: "{a.instruct}"')
    if a.reference_audio:
        print(f"Reference Audio: {a.reference_audio}")
    if a.reference_text:
        print(f'Reference Text: "{a.reference_text}"')
    print(f"Model: {a.model}")
    print(f"Output: {a.output}")
    print()
    print("Loading model...")
    t0 = time.time()
    model = Qwen3TTSModel.from_pretrained(a.model, device=a.device, max_batch=1, max_frames=min(a.max_tokens, 2048) + 8,
                                          max_prompt=1024)
    print(f"Model loaded in {time.time() - t0:.2f}s")
    print()
    print("Generating audio...")
    t1 = time.time()
    if a.reference_audio and a.reference_text:
        if not model.supports_voice_cloning:
            print("Error: This model doesn't support voice cloning. Use a Base model.")
            return 1
        print("Mode: Voice Cloning")
        sr, ref = read_wav(a.reference_audio)
        if sr != 24000:
            print(f"Warning: Reference audio is {sr}Hz, expected 24000Hz. Results may vary.")
        audio = model.generate_voice_clone(text=a.text, reference_audio=ref, reference_text=a.reference_text,
                                           language=a.language, temperature=a.temperature, top_k=a.top_k,
                                           max_tokens=a.max_tokens, seed=a.seed)
    else:
        audio = model.generate(text=a.text, speaker=a.speaker, instruct=a.instruct, language=a.language,
                               temperature=a.temperature, top_k=a.top_k, max_tokens=a.max_tokens, seed=a.seed)
    gen = time.time() - t1
    dur = audio.size / float(model.sample_rate)
    print(f"Generated {audio.size} samples ({dur:.2f}s audio)")
    print(f"Generation time: {gen:.2f}s")
    print(f"Real-time factor: {dur / gen:.2f}x")
    print()
    write_wav(a.output, audio, model.sample_rate)
    print(f"Saved to: {a.output}")
    print(f"Peak memory: {model.last_info.peak_memory_usage if model.last_info else 0.0:.2f} GB")
    model.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
