"""Writes tests/golden/tokenizer.json (a small Qwen2-style byte-level BPE trained offline on a fixed multilingual corpus)
and tests/golden/tokenizer_cases.json (strings -> ids) with the Hugging Face `tokenizers` wheel: the independent
implementation of the tokenizer.json semantics that pins csrc/tokenizer.cc. The real Qwen tokenizer files are not
available offline; the pre-tokeniser pattern, normaliser, byte-level mapping and special tokens are Qwen2's.
Run from the repo root:  python tests/golden/make_tokenizer_golden.py"""
import json
import os

from tokenizers import Regex, Tokenizer, decoders, models, normalizers, pre_tokenizers, trainers

OUT = os.path.dirname(os.path.abspath(__file__))
PAT = r"""(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+"""

CORPUS = [
    "Hello, world! It's a test of the tokenizer's behaviour; we'll see what it'd do. I'm sure they've done it.",
    "The quick brown fox jumps over the lazy dog. THE QUICK BROWN FOX. He's, SHE'S, they'RE, we'Ll.",
    "你好，世界！今天天气怎么样？我们去公园散步吧。语音合成系统需要分词器。",
    "Ça va très bien, merci. Où est l'école? Noël, garçon, œuvre, crème brûlée.",
    "12345 67,890.12 numbers 3.14159 and ٣٤٥ and ⅓ Ⅻ", "  leading spaces\n\nnewlines\r\n tabs\tand   more  \n",
    "こんにちは世界、元気ですか？ 한국어 텍스트입니다. Привет мир, как дела? Γειά σου κόσμε.",
    "é café ñ composed vs decomposed: é café ñ Å Å 각 각",
    "emoji 😀 👍🏽 and symbols © ® ™ € £ ¥ → ← ≈ ≠ — – … «quotes» “curly” ‘single’",
    "<|im_start|>assistant\nSome text here<|im_end|>\n<|im_start|>user\nA calm, warm voice.<|im_end|>\n",
] * 30

CASES = [
    "", " ", "  ", "\n", " \n ", "a", "Hello", "Hello world", " Hello  world ", "Hello\nworld\n\n", "tabs\t\tand\r\nCRLF\r\n\r\n end",
    "It's they're WE'LL I'M you'D he'S 'tis 'Re 'VE don't can't", "x'sy 'ſ long s", "123 4567 8 ٣٤٥ ⅓Ⅻ 1a2b", "a1 b22 c333",
    "foo.bar,baz!qux?  ...!!!\n\n\nnext", " !@# $%^ &*()_+\n", "trailing spaces   ", "   leading", "mid   dle", "a \n b", "a  \n  b",
    "你好，世界！", "今天天气怎么样？我们去公园。", "こんにちは 世界", "한국어 텍스트 각", "Привет, мир!", "Γειά σου",
    "café café é́ Å Å ñ 각 각 ộ ộ ̈́ क़ ﬁ",
    "😀 👍🏽 👨‍👩‍👧 emoji", "© ® ™ € → ≈ — … «a» “b”", "nbsp here thin space ideographic　space zwsp​here line sep",
    "<|im_start|>assistant\nHello there, it's 2024!<|im_end|>\n<|im_start|>assistant\n", "<|im_start|>user\nA calm voice.<|im_end|>\n",
    "text<|endoftext|>more<|im_start|><|im_end|>", "<|im_start", "< |im_start|>", "MiXeD CaSe WoRdS and ALLCAPS and snake_case_name and kebab-case",
    "https://example.com/path?query=1&x=y#frag user@example.org", "def f(x):\n    return x ** 2  # comment\n", "\t\tindented\n\t\tmore",
    "The year 1999, the price $12.50, 50% off!", "ÀÉÎÕÜ àéîõü ß ẞ İ ı", "́lone combining ́́", "à́̂b",
]


def main():
    tok = Tokenizer(models.BPE())
    tok.normalizer = normalizers.NFC()
    tok.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.Split(Regex(PAT), behavior="isolated", invert=False),
                                                 pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)])
    tok.decoder = decoders.ByteLevel()
    trainer = trainers.BpeTrainer(vocab_size=1200, special_tokens=["<|endoftext|>", "<|im_start|>", "<|im_end|>"],
                                  initial_alphabet=pre_tokenizers.ByteLevel.alphabet(), show_progress=False)
    tok.train_from_iterator(CORPUS, trainer)
    tok.save(os.path.join(OUT, "tokenizer.json"))
    cases = [{"text": t, "ids": tok.encode(t).ids} for t in CASES]
    json.dump(cases, open(os.path.join(OUT, "tokenizer_cases.json"), "w"), ensure_ascii=True, indent=0)
    print("vocab", tok.get_vocab_size(), "cases", len(cases))


if __name__ == "__main__":
    main()
