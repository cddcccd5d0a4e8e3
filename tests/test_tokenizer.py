"""csrc/tokenizer.cc (the engine's Qwen2 byte-level BPE) against the Hugging Face `tokenizers` wheel --
an independent implementation of the tokenizer.json semantics the reference reaches through swift-transformers -- on a
synthetic Qwen2-style tokenizer.json: committed cases (tests/golden) and, when the wheel is importable, live comparisons
on generated strings. Host code only: runs without a GPU."""
import json
import os
import random

import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def native():
    from qwen3tts.model import NativeTokenizer
    t = NativeTokenizer(os.path.join(GOLD, "tokenizer.json"))
    yield t
    t.close()


def test_committed_cases(native):
    cases = json.load(open(os.path.join(GOLD, "tokenizer_cases.json")))
    assert len(cases) > 40
    for c in cases:
        assert native(c["text"]) == c["ids"], repr(c["text"])


def test_live_against_hf_tokenizers(native):
    tokenizers = pytest.importorskip("tokenizers")
    hf = tokenizers.Tokenizer.from_file(os.path.join(GOLD, "tokenizer.json"))
    rng = random.Random(5)
    alphabet = (list("abcXYZ 019\n\t\r'.,!?-_") + ["'s", "'T", "'re", "  ", "\n\n", " \n", "é", "é", "ñ", "你", "好", "世界", "こん", "한", "각",
                "가", "Ж", "д", "😀", "👍🏽", " ", "　", " ", "​", "©", "€", "<|im_start|>", "<|im_end|>",
                "<|", "|>", "٣", "⅓", "ß", "İ", "́", "̣", "̂", "o", "Å"])
    for _ in range(600):
        s = "".join(rng.choice(alphabet) for _ in range(rng.randint(0, 24)))
        assert native(s) == hf.encode(s).ids, repr(s)


def test_loads_next_to_a_checkpoint(tmp_path):
    """from_pretrained picks the tokenizer up from the model folder like AutoTokenizer.from(modelFolder:) (Qwen3.swift:1458);
    the chat-template helper then produces the three id lists of a request."""
    import shutil
    from qwen3tts.model import NativeTokenizer, chat_template_ids
    shutil.copy(os.path.join(GOLD, "tokenizer.json"), tmp_path / "tokenizer.json")
    t = NativeTokenizer(str(tmp_path))
    ids = chat_template_ids(t, "Hello there.", "A calm voice.")
    assert ids["text_ids"][0] == 1 and ids["text_ids"][-3] == 1 and ids["instruct_ids"][0] == 1  # <|im_start|> = 1
    assert ids["target_token_count"] == len(t("Hello there."))
    t.close()


def test_random_unicode_against_hf_tokenizers(native):
    """Random code points from scripts with letters, digits of several kinds, combining marks (NFC), Hangul, spaces of
    every White_Space kind, symbols and emoji."""
    tokenizers = pytest.importorskip("tokenizers")
    hf = tokenizers.Tokenizer.from_file(os.path.join(GOLD, "tokenizer.json"))
    rng = random.Random(1)
    pools = [(0x20, 0x7f), (0xa0, 0x24f), (0x300, 0x36f), (0x370, 0x3ff), (0x400, 0x4ff), (0x590, 0x6ff), (0x900, 0x97f),
             (0xe00, 0xe7f), (0x1100, 0x11ff), (0x1e00, 0x1eff), (0x2000, 0x206f), (0x2150, 0x218f), (0x3000, 0x30ff),
             (0x4e00, 0x4fff), (0xac00, 0xacff), (0xfb00, 0xfb06), (0xff00, 0xffef), (0x1f300, 0x1f64f), (0x1d400, 0x1d4ff)]
    for _ in range(1500):
        s = "".join(chr(rng.randint(*rng.choice(pools))) for _ in range(rng.randint(1, 16)))
        if rng.random() < 0.3:
            s = s + rng.choice([" ", "\n", "  \n", " \t"]) + s[:3]
        assert native(s) == hf.encode(s).ids, [hex(ord(c)) for c in s]


def test_slow_format_files(tmp_path, native):
    """vocab.json + merges.txt (+ tokenizer_config.json added_tokens_decoder): the other layout Qwen checkpoints ship."""
    from qwen3tts.model import NativeTokenizer
    j = json.load(open(os.path.join(GOLD, "tokenizer.json")))
    json.dump(j["model"]["vocab"], open(tmp_path / "vocab.json", "w"), ensure_ascii=False)
    with open(tmp_path / "merges.txt", "w") as f:
        f.write("#version: 0.2\n")
        for m in j["model"]["merges"]:
            f.write((m if isinstance(m, str) else " ".join(m)) + "\n")
    json.dump({"added_tokens_decoder": {str(a["id"]): {"content": a["content"], "special": True} for a in j["added_tokens"]}},
              open(tmp_path / "tokenizer_config.json", "w"))
    t = NativeTokenizer(str(tmp_path))
    for s in ("<|im_start|>assistant\nHello there, it's 2024!<|im_end|>\n", "你好，世界！ café Å 각", "  a \n\n b  "):
        assert t(s) == native(s)
    t.close()


def test_errors():
    from qwen3tts.model import NativeTokenizer, Qwen3TTSError
    with pytest.raises(Qwen3TTSError) as e:
        NativeTokenizer("/nonexistent/dir")
    assert e.value.status == 1 and "Tokenizer not loaded" in str(e.value)


@pytest.mark.gpu
def test_committed_cases_on_the_gpu_box(native):
    """The same committed cases again under the gpu marker, so that the tokeniser (row f2) is exercised by the GPU-box
    run as well: it is host code inside libq3tts_hip.so and needs neither the tokenizers wheel nor the card."""
    test_committed_cases(native)
