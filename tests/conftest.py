import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "swift-qwen3-tts_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def bf16_to_f32(b):
    return (np.asarray(b, np.uint16).astype(np.uint32) << 16).view(np.float32)


@pytest.fixture(scope="session")
def ckpt_dirs(tmp_path_factory):
    """Synthetic tiny checkpoints in the reference's on-disk layout (qwen3tts/synth.py)."""
    from qwen3tts import synth
    out = {}
    for name in ("tiny-a", "tiny-b"):
        d = str(tmp_path_factory.mktemp(name.replace("-", "_")))
        synth.write_checkpoint(d, name, seed=1234)
        out[name] = d
    return out


def tiny_request(row=0, n_text=12, n_instruct=0, speaker="aiden", language="english"):
    from qwen3tts import synth
    pr = synth.synthetic_prompt(row, n_text=n_text, n_instruct=n_instruct, text_vocab=1000, im_start=1000, im_end=1001)
    return dict(text_ids=pr["text_ids"], target_token_count=pr["target_token_count"],
                instruct_ids=pr.get("instruct_ids"), speaker=speaker, language=language)
