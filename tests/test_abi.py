"""The C-ABI library loads and exports every symbol include/q3tts.h declares (no compute, no GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "q3tts.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(q3tts_[a-z0-9_]+)\s*\(", src)) - {"q3tts_event_cb"})


def test_header_declares_the_surface():
    fns = declared_functions()
    for must in ("q3tts_model_load", "q3tts_model_free", "q3tts_generate", "q3tts_result_free", "q3tts_codec_decode",
                 "q3tts_last_error", "q3tts_model_get_info", "q3tts_model_arena"):
        assert must in fns


def test_library_exports_every_declared_symbol():
    from qwen3tts import _lib
    L = _lib.lib()  # raises loudly if the HIP extension was not built
    missing = [f for f in declared_functions() if not hasattr(L, f)]
    assert not missing, f"symbols declared in q3tts.h but not exported: {missing}"


def test_defaults_match_the_reference():
    from qwen3tts import _lib
    L = _lib.lib()
    s = _lib.Sampling()
    L.q3tts_default_sampling(ctypes.byref(s))
    # generate() defaults, Qwen3.swift:1296-1299
    assert (round(s.temperature, 3), s.top_k, s.top_p, round(s.repetition_penalty, 3)) == (0.9, 50, 1.0, 1.05)
    o = _lib.LoadOpts()
    L.q3tts_default_load_opts(ctypes.byref(o))
    assert o.max_frames == 2048 and o.use_graph == 1


def test_no_gpu_fails_loudly(tmp_path):
    """Without a HIP device the product path refuses to load (no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from qwen3tts import Qwen3TTSError, Qwen3TTSModel, synth
    d = str(tmp_path / "m")
    synth.write_checkpoint(d, "tiny-a")
    with pytest.raises(Qwen3TTSError) as e:
        Qwen3TTSModel.from_pretrained(d)
    assert e.value.status == 7 and "no CPU fallback" in str(e.value)


def test_product_never_references_the_oracle():
    pkg = os.path.join(ROOT, "swift-qwen3-tts_amd")
    bad = []
    for dp, _, fs in os.walk(pkg):
        if os.sep + "build" in dp:
            continue
        for f in fs:
            if f.endswith((".py", ".cc", ".h", ".hip")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle|oracle/_build|libq3tts_oracle", txt, flags=re.M):
                    bad.append(os.path.join(dp, f))
    assert not bad, f"product files reference the oracle: {bad}"


def test_ctypes_mirror_has_the_headers_layout(tmp_path):
    """Every struct of include/q3tts.h against its ctypes mirror (qwen3tts/_lib.py): size and the offset of every field, as a C
    compiler lays the header out (which also shows that the header is plain C). A field appended on one side only -- the way
    q3tts_timing grew in round 4 -- fails here and not as a garbled read on the GPU box."""
    import subprocess
    from qwen3tts import _lib
    pairs = [("q3tts_load_opts", _lib.LoadOpts), ("q3tts_comm_id", _lib.CommId), ("q3tts_model_info", _lib.ModelInfo),
             ("q3tts_request", _lib.Request), ("q3tts_sampling", _lib.Sampling), ("q3tts_gen_info", _lib.GenInfo),
             ("q3tts_event", _lib.Event), ("q3tts_result", _lib.Result), ("q3tts_timing", _lib.Timing)]
    lines = ['#include <stddef.h>', '#include <stdio.h>', '#include "q3tts.h"', 'int main(void) {']
    for cname, cls in pairs:
        lines.append('  printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, *_ in cls._fields_:
            lines.append('  printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    c_layout = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, cls in pairs:
        assert int(c_layout[cname]) == ctypes.sizeof(cls), cname
        for fname, *_ in cls._fields_:
            assert int(c_layout["%s.%s" % (cname, fname)]) == getattr(cls, fname).offset, "%s.%s" % (cname, fname)
    # and no C field is missing from a mirror: the last mirrored field ends where the struct does (up to tail padding)
    for cname, cls in pairs:
        fname, ftype = cls._fields_[-1][0], cls._fields_[-1][1]
        assert getattr(cls, fname).offset + ctypes.sizeof(ftype) > ctypes.sizeof(cls) - 8, cname
