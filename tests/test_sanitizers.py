"""AddressSanitizer / UndefinedBehaviorSanitizer over the code that can run without a GPU ("race detection /
sanitizers": the reference has none; GPU ASan and XNACK are not available on this pool, so the sanitizers run on the CPU builds).

  * the product's HOST-ONLY sources -- the tokeniser (csrc/tokenizer.cc), the JSON reader (csrc/json.h) and the safetensors reader
    (csrc/safetensors.h): everything that parses a file a user hands to q3tts_model_load / q3tts_tokenizer_load -- compiled with
    g++ -fsanitize=address,undefined,float-cast-overflow into tests/native/_build/host_san and fed the committed tokenizer cases
    plus seeded manglings of config.json, a safetensors header, tokenizer.json and input text. Every input ends in a result or
    in a q3::Error; a sanitizer report fails the test (the first run of this file found four: header numbers cast to integers
    without a range check, `8 + header_length` wrapping, unbounded nesting).
  * the C oracle (oracle/q3tts_oracle.c) built the same way and driven through the golden-vector and block tests in a child
    interpreter with the sanitizer runtime preloaded.
"""
import json
import os
import shutil
import struct
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NATIVE = os.path.join(ROOT, "tests", "native")
SAN = ["-fsanitize=address,undefined,float-cast-overflow", "-fno-sanitize-recover=undefined,float-cast-overflow"]
# halt on the first report with a non-zero exit; leaks are not what is being looked for (CPython itself "leaks" at exit)
SAN_ENV = {"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0:exitcode=97", "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1:exitcode=98"}


def _have_sanitizers():
    if not shutil.which("g++") or not shutil.which("gcc"):
        return False
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return os.path.isabs(asan) and os.path.exists(asan)


pytestmark = pytest.mark.skipif(not _have_sanitizers(), reason="gcc / g++ with libasan are not installed")


@pytest.fixture(scope="module")
def host_san():
    out = os.path.join(NATIVE, "_build", "host_san")
    srcs = [os.path.join(NATIVE, "host_san.cc"), os.path.join(ROOT, "swift-qwen3-tts_amd", "csrc", "tokenizer.cc")]
    deps = srcs + [os.path.join(ROOT, "swift-qwen3-tts_amd", "csrc", h) for h in ("json.h", "safetensors.h", "tokenizer.h", "common.h", "config.h")]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        # common.h includes the HIP runtime header for the error macro; with g++ that needs the platform define and nothing else
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", *SAN, "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                               "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "swift-qwen3-tts_amd", "csrc"),
                               *srcs, "-o", out])
    return out


def _run(cmd, **kw):
    r = subprocess.run(cmd, capture_output=True, text=True, env={**os.environ, **SAN_ENV}, timeout=600, **kw)
    assert r.returncode == 0, "exit %d\n%s\n%s" % (r.returncode, r.stdout[-2000:], r.stderr[-6000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-6000:]
    return r.stdout


def _counts(out):
    w = out.split()
    assert w[0] == "ok" and w[2] == "rejected", out
    return int(w[1]), int(w[3])


def test_committed_tokenizer_cases_under_sanitizers(host_san, tmp_path):
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "tokenizer_cases.json")))
    blob = struct.pack("<I", len(cases))
    for c in cases:
        b = c["text"].encode("utf-8")
        blob += struct.pack("<I", len(b)) + b
    p = tmp_path / "cases.bin"
    p.write_bytes(blob)
    out = _run([host_san, "tok", os.path.join(ROOT, "tests", "golden", "tokenizer.json"), str(p)])
    lines = out.split("\n")[:len(cases)]
    for c, line in zip(cases, lines):
        assert [int(x) for x in line.split()] == c["ids"], c["text"]


def test_mangled_config_json_is_parsed_or_rejected(host_san, ckpt_dirs):
    ok, rejected = _counts(_run([host_san, "jsonfuzz", os.path.join(ckpt_dirs["tiny-a"], "config.json"), "1", "4000"]))
    assert ok + rejected == 4000 and ok > 100 and rejected > 1000   # both outcomes are exercised


def test_mangled_model_configs_validate_or_are_rejected(host_san, ckpt_dirs):
    """ModelConfig::validate (csrc/config.h) is where a damaged config.json stops: after it, every size is a size, there is one
    intermediate size per layer and every special id lies inside the table it indexes -- the loader and the engine index with
    these numbers without looking again (an id beyond the codec vocabulary would be a gather outside a table on the GPU)."""
    d = ckpt_dirs["tiny-a"]
    ok, rejected = _counts(_run([host_san, "cfgfuzz", os.path.join(d, "config.json"), os.path.join(d, "speech_tokenizer", "config.json"), "11", "4000"]))
    assert ok + rejected == 4000 and ok > 100 and rejected > 1000


def test_mangled_safetensors_headers_are_opened_or_rejected(host_san, tmp_path):
    # a small file of its own (every dtype the reader knows, an empty tensor, a scalar, metadata): the driver rewrites the file
    # once per case, and a real checkpoint's megabytes would make that the slowest test of the CPU suite
    import numpy as np
    from safetensors.numpy import save_file
    rng = np.random.default_rng(0)
    tensors = {
        "talker.model.layers.0.mlp.gate_proj.weight": rng.standard_normal((24, 16)).astype(np.float32),
        "decoder.decoder.1.block.1.conv.weight": rng.standard_normal((8, 3, 4)).astype(np.float16),
        "text_token_map": rng.integers(0, 1000, (40,), dtype=np.int32),
        "ids64": rng.integers(0, 1 << 40, (5,), dtype=np.int64),
        "packed.weight": rng.integers(0, 1 << 32, (6, 4), dtype=np.uint32),
        "bytes": rng.integers(0, 256, (33,), dtype=np.uint8),
        "empty": np.zeros((0, 7), np.float32),
        "scalar": np.array(3.5, np.float32),
    }
    src = tmp_path / "src.safetensors"
    save_file(tensors, str(src), metadata={"format": "mlx", "note": "x" * 40})
    work = tmp_path / "work"
    work.mkdir()
    ok, rejected = _counts(_run([host_san, "stfuzz", str(src), str(work), "7", "3000"]))
    assert ok + rejected == 3000 and ok > 50 and rejected > 1500


def test_mangled_tokenizer_json_loads_or_is_rejected(host_san, tmp_path):
    ok, rejected = _counts(_run([host_san, "tokfuzz", os.path.join(ROOT, "tests", "golden", "tokenizer.json"), str(tmp_path), "3", "600"]))
    assert ok + rejected == 600 and rejected > 300


def test_text_that_is_not_utf8_is_encoded_or_rejected(host_san):
    ok, rejected = _counts(_run([host_san, "textfuzz", os.path.join(ROOT, "tests", "golden", "tokenizer.json"), "5", "5000"]))
    assert ok + rejected == 5000


def test_oracle_c_restatement_under_sanitizers(tmp_path):
    """The golden vectors and the block tests through an ASan / UBSan build of oracle/q3tts_oracle.c (child interpreter, sanitizer
    runtime preloaded): the checker itself reads and writes inside its buffers on the cases the parity tests lean on."""
    src = os.path.join(ROOT, "oracle", "q3tts_oracle.c")
    lib = os.path.join(ROOT, "oracle", "_build", "libq3tts_oracle_san.so")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(lib), exist_ok=True)
        subprocess.check_call(["gcc", "-O1", "-g", *SAN, "-march=x86-64-v3", "-fopenmp", "-ffp-contract=off", "-fno-math-errno", "-fPIC",
                               "-std=gnu11", "-shared", "-fvisibility=hidden", "-o", lib, src, "-lm"])
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    env = {**os.environ, **SAN_ENV, "LD_PRELOAD": asan, "Q3TTS_ORACLE_LIB": lib, "OMP_NUM_THREADS": "4"}
    probe = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from oracle import oracle; print(oracle.lib()._name)" % ROOT],
                           capture_output=True, text=True, env=env, timeout=600)
    assert probe.returncode == 0 and probe.stdout.strip() == lib, probe.stderr[-3000:]
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_golden.py"), os.path.join(ROOT, "tests", "test_oracle_blocks.py")],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-6000:]
