#!/usr/bin/env python3
"""What kind of neighbour slows the frame loop? The AR loop of the headline workload (1.7B, batch 32, F forced frames, codec
decode excluded from the timed part) is run alone and then beside a synthetic workload on a second (torch) stream of the same
process: matrix-core bound, HBM streaming, L2-resident streaming, or a stream of tiny launches. Prints the loop's ms per frame
step in each case (engine timing: HIP events on its own stream).   usage: neighbour_probe.py [frames]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "swift-qwen3-tts_amd"))
import bench  # noqa: E402
from qwen3tts import Qwen3TTSModel  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ckpt = bench.ensure_checkpoint("1.7b", 0, None)
m = Qwen3TTSModel.from_pretrained(ckpt, max_batch=32, max_frames=F + 8, max_prompt=128)
reqs = bench.build_requests("1.7b", 0, 32, 32, 16)
kw = dict(temperature=0.9, top_k=50, top_p=1.0, repetition_penalty=1.05, seed=1234, force_frames=F)
m.generate_batch(reqs, **kw)  # warm-up (graph capture)

dev = torch.device("cuda:0")
if os.environ.get("PROBE_UNMASKED"):
    side = torch.cuda.Stream(device=dev)
else:  # the same CU mask as the engine's overlapped decode stream: the first 128 mask bits = half of every XCD
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    words = (ncu + 31) // 32
    maskv = (ctypes.c_uint32 * words)()
    for i in range(int(os.environ.get("PROBE_CUS", ncu // 2))):
        maskv[i // 32] |= 1 << (i % 32)
    sp = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(sp), ctypes.c_uint32(words), maskv)
    assert rc == 0, rc
    side = torch.cuda.ExternalStream(sp.value, device=dev)
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
big = torch.empty(1 << 30, device=dev, dtype=torch.uint8)       # 1 GiB: HBM streaming
big2 = torch.empty_like(big)
small = torch.empty(4 << 20, device=dev, dtype=torch.uint8)     # 4 MiB: stays in the L2s / Infinity Cache
small2 = torch.empty_like(small)
tiny = torch.zeros(64, device=dev)


def neighbour(kind, seconds):
    """enqueue roughly `seconds` of work on the side stream (sizes calibrated below)"""
    with torch.cuda.stream(side):
        if kind == "mfma":
            for _ in range(int(seconds / cal["mfma"])):
                torch.mm(a, b)
        elif kind == "hbm":
            for _ in range(int(seconds / cal["hbm"])):
                big2.copy_(big)
        elif kind == "l2":
            for _ in range(int(seconds / cal["l2"])):
                small2.copy_(small)
        elif kind == "launches":
            for _ in range(int(seconds / cal["launches"])):
                tiny.add_(1.0)


cal = {}
for kind, fn in (("mfma", lambda: torch.mm(a, b)), ("hbm", lambda: big2.copy_(big)), ("l2", lambda: small2.copy_(small)),
                 ("launches", lambda: tiny.add_(1.0))):
    with torch.cuda.stream(side):
        fn()
        side.synchronize()
        t0 = time.time()
        n = 20 if kind != "launches" and kind != "l2" else 2000
        for _ in range(n):
            fn()
        side.synchronize()
        cal[kind] = (time.time() - t0) / n
print("neighbour unit times (s):", {k: round(v, 6) for k, v in cal.items()}, flush=True)

for kind in ("none", "mfma", "hbm", "l2", "launches", "none"):
    if kind != "none":
        neighbour(kind, 0.9)
    t0 = time.time()
    m.generate_batch(reqs, **kw)
    wall = time.time() - t0
    tm = m.last_timing()
    side.synchronize()
    print(f"{kind:9s} frame step {tm.decode_ms / F:.3f} ms   (AR {tm.decode_ms:.1f} ms, prefill {tm.prefill_ms:.1f} ms, codec {tm.codec_ms:.1f} ms, wall {1e3 * wall:.0f} ms)",
          flush=True)
m.close()
