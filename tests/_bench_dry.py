"""Stand-in for the engine in tests/test_distributed_cpu.py::test_bench_world2_control_path_end_to_end (bench.py imports it only
under Q3TTS_BENCH_DRY=1): there is no GPU in the CPU test environment, so the N > 1 CONTROL path of bench.py -- shard, broadcast,
checksum, pipelined begin / end, MAX / SUM reduction, the JSON line -- runs with a model that decodes nothing: rows come back
with the requested number of frames. Test infrastructure; never used on a GPU box (the product path fails loudly when the HIP
engine is missing)."""
import time

import numpy as np


class DryModel:
    class _Info:
        weight_bytes = 1 << 20

    class _Res:
        def __init__(self, frames):
            self.codes = np.zeros((frames, 16), np.int32)

    def __init__(self, rank, empty):
        import torch
        g = torch.Generator().manual_seed(5)
        ref = torch.randint(0, 256, (1 << 16,), dtype=torch.uint8, generator=g)
        self.arena_t = torch.zeros_like(ref) if empty else ref
        self.info = self._Info()
        self._frames = 0

    def arena_checksum(self):
        return int(self.arena_t.view(dtype=__import__("torch").int32).to(__import__("torch").int64).sum().item())

    def generate_batch(self, reqs, force_frames=0, **kw):
        self._frames = force_frames
        time.sleep(0.01)
        return [self._Res(force_frames) for _ in reqs]

    def generate_batch_begin(self, reqs, force_frames=0, **kw):
        return (len(reqs), force_frames)

    def generate_batch_end(self, job):
        time.sleep(0.01)
        return [self._Res(job[1]) for _ in range(job[0])]

    def last_timing(self):
        from types import SimpleNamespace
        return SimpleNamespace(prefill_ms=1.0, decode_ms=8.0, codec_ms=2.0, frontend_ms=0.0, frame_steps=max(self._frames, 1),
                               kv_bytes_read=0, first_audio_ms=0.0)
