"""The N > 1 path of bench.py on CPU: world_size 2 over gloo. Covers the batch-shard partition, the
one-time weight broadcast (the job's only collective) and the timing reduction."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # 1. weight arena: rank 0 "loaded" it, replicas allocated it empty (weights_from_broadcast)
        g = torch.Generator().manual_seed(5)
        ref = torch.randint(0, 256, (1 << 16,), dtype=torch.uint8, generator=g)
        arena = ref.clone() if rank == 0 else torch.zeros_like(ref)
        bench.broadcast_weights(dist, arena, src=0)
        assert torch.equal(arena, ref)
        # 2. batch shard of 2 x 32 rows: contiguous, disjoint, complete
        lo, hi = bench.shard_rows(32 * world, rank, world)
        rows = torch.zeros(32 * world, dtype=torch.int32)
        rows[lo:hi] = 1
        dist.all_reduce(rows)
        assert bool((rows == 1).all())
        # the same request builder as the GPU run: row r gets prompt seed 7 + r whatever the rank
        reqs = bench.build_requests("1.7b", lo, hi, 32, 16)
        assert len(reqs) == 32 and len(reqs[0].text_ids) == 3 + 32 + 5 and len(reqs[0].instruct_ids) == 3 + 16 + 2
        # 3. timing contract: MAX of elapsed, SUM of frames
        el, fr = bench.reduce_job_stats(dist, 1.0 + rank, 100 * (rank + 1), torch.device("cpu"))
        assert el == float(world) and fr == 100 * world * (world + 1) // 2
        np.save(os.path.join(out_dir, f"ok{rank}.npy"), np.array([lo, hi]))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    spans = [np.load(tmp_path / f"ok{r}.npy") for r in range(world)]
    assert spans[0][1] == spans[1][0] == 32


def test_bench_world2_control_path_end_to_end(tmp_path):
    """bench.py launched the way the driver launches it for N = 2 (torch.distributed.run, one process per rank) under gloo,
    BASELINE configs[4]'s preset (64 rows per rank): the whole N > 1 branch -- shard, weight broadcast + checksum, pipelined
    begin / end, MAX-over-ranks elapsed, SUM-over-ranks frames, ONE JSON line from rank 0 -- with the engine replaced by a
    stand-in that decodes nothing (Q3TTS_BENCH_DRY=1: there is no GPU here; the GPU box runs the real thing)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, Q3TTS_BENCH_DRY="1", Q3TTS_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--preset", "0.6b-q4", "--frames", "8"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["unit"] == "frames/s"
    assert d["config"]["batch_per_gpu"] == 64 and d["config"]["parallelism"] == "batch-shard x2"
    assert "gloo" in d["config"]["weight_broadcast"]
    frames = 2 * 64 * 8 * 3                               # SUM over ranks of rows x frames x steps
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 * d["steps"] - frames) < 1e-6 * frames
    assert d["vs_baseline"] is None and d["higher_is_better"] is True
