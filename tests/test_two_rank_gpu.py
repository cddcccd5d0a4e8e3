"""SURVEY 8(e) on the hardware this pool offers (one card): two fresh processes -- started before either touches the GPU --
share device 0 over gloo. Rank 0 loads the checkpoint, rank 1 receives the weight arena through bench.broadcast_weights
(the only collective of the job), checksums agree, and each rank's shard of the rows equals the single-process result for
the same global row indices bit for bit (prompts AND random streams are keyed by the global row index)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import tiny_request

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.gpu
def test_two_ranks_one_device_equal_the_single_process_rows(ckpt_dirs, tmp_path):
    from qwen3tts import GenerationRequest, Qwen3TTSModel
    total, world = 4, 2
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_two_rank_worker.py"), ckpt_dirs["tiny-b"], str(tmp_path),
                                       str(total)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode(errors="replace"))
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    r0, r1 = (np.load(tmp_path / f"rank{r}.npz") for r in range(world))
    assert (int(r0["lo"]), int(r0["hi"]), int(r1["lo"]), int(r1["hi"])) == (0, 2, 2, 4)
    assert int(r0["nbytes"]) == int(r1["nbytes"]) and int(r0["after"]) == int(r1["after"]) == int(r0["before"])
    assert int(r1["before"]) != int(r1["after"])          # rank 1 really started without the weights
    assert int(r0["frames"]) == int(r1["frames"]) == total * 12
    # the same four rows in ONE process
    m = Qwen3TTSModel.from_pretrained(ckpt_dirs["tiny-b"], max_batch=total, max_frames=32, max_prompt=96)
    try:
        reqs = []
        for row in range(total):
            r = tiny_request(row=row, n_text=6 + row)
            reqs.append(GenerationRequest(r["text_ids"], r["target_token_count"], r["instruct_ids"], r["speaker"], r["language"]))
        want = m.generate_batch(reqs, temperature=0.9, top_k=40, repetition_penalty=1.05, seed=77, force_frames=12)
    finally:
        m.close()
    got_codes = np.concatenate([r0["codes"], r1["codes"]])
    got_audio = np.concatenate([r0["audio"], r1["audio"]])
    for i, w in enumerate(want):
        assert (got_codes[i] == w.codes).all() and (got_audio[i] == w.audio).all(), i
    assert len({tuple(c[:, 0]) for c in got_codes}) == total   # and the rows are not copies of each other


@pytest.mark.gpu
def test_native_broadcast_on_a_single_rank_communicator(ckpt_dirs):
    """q3tts_comm_get_unique_id / q3tts_model_broadcast / q3tts_model_arena_checksum (csrc/comm.cc): RCCL opened at run time from
    the C ABI, a communicator of one rank on this box's one GPU, the arena broadcast onto itself. What this pins on hardware:
    the library finds RCCL, the call sequence (ncclCommInitRank -> ncclBroadcast on the arena -> sync -> ncclCommDestroy) is
    accepted and the arena is unchanged by it (checksum, and the rows a generate call produces). The checksum itself is
    compared with a host-side sum of the same bytes in _two_rank_worker.py, where torch owns the process's GPU runtime."""
    from qwen3tts import GenerationRequest, Qwen3TTSModel
    m = Qwen3TTSModel.from_pretrained(ckpt_dirs["tiny-b"], max_batch=2, max_frames=16, max_prompt=64)
    e = Qwen3TTSModel.from_pretrained(ckpt_dirs["tiny-b"], max_batch=2, max_frames=16, max_prompt=64, weights_from_broadcast=True)
    try:
        r = tiny_request(row=0, n_text=7)
        reqs = [GenerationRequest(r["text_ids"], r["target_token_count"], r["instruct_ids"], r["speaker"], r["language"])]
        kw = dict(temperature=0.9, top_k=40, seed=3, force_frames=6)
        want = m.generate_batch(reqs, **kw)[0]
        before = m.arena_checksum()
        assert before != 0 and e.arena_checksum() != before      # a replica starts without the weights
        cid = Qwen3TTSModel.comm_unique_id()
        assert len(cid) == 128 and any(cid)
        m.broadcast_weights(cid, 0, 1, 0)
        assert m.arena_checksum() == before
        got = m.generate_batch(reqs, **kw)[0]
        assert (got.codes == want.codes).all() and (got.audio == want.audio).all()
    finally:
        m.close()
        e.close()


@pytest.mark.gpu
def test_bench_two_ranks_on_one_device_agree_to_fall_back_from_the_native_broadcast(tmp_path):
    """bench.py itself, launched the way the driver launches it for N > 1 (torch.distributed.run, one process per rank), on the one
    card this pool offers: both ranks share device 0 over gloo and ask for the library's own RCCL broadcast. RCCL must refuse a
    communicator with two ranks on one GPU; what this run pins ON HARDWARE is everything around that refusal -- the pre-flight
    (every rank can open RCCL), the 128-byte id travelling through torch, the every-rank agreement that the native path failed,
    the torch fallback over the zero-copy arena view, the checksum agreement, the sharded rows, the MAX / SUM reduction and ONE
    JSON line from rank 0. (The RCCL broadcast proper is covered on a one-rank communicator above; > 1 GPU is the driver's.)"""
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, Q3TTS_BENCH_ONE_DEVICE="1", Q3TTS_BENCH_BACKEND="gloo", Q3TTS_BENCH_BROADCAST="native-force",
               Q3TTS_BENCH_CKPT=str(tmp_path / "ckpt"), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--preset", "0.6b", "--batch", "4", "--frames", "8", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak"
    assert d["config"]["weight_broadcast"].startswith("torch.distributed (gloo)")     # the agreed fallback
    assert "native broadcast failed" in r.stderr                                       # ... after RCCL refused on both ranks
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 2 * 4 * 8) < 1e-6 * 64            # both ranks' frames are in the aggregate
