"""One rank of tests/test_two_rank_gpu.py: a fresh process (nothing of the parent's GPU state), gloo rendezvous on
127.0.0.1, every rank on device 0. Rank 0 loads the checkpoint from disk; rank 1 allocates its weight arena empty
(weights_from_broadcast) and receives it through bench.broadcast_weights -- the job's only collective -- then both decode
their own shard of the rows."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "swift-qwen3-tts_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    ckpt, out_dir, total_rows = sys.argv[1], sys.argv[2], int(sys.argv[3])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch
    import torch.distributed as dist
    import bench
    from conftest import tiny_request
    from qwen3tts import GenerationRequest, Qwen3TTSModel
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        m = Qwen3TTSModel.from_pretrained(ckpt, device=0, max_batch=total_rows, max_frames=32, max_prompt=96,
                                          weights_from_broadcast=(rank != 0))
        ptr, nbytes = m.arena()

        class _Arena:
            __cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}

        arena = torch.as_tensor(_Arena(), device=torch.device("cuda", 0))
        before = int(arena.view(torch.int32)[: nbytes // 4].to(torch.int64).sum().item())
        bench.broadcast_weights(dist, arena, src=0)
        torch.cuda.synchronize()
        after = int(arena.view(torch.int32)[: nbytes // 4].to(torch.int64).sum().item())
        # the library's own device-side checksum (q3tts_model_arena_checksum: 64-bit sum of the unsigned words) against the host's
        usum = int(arena.view(torch.int32)[: nbytes // 4].to(torch.int64).bitwise_and(0xffffffff).sum().item())
        assert m.arena_checksum() == usum % (1 << 64), "q3tts_model_arena_checksum differs from the host-side sum"
        ck = torch.tensor([after], dtype=torch.int64)
        lo_ck, hi_ck = ck.clone(), ck.clone()
        dist.all_reduce(lo_ck, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi_ck, op=dist.ReduceOp.MAX)
        assert int(lo_ck.item()) == int(hi_ck.item()), "weight arena differs between ranks after the broadcast"
        lo, hi = bench.shard_rows(total_rows, rank, world)
        reqs = []
        for row in range(lo, hi):
            r = tiny_request(row=row, n_text=6 + row)
            reqs.append(GenerationRequest(r["text_ids"], r["target_token_count"], r["instruct_ids"], r["speaker"], r["language"]))
        res = m.generate_batch(reqs, temperature=0.9, top_k=40, repetition_penalty=1.05, seed=77, force_frames=12, row_base=lo)
        el, fr = bench.reduce_job_stats(dist, 1.0, sum(r.codes.shape[0] for r in res), torch.device("cpu"))
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), lo=lo, hi=hi, nbytes=nbytes, before=before, after=after, frames=fr,
                 codes=np.stack([r.codes for r in res]), audio=np.stack([r.audio for r in res]))
        m.close()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
