#!/usr/bin/env python3
"""bench.py -- headline benchmark: codec tokens/s + real-time factor @24 kHz (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[2], the config the metric is quoted on): Qwen3-TTS-1.7B-VoiceDesign
bf16, batch 32 fixed-length synthetic prompts per GPU, 200 codec frames per utterance (EOS masked),
reference sampling defaults (T=0.9, top-k 50, rep-penalty 1.05), prompt assembly + prefill + AR
decode (hipGraph frame step) + codec decode to 24 kHz PCM. One "step" = one such batch end to end.
Utterances are batch-sharded: every rank holds a full replica (weights broadcast over RCCL at load)
and decodes its own 32 rows; there is no data-path collective, scaling "weak".

Prints ONE JSON line on rank 0. `roofline` describes the hipGraph-captured frame step (the unit the
AR loop launches): algorithmic bytes = distinct weight bytes + KV bytes read, over the average
frame-step duration measured with HIP events on the engine's stream. `cpu_baseline` times the oracle
(oracle/, a C restatement -- kind "port") on the host cores for a bounded sample of the same
workload (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "swift-qwen3-tts_amd"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (16x the 157.3 TFLOP/s fp32 matrix rate, MI355X_MICROARCH.md)
FRAME_SECONDS = 0.08   # 1 codec frame = 1920 samples @ 24 kHz


def kernel_sources_sha16() -> str:
    """Hash of the sources that decide what a frame step launches (kernels + engine): ties a committed PMC measurement to a build."""
    import glob
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "swift-qwen3-tts_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(base, "kernels", "*")) + [os.path.join(base, "engine.cc")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def shard_rows(total_rows: int, rank: int, world: int):
    """Contiguous batch shard of `total_rows` utterances for `rank`."""
    per = total_rows // world
    rem = total_rows % world
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)


def broadcast_weights(dist, arena_tensor, src: int = 0) -> None:
    """The only collective of the whole job: rank `src` holds the loaded weight arena, replicas receive
    it in one broadcast (RCCL over xGMI on the GPU box, gloo in the CPU tests). No per-step collectives."""
    dist.broadcast(arena_tensor, src=src)


def reduce_job_stats(dist, elapsed_s: float, frames_done: int, device):
    """Timing contract: elapsed = MAX over ranks, frames = SUM over ranks."""
    import torch
    t = torch.tensor([elapsed_s], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    f = torch.tensor([frames_done], device=device, dtype=torch.int64)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    return float(t.item()), int(f.item())


def build_requests(preset: str, lo: int, hi: int, n_text: int, n_instruct: int, ref_seconds: float = 3.0):
    from qwen3tts import GenerationRequest, synth
    reqs = []
    for row in range(lo, hi):
        p = synth.synthetic_prompt(row, n_text=n_text, n_instruct=n_instruct)
        if preset.endswith("-base"):  # BASELINE configs[4]: voice clone, 3.0 s synthetic reference clip per row
            reqs.append(GenerationRequest(p["text_ids"], p["target_token_count"], None, None, "english",
                                          ref_audio=synth.synthetic_reference_audio(row, ref_seconds),
                                          ref_text_ids=p["ref_text_ids"]))
            continue
        reqs.append(GenerationRequest(p["text_ids"], p["target_token_count"], p.get("instruct_ids"),
                                      "aiden" if preset.startswith("0.6b") else None, "english"))
    return reqs


def _tensor_files(d: str):
    out = {}
    for root, _, files in os.walk(d):
        for f in files:
            if f.endswith(".safetensors") or f.endswith(".json"):
                p = os.path.join(root, f)
                out[os.path.relpath(p, d)] = os.path.getsize(p)
    return out


def ensure_checkpoint(preset: str, rank: int, dist) -> str:
    """Synthetic checkpoint directory shared by bench.py and tests/test_full_size.py. The `.complete` marker is written
    last and lists every file with its size; a directory whose files do not match it (a writer that died half way, or
    two writers) is rebuilt instead of being trusted."""
    from qwen3tts import synth
    d = os.environ.get("Q3TTS_BENCH_CKPT", f"/tmp/q3tts_synth_{preset}_seed1234")
    marker = os.path.join(d, ".complete")
    if rank == 0:
        ok = False
        if os.path.exists(marker):
            try:
                ok = json.load(open(marker)) == _tensor_files(d)
            except ValueError:
                ok = False
        if not ok:
            if os.path.exists(marker):
                os.remove(marker)
            synth.write_checkpoint(d, preset, seed=1234)
            tmp = marker + f".tmp{os.getpid()}"
            json.dump(_tensor_files(d), open(tmp, "w"))
            os.replace(tmp, marker)
    if dist is not None:
        dist.barrier()
    return d


def cpu_baseline(ckpt: str, preset: str, n_text: int, n_instruct: int, frames: int) -> dict:
    """Oracle (CPU restatement, OpenMP) on a bounded sample: batch 1, same prompt shape, `frames`
    frames, greedy, end to end (prompt assembly + prefill + AR loop + codec decode)."""
    from oracle import oracle as O
    from qwen3tts import synth
    om = O.OracleModel(ckpt)
    p = synth.synthetic_prompt(0, n_text=n_text, n_instruct=n_instruct)
    if preset.endswith("-base"):
        req = O.Request(text_ids=p["text_ids"], target_token_count=p["target_token_count"], language="english",
                        ref_audio=synth.synthetic_reference_audio(0, 3.0), ref_text_ids=p["ref_text_ids"])
    else:
        req = O.Request(text_ids=p["text_ids"], target_token_count=p["target_token_count"],
                        instruct_ids=p.get("instruct_ids"), speaker="aiden" if preset.startswith("0.6b") else None,
                        language="english")
    t0 = time.time()
    tr = om.generate_codes(req, O.Sampling(temperature=0.0, force_frames=frames))
    t1 = time.time()
    om.codec_decode(tr.codes if tr.ref_codes is None else np.concatenate([tr.ref_codes.T, tr.codes], 0))
    t2 = time.time()
    return {"value": frames / (t2 - t0), "unit": "frames/s", "cores": int(O.lib().o_num_threads()), "kind": "port",
            "sample": f"oracle (C restatement, OpenMP), batch 1, {frames} frames greedy end-to-end: "
                      f"AR {t1 - t0:.1f}s + codec {t2 - t1:.1f}s; MLX-CPU reference unavailable offline"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--preset", default="1.7b", choices=["1.7b", "0.6b", "0.6b-q4", "0.6b-base", "1.7b-base"],
                    help="*-base: voice clone (BASELINE configs[4]): 3 s reference clip per row, repetition penalty 1.5")
    ap.add_argument("--batch", type=int, default=0, help="utterances per GPU (default 32; 16 for *-base; 64 for 0.6b-q4)")
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--n-text", type=int, default=32)
    ap.add_argument("--cpu-frames", type=int, default=75, help="frames of the bounded CPU-oracle sample: 75 = the reference's floor (SURVEY 8d), about 20 s of CPU work on the box's 16-core share")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--greedy", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay (profiling)")
    ap.add_argument("--streams", type=int, default=0, help="lanes per GPU (0 = engine default)")
    ap.add_argument("--no-pipeline", action="store_true", help="one batch at a time (no codec / AR overlap between steps)")
    ap.add_argument("--no-streaming", action="store_true", help="skip the streamed-audio batch behind the timed region (profiling runs)")
    ap.add_argument("--codec-cus", type=int, default=0, help="q3tts_load_opts.codec_overlap_cus (0 = engine default, -1 = no mask)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the N > 1 path on a one-GPU box: every rank on device 0, gloo instead of RCCL (which needs one
    # device per rank). The driver's runs use neither variable.
    if os.environ.get("Q3TTS_BENCH_ONE_DEVICE") == "1":
        local = 0
    backend = os.environ.get("Q3TTS_BENCH_BACKEND", "nccl")
    dist = None
    dry = os.environ.get("Q3TTS_BENCH_DRY") == "1"  # no GPU: the control path only (_DryModel)
    if world > 1:
        import torch
        import torch.distributed as dist_mod
        if not dry:
            torch.cuda.set_device(local)
        if backend == "nccl":
            dist_mod.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist_mod.init_process_group(backend)
        dist = dist_mod
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N for N > 1"

    from qwen3tts import Qwen3TTSModel
    n_instruct = 16 if args.preset == "1.7b" else 0
    clone = args.preset.endswith("-base")
    B = args.batch or (16 if clone else 64 if args.preset == "0.6b-q4" else 32)
    rep = 1.5 if clone else 1.05  # generateVoiceClone's default (Qwen3.swift:1017)
    if dry:
        # the stand-in lives with the tests (tests/_bench_dry.py): it decodes nothing and exists so that this file's N > 1 control
        # path can run under gloo on a machine without a GPU. Nothing on a GPU box goes near it.
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from _bench_dry import DryModel
        ckpt = None
        model = DryModel(rank, empty=(world > 1 and rank != 0))
    else:
        ckpt = ensure_checkpoint(args.preset, rank, dist)
        # rank 0 reads the checkpoint; replicas receive the weight arena by one RCCL broadcast over xGMI
        model = Qwen3TTSModel.from_pretrained(ckpt, device=local, max_batch=B, max_frames=args.frames + 8, max_prompt=192 if clone else 128,
                                              use_graph=not args.no_graph, n_streams=args.streams, codec_overlap_cus=args.codec_cus,
                                              weights_from_broadcast=(world > 1 and rank != 0))
    weight_broadcast = None
    if world > 1:
        import torch
        dev = torch.device("cpu") if dry else torch.device("cuda", local)  # where the small tensors of the collectives live
        # The job's only collective. Default: the library's own broadcast (q3tts_model_broadcast: RCCL straight from
        # the C ABI, the path a Swift or C host takes -- torch only carries the 128-byte communicator id). Fallback, and the
        # one-device gloo rehearsal: torch.distributed over a zero-copy view of the arena.
        # ("native-force": also in the one-device gloo rehearsal, where RCCL must refuse two ranks on one GPU -- that run
        # exercises the every-rank-agrees fallback below on hardware)
        how = os.environ.get("Q3TTS_BENCH_BROADCAST", "native")
        want_native = ((how == "native" and backend == "nccl") or how == "native-force") and not dry
        ok = 0
        if want_native:
            # Pre-flight on EVERY rank before anything collective: a rank whose process cannot open RCCL must say so here -- once the
            # others are inside ncclCommInitRank they wait for it for ever. (ncclGetUniqueId is local; rank 0's id is the one used.)
            ids = [None]
            avail = 1
            try:
                ids[0] = Qwen3TTSModel.comm_unique_id()
            except Exception as e:
                avail = 0
                print(f"[bench] rank {rank}: native broadcast unavailable: {e}", file=sys.stderr)
            every = torch.tensor([avail], device=dev, dtype=torch.int32)
            dist.all_reduce(every, op=dist.ReduceOp.MIN)
            if int(every.item()):
                dist.broadcast_object_list(ids, src=0)
                try:
                    model.broadcast_weights(ids[0], rank, world, 0)
                    ok = 1
                except Exception as e:
                    print(f"[bench] rank {rank}: native broadcast failed: {e}", file=sys.stderr)
            agreed = torch.tensor([ok], device=dev, dtype=torch.int32)
            dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
            ok = int(agreed.item())
        if ok:
            weight_broadcast = "q3tts_model_broadcast (RCCL from the C ABI)"
        elif dry:
            broadcast_weights(dist, model.arena_t, src=0)
            weight_broadcast = f"torch.distributed ({backend}), dry run"
        else:
            ptr, nbytes = model.arena()

            class _Arena:  # zero-copy view of the engine's weight arena for torch.distributed
                __cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}

            arena = torch.as_tensor(_Arena(), device=torch.device("cuda", local))
            broadcast_weights(dist, arena, src=0)
            torch.cuda.synchronize()
            weight_broadcast = f"torch.distributed ({backend})"
        # every replica now holds rank 0's bytes: compare a checksum of the arena across ranks
        ck = torch.tensor([model.arena_checksum() & ((1 << 62) - 1)], device=dev, dtype=torch.int64)
        lo_ck, hi_ck = ck.clone(), ck.clone()
        dist.all_reduce(lo_ck, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi_ck, op=dist.ReduceOp.MAX)
        assert int(lo_ck.item()) == int(hi_ck.item()), "weight arena differs between ranks after the broadcast"
    lo, hi = shard_rows(B * world, rank, world)
    reqs = build_requests(args.preset, lo, hi, args.n_text, n_instruct)
    temp = 0.0 if args.greedy else 0.9

    # row_base: a row's random stream is keyed by its GLOBAL index, so the sharded job draws what one 32 x N-row call would
    gen_kw = dict(temperature=temp, top_k=50, top_p=1.0, repetition_penalty=rep, seed=1234, force_frames=args.frames, row_base=lo)

    def step():
        return model.generate_batch(reqs, **gen_kw)

    def sync_all():
        if dist is not None:
            import torch
            dist.barrier()
            if not dry:
                torch.cuda.synchronize()

    solo = None  # phase times of one batch running alone (last warm-up step): nothing overlaps there
    for _ in range(args.warmup):
        step()
        tm = model.last_timing()
        solo = {"prefill": tm.prefill_ms, "ar_decode": tm.decode_ms, "codec_decode": tm.codec_ms, "voice_frontend": tm.frontend_ms,
                "frame_step": tm.decode_ms / max(tm.frame_steps, 1)}
    sync_all()
    t0 = time.perf_counter()
    dec_ms = pre_ms = codec_ms = fe_ms = 0.0
    launches = 0
    frame_steps = 0
    kv_bytes = 0
    frames_done = 0

    def account(res):
        nonlocal dec_ms, pre_ms, codec_ms, fe_ms, frame_steps, kv_bytes, frames_done, launches
        tm = model.last_timing()  # timing of the job that just ended
        launches = max(launches, getattr(tm, "launches_per_frame_step", 0))
        pre_ms += tm.prefill_ms
        dec_ms += tm.decode_ms
        codec_ms += tm.codec_ms
        fe_ms += tm.frontend_ms
        frame_steps += tm.frame_steps
        kv_bytes += tm.kv_bytes_read
        frames_done += sum(r.codes.shape[0] for r in res)

    # K steps, all of their work inside the timed region. Back-to-back batches are pipelined two deep: step i+1's prompt
    # assembly, prefill and AR loop are issued while step i's codec decode runs on the engine's second stream; a step's
    # PCM is on the host when its generate_batch_end returns. --no-pipeline issues the steps strictly one after another.
    iter_ms = []  # wall time of each pipelined iteration (a begin + the previous job's end): the steady state, reported beside `value`
    if args.no_pipeline:
        for _ in range(args.steps):
            account(step())
    else:
        trace = os.environ.get("Q3TTS_BENCH_TRACE") == "1"  # host wall time of each half, to stderr
        def timed(what, f, *a, **k):
            t = time.perf_counter()
            r = f(*a, **k)
            if trace:
                print(f"[trace] {what} {1e3 * (time.perf_counter() - t):.1f} ms", file=sys.stderr)
            return r
        job = timed("begin", model.generate_batch_begin, reqs, more_follows=(args.steps > 1), **gen_kw)
        for i in range(args.steps - 1):
            t_it = time.perf_counter()
            nxt = timed("begin", model.generate_batch_begin, reqs, more_follows=(i + 2 < args.steps), **gen_kw)
            account(timed("end", model.generate_batch_end, job))
            job = nxt
            iter_ms.append(1e3 * (time.perf_counter() - t_it))
        account(timed("end", model.generate_batch_end, job))
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        elapsed, frames_done = reduce_job_stats(dist, elapsed, frames_done, torch.device("cpu") if dry else torch.device("cuda", local))
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    value = frames_done / elapsed
    # memory-side bytes per frame step from the PMC passes committed under profiles/ (default workload only); the file
    # names the kernel sources it was measured on, and a figure from other sources is not reported
    traffic = traffic_src = None
    tpath = os.path.join(ROOT, "profiles", "frame_traffic.json")
    if args.preset == "1.7b" and B == 32 and os.path.exists(tpath):
        tj = json.load(open(tpath))
        if tj.get("kernel_sources_sha16") == kernel_sources_sha16():
            traffic = tj["traffic_bytes_per_frame_step"]
            traffic_src = "profiles/frame_traffic.json (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, Infinity-Cache hits included; " \
                          "measured on these kernel sources)"
        else:
            traffic_src = "profiles/frame_traffic.json is from other kernel sources: not reported"
    step_ms = dec_ms / max(frame_steps, 1)
    algo_bytes = model.info.weight_bytes + kv_bytes / max(frame_steps, 1)
    achieved = algo_bytes / (step_ms * 1e-3) / 1e9
    pipelined = not args.no_pipeline and args.steps > 1
    # codec decoder: 2.484 GMAC per frame; every fp32 product block is executed as THREE fp16 MFMA products
    # (two-plane split, codec_conv.hip) -- `achieved` counts exactly those, `fp32_equivalent_tflops` the useful work
    n_frames_step = B * args.frames
    codec_solo_ms = solo["codec_decode"] if solo else codec_ms / args.steps
    # a float16 ("lite") speech tokenizer decodes in float16 like the reference's: ONE fp16 product per block from initConv on
    # (codec_conv_h1.hip; 96 % of the decoder's MACs), three in front of it
    codec_f16 = args.preset == "0.6b-q4"
    codec_products = 1 if codec_f16 else 3
    codec_flops_bf16 = codec_products * 2 * 2.484e9 * n_frames_step
    latency_ms = (solo["voice_frontend"] + solo["prefill"] + solo["ar_decode"] + solo["codec_decode"]) if solo else None
    # HBM GB/s of the decoder's two narrow stages from the PMC passes committed under profiles/ (FETCH_SIZE x 2 +
    # WRITE_SIZE per launch over un-profiled durations, tools/measure_codec_traffic.sh); reported while the kernel sources match
    codec_hbm = None
    cpath = os.path.join(ROOT, "profiles", "codec_traffic.json")
    if os.path.exists(cpath):
        cj = json.load(open(cpath))
        codec_hbm = {k: {"hbm_gbs": v["hbm_gbs"], "gb": v["bytes"] / 1e9, "ms": v["ms"]} for k, v in cj["stages"].items()}
        codec_hbm["source"] = "profiles/codec_traffic.json (%d rows x %d frames; rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE)%s" % (
            cj["rows"], cj["frames"], "" if cj.get("kernel_sources_sha16") == kernel_sources_sha16() else "; measured on EARLIER kernel sources")
    out = {
        "metric": "codec_tokens_per_s", "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16 (int4-g64 weights)" if args.preset.endswith("q4") else "bf16", "data": "synthetic",
        "config": {"workload": f"Qwen3-TTS-{args.preset.upper()} bf16, batch {B}/GPU x {args.frames} frames, "
                               f"{args.n_text} text + {n_instruct} instruct tokens, T={temp} top-k 50, "
                               + ("voice clone: 3.0 s reference clip per row -> codec encoder + speaker encoder + ICL prompt, "
                                  if clone else "")
                               + "prompt assembly + prefill + hipGraph AR decode + "
                               + ("float16 codec decode (float16 speech tokenizer, as the reference computes it)" if codec_f16
                                  else "fp32-equivalent (two-plane fp16 split) codec decode") + " -> 24 kHz PCM"
                               + ("; steps pipelined two deep (a step's codec decode overlaps the next step's AR loop)" if pipelined else ""),
                   "batch_per_gpu": B, "frames_per_utterance": args.frames, "parallelism": f"batch-shard x{world}",
                   "weight_broadcast": weight_broadcast},
        "rtf_audio_s_per_wall_s": value * FRAME_SECONDS, "rtf_wall_s_per_audio_s": 1.0 / (value * FRAME_SECONDS),
        # throughput reading: audio seconds per wall second, per utterance of the batch
        "rtf_per_utterance": value * FRAME_SECONDS / (B * world),
        # latency reading: one batch running alone, request in -> PCM out (last warm-up step)
        "utterance_latency_ms": latency_ms,
        "rtf_per_utterance_latency": (args.frames * FRAME_SECONDS * 1e3 / latency_ms) if latency_ms else None,
        "phase_ms_per_step": {"voice_frontend": fe_ms / args.steps, "prefill": pre_ms / args.steps,
                              "ar_decode": dec_ms / args.steps, "codec_decode": codec_ms / args.steps,
                              "note": ("HIP events per phase on the phase's own stream; codec_decode of step i runs beside "
                                       "prefill / ar_decode of step i+1, so the phases add up to more than ms_per_step")
                              if pipelined else "phases run back to back on one batch"},
        "phase_ms_alone": solo,
        # informational: the median pipelined iteration. `value` / `ms_per_step` above are the contract's K steps as a whole, which
        # also hold the pipeline's fill (a first step nothing overlaps) and drain (the last batch's decode, alone)
        "steady_state": ({"ms_per_step": sorted(iter_ms)[len(iter_ms) // 2],
                          "frames_per_s": world * B * args.frames / (sorted(iter_ms)[len(iter_ms) // 2] / 1e3)} if len(iter_ms) >= 3 else None),
        "roofline": {"bound": "hbm", "kernel": "frame_step (hipGraph: talker step + 16 code-predictor passes + samplers)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": step_ms,
                     "avg_launch_ms_alone": solo["frame_step"] if solo else None,
                     # what the frame step is made of: a chain of dependent kernel launches (graph nodes). On this part a dependent
                     # kernel that loads its input and stores its output costs ~4.5 us however small: launch
                     # count x that floor, not bytes, bounds the step
                     "chain": ({"launches_per_frame_step": launches, "us_per_launch": step_ms * 1e3 / launches,
                                "us_per_launch_alone": (solo["frame_step"] * 1e3 / launches) if solo else None} if launches else None),
                     # the same kernel with nothing beside it (last warm-up step): in the pipelined region the frame loop
                     # shares the chip with the previous batch's codec decode, which is what `frac` above includes
                     "frac_alone": (algo_bytes / (solo["frame_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if solo and solo["frame_step"] else None},
        "roofline_codec": {"bound": "mfma", "kernel": ("codec decoder (conv_gemm_h1: float16 tensors, one fp16 MFMA product per block)" if codec_f16 else
                                                        "codec decoder (conv_gemm_h2 / resunit_h2: fp16 MFMA, %d executed products per fp32 product block)" % codec_products), "executed_products_per_fp32_product": codec_products, "achieved": codec_flops_bf16 / (codec_solo_ms * 1e-3) / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS,
                           "unit": "TFLOP/s", "frac": codec_flops_bf16 / (codec_solo_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                           "fp32_equivalent_tflops": 2 * 2.484e9 * n_frames_step / (codec_solo_ms * 1e-3) / 1e12,
                           "algorithmic_gmac_per_frame": 2.484, "frames_per_launch_sequence": n_frames_step,
                           "narrow_stages_hbm": codec_hbm, "hbm_peak_gbs": HBM_PEAK_GBS,
                           "ms": codec_solo_ms, "measured": "one batch alone (last warm-up step)" if solo else "overlapped steps"},
    }
    if world == 1 and not clone and not args.no_streaming and not dry:
        # row f1, outside the timed region: one batch with the waveform streamed while the tokens are generated (16-frame
        # chunks, 32 frames of left context, 4 of look-ahead) -- when does the first audio reach the host?
        sk = dict(gen_kw, audio_chunk_frames=16, audio_window_frames=32, audio_lookahead_frames=4)
        model.generate_batch(reqs, **sk)
        t1 = time.perf_counter()
        model.generate_batch(reqs, **sk)
        t2 = time.perf_counter()
        tm = model.last_timing()
        out["streaming"] = {"first_audio_ms": tm.first_audio_ms, "batch_ms": (t2 - t1) * 1e3,
                            "one_shot_latency_ms": latency_ms, "chunk_frames": 16, "window_frames": 32, "lookahead_frames": 4,
                            "note": "time from the request to the first 16 frames (1.28 s) of audio of all rows on the host; "
                                    "the exact (one-shot) mode delivers everything after one_shot_latency_ms"}
    if world == 1 and not args.no_cpu_baseline and not dry:
        out["cpu_baseline"] = cpu_baseline(ckpt, args.preset, args.n_text, n_instruct, args.cpu_frames)
    if dist is not None:
        dist.destroy_process_group()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
