"""Assemble profiles/frame_traffic.json from the two PMC differences written by tools/measure_frame_traffic.sh
(gpurun_out/traffic/FETCH_SIZE.json, WRITE_SIZE.json) and stamp it with the kernel sources it was measured on."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "traffic")
fe = json.load(open(os.path.join(src, "FETCH_SIZE.json")))
wr = json.load(open(os.path.join(src, "WRITE_SIZE.json")))
fetch_bytes = fe["per_frame_step"] * 1024 * 2   # KB -> bytes, x2: gfx950 tallies 128-byte requests of 16-B/lane reads at 64 B
write_bytes = wr["per_frame_step"] * 1024
out = {
    "what": "memory-side (L2 <-> fabric) bytes of ONE frame step of bench.py's default workload (Qwen3-TTS-1.7B bf16, batch 32), eager launches",
    "method": "tools/measure_frame_traffic.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in separate passes, --pmc WRITE_SIZE, "
              "each at 2 and at 6 frames per utterance; per-frame value = (sum over the engine's kernels at 6 frames - at 2 frames) / 4 "
              "(tools/frame_traffic.py); runs stay below 16384 AQL packets (DESIGN.md section 5: profiler ring-wrap fault)",
    "kernel_sources_sha16": bench.kernel_sources_sha16(),
    "fetch_size_kb_raw": fe["per_frame_step"],
    "fetch_correction": "x2: on gfx950 FETCH_SIZE tallies the 128-byte requests of 16-byte-per-lane streaming reads at 64 bytes (MI355X_MICROARCH.md, HBM section)",
    "fetch_bytes": fetch_bytes, "write_size_kb": wr["per_frame_step"], "write_bytes": write_bytes,
    "traffic_bytes_per_frame_step": fetch_bytes + write_bytes,
    "top_fetch_kernels_kb_raw": fe["top_kernels_per_frame"], "top_write_kernels_kb": wr["top_kernels_per_frame"],
}
json.dump(out, open(os.path.join(ROOT, "profiles", "frame_traffic.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("kernel_sources_sha16", "fetch_bytes", "write_bytes", "traffic_bytes_per_frame_step")}))
