"""HBM-side traffic of one frame step from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE are derived from the L2's
memory-side request counters; MI355X_MICROARCH.md, HBM section). Usage:
    frame_traffic.py <counter_collection.csv at F1 frames> <same at F2 frames> F1 F2 <counter name>
Prints the counter's sum over the frame loop's kernels, per frame step = (sum(F2) - sum(F1)) / (F2 - F1),
so that load-time kernels, prompt assembly and prefill cancel."""
import csv
import json
import re
import sys

# kernels of the frame loop (prefill uses some of them too: identical in both runs, so it cancels)
FRAME = ("gemm_skinny_kernel", "attn_decode_kernel", "attn_chunk_kernel", "sampler_kernel", "frame_end_kernel", "norm_rows_kernel",
         "advance_len_kernel", "stamp_kernel")


def total(path, counter):
    s = 0.0
    per = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or not any(c in r["Kernel_Name"] for c in FRAME):
            continue
        v = float(r["Counter_Value"])
        s += v
        k = re.sub(r"\(anonymous namespace\)::|q3::|void |\(.*\)$", "", r["Kernel_Name"])[:60]
        per[k] = per.get(k, 0.0) + v
    return s, per


f1, f2, F1, F2, counter = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
(s1, p1), (s2, p2) = total(f1, counter), total(f2, counter)
per_frame = (s2 - s1) / (F2 - F1)
top = sorted(((p2.get(k, 0) - p1.get(k, 0)) / (F2 - F1), k) for k in p2)[-12:]
print(json.dumps({"counter": counter, "per_frame_step": per_frame, "frames": [F1, F2],
                  "top_kernels_per_frame": [[k, v] for v, k in reversed(top)]}))
