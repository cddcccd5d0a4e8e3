#!/usr/bin/env python3
"""Per-launch durations of one prefill layer pass, in launch order, from a rocprofv3 kernel trace of bench.py
(tools/profile_frame.sh): the prefill GEMMs share kernel templates (o_proj and down_proj), so the by-name statistics of
tools/kstats.py mix them. usage: prefill_trace.py <kernel_trace.csv>"""
import csv, statistics, sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
names = ["norm_rows", "qkv", "attn_chunk", "o_proj", "norm_rows", "gate_up", "down_proj"]
i = next(k for k, (n, _) in enumerate(seq) if "prefill_chunk_load" in n) + 1
per = [[] for _ in names]
while i + len(names) <= len(seq) and "norm_rows" in seq[i][0] and "attn_chunk" in seq[i + 2][0]:
    for k in range(len(names)):
        per[k].append(seq[i + k][1])
    i += len(names)
tot = 0.0
for n, v in zip(names, per):
    tot += statistics.mean(v)
    print(f"{n:12s} {statistics.mean(v):7.2f} us  (n={len(v)})  {seq[i - len(names) + names.index(n)][0][:70] if v else ''}")
print(f"layer pass   {tot:7.2f} us")
