"""Exclusive cost of every kernel of a serial launch chain from a rocprofv3 kernel trace: in a chain each launch can only
end after its predecessor, so end[i] - end[i-1] is what launch i added to the chain (its own work plus the boundary in
front of it), whatever the tool stamps as its start. Usage: chain_costs.py <kernel_trace.csv> [name filter]"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["End_Timestamp"]))
agg = defaultdict(lambda: [0, 0.0, 0.0])
prev = None
for r in rows:
    e, s = int(r["End_Timestamp"]), int(r["Start_Timestamp"])
    n = re.sub(r"\(anonymous namespace\)::|q3::|void |\(.*\)$", "", r["Kernel_Name"])[:64]
    if prev is not None and e - prev < 200000:  # gaps above 0.2 ms are host pauses, not chain links
        a = agg[n]
        a[0] += 1
        a[1] += (e - prev) / 1e3
        a[2] += (e - s) / 1e3
    prev = e
tot = sum(a[1] for a in agg.values())
print(f"{'kernel':64s} {'calls':>7s} {'chain us':>9s} {'stamped us':>10s} {'total ms':>9s} {'share':>6s}")
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if len(sys.argv) > 2 and sys.argv[2] not in n:
        continue
    print(f"{n:64s} {a[0]:7d} {a[1] / a[0]:9.2f} {a[2] / a[0]:10.2f} {a[1] / 1e3:9.2f} {100 * a[1] / tot:5.1f}%")
