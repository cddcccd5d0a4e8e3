#!/bin/bash
# shader clock (tools/clockprobe, its own process) while bench.py runs: sequential steps (frame loop, then decode, in turn),
# then pipelined steps (decode beside the next frame loop). Output: gpurun_out/clock_{seq,pipe}.txt + the bench lines.
cd "$GRAFT_REPO_ROOT"
for mode in seq pipe; do
  flag=""; [ $mode = seq ] && flag="--no-pipeline"
  ./tools/clockprobe 2400 10 > gpurun_out/clock_$mode.txt &
  pp=$!
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-streaming $flag > gpurun_out/clock_bench_$mode.json 2> gpurun_out/clock_bench_$mode.err
  kill $pp 2>/dev/null; wait $pp 2>/dev/null
  python - gpurun_out/clock_$mode.txt <<'PY'
import sys
rows=[l.split() for l in open(sys.argv[1]) if "MHz" in l]
t=[float(r[0]) for r in rows]; f=[float(r[2]) for r in rows]
import statistics
print(sys.argv[1], "samples", len(f), "min %.0f median %.0f max %.0f MHz" % (min(f), statistics.median(f), max(f)))
# histogram
import collections
h=collections.Counter(int(x//100)*100 for x in f)
print("  " + "  ".join(f"{k}:{v}" for k,v in sorted(h.items())))
PY
done
