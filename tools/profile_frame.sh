#!/bin/bash
# Per-kernel durations of the default bench workload (eager launches: rocprofv3 cannot follow a multi-packet graph batch
# across the AQL ring wrap). Run on the GPU box from the repo root:
#   tools/profile_frame.sh <tag> [extra bench.py flags]   ->  gpurun_out/prof_<tag>/stats.txt, stats.csv
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 bench.py --no-pipeline --frames ${FRAMES:-40} --steps ${STEPS:-2} --warmup ${WARMUP:-1} ${GRAPHFLAG---no-graph} --no-cpu-baseline --no-streaming "$@" > $out/bench.json 2> $out/bench.err
echo "rc=$?" >> $out/bench.err
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp "$f" $out/stats.csv
python tools/kstats.py $out/stats.csv > $out/stats.txt
find $out -name "*kernel_trace.csv" -size +60M -delete
head -45 $out/stats.txt
