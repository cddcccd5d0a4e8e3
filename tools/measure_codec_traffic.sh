#!/bin/bash
# HBM-side bytes of every conv launch of the codec decoder ("rocprof HBM GB/s for the two narrow stages"):
# rocprofv3 PMC passes FETCH_SIZE and WRITE_SIZE (separate passes, --kernel-trace only beside them) over
# tools/codec_only.py B F, plus one un-profiled kernel trace for the durations (a PMC pass serialises and slows the
# dispatches). tools/codec_traffic.py joins the three by launch order. Run on the GPU box from the repo root.
set -u
B=${1:-32}; F=${2:-200}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/codec_traffic
rm -rf $out && mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -o t -- python3 tools/codec_only.py $B $F > $out/$c.log 2>&1
  echo "$c rc=$?" >> $out/progress.log
done
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o t -- python3 tools/codec_only.py $B $F > $out/trace.log 2>&1
echo "trace rc=$?" >> $out/progress.log
python tools/codec_traffic.py $out/FETCH_SIZE/t_counter_collection.csv $out/WRITE_SIZE/t_counter_collection.csv \
    $(find $out/trace -name "*kernel_trace.csv" | head -1) $B $F > $out/summary.txt 2>&1
find $out -name "*kernel_trace.csv" -delete
find $out -name "*counter_collection.csv" -size +20M -delete
cat $out/summary.txt
