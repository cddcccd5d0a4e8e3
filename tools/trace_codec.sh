#!/bin/bash
# per-launch trace of one codec decode (tools/codec_only.py B F) -> gpurun_out/<tag>_conv_trace.txt
set -u
tag=${1:-codec}; B=${2:-32}; F=${3:-200}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${tag}_trace
rm -rf $out && mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 tools/codec_only.py $B $F > $out/run.log 2>&1
echo "rc=$?" >> $out/run.log
f=$(find $out -name "*kernel_trace.csv" | head -1)
python tools/conv_trace.py "$f" 3 $B $F > gpurun_out/${tag}_conv_trace.txt 2>&1
s=$(find $out -name "*kernel_stats.csv" | head -1)
cp "$s" gpurun_out/${tag}_kernel_stats.csv
find $out -name "*kernel_trace.csv" -delete
cat gpurun_out/${tag}_conv_trace.txt
