#!/bin/bash
# Memory-side bytes of one frame step (roofline.traffic): four rocprofv3 PMC passes of the default bench workload with
# eager launches, FETCH_SIZE and WRITE_SIZE each at 2 and 6 frames per utterance (tools/frame_traffic.py differences
# them). Run on the GPU box from the repo root; writes gpurun_out/traffic/*.json and a progress log.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/traffic
mkdir -p $out
: > $out/progress.log
for c in FETCH_SIZE WRITE_SIZE; do
  for f in 2 6; do
    timeout -k 10 280 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/${c}_$f -o t -- python3 bench.py --no-graph --no-pipeline --frames $f --steps 1 --warmup 0 --no-cpu-baseline --no-streaming > $out/${c}_$f.log 2>&1
    echo "$c $f rc=$?" >> $out/progress.log
  done
  python tools/frame_traffic.py $out/${c}_2/t_counter_collection.csv $out/${c}_6/t_counter_collection.csv 2 6 $c > $out/$c.json
done
cat $out/FETCH_SIZE.json $out/WRITE_SIZE.json
