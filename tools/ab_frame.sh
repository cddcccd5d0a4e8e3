#!/bin/bash
# A/B of the frame step on the GPU box: the headline workload, one batch at a time, per library / environment variant.
#   tools/ab_frame.sh <tag> "<VAR=val ...>" ...     ->  gpurun_out/r04/ab_<tag>.txt (one line per variant)
tag=$1; shift
out=gpurun_out/r04/ab_$tag.txt
mkdir -p gpurun_out/r04
: > $out
for variant in "$@"; do
  env $variant python bench.py --no-pipeline --steps ${STEPS:-2} --warmup 2 --no-cpu-baseline --no-streaming ${BENCH_FLAGS:-} > /tmp/ab.json 2> /tmp/ab.err || { echo "$variant FAILED" >> $out; tail -3 /tmp/ab.err >> $out; continue; }
  python - "$variant" >> $out <<'PY'
import json,sys
d=json.loads([l for l in open('/tmp/ab.json') if l.startswith('{')][-1])
a=d['phase_ms_alone']
print("%-60s frame_step %.4f ms  ar %.1f  prefill %.2f  codec %.1f  step %.1f" % (sys.argv[1], a['frame_step'], a['ar_decode'], a['prefill'], a['codec_decode'], d['ms_per_step']))
PY
done
cat $out
