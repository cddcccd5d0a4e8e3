#!/bin/bash
# Round 4's final artefacts in one gpurun call: the GPU suite, then (only if it is green) the frame profiles on THESE kernel
# sources, the float16 decoder's per-launch trace, configs[4] and the headline bench with the driver's flags.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
python -m pytest tests -x -q -m gpu > gpurun_out/r04/full_final.log 2>&1
rc=$?
tail -3 gpurun_out/r04/full_final.log
[ $rc -eq 0 ] || exit $rc
bash tools/collect_round.sh r04 a > gpurun_out/r04_collect_a.log 2>&1
tail -1 gpurun_out/r04_collect_a.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04/h1_trace; rm -rf $out; mkdir -p $out
Q3TTS_CODEC_ONLY_F16=1 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 tools/codec_only.py 32 200 > $out/run.log 2>&1
f=$(find $out -name "*kernel_trace.csv" | head -1)
python tools/h1_trace.py "$f" > gpurun_out/r04/h1_launches.txt
find $out -name "*kernel_trace.csv" -delete
tail -2 gpurun_out/r04/h1_launches.txt
python bench.py --preset 0.6b-q4 --steps 6 --warmup 2 --no-cpu-baseline --no-streaming > gpurun_out/r04/bench_0.6b-q4_64.json 2> /dev/null
python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench.json 2> gpurun_out/r04/bench.err
python - <<'PY'
import json
for f in ("gpurun_out/r04/bench_0.6b-q4_64.json", "gpurun_out/r04/bench.json"):
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    a = d["phase_ms_alone"]
    print(f, round(d["value"]), round(d["ms_per_step"], 1), "alone: prefill %.1f ar %.1f codec %.1f frame %.3f" % (a["prefill"], a["ar_decode"], a["codec_decode"], a["frame_step"]),
          "traffic", d["roofline"]["traffic"], "frac", round(d["roofline"]["frac"], 4), round(d["roofline"]["frac_alone"], 4))
PY
