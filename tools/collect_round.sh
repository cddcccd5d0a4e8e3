#!/bin/bash
# Collects the round's profile artefacts on the GPU box (run from the repo root through gpurun): part = a | b | c
set -u
tag=${1:-r03}; part=${2:-a}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$tag
if [ "$part" = a ]; then
  bash tools/profile_frame.sh ${tag} > gpurun_out/$tag/eager_stats.log 2>&1
  cp gpurun_out/prof_${tag}/stats.txt gpurun_out/$tag/kernel_stats.txt; cp gpurun_out/prof_${tag}/stats.csv gpurun_out/$tag/kernel_stats.csv
  GRAPHFLAG= FRAMES=12 STEPS=1 WARMUP=0 bash tools/profile_frame.sh ${tag}g > gpurun_out/$tag/graph_stats.log 2>&1
  cp gpurun_out/prof_${tag}g/stats.txt gpurun_out/$tag/kernel_stats_graph.txt
  bash tools/measure_frame_traffic.sh > gpurun_out/$tag/traffic.log 2>&1
  python tools/make_frame_traffic_json.py > gpurun_out/$tag/traffic_json.log 2>&1
  cp profiles/frame_traffic.json gpurun_out/$tag/frame_traffic.json
  tail -3 gpurun_out/$tag/traffic_json.log
elif [ "$part" = b ]; then
  bash tools/measure_codec_traffic.sh 32 200 > gpurun_out/$tag/codec_traffic.log 2>&1
  cp gpurun_out/codec_traffic/summary.txt gpurun_out/$tag/codec_traffic.txt; cp gpurun_out/codec_traffic/codec_traffic.json gpurun_out/$tag/codec_traffic.json
  bash tools/measure_codec_pmc.sh > gpurun_out/$tag/codec_pmc.log 2>&1
  cp gpurun_out/codec_pmc/summary.txt gpurun_out/$tag/codec_pmc_summary.txt
  bash tools/pmc_run.sh insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES > gpurun_out/$tag/codec_pmc_insts.txt 2>&1
  bash tools/trace_codec.sh ${tag} 32 200 > gpurun_out/$tag/conv_trace.log 2>&1
  cp gpurun_out/${tag}_conv_trace.txt gpurun_out/$tag/conv_trace.txt
  tail -4 gpurun_out/$tag/conv_trace.txt
else
  for cfg in "0.6b 32" "0.6b 8" "0.6b-q4 64" "1.7b-base 16" "0.6b-base 16"; do
    set -- $cfg
    python bench.py --preset $1 --batch $2 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/$tag/bench_$1_$2.json 2>/dev/null
    python - gpurun_out/$tag/bench_$1_$2.json "$1 $2" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
a=d["phase_ms_alone"]
print(sys.argv[2], round(d["value"]), round(d["value"]*0.08), round(d["ms_per_step"],1), "prefill", round(a["prefill"],1), "fe", round(a["voice_frontend"],1), "ar", round(a["ar_decode"]), "codec", round(a["codec_decode"]), "frame", round(a["frame_step"],3))
PY
  done
fi
