#!/bin/bash
# libq3tts_hip_<name>.so per "name:flags" argument (e.g. pf0:-DQ3_PF_MODE=0): only the two kernel files that carry the touch
# code and the engine (its planner) are rebuilt, everything else is linked from build/. For A/B runs through Q3TTS_LIB (tools/ab_frame.sh).
set -e
cd "$(dirname "$0")/../swift-qwen3-tts_amd"
make -j8 > /dev/null
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I../include -Icsrc"
for spec in "$@"; do
  name=${spec%%:*}; extra=${spec#*:}
  (
  mkdir -p build_$name
  /opt/rocm/bin/hipcc $FLAGS $extra -c csrc/kernels/gemm_decode.hip -o build_$name/k_gemm_decode.o &
  /opt/rocm/bin/hipcc $FLAGS $extra -c csrc/kernels/attn_decode.hip -o build_$name/k_attn_decode.o &
  /opt/rocm/bin/hipcc $FLAGS $extra -c csrc/engine.cc -o build_$name/engine.o &  # (the planner is compiled out with the touch code)
  wait
  objs=$(ls build/*.o | grep -v -e k_gemm_decode.o -e k_attn_decode.o -e build/engine.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o qwen3tts/libq3tts_hip_$name.so $objs build_$name/k_gemm_decode.o build_$name/k_attn_decode.o build_$name/engine.o
  ) &
done
wait
ls -la qwen3tts/*.so
