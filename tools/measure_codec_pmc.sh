#!/bin/bash
# Matrix-pipe occupancy of the codec decoder's kernels: two rocprofv3 PMC passes over tools/codec_only.py (16 rows x 100
# frames), values normalised by SQ_WAVE_CYCLES (tools/pmc_summarise.py). Run on the GPU box from the repo root; writes
# gpurun_out/codec_pmc/summary.txt. PMC passes only carry --kernel-trace (gpurun refuses other trace domains with --pmc).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/codec_pmc
rm -rf $out && mkdir -p $out
sets=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"
      "SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE")
echo "# tools/codec_only.py 16 100 (fp16x2 kernels of this build), rocprofv3 --kernel-trace --pmc, two passes; values / SQ_WAVE_CYCLES" > $out/summary.txt
i=0
for s in "${sets[@]}"; do
  timeout -k 10 280 rocprofv3 --kernel-trace --pmc $s --output-format csv -d $out/p$i -o t -- python3 tools/codec_only.py 16 100 > $out/p$i.log 2>&1
  echo "pass $i rc=$?" >> $out/progress.log
  f=$(find $out/p$i -name "*counter_collection.csv" | head -1)
  python tools/pmc_summarise.py "$f" >> $out/summary.txt
  echo >> $out/summary.txt
  i=$((i+1))
done
cat $out/summary.txt
