#!/bin/bash
# one rocprofv3 PMC pass over a two-frame eager bench.py run (prefill + frame loop kernels) with the counters given as
# arguments; prints per-kernel sums for the kernels whose name matches $MATCH (default: every kernel, top 12)
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmcb_$tag
rm -rf $out && mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out -o t -- python3 bench.py --no-graph --no-pipeline --frames 2 --steps 1 --warmup 0 --no-cpu-baseline --no-streaming > $out/run.log 2>&1
echo "rc=$?"
f=$(find $out -name "*counter_collection.csv" | head -1)
python - "$f" "${MATCH:-}" <<'PY'
import csv, collections, sys
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Kernel_Name"].replace("q3::(anonymous namespace)::","").replace("void ","").split("(")[0][:48]
    if sys.argv[2] and sys.argv[2] not in n: continue
    agg[n][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[n]+=1
rows=sorted(agg.items(), key=lambda kv: -max(kv[1].values()))[:12]
for k,v in rows:
    d=cnt[k]//max(1,len(v))
    print(k, "dispatches", d, " ".join(f"{n}={x/d:.4g}" for n,x in sorted(v.items())), "(per dispatch)")
PY
find $out -name "*.csv" -size +5M -delete
