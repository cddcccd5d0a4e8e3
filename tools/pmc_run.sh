#!/bin/bash
# one rocprofv3 PMC pass over tools/codec_only.py B F with the counters given as arguments; prints per-kernel sums
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$tag
rm -rf $out && mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out -o t -- python3 tools/codec_only.py 16 100 > $out/run.log 2>&1
echo "rc=$?"
f=$(find $out -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, collections, sys
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Kernel_Name"].replace("q3::(anonymous namespace)::","").replace("void ","").split("(")[0][:48]
    agg[n][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[n]+=1
rows=sorted(agg.items(), key=lambda kv: -max(kv[1].values()))[:8]
for k,v in rows:
    print(k, "dispatches", cnt[k]//max(1,len(v)), " ".join(f"{n}={x:.4g}" for n,x in sorted(v.items())))
PY
find $out -name "*.csv" -size +5M -delete
