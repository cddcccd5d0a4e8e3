import csv, collections, sys
f=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=(r["Kernel_Name"][:60], r.get("Grid_Size_X",""), r.get("Grid_Size_Y",""))
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k]+=1
rows=sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES",0))[:12]
for k,v in rows:
    w=v.get("SQ_WAVE_CYCLES",1)
    print(k, "calls", cnt[k]//max(1,len(v)), " ".join(f"{n[3:]}={v[n]/w:.2f}" for n in v if n!="SQ_WAVE_CYCLES"), f"WAVE_CYCLES={w:.3g}")
