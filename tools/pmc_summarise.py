import csv, collections, sys
f=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"].replace("q3::(anonymous namespace)::","").replace("void ","")
    k=(n.split("(")[0][:64],)  # keep the template arguments: they tell the K = 7 / K = 1 / fused variants apart
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k]+=1
rows=sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES",0))[:12]
for k,v in rows:
    w=v.get("SQ_WAVE_CYCLES",1)
    print(k, "calls", cnt[k]//max(1,len(v)), " ".join(f"{n[3:]}={v[n]/w:.2f}" for n in v if n!="SQ_WAVE_CYCLES"), f"WAVE_CYCLES={w:.3g}")
