"""Per-dispatch durations of the codec decoder's convs from a rocprofv3 kernel trace (last decode of the run), next to
the bf16x6 matrix-core time of each (MACs * 2 * 6 / 2.5 PFLOP/s) -- tools/codec_only.py B F shapes are passed in."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
convs = [r for r in rows if "conv_gemm" in r["Kernel_Name"] or "out_conv" in r["Kernel_Name"] or "resunit" in r["Kernel_Name"]]
last = convs[-(len(convs) // reps):]
tot = 0.0
for r in last:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    k = r["Kernel_Name"]
    nm = "resu" if "resunit" in k else "split" if "split" in k else "fp32" if "conv_gemm" in k else "out"
    bn = k[k.index("<") + 1:k.index(">")] if "<" in k else ""
    print(f"{nm:5s} {bn:4s} ntiles {int(r['Grid_Size_X']) // 256:4d} mtiles {r['Grid_Size_Y']:5s} rows {r['Grid_Size_Z']:3s} {d:9.1f} us")
print(f"total {tot / 1e3:.1f} ms")
