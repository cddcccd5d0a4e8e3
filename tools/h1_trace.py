"""Per-launch durations of the float16 MainDecoder (conv_gemm_h1 / out_conv_h1) from a rocprofv3 kernel trace of
tools/codec_only.py: the launches of the LAST decode in order, with the layer each one is (known from the launch order)."""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "h1" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = ["initConv"]
for b in range(4):
    names.append(f"block{b} tconv")
    for j in range(3):
        names += [f"block{b} res{j + 1} (fused unit)"] if b == 3 else [f"block{b} res{j + 1} conv1 (k7)", f"block{b} res{j + 1} conv2 (k1)"]
names.append("outConv")
per = len(names)  # initConv + 3 x (tconv + 3 x (conv1, conv2)) + (tconv + 3 fused units) + outConv
last = rows[-per:]
tot = 0.0
for n, r in zip(names, last):
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += us
    k = r["Kernel_Name"]
    k = (k[k.find("conv_gemm_h1_kernel"):k.find(">") + 1] if "conv_gemm_h1" in k else
         k[k.find("resunit_h1_kernel"):k.find(">") + 1] if "resunit_h1" in k else "out_conv_h1")
    print(f"{n:28s} {k:32s} {us:9.1f} us")
print(f"{'sum':28s} {'':32s} {tot:9.1f} us")
