"""Per-dispatch durations of the codec decoder's convs from a rocprofv3 kernel trace (last decode of the run) with each
launch labelled by walking CodecRunner::run_front / run_tail's order for the shipped decoder geometry, next to its
matrix-core time (MACs * 2 * products / 2.5 PFLOP/s) and its minimum HBM time (input + output tensors once at 8 TB/s).
usage: conv_trace.py <kernel_trace.csv> [reps] [B] [F] [products]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
F = int(sys.argv[4]) if len(sys.argv) > 4 else 200
PROD = int(sys.argv[5]) if len(sys.argv) > 5 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
convs = [r for r in rows if "conv_gemm" in r["Kernel_Name"] or "conv_pw" in r["Kernel_Name"] or "out_conv" in r["Kernel_Name"] or "resunit" in r["Kernel_Name"]]
last = convs[-(len(convs) // reps):]

# (label, Cin, N, K, positions per frame, extra output copies, fused-unit flag)
seq = [("rvq_out", 512, 512, 1, 1), ("pre_conv", 512, 1024, 3, 1), ("t_in", 1024, 512, 1, 1)]
for l in range(8):
    seq += [(f"t{l}.qkv", 512, 1536, 1, 1), (f"t{l}.o", 512, 512, 1, 1), (f"t{l}.gateup", 512, 2048, 1, 1), (f"t{l}.down", 1024, 512, 1, 1)]
seq += [("t_out", 512, 1024, 1, 1)]
ppf = 1
for i in range(2):
    seq += [(f"up{i}.tconv", 1024, 2048, 1, ppf)]
    ppf *= 2
    seq += [(f"up{i}.pw1", 1024, 4096, 1, ppf), (f"up{i}.pw2", 4096, 1024, 1, ppf)]
seq += [("init_conv", 1024, 1536, 7, ppf)]
C = 1536
for i, s in enumerate((8, 5, 4, 3)):
    seq += [(f"b{i}.tconv", C, s * (C // 2), 2, ppf)]
    ppf *= s
    C //= 2
    for j in range(3):
        if C <= 96 or C == 192:
            seq += [(f"b{i}.res{j}.fused", C, C, 8, ppf)]
        else:
            seq += [(f"b{i}.res{j}.conv1", C, C, 7, ppf), (f"b{i}.res{j}.conv2", C, C, 1, ppf)]
seq += [("out_conv", 96, 1, 7, ppf)]
assert len(seq) == len(last), (len(seq), len(last))
tot = 0.0
print(f"{'launch':18s} {'kernel':10s} {'us':>9s} {'mfma us':>8s} {'hbm us':>7s}  PF/s   TB/s(min bytes)")
groups = {}
for (name, cin, n, k, p), r in zip(seq, last):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    kn = r["Kernel_Name"]
    nm = "resu" if "resunit" in kn else "pw" if "conv_pw" in kn else "h2" if "h2" in kn else "split" if "split" in kn else "fp32" if "conv_gemm" in kn else "out"
    bn = kn[kn.index("<") + 1:kn.index(">")] if "<" in kn else ""
    pos = B * F * p
    macs = pos * cin * n * k
    flops = macs * 2 * (PROD if nm != "out" else 1)
    byts = pos * (cin + n) * 4 if "fused" not in name else pos * 2 * n * 4
    if "tconv" in name: byts = pos * (cin + n) * 4
    print(f"{name:18s} {nm + ' ' + bn:10s} {d:9.1f} {flops / 2.5e15 * 1e6:8.1f} {byts / 8e12 * 1e6:7.1f}  {flops / d / 1e9:5.2f}  {byts / d / 1e6:5.2f}")
    g = name.split(".")[0] if name[0] in "bu" else ("transformer" if name[0] == "t" else name)
    groups[g] = groups.get(g, 0.0) + d
print(f"total {tot / 1e3:.1f} ms")
print(" ".join(f"{g}={v / 1e3:.1f}ms" for g, v in groups.items()))
