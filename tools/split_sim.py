import numpy as np
rng = np.random.default_rng(0)
K, N = 5376, 2000
x = rng.standard_normal((N, K)).astype(np.float32) * rng.choice([0.01, 1.0, 30.0], size=(N, 1)).astype(np.float32)
w = (rng.standard_normal((K,)) * 0.02).astype(np.float32)
truth = (x.astype(np.float64) * w.astype(np.float64)).sum(1)
scale = np.sqrt((truth ** 2).mean())
# fp32 sequential-ish chain (blocks of 4 then chain) emulate with float32 cumulative sum
def chain32(p):
    acc = np.zeros(p.shape[0], np.float32)
    for k in range(p.shape[1]):
        acc = (acc + p[:, k]).astype(np.float32)
    return acc
p32 = (x.astype(np.float64) * w.astype(np.float64)).astype(np.float32)  # fmaf: single rounding on add really; approx
# exact fmaf chain
def fma_chain(x, w):
    acc = np.zeros(x.shape[0], np.float64)
    for k in range(x.shape[1]):
        acc = (acc + x[:, k].astype(np.float64) * np.float64(w[k])).astype(np.float32).astype(np.float64)
    return acc
r_fma = fma_chain(x, w)
# fp16x2: weights prescaled to max in [2^13, 2^14)
s = 13 - int(np.floor(np.log2(np.abs(w).max())))
ws = (w * np.float32(2.0 ** s)).astype(np.float32)
A = ws.astype(np.float16); C = (ws - A.astype(np.float32)).astype(np.float16)
B = (A.astype(np.float32) * np.float32(2.0 ** -11)).astype(np.float16)
xh = x.astype(np.float16); xl = ((x - xh.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
assert np.isfinite(xh.astype(np.float32)).all()
def mfma_sum(terms, kblk=32):
    # per 32-k block exact-ish (fp64) then fp32 accumulate across blocks and products, smallest first
    acc = np.zeros(x.shape[0], np.float32)
    for k0 in range(0, K, kblk):
        for (a, b) in terms:
            blk = (a[:, k0:k0 + kblk].astype(np.float64) * b[k0:k0 + kblk].astype(np.float64)).sum(1)
            acc = (acc.astype(np.float64) + blk).astype(np.float32)
    return acc.astype(np.float64)
r_h2 = mfma_sum([(xl, B), (xh, C), (xh, A)]) * 2.0 ** -s
# bf16x3
def trunc_bf16(v):
    u = v.view(np.uint32) & np.uint32(0xffff0000)
    return u.view(np.float32)
def split3(v):
    h = trunc_bf16(v.copy()); r1 = (v - h).astype(np.float32); m = trunc_bf16(r1.copy()); r2 = (r1 - m).astype(np.float32); l = trunc_bf16(r2.copy())
    return h, m, l
xh3, xm3, xl3 = split3(x); wh3, wm3, wl3 = split3(w)
r_b3 = mfma_sum([(xh3, wl3), (xl3, wh3), (xm3, wm3), (xh3, wm3), (xm3, wh3), (xh3, wh3)])
for name, r in (("fmaf chain", r_fma), ("fp16x2 (3 products)", r_h2), ("bf16x3 (6 products)", r_b3)):
    e = r - truth
    rel = np.abs(e) / np.abs(x).max(1) / 0.02 / np.sqrt(K)
    print(f"{name:22s} rms err/scale_row {np.sqrt((rel**2).mean()):.3e} max {rel.max():.3e}")
