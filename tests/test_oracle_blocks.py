"""Secondary cross-checks of the oracle (oracle/q3tts_oracle.c) against torch CPU ops and the local
HF Qwen3 modules, block by block. These do not pin parity with the reference
(no reference-run fixture exists here: the oracle header says "parity unpinned"); they catch
restatement mistakes where the reference's semantics coincide with a well-known torch op."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle as O

L = O.lib()
rng = np.random.default_rng(1234)


def bf(x):  # f32 -> bf16 bits
    return O.f32_to_bf16(np.asarray(x, np.float32))


def f32(b):
    return O.bf16_to_f32(b)


def tbf(x):  # round a torch tensor through bf16
    return x.to(torch.bfloat16).to(torch.float32)


def test_bf16_rounding_matches_torch():
    x = rng.standard_normal(10000).astype(np.float32) * 10
    x[:4] = [0.0, -0.0, 1e-40, 3.0e38]
    got = f32(bf(x))
    exp = torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy()
    assert np.array_equal(got, exp)


def test_linear_bf16():
    M, K, N = 5, 384, 96
    x, W, b = bf(rng.standard_normal((M, K))), bf(rng.standard_normal((N, K)) * 0.05), bf(rng.standard_normal(N) * 0.1)
    out = np.empty((M, N), np.uint16)
    L.o_linear_bf16(O._p16(x), O._p16(W), O._p16(b), M, K, N, O._p16(out))
    ref = torch.from_numpy(f32(x)).double() @ torch.from_numpy(f32(W)).double().T + torch.from_numpy(f32(b)).double()
    # fp32 accumulation vs exact: at most one bf16 ulp apart after rounding
    err = np.abs(f32(out) - tbf(ref.float()).numpy())
    assert (err <= np.abs(ref.numpy()) * 2.0 ** -7 + 1e-6).all()


def test_rmsnorm_bf16_rounding_points():
    rows, dim = 7, 256
    x, w = bf(rng.standard_normal((rows, dim)) * 3), bf(1 + 0.1 * rng.standard_normal(dim))
    out = np.empty((rows, dim), np.uint16)
    L.o_rmsnorm_bf16(O._p16(x), O._p16(w), C.c_float(1e-6), rows, dim, O._p16(out))
    xf = torch.from_numpy(f32(x))
    n = tbf(xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6))  # MLX fast.rms_norm fallback: cast, then * w
    ref = tbf(n * torch.from_numpy(f32(w)))
    assert np.abs(f32(out) - ref.numpy()).max() <= 2.0 ** -6 * 4  # <= 1 bf16 ulp at |x| < 4


def test_rope_tables_match_hf_qwen3():
    from transformers.models.qwen3.configuration_qwen3 import Qwen3Config
    from transformers.models.qwen3.modeling_qwen3 import Qwen3RotaryEmbedding
    cfg = Qwen3Config(hidden_size=256, num_attention_heads=2, num_key_value_heads=1, head_dim=128, rope_theta=1e6,
                      max_position_embeddings=4096)
    rot = Qwen3RotaryEmbedding(cfg)
    pos = torch.arange(3, 40)[None]
    cos, sin = rot(torch.zeros(1, pos.shape[1], 256), pos)
    oc, osn = np.empty((37, 128), np.uint16), np.empty((37, 128), np.uint16)
    L.o_rope_tables(C.c_float(1e6), 128, 3, 37, O._p16(oc), O._p16(osn))
    assert np.abs(f32(oc) - tbf(cos[0]).numpy()).max() <= 2.0 ** -8
    assert np.abs(f32(osn) - tbf(sin[0]).numpy()).max() <= 2.0 ** -8


def _stack_weights(H, I, nh, nkv, hd, layers):
    w = {}
    for l in range(layers):
        p = f"m.layers.{l}"
        w[p + ".self_attn.q_proj.weight"] = bf(rng.standard_normal((nh * hd, H)) * 0.05)
        w[p + ".self_attn.k_proj.weight"] = bf(rng.standard_normal((nkv * hd, H)) * 0.05)
        w[p + ".self_attn.v_proj.weight"] = bf(rng.standard_normal((nkv * hd, H)) * 0.05)
        w[p + ".self_attn.o_proj.weight"] = bf(rng.standard_normal((H, nh * hd)) * 0.05)
        w[p + ".self_attn.q_norm.weight"] = bf(1 + 0.1 * rng.standard_normal(hd))
        w[p + ".self_attn.k_norm.weight"] = bf(1 + 0.1 * rng.standard_normal(hd))
        w[p + ".mlp.gate_proj.weight"] = bf(rng.standard_normal((I, H)) * 0.05)
        w[p + ".mlp.up_proj.weight"] = bf(rng.standard_normal((I, H)) * 0.05)
        w[p + ".mlp.down_proj.weight"] = bf(rng.standard_normal((H, I)) * 0.05)
        w[p + ".input_layernorm.weight"] = bf(1 + 0.1 * rng.standard_normal(H))
        w[p + ".post_attention_layernorm.weight"] = bf(1 + 0.1 * rng.standard_normal(H))
    w["m.norm.weight"] = bf(1 + 0.1 * rng.standard_normal(H))
    return w


def test_decoder_stack_matches_hf_qwen3_layer():
    """The talker / code-predictor block (Talker.swift:435-470) is the HF Qwen3 decoder layer when
    positions are sequential: QK-norm, rotate-half RoPE, GQA, SwiGLU, pre-norm residuals. Prefill of L
    tokens followed by single-token steps must agree with HF run on the whole sequence."""
    from transformers.models.qwen3.configuration_qwen3 import Qwen3Config
    from transformers.models.qwen3.modeling_qwen3 import Qwen3Model
    H, I, nh, nkv, hd, layers, T = 256, 512, 4, 2, 128, 2, 9
    w = _stack_weights(H, I, nh, nkv, hd, layers)
    holder = O._StackHolder(w, "m", H, [I] * layers, layers, nh, nkv, hd, 1e-6, 1e6)
    cfg = Qwen3Config(vocab_size=8, hidden_size=H, intermediate_size=I, num_hidden_layers=layers, num_attention_heads=nh,
                      num_key_value_heads=nkv, head_dim=hd, rms_norm_eps=1e-6, rope_theta=1e6, max_position_embeddings=512,
                      attention_bias=False, use_sliding_window=False)
    cfg._attn_implementation = "eager"
    hf = Qwen3Model(cfg).float().eval()
    sd = {}
    for k, v in w.items():
        sd[k[2:]] = torch.from_numpy(f32(v))
    sd["embed_tokens.weight"] = hf.embed_tokens.weight.detach()
    hf.load_state_dict(sd, strict=True)
    x = bf(rng.standard_normal((T, H)))
    with torch.no_grad():
        ref = hf(inputs_embeds=torch.from_numpy(f32(x))[None]).last_hidden_state[0].numpy()
    cache = holder.new_cache(T + 1)
    out_prefill = holder.forward(cache, x[:6])        # L > 1: causal mask path (Talker.swift:559-566)
    outs = [out_prefill] + [holder.forward(cache, x[i:i + 1]) for i in range(6, T)]  # decode steps
    got = f32(np.concatenate(outs, 0))
    # bf16 storage at every op vs fp32 HF: a few bf16 ulps of O(1) activations
    assert np.abs(got - ref).max() < 0.08, np.abs(got - ref).max()
    assert np.corrcoef(got.ravel(), ref.ravel())[0, 1] > 0.9995


@pytest.mark.parametrize("K,dil,groups", [(7, 1, 1), (7, 3, 1), (7, 9, 1), (3, 1, 1), (1, 1, 1), (7, 1, 16)])
def test_causal_conv1d(K, dil, groups):
    T, Cin, Cout = 50, 16, 24 if groups == 1 else 16
    x = rng.standard_normal((T, Cin)).astype(np.float32)
    W = (rng.standard_normal((Cout, K, Cin // groups)) * 0.2).astype(np.float32)  # MLX [O][K][I/g]
    b = rng.standard_normal(Cout).astype(np.float32)
    out = np.empty((T, Cout), np.float32)
    L.o_conv1d_causal(O._pf(x), O._pf(W), O._pf(b), T, Cin, Cout, K, dil, groups, O._pf(out))
    xt = F.pad(torch.from_numpy(x).T[None], ((K - 1) * dil, 0))  # left pad only (SpeechTokenizer.swift:298-301)
    ref = F.conv1d(xt, torch.from_numpy(W).permute(0, 2, 1).contiguous(), torch.from_numpy(b), dilation=dil, groups=groups)[0].T
    assert np.abs(out - ref.numpy()).max() < 2e-5


@pytest.mark.parametrize("K,stride", [(16, 8), (10, 5), (8, 4), (6, 3), (2, 2)])
def test_causal_transposed_conv_trims_right_only(K, stride):
    """ConvTransposed1d then drop the LAST K-s samples (SpeechTokenizer.swift:346-351). HF trims both
    sides; the reference does not (SURVEY.md section 8c known divergences)."""
    T, Cin, Cout = 11, 12, 20
    x = rng.standard_normal((T, Cin)).astype(np.float32)
    W = (rng.standard_normal((Cout, K, Cin)) * 0.2).astype(np.float32)  # MLX [O][K][I]
    b = rng.standard_normal(Cout).astype(np.float32)
    out = np.empty((T * stride, Cout), np.float32)
    L.o_convtr1d_causal(O._pf(x), O._pf(W), O._pf(b), T, Cin, Cout, K, stride, O._pf(out))
    wt = torch.from_numpy(W).permute(2, 0, 1).contiguous()  # torch ConvTranspose1d [in][out][k]
    full = F.conv_transpose1d(torch.from_numpy(x).T[None], wt, torch.from_numpy(b), stride=stride)[0].T
    assert full.shape[0] == (T - 1) * stride + K
    ref = full[: T * stride]
    assert np.abs(out - ref.numpy()).max() < 2e-5


def test_snake_layernorm_gelu_rms_f32():
    T, Cc = 9, 32
    x = rng.standard_normal((T, Cc)).astype(np.float32) * 2
    a, b = (rng.standard_normal(Cc) * 0.3).astype(np.float32), (rng.standard_normal(Cc) * 0.3).astype(np.float32)
    out = np.empty_like(x)
    L.o_snake(O._pf(x), O._pf(a), O._pf(b), T, Cc, O._pf(out))
    xt, at, bt = torch.from_numpy(x), torch.from_numpy(a), torch.from_numpy(b)
    ref = xt + (1.0 / (bt.exp() + 1e-9)) * torch.sin(xt * at.exp()) ** 2  # SpeechTokenizer.swift:246-253
    assert np.abs(out - ref.numpy()).max() < 1e-5
    w, bb = rng.standard_normal(Cc).astype(np.float32), rng.standard_normal(Cc).astype(np.float32)
    L.o_layernorm_f32(O._pf(x), O._pf(w), O._pf(bb), C.c_float(1e-6), T, Cc, O._pf(out))
    assert np.abs(out - F.layer_norm(xt, (Cc,), torch.from_numpy(w), torch.from_numpy(bb), 1e-6).numpy()).max() < 1e-5
    L.o_gelu_f32(O._pf(x), C.c_int64(x.size), O._pf(out))
    assert np.abs(out - F.gelu(xt).numpy()).max() < 1e-6
    L.o_rmsnorm_f32(O._pf(x), O._pf(w), C.c_float(1e-5), T, Cc, O._pf(out))
    ref = xt * torch.rsqrt(xt.pow(2).mean(-1, keepdim=True) + 1e-5) * torch.from_numpy(w)
    assert np.abs(out - ref.numpy()).max() < 1e-5


def test_full_attention_no_mask_no_positions():
    T, heads, D = 13, 3, 64
    q, k, v = (rng.standard_normal((T, heads, D)).astype(np.float32) for _ in range(3))
    out = np.empty_like(q)
    L.o_attention_full_f32(O._pf(q), O._pf(k), O._pf(v), T, heads, D, O._pf(out))
    tq, tk, tv = (torch.from_numpy(a).permute(1, 0, 2)[None] for a in (q, k, v))
    ref = F.scaled_dot_product_attention(tq, tk, tv)[0].permute(1, 0, 2)
    assert np.abs(out - ref.numpy()).max() < 1e-5


def test_q3_log_exp_accuracy():
    L.o_expf.restype = C.c_float
    L.o_expf.argtypes = [C.c_float]
    for x in np.concatenate([rng.uniform(1e-6, 1.0, 200), rng.uniform(1, 50, 50)]):
        assert abs(L.o_logf(float(x)) - math.log(np.float32(x))) <= 2e-6 * max(1.0, abs(math.log(x)))
    for x in rng.uniform(-30, 10, 200):
        assert abs(L.o_expf(float(x)) - math.exp(np.float32(x))) <= 3e-7 * math.exp(x) + 1e-30


def _sample(logits, **kw):
    lr = np.ascontiguousarray(logits, np.uint16)
    seen = kw.get("seen")
    return int(L.o_sample_token(O._p16(lr), lr.size, C.c_float(kw.get("temperature", 0.0)), kw.get("top_k", 50),
                                C.c_float(kw.get("top_p", 1.0)), C.c_float(kw.get("rep", 1.0)),
                                seen.ctypes.data_as(O.u8p) if seen is not None else None, kw.get("slo", 0), kw.get("shi", 0),
                                kw.get("eos", -1), kw.get("mask_eos", 0), C.c_uint64(kw.get("seed", 0)), kw.get("row", 0),
                                kw.get("draw", 0)))


def test_sampler_greedy_suppress_and_penalty():
    V = 3072
    lg = (rng.standard_normal(V) * 0.5).astype(np.float32)
    lg[2500] = 9.0   # inside the suppressed range [V-1024, V)
    lg[2150] = 5.0   # EOS stays available (Qwen3.swift:829-835)
    lg[100] = 4.0
    lg[7] = 4.0      # tie with a larger index: first maximum wins
    b = bf(lg)
    assert _sample(b) == 2500
    assert _sample(b, slo=V - 1024, shi=V, eos=2150) == 2150
    assert _sample(b, slo=V - 1024, shi=V, eos=2150, mask_eos=1) == 7
    seen = np.zeros(V, np.uint8)
    seen[7] = 1  # penalised: 4.0 / bf16(1.05) < 4.0 -> token 100 now wins
    assert _sample(b, slo=V - 1024, shi=V, eos=2150, mask_eos=1, seen=seen, rep=1.05) == 100
    neg = bf(np.full(V, -3.0, np.float32))
    seen2 = np.zeros(V, np.uint8)
    seen2[0] = 1     # negative logits are multiplied (Qwen3.swift:171-175): -3 * 1.05 < -3
    assert _sample(neg, seen=seen2, rep=1.05) == 1


def test_sampler_topk_support_and_distribution():
    V, k = 64, 5
    lg = np.linspace(-2, 2, V).astype(np.float32)
    b = bf(lg)
    draws = [_sample(b, temperature=0.9, top_k=k, seed=11, draw=d) for d in range(4000)]
    assert set(draws) <= set(range(V - k, V))           # only the k largest survive (Qwen3.swift:68-89)
    # categorical(logits / T): frequencies follow softmax over the kept logits
    kept = f32(bf(f32(b)[-k:] * f32(bf(np.float32(1 / 0.9)))))
    p = np.exp(kept - kept.max())
    p /= p.sum()
    freq = np.bincount(np.array(draws) - (V - k), minlength=k) / len(draws)
    assert np.abs(freq - p).max() < 0.03
    # same seed and draw index -> same token; EOS logit survives top-k (Qwen3.swift:188-207)
    assert _sample(b, temperature=0.9, top_k=k, seed=3, draw=9) == _sample(b, temperature=0.9, top_k=k, seed=3, draw=9)
    lg2 = lg.copy()
    lg2[0] = 2.5
    got = {_sample(bf(lg2), temperature=0.9, top_k=1, eos=3, seed=5, draw=d) for d in range(300)}
    assert got <= {0, 3} and 0 in got


def test_sampler_topp_keeps_the_head():
    V = 32
    lg = np.full(V, -8.0, np.float32)
    lg[[3, 9]] = [2.0, 1.5]
    draws = {_sample(bf(lg), temperature=1.0, top_k=0, top_p=0.9, seed=1, draw=d) for d in range(300)}
    assert draws <= {3, 9}
