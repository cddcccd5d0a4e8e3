/*
 * q3tts_oracle.c -- CPU restatement of the Qwen3-TTS hot path (TEST INFRASTRUCTURE, NOT PRODUCT).
 *
 * This file is the parity oracle for the HIP engine under swift-qwen3-tts_amd/csrc. Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it. The product path never
 * links, imports or falls back to anything in oracle/.
 *
 * PARITY UNPINNED: the reference (Swift on mlx-swift 0.29.1 / mlx-swift-examples 2.29.1, see
 * /root/reference/Package.resolved:13-28) cannot be built or run here (no Swift toolchain, no MLX,
 * no checkpoints), and its only known-answer test needs real 1.7B weights
 * (Tests/Qwen3TTSTests/Qwen3TTSTests.swift:25-282). The arithmetic below restates the Swift
 * sources line by line and, for MLX-internal semantics the Swift does not spell out, the
 * documented MLX behaviour (fp32 reduction in rms_norm / softmax / matmul accumulate, results
 * rounded to the array dtype after every op). It is cross-checked block by block against
 * torch CPU ops in tests/test_oracle_blocks.py (a secondary check, not a pin).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/Sources/Qwen3TTS/Models/).
 *
 * Arithmetic contract ("rounding edges"), shared with DESIGN.md section 3:
 *   talker / code predictor: storage bf16, every op accumulates in fp32 and rounds its result to
 *   bf16 once (RNE); codec decoder: fp32 throughout.
 *
 * Build: see oracle/Makefile (gcc -O3 -fopenmp -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define O_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------ */
/* bf16 helpers                                                                               */
/* ------------------------------------------------------------------------------------------ */
static inline float bf2f(uint16_t h) {
    uint32_t u = ((uint32_t)h) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
/* round-to-nearest-even; NaN stays NaN (quiet) */
static inline uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float rbf(float f) { return bf2f(f2bf(f)); }

O_API int o_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
O_API void o_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
O_API void o_f32_to_bf16(const float* x, int64_t n, uint16_t* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = f2bf(x[i]);
}
O_API void o_bf16_to_f32(const uint16_t* x, int64_t n, float* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = bf2f(x[i]);
}

/* ------------------------------------------------------------------------------------------ */
/* bf16 LM blocks                                                                             */
/* ------------------------------------------------------------------------------------------ */

/* RMSNorm as MLXNN.RMSNorm -> MLXFast.rmsNorm (call sites Talker.swift:189-190,447-448,520;
 * CodePredictor.swift:95-96,174-175,226): fp32 reduction; x*rsqrt(mean(x^2)+eps) rounded to the
 * activation dtype, then multiplied by the weight in the activation dtype. */
O_API void o_rmsnorm_bf16(const uint16_t* x, const uint16_t* w, float eps, int rows, int dim,
                          uint16_t* out) {
    for (int r = 0; r < rows; ++r) {
        const uint16_t* xr = x + (size_t)r * dim;
        float ss = 0.f;
        for (int i = 0; i < dim; ++i) {
            float v = bf2f(xr[i]);
            ss += v * v;
        }
        float rstd = 1.0f / sqrtf(ss / (float)dim + eps);
        for (int i = 0; i < dim; ++i) {
            float n = rbf(bf2f(xr[i]) * rstd);
            out[(size_t)r * dim + i] = f2bf(n * bf2f(w[i]));
        }
    }
}

/* y = x W^T (+ b): MLXNN.Linear (Talker.swift:183-186,413-415,480-481,607;
 * CodePredictor.swift:90-93,152-154,296,305). fp32 accumulate over k in index order, bias added
 * in fp32, one rounding to bf16. W is [N][K] row-major (out, in). */
/* Test switch: 0 = sum over k in index order (the restatement), 1 = in reverse order. MLX's own accumulation order is
 * not visible from the Swift, so any fp32 order is an equally valid reading of "fp32 accumulate"; the distance between
 * the two orders after 28 layers of bf16 roundings is the noise floor the full-size parity tests measure their
 * tolerance against (tests/test_full_size.py). */
static int g_sum_order = 0;
O_API void o_set_sum_order(int mode) { g_sum_order = mode; }

O_API void o_linear_bf16(const uint16_t* x, const uint16_t* W, const uint16_t* bias, int M, int K,
                         int N, uint16_t* out) {
    const int rev = g_sum_order;
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n) {
        const uint16_t* wr = W + (size_t)n * K;
        for (int m = 0; m < M; ++m) {
            const uint16_t* xr = x + (size_t)m * K;
            float acc = 0.f;
            if (rev) for (int k = K - 1; k >= 0; --k) acc += bf2f(xr[k]) * bf2f(wr[k]);
            else for (int k = 0; k < K; ++k) acc += bf2f(xr[k]) * bf2f(wr[k]);
            if (bias) acc += bf2f(bias[n]);
            out[(size_t)m * N + n] = f2bf(acc);
        }
    }
}

/* MLX affine int4 (group 64) Linear: QuantizedLinear installed by quantize(model:...) at
 * Qwen3.swift:1412-1425. w[n][k] = q*scale + bias with q the k-th nibble (little-endian, 8 per
 * uint32) of row n; scales/biases are bf16 [N][K/group]. Dequantised weight is rounded to bf16
 * (MLX dequantises into the activation dtype), fp32 accumulate. */
O_API void o_qlinear_bf16(const uint16_t* x, const uint32_t* Wq, const uint16_t* scales,
                          const uint16_t* qbiases, const uint16_t* bias, int M, int K, int N,
                          int group, uint16_t* out) {
    int gpr = K / group;
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n) {
        const uint32_t* wr = Wq + (size_t)n * (K / 8);
        for (int m = 0; m < M; ++m) {
            const uint16_t* xr = x + (size_t)m * K;
            float acc = 0.f;
            for (int kk = 0; kk < K; ++kk) {
                const int k = g_sum_order ? K - 1 - kk : kk;
                uint32_t q = (wr[k >> 3] >> (4 * (k & 7))) & 0xFu;
                int g = k / group;
                float w = rbf((float)q * bf2f(scales[(size_t)n * gpr + g]) +
                              bf2f(qbiases[(size_t)n * gpr + g]));
                acc += bf2f(xr[k]) * w;
            }
            if (bias) acc += bf2f(bias[n]);
            out[(size_t)m * N + n] = f2bf(acc);
        }
    }
}

static inline float silu_f(float v) { return v / (1.0f + expf(-v)); }

/* silu(gate) * up: Talker.swift:420, CodePredictor.swift:158. silu result rounded to bf16, the
 * product rounded again (two array ops in the reference). */
O_API void o_silu_mul_bf16(const uint16_t* gate, const uint16_t* up, int64_t n, uint16_t* out) {
    for (int64_t i = 0; i < n; ++i) {
        float s = rbf(silu_f(bf2f(gate[i])));
        out[i] = f2bf(s * bf2f(up[i]));
    }
}
/* silu alone: ResizeMLP, Talker.swift:485 */
O_API void o_silu_bf16(const uint16_t* x, int64_t n, uint16_t* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = f2bf(silu_f(bf2f(x[i])));
}
/* elementwise add in bf16: residual adds Talker.swift:461,466; embedding sums Qwen3.swift:379,390,
 * 721-728 */
O_API void o_add_bf16(const uint16_t* a, const uint16_t* b, int64_t n, uint16_t* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = f2bf(bf2f(a[i]) + bf2f(b[i]));
}

/* RoPE tables: TalkerRotaryEmbedding (Talker.swift:42-44,103-117) with all three position rows
 * equal (Talker.swift:93-96 => the interleave at :76-79 is the identity), and
 * CodePredictorRotaryEmbedding (CodePredictor.swift:38-39,44-56). inv_freq = 1/pow(base, i/dim),
 * i = 0,2,4..; angle = pos*inv_freq in fp32; emb = cat(f,f); cos/sin cast to bf16. */
O_API void o_rope_tables(float base, int dim, int pos0, int npos, uint16_t* cos_out,
                         uint16_t* sin_out) {
    int half = dim / 2;
    for (int p = 0; p < npos; ++p) {
        for (int i = 0; i < half; ++i) {
            float inv = 1.0f / powf(base, (float)(2 * i) / (float)dim);
            float ang = (float)(pos0 + p) * inv;
            uint16_t c = f2bf(cosf(ang)), s = f2bf(sinf(ang));
            cos_out[(size_t)p * dim + i] = c;
            cos_out[(size_t)p * dim + half + i] = c;
            sin_out[(size_t)p * dim + i] = s;
            sin_out[(size_t)p * dim + half + i] = s;
        }
    }
}

/* applyRotaryPosEmb + rotateHalf (Talker.swift:125-152): q*cos + [-q2, q1]*sin, each array op
 * rounded to bf16. x: [heads][dim] for one position, in place. */
static void rope_apply(uint16_t* x, int heads, int dim, const uint16_t* cosr, const uint16_t* sinr) {
    int half = dim / 2;
    uint16_t tmp[512];
    for (int h = 0; h < heads; ++h) {
        uint16_t* v = x + (size_t)h * dim;
        for (int i = 0; i < dim; ++i) {
            float rot = (i < half) ? -bf2f(v[i + half]) : bf2f(v[i - half]);
            float a = rbf(bf2f(v[i]) * bf2f(cosr[i]));
            float b = rbf(rot * bf2f(sinr[i]));
            tmp[i] = f2bf(a + b);
        }
        memcpy(v, tmp, (size_t)dim * 2);
    }
}

/* One Qwen3 decoder stack (talker: Talker.swift:435-470,157-241,402-430,532-574;
 * code predictor: CodePredictor.swift:64-196,236-268). Pointers are arrays over layers. */
typedef struct {
    int hidden, n_layers, n_heads, n_kv, head_dim;
    float eps, rope_base;
    const int* inter;            /* per-layer intermediate size (Talker.swift:514-518) */
    const uint16_t** ln1;        /* input_layernorm.weight [hidden] */
    const uint16_t** ln2;        /* post_attention_layernorm.weight */
    const uint16_t** qw;         /* [n_heads*head_dim][hidden] */
    const uint16_t** kw;         /* [n_kv*head_dim][hidden] */
    const uint16_t** vw;
    const uint16_t** ow;         /* [hidden][n_heads*head_dim] */
    const uint16_t** qn;         /* q_norm.weight [head_dim] */
    const uint16_t** kn;
    const uint16_t** gw;         /* gate [inter][hidden] */
    const uint16_t** uw;
    const uint16_t** dw;         /* down [hidden][inter] */
    const uint16_t* norm;        /* final norm [hidden] */
} o_stack;

/* KVCacheSimple restated (MLXLMCommon; used at Talker.swift:224-226,577): append-only
 * contiguous K,V per layer; offset = tokens seen. k,v: [layer][cap][n_kv][head_dim]. */
typedef struct {
    uint16_t* k;
    uint16_t* v;
    int len, cap;
} o_cache;

/* Attention for new tokens i=0..L-1 at positions len+i over keys 0..len+i (additive causal mask
 * only when L>1: Talker.swift:559-566). MLXFast.scaledDotProductAttention (Talker.swift:229):
 * softmax(scale*q.k) in fp32, GQA by head repetition, output rounded to bf16. */
static void attn_tokens(const o_stack* s, const uint16_t* q, const uint16_t* kc, const uint16_t* vc,
                        int len, int L, uint16_t* out) {
    int D = s->head_dim, Hq = s->n_heads, Hk = s->n_kv, rep = Hq / Hk;
    float scale = powf((float)D, -0.5f);
#pragma omp parallel for collapse(2) schedule(static)
    for (int i = 0; i < L; ++i) {
        for (int h = 0; h < Hq; ++h) {
            int T = len + i + 1, kh = h / rep;
            const uint16_t* qv = q + ((size_t)i * Hq + h) * D;
            float* sc = (float*)malloc(sizeof(float) * (size_t)T);
            float mx = -INFINITY;
            for (int t = 0; t < T; ++t) {
                const uint16_t* kv = kc + ((size_t)t * Hk + kh) * D;
                float a = 0.f;
                for (int d = 0; d < D; ++d) a += bf2f(qv[d]) * bf2f(kv[d]);
                a *= scale;
                sc[t] = a;
                if (a > mx) mx = a;
            }
            float sum = 0.f;
            for (int t = 0; t < T; ++t) {
                sc[t] = expf(sc[t] - mx);
                sum += sc[t];
            }
            float inv = 1.0f / sum;
            for (int d = 0; d < D; ++d) {
                float a = 0.f;
                for (int t = 0; t < T; ++t) a += sc[t] * bf2f(vc[((size_t)t * Hk + kh) * D + d]);
                out[((size_t)i * Hq + h) * D + d] = f2bf(a * inv);
            }
            free(sc);
        }
    }
}

/* Forward L new tokens through the stack, appending to the cache. x: [L][hidden] bf16.
 * out: [L][hidden] after the final norm (Talker.swift:573 / CodePredictor.swift:267). */
O_API int o_stack_forward(const o_stack* s, o_cache* c, const uint16_t* x, int L, uint16_t* out) {
    int H = s->hidden, D = s->head_dim, Hq = s->n_heads, Hk = s->n_kv;
    int qd = Hq * D, kd = Hk * D;
    if (c->len + L > c->cap) return -1;
    size_t layer_stride = (size_t)c->cap * kd;
    uint16_t* h = (uint16_t*)malloc((size_t)L * H * 2);
    uint16_t* xn = (uint16_t*)malloc((size_t)L * H * 2);
    uint16_t* q = (uint16_t*)malloc((size_t)L * qd * 2);
    uint16_t* k = (uint16_t*)malloc((size_t)L * kd * 2);
    uint16_t* v = (uint16_t*)malloc((size_t)L * kd * 2);
    uint16_t* ao = (uint16_t*)malloc((size_t)L * qd * 2);
    uint16_t* y = (uint16_t*)malloc((size_t)L * H * 2);
    uint16_t* cosr = (uint16_t*)malloc((size_t)L * D * 2);
    uint16_t* sinr = (uint16_t*)malloc((size_t)L * D * 2);
    memcpy(h, x, (size_t)L * H * 2);
    o_rope_tables(s->rope_base, D, c->len, L, cosr, sinr);
    for (int l = 0; l < s->n_layers; ++l) {
        int I = s->inter[l];
        uint16_t* g = (uint16_t*)malloc((size_t)L * I * 2);
        uint16_t* u = (uint16_t*)malloc((size_t)L * I * 2);
        /* Talker.swift:458-461 */
        o_rmsnorm_bf16(h, s->ln1[l], s->eps, L, H, xn);
        o_linear_bf16(xn, s->qw[l], NULL, L, H, qd, q);
        o_linear_bf16(xn, s->kw[l], NULL, L, H, kd, k);
        o_linear_bf16(xn, s->vw[l], NULL, L, H, kd, v);
        /* per-head QK RMSNorm before RoPE: Talker.swift:207-213 */
        o_rmsnorm_bf16(q, s->qn[l], s->eps, L * Hq, D, q);
        o_rmsnorm_bf16(k, s->kn[l], s->eps, L * Hk, D, k);
        for (int i = 0; i < L; ++i) {
            rope_apply(q + (size_t)i * qd, Hq, D, cosr + (size_t)i * D, sinr + (size_t)i * D);
            rope_apply(k + (size_t)i * kd, Hk, D, cosr + (size_t)i * D, sinr + (size_t)i * D);
        }
        uint16_t* kc = c->k + (size_t)l * layer_stride;
        uint16_t* vc = c->v + (size_t)l * layer_stride;
        memcpy(kc + (size_t)c->len * kd, k, (size_t)L * kd * 2);
        memcpy(vc + (size_t)c->len * kd, v, (size_t)L * kd * 2);
        attn_tokens(s, q, kc, vc, c->len, L, ao);
        o_linear_bf16(ao, s->ow[l], NULL, L, qd, H, y);
        o_add_bf16(h, y, (int64_t)L * H, h);
        /* Talker.swift:463-466 */
        o_rmsnorm_bf16(h, s->ln2[l], s->eps, L, H, xn);
        o_linear_bf16(xn, s->gw[l], NULL, L, H, I, g);
        o_linear_bf16(xn, s->uw[l], NULL, L, H, I, u);
        o_silu_mul_bf16(g, u, (int64_t)L * I, g);
        o_linear_bf16(g, s->dw[l], NULL, L, I, H, y);
        o_add_bf16(h, y, (int64_t)L * H, h);
        free(g);
        free(u);
    }
    o_rmsnorm_bf16(h, s->norm, s->eps, L, H, out);
    c->len += L;
    free(h); free(xn); free(q); free(k); free(v); free(ao); free(y); free(cosr); free(sinr);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Sampler                                                                                    */
/* ------------------------------------------------------------------------------------------ */

/* Philox4x32-10 (Salmon et al. 2011), the engine's counter-based RNG. The reference draws from
 * MLX's global key (no seed API: Qwen3.swift:120-126), so sampled streams are specified by this
 * build, not by the reference; greedy decoding does not touch the RNG. */
static inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                 uint32_t k1, uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* natural log built from +,*,fma and bit operations only (Cephes logf coefficients), so that the
 * HIP sampler can reproduce it bit for bit. Valid for normal positive x. */
static inline float q3_logf(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    int e = (int)((u >> 23) & 0xff) - 126;
    u = (u & 0x007fffffu) | 0x3f000000u; /* mantissa in [0.5,1) */
    float m;
    memcpy(&m, &u, 4);
    if (m < 0.70710678118654752440f) {
        e -= 1;
        m = m + m;
    }
    float f = m - 1.0f;
    float z = f * f;
    float y = 7.0376836292E-2f;
    y = fmaf(y, f, -1.1514610310E-1f);
    y = fmaf(y, f, 1.1676998740E-1f);
    y = fmaf(y, f, -1.2420140846E-1f);
    y = fmaf(y, f, 1.4249322787E-1f);
    y = fmaf(y, f, -1.6668057665E-1f);
    y = fmaf(y, f, 2.0000714765E-1f);
    y = fmaf(y, f, -2.4999993993E-1f);
    y = fmaf(y, f, 3.3333331174E-1f);
    y = y * f;
    y = y * z;
    float fe = (float)e;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(z, -0.5f, y);
    float r = f + y;
    r = fmaf(fe, 0.693359375f, r);
    return r;
}

/* exp with the same construction rules (Cephes expf coefficients); used by top-p */
static inline float q3_expf(float x) {
    if (x < -87.0f) return 0.0f;
    if (x > 88.0f) return INFINITY;
    float fx = floorf(fmaf(x, 1.44269504088896341f, 0.5f));
    float r = fmaf(fx, -0.693359375f, x);
    r = fmaf(fx, 2.12194440e-4f, r);
    float z = r * r;
    float y = 1.9875691500E-4f;
    y = fmaf(y, r, 1.3981999507E-3f);
    y = fmaf(y, r, 8.3334519073E-3f);
    y = fmaf(y, r, 4.1665795894E-2f);
    y = fmaf(y, r, 1.6666665459E-1f);
    y = fmaf(y, r, 5.0000001201E-1f);
    y = fmaf(y, z, r);
    y = y + 1.0f;
    int n = (int)fx;
    uint32_t sb = (uint32_t)(n + 127) << 23;
    float sc;
    memcpy(&sc, &sb, 4);
    return y * sc;
}
O_API float o_expf(float x) { return q3_expf(x); }

/* Gumbel noise for element i of row `row` at draw `draw` (mx.random.categorical =
 * argmax(logits + gumbel), called from Qwen3.swift:124-125). u in (0,1) from 24 random bits. */
static inline float gumbel_noise(uint64_t seed, uint32_t row, uint32_t draw, uint32_t i) {
    uint32_t r[4];
    philox4x32_10(i, row, draw, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    float u = ((float)(r[0] >> 8) + 0.5f) * 5.9604644775390625e-08f; /* 2^-24 */
    return -q3_logf(-q3_logf(u));
}
O_API float o_gumbel(uint64_t seed, uint32_t row, uint32_t draw, uint32_t i) {
    return gumbel_noise(seed, row, draw, i);
}
O_API float o_logf(float x) { return q3_logf(x); }

/* order key: larger value first, lower index first among equals */
static inline int before_desc(float va, int ia, float vb, int ib) {
    return (va > vb) || (va == vb && ia < ib);
}

/* sampleToken (Qwen3.swift:130-213) on one row of bf16 logits. All array arithmetic is in the
 * logits dtype (bf16): Float scalars are converted to bf16 arrays by the mlx-swift operators.
 *   seen:      uint8[V], 1 for every previously generated first-codebook token (the Set at :165)
 *   suppress:  [lo,hi) range set to -inf except eos (:153-161 with the list built at :829-835)
 *   temperature <= 0 -> argmax (first maximum) (:182-185)
 * Returns the token id. */
O_API int o_sample_token(const uint16_t* logits, int V, float temperature, int top_k, float top_p,
                         float rep_penalty, const uint8_t* seen, int suppress_lo, int suppress_hi,
                         int eos_id, int mask_eos, uint64_t seed, uint32_t row, uint32_t draw) {
    float* l = (float*)malloc(sizeof(float) * (size_t)V);
    for (int i = 0; i < V; ++i) l[i] = bf2f(logits[i]);
    /* 1. suppress */
    for (int i = suppress_lo; i < suppress_hi && i < V; ++i)
        if (i >= 0 && i != eos_id) l[i] = -INFINITY;
    if (mask_eos && eos_id >= 0 && eos_id < V) l[eos_id] = -INFINITY; /* bench force_frames only */
    /* 2. repetition penalty (:164-179) */
    if (seen && rep_penalty != 1.0f) {
        float p = rbf(rep_penalty);
        for (int i = 0; i < V; ++i)
            if (seen[i]) l[i] = (l[i] < 0.f) ? rbf(l[i] * p) : rbf(l[i] / p);
    }
    int tok = 0;
    if (temperature <= 0.f) { /* 3. greedy */
        for (int i = 1; i < V; ++i)
            if (l[i] > l[tok]) tok = i;
        free(l);
        return tok;
    }
    /* 4. save EOS logit (:188-191) */
    int have_eos = (eos_id >= 0 && eos_id < V);
    float eos_logit = have_eos ? l[eos_id] : 0.f;
    /* 5. top-k (:68-89): keep the k largest, ties resolved by lower index */
    if (top_k > 0 && top_k < V) {
        /* two passes so ranks are computed on the unmodified row */
        uint8_t* drop = (uint8_t*)calloc((size_t)V, 1);
        for (int i = 0; i < V; ++i) {
            int rank = 0;
            for (int j = 0; j < V; ++j)
                if (j != i && before_desc(l[j], j, l[i], i)) rank++;
            drop[i] = (uint8_t)(rank >= top_k);
        }
        for (int i = 0; i < V; ++i)
            if (drop[i]) l[i] = -INFINITY;
        free(drop);
    }
    /* 6. top-p (:92-117): probs = exp(logits) (not normalised, as written), ascending sort,
     * cumulative sum, keep where cumsum > 1 - top_p */
    if (top_p > 0.f && top_p < 1.0f) {
        int* idx = (int*)malloc(sizeof(int) * (size_t)V);
        for (int i = 0; i < V; ++i) idx[i] = i;
        /* ascending by (value, index): insertion sort is fine at V <= 4096 */
        for (int i = 1; i < V; ++i) {
            int t = idx[i], j = i - 1;
            while (j >= 0 && (l[idx[j]] > l[t] || (l[idx[j]] == l[t] && idx[j] > t))) {
                idx[j + 1] = idx[j];
                --j;
            }
            idx[j + 1] = t;
        }
        float thr = rbf(1.0f - top_p);
        float run = 0.f;
        uint8_t* keep = (uint8_t*)calloc((size_t)V, 1);
        for (int r = 0; r < V; ++r) {
            run += rbf(q3_expf(l[idx[r]]));
            keep[idx[r]] = (uint8_t)(rbf(run) > thr);
        }
        for (int i = 0; i < V; ++i)
            if (!keep[i]) l[i] = -INFINITY;
        free(keep);
        free(idx);
    }
    /* 7. restore EOS (:204-207) */
    if (have_eos && !mask_eos) l[eos_id] = eos_logit;
    /* 8. categorical(logits * (1/T)) (:120-126): scale in bf16, add fp32 Gumbel noise, argmax */
    float invt = rbf(1.0f / temperature);
    float best = -INFINITY;
    tok = -1;
    for (int i = 0; i < V; ++i) {
        if (l[i] == -INFINITY) continue;
        float v = rbf(l[i] * invt) + gumbel_noise(seed, row, draw, (uint32_t)i);
        if (tok < 0 || v > best) {
            best = v;
            tok = i;
        }
    }
    if (tok < 0) tok = 0;
    free(l);
    return tok;
}

/* ------------------------------------------------------------------------------------------ */
/* Codec decoder blocks (fp32, channels-last [T][C] like the MLX NLC arrays)                  */
/* ------------------------------------------------------------------------------------------ */

/* CausalConv1d (SpeechTokenizer.swift:259-306): left zero-pad (K-1)*dil, stride 1.
 * x [T][Cin], W [Cout][K][Cin/groups] (MLX layout), out [T][Cout]. */
O_API void o_conv1d_causal(const float* x, const float* W, const float* bias, int T, int Cin,
                           int Cout, int K, int dil, int groups, float* out) {
    int cig = Cin / groups, cog = Cout / groups;
#pragma omp parallel for schedule(static)
    for (int t = 0; t < T; ++t) {
        for (int co = 0; co < Cout; ++co) {
            int g = co / cog;
            float acc = 0.f;
            for (int k = 0; k < K; ++k) {
                int ti = t - (K - 1 - k) * dil;
                if (ti < 0) continue;
                const float* xr = x + (size_t)ti * Cin + (size_t)g * cig;
                const float* wr = W + ((size_t)co * K + k) * cig;
                for (int ci = 0; ci < cig; ++ci) acc += xr[ci] * wr[ci];
            }
            if (bias) acc += bias[co];
            out[(size_t)t * Cout + co] = acc;
        }
    }
}

/* CausalTransposeConv1d (SpeechTokenizer.swift:311-354): ConvTransposed1d(padding 0) producing
 * (T-1)*s+K samples, then the last K-s are dropped => T*s. W [Cout][K][Cin] (MLX layout):
 * y[t*s + k][co] += x[t][ci] * W[co][k][ci]. */
O_API void o_convtr1d_causal(const float* x, const float* W, const float* bias, int T, int Cin,
                             int Cout, int K, int stride, float* out) {
    int Tfull = (T - 1) * stride + K, Tout = T * stride;
    float* full = (float*)calloc((size_t)Tfull * Cout, sizeof(float));
#pragma omp parallel for schedule(static)
    for (int co = 0; co < Cout; ++co) {
        for (int t = 0; t < T; ++t) {
            const float* xr = x + (size_t)t * Cin;
            for (int k = 0; k < K; ++k) {
                const float* wr = W + ((size_t)co * K + k) * Cin;
                float acc = 0.f;
                for (int ci = 0; ci < Cin; ++ci) acc += xr[ci] * wr[ci];
                full[(size_t)(t * stride + k) * Cout + co] += acc;
            }
        }
    }
    for (int t = 0; t < Tout; ++t)
        for (int co = 0; co < Cout; ++co)
            out[(size_t)t * Cout + co] = full[(size_t)t * Cout + co] + (bias ? bias[co] : 0.f);
    free(full);
}

/* SnakeBeta (SpeechTokenizer.swift:232-254): x + 1/(exp(beta)+1e-9) * sin^2(x*exp(alpha)) */
O_API void o_snake(const float* x, const float* alpha, const float* beta, int T, int C, float* out) {
#pragma omp parallel for schedule(static)
    for (int t = 0; t < T; ++t)
        for (int c = 0; c < C; ++c) {
            float v = x[(size_t)t * C + c];
            float s = sinf(v * expf(alpha[c]));
            out[(size_t)t * C + c] = v + (1.0f / (expf(beta[c]) + 1e-9f)) * (s * s);
        }
}

/* fp32 Linear (codec transformer / ConvNeXt pointwise: SpeechTokenizer.swift:380-381,506-509,
 * 555-557,620-621). W [N][K]. */
O_API void o_linear_f32(const float* x, const float* W, const float* bias, int M, int K, int N,
                        float* out) {
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            float acc = 0.f;
            const float* xr = x + (size_t)m * K;
            const float* wr = W + (size_t)n * K;
            for (int k = 0; k < K; ++k) acc += xr[k] * wr[k];
            out[(size_t)m * N + n] = acc + (bias ? bias[n] : 0.f);
        }
}

/* fp32 RMSNorm (codec transformer, eps 1e-5: SpeechTokenizer.swift:581-582,626) */
O_API void o_rmsnorm_f32(const float* x, const float* w, float eps, int rows, int dim, float* out) {
    for (int r = 0; r < rows; ++r) {
        float ss = 0.f;
        for (int i = 0; i < dim; ++i) ss += x[(size_t)r * dim + i] * x[(size_t)r * dim + i];
        float rstd = 1.0f / sqrtf(ss / (float)dim + eps);
        for (int i = 0; i < dim; ++i) out[(size_t)r * dim + i] = (x[(size_t)r * dim + i] * rstd) * w[i];
    }
}

/* LayerNorm (ConvNeXt, eps 1e-6: SpeechTokenizer.swift:379,393) */
O_API void o_layernorm_f32(const float* x, const float* w, const float* b, float eps, int rows,
                           int dim, float* out) {
    for (int r = 0; r < rows; ++r) {
        const float* xr = x + (size_t)r * dim;
        float mean = 0.f;
        for (int i = 0; i < dim; ++i) mean += xr[i];
        mean /= (float)dim;
        float var = 0.f;
        for (int i = 0; i < dim; ++i) var += (xr[i] - mean) * (xr[i] - mean);
        var /= (float)dim;
        float rstd = 1.0f / sqrtf(var + eps);
        for (int i = 0; i < dim; ++i) out[(size_t)r * dim + i] = (xr[i] - mean) * rstd * w[i] + b[i];
    }
}

/* gelu, exact erf form (MLXNN gelu; SpeechTokenizer.swift:395) */
O_API void o_gelu_f32(const float* x, int64_t n, float* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = 0.5f * x[i] * (1.0f + erff(x[i] * 0.70710678118654752440f));
}
O_API void o_silu_mul_f32(const float* g, const float* u, int64_t n, float* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = silu_f(g[i]) * u[i];
}

/* Scaled-dot-product attention, q,k,v [T][heads][D] -> out [T][heads][D]; causal = additive
 * -inf mask above the diagonal (SpeechTokenizerEncoder.swift:1039-1043). */
static void attention_f32(const float* q, const float* k, const float* v, int T, int heads, int D,
                          int causal, float* out) {
    float scale = powf((float)D, -0.5f);
#pragma omp parallel for collapse(2) schedule(static)
    for (int i = 0; i < T; ++i)
        for (int h = 0; h < heads; ++h) {
            int Tk = causal ? i + 1 : T;
            float* sc = (float*)malloc(sizeof(float) * (size_t)T);
            float mx = -INFINITY;
            for (int t = 0; t < Tk; ++t) {
                float a = 0.f;
                for (int d = 0; d < D; ++d)
                    a += q[((size_t)i * heads + h) * D + d] * k[((size_t)t * heads + h) * D + d];
                a *= scale;
                sc[t] = a;
                if (a > mx) mx = a;
            }
            float sum = 0.f;
            for (int t = 0; t < Tk; ++t) {
                sc[t] = expf(sc[t] - mx);
                sum += sc[t];
            }
            for (int d = 0; d < D; ++d) {
                float a = 0.f;
                for (int t = 0; t < Tk; ++t) a += sc[t] * v[((size_t)t * heads + h) * D + d];
                out[((size_t)i * heads + h) * D + d] = a / sum;
            }
            free(sc);
        }
}

/* DecoderTransformerAttention core (SpeechTokenizer.swift:512-528): full bidirectional attention,
 * no positional encoding, no mask. */
O_API void o_attention_full_f32(const float* q, const float* k, const float* v, int T, int heads,
                                int D, float* out) {
    attention_f32(q, k, v, T, heads, D, 0, out);
}

/* ------------------------------------------------------------------------------------------ */
/* Voice-clone front end (fp32): codec encoder + speaker encoder blocks                       */
/* ------------------------------------------------------------------------------------------ */

/* EncoderAttention core (SpeechTokenizerEncoder.swift:497-526) under the causal mask built in
 * encode() (:1039-1043); RoPE is applied by the caller. */
O_API void o_attention_causal_f32(const float* q, const float* k, const float* v, int T, int heads,
                                  int D, float* out) {
    attention_f32(q, k, v, T, heads, D, 1, out);
}

/* General 1-D convolution on channels-last data, MLX.conv1d semantics after explicit padding:
 *   StreamableConv1d (SpeechTokenizerEncoder.swift:163-186): zero pad (left, right), stride, dilation;
 *   TimeDelayNetBlock (SpeakerEncoder.swift:62-69) : reflect pad (pad_mode 1, SpeakerEncoder.swift:26-40).
 * x [T][Cin], W [Cout][K][Cin], out [Tout][Cout], Tout = (T + pl + pr - (K-1)*dil - 1)/stride + 1. */
O_API int o_conv1d_f32(const float* x, const float* W, const float* bias, int T, int Cin, int Cout, int K,
                       int stride, int dil, int pad_left, int pad_right, int pad_mode, float* out) {
    int Tp = T + pad_left + pad_right;
    int Tout = (Tp - (K - 1) * dil - 1) / stride + 1;
    if (Tout <= 0) return 0;
    if (!out) return Tout;
#pragma omp parallel for schedule(static)
    for (int t = 0; t < Tout; ++t) {
        for (int co = 0; co < Cout; ++co) {
            float acc = 0.f;
            for (int k = 0; k < K; ++k) {
                int ti = t * stride + k * dil - pad_left;
                if (ti < 0 || ti >= T) {
                    if (pad_mode == 0) continue;
                    ti = ti < 0 ? -ti : 2 * (T - 1) - ti;
                }
                const float* xr = x + (size_t)ti * Cin;
                const float* wr = W + ((size_t)co * K + k) * Cin;
                for (int ci = 0; ci < Cin; ++ci) acc += xr[ci] * wr[ci];
            }
            if (bias) acc += bias[co];
            out[(size_t)t * Cout + co] = acc;
        }
    }
    return Tout;
}

/* melFilterbank (SpeakerEncoder.swift:493-550), all arithmetic in Float as in the Swift source:
 * HTK mel scale, integer bin edges floor((nfft+1)*hz/sr), triangular slopes. out [nfft/2+1][n_mels]. */
O_API void o_mel_filterbank(int nfft, int n_mels, int sr, float fmin, float fmax, float* out) {
    int nfreq = nfft / 2 + 1;
    float mel_min = 2595.0f * log10f(1.0f + fmin / 700.0f);
    float mel_max = 2595.0f * log10f(1.0f + fmax / 700.0f);
    int* bins = (int*)malloc(sizeof(int) * (size_t)(n_mels + 2));
    for (int i = 0; i <= n_mels + 1; ++i) {
        float mel = mel_min + (float)i * (mel_max - mel_min) / (float)(n_mels + 1);
        float hz = 700.0f * (powf(10.0f, mel / 2595.0f) - 1.0f);
        bins[i] = (int)floorf((float)(nfft + 1) * hz / (float)sr);
    }
    memset(out, 0, sizeof(float) * (size_t)nfreq * n_mels);
    for (int m = 0; m < n_mels; ++m) {
        int left = bins[m], center = bins[m + 1], right = bins[m + 2];
        for (int k = left; k < center; ++k)
            if (k < nfreq && center > left) out[(size_t)k * n_mels + m] = (float)(k - left) / (float)(center - left);
        for (int k = center; k < right; ++k)
            if (k < nfreq && right > center) out[(size_t)k * n_mels + m] = (float)(right - k) / (float)(right - center);
    }
    free(bins);
}
