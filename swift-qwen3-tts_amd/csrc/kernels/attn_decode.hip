// attn_decode.hip -- one decode step of TalkerAttention / CodePredictorAttention after the fused
// QKV projection: per-head QK RMSNorm -> RoPE -> KV append -> GQA attention over the paged cache.
//
// Reference: /root/reference/Sources/Qwen3TTS/Models/Talker.swift:207-235 (norm :212-213, RoPE
// :221 with applyRotaryPosEmb :139-152, cache.update :224-226, SDPA :229-235) and
// CodePredictor.swift:111-134. KVCacheSimple (contiguous, grown by concatenation) becomes a paged
// pool: page = 64 tokens x 128 dims per kv head, block table per row.
//
// Roofline: HBM (KV read: 2 * T * 256 B per (row, kv head, layer)). One workgroup per
// (kv head, row); the `rep` query heads that share the kv head are processed together so K/V are
// read once. 16 lanes cover one 256-byte K (or V) row with 16-byte loads, so a wave-instruction
// reads 4 consecutive cache rows = 1 KiB contiguous; the 16 lane-groups of the workgroup stride
// over T and are merged with a log-sum-exp reduction through LDS.
#include <algorithm>
#include <cstdlib>

#include <type_traits>

#include "../common.h"
#include "../kernels.h"

namespace q3 {
namespace {

#include "attn_body.inc"

template <int REP, int NTH, bool NTKV = false>
__global__ __launch_bounds__(NTH) void attn_decode_kernel(AttnArgs a) {
    __builtin_amdgcn_s_setprio(3);
    attn_decode_body<REP, NTH, NTKV>(a, blockIdx.x, blockIdx.y);
}

// Several consecutive positions of every row in one launch (chunked prompt prefill; the code predictor's step 0,
// whose two positions [hidden, embed(code0)] arrive together: Qwen3.swift:884-887). Row m = p * B + b carries chunk
// element p of batch row b. Elements p < p0(b) are padding in front of a right-aligned prompt and are skipped; element p
// sits at cache position len0 + (p - p0). A query attends to the cache and to the chunk elements up to itself, which are
// kept in LDS; every lane group walks its positions (t mod NG) in increasing order exactly as the one-position kernel
// would if the elements arrived one launch at a time, so the results do not depend on where a chunk boundary falls.
template <int REP, int NTH, int CMAX>
__global__ __launch_bounds__(NTH) void attn_chunk_kernel(AttnArgs a) {
    __builtin_amdgcn_s_setprio(3);
    constexpr int NG = NTH / 16;
    constexpr int NWV = NTH / 64;
    __shared__ __attribute__((aligned(16))) float q_s[CMAX][REP][D];
    // the chunk's keys / values are bf16-exact (norm_rope rounds, V is a copy): above eight positions they are kept as bf16 so
    // that sixteen positions still fit the 64 KiB of static LDS next to the lane groups' partial results
    using KV = typename std::conditional<(CMAX > 8), uint16_t, float>::type;
    auto kv_put = [](float v) -> KV { if constexpr (CMAX > 8) return f2bf(v); else return v; };
    auto kv_get = [](KV v) -> float { if constexpr (CMAX > 8) return bf2f(v); else return v; };
    __shared__ KV k_s[CMAX][D];
    __shared__ KV v_s[CMAX][D];
    __shared__ float m_s[NG][REP];
    __shared__ float l_s[NG][REP];
    __shared__ float acc_s[NG][REP][D];
    __shared__ __attribute__((aligned(16))) uint16_t out_s[REP * D];

    const int kvh = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = a.chunk;
    const int len0 = a.fixed_len >= 0 ? a.fixed_len : a.kv_len[b];
    const int r0 = a.chunk_n_prompt ? a.chunk_r_base + a.chunk_n_prompt[b] : 0;  // prompt index of chunk element 0
    const int p0 = r0 < 0 ? (-r0 < C ? -r0 : C) : 0;
    const int qdim = a.n_heads * D, kdim = a.n_kv * D;
    const int32_t* bt = a.block_table + (size_t)b * a.max_pages;

    // The queries of a chunk are independent of one another and one workgroup walks them in turn (two barriers and a
    // reduction over all lane groups per query). gridDim.z workgroups share them, query p going to workgroup p mod gridDim.z:
    // each stages the whole chunk's keys / values for itself and only its own queries; the cache is appended by workgroup 0
    // alone. Nothing a query computes depends on the split. (Measured at sixteen positions x 32 rows, 1.7B: 30.2 us with
    // one workgroup per (row, kv head), 26.1 with two, 31.6 with four -- the part is full either way. Requesting phase 1's
    // operands up front and the cached keys / values once per chunk instead of once per query bought 2 us more and cost the
    // predictor's two-position step 0.25 us per launch: not kept.)
    const int qs = blockIdx.z, nqs = gridDim.z;
    // ---- phase 1: q/k norm + rope, v copy, cache append for every live element; vectors round-robin over the waves ----
    for (int vtx = wave; vtx < (C - p0) * (REP + 2); vtx += NWV) {
        const int p = p0 + vtx / (REP + 2), j = vtx % (REP + 2);
        if (j < REP && p % nqs != qs) continue;  // (wave-uniform)
        const int pos = len0 + (p - p0);
        const uint16_t* row = a.qkv + (size_t)(p * a.B + b) * a.ld;
        const uint16_t* cosr = a.rope_cos + (size_t)pos * D;
        const uint16_t* sinr = a.rope_sin + (size_t)pos * D;
        const int npage = a.identity_pages ? b : bt[pos / kPageTokens];
        const size_t nslot = (((size_t)npage * a.n_kv + kvh) * kPageTokens + (pos % kPageTokens)) * D;
        if (j < REP) {
            const uint16_t* qp = row + (size_t)(kvh * REP + j) * D;
            float o0, o1;
            norm_rope(bf2f(qp[lane]), bf2f(qp[lane + 64]), a.qn_w, a.eps, cosr, sinr, lane, o0, o1);
            q_s[p][j][lane] = o0;
            q_s[p][j][lane + 64] = o1;
        } else if (j == REP) {
            const uint16_t* kp = row + qdim + (size_t)kvh * D;
            float o0, o1;
            norm_rope(bf2f(kp[lane]), bf2f(kp[lane + 64]), a.kn_w, a.eps, cosr, sinr, lane, o0, o1);
            k_s[p][lane] = kv_put(o0);
            k_s[p][lane + 64] = kv_put(o1);
            if (qs == 0) {
                a.kpool[nslot + lane] = f2bf(o0);
                a.kpool[nslot + lane + 64] = f2bf(o1);
            }
        } else {
            const uint16_t* vp = row + qdim + kdim + (size_t)kvh * D;
            const uint16_t v0 = vp[lane], v1 = vp[lane + 64];
            v_s[p][lane] = kv_put(bf2f(v0));
            v_s[p][lane + 64] = kv_put(bf2f(v1));
            if (qs == 0) {
                a.vpool[nslot + lane] = v0;
                a.vpool[nslot + lane + 64] = v1;
            }
        }
    }
    __syncthreads();

    // ---- phase 2: one query after the other ----
    const int g = tid >> 4, c = tid & 15;
    for (int p = p0; p < C; ++p) {
        if (p % nqs != qs) continue;  // (uniform: the barriers below are per query)
        float q[REP][8];
#pragma unroll
        for (int h = 0; h < REP; ++h)
#pragma unroll
            for (int j = 0; j < 8; ++j) q[h][j] = q_s[p][h][8 * c + j];
        float m[REP], l[REP], acc[REP][8];
#pragma unroll
        for (int h = 0; h < REP; ++h) {
            m[h] = -INFINITY;
            l[h] = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[h][j] = 0.f;
        }
        auto step = [&](const float (&kf)[8], const float (&vf)[8]) {
#pragma unroll
            for (int h = 0; h < REP; ++h) {
                float d = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) d += q[h][j] * kf[j];
                d = row16_sum(d);
                const float sc = d * a.scale;
                const float mn = fmaxf(m[h], sc);
                const float alpha = __expf(m[h] - mn);
                const float pr = __expf(sc - mn);
                l[h] = l[h] * alpha + pr;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[h][j] = acc[h][j] * alpha + pr * vf[j];
                m[h] = mn;
            }
        };
        const int len = len0 + (p - p0);  // tokens before this query: len0 from the cache, the rest from this chunk
        for (int t = g; t <= len; t += NG) {
            float kf[8], vf[8];
            if (t < len0) {
                const int page = a.identity_pages ? b : bt[t / kPageTokens];
                const size_t off = (((size_t)page * a.n_kv + kvh) * kPageTokens + (t % kPageTokens)) * D + 8 * c;
                const uint4 kr = *reinterpret_cast<const uint4*>(a.kpool + off);
                const uint4 vr = *reinterpret_cast<const uint4*>(a.vpool + off);
                kf[0] = lo_bf(kr.x); kf[1] = hi_bf(kr.x); kf[2] = lo_bf(kr.y); kf[3] = hi_bf(kr.y);
                kf[4] = lo_bf(kr.z); kf[5] = hi_bf(kr.z); kf[6] = lo_bf(kr.w); kf[7] = hi_bf(kr.w);
                vf[0] = lo_bf(vr.x); vf[1] = hi_bf(vr.x); vf[2] = lo_bf(vr.y); vf[3] = hi_bf(vr.y);
                vf[4] = lo_bf(vr.z); vf[5] = hi_bf(vr.z); vf[6] = lo_bf(vr.w); vf[7] = hi_bf(vr.w);
            } else {
                const int e = p0 + (t - len0);  // chunk element that sits at position t
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    kf[j] = kv_get(k_s[e][8 * c + j]);
                    vf[j] = kv_get(v_s[e][8 * c + j]);
                }
            }
            step(kf, vf);
        }
#pragma unroll
        for (int h = 0; h < REP; ++h) {
            if (c == 0) {
                m_s[g][h] = m[h];
                l_s[g][h] = l[h];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc_s[g][h][8 * c + j] = acc[h][j];
        }
        __syncthreads();
        for (int o = tid; o < REP * D; o += NTH) {
            const int h = o / D, d = o % D;
            float M = -INFINITY;
#pragma unroll
            for (int gg = 0; gg < NG; ++gg) M = fmaxf(M, m_s[gg][h]);
            float num = 0.f, den = 0.f;
#pragma unroll
            for (int gg = 0; gg < NG; ++gg) {
                const float w = (m_s[gg][h] == -INFINITY) ? 0.f : __expf(m_s[gg][h] - M);
                num += acc_s[gg][h][d] * w;
                den += l_s[gg][h] * w;
            }
            out_s[o] = f2bf(num / den);
        }
        __syncthreads();
        for (int pc = tid; pc < REP * D / 8; pc += NTH) {
            const int col = (kvh * REP) * D + 8 * pc;
            *reinterpret_cast<uint4*>(a.out + act_tiled_offset(p * a.B + b, col, a.outMB)) = *reinterpret_cast<const uint4*>(out_s + 8 * pc);
        }
        __syncthreads();
    }
}

}  // namespace

int attn_decode_threads(const AttnArgs& a) {
    const int rep = a.n_heads / a.n_kv;
    return (a.max_pages > 1 && rep <= 2) ? 512 : 256;
}

void launch_attn_decode(const AttnArgs& a0, hipStream_t st) {
    AttnArgs a = a0;
    if (!a.pf.base) {  // a caller outside the frame step's plan: an empty range on a valid address
        a.pf = PfArgs{};
        a.pf.base = reinterpret_cast<const uint8_t*>(a.qkv);
        a.pf.span = 128; a.pf.lines = 1; a.pf.inv_lines = 1.0f;
    }
    a.pf.gx = uint32_t(a.n_kv);  // this launch's own geometry for its touch descriptor (prefetch.h)
    a.pf.wg_per_xcd = uint32_t((a.n_kv * a.B + 7) / 8);
    const int rep = a.n_heads / a.n_kv;
    Q3_CHECK(rep * a.n_kv == a.n_heads && rep >= 1 && rep <= kMaxRep, 3, "attn_decode: unsupported GQA ratio");
    dim3 grid(a.n_kv, a.B);
    // Short caches (the code predictor never holds more than 17 tokens; max_pages == 1): 256 threads -- phase 1's four
    // vectors on four waves, every cached position on its own lane group. One wave per workgroup (free barriers, nothing
    // waits on other waves) looked right on paper and measured 4 % slower on the whole frame step. Long caches: 512
    // threads, 32 lane groups walk the positions (1.3 % on the frame step over 256; LDS allows it up to two query heads per
    // kv head). The chunk kernel uses the same counts, so both round identically.
    const bool wide = a.max_pages > 1 && rep <= 2;
    if (a.chunk > 1) {
        Q3_CHECK(a.chunk <= 16, 3, "attn_decode: at most 16 positions per launch");
        // two workgroups share the queries of a long chunk (prefill), one takes a short one (the predictor's step 0)
        const int qenv = debug_env().chunk_qsplit;
        const int nqs = qenv > 0 ? std::max(1, std::min(a.chunk, qenv)) : (a.chunk > 8 ? 2 : 1);
        const dim3 cgrid(a.n_kv, a.B, nqs);
#define Q3_CHUNK(REPv, NTHv)                                                                                        \
    do {                                                                                                            \
        if (a.chunk > 8) hipLaunchKernelGGL((attn_chunk_kernel<REPv, NTHv, 16>), cgrid, dim3(NTHv), 0, st, a);       \
        else hipLaunchKernelGGL((attn_chunk_kernel<REPv, NTHv, 8>), cgrid, dim3(NTHv), 0, st, a);                    \
    } while (0)
        switch (rep) {
            case 1: if (wide) Q3_CHUNK(1, 512); else Q3_CHUNK(1, 256); break;
            case 2: if (wide) Q3_CHUNK(2, 512); else Q3_CHUNK(2, 256); break;
            case 3: Q3_CHUNK(3, 256); break;
            case 4: Q3_CHUNK(4, 256); break;
        }
#undef Q3_CHUNK
        return;
    }
    switch (rep) {
        case 1:
            if (wide) hipLaunchKernelGGL((attn_decode_kernel<1, 512>), grid, dim3(512), 0, st, a);
            else hipLaunchKernelGGL((attn_decode_kernel<1, 256>), grid, dim3(256), 0, st, a);
            break;
        case 2:
            if (wide && a.nt_kv) hipLaunchKernelGGL((attn_decode_kernel<2, 512, true>), grid, dim3(512), 0, st, a);
            else if (wide) hipLaunchKernelGGL((attn_decode_kernel<2, 512>), grid, dim3(512), 0, st, a);
            else hipLaunchKernelGGL((attn_decode_kernel<2, 256>), grid, dim3(256), 0, st, a);
            break;
        case 3: hipLaunchKernelGGL((attn_decode_kernel<3, 256>), grid, dim3(256), 0, st, a); break;
        case 4: hipLaunchKernelGGL((attn_decode_kernel<4, 256>), grid, dim3(256), 0, st, a); break;
    }
}

}  // namespace q3
