// row_jobs.h -- per-row jobs of the frame step that ride along in another kernel's launch instead of paying a launch of
// their own (a launch boundary costs 2.65 us in the replayed graph whatever the kernel does):
//   norm_row_job      RMSNorm of one hidden-state row (the talker's final norm in front of the code predictor): extra
//                     workgroups of the codec_head GEMM's launch -- both only read the talker's output;
//   frame_end_job     end of a frame (next talker input + loop bookkeeping): tail of the LAST code-predictor sampler, which
//                     is its only predecessor and works on the same row.
// Both are executed by the first 256 threads of a workgroup; EVERY thread of the workgroup must call them (barriers
// inside). Multiply-adds are spelled out (__fmaf_rn) so that the result does not depend on the translation unit's
// -ffp-contract setting: the same row must round identically whichever kernel carries the job.
#pragma once
#include "../common.h"
#include "../kernels.h"

namespace q3 {

// sum over threads 0..255 in the order wave sums -> ((s0 + s1) + s2) + s3; sh: 4 floats of LDS
__device__ __forceinline__ float block_sum_first256(float v, float* sh, int tid) {
    v = wave_sum(v);
    __syncthreads();
    if (tid < 256 && (tid & 63) == 0) sh[tid >> 6] = v;
    __syncthreads();
    return ((sh[0] + sh[1]) + sh[2]) + sh[3];
}

//   out <- bf16( bf16(h * rstd) * w )            MLXNN.RMSNorm, Talker.swift:520,573
__device__ __forceinline__ void norm_row_job(const NormRowsArgs& a, int m, int tid, float* sh /*[4]*/, float* parts /*[8]*/) {
    constexpr int kTrips = 2;  // H <= 4096
    const bool on = tid < 256;
    float v[kTrips][8];
    uint4 wv[kTrips];
    float ss = 0.f;
#pragma unroll
    for (int tr = 0; tr < kTrips; ++tr) {
        const int i0 = tid * 8 + tr * 2048;
        wv[tr] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[tr][j] = 0.f;
        if (on && i0 < a.H) {
            const uint4 hv = *reinterpret_cast<const uint4*>(a.h + act_tiled_offset(m, i0, a.hMB));
            wv[tr] = *reinterpret_cast<const uint4*>(a.w + i0);
            const uint32_t hw[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[tr][2 * j] = lo_bf(hw[j]);
                v[tr][2 * j + 1] = hi_bf(hw[j]);
                ss = __fmaf_rn(v[tr][2 * j], v[tr][2 * j], ss);
                ss = __fmaf_rn(v[tr][2 * j + 1], v[tr][2 * j + 1], ss);
            }
        }
    }
    float tot;
    if (a.ss_in) {  // the GEMM prologue's order: eight strided partial sums, then those eight in order
        if (tid < 8) {
            // all of a thread's partials are requested before the first add (as a loop of load-then-add this was sixteen
            // memory round trips in a row, 8 of the kernel's 10 us at 512 rows); H <= 4096: at most 32 per thread
            float t[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                const int j = tid + 8 * u;
                t[u] = a.ss_in[(size_t)(j < a.ss_count ? j : 0) * a.ss_ld + m];
            }
            float s = 0.f;
#pragma unroll
            for (int u = 0; u < 32; ++u)
                if (tid + 8 * u < a.ss_count) s += t[u];
            parts[tid] = s;
        }
        __syncthreads();
        tot = 0.f;
#pragma unroll
        for (int p = 0; p < 8; ++p) tot += parts[p];
    } else {
        tot = block_sum_first256(ss, sh, tid);
    }
    const float rstd = 1.0f / sqrtf(tot / (float)a.H + a.eps);
    float so = 0.f;
#pragma unroll
    for (int tr = 0; tr < kTrips; ++tr) {
        const int i0 = tid * 8 + tr * 2048;
        if (on && i0 < a.H) {
            const uint32_t ww[4] = {wv[tr].x, wv[tr].y, wv[tr].z, wv[tr].w};
            uint32_t ow[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float o0 = rbf(rbf(v[tr][2 * j] * rstd) * lo_bf(ww[j]));
                const float o1 = rbf(rbf(v[tr][2 * j + 1] * rstd) * hi_bf(ww[j]));
                so = __fmaf_rn(o0, o0, so);
                so = __fmaf_rn(o1, o1, so);
                ow[j] = pack_bf(o0, o1);
            }
            *reinterpret_cast<uint4*>(a.out + act_tiled_offset(m, i0, a.outMB)) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
        }
    }
    if (a.ss_out) {
        const float t2 = block_sum_first256(so, sh, tid);
        if (tid == 0) a.ss_out[m] = t2;
    }
}

// End of a frame (Qwen3.swift:914-935): next talker input = text embed (next trailing text row or tts_pad) + sum of the 16
// codebook embeddings, every add rounded to bf16 in the reference's left-to-right order; then the loop bookkeeping that the
// Swift loop keeps on the host. `last_code` >= 0 replaces cur_codes[b][groups - 1] (the caller has just decided it and its
// store may not be visible to the other threads yet).
__device__ __forceinline__ void frame_end_job(const FrameEndArgs& a, int b, int tid, int last_code, float* sh /*[4]*/) {
    const bool on = tid < 256;
    const int32_t* cc = a.cur_codes + (size_t)b * 16;
    const int ti = a.trailing_idx[b];
    const bool has_text = ti < a.n_trailing[b];
    const uint16_t* text = has_text ? a.trailing + ((size_t)b * a.Tmax + ti) * a.H : a.tts_pad;
    const uint16_t* rows[16];
    rows[0] = a.codec_emb + (size_t)cc[0] * a.H;
#pragma unroll
    for (int g = 1; g < 16; ++g) {
        const int code = (last_code >= 0 && g == a.groups - 1) ? last_code : cc[g];
        rows[g] = (g < a.groups) ? a.cp_emb[g - 1] + (size_t)code * a.H : rows[0];
    }
    float ss = 0.f;
    if (on) {
        for (int i0 = tid * 8; i0 < a.H; i0 += 256 * 8) {
            uint4 e[16];
#pragma unroll
            for (int g = 0; g < 16; ++g) e[g] = *reinterpret_cast<const uint4*>(rows[g] + i0);  // 16 independent loads
            const uint4 tx = *reinterpret_cast<const uint4*>(text + i0);
            const uint32_t tw[4] = {tx.x, tx.y, tx.z, tx.w};
            uint32_t ow[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t first = (&e[0].x)[j];
                float c0 = lo_bf(first), c1 = hi_bf(first);
#pragma unroll
                for (int g = 1; g < 16; ++g)
                    if (g < a.groups) {
                        const uint32_t w = (&e[g].x)[j];
                        c0 = rbf(c0 + lo_bf(w));
                        c1 = rbf(c1 + hi_bf(w));
                    }
                const float h0 = rbf(lo_bf(tw[j]) + c0), h1 = rbf(hi_bf(tw[j]) + c1);
                ss = __fmaf_rn(h0, h0, ss);
                ss = __fmaf_rn(h1, h1, ss);
                ow[j] = pack_bf(h0, h1);
            }
            *reinterpret_cast<uint4*>(a.h + act_tiled_offset(b, i0, a.hMB)) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
        }
    }
    const float tot = block_sum_first256(ss, sh, tid);  // (its barriers also order every thread's read of trailing_idx before the update)
    if (tid == 0) {
        a.ss_out[b] = tot;
        if (has_text) a.trailing_idx[b] = ti + 1;
        const int nf = a.n_frames[b] + 1;
        a.n_frames[b] = nf;
        if (nf >= a.max_frames[b]) {  // for _ in 0..<effectiveMaxTokens (Qwen3.swift:847)
            a.finished[b] = 1;
            a.active[b] = 0;
        }
        a.cp_len[b] = 0;  // fresh code-predictor cache per frame (Qwen3.swift:879)
    }
}

}  // namespace q3
