// sampler.hip -- fused per-row token sampler (one workgroup per row; vocab <= 4096 fits LDS).
//
// Restates sampleToken / applyTopK / applyTopP / categoricalSampling
// (/root/reference/Sources/Qwen3TTS/Models/Qwen3.swift:130-213, 68-89, 92-117, 120-126) as ONE
// kernel instead of ~16 MLX op launches plus eval()/.item() host syncs per codebook:
//   suppress -> repetition penalty -> [T<=0: argmax] -> save EOS -> top-k -> top-p -> restore EOS
//   -> categorical(logits * 1/T) = argmax(scaled + Gumbel noise).
// Array arithmetic is in bf16 like the reference (Float scalars become bf16 arrays in mlx-swift).
// Integer / compare work is bit-exact against the oracle; the two transcendental helpers
// (q3_logf, q3_expf) are written with explicit fma so that host and device agree bit for bit.
// This translation unit is compiled with -ffp-contract=off.
//
// Also folded in (so the frame step needs no extra launches): teacher forcing for tests, the
// code/flag bookkeeping of the Swift loop (Qwen3.swift:864-871), the embedding gather that feeds
// the next code-predictor pass (:884-892) and the KV-length advance.
#include "../common.h"
#include "../kernels.h"
#include "row_jobs.h"

namespace q3 {
namespace {

constexpr int kMaxVAll = 4096;

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// natural log, Cephes logf coefficients, +,*,fma and bit operations only
__device__ __forceinline__ float q3_logf(float x) {
    uint32_t u = __float_as_uint(x);
    int e = (int)((u >> 23) & 0xff) - 126;
    u = (u & 0x007fffffu) | 0x3f000000u;
    float m = __uint_as_float(u);
    if (m < 0.70710678118654752440f) {
        e -= 1;
        m = m + m;
    }
    const float f = m - 1.0f;
    const float z = f * f;
    float y = 7.0376836292E-2f;
    y = __fmaf_rn(y, f, -1.1514610310E-1f);
    y = __fmaf_rn(y, f, 1.1676998740E-1f);
    y = __fmaf_rn(y, f, -1.2420140846E-1f);
    y = __fmaf_rn(y, f, 1.4249322787E-1f);
    y = __fmaf_rn(y, f, -1.6668057665E-1f);
    y = __fmaf_rn(y, f, 2.0000714765E-1f);
    y = __fmaf_rn(y, f, -2.4999993993E-1f);
    y = __fmaf_rn(y, f, 3.3333331174E-1f);
    y = y * f;
    y = y * z;
    const float fe = (float)e;
    y = __fmaf_rn(fe, -2.12194440e-4f, y);
    y = __fmaf_rn(z, -0.5f, y);
    float r = f + y;
    r = __fmaf_rn(fe, 0.693359375f, r);
    return r;
}

// exp, Cephes expf coefficients, same construction rules
__device__ __forceinline__ float q3_expf(float x) {
    if (x < -87.0f) return 0.0f;
    if (x > 88.0f) return INFINITY;
    const float fx = floorf(__fmaf_rn(x, 1.44269504088896341f, 0.5f));
    float r = __fmaf_rn(fx, -0.693359375f, x);
    r = __fmaf_rn(fx, 2.12194440e-4f, r);
    const float z = r * r;
    float y = 1.9875691500E-4f;
    y = __fmaf_rn(y, r, 1.3981999507E-3f);
    y = __fmaf_rn(y, r, 8.3334519073E-3f);
    y = __fmaf_rn(y, r, 4.1665795894E-2f);
    y = __fmaf_rn(y, r, 1.6666665459E-1f);
    y = __fmaf_rn(y, r, 5.0000001201E-1f);
    y = __fmaf_rn(y, z, r);
    y = y + 1.0f;
    const int n = (int)fx;
    return y * __uint_as_float((uint32_t)(n + 127) << 23);
}

__device__ __forceinline__ float gumbel_noise(uint64_t seed, uint32_t row, uint32_t draw, uint32_t i) {
    uint32_t r[4];
    philox4x32_10(i, row, draw, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const float u = ((float)(r[0] >> 8) + 0.5f) * 5.9604644775390625e-08f;
    return -q3_logf(-q3_logf(u));
}

// monotone 16-bit key of a bf16-exact float (ascending)
__device__ __forceinline__ uint32_t key16(float v) {
    const uint32_t u = __float_as_uint(v) >> 16;
    return (u & 0x8000u) ? (~u & 0xffffu) : (u | 0x8000u);
}

struct Best {
    float v;
    int i;
};
__device__ __forceinline__ Best better(Best a, Best b) {  // larger value, then lower index
    return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}

// lane i ^ o's copy of (v, i) for the butterfly below, on DPP / permlane swaps (common.h XorPartner) instead of ds_bpermute
template <int O>
__device__ __forceinline__ Best partner(Best x) {
    Best y;
    const float fi = __int_as_float(x.i);
    if constexpr (O == 32 || O == 16) {
        float a, b, c, d;
        if constexpr (O == 32) { XorPartner::swap32(x.v, a, b); XorPartner::swap32(fi, c, d); }
        else { XorPartner::swap16(x.v, a, b); XorPartner::swap16(fi, c, d); }
        // a = {lo, lo}, b = {hi, hi}: the partner's value is whichever of the two is not this lane's own copy
        const bool upper = O == 32 ? ((threadIdx.x & 32) != 0) : ((threadIdx.x & 16) != 0);
        y.v = upper ? a : b;
        y.i = __float_as_int(upper ? c : d);
    } else if constexpr (O == 8) {
        y.v = XorPartner::x8(x.v); y.i = __float_as_int(XorPartner::x8(fi));
    } else if constexpr (O == 4) {
        y.v = XorPartner::x4(x.v); y.i = __float_as_int(XorPartner::x4(fi));
    } else if constexpr (O == 2) {
        y.v = XorPartner::x2(x.v); y.i = __float_as_int(XorPartner::x2(fi));
    } else {
        y.v = XorPartner::x1(x.v); y.i = __float_as_int(XorPartner::x1(fi));
    }
    return y;
}

template <int kThreads>
__device__ Best block_argmax(Best x, Best* scratch) {
    x = better(x, partner<32>(x));
    x = better(x, partner<16>(x));
    x = better(x, partner<8>(x));
    x = better(x, partner<4>(x));
    x = better(x, partner<2>(x));
    x = better(x, partner<1>(x));
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[wave] = x;
    __syncthreads();
    Best r = scratch[0];
    for (int w = 1; w < kThreads / 64; ++w) r = better(r, scratch[w]);
    return r;
}

// kMaxV >= V elements over kThreads threads. <1024, 4096, *>: any vocabulary up to 4096 (the talker's 3072). <1024, 2048,
// false>: the code predictor's fifteen draws per frame (V = 2048, no suppress range, no repetition penalty, no EOS:
// Qwen3.swift:904-909) -- two elements per thread instead of four and the talker-only branches compiled out: frame step
// 3.413 -> 3.375 ms (512 threads x 4 elements: 3.380; 256 x 8: 3.426). TALKER must equal a.is_talker.
template <int kThreads, int kMaxV, bool TALKER, bool FEND>
__device__ __forceinline__ void sampler_body(const SamplerArgs& a, const FrameEndArgs* fe) {
    constexpr int kElems = kMaxV / kThreads;
    __builtin_amdgcn_s_setprio(3);
    __shared__ float vals[kMaxV];
    __shared__ uint32_t sortbuf[kMaxV];
    __shared__ float sval[kMaxV];   // surviving logits, compacted (indices in sortbuf)
    __shared__ int n_surv;
    __shared__ uint32_t hist[2][256];
    __shared__ Best scratch[kThreads / 64];
    __shared__ uint32_t wcount[kElems][kThreads / 64];
    __shared__ int sel[2][2];

    const int b = blockIdx.x, tid = threadIdx.x;
    // Every operand is requested before the first branch: the row's logits and repetition flags do not depend on the
    // flags that decide whether the row is still live, and a finished row wastes a few loads instead of every live row
    // paying a second memory round trip.
    const int V = a.V;
    const uint16_t* lrow = a.logits + (size_t)b * a.ldl;
    uint16_t raw[kElems];
    uint8_t was_seen[kElems];
    const bool want_seen = TALKER && a.seen;
#pragma unroll
    for (int k = 0; k < kElems; ++k) {
        const int i = k * kThreads + tid;
        // (addresses clamped, not predicated: behind a predicated load hipcc waits -- s_waitcnt vmcnt(0) -- before it issues
        // the next one, and the row's two or four logits became as many memory round trips in a row)
        const int ic = i < V ? i : V - 1;
        raw[k] = lrow[ic];
        was_seen[k] = want_seen ? a.seen[(size_t)b * V + ic] : (uint8_t)0;
    }
    const SamplingParams sp = *a.sp;
    const bool row_done = a.finished[b] != 0;
    const int frame = a.n_frames[b];
    // (read from an address that is always valid, with the other operands: under a branch on the pointer this byte was a
    // memory round trip of its own behind them)
    const uint8_t gate_raw = *(a.advance_gate ? a.advance_gate + b : a.finished + b);
    const bool gate = a.advance_gate ? (gate_raw != 0) : true;
    if (tid < 256) {  // both radix levels' histograms, zeroed under the loads
        hist[0][tid] = 0;
        hist[1][tid] = 0;
    }
    if (tid == 0) n_surv = 0;
    if (row_done) {
        if (tid == 0 && a.advance && gate && !a.advance_gate) a.kv_len[b] += 1;  // predictor rows keep in step
        return;
    }
    if (a.logits_dump && frame < a.forced_frames)
        for (int i = tid; i < V; i += kThreads)
            a.logits_dump[((size_t)b * a.forced_frames + frame) * a.dump_ld + a.dump_off + i] = lrow[i];

    // ---- 1+2: suppress, repetition penalty ----
    // the LDS copy of the row is read by the EOS save / restore (talker) and by top-p only: the predictor's draws skip it
    const bool use_top_p = sp.top_p > 0.f && sp.top_p < 1.0f;
    const bool need_vals = TALKER || use_top_p;
    const float pen = rbf(sp.rep_penalty);
    const bool use_pen = want_seen && sp.rep_penalty != 1.0f;
    float l[kElems];
#pragma unroll
    for (int k = 0; k < kElems; ++k) {
        const int i = k * kThreads + tid;
        float v = -INFINITY;
        if (i < V) {
            v = bf2f(raw[k]);
            if (TALKER) {
                if (i >= a.suppress_lo && i < a.suppress_hi && i != a.eos_id) v = -INFINITY;
                if (sp.mask_eos && i == a.eos_id) v = -INFINITY;
                if (use_pen && was_seen[k]) v = (v < 0.f) ? rbf(v * pen) : rbf(v / pen);
            }
        }
        l[k] = v;
        if (need_vals) vals[k * kThreads + tid] = v;
    }
    __syncthreads();  // also orders the zeroed histograms / n_surv before the atomics below

    int tok;
    if (sp.temperature <= 0.f) {
        // ---- 3: greedy = first maximum ----
        Best x{-INFINITY, 0x7fffffff};
#pragma unroll
        for (int k = 0; k < kElems; ++k) {
            const int i = k * kThreads + tid;
            if (i < V) x = better(x, Best{l[k], i});
        }
        tok = block_argmax<kThreads>(x, scratch).i;
        if (tok == 0x7fffffff) tok = 0;  // every logit NaN (a damaged checkpoint): no maximum exists; the token is a table row next step
    } else {
        const bool have_eos = TALKER && a.eos_id >= 0 && a.eos_id < V;
        const float eos_logit = have_eos ? vals[a.eos_id] : 0.f;
        // ---- 5: top-k, two-level radix select on the 16-bit key; ties by lower index ----
        if (sp.top_k > 0 && sp.top_k < V) {
            uint32_t key[kElems];
#pragma unroll
            for (int k = 0; k < kElems; ++k) key[k] = key16(l[k]);
            int need = sp.top_k;
            uint32_t prefix = 0;  // selected high byte
            for (int level = 0; level < 2; ++level) {  // hist[level] was zeroed at the top; two barriers per level
#pragma unroll
                for (int k = 0; k < kElems; ++k) {
                    const int i = k * kThreads + tid;
                    if (i < V && (level == 0 || (key[k] >> 8) == prefix))
                        atomicAdd(&hist[level][level == 0 ? (key[k] >> 8) : (key[k] & 0xff)], 1u);
                }
                __syncthreads();
                if (tid < 64) {  // one wave: lane l owns bins 4l..4l+3; inclusive suffix sums by shuffles
                    const uint32_t* hl = hist[level];
                    const uint32_t h0 = hl[4 * tid], h1 = hl[4 * tid + 1], h2 = hl[4 * tid + 2], h3 = hl[4 * tid + 3];
                    const int own = (int)(h0 + h1 + h2 + h3);
                    int suf = own;  // sum over lanes >= tid
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const int v = __shfl_down(suf, o, 64);
                        if (tid + o < 64) suf += v;
                    }
                    const int above = suf - own;  // elements in bins owned by higher lanes
                    // the crossing bin is the highest bin whose cumulative count from the top reaches `need`
                    const bool mine = above < need && suf >= need;
                    const bool none = __ballot(mine) == 0ull;  // fewer than `need` elements in total: take bin 0
                    if (mine) {
                        int cum = above, bin = 4 * tid + 3;
                        const int hh[4] = {(int)h0, (int)h1, (int)h2, (int)h3};
                        for (; bin > 4 * tid; --bin) {
                            if (cum + hh[bin & 3] >= need) break;
                            cum += hh[bin & 3];
                        }
                        sel[level][0] = bin;
                        sel[level][1] = need - cum;
                    }
                    if (none && tid == 0) {
                        sel[level][0] = 0;
                        sel[level][1] = need - (suf - (int)h0);
                    }
                }
                __syncthreads();
                if (level == 0) prefix = (uint32_t)sel[0][0];
                else prefix = (prefix << 8) | (uint32_t)sel[1][0];
                need = sel[level][1];
            }
            const uint32_t kt = prefix;  // threshold key; `need` equal-key elements survive
            // rank of every threshold-key element in index order: per-wave counts of all kElems slices, one barrier
            unsigned long long bal[kElems];
            {
                const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
                for (int k = 0; k < kElems; ++k) {
                    const int i = k * kThreads + tid;
                    bal[k] = __ballot((i < V) && key[k] == kt);
                    if (lane == 0) wcount[k][wave] = (uint32_t)__popcll(bal[k]);
                }
                __syncthreads();
                int base = 0;
#pragma unroll
                for (int k = 0; k < kElems; ++k) {
                    const int i = k * kThreads + tid;
                    const bool eq = (i < V) && key[k] == kt;
                    const int below = __popcll(bal[k] & ((1ull << lane) - 1ull));
                    int woff = 0, total = 0;
                    for (int w = 0; w < kThreads / 64; ++w) {
                        if (w < wave) woff += (int)wcount[k][w];
                        total += (int)wcount[k][w];
                    }
                    const int rank = base + woff + below;
                    if (i < V && (key[k] < kt || (eq && rank >= need))) l[k] = -INFINITY;
                    base += total;
                }
            }
            if (use_top_p) {
#pragma unroll
                for (int k = 0; k < kElems; ++k) vals[k * kThreads + tid] = l[k];
                __syncthreads();
            }
        }
        // ---- 6: top-p (rare path): ascending sort, sequential cumulative sum like the oracle ----
        if (use_top_p) {
#pragma unroll
            for (int k = 0; k < kElems; ++k) {
                const int i = k * kThreads + tid;
                sortbuf[i] = (i < V) ? ((key16(l[k]) << 16) | (uint32_t)i) : 0xffffffffu;
            }
            __syncthreads();
            for (int size = 2; size <= kMaxV; size <<= 1) {
                for (int stride = size >> 1; stride > 0; stride >>= 1) {
                    for (int t = tid; t < kMaxV / 2; t += kThreads) {
                        const int lo = 2 * t - (t & (stride - 1));
                        const int hi = lo + stride;
                        const bool up = ((lo & size) == 0);
                        const uint32_t x = sortbuf[lo], y = sortbuf[hi];
                        if ((x > y) == up) {
                            sortbuf[lo] = y;
                            sortbuf[hi] = x;
                        }
                    }
                    __syncthreads();
                }
            }
            if (tid == 0) {
                const float thr = rbf(1.0f - sp.top_p);
                float run = 0.f;
                for (int r = 0; r < V; ++r) {
                    const int i = (int)(sortbuf[r] & 0xffffu);
                    run += rbf(q3_expf(vals[i]));
                    if (!(rbf(run) > thr)) vals[i] = -INFINITY;
                }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < kElems; ++k) l[k] = vals[k * kThreads + tid];
        }
        // ---- 7: restore EOS ----
        if (have_eos && !sp.mask_eos) {
#pragma unroll
            for (int k = 0; k < kElems; ++k)
                if (k * kThreads + tid == a.eos_id) l[k] = eos_logit;
        }
        // ---- 8: categorical(logits * (1/T)) ----
        const float invt = rbf(1.0f / sp.temperature);
        const uint32_t draw = (uint32_t)frame * 16u + (uint32_t)a.cb;
        // The survivors (top_k of them, scattered over the 16 waves) are compacted first: Philox + two logarithms per element
        // is ~250 instructions, and a wave executes them once per slice in which ANY of its lanes holds a survivor --
        // nearly every (wave, slice) pair for 50 survivors. Compacted, one wave does it once. The noise is keyed by the
        // element index, and the arg-max breaks ties by lower index, so the order of evaluation does not matter.
#pragma unroll
        for (int k = 0; k < kElems; ++k) {
            const int i = k * kThreads + tid;
            if (i < V && l[k] != -INFINITY) {
                const int slot = atomicAdd(&n_surv, 1);
                sortbuf[slot] = (uint32_t)i;
                sval[slot] = l[k];
            }
        }
        __syncthreads();
        const int ns = n_surv;
        Best x{-INFINITY, 0x7fffffff};
        for (int q = tid; q < ns; q += kThreads) {
            const int i = (int)sortbuf[q];
            const float v = rbf(sval[q] * invt) + gumbel_noise(sp.seed, sp.row0 + (uint32_t)b, draw, (uint32_t)i);
            x = better(x, Best{v, i});
        }
        Best r = block_argmax<kThreads>(x, scratch);
        tok = (r.i == 0x7fffffff) ? 0 : r.i;
    }

    // ---- bookkeeping ----
    int used = tok;
    if (a.forced && frame < a.forced_frames) used = a.forced[((size_t)b * a.forced_frames + frame) * 16 + a.cb];
    if (tid == 0) {
        if (a.sampled && frame < a.forced_frames) a.sampled[((size_t)b * a.forced_frames + frame) * 16 + a.cb] = tok;
        a.cur_codes[(size_t)b * 16 + a.cb] = used;
        if (TALKER) {
            if (a.seen && used >= 0 && used < V) a.seen[(size_t)b * V + used] = 1;  // generatedTokens (:865)
            if (a.advance && a.active[b]) a.kv_len[b] += 1;
            if (used == a.eos_id) {  // :868-870: EOS ends the row before any code is stored
                a.finished[b] = 1;
                a.active[b] = 0;
            } else if (frame < a.Fmax) {
                a.codes[((size_t)b * a.Fmax + frame) * 16] = used;
            }
        } else {
            if (a.advance && gate) a.kv_len[b] += 1;
            if (frame < a.Fmax) a.codes[((size_t)b * a.Fmax + frame) * 16 + a.cb] = used;
        }
    }
    if (a.next_x) {
        const uint4* src = reinterpret_cast<const uint4*>(a.emb + (size_t)used * a.emb_ld);
        float ss = 0.f;
        for (int i = tid; i < a.H / 8; i += kThreads) {
            const uint4 v = src[i];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) ss += lo_bf(w[j]) * lo_bf(w[j]) + hi_bf(w[j]) * hi_bf(w[j]);
            *reinterpret_cast<uint4*>(a.next_x + act_tiled_offset(b + a.next_row0, 8 * i, a.next_MB)) = v;
        }
        if (a.emb_ss) {  // projected table: the per-tile partials the projection GEMM's epilogue would have written
            for (int j = tid; j < a.nss; j += kThreads)
                a.next_ss[(size_t)j * a.next_ss_ld + b + a.next_row0] = a.emb_ss[(size_t)used * a.nss + j];
        } else if (a.next_ss) {  // sum of squares of the row: first (and only) partial of the consumer's norm prologue
            ss = wave_sum(ss);
            __syncthreads();
            if ((tid & 63) == 0) vals[tid >> 6] = ss;
            __syncthreads();
            if (tid == 0) {
                float t = 0.f;
                for (int w = 0; w < kThreads / 64; ++w) t += vals[w];
                a.next_ss[b] = t;
            }
        }
    }
    // the frame's last draw: its row's end-of-frame job rides along (next talker input from all sixteen codes + loop state)
    if constexpr (FEND) {
        __shared__ float fe_sh[4];
        frame_end_job(*fe, b, tid, used, fe_sh);
    }
}

template <int kThreads, int kMaxV, bool TALKER>
__global__ __launch_bounds__(kThreads) void sampler_kernel(SamplerArgs a) {
    sampler_body<kThreads, kMaxV, TALKER, false>(a, nullptr);
}
struct SamplerFendArgs {
    SamplerArgs s;
    FrameEndArgs fe;
};
__global__ __launch_bounds__(1024) void sampler_fend_kernel(SamplerFendArgs a) {
    sampler_body<1024, 2048, false, true>(a.s, &a.fe);
}

}  // namespace

void launch_sampler_with_frame_end(const SamplerArgs& a, const FrameEndArgs& fe, hipStream_t st) {
    Q3_CHECK(!a.is_talker && a.V <= 2048, 3, "sampler: the end-of-frame rider belongs to a code-predictor draw");
    SamplerFendArgs s{a, fe};
    hipLaunchKernelGGL(sampler_fend_kernel, dim3(a.B), dim3(1024), 0, st, s);
}

void launch_sampler(const SamplerArgs& a, hipStream_t st) {
    Q3_CHECK(a.V <= kMaxVAll, 3, "sampler: vocabulary larger than 4096 is not supported");
    if (!a.is_talker && a.V <= 2048) hipLaunchKernelGGL((sampler_kernel<1024, 2048, false>), dim3(a.B), dim3(1024), 0, st, a);
    else if (a.is_talker) hipLaunchKernelGGL((sampler_kernel<1024, 4096, true>), dim3(a.B), dim3(1024), 0, st, a);
    else hipLaunchKernelGGL((sampler_kernel<1024, 4096, false>), dim3(a.B), dim3(1024), 0, st, a);
}

}  // namespace q3
