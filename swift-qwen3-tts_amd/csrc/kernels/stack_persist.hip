// stack_persist.hip -- one decoder-stack forward (all layers + head) of the AR loop as ONE persistent kernel.
//
// Same arithmetic as the per-op kernels (gemm_decode.hip, attn_decode.hip; reference
// /root/reference/Sources/Qwen3TTS/Models/Talker.swift:451-469, 207-235, CodePredictor.swift:111-134, 320-339),
// different schedule: the frame step of the launch-per-op path is a chain of ~650 dependent kernels of 5-8 us each
// (DESIGN.md section 5); here the chain links are phases of a resident grid separated by a device-wide flag barrier
// (~3 us on 256 workgroups, tools/gridbar.hip), and the weights of the next phase are requested from HBM before
// the barrier so that their latency hides behind it.
//
// Coherence on the 8-XCD part (L2s are not coherent with each other): every buffer that one workgroup writes and
// another reads inside the launch (h, qkv, ao, act, the per-tile sums of squares, the barrier flags) lives in
// uncached device memory (hipDeviceMallocUncached) and is read with agent-scope loads, which miss the CU's L1;
// writers drain their stores (s_waitcnt) before they publish their flag. No L2 writeback/invalidate is issued
// (measured: 14-30 us per barrier when every wave does one). Weights, norm vectors, RoPE tables and the KV pool
// are ordinary cached memory: nothing in the launch writes data that another workgroup reads from them.
//
// Liveness: the grid is sized to be co-resident (<= one workgroup per CU, 512 threads), a workgroup only ever waits
// for flags of workgroups of the same launch, and every wait is bounded by a wall-clock timeout that raises an
// error word and makes the workgroup leave; a scheduling surprise costs 50 ms, not the GPU.
#include "../common.h"
#include "../kernels.h"

namespace q3 {
namespace {

constexpr int D = kHeadDim;
constexpr int NW = 8;  // waves per workgroup
constexpr unsigned long long kTimeoutTicks = 5000000ull;  // 50 ms of the 100 MHz wall clock

// threadIdx.x behind an opaque asm: lane-derived offsets are then recomputed inside each phase instead of being
// hoisted to the top of the kernel and kept (or spilled) for its whole lifetime.
__device__ __forceinline__ int phase_tid() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

// ---- coherent (agent-scope) loads of data written by other workgroups of this launch -----------------------------
__device__ __forceinline__ uint4 ld_coh16(const void* p) {
    const uint64_t* q = reinterpret_cast<const uint64_t*>(p);
    const uint64_t a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
}
__device__ __forceinline__ uint32_t ld_coh32(const void* p) {
    return __hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_coh_f32(const float* p) { return __uint_as_float(ld_coh32(p)); }
// element i of a row-major bf16 row (4-byte aligned row start)
__device__ __forceinline__ uint16_t ld_coh_bf(const uint16_t* row, int i) {
    const uint32_t w = ld_coh32(row + (i & ~1));
    return (uint16_t)((i & 1) ? (w >> 16) : (w & 0xffffu));
}

// ---- device-wide barrier ------------------------------------------------------------------------------------------
struct GridSync {
    unsigned* flags;  // [G], uncached
    int* err;         // [0]: timeouts
    unsigned target;  // value of the last barrier
    int G;
};
// arrive: every store of this workgroup is acknowledged, then its flag moves on
__device__ __forceinline__ void barrier_arrive(GridSync& s) {
    ++s.target;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&s.flags[blockIdx.x], s.target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wait: wave 0 polls every flag (all polls of a round in flight together); false = timeout, the caller leaves
__device__ __forceinline__ bool barrier_wait(GridSync& s, int* ok_s) {
    if (threadIdx.x < 64) {
        const unsigned long long t0 = wall_clock64();
        bool ok = true;
        for (;;) {
            unsigned v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = threadIdx.x + 64 * j;
                v[j] = i < s.G ? __hip_atomic_load(&s.flags[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : s.target;
            }
            bool all = true;
#pragma unroll
            for (int j = 0; j < 8; ++j) all = all && ((int)(v[j] - s.target) >= 0);
            if (__all(all)) break;
            if (wall_clock64() - t0 > kTimeoutTicks) {
                if (threadIdx.x == 0) atomicAdd(s.err, 1);
                ok = false;
                break;
            }
        }
        if (threadIdx.x == 0) *ok_s = ok ? 1 : 0;
    }
    __syncthreads();
    return *ok_s != 0;
}

// ---- GEMM pieces (gemm_decode.hip arithmetic, tile index and weight registers passed in) --------------------------
__device__ __forceinline__ f32x4 mfma16(const uint4& a, const uint4& b, f32x4 c) {
    bf16x8 av, bv;
    __builtin_memcpy(&av, &a, 16);
    __builtin_memcpy(&bv, &b, 16);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c, 0, 0, 0);
}
__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }
__device__ __forceinline__ uint4 norm8(const uint4& hx, const uint4& wx, float rstd) {
    const uint32_t hw[4] = {hx.x, hx.y, hx.z, hx.w}, ww[4] = {wx.x, wx.y, wx.z, wx.w};
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float n0 = rbf(lo_bf(hw[j]) * rstd), n1 = rbf(hi_bf(hw[j]) * rstd);
        o[j] = pack_bf(n0 * lo_bf(ww[j]), n1 * hi_bf(ww[j]));
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}

struct Smem {  // one workgroup's LDS, reused by every phase
    union {
        struct {
            float red[NW][2][4][4][64];                               // [wave][tile of the pair][mb][q][lane]
            __attribute__((aligned(16))) uint16_t ys[4][16][16];
            float rstd_s[64];
            float ssp_s[8][64];
        } g;
        struct {
            __attribute__((aligned(16))) float q_s[kMaxRepPersist][D];
            float k_s[D];
            float v_s[D];
            float m_s[32][kMaxRepPersist];
            float l_s[32][kMaxRepPersist];
            float acc_s[32][kMaxRepPersist][D];
        } a;
    };
    int ok;
};

struct GemmDesc {       // one GEMM phase
    const uint16_t* W;  // tiled bf16 weights
    const uint16_t* x;  // fragment-major activations (uncached)
    int xMB, M, N, K;
    uint16_t* y;
    int ldy, y_tiled, yMB;
    const uint16_t* bias;
    const uint16_t* norm_w;
    const float* ss_in;
    int ss_count, ss_ld, norm_dim;
    float norm_eps;
    int resid;
    float* ss_out;
};

template <int NT, int CH>
__device__ __forceinline__ void gemm_load_w(const GemmDesc& g, int tile, uint4 (&wf)[CH][NT][4]) {
    const int tid = phase_tid();
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KC = g.K >> 7;
    const uint4* Wt = reinterpret_cast<const uint4*>(g.W);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int kl = wave + c * NW;
        const int kc = kl < KC ? kl : 0;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const uint4* wp = Wt + ((size_t)(tile * NT + t) * KC + kc) * 256 + lane;
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[c][t][i] = wp[i * 64];
        }
    }
}

// EPI 0: bf16 store (+bias), row-major or fragment-major; 2: gate/up pair -> SwiGLU; 3: hidden-state store (+residual) + sum(h^2)
template <int MB, int EPI, int CH, bool NORM>
__device__ __forceinline__ void gemm_tile(const GemmDesc& a, int tile, const uint4 (&wf)[CH][(EPI == 2) ? 2 : 1][4], Smem& sm) {
    constexpr int NT = (EPI == 2) ? 2 : 1;
    const int tid = phase_tid();
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KC = a.K >> 7;
    f32x4 acc[NT][MB];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[t][mb] = f32x4{0.f, 0.f, 0.f, 0.f};

    float rstd[MB];
    if constexpr (NORM) {  // rstd per row from the producer's per-tile sums of squares, 8 strided partials in tile order
        const int rows = 16 * MB;
        for (int idx = tid; idx < rows * 8; idx += NW * 64) {
            const int row = idx % rows, part = idx / rows;
            float tmp[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {  // all partials of this (row, part) requested before the first add
                const int j = part + 8 * u;
                const float* p = a.ss_in + (size_t)(j < a.ss_count ? j : 0) * a.ss_ld + row;
                tmp[u] = ld_coh_f32(p);
            }
            float s = 0.f;
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (part + 8 * u < a.ss_count) s += tmp[u];
            sm.g.ssp_s[part][row] = s;
        }
        __syncthreads();
        if (tid < rows) {
            float s = 0.f;
#pragma unroll
            for (int p = 0; p < 8; ++p) s += sm.g.ssp_s[p][tid];
            sm.g.rstd_s[tid] = 1.0f / sqrtf(s / (float)a.norm_dim + a.norm_eps);
        }
        __syncthreads();
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) rstd[mb] = sm.g.rstd_s[16 * mb + (lane & 15)];
    }

    const uint4* Xt = reinterpret_cast<const uint4*>(a.x);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int kc = wave + c * NW;
        if (kc < KC) {  // wave-uniform
            uint4 nw[4];
            if constexpr (NORM) {
                const uint4* np = reinterpret_cast<const uint4*>(a.norm_w + kc * 128 + 32 * (lane >> 4));
#pragma unroll
                for (int i = 0; i < 4; ++i) nw[i] = np[i];
            }
            uint4 xf[MB][4];  // every x fragment of this chunk in flight together
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const uint4* xp = Xt + ((size_t)(kc * a.xMB + mb) * 4) * 64 + lane;
#pragma unroll
                for (int i = 0; i < 4; ++i) xf[mb][i] = ld_coh16(xp + i * 64);
            }
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                if constexpr (NORM) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) xf[mb][i] = norm8(xf[mb][i], nw[i], rstd[mb]);
                }
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[t][mb] = mfma16(wf[c][t][i], xf[mb][i], acc[t][mb]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // one chunk's fragments at a time: the weight registers leave little room
    }

    // cross-wave K reduction through LDS, fixed wave order
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int q = 0; q < 4; ++q) sm.g.red[wave][t][mb][q][lane] = acc[t][mb][q];
    __syncthreads();
    for (int o = tid; o < 256 * MB; o += NW * 64) {
        const int mb = o >> 8, rem = o & 255;
        const int b = rem >> 4, f = rem & 15;
        const int src_lane = (f >> 2) * 16 + b, q = f & 3;
        float v[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) sum += sm.g.red[w][t][mb][q][src_lane];
            v[t] = sum;
        }
        const int n = tile * 16 + f;
        if constexpr (EPI == 2) {
            const float gg = rbf(v[0]), u = rbf(v[1]);
            sm.g.ys[mb][b][f] = f2bf(rbf(silu_f(gg)) * u);
        } else {
            float y = v[0];
            if (a.bias) y += bf2f(a.bias[n]);
            sm.g.ys[mb][b][f] = f2bf(y);
        }
    }
    __syncthreads();
    for (int o = tid; o < 32 * MB; o += NW * 64) {
        const int mb = o >> 5, b = (o >> 1) & 15, p = o & 1;
        const int m = 16 * mb + b;
        uint4 v = *reinterpret_cast<const uint4*>(&sm.g.ys[mb][b][8 * p]);
        const int n = tile * 16 + 8 * p;
        if constexpr (EPI == 3) {
            uint16_t* hp = a.y + act_tiled_offset(m, n, a.yMB);
            float ss = 0.f;
            const uint32_t yw[4] = {v.x, v.y, v.z, v.w};
            uint32_t ow[4];
            uint4 hv = make_uint4(0, 0, 0, 0);
            if (a.resid) hv = ld_coh16(hp);
            const uint32_t hw[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float h0 = lo_bf(yw[j]), h1 = hi_bf(yw[j]);
                if (a.resid) {
                    h0 = rbf(lo_bf(hw[j]) + h0);
                    h1 = rbf(hi_bf(hw[j]) + h1);
                }
                ss += h0 * h0;
                ss += h1 * h1;
                ow[j] = pack_bf(h0, h1);
            }
            const float other = __shfl_xor(ss, 1, 64);
            if (m < a.M) {
                *reinterpret_cast<uint4*>(hp) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
                if (p == 0 && a.ss_out) a.ss_out[(size_t)tile * a.ss_ld + m] = ss + other;
            }
        } else {
            if (m >= a.M) continue;
            if (EPI == 2 || a.y_tiled)
                *reinterpret_cast<uint4*>(a.y + act_tiled_offset(m, n, a.yMB)) = v;
            else
                *reinterpret_cast<uint4*>(a.y + (size_t)m * a.ldy + n) = v;
        }
    }
    __syncthreads();  // LDS is reused by the next tile / phase
}

// all tiles of one GEMM phase owned by this workgroup; the first one uses the prefetched weights
template <int MB, int EPI, int CH, bool NORM>
__device__ __forceinline__ void gemm_phase(const GemmDesc& g, int G, uint4 (&wf)[CH][(EPI == 2) ? 2 : 1][4], Smem& sm) {
    constexpr int NT = (EPI == 2) ? 2 : 1;
    const int ntiles = g.N / (16 * NT);
    bool first = true;
    for (int tile = blockIdx.x; tile < ntiles; tile += G) {
        if (!first) gemm_load_w<NT, CH>(g, tile, wf);
        gemm_tile<MB, EPI, CH, NORM>(g, tile, wf, sm);
        first = false;
    }
}
template <int NT, int CH>
__device__ __forceinline__ void gemm_prefetch(const GemmDesc& g, uint4 (&wf)[CH][NT][4]) {
    if ((int)blockIdx.x < g.N / (16 * NT)) {
        gemm_load_w<NT, CH>(g, blockIdx.x, wf);
    } else {  // fully (re)defined on every path: the registers are dead between a phase and the next prefetch
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[c][t][i] = make_uint4(0, 0, 0, 0);
    }
}

// ---- attention unit (attn_decode.hip arithmetic) on the first NTH threads of the workgroup ---------------------------
__device__ __forceinline__ void norm_rope(float x0, float x1, const uint16_t* w, float eps, const uint16_t* cosr,
                                          const uint16_t* sinr, int lane, float& o0, float& o1) {
    float ss = wave_sum(x0 * x0 + x1 * x1);
    float rstd = 1.0f / sqrtf(ss / (float)D + eps);
    float y0 = rbf(rbf(x0 * rstd) * bf2f(w[lane]));
    float y1 = rbf(rbf(x1 * rstd) * bf2f(w[lane + 64]));
    o0 = rbf(rbf(y0 * bf2f(cosr[lane])) + rbf(-y1 * bf2f(sinr[lane])));
    o1 = rbf(rbf(y1 * bf2f(cosr[lane + 64])) + rbf(y0 * bf2f(sinr[lane + 64])));
}

template <int REP, int NTH>
__device__ __forceinline__ void attn_unit(const StackPersistArgs& a, const PersistLayer& L, int kvh, int b, Smem& sm) {
    constexpr int NG = NTH / 16, NWV = NTH / 64;
    const int tid = phase_tid(), lane = tid & 63, wave = tid >> 6;
    const bool on = tid < NTH;
    const int len = a.kv_len[b];
    const bool append = a.active ? (a.active[b] != 0) : true;
    const uint16_t* row = a.qkv + (size_t)b * a.ld_qkv;
    const uint16_t* cosr = a.rope_cos + (size_t)len * D;
    const uint16_t* sinr = a.rope_sin + (size_t)len * D;
    const int qdim = a.n_heads * D, kdim = a.n_kv * D;
    const int32_t* bt = a.block_table + (size_t)b * a.max_pages;
    const int npage = bt[len / kPageTokens];
    const size_t nslot = (((size_t)npage * a.n_kv + kvh) * kPageTokens + (len % kPageTokens)) * D;

    if (on) {
        for (int j = wave; j < REP + 2; j += NWV) {
            if (j < REP) {
                const uint16_t* qp = row + (size_t)(kvh * REP + j) * D;
                float o0, o1;
                norm_rope(bf2f(ld_coh_bf(qp, lane)), bf2f(ld_coh_bf(qp, lane + 64)), L.qn, a.eps, cosr, sinr, lane, o0, o1);
                sm.a.q_s[j][lane] = o0;
                sm.a.q_s[j][lane + 64] = o1;
            } else if (j == REP) {
                const uint16_t* kp = row + qdim + (size_t)kvh * D;
                float o0, o1;
                norm_rope(bf2f(ld_coh_bf(kp, lane)), bf2f(ld_coh_bf(kp, lane + 64)), L.kn, a.eps, cosr, sinr, lane, o0, o1);
                sm.a.k_s[lane] = o0;
                sm.a.k_s[lane + 64] = o1;
                if (append) {
                    L.kpool[nslot + lane] = f2bf(o0);
                    L.kpool[nslot + lane + 64] = f2bf(o1);
                }
            } else {
                const uint16_t* vp = row + qdim + kdim + (size_t)kvh * D;
                const uint16_t v0 = ld_coh_bf(vp, lane), v1 = ld_coh_bf(vp, lane + 64);
                sm.a.v_s[lane] = bf2f(v0);
                sm.a.v_s[lane + 64] = bf2f(v1);
                if (append) {
                    L.vpool[nslot + lane] = v0;
                    L.vpool[nslot + lane + 64] = v1;
                }
            }
        }
    }
    __syncthreads();

    const int g = tid >> 4, c = tid & 15;
    if (on) {
        float q[REP][8];
#pragma unroll
        for (int h = 0; h < REP; ++h)
#pragma unroll
            for (int j = 0; j < 8; ++j) q[h][j] = sm.a.q_s[h][8 * c + j];
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
                const float p = __expf(sc - mn);
                l[h] = l[h] * alpha + p;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[h][j] = acc[h][j] * alpha + p * vf[j];
                m[h] = mn;
            }
        };
        for (int t = g; t < len; t += NG) {
            const int page = bt[t / kPageTokens];
            const size_t off = (((size_t)page * a.n_kv + kvh) * kPageTokens + (t % kPageTokens)) * D + 8 * c;
            const uint4 kr = *reinterpret_cast<const uint4*>(L.kpool + off);
            const uint4 vr = *reinterpret_cast<const uint4*>(L.vpool + off);
            float kf[8] = {lo_bf(kr.x), hi_bf(kr.x), lo_bf(kr.y), hi_bf(kr.y), lo_bf(kr.z), hi_bf(kr.z), lo_bf(kr.w), hi_bf(kr.w)};
            float vf[8] = {lo_bf(vr.x), hi_bf(vr.x), lo_bf(vr.y), hi_bf(vr.y), lo_bf(vr.z), hi_bf(vr.z), lo_bf(vr.w), hi_bf(vr.w)};
            step(kf, vf);
        }
        if (g == (len % NG)) {
            float kf[8], vf[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                kf[j] = sm.a.k_s[8 * c + j];
                vf[j] = sm.a.v_s[8 * c + j];
            }
            step(kf, vf);
        }
#pragma unroll
        for (int h = 0; h < REP; ++h) {
            if (c == 0) {
                sm.a.m_s[g][h] = m[h];
                sm.a.l_s[g][h] = l[h];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) sm.a.acc_s[g][h][8 * c + j] = acc[h][j];
        }
    }
    __syncthreads();

    uint16_t* out_s = reinterpret_cast<uint16_t*>(&sm.a.q_s[0][0]);
    if (on) {
        for (int o = tid; o < REP * D; o += NTH) {
            const int h = o / D, d = o % D;
            float M = -INFINITY;
#pragma unroll
            for (int gg = 0; gg < NG; ++gg) M = fmaxf(M, sm.a.m_s[gg][h]);
            float num = 0.f, den = 0.f;
#pragma unroll
            for (int gg = 0; gg < NG; ++gg) {
                const float w = (sm.a.m_s[gg][h] == -INFINITY) ? 0.f : __expf(sm.a.m_s[gg][h] - M);
                num += sm.a.acc_s[gg][h][d] * w;
                den += sm.a.l_s[gg][h] * w;
            }
            out_s[o] = f2bf(num / den);
        }
    }
    __syncthreads();
    if (on) {
        for (int p = tid; p < REP * D / 8; p += NTH) {
            const int col = (kvh * REP) * D + 8 * p;
            *reinterpret_cast<uint4*>(a.ao + act_tiled_offset(b, col, a.MBL)) = *reinterpret_cast<const uint4*>(out_s + 8 * p);
        }
    }
    __syncthreads();
}

// ---- the persistent stack forward ---------------------------------------------------------------------------------
// CHH / CHQ / CHI: 128-wide K chunks per wave for K = hidden, K = n_heads*128, K = padded intermediate size.
template <int MB, int CHH, int CHQ, int CHI, int CHP>
__global__ __launch_bounds__(NW * 64) void stack_persist_kernel(StackPersistArgs a) {
    __shared__ Smem sm;
    const int G = gridDim.x;
    GridSync gs{a.flags, a.err, 0u, G};
    gs.target = __hip_atomic_load(a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned epoch0 = gs.target;

    uint4 w_h1[CHH][1][4];  // qkv / head tiles (K = hidden)
    uint4 w_q1[CHQ][1][4];  // o_proj tiles (K = n_heads * 128)
    uint4 w_h2[CHH][2][4];  // gate/up tile pairs
    uint4 w_i1[CHI][1][4];  // down_proj tiles

    auto base_desc = [&]() {
        GemmDesc g{};
        g.xMB = a.MBL; g.M = a.M; g.ss_ld = a.ss_ld; g.yMB = a.MBL;
        return g;
    };
    int ss_count = a.ss_count_in;

    // optional input projection (small_to_mtp_projection, CodePredictor.swift:327-330): h <- bf16(W x + b)
    if (a.proj_W) {
        GemmDesc p = base_desc();
        p.W = a.proj_W; p.x = a.proj_x; p.N = a.H; p.K = a.proj_K; p.y = a.h; p.bias = a.proj_bias; p.resid = 0; p.ss_out = a.ss_a;
        p.norm_w = a.proj_norm_w; p.ss_in = a.proj_ss_in; p.ss_count = a.proj_ss_count; p.norm_dim = a.proj_norm_dim;
        p.norm_eps = a.proj_norm_eps;
        uint4 w_p[CHP][1][4];
        gemm_prefetch<1, CHP>(p, w_p);
        if (a.proj_norm_w) gemm_phase<MB, 3, CHP, true>(p, G, w_p, sm);
        else gemm_phase<MB, 3, CHP, false>(p, G, w_p, sm);
        ss_count = a.H / 16;
    }

    for (int l = 0; l < a.n_layers; ++l) {
        const PersistLayer& L = a.layers[l];
        // The activation buffers are the same in every layer; without this the compiler hoists every load address of
        // every phase out of the layer loop and spills ~200 registers of them.
        uint16_t *xh = a.h, *xao = a.ao, *xact = a.act, *xqkv = a.qkv;
        float *sa = a.ss_a, *sb = a.ss_b;
        asm volatile("" : "+s"(xh), "+s"(xao), "+s"(xact), "+s"(xqkv), "+s"(sa), "+s"(sb));
        // ---- P1: qkv = RMSNorm(h) Wqkv^T ----
        GemmDesc q = base_desc();
        q.W = L.qkv; q.x = xh; q.N = a.QD + 2 * a.KD; q.K = a.H; q.y = xqkv; q.ldy = a.ld_qkv;
        q.norm_w = L.ln1; q.ss_in = sa; q.ss_count = (l == 0) ? ss_count : a.H / 16; q.norm_dim = a.H; q.norm_eps = a.eps;
        if (l == 0) {
            if (a.proj_W) {  // the projection's outputs must be complete first
                barrier_arrive(gs);
                gemm_prefetch<1, CHH>(q, w_h1);
                if (!barrier_wait(gs, &sm.ok)) return;
            } else {
                gemm_prefetch<1, CHH>(q, w_h1);
            }
        }
        gemm_phase<MB, 0, CHH, true>(q, G, w_h1, sm);
        barrier_arrive(gs);
        if (!barrier_wait(gs, &sm.ok)) return;
        // ---- P2: attention ----
        for (int u = blockIdx.x; u < a.n_kv * a.M; u += G) {
            const int kvh = u % a.n_kv, b = u / a.n_kv;
            // the lane-group counts of attn_decode.hip: 16 for the code predictor's one-page cache, 32 for long caches
            if (a.max_pages == 1) attn_unit<kRepPersist, 256>(a, L, kvh, b, sm);
            else attn_unit<kRepPersist, 512>(a, L, kvh, b, sm);
        }
        GemmDesc o = base_desc();
        o.W = L.o; o.x = xao; o.N = a.H; o.K = a.QD; o.y = xh; o.resid = 1; o.ss_out = sb;
        barrier_arrive(gs);
        gemm_prefetch<1, CHQ>(o, w_q1);
        if (!barrier_wait(gs, &sm.ok)) return;
        // ---- P3: h += attn Wo^T ----
        gemm_phase<MB, 3, CHQ, false>(o, G, w_q1, sm);
        GemmDesc gu = base_desc();
        gu.W = L.gateup; gu.x = xh; gu.N = 2 * L.inter_p; gu.K = a.H; gu.y = xact;
        gu.norm_w = L.ln2; gu.ss_in = sb; gu.ss_count = a.H / 16; gu.norm_dim = a.H; gu.norm_eps = a.eps;
        barrier_arrive(gs);
        gemm_prefetch<2, CHH>(gu, w_h2);
        if (!barrier_wait(gs, &sm.ok)) return;
        // ---- P4: act = silu(g) * u ----
        gemm_phase<MB, 2, CHH, true>(gu, G, w_h2, sm);
        GemmDesc dn = base_desc();
        dn.W = L.down; dn.x = xact; dn.N = a.H; dn.K = L.inter_p; dn.y = xh; dn.resid = 1; dn.ss_out = sa;
        barrier_arrive(gs);
        gemm_prefetch<1, CHI>(dn, w_i1);
        if (!barrier_wait(gs, &sm.ok)) return;
        // ---- P5: h += act Wdown^T ----
        gemm_phase<MB, 3, CHI, false>(dn, G, w_i1, sm);
        // next: qkv of layer l+1, or the head
        const bool last = (l + 1 == a.n_layers);
        if (!last || a.head_W) {
            GemmDesc nx = base_desc();
            nx.x = xh; nx.K = a.H; nx.ss_in = sa; nx.ss_count = a.H / 16; nx.norm_dim = a.H; nx.norm_eps = a.eps;
            if (!last) { nx.W = a.layers[l + 1].qkv; nx.N = a.QD + 2 * a.KD; }
            else { nx.W = a.head_W; nx.N = a.head_N; }
            barrier_arrive(gs);
            gemm_prefetch<1, CHH>(nx, w_h1);
            if (!barrier_wait(gs, &sm.ok)) return;
        }
    }
    if (a.head_W) {  // logits = RMSNorm_final(h) Whead^T (Talker.swift:573,644; CodePredictor.swift:335-338)
        GemmDesc hd = base_desc();
        hd.W = a.head_W; hd.x = a.h; hd.N = a.head_N; hd.K = a.H; hd.y = a.logits; hd.ldy = a.ld_logits;
        hd.norm_w = a.head_norm_w; hd.ss_in = a.ss_a; hd.ss_count = a.H / 16; hd.norm_dim = a.H; hd.norm_eps = a.eps;
        gemm_phase<MB, 0, CHH, true>(hd, G, w_h1, sm);
    }
    // the next launch continues the flag sequence where this one stopped (every workgroup has read epoch0 long ago)
    if (blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(a.epoch, gs.target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    (void)epoch0;
}

template <int MB>
void launch_mb(const StackPersistArgs& a, int grid, hipStream_t st) {
    const int chh = (a.H / 128 + NW - 1) / NW, chq = (a.QD / 128 + NW - 1) / NW, chi = (a.I_p / 128 + NW - 1) / NW;
    const int chp = a.proj_W ? (a.proj_K / 128 + NW - 1) / NW : 1;
#define Q3_SP(A, B, C, P) hipLaunchKernelGGL((stack_persist_kernel<MB, A, B, C, P>), dim3(grid), dim3(NW * 64), 0, st, a)
    if (chh == 1 && chq == 1 && chi == 1 && chp == 1) Q3_SP(1, 1, 1, 1);       // tiny test models
    else if (chh == 1 && chq == 2 && chi == 3 && chp == 1) Q3_SP(1, 2, 3, 1);  // 0.6B talker; code predictor without projection
    else if (chh == 1 && chq == 2 && chi == 3 && chp == 2) Q3_SP(1, 2, 3, 2);  // code predictor behind a 2048-wide talker
    else if (chh == 2 && chq == 2 && chi == 6 && chp == 1) Q3_SP(2, 2, 6, 1);  // 1.7B talker
    else throw Error(3, "stack_persist: no kernel instance for these layer widths");
#undef Q3_SP
}

}  // namespace

bool stack_persist_supported(int H, int QD, int I_p, int proj_K, int n_heads, int n_kv) {
    const int chh = (H / 128 + NW - 1) / NW, chq = (QD / 128 + NW - 1) / NW, chi = (I_p / 128 + NW - 1) / NW;
    const int chp = proj_K ? (proj_K / 128 + NW - 1) / NW : 1;
    if (n_kv <= 0 || n_heads != kRepPersist * n_kv) return false;
    return (chh == 1 && chq == 1 && chi == 1 && chp == 1) || (chh == 1 && chq == 2 && chi == 3 && chp <= 2) ||
           (chh == 2 && chq == 2 && chi == 6 && chp == 1);
}

void launch_stack_persist(const StackPersistArgs& a, int grid, hipStream_t st) {
    Q3_CHECK(a.H % 128 == 0 && a.QD % 128 == 0 && a.I_p % 128 == 0, 3, "stack_persist: widths must be multiples of 128");
    Q3_CHECK(a.M >= 1 && a.M <= 64 && a.MBL * 16 >= a.M, 3, "stack_persist: bad batch");
    Q3_CHECK(grid >= 1 && grid <= 512, 3, "stack_persist: grid out of range");
    Q3_CHECK(a.n_heads == kRepPersist * a.n_kv, 3, "stack_persist: GQA ratio must be 2");
    switch ((a.M + 15) / 16) {
        case 1: launch_mb<1>(a, grid, st); break;
        case 2: launch_mb<2>(a, grid, st); break;
        case 3: launch_mb<3>(a, grid, st); break;
        default: launch_mb<4>(a, grid, st); break;
    }
}

}  // namespace q3
