// gemm_prefill.hip -- the tall form of the decode GEMM for prefill chunks (more than 64 activation rows per launch).
//
// Same operation as gemm_decode.hip (MLXNN.Linear on the talker / code-predictor stacks, Talker.swift:183-186, 413-415;
// CodePredictor.swift:90-93, 152-154) and the SAME result bit for bit, so a row's prompt may be prefilled sixteen positions
// at a time or one at a time (row independence). What fixes a result in the skinny kernel is, per
// output element: the MFMA sequence of each of its NW wave partials (wave w takes the 128-wide k chunks w, w + NW, ... in
// order, four 32-wide MFMAs per chunk in order) and then the fp32 sum of the partials in wave order starting from 0.
// A workgroup here runs those partials one after the other -- phase w accumulates chunks w, w + NW, ... into `acc`, then
// `total += acc` -- which is the same arithmetic in the same order with TWO accumulator sets instead of NW, so an output
// tile can be as tall and wide as a wave's registers allow and both operands can be shared through LDS:
//   * 4 waves as 2 x 2, each WM row blocks (16 activation rows) x WN weight tiles (16 weight rows); the workgroup stages
//     half a k chunk (64 k) of its 2 WN weight tiles and 2 WM row blocks per step, double-buffered. Both operands already
//     lie in MFMA fragment order in global memory (repack.hip, common.h act_tiled_offset), so a 1 KiB fragment is one
//     wave-wide 16-byte load, one ds_write_b128 and later one conflict-free ds_read_b128 per lane -- no swizzle anywhere.
//   * weight rows are the MFMA A operand, activation rows the B operand, exactly as in the skinny kernel.
//   * the epilogues (EPI 0 / 2 / 3 of gemm_body.inc) run from the accumulator registers with the skinny kernel's
//     expressions; the lane exchanges they need (gate | up halves of a tile, the two 4-feature halves of an 8-feature
//     piece, the two pieces of a tile's sum of squares) are wave shuffles.
// Roofline: these are small GEMMs for the part (51 GFLOP per 512-row layer pass, 20 us at the bf16 peak): with one to three
// workgroups per CU each streams (TM + TN) x K x 2 bytes from L2 at ~64 B/clk/CU, which bounds them before the matrix
// pipe does; the skinny form re-read x once per 64 columns and ran at 0.4 PF/s.
#include <cstdlib>

#include "../common.h"
#include "../kernels.h"

namespace q3 {

namespace {

// fragments travel as a plain vector type: with HIP's uint4 (a struct) every copy is a memcpy, and hipcc then kept the
// prefetch ring in scratch memory
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
__device__ __forceinline__ f32x4 mfma16t(const u32x4& a, const u32x4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ float silu_t(float v) { return v / (1.0f + __expf(-v)); }  // gemm_body.inc silu_f

struct Stage {  // position in the skinny kernel's k order: phase (wave) w, chunk kc, half of the chunk
    int w, kc, half;
};

template <int GM, int GN, int WM, int WN, int EPI, int D>
__global__ __launch_bounds__(64 * GM * GN, 2) void gemm_tall_kernel(GemmArgs a, int nphase) {
    constexpr int NWV = GM * GN;                 // waves: GM along the rows x GN along the weight tiles
    constexpr int TMB = GM * WM, TNT = GN * WN;  // row blocks / weight tiles per workgroup
    constexpr int FR = (TMB + TNT) * 2;          // 1 KiB fragments per stage: two per weight tile, then two per row block
    constexpr int LV = FR / NWV;                 // fragments each wave moves per stage
    static_assert((2 * TNT) % NWV == 0 && (2 * TMB) % NWV == 0, "whole weight / activation fragments per wave");
    static_assert(D == 2 || D == 4 || D == 8 || D == 16, "ring depth (even: the LDS buffer of a ring slot is a compile-time value)");
    __shared__ u32x4 lds[2][FR][64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / GN, wn = wave % GN;
    const int n_tiles = (EPI == 2) ? (a.N >> 3) : (a.N >> 4);
    const int KC = a.K >> 7, MBt = a.Mpad >> 4;
    const int tile0 = blockIdx.x * TNT, mb0 = blockIdx.y * TMB;

    // the fragments this wave moves: f = v * NWV + wave, the first 2 TNT of them weights (v < LW: a compile-time split);
    // ragged edges re-read the last tile / row block (never stored)
    constexpr int LW = 2 * TNT / NWV;
    const u32x4* src[LV];
#pragma unroll
    for (int v = 0; v < LV; ++v) {
        if (v < LW) {
            const int f = v * NWV + wave, t = tile0 + (f >> 1);
            src[v] = reinterpret_cast<const u32x4*>(a.W) + ((size_t)(t < n_tiles ? t : n_tiles - 1) * KC * 4 + (f & 1)) * 64 + lane;
        } else {
            const int f = (v - LW) * NWV + wave, mb = mb0 + (f >> 1);
            src[v] = reinterpret_cast<const u32x4*>(a.x) + ((size_t)(mb < MBt ? mb : MBt - 1) * 4 + (f & 1)) * 64 + lane;
        }
    }
    const size_t xstride = (size_t)a.xMB * 256;
    // A stage is requested D stages before its MFMAs: a workgroup's K loop is a chain of first-touch reads (every weight
    // byte is used once per workgroup), and with one stage in flight each link cost a whole memory round trip -- 2.3 us per
    // stage measured, 70-150 us per GEMM. ring[j] holds stage s + 1 while stage s (s % D == j) is at the matrix pipe.
    // (Macros, and loads / stores that are never conditional -- trips past the end re-read their own stage: lambdas that
    // capture the arrays send them to scratch, and a load under a branch makes hipcc wait for everything in flight.)
    u32x4 ring[D][LV];
#define Q3_LOAD_STAGE(J, S)                                          \
    _Pragma("unroll") for (int v = 0; v < LV; ++v) ring[J][v] = src[v][(size_t)(S).kc * (v < LW ? (size_t)256 : xstride) + (S).half * 128]
#define Q3_STORE_STAGE(BUF, J) \
    _Pragma("unroll") for (int v = 0; v < LV; ++v) lds[BUF][v * NWV + wave][lane] = ring[J][v]
    auto advance = [nphase, KC](Stage& s) {
        if (s.half == 0) { s.half = 1; return; }
        s.half = 0;
        s.kc += nphase;
        if (s.kc >= KC) {  // next phase; if that one starts past the last chunk so do all later ones
            ++s.w;
            s.kc = s.w;
            if (s.kc >= KC) s.w = nphase;
        }
    };

    f32x4 acc[WN][WM], total[WN][WM];
#pragma unroll
    for (int c = 0; c < WN; ++c)
#pragma unroll
        for (int p = 0; p < WM; ++p) {
            acc[c][p] = f32x4{0.f, 0.f, 0.f, 0.f};
            total[c][p] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

    const int stages = 2 * KC;
    Stage cur{0, 0, 0}, nxt{0, 0, 0}, pre{0, 0, 0};
    int pre_s = 0;  // the stage `pre` points at
    Q3_LOAD_STAGE(0, cur);
    Q3_STORE_STAGE(0, 0);
    advance(nxt);
    // (the ring steps are spelled out by macro, not by an unrolled loop over j: hipcc left `ring` in scratch memory with the loop)
#define Q3_FILL(J)                                             \
    if (pre_s + 1 < stages) { advance(pre); ++pre_s; }         \
    Q3_LOAD_STAGE(J, pre)
    Q3_FILL(0); Q3_FILL(1);
    if constexpr (D > 2) { Q3_FILL(2); Q3_FILL(3); }
    if constexpr (D > 4) { Q3_FILL(4); Q3_FILL(5); Q3_FILL(6); Q3_FILL(7); }
    if constexpr (D > 8) { Q3_FILL(8); Q3_FILL(9); Q3_FILL(10); Q3_FILL(11); Q3_FILL(12); Q3_FILL(13); Q3_FILL(14); Q3_FILL(15); }
#undef Q3_FILL
    __syncthreads();
    // one stage: stage s + 1 leaves ring[J] for the other LDS buffer (last read one trip ago, which every wave left through
    // the barrier) and ring[J] takes stage s + 1 + D; THEN the MFMAs of stage s on buffer J & 1, so that the writes' latency
    // hides behind the matrix pipe (written the other way round -- MFMAs, then the parking -- a stage was a serial chain
    // ds_write -> barrier -> ds_read -> MFMA of ~1000 clocks whatever the ring depth); fold at the end of a phase
#define Q3_STEP(J, GUARD)                                                                                            \
    if (!(GUARD) || s0 + (J) < stages) { /* uniform */                                                              \
        const bool more = s0 + (J) + 1 < stages;                                                                    \
        Q3_STORE_STAGE(((J) & 1) ^ 1, J);                                                                           \
        if (pre_s + 1 < stages) { advance(pre); ++pre_s; }                                                          \
        Q3_LOAD_STAGE(J, pre);                                                                                      \
        _Pragma("unroll") for (int hf = 0; hf < 2; ++hf) {                                                          \
            u32x4 wv[WN], xv[WM];                                                                                   \
            _Pragma("unroll") for (int c = 0; c < WN; ++c) wv[c] = lds[(J) & 1][(wn * WN + c) * 2 + hf][lane];          \
            _Pragma("unroll") for (int p = 0; p < WM; ++p) xv[p] = lds[(J) & 1][2 * TNT + (wm * WM + p) * 2 + hf][lane]; \
            _Pragma("unroll") for (int c = 0; c < WN; ++c)                                                          \
                _Pragma("unroll") for (int p = 0; p < WM; ++p) acc[c][p] = mfma16t(wv[c], xv[p], acc[c][p]);        \
        }                                                                                                           \
        if (cur.half == 1 && (!more || nxt.w != cur.w)) {                                                           \
            _Pragma("unroll") for (int c = 0; c < WN; ++c)                                                          \
                _Pragma("unroll") for (int p = 0; p < WM; ++p) {                                                    \
                    total[c][p] += acc[c][p];                                                                       \
                    acc[c][p] = f32x4{0.f, 0.f, 0.f, 0.f};                                                          \
                }                                                                                                   \
        }                                                                                                           \
        cur = nxt;                                                                                                  \
        advance(nxt);                                                                                               \
        __syncthreads();                                                                                            \
    }
    // Whole groups of D stages run without a guard around the steps: with one, hipcc's wait-count bookkeeping merged the
    // skipped and the taken paths at the loop header and drained the whole ring (s_waitcnt vmcnt(0)) once per trip.
#define Q3_GROUP(G)                                                                                       \
    Q3_STEP(0, G) Q3_STEP(1, G)                                                                           \
    if constexpr (D > 2) { Q3_STEP(2, G) Q3_STEP(3, G) }                                                  \
    if constexpr (D > 4) { Q3_STEP(4, G) Q3_STEP(5, G) Q3_STEP(6, G) Q3_STEP(7, G) }                      \
    if constexpr (D > 8) { Q3_STEP(8, G) Q3_STEP(9, G) Q3_STEP(10, G) Q3_STEP(11, G) Q3_STEP(12, G) Q3_STEP(13, G) Q3_STEP(14, G) Q3_STEP(15, G) }
    int s0 = 0;
    for (; s0 + D <= stages; s0 += D) { Q3_GROUP(false) }
    if (s0 < stages) { Q3_GROUP(true) }
#undef Q3_GROUP
#undef Q3_STEP

#undef Q3_LOAD_STAGE
#undef Q3_STORE_STAGE

    // epilogue: lane holds rows n = 4 * (lane >> 4) + q (q = 0..3) of weight tile c at activation row m = 16 * mb + (lane & 15)
    const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int c = 0; c < WN; ++c) {
        const int otile = tile0 + wn * WN + c;
#pragma unroll
        for (int p = 0; p < WM; ++p) {
            const int m = 16 * (mb0 + wm * WM + p) + lr;
            const bool live = otile < n_tiles && m < a.M;
            const f32x4 t = total[c][p];
            if constexpr (EPI == 0) {
                const int n = otile * 16 + 4 * lq;
                const uint2 v = make_uint2(pack_bf(t[0], t[1]), pack_bf(t[2], t[3]));
                if (live) {
                    if (a.y_tiled) *reinterpret_cast<uint2*>(a.y + act_tiled_offset(m, n, a.yMB)) = v;
                    else *reinterpret_cast<uint2*>(a.y + (size_t)m * a.ldy + n) = v;
                }
            } else if constexpr (EPI == 2) {
                // tile rows 0..7 gate, 8..15 up of the same eight columns: lanes 32..63 hold the ups of lanes 0..31's gates
                float y[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float u = __shfl_xor(t[q], 32, 64);
                    y[q] = rbf(silu_t(rbf(t[q]))) * rbf(u);
                }
                const uint2 v = make_uint2(pack_bf(y[0], y[1]), pack_bf(y[2], y[3]));
                if (live && lq < 2) *reinterpret_cast<uint2*>(a.y + act_tiled_offset(m, otile * 8 + 4 * lq, a.yMB)) = v;
            } else {
                // hidden-state piece of eight features (gemm_body.inc EPI 3): lanes with an even lane >> 4 take the four
                // features of the lane sixteen up as well
                const uint32_t y01 = pack_bf(t[0], t[1]), y23 = pack_bf(t[2], t[3]);
                const uint32_t z01 = __shfl_down(y01, 16, 64), z23 = __shfl_down(y23, 16, 64);
                const uint32_t yw[4] = {y01, y23, z01, z23};
                const int pc = lq >> 1;
                const bool mine = live && (lq & 1) == 0;
                uint16_t* hp = a.y + act_tiled_offset(m, otile * 16 + 8 * pc, a.yMB);
                uint4 hv = make_uint4(0, 0, 0, 0);
                if (a.resid && mine) hv = *reinterpret_cast<const uint4*>(hp);
                const uint32_t hw[4] = {hv.x, hv.y, hv.z, hv.w};
                uint32_t ow[4];
                float ss = 0.f;
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
                const float other = __shfl_xor(ss, 32, 64);  // the tile's second eight-feature piece of this row
                if (mine) {
                    *reinterpret_cast<uint4*>(hp) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
                    if (pc == 0 && a.ss_out) a.ss_out[(size_t)otile * a.ss_ld + m] = ss + other;
                }
            }
        }
    }
}

template <int GM, int GN, int WM, int WN, int D>
void launch_shape(const GemmArgs& a, int n_tiles, int nphase, hipStream_t st) {
    const dim3 grid((n_tiles + GN * WN - 1) / (GN * WN), ((a.Mpad >> 4) + GM * WM - 1) / (GM * WM)), block(64 * GM * GN);
    switch (a.epi) {
        case 0: hipLaunchKernelGGL((gemm_tall_kernel<GM, GN, WM, WN, 0, D>), grid, block, 0, st, a, nphase); break;
        case 2: hipLaunchKernelGGL((gemm_tall_kernel<GM, GN, WM, WN, 2, D>), grid, block, 0, st, a, nphase); break;
        default: hipLaunchKernelGGL((gemm_tall_kernel<GM, GN, WM, WN, 3, D>), grid, block, 0, st, a, nphase); break;
    }
}

}  // namespace

bool gemm_tall_takes(const GemmArgs& a) {
    if (debug_env().no_tall_gemm || a.Mpad <= 64 || a.Wsb || a.norm_w || a.bias || a.act_silu) return false;
    if (a.epi != 0 && a.epi != 2 && a.epi != 3) return false;
    if (a.epi == 0 && a.N % 16 != 0) return false;
    return true;
}

bool launch_gemm_tall(const GemmArgs& a, hipStream_t st) {
    if (!gemm_tall_takes(a)) return false;
    const int KC = a.K / 128;
    const int nphase = KC <= 4 ? 4 : 8;  // gemm_decode.hip launch_mb: waves that split K
    const int n_tiles = a.epi == 2 ? a.N / 8 : a.N / 16;
    const int MBt = a.Mpad / 16;
    // 128 activation rows x 64 weight rows where that still gives (nearly) every CU a workgroup, else 64 x 64. Measured per
    // launch at 512 rows, 1.7B (qkv / o_proj / gate-up / down_proj, us): 128 x 64 ring 4: 20.9 / 21.1 / 49.5 / 47.3;
    // 64 x 64 ring 8: - / 16.1 / - / 35.1; 128 x 128 ring 2: 38 / 40 / 55 / 83; eight-wave workgroups and deeper rings
    // (128 x 64 ring 8: 66 / 56 / 171 / 149) lose: past ~32 wave-wide loads in flight per wave the CU's own miss queue is the
    // limit and further loads stall the wave at issue. Q3TTS_TALL_SHAPE = 2 | 3 forces one (diagnostics).
    const int force = debug_env().tall_shape;
    auto wgs = [&](int tm, int tn) { return ((MBt + tm - 1) / tm) * ((n_tiles + tn - 1) / tn); };
    int shape = wgs(8, 4) >= 192 ? 2 : 3;
    if (force == 2 || force == 3) shape = force;
    if (shape == 2) launch_shape<2, 2, 4, 2, 4>(a, n_tiles, nphase, st);
    else launch_shape<2, 2, 2, 2, 8>(a, n_tiles, nphase, st);
    return true;
}

}  // namespace q3
