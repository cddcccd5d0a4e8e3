// gemm_decode.hip -- skinny (M <= 64) bf16 weight-streaming GEMM for the decode path.
//
// Replaces every MLXNN.Linear on the AR loop (/root/reference/Sources/Qwen3TTS/Models/
// Talker.swift:183-186,413-415,480-481,607; CodePredictor.swift:90-93,152-154,296,305):
//   y[m][n] = sum_k x[m][k] * W[n][k]   (fp32 accumulate, one rounding to bf16 per output)
// together with the ops around it in the pre-norm block (Talker.swift:451-469), so that a decoder
// layer is 5 launches (qkv, attention, o_proj, gate/up, down) instead of 7:
//   NORM prologue : x = RMSNorm(h) applied to the B fragments in registers while the weight loads
//                   are in flight:  bf16( bf16(h * rstd) * w )   (MLXNN.RMSNorm, Talker.swift:447-448)
//   EPI 3 epilogue: h <- bf16(h + bf16(acc))  (residual add, Talker.swift:461,466) plus this tile's
//                   share of sum(h^2) per row, which the next NORM prologue reduces in tile order.
//
// Roofline: HBM. Every weight byte is read exactly once per launch; x (<= 64 rows) comes from L2.
// MI355X mapping:
//   * W is re-tiled at load time (repack.hip) into 4 KiB tiles of 16 rows x 128 k laid out
//     [instr i=0..3][lane 0..63][8 bf16], so each global_load_dwordx4 wave-instruction reads 1 KiB
//     contiguous and lands directly in the A fragment of one v_mfma_f32_16x16x32_bf16 (no LDS round
//     trip: cdna_hip_programming.md, "GEMV / M <= 16" row). The k index inside a tile is permuted
//     (k = 32*(lane>>4) + 8*i + j); x fragments use the same permutation.
//   * x (the <= 64 activation rows) is kept in the same fragment-major order by its producers
//     (common.h: act_tiled_offset), so the B-fragment loads are 1 KiB contiguous too. Row-major x
//     costs as much L2->CU time as streaming the weights (16 rows x 64 B per wave-instruction).
//   * out^T tile = W(16 x K) . x^T(K x 16*MB): W rows are the MFMA M dimension, batch rows the N
//     dimension, so one weight fragment feeds MB MFMAs.
//   * a workgroup = NW waves that split K (chunk-interleaved) for one 16-row weight tile and reduce
//     through LDS in fixed wave order -> results do not depend on batch size or launch geometry
//     (row independence is the batching contract). CH (chunks per wave) is a template
//     parameter so every weight load of a wave is issued before its first MFMA.
#include <cstdlib>

#include "../common.h"
#include "../kernels.h"

namespace q3 {

namespace {

__device__ __forceinline__ f32x4 mfma16(const uint4& a, const uint4& b, f32x4 c) {
    bf16x8 av, bv;
    __builtin_memcpy(&av, &a, 16);
    __builtin_memcpy(&bv, &b, 16);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c, 0, 0, 0);
}

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }

// bf16( bf16(h * rstd) * w ) on 8 packed elements
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

// MLX affine int4, group 64 (QuantizedLinear installed by quantize(model:...) at Qwen3.swift:1412-1425):
// w = bf16(q * scale + bias). One dwordx4 per lane holds the 32 nibbles of its four A fragments of a chunk
// (8 consecutive k per uint32, little-endian nibbles), so an int4 weight tile is 1 KiB instead of 4 KiB.
__device__ __forceinline__ void dequant_chunk(const uint4& qw, uint32_t sb, uint4 (&out)[4]) {
    const float sc = lo_bf(sb), bi = hi_bf(sb);
    const uint32_t w[4] = {qw.x, qw.y, qw.z, qw.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float q0 = (float)((w[i] >> (8 * j)) & 15u), q1 = (float)((w[i] >> (8 * j + 4)) & 15u);
            // mul and add rounded separately like the oracle (no fma contraction)
            o[j] = pack_bf(__fadd_rn(__fmul_rn(q0, sc), bi), __fadd_rn(__fmul_rn(q1, sc), bi));
        }
        out[i] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// EPI: 0 = bf16 store (+bias, +optional silu), row-major or fragment-major
//      2 = gate/up tile pair -> bf16(bf16(silu(g)) * u), fragment-major
//      3 = hidden-state store, fragment-major, in place: h = bf16((resid ? h : 0) + bf16(acc + bias)),
//          plus ss_out[tile][m] = sum over the tile's 16 features of h^2
// NORM: RMSNorm prologue on x (x is then the raw residual stream h)
// NP (EPI 2 only): gate/up tile pairs per workgroup. A 6144-wide MLP is 384 pairs: one pair per workgroup runs as a full
// round of 256 workgroups plus a half-empty one; two pairs per workgroup is one round of 192.
template <int MB, int EPI, int NW, int CH, bool NORM, bool QUANT, int NP = 1>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_kernel(GemmArgs a) {
    constexpr int NT = (EPI == 2) ? 2 * NP : 1;  // weight tiles per workgroup
    constexpr int NR = (EPI == 2) ? 2 : 1;       // tiles reduced per pass of the epilogue
    __shared__ float red[NW][NR][MB][4][64];
    __shared__ __attribute__((aligned(16))) uint16_t ys[MB][16][16];
    __shared__ float rstd_s[NORM ? 16 * MB : 1];
    __shared__ float ssp_s[NORM ? 8 : 1][NORM ? 16 * MB : 1];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int tile = blockIdx.x;
    const int mb0 = blockIdx.y * MB;  // first 16-row block of this workgroup (launch_q splits the rows of narrow layers)
    const int KC = a.K >> 7;  // 128-wide k chunks

    f32x4 acc[NT][MB];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[t][mb] = f32x4{0.f, 0.f, 0.f, 0.f};

    const uint4* Wt = reinterpret_cast<const uint4*>(a.W);
    const uint4* Xt = reinterpret_cast<const uint4*>(a.x);

    // 0. NORM: the producer's per-tile sums of squares are requested before anything else. They gate the whole prologue
    //    (rstd -> normalised x fragments), and loads return in issue order: with the sums first, the reduction and the
    //    VALU work on x run while the weight tiles are still arriving instead of after the last of them.
    constexpr int kSsIter = NORM ? (16 * MB * 8 + NW * 64 - 1) / (NW * 64) : 1;
    float sst[kSsIter][16];
    if constexpr (NORM) {
        const int rows = 16 * MB;
#pragma unroll
        for (int it = 0; it < kSsIter; ++it) {
            const int idx = threadIdx.x + it * NW * 64;
            const int row = idx % rows, part = idx / rows;  // (row, part): 8 strided partial sums per row
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int j = part + 8 * u;
                sst[it][u] = (idx < rows * 8) ? a.ss_in[(size_t)(j < a.ss_count ? j : 0) * a.ss_ld + 16 * mb0 + row] : 0.f;
            }
        }
    }

    // 1. weight loads: they depend on nothing
    constexpr int CHR = CH > 0 ? CH : 1;
    constexpr int WL = QUANT ? 1 : 4;  // dwordx4 loads per (chunk, tile): packed int4 needs one
    uint4 wf[CHR][NT][WL];
    uint32_t wsb[CHR][NT];
    auto load_w = [&](int c, int kc) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const size_t blk = (size_t)(tile * NT + t) * KC + kc;
            if constexpr (QUANT) {
                wf[c][t][0] = Wt[blk * 64 + lane];
                wsb[c][t] = a.Wsb[blk * 64 + lane];
            } else {
                const uint4* wp = Wt + blk * 256 + lane;
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[c][t][i] = wp[i * 64];
            }
        }
    };

    // 1b. every other operand that does not depend on arithmetic is requested now as well, so that the kernel pays
    //     ONE memory round trip instead of a chain of three (sums of squares -> x fragments -> residual tile):
    //     the raw x fragments of the first XG chunks, and (EPI 3) the old hidden-state piece this thread will update.
    // chunks whose x fragments are prefetched: as many as fit next to the weight registers (16 VGPRs per fragment set)
    constexpr int kWRegs = CH * NT * (QUANT ? 5 : 16);
    // (one row block leaves room for every chunk of the widest layer, K = 6144: one round trip for the whole workgroup)
    constexpr int kRoom = (((MB == 1 && !NORM) ? 200 : 176) - kWRegs) / (MB * 16);
    constexpr int kCap = (MB == 1 && !NORM) ? 6 : 3;
    constexpr int XG = (CH == 0) ? 0 : (kRoom < 1 ? 1 : (kRoom > kCap ? (CH < kCap ? CH : kCap) : (kRoom < CH ? kRoom : CH)));
    uint4 xr[XG > 0 ? XG : 1][MB][4];
    uint4 nwr[(NORM && XG > 0) ? XG : 1][4];
    // issue order: weights then x, or (NORM) x then weights so that normalising x overlaps the weights' arrival.
    // Written as a two-trip unrolled loop instead of lambdas: capturing the fragment arrays by reference sends them to scratch.
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
        if ((ph == 0) != NORM) {
    if constexpr (CH > 0) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int kl = wave + c * NW;
                load_w(c, kl < KC ? kl : 0);
            }
        }
        } else {
    if constexpr (XG > 0) {
#pragma unroll
        for (int c = 0; c < XG; ++c) {
            const int kl = wave + c * NW;
            const int kc = kl < KC ? kl : 0;
            if constexpr (NORM) {
                const uint4* np = reinterpret_cast<const uint4*>(a.norm_w + kc * 128 + 32 * (lane >> 4));
#pragma unroll
                for (int i = 0; i < 4; ++i) nwr[c][i] = np[i];
            }
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const uint4* xp = Xt + ((size_t)(kc * a.xMB + mb0 + mb) * 4) * 64 + lane;
#pragma unroll
                for (int i = 0; i < 4; ++i) xr[c][mb][i] = xp[i * 64];
            }
        }
    }
        }
    }
    uint4 hv_pre = make_uint4(0, 0, 0, 0);
    if constexpr (EPI == 3) {
        if (a.resid && threadIdx.x < 32 * MB) {
            const int o = threadIdx.x, mb = o >> 5, b = (o >> 1) & 15, p = o & 1;
            hv_pre = *reinterpret_cast<const uint4*>(a.y + act_tiled_offset(16 * (mb0 + mb) + b, tile * 16 + 8 * p, a.yMB));
        }
    }

    // 2. NORM: rstd per row from the producer's per-tile sums of squares, summed in tile order
    float rstd[MB];
    if constexpr (NORM) {
        const int rows = 16 * MB;
#pragma unroll
        for (int it = 0; it < kSsIter; ++it) {
            const int idx = threadIdx.x + it * NW * 64;
            if (idx < rows * 8) {
                const int row = idx % rows, part = idx / rows;
                float s = 0.f;
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (part + 8 * u < a.ss_count) s += sst[it][u];
                for (int j = part + 128; j < a.ss_count; j += 8) s += a.ss_in[(size_t)j * a.ss_ld + 16 * mb0 + row];
                ssp_s[part][row] = s;
            }
        }
        __syncthreads();
        if (threadIdx.x < rows) {
            float s = 0.f;
#pragma unroll
            for (int p = 0; p < 8; ++p) s += ssp_s[p][threadIdx.x];
            rstd_s[threadIdx.x] = 1.0f / sqrtf(s / (float)a.norm_dim + a.norm_eps);
        }
        __syncthreads();
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) rstd[mb] = rstd_s[16 * mb + (lane & 15)];
    }

    // 3. x fragments (+ norm) and MFMAs
    auto chunk = [&](int kc, int c, const uint4 (*xpre)[4], const uint4* nwpre) {
        uint4 w[NT][4];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if constexpr (QUANT) {
                dequant_chunk(wf[c][t][0], wsb[c][t], w[t]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) w[t][i] = wf[c][t][i];
            }
        }
        uint4 nw[4];
        if constexpr (NORM) {
            if (nwpre) {
#pragma unroll
                for (int i = 0; i < 4; ++i) nw[i] = nwpre[i];
            } else {
                const uint4* np = reinterpret_cast<const uint4*>(a.norm_w + kc * 128 + 32 * (lane >> 4));
#pragma unroll
                for (int i = 0; i < 4; ++i) nw[i] = np[i];
            }
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            uint4 xf[4];
            if (xpre) {
#pragma unroll
                for (int i = 0; i < 4; ++i) xf[i] = xpre[mb][i];
            } else {
                const uint4* xp = Xt + ((size_t)(kc * a.xMB + mb0 + mb) * 4) * 64 + lane;
#pragma unroll
                for (int i = 0; i < 4; ++i) xf[i] = xp[i * 64];
            }
            if constexpr (NORM) {
#pragma unroll
                for (int i = 0; i < 4; ++i) xf[i] = norm8(xf[i], nw[i], rstd[mb]);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[t][mb] = mfma16(w[t][i], xf[i], acc[t][mb]);
        }
    };
    if constexpr (CH > 0) {
        // chunks in groups of XG: the first group's fragments are already in flight; each later group is requested as a
        // whole (one round trip per group) into the same registers
#pragma unroll
        for (int g0 = 0; g0 < CH; g0 += XG) {
            if (g0 > 0) {
#pragma unroll
                for (int c = 0; c < XG; ++c) {
                    if (g0 + c < CH) {
                        const int kl = wave + (g0 + c) * NW;
                        const int kc = kl < KC ? kl : 0;
                        if constexpr (NORM) {
                            const uint4* np = reinterpret_cast<const uint4*>(a.norm_w + kc * 128 + 32 * (lane >> 4));
#pragma unroll
                            for (int i = 0; i < 4; ++i) nwr[c][i] = np[i];
                        }
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb) {
                            const uint4* xp = Xt + ((size_t)(kc * a.xMB + mb0 + mb) * 4) * 64 + lane;
#pragma unroll
                            for (int i = 0; i < 4; ++i) xr[c][mb][i] = xp[i * 64];
                        }
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < XG; ++c) {
                if (g0 + c < CH) {
                    const int kl = wave + (g0 + c) * NW;
                    if (kl < KC) chunk(kl, g0 + c, xr[c], NORM ? nwr[c] : nullptr);  // wave-uniform
                }
            }
        }
    } else {
        for (int kl = wave; kl < KC; kl += NW) {
            load_w(0, kl);
            chunk(kl, 0, nullptr, nullptr);
        }
    }

    // 4. cross-wave K reduction through LDS, fixed wave order; one tile (EPI 2: one gate/up pair) per pass
#pragma unroll
    for (int pr = 0; pr < NT / NR; ++pr) {
    if (pr > 0) __syncthreads();  // the previous pair's red / ys reads
#pragma unroll
    for (int t = 0; t < NR; ++t)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int q = 0; q < 4; ++q) red[wave][t][mb][q][lane] = acc[pr * NR + t][mb][q];
    __syncthreads();
    const int otile = tile * (NT / NR) + pr;  // 16-column output tile

    // 256*MB outputs per tile: thread -> (mb, batch row b, feature f)
    for (int o = threadIdx.x; o < 256 * MB; o += NW * 64) {
        const int mb = o >> 8, rem = o & 255;
        const int b = rem >> 4, f = rem & 15;
        const int src_lane = (f >> 2) * 16 + b, q = f & 3;
        float v[NR];
#pragma unroll
        for (int t = 0; t < NR; ++t) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) sum += red[w][t][mb][q][src_lane];
            v[t] = sum;
        }
        const int n = otile * 16 + f;
        if constexpr (EPI == 2) {
            const float g = rbf(v[0]), u = rbf(v[1]);
            ys[mb][b][f] = f2bf(rbf(silu_f(g)) * u);
        } else {
            float y = v[0];
            if (a.bias) y += bf2f(a.bias[n]);
            uint16_t yb = f2bf(y);
            if (EPI == 0 && a.act_silu) yb = f2bf(silu_f(bf2f(yb)));
            ys[mb][b][f] = yb;
        }
    }
    __syncthreads();
    // 16-byte stores: thread -> (mb, row b, 8-feature piece p)
    for (int o = threadIdx.x; o < 32 * MB; o += NW * 64) {
        const int mb = o >> 5, b = (o >> 1) & 15, p = o & 1;
        const int m = 16 * (mb0 + mb) + b;
        uint4 v = *reinterpret_cast<const uint4*>(&ys[mb][b][8 * p]);
        const int n = otile * 16 + 8 * p;
        if constexpr (EPI == 3) {
            uint16_t* hp = a.y + act_tiled_offset(m, n, a.yMB);
            float ss = 0.f;
            const uint32_t yw[4] = {v.x, v.y, v.z, v.w};
            uint32_t ow[4];
            const uint4 hv = hv_pre;  // o == threadIdx.x here (32 * MB <= 128 threads, one trip)
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
            const float other = __shfl_xor(ss, 1, 64);  // the tile's second 8-feature piece of this row
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
    }
}

template <int MB, int EPI, bool NORM, bool QUANT>
void launch_mb(const GemmArgs& a, int split, hipStream_t st) {
    const int KC = a.K / 128;
    const int nw = KC <= 4 ? 4 : 8;  // no idle waves on short K
    const int ch = (KC + nw - 1) / nw;
    const int tiles = a.N / 16;
#define Q3_GEMM(NWv, CHv) \
    hipLaunchKernelGGL((gemm_skinny_kernel<MB, EPI, NWv, CHv, NORM, QUANT>), dim3(tiles, split), dim3(NWv * 64), 0, st, a)
    if constexpr (EPI == 2 && !QUANT && MB <= 2) {
        // more than one round of workgroups on 256 CUs, and an even pair count: two pairs per workgroup, one round
        static const bool one_pair = std::getenv("Q3TTS_GEMM_ONE_PAIR") != nullptr;
        if (!one_pair && nw == 8 && ch == 2 && tiles > 256 && tiles <= 512 && tiles % 2 == 0) {
            hipLaunchKernelGGL((gemm_skinny_kernel<MB, 2, 8, 2, NORM, false, 2>), dim3(tiles / 2, split), dim3(512), 0, st, a);
            return;
        }
    }
    if (nw == 4) {
        Q3_GEMM(4, 1);
    } else {
        switch (ch) {
            case 1: Q3_GEMM(8, 1); break;
            case 2: Q3_GEMM(8, 2); break;
            case 3: Q3_GEMM(8, 3); break;
            case 6: Q3_GEMM(8, 6); break;
            default: Q3_GEMM(8, 0); break;
        }
    }
#undef Q3_GEMM
}

template <int EPI, bool NORM, bool QUANT>
void launch_q(const GemmArgs& a, hipStream_t st) {
    const int MBt = (a.Mpad + 15) / 16;
    // Narrow layers (o_proj, down_proj: N / 16 <= 128 column tiles) leave CUs idle with one workgroup per tile, and a
    // workgroup that carries two row blocks cannot keep every x fragment of a long K in registers. Their row blocks go
    // to separate workgroups instead (grid.y); tile x of both lands on the same XCD (128 = 0 mod 8), so the second
    // read of the weight tile is an L2 hit. Per-row arithmetic does not depend on the grouping: results are unchanged.
    static const bool no_split = std::getenv("Q3TTS_GEMM_NO_ROW_SPLIT") != nullptr;
    const int tiles = a.N / 16;
    int split = 1;
    if (!no_split) {
        if (MBt % 4 == 0 && tiles * 4 <= 256) split = 4;
        else if (MBt % 2 == 0 && tiles * 2 <= 256) split = 2;
    }
    // More than 64 rows (prefill chunks, the predictor's two-position step): at most 4 row blocks per workgroup -- 2 with
    // the norm prologue, whose VALU work on the x fragments (critical path, per workgroup) grows with the row blocks.
    if (!no_split) {
        const int cap = NORM ? 2 : 4;
        while (MBt / split > cap || MBt % split != 0) ++split;
    }
    while (MBt / split > 4 || MBt % split != 0) ++split;
    switch (MBt / split) {
        case 1: launch_mb<1, EPI, NORM, QUANT>(a, split, st); break;
        case 2: launch_mb<2, EPI, NORM, QUANT>(a, split, st); break;
        case 3: launch_mb<3, EPI, NORM, QUANT>(a, split, st); break;
        case 4: launch_mb<4, EPI, NORM, QUANT>(a, split, st); break;
        default: throw Error(3, "gemm_skinny: more than 4 row blocks per workgroup");
    }
}

template <int EPI, bool NORM>
void launch_epi(const GemmArgs& a, hipStream_t st) {
    if (a.Wsb) launch_q<EPI, NORM, true>(a, st);
    else launch_q<EPI, NORM, false>(a, st);
}

}  // namespace

void launch_gemm_skinny(const GemmArgs& a, hipStream_t st) {
    Q3_CHECK(a.K % 128 == 0 && a.N % 16 == 0, 3, "gemm_skinny: K must be a multiple of 128 and N of 16");
    Q3_CHECK(a.Mpad % 16 == 0 && a.M <= a.Mpad && a.Mpad <= 256, 3, "gemm_skinny: bad M padding");
    Q3_CHECK(a.xMB * 16 >= a.Mpad, 3, "gemm_skinny: x allocation has fewer row blocks than the batch");
    const bool norm = a.norm_w != nullptr;
    if (norm) Q3_CHECK(a.ss_in && a.ss_count >= 1 && a.ss_ld >= a.Mpad, 3, "gemm_skinny: norm prologue needs sums of squares");
    switch (a.epi) {
        case 0: norm ? launch_epi<0, true>(a, st) : launch_epi<0, false>(a, st); break;
        case 2: norm ? launch_epi<2, true>(a, st) : launch_epi<2, false>(a, st); break;
        case 3:
            Q3_CHECK(a.yMB * 16 >= a.Mpad, 3, "gemm_skinny: h allocation has fewer row blocks than the batch");
            norm ? launch_epi<3, true>(a, st) : launch_epi<3, false>(a, st);
            break;
        default: throw Error(3, "gemm_skinny: unknown epilogue");
    }
}

}  // namespace q3
