// gemm_decode.hip -- skinny (M <= 64) bf16 weight-streaming GEMM for the decode path.
//
// Replaces every MLXNN.Linear on the AR loop (/root/reference/Sources/Qwen3TTS/Models/
// Talker.swift:183-186,413-415,480-481,607; CodePredictor.swift:90-93,152-154,296,305):
//   y[m][n] = sum_k x[m][k] * W[n][k]   (fp32 accumulate, one rounding to bf16 per output)
//
// Roofline: HBM. Every weight byte is read exactly once per launch; x (<= 64 rows) comes from L2.
// MI355X mapping:
//   * W is re-tiled at load time (weights.cc: tile_weights) into 4 KiB tiles of 16 rows x 128 k
//     laid out [instr i=0..3][lane 0..63][8 bf16], so each global_load_dwordx4 wave-instruction
//     reads 1 KiB contiguous and lands directly in the A fragment of one
//     v_mfma_f32_16x16x32_bf16 (no LDS round trip: cdna_hip_programming.md, "GEMV / M <= 16" row).
//     The k index inside a tile is permuted (k = 32*(lane>>4) + 8*i + j); x fragments use the same
//     permutation, and a dot product does not care about the order of its terms.
//   * x (the <= 64 activation rows) is kept in the same fragment-major order by its producers
//     (common.h: act_tiled_offset), so the B-fragment loads are 1 KiB contiguous too. Row-major x
//     costs as much L2->CU time as streaming the weights (16 rows x 64 B per wave-instruction).
//   * out^T tile = W(16 x K) . x^T(K x 16*MB): W rows are the MFMA M dimension, batch rows the N
//     dimension, so one weight fragment feeds MB MFMAs.
//   * a workgroup = 8 waves that split K (chunk-interleaved) for one 16-row weight tile and
//     reduce through LDS in fixed wave order -> results do not depend on batch size or launch
//     geometry (row independence is the batching contract).
//   * grid = (N/16, S): S > 1 splits K across workgroups for the small-N projections (o_proj,
//     down_proj) and writes fp32 partial slabs that the following resid_norm kernel sums in
//     fixed order.
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

// EPI: 0 = bf16 store (+bias, +optional silu), 1 = fp32 partial slab, 2 = gate/up -> silu(g)*u
// NW : waves per workgroup (they split the K slice chunk-interleaved)
// CH : 128-wide k chunks per wave (compile time, so every weight load of the wave is issued before
//      the first MFMA: a wave sees ONE exposed HBM latency however long its K range is);
//      CH == 0 is the generic runtime loop for shapes the launcher cannot unroll.
template <int MB, int EPI, int NW, int CH>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_kernel(GemmArgs a) {
    constexpr int NT = (EPI == 2) ? 2 : 1;  // weight tiles per workgroup
    __shared__ float red[NW][NT][MB][4][64];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int tile = blockIdx.x;
    const int s = blockIdx.y;
    const int KC = a.K >> 7;             // 128-wide k chunks in total
    const int cps = KC / a.S;            // chunks per K slice
    f32x4 acc[NT][MB];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[t][mb] = f32x4{0.f, 0.f, 0.f, 0.f};

    const uint4* Wt = reinterpret_cast<const uint4*>(a.W);
    const uint4* Xt = reinterpret_cast<const uint4*>(a.x);
    if constexpr (CH > 0) {
        uint4 wf[CH][NT][4];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int kl = wave + c * NW;
            const int kc = s * cps + (kl < cps ? kl : 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const uint4* wp = Wt + ((size_t)(tile * NT + t) * KC + kc) * 256 + lane;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#if defined(Q3_GEMM_ABLATE) && Q3_GEMM_ABLATE == 2
                    wf[c][t][i] = make_uint4(lane, i, t, c);
#else
                    wf[c][t][i] = wp[i * 64];
#endif
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int kl = wave + c * NW;
            if (kl < cps) {  // wave-uniform
                const int kc = s * cps + kl;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const uint4* xp = Xt + ((size_t)(kc * a.xMB + mb) * 4) * 64 + lane;
                    uint4 xf[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
#if defined(Q3_GEMM_ABLATE) && Q3_GEMM_ABLATE == 1
                        xf[i] = make_uint4(lane, i, mb, c);
                        (void)xp;
#else
                        xf[i] = xp[i * 64];
#endif
                    }
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[t][mb] = mfma16(wf[c][t][i], xf[i], acc[t][mb]);
                }
            }
        }
    } else {
        for (int kl = wave; kl < cps; kl += NW) {
            const int kc = s * cps + kl;
            uint4 wf[NT][4];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const uint4* wp = Wt + ((size_t)(tile * NT + t) * KC + kc) * 256 + lane;
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[t][i] = wp[i * 64];
            }
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const uint4* xp = Xt + ((size_t)(kc * a.xMB + mb) * 4) * 64 + lane;
                uint4 xf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) xf[i] = xp[i * 64];
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[t][mb] = mfma16(wf[t][i], xf[i], acc[t][mb]);
            }
        }
    }

#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int q = 0; q < 4; ++q) red[wave][t][mb][q][lane] = acc[t][mb][q];
    __syncthreads();

    // 256*MB outputs per tile: thread -> (mb, batch row b, feature f)
    __shared__ uint16_t ys[MB][16][16];
    for (int o = threadIdx.x; o < 256 * MB; o += NW * 64) {
        const int mb = o >> 8, rem = o & 255;
        const int b = rem >> 4, f = rem & 15;
        const int src_lane = (f >> 2) * 16 + b, q = f & 3;
        const int m = 16 * mb + b;
        float v[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) sum += red[w][t][mb][q][src_lane];
            v[t] = sum;
        }
        const int n = tile * 16 + f;
        if constexpr (EPI == 1) {
            // padded rows are written too (zeros from zero-padded x), the consumer ignores them
            a.part[((size_t)s * a.Mpad + m) * a.N + n] = v[0];
        } else if constexpr (EPI == 0) {
            float y = v[0];
            if (a.bias) y += bf2f(a.bias[n]);
            uint16_t yb = f2bf(y);
            if (a.act_silu) yb = f2bf(silu_f(bf2f(yb)));
            ys[mb][b][f] = yb;
        } else {
            float g = rbf(v[0]), u = rbf(v[1]);
            float sg = rbf(silu_f(g));
            ys[mb][b][f] = f2bf(sg * u);
        }
    }
    if constexpr (EPI != 1) {
        __syncthreads();
        // 16-byte stores: thread -> (mb, row b, 8-feature piece p)
        for (int o = threadIdx.x; o < 32 * MB; o += NW * 64) {
            const int mb = o >> 5, b = (o >> 1) & 15, p = o & 1;
            const int m = 16 * mb + b;
            if (m >= a.M) continue;
            const uint4 v = *reinterpret_cast<const uint4*>(&ys[mb][b][8 * p]);
            const int n = tile * 16 + 8 * p;
            if (EPI == 2 || a.y_tiled)
                *reinterpret_cast<uint4*>(a.y + act_tiled_offset(m, n, a.yMB)) = v;
            else
                *reinterpret_cast<uint4*>(a.y + (size_t)m * a.ldy + n) = v;
        }
    }
}

template <int MB, int EPI>
void launch_mb(const GemmArgs& a, hipStream_t st) {
    const int cps = (a.K / 128) / a.S;
    // waves: 4 when the slice has <= 4 chunks (no idle waves), else 8; CH chunks per wave (<= 3 unrolled)
    const int nw = cps <= 4 ? 4 : 8;
    const int ch = (cps + nw - 1) / nw;
    dim3 grid(a.N / 16, a.S);
#define Q3_GEMM(NWv, CHv) \
    hipLaunchKernelGGL((gemm_skinny_kernel<MB, EPI, NWv, CHv>), grid, dim3(NWv * 64), 0, st, a)
    if (nw == 4) {
        Q3_GEMM(4, 1);
    } else {
        switch (ch) {
            case 1: Q3_GEMM(8, 1); break;
            case 2: Q3_GEMM(8, 2); break;
            case 3: Q3_GEMM(8, 3); break;
            default: Q3_GEMM(8, 0); break;
        }
    }
#undef Q3_GEMM
}

template <int EPI>
void launch_epi(const GemmArgs& a, hipStream_t st) {
    const int MB = (a.Mpad + 15) / 16;
    switch (MB) {
        case 1: launch_mb<1, EPI>(a, st); break;
        case 2: launch_mb<2, EPI>(a, st); break;
        case 3: launch_mb<3, EPI>(a, st); break;
        case 4: launch_mb<4, EPI>(a, st); break;
        default: throw Error(3, "gemm_skinny: M > 64 is not supported");
    }
}

}  // namespace

void launch_gemm_skinny(const GemmArgs& a, hipStream_t st) {
    Q3_CHECK(a.K % 128 == 0 && a.N % 16 == 0, 3, "gemm_skinny: K must be a multiple of 128 and N of 16");
    Q3_CHECK(a.S >= 1 && (a.K / 128) % a.S == 0, 3, "gemm_skinny: K chunks must divide by the split");
    Q3_CHECK(a.Mpad % 16 == 0 && a.M <= a.Mpad && a.Mpad <= 64, 3, "gemm_skinny: bad M padding");
    Q3_CHECK(a.xMB * 16 >= a.Mpad, 3, "gemm_skinny: x allocation has fewer row blocks than the batch");
    switch (a.epi) {
        case 0: Q3_CHECK(a.S == 1, 3, "gemm_skinny: bf16 epilogue needs S == 1"); launch_epi<0>(a, st); break;
        case 1: launch_epi<1>(a, st); break;
        case 2: Q3_CHECK(a.S == 1, 3, "gemm_skinny: gate/up epilogue needs S == 1"); launch_epi<2>(a, st); break;
        default: throw Error(3, "gemm_skinny: unknown epilogue");
    }
}

}  // namespace q3
