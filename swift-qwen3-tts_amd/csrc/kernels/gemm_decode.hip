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
#include "../common.h"
#include "../kernels.h"
#include "row_jobs.h"

namespace q3 {

namespace {

#include "gemm_body.inc"

template <int MB, int EPI, int NW, int CH, bool NORM, bool QUANT, int NP = 1, bool NTW = false>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_kernel(GemmArgs a) {
    __builtin_amdgcn_s_setprio(3);  // the AR chain's waves go first where they share a CU with a codec decode (engine.cc)
    gemm_skinny_body<MB, EPI, NW, CH, NORM, QUANT, NP, NTW>(a, blockIdx.x, blockIdx.y * MB);
}

// gemm_skinny_kernel<MB, 0, 8, CH, true, QUANT> plus riders: workgroups tiles .. tiles + n.M - 1 of grid.x normalise one row each
struct GemmSideArgs {
    GemmArgs g;
    NormRowsArgs n;
    int tiles;
};
template <int MB, int CH, bool QUANT>
__global__ __launch_bounds__(512) void gemm_skinny_side_kernel(GemmSideArgs s) {
    __builtin_amdgcn_s_setprio(3);
    if (int(blockIdx.x) >= s.tiles) {
        __shared__ float side_sh[4];
        __shared__ float side_parts[8];
        if (blockIdx.y == 0) norm_row_job(s.n, int(blockIdx.x) - s.tiles, threadIdx.x, side_sh, side_parts);
        return;
    }
    gemm_skinny_body<MB, 0, 8, CH, true, QUANT, 1, false>(s.g, blockIdx.x, blockIdx.y * MB);
}

// a with the launch's own geometry written into its touch descriptor (prefetch.h: gx, wg_per_xcd)
inline GemmArgs with_geometry(const GemmArgs& a, const dim3& grid) {
    GemmArgs b = a;
    if (!b.pf.base) {  // a caller outside the frame step's plan: an empty range on a valid address (the touch loads are unconditional)
        b.pf = PfArgs{};
        b.pf.base = reinterpret_cast<const uint8_t*>(a.W);
        b.pf.span = 128; b.pf.lines = 1; b.pf.inv_lines = 1.0f;
    }
    b.pf.gx = grid.x;
    b.pf.wg_per_xcd = (grid.x * grid.y + 7) / 8;
    return b;
}

template <int MB, int EPI, bool NORM, bool QUANT, bool NTW>
void launch_mb(const GemmArgs& a0, const SkinnyGeom& g, hipStream_t st) {
    const dim3 grid(g.gx, g.split);
    const GemmArgs a = with_geometry(a0, grid);
#define Q3_GEMM(NWv, CHv) \
    hipLaunchKernelGGL((gemm_skinny_kernel<MB, EPI, NWv, CHv, NORM, QUANT, 1, NTW>), grid, dim3(NWv * 64), 0, st, a)
    if constexpr (MB == 4 && !NORM && !NTW) {
        // prefill chunks: several tiles per workgroup through the chunk-streaming form (CH = 0): they quarter the x traffic
        // (every workgroup reads all K of its 64 rows), but a narrow layer (o_proj / down_proj: 128 tiles) is then 128
        // workgroups on 256 CUs: two tiles each there (prefill 17.0 -> 14.2 ms at 1.7B / batch 32; one tile each: 15.5)
        if (g.ch == 0 && g.np == 2) { hipLaunchKernelGGL((gemm_skinny_kernel<4, EPI, 8, 0, false, QUANT, 2, false>), grid, dim3(512), 0, st, a); return; }
        if (g.ch == 0 && g.np == 4) { hipLaunchKernelGGL((gemm_skinny_kernel<4, EPI, 8, 0, false, QUANT, 4, false>), grid, dim3(512), 0, st, a); return; }
    }
    if constexpr (EPI == 2 && MB <= 2) {
        // gate/up tiles are self-contained, so a workgroup takes as many as it needs for the launch to be ONE round of at
        // most 256 workgroups: 768 tiles (6144 columns) -> 3 each, 384 (3072) -> 2 each
        if (g.np > 1) {
#define Q3_GEMM_NP(CHv, NPv) hipLaunchKernelGGL((gemm_skinny_kernel<MB, 2, 8, CHv, NORM, QUANT, NPv, NTW>), grid, dim3(512), 0, st, a)
            if (g.ch == 1) { if (g.np == 2) Q3_GEMM_NP(1, 2); else Q3_GEMM_NP(1, 3); }
            else { if (g.np == 2) Q3_GEMM_NP(2, 2); else Q3_GEMM_NP(2, 3); }
#undef Q3_GEMM_NP
            return;
        }
    }
    if (g.nw == 4) {
        Q3_GEMM(4, 1);
    } else {
        switch (g.ch) {
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
    const SkinnyGeom g = skinny_geometry(a);
    // non-temporal weight loads: only the one- and two-row-block forms the talker's decode step uses are instantiated
    if (g.ntw) {
        if (g.mbw == 1) launch_mb<1, EPI, NORM, QUANT, true>(a, g, st);
        else launch_mb<2, EPI, NORM, QUANT, true>(a, g, st);
        return;
    }
    switch (g.mbw) {
        case 1: launch_mb<1, EPI, NORM, QUANT, false>(a, g, st); break;
        case 2: launch_mb<2, EPI, NORM, QUANT, false>(a, g, st); break;
        case 3: launch_mb<3, EPI, NORM, QUANT, false>(a, g, st); break;
        case 4: launch_mb<4, EPI, NORM, QUANT, false>(a, g, st); break;
        default: throw Error(3, "gemm_skinny: more than 4 row blocks per workgroup");
    }
}

template <int EPI, bool NORM>
void launch_epi(const GemmArgs& a, hipStream_t st) {
    if (a.Wsb) launch_q<EPI, NORM, true>(a, st);
    else launch_q<EPI, NORM, false>(a, st);
}

}  // namespace

// The ONE place that decides a skinny GEMM's launch (kernels.h SkinnyGeom).
SkinnyGeom skinny_geometry(const GemmArgs& a) {
    SkinnyGeom g{};
    if (gemm_tall_takes(a)) {
        g.tall = true;
        return g;
    }
    const DebugEnv& env = debug_env();
    const bool norm = a.norm_w != nullptr;
    const int MBt = (a.Mpad + 15) / 16;
    // Narrow layers (o_proj, down_proj: N / 16 <= 128 column tiles) leave CUs idle with one workgroup per tile, and a
    // workgroup that carries two row blocks cannot keep every x fragment of a long K in registers. Their row blocks go
    // to separate workgroups instead (grid.y); tile x of both lands on the same XCD (128 = 0 mod 8), so the second
    // read of the weight tile is an L2 hit. Per-row arithmetic does not depend on the grouping: results are unchanged.
    const bool no_split = env.gemm_no_row_split;
    int tiles = a.N / 16;  // workgroups along x
    if (a.epi == 2) {      // eight columns per tile, up to three tiles per workgroup
        const int t8 = a.N / 8;
        tiles = t8 > 512 ? (t8 + 2) / 3 : (t8 > 256 ? (t8 + 1) / 2 : t8);
    }
    int split = 1;
    if (!no_split) {
        if (MBt % 4 == 0 && tiles * 4 <= 256) split = 4;
        else if (MBt % 2 == 0 && tiles * 2 <= 256) split = 2;
        // More than 64 rows (prefill chunks, the predictor's two-position step): at most 4 row blocks per workgroup -- 2 with
        // the norm prologue, whose VALU work on the x fragments (critical path, per workgroup) grows with the row blocks.
        const int cap = norm ? 2 : 4;
        while (MBt / split > cap || MBt % split != 0) ++split;
    }
    while (MBt / split > 4 || MBt % split != 0) ++split;
    g.split = split;
    g.mbw = MBt / split;
    g.ntw = a.nt_weights && g.mbw <= 2;
    const int KC = a.K / 128;
    g.nw = KC <= 4 ? 4 : 8;  // no idle waves on short K
    g.ch = (KC + g.nw - 1) / g.nw;
    g.np = 1;
    const int t = a.epi == 2 ? a.N / 8 : a.N / 16;  // EPI 2: eight columns per (gate | up) tile
    g.gx = t;
    bool shaped = false;
    if (g.mbw == 4 && !norm && !g.ntw && !env.gemm_one_pair && g.nw == 8 && t >= 64) {  // prefill chunks (launch_mb)
        g.np = (((t + 3) / 4) * split <= 128) ? 2 : 4;
        g.ch = 0;
        g.gx = (t + g.np - 1) / g.np;
        shaped = true;
    }
    if (!shaped && a.epi == 2 && g.mbw <= 2) {
        const int np = env.gemm_one_pair ? 1 : (t > 512 ? 3 : (t > 256 ? 2 : 1));
        if (g.nw == 8 && (g.ch == 1 || g.ch == 2) && np > 1) {
            g.np = np;
            g.gx = (t + np - 1) / np;
            shaped = true;
        }
    }
    if (!shaped) {
        if (g.nw == 4) g.ch = 1;
        else if (g.ch != 1 && g.ch != 2 && g.ch != 3 && g.ch != 6) g.ch = 0;
    }
    g.touches = gemm_touches(g.mbw, g.ch);
    return g;
}

bool gemm_norm_rows_rides(const GemmArgs& a, const NormRowsArgs& n) {
    if (a.epi != 0 || !a.norm_w || a.nt_weights || n.M <= 0 || n.H > 4096) return false;
    if (a.K % 128 != 0 || a.N % 16 != 0) return false;
    const SkinnyGeom g = skinny_geometry(a);
    return !(g.tall || g.nw != 8 || g.np != 1 || (g.ch != 1 && g.ch != 2) || g.mbw < 1 || g.mbw > 2 || g.ntw);
}

bool launch_gemm_skinny_with_norm_rows(const GemmArgs& a, const NormRowsArgs& n, hipStream_t st) {
    // exactly the launch launch_gemm_skinny would make for these arguments, plus n.M workgroups along x -- or nothing
    if (!gemm_norm_rows_rides(a, n)) return false;
    const SkinnyGeom g = skinny_geometry(a);
    Q3_CHECK(a.ss_in && a.ss_count >= 1 && a.ss_ld >= a.Mpad && a.xMB * 16 >= a.Mpad, 3, "gemm_skinny: norm prologue needs sums of squares");
    const int tiles = g.gx;
    const dim3 grid(tiles + n.M, g.split), block(512);
    GemmSideArgs s{with_geometry(a, dim3(tiles, g.split)), n, tiles};
    s.g.pf.gx = grid.x;  // the riders sit behind the tiles along x; they touch nothing themselves
    const bool q = a.Wsb != nullptr;
#define Q3_SIDE(MBv, CHv)                                                                              \
    do {                                                                                               \
        if (q) hipLaunchKernelGGL((gemm_skinny_side_kernel<MBv, CHv, true>), grid, block, 0, st, s);   \
        else hipLaunchKernelGGL((gemm_skinny_side_kernel<MBv, CHv, false>), grid, block, 0, st, s);    \
    } while (0)
    if (g.mbw == 1) { if (g.ch == 1) Q3_SIDE(1, 1); else Q3_SIDE(1, 2); }
    else { if (g.ch == 1) Q3_SIDE(2, 1); else Q3_SIDE(2, 2); }
#undef Q3_SIDE
    return true;
}

void launch_gemm_skinny(const GemmArgs& a, hipStream_t st) {
    Q3_CHECK(a.K % 128 == 0 && a.N % (a.epi == 2 ? 8 : 16) == 0, 3, "gemm_skinny: K must be a multiple of 128 and N of 16");
    Q3_CHECK(a.Mpad % 16 == 0 && a.M <= a.Mpad && a.Mpad <= 1024, 3, "gemm_skinny: bad M padding");
    Q3_CHECK(a.xMB * 16 >= a.Mpad, 3, "gemm_skinny: x allocation has fewer row blocks than the batch");
    const bool norm = a.norm_w != nullptr;
    if (a.epi == 3) Q3_CHECK(a.yMB * 16 >= a.Mpad, 3, "gemm_skinny: h allocation has fewer row blocks than the batch");
    if (launch_gemm_tall(a, st)) return;  // prefill chunks: same results from LDS-shared tiles (gemm_prefill.hip)
    if (norm) Q3_CHECK(a.ss_in && a.ss_count >= 1 && a.ss_ld >= a.Mpad, 3, "gemm_skinny: norm prologue needs sums of squares");
    switch (a.epi) {
        case 0: norm ? launch_epi<0, true>(a, st) : launch_epi<0, false>(a, st); break;
        case 2: norm ? launch_epi<2, true>(a, st) : launch_epi<2, false>(a, st); break;
        case 3: norm ? launch_epi<3, true>(a, st) : launch_epi<3, false>(a, st); break;
        default: throw Error(3, "gemm_skinny: unknown epilogue");
    }
}

}  // namespace q3
