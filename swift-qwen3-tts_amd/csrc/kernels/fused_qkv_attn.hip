// fused_qkv_attn.hip -- the code predictor's qkv projection and attention as ONE launch.
//
// The frame step is a chain of ~600 dependent launches and each boundary costs ~4.5 us at best; a
// producer -> consumer hand-off between workgroups of one launch costs 1.1-1.5 us (tools/xcd_handoff.hip). Here
// workgroups [0, tiles) are the qkv GEMM's column tiles (gemm_body.inc, unchanged arithmetic, result stored with
// agent-scope atomics, one flag per tile) and workgroups [tiles, tiles + n_kv * B) are the attention units
// (attn_body.inc, unchanged arithmetic): they start with the GEMM, request everything that does not depend on it (cache
// rows, norm weights, RoPE row), wait for the flags of the 32 tiles they read and fetch their row with agent-scope
// loads. Workgroups are dispatched in index order, so every producer is resident or finished before any consumer
// starts: no deadlock whatever the occupancy. Flags are zeroed by a memset at the head of the frame graph.
#include "../common.h"
#include "../kernels.h"

namespace q3 {
namespace {

#include "gemm_body.inc"
#include "attn_body.inc"

template <int MB>
__global__ __launch_bounds__(512) void qkv_attn_fused_kernel(GemmArgs g, AttnArgs at, unsigned* flags, int* err, int tiles) {
    if ((int)blockIdx.x < tiles) {
        gemm_skinny_body<MB, 0, 8, 1, true, false, 1, true>(g, blockIdx.x, 0, flags);
    } else {
        if (threadIdx.x >= 256) return;  // the attention unit is a 256-thread workgroup (16 lane groups, as attn_decode.hip)
        const int u = blockIdx.x - tiles;
        attn_decode_body<2, 256, true>(at, u % at.n_kv, u / at.n_kv, flags, err);
    }
}

}  // namespace

bool qkv_attn_fused_supported(const GemmArgs& g, const AttnArgs& at) {
    const int MBt = (g.Mpad + 15) / 16;
    return g.K == 1024 && g.norm_w && !g.Wsb && g.epi == 0 && !g.y_tiled && !g.act_silu && (MBt == 1 || MBt == 2) &&
           at.n_heads == 2 * at.n_kv && at.chunk <= 1 && at.max_pages == 1 && g.y == at.qkv && g.ldy == at.ld &&
           g.N == (at.n_heads + 2 * at.n_kv) * kHeadDim;
}

void launch_qkv_attn_fused(const GemmArgs& g, const AttnArgs& at, unsigned* flags, int* err, hipStream_t st) {
    Q3_CHECK(qkv_attn_fused_supported(g, at), 3, "qkv_attn_fused: unsupported shape");
    const int tiles = g.N / 16, units = at.n_kv * at.B;
    const int MBt = (g.Mpad + 15) / 16;
    dim3 grid(tiles + units), block(512);
    if (MBt == 1) hipLaunchKernelGGL(qkv_attn_fused_kernel<1>, grid, block, 0, st, g, at, flags, err, tiles);
    else hipLaunchKernelGGL(qkv_attn_fused_kernel<2>, grid, block, 0, st, g, at, flags, err, tiles);
}

}  // namespace q3
