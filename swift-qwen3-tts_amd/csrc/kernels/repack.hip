// repack.hip -- load-time weight re-tiling on the GPU (byte permutation only).
// [N][K] row-major bf16 (MLXNN.Linear.weight, /root/reference/.../Talker.swift:183-186) ->
// 4 KiB tiles [16 rows x 128 k] stored as [instr i=0..3][lane 0..63][8 bf16], where
// lane = (row & 15) + 16 * h and the 8 elements are k = kc*128 + 32*h + 8*i + (0..7).
// One global_load_dwordx4 per lane then fills one MFMA A fragment (gemm_decode.hip).
#include "../common.h"
#include "../model.h"

namespace q3 {
namespace {

// rpt / row_off: source rows per destination tile (16, or 8 when two matrices share every tile: gate rows in tile rows
// 0..7, up rows in 8..15) and the tile row the first of them lands on; lanes outside [row_off, row_off + rpt) leave
// their part of the tile alone.
__global__ __launch_bounds__(256) void tile_weights_kernel(const uint16_t* src, int N, int K, uint16_t* dst, int KC,
                                                           int tile_off, int tile_stride, int rpt, int row_off) {
    const int kc = blockIdx.x, j = blockIdx.y;
    const int i = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pos = (lane & 15) - row_off;
    if (pos < 0 || pos >= rpt) return;
    const int row = rpt * j + pos;
    const int k = kc * 128 + 32 * (lane >> 4) + 8 * i;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < N && k + 8 <= K) {
        v = *reinterpret_cast<const uint4*>(src + (size_t)row * K + k);
    } else if (row < N && k < K) {
        uint16_t tmp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int e = 0; e < 8 && k + e < K; ++e) tmp[e] = src[(size_t)row * K + k + e];
        __builtin_memcpy(&v, tmp, 16);
    }
    const size_t tile = (size_t)(tile_off + tile_stride * j) * KC + kc;
    *reinterpret_cast<uint4*>(dst + tile * 2048 + (size_t)i * 512 + lane * 8) = v;
}

// MLX affine int4 (group 64): packed [N][K/8] uint32 + bf16 scales/biases [N][K/64] ->
//   q tiles  [tile][kc][lane] uint4 : the 16 bytes of row (lane&15) at k = kc*128 + 32*(lane>>4) .. +31
//   sb tiles [tile][kc][lane] uint32: {scale, bias} of that lane's group (k/64)
__global__ __launch_bounds__(64) void tile_int4_kernel(const uint32_t* wq, const uint16_t* scales, const uint16_t* biases,
                                                       int N, int K, uint4* dq, uint32_t* dsb, int KC, int tile_off,
                                                       int tile_stride, int rpt, int row_off) {
    const int kc = blockIdx.x, j = blockIdx.y, lane = threadIdx.x;
    const int pos = (lane & 15) - row_off;
    if (pos < 0 || pos >= rpt) return;
    const int row = rpt * j + pos, h = lane >> 4;
    const int k = kc * 128 + 32 * h;
    uint4 v = make_uint4(0, 0, 0, 0);
    uint32_t sb = 0;
    if (row < N && k < K) {
        v = *reinterpret_cast<const uint4*>(wq + (size_t)row * (K / 8) + k / 8);
        const int g = k / 64, gpr = K / 64;
        sb = (uint32_t)scales[(size_t)row * gpr + g] | ((uint32_t)biases[(size_t)row * gpr + g] << 16);
    }
    const size_t blk = (size_t)(tile_off + tile_stride * j) * KC + kc;
    dq[blk * 64 + lane] = v;
    dsb[blk * 64 + lane] = sb;
}

}  // namespace

void launch_tile_int4(const uint32_t* wq, const uint16_t* scales, const uint16_t* biases, int N, int K, void* dq,
                      uint32_t* dsb, int KC, int tile_off, int tile_stride, hipStream_t st, int rpt, int row_off) {
    Q3_CHECK(K % 64 == 0, 6, "int4 weights need an inner size that is a multiple of the group size 64");
    Q3_CHECK((rpt == 16 && row_off == 0) || (rpt == 8 && (row_off == 0 || row_off == 8)), 7, "tile_int4: bad row mapping");
    dim3 grid(KC, (N + rpt - 1) / rpt);
    hipLaunchKernelGGL(tile_int4_kernel, grid, dim3(64), 0, st, wq, scales, biases, N, K, reinterpret_cast<uint4*>(dq), dsb, KC,
                       tile_off, tile_stride, rpt, row_off);
}

void launch_tile_weights(const uint16_t* src, int N, int K, uint16_t* dst, int KC, int tile_off, int tile_stride,
                         hipStream_t st, int rpt, int row_off) {
    Q3_CHECK(K % 8 == 0, 6, "linear weights need an inner size that is a multiple of 8");
    Q3_CHECK((rpt == 16 && row_off == 0) || (rpt == 8 && (row_off == 0 || row_off == 8)), 7, "tile_weights: bad row mapping");
    dim3 grid(KC, (N + rpt - 1) / rpt);
    hipLaunchKernelGGL(tile_weights_kernel, grid, dim3(256), 0, st, src, N, K, dst, KC, tile_off, tile_stride, rpt, row_off);
}

}  // namespace q3
