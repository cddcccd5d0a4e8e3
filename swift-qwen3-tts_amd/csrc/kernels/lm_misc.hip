// lm_misc.hip -- small row kernels of the AR loop: residual+RMSNorm, embedding gathers, the
// next-input embedding sum and the per-row loop bookkeeping. All are HBM/L2 trivial (<= 64 rows of
// <= 2048 bf16); they exist to keep the frame step free of host round trips (the reference pays
// >= 17 eval()/.item() syncs per frame: SURVEY.md section 3.2).
#include "../common.h"
#include "../kernels.h"

namespace q3 {
namespace {

// Residual add + RMSNorm. One workgroup per row.
//   h  <- bf16(h + bf16(sum_s part[s]))          residual adds Talker.swift:461,466
//   xn <- bf16( bf16(h * rstd) * w )             MLXNN.RMSNorm, Talker.swift:447-448,520
// The split-K partial slabs of the preceding o_proj / down_proj GEMM are summed here in fixed
// order, so the projection's rounding point (bf16 of the full fp32 sum) is the reference's.
__global__ __launch_bounds__(256) void resid_norm_kernel(ResidNormArgs a) {
    __shared__ float wsum[4];
    constexpr int kTrips = 2;  // H <= 4096
    const int m = blockIdx.x;
    const int tid = threadIdx.x;
    uint16_t* hrow = a.h + (size_t)m * a.ldh;
    float v[kTrips][8];
    uint4 wv[kTrips];
    float ss = 0.f;
    // each thread owns elements tid*8 .. tid*8+7 (+2048 per trip); everything is loaded up front
#pragma unroll
    for (int tr = 0; tr < kTrips; ++tr) {
        const int i0 = tid * 8 + tr * 2048;
        if (i0 < a.H) {
            const uint4 hv = *reinterpret_cast<const uint4*>(hrow + i0);
            if (a.w) wv[tr] = *reinterpret_cast<const uint4*>(a.w + i0);
            const uint32_t hw[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[tr][2 * j] = lo_bf(hw[j]);
                v[tr][2 * j + 1] = hi_bf(hw[j]);
            }
            if (a.part) {
                float y[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                for (int s = 0; s < a.S; ++s) {
                    const float4* pp = reinterpret_cast<const float4*>(a.part + ((size_t)s * a.Mpad + m) * a.H + i0);
                    const float4 p0 = pp[0], p1 = pp[1];
                    y[0] += p0.x; y[1] += p0.y; y[2] += p0.z; y[3] += p0.w;
                    y[4] += p1.x; y[5] += p1.y; y[6] += p1.z; y[7] += p1.w;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) v[tr][j] = rbf(v[tr][j] + rbf(y[j]));
                uint4 o;
                o.x = pack_bf(v[tr][0], v[tr][1]); o.y = pack_bf(v[tr][2], v[tr][3]);
                o.z = pack_bf(v[tr][4], v[tr][5]); o.w = pack_bf(v[tr][6], v[tr][7]);
                *reinterpret_cast<uint4*>(hrow + i0) = o;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += v[tr][j] * v[tr][j];
        }
    }
    if (!a.w) return;
    ss = wave_sum(ss);
    if ((tid & 63) == 0) wsum[tid >> 6] = ss;
    __syncthreads();
    const float tot = ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
    const float rstd = 1.0f / sqrtf(tot / (float)a.H + a.eps);
#pragma unroll
    for (int tr = 0; tr < kTrips; ++tr) {
        const int i0 = tid * 8 + tr * 2048;
        if (i0 < a.H) {
            const uint32_t ww[4] = {wv[tr].x, wv[tr].y, wv[tr].z, wv[tr].w};
            uint32_t ow[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float n0 = rbf(v[tr][2 * j] * rstd), n1 = rbf(v[tr][2 * j + 1] * rstd);
                ow[j] = pack_bf(n0 * lo_bf(ww[j]), n1 * hi_bf(ww[j]));
            }
            *reinterpret_cast<uint4*>(a.xn + act_tiled_offset(m, i0, a.xnMB)) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
        }
    }
}

__global__ void gather_rows_kernel(const uint16_t* table, int ld, const int32_t* ids, const int32_t* token_map,
                                   int dim, uint16_t* out, int ldo, int out_MB) {
    const int r = blockIdx.x;
    int id = ids[r];
    if (token_map) id = token_map[id];  // embedText, Talker.swift:627-633
    const uint4* src = reinterpret_cast<const uint4*>(table + (size_t)id * ld);
    for (int i = threadIdx.x; i < dim / 8; i += blockDim.x) {
        if (out_MB > 0) *reinterpret_cast<uint4*>(out + act_tiled_offset(r, 8 * i, out_MB)) = src[i];
        else reinterpret_cast<uint4*>(out + (size_t)r * ldo)[i] = src[i];
    }
}

__global__ void tile_rows_kernel(const uint16_t* src, int lds, uint16_t* dst, int dstMB, int dim) {
    const int r = blockIdx.x;
    const uint4* s = reinterpret_cast<const uint4*>(src + (size_t)r * lds);
    for (int i = threadIdx.x; i < dim / 8; i += blockDim.x)
        *reinterpret_cast<uint4*>(dst + act_tiled_offset(r, 8 * i, dstMB)) = s[i];
}
__global__ void untile_rows_kernel(const uint16_t* src, int srcMB, uint16_t* dst, int ldd, int dim) {
    const int r = blockIdx.x;
    uint4* d = reinterpret_cast<uint4*>(dst + (size_t)r * ldd);
    for (int i = threadIdx.x; i < dim / 8; i += blockDim.x)
        d[i] = *reinterpret_cast<const uint4*>(src + act_tiled_offset(r, 8 * i, srcMB));
}

__global__ void add_rows_kernel(const uint16_t* a, int lda, const uint16_t* b, int ldb, int dim, uint16_t* out,
                                int ldo) {
    const int r = blockIdx.x;
    const uint16_t* ar = a + (size_t)r * lda;
    const uint16_t* br = b + (size_t)r * ldb;
    uint16_t* orow = out + (size_t)r * ldo;
    for (int i = threadIdx.x; i < dim; i += blockDim.x) orow[i] = f2bf(bf2f(ar[i]) + bf2f(br[i]));
}

__global__ void compose_rows_kernel(const uint16_t* proj, int ldp, const uint16_t* table, int ldt, const int32_t* a,
                                    const int32_t* b, const int32_t* dst_row, uint16_t* dst, int ldd, int dim) {
    const int r = blockIdx.x;
    const uint16_t* pa = proj + (size_t)a[r] * ldp;
    const int bi = b[r];
    uint16_t* o = dst + (size_t)dst_row[r] * ldd;
    if (bi < 0) {
        for (int i = threadIdx.x; i < dim; i += blockDim.x) o[i] = pa[i];
    } else {
        const uint16_t* pb = table + (size_t)bi * ldt;
        for (int i = threadIdx.x; i < dim; i += blockDim.x) o[i] = f2bf(bf2f(pa[i]) + bf2f(pb[i]));
    }
}

__global__ void copy_rows_kernel(const uint16_t* src, int lds, uint16_t* dst, int ldd, int dim) {
    const int r = blockIdx.x;
    const uint4* s = reinterpret_cast<const uint4*>(src + (size_t)r * lds);
    uint4* d = reinterpret_cast<uint4*>(dst + (size_t)r * ldd);
    for (int i = threadIdx.x; i < dim / 8; i += blockDim.x) d[i] = s[i];
}

// Right-aligned position-by-position prefill: row b starts at step Pmax - n_prompt[b], so every
// row finishes its prompt at the same step and enters the frame loop together.
__global__ void prefill_load_kernel(PrefillLoadArgs a) {
    const int b = blockIdx.x;
    const int idx = a.step - (a.Pmax - a.n_prompt[b]);
    const bool on = idx >= 0;
    if (threadIdx.x == 0) a.active[b] = on ? 1 : 0;
    const uint4* src = reinterpret_cast<const uint4*>(a.prompt + ((size_t)b * a.Pmax + (on ? idx : 0)) * a.H);
    uint4* dst = reinterpret_cast<uint4*>(a.h + (size_t)b * a.ldh);
    for (int i = threadIdx.x; i < a.H / 8; i += blockDim.x) dst[i] = src[i];
}

__global__ void advance_len_kernel(int32_t* kv_len, const uint8_t* active, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B && (!active || active[b])) kv_len[b] += 1;
}

// End of a frame (Qwen3.swift:914-935): next talker input = text embed (next trailing text row or
// tts_pad) + sum of the 16 codebook embeddings, every add rounded to bf16 in the reference's
// left-to-right order; then the loop bookkeeping that the Swift loop keeps on the host.
__global__ __launch_bounds__(256) void frame_end_kernel(FrameEndArgs a) {
    const int b = blockIdx.x;
    if (a.finished[b]) {
        if (threadIdx.x == 0) a.cp_len[b] = 0;
        return;
    }
    const int32_t* cc = a.cur_codes + (size_t)b * 16;
    const int ti = a.trailing_idx[b];
    const bool has_text = ti < a.n_trailing[b];
    const uint16_t* text = has_text ? a.trailing + ((size_t)b * a.Tmax + ti) * a.H : a.tts_pad;
    const uint16_t* rows[16];
    rows[0] = a.codec_emb + (size_t)cc[0] * a.H;
#pragma unroll
    for (int g = 1; g < 16; ++g) rows[g] = (g < a.groups) ? a.cp_emb[g - 1] + (size_t)cc[g] * a.H : rows[0];
    for (int i = threadIdx.x; i < a.H; i += blockDim.x) {
        float e[16];
#pragma unroll
        for (int g = 0; g < 16; ++g) e[g] = bf2f(rows[g][i]);  // 16 independent loads in flight
        float ce = e[0];
#pragma unroll
        for (int g = 1; g < 16; ++g)
            if (g < a.groups) ce = rbf(ce + e[g]);
        a.h[(size_t)b * a.ldh + i] = f2bf(bf2f(text[i]) + ce);
    }
    __syncthreads();  // every thread has read trailing_idx before it moves
    if (threadIdx.x == 0) {
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

}  // namespace

void launch_resid_norm(const ResidNormArgs& a, hipStream_t st) {
    Q3_CHECK(a.H % 128 == 0 && a.H <= 4096 && a.ldh % 8 == 0, 3, "resid_norm: H must be a multiple of 128, at most 4096");
    hipLaunchKernelGGL(resid_norm_kernel, dim3(a.M), dim3(256), 0, st, a);
}
void launch_gather_rows(const uint16_t* table, int ld, const int32_t* ids, const int32_t* token_map, int n,
                        int dim, uint16_t* out, int ldo, int out_MB, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(n), dim3(256), 0, st, table, ld, ids, token_map, dim, out, ldo, out_MB);
}
void launch_tile_rows(const uint16_t* src, int lds, uint16_t* dst, int dstMB, int rows, int dim, hipStream_t st) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(tile_rows_kernel, dim3(rows), dim3(256), 0, st, src, lds, dst, dstMB, dim);
}
void launch_untile_rows(const uint16_t* src, int srcMB, uint16_t* dst, int ldd, int rows, int dim, hipStream_t st) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(untile_rows_kernel, dim3(rows), dim3(256), 0, st, src, srcMB, dst, ldd, dim);
}
void launch_add_rows(const uint16_t* a, int lda, const uint16_t* b, int ldb, int rows, int dim, uint16_t* out,
                     int ldo, hipStream_t st) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(add_rows_kernel, dim3(rows), dim3(256), 0, st, a, lda, b, ldb, dim, out, ldo);
}
void launch_compose_rows(const uint16_t* proj, int ldp, const uint16_t* table, int ldt, const int32_t* a,
                         const int32_t* b, const int32_t* dst_row, uint16_t* dst, int ldd, int n, int dim,
                         hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(compose_rows_kernel, dim3(n), dim3(256), 0, st, proj, ldp, table, ldt, a, b, dst_row, dst, ldd, dim);
}
void launch_copy_rows(const uint16_t* src, int lds, uint16_t* dst, int ldd, int rows, int dim, hipStream_t st) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(copy_rows_kernel, dim3(rows), dim3(256), 0, st, src, lds, dst, ldd, dim);
}
void launch_prefill_load(const PrefillLoadArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(prefill_load_kernel, dim3(a.B), dim3(256), 0, st, a);
}
void launch_advance_len(int32_t* kv_len, const uint8_t* active, int B, hipStream_t st) {
    hipLaunchKernelGGL(advance_len_kernel, dim3(1), dim3(64), 0, st, kv_len, active, B);
}
void launch_frame_end(const FrameEndArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(frame_end_kernel, dim3(a.B), dim3(256), 0, st, a);
}

}  // namespace q3
