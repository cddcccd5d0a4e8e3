// lm_misc.hip -- small row kernels of the AR loop: residual+RMSNorm, embedding gathers, the
// next-input embedding sum and the per-row loop bookkeeping. All are HBM/L2 trivial (<= 64 rows of
// <= 2048 bf16); they exist to keep the frame step free of host round trips (the reference pays
// >= 17 eval()/.item() syncs per frame: SURVEY.md section 3.2).
#include "../common.h"
#include "../kernels.h"
#include "row_jobs.h"

namespace q3 {
namespace {

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((sh[0] + sh[1]) + sh[2]) + sh[3];
}

// RMSNorm of whole rows, fragment-major in and out. One workgroup per row (the arithmetic lives in row_jobs.h: the frame
// step carries the same job inside the codec_head launch).
__global__ __launch_bounds__(256) void norm_rows_kernel(NormRowsArgs a) {
    __shared__ float sh[4];
    __shared__ float parts[8];
    norm_row_job(a, blockIdx.x, threadIdx.x, sh, parts);
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

__global__ void compose_rows_kernel(const uint16_t* proj, int ldp, const uint16_t* table, int ldt, const uint16_t* extra,
                                    int lde, const int32_t* a, const int32_t* b, const int32_t* dst_row, uint16_t* dst,
                                    int ldd, int dim) {
    const int r = blockIdx.x;
    const uint16_t* pa = proj + (size_t)a[r] * ldp;
    const int bi = b[r];
    uint16_t* o = dst + (size_t)dst_row[r] * ldd;
    if (bi == -1) {
        for (int i = threadIdx.x; i < dim; i += blockDim.x) o[i] = pa[i];
    } else {
        const uint16_t* pb = bi >= 0 ? table + (size_t)bi * ldt : extra + (size_t)(-2 - bi) * lde;
        for (int i = threadIdx.x; i < dim; i += blockDim.x) o[i] = f2bf(bf2f(pa[i]) + bf2f(pb[i]));
    }
}

// Voice-clone prompt rows (Qwen3.swift:485-491): out[t] = codec_emb[c0[t]] + cp_emb[0][c1[t]] + ... left to right,
// each add rounded to bf16. codes [groups][T].
__global__ void ref_embed_rows_kernel(const int32_t* codes, int T, int groups, const uint16_t* codec_emb,
                                      const uint16_t* const* cp_emb, int H, uint16_t* out, int ldo) {
    const int t = blockIdx.x;
    for (int i = threadIdx.x; i < H; i += blockDim.x) {
        float v = bf2f(codec_emb[(size_t)codes[t] * H + i]);
        for (int g = 1; g < groups; ++g) v = rbf(v + bf2f(cp_emb[g - 1][(size_t)codes[(size_t)g * T + t] * H + i]));
        out[(size_t)t * ldo + i] = f2bf(v);
    }
}

// decoder input of a voice-clone row (Qwen3.swift:1176-1180): reference frames ([16][Tref]) then generated ([F][16])
__global__ void build_decode_codes_kernel(const int32_t* ref, int Tref, const int32_t* gen, int F, int32_t* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (Tref + F) * 16) return;
    const int f = i >> 4, g = i & 15;
    out[i] = f < Tref ? ref[(size_t)g * Tref + f] : gen[(size_t)(f - Tref) * 16 + g];
}

__global__ void f32_to_bf16_kernel(const float* x, uint16_t* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = f2bf(x[i]);
}

__global__ void ss_to_table_kernel(const float* ss, int ss_ld, float* dst, int nss, int rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows * nss) dst[i] = ss[(size_t)(i % nss) * ss_ld + i / nss];
}

__global__ void copy_rows_kernel(const uint16_t* src, int lds, uint16_t* dst, int ldd, int dim) {
    const int r = blockIdx.x;
    const uint4* s = reinterpret_cast<const uint4*>(src + (size_t)r * lds);
    uint4* d = reinterpret_cast<uint4*>(dst + (size_t)r * ldd);
    for (int i = threadIdx.x; i < dim / 8; i += blockDim.x) d[i] = s[i];
}

// Right-aligned position-by-position prefill: row b starts at step Pmax - n_prompt[b], so every
// row finishes its prompt at the same step and enters the frame loop together.
__global__ __launch_bounds__(256) void prefill_load_kernel(PrefillLoadArgs a) {
    __shared__ float sh[4];
    const int b = blockIdx.x;
    const int idx = a.step - (a.Pmax - a.n_prompt[b]);
    const bool on = idx >= 0;
    if (threadIdx.x == 0) a.active[b] = on ? 1 : 0;
    const uint4* src = reinterpret_cast<const uint4*>(a.prompt + ((size_t)b * a.Pmax + (on ? idx : 0)) * a.H);
    float ss = 0.f;
    for (int i = threadIdx.x; i < a.H / 8; i += 256) {
        const uint4 v = src[i];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) ss += lo_bf(w[j]) * lo_bf(w[j]) + hi_bf(w[j]) * hi_bf(w[j]);
        *reinterpret_cast<uint4*>(a.h + act_tiled_offset(b, 8 * i, a.hMB)) = v;
    }
    const float tot = block_sum_256(ss, sh);
    if (threadIdx.x == 0) a.ss_out[b] = tot;
}

// Chunked prefill: element p of row b is prompt position r = r_base + n_prompt[b] + p (right-aligned so that the last
// chunk ends at the row's second-to-last prompt position); r < 0 is padding. Row m = p * B + b of the residual stream.
__global__ __launch_bounds__(256) void prefill_chunk_load_kernel(PrefillLoadArgs a) {
    __shared__ float sh[4];
    const int b = blockIdx.x, p = blockIdx.y;
    const int r = a.step + a.n_prompt[b] + p;  // `step` carries r_base
    const bool on = r >= 0;
    const int m = p * a.B + b;
    const uint4* src = reinterpret_cast<const uint4*>(a.prompt + ((size_t)b * a.Pmax + (on ? r : 0)) * a.H);
    float ss = 0.f;
    for (int i = threadIdx.x; i < a.H / 8; i += 256) {
        const uint4 v = on ? src[i] : make_uint4(0, 0, 0, 0);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) ss += lo_bf(w[j]) * lo_bf(w[j]) + hi_bf(w[j]) * hi_bf(w[j]);
        *reinterpret_cast<uint4*>(a.h + act_tiled_offset(m, 8 * i, a.hMB)) = v;
    }
    const float tot = block_sum_256(ss, sh);
    if (threadIdx.x == 0) a.ss_out[m] = tot;
}
__global__ void advance_len_chunk_kernel(int32_t* kv_len, const int32_t* n_prompt, int r_base, int C, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int r0 = r_base + n_prompt[b];
    const int p0 = r0 < 0 ? (-r0 < C ? -r0 : C) : 0;
    kv_len[b] += C - p0;
}

// Diagnostics (Q3TTS_FRAME_STAMPS=1): slot[k] accumulates the time between this stamp and the previous one of the same
// frame step, so that the phases of a REPLAYED graph can be timed (a tracing profiler perturbs a chain of 5 us launches).
__global__ void stamp_kernel(unsigned long long* acc, unsigned long long* last, int k) {
    const unsigned long long t = wall_clock64();  // 100 MHz constant clock
    if (k > 0) acc[k] += t - *last;
    else acc[0] += 1;  // frame steps seen
    *last = t;
}

__global__ void advance_len_kernel(int32_t* kv_len, const uint8_t* active, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B && (!active || active[b])) kv_len[b] += 1;
}

// End of a frame as a launch of its own (row_jobs.h frame_end_job; the frame step runs it in the last sampler's launch)
__global__ __launch_bounds__(256) void frame_end_kernel(FrameEndArgs a) {
    __shared__ float sh[4];
    const int b = blockIdx.x;
    if (a.finished[b]) {
        if (threadIdx.x == 0) a.cp_len[b] = 0;
        return;
    }
    frame_end_job(a, b, threadIdx.x, -1, sh);
}

}  // namespace

void launch_norm_rows(const NormRowsArgs& a, hipStream_t st) {
    Q3_CHECK(a.H % 128 == 0 && a.H <= 4096, 3, "norm_rows: H must be a multiple of 128, at most 4096");
    hipLaunchKernelGGL(norm_rows_kernel, dim3(a.M), dim3(256), 0, st, a);
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
void launch_compose_rows(const uint16_t* proj, int ldp, const uint16_t* table, int ldt, const uint16_t* extra, int lde,
                         const int32_t* a, const int32_t* b, const int32_t* dst_row, uint16_t* dst, int ldd, int n, int dim,
                         hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(compose_rows_kernel, dim3(n), dim3(256), 0, st, proj, ldp, table, ldt, extra, lde, a, b, dst_row, dst,
                       ldd, dim);
}
void launch_ref_embed_rows(const int32_t* codes, int T, int groups, const uint16_t* codec_emb, const uint16_t* const* cp_emb,
                           int H, uint16_t* out, int ldo, hipStream_t st) {
    if (T <= 0) return;
    hipLaunchKernelGGL(ref_embed_rows_kernel, dim3(T), dim3(256), 0, st, codes, T, groups, codec_emb, cp_emb, H, out, ldo);
}
void launch_build_decode_codes(const int32_t* ref, int Tref, const int32_t* gen, int F, int32_t* out, hipStream_t st) {
    const int n = (Tref + F) * 16;
    if (n <= 0) return;
    hipLaunchKernelGGL(build_decode_codes_kernel, dim3((n + 255) / 256), dim3(256), 0, st, ref, Tref, gen, F, out);
}
void launch_f32_to_bf16(const float* x, uint16_t* out, int n, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((n + 255) / 256), dim3(256), 0, st, x, out, n);
}
void launch_ss_to_table(const float* ss, int ss_ld, float* dst, int nss, int rows, hipStream_t st) {
    hipLaunchKernelGGL(ss_to_table_kernel, dim3((rows * nss + 255) / 256), dim3(256), 0, st, ss, ss_ld, dst, nss, rows);
}
void launch_copy_rows(const uint16_t* src, int lds, uint16_t* dst, int ldd, int rows, int dim, hipStream_t st) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(copy_rows_kernel, dim3(rows), dim3(256), 0, st, src, lds, dst, ldd, dim);
}
void launch_prefill_load(const PrefillLoadArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(prefill_load_kernel, dim3(a.B), dim3(256), 0, st, a);
}
void launch_prefill_chunk_load(const PrefillLoadArgs& a, int C, hipStream_t st) {
    hipLaunchKernelGGL(prefill_chunk_load_kernel, dim3(a.B, C), dim3(256), 0, st, a);
}
void launch_advance_len_chunk(int32_t* kv_len, const int32_t* n_prompt, int r_base, int C, int B, hipStream_t st) {
    hipLaunchKernelGGL(advance_len_chunk_kernel, dim3(1), dim3(64), 0, st, kv_len, n_prompt, r_base, C, B);
}
void launch_stamp(unsigned long long* acc, unsigned long long* last, int k, hipStream_t st) {
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, st, acc, last, k);
}
void launch_advance_len(int32_t* kv_len, const uint8_t* active, int B, hipStream_t st) {
    hipLaunchKernelGGL(advance_len_kernel, dim3(1), dim3(64), 0, st, kv_len, active, B);
}
void launch_frame_end(const FrameEndArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(frame_end_kernel, dim3(a.B), dim3(256), 0, st, a);
}

}  // namespace q3
