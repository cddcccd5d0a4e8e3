// codec_misc.hip -- the non-GEMM pieces of the codec decoder
// (/root/reference/Sources/Qwen3TTS/Models/SpeechTokenizer.swift): Split-RVQ gather, fp32 RMSNorm,
// depthwise conv + LayerNorm, SwiGLU gating, full attention of the 8-layer pre-transformer and the
// Snake -> conv(C->1) -> clip tail. All are HBM/L2 bound row kernels over channels-last tensors.
#include <algorithm>

#include "../common.h"
#include "snake.h"
#include "../codec_kernels.h"

namespace q3 {
namespace {

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((sh[0] + sh[1]) + sh[2]) + sh[3];
}

// one workgroup per (frame, row)
__global__ __launch_bounds__(256) void rvq_gather_kernel(const int32_t* codes, int code_stride_frames,
                                                         const float* cb_first, const float* const* cb_rest, int n_rest,
                                                         int inner, const int32_t* frames, int Fmax, float* out,
                                                         int rows_first, int rows_rest) {
    const int f = blockIdx.x, b = blockIdx.y;
    if (f >= frames[b]) return;
    const int32_t* c = codes + ((size_t)b * code_stride_frames + f) * 16;
    float* o = out + ((size_t)b * Fmax + f) * 2 * inner;
    // a code is a row index: kept inside the table whatever produced it (caller codes are rejected on the host before they get
    // here, Engine::check_caller_codes; sampled codes are inside by construction -- the clamp is the seat belt, not the rule)
    auto row = [](int32_t code, int rows) { return (size_t)(code < 0 ? 0 : (code < rows ? code : rows - 1)); };
    for (int i = threadIdx.x; i < inner; i += 256) {
        o[i] = cb_first[row(c[0], rows_first) * inner + i];
        float q = cb_rest[0][row(c[1], rows_rest) * inner + i];  // layers summed in index order (:84-93)
        for (int j = 1; j < n_rest; ++j) q = q + cb_rest[j][row(c[1 + j], rows_rest) * inner + i];
        o[inner + i] = q;
    }
}

__global__ __launch_bounds__(256) void rmsnorm_f32_kernel(const float* x, const float* w, float eps, int C,
                                                          const int32_t* frames, int ppf, int Tmax, float* out) {
    __shared__ float sh[4];
    const int t = blockIdx.x, b = blockIdx.y;
    if (t >= frames[b] * ppf) return;
    const float* xr = x + ((size_t)b * Tmax + t) * C;
    float* orow = out + ((size_t)b * Tmax + t) * C;
    float ss = 0.f;
    for (int i = threadIdx.x; i < C; i += 256) ss += xr[i] * xr[i];
    const float tot = block_sum_256(ss, sh);
    const float rstd = 1.0f / sqrtf(tot / (float)C + eps);
    for (int i = threadIdx.x; i < C; i += 256) orow[i] = (xr[i] * rstd) * w[i];
}

// ConvNeXt front half: y = dwconv_k7_causal(x) + b ; out = LayerNorm(y) (eps 1e-6)
__global__ __launch_bounds__(256) void dwconv_ln_kernel(const float* x, const float* dw_w, const float* dw_b,
                                                        const float* ln_w, const float* ln_b, float eps, int C,
                                                        const int32_t* frames, int ppf, int Tmax, float* out, int hist) {
    __shared__ float sh[4];
    constexpr int kMaxPer = 16;  // C <= 4096
    const int t = blockIdx.x, b = blockIdx.y;
    if (t >= frames[b] * ppf) return;
    const float* xb = x + (size_t)b * Tmax * C;
    float y[kMaxPer];
    float s = 0.f;
    int cnt = 0;
    for (int c = threadIdx.x; c < C; c += 256, ++cnt) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int ti = t - 6 + k;
            if (ti >= -hist) acc += xb[(int64_t)ti * C + c] * dw_w[c * 7 + k];
        }
        acc += dw_b[c];
        y[cnt] = acc;
        s += acc;
    }
    const float mean = block_sum_256(s, sh) / (float)C;
    float v = 0.f;
    for (int i = 0; i < cnt; ++i) v += (y[i] - mean) * (y[i] - mean);
    const float var = block_sum_256(v, sh) / (float)C;
    const float rstd = 1.0f / sqrtf(var + eps);
    float* orow = out + ((size_t)b * Tmax + t) * C;
    cnt = 0;
    for (int c = threadIdx.x; c < C; c += 256, ++cnt) orow[c] = (y[cnt] - mean) * rstd * ln_w[c] + ln_b[c];
}

__global__ void silu_mul_f32_kernel(const float* gu, int I, const int32_t* frames, int ppf, int Tmax, float* out) {
    const int t = blockIdx.x, b = blockIdx.y;
    if (t >= frames[b] * ppf) return;
    const float* g = gu + ((size_t)b * Tmax + t) * 2 * I;
    float* o = out + ((size_t)b * Tmax + t) * I;
    for (int i = threadIdx.x; i < I; i += blockDim.x) {
        const float gv = g[i];
        o[i] = (gv / (1.0f + expf(-gv))) * g[I + i];
    }
}

// Full attention, head_dim 64, one query per thread, keys/values streamed through LDS in tiles of 64.
__global__ __launch_bounds__(64) void attn_full_f32_kernel(const float* qkv, int heads, const int32_t* frames,
                                                           int Tmax, float* out) {
    constexpr int Dh = 64, TK = 64;
    __shared__ __attribute__((aligned(16))) float Ks[TK][Dh];
    __shared__ __attribute__((aligned(16))) float Vs[TK][Dh];
    const int b = blockIdx.z, h = blockIdx.y;
    const int T = frames[b];
    const int q0 = blockIdx.x * 64;
    if (q0 >= T) return;
    const int ld = 3 * heads * Dh;
    const float* base = qkv + (size_t)b * Tmax * ld;
    const int qi = q0 + threadIdx.x;
    const bool qvalid = qi < T;
    float q[Dh], acc[Dh];
    const float scale = 0.125f;  // 64^-0.5 (SpeechTokenizer.swift:502)
#pragma unroll
    for (int d = 0; d < Dh; ++d) {
        q[d] = qvalid ? base[(size_t)qi * ld + h * Dh + d] : 0.f;
        acc[d] = 0.f;
    }
    float m = -INFINITY, l = 0.f;
    for (int k0 = 0; k0 < T; k0 += TK) {
        __syncthreads();
        for (int i = threadIdx.x; i < TK * Dh / 4; i += 64) {
            const int r = i / (Dh / 4), c4 = (i % (Dh / 4)) * 4;
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
            if (k0 + r < T) {
                kv = *reinterpret_cast<const float4*>(base + (size_t)(k0 + r) * ld + (heads + h) * Dh + c4);
                vv = *reinterpret_cast<const float4*>(base + (size_t)(k0 + r) * ld + (2 * heads + h) * Dh + c4);
            }
            *reinterpret_cast<float4*>(&Ks[r][c4]) = kv;
            *reinterpret_cast<float4*>(&Vs[r][c4]) = vv;
        }
        __syncthreads();
        const int kn = min(TK, T - k0);
        for (int j = 0; j < kn; ++j) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < Dh; ++d) s += q[d] * Ks[j][d];
            s *= scale;
            const float mn = fmaxf(m, s);
            const float alpha = expf(m - mn), p = expf(s - mn);
            l = l * alpha + p;
#pragma unroll
            for (int d = 0; d < Dh; ++d) acc[d] = acc[d] * alpha + p * Vs[j][d];
            m = mn;
        }
    }
    if (qvalid) {
        float* o = out + ((size_t)b * Tmax + qi) * heads * Dh + h * Dh;
        const float inv = 1.0f / l;
#pragma unroll
        for (int d = 0; d < Dh; ++d) o[d] = acc[d] * inv;
    }
}

// out[t] = clip(bias + sum_{k,c} snake(x[t-6+k][c]) * w[k][c]); 64 positions per workgroup, 4 lanes each.
// The activated tile (70 rows, stride C + 4 floats so that the 16 positions of a ds_read_b128 lane group spread over
// the banks) and the 7 x C taps sit in LDS; a lane owns every fourth 4-channel group of its position.
__global__ __launch_bounds__(256) void out_conv_kernel(const float* x, int C, const float* ea, const float* ib,
                                                       const float* w, const float* bias, const int32_t* frames, int ppf,
                                                       int Tmax, float* pcm, int hist, int32_t* nonfinite) {
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [(64+6)][C + 4] snake(x), then [7][C] taps
    const int ld = C + 4, C4 = C >> 2;
    float* ws = xs + 70 * ld;
    const int b = blockIdx.y, t0 = blockIdx.x * 64;
    const int T = frames[b] * ppf;
    if (t0 >= T) return;
    const float* xb = x + (size_t)b * Tmax * C;
    for (int i = threadIdx.x; i < 7 * C4; i += 256) *reinterpret_cast<float4*>(ws + 4 * i) = *reinterpret_cast<const float4*>(w + 4 * i);
    {   // (row, 4-channel group) walk without a division per element. Eight pieces are requested before the first of them is
        // activated: with SnakeBeta between one load and the next every piece was a memory round trip of its own.
        int r = threadIdx.x / C4, c4 = threadIdx.x % C4;
        const int dr = 256 / C4, dc = 256 % C4;
        while (r < 70) {
            float4 raw[8];
            int rr[8], cc[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                rr[u] = r;
                cc[u] = c4;
                const int t = t0 - 6 + r;
                raw[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < 70 && t >= -hist && t < T) raw[u] = *reinterpret_cast<const float4*>(xb + (int64_t)t * C + 4 * c4);
                r += dr;
                c4 += dc;
                if (c4 >= C4) {
                    c4 -= C4;
                    ++r;
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (rr[u] >= 70) continue;
                const int t = t0 - 6 + rr[u];
                float4 v = raw[u];
                if (t >= -hist && t < T) {
                    const float4 e = *reinterpret_cast<const float4*>(ea + 4 * cc[u]), q = *reinterpret_cast<const float4*>(ib + 4 * cc[u]);
                    v.x = v.x + q.x * snake_sin2(v.x * e.x);
                    v.y = v.y + q.y * snake_sin2(v.y * e.y);
                    v.z = v.z + q.z * snake_sin2(v.z * e.z);
                    v.w = v.w + q.w * snake_sin2(v.w * e.w);
                }
                *reinterpret_cast<float4*>(xs + rr[u] * ld + 4 * cc[u]) = v;
            }
        }
    }
    __syncthreads();
    const int p = threadIdx.x >> 2, part = threadIdx.x & 3;
    float acc = 0.f;
    for (int k = 0; k < 7; ++k) {
        const float* xr = xs + (p + k) * ld;
        const float* wr = ws + k * C;
        for (int g = part; g < C4; g += 4) {
            const float4 xv = *reinterpret_cast<const float4*>(xr + 4 * g), wv = *reinterpret_cast<const float4*>(wr + 4 * g);
            acc += xv.x * wv.x;
            acc += xv.y * wv.y;
            acc += xv.z * wv.z;
            acc += xv.w * wv.w;
        }
    }
    acc += XorPartner::x1(acc);
    acc += XorPartner::x2(acc);
    const int t = t0 + p;
    if (part == 0 && t < T) {
        const float v = acc + bias[0];
        // an activation beyond the fp16 range of the two-plane convs (codec_conv.hip) arrives here as inf / NaN: tell the host
        if (nonfinite && !(fabsf(v) < INFINITY)) atomicOr(nonfinite + b, 1);
        pcm[(size_t)b * Tmax + t] = fminf(fmaxf(v, -1.0f), 1.0f);
    }
}

// streamed decode: the last `keep` floats of a chunk's valid region become the history in front of the next chunk
__global__ __launch_bounds__(256) void roll_history_kernel(float* cur, int64_t bstride, int64_t keep, int64_t chunk) {
    float* row = cur + (int64_t)blockIdx.y * bstride;
    const float4* src = reinterpret_cast<const float4*>(row + chunk - keep);
    float4* dst = reinterpret_cast<float4*>(row - keep);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < keep / 4; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}

}  // namespace

void launch_roll_history(float* cur, int64_t bstride, int64_t keep_floats, int64_t chunk_floats, int B, hipStream_t st) {
    Q3_CHECK(keep_floats % 4 == 0 && chunk_floats >= keep_floats, 3, "roll_history: chunk shorter than the history");
    if (keep_floats == 0 || B <= 0) return;
    const int gx = int(std::min<int64_t>(64, (keep_floats / 4 + 255) / 256));
    hipLaunchKernelGGL(roll_history_kernel, dim3(gx, B), dim3(256), 0, st, cur, bstride, keep_floats, chunk_floats);
}

void launch_rvq_gather(const int32_t* codes, int code_stride_frames, const float* cb_first, const float* const* cb_rest,
                       int n_rest, int inner, const int32_t* frames, int Fmax, int B, float* out, int rows_first, int rows_rest,
                       hipStream_t st) {
    Q3_CHECK(rows_first >= 1 && rows_rest >= 1 && n_rest >= 1 && n_rest <= 15, 3, "rvq_gather: empty codebook");
    hipLaunchKernelGGL(rvq_gather_kernel, dim3(Fmax, B), dim3(256), 0, st, codes, code_stride_frames, cb_first, cb_rest,
                       n_rest, inner, frames, Fmax, out, rows_first, rows_rest);
}
void launch_rmsnorm_f32(const float* x, const float* w, float eps, int C, const int32_t* frames, int ppf, int Tmax, int B,
                        float* out, hipStream_t st) {
    hipLaunchKernelGGL(rmsnorm_f32_kernel, dim3(Tmax, B), dim3(256), 0, st, x, w, eps, C, frames, ppf, Tmax, out);
}
void launch_dwconv_ln(const float* x, const float* dw_w, const float* dw_b, const float* ln_w, const float* ln_b,
                      float eps, int C, const int32_t* frames, int ppf, int Tmax, int B, float* out, hipStream_t st, int hist) {
    Q3_CHECK(C <= 4096, 3, "dwconv_ln: more than 4096 channels");
    hipLaunchKernelGGL(dwconv_ln_kernel, dim3(Tmax, B), dim3(256), 0, st, x, dw_w, dw_b, ln_w, ln_b, eps, C, frames, ppf,
                       Tmax, out, hist);
}
void launch_silu_mul_f32(const float* gu, int I, const int32_t* frames, int ppf, int Tmax, int B, float* out,
                         hipStream_t st) {
    hipLaunchKernelGGL(silu_mul_f32_kernel, dim3(Tmax, B), dim3(256), 0, st, gu, I, frames, ppf, Tmax, out);
}
void launch_attn_full_f32(const float* qkv, int heads, const int32_t* frames, int Tmax, int B, float* out,
                          hipStream_t st) {
    hipLaunchKernelGGL(attn_full_f32_kernel, dim3((Tmax + 63) / 64, heads, B), dim3(64), 0, st, qkv, heads, frames, Tmax,
                       out);
}
void launch_out_conv(const float* x, int C, const float* ea, const float* ib, const float* w, const float* bias,
                     const int32_t* frames, int ppf, int Tmax, int B, float* pcm, hipStream_t st, int hist, int32_t* nonfinite) {
    const size_t smem = (size_t(70) * (C + 4) + size_t(7) * C) * sizeof(float);
    Q3_CHECK(smem <= 64 * 1024 && C % 4 == 0 && C >= 4 && C <= 1024, 3, "out_conv: unsupported channel count");
    hipLaunchKernelGGL(out_conv_kernel, dim3((Tmax + 63) / 64, B), dim3(256), smem, st, x, C, ea, ib, w, bias, frames, ppf,
                       Tmax, pcm, hist, nonfinite);
}

}  // namespace q3
