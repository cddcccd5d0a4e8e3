// voice_frontend.hip -- the non-GEMM kernels of the voice-clone front end:
//   codec encoder  (/root/reference/Sources/Qwen3TTS/Models/SpeechTokenizerEncoder.swift): first SEANet conv,
//                  LayerNorm, RoPE, causal attention, residual-VQ nearest-neighbour search;
//   speaker encoder (/root/reference/Sources/Qwen3TTS/Models/SpeakerEncoder.swift): log-mel, squeeze-excitation,
//                  attentive statistics pooling.
// Every dense contraction of both networks (convs, Linears, the windowed DFT) runs through conv_gemm_kernel
// (codec_conv.hip). These stages run once per request on a few hundred positions: they are written for
// correctness and coalesced access, not tuned against a roofline.
// Activations are fp32 channels-last [T][C], one utterance at a time.
#include <algorithm>

#include "../common.h"
#include "../codec_kernels.h"

namespace q3 {
namespace {

__device__ __forceinline__ float block_sum(float v, float* sh) {  // blockDim.x = 256
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((sh[0] + sh[1]) + sh[2]) + sh[3];
}

// out[t][c] = bias[c] + sum_k w[c][k] * audio[t - (K-1) + k]   (zero left padding; stride 1 needs no right padding)
__global__ __launch_bounds__(256) void enc_init_conv_kernel(const float* audio, int64_t S, const float* w, const float* bias,
                                                            int C, int K, float* out, int64_t out_bstride) {
    __shared__ float xs[256 + 16];
    audio += (int64_t)blockIdx.y * S;
    out += (int64_t)blockIdx.y * out_bstride;
    const int64_t t0 = (int64_t)blockIdx.x * 256;
    for (int i = threadIdx.x; i < 256 + K - 1; i += 256) {
        const int64_t t = t0 - (K - 1) + i;
        xs[i] = (t >= 0 && t < S) ? audio[t] : 0.f;
    }
    __syncthreads();
    const int n = (int)min((int64_t)256, S - t0);
    for (int i = threadIdx.x; i < n * C; i += 256) {
        const int tl = i / C, c = i % C;
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc += w[c * K + k] * xs[tl + k];
        out[(t0 + tl) * C + c] = acc + bias[c];
    }
}

__global__ __launch_bounds__(256) void layernorm_f32_kernel(const float* x, int64_t x_bstride, const float* w, const float* b,
                                                            float eps, int C, float* out, int64_t out_bstride) {
    __shared__ float sh[4];
    const float* xr = x + (size_t)blockIdx.y * x_bstride + (size_t)blockIdx.x * C;
    float* orow = out + (size_t)blockIdx.y * out_bstride + (size_t)blockIdx.x * C;
    float s = 0.f;
    for (int i = threadIdx.x; i < C; i += 256) s += xr[i];
    const float mean = block_sum(s, sh) / (float)C;
    float v = 0.f;
    for (int i = threadIdx.x; i < C; i += 256) v += (xr[i] - mean) * (xr[i] - mean);
    const float var = block_sum(v, sh) / (float)C;
    const float rstd = 1.0f / sqrtf(var + eps);
    for (int i = threadIdx.x; i < C; i += 256) orow[i] = (xr[i] - mean) * rstd * w[i] + b[i];
}

// halves layout: (x[d], x[d+32]) -> (x1*c - x2*s, x1*s + x2*c)
__global__ void rope_qk_f32_kernel(float* qkv, int heads, int T, const float* cos_t, const float* sin_t) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int per_t = 2 * heads * 32;
    if (idx >= T * per_t) return;
    const int t = idx / per_t, r = idx % per_t;
    const int hh = r >> 5, d = r & 31;  // hh in [0, 2*heads): q heads then k heads (contiguous in qkv)
    float* p = qkv + ((size_t)blockIdx.y * T + t) * 3 * heads * 64 + hh * 64;
    const float c = cos_t[t * 32 + d], s = sin_t[t * 32 + d];
    const float x1 = p[d], x2 = p[d + 32];
    p[d] = x1 * c - x2 * s;
    p[d + 32] = x1 * s + x2 * c;
}

// Causal attention, head_dim 64, one query per thread, keys/values streamed through LDS in tiles of 64.
__global__ __launch_bounds__(64) void attn_causal_f32_kernel(const float* qkv, int heads, int T, float* out) {
    constexpr int Dh = 64, TK = 64;
    __shared__ __attribute__((aligned(16))) float Ks[TK][Dh];
    __shared__ __attribute__((aligned(16))) float Vs[TK][Dh];
    const int h = blockIdx.y;
    const int q0 = blockIdx.x * 64;
    const int ld = 3 * heads * Dh;
    qkv += (size_t)blockIdx.z * T * ld;
    out += (size_t)blockIdx.z * T * heads * Dh;
    const int qi = q0 + threadIdx.x;
    const bool qvalid = qi < T;
    float q[Dh], acc[Dh];
    const float scale = 0.125f;  // headDim^-0.5 (SpeechTokenizerEncoder.swift:485)
#pragma unroll
    for (int d = 0; d < Dh; ++d) {
        q[d] = qvalid ? qkv[(size_t)qi * ld + h * Dh + d] : 0.f;
        acc[d] = 0.f;
    }
    float m = -INFINITY, l = 0.f;
    const int kend = min(T, q0 + 64);
    for (int k0 = 0; k0 < kend; k0 += TK) {
        __syncthreads();
        for (int i = threadIdx.x; i < TK * Dh / 4; i += 64) {
            const int r = i / (Dh / 4), c4 = (i % (Dh / 4)) * 4;
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
            if (k0 + r < T) {
                kv = *reinterpret_cast<const float4*>(qkv + (size_t)(k0 + r) * ld + (heads + h) * Dh + c4);
                vv = *reinterpret_cast<const float4*>(qkv + (size_t)(k0 + r) * ld + (2 * heads + h) * Dh + c4);
            }
            *reinterpret_cast<float4*>(&Ks[r][c4]) = kv;
            *reinterpret_cast<float4*>(&Vs[r][c4]) = vv;
        }
        __syncthreads();
        const int kn = min(TK, kend - k0);
        for (int j = 0; j < kn; ++j) {
            if (k0 + j > qi) continue;  // -inf above the diagonal
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
        float* o = out + (size_t)qi * heads * Dh + h * Dh;
        const float inv = 1.0f / l;
#pragma unroll
        for (int d = 0; d < Dh; ++d) o[d] = acc[d] * inv;
    }
}

// One workgroup per frame. The residual lives in LDS; every thread scans bins tid, tid+256, ... with the oracle's
// summation order (sequential over the dimensions, unfused multiply-add), then the block picks the smallest distance,
// lowest index on ties (argMin), and subtracts that code vector. The codebooks are stored transposed, [dim][bins], so
// that the 64 lanes of a wave read 64 consecutive floats per dimension (row-major rows are 1 KiB apart: one cache
// line per lane per load made this kernel 40 % of the encoder's time).
__global__ __launch_bounds__(256) void rvq_encode_kernel(const float* x, int ldx, int64_t x_bstride, const int32_t* valid,
                                                         const int64_t* code_off, int dim, int bins,
                                                         const float* const* cbt, const float* const* c2, int n_layers,
                                                         int layer0, int32_t* codes) {
    extern __shared__ float rs[];  // [dim]
    __shared__ float bv[4];
    __shared__ int bi[4];
    __shared__ int best_s;
    const int t = blockIdx.x, row = blockIdx.y;
    const int T = valid[row];  // frames of this clip (the batch is padded to a common length)
    if (t >= T) return;
    x += (size_t)row * x_bstride;
    codes += code_off[row] + (size_t)layer0 * T;
    for (int d = threadIdx.x; d < dim; d += 256) rs[d] = x[(size_t)t * ldx + d];
    __syncthreads();
    for (int layer = 0; layer < n_layers; ++layer) {
        const float* embt = cbt[layer];
        const float* cc = c2[layer];
        float best = INFINITY;
        int besti = 0x7fffffff;
        for (int j0 = 0; j0 < bins; j0 += 1024) {  // four bins per thread at a time: four independent add chains
            float dot[4] = {0.f, 0.f, 0.f, 0.f};
            int jj[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) jj[u] = j0 + u * 256 + threadIdx.x;
            int d = 0;
            for (; d + 8 <= dim; d += 8) {  // 32 loads in flight, then the adds in dimension order
                float e[8][4];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float* row = embt + (size_t)(d + k) * bins;
#pragma unroll
                    for (int u = 0; u < 4; ++u) e[k][u] = jj[u] < bins ? row[jj[u]] : 0.f;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float r = rs[d + k];
#pragma unroll
                    for (int u = 0; u < 4; ++u) dot[u] = __fadd_rn(dot[u], __fmul_rn(r, e[k][u]));
                }
            }
            for (; d < dim; ++d) {
                const float r = rs[d];
                const float* row = embt + (size_t)d * bins;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (jj[u] < bins) dot[u] = __fadd_rn(dot[u], __fmul_rn(r, row[jj[u]]));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (jj[u] < bins) {
                    const float dist = __fsub_rn(cc[jj[u]], dot[u]);
                    if (dist < best) {  // bins ascend per thread: strict < keeps the lowest index
                        best = dist;
                        besti = jj[u];
                    }
                }
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(besti, off, 64);
            if (ov < best || (ov == best && oi < besti)) {
                best = ov;
                besti = oi;
            }
        }
        if ((threadIdx.x & 63) == 0) {
            bv[threadIdx.x >> 6] = best;
            bi[threadIdx.x >> 6] = besti;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            float v = bv[0];
            int i = bi[0];
            for (int w = 1; w < 4; ++w)
                if (bv[w] < v || (bv[w] == v && bi[w] < i)) {
                    v = bv[w];
                    i = bi[w];
                }
            best_s = i;
            codes[(size_t)layer * T + t] = i;
        }
        __syncthreads();
        for (int d = threadIdx.x; d < dim; d += 256) rs[d] = __fsub_rn(rs[d], embt[(size_t)d * bins + best_s]);
        __syncthreads();
    }
}

// |X|^2 -> mel filterbank -> log(max(., 1e-10)); one workgroup per frame, one thread per mel bin
__global__ __launch_bounds__(128) void log_mel_kernel(const float* spec, int ld, int nfreq, const float* fb, int n_mels,
                                                      float* out) {
    extern __shared__ float pw[];  // [nfreq]
    const float* sr = spec + (size_t)blockIdx.x * ld;
    for (int k = threadIdx.x; k < nfreq; k += 128) {
        const float re = sr[k], im = sr[nfreq + k];
        const float mag = sqrtf(re * re + im * im);  // MLX.abs, then pow 2 (SpeakerEncoder.swift:437)
        pw[k] = mag * mag;
    }
    __syncthreads();
    for (int m = threadIdx.x; m < n_mels; m += 128) {
        float acc = 0.f;
        for (int k = 0; k < nfreq; ++k) acc = __fadd_rn(acc, __fmul_rn(pw[k], fb[(size_t)k * n_mels + m]));
        out[(size_t)blockIdx.x * n_mels + m] = logf(fmaxf(acc, 1e-10f));
    }
}

// per channel: mean over time and (optionally) sqrt(var + eps) with var = mean((x - mean)^2)
__global__ __launch_bounds__(256) void time_stats_kernel(const float* x, int ld, int T, int C, float* mean, float* stdv,
                                                         float eps) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int part = threadIdx.x >> 6;  // 4 time slices per channel
    __shared__ float sh[4][64];
    float s = 0.f;
    if (c < C)
        for (int t = part; t < T; t += 4) s += x[(size_t)t * ld + c];
    sh[part][threadIdx.x & 63] = s;
    __syncthreads();
    const float mu = (((sh[0][threadIdx.x & 63] + sh[1][threadIdx.x & 63]) + sh[2][threadIdx.x & 63]) + sh[3][threadIdx.x & 63]) / (float)T;
    __syncthreads();
    if (!stdv) {
        if (part == 0 && c < C) mean[c] = mu;
        return;
    }
    float v = 0.f;
    if (c < C)
        for (int t = part; t < T; t += 4) {
            const float d = x[(size_t)t * ld + c] - mu;
            v += d * d;
        }
    sh[part][threadIdx.x & 63] = v;
    __syncthreads();
    if (part == 0 && c < C) {
        const float var = (((sh[0][threadIdx.x] + sh[1][threadIdx.x]) + sh[2][threadIdx.x]) + sh[3][threadIdx.x]) / (float)T;
        mean[c] = mu;
        stdv[c] = sqrtf(var + eps);
    }
}

__global__ void scale_res_kernel(const float* x, int ldx, const float* se, const float* res, int ldr, float* out, int ldo,
                                 int T, int C) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T * C) return;
    const int t = idx / C, c = idx % C;
    out[(size_t)t * ldo + c] = x[(size_t)t * ldx + c] * se[c] + res[(size_t)t * ldr + c];
}

__global__ void asp_concat_kernel(const float* x, const float* mean, const float* stdv, int T, int C, float* out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T * 3 * C) return;
    const int t = idx / (3 * C), c = idx % (3 * C);
    out[idx] = c < C ? x[(size_t)t * C + c] : (c < 2 * C ? mean[c - C] : stdv[c - 2 * C]);
}

// one thread per channel: softmax over time, weighted mean and std
__global__ void asp_pool_kernel(const float* att, const float* x, int T, int C, float eps, float* pooled) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float mx = -INFINITY;
    for (int t = 0; t < T; ++t) mx = fmaxf(mx, att[(size_t)t * C + c]);
    float sum = 0.f;
    for (int t = 0; t < T; ++t) sum += expf(att[(size_t)t * C + c] - mx);
    float mean = 0.f;
    for (int t = 0; t < T; ++t) mean += (expf(att[(size_t)t * C + c] - mx) / sum) * x[(size_t)t * C + c];
    float var = 0.f;
    for (int t = 0; t < T; ++t) {
        const float d = x[(size_t)t * C + c] - mean;
        var += (expf(att[(size_t)t * C + c] - mx) / sum) * (d * d);
    }
    pooled[c] = mean;
    pooled[C + c] = sqrtf(fmaxf(var, eps));
}

// zero rows valid[b] .. Tpad-1 of clip b: what lies behind a clip's own end must be the zero padding the reference adds
// in front of a strided conv (SpeechTokenizerEncoder.swift:114-118, :184), not the activations of the padded batch
__global__ void mask_tail_kernel(float* x, int64_t bstride, const int32_t* valid, int Tpad, int C) {
    const int b = blockIdx.y;
    const int t0 = valid[b];
    const int64_t n = (int64_t)(Tpad - t0) * C;
    float* p = x + (size_t)b * bstride + (size_t)t0 * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0.f;
}

__global__ void copy2d_f32_kernel(const float* src, int lds, float* dst, int ldd, int T, int C) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T * C) return;
    const int t = idx / C, c = idx % C;
    dst[(size_t)t * ldd + c] = src[(size_t)t * lds + c];
}

inline int blocks_for(int64_t n, int per) { return (int)((n + per - 1) / per); }

}  // namespace

void launch_enc_init_conv(const float* audio, int64_t S, int B, const float* w, const float* bias, int C, int K, float* out,
                          int64_t out_bstride, hipStream_t st) {
    Q3_CHECK(K >= 1 && K <= 17, 3, "enc_init_conv: kernel size out of range");
    if (S <= 0 || B <= 0) return;
    hipLaunchKernelGGL(enc_init_conv_kernel, dim3(blocks_for(S, 256), B), dim3(256), 0, st, audio, S, w, bias, C, K, out, out_bstride);
}
void launch_layernorm_f32(const float* x, int64_t x_bstride, const float* w, const float* b, float eps, int C, int T, int B,
                          float* out, int64_t out_bstride, hipStream_t st) {
    if (T <= 0 || B <= 0) return;
    hipLaunchKernelGGL(layernorm_f32_kernel, dim3(T, B), dim3(256), 0, st, x, x_bstride, w, b, eps, C, out, out_bstride);
}
void launch_rope_qk_f32(float* qkv, int heads, int T, int B, const float* cos_t, const float* sin_t, hipStream_t st) {
    const int64_t n = (int64_t)T * 2 * heads * 32;
    if (n <= 0 || B <= 0) return;
    hipLaunchKernelGGL(rope_qk_f32_kernel, dim3(blocks_for(n, 256), B), dim3(256), 0, st, qkv, heads, T, cos_t, sin_t);
}
void launch_attn_causal_f32(const float* qkv, int heads, int T, int B, float* out, hipStream_t st) {
    if (T <= 0 || B <= 0) return;
    hipLaunchKernelGGL(attn_causal_f32_kernel, dim3((T + 63) / 64, heads, B), dim3(64), 0, st, qkv, heads, T, out);
}
void launch_rvq_encode(const float* x, int ldx, int64_t x_bstride, int Tmax, int B, const int32_t* valid, const int64_t* code_off,
                       int dim, int bins, const float* const* cb, const float* const* c2, int n_layers, int layer0,
                       int32_t* codes, hipStream_t st) {
    if (Tmax <= 0 || n_layers <= 0 || B <= 0) return;
    Q3_CHECK(dim <= 4096, 3, "rvq_encode: codebook dimension too large");
    hipLaunchKernelGGL(rvq_encode_kernel, dim3(Tmax, B), dim3(256), size_t(dim) * sizeof(float), st, x, ldx, x_bstride, valid,
                       code_off, dim, bins, cb, c2, n_layers, layer0, codes);
}
void launch_log_mel(const float* spec, int ld, int T, int nfreq, const float* fb, int n_mels, float* out, hipStream_t st) {
    if (T <= 0) return;
    hipLaunchKernelGGL(log_mel_kernel, dim3(T), dim3(128), size_t(nfreq) * sizeof(float), st, spec, ld, nfreq, fb, n_mels, out);
}
void launch_time_stats(const float* x, int ld, int T, int C, float* mean, float* std_or_null, float eps, hipStream_t st) {
    hipLaunchKernelGGL(time_stats_kernel, dim3((C + 63) / 64), dim3(256), 0, st, x, ld, T, C, mean, std_or_null, eps);
}
void launch_scale_res(const float* x, int ldx, const float* se, const float* res, int ldr, float* out, int ldo, int T, int C,
                      hipStream_t st) {
    hipLaunchKernelGGL(scale_res_kernel, dim3(blocks_for((int64_t)T * C, 256)), dim3(256), 0, st, x, ldx, se, res, ldr, out, ldo,
                       T, C);
}
void launch_asp_concat(const float* x, const float* mean, const float* stdv, int T, int C, float* out, hipStream_t st) {
    hipLaunchKernelGGL(asp_concat_kernel, dim3(blocks_for((int64_t)T * 3 * C, 256)), dim3(256), 0, st, x, mean, stdv, T, C, out);
}
void launch_asp_pool(const float* att, const float* x, int T, int C, float eps, float* pooled, hipStream_t st) {
    hipLaunchKernelGGL(asp_pool_kernel, dim3((C + 63) / 64), dim3(64), 0, st, att, x, T, C, eps, pooled);
}
void launch_mask_tail(float* x, int64_t bstride, const int32_t* valid, int Tpad, int C, int B, hipStream_t st) {
    if (Tpad <= 0 || B <= 0) return;
    const int blocks = (int)std::min<int64_t>(1024, ((int64_t)Tpad * C + 255) / 256);
    hipLaunchKernelGGL(mask_tail_kernel, dim3(blocks, B), dim3(256), 0, st, x, bstride, valid, Tpad, C);
}
void launch_copy2d_f32(const float* src, int lds, float* dst, int ldd, int T, int C, hipStream_t st) {
    hipLaunchKernelGGL(copy2d_f32_kernel, dim3(blocks_for((int64_t)T * C, 256)), dim3(256), 0, st, src, lds, dst, ldd, T, C);
}

}  // namespace q3
