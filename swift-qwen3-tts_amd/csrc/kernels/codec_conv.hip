// codec_conv.hip -- causal (dilated) 1-D convolution as an implicit GEMM on the matrix cores: conv_gemm_h2_kernel (fp16
// MFMA, every fp32 operand split into two fp16 planes: the decoder's default), resunit_h2_kernel (a whole residual unit of
// the narrow blocks in one launch) and conv_gemm_kernel (plain fp32 MFMA: the voice-clone front end, and the decoder when
// an activation leaves the fp16 range or Q3TTS_CODEC_FP32=1 asks for it).
//
// One kernel body serves every dense contraction of the codec decoder
// (/root/reference/Sources/Qwen3TTS/Models/SpeechTokenizer.swift): CausalConv1d (:259-306),
// CausalTransposeConv1d (:311-354, stored in polyphase form by model.cc so that it is a causal
// K<=2 conv whose N = stride*Cout outputs ARE the upsampled rows in channels-last memory), the
// pointwise convs / Linears of the ConvNeXt blocks and the transformer (K = 1), with
//   prologue : SnakeBeta on the input (:232-254) while staging the tile,
//   epilogue : + bias, exact-erf GELU, per-channel scale (LayerScale / gamma), + residual.
// The voice-clone front end (SpeechTokenizerEncoder.swift, SpeakerEncoder.swift) reuses it with an ELU or
// pre-add prologue, "same" (shifted, reflect-padded) windows and ReLU / sigmoid / tanh epilogues; strided
// convs arrive as K = 2 causal convs over a [T/r][r*C] view of the input (model.cc).
//
//   out[b][t][n] = res[b][t][n] + scale[n] * act( bias[n] + sum_{tap,ci} W[n][tap][ci] * A(b, t-(K-1-tap)*dil, ci) )
//
// Roofline: fp32 MFMA (v_mfma_f32_16x16x4_f32, exact fmaf chain; 157 TFLOP/s peak). Activations are
// channels-last [b][t][c]. A 128-position input tile plus its causal halo ((K-1)*dil rows) is
// staged ONCE per 32-channel chunk into LDS and all K taps read shifted windows of it (the "LDS
// ring buffer" of the north star); the weight tile of each (tap, chunk) is double-buffered in LDS
// with the next one prefetched into registers under the MFMAs. Weights are the MFMA A operand
// (M = out channels), activations the B operand (N = positions), so each lane ends with 4
// consecutive channels of one position -> 16-byte epilogue loads/stores.
// LDS rows are padded to 40 floats: conflict-free for ds_read_b128 (MI355X_MICROARCH.md LDS table).
#include <cstdlib>
#include <mutex>

#include "../common.h"
#include "../codec_kernels.h"
#include "../kernels.h"
#include "snake.h"

namespace q3 {
namespace {

constexpr int BM = 128;        // positions per workgroup
constexpr int KC = 32;         // input channels per chunk
constexpr int LDS_LD = 40;     // padded row (floats)
constexpr int MAX_HALO = 56;   // (K-1)*dil <= 54 in the decoder (k7, dil 9)

// 16-byte tensor store. (Written through to memory -- `sc0 sc1`, with or without `nt` -- the decode's stores leave no
// dirty lines in the XCD L2s; measured beside a frame loop that changes nothing: 775-778 ms per pipelined step either way.)
__device__ __forceinline__ void st16(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
// (Non-temporal tensor loads and stores were measured too: the decode alone 125 -> 133 ms, and the frame loop beside it no
// faster -- what the loop loses beside a decode is not cache space)
__device__ __forceinline__ float4 ld16f(const float* p) { return *reinterpret_cast<const float4*>(p); }

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

// geluApprox (SpeechTokenizerEncoder.swift:1080-1082), relu / sigmoid / tanh(relu) (SpeakerEncoder.swift:68, 151, 255-256)
__device__ __forceinline__ float act_other(float v, int act) {
    switch (act) {
        case 2: return v * 0.5f * (1.0f + tanhf(0.7978845608f * (v + 0.044715f * (v * v * v))));
        case 3: return fmaxf(v, 0.f);
        case 4: return 1.0f / (1.0f + expf(-v));
        default: return tanhf(fmaxf(v, 0.f));
    }
}

// Epilogue of a 64-position x CT*16-channel wave tile: + bias, activation, scale, + residual -> out; and, when the
// consumer of this tensor is a conv behind a SnakeBeta (DecoderResidualUnit act1/act2, DecoderBlock snake:
// SpeechTokenizer.swift:430-437, 474-475), the activated copy -> out2 (snake_pass), so that each element goes through
// sinf once here instead of once per output-channel tile (and halo overlap) in the consumer's staging loop.
// Lane: channels n..n+3 (rows of D), n = n_w + 16 c + 4 (lane >> 4), of positions t = t_w + 16 p + (lane & 15).
// Written in phases -- every bias / scale / residual request first, arithmetic and stores after the last of them --
// because out and res may alias as far as the compiler knows: per (p, c) block it had emitted load-wait-...-store, up to
// 3 x 16 dependent memory round trips per thread, which was most of a K = 1 launch's time (ACT: activation code compiled in).
template <int CT, bool ACT>
__device__ __forceinline__ void epilogue_tile(const ConvGemmArgs& a, int b, int t_w, int n_w, int T, int lane, f32x4 (&acc)[4][CT]) {
    const int nq = n_w + 4 * (lane >> 4);
    float4 bv[CT], sv[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int n = nq + 16 * c;
        bv[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        sv[c] = make_float4(1.f, 1.f, 1.f, 1.f);
        if (a.bias && n < a.N) bv[c] = *reinterpret_cast<const float4*>(a.bias + n);
        if (a.scale && n < a.N) sv[c] = *reinterpret_cast<const float4*>(a.scale + n);
    }
    float4 rv[4][CT];
    if (a.res) {
        const float* rb = a.res + (size_t)b * a.res_bstride;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int t = t_w + 16 * p + (lane & 15);
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const int n = nq + 16 * c;
                rv[p][c] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t < T && n < a.N) rv[p][c] = *reinterpret_cast<const float4*>(rb + (size_t)t * a.ldr + n);
            }
        }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int t = t_w + 16 * p + (lane & 15);
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int n = nq + 16 * c;
            float v[4] = {acc[p][c][0], acc[p][c][1], acc[p][c][2], acc[p][c][3]};
            if (a.bias) { v[0] += bv[c].x; v[1] += bv[c].y; v[2] += bv[c].z; v[3] += bv[c].w; }
            if constexpr (ACT) {
                if (a.act == 1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = gelu_erf(v[j]);
                } else if (a.act != 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = act_other(v[j], a.act);
                }
            }
            if (a.scale) { v[0] *= sv[c].x; v[1] *= sv[c].y; v[2] *= sv[c].z; v[3] *= sv[c].w; }
            if (a.res) { v[0] += rv[p][c].x; v[1] += rv[p][c].y; v[2] += rv[p][c].z; v[3] += rv[p][c].w; }
            if (a.out && t < T && n < a.N)
                st16(a.out + (size_t)b * a.out_bstride + (size_t)t * a.ldo + n, make_float4(v[0], v[1], v[2], v[3]));
            acc[p][c] = f32x4{v[0], v[1], v[2], v[3]};  // kept for snake_pass
        }
    }
}

// Second output: SnakeBeta of the finished tile. The accumulators are parked in LDS (each lane its own slots, so no
// barrier beyond the one that retires the main loop's tiles) and walked by a ROLLED loop: one inlined sinf body per
// 16-position slice instead of one per accumulator register (64 copies demote the accumulators to scratch).
template <int CT>
__device__ __forceinline__ void snake_pass(const ConvGemmArgs& a, float* smem_base, int b, int t0, int n0, int wn, int T, int wave,
                                           int lane, f32x4 (&acc)[4][CT], int wm) {
    constexpr int BNl = CT * 32;
    float4* stash = reinterpret_cast<float4*>(smem_base) + wave * (CT * 64);
    float4* par = reinterpret_cast<float4*>(smem_base) + 4 * (CT * 64);  // [BN / 4] exp(alpha), then [BN / 4] 1/(exp(beta)+eps)
    __syncthreads();
    if (threadIdx.x < BNl / 4) {  // this tile's SnakeBeta parameters, once per workgroup
        const int n = n0 + 4 * threadIdx.x;
        if (n < a.N) {
            const int ch = n % a.post_C;  // transposed convs: n = phase * Cout + channel
            par[threadIdx.x] = *reinterpret_cast<const float4*>(a.post_ea + ch);
            par[BNl / 4 + threadIdx.x] = *reinterpret_cast<const float4*>(a.post_ib + ch);
        }
    }
    __syncthreads();
    const int nl0 = wn * (BNl / 2) + 4 * (lane >> 4);  // this lane's first channel within the tile
#pragma unroll
    for (int p = 0; p < 4; ++p) {
#pragma unroll
        for (int c = 0; c < CT; ++c) stash[c * 64 + lane] = make_float4(acc[p][c][0], acc[p][c][1], acc[p][c][2], acc[p][c][3]);
        const int t = t0 + wm * 64 + p * 16 + (lane & 15);
        if (t >= T) continue;
        float* dst = a.out2 + (size_t)b * a.out_bstride + (size_t)t * a.ldo + n0;
#pragma unroll 1
        for (int c = 0; c < CT; ++c) {
            const int nl = nl0 + c * 16;
            if (n0 + nl >= a.N) break;
            float4 v = stash[c * 64 + lane];
            const float4 ea = par[nl >> 2], ib = par[BNl / 4 + (nl >> 2)];
            v.x = v.x + ib.x * snake_sin2(v.x * ea.x);
            v.y = v.y + ib.y * snake_sin2(v.y * ea.y);
            v.z = v.z + ib.z * snake_sin2(v.z * ea.z);
            v.w = v.w + ib.w * snake_sin2(v.w * ea.w);
            st16(dst + nl, v);
        }
    }
}

template <int BN>
__global__ __launch_bounds__(256) void conv_gemm_kernel(ConvGemmArgs a) {
    constexpr int CT = BN / 32;  // 16-channel tiles per wave (wave tile = 64 positions x BN/2 channels)
    extern __shared__ __attribute__((aligned(16))) float smem[];  // > 64 KiB for BN = 128: dynamic LDS
    float* As = smem;                                        // [(BM + halo)][LDS_LD]: sized by this conv's own halo, so that
    float* Ws0 = smem + (BM + (a.K - 1) * a.dil) * LDS_LD;   // [2][BN][LDS_LD]  three workgroups fit a CU when it is short

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;  // position half, channel half
    const int b = blockIdx.z;
    const int n0 = blockIdx.x * BN;
    const int t0 = blockIdx.y * BM;
    const int T = a.frames[b] * a.ppf;
    if (t0 >= T) return;
    const int halo = (a.K - 1) * a.dil;
    const int rows = BM + halo;
    const float* xb = a.x + (size_t)b * a.x_bstride;
    const float* x2b = a.x2;  // front-end calls are single-row (B = 1)
    const int nchunks = (a.Cin + KC - 1) / KC;
    const int steps = nchunks * a.K;

    f32x4 acc[4][CT];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    // weight tile prefetch registers: BN rows x 8 float4 per row / 256 threads
    constexpr int WV = BN * 8 / 256;
    float4 wreg[WV];
    auto load_w = [&](int step) {
        const int chunk = step / a.K, tap = step % a.K;
        const int c0 = chunk * KC;
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int item = i * 256 + tid;
            const int n = item >> 3, c4 = (item & 7) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n0 + n < a.N && c0 + c4 < a.Cin)
                v = *reinterpret_cast<const float4*>(a.w + ((size_t)(n0 + n) * a.K + tap) * a.Cin + c0 + c4);
            wreg[i] = v;
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int item = i * 256 + tid;
            const int n = item >> 3, c4 = (item & 7) * 4;
            *reinterpret_cast<float4*>(&Ws0[(buf * BN + n) * LDS_LD + c4]) = wreg[i];
        }
    };

    // input tile of one 32-channel chunk: global -> registers (prologue applied) -> LDS. The registers of chunk c+1 are
    // requested while the MFMAs of chunk c run, so only the first chunk exposes the global-memory latency (pointwise
    // convs have a single tap per chunk: their 0.85 us of MFMA work used to sit behind a 1.5 us load every chunk).
    constexpr int AV = ((BM + MAX_HALO) * 8 + 255) / 256;  // float4 per thread per chunk
    float4 areg[AV], areg2[AV];
    // load_a only REQUESTS the pieces (and the optional second input's); the prologue (pre-add, ELU, SnakeBeta) is applied
    // when the tile is staged, a chunk of MFMAs later: with expf / sinf between one piece's load and the next the compiler
    // waited for every piece right after asking for it -- one memory round trip per piece, ahead of the chunk's MFMAs.
    auto row_of = [&](int r) {
        int t = t0 - halo + a.shift + r;
        if (a.reflect) t = t < 0 ? -t : (t >= T ? 2 * (T - 1) - t : t);
        return t;
    };
    auto load_a = [&](int chunk) {
        const int c0 = chunk * KC;
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            const int item = i * 256 + tid;
            const int r = item >> 3, c4 = (item & 7) * 4;
            const int t = row_of(r);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f), u = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < rows && t >= -a.hist && t < T && c0 + c4 < a.Cin) {
                v = ld16f(xb + (int64_t)t * a.ldx + c0 + c4);
                if (x2b) u = *reinterpret_cast<const float4*>(x2b + (int64_t)t * a.ldx2 + c0 + c4);
            }
            areg[i] = v;
            areg2[i] = u;
        }
    };
    auto store_a = [&](int chunk) {
        const int c0 = chunk * KC, c4 = (tid & 7) * 4;  // a thread's pieces all sit in the same four channels
        float4 ea = make_float4(0.f, 0.f, 0.f, 0.f), ib = ea;
        if (a.snake_ea && c0 + c4 < a.Cin) {
            ea = *reinterpret_cast<const float4*>(a.snake_ea + c0 + c4);
            ib = *reinterpret_cast<const float4*>(a.snake_ib + c0 + c4);
        }
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            const int item = i * 256 + tid;
            const int r = item >> 3;
            if (r >= rows) continue;
            const int t = row_of(r);
            float4 v = areg[i];
            if (t >= -a.hist && t < T && c0 + c4 < a.Cin) {  // (padding rows and channels stay zero: not activated)
                if (x2b) { v.x += areg2[i].x; v.y += areg2[i].y; v.z += areg2[i].z; v.w += areg2[i].w; }
                if (a.pre_act == 1) {  // ELU, alpha 1 (SpeechTokenizerEncoder.swift:1075-1077)
                    v.x = v.x > 0.f ? v.x : expf(v.x) - 1.0f;
                    v.y = v.y > 0.f ? v.y : expf(v.y) - 1.0f;
                    v.z = v.z > 0.f ? v.z : expf(v.z) - 1.0f;
                    v.w = v.w > 0.f ? v.w : expf(v.w) - 1.0f;
                }
                if (a.snake_ea) {
                    float s;
                    s = sinf(v.x * ea.x); v.x = v.x + ib.x * (s * s);
                    s = sinf(v.y * ea.y); v.y = v.y + ib.y * (s * s);
                    s = sinf(v.z * ea.z); v.z = v.z + ib.z * (s * s);
                    s = sinf(v.w * ea.w); v.w = v.w + ib.w * (s * s);
                }
            }
            *reinterpret_cast<float4*>(&As[r * LDS_LD + c4]) = v;
        }
    };

    load_w(0);
    load_a(0);
    int buf = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        __syncthreads();  // previous chunk's MFMAs are done with As
        store_a(chunk);
        if (chunk + 1 < nchunks) load_a(chunk + 1);
        for (int tap = 0; tap < a.K; ++tap) {
            const int step = chunk * a.K + tap;
            store_w(buf);
            __syncthreads();
            if (step + 1 < steps) load_w(step + 1);
            const float* arow = &As[(wm * 64 + tap * a.dil + (lane & 15)) * LDS_LD + 4 * (lane >> 4)];
            const float* wrow = &Ws0[(buf * BN + wn * (BN / 2) + (lane & 15)) * LDS_LD + 4 * (lane >> 4)];
            // both k16 halves' fragments are read from LDS before the first MFMA of the step: the second half's reads
            // complete under the first half's 64 MFMAs instead of in front of its own
            float4 xa[2][4], wa[2][CT];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
#pragma unroll
                for (int p = 0; p < 4; ++p) xa[g][p] = *reinterpret_cast<const float4*>(arow + p * 16 * LDS_LD + g * 16);
#pragma unroll
                for (int c = 0; c < CT; ++c) wa[g][c] = *reinterpret_cast<const float4*>(wrow + c * 16 * LDS_LD + g * 16);
            }
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                // k-step outermost: consecutive MFMAs hit different accumulators (a dependent v_mfma_f32_16x16x4_f32
                // issues after 40 cycles, an independent one after 32: MI355X_MICROARCH.md). Each accumulator still sees
                // its k-steps in the order x, y, z, w, so the fmaf chain and the results are unchanged.
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int c = 0; c < CT; ++c) acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[g][c].x, xa[g][p].x, acc[p][c], 0, 0, 0);
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int c = 0; c < CT; ++c) acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[g][c].y, xa[g][p].y, acc[p][c], 0, 0, 0);
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int c = 0; c < CT; ++c) acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[g][c].z, xa[g][p].z, acc[p][c], 0, 0, 0);
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int c = 0; c < CT; ++c) acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[g][c].w, xa[g][p].w, acc[p][c], 0, 0, 0);
            }
            buf ^= 1;
        }
    }

    epilogue_tile<CT, true>(a, b, t0 + wm * 64, n0 + wn * (BN / 2), T, lane, acc);
    if (a.out2) snake_pass<CT>(a, smem, b, t0, n0, wn, T, wave, lane, acc, wm);
}

// ---- fp16x2 variant ------------------------------------------------------------------------------------
// Three matrix-core products per block instead of six. fp16 carries 11 significand bits, so two planes hold 22:
//   activations  x = xh + 2^-11 xl'   xh = fp16(x) (round to nearest), xl' = fp16((x - xh) * 2^11)   -- split while staging;
//                the 2^11 keeps xl' a NORMAL fp16 wherever x itself is one (|x| >= 2^-14), independent of denormal modes
//   weights      w 2^s = A + C        A = fp16(w 2^s), C = fp16(w 2^s - A), s per output channel such that the row's
//                largest |w 2^s| lies in [2^13, 2^14): C (~2^-12 of the element) is normal for every element within
//                2^-15 of the row maximum and carries an absolute error below 2^-38 of that maximum otherwise (model.cc
//                attach_h2). 2^-s is applied to the accumulators in the epilogue (exact).
//   product      w x 2^s = A xh + C xh + (A 2^-11) xl'  [+ C xl' 2^-11, dropped: ~2^-24.8 of the product, r.m.s.]
// A 2^-11 is formed in registers from the A fragment (v_pk_mul_f16, exact unless A < 2^-3, i.e. 2^-16 of the row maximum).
// Each fp16 x fp16 product is exact in fp32; what is lost is the rounding of xl' and C (operand error ~0.6 x 2^-24
// r.m.s., 2^-22 worst case) and the dropped term -- per-product noise at the level of ONE fp32 rounding, which adds up
// as sqrt(K) while an fp32 fmaf chain's own rounding adds up as K: on a K = 5376 contraction the simulated error against
// double is 0.3x the fmaf chain's (tools/split_sim.py). Range: |x| must stay below 65504 (fp16); a decoder activation
// beyond that turns into inf/NaN and reaches the PCM as a non-finite sample, which out_conv reports per row; the engine then
// decodes the flagged rows once more on the fp32 matrix-core kernel below (Engine::redo_rows_fp32, Engine::codec_decode).
// An LDS row holds the two planes of 32 input channels (2 x 64 B) + 32 B of padding = 10 sixteen-byte units (10 = 2 mod 4:
// conflict-free ds_read_b128); weight tiles are double-buffered (one barrier per tap): (128 + halo + 2 BN) x 160 B <= 70 KiB.
// (Round 4 tried this kernel with its weight fragments fetched per wave straight from the L2 into registers and one barrier
// pair per chunk, the form that made codec_conv_h1.hip 1.8x faster: bit-identical, and SLOWER here, 113.9 -> 125.8 ms per
// 32 x 200-frame decode -- with two planes the two position halves of a workgroup fetch 32 KB of identical fragments per step,
// and at 48 MFMAs per step the shared LDS tile's barrier is the cheaper of the two. Removed; tools/codec_ab.py measured it.)
constexpr int ROWH = 40;  // dwords per LDS row

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split_h2(const float4& v, uint2& hi, uint2& lo) {
    const f32x4v x = {v.x, v.y, v.z, v.w};
    const f16x4 h = __builtin_convertvector(x, f16x4);
    const f32x4v r = (x - __builtin_convertvector(h, f32x4v)) * 2048.0f;
    const f16x4 l = __builtin_convertvector(r, f16x4);
    __builtin_memcpy(&hi, &h, 8);
    __builtin_memcpy(&lo, &l, 8);
}

__device__ __forceinline__ f32x4 mfma_f16(const uint4& a, const uint4& b, f32x4 c) {
    f16x8 av, bv;
    __builtin_memcpy(&av, &a, 16);
    __builtin_memcpy(&bv, &b, 16);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, c, 0, 0, 0);
}

__device__ __forceinline__ uint4 f16x8_scale_m11(const uint4& a) {  // every fp16 of the fragment times 2^-11
    f16x8 v;
    __builtin_memcpy(&v, &a, 16);
    v = v * static_cast<_Float16>(0.00048828125f);
    uint4 r;
    __builtin_memcpy(&r, &v, 16);
    return r;
}

template <int CT>
__device__ __forceinline__ void scale_acc(const float* wsc, int N, int n_w, int lane, f32x4 (&acc)[4][CT]) {
    const int nq = n_w + 4 * (lane >> 4);
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int n = nq + 16 * c;
        float4 s = make_float4(1.f, 1.f, 1.f, 1.f);
        if (n < N) s = *reinterpret_cast<const float4*>(wsc + n);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            acc[p][c][0] *= s.x; acc[p][c][1] *= s.y; acc[p][c][2] *= s.z; acc[p][c][3] *= s.w;
        }
    }
}

template <int BN, bool PRO>
__global__ __launch_bounds__(256, 2) void conv_gemm_h2_kernel(ConvGemmArgs a) {
    constexpr int CT = BN / 32;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem3[];
    const int halo = (a.K - 1) * a.dil;
    uint32_t* As = smem3;                        // [(BM + halo)][ROWH]
    uint32_t* Ws = smem3 + (BM + halo) * ROWH;   // [2][BN][ROWH]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // Workgroup -> tile. The dispatcher hands consecutive workgroup ids to the eight XCDs in turn (cdna_hip_programming.md T1),
    // each with its own L2: with the N tiles of a position tile on consecutive ids every L2 fetched every input tile (the
    // K = 7 convs read 6-10 GB for 0.6-3 GB of input, profiles/r03_codec_traffic.txt). Each XCD now takes a contiguous run of
    // the (row, position tile, N tile) list, so the N tiles of a position tile run side by side on one L2: K = 7 convs 2-4 %
    // faster, the decode 121.2 -> 119.2 ms. (Blocks of 8 position tiles x 8 N tiles per L2 -- weights shared eight ways as
    // well -- measured no better: 121.6 ms; the transposed convs, 15-48 N tiles wide, are not bound by these fetches.)
    // Bijective for any grid; placement is a speed matter only.
    int b, n_tile, m_tile;
    {
        const uint32_t gx = gridDim.x, gy = gridDim.y, nwg = gx * gy * gridDim.z;
        const uint32_t bid = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        const uint32_t q = nwg >> 3, r = nwg & 7u, xcd = bid & 7u;
        const uint32_t swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        n_tile = int(swz % gx);
        const uint32_t rest = swz / gx;
        m_tile = int(rest % gy);
        b = int(rest / gy);
    }
    const int n0 = n_tile * BN;
    const int t0 = m_tile * BM;
    const int T = a.frames[b] * a.ppf;
    if (t0 >= T) return;
    const int rows = BM + halo;
    const float* xb = a.x + (size_t)b * a.x_bstride;
    const float* x2b = a.x2;
    const int nchunks = (a.Cin + KC - 1) / KC;
    const int steps = nchunks * a.K;

    f32x4 acc[4][CT];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    // weight tile of one (tap, chunk) step: BN x 8 sixteen-byte pieces, contiguous in global memory
    constexpr int WV = BN * 8 / 256;
    uint4 wreg[WV];
    auto load_w = [&](int step) {
        const int chunk = step / a.K, tap = step % a.K;
        const uint4* src = reinterpret_cast<const uint4*>(a.wh + ((size_t)(tap * nchunks + chunk) * a.N + n0) * 64);
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int item = i * 256 + tid;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (n0 + (item >> 3) < a.N) v = src[item];
            wreg[i] = v;
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int item = i * 256 + tid;
            *reinterpret_cast<uint4*>(&Ws[(buf * BN + (item >> 3)) * ROWH + (item & 7) * 4]) = wreg[i];
        }
    };

    constexpr int AV = ((BM + MAX_HALO) * 8 + 255) / 256;
    float4 areg[AV];
    auto load_a = [&](int chunk) {
        const int c0 = chunk * KC;
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            const int item = i * 256 + tid;
            const int r = item >> 3, c4 = (item & 7) * 4;
            int t = t0 - halo + r;
            if constexpr (PRO) {
                t += a.shift;
                if (a.reflect) t = t < 0 ? -t : (t >= T ? 2 * (T - 1) - t : t);
            }
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < rows && t >= -a.hist && t < T && c0 + c4 < a.Cin) {
                v = ld16f(xb + (int64_t)t * a.ldx + c0 + c4);
                if constexpr (PRO) {
                    if (x2b) {
                        const float4 u = *reinterpret_cast<const float4*>(x2b + (int64_t)t * a.ldx2 + c0 + c4);
                        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
                    }
                    if (a.pre_act == 1) {
                        v.x = v.x > 0.f ? v.x : expf(v.x) - 1.0f;
                        v.y = v.y > 0.f ? v.y : expf(v.y) - 1.0f;
                        v.z = v.z > 0.f ? v.z : expf(v.z) - 1.0f;
                        v.w = v.w > 0.f ? v.w : expf(v.w) - 1.0f;
                    }
                    if (a.snake_ea) {
                        const float4 ea = *reinterpret_cast<const float4*>(a.snake_ea + c0 + c4);
                        const float4 ib = *reinterpret_cast<const float4*>(a.snake_ib + c0 + c4);
                        float s;
                        s = sinf(v.x * ea.x); v.x = v.x + ib.x * (s * s);
                        s = sinf(v.y * ea.y); v.y = v.y + ib.y * (s * s);
                        s = sinf(v.z * ea.z); v.z = v.z + ib.z * (s * s);
                        s = sinf(v.w * ea.w); v.w = v.w + ib.w * (s * s);
                    }
                }
            }
            areg[i] = v;
        }
    };
    auto store_a = [&]() {
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            const int item = i * 256 + tid;
            const int r = item >> 3, c4 = (item & 7) * 4;
            if (r >= rows) continue;
            uint2 hi, lo;
            split_h2(areg[i], hi, lo);
            uint32_t* dst = &As[r * ROWH + (c4 >> 1)];
            *reinterpret_cast<uint2*>(dst) = hi;
            *reinterpret_cast<uint2*>(dst + 16) = lo;
        }
    };

    load_w(0);
    load_a(0);
    int buf = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        __syncthreads();  // the previous chunk's MFMAs are done with As
        store_a();
        if (chunk + 1 < nchunks) load_a(chunk + 1);
        for (int tap = 0; tap < a.K; ++tap) {
            const int step = chunk * a.K + tap;
            store_w(buf);     // the other buffer may still be read by a wave that is behind: double-buffered
            __syncthreads();
            if (step + 1 < steps) load_w(step + 1);
            const uint32_t* arow = &As[(wm * 64 + tap * a.dil + (lane & 15)) * ROWH + 4 * (lane >> 4)];
            const uint32_t* wrow = &Ws[(buf * BN + wn * (BN / 2) + (lane & 15)) * ROWH + 4 * (lane >> 4)];
            uint4 xa[2][4], wa[2][CT], wb[CT];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                for (int p = 0; p < 4; ++p) xa[pl][p] = *reinterpret_cast<const uint4*>(arow + p * 16 * ROWH + pl * 16);
#pragma unroll
                for (int c = 0; c < CT; ++c) wa[pl][c] = *reinterpret_cast<const uint4*>(wrow + c * 16 * ROWH + pl * 16);
            }
#pragma unroll
            for (int c = 0; c < CT; ++c) wb[c] = f16x8_scale_m11(wa[0][c]);
            // smallest products first: (A 2^-11) xl', C xh, A xh
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[p][c] = mfma_f16(wb[c], xa[1][p], acc[p][c]);
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[p][c] = mfma_f16(wa[1][c], xa[0][p], acc[p][c]);
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[p][c] = mfma_f16(wa[0][c], xa[0][p], acc[p][c]);
            buf ^= 1;
        }
    }

    scale_acc<CT>(a.wsc, a.N, n0 + wn * (BN / 2), lane, acc);
    if (a.act != 0) epilogue_tile<CT, true>(a, b, t0 + wm * 64, n0 + wn * (BN / 2), T, lane, acc);
    else epilogue_tile<CT, false>(a, b, t0 + wm * 64, n0 + wn * (BN / 2), T, lane, acc);
    if (a.out2) snake_pass<CT>(a, reinterpret_cast<float*>(smem3), b, t0, n0, wn, T, wave, lane, acc, wm);
}

// ---- pointwise form of the fp16x2 kernel ---------------------------------------------------------------
// K = 1 convs (the 768- / 384-channel residual units' second conv, the ConvNeXt linears, the pre-transformer's linears) are
// steps of 32 input channels with nothing in between: in conv_gemm_h2_kernel both tiles of step s + 1 are requested during
// step s, i.e. behind 0.3 us of MFMAs against 2-3 us of first-touch latency for the input tile -- 3.7 us per step measured, the
// matrix pipe 17 % busy. Here a step's two tiles are requested D steps ahead into a register ring. Every load is
// unconditional (addresses clamped, values masked when they are staged) and whole groups of D steps run without guards, so
// that hipcc's wait counts stay exact (s_waitcnt vmcnt((D - 1) x loads per step)) instead of draining the ring -- the two
// lessons of gemm_prefill.hip. Same tiles, same LDS layout, same MFMA order per accumulator and the same epilogue as
// conv_gemm_h2_kernel: results are bit-identical to it.
template <int BN, int D>
__global__ __launch_bounds__(256, 2) void conv_pw_h2_kernel(ConvGemmArgs a) {
    constexpr int CT = BN / 32;
    constexpr int WV = BN * 8 / 256, AV = BM * 8 / 256;
    using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
    using f32x4r = __attribute__((ext_vector_type(4))) float;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem3[];
    uint32_t* As = smem3;              // [BM][ROWH]
    uint32_t* Ws = smem3 + BM * ROWH;  // [BN][ROWH]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int b, n_tile, m_tile;
    {   // XCD-contiguous tile order (conv_gemm_h2_kernel)
        const uint32_t gx = gridDim.x, gy = gridDim.y, nwg = gx * gy * gridDim.z;
        const uint32_t bid = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        const uint32_t q = nwg >> 3, r = nwg & 7u, xcd = bid & 7u;
        const uint32_t swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        n_tile = int(swz % gx);
        const uint32_t rest = swz / gx;
        m_tile = int(rest % gy);
        b = int(rest / gy);
    }
    const int n0 = n_tile * BN, t0 = m_tile * BM;
    const int T = a.frames[b] * a.ppf;
    if (t0 >= T) return;
    const float* xb = a.x + (size_t)b * a.x_bstride;
    const int steps = (a.Cin + KC - 1) / KC;

    f32x4 acc[4][CT];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    // this thread's pieces: input rows (clamped into the row's valid range; rows outside it are zeroed when staged) and
    // weight rows (clamped; columns past N are never stored)
    const float* asrc[AV];
    bool aok[AV];
#pragma unroll
    for (int i = 0; i < AV; ++i) {
        const int item = i * 256 + tid, r = item >> 3;
        const int t = t0 + r;
        aok[i] = t >= -a.hist && t < T;
        const int tc = t < T ? t : T - 1;
        asrc[i] = xb + (int64_t)tc * a.ldx + (item & 7) * 4;
    }
    const u32x4* wsrc[WV];
#pragma unroll
    for (int i = 0; i < WV; ++i) {
        const int item = i * 256 + tid;
        const int n = n0 + (item >> 3);
        wsrc[i] = reinterpret_cast<const u32x4*>(a.wh) + (size_t)(n < a.N ? n : a.N - 1) * 8 + (item & 7);
    }
    const size_t wstep = (size_t)a.N * 8;  // sixteen-byte pieces per chunk of the weight planes
    const int cin4 = a.Cin - 4;

    f32x4r ra[D][AV];
    u32x4 rw[D][WV];
#define Q3_PW_LOAD(J, S)                                                                          \
    {                                                                                             \
        const int c0_ = (S) * KC;                                                                 \
        _Pragma("unroll") for (int i = 0; i < AV; ++i) {                                          \
            const int c_ = c0_ + ((i * 256 + tid) & 7) * 4;                                       \
            ra[J][i] = *reinterpret_cast<const f32x4r*>(asrc[i] + (c_ <= cin4 ? c0_ : cin4 - ((i * 256 + tid) & 7) * 4)); \
        }                                                                                         \
        _Pragma("unroll") for (int i = 0; i < WV; ++i) rw[J][i] = wsrc[i][(size_t)(S) * wstep];   \
    }
#define Q3_PW_STORE(J, S)                                                                         \
    {                                                                                             \
        const int c0_ = (S) * KC;                                                                 \
        _Pragma("unroll") for (int i = 0; i < AV; ++i) {                                          \
            const int item = i * 256 + tid, r = item >> 3, c4 = (item & 7) * 4;                   \
            const bool ok_ = aok[i] && c0_ + c4 <= cin4;                                          \
            const f32x4r v_ = ra[J][i];                                                           \
            const float4 f_ = ok_ ? make_float4(v_.x, v_.y, v_.z, v_.w) : make_float4(0.f, 0.f, 0.f, 0.f); \
            uint2 hi, lo;                                                                         \
            split_h2(f_, hi, lo);                                                                 \
            uint32_t* dst = &As[r * ROWH + (c4 >> 1)];                                            \
            *reinterpret_cast<uint2*>(dst) = hi;                                                  \
            *reinterpret_cast<uint2*>(dst + 16) = lo;                                             \
        }                                                                                         \
        _Pragma("unroll") for (int i = 0; i < WV; ++i) {                                          \
            const int item = i * 256 + tid;                                                       \
            *reinterpret_cast<u32x4*>(&Ws[(item >> 3) * ROWH + (item & 7) * 4]) = rw[J][i];       \
        }                                                                                         \
    }
#define Q3_PW_STEP(J, GUARD)                                                                      \
    if (!(GUARD) || s0 + (J) < steps) {                                                           \
        __syncthreads(); /* the previous step's MFMAs are done with As / Ws */                    \
        Q3_PW_STORE(J, s0 + (J))                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        { const int nx_ = s0 + (J) + D; Q3_PW_LOAD(J, nx_ < steps ? nx_ : steps - 1) }            \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        __syncthreads();                                                                          \
        const uint32_t* arow = &As[(wm * 64 + (lane & 15)) * ROWH + 4 * (lane >> 4)];             \
        const uint32_t* wrow = &Ws[(wn * (BN / 2) + (lane & 15)) * ROWH + 4 * (lane >> 4)];       \
        uint4 xa[2][4], wa[2][CT], wb[CT];                                                        \
        _Pragma("unroll") for (int pl = 0; pl < 2; ++pl) {                                        \
            _Pragma("unroll") for (int p = 0; p < 4; ++p) xa[pl][p] = *reinterpret_cast<const uint4*>(arow + p * 16 * ROWH + pl * 16); \
            _Pragma("unroll") for (int c = 0; c < CT; ++c) wa[pl][c] = *reinterpret_cast<const uint4*>(wrow + c * 16 * ROWH + pl * 16); \
        }                                                                                         \
        _Pragma("unroll") for (int c = 0; c < CT; ++c) wb[c] = f16x8_scale_m11(wa[0][c]);         \
        _Pragma("unroll") for (int p = 0; p < 4; ++p)                                             \
            _Pragma("unroll") for (int c = 0; c < CT; ++c) acc[p][c] = mfma_f16(wb[c], xa[1][p], acc[p][c]);    \
        _Pragma("unroll") for (int p = 0; p < 4; ++p)                                             \
            _Pragma("unroll") for (int c = 0; c < CT; ++c) acc[p][c] = mfma_f16(wa[1][c], xa[0][p], acc[p][c]); \
        _Pragma("unroll") for (int p = 0; p < 4; ++p)                                             \
            _Pragma("unroll") for (int c = 0; c < CT; ++c) acc[p][c] = mfma_f16(wa[0][c], xa[0][p], acc[p][c]); \
    }
#define Q3_PW_GROUP(G)                                       \
    Q3_PW_STEP(0, G) Q3_PW_STEP(1, G)                        \
    if constexpr (D > 2) { Q3_PW_STEP(2, G) }                \
    if constexpr (D > 3) { Q3_PW_STEP(3, G) }

    // (the ring is filled in the order the loop refills it -- slot by slot, input pieces then weight pieces -- and the
    // scheduler may not interleave the slots: the wait counts at the loop header are the worse of the two orders)
    Q3_PW_LOAD(0, 0)
    __builtin_amdgcn_sched_barrier(0);
    Q3_PW_LOAD(1, 1 < steps ? 1 : steps - 1)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (D > 2) { Q3_PW_LOAD(2, 2 < steps ? 2 : steps - 1) __builtin_amdgcn_sched_barrier(0); }
    if constexpr (D > 3) { Q3_PW_LOAD(3, 3 < steps ? 3 : steps - 1) __builtin_amdgcn_sched_barrier(0); }
    int s0 = 0;
    for (; s0 + D <= steps; s0 += D) { Q3_PW_GROUP(false) }
    if (s0 < steps) { Q3_PW_GROUP(true) }
#undef Q3_PW_GROUP
#undef Q3_PW_STEP
#undef Q3_PW_STORE
#undef Q3_PW_LOAD

    scale_acc<CT>(a.wsc, a.N, n0 + wn * (BN / 2), lane, acc);
    if (a.act != 0) epilogue_tile<CT, true>(a, b, t0 + wm * 64, n0 + wn * (BN / 2), T, lane, acc);
    else epilogue_tile<CT, false>(a, b, t0 + wm * 64, n0 + wn * (BN / 2), T, lane, acc);
    if (a.out2) snake_pass<CT>(a, reinterpret_cast<float*>(smem3), b, t0, n0, wn, T, wave, lane, acc, wm);
}

// ---- fused residual unit -------------------------------------------------------------------------------
// out = y + conv2(act2(conv1(act1(y)))) (DecoderResidualUnit, SpeechTokenizer.swift:430-437) for the narrow, long blocks
// (C <= 192 channels at up to 384 k positions per row), which are HBM-bound when every conv is its own launch: six
// tensor passes per unit (act1 copy in, conv1 out, conv1 out back in, y in, y out, act1 copy out) become two.
//  * act1 is applied while the raw tile of y (+ causal halo) is staged -- one N tile, so nothing is evaluated twice
//    except the halo rows;
//  * each wave owns 16 PT positions x ALL channels, so conv1's accumulators already hold conv2's whole reduction dimension:
//    an accumulator pair (tiles 2m, 2m+1) of a lane is exactly one MFMA B fragment (8 k values of its position) once
//    conv2's weights are stored with the matching k order (model.cc attach_h2_perm) -- act2 and the two-plane split
//    happen in registers and conv1's output never leaves the wave;
//  * the residual is the raw y tile again (L2-warm), the sum goes to a second buffer because neighbouring workgroups
//    still need the old halo rows.
// Numerics and LDS layout as conv_gemm_h2_kernel: three fp16 products per block, rows of 160 B; conv1's weight tiles are
// double-buffered where two workgroups still fit a CU (launch_resunit).
// Tile shape as template parameters: CT2 = C / 16 channel tiles, PT = 16-position tiles per wave (a workgroup covers 64 PT
// positions; the first build of this kernel had PT = 2 for every width). Two instantiations matter:
//   <6, 4>  C = 96, 256 positions per workgroup. With 32 positions per wave a step was 36 MFMAs per wave
//           between barriers and every wave re-read all of the weight fragments: matrix pipe 27 % busy, waves parked on
//           barriers / LDS 42 % of the time (profiles/r03_codec_pmc_summary.txt). 64 positions per wave double the work per
//           barrier and per weight fragment and halve the halo's share of the staging.
//   <12, 2> C = 192, 128 positions per workgroup: the unit that used to be two launches (k7 conv over two 96-wide N tiles,
//           then an HBM-bound pointwise conv: 5.5 TB/s of tensor traffic, profiles/r03_codec_traffic.txt) -- the input is
//           staged once instead of once per N tile and conv1's output never leaves the registers.
// conv2 runs in passes over CG output-channel tiles (its accumulators would not fit beside conv1's otherwise); each pass
// stages only its own rows of conv2's weights, so nothing is staged twice. Per accumulator the MFMA sequence -- products
// smallest first, steps in order -- does not depend on the tile shape: same results bit for bit as the PT = 2 build.
// Where the time goes (profiles/r03_codec_pmc_summary.txt, <6, 4>): 1728 MFMAs but also ~6300 other vector instructions per
// wave (two SnakeBeta evaluations and two fp16 splits per element) -- the matrix pipe shares its issue port with them.
template <int CT2, int PT>
__global__ __launch_bounds__(256, 2) void resunit_h2_kernel(ResUnitArgs a) {
    constexpr int C = 16 * CT2, NCH = CT2 / 2, PW = 16 * PT, BMU = 4 * PW;
    constexpr int CG = CT2 >= 12 ? 4 : (CT2 == 6 ? 3 : CT2);  // channel tiles per fragment group / conv2 pass
    constexpr int NG = CT2 / CG;
    static_assert(CT2 % CG == 0 && CT2 % 2 == 0, "channel tiles must split into groups and into 32-channel chunks");
    extern __shared__ __attribute__((aligned(16))) uint32_t smem3[];
    const int halo = (a.K - 1) * a.dil;
    uint32_t* As = smem3;                         // [(BMU + halo)][ROWH]
    uint32_t* Ws = smem3 + (BMU + halo) * ROWH;   // [wdb ? 2 : 1][C][ROWH]
    const bool wdb = a.wdb != 0;                  // conv1's weight tiles double-buffered (launch_resunit: when LDS allows)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z;
    const int t0 = blockIdx.y * BMU;
    const int T = a.frames[b] * a.ppf;
    if (t0 >= T) return;
    const int rows = BMU + halo;
    const size_t boff = (size_t)b * a.Tmax * C;
    const float* yb = a.y + boff;
    const int S1 = NCH * a.K;  // conv1 steps; conv2 adds NG * NCH more (pass-major)

    f32x4 acc1[PT][CT2];
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int c = 0; c < CT2; ++c) acc1[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int WV = (C * 8 + 255) / 256;
    uint4 wreg[WV];
    // step < S1: conv1's (tap, chunk) tile, C rows; else conv2's rows of pass h for chunk m, 16 CG rows
    auto load_w = [&](int step) {
        const uint4* src;
        int items;
        if (step < S1) {
            const int chunk = step / a.K, tap = step % a.K;
            src = reinterpret_cast<const uint4*>(a.w1h + (size_t)(tap * NCH + chunk) * C * 64);
            items = C * 8;
        } else {
            const int j = step - S1, h = j / NCH, m = j % NCH;
            src = reinterpret_cast<const uint4*>(a.w2ph + ((size_t)m * C + h * CG * 16) * 64);
            items = CG * 16 * 8;
        }
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int item = i * 256 + tid;
            wreg[i] = item < items ? src[item] : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto store_w = [&](int buf, int items) {
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int item = i * 256 + tid;
            if (item < items) *reinterpret_cast<uint4*>(&Ws[(buf * C + (item >> 3)) * ROWH + (item & 7) * 4]) = wreg[i];
        }
    };

    constexpr int AV = ((BMU + MAX_HALO) * 8 + 255) / 256;
    float4 areg[AV];
    // The raw tile of the NEXT chunk is only requested here; act1 is applied when the tile is staged (store_a), a chunk of
    // MFMAs later. (With SnakeBeta evaluated inside the load loop the compiler waited for every piece right after asking
    // for it: ten memory round trips in a row per chunk, in front of the chunk's MFMAs -- most of this kernel's time.)
    auto load_a = [&](int chunk) {
        const int c0 = chunk * KC;
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            const int item = i * 256 + tid;
            const int r = item >> 3, c4 = (item & 7) * 4;
            const int t = t0 - halo + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < rows && t >= -a.hist && t < T) v = ld16f(yb + (int64_t)t * C + c0 + c4);
            areg[i] = v;
        }
    };
    auto store_a = [&](int chunk) {
        const int c0 = chunk * KC, c4 = (tid & 7) * 4;  // a thread's pieces all sit in the same four channels
        const float4 ea = *reinterpret_cast<const float4*>(a.ea1 + c0 + c4);
        const float4 ib = *reinterpret_cast<const float4*>(a.ib1 + c0 + c4);
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            const int item = i * 256 + tid;
            const int r = item >> 3;
            if (r >= rows) continue;
            const int t = t0 - halo + r;
            float4 v = areg[i];
            if (t >= -a.hist && t < T) {  // (rows outside the sequence stay zero: causal padding is not activated)
                v.x = v.x + ib.x * snake_sin2(v.x * ea.x);
                v.y = v.y + ib.y * snake_sin2(v.y * ea.y);
                v.z = v.z + ib.z * snake_sin2(v.z * ea.z);
                v.w = v.w + ib.w * snake_sin2(v.w * ea.w);
            }
            uint2 hi, lo;
            split_h2(v, hi, lo);
            uint32_t* dst = &As[r * ROWH + (c4 >> 1)];
            *reinterpret_cast<uint2*>(dst) = hi;
            *reinterpret_cast<uint2*>(dst + 16) = lo;
        }
    };

    // ---- conv1 ----
    load_w(0);
    load_a(0);
    int buf = 0;
    for (int chunk = 0; chunk < NCH; ++chunk) {
        __syncthreads();  // the previous chunk's MFMAs are done with As (and, single-buffered, with Ws)
        store_a(chunk);
        if (chunk + 1 < NCH) load_a(chunk + 1);
        for (int tap = 0; tap < a.K; ++tap) {
            const int step = chunk * a.K + tap;
            if (!wdb && tap > 0) __syncthreads();  // one weight buffer: the previous tap's reads
            store_w(buf, C * 8);
            __syncthreads();
            load_w(step + 1);  // the step after conv1's last one is conv2's first
            const uint32_t* arow = &As[(PW * wave + tap * a.dil + (lane & 15)) * ROWH + 4 * (lane >> 4)];
            const uint32_t* wrow = &Ws[(buf * C + (lane & 15)) * ROWH + 4 * (lane >> 4)];
            uint4 xa[2][PT];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int p = 0; p < PT; ++p) xa[pl][p] = *reinterpret_cast<const uint4*>(arow + p * 16 * ROWH + pl * 16);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                uint4 wa[2][CG], wb[CG];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                    for (int c = 0; c < CG; ++c) wa[pl][c] = *reinterpret_cast<const uint4*>(wrow + (g * CG + c) * 16 * ROWH + pl * 16);
#pragma unroll
                for (int c = 0; c < CG; ++c) wb[c] = f16x8_scale_m11(wa[0][c]);
#pragma unroll
                for (int p = 0; p < PT; ++p)
#pragma unroll
                    for (int c = 0; c < CG; ++c) acc1[p][g * CG + c] = mfma_f16(wb[c], xa[1][p], acc1[p][g * CG + c]);
#pragma unroll
                for (int p = 0; p < PT; ++p)
#pragma unroll
                    for (int c = 0; c < CG; ++c) acc1[p][g * CG + c] = mfma_f16(wa[1][c], xa[0][p], acc1[p][g * CG + c]);
#pragma unroll
                for (int p = 0; p < PT; ++p)
#pragma unroll
                    for (int c = 0; c < CG; ++c) acc1[p][g * CG + c] = mfma_f16(wa[0][c], xa[0][p], acc1[p][g * CG + c]);
            }
            if (wdb) buf ^= 1;
        }
    }

    // ---- 2^-s, + bias1, act2 in registers (lane: channels 16c + 4(lane >> 4) + j of its PT position tiles) ----
    const int q4 = 4 * (lane >> 4);
    int big = 0;
#pragma unroll
    for (int c = 0; c < CT2; ++c) {
        const float4 sv = *reinterpret_cast<const float4*>(a.wsc1 + 16 * c + q4);
        const float4 bv = a.b1 ? *reinterpret_cast<const float4*>(a.b1 + 16 * c + q4) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 ea = *reinterpret_cast<const float4*>(a.ea2 + 16 * c + q4);
        const float e[4] = {ea.x, ea.y, ea.z, ea.w}, bb[4] = {bv.x, bv.y, bv.z, bv.w}, ss[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc1[p][c][j] = acc1[p][c][j] * ss[j] + bb[j];
                big |= !(fabsf(acc1[p][c][j] * e[j]) < 1.0e6f);
            }
    }
    // block-wide vote (also the barrier that retires conv1's reads of As and Ws): arguments beyond the polynomial's range
    // are possible only in a diverged model; the whole workgroup then takes the libm path, one tile at a time via LDS
    if (__syncthreads_or(big)) {
        // (the stash may run over As into Ws: nobody reads either between the vote and conv2's first store below, which
        // every wave reaches only after its own stash traffic)
        float4* stash = reinterpret_cast<float4*>(As) + wave * (CT2 * 64);
#pragma unroll
        for (int p = 0; p < PT; ++p) {
#pragma unroll
            for (int c = 0; c < CT2; ++c) stash[c * 64 + lane] = make_float4(acc1[p][c][0], acc1[p][c][1], acc1[p][c][2], acc1[p][c][3]);
#pragma unroll 1
            for (int c = 0; c < CT2; ++c) {
                float4 v = stash[c * 64 + lane];
                const float4 ea = *reinterpret_cast<const float4*>(a.ea2 + 16 * c + q4);
                const float4 ib = *reinterpret_cast<const float4*>(a.ib2 + 16 * c + q4);
                v.x = v.x + ib.x * snake_sin2(v.x * ea.x);
                v.y = v.y + ib.y * snake_sin2(v.y * ea.y);
                v.z = v.z + ib.z * snake_sin2(v.z * ea.z);
                v.w = v.w + ib.w * snake_sin2(v.w * ea.w);
                stash[c * 64 + lane] = v;
            }
#pragma unroll
            for (int c = 0; c < CT2; ++c) {
                const float4 v = stash[c * 64 + lane];
                acc1[p][c] = f32x4{v.x, v.y, v.z, v.w};
            }
        }
        __syncthreads();  // the stash of a slow wave must not meet a fast wave's weight store
    } else {
#pragma unroll
        for (int c = 0; c < CT2; ++c) {
            const float4 ea = *reinterpret_cast<const float4*>(a.ea2 + 16 * c + q4);
            const float4 ib = *reinterpret_cast<const float4*>(a.ib2 + 16 * c + q4);
            const float e[4] = {ea.x, ea.y, ea.z, ea.w}, ii[4] = {ib.x, ib.y, ib.z, ib.w};
#pragma unroll
            for (int p = 0; p < PT; ++p)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc1[p][c][j] = acc1[p][c][j] + ii[j] * snake_sin2_poly(acc1[p][c][j] * e[j]);
        }
    }

    // ---- conv2 in NG passes of CG output-channel tiles; the B fragments come out of acc1 ----
    const uint32_t* wrow2 = &Ws[(lane & 15) * ROWH + 4 * (lane >> 4)];
    float4* stash = reinterpret_cast<float4*>(As) + wave * (CG * 64);  // As is free since the vote; per-lane slots
#pragma unroll
    for (int h = 0; h < NG; ++h) {
        f32x4 acc2[PT][CG];
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int c = 0; c < CG; ++c) acc2[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < NCH; ++m) {
            const int j = h * NCH + m;
            if (j > 0) __syncthreads();  // the previous tile's reads of Ws (tile 0: the vote above)
            store_w(0, CG * 16 * 8);
            __syncthreads();
            if (j + 1 < NG * NCH) load_w(S1 + j + 1);
            uint4 xb[2][PT], wa[2][CG], wb[CG];
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                const float4 v0 = make_float4(acc1[p][2 * m][0], acc1[p][2 * m][1], acc1[p][2 * m][2], acc1[p][2 * m][3]);
                const float4 v1 = make_float4(acc1[p][2 * m + 1][0], acc1[p][2 * m + 1][1], acc1[p][2 * m + 1][2], acc1[p][2 * m + 1][3]);
                uint2 h0, l0, h1, l1;
                split_h2(v0, h0, l0);
                split_h2(v1, h1, l1);
                xb[0][p] = make_uint4(h0.x, h0.y, h1.x, h1.y);
                xb[1][p] = make_uint4(l0.x, l0.y, l1.x, l1.y);
            }
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int c = 0; c < CG; ++c) wa[pl][c] = *reinterpret_cast<const uint4*>(wrow2 + c * 16 * ROWH + pl * 16);
#pragma unroll
            for (int c = 0; c < CG; ++c) wb[c] = f16x8_scale_m11(wa[0][c]);
#pragma unroll
            for (int p = 0; p < PT; ++p)
#pragma unroll
                for (int c = 0; c < CG; ++c) acc2[p][c] = mfma_f16(wb[c], xb[1][p], acc2[p][c]);
#pragma unroll
            for (int p = 0; p < PT; ++p)
#pragma unroll
                for (int c = 0; c < CG; ++c) acc2[p][c] = mfma_f16(wa[1][c], xb[0][p], acc2[p][c]);
#pragma unroll
            for (int p = 0; p < PT; ++p)
#pragma unroll
                for (int c = 0; c < CG; ++c) acc2[p][c] = mfma_f16(wa[0][c], xb[0][p], acc2[p][c]);
        }
        // ---- this pass's channels: 2^-s, + bias2, + y -> out; optionally the next block's SnakeBeta of the sum -> out2 ----
        // (every residual request before the first store: out and y could alias as far as the compiler knows)
        float4 rvs[PT][CG], bv2[CG], sv2[CG];
#pragma unroll
        for (int c = 0; c < CG; ++c) {
            const int n = 16 * (h * CG + c) + q4;
            bv2[c] = a.b2 ? *reinterpret_cast<const float4*>(a.b2 + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            sv2[c] = *reinterpret_cast<const float4*>(a.wsc2 + n);
        }
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            const int t = t0 + PW * wave + 16 * p + (lane & 15);
#pragma unroll
            for (int c = 0; c < CG; ++c) {
                rvs[p][c] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t < T) rvs[p][c] = *reinterpret_cast<const float4*>(yb + (size_t)t * C + 16 * (h * CG + c) + q4);
            }
        }
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            const int t = t0 + PW * wave + 16 * p + (lane & 15);
#pragma unroll
            for (int c = 0; c < CG; ++c) {
                const int n = 16 * (h * CG + c) + q4;
                const float4 rv = rvs[p][c], bv = bv2[c], sv = sv2[c];
                float4 v = make_float4(acc2[p][c][0] * sv.x + rv.x, acc2[p][c][1] * sv.y + rv.y, acc2[p][c][2] * sv.z + rv.z,
                                       acc2[p][c][3] * sv.w + rv.w);
                if (a.b2)
                    v = make_float4((acc2[p][c][0] * sv.x + bv.x) + rv.x, (acc2[p][c][1] * sv.y + bv.y) + rv.y,
                                    (acc2[p][c][2] * sv.z + bv.z) + rv.z, (acc2[p][c][3] * sv.w + bv.w) + rv.w);
                if (t < T) st16(a.out + boff + (size_t)t * C + n, v);
                acc2[p][c] = f32x4{v.x, v.y, v.z, v.w};
            }
        }
        if (a.out2) {
#pragma unroll
            for (int p = 0; p < PT; ++p) {
#pragma unroll
                for (int c = 0; c < CG; ++c) stash[c * 64 + lane] = make_float4(acc2[p][c][0], acc2[p][c][1], acc2[p][c][2], acc2[p][c][3]);
                const int t = t0 + PW * wave + 16 * p + (lane & 15);
                if (t >= T) continue;
#pragma unroll 1
                for (int c = 0; c < CG; ++c) {
                    const int n = 16 * (h * CG + c) + q4;
                    float4 v = stash[c * 64 + lane];
                    const float4 ea = *reinterpret_cast<const float4*>(a.post_ea + n);
                    const float4 ib = *reinterpret_cast<const float4*>(a.post_ib + n);
                    v.x = v.x + ib.x * snake_sin2(v.x * ea.x);
                    v.y = v.y + ib.y * snake_sin2(v.y * ea.y);
                    v.z = v.z + ib.z * snake_sin2(v.z * ea.z);
                    v.w = v.w + ib.w * snake_sin2(v.w * ea.w);
                    st16(a.out2 + boff + (size_t)t * C + n, v);
                }
            }
        }
    }
}

}  // namespace

bool resunit_supported(int C, int K, int dil) {
    return (C == 32 || C == 64 || C == 96 || C == 192) && K >= 1 && (K - 1) * dil <= MAX_HALO;
}

void launch_resunit(const ResUnitArgs& a0, hipStream_t st) {
    static std::once_flag attr_once[kMaxDevices];
    ResUnitArgs a = a0;
    Q3_CHECK(resunit_supported(a.C, a.K, a.dil) && a.out != a.y, 3, "resunit: unsupported geometry");
    Q3_CHECK(a.w1h && a.w2ph && a.wsc1 && a.wsc2, 3, "resunit: incomplete fp16x2 weights");
    if (a.Tmax <= 0 || a.B <= 0) return;
    // once per DEVICE (a process may hold one handle per GPU; the attribute belongs to the device's copy of the kernel) and
    // thread-safe (lanes launch from their own threads)
    std::call_once(device_once(attr_once), [] {
        void (*ks[4])(ResUnitArgs) = {&resunit_h2_kernel<2, 2>, &resunit_h2_kernel<4, 2>, &resunit_h2_kernel<6, 4>, &resunit_h2_kernel<12, 2>};
        for (auto k : ks)
            Q3_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    });
    const int bmu = a.C == 96 ? 256 : 128;  // positions per workgroup (64 PT)
    const int rows = bmu + (a.K - 1) * a.dil;
    // conv1's weight tiles are double-buffered (one barrier per tap instead of two) where two workgroups still share a CU
    a.wdb = size_t(rows + 2 * a.C) * ROWH * sizeof(uint32_t) <= 80 * 1024 ? 1 : 0;
    const size_t smemh = size_t(rows + (a.wdb ? 2 : 1) * a.C) * ROWH * sizeof(uint32_t);
    dim3 grid(1, (a.Tmax + bmu - 1) / bmu, a.B), block(256);
    switch (a.C) {
        case 32: hipLaunchKernelGGL((resunit_h2_kernel<2, 2>), grid, block, smemh, st, a); break;
        case 64: hipLaunchKernelGGL((resunit_h2_kernel<4, 2>), grid, block, smemh, st, a); break;
        case 96: hipLaunchKernelGGL((resunit_h2_kernel<6, 4>), grid, block, smemh, st, a); break;
        default: hipLaunchKernelGGL((resunit_h2_kernel<12, 2>), grid, block, smemh, st, a); break;
    }
}

void launch_conv_gemm(const ConvGemmArgs& a, hipStream_t st) {
    static std::once_flag attr_once[kMaxDevices];
    Q3_CHECK((a.K - 1) * a.dil <= MAX_HALO, 3, "conv_gemm: receptive field too large");
    Q3_CHECK(!a.x2 || (a.B == 1 && a.ldx2 % 4 == 0), 3, "conv_gemm: the pre-add input is single-row only");
    Q3_CHECK(a.Cin % 4 == 0 && a.N % 4 == 0 && a.ldx % 4 == 0 && a.ldo % 4 == 0, 3, "conv_gemm: channels must be multiples of 4");
    const int mt = (a.Tmax + BM - 1) / BM;
    if (mt <= 0 || a.B <= 0) return;
    // tile width: 128 when it divides evenly, 96 for the 96/192/288-channel stages, else 64 (masked tail)
    int BN = 64;
    if (a.N % 128 == 0) BN = 128;
    else if (a.N % 96 == 0) BN = 96;
    else if (a.N > 128 && (a.N % 64) != 0) BN = 128;
    dim3 grid((a.N + BN - 1) / BN, mt, a.B), block(256);
    // once per DEVICE (a process may hold one handle per GPU; the attribute belongs to the device's copy of the kernel) and
    // thread-safe (lanes launch from their own threads)
    std::call_once(device_once(attr_once), [] {  // 160 KiB of LDS per CU on gfx950; the default static cap is 64 KiB
        Q3_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        Q3_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_kernel<96>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        Q3_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        void (*h2_kernels[6])(ConvGemmArgs) = {&conv_gemm_h2_kernel<128, true>, &conv_gemm_h2_kernel<128, false>,
                                                &conv_gemm_h2_kernel<96, true>,  &conv_gemm_h2_kernel<96, false>,
                                                &conv_gemm_h2_kernel<64, true>,  &conv_gemm_h2_kernel<64, false>};
        for (auto k : h2_kernels)
            Q3_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    });
    if (a.wh) {
        Q3_CHECK(a.wsc != nullptr, 3, "conv_gemm: fp16x2 weights without their row scales");
        const size_t smemh = size_t(BM + (a.K - 1) * a.dil + 2 * BN) * ROWH * sizeof(uint32_t);
        const bool pro = a.x2 || a.pre_act || a.snake_ea || a.shift || a.reflect;
        if (!pro && a.K == 1 && !debug_env().conv_no_pw) {  // pointwise: both tiles requested PWD steps ahead (kernel)
            constexpr int PWD = 2;
            const size_t smpw = size_t(BM + BN) * ROWH * sizeof(uint32_t);
            switch (BN) {
                case 128: hipLaunchKernelGGL((conv_pw_h2_kernel<128, PWD>), grid, block, smpw, st, a); break;
                case 96: hipLaunchKernelGGL((conv_pw_h2_kernel<96, PWD>), grid, block, smpw, st, a); break;
                default: hipLaunchKernelGGL((conv_pw_h2_kernel<64, PWD>), grid, block, smpw, st, a); break;
            }
            return;
        }
        auto go = [&](void (*kern)(ConvGemmArgs)) { hipLaunchKernelGGL(kern, grid, block, smemh, st, a); };
        switch (BN) {
            case 128: pro ? go(&conv_gemm_h2_kernel<128, true>) : go(&conv_gemm_h2_kernel<128, false>); break;
            case 96: pro ? go(&conv_gemm_h2_kernel<96, true>) : go(&conv_gemm_h2_kernel<96, false>); break;
            default: pro ? go(&conv_gemm_h2_kernel<64, true>) : go(&conv_gemm_h2_kernel<64, false>); break;
        }
        return;
    }
    const size_t smem = size_t(BM + (a.K - 1) * a.dil + 2 * BN) * LDS_LD * sizeof(float);
    switch (BN) {
        case 128: hipLaunchKernelGGL(conv_gemm_kernel<128>, grid, block, smem, st, a); break;
        case 96: hipLaunchKernelGGL(conv_gemm_kernel<96>, grid, block, smem, st, a); break;
        default: hipLaunchKernelGGL(conv_gemm_kernel<64>, grid, block, smem, st, a); break;
    }
}

}  // namespace q3
