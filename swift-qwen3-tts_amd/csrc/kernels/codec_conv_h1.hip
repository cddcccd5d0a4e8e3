// codec_conv_h1.hip -- the MainDecoder's convolutions for FLOAT16 speech tokenizers ("lite" checkpoints store every
// speech-tokenizer tensor in float16, /root/reference/docs/paper.tex:207).
//
// What the reference computes there: MLX evaluates every op in the arrays' dtype, so with float16 weights the decoder's
// tensors ARE float16 and every op result is rounded to float16 (SpeechTokenizer.swift: CausalConv1d :298-306 = conv, then
// + bias; CausalTransposeConv1d :346-352; SnakeBeta :246-253 = x * alpha, sin, s * s, (1 / beta) * q, x + r;
// DecoderResidualUnit :430-437 = residual + h). Up-casting such a checkpoint to fp32 and running the two-plane kernels
// (codec_conv.hip) is both wider than the reference and three matrix-core products where one does: here activations stay
// float16 in HBM (half the bytes), a product block is ONE v_mfma_f32_16x16x32_f16 (exact fp16 x fp16 products, fp32
// accumulation), and the epilogue rounds where MLX rounds: fp16(acc), fp16(+ bias), fp16(res + .), and the five roundings of
// SnakeBeta for the activated copy the next conv reads. Oracle: OracleModel._main_decoder16.
//
// Same tiling as conv_gemm_h2_kernel: 128 positions x {128, 96, 64} channels per workgroup of four waves, input tile + causal
// halo staged once per 32-channel chunk, all K taps read shifted windows of it, weight tiles double-buffered (one barrier per
// tap), XCD-contiguous tile order. An LDS row is 32 halfs (64 B) + 16 B of padding = 5 sixteen-byte units: the 16 rows x 4
// units of a ds_read_b128 wave access fall on distinct bank groups. X32: the input tensor is still fp32 (initConv reads
// the last ConvNeXt stage's output) and is rounded to float16 while it is staged -- the tensor the reference holds.
// Roofline: fp16 MFMA (2.5 PFLOP/s dense) for the K = 7 convs, HBM for the pointwise and transposed ones.
#include <algorithm>

#include "../codec_kernels.h"
#include "../common.h"
#include "snake.h"

namespace q3 {
namespace {

constexpr int BM = 128;       // positions per workgroup
constexpr int KC = 32;        // input channels per chunk
constexpr int RH = 20;        // dwords per LDS row: 64 B of data + 16 B padding
constexpr int MAX_HALO = 56;  // (K - 1) * dil <= 54 in the decoder

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float r16(float v) { return static_cast<float>(static_cast<_Float16>(v)); }

__device__ __forceinline__ f32x4 mfma_h(const uint4& a, const uint4& b, f32x4 c) {
    f16x8 av, bv;
    __builtin_memcpy(&av, &a, 16);
    __builtin_memcpy(&bv, &b, 16);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, c, 0, 0, 0);
}

__device__ __forceinline__ uint2 pack_h4(float a, float b, float c, float d) {
    const f32x4v x = {a, b, c, d};
    const f16x4 h = __builtin_convertvector(x, f16x4);
    uint2 r;
    __builtin_memcpy(&r, &h, 8);
    return r;
}
__device__ __forceinline__ void unpack_h4(const uint2& u, float (&o)[4]) {
    f16x4 h;
    __builtin_memcpy(&h, &u, 8);
    const f32x4v x = __builtin_convertvector(h, f32x4v);
    o[0] = x[0]; o[1] = x[1]; o[2] = x[2]; o[3] = x[3];
}

typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

// |sin(t)| is all SnakeBeta needs (it squares it): reduce by multiples of pi to [-pi/2, pi/2] (Cody-Waite, two FMAs; t is a
// float16 value, |t| <= 65504) and take the odd degree-11 polynomial there (< 1 ulp of fp32); the sign is dropped by the square
__device__ __forceinline__ float sin_mod_pi(float t) {
    const float k = __builtin_rintf(t * 0.318309886183790672f);
    float r = __builtin_fmaf(k, -3.14159274101257324f, t);
    r = __builtin_fmaf(k, 8.74227765734758577e-8f, r);
    const float r2 = r * r;
    float p = __builtin_fmaf(r2, -2.50521083854417188e-8f, 2.75573192239858907e-6f);
    p = __builtin_fmaf(r2, p, -1.98412698412698413e-4f);
    p = __builtin_fmaf(r2, p, 8.33333333333333333e-3f);
    p = __builtin_fmaf(r2, p, -1.66666666666666667e-1f);
    return __builtin_fmaf(r * r2, p, r);
}

// SnakeBeta the way MLX evaluates it on float16 arrays (SpeechTokenizer.swift:251-252), two elements at a time on the packed
// float16 ALU: every mul / add below IS one float16 op with one rounding (x * alpha, s * s, (1 / beta) * q, x + r); the sine is
// taken in fp32 and rounded once. ea / ib: the float16-rounded exp(alpha) and 1 / exp(beta) (model.cc put_snake).
__device__ __forceinline__ h2v snake_h2(h2v x, h2v ea, h2v ib) {
    const h2v t = x * ea;
    const f32x2v tf = __builtin_convertvector(t, f32x2v);
    const f32x2v sf = {sin_mod_pi(tf[0]), sin_mod_pi(tf[1])};
    const h2v s = __builtin_convertvector(sf, h2v);
    const h2v q = s * s;
    return x + ib * q;
}
__device__ __forceinline__ float snake_h(float x, float ea, float ib) {  // one element (out_conv_h1's staging loop)
    const h2v r = snake_h2(h2v{static_cast<_Float16>(x), static_cast<_Float16>(0.f)}, h2v{static_cast<_Float16>(ea), static_cast<_Float16>(0.f)},
                           h2v{static_cast<_Float16>(ib), static_cast<_Float16>(0.f)});
    return static_cast<float>(r[0]);
}

// KT = taps (a template parameter: the tap loop is unrolled and every weight fragment has a register of its own).
// Weights never touch LDS: a wave's A fragment of 16 output channels x 32 input channels is one 16-byte load per lane, 1 KiB
// contiguous per wave (the [tap][chunk][N][32] layout of model.cc attach_h1), served by the L2 -- a conv's whole weight set is
// at most 8 MB and every workgroup of a launch reads the same bytes. The fragments of ALL taps of a chunk live in registers
// (KT x CT x 4 VGPRs) and each is re-requested for the next chunk right after its last use, a whole chunk of MFMAs ahead of
// its next one. Only the input tile (+ causal halo) goes through LDS: one barrier pair per 32-channel CHUNK instead of one per
// tap. (The first build staged the weight tiles through LDS with a barrier per tap like conv_gemm_h2_kernel: with a third of
// that kernel's MFMAs per step the barriers were 85 % of the launch -- block0's k7 conv 3.48 ms against 4.5-4.8 ms.)
template <int BN, int KT, bool X32>
__global__ __launch_bounds__(256, 2) void conv_gemm_h1_kernel(ConvH1Args a) {
    constexpr int CT = BN / 32;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem1[];
    const int halo = (KT - 1) * a.dil;
    uint32_t* As = smem1;  // [(BM + halo)][RH]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int b, n_tile, m_tile;
    {   // XCD-contiguous tile order (codec_conv.hip conv_gemm_h2_kernel): bijective for any grid, a speed matter only
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
    const int rows = BM + halo;
    const int nchunks = (a.Cin + KC - 1) / KC;

    f32x4 acc[4][CT];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    // this lane's piece of the wave's weight fragments: row n_w + 16 c + (lane & 15) (clamped: rows past N are computed and never
    // stored), input channels 8 (lane >> 4) .. + 7 of the chunk
    const uint4* wlane[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int n = n0 + wn * (BN / 2) + 16 * c + (lane & 15);
        wlane[c] = reinterpret_cast<const uint4*>(a.w1 + (size_t)(n < a.N ? n : a.N - 1) * 32) + (lane >> 4);
    }
    const size_t wchunk = (size_t)a.N * 4;            // sixteen-byte pieces per (tap, chunk) block
    uint4 wf[KT][CT];
    auto load_w = [&](int tap, int chunk) {
#pragma unroll
        for (int c = 0; c < CT; ++c) wf[tap][c] = wlane[c][(size_t)(tap * nchunks + chunk) * wchunk];
    };

    // input tile: (BM + halo) rows x 4 pieces of 8 channels
    constexpr int AV = ((BM + MAX_HALO) * 4 + 255) / 256;
    uint4 areg[AV];
    float4 areg32[X32 ? AV : 1][2];
    const uint16_t* xh = reinterpret_cast<const uint16_t*>(a.x) + (size_t)b * a.x_bstride;
    const float* xf = reinterpret_cast<const float*>(a.x) + (size_t)b * a.x_bstride;
    auto load_a = [&](int chunk) {
        const int c0 = chunk * KC;
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            const int item = i * 256 + tid;
            const int r = item >> 2, c8 = (item & 3) * 8;
            const int t = t0 - halo + r;
            // clamped address, value masked when it is staged (hipcc waits on the spot for a load it has to predicate)
            const int tc = t < -a.hist ? -a.hist : (t < T ? t : T - 1);  // (hist: rows in front of x that hold the previous chunk's last rows)
            const int cc = c0 + c8 < a.Cin ? c0 + c8 : 0;
            if constexpr (X32) {
                const float* p = xf + (int64_t)tc * a.ldx + cc;
                areg32[i][0] = *reinterpret_cast<const float4*>(p);
                areg32[i][1] = *reinterpret_cast<const float4*>(p + 4);
            } else {
                areg[i] = *reinterpret_cast<const uint4*>(xh + (int64_t)tc * a.ldx + cc);
            }
        }
    };
    auto store_a = [&](int chunk) {
        const int c0 = chunk * KC;
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            const int item = i * 256 + tid;
            const int r = item >> 2, c8 = (item & 3) * 8;
            if (r >= rows) continue;
            const int t = t0 - halo + r;
            uint4 v;
            if constexpr (X32) {
                const uint2 lo = pack_h4(areg32[i][0].x, areg32[i][0].y, areg32[i][0].z, areg32[i][0].w);
                const uint2 hi = pack_h4(areg32[i][1].x, areg32[i][1].y, areg32[i][1].z, areg32[i][1].w);
                v = make_uint4(lo.x, lo.y, hi.x, hi.y);
            } else {
                v = areg[i];
            }
            if (t < -a.hist || t >= T || c0 + c8 >= a.Cin) v = make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4*>(&As[r * RH + (item & 3) * 4]) = v;
        }
    };

    load_a(0);
#pragma unroll
    for (int tap = 0; tap < KT; ++tap) load_w(tap, 0);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        __syncthreads();  // the previous chunk's MFMAs are done with As
        store_a(chunk);
        const int nxt = chunk + 1 < nchunks ? chunk + 1 : chunk;  // (the last chunk re-requests itself: unconditional loads)
        load_a(nxt);
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < KT; ++tap) {
            const uint32_t* arow = &As[(wm * 64 + tap * a.dil + (lane & 15)) * RH + 4 * (lane >> 4)];
            uint4 xa[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) xa[p] = *reinterpret_cast<const uint4*>(arow + p * 16 * RH);
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[p][c] = mfma_h(wf[tap][c], xa[p], acc[p][c]);
            load_w(tap, nxt);  // this tap's fragments for the next chunk: a whole chunk of MFMAs ahead of their use
        }
    }

    // ---- epilogue: the reference's op sequence on the packed float16 ALU, one rounding per op. Lane: channels n .. n + 3 of
    //      position t; `yh` keeps the finished float16 values for the SnakeBeta pass ----
    const int n_w = n0 + wn * (BN / 2), t_w = t0 + wm * 64;
    const int nq = n_w + 4 * (lane >> 4);
    h2v bh[CT][2];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int n = nq + 16 * c;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.bias && n < a.N) bv = *reinterpret_cast<const float4*>(a.bias + n);
        bh[c][0] = h2v{static_cast<_Float16>(bv.x), static_cast<_Float16>(bv.y)};  // (float16-exact values)
        bh[c][1] = h2v{static_cast<_Float16>(bv.z), static_cast<_Float16>(bv.w)};
    }
    uint2 rv[4][CT];
    if (a.res) {
        const uint16_t* rb = a.res + (size_t)b * a.res_bstride;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int t = t_w + 16 * p + (lane & 15);
            const int tc = t < T ? t : T - 1;
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const int n = nq + 16 * c;
                rv[p][c] = *reinterpret_cast<const uint2*>(rb + (size_t)tc * a.ldr + (n < a.N ? n : 0));
            }
        }
    }
    uint2 yh[4][CT];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int t = t_w + 16 * p + (lane & 15);
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int n = nq + 16 * c;
            const f32x2v a0 = {acc[p][c][0], acc[p][c][1]}, a1 = {acc[p][c][2], acc[p][c][3]};
            h2v v0 = __builtin_convertvector(a0, h2v), v1 = __builtin_convertvector(a1, h2v);  // conv(x, w)
            if (a.bias) { v0 = v0 + bh[c][0]; v1 = v1 + bh[c][1]; }                             // + bias
            if (a.res) {                                                                        // residual + h
                h2v r0, r1;
                __builtin_memcpy(&r0, &rv[p][c].x, 4);
                __builtin_memcpy(&r1, &rv[p][c].y, 4);
                v0 = r0 + v0;
                v1 = r1 + v1;
            }
            uint2 pk;
            __builtin_memcpy(&pk.x, &v0, 4);
            __builtin_memcpy(&pk.y, &v1, 4);
            yh[p][c] = pk;
            if (a.out && t < T && n < a.N) *reinterpret_cast<uint2*>(a.out + (size_t)b * a.out_bstride + (size_t)t * a.ldo + n) = pk;
        }
    }
    if (a.out2) {
        // Second output: SnakeBeta of the finished tile, for the conv that consumes it. The values are parked in LDS (each lane
        // its own slots) and walked by a rolled loop: one inlined sine body per 16-position slice (codec_conv.hip snake_pass).
        uint2* stash = reinterpret_cast<uint2*>(smem1) + wave * (CT * 64);
        uint2* par = reinterpret_cast<uint2*>(smem1) + 4 * (CT * 64);  // [BN / 4] ea as four halfs, then [BN / 4] ib
        __syncthreads();
        if (tid < BN / 4) {
            const int n = n0 + 4 * tid;
            if (n < a.N) {
                const int ch = n % a.post_C;  // transposed convs: n = phase * Cout + channel
                const float4 e = *reinterpret_cast<const float4*>(a.post_ea + ch), q = *reinterpret_cast<const float4*>(a.post_ib + ch);
                par[tid] = pack_h4(e.x, e.y, e.z, e.w);
                par[BN / 4 + tid] = pack_h4(q.x, q.y, q.z, q.w);
            }
        }
        __syncthreads();
        const int nl0 = wn * (BN / 2) + 4 * (lane >> 4);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#pragma unroll
            for (int c = 0; c < CT; ++c) stash[c * 64 + lane] = yh[p][c];
            const int t = t_w + p * 16 + (lane & 15);
            if (t >= T) continue;
            uint16_t* dst = a.out2 + (size_t)b * a.out_bstride + (size_t)t * a.ldo + n0;
#pragma unroll 1
            for (int c = 0; c < CT; ++c) {
                const int nl = nl0 + c * 16;
                if (n0 + nl >= a.N) break;
                const uint2 v = stash[c * 64 + lane], e = par[nl >> 2], q = par[BN / 4 + (nl >> 2)];
                h2v x0, x1, e0, e1, q0, q1;
                __builtin_memcpy(&x0, &v.x, 4); __builtin_memcpy(&x1, &v.y, 4);
                __builtin_memcpy(&e0, &e.x, 4); __builtin_memcpy(&e1, &e.y, 4);
                __builtin_memcpy(&q0, &q.x, 4); __builtin_memcpy(&q1, &q.y, 4);
                const h2v y0 = snake_h2(x0, e0, q0), y1 = snake_h2(x1, e1, q1);
                uint2 o;
                __builtin_memcpy(&o.x, &y0, 4);
                __builtin_memcpy(&o.y, &y1, 4);
                *reinterpret_cast<uint2*>(dst + nl) = o;
            }
        }
    }
}

// ---- a whole DecoderResidualUnit of the two narrow blocks in one launch -------------------------------------------------
// out = y + conv2(act2(conv1(act1(y))))  (SpeechTokenizer.swift:430-437) for C = 96 on float16 tensors (the template takes 192 too; see resunit_h1_supported). As separate
// launches the unit moves its tensor six times (act1(y) in, act2(conv1) out and back in, y in, y out, the activated copy out) and
// its pointwise half is HBM-bound (9.4 GB in 2.4 ms at 96 channels); here y is read once (+ halo; the raw tile a second time
// from the L2 for the residual) and the sum is written once. The structure is resunit_h2_kernel's (codec_conv.hip) with one
// plane: every wave owns 16 PT positions x ALL channels, act1 is applied while the raw tile is staged (one 32-channel chunk at a
// time, the only thing that goes through LDS), conv1's accumulators -- rounded and activated the way MLX rounds them -- ARE
// conv2's B fragments once conv2's weights are stored in the matching k order (model.cc attach_h1_perm: an accumulator pair of
// a lane holds 8 k values of its position), so conv1's output never leaves the registers. Weight fragments come per wave
// straight from the L2 (conv_gemm_h1_kernel), three steps deep in registers; the whole (chunk, tap) loop is unrolled so that
// the ring's slots are compile-time registers and hipcc's wait counts are exact. The sum goes to a second buffer because
// neighbouring workgroups still need the old halo rows. Rounding points: _main_decoder16's, op for op.
template <int CT2, int PT>
__global__ __launch_bounds__(256, 2) void resunit_h1_kernel(ResUnitH1Args a) {
    constexpr int C = 16 * CT2, NCH = CT2 / 2, PW = 16 * PT, BMU = 4 * PW, KT = 7, S1 = NCH * KT;
    constexpr int D = CT2 >= 12 ? 2 : 3;  // ring depth in steps: 12 fragments per step at 192 channels leave room for two
    constexpr int CG = 6, NG = CT2 / CG;  // conv2 in passes of six output-channel tiles
    static_assert(CT2 % CG == 0 && CT2 % 2 == 0, "channel tiles must split into passes and into 32-channel chunks");
    extern __shared__ __attribute__((aligned(16))) uint32_t smem1[];
    const int halo = (KT - 1) * a.dil;
    uint32_t* As = smem1;  // [(BMU + halo)][RH]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z;
    const int t0 = blockIdx.y * BMU;
    const int T = a.frames[b] * a.ppf;
    if (t0 >= T) return;
    const int rows = BMU + halo;
    const size_t boff = (size_t)b * a.Tmax * C;
    const uint16_t* yb = a.y + boff;

    f32x4 acc1[PT][CT2];
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int c = 0; c < CT2; ++c) acc1[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    // conv1's fragments: (tap, chunk) blocks of C rows x 4 sixteen-byte pieces; this lane: row 16 c + (lane & 15), piece lane >> 4
    const uint4* w1l = reinterpret_cast<const uint4*>(a.w1) + (size_t)(lane & 15) * 4 + (lane >> 4);
    uint4 wf[D][CT2];
    auto load_w1 = [&](int slot, int step) {
        const int chunk = step / KT, tap = step - chunk * KT;
        const uint4* src = w1l + (size_t)(tap * NCH + chunk) * C * 4;
#pragma unroll
        for (int c = 0; c < CT2; ++c) wf[slot][c] = src[c * 64];
    };

    // input tile, one 32-channel chunk at a time: (BMU + halo) rows x 4 pieces of 8 channels
    constexpr int AV = ((BMU + MAX_HALO) * 4 + 255) / 256;
    uint4 areg[AV];
    auto load_a = [&](int chunk) {
        const int c0 = chunk * KC;
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            const int item = i * 256 + tid;
            const int r = item >> 2, c8 = (item & 3) * 8;
            const int t = t0 - halo + r;
            const int tc = t < -a.hist ? -a.hist : (t < T ? t : T - 1);  // clamped address, value masked when it is staged
            areg[i] = *reinterpret_cast<const uint4*>(yb + (int64_t)tc * C + c0 + c8);
        }
    };
    auto store_a = [&](int chunk) {
        const int c0 = chunk * KC, c8 = (tid & 3) * 8;  // a thread's pieces all sit in the same eight channels
        const float4 e0 = *reinterpret_cast<const float4*>(a.ea1 + c0 + c8), e1 = *reinterpret_cast<const float4*>(a.ea1 + c0 + c8 + 4);
        const float4 i0 = *reinterpret_cast<const float4*>(a.ib1 + c0 + c8), i1 = *reinterpret_cast<const float4*>(a.ib1 + c0 + c8 + 4);
        const uint2 eh0 = pack_h4(e0.x, e0.y, e0.z, e0.w), eh1 = pack_h4(e1.x, e1.y, e1.z, e1.w);
        const uint2 ih0 = pack_h4(i0.x, i0.y, i0.z, i0.w), ih1 = pack_h4(i1.x, i1.y, i1.z, i1.w);
        const uint32_t ew[4] = {eh0.x, eh0.y, eh1.x, eh1.y}, iw[4] = {ih0.x, ih0.y, ih1.x, ih1.y};
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            const int item = i * 256 + tid;
            const int r = item >> 2;
            if (r >= rows) continue;
            const int t = t0 - halo + r;
            uint32_t v[4] = {areg[i].x, areg[i].y, areg[i].z, areg[i].w};
            if (t >= -a.hist && t < T) {  // act1 on the raw tile (rows outside the sequence stay zero: causal padding is not activated)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    h2v x, e, q;
                    __builtin_memcpy(&x, &v[j], 4);
                    __builtin_memcpy(&e, &ew[j], 4);
                    __builtin_memcpy(&q, &iw[j], 4);
                    const h2v y = snake_h2(x, e, q);
                    __builtin_memcpy(&v[j], &y, 4);
                }
            } else {
                v[0] = v[1] = v[2] = v[3] = 0u;
            }
            *reinterpret_cast<uint4*>(&As[r * RH + (item & 3) * 4]) = make_uint4(v[0], v[1], v[2], v[3]);
        }
    };

    // ---- conv1: fully unrolled over (chunk, tap); fragments of step s + D - 1 are requested at the top of step s ----
    load_a(0);
#pragma unroll
    for (int s = 0; s < D - 1; ++s) load_w1(s, s);
#pragma unroll
    for (int chunk = 0; chunk < NCH; ++chunk) {
        __syncthreads();  // the previous chunk's MFMAs are done with As
        store_a(chunk);
        if (chunk + 1 < NCH) load_a(chunk + 1);
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < KT; ++tap) {
            const int step = chunk * KT + tap;
            if (step + D - 1 < S1) load_w1((step + D - 1) % D, step + D - 1);
            const uint32_t* arow = &As[(PW * wave + tap * a.dil + (lane & 15)) * RH + 4 * (lane >> 4)];
            uint4 xa[PT];
#pragma unroll
            for (int p = 0; p < PT; ++p) xa[p] = *reinterpret_cast<const uint4*>(arow + p * 16 * RH);
#pragma unroll
            for (int p = 0; p < PT; ++p)
#pragma unroll
                for (int c = 0; c < CT2; ++c) acc1[p][c] = mfma_h(wf[step % D][c], xa[p], acc1[p][c]);
        }
    }

    // ---- fp16(acc), + bias1, act2 (the reference's op order), packed into conv2's B fragments: pair (2m, 2m + 1) of a lane ----
    const int q4 = 4 * (lane >> 4);
    uint4 xb[PT][NCH];
#pragma unroll
    for (int m = 0; m < NCH; ++m) {
        uint32_t bw[2][2], ew[2][2], iw[2][2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int n = 16 * (2 * m + e) + q4;
            const float4 bv = a.b1 ? *reinterpret_cast<const float4*>(a.b1 + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 ev = *reinterpret_cast<const float4*>(a.ea2 + n), iv = *reinterpret_cast<const float4*>(a.ib2 + n);
            const uint2 bp = pack_h4(bv.x, bv.y, bv.z, bv.w), ep = pack_h4(ev.x, ev.y, ev.z, ev.w), ip = pack_h4(iv.x, iv.y, iv.z, iv.w);
            bw[e][0] = bp.x; bw[e][1] = bp.y; ew[e][0] = ep.x; ew[e][1] = ep.y; iw[e][0] = ip.x; iw[e][1] = ip.y;
        }
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            uint32_t o[4];
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const f32x2v av = {acc1[p][2 * m + e][2 * hh], acc1[p][2 * m + e][2 * hh + 1]};
                    h2v v = __builtin_convertvector(av, h2v);  // conv1(x, w)
                    h2v bh, eh, ih;
                    __builtin_memcpy(&bh, &bw[e][hh], 4);
                    __builtin_memcpy(&eh, &ew[e][hh], 4);
                    __builtin_memcpy(&ih, &iw[e][hh], 4);
                    if (a.b1) v = v + bh;                       // + bias
                    v = snake_h2(v, eh, ih);                    // act2
                    __builtin_memcpy(&o[2 * e + hh], &v, 4);
                }
            xb[p][m] = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }

    // ---- conv2 in NG passes of CG output-channel tiles; fragments of chunk m + 1 are requested while chunk m multiplies ----
    const uint4* w2l = reinterpret_cast<const uint4*>(a.w2p) + (size_t)(lane & 15) * 4 + (lane >> 4);
    __syncthreads();  // every wave is done with As: the SnakeBeta pass below parks values there
    uint2* stash = reinterpret_cast<uint2*>(smem1) + wave * (CG * 64);
#pragma unroll
    for (int h = 0; h < NG; ++h) {
        f32x4 acc2[PT][CG];
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int c = 0; c < CG; ++c) acc2[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        uint4 w2[2][CG];
        auto load_w2 = [&](int slot, int m) {
#pragma unroll
            for (int c = 0; c < CG; ++c) w2[slot][c] = w2l[((size_t)m * C + 16 * (h * CG + c)) * 4];
        };
        load_w2(0, 0);
#pragma unroll
        for (int m = 0; m < NCH; ++m) {
            if (m + 1 < NCH) load_w2((m + 1) & 1, m + 1);
#pragma unroll
            for (int p = 0; p < PT; ++p)
#pragma unroll
                for (int c = 0; c < CG; ++c) acc2[p][c] = mfma_h(w2[m & 1][c], xb[p][m], acc2[p][c]);
        }
        // ---- this pass's channels: fp16(acc), + bias2, residual + . -> out; optionally the next block's SnakeBeta -> out2 ----
        uint2 rvs[PT][CG];
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            const int t = t0 + PW * wave + 16 * p + (lane & 15);
            const int tc = t < T ? t : T - 1;
#pragma unroll
            for (int c = 0; c < CG; ++c) rvs[p][c] = *reinterpret_cast<const uint2*>(yb + (size_t)tc * C + 16 * (h * CG + c) + q4);
        }
        uint2 yh[PT][CG];
#pragma unroll
        for (int c = 0; c < CG; ++c) {
            const int n = 16 * (h * CG + c) + q4;
            const float4 bv = a.b2 ? *reinterpret_cast<const float4*>(a.b2 + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            const h2v b0 = {static_cast<_Float16>(bv.x), static_cast<_Float16>(bv.y)}, b1h = {static_cast<_Float16>(bv.z), static_cast<_Float16>(bv.w)};
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                const int t = t0 + PW * wave + 16 * p + (lane & 15);
                const f32x2v a0 = {acc2[p][c][0], acc2[p][c][1]}, a1 = {acc2[p][c][2], acc2[p][c][3]};
                h2v v0 = __builtin_convertvector(a0, h2v), v1 = __builtin_convertvector(a1, h2v);  // conv2(x, w)
                if (a.b2) { v0 = v0 + b0; v1 = v1 + b1h; }                                         // + bias
                h2v r0, r1;
                __builtin_memcpy(&r0, &rvs[p][c].x, 4);
                __builtin_memcpy(&r1, &rvs[p][c].y, 4);
                v0 = r0 + v0;                                                                      // residual + h
                v1 = r1 + v1;
                uint2 pk;
                __builtin_memcpy(&pk.x, &v0, 4);
                __builtin_memcpy(&pk.y, &v1, 4);
                yh[p][c] = pk;
                if (t < T) *reinterpret_cast<uint2*>(a.out + boff + (size_t)t * C + n) = pk;
            }
        }
        if (a.out2) {
#pragma unroll
            for (int p = 0; p < PT; ++p) {
#pragma unroll
                for (int c = 0; c < CG; ++c) stash[c * 64 + lane] = yh[p][c];
                const int t = t0 + PW * wave + 16 * p + (lane & 15);
                if (t >= T) continue;
#pragma unroll 1
                for (int c = 0; c < CG; ++c) {
                    const int n = 16 * (h * CG + c) + q4;
                    const uint2 v = stash[c * 64 + lane];
                    const float4 ev = *reinterpret_cast<const float4*>(a.post_ea + n), iv = *reinterpret_cast<const float4*>(a.post_ib + n);
                    const uint2 ep = pack_h4(ev.x, ev.y, ev.z, ev.w), ip = pack_h4(iv.x, iv.y, iv.z, iv.w);
                    h2v x0, x1, e0, e1, i0, i1;
                    __builtin_memcpy(&x0, &v.x, 4); __builtin_memcpy(&x1, &v.y, 4);
                    __builtin_memcpy(&e0, &ep.x, 4); __builtin_memcpy(&e1, &ep.y, 4);
                    __builtin_memcpy(&i0, &ip.x, 4); __builtin_memcpy(&i1, &ip.y, 4);
                    const h2v y0 = snake_h2(x0, e0, i0), y1 = snake_h2(x1, e1, i1);
                    uint2 o;
                    __builtin_memcpy(&o.x, &y0, 4);
                    __builtin_memcpy(&o.y, &y1, 4);
                    *reinterpret_cast<uint2*>(a.out2 + boff + (size_t)t * C + n) = o;
                }
            }
        }
    }
}

// out[t] = clip(fp16(fp16(sum_{k,c} snake(x[t-6+k][c]) * w[k][c]) + bias)): the MainDecoder's tail on a float16 tensor
// (SpeechTokenizer.swift:687-688, 781). 64 positions per workgroup, 4 lanes each, activated tile and taps in LDS (the layout of
// codec_misc.hip out_conv_kernel).
__global__ __launch_bounds__(256) void out_conv_h1_kernel(const uint16_t* x, int C, const float* ea, const float* ib, const float* w,
                                                          const float* bias, const int32_t* frames, int ppf, int Tmax, float* pcm,
                                                          int32_t* nonfinite, int hist) {
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [(64 + 6)][C + 4] snake(x), then [7][C] taps
    const int ld = C + 4, C4 = C >> 2;
    float* ws = xs + 70 * ld;
    const int b = blockIdx.y, t0 = blockIdx.x * 64;
    const int T = frames[b] * ppf;
    if (t0 >= T) return;
    const uint16_t* xb = x + (size_t)b * Tmax * C;
    for (int i = threadIdx.x; i < 7 * C4; i += 256) *reinterpret_cast<float4*>(ws + 4 * i) = *reinterpret_cast<const float4*>(w + 4 * i);
    for (int i = threadIdx.x; i < 70 * C4; i += 256) {
        const int r = i / C4, c4 = i % C4;
        const int t = t0 - 6 + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t >= -hist && t < T) {
            float xv[4];
            unpack_h4(*reinterpret_cast<const uint2*>(xb + (int64_t)t * C + 4 * c4), xv);
            const float4 e = *reinterpret_cast<const float4*>(ea + 4 * c4), q = *reinterpret_cast<const float4*>(ib + 4 * c4);
            v = make_float4(snake_h(xv[0], e.x, q.x), snake_h(xv[1], e.y, q.y), snake_h(xv[2], e.z, q.z), snake_h(xv[3], e.w, q.w));
        }
        *reinterpret_cast<float4*>(xs + r * ld + 4 * c4) = v;
    }
    __syncthreads();
    const int pos = threadIdx.x >> 2, sub = threadIdx.x & 3;
    float acc = 0.f;
    for (int k = 0; k < 7; ++k) {
        const float* xr = xs + (pos + k) * ld;
        const float* wr = ws + k * C;
        for (int c4 = sub; c4 < C4; c4 += 4) {
            const float4 xv = *reinterpret_cast<const float4*>(xr + 4 * c4), wv = *reinterpret_cast<const float4*>(wr + 4 * c4);
            acc += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
        }
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    const int t = t0 + pos;
    if (sub == 0 && t < T) {
        float v = r16(acc);
        if (bias) v = r16(v + bias[0]);
        if (nonfinite && !(fabsf(v) <= 3.0e38f)) atomicOr(reinterpret_cast<unsigned int*>(nonfinite + b), 1u);
        pcm[(size_t)b * Tmax + t] = fminf(fmaxf(v, -1.0f), 1.0f);
    }
}

}  // namespace

void launch_conv_gemm_h1(const ConvH1Args& a, hipStream_t st) {
    Q3_CHECK((a.K - 1) * a.dil <= MAX_HALO, 3, "conv_gemm_h1: receptive field too large");
    Q3_CHECK(a.Cin % 8 == 0 && a.N % 4 == 0 && a.ldx % 8 == 0 && a.ldo % 4 == 0, 3, "conv_gemm_h1: channel counts must be multiples of 8 / 4");
    Q3_CHECK(!a.out2 || (a.post_ea && a.post_ib && a.post_C > 0 && a.post_C % 4 == 0), 3, "conv_gemm_h1: activated output without its parameters");
    const int mt = (a.Tmax + BM - 1) / BM;
    if (mt <= 0 || a.B <= 0) return;
    int BN = 64;
    if (a.N % 128 == 0) BN = 128;
    else if (a.N % 96 == 0) BN = 96;
    else if (a.N > 128 && (a.N % 64) != 0) BN = 128;
    const dim3 grid((a.N + BN - 1) / BN, mt, a.B), block(256);
    // LDS: the input tile, or the SnakeBeta pass's stash (4 waves x CT x 64 uint2) + parameters, whichever is larger
    const size_t tiles = size_t(BM + (a.K - 1) * a.dil) * RH * sizeof(uint32_t);
    const size_t pass = size_t(4 * (BN / 32) * 64 + 2 * (BN / 4)) * sizeof(uint2);
    const size_t smem = std::max(tiles, pass);
    Q3_CHECK(smem <= 64 * 1024, 3, "conv_gemm_h1: tile does not fit the static LDS limit");
    Q3_CHECK(a.K == 1 || a.K == 2 || a.K == 7, 3, "conv_gemm_h1: 1, 2 or 7 taps (the MainDecoder's convs)");
    Q3_CHECK(!a.x_f32 || (a.K == 7 && BN == 128), 3, "conv_gemm_h1: the fp32-input form is initConv's");
#define Q3_H1(BNv)                                                                                              \
    do {                                                                                                        \
        if (a.x_f32) hipLaunchKernelGGL((conv_gemm_h1_kernel<128, 7, true>), grid, block, smem, st, a);         \
        else if (a.K == 7) hipLaunchKernelGGL((conv_gemm_h1_kernel<BNv, 7, false>), grid, block, smem, st, a);  \
        else if (a.K == 2) hipLaunchKernelGGL((conv_gemm_h1_kernel<BNv, 2, false>), grid, block, smem, st, a);  \
        else hipLaunchKernelGGL((conv_gemm_h1_kernel<BNv, 1, false>), grid, block, smem, st, a);                \
    } while (0)
    switch (BN) {
        case 128: Q3_H1(128); break;
        case 96: Q3_H1(96); break;
        default: Q3_H1(64); break;
    }
#undef Q3_H1
}

// Measured at 32 x 200 frames (profiles/r04_codec_f16_launches.txt): at 96 channels the fused unit takes 3.2-3.4 ms against 2.72 +
// 2.43 for its two convs as launches. At 192 channels the <12, 2> form (32 positions per wave: acc1 for 64 would not fit) took
// 5.22 ms against 2.77 + 1.90: every wave fetches twelve weight fragments per 24 MFMAs, twice the L2 traffic per MFMA of the other
// kernels, four waves fetching the same ones -- not instantiated; the 192-channel units stay two launches each.
bool resunit_h1_supported(int C, int K, int dil) { return C == 96 && K == 7 && 6 * dil <= MAX_HALO; }

void launch_resunit_h1(const ResUnitH1Args& a, hipStream_t st) {
    Q3_CHECK(resunit_h1_supported(a.C, 7, a.dil) && a.out != a.y, 3, "resunit_h1: unsupported geometry");
    Q3_CHECK(a.w1 && a.w2p && a.ea1 && a.ib1 && a.ea2 && a.ib2, 3, "resunit_h1: incomplete float16 weights");
    Q3_CHECK(!a.out2 || (a.post_ea && a.post_ib), 3, "resunit_h1: activated output without its parameters");
    if (a.Tmax <= 0 || a.B <= 0) return;
    const int bmu = a.C == 96 ? 256 : 128;  // positions per workgroup (64 PT)
    // LDS: the input tile of one chunk, or the SnakeBeta pass's stash (4 waves x 6 tiles x 64 uint2), whichever is larger
    const size_t smem = std::max(size_t(bmu + 6 * a.dil) * RH * sizeof(uint32_t), size_t(4 * 6 * 64) * sizeof(uint2));
    const dim3 grid(1, (a.Tmax + bmu - 1) / bmu, a.B), block(256);
    hipLaunchKernelGGL((resunit_h1_kernel<6, 4>), grid, block, smem, st, a);
}

void launch_out_conv_h1(const uint16_t* x, int C, const float* ea, const float* ib, const float* w, const float* bias,
                        const int32_t* frames, int ppf, int Tmax, int B, float* pcm, hipStream_t st, int32_t* nonfinite, int hist) {
    const size_t smem = size_t(70 * (C + 4) + 7 * C) * sizeof(float);
    Q3_CHECK(smem <= 64 * 1024 && C % 4 == 0 && C >= 4 && C <= 1024, 3, "out_conv_h1: unsupported channel count");
    hipLaunchKernelGGL(out_conv_h1_kernel, dim3((Tmax + 63) / 64, B), dim3(256), smem, st, x, C, ea, ib, w, bias, frames, ppf, Tmax, pcm,
                       nonfinite, hist);
}

}  // namespace q3
