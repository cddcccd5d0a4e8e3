// common.h -- shared host/device helpers for the q3tts HIP engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>

namespace q3 {

struct Error : std::runtime_error {
    int status;  // q3tts_status
    Error(int st, const std::string& msg) : std::runtime_error(msg), status(st) {}
};

#define Q3_HIP(expr)                                                                           \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            throw ::q3::Error(7, std::string("HIP error: ") + hipGetErrorString(_e) + " at " + \
                                     __FILE__ + ":" + std::to_string(__LINE__) + " (" #expr ")"); \
    } while (0)

#define Q3_CHECK(cond, st, msg)                         \
    do {                                                \
        if (!(cond)) throw ::q3::Error((st), (msg));    \
    } while (0)

using bf16_t = uint16_t;  // raw bf16 bit pattern everywhere on the host side

// host-side bf16 conversion (round to nearest even, NaN preserved)
inline float bf16_to_f32_host(bf16_t h) {
    uint32_t u = uint32_t(h) << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
inline bf16_t f32_to_bf16_host(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return bf16_t((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return bf16_t(u >> 16);
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Fragment-major ("tiled") activation layout. Every GEMM x operand [rows][K] (K % 128 == 0) is
// stored so that the 64 lanes of a wave read one MFMA B fragment as 1 KiB contiguous:
//   block (kc = k/128, mb = m/16, i = (k/8)%4) holds [lane = ((k/32)%4)*16 + m%16][k%8].
// MBL = row blocks of the allocation (padded batch / 16). Element offset of (m, k):
#if defined(__HIPCC__)
__host__ __device__
#endif
inline size_t act_tiled_offset(int m, int k, int MBL) {
    return (((size_t(k >> 7) * MBL + (m >> 4)) * 4 + ((k >> 3) & 3)) * 64 + (((k >> 5) & 3) * 16 + (m & 15))) * 8 + (k & 7);
}
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace q3

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
#if defined(__HIPCC__)
namespace q3 {

__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float(uint32_t(h) << 16); }
// fp32 -> bf16, round to nearest even. A plain cast compiles to v_cvt_pk_bf16_f32 on gfx950 (one VALU op for
// two values); an integer-arithmetic RNE costs ~10 ops per value and made the RMSNorm prologue VALU-bound.
// Same results as the oracle's f2bf for every non-NaN input (NaN stays NaN).
using bf16x2_t = __attribute__((ext_vector_type(2))) __bf16;
using f32x2_t = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ uint16_t f2bf(float f) { return __builtin_bit_cast(uint16_t, static_cast<__bf16>(f)); }
__device__ __forceinline__ float rbf(float f) { return static_cast<float>(static_cast<__bf16>(f)); }
__device__ __forceinline__ float lo_bf(uint32_t packed) { return __uint_as_float(packed << 16); }
__device__ __forceinline__ float hi_bf(uint32_t packed) { return __uint_as_float(packed & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack_bf(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

using bf16x8 = __attribute__((ext_vector_type(8))) short;  // MFMA bf16 A/B fragment (4 VGPRs)
using f32x4 = __attribute__((ext_vector_type(4))) float;   // 16x16 accumulator fragment
using f32x16 = __attribute__((ext_vector_type(16))) float; // 32x32 accumulator fragment

}  // namespace q3
#endif
