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

// per-device one-time setup (hipFuncSetAttribute and the like): the flag of the CURRENT device out of an array of kMaxDevices
constexpr int kMaxDevices = 64;
template <class Flag>
inline Flag& device_once(Flag (&flags)[kMaxDevices]) {
    int dev = 0;
    Q3_HIP(hipGetDevice(&dev));
    Q3_CHECK(dev >= 0 && dev < kMaxDevices, 7, "device ordinal beyond kMaxDevices");
    return flags[dev];
}

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

// Cross-lane reductions on the VALU (DPP row operations + gfx950's v_permlane{16,32}_swap) instead of ds_bpermute: the
// LDS crossbar costs a full LDS round trip per butterfly step, and the decode kernels are chains of such steps. Every
// function pairs exactly the lanes the xor butterfly it replaces paired (i with i ^ o), so results are bit-identical
// to the __shfl_xor forms (checked on the GPU by tools/dpp_reduce.hip).
template <int CTRL, int BANK = 0xF>
__device__ __forceinline__ float dpp_mov(float old, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, 0xF, BANK, false));
}
struct XorPartner {  // v's value in lane i ^ o for o = 32, 16, 8, 4, 2, 1 (in that order)
    static __device__ __forceinline__ void swap32(float v, float& a, float& b) {
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
        a = __int_as_float(r[0]); b = __int_as_float(r[1]);  // {lo, lo}, {hi, hi}: a op b = v[i] op v[i ^ 32] for a commutative op
    }
    static __device__ __forceinline__ void swap16(float v, float& a, float& b) {
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(v), __float_as_int(v), false, false);
        a = __int_as_float(r[0]); b = __int_as_float(r[1]);
    }
    static __device__ __forceinline__ float x8(float v) { return dpp_mov<0x128>(v, v); }  // row_ror:8
    static __device__ __forceinline__ float x4(float v) {                                   // row_shl:4 | row_shr:4 by bank
        const float t = dpp_mov<0x104, 0x5>(v, v);
        return dpp_mov<0x114, 0xA>(t, v);
    }
    static __device__ __forceinline__ float x2(float v) { return dpp_mov<0x4E>(v, v); }   // quad_perm [2,3,0,1]
    static __device__ __forceinline__ float x1(float v) { return dpp_mov<0xB1>(v, v); }   // quad_perm [1,0,3,2]
};
__device__ __forceinline__ float wave_sum(float v) {  // butterfly o = 32, 16, ..., 1; every lane gets the sum
    float a, b;
    XorPartner::swap32(v, a, b); v = a + b;
    XorPartner::swap16(v, a, b); v = a + b;
    v += XorPartner::x8(v);
    v += XorPartner::x4(v);
    v += XorPartner::x2(v);
    v += XorPartner::x1(v);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    float a, b;
    XorPartner::swap32(v, a, b); v = fmaxf(a, b);
    XorPartner::swap16(v, a, b); v = fmaxf(a, b);
    v = fmaxf(v, XorPartner::x8(v));
    v = fmaxf(v, XorPartner::x4(v));
    v = fmaxf(v, XorPartner::x2(v));
    v = fmaxf(v, XorPartner::x1(v));
    return v;
}
// sum over a 16-lane DPP row, butterfly o = 1, 2, 4, 8 (the attention kernels' per-position dot products). After the
// first two steps every quad is uniform, so the mirrors pair the same VALUES as xor 4 and xor 8 would.
__device__ __forceinline__ float row16_sum(float v) {
    v += XorPartner::x1(v);
    v += XorPartner::x2(v);
    v += dpp_mov<0x141>(v, v);  // row_half_mirror
    v += dpp_mov<0x140>(v, v);  // row_mirror
    return v;
}

using bf16x8 = __attribute__((ext_vector_type(8))) short;  // MFMA bf16 A/B fragment (4 VGPRs)
// 16-byte global load; NT = non-temporal (`global_load_dwordx4 ... nt`): for bytes that are read once per frame step out
// of a working set far larger than the 256 MB Infinity Cache (the talker's layer weights, its KV cache). The hint has to
// be a compile-time property of the load: hipcc merges the two arms of a run-time select into one plain load.
template <bool NT>
__device__ __forceinline__ uint4 ld16(const uint4* p) {
    using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
    if constexpr (NT) {
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
        return make_uint4(v.x, v.y, v.z, v.w);
    } else {
        return *p;
    }
}
using f32x4 = __attribute__((ext_vector_type(4))) float;   // 16x16 accumulator fragment
using f32x16 = __attribute__((ext_vector_type(16))) float; // 32x32 accumulator fragment

}  // namespace q3
#endif
