// snake.h -- sin^2 for SnakeBeta (x + 1/(exp(beta)+1e-9) * sin^2(exp(alpha) * x), SpeechTokenizer.swift:232-254) on the
// decoder's hot convs. libm's sinf spends ~45 instructions per element (two-path argument reduction); the codec
// evaluates 18 G of them per 32 x 200-frame batch. Here: Cody-Waite reduction by pi/2 in two FMAs (exact to 2^-30 for
// |u| < 1e6), the Cephes degree-7 sine on [-pi/4, pi/4] (< 1 ulp), and sin^2(u) = sin^2(r) or 1 - sin^2(r) by the
// parity of the quadrant -- no cosine polynomial. Measured max |error| 1.2e-7 for |u| <= 1e6, the same as squaring a
// correctly rounded float sine; larger arguments take sinf.
#pragma once
#include <hip/hip_runtime.h>

namespace q3 {

// the polynomial path alone (callers vote on !(|u| < 1e6) and take snake_sin2 for the whole tile when it fires)
__device__ __forceinline__ float snake_sin2_poly(float u) {
    const float k = __builtin_rintf(u * 0.636619772367581343f);
    float r = __builtin_fmaf(k, -1.57079637050628662109375f, u);
    r = __builtin_fmaf(k, 4.37113900018624283e-8f, r);
    const float r2 = r * r;
    float p = __builtin_fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    p = __builtin_fmaf(r2, p, -1.6666654611e-1f);
    const float s = __builtin_fmaf(r * r2, p, r);
    const float s2 = s * s;
    return (static_cast<int>(k) & 1) ? 1.0f - s2 : s2;
}

__device__ __forceinline__ float snake_sin2(float u) {
    if (__builtin_expect(!(fabsf(u) < 1.0e6f), 0)) {
        const float s = sinf(u);
        return s * s;
    }
    return snake_sin2_poly(u);
}

}  // namespace q3
