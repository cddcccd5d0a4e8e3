#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
template <int CTRL, int BANK = 0xF>
__device__ __forceinline__ float dpp_mov(float old, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, 0xF, BANK, false));
}
__device__ __forceinline__ float xor32(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
    return __int_as_float(r[0]) + __int_as_float(r[1]);
}
__device__ __forceinline__ float xor16(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(v), __float_as_int(v), false, false);
    return __int_as_float(r[0]) + __int_as_float(r[1]);
}
__device__ __forceinline__ float wave_sum2(float v) {
    v = xor32(v);
    v = xor16(v);
    v += dpp_mov<0x128>(v, v);                               // row_ror:8  = xor 8
    { float t = dpp_mov<0x104, 0x5>(v, v); t = dpp_mov<0x114, 0xA>(t, v); v += t; }  // row_shl:4 on banks 0,2; row_shr:4 on banks 1,3 = xor 4
    v += dpp_mov<0x4E>(v, v);                                // quad_perm [2,3,0,1] = xor 2
    v += dpp_mov<0xB1>(v, v);                                // quad_perm [1,0,3,2] = xor 1
    return v;
}
__device__ __forceinline__ float wave_sum1(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__global__ void k(const float* x, float* y, float* z) {
    float v = x[threadIdx.x];
    y[threadIdx.x] = wave_sum1(v);
    z[threadIdx.x] = wave_sum2(v);
}
int main() {
    float *x, *y, *z; (void)hipMalloc(&x, 256); (void)hipMalloc(&y, 256); (void)hipMalloc(&z, 256);
    float h[64]; for (int i = 0; i < 64; ++i) h[i] = 1.0f / (i + 3) * ((i * 7) % 5 - 2.3f);
    (void)hipMemcpy(x, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(x, y, z);
    float a[64], b[64]; (void)hipMemcpy(a, y, 256, hipMemcpyDeviceToHost); (void)hipMemcpy(b, z, 256, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 64; ++i) if (memcmp(&a[i], &b[i], 4)) ++bad;
    printf("bad %d  %.9g %.9g\n", bad, a[0], b[0]);
    return bad != 0;
}
