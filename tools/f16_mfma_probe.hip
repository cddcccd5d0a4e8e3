// Does v_mfma_f32_16x16x32_f16 honour fp16 subnormal A/B inputs on gfx950? And what does a two-plane fp16 split
// (hi + 2^-11 lo') of fp32 operands cost in accuracy against a double reference, next to the three-plane bf16 split?
// hipcc --offload-arch=gfx950 -O2 tools/f16_mfma_probe.hip -o tools/f16_mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const _Float16* a, const _Float16* b, float* d) {
    // A[row l&15][k = 8(l>>4)+j], B[k][col l&15]; D col = lane&15, row = 4(lane>>4)+reg
    const int lane = threadIdx.x;
    f16x8 av, bv;
    for (int j = 0; j < 8; ++j) {
        av[j] = a[(lane & 15) * 32 + 8 * (lane >> 4) + j];
        bv[j] = b[(8 * (lane >> 4) + j) * 16 + (lane & 15)];
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[(4 * (lane >> 4) + r) * 16 + (lane & 15)] = c[r];
}

int main() {
    std::vector<_Float16> a(16 * 32, (_Float16)0.f), b(32 * 16, (_Float16)0.f);
    // row 0: a subnormal (2^-20) times 2^10 -> 2^-10 if honoured, 0 if flushed
    a[0 * 32 + 0] = (_Float16)9.5367431640625e-07f;
    b[0 * 16 + 0] = (_Float16)1024.f;
    // row 1 col 1: subnormal B operand
    a[1 * 32 + 0] = (_Float16)1024.f;
    b[0 * 16 + 1] = (_Float16)9.5367431640625e-07f;
    // row 2 col 2: normal reference
    a[2 * 32 + 0] = (_Float16)0.5f;
    b[0 * 16 + 2] = (_Float16)0.25f;
    _Float16 *da, *db;
    float* dd;
    hipMalloc(&da, a.size() * 2); hipMalloc(&db, b.size() * 2); hipMalloc(&dd, 256 * 4);
    hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), b.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dd);
    std::vector<float> d(256);
    hipMemcpy(d.data(), dd, 256 * 4, hipMemcpyDeviceToHost);
    printf("subnormal A x 1024 = %g (honoured: %g)\n", d[0 * 16 + 0], 9.5367431640625e-07 * 1024);
    printf("1024 x subnormal B = %g (honoured: %g)\n", d[1 * 16 + 1], 9.5367431640625e-07 * 1024);
    printf("0.5 x 0.25 = %g\n", d[2 * 16 + 2]);
    return 0;
}
