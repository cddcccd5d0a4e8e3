// l2keep.hip -- does data read by one kernel stay in the XCD L2s for the next kernel of the same stream?
// Decides whether "kernel k also requests kernel k+1's weight tiles" can hide the HBM fetch of the decode GEMMs.
//   hipcc --offload-arch=gfx950 -O3 -o tools/l2keep tools/l2keep.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// workgroup g reads `per_wg` bytes at base + g * per_wg (same g -> same XCD in every launch), writes 4 bytes
__global__ __launch_bounds__(512) void reader(const uint4* base, size_t per_wg_vec, float* out) {
    const uint4* p = base + (size_t)blockIdx.x * per_wg_vec;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t i = threadIdx.x; i < per_wg_vec; i += 512) {
        const uint4 v = p[i];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[blockIdx.x] = 1.f;
}

int main() {
    const int G = 256;
    float* out;
    CK(hipMalloc(&out, G * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (size_t per_wg : {size_t(16) << 10, size_t(32) << 10, size_t(64) << 10, size_t(128) << 10}) {
        const size_t bytes = per_wg * G;
        const int NB = 64;  // distinct buffers: 64 x (4..32 MB) defeats L2 and MALL for the "cold" case
        uint4* buf;
        CK(hipMalloc(&buf, bytes * NB));
        CK(hipMemset(buf, 1, bytes * NB));
        for (int mode = 0; mode < 3; ++mode) {
            // mode 0: every launch reads a different buffer (cold); 1: the same buffer every time (hot in L2 if kept);
            // 2: pairs (A, A, B, B, ...): every second launch re-reads what the previous one read
            const int iters = 400;
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < iters; ++i) {
                const int b = mode == 0 ? i % NB : mode == 1 ? 0 : (i / 2) % NB;
                hipLaunchKernelGGL(reader, dim3(G), dim3(512), 0, 0, buf + (size_t)b * (bytes / 16), per_wg / 16, out);
            }
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("per-WG %4zu KiB (total %5.1f MB) %s: %.2f us per launch\n", per_wg >> 10, bytes / 1e6,
                   mode == 0 ? "different buffer each launch" : mode == 1 ? "same buffer every launch   " : "each buffer read twice      ",
                   ms * 1e3f / iters);
        }
        CK(hipFree(buf));
    }
    return 0;
}
