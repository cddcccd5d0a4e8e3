// pfprobe.hip -- two questions behind "a launch touches the NEXT launch's weight tiles" (DESIGN.md section 5, round 4):
//   1. does workgroup b of EVERY launch land on XCD (b + const) % 8, whatever the grids of the launches before it?
//   2. how wide must a touch be for the XCD's L2 to hold the whole line afterwards (one dword per 32 / 64 / 128 bytes, or
//      every byte), and is the gain really the XCD-local L2 (a touch from the wrong XCD must gain nothing)?
//   hipcc --offload-arch=gfx950 -O3 -o tools/pfprobe tools/pfprobe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void xcc_probe(int* out) {
    if (threadIdx.x == 0) out[blockIdx.x + blockIdx.y * gridDim.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20);  // HW_REG_XCC_ID[3:0]
}

// workgroup b reads its span (span_vec uint4) completely
__global__ __launch_bounds__(512) void reader(const uint4* base, int span_vec, float* out) {
    const uint4* p = base + (size_t)blockIdx.x * span_vec;
    uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll 4
    for (int i = threadIdx.x; i < span_vec; i += 512) {
        const uint4 v = p[i];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[blockIdx.x] = 1.f;
}

// workgroup L touches the spans that the reader's workgroups on XCD (L + shift) % 8 will read: one dword per GRAN bytes
// (GRAN = 16: every byte, dwordx4). R = toucher workgroups per XCD, nspan = reader workgroups.
template <int GRAN>
__global__ __launch_bounds__(512) void toucher(const char* base, int span_bytes, int nspan, int R, int shift, float* out) {
    const int L = blockIdx.x, c = (L + shift) & 7, r = L >> 3;
    const int n_c = (nspan - c + 7) >> 3;
    const int units_per_span = span_bytes / (GRAN == 16 ? 16 : GRAN);
    const int total = n_c * units_per_span;
    uint32_t acc = 0;
    for (int q = r * 512 + threadIdx.x; q < total; q += R * 512) {
        const int s = q / units_per_span, u = q - s * units_per_span;
        const char* p = base + (size_t)(c + 8 * s) * span_bytes + (size_t)u * (GRAN == 16 ? 16 : GRAN);
        if constexpr (GRAN == 16) {
            const uint4 v = *reinterpret_cast<const uint4*>(p);
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        } else {
            acc ^= *reinterpret_cast<const uint32_t*>(p);
        }
    }
    if (acc == 0x12345678u) out[blockIdx.x] = 1.f;
}

int main() {
    // ---- 1. workgroup -> XCD across launches of different grids, eager and as one graph ----
    int* xo;
    CK(hipMalloc(&xo, 8 * 1024 * 4));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const int grids[8] = {256, 128, 100, 192, 33, 224, 7, 256};
    for (int mode = 0; mode < 2; ++mode) {
        CK(hipMemset(xo, 0xff, 8 * 1024 * 4));
        hipGraph_t g = nullptr;
        hipGraphExec_t ge = nullptr;
        if (mode) CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(xcc_probe, dim3(grids[i]), dim3(64), 0, st, xo + 1024 * i);
        if (mode) {
            CK(hipStreamEndCapture(st, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, st));
            CK(hipGraphLaunch(ge, st));
        }
        CK(hipStreamSynchronize(st));
        std::vector<int> h(8 * 1024);
        CK(hipMemcpy(h.data(), xo, h.size() * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < 8; ++i) {
            int bad = 0;
            for (int b = 0; b < grids[i]; ++b) bad += ((h[1024 * i + b] - h[1024 * i]) & 7) != (b & 7);
            printf("%s launch %d grid %3d: block 0 on XCD %d, blocks off the round-robin: %d\n", mode ? "graph" : "eager", i, grids[i], h[1024 * i], bad);
        }
    }
    {   // 2-D grid: linear id = x + y * gridDim.x ?
        CK(hipMemset(xo, 0xff, 8 * 1024 * 4));
        hipLaunchKernelGGL(xcc_probe, dim3(64, 2), dim3(64), 0, st, xo);
        hipLaunchKernelGGL(xcc_probe, dim3(12, 4), dim3(64), 0, st, xo + 1024);
        CK(hipStreamSynchronize(st));
        std::vector<int> h(2048);
        CK(hipMemcpy(h.data(), xo, h.size() * 4, hipMemcpyDeviceToHost));
        int bad = 0, bad2 = 0;
        for (int b = 0; b < 128; ++b) bad += ((h[b] - h[0]) & 7) != (b & 7);
        for (int b = 0; b < 48; ++b) bad2 += ((h[1024 + b] - h[1024]) & 7) != (b & 7);
        printf("2-D grid (64,2): block 0 on XCD %d, off the x-fastest round-robin: %d; (12,4): block 0 on %d, off: %d\n", h[0], bad, h[1024], bad2);
    }

    // ---- 2. touch width and locality ----
    const int G = 256;
    float* out;
    CK(hipMalloc(&out, G * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int span : {32 << 10, 96 << 10}) {
        const size_t bytes = (size_t)span * G;
        const int NB = (int)((size_t(640) << 20) / bytes);  // > Infinity Cache in total
        char* buf;
        CK(hipMalloc(&buf, bytes * NB));
        CK(hipMemset(buf, 1, bytes * NB));
        auto run = [&](int variant, int shift, bool with_reader, bool with_toucher) {
            const int iters = 300;
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i) {
                char* b = buf + (size_t)(i % NB) * bytes;
                if (with_toucher) {
                    switch (variant) {
                        case 16: hipLaunchKernelGGL(toucher<16>, dim3(G), dim3(512), 0, st, b, span, G, G / 8, shift, out); break;
                        case 32: hipLaunchKernelGGL(toucher<32>, dim3(G), dim3(512), 0, st, b, span, G, G / 8, shift, out); break;
                        case 64: hipLaunchKernelGGL(toucher<64>, dim3(G), dim3(512), 0, st, b, span, G, G / 8, shift, out); break;
                        case 128: hipLaunchKernelGGL(toucher<128>, dim3(G), dim3(512), 0, st, b, span, G, G / 8, shift, out); break;
                    }
                }
                if (with_reader) hipLaunchKernelGGL(reader, dim3(G), dim3(512), 0, st, reinterpret_cast<const uint4*>(b), span / 16, out);
            }
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            return ms * 1e3f / iters;
        };
        const float cold = run(0, 0, true, false);
        printf("span %3d KiB (%5.1f MB per launch): reader alone, cold %.2f us\n", span >> 10, bytes / 1e6, cold);
        for (int variant : {16, 32, 64, 128}) {
            for (int shift : {0, 1}) {
                const float t = run(variant, shift, false, true), tr = run(variant, shift, true, true);
                printf("  touch every %3d B from XCD%+d: toucher %.2f us, toucher + reader %.2f us -> reader %.2f us (cold %.2f)\n", variant, shift, t, tr, tr - t, cold);
            }
        }
        CK(hipFree(buf));
    }
    return 0;
}
