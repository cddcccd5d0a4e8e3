// Shader clock of the chip while something else runs on it: a one-wave kernel stamps s_memtime (shader cycles) and
// s_memrealtime (100 MHz) around a ~100 us spin; their ratio is the clock the wave ran at (MI355X_MICROARCH.md, DVFS item 6).
// Run beside bench.py (another process) to see what the frame loop's clock does when a codec decode starts:
//   ./tools/clockprobe 400 25 > clocks.txt &   python bench.py ...
// hipcc --offload-arch=gfx950 -O2 tools/clockprobe.hip -o tools/clockprobe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

__global__ void probe(unsigned long long* out, int spin) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    float x = 1.0f;
    for (int i = 0; i < spin; ++i) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = r1 - r0;
        out[2] = (unsigned long long)x;
    }
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 400, period_ms = argc > 2 ? atoi(argv[2]) : 25;
    unsigned long long *d, h[3];
    if (hipMalloc(&d, 24) != hipSuccess) return 1;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 60000);
        if (hipMemcpy(h, d, 24, hipMemcpyDeviceToHost) != hipSuccess) return 2;
        const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%8.3f s  %7.1f MHz  (%llu cycles / %llu ticks)\n", t, h[1] ? 100.0 * double(h[0]) / double(h[1]) : 0.0, h[0], h[1]);
        fflush(stdout);
        std::this_thread::sleep_for(std::chrono::milliseconds(period_ms));
    }
    return 0;
}
