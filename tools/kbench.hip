// kbench.hip -- standalone microbenchmark for the decode-path kernels (development tool, not product).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I swift-qwen3-tts_amd/csrc tools/kbench.hip \
//        swift-qwen3-tts_amd/csrc/kernels/gemm_decode.hip swift-qwen3-tts_amd/csrc/kernels/lm_misc.hip -o tools/kbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <functional>
#include <cstdlib>
#include <vector>
#include "common.h"
#include "kernels.h"
using namespace q3;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)

static float time_it(std::function<void(hipStream_t)> f, hipStream_t st, int iters) {
    hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i=0;i<5;++i) f(st);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(a,st));
    for (int i=0;i<iters;++i) f(st);
    CK(hipEventRecord(b,st));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms,a,b));
    return ms*1000.f/iters;
}

int main(int argc, char** argv) {
    int M = argc > 1 ? atoi(argv[1]) : 32;
    hipStream_t st; CK(hipStreamCreate(&st));
    struct Shape { const char* name; int N, K, epi, S; };
    std::vector<Shape> shapes = {
        {"tk qkv    ", 4096, 2048, 0, 1}, {"tk o   S2 ", 2048, 2048, 1, 2}, {"tk o   S4 ", 2048, 2048, 1, 4}, {"tk gateup ", 6144, 2048, 2, 1},
        {"tk down S2", 2048, 6144, 1, 2}, {"tk down S3", 2048, 6144, 1, 3}, {"tk down S6", 2048, 6144, 1, 6}, {"tk head   ", 3072, 2048, 0, 1},
        {"cp qkv    ", 4096, 1024, 0, 1}, {"cp o      ", 1024, 2048, 1, 4}, {"cp gateup ", 3072, 1024, 2, 1},
        {"cp down S3", 1024, 3072, 1, 3}, {"cp down S6", 1024, 3072, 1, 6}, {"cp head   ", 2048, 1024, 0, 1}, {"cp proj   ", 1024, 2048, 0, 1}};
    const int Mp = (M + 15) / 16 * 16;
    // rotate over several weight copies so that the Infinity Cache does not serve the weights
    const int copies = 24;
    for (auto& s : shapes) {
        const int tiles_rows = (s.epi == 2 ? 2 * s.N : s.N);
        size_t welems = (size_t)tiles_rows * s.K;
        uint16_t* W; CK(hipMalloc(&W, welems * 2 * copies)); CK(hipMemset(W, 0x3c, welems * 2 * copies));
        uint16_t* x; CK(hipMalloc(&x, (size_t)Mp * s.K * 2)); CK(hipMemset(x, 0x3c, (size_t)Mp * s.K * 2));
        uint16_t* y; CK(hipMalloc(&y, (size_t)Mp * s.N * 2));
        float* part; CK(hipMalloc(&part, (size_t)8 * Mp * s.N * 4)); (void)0;
        int it = 0;
        auto f = [&](hipStream_t q) {
            GemmArgs a{}; a.W = W + (size_t)(it++ % copies) * welems; a.x = x; a.xMB = Mp / 16; a.M = M; a.Mpad = Mp; a.N = s.N; a.K = s.K;
            a.S = s.S; a.epi = s.epi; a.y = y; a.ldy = s.N; a.yMB = Mp / 16; a.part = part;
            launch_gemm_skinny(a, q);
        };
        float us = time_it(f, st, 200);
        double mb = welems * 2 / 1e6;
        printf("%s M=%d N=%5d K=%5d epi=%d S=%d  %7.2f us  %6.1f MB  %6.2f TB/s\n", s.name, M, s.N, s.K, s.epi, s.S, us, mb, mb * 1e6 / (us * 1e-6) / 1e12);
        CK(hipFree(W)); CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(part));
    }
    {   // resid_norm
        const int H = 2048; uint16_t *h, *xn, *w; float* part;
        CK(hipMalloc(&h, Mp*H*2)); CK(hipMalloc(&xn, Mp*H*2)); CK(hipMalloc(&w, H*2)); CK(hipMalloc(&part, 8*Mp*H*4));
        CK(hipMemset(h,0x3c,Mp*H*2)); CK(hipMemset(w,0x3c,H*2)); CK(hipMemset(part,0,8*Mp*H*4));
        for (int S : {0, 2, 4}) {
            auto f = [&](hipStream_t q){ ResidNormArgs a{}; a.h=h; a.ldh=H; a.part=S?part:nullptr; a.S=S; a.Mpad=Mp; a.w=w; a.eps=1e-6f; a.xn=xn; a.xnMB=Mp/16; a.M=M; a.H=H; launch_resid_norm(a,q); };
            printf("resid_norm H=%d S=%d: %.2f us\n", H, S, time_it(f, st, 500));
        }
        auto g = [&](hipStream_t q){ launch_advance_len((int32_t*)part, nullptr, M, q); };
        printf("advance_len (empty-ish kernel): %.2f us\n", time_it(g, st, 500));
    }
    return 0;
}
