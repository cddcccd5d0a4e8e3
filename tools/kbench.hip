// kbench.hip -- standalone microbenchmark for the decode-path GEMM (development tool, not product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <functional>
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
    const int nt = argc > 2 ? atoi(argv[2]) : 0;  // 1: non-temporal weight loads
    hipStream_t st; CK(hipStreamCreate(&st));
    struct Shape { const char* name; int N, K, epi, norm; };
    std::vector<Shape> shapes = {
        {"tk qkv  +norm", 4096, 2048, 0, 1}, {"tk qkv       ", 4096, 2048, 0, 0}, {"tk o    epi3 ", 2048, 2048, 3, 0},
        {"tk gateup+nrm", 6144, 2048, 2, 1}, {"tk gateup    ", 6144, 2048, 2, 0}, {"tk down epi3 ", 2048, 6144, 3, 0},
        {"tk head +norm", 3072, 2048, 0, 1},
        {"cp qkv  +norm", 4096, 1024, 0, 1}, {"cp qkv       ", 4096, 1024, 0, 0}, {"cp o    epi3 ", 1024, 2048, 3, 0},
        {"cp gateup+nrm", 3072, 1024, 2, 1}, {"cp down epi3 ", 1024, 3072, 3, 0}, {"cp head +norm", 2048, 1024, 0, 1},
        {"cp proj epi3 ", 1024, 2048, 3, 0},
        // half the bytes per workgroup at the same workgroup count: what an 8-column tile split of o_proj / down_proj could gain
        {"cp o   K/2   ", 1024, 1024, 3, 0}, {"cp down K/2  ", 1024, 1536, 3, 0}};
    const int Mp = (M + 15) / 16 * 16;
    const int copies = argc > 3 ? atoi(argv[3]) : 24;  // rotate weight copies so the Infinity Cache does not serve them (1: warm)
    for (auto& s : shapes) {
        const int rows = (s.epi == 2 ? 2 * s.N : s.N);
        size_t welems = (size_t)rows * s.K;
        uint16_t* W; CK(hipMalloc(&W, welems * 2 * copies)); CK(hipMemset(W, 0x3c, welems * 2 * copies));
        uint16_t* x; CK(hipMalloc(&x, (size_t)Mp * s.K * 2)); CK(hipMemset(x, 0x3c, (size_t)Mp * s.K * 2));
        uint16_t* y; CK(hipMalloc(&y, (size_t)Mp * s.N * 2)); CK(hipMemset(y, 0, (size_t)Mp * s.N * 2));
        uint16_t* nw; CK(hipMalloc(&nw, (size_t)s.K * 2)); CK(hipMemset(nw, 0x3c, (size_t)s.K * 2));
        float *ssi, *sso; CK(hipMalloc(&ssi, 512 * Mp * 4)); CK(hipMalloc(&sso, 512 * Mp * 4)); CK(hipMemset(ssi, 0, 512 * Mp * 4));
        int it = 0;
        auto f = [&](hipStream_t q) {
            GemmArgs a{}; a.W = W + (size_t)(it++ % copies) * welems; a.x = x; a.xMB = Mp / 16; a.M = M; a.Mpad = Mp; a.N = s.N; a.K = s.K;
            a.epi = s.epi; a.y = y; a.ldy = s.N; a.yMB = Mp / 16; a.ss_ld = Mp;
            if (s.norm) { a.norm_w = nw; a.ss_in = ssi; a.ss_count = s.K / 16; a.norm_dim = s.K; a.norm_eps = 1e-6f; }
            if (s.epi == 3) { a.resid = 1; a.ss_out = sso; }
            a.nt_weights = nt;
            launch_gemm_skinny(a, q);
        };
        float us = time_it(f, st, 200);
        double mb = welems * 2 / 1e6;
        printf("%s M=%d N=%5d K=%5d  %7.2f us  %6.1f MB  %6.2f TB/s\n", s.name, M, s.N, s.K, us, mb, mb * 1e6 / (us * 1e-6) / 1e12);
        CK(hipFree(W)); CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(nw)); CK(hipFree(ssi)); CK(hipFree(sso));
    }
    return 0;
}
