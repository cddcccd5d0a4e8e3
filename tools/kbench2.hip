// kbench2.hip -- do two hipGraphs of dependent skinny-GEMM chains on two streams overlap on the GPU?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include "common.h"
#include "kernels.h"
using namespace q3;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct Chain {
    hipStream_t st; hipGraphExec_t ge; uint16_t *W, *x, *y, *nw; float* ss;
    void build(int M, int N, int K, int nodes) {
        const int Mp = (M + 15) / 16 * 16;
        CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        CK(hipMalloc(&W, (size_t)N * K * 2 * 8)); CK(hipMemset(W, 0x3c, (size_t)N * K * 2 * 8));
        CK(hipMalloc(&x, (size_t)Mp * K * 2)); CK(hipMemset(x, 0x3c, (size_t)Mp * K * 2));
        CK(hipMalloc(&y, (size_t)Mp * N * 2)); CK(hipMalloc(&nw, K * 2)); CK(hipMemset(nw, 0x3c, K * 2));
        CK(hipMalloc(&ss, 512 * Mp * 4)); CK(hipMemset(ss, 0, 512 * Mp * 4));
        hipGraph_t g;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < nodes; ++i) {
            GemmArgs a{}; a.W = W + (size_t)(i % 8) * N * K; a.x = x; a.xMB = Mp / 16; a.M = M; a.Mpad = Mp; a.N = N; a.K = K;
            a.epi = 0; a.y = y; a.ldy = N; a.ss_ld = Mp; a.norm_w = nw; a.ss_in = ss; a.ss_count = K / 16; a.norm_dim = K; a.norm_eps = 1e-6f;
            launch_gemm_skinny(a, st);
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    }
};
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 16, nodes = 400, reps = 10;
    for (int nch : {1, 2, 4}) {
        std::vector<Chain> ch((size_t)nch);
        for (auto& c : ch) c.build(M, 4096, 1024, nodes);
        for (auto& c : ch) { CK(hipGraphLaunch(c.ge, c.st)); CK(hipStreamSynchronize(c.st)); }
        double t0 = now();
        std::vector<std::thread> th;
        for (auto& c : ch) th.emplace_back([&c, reps] { for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(c.ge, c.st)); CK(hipStreamSynchronize(c.st)); });
        for (auto& t : th) t.join();
        double t1 = now();
        printf("chains=%d M=%d: %.2f us per node per chain (wall %.1f ms for %d nodes each)\n", nch, M, (t1 - t0) * 1e6 / (nodes * reps), (t1 - t0) * 1e3, nodes * reps);
    }
    return 0;
}
