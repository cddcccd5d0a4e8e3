#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
struct Big { int* p; int v; char pad[200]; };
template <int ID, int LDS> __global__ void k(Big a) {
    __shared__ float sh[LDS / 4 + 1];
    sh[threadIdx.x % (LDS / 4 + 1)] = (float)a.v;
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) a.p[ID] += (int)sh[0];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <int LDS> void enqueue(hipStream_t st, Big a, int nodes, int distinct, int wg, int thr) {
    for (int i = 0; i < nodes; ++i) {
        switch (distinct > 1 ? i % distinct : 0) {
            case 0: hipLaunchKernelGGL((k<0, LDS>), dim3(wg), dim3(thr), 0, st, a); break;
            case 1: hipLaunchKernelGGL((k<1, LDS>), dim3(wg), dim3(thr), 0, st, a); break;
            case 2: hipLaunchKernelGGL((k<2, LDS>), dim3(wg), dim3(thr), 0, st, a); break;
            case 3: hipLaunchKernelGGL((k<3, LDS>), dim3(wg), dim3(thr), 0, st, a); break;
            case 4: hipLaunchKernelGGL((k<4, LDS>), dim3(wg), dim3(thr), 0, st, a); break;
            case 5: hipLaunchKernelGGL((k<5, LDS>), dim3(wg), dim3(thr), 0, st, a); break;
            case 6: hipLaunchKernelGGL((k<6, LDS>), dim3(wg), dim3(thr), 0, st, a); break;
            default: hipLaunchKernelGGL((k<7, LDS>), dim3(wg), dim3(thr), 0, st, a); break;
        }
    }
}
template <int LDS> void run(const char* name, int nodes, int distinct, int wg, int thr) {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    int* d; CK(hipMalloc(&d, 64)); CK(hipMemset(d, 0, 64));
    Big a{}; a.p = d; a.v = 1;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    enqueue<LDS>(st, a, nodes, distinct, wg, thr);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    double t0 = now();
    for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, st));
    double t1 = now();
    CK(hipStreamSynchronize(st));
    double t2 = now();
    printf("%-28s nodes=%4d distinct=%d wg=%4d thr=%4d lds=%6d: enqueue %.2f us/node, total %.2f us/node\n", name, nodes, distinct, wg, thr, LDS,
           (t1 - t0) * 1e6 / (nodes * 20), (t2 - t0) * 1e6 / (nodes * 20));
}
int main() {
    run<16>("baseline", 600, 1, 256, 512);
    run<16>("8 distinct kernels", 600, 8, 256, 512);
    run<32768>("32K LDS", 600, 1, 256, 512);
    run<32768>("32K LDS, 8 distinct", 600, 8, 256, 512);
    run<16>("1024 threads", 600, 1, 32, 1024);
    run<16>("840 nodes", 840, 8, 256, 512);
    run<16>("2000 nodes", 2000, 8, 256, 512);
    return 0;
}
