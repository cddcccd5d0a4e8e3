// launchbench.hip -- per-node cost of dependent kernel chains: eager vs hipGraph, small vs big kernargs.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
struct Small { int* p; int v; };
struct Big { int* p; int v; char pad[240]; };
template <class A> __global__ void k(A a) { if (threadIdx.x == 0 && blockIdx.x == 0) a.p[0] += a.v; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <class A> void run(const char* name, int nodes, int wg, int threads) {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    int* d; CK(hipMalloc(&d, 4)); CK(hipMemset(d, 0, 4));
    A a{}; a.p = d; a.v = 1;
    // eager
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k<A>, dim3(wg), dim3(threads), 0, st, a);
    CK(hipStreamSynchronize(st));
    double t0 = now();
    for (int i = 0; i < nodes * 20; ++i) hipLaunchKernelGGL(k<A>, dim3(wg), dim3(threads), 0, st, a);
    double t1 = now();
    CK(hipStreamSynchronize(st));
    double t2 = now();
    printf("%-6s wg=%4d thr=%4d eager : enqueue %.2f us/node, total %.2f us/node\n", name, wg, threads, (t1 - t0) * 1e6 / (nodes * 20), (t2 - t0) * 1e6 / (nodes * 20));
    // graph
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < nodes; ++i) hipLaunchKernelGGL(k<A>, dim3(wg), dim3(threads), 0, st, a);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    t0 = now();
    for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, st));
    t1 = now();
    CK(hipStreamSynchronize(st));
    t2 = now();
    printf("%-6s wg=%4d thr=%4d graph : enqueue %.2f us/node, total %.2f us/node\n", name, wg, threads, (t1 - t0) * 1e6 / (nodes * 20), (t2 - t0) * 1e6 / (nodes * 20));
}
int main() {
    run<Small>("small", 600, 1, 64);
    run<Big>("big", 600, 1, 64);
    run<Small>("small", 600, 256, 512);
    run<Big>("big", 600, 256, 512);
    return 0;
}
