// kargbench.hip -- does kernarg preloading (-mllvm -amdgpu-kernarg-preload-count=N, scalar kernel arguments) shorten a
// chain of short dependent kernels? Each node: 256 workgroups x 512 threads load in[i], add, store out[i] (ping-pong).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
struct Args { const float* in; float* out; const float* w; float s; int n; char pad[96]; };
__global__ __launch_bounds__(512) void k_struct(Args a) {
    const int i = blockIdx.x * 512 + threadIdx.x;
    if (i < a.n) a.out[i] = a.in[i] * a.s + a.w[i];
}
__global__ __launch_bounds__(512) void k_scalar(const float* in, float* out, const float* w, float s, int n) {
    const int i = blockIdx.x * 512 + threadIdx.x;
    if (i < n) out[i] = in[i] * s + w[i];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int n = 256 * 512, nodes = 600;
    float *a, *b, *w; CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&w, n * 4));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4)); CK(hipMemset(w, 0, n * 4));
    for (int mode = 0; mode < 2; ++mode) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < nodes; ++i) {
            const float* in = (i & 1) ? b : a; float* out = (i & 1) ? a : b;
            if (mode == 0) { Args x{}; x.in = in; x.out = out; x.w = w; x.s = 0.5f; x.n = n; hipLaunchKernelGGL(k_struct, dim3(256), dim3(512), 0, st, x); }
            else hipLaunchKernelGGL(k_scalar, dim3(256), dim3(512), 0, st, in, out, (const float*)w, 0.5f, n);
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        double t0 = now();
        for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        double t1 = now();
        printf("%s: %.3f us/node\n", mode == 0 ? "struct by value" : "scalar args    ", (t1 - t0) * 1e6 / (nodes * 20));
    }
    return 0;
}
