// xcd_handoff.hip -- probe for an XCD-local cluster kernel: (1) which XCD does workgroup i of a 1-D grid land on
// (HW_REG_XCC_ID), (2) what does a producer -> consumer hand-off between two workgroups cost when both sit on ONE
// XCD and talk through that XCD's L2 (plain cached memory, sc1 = agent-scope loads/stores that bypass only the
// per-CU L1), against partners on different XCDs (where cached memory is expected to go stale).
//   hipcc --offload-arch=gfx950 -O3 -o tools/xcd_handoff tools/xcd_handoff.hip && tools/xcd_handoff
// Every spin is bounded by the wall clock: a surprise ends in an error count, not a hang.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

struct Args {
    unsigned* flag;   // [G] producer -> consumer sequence numbers
    unsigned* ack;    // [G] consumer -> producer
    float* data;      // [G][256] payload (1 KiB per pair)
    int* err;         // [0] timeouts, [1] stale payload reads
    unsigned* xcc;    // [G]
    unsigned long long* ticks;  // [G] wall-clock ticks of the ping-pong loop (100 MHz)
    int iters, stride;          // partner of producer b is b + stride
};

__global__ __launch_bounds__(256) void handoff(Args a) {
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) a.xcc[b] = xcc_id();
    const int pair = (b / a.stride) % 2;  // 0: producer, 1: consumer of workgroup b - stride
    const int p = pair == 0 ? b : b - a.stride;
    __shared__ int ok_s;
    const unsigned long long t_begin = wall_clock64();
    for (int it = 1; it <= a.iters; ++it) {
        if (pair == 0) {
            // payload: every thread one float, agent-scope (L1-bypassing) store into this XCD's L2
            __hip_atomic_store(&a.data[p * 256 + tid], (float)(it * 1000 + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_waitcnt(0);  // stores acknowledged
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_store(&a.flag[p], (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long t0 = wall_clock64();
                int ok = 1;
                while (__hip_atomic_load(&a.ack[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it) {
                    if (wall_clock64() - t0 > 5000000ull) { atomicAdd(&a.err[0], 1); ok = 0; break; }  // 50 ms
                }
                ok_s = ok;
            }
            __syncthreads();
            if (!ok_s) return;
        } else {
            if (tid == 0) {
                const unsigned long long t0 = wall_clock64();
                int ok = 1;
                while (__hip_atomic_load(&a.flag[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it) {
                    if (wall_clock64() - t0 > 5000000ull) { atomicAdd(&a.err[0], 1); ok = 0; break; }
                }
                ok_s = ok;
            }
            __syncthreads();
            if (!ok_s) return;
            const float v = __hip_atomic_load(&a.data[p * 256 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v != (float)(it * 1000 + tid)) atomicAdd(&a.err[1], 1);
            __syncthreads();
            if (tid == 0) __hip_atomic_store(&a.ack[p], (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (tid == 0) a.ticks[b] = wall_clock64() - t_begin;
}

int main() {
    const int G = 256, iters = 2000;
    Args a{};
    CK(hipMalloc(&a.flag, G * 4)); CK(hipMalloc(&a.ack, G * 4)); CK(hipMalloc(&a.data, G * 256 * 4));
    CK(hipMalloc(&a.err, 8)); CK(hipMalloc(&a.xcc, G * 4)); CK(hipMalloc(&a.ticks, G * 8));
    a.iters = iters;
    for (int stride : {8, 1, 4, 64}) {   // 8 and 64: partner on the same XCD if placement is round-robin; 1, 4: other XCDs
        CK(hipMemset(a.flag, 0, G * 4)); CK(hipMemset(a.ack, 0, G * 4)); CK(hipMemset(a.data, 0, G * 256 * 4));
        CK(hipMemset(a.err, 0, 8)); CK(hipMemset(a.ticks, 0, G * 8));
        a.stride = stride;
        hipLaunchKernelGGL(handoff, dim3(G), dim3(256), 0, 0, a);
        CK(hipDeviceSynchronize());
        int err[2];
        std::vector<unsigned> xcc(G);
        std::vector<unsigned long long> ticks(G);
        CK(hipMemcpy(err, a.err, 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(xcc.data(), a.xcc, G * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ticks.data(), a.ticks, G * 8, hipMemcpyDeviceToHost));
        int rr = 0, same = 0, pairs = 0;
        unsigned long long tmax = 0;
        for (int b = 0; b < G; ++b) {
            rr += (int)(xcc[b] == (unsigned)(b % 8));
            if ((b / stride) % 2 == 0 && b + stride < G) { ++pairs; same += (int)(xcc[b] == xcc[b + stride]); }
            if (ticks[b] > tmax) tmax = ticks[b];
        }
        printf("stride %3d: xcc == blockIdx %% 8 for %d/%d workgroups; partner on same XCD %d/%d; timeouts %d stale %d; "
               "%.0f ns per round trip (two hand-offs)\n", stride, rr, G, same, pairs, err[0], err[1], tmax * 10.0 / iters);
    }
    return 0;
}
