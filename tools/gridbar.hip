// gridbar.hip -- feasibility probe for a persistent decode kernel: cost of a device-wide barrier between
// dependent phases on MI355X (8 XCDs, non-coherent L2s), with cross-XCD visibility checked, and the same
// with a 32 KiB-per-workgroup weight prefetch issued before each barrier.
//   hipcc --offload-arch=gfx950 -O3 -o tools/gridbar tools/gridbar.hip && tools/gridbar
// Every spin is bounded (wall clock), so a scheduling surprise ends in an error flag, not a hang.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Args {
    unsigned* counter;   // monotonic arrival counter
    int* err;            // [0] timeouts, [1] stale reads
    float* xbuf;         // [2][G][64] exchange buffer
    const uint4* w;      // weights: [nslots][G][2048] uint4 (32 KiB per workgroup per slot)
    float* sink;
    float* big;          // [2][16384] floats, uncached
    int phases, mode, nslots;
    unsigned* flags;
    int ld;              // exchange loads: 0 plain, 1 4-byte agent-scope, 2 16-byte agent-scope (asm)
    int kind;            // 0 counter + agent fences, 1 flags + agent fences, 2 flags on uncached memory, no cache maintenance
};

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, int* err) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // every wave: its stores are written back past the XCD's L2
    __syncthreads();
    __shared__ int ok_s;
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (wall_clock64() - t0 > 20000000ull) {  // 0.2 s at 100 MHz
                atomicAdd(&err[0], 1);
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        ok_s = ok ? 1 : 0;
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // every wave: drop stale lines before reading other XCDs' data
    return ok_s != 0;
}

// Flag barrier: no read-modify-write. Workgroup g publishes flags[g] = phase; wave 0 of every workgroup polls all
// flags with coherent loads (4 per lane for up to 256 workgroups) until every one has reached the phase.
__device__ __forceinline__ bool flag_barrier(unsigned* flags, int G, unsigned phase, int* err) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    __shared__ int ok_s;
    if (threadIdx.x < 64) {
        if (threadIdx.x == 0) __hip_atomic_store(&flags[blockIdx.x], phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = wall_clock64();
        bool ok = true;
        for (;;) {
            bool all = true;
            for (int i = threadIdx.x; i < G; i += 64)
                all = all && (__hip_atomic_load(&flags[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= phase);
            if (__all(all)) break;
            if (wall_clock64() - t0 > 20000000ull) {
                if (threadIdx.x == 0) atomicAdd(&err[0], 1);
                ok = false;
                break;
            }
        }
        if (threadIdx.x == 0) ok_s = ok ? 1 : 0;
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return ok_s != 0;
}

// Same flag barrier without agent-scope cache maintenance: valid when every buffer exchanged between workgroups
// (and the flags) lives in uncached device memory (hipDeviceMallocUncached), so plain loads/stores are coherent.
__device__ __forceinline__ bool flag_barrier_uc(unsigned* flags, int G, unsigned phase, int* err, int inv) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // s_waitcnt: this wave's stores have been acknowledged
    __syncthreads();
    __shared__ int ok_s;
    if (threadIdx.x < 64) {
        if (threadIdx.x == 0) __hip_atomic_store(&flags[blockIdx.x], phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = wall_clock64();
        bool ok = true;
        for (;;) {
            unsigned v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {  // all polls in flight at once
                const int i = threadIdx.x + 64 * j;
                v[j] = i < G ? __hip_atomic_load(&flags[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : phase;
            }
            const bool all = v[0] >= phase && v[1] >= phase && v[2] >= phase && v[3] >= phase;
            if (__all(all)) break;
            if (wall_clock64() - t0 > 20000000ull) {
                if (threadIdx.x == 0) atomicAdd(&err[0], 1);
                ok = false;
                break;
            }
        }
        if (threadIdx.x == 0) ok_s = ok ? 1 : 0;
    }
    __syncthreads();
    if (inv == 1) asm volatile("buffer_inv sc0" ::: "memory");        // L1 only
    else if (inv == 2) asm volatile("buffer_inv sc1" ::: "memory");   // L1 + non-coherent L2 lines
    else if (inv == 3) asm volatile("buffer_inv sc0 sc1" ::: "memory");
    return ok_s != 0;
}

__global__ __launch_bounds__(512) void persist(Args a) {
    const int G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
    float acc = 0.f;
    uint4 pre[4] = {};
    for (int p = 0; p < a.phases; ++p) {
        if (a.mode >= 1) {  // produce: 64 floats per workgroup, then fence (release is on the atomic)
            if (tid < 64) a.xbuf[((size_t)(p & 1) * G + wg) * 64 + tid] = (float)(p * 1000 + wg);
        }
        if (a.mode >= 2) {  // prefetch the next phase's weights (independent of the barrier)
            const uint4* src = a.w + ((size_t)(p % a.nslots) * G + wg) * 2048;
#pragma unroll
            for (int i = 0; i < 4; ++i) pre[i] = src[i * 512 + tid];
        }
        if (a.kind >= 2 ? !flag_barrier_uc(a.flags, G, (unsigned)(p + 1), a.err, a.kind - 2)
            : a.kind == 1 ? !flag_barrier(a.flags, G, (unsigned)(p + 1), a.err)
                          : !grid_barrier(a.counter, (unsigned)(G * (p + 1)), a.err)) return;
        if (a.mode >= 1) {  // consume another workgroup's data (different XCD: wg ids round-robin over XCDs)
            const int other = (wg + 37) % G;
            const float* src = &a.xbuf[((size_t)(p & 1) * G + other) * 64];
            float v;
            if (a.ld == 0) v = src[tid & 63];
            else if (a.ld == 1) v = __hip_atomic_load(&src[tid & 63], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else {  // 16-byte load with agent scope bits
                float4 q;
                const float* ptr = src + 4 * (tid & 15);
                asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(q) : "v"(ptr) : "memory");
                v = (tid & 3) == 0 ? q.x : (tid & 3) == 1 ? q.y : (tid & 3) == 2 ? q.z : q.w;
                if (v != (float)(p * 1000 + other)) atomicAdd(&a.err[1], 1);
                v = src[tid & 63] * 0.f + v;
            }
            if (v != (float)(p * 1000 + other)) atomicAdd(&a.err[1], 1);
            acc += v;
        }
        if (a.mode >= 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc += (float)(pre[i].x ^ pre[i].y ^ pre[i].z ^ pre[i].w);
        }
        if (a.mode >= 3) {  // 64 KiB of activations per workgroup through agent-scope loads (8 x 16 B per thread)
            float4 q[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float* ptr = a.big + ((size_t)(p & 1) * 16384) + (size_t)(i * 512 + tid) * 4;
                asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(q[i]) : "v"(ptr) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += q[i].x + q[i].w;
            if (tid < 64) a.big[((size_t)((p + 1) & 1) * 16384) + (size_t)(wg % 256) * 64 + tid] = acc;
        }
    }
    if (acc == 123.456f) a.sink[0] = acc;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int G = prop.multiProcessorCount;
    printf("CUs %d\n", G);
    Args a{};
    const int nslots = 24;
    CK(hipMalloc(&a.counter, 4));
    CK(hipMalloc(&a.err, 8));
    float* xbuf_c;
    float* xbuf_uc;
    CK(hipMalloc(&xbuf_c, sizeof(float) * 2 * G * 64));
    CK(hipExtMallocWithFlags((void**)&xbuf_uc, sizeof(float) * 2 * G * 64, hipDeviceMallocUncached));
    CK(hipMalloc(&a.sink, 4));
    uint4* w;
    CK(hipMalloc(&w, (size_t)nslots * G * 2048 * sizeof(uint4)));
    CK(hipMemset(w, 1, (size_t)nslots * G * 2048 * sizeof(uint4)));
    a.w = w;
    a.nslots = nslots;
    unsigned* flags;
    CK(hipExtMallocWithFlags((void**)&flags, 4 * 1024, hipDeviceMallocUncached));
    CK(hipExtMallocWithFlags((void**)&a.big, 2 * 16384 * 4 + 4096, hipDeviceMallocUncached));
    CK(hipMemset(a.big, 0, 2 * 16384 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float* big_uc = a.big;
    float* big_c;
    CK(hipMalloc(&big_c, 2 * 16384 * 4 + 4096));
    CK(hipMemset(big_c, 0, 2 * 16384 * 4));
    for (int mk = 0; mk < 2; ++mk)
    for (int ld = 1; ld < 3; ++ld)
    for (int kind = 2; kind >= 2; --kind)
    for (int grid : {G}) {
        for (int mode = 0; mode < 4; ++mode) {
            for (int phases : {1000}) {
                a.flags = flags;
                a.kind = kind;
                a.ld = ld;
                a.xbuf = mk == 0 ? xbuf_uc : xbuf_c;
                a.big = mk == 0 ? big_uc : big_c;
                a.phases = phases;
                a.mode = mode;
                float best = 1e9f;
                int herr[2] = {0, 0};
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipMemset(a.counter, 0, 4));
                    CK(hipMemset(flags, 0, 4 * 1024));
                    CK(hipMemset(a.err, 0, 8));
                    CK(hipEventRecord(e0, 0));
                    hipLaunchKernelGGL(persist, dim3(grid), dim3(512), 0, 0, a);
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    best = ms < best ? ms : best;
                    CK(hipMemcpy(herr, a.err, 8, hipMemcpyDeviceToHost));
                    if (herr[0]) break;
                }
                printf("%s ld%d %s grid %3d mode %d phases %4d: %.3f ms -> %.2f us/phase  timeouts %d stale %d\n", mk ? "cached-mem" : "uncached  ", ld, kind == 2 ? "uc      " : kind == 3 ? "uc+inv0 " : kind == 4 ? "uc+inv1 " : kind == 5 ? "uc+inv01" : "flags   ", grid, mode, phases, best,
                       best * 1e3f / phases, herr[0], herr[1]);
                if (herr[0]) return 1;
            }
        }
    }
    return 0;
}
