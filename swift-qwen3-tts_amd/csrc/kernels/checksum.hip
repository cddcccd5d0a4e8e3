// checksum.hip -- 64-bit sum of a device buffer's 32-bit words: what every rank compares after the load-time broadcast of the
// weight arena (csrc/comm.cc, include/q3tts.h q3tts_model_arena_checksum). Integer adds commute, so the grid shape does not
// matter: partial sums per workgroup, one 64-bit atomic each. HBM-bound, once per load.
#include "../comm.h"
#include "../common.h"

namespace q3 {
namespace {

__global__ __launch_bounds__(256) void checksum_kernel(const uint4* p, size_t n_vec, const uint32_t* tail, int n_tail, unsigned long long* out) {
    unsigned long long s = 0;
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n_vec; i += size_t(gridDim.x) * 256) {
        const uint4 v = p[i];
        s += (unsigned long long)v.x + v.y + v.z + v.w;
    }
    if (blockIdx.x == 0 && int(threadIdx.x) < n_tail) s += tail[threadIdx.x];
    __shared__ unsigned long long part[256];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (int(threadIdx.x) < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(out, part[0]);
}

}  // namespace

uint64_t arena_checksum(int device, const void* arena, size_t bytes) {
    Q3_CHECK(arena && bytes >= 4, 1, "Model not initialized: no weight arena");
    Q3_HIP(hipSetDevice(device));
    unsigned long long* d = nullptr;
    Q3_HIP(hipMalloc(reinterpret_cast<void**>(&d), 8));
    unsigned long long h = 0;
    try {
        Q3_HIP(hipMemset(d, 0, 8));
        const size_t words = bytes / 4, n_vec = words / 4;
        const int n_tail = int(words - n_vec * 4);
        (void)hipGetLastError();  // the runtime's last-error slot is sticky: an earlier, handled failure must not be read as this launch's
        hipLaunchKernelGGL(checksum_kernel, dim3(2048), dim3(256), 0, nullptr, static_cast<const uint4*>(arena), n_vec,
                           static_cast<const uint32_t*>(arena) + n_vec * 4, n_tail, d);
        Q3_HIP(hipGetLastError());
        Q3_HIP(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
    } catch (...) {
        (void)hipFree(d);
        throw;
    }
    (void)hipFree(d);
    return uint64_t(h);
}

}  // namespace q3
