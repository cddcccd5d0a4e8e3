// prefetch.h -- a launch of the frame step touches weight tiles that a LATER launch of the chain will stream (round 4).
//
// Why: the frame step is ~550 dependent launches, each of which starts by requesting its weight tiles and then waits one
// memory round trip (plus the streaming time of 16-200 KB per CU) before its first MFMA, while HBM and the fabric sit idle
// through every launch's ramp, epilogue and boundary. Measured on the part (tools/pfprobe.hip): workgroup b of EVERY launch
// lands on XCD (b + const) % 8 whatever ran before it, data stays in an XCD's L2 across kernel boundaries, ONE dword per
// 128-byte line pulls the whole line in, and a launch that finds its 8 / 25 MB of weights in its own XCD's L2 takes
// 2.0 / 2.5 us instead of 3.2 / 6.4 us (from the Infinity Cache, i.e. touched by the wrong XCD: 2.6 / 5.0).
//
// How: the host knows the whole chain when it enqueues (captures) it. Every decode-shaped launch carries a PfArgs that names
// a byte stream of a later launch: `nspan` contiguous spans of `span` bytes, span x' being what workgroup x' of that later
// launch reads (its weight tiles over all of K). The spans with x' = c (mod 8) are the ones XCD c will read; the workgroups
// of THIS launch that sit on XCD c (linear id = c mod 8) share lines [u0, u1) of that XCD's part of the stream among
// themselves: P touches per thread, issued right behind the launch's own operand requests (loads return in issue order, so
// the launch's own data is not delayed by more than the issue slots) and retired by the last instruction of the kernel.
// The touches are ordinary tracked loads (one VGPR each), so hipcc's wait counts stay exact.
//
// Nothing here changes what any kernel computes: results are bit-identical with and without (tests/test_gpu_parity.py
// runs both ways, Q3TTS_PF=0 hands every launch an empty range).
#pragma once
#include <cstdint>

// MEASURED AND LOST (profiles/r04_touch_ahead_ab.txt): the shipped build compiles the touch
// code OUT (Q3_PF_MODE 0). Build-time variants for the A/B runs (tools/build_pf_variants.sh): 0 = no touch code at all,
// 1 = touches behind the launch's own requests, 2 = in front of them, 3 = behind, as agent-scope loads (`sc1`: past the
// CU's L1); Q3_PF_GEMM=0 leaves the GEMMs out (the attention launches alone touch).
#ifndef Q3_PF_MODE
#define Q3_PF_MODE 0
#endif
#ifndef Q3_PF_GEMM
#define Q3_PF_GEMM 1
#endif

namespace q3 {

struct PfArgs {
    const uint8_t* base;   // first byte of the later launch's stream (always a valid address: an empty range still loads)
    uint32_t span;         // bytes per span, a multiple of 128
    uint32_t lines;        // span / 128
    float inv_lines;       // 1 / lines
    uint32_t u0, u1;       // lines [u0, u1) of each XCD's part of the stream (span-major); u1 > u0 >= 0 unless empty (u1 == 0)
    uint32_t wg_per_xcd;   // workgroups of THIS launch per XCD (filled in by the launcher, like gx)
    uint32_t gx;           // grid.x of THIS launch: linear workgroup id = blockIdx.x + blockIdx.y * gx (no dispatch-packet read)
};

#if defined(__HIPCC__)
constexpr int kPfTouches = 4;  // per thread: 4 x 128 B x 512 threads = 256 KiB per workgroup at most

// lin = linear workgroup id of this launch (x fastest), nthr = its threads
template <int P>
__device__ __forceinline__ void pf_issue(const PfArgs& p, uint32_t lin, uint32_t nthr, uint32_t tid, uint32_t (&v)[P]) {
    const uint32_t c = lin & 7u, r = lin >> 3;
    const uint32_t last = p.u1 ? p.u1 - 1u : 0u;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        uint32_t q = p.u0 + (uint32_t(i) * p.wg_per_xcd + r) * nthr + tid;
        q = q < last ? q : last;  // surplus threads re-touch the last line (an L1 hit); the load itself stays unconditional
        const uint32_t s = uint32_t((float(q) + 0.5f) * p.inv_lines);  // q / lines, exact below 2^22 lines
        const uint32_t u = q - s * p.lines;
        // a 32-bit offset from a uniform base: `global_load_dword v, v, s[base:base+1]`, one VGPR per touch (streams < 4 GiB)
        const uint32_t off = (c + 8u * s) * p.span + u * 128u;
#if Q3_PF_MODE == 3
        v[i] = __hip_atomic_load(reinterpret_cast<const uint32_t*>(p.base + off), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
        v[i] = *reinterpret_cast<const uint32_t*>(p.base + off);
#endif
    }
}

template <int P>
__device__ __forceinline__ void pf_retire(const uint32_t (&v)[P]) {
#pragma unroll
    for (int i = 0; i < P; ++i) asm volatile("" ::"v"(v[i]));
}
#endif

}  // namespace q3
