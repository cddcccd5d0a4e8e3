// comm.cc -- the one collective of a batch-sharded job, behind the C ABI: the load-time broadcast of the weight arena
// (DESIGN.md section 6). One process -- or one handle -- per GPU; rank `root` has read the checkpoint,
// the others were loaded with q3tts_load_opts.weights_from_broadcast and receive the arena (a pure function of the config,
// csrc/model.cc) in ONE ncclBroadcast over xGMI. A Swift or C host needs nothing but this library: RCCL is opened at run time
// (librccl.so.1, the ROCm image's or the one a host process has already loaded), so single-GPU users never touch it.
// Reference: none -- the reference is single-device (Qwen3.swift:1382-1470 loads one model into one MLX device).
#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <string>

#include "../../include/q3tts.h"
#include "comm.h"
#include "common.h"

namespace q3 {
namespace {

// the slice of rccl.h this file uses (rccl/rccl.h: ncclUniqueId :43, ncclCommInitRank :220, ncclBroadcast :591)
struct NcclUniqueId { char internal[128]; };
using NcclComm = void*;
constexpr int kNcclUint8 = 1;
struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(NcclUniqueId*) = nullptr;
    int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // A copy the process has already loaded (a host with its own RCCL, torch's bundled one) is reused; otherwise the ROCm
        // image's. RTLD_LOCAL: RCCL drags librocm_smi64 in, and a process that later loads ANOTHER copy of that library (torch
        // bundles one under a different soname) must not have the two interposed on each other -- with RTLD_GLOBAL such a
        // process aborted at exit ("double free or corruption": one set of globals destructed twice).
        r.h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (r.h) break;
            r.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        }
        if (!r.h) {
            r.err = std::string("RCCL is not available (dlopen librccl.so.1: ") + (dlerror() ? dlerror() : "?") + ")";
            return;
        }
        auto sym = [&](const char* s) {
            void* p = dlsym(r.h, s);
            if (!p && r.err.empty()) r.err = std::string("RCCL symbol missing: ") + s;
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(sym("ncclBroadcast"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return r;
}

void nccl_check(int rc, const char* what) {
    if (rc == 0) return;
    Rccl& r = rccl();
    throw Error(7, std::string("RCCL error in ") + what + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "?"));
}

}  // namespace

void comm_unique_id(q3tts_comm_id* out) {
    static_assert(sizeof(q3tts_comm_id) == sizeof(NcclUniqueId), "q3tts_comm_id must hold an ncclUniqueId");
    Rccl& r = rccl();
    Q3_CHECK(r.err.empty(), 7, r.err);
    NcclUniqueId id;
    nccl_check(r.GetUniqueId(&id), "ncclGetUniqueId");
    std::memcpy(out, &id, sizeof(id));
}

// arena: this rank's weight arena on `device`; every rank calls with the same id / world / root
void comm_broadcast_arena(int device, void* arena, size_t bytes, const q3tts_comm_id& cid, int rank, int world, int root) {
    Rccl& r = rccl();
    Q3_CHECK(r.err.empty(), 7, r.err);
    Q3_CHECK(world >= 1 && rank >= 0 && rank < world && root >= 0 && root < world, 3, "Invalid input: rank / world / root of the broadcast");
    Q3_CHECK(arena && bytes > 0, 1, "Model not initialized: no weight arena to broadcast");
    Q3_HIP(hipSetDevice(device));
    NcclUniqueId id;
    std::memcpy(&id, &cid, sizeof(id));
    NcclComm comm = nullptr;
    nccl_check(r.CommInitRank(&comm, world, id, rank), "ncclCommInitRank");
    hipStream_t st = nullptr;
    try {
        Q3_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        // one message: 0.8-4.6 GB; a ring over xGMI is bound by one link (~153 GB/s): tens of milliseconds, once per load
        nccl_check(r.Broadcast(arena, arena, bytes, kNcclUint8, root, comm, st), "ncclBroadcast");
        Q3_HIP(hipStreamSynchronize(st));
    } catch (...) {
        if (st) (void)hipStreamDestroy(st);
        (void)r.CommDestroy(comm);
        throw;
    }
    (void)hipStreamDestroy(st);
    nccl_check(r.CommDestroy(comm), "ncclCommDestroy");
}

}  // namespace q3
