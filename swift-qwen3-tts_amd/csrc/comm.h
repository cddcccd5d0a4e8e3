// comm.h -- the load-time weight broadcast and the arena checksum behind q3tts_model_broadcast (csrc/comm.cc, kernels/checksum.hip).
#pragma once
#include <cstddef>
#include <cstdint>

#include "../../include/q3tts.h"

namespace q3 {
void comm_unique_id(q3tts_comm_id* out);
void comm_broadcast_arena(int device, void* arena, size_t bytes, const q3tts_comm_id& id, int rank, int world, int root);
uint64_t arena_checksum(int device, const void* arena, size_t bytes);
}  // namespace q3
