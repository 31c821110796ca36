// Two-slot LDS ring through which the node kernels stream their (three-term) weights: chunk i+1 is copied by
// LDS-DMA (global_load_lds_dwordx4: no registers, asynchronous) into one half of the ring while every wave of
// the workgroup runs the MFMAs of chunk i out of the other half; one barrier per chunk.
#pragma once
#include "mlp_device.hpp"

namespace cgnn {

#define CGNN_X3_CHUNK_FRAGS 16
#define CGNN_X3_CHUNK_BYTES (CGNN_X3_CHUNK_FRAGS * 3 * 1024)
#define CGNN_X3_MAX_CHUNKS 24

struct X3Chunks {
    const char* src[CGNN_X3_MAX_CHUNKS];   // packed bytes of each chunk, in consumption order
    uint32_t bytes[CGNN_X3_MAX_CHUNKS];
    int32_t count;
};

typedef __attribute__((address_space(3))) void* LdsVoidPtr;
typedef const __attribute__((address_space(1))) void* GlobalVoidPtr;

template <int SLOT_BYTES, int SLOTS>
struct WeightRingT {
    const X3Chunks& c;
    int wave, lane, next;   // next = index of the chunk to be consumed next (its DMA is already in flight)
    __device__ __forceinline__ WeightRingT(const X3Chunks& cc, int w, int l) : c(cc), wave(w), lane(l), next(0) {}
    // every wave copies its share of chunk `idx` (1-KiB pieces wave, wave + #waves, ...) into ring slot idx & 1
    __device__ __forceinline__ void issue(int idx) const {
        const char* src = c.src[idx];
        const uint32_t nb = c.bytes[idx];
        char* dst = cgnn_smem + (idx % SLOTS) * SLOT_BYTES;
        const uint32_t step = (blockDim.x >> 6) * 1024u;
        for (uint32_t off = wave * 1024u; off < nb; off += step)
            __builtin_amdgcn_global_load_lds((GlobalVoidPtr)(src + off + lane * 16), (LdsVoidPtr)(dst + off), 16, 0, 0);
    }
    // Make chunk `next` readable, start the copy of the one after it, return its LDS base.
    __device__ __forceinline__ LdsWeightPtr acquire(bool more_tiles) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces have landed
        __syncthreads();                                    // ... and everybody else's; slot (next+1)&1 is free
        const int cur = next;
        next = (next + 1 == c.count) ? 0 : next + 1;
        if (next != 0 || more_tiles) issue(next);
        return (LdsWeightPtr)(cgnn_smem + (cur % SLOTS) * SLOT_BYTES);
    }
};
typedef WeightRingT<CGNN_X3_CHUNK_BYTES, 2> WeightRing;

}  // namespace cgnn
