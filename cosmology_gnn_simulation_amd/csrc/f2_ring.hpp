// Shared by the two-fp16-term ring kernels (node_block_f2.hip, edge_block_f2.hip): 512-thread workgroups, 16 rows per
// wave, latent = hidden = 128; the weights of a step stream through an LDS ring of 16-KiB chunks (two 16-feature output
// tiles over K = 128, packed CGNN_F16X2_N16) by LDS-DMA, `PD` chunks ahead, with counted waits.  See node_block_f2.hip
// for the protocol.  The chunk macros expect in scope: NS, PD, NC (constants), slot, ring_lds, lane, pipe and the
// lambda issue(chunk, slot).
#pragma once
#include "n16.hpp"

namespace cgnn {

#define CGNN_F2R_BLOCK 512

typedef __attribute__((address_space(3))) f32x4* LdsF4Ptr;
typedef __attribute__((address_space(3))) u32x4* LdsU4Ptr;
typedef __attribute__((address_space(3))) bf16x4* LdsB4Ptr;

namespace f2r {
constexpr int D = 128, OT = 8, KS = 4;
constexpr int CF = 8;                         // fragments per chunk: two output tiles x four k-steps
constexpr int CHUNK = CF * 2048;              // 16 KiB
constexpr int UNIT_CHUNKS = OT * KS / CF;     // 4
constexpr int WAVES = CGNN_F2R_BLOCK / 64;
constexpr int PC = CHUNK / 1024 / WAVES;      // 1-KiB DMA pieces per wave per chunk
constexpr int VEC_BYTES = 4096;               // up to 8 vectors of 128 floats, first thing in LDS
static_assert(PC * 1024 * WAVES == CHUNK, "a chunk is a whole number of pieces per wave");

// one 1-KiB piece: 64 lanes x 16 B from sbase + voff to LDS address lds (wave-uniform) + lane * 16.
// The s_nop is not decoration.  A vector-memory instruction that reads an SGPR needs five wait states after a VALU
// instruction wrote it (v_readlane reloading a spilled base pointer, v_readfirstlane); the compiler pads that for its
// own instructions but cannot see inside an asm block, and with ~40 chunk base pointers live it does keep some of them
// in VGPR lanes.  s_mov + s_nop 3 are the five states whatever stands before the block
// (scripts/dev/scan_asm_hazards.py checks the listing; tests/test_host_logic.py runs it).
__device__ __forceinline__ void dma_piece(const char* sbase, unsigned voff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds)
                 : "memory");
}

template <int IMM>
__device__ __forceinline__ f32x4 row_load(const float* p) {
    f32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(r) : "v"(p), "n"(IMM) : "memory");
    return r;
}

// the wait that hands the prefetched rows over: at most N younger vector-memory operations may still be in flight
template <int N>
__device__ __forceinline__ void rows_ready(f32x4 (&a)[OT], f32x4 (&b)[OT]) {
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                 : "n"(N)
                 : "memory");
    asm volatile("" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
}
}  // namespace f2r

#ifdef CGNN_F2R_STAMPS   // developer build: cycle stamps of one workgroup's waves over one step (printed by the launcher)
__device__ unsigned long long cgnn_f2r_stamps[8 * 64];
#define F2R_STAMP(k)                                                             \
    if (stamp_on && lane == 0) {                                                 \
        __builtin_amdgcn_sched_barrier(0);                                       \
        cgnn_f2r_stamps[wave * 64 + (k)] = __builtin_readcyclecounter();         \
        __builtin_amdgcn_sched_barrier(0);                                       \
    }
#else
#define F2R_STAMP(k)
#endif


#define F2R_ISSUE(C, S) issue(C, S)
#define F2R_CHUNK_WAIT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#define F2R_BARRIER() asm volatile("s_barrier" ::: "memory")
#define F2R_SPLIT(RELU, OP, SRC) operand16f2<RELU, KS>(OP, SRC)
// Chunk Q of the step.  The barrier vouches for chunks Q and Q + 1 (so that the fragment reads can run into the next
// chunk), then chunk Q + PD starts into the slot chunk Q - 1 was read from.  Group g of the chunk is k-step g of its two
// output tiles; the reads of group g + 2 (of this chunk or the next) go out before group g's MFMAs.
#define CGNN_F2R_CHUNK(Q, C0, C1, OP)                                                                               \
    {                                                                                                               \
        if ((Q) + 1 >= PD) F2R_CHUNK_WAIT((PD - 2) * PC);                                                            \
        F2R_BARRIER();                                                                                               \
        F2R_ISSUE(((Q) + PD) % NC, slot == 0 ? NS - 1 : slot - 1);                                                   \
        const unsigned cur_ = ring_lds + slot * CHUNK + lane * 16;                                                   \
        slot = slot + 1 == NS ? 0 : slot + 1;                                                                        \
        const unsigned nxt_ = ring_lds + slot * CHUNK + lane * 16;                                                   \
        constexpr int o0_ = 2 * ((Q) % UNIT_CHUNKS), g0_ = 4 * (Q);                                                  \
        constexpr bool last_ = (Q) == NC - 1;                                                                        \
        if ((Q) == 0) {                                                                                              \
            pipe.template request<0, 0>(cur_);                                                                       \
            pipe.template request<1, 1>(cur_);                                                                       \
        }                                                                                                            \
        pipe.template request<(g0_ + 2) % 3, 2>(cur_);                                                               \
        pipe.template run<(g0_ + 0) % 3, 8, OT, KS>(C0, C1, OP, o0_, 0);                                             \
        pipe.template request<(g0_ + 3) % 3, 3>(cur_);                                                               \
        pipe.template run<(g0_ + 1) % 3, 8, OT, KS>(C0, C1, OP, o0_, 1);                                             \
        if (!last_) pipe.template request<(g0_ + 4) % 3, 0>(nxt_);                                                   \
        pipe.template run<(g0_ + 2) % 3, (last_ ? 4 : 8), OT, KS>(C0, C1, OP, o0_, 2);                               \
        if (!last_) pipe.template request<(g0_ + 5) % 3, 1>(nxt_);                                                   \
        pipe.template run<(g0_ + 3) % 3, (last_ ? 0 : 8), OT, KS>(C0, C1, OP, o0_, 3);                               \
    }
#define CGNN_F2R_UNIT(U, C0, C1, OP)                                                                   \
    CGNN_F2R_CHUNK((U) * UNIT_CHUNKS + 0, C0, C1, OP) CGNN_F2R_CHUNK((U) * UNIT_CHUNKS + 1, C0, C1, OP) \
    CGNN_F2R_CHUNK((U) * UNIT_CHUNKS + 2, C0, C1, OP) CGNN_F2R_CHUNK((U) * UNIT_CHUNKS + 3, C0, C1, OP)


}  // namespace cgnn
