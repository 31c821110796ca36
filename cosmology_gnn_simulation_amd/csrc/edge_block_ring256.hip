// cgnn_edge_block, CGNN_BF16_N16 weights, latent = hidden = 256 (BASELINE cfg5; reference README.md:59-62 allows latent
// sizes up to 256): one round's edge update
//
//   e' = e + LayerNorm(W3 relu(W2 relu(Ps[src] + Pd[dst] + We e) + b2) + b3)          graph_network.py:89-90, :182
//
// with bf16 operands, f32 accumulation, f32 LayerNorm and residual -- the arithmetic of edge_block_n16_kernel
// (edge_block.hip), whose weights must be LDS resident: at 256 a layer is 128 KiB, three of them 384 KiB.  Here the
// layers stream through an eight-slot LDS ring of 16-KiB chunks (two 16-feature output tiles over K = 256) exactly as in
// edge_block_f2.hip / node_block_f2.hip (LDS-DMA seven chunks ahead, counted vmcnt waits, raw s_barrier per chunk, LDS
// fragment reads one MFMA group ahead across chunk boundaries).  16 edges per wave (v_mfma_f32_16x16x32_bf16), eight
// waves per workgroup (two per SIMD), 128 edges per step; the P rows (CGNN_P_BF16_S16) enter the accumulators through
// selector MFMAs.  Before this kernel the 256-wide edge update ran on the 32-row kernel with its weights read from L2 by
// every wave: 50.6 ms per round at cfg5's shape (32 M edges), a tenth of the MFMA peak.
//
// Registers set the order of the tail: 64 each for the accumulators, the f32 tile kept for the residual and the next
// tile's prefetch do not leave room for its P rows (32 + 32) as well, so the next tile's latents are requested first
// (into the registers the operand and the fragment pipeline have just vacated) and its P rows are requested one by one
// behind the stores, as the stores release the accumulator and tile registers.
#include <string.h>

#include "n16.hpp"

namespace cgnn {

int num_compute_units();   // runtime.hip

#define CGNN_R256_BLOCK 512
#define CGNN_R256_MAX_UNITS 4      // hidden layers <= 3

struct Ring256Args {
    const char* unit[CGNN_R256_MAX_UNITS];   // packed CGNN_BF16_N16: We (the e third of Linear 0), hidden..., output
    const float* bias[CGNN_R256_MAX_UNITS];  // bias[l] of Linear l = 1 .. nh (Linear 0's lives in Pd)
    const float* gamma;
    const float* beta;
    const __bf16* ps;                        // [n, 256] CGNN_P_BF16_S16
    const __bf16* pd;
    const int32_t* src;
    const int32_t* dst;
    const float* e_in;                       // CGNN_TILED32
    float* e_out;
    float* e_upd;
    int64_t num_edges;
    int64_t first_half_tile;
    int64_t half_tiles;
    int64_t steps;
    int32_t residual;
};

namespace r256 {
constexpr int D = 256, OT = 16, KS = 8;
constexpr int CF = 16;                        // fragments per chunk: two output tiles x eight k-steps
constexpr int CHUNK = CF * 1024;              // 16 KiB
constexpr int UNIT_CHUNKS = OT * KS / CF;     // 8
constexpr int NS = 8, PD = NS - 1;
constexpr int WAVES = CGNN_R256_BLOCK / 64;
constexpr int PC = CHUNK / 1024 / WAVES;
constexpr int VEC_BYTES = 8192;               // up to 8 vectors of 256 floats
constexpr int RING_OFF = VEC_BYTES;
constexpr int LDS_BYTES = RING_OFF + NS * CHUNK;    // 136 KiB

// s_mov + s_nop 3: five wait states between any VALU write of the base SGPRs (a v_readlane spill reload) and the load
// that reads them -- the compiler cannot pad inside an asm block (see f2_ring.hpp)
__device__ __forceinline__ void dma_piece(const char* sbase, unsigned voff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds)
                 : "memory");
}
template <int IMM>
__device__ __forceinline__ u32x4 load16(const void* p) {
    u32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(r) : "v"(p), "n"(IMM) : "memory");
    return r;
}
// wave-uniform base + one 32-bit lane offset: no 64-bit address per load (sixteen of them, computed ahead of the burst,
// were what pushed the kernel over its 256 registers)
__device__ __forceinline__ u32x4 load16_s(const void* base, unsigned voff) {
    // the base is wave-uniform by construction; say so (the compiler does not always see it through the tile arithmetic)
    const uint64_t b = (uint64_t)(uintptr_t)base;
    const uint64_t sb = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
    u32x4 r;
    // s_nop 4: the v_readfirstlane above is a VALU write of the SGPR pair this load reads (five wait states)
    // nt: the edge latents stream through once per round (32.8 GB at cfg5's shape); kept out of the caches they leave L2 to
    // the gathered P rows (loads and stores nt: 19.09 -> 18.52 ms per round)
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 nt" : "=v"(r) : "v"(voff), "s"(sb) : "memory");
    return r;
}
__device__ __forceinline__ int idx_load(const int32_t* p) {
    int r;
    asm volatile("global_load_dword %0, %1, off" : "=v"(r) : "v"(p) : "memory");
    return r;
}
// the wait that hands the prefetched tile over: at most N younger vector-memory operations may still be in flight
template <int N>
__device__ __forceinline__ void tile_ready(u32x4 (&e)[OT], u32x4 (&a)[KS], u32x4 (&b)[KS], int& s, int& d) {
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7])
                 : "n"(N)
                 : "memory");
    asm volatile("" : "+v"(e[8]), "+v"(e[9]), "+v"(e[10]), "+v"(e[11]), "+v"(e[12]), "+v"(e[13]), "+v"(e[14]), "+v"(e[15]));
    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
    asm volatile(""
                 : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]), "+v"(s),
                   "+v"(d));
}

// LDS fragment pipeline: a chunk holds [tile 0: k-steps 0..7][tile 1: k-steps 0..7]; a group = two k-steps of both tiles
struct FragPipe {
    u32x4 buf[2][4];     // [group % 2][(t0, 2G), (t0, 2G+1), (t1, 2G), (t1, 2G+1)]: one group ahead (registers)
    template <int SLOT, int G>
    __device__ __forceinline__ void request(unsigned addr) {
        buf[SLOT][0] = lds_read_b128<(0 * KS + 2 * G) * 1024>(addr);
        buf[SLOT][1] = lds_read_b128<(0 * KS + 2 * G + 1) * 1024>(addr);
        buf[SLOT][2] = lds_read_b128<(1 * KS + 2 * G) * 1024>(addr);
        buf[SLOT][3] = lds_read_b128<(1 * KS + 2 * G + 1) * 1024>(addr);
    }
    // the encoder's first Linear (K <= 32: one k-step): its chunk holds the sixteen fragments (o, k-step 0), o = 0 .. 15;
    // a group = four consecutive output tiles
    template <int SLOT, int G>
    __device__ __forceinline__ void request_lin(unsigned addr) {
        buf[SLOT][0] = lds_read_b128<(4 * G + 0) * 1024>(addr);
        buf[SLOT][1] = lds_read_b128<(4 * G + 1) * 1024>(addr);
        buf[SLOT][2] = lds_read_b128<(4 * G + 2) * 1024>(addr);
        buf[SLOT][3] = lds_read_b128<(4 * G + 3) * 1024>(addr);
    }
    template <int SLOT, int NEWER>
    __device__ __forceinline__ void run_lin(f32x4 (&c)[OT], bf16x8 in0, int o0) {
        lds_wait4<NEWER>(buf[SLOT][0], buf[SLOT][1], buf[SLOT][2], buf[SLOT][3]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            c[o0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, buf[SLOT][j]), in0, c[o0 + j], 0, 0, 0);
    }
    template <int SLOT, int NEWER>
    __device__ __forceinline__ void run(f32x4 (&c)[OT], const bf16x8 (&in)[KS], int o0, int g) {
        lds_wait4<NEWER>(buf[SLOT][0], buf[SLOT][1], buf[SLOT][2], buf[SLOT][3]);
        c[o0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, buf[SLOT][0]), in[2 * g], c[o0], 0, 0, 0);
        c[o0 + 1] =
            __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, buf[SLOT][2]), in[2 * g], c[o0 + 1], 0, 0, 0);
        c[o0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, buf[SLOT][1]), in[2 * g + 1], c[o0], 0, 0, 0);
        c[o0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, buf[SLOT][3]), in[2 * g + 1], c[o0 + 1],
                                                            0, 0, 0);
    }
};
}  // namespace r256

// Chunk Q of the step (see f2_ring.hpp for the protocol): the barrier vouches for chunks Q and Q + 1, chunk Q + PD starts
// into the slot chunk Q - 1 was read from, the reads of group g + 1 go out before group g's MFMAs.
#define CGNN_R256_CHUNK_O(Q, O0, C, OP)                                                                             \
    {                                                                                                               \
        if ((Q) + 1 >= PD) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PD - 2) * PC) : "memory");                      \
        asm volatile("s_barrier" ::: "memory");                                                                      \
        issue(((Q) + PD) % NC, slot == 0 ? NS - 1 : slot - 1);                                                       \
        const unsigned cur_ = ring_lds + slot * CHUNK + lane * 16;                                                   \
        slot = slot + 1 == NS ? 0 : slot + 1;                                                                        \
        const unsigned nxt_ = ring_lds + slot * CHUNK + lane * 16;                                                   \
        constexpr int o0_ = (O0);                                                                                    \
        constexpr bool last_ = (Q) == NC - 1;                                                                        \
        if ((Q) == 0) pipe.template request<0, 0>(cur_);                                                             \
        pipe.template request<1, 1>(cur_);                                                                           \
        pipe.template run<0, 4>(C, OP, o0_, 0);                                                                      \
        pipe.template request<0, 2>(cur_);                                                                           \
        pipe.template run<1, 4>(C, OP, o0_, 1);                                                                      \
        pipe.template request<1, 3>(cur_);                                                                           \
        pipe.template run<0, 4>(C, OP, o0_, 2);                                                                      \
        if (!last_) pipe.template request<0, 0>(nxt_);                                                               \
        pipe.template run<1, (last_ ? 0 : 4)>(C, OP, o0_, 3);                                                        \
    }
#define CGNN_R256_CHUNK(Q, C, OP) CGNN_R256_CHUNK_O(Q, 2 * ((Q) % UNIT_CHUNKS), C, OP)
#define CGNN_R256_UNIT(U, C, OP)                                                                               \
    CGNN_R256_CHUNK((U) * 8 + 0, C, OP) CGNN_R256_CHUNK((U) * 8 + 1, C, OP) CGNN_R256_CHUNK((U) * 8 + 2, C, OP)  \
    CGNN_R256_CHUNK((U) * 8 + 3, C, OP) CGNN_R256_CHUNK((U) * 8 + 4, C, OP) CGNN_R256_CHUNK((U) * 8 + 5, C, OP)  \
    CGNN_R256_CHUNK((U) * 8 + 6, C, OP) CGNN_R256_CHUNK((U) * 8 + 7, C, OP)

template <int NH, bool RAGGED>
__global__ __launch_bounds__(CGNN_R256_BLOCK) void edge_block_ring256_kernel(Ring256Args a) {
    using namespace r256;
    constexpr int NU = NH + 1, NC = NU * UNIT_CHUNKS;
    static_assert(NU <= CGNN_R256_MAX_UNITS && UNIT_CHUNKS == 8, "layer count / chunking");
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    {   // resident: bias of Linear 1 .. NH at vec[l], LayerNorm vectors behind them
        float* vec = reinterpret_cast<float*>(cgnn_smem);
        for (int i = threadIdx.x; i < D; i += blockDim.x) {
#pragma unroll
            for (int l = 1; l <= NH; ++l) vec[l * D + i] = a.bias[l][i];
            vec[(NH + 1) * D + i] = a.gamma[i];
            vec[(NH + 2) * D + i] = a.beta[i];
        }
    }
    __syncthreads();
    const LdsVecPtr vec = (LdsVecPtr)cgnn_smem;
    const unsigned ring_lds = (unsigned)(uintptr_t)(cgnn_smem + RING_OFF);

    const unsigned voff = (unsigned)wave * 1024u + (unsigned)lane * 16u;
    int slot = 0;
    auto issue = [&](int chunk, int into_slot) {
        const char* src = a.unit[chunk / UNIT_CHUNKS] + (chunk % UNIT_CHUNKS) * CHUNK;
#pragma unroll
        for (int i = 0; i < PC; ++i)
            dma_piece(src + i * (WAVES * 1024), voff, ring_lds + into_slot * CHUNK + (wave + WAVES * i) * 1024);
    };
#pragma unroll
    for (int i = 0; i < PD; ++i) issue(i, i);

    const int64_t last_ht = a.half_tiles - 1;
    auto edge_of = [&](int64_t ht) {
        const int64_t e = ht * 16 + c;
        return e < a.num_edges ? e : a.num_edges - 1;
    };
    auto clip = [&](int64_t ht) { return RAGGED && ht > last_ht ? last_ht : ht; };
    auto lane_off = [&](int64_t ht) { return (unsigned)n16_lane_offset(c, q, (int)(ht & 1)) * 4u; };   // bytes inside the tile

    const int nb = gridDim.x;
    int64_t step = blockIdx.x;
    u32x4 en[OT], psn[KS], pdn[KS];
    int sn, dn;
    {
        const int64_t ht = clip(a.first_half_tile + step * WAVES + wave);
        const int64_t ec = edge_of(ht);
        int s0 = idx_load(a.src + ec), d0 = idx_load(a.dst + ec);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(s0), "+v"(d0)::"memory");
        const float* ep = a.e_in + (ht >> 1) * (32 * D);         // wave-uniform tile base
        const __bf16* pp = a.ps + (int64_t)s0 * D + 8 * q;
        const __bf16* dp = a.pd + (int64_t)d0 * D + 8 * q;
        static_for_each([&](auto oc) {
            en[decltype(oc)::value] = load16_s(ep + n16_tile_offset(decltype(oc)::value), lane_off(ht));
        }, std::make_integer_sequence<int, OT>{});
        static_for_each([&](auto sc) { psn[decltype(sc)::value] = load16<decltype(sc)::value * 64>(pp); },
                        std::make_integer_sequence<int, KS>{});
        static_for_each([&](auto sc) { pdn[decltype(sc)::value] = load16<decltype(sc)::value * 64>(dp); },
                        std::make_integer_sequence<int, KS>{});
        const int64_t nstep = step + nb < a.steps ? step + nb : step;
        const int64_t ec1 = edge_of(clip(a.first_half_tile + nstep * WAVES + wave));
        sn = idx_load(a.src + ec1);
        dn = idx_load(a.dst + ec1);
        tile_ready<0>(en, psn, pdn, sn, dn);
    }

    for (; step < a.steps; step += nb) {
        const int64_t ht_raw = a.first_half_tile + step * WAVES + wave;
        const bool valid = !RAGGED || ht_raw <= last_ht;           // wave-uniform
        const int64_t ht = clip(ht_raw);
        const int64_t next_step = step + nb < a.steps ? step + nb : step;
        const int64_t next2_step = next_step + nb < a.steps ? next_step + nb : next_step;
        const int64_t ht1 = clip(a.first_half_tile + next_step * WAVES + wave);

        FragPipe pipe;
        f32x4 ev[OT], acc[OT];
        bf16x8 op[KS];
#pragma unroll
        for (int o = 0; o < OT; ++o) ev[o] = __builtin_bit_cast(f32x4, en[o]);
        {
            bf16x8 pso[KS], pdo[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                pso[s] = __builtin_bit_cast(bf16x8, psn[s]);
                pdo[s] = __builtin_bit_cast(bf16x8, pdn[s]);
            }
            // the two constant selector fragments are rebuilt here every step (a dozen vector instructions) instead of
            // living in eight registers across it: the kernel sits at the 256-register limit
            int lane_ = lane;
            asm volatile("" : "+v"(lane_));
            p16_accumulate<KS>(acc, pso, pdo, p16_selector(lane_, 0), p16_selector(lane_, 1));      // Linear 0's node thirds (+ b1)
        }
        operand16<false, KS>(op, ev);
        CGNN_R256_UNIT(0, acc, op)
        operand16<true, KS>(op, acc);
        if constexpr (NH >= 2) {
            fill16<OT>(acc, vec + 1 * D, q);
            CGNN_R256_UNIT(1, acc, op)
            operand16<true, KS>(op, acc);
        }
        if constexpr (NH >= 3) {
            fill16<OT>(acc, vec + 2 * D, q);
            CGNN_R256_UNIT(2, acc, op)
            operand16<true, KS>(op, acc);
        }
        fill16<OT>(acc, vec + NH * D, q);
        CGNN_R256_UNIT(NU - 1, acc, op)

        // ---- tail ----
        // The lane's constants (row in the tile, feature quarter, byte offset in a tile) are recomputed here from the lane
        // id instead of living in registers through the MFMA units: at the 256-register limit the compiler otherwise
        // spills them, and each reload waits vmcnt(0), draining the prefetches below.
        int tl = threadIdx.x & 63;
        asm volatile("" : "+v"(tl));
        const int tc = tl & 15, tq = tl >> 4;
        const __bf16* pp = a.ps + (int64_t)sn * D + 8 * tq;     // the next tile's P rows (indices from a step ago)
        const __bf16* dp = a.pd + (int64_t)dn * D + 8 * tq;
        {
            const int64_t e2 = clip(a.first_half_tile + next2_step * WAVES + wave) * 16 + tc;
            const int64_t ec2 = e2 < a.num_edges ? e2 : a.num_edges - 1;
            sn = idx_load(a.src + ec2);                         // indices of the tile after it
            dn = idx_load(a.dst + ec2);
        }
        {
            const float* ep = a.e_in + (ht1 >> 1) * (32 * D);
            const unsigned loff = (unsigned)n16_lane_offset(tc, tq, (int)(ht1 & 1)) * 4u;
            static_for_each([&](auto oc) {
                en[decltype(oc)::value] = load16_s(ep + n16_tile_offset(decltype(oc)::value), loff);
            }, std::make_integer_sequence<int, OT>{});
        }
        layer_norm16<OT>(acc, vec + (NH + 1) * D, vec + (NH + 2) * D, tq);
        const int64_t tb = (ht >> 1) * (32 * D) + n16_lane_offset(tc, tq, (int)(ht & 1));
        if (valid && a.e_upd != nullptr) {   // block-uniform pointer
#pragma unroll
            for (int o = 0; o < OT; ++o) *reinterpret_cast<f32x4*>(a.e_upd + tb + n16_tile_offset(o)) = acc[o];
        }
        // stores and the next tile's P-row requests alternate (a store releases the registers the next request lands in);
        // the last request goes out before the last store, so that the closing wait can leave that store in flight
        static_for_each([&](auto oc) {
            constexpr int o = decltype(oc)::value;
            if constexpr (o == OT - 1) pdn[KS - 1] = load16<(KS - 1) * 64>(dp);
            if (a.residual) acc[o] += ev[o];
            if (valid) __builtin_nontemporal_store(acc[o], reinterpret_cast<f32x4*>(a.e_out + tb + n16_tile_offset(o)));
            if constexpr (o < KS) psn[o] = load16<o * 64>(pp);
            else if constexpr (o < OT - 1) pdn[o - KS] = load16<(o - KS) * 64>(dp);
        }, std::make_integer_sequence<int, OT>{});
        if (RAGGED)
            tile_ready<0>(en, psn, pdn, sn, dn);      // a wave past the end issues no stores
        else
            tile_ready<1>(en, psn, pdn, sn, dn);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// ---- the edge encoder at latent = hidden = 256 (reference graph_network.py:57: MLP + LayerNorm of the edge features) ----
// Same ring, same 16-edge tiles, no P rows, no residual: chunk 0 is the whole first Linear (K <= 32 -> one k-step, sixteen
// output tiles), then the hidden and output Linears as eight-chunk units.  The per-edge input is one aligned 16-byte
// load (<= 4 features), requested a step ahead.  Before it the 256-wide encoder ran on the 32-row kernel with every
// wave streaming its weights from L2: 31 ms at cfg5's shape (32 M edges).
struct Enc256Args {
    const char* unit[CGNN_R256_MAX_UNITS];   // packed CGNN_BF16_N16: Linear 0 (one chunk), hidden..., output
    const float* bias[CGNN_R256_MAX_UNITS];
    const float* gamma;
    const float* beta;
    const float* x;                          // [n, ld_x] edge features
    float* y;                                // CGNN_TILED32 edge latents
    int64_t n;
    int64_t first_half_tile, half_tiles, steps;
    int32_t ld_x, in_dim;
};

#define CGNN_R256E_CHUNK0(C, IN0)                                                                                   \
    {                                                                                                               \
        asm volatile("s_barrier" ::: "memory");                                                                      \
        issue(PD % NC, slot == 0 ? NS - 1 : slot - 1);                                                               \
        const unsigned cur_ = ring_lds + slot * CHUNK + lane * 16;                                                   \
        slot = slot + 1 == NS ? 0 : slot + 1;                                                                        \
        const unsigned nxt_ = ring_lds + slot * CHUNK + lane * 16;                                                   \
        pipe.template request_lin<0, 0>(cur_);                                                                       \
        pipe.template request_lin<1, 1>(cur_);                                                                       \
        pipe.template run_lin<0, 4>(C, IN0, 0);                                                                      \
        pipe.template request_lin<0, 2>(cur_);                                                                       \
        pipe.template run_lin<1, 4>(C, IN0, 4);                                                                      \
        pipe.template request_lin<1, 3>(cur_);                                                                       \
        pipe.template run_lin<0, 4>(C, IN0, 8);                                                                      \
        pipe.template request<0, 0>(nxt_);                                                                           \
        pipe.template run_lin<1, 4>(C, IN0, 12);                                                                     \
    }
// unit U >= 1 of the encoder = chunks 1 + 8 (U - 1) .. 8 U (chunk 0 is the first Linear)
#define CGNN_R256E_CH(U, I, C, OP) CGNN_R256_CHUNK_O(1 + 8 * ((U)-1) + (I), 2 * (I), C, OP)
#define CGNN_R256E_UNIT(U, C, OP)                                                                       \
    CGNN_R256E_CH(U, 0, C, OP) CGNN_R256E_CH(U, 1, C, OP) CGNN_R256E_CH(U, 2, C, OP) CGNN_R256E_CH(U, 3, C, OP) \
    CGNN_R256E_CH(U, 4, C, OP) CGNN_R256E_CH(U, 5, C, OP) CGNN_R256E_CH(U, 6, C, OP) CGNN_R256E_CH(U, 7, C, OP)

template <int NH, bool RAGGED>
__global__ __launch_bounds__(CGNN_R256_BLOCK) void edge_encode_ring256_kernel(Enc256Args a) {
    using namespace r256;
    constexpr int NC = 1 + NH * UNIT_CHUNKS;
    static_assert(NH + 1 <= CGNN_R256_MAX_UNITS && UNIT_CHUNKS == 8 && NC > PD, "layer count / chunking");
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    {   // resident: bias of Linear 0 .. NH at vec[l], LayerNorm vectors behind them
        float* vec = reinterpret_cast<float*>(cgnn_smem);
        for (int i = threadIdx.x; i < D; i += blockDim.x) {
#pragma unroll
            for (int l = 0; l <= NH; ++l) vec[l * D + i] = a.bias[l][i];
            vec[(NH + 1) * D + i] = a.gamma[i];
            vec[(NH + 2) * D + i] = a.beta[i];
        }
    }
    __syncthreads();
    const LdsVecPtr vec = (LdsVecPtr)cgnn_smem;
    const unsigned ring_lds = (unsigned)(uintptr_t)(cgnn_smem + RING_OFF);

    const unsigned voff = (unsigned)wave * 1024u + (unsigned)lane * 16u;
    int slot = 0;
    auto issue = [&](int chunk, int into_slot) {
        const char* src = chunk == 0 ? a.unit[0] : a.unit[1 + (chunk - 1) / UNIT_CHUNKS] + ((chunk - 1) % UNIT_CHUNKS) * CHUNK;
#pragma unroll
        for (int i = 0; i < PC; ++i)
            dma_piece(src + i * (WAVES * 1024), voff, ring_lds + into_slot * CHUNK + (wave + WAVES * i) * 1024);
    };
#pragma unroll
    for (int i = 0; i < PD; ++i) issue(i, i);

    const int64_t last_ht = a.half_tiles - 1;
    auto row_of = [&](int64_t ht) {      // the lane's edge: its feature row's address
        const int64_t ht_c = RAGGED && ht > last_ht ? last_ht : ht;
        const int64_t e = ht_c * 16 + c;
        return a.x + (e < a.n ? e : a.n - 1) * a.ld_x;
    };
    const int nb = gridDim.x;
    int64_t step = blockIdx.x;
    u32x4 xn = load16<0>(row_of(a.first_half_tile + step * WAVES + wave));
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(xn)::"memory");

    for (; step < a.steps; step += nb) {
        const int64_t ht_raw = a.first_half_tile + step * WAVES + wave;
        const bool valid = !RAGGED || ht_raw <= last_ht;           // wave-uniform
        const int64_t ht = RAGGED && ht_raw > last_ht ? last_ht : ht_raw;
        const int64_t next_step = step + nb < a.steps ? step + nb : step;

        FragPipe pipe;
        f32x4 acc[OT];
        bf16x8 op[KS];
        bf16x8 in0;
        {   // k-step 0 of the N16 operand: element j of lane (c, q) is feature 16 (j >> 2) + 4 q + (j & 3): features 0 .. 3 sit
            // in elements 0 .. 3 of the q = 0 lanes
            const f32x4 xv = __builtin_bit_cast(f32x4, xn);
#pragma unroll
            for (int j = 0; j < 8; ++j) in0[j] = (__bf16)((j < 4 && q == 0 && j < a.in_dim) ? xv[j & 3] : 0.f);
        }
        fill16<OT>(acc, vec, q);
        CGNN_R256E_CHUNK0(acc, in0)
        operand16<true, KS>(op, acc);
        if constexpr (NH >= 2) {
            fill16<OT>(acc, vec + 1 * D, q);
            CGNN_R256E_UNIT(1, acc, op)
            operand16<true, KS>(op, acc);
        }
        if constexpr (NH >= 3) {
            fill16<OT>(acc, vec + 2 * D, q);
            CGNN_R256E_UNIT(2, acc, op)
            operand16<true, KS>(op, acc);
        }
        fill16<OT>(acc, vec + NH * D, q);
        CGNN_R256E_UNIT(NH, acc, op)

        // ---- tail: the next tile's features, LayerNorm, stores ----
        int tl = threadIdx.x & 63;
        asm volatile("" : "+v"(tl));
        const int tc = tl & 15, tq = tl >> 4;
        xn = load16<0>(row_of(a.first_half_tile + next_step * WAVES + wave));
        layer_norm16<OT>(acc, vec + (NH + 1) * D, vec + (NH + 2) * D, tq);
        const int64_t tb = (ht >> 1) * (32 * D) + n16_lane_offset(tc, tq, (int)(ht & 1));
        if (valid) {
#pragma unroll
            for (int o = 0; o < OT; ++o) *reinterpret_cast<f32x4*>(a.y + tb + n16_tile_offset(o)) = acc[o];
        }
        // the feature load is older than the stores: everything behind it may stay in flight (a wave past the end of a
        // ragged step issues no stores: it drains)
        if (RAGGED)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(xn)::"memory");
        else
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(xn) : "n"(OT) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

template <int NH, bool RAGGED>
static int launch_enc256(const Enc256Args& a, hipStream_t st) {
    auto kern = edge_encode_ring256_kernel<NH, RAGGED>;
    int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (size_t)(r256::LDS_BYTES), "hipFuncSetAttribute(edge_encode_ring256)");
    if (rc != CGNN_OK) return rc;
    const int grid = (int)(a.steps < (int64_t)num_compute_units() ? a.steps : (int64_t)num_compute_units());
    kern<<<grid, CGNN_R256_BLOCK, r256::LDS_BYTES, st>>>(a);
    return check_hip(hipGetLastError(), "cgnn_mlp_rows(ring 256 encoder) launch");
}

// Called by cgnn_mlp_rows (through mlp_rows_n16_encoder, edge_block.hip) for a CGNN_BF16_N16 encoder with latent = hidden =
// 256.  *handled = 0: the shape or the feature layout is not this kernel's (nothing launched).
int edge_encode_ring256(const MlpDev& m, const float* x, int64_t n, int ld_x, float* y, hipStream_t st, int* handled) {
    *handled = 0;
    if (m.nh < 1 || m.nh > 3 || !m.gamma || !m.beta || m.in_dim[0] > 4 || (ld_x & 3) != 0 || ((uintptr_t)x & 15) != 0 || n <= 0)
        return CGNN_OK;
    Enc256Args a;
    memset(&a, 0, sizeof(a));
    for (int l = 0; l <= m.nh; ++l) {
        a.unit[l] = reinterpret_cast<const char*>(m.w[l]);
        a.bias[l] = m.b[l];
        if (!m.b[l]) return CGNN_OK;
    }
    a.gamma = m.gamma;
    a.beta = m.beta;
    a.x = x;
    a.y = y;
    a.n = n;
    a.ld_x = ld_x;
    a.in_dim = m.in_dim[0];
    a.half_tiles = 2 * ((n + 31) / 32);
    const int64_t full = a.half_tiles / 8;
    int rc = CGNN_OK;
#define CGNN_GO(NHh, RAG) \
    if (rc == CGNN_OK && m.nh == NHh) rc = launch_enc256<NHh, RAG>(a, st);
    if (full > 0) {
        a.first_half_tile = 0;
        a.steps = full;
        CGNN_GO(1, false) CGNN_GO(2, false) CGNN_GO(3, false)
    }
    if (rc == CGNN_OK && a.half_tiles % 8 != 0) {
        a.first_half_tile = full * 8;
        a.steps = 1;
        CGNN_GO(1, true) CGNN_GO(2, true) CGNN_GO(3, true)
    }
#undef CGNN_GO
    if (rc == CGNN_OK) *handled = 1;
    return rc;
}

template <int NH, bool RAGGED>
static int launch_ring256(const Ring256Args& a, hipStream_t st) {
    auto kern = edge_block_ring256_kernel<NH, RAGGED>;
    int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (size_t)(r256::LDS_BYTES), "hipFuncSetAttribute(edge_block_ring256)");
    if (rc != CGNN_OK) return rc;
    const int grid = (int)(a.steps < (int64_t)num_compute_units() ? a.steps : (int64_t)num_compute_units());
    kern<<<grid, CGNN_R256_BLOCK, r256::LDS_BYTES, st>>>(a);
    return check_hip(hipGetLastError(), "cgnn_edge_block(ring 256) launch");
}

// Called by cgnn_edge_block (edge_block.hip) for CGNN_BF16_N16 models with latent = hidden = 256.
int edge_block_ring256(const MlpDev& m, const __bf16* ps, const __bf16* pd, const int32_t* src, const int32_t* dst,
                       int64_t num_edges, const float* e_in, float* e_out, float* e_upd, int residual, hipStream_t st) {
    if (m.nh < 1 || m.nh > 3 || !m.gamma || !m.beta) {
        set_error("cgnn_edge_block: the 256-wide CGNN_BF16_N16 kernel needs 1..3 hidden layers and LayerNorm");
        return CGNN_ERR_UNSUPPORTED;
    }
    Ring256Args a;
    memset(&a, 0, sizeof(a));
    for (int l = 0; l <= m.nh; ++l) {
        a.unit[l] = reinterpret_cast<const char*>(m.w[l]);
        a.bias[l] = m.b[l];
        if (l >= 1 && !m.b[l]) {
            set_error("cgnn_edge_block: the 256-wide CGNN_BF16_N16 kernel needs a bias on every Linear");
            return CGNN_ERR_UNSUPPORTED;
        }
    }
    a.gamma = m.gamma;
    a.beta = m.beta;
    a.ps = ps;
    a.pd = pd;
    a.src = src;
    a.dst = dst;
    a.e_in = e_in;
    a.e_out = e_out;
    a.e_upd = e_upd;
    a.num_edges = num_edges;
    a.residual = residual;
    a.half_tiles = 2 * ((num_edges + 31) / 32);
    const int64_t full = a.half_tiles / 8;
    int rc = CGNN_OK;
#define CGNN_GO(NHh, RAG) \
    if (rc == CGNN_OK && m.nh == NHh) rc = launch_ring256<NHh, RAG>(a, st);
    if (full > 0) {
        a.first_half_tile = 0;
        a.steps = full;
        CGNN_GO(1, false) CGNN_GO(2, false) CGNN_GO(3, false)
    }
    if (rc == CGNN_OK && a.half_tiles % 8 != 0) {
        a.first_half_tile = full * 8;
        a.steps = 1;
        CGNN_GO(1, true) CGNN_GO(2, true) CGNN_GO(3, true)
    }
#undef CGNN_GO
    return rc;
}

}  // namespace cgnn
