// EXPERIMENT RECORD (round 4) -- not part of the library.  Measured on one MI355X at 1 M rows (gpurun_out/r04_i_node.log):
// node block + projections 0.842 ms against 0.791 ms for node_block_f2ring_kernel, 0.735-0.768 against 0.665 without the
// projections; all 60 tests of tests/test_gpu_node_block.py / test_gpu_ring_kernels_repeat.py passed with it.  The MFMA units
// do get their LDS traffic halved, but with ONE wave per SIMD nothing runs under the tail (row loads, LayerNorm, the
// LDS-transposed stores, the projection epilogue): what eight waves overlapped by themselves is serial here.  To build it
// again: move F2RingArgs / CGNN_F2R_MAX_UNITS from node_block_f2.hip into f2_ring.hpp, add this file to csrc/Makefile's SRCS
// and call node_block_f2w4_launch from node_block_f2ring().  DESIGN.md section 6 (round 4) has the numbers.
//
// cgnn_node_block, CGNN_F16X2_N16 weights, latent = hidden = 128, second form of the five-slot-ring kernel
// (node_block_f2.hip): ONE wave per SIMD, TWO 16-row tiles per wave that share every weight fragment.
//
// Why.  In node_block_f2ring_kernel (eight waves, 16 rows each) a wave reads a 2-KiB two-term fragment from LDS for every
// 48 cycles of matrix work: 8 waves x 16 KiB per chunk plus the ring's own 16-KiB LDS-DMA write = 144 KiB per 768 matrix
// cycles, 75 % of the LDS port -- a unit took 1.2 k cycles per chunk instead of 0.77 k.
// Here a wave owns 32 rows as two MFMA column tiles: the fragment it has just read feeds twelve v_mfma_f32_16x16x32_f16
// instead of six, the workgroup is four waves (256 threads) with all 512 registers each, and the LDS traffic per matrix
// cycle halves (80 KiB per 768 cycles).  Same ring, same chunk images, same arithmetic per row, instruction for
// instruction: results are bit-identical to node_block_f2ring_kernel (tests/test_gpu_node_block.py).
//
// Registers.  Per tile: operand 32, two accumulators 64, the tile's f32 rows (residual) 32, the next tile's x / agg rows
// 64; twice that per wave plus 48 of weight fragments: more than the 256 registers vector instructions can address.  The
// prefetched rows live in the accumulation half: they are loaded there (inline asm, "a" operands), handed over behind a
// counted wait and only then read by the operand split; nothing the hardware is still writing is ever copied.
#include <string.h>

#include "f2_ring.hpp"

namespace cgnn {

int num_compute_units();   // runtime.hip

namespace f2w {
constexpr int WAVES = 4, BLOCK = WAVES * 64;
constexpr int TILES = 2;                        // 16-row tiles per wave
constexpr int NS = 5, PD = NS - 1;              // ring slots, chunks in flight ahead of the one being read
constexpr int PC = f2r::CHUNK / 1024 / WAVES;   // 1-KiB DMA pieces per wave per chunk (4)
constexpr int PROJ_BYTES = 2 * f2r::OT * f2r::KS * 1024;
constexpr int RING_OFF = f2r::VEC_BYTES + PROJ_BYTES;
constexpr int LDS_BYTES = RING_OFF + NS * f2r::CHUNK;
static_assert(PC * 1024 * WAVES == f2r::CHUNK, "a chunk is a whole number of pieces per wave");

// a row piece into the accumulation half of the register file (see the header)
template <int IMM>
__device__ __forceinline__ f32x4 row_load_a(const float* p) {
    f32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=a"(r) : "v"(p), "n"(IMM) : "memory");
    return r;
}
// the wait that hands the prefetched rows over: at most N younger vector-memory operations may still be in flight
template <int N>
__device__ __forceinline__ void rows_ready_a(f32x4 (&a)[f2r::OT], f32x4 (&b)[f2r::OT], f32x4 (&c)[f2r::OT], f32x4 (&d)[f2r::OT]) {
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+a"(a[0]), "+a"(a[1]), "+a"(a[2]), "+a"(a[3]), "+a"(a[4]), "+a"(a[5]), "+a"(a[6]), "+a"(a[7])
                 : "n"(N)
                 : "memory");
    asm volatile("" : "+a"(b[0]), "+a"(b[1]), "+a"(b[2]), "+a"(b[3]), "+a"(b[4]), "+a"(b[5]), "+a"(b[6]), "+a"(b[7]));
    asm volatile("" : "+a"(c[0]), "+a"(c[1]), "+a"(c[2]), "+a"(c[3]), "+a"(c[4]), "+a"(c[5]), "+a"(c[6]), "+a"(c[7]));
    asm volatile("" : "+a"(d[0]), "+a"(d[1]), "+a"(d[2]), "+a"(d[3]), "+a"(d[4]), "+a"(d[5]), "+a"(d[6]), "+a"(d[7]));
}

// FragPipe16f2 (n16.hpp) for two column tiles: the group's four fragment registers feed both tiles' MFMAs.  Per tile and
// accumulator the products come in FragPipe16f2::run's order (c1[o]: hi.lo' then lo.hi'): the same bits.
struct FragPipe2 {
    u32x4 buf[3][4];     // [group % 3][hi(O0), lo(O0), hi(O0+1), lo(O0+1)]
    template <int SLOT, int G>
    __device__ __forceinline__ void request(unsigned addr) {
        buf[SLOT][0] = lds_read_b128<(0 * 4 + G) * 2048>(addr);
        buf[SLOT][1] = lds_read_b128<(0 * 4 + G) * 2048 + 1024>(addr);
        buf[SLOT][2] = lds_read_b128<(1 * 4 + G) * 2048>(addr);
        buf[SLOT][3] = lds_read_b128<(1 * 4 + G) * 2048 + 1024>(addr);
    }
    template <int SLOT, int NEWER, int OT, int KS>
    __device__ __forceinline__ void run(f32x4 (&c0)[TILES][OT], f32x4 (&c1)[TILES][OT], const f16x8 (&in)[TILES][2][KS], int o0,
                                        int s) {
        lds_wait4<NEWER>(buf[SLOT][0], buf[SLOT][1], buf[SLOT][2], buf[SLOT][3]);
        const f16x8 h0 = __builtin_bit_cast(f16x8, buf[SLOT][0]), l0 = __builtin_bit_cast(f16x8, buf[SLOT][1]);
        const f16x8 h1 = __builtin_bit_cast(f16x8, buf[SLOT][2]), l1 = __builtin_bit_cast(f16x8, buf[SLOT][3]);
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            c0[t][o0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, in[t][0][s], c0[t][o0], 0, 0, 0);
            c0[t][o0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, in[t][0][s], c0[t][o0 + 1], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            c1[t][o0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, in[t][1][s], c1[t][o0], 0, 0, 0);
            c1[t][o0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, in[t][1][s], c1[t][o0 + 1], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            c1[t][o0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(l0, in[t][0][s], c1[t][o0], 0, 0, 0);
            c1[t][o0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(l1, in[t][0][s], c1[t][o0 + 1], 0, 0, 0);
        }
    }
};

// dense16_pipelined (n16.hpp) for two column tiles: out[t][O] += W[16 O .. 16 O + 15, :] . in[t]
template <int KS, int OT, int NB = 3>
__device__ __forceinline__ void dense16_pipelined2(f32x4 (&out)[TILES][OT], const bf16x8 (&in)[TILES][KS], const LdsW& wp, int lane) {
    constexpr int M = OT * KS, GS = 4, NG = M / GS;
    static_assert(OT % 4 == 0 && NB >= 2 && (NB - 1) * GS <= 15, "blocks of four output tiles; lgkmcnt is 4 bits");
#define CGNN_D16_O(t) (((t) / (KS * 4)) * 4 + (t) % 4)
#define CGNN_D16_S(t) (((t) / 4) % KS)
    const unsigned addr = (unsigned)(uintptr_t)wp.p + (unsigned)lane * 16u;
    u32x4 buf[NB][GS];
    static_for_each([&](auto pc) {
        constexpr int p = decltype(pc)::value;
        if constexpr (p < NG) {
            static_for_each([&](auto jc) {
                constexpr int t = p * GS + decltype(jc)::value;
                buf[p][decltype(jc)::value] = lds_read_b128<(CGNN_D16_O(t) * KS + CGNN_D16_S(t)) * 1024>(addr);
            }, std::make_integer_sequence<int, GS>{});
        }
    }, std::make_integer_sequence<int, NB - 1>{});
    static_for_each([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        if constexpr (g + NB - 1 < NG) {
            static_for_each([&](auto jc) {
                constexpr int t = (g + NB - 1) * GS + decltype(jc)::value;
                buf[(g + NB - 1) % NB][decltype(jc)::value] =
                    lds_read_b128<(CGNN_D16_O(t) * KS + CGNN_D16_S(t)) * 1024>(addr);
            }, std::make_integer_sequence<int, GS>{});
        }
        constexpr int newer = ((g + NB - 1 < NG ? g + NB - 1 : NG - 1) - g) * GS;     // reads issued after group g's
        lds_wait4<newer>(buf[g % NB][0], buf[g % NB][1], buf[g % NB][2], buf[g % NB][3]);
        static_for_each([&](auto jc) {
            constexpr int j = decltype(jc)::value, t = g * GS + j;
#pragma unroll
            for (int tl = 0; tl < TILES; ++tl)
                out[tl][CGNN_D16_O(t)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, buf[g % NB][j]), in[tl][CGNN_D16_S(t)], out[tl][CGNN_D16_O(t)], 0, 0, 0);
        }, std::make_integer_sequence<int, GS>{});
    }, std::make_integer_sequence<int, NG>{});
#undef CGNN_D16_O
#undef CGNN_D16_S
}
}  // namespace f2w

// Chunk Q of the step, as CGNN_F2R_CHUNK (f2_ring.hpp) with both tiles behind every fragment group.
#define CGNN_F2W_CHUNK(Q, C0, C1, OP)                                                                               \
    {                                                                                                               \
        if ((Q) + 1 >= PD) F2R_CHUNK_WAIT((PD - 2) * PC);                                                            \
        F2R_BARRIER();                                                                                               \
        issue(((Q) + PD) % NC, slot == 0 ? NS - 1 : slot - 1);                                                       \
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
#define CGNN_F2W_UNIT(U, C0, C1, OP)                                                                   \
    CGNN_F2W_CHUNK((U) * UNIT_CHUNKS + 0, C0, C1, OP) CGNN_F2W_CHUNK((U) * UNIT_CHUNKS + 1, C0, C1, OP) \
    CGNN_F2W_CHUNK((U) * UNIT_CHUNKS + 2, C0, C1, OP) CGNN_F2W_CHUNK((U) * UNIT_CHUNKS + 3, C0, C1, OP)

template <int NH, int PFMT>
__global__ __launch_bounds__(f2w::BLOCK, 1) void node_block_f2w4_kernel(F2RingArgs a) {
    using namespace f2r;
    using f2w::NS;
    using f2w::PC;
    using f2w::PD;
    using f2w::RING_OFF;
    using f2w::TILES;
    using f2w::WAVES;
    constexpr int NU = NH + 2, NC = NU * UNIT_CHUNKS;
    static_assert(NU <= CGNN_F2R_MAX_UNITS, "too many layers");
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool proj = a.ps_next != nullptr;    // block-uniform

    // ---- resident part: bias / LayerNorm vectors, projection weights ----
    {
        float* vec = reinterpret_cast<float*>(cgnn_smem);
        for (int i = threadIdx.x; i < D; i += blockDim.x) {
#pragma unroll
            for (int l = 0; l <= NH; ++l) vec[l * D + i] = a.bias[l][i];
            vec[(NH + 1) * D + i] = a.gamma[i];
            vec[(NH + 2) * D + i] = a.beta[i];
            vec[(NH + 3) * D + i] = a.bd_next ? a.bd_next[i] : 0.f;
        }
        if (proj) {
            const u32x4* s0 = reinterpret_cast<const u32x4*>(a.ws_w);
            const u32x4* s1 = reinterpret_cast<const u32x4*>(a.wd_w);
            u32x4* d0 = reinterpret_cast<u32x4*>(cgnn_smem + VEC_BYTES);
            for (int i = threadIdx.x; i < OT * KS * 64; i += blockDim.x) {
                d0[i] = s0[i];
                d0[OT * KS * 64 + i] = s1[i];
            }
        }
    }
    __syncthreads();
    const LdsVecPtr vec = (LdsVecPtr)cgnn_smem;
    const LdsWeightPtr proj_w = (LdsWeightPtr)(cgnn_smem + VEC_BYTES);
    const unsigned ring_lds = (unsigned)(uintptr_t)(cgnn_smem + RING_OFF);

    // ---- the ring ----
    const unsigned voff = (unsigned)wave * 1024u + (unsigned)lane * 16u;
    int slot = 0;                              // slot of the chunk about to be read
    auto issue = [&](int chunk /* 0 .. NC-1 */, int into_slot) {
        const char* src = a.unit[chunk / UNIT_CHUNKS] + (chunk % UNIT_CHUNKS) * CHUNK;
#pragma unroll
        for (int i = 0; i < PC; ++i)
            dma_piece(src + i * (WAVES * 1024), voff, ring_lds + into_slot * CHUNK + (wave + WAVES * i) * 1024);
    };
#pragma unroll
    for (int i = 0; i < PD; ++i) issue(i, i);

    // ---- first tiles' rows.  Tile t of wave w in step s: rows ((8 s + 2 w + t) * 16 ..+15) ----
    const int nb = gridDim.x;
    int64_t step = blockIdx.x;
    f32x4 xn[TILES][OT], an[TILES][OT];
    auto request_rows = [&](int64_t st) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            const int64_t row0 = (st * 8 + 2 * wave + t) * 16 + c;
            const int64_t row = row0 < a.n ? row0 : a.n - 1;      // rows past the end: the last row again
            const float* xp = a.x + row * D + 4 * q;
            const float* ap = a.agg + row * D + 4 * q;
            static_for_each([&](auto oc) { xn[t][decltype(oc)::value] = f2w::row_load_a<decltype(oc)::value * 64>(xp); },
                            std::make_integer_sequence<int, OT>{});
            static_for_each([&](auto oc) { an[t][decltype(oc)::value] = f2w::row_load_a<decltype(oc)::value * 64>(ap); },
                            std::make_integer_sequence<int, OT>{});
        }
    };
    request_rows(step);
    f2w::rows_ready_a<0>(xn[0], an[0], xn[1], an[1]);

    for (; step < a.steps; step += nb) {
        const int64_t next_step = step + nb < a.steps ? step + nb : step;     // last step: re-read its own rows
        // the last step may hold fewer than 128 rows: loads are clamped, stores predicated, and its closing wait
        // drains everything (a wave without live rows issues no stores for the counted wait to lean on)
        const bool partial = step == a.steps - 1 && (a.n & 127) != 0;

        f2w::FragPipe2 pipe;
        f16x8 op[TILES][2][KS];
        f32x4 c0[TILES][OT], c1[TILES][OT];
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            operand16f2<false, KS>(op[t], xn[t]);      // (xn stays where it is -- the accumulation half -- for the residual)
            fill16<OT>(c0[t], vec, q);
            fill16_global<OT>(c1[t], nullptr, q);
        }
        CGNN_F2W_UNIT(0, c0, c1, op)
#pragma unroll
        for (int t = 0; t < TILES; ++t) operand16f2<false, KS>(op[t], an[t]);
        CGNN_F2W_UNIT(1, c0, c1, op)
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            fold16f2<OT>(c0[t], c1[t]);
            operand16f2<true, KS>(op[t], c0[t]);
        }
        if constexpr (NH >= 2) {
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                fill16<OT>(c0[t], vec + 1 * D, q);
                fill16_global<OT>(c1[t], nullptr, q);
            }
            CGNN_F2W_UNIT(2, c0, c1, op)
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                fold16f2<OT>(c0[t], c1[t]);
                operand16f2<true, KS>(op[t], c0[t]);
            }
        }
        if constexpr (NH >= 3) {
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                fill16<OT>(c0[t], vec + 2 * D, q);
                fill16_global<OT>(c1[t], nullptr, q);
            }
            CGNN_F2W_UNIT(3, c0, c1, op)
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                fold16f2<OT>(c0[t], c1[t]);
                operand16f2<true, KS>(op[t], c0[t]);
            }
        }
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            fill16<OT>(c0[t], vec + NH * D, q);
            fill16_global<OT>(c1[t], nullptr, q);
        }
        CGNN_F2W_UNIT(NU - 1, c0, c1, op)

        // ---- tail: LayerNorm and the residual (the tile's own rows are still in xn), THEN the next tiles' rows are requested
        // into the same registers; stores and projections follow under their flight ----
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            fold16f2<OT>(c0[t], c1[t]);
            layer_norm16<OT>(c0[t], vec + (NH + 1) * D, vec + (NH + 2) * D, q);
            if (a.residual) {
#pragma unroll
                for (int o = 0; o < OT; ++o) c0[t][o] += xn[t][o];
            }
        }
        request_rows(next_step);
        // Every wave is done reading the step's last chunk: until the next step's first barrier its slot is the staging
        // area of the stores (4 KiB per wave, 2 KiB per tile): through LDS each store instruction writes 8 x 128
        // contiguous bytes of x_out, or 16 x 64 of a P table.
        F2R_BARRIER();
        char* const stage_w = cgnn_smem + RING_OFF + (slot == 0 ? NS - 1 : slot - 1) * CHUNK + wave * 4096;
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            char* const stage = stage_w + t * 2048;
            const int64_t tile_row = (step * 8 + 2 * wave + t) * 16;
            const bool ok0 = tile_row + (lane >> 3) < a.n, ok1 = tile_row + (lane >> 3) + 8 < a.n;    // rows of the staged stores
            float* const xo = a.x_out + (tile_row + (lane >> 3)) * D + (lane & 7) * 4;
#pragma unroll
            for (int p = 0; p < OT / 2; ++p) {       // features 32 p .. 32 p + 31 of the 16 rows: 16 x 128 B
                const LdsF4Ptr w = (LdsF4Ptr)(stage + c * 128 + q * 16);
                w[0] = c0[t][2 * p];
                w[4] = c0[t][2 * p + 1];
                const LdsF4Ptr r = (LdsF4Ptr)(stage + lane * 16);
                const f32x4 v0 = r[0], v1 = r[64];
                if (ok0) *reinterpret_cast<f32x4*>(xo + p * 32) = v0;
                if (ok1) *reinterpret_cast<f32x4*>(xo + p * 32 + 8 * D) = v1;
            }
        }
        // CGNN_P_BF16_S32 / CGNN_P_F16_S32 rows (feature 32t + 8g + 4h + i at h * 64 + (4t + g) * 4 + i): tile O of lane (c, q)
        // is 8 bytes at h = q & 1, 4t + g = 4 (O >> 1) + 2 (O & 1) + (q >> 1); four tiles fill 64 bytes of each half of the row
        auto store_p = [&](const f32x4 (&acc)[OT], __bf16* base, int t) __attribute__((always_inline)) {
            char* const stage = stage_w + t * 2048;
            const int64_t tile_row = (step * 8 + 2 * wave + t) * 16;
            const bool ok0 = tile_row + (lane >> 3) < a.n, ok1 = tile_row + (lane >> 3) + 8 < a.n;
            if constexpr (PFMT == CGNN_P_BF16_S32 || PFMT == CGNN_P_F16_S32) {
                char* const pt = reinterpret_cast<char*>(base + (tile_row + (lane >> 3)) * D) + ((lane & 7) >> 2) * 128 +
                                 (lane & 3) * 16;
#pragma unroll
                for (int pp = 0; pp < OT / 4; ++pp) {
#pragma unroll
                    for (int oo = 0; oo < 4; ++oo) {
                        char* const sp = stage + c * 128 + (q & 1) * 64 + (4 * (oo >> 1) + 2 * (oo & 1) + (q >> 1)) * 8;
                        if constexpr (PFMT == CGNN_P_F16_S32) {      // the same order, fp16 values
                            typedef _Float16 f16x4v __attribute__((ext_vector_type(4)));
                            f16x4v v;
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = (_Float16)acc[4 * pp + oo][i];
                            *(__attribute__((address_space(3))) f16x4v*)sp = v;
                        } else {
                            bf16x4 v;
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = (__bf16)acc[4 * pp + oo][i];
                            *(LdsB4Ptr)sp = v;
                        }
                    }
                    const LdsU4Ptr r = (LdsU4Ptr)(stage + lane * 16);
                    const u32x4 v0 = r[0], v1 = r[64];
                    if (ok0) *reinterpret_cast<u32x4*>(pt + pp * 64) = v0;
                    if (ok1) *reinterpret_cast<u32x4*>(pt + pp * 64 + 8 * D * 2) = v1;
                }
            } else {
                const int64_t row = tile_row + c;
                if (row < a.n) store_p16<PFMT, OT>(acc, base, row, q);
            }
        };
        if (proj) {   // block-uniform
            bf16x8 opb[TILES][KS];
#pragma unroll
            for (int t = 0; t < TILES; ++t) operand16<false, KS>(opb[t], c0[t]);
            {
                f32x4 acc[TILES][OT];
#pragma unroll
                for (int t = 0; t < TILES; ++t) fill16_global<OT>(acc[t], nullptr, q);
                f2w::dense16_pipelined2<KS, OT, 3>(acc, opb, LdsW(proj_w), lane);
#pragma unroll
                for (int t = 0; t < TILES; ++t) store_p(acc[t], a.ps_next, t);
            }
            {
                f32x4 acc[TILES][OT];
#pragma unroll
                for (int t = 0; t < TILES; ++t) fill16<OT>(acc[t], vec + (NH + 3) * D, q);
                f2w::dense16_pipelined2<KS, OT, 3>(acc, opb, LdsW(proj_w + OT * KS * 64), lane);
#pragma unroll
                for (int t = 0; t < TILES; ++t) store_p(acc[t], a.pd_next, t);
            }
        }
        // younger than the row loads: 16 x_out stores (and the P-row stores); the count names fewer than were issued, the
        // safe side
        if (partial)
            f2w::rows_ready_a<0>(xn[0], an[0], xn[1], an[1]);
        else
            f2w::rows_ready_a<16>(xn[0], an[0], xn[1], an[1]);
    }
    // the last steps' wrapped chunks are still on their way into this workgroup's LDS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

template <int NH, int PFMT>
int launch_f2w4(const F2RingArgs& a, hipStream_t st) {
    auto kern = node_block_f2w4_kernel<NH, PFMT>;
    int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (size_t)(f2w::LDS_BYTES), "hipFuncSetAttribute(node_block_f2w4)");
    if (rc != CGNN_OK) return rc;
    const int grid = (int)(a.steps < (int64_t)num_compute_units() ? a.steps : (int64_t)num_compute_units());
    kern<<<grid, f2w::BLOCK, f2w::LDS_BYTES, st>>>(a);
    return check_hip(hipGetLastError(), "cgnn_node_block(f16x2 ring, two tiles per wave) launch");
}

// one 16-row-tile-per-wave form or this one: node_block_f2.hip decides
int node_block_f2w4_launch(const F2RingArgs& a, int nh, int p_format, bool fuse, hipStream_t st) {
    const bool s16 = fuse && p_format == CGNN_P_BF16_S16, f16 = fuse && p_format == CGNN_P_F16_S32;
#define CGNN_GO(NHh)                                                                                     \
    if (nh == NHh)                                                                                        \
        return s16 ? launch_f2w4<NHh, CGNN_P_BF16_S16>(a, st)                                             \
                   : (f16 ? launch_f2w4<NHh, CGNN_P_F16_S32>(a, st) : launch_f2w4<NHh, CGNN_P_BF16_S32>(a, st));
    CGNN_GO(1) CGNN_GO(2) CGNN_GO(3)
#undef CGNN_GO
    return CGNN_ERR_UNSUPPORTED;
}

}  // namespace cgnn
