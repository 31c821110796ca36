// cgnn_node_block, CGNN_F16X2_N16 weights, latent = hidden = 128: the node update on two-fp16-term arithmetic (n16.hpp)
// with the weights streamed through a FIVE-slot LDS ring that is never drained inside a step.
//
// Shape: 512-thread workgroups, one per CU, two waves per SIMD; a wave owns 16 nodes per step (v_mfma_f32_16x16x32_f16),
// a workgroup step is 128 nodes.  Per step every wave walks all weights of the block once -- (hidden layers + 2) units
// of 64 KiB (x and agg halves of the first Linear, the hidden Linears, the output Linear), cut into 16-KiB chunks (two
// 16-feature output tiles over the whole K = 128) -- while the two bf16 projection matrices of the next round's Ps / Pd
// (64 KiB) and the bias / LayerNorm vectors stay resident.
//
// The two-slot ring of node_block_n16.hip waits `vmcnt(0)` + barrier at every chunk: the LDS-DMA of chunk i+1 is issued
// when chunk i starts and must have landed one chunk (0.3 us of MFMA work) later, less than an L2 round trip, so every
// chunk stalled (measured 79 k cycles per step against 14 k of MFMA issue).  Here chunk q+4 is issued when chunk q
// starts, the wave waits with a COUNTED `s_waitcnt vmcnt(4)` (two younger chunks, two 1-KiB pieces per wave each, stay
// in flight; chunks q and q+1 have landed, so the LDS fragment reads run two MFMA groups ahead across chunk boundaries)
// and a raw `s_barrier`; the next tile's x / agg rows are requested at the start of the tail (LayerNorm,
// stores, projections) of the current one and are the only other loads, waited for once per step with a count that
// leaves the step's stores in flight.  All vector-memory loads are inline asm so that the compiler inserts no waits of
// its own; vector-memory operations retire in order, so each counted wait names exactly the operations issued after
// the one it needs (a smaller count is always safe, a larger one never is: in a full step every wave issues every store;
// the one step that may be partial -- the last -- drains instead).
#include <string.h>

#include "f2_ring.hpp"

namespace cgnn {

int num_compute_units();   // runtime.hip

#define CGNN_F2R_MAX_UNITS 5      // hidden layers <= 3

struct F2RingArgs {
    const char* unit[CGNN_F2R_MAX_UNITS];   // packed CGNN_F16X2_N16 units in consumption order: Wx, Wa, hidden..., output
    const float* bias[CGNN_F2R_MAX_UNITS];  // bias of Linear 0 .. nh
    const float* gamma;
    const float* beta;
    const float* bd_next;                    // bias of the next round's Pd (the round's first edge Linear), or null
    const void* ws_w;                        // CGNN_BF16_N16 projection weights, or null
    const void* wd_w;
    const float* x;
    const float* agg;
    float* x_out;
    __bf16* ps_next;
    __bf16* pd_next;
    int64_t n;                               // rows
    int64_t steps;                           // 128-row steps, the last one may be partial
    int32_t residual;
};

namespace f2r_node {
constexpr int NS = 5, PD = NS - 1;            // ring slots, chunks in flight ahead of the one being read
constexpr int PROJ_BYTES = 2 * f2r::OT * f2r::KS * 1024;
constexpr int RING_OFF = f2r::VEC_BYTES + PROJ_BYTES;
constexpr int LDS_BYTES = RING_OFF + NS * f2r::CHUNK;
}  // namespace f2r_node

template <int NH, int PFMT>
__global__ __launch_bounds__(CGNN_F2R_BLOCK) void node_block_f2ring_kernel(F2RingArgs a) {
    using namespace f2r;
    using namespace f2r_node;
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

    // ---- first tile's rows ----
    const int nb = gridDim.x;
    int64_t step = blockIdx.x;
    f32x4 xn[OT], an[OT];
    {
        const int64_t row0 = (step * WAVES + wave) * 16 + c;
        const int64_t row = row0 < a.n ? row0 : a.n - 1;
        const float* xp = a.x + row * D + 4 * q;
        const float* ap = a.agg + row * D + 4 * q;
        static_for_each([&](auto oc) { xn[decltype(oc)::value] = row_load<decltype(oc)::value * 64>(xp); },
                        std::make_integer_sequence<int, OT>{});
        static_for_each([&](auto oc) { an[decltype(oc)::value] = row_load<decltype(oc)::value * 64>(ap); },
                        std::make_integer_sequence<int, OT>{});
        rows_ready<0>(xn, an);
    }

    for (; step < a.steps; step += nb) {
        const int64_t row = (step * WAVES + wave) * 16 + c;
        const int64_t next_step = step + nb < a.steps ? step + nb : step;     // last step: re-read its own rows
        const int64_t next_row0 = (next_step * WAVES + wave) * 16 + c;
        const int64_t next_row = next_row0 < a.n ? next_row0 : a.n - 1;      // rows past the end: the last row again
        // the last step may hold fewer than 128 rows: loads are clamped, stores predicated, and its closing wait
        // drains everything (a wave without live rows issues no stores for the counted wait to lean on)
        const bool partial = step == a.steps - 1 && (a.n & 127) != 0;

#ifdef CGNN_F2R_STAMPS
        const bool stamp_on = blockIdx.x == 9 && step == blockIdx.x + 3 * (int64_t)nb;
#endif
        F2R_STAMP(0);
        FragPipe16f2 pipe;
        f32x4 xv[OT];
#pragma unroll
        for (int o = 0; o < OT; ++o) xv[o] = xn[o];
        f16x8 op[2][KS];
        f32x4 c0[OT], c1[OT];
        operand16f2<false, KS>(op, xv);
        fill16<OT>(c0, vec, q);
        fill16_global<OT>(c1, nullptr, q);
        F2R_STAMP(1);
        CGNN_F2R_UNIT(0, c0, c1, op)
        F2R_STAMP(2);
        F2R_SPLIT(false, op, an);
        F2R_STAMP(3);
        CGNN_F2R_UNIT(1, c0, c1, op)
        F2R_STAMP(4);
        fold16f2<OT>(c0, c1);
        F2R_SPLIT(true, op, c0);
        F2R_STAMP(5);
        if constexpr (NH >= 2) {
            fill16<OT>(c0, vec + 1 * D, q);
            fill16_global<OT>(c1, nullptr, q);
            CGNN_F2R_UNIT(2, c0, c1, op)
            F2R_STAMP(6);
            fold16f2<OT>(c0, c1);
            F2R_SPLIT(true, op, c0);
            F2R_STAMP(7);
        }
        if constexpr (NH >= 3) {
            fill16<OT>(c0, vec + 2 * D, q);
            fill16_global<OT>(c1, nullptr, q);
            CGNN_F2R_UNIT(3, c0, c1, op)
            fold16f2<OT>(c0, c1);
            F2R_SPLIT(true, op, c0);
        }
        fill16<OT>(c0, vec + NH * D, q);
        fill16_global<OT>(c1, nullptr, q);
        CGNN_F2R_UNIT(NU - 1, c0, c1, op)
        F2R_STAMP(8);

        // ---- tail: the next tile's rows are requested first, then LayerNorm, residual, stores, projections ----
        {
            // (spreading these loads over the last chunks' MFMA groups moved their issue time, 350-450 cycles per
            // instruction with eight waves at it, into those chunks one for one: 0.85 ms either way)
            const float* xp = a.x + next_row * D + 4 * q;
            const float* ap = a.agg + next_row * D + 4 * q;
            static_for_each([&](auto oc) { xn[decltype(oc)::value] = row_load<decltype(oc)::value * 64>(xp); },
                            std::make_integer_sequence<int, OT>{});
            static_for_each([&](auto oc) { an[decltype(oc)::value] = row_load<decltype(oc)::value * 64>(ap); },
                            std::make_integer_sequence<int, OT>{});
        }
        // Every wave is done reading the step's last chunk: until the next step's first barrier its slot is the staging
        // area of the stores (2 KiB per wave).  The MFMA layout gives a lane 16 bytes of ITS row (64 lanes = 64 cache
        // lines per store instruction, and the stores were a fifth of the kernel); through LDS each instruction writes
        // 8 x 128 contiguous bytes of x_out, or 16 x 64 of a P table.
        F2R_BARRIER();
        F2R_STAMP(9);
        char* const stage = cgnn_smem + RING_OFF + (slot == 0 ? NS - 1 : slot - 1) * CHUNK + wave * 2048;
        // Staging layout: row R of the tile at R * 128, its 16-byte piece j at slot j ^ ((R >> 1) & 7).  Unswizzled, the
        // sixteen lanes of a write pass (one piece of sixteen rows: addresses 128 bytes apart) met in two bank groups of
        // four -- an 8-way conflict on every staging write: SQ_LDS_BANK_CONFLICT 8.8e7 cycles per launch, a quarter of
        // the kernel's time on the port that bounds it (round 4: 8.0e6 left, the 8-byte P writes at 2-way; -9.5 %).
        // Rotated by the row PAIR (even rows start at bank 0, odd rows at bank 32) they cover all 64 banks, and a read
        // pass (two whole rows) still does.
        const unsigned sw_w = (unsigned)((c >> 1) & 7);                       // writer: row c
        const unsigned sw_r = (unsigned)((lane >> 4) & 7);                    // reader: rows lane >> 3 and (lane >> 3) + 8
        char* const stage_rd0 = stage + (lane >> 3) * 128 + (((unsigned)(lane & 7) ^ sw_r) << 4);
        char* const stage_rd1 = stage + ((lane >> 3) + 8) * 128 + (((unsigned)(lane & 7) ^ sw_r ^ 4u) << 4);
        const int64_t tile_row = (step * WAVES + wave) * 16;
        const bool ok0 = tile_row + (lane >> 3) < a.n, ok1 = tile_row + (lane >> 3) + 8 < a.n;    // rows of the staged stores
        fold16f2<OT>(c0, c1);
        layer_norm16<OT>(c0, vec + (NH + 1) * D, vec + (NH + 2) * D, q);
        F2R_STAMP(10);
        {
            float* const xo = a.x_out + (tile_row + (lane >> 3)) * D + (lane & 7) * 4;
#pragma unroll
            for (int p = 0; p < OT / 2; ++p) {       // features 32 p .. 32 p + 31 of the 16 rows: 16 x 128 B
                if (a.residual) {
                    c0[2 * p] += xv[2 * p];
                    c0[2 * p + 1] += xv[2 * p + 1];
                }
                *(LdsF4Ptr)(stage + c * 128 + (((unsigned)q ^ sw_w) << 4)) = c0[2 * p];
                *(LdsF4Ptr)(stage + c * 128 + (((unsigned)q ^ sw_w ^ 4u) << 4)) = c0[2 * p + 1];
                const f32x4 v0 = *(LdsF4Ptr)stage_rd0, v1 = *(LdsF4Ptr)stage_rd1;
                if (ok0) *reinterpret_cast<f32x4*>(xo + p * 32) = v0;
                if (ok1) *reinterpret_cast<f32x4*>(xo + p * 32 + 8 * D) = v1;
            }
        }
        F2R_STAMP(11);
#define F2R_PMFMA(ACC, OPB, W) dense16_pipelined<KS, OT, 3>(ACC, OPB, W, lane)
#define F2R_PSTORE(ACC, BASE) store_p(ACC, BASE)
        // CGNN_P_BF16_S32 rows (feature 32t + 8g + 4h + i at h * 64 + (4t + g) * 4 + i): tile O of lane (c, q) is 8 bytes at
        // h = q & 1, 4t + g = 4 (O >> 1) + 2 (O & 1) + (q >> 1); four tiles fill 64 bytes of each half of the row
        auto store_p = [&](const f32x4 (&acc)[OT], __bf16* base) {
            if constexpr (PFMT == CGNN_P_BF16_S32 || PFMT == CGNN_P_F16_S32) {
                // bf16: 64 bytes into each half of the row; fp16 (CGNN_P_F16_S32): the staged 128 bytes of a row ARE one line of it
                char* const pt = reinterpret_cast<char*>(base + (tile_row + (lane >> 3)) * D) +
                                 (PFMT == CGNN_P_F16_S32 ? (lane & 7) * 16 : ((lane & 7) >> 2) * 128 + (lane & 3) * 16);
                constexpr int PP_STRIDE = PFMT == CGNN_P_F16_S32 ? 128 : 64;
#pragma unroll
                for (int pp = 0; pp < OT / 4; ++pp) {
#pragma unroll
                    for (int oo = 0; oo < 4; ++oo) {
                        // byte (q & 1) * 64 + (4 (oo >> 1) + 2 (oo & 1) + (q >> 1)) * 8 of the row: 8-byte half q >> 1 of piece
                        // 4 (q & 1) + 2 (oo >> 1) + (oo & 1)
                        char* const sp = stage + c * 128 + (((unsigned)(4 * (q & 1) + 2 * (oo >> 1) + (oo & 1)) ^ sw_w) << 4) + (q >> 1) * 8;
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
                    const u32x4 v0 = *(LdsU4Ptr)stage_rd0, v1 = *(LdsU4Ptr)stage_rd1;
                    if (ok0) *reinterpret_cast<u32x4*>(pt + pp * PP_STRIDE) = v0;
                    if (ok1) *reinterpret_cast<u32x4*>(pt + pp * PP_STRIDE + 8 * D * 2) = v1;
                }
            } else {
                if (row < a.n) store_p16<PFMT, OT>(acc, base, row, q);
            }
        };
        if (proj) {   // block-uniform
            bf16x8 opb[KS];
            operand16<false, KS>(opb, c0);
            {
                f32x4 acc[OT];
                fill16_global<OT>(acc, nullptr, q);
                F2R_PMFMA(acc, opb, LdsW(proj_w));
                F2R_PSTORE(acc, a.ps_next);
            }
            {
                f32x4 acc[OT];
                fill16<OT>(acc, vec + (NH + 3) * D, q);
                F2R_PMFMA(acc, opb, LdsW(proj_w + OT * KS * 64));
                F2R_PSTORE(acc, a.pd_next);
            }
        }
        F2R_STAMP(12);
        // younger than the row loads: 8 x_out stores and the P-row stores (4 per table); the count names fewer than were
        // issued, the safe side
        if (partial)
            rows_ready<0>(xn, an);
        else
            rows_ready<8>(xn, an);
        F2R_STAMP(13);
    }
    // the last steps' wrapped chunks are still on their way into this workgroup's LDS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

template <int NH, int PFMT>
static int launch_f2ring(const F2RingArgs& a, hipStream_t st) {
    auto kern = node_block_f2ring_kernel<NH, PFMT>;
    int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (size_t)(f2r_node::LDS_BYTES), "hipFuncSetAttribute(node_block_f2ring)");
    if (rc != CGNN_OK) return rc;
    const int grid = (int)(a.steps < (int64_t)num_compute_units() ? a.steps : (int64_t)num_compute_units());
    kern<<<grid, CGNN_F2R_BLOCK, f2r_node::LDS_BYTES, st>>>(a);
#ifdef CGNN_F2R_STAMPS
    {
        static int printed = 0;
        (void)hipStreamSynchronize(st);
        if (printed++ == 3) {
            static unsigned long long hs[8 * 64];
            (void)hipMemcpyFromSymbol(hs, HIP_SYMBOL(cgnn_f2r_stamps), sizeof(hs));
            const char* names[14] = {"start", "x split", "unit x", "agg split", "unit agg", "fold+split", "unit h", "fold+split",
                                     "unit out", "barrier", "fold+LN", "x_out stores", "proj", "rows ready"};
            for (int k = 0; k < 14; ++k) {
                printf("stamp %2d %-13s", k, names[k]);
                for (int w = 0; w < 8; w += 1)
                    printf(" %6lld(+%5lld)", (long long)(hs[w * 64 + k] - hs[0]),
                           k ? (long long)(hs[w * 64 + k] - hs[w * 64 + k - 1]) : 0LL);
                printf("\n");
            }
        }
    }
#endif
    return check_hip(hipGetLastError(), "cgnn_node_block(f16x2 ring) launch");
}

// A CGNN_F16X2_N16 node block with latent = hidden = 128 and nh <= 3 hidden layers.  Returns CGNN_OK and the number of
// rows done in *rows_done (n, or 0: shape not covered, nothing launched).
int node_block_f2ring(const MlpDev& m, const cgnn_linear* w_x, const cgnn_linear* w_agg, const float* x, const float* agg,
                      int64_t n, float* x_out, int residual, bool fuse, const cgnn_linear* ws_next,
                      const cgnn_linear* wd_next, void* ps_next, void* pd_next, int p_format, hipStream_t st,
                      int64_t* rows_done) {
    *rows_done = 0;
    const int64_t steps = (n + 127) / 128;
    if (m.nh < 1 || m.nh > 3 || steps == 0 || !m.gamma || !m.beta) return CGNN_OK;
    F2RingArgs a;
    memset(&a, 0, sizeof(a));
    a.unit[0] = reinterpret_cast<const char*>(w_x->w);
    a.unit[1] = reinterpret_cast<const char*>(w_agg->w);
    for (int l = 1; l <= m.nh; ++l) a.unit[1 + l] = reinterpret_cast<const char*>(m.w[l]);
    a.bias[0] = w_x->b ? w_x->b : w_agg->b;
    for (int l = 1; l <= m.nh; ++l) a.bias[l] = m.b[l];
    for (int l = 0; l <= m.nh; ++l)
        if (!a.bias[l]) {
            set_error("cgnn_node_block: CGNN_F16X2_N16 needs a bias on every Linear");
            return CGNN_ERR_UNSUPPORTED;
        }
    a.gamma = m.gamma;
    a.beta = m.beta;
    a.bd_next = fuse ? wd_next->b : nullptr;
    a.ws_w = fuse ? ws_next->w : nullptr;
    a.wd_w = fuse ? wd_next->w : nullptr;
    a.x = x;
    a.agg = agg;
    a.x_out = x_out;
    a.ps_next = fuse ? (__bf16*)ps_next : nullptr;
    a.pd_next = fuse ? (__bf16*)pd_next : nullptr;
    a.n = n;
    a.steps = steps;
    a.residual = residual;
    const bool s16 = fuse && p_format == CGNN_P_BF16_S16, f16 = fuse && p_format == CGNN_P_F16_S32;
    int rc = CGNN_ERR_UNSUPPORTED;
#define CGNN_GO(NHh)                                                                                       \
    if (m.nh == NHh)                                                                                        \
        rc = s16 ? launch_f2ring<NHh, CGNN_P_BF16_S16>(a, st)                                               \
                 : (f16 ? launch_f2ring<NHh, CGNN_P_F16_S32>(a, st) : launch_f2ring<NHh, CGNN_P_BF16_S32>(a, st));
    CGNN_GO(1) CGNN_GO(2) CGNN_GO(3)
#undef CGNN_GO
    if (rc == CGNN_OK) *rows_done = n;
    return rc;
}

}  // namespace cgnn
