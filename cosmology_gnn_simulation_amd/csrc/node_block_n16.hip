// cgnn_node_block, CGNN_F32X3_N16 variant: 16 nodes per wave (v_mfma_f32_16x16x32_bf16, see n16.hpp), 512-thread
// workgroups = two waves per SIMD, three-term weights streamed through the LDS ring (weight_ring.hpp).
//
// The 32-row F32X3 kernel needs ~380 registers per wave, i.e. one wave per SIMD, and its VALU work (splitting every
// activation into three bf16 terms: ~2900 instructions per tile against 832 MFMAs) cannot overlap its own MFMAs.
// Halving the rows per wave fits two waves per SIMD, so one wave's splitting and LayerNorm run under the other
// wave's MFMAs.
#include <string.h>

#include "n16.hpp"
#include "weight_ring.hpp"

namespace cgnn {

#define CGNN_NODE_N16_BLOCK 512          // measured alternatives: two 256-thread workgroups per CU with 24-KB chunks
                                         // (0.89-1.20 ms vs 0.84-0.88 ms), s_setprio around the MFMA groups (no change)
#define CGNN_NODE_N16_CHUNK_FRAGS 16
#define CGNN_NODE_N16_BPC 1
#define CGNN_NODE_N16_CHUNK_BYTES (CGNN_NODE_N16_CHUNK_FRAGS * 3 * 1024)

template <int T, int PFMT>
__global__ __launch_bounds__(CGNN_NODE_N16_BLOCK) void node_block_x3n16_kernel(
    MlpDev m, X3Chunks chunks, const float* __restrict__ b1, const float* x, const float* __restrict__ agg, int64_t n,
    float* x_out, int residual, const float* __restrict__ bd_next, __bf16* __restrict__ ps_next,
    __bf16* __restrict__ pd_next, const void* __restrict__ ws_w, const void* __restrict__ wd_w) {
    constexpr int D = 32 * T, OT = 2 * T, KS = T;
    constexpr int M = OT * KS;                                     // fragments per layer
    constexpr int CH = (M < CGNN_NODE_N16_CHUNK_FRAGS) ? M : CGNN_NODE_N16_CHUNK_FRAGS;
    constexpr int NCH = M / CH;                                    // 1, 1, 4 for T = 1, 2, 4
    constexpr int WAVES = CGNN_NODE_N16_BLOCK / 64;
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t tiles = (n + 15) / 16;
    const int nb = gridDim.x, b = blockIdx.x;
    int64_t bt, bend, bstride;                                     // block-uniform loop: every wave meets every barrier
    if ((nb & 7) == 0) {
        const int xcd = b & 7, slot = b >> 3, per = nb >> 3;
        bt = tiles * xcd / 8 + (int64_t)slot * WAVES;
        bend = tiles * (xcd + 1) / 8;
        bstride = (int64_t)per * WAVES;
    } else {
        bt = (int64_t)b * WAVES;
        bend = tiles;
        bstride = (int64_t)nb * WAVES;
    }
    // The two one-term projection layers (next round's Ws / Wd, M KiB each) stay resident behind the ring's two
    // slots for the whole launch: they were 64 KB of the 448 KB each workgroup step pulled through the ring, and four of
    // its twelve barriers.
    const LdsWeightPtr proj_w = (LdsWeightPtr)(cgnn_smem + 2 * CGNN_NODE_N16_CHUNK_BYTES);
    if (ps_next != nullptr) {
        const u32x4* s0 = reinterpret_cast<const u32x4*>(ws_w);
        const u32x4* s1 = reinterpret_cast<const u32x4*>(wd_w);
        u32x4* d0 = reinterpret_cast<u32x4*>(cgnn_smem + 2 * CGNN_NODE_N16_CHUNK_BYTES);
        for (int i = threadIdx.x; i < M * 64; i += blockDim.x) {
            d0[i] = s0[i];
            d0[M * 64 + i] = s1[i];
        }
        __syncthreads();
    }
    WeightRingT<CGNN_NODE_N16_CHUNK_BYTES, 2> ring(chunks, wave, lane);
    if (bt < bend) ring.issue(0);
    for (; bt < bend; bt += bstride) {
        const bool more = bt + bstride < bend;
        const int64_t tile = bt + wave;
        const bool valid = tile < bend;
        const int64_t row = tile * 16 + c;
        const bool live = valid && row < n;
        const int64_t rowc = live ? row : n - 1;
        const int foff = 4 * q;                                    // lane's feature offset inside a 16-feature tile

#define CGNN_N16_CHUNK(ACC, OP, I)                                                                    \
    if (NCH > I) {                                                                                    \
        const LdsWx3 w_(ring.acquire(more));                                                          \
        dense16x3_part<KS, OT, (NCH > I ? I * CH : 0), (NCH > I ? (I + 1) * CH : CH)>(ACC, OP, w_, lane); \
    }
#define CGNN_N16_LAYER(ACC, OP) \
    CGNN_N16_CHUNK(ACC, OP, 0) CGNN_N16_CHUNK(ACC, OP, 1) CGNN_N16_CHUNK(ACC, OP, 2) CGNN_N16_CHUNK(ACC, OP, 3)

        f32x4 xv[OT], av[OT];      // both row tiles are requested up front: the agg rows arrive under the Wx MFMAs
#pragma unroll
        for (int o = 0; o < OT; ++o) xv[o] = *reinterpret_cast<const f32x4*>(x + rowc * D + 16 * o + foff);
#pragma unroll
        for (int o = 0; o < OT; ++o) av[o] = *reinterpret_cast<const f32x4*>(agg + rowc * D + 16 * o + foff);
        bf16x8 oph[3][KS];
        {
            f32x4 acc[OT];
            fill16_global<OT>(acc, b1, q);
            {
                bf16x8 op[3][KS];
                operand16x3<false, KS>(op, xv);
                CGNN_N16_LAYER(acc, op)
            }
            {
                bf16x8 op[3][KS];
                operand16x3<false, KS>(op, av);
                CGNN_N16_LAYER(acc, op)
            }
            operand16x3<true, KS>(oph, acc);
        }
        for (int l = 1; l < m.nh; ++l) {
            f32x4 acc[OT];
            fill16_global<OT>(acc, m.b[l], q);
            CGNN_N16_LAYER(acc, oph)
            operand16x3<true, KS>(oph, acc);
        }
        f32x4 out[OT];
        fill16_global<OT>(out, m.b[m.nh], q);
        CGNN_N16_LAYER(out, oph)
#undef CGNN_N16_LAYER
#undef CGNN_N16_CHUNK
        layer_norm16_global<OT>(out, m.gamma, m.beta, q);
#pragma unroll
        for (int o = 0; o < OT; ++o) {
            if (residual) out[o] += xv[o];
            if (live) *reinterpret_cast<f32x4*>(x_out + row * D + 16 * o + foff) = out[o];
        }
        if (ps_next != nullptr) {   // block-uniform
            bf16x8 opb[KS];
            operand16<false, KS>(opb, out);
            {
                f32x4 acc[OT];
                fill16_global<OT>(acc, nullptr, q);
                dense16<KS, OT>(acc, opb, LdsW(proj_w), lane);
                if (live) store_p16<PFMT, OT>(acc, ps_next, row, q);
            }
            {
                f32x4 acc[OT];
                fill16_global<OT>(acc, bd_next, q);
                dense16<KS, OT>(acc, opb, LdsW(proj_w + M * 64), lane);
                if (live) store_p16<PFMT, OT>(acc, pd_next, row, q);
            }
        }
    }
}

// CGNN_F16X2_N16: the same kernel shape on two-fp16-term arithmetic (n16.hpp): three v_mfma_f32_16x16x32_f16 per
// fragment instead of six bf16 ones, 2 KiB of weights per fragment instead of 3, two conversions per activation
// instead of three.  Two accumulators per layer (c0: hi.hi, c1: the products carrying one 2^11-scaled term).
#define CGNN_NODE_F2_CHUNK_BYTES (CGNN_NODE_N16_CHUNK_FRAGS * 2 * 1024)

template <int T, int PFMT>
__global__ __launch_bounds__(CGNN_NODE_N16_BLOCK) void node_block_f2n16_kernel(
    MlpDev m, X3Chunks chunks, const float* __restrict__ b1, const float* x, const float* __restrict__ agg, int64_t n,
    float* x_out, int residual, const float* __restrict__ bd_next, __bf16* __restrict__ ps_next,
    __bf16* __restrict__ pd_next, const void* __restrict__ ws_w, const void* __restrict__ wd_w) {
    constexpr int D = 32 * T, OT = 2 * T, KS = T;
    constexpr int M = OT * KS;
    constexpr int CH = (M < CGNN_NODE_N16_CHUNK_FRAGS) ? M : CGNN_NODE_N16_CHUNK_FRAGS;
    constexpr int NCH = M / CH;
    constexpr int WAVES = CGNN_NODE_N16_BLOCK / 64;
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t tiles = (n + 15) / 16;
    const int nb = gridDim.x, b = blockIdx.x;
    int64_t bt, bend, bstride;
    if ((nb & 7) == 0) {
        const int xcd = b & 7, slot = b >> 3, per = nb >> 3;
        bt = tiles * xcd / 8 + (int64_t)slot * WAVES;
        bend = tiles * (xcd + 1) / 8;
        bstride = (int64_t)per * WAVES;
    } else {
        bt = (int64_t)b * WAVES;
        bend = tiles;
        bstride = (int64_t)nb * WAVES;
    }
    const LdsWeightPtr proj_w = (LdsWeightPtr)(cgnn_smem + 2 * CGNN_NODE_F2_CHUNK_BYTES);
    if (ps_next != nullptr) {
        const u32x4* s0 = reinterpret_cast<const u32x4*>(ws_w);
        const u32x4* s1 = reinterpret_cast<const u32x4*>(wd_w);
        u32x4* d0 = reinterpret_cast<u32x4*>(cgnn_smem + 2 * CGNN_NODE_F2_CHUNK_BYTES);
        for (int i = threadIdx.x; i < M * 64; i += blockDim.x) {
            d0[i] = s0[i];
            d0[M * 64 + i] = s1[i];
        }
        __syncthreads();
    }
    WeightRingT<CGNN_NODE_F2_CHUNK_BYTES, 2> ring(chunks, wave, lane);
    if (bt < bend) ring.issue(0);
    for (; bt < bend; bt += bstride) {
        const bool more = bt + bstride < bend;
        const int64_t tile = bt + wave;
        const bool valid = tile < bend;
        const int64_t row = tile * 16 + c;
        const bool live = valid && row < n;
        const int64_t rowc = live ? row : n - 1;
        const int foff = 4 * q;

#define CGNN_F2_CHUNK(C0, C1, OP, I)                                                                  \
    if (NCH > I) {                                                                                    \
        const LdsWf2 w_(ring.acquire(more));                                                          \
        dense16f2_part<KS, OT, (NCH > I ? I * CH : 0), (NCH > I ? (I + 1) * CH : CH)>(C0, C1, OP, w_, lane); \
    }
#define CGNN_F2_LAYER(C0, C1, OP) \
    CGNN_F2_CHUNK(C0, C1, OP, 0) CGNN_F2_CHUNK(C0, C1, OP, 1) CGNN_F2_CHUNK(C0, C1, OP, 2) CGNN_F2_CHUNK(C0, C1, OP, 3)

        f32x4 xv[OT], av[OT];
#pragma unroll
        for (int o = 0; o < OT; ++o) xv[o] = *reinterpret_cast<const f32x4*>(x + rowc * D + 16 * o + foff);
#pragma unroll
        for (int o = 0; o < OT; ++o) av[o] = *reinterpret_cast<const f32x4*>(agg + rowc * D + 16 * o + foff);
        f16x8 oph[2][KS];
        f32x4 c1[OT];
        {
            f32x4 c0[OT];
            fill16_global<OT>(c0, b1, q);
            fill16_global<OT>(c1, nullptr, q);
            {
                f16x8 op[2][KS];
                operand16f2<false, KS>(op, xv);
                CGNN_F2_LAYER(c0, c1, op)
            }
            {
                f16x8 op[2][KS];
                operand16f2<false, KS>(op, av);
                CGNN_F2_LAYER(c0, c1, op)
            }
            fold16f2<OT>(c0, c1);
            operand16f2<true, KS>(oph, c0);
        }
        for (int l = 1; l < m.nh; ++l) {
            f32x4 c0[OT];
            fill16_global<OT>(c0, m.b[l], q);
            fill16_global<OT>(c1, nullptr, q);
            CGNN_F2_LAYER(c0, c1, oph)
            fold16f2<OT>(c0, c1);
            operand16f2<true, KS>(oph, c0);
        }
        f32x4 out[OT];
        fill16_global<OT>(out, m.b[m.nh], q);
        fill16_global<OT>(c1, nullptr, q);
        CGNN_F2_LAYER(out, c1, oph)
        fold16f2<OT>(out, c1);
#undef CGNN_F2_LAYER
#undef CGNN_F2_CHUNK
        layer_norm16<OT, const float*>(out, m.gamma, m.beta, q);   // the ring kernel's arithmetic (node_block_f2.hip): same bits
#pragma unroll
        for (int o = 0; o < OT; ++o) {
            if (residual) out[o] += xv[o];
            if (live) *reinterpret_cast<f32x4*>(x_out + row * D + 16 * o + foff) = out[o];
        }
        if (ps_next != nullptr) {   // block-uniform
            bf16x8 opb[KS];
            operand16<false, KS>(opb, out);
            {
                f32x4 acc[OT];
                fill16_global<OT>(acc, nullptr, q);
                dense16<KS, OT>(acc, opb, LdsW(proj_w), lane);
                if (live) store_p16<PFMT, OT>(acc, ps_next, row, q);
            }
            {
                f32x4 acc[OT];
                fill16_global<OT>(acc, bd_next, q);
                dense16<KS, OT>(acc, opb, LdsW(proj_w + M * 64), lane);
                if (live) store_p16<PFMT, OT>(acc, pd_next, row, q);
            }
        }
    }
}

template <int T, int PFMT, int TERMS>
static int launch(const MlpDev& m, const X3Chunks& ch, const float* b1, const float* x, const float* agg, int64_t n,
                  float* x_out, int residual, const float* bd, void* ps, void* pd, const void* ws_w, const void* wd_w,
                  hipStream_t st) {
    auto kern = TERMS == 3 ? node_block_x3n16_kernel<T, PFMT> : node_block_f2n16_kernel<T, PFMT>;
    const int lds = 2 * (TERMS == 3 ? CGNN_NODE_N16_CHUNK_BYTES : CGNN_NODE_F2_CHUNK_BYTES) + (ps ? 2 * (2 * T * T) * 1024 : 0);
    int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (size_t)(lds), "hipFuncSetAttribute(node_block_x3n16)");
    if (rc != CGNN_OK) return rc;
    const int grid = grid_for_tiles((n + 15) / 16, CGNN_NODE_N16_BPC, CGNN_NODE_N16_BLOCK / 64);
    kern<<<grid, CGNN_NODE_N16_BLOCK, lds, st>>>(m, ch, b1, x, agg, n, x_out, residual, bd, (__bf16*)ps, (__bf16*)pd, ws_w,
                                                 wd_w);
    return check_hip(hipGetLastError(), "cgnn_node_block(x3 n16) launch");
}

int node_block_f2ring(const MlpDev& m, const cgnn_linear* w_x, const cgnn_linear* w_agg, const float* x, const float* agg,
                      int64_t n, float* x_out, int residual, bool fuse, const cgnn_linear* ws_next,
                      const cgnn_linear* wd_next, void* ps_next, void* pd_next, int p_format, hipStream_t st,
                      int64_t* rows_done);   // node_block_f2.hip

// Called by cgnn_node_block (node_block.hip) after argument validation; precision = CGNN_F32X3_N16 or CGNN_F16X2_N16.  `fuse`: emit the projections in-kernel.
int node_block_x3n16(const MlpDev& m, int precision, const cgnn_linear* w_x, const cgnn_linear* w_agg, const float* x,
                     const float* agg, int64_t n, float* x_out, int residual, int T, bool fuse,
                     const cgnn_linear* ws_next, const cgnn_linear* wd_next, void* ps_next, void* pd_next, int p_format,
                     hipStream_t st) {
    const bool f2 = precision == CGNN_F16X2_N16;
    if (f2 && T == 4) {
        // the five-slot-ring kernel (node_block_f2.hip) takes every row when it covers the shape (1..3 hidden layers,
        // LayerNorm, a bias on every Linear); otherwise (done == 0) the two-slot kernel below runs
        int64_t done = 0;
        int rc = node_block_f2ring(m, w_x, w_agg, x, agg, n, x_out, residual, fuse, ws_next, wd_next, ps_next, pd_next,
                                   p_format, st, &done);
        if (rc != CGNN_OK || done == n) return rc;
        x += done * 128;
        agg += done * 128;
        x_out += done * 128;
        if (fuse) {
            ps_next = reinterpret_cast<__bf16*>(ps_next) + done * 128;
            pd_next = reinterpret_cast<__bf16*>(pd_next) + done * 128;
        }
        n -= done;
    }
    X3Chunks ch;
    memset(&ch, 0, sizeof(ch));
    const int M = 2 * T * T;
    const int CH = M < CGNN_NODE_N16_CHUNK_FRAGS ? M : CGNN_NODE_N16_CHUNK_FRAGS;
    const size_t frag_bytes = f2 ? 2048 : 3072;
    auto add_layer = [&](const void* w) {
        for (int c = 0; c < M / CH; ++c) {
            ch.src[ch.count] = reinterpret_cast<const char*>(w) + (size_t)c * CH * frag_bytes;
            ch.bytes[ch.count++] = (uint32_t)(CH * frag_bytes);
        }
    };
    add_layer(w_x->w);
    add_layer(w_agg->w);
    for (int l = 1; l <= m.nh; ++l) add_layer(m.w[l]);
    const float* b1 = w_x->b ? w_x->b : w_agg->b;
    const float* bd = fuse ? wd_next->b : nullptr;
    void* ps = fuse ? ps_next : nullptr;
    void* pd = fuse ? pd_next : nullptr;
    const bool s16 = fuse && p_format == CGNN_P_BF16_S16;
    const void* wsw = fuse ? ws_next->w : nullptr;
    const void* wdw = fuse ? wd_next->w : nullptr;
#define CGNN_GO(Tt, TERMS)                                                                                           \
    if (T == Tt)                                                                                                      \
        return s16 ? launch<Tt, CGNN_P_BF16_S16, TERMS>(m, ch, b1, x, agg, n, x_out, residual, bd, ps, pd, wsw, wdw, st) \
                   : launch<Tt, CGNN_P_BF16_S32, TERMS>(m, ch, b1, x, agg, n, x_out, residual, bd, ps, pd, wsw, wdw, st);
    if (fuse && p_format == CGNN_P_F16_S32) {      // cgnn_edge_stream_run_w8's tables: latent 128 only
        if (T != 4) {
            set_error("cgnn_node_block: CGNN_P_F16_S32 tables are built for latent 128 (got %d)", 32 * T);
            return CGNN_ERR_UNSUPPORTED;
        }
        return f2 ? launch<4, CGNN_P_F16_S32, 2>(m, ch, b1, x, agg, n, x_out, residual, bd, ps, pd, wsw, wdw, st)
                  : launch<4, CGNN_P_F16_S32, 3>(m, ch, b1, x, agg, n, x_out, residual, bd, ps, pd, wsw, wdw, st);
    }
    if (f2) {
        CGNN_GO(1, 2) CGNN_GO(2, 2) CGNN_GO(4, 2)
    } else {
        CGNN_GO(1, 3) CGNN_GO(2, 3) CGNN_GO(4, 3)
    }
#undef CGNN_GO
    set_error("cgnn_node_block: no 16-row kernel for latent=%d", 32 * T);
    return CGNN_ERR_UNSUPPORTED;
}

}  // namespace cgnn
