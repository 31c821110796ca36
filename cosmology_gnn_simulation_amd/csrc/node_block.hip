// cgnn_node_block: the fused node update of one message-passing round
// (reference graph_network.py:94-96 + the residual at :181).
//
//   u = LayerNorm(W3 relu(W2 relu(Wx x + Wa agg + b1) + b2) + b3) ;  x_out = x + u
//
// [Wx | Wa] is the column split of the first Linear following cat([x, agg]) at :94.
#include <string.h>

#include "mlp_device.hpp"
#include "weight_ring.hpp"

namespace cgnn {

template <int PREC, int HT, int DT>
__global__ __launch_bounds__(CGNN_BLOCK) void node_block_kernel(MlpDev m, const void* wx, const void* wa,
                                                                const float* __restrict__ b1, const float* x,
                                                                const float* __restrict__ agg, int64_t n,
                                                                float* x_out, int residual) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int64_t tiles = (n + 31) / 32;
    constexpr int D = 32 * DT, H = 32 * HT;
    constexpr unsigned wbytes = DT * HT * 1024 * (PREC == CGNN_BF16 ? 2 : (PREC == CGNN_F32X3 ? 6 : 4));
    const BufW<PREC> wsrc_x(wx, wbytes), wsrc_a(wa, wbytes);
    const TileRange tr = tile_range(tiles);
    for (int64_t tile = tr.first; tile < tr.end; tile += tr.stride) {
        const int64_t row = tile * 32 + r;
        const int64_t rowc = row < n ? row : n - 1;
        // At latent 256 the f32 tile of x (128 registers) is NOT kept for the residual: it is read again at the end (from L2,
        // 32 KiB per tile) -- kept, the two-fp16-term kernel spilled 192 registers (772 bytes of scratch per lane and tile)
        constexpr bool RELOAD_X = DT >= 8;
        f32x16 xv[DT];
        load_rows_full<DT>(xv, x + rowc * D, h);
        Operand<PREC, HT> oph;
        {
            f32x16 acc[HT];
            acc_fill_bias<HT>(acc, b1, H, h);
            {
                Operand<PREC, DT> op;
                op.template from_acc<false>(xv);
                dense<DT, HT>(acc, op, wsrc_x, lane);
            }
            {
                f32x16 av[DT];
                load_rows_full<DT>(av, agg + rowc * D, h);
                Operand<PREC, DT> op;
                op.template from_acc<false>(av);
                dense<DT, HT>(acc, op, wsrc_a, lane);
            }
            oph.template from_acc<true>(acc);
        }
        f32x16 out[DT];
        mlp_tail<PREC, false, HT, DT>(m, oph, out, lane);
        layer_norm_rows<DT>(out, m.gamma, m.beta, h);
        if (row < n) {
            if (residual) {
                if constexpr (RELOAD_X) load_rows_full<DT>(xv, x + rowc * D, h);
#pragma unroll
                for (int t = 0; t < DT; ++t) out[t] += xv[t];
            }
            store_rows_full<DT>(out, x_out + row * D, h);
        }
    }
}

// --------------------------------------------------------------------------------------------------------
// CGNN_F32X3, weights cycled through LDS.
//
// With three bf16 terms per weight a 128x128 layer is 96 KB and each wave would stream 3 KB of fragments per
// 6 MFMAs from L2 -- exactly the L1 bandwidth of a CU, so the generic kernel ran the matrix cores at a third
// of their rate.  Here the four waves of a workgroup share every fragment: a layer is cut into chunks of 16
// fragments (48 KB), chunk i+1 is copied into one half of a two-slot LDS ring by LDS-DMA
// (global_load_lds_dwordx4, no registers, one barrier per chunk) while the waves run the MFMAs of chunk i from
// the other half.  Optionally the kernel also emits the next round's sender / receiver projections
// (cgnn_project_nodes fused in: saves one launch and one read of x per round).
template <int T, int PFMT>
__global__ __launch_bounds__(CGNN_BLOCK) void node_block_x3_kernel(MlpDev m, X3Chunks chunks,
                                                                   const float* __restrict__ b1, const float* x,
                                                                   const float* __restrict__ agg, int64_t n,
                                                                   float* x_out, int residual,
                                                                   const float* __restrict__ bd_next,
                                                                   typename PFmt<PFMT>::elem* __restrict__ ps_next,
                                                                   typename PFmt<PFMT>::elem* __restrict__ pd_next) {
    constexpr int D = 32 * T;
    constexpr int M = T * T * 2;                                   // fragments per layer
    constexpr int CH = (M < CGNN_X3_CHUNK_FRAGS) ? M : CGNN_X3_CHUNK_FRAGS;
    constexpr int NCH = M / CH;                                    // chunks per X3 layer
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t tiles = (n + 31) / 32;
    // block-uniform tile loop (every wave must reach every barrier): wave w of the block takes tile bt + w
    const int nb = gridDim.x, b = blockIdx.x;
    int64_t bt, bend, bstride;
    if ((nb & 7) == 0) {
        const int xcd = b & 7, slot = b >> 3, per = nb >> 3;
        bt = tiles * xcd / 8 + (int64_t)slot * CGNN_WAVES_PER_BLOCK;
        bend = tiles * (xcd + 1) / 8;
        bstride = (int64_t)per * CGNN_WAVES_PER_BLOCK;
    } else {
        bt = (int64_t)b * CGNN_WAVES_PER_BLOCK;
        bend = tiles;
        bstride = (int64_t)nb * CGNN_WAVES_PER_BLOCK;
    }
    WeightRing ring(chunks, wave, lane);
    if (bt < bend) ring.issue(0);
    for (; bt < bend; bt += bstride) {
        const bool more = bt + bstride < bend;
        const int64_t tile = bt + wave;
        const bool valid = tile < bend;
        const int64_t row = tile * 32 + r;
        const int64_t rowc = (valid && row < n) ? row : n - 1;
        f32x16 xv[T];
        load_rows_full<T>(xv, x + rowc * D, h);
        Operand<CGNN_F32X3, T> oph;
        {
            f32x16 acc[T];
            acc_fill_bias<T>(acc, b1, D, h);
            {
                Operand<CGNN_F32X3, T> op;
                op.template from_acc<false>(xv);
#define CGNN_X3_LAYER(ACC, OP)                                                              \
    {                                                                                        \
        const LdsWx3 w0(ring.acquire(more));                                                 \
        dense_part<T, T, 0, CH>(ACC, OP, w0, lane);                                          \
        if (NCH == 2) {                                                                      \
            const LdsWx3 w1(ring.acquire(more));                                             \
            dense_part<T, T, (NCH == 2 ? CH : 0), (NCH == 2 ? 2 * CH : CH)>(ACC, OP, w1, lane); \
        }                                                                                    \
    }
                CGNN_X3_LAYER(acc, op)
            }
            {
                f32x16 av[T];
                load_rows_full<T>(av, agg + rowc * D, h);
                Operand<CGNN_F32X3, T> op;
                op.template from_acc<false>(av);
                CGNN_X3_LAYER(acc, op)
            }
            oph.template from_acc<true>(acc);
        }
        for (int l = 1; l < m.nh; ++l) {
            f32x16 acc[T];
            acc_fill_bias<T>(acc, m.b[l], D, h);
            CGNN_X3_LAYER(acc, oph)
            oph.template from_acc<true>(acc);
        }
        f32x16 out[T];
        acc_fill_bias<T>(out, m.b[m.nh], D, h);
        CGNN_X3_LAYER(out, oph)
#undef CGNN_X3_LAYER
        layer_norm_rows<T>(out, m.gamma, m.beta, h);
        if (residual) {
#pragma unroll
            for (int t = 0; t < T; ++t) out[t] += xv[t];
        }
        if (valid && row < n) store_rows_full<T>(out, x_out + row * D, h);
        if (ps_next != nullptr) {   // block-uniform: the projection chunks are part of the ring sequence
            Operand<CGNN_BF16, T> opb;
            opb.template from_acc<false>(out);
            {
                f32x16 acc[T];
                acc_fill_bias<T>(acc, (const float*)nullptr, D, h);
                const LdsW ws(ring.acquire(more));
                dense<T, T>(acc, opb, ws, lane);
                if (valid && row < n) PFmt<PFMT>::template store<T>(acc, ps_next, row, h);
            }
            {
                f32x16 acc[T];
                acc_fill_bias<T>(acc, bd_next, D, h);
                const LdsW wd(ring.acquire(more));
                dense<T, T>(acc, opb, wd, lane);
                if (valid && row < n) PFmt<PFMT>::template store<T>(acc, pd_next, row, h);
            }
        }
    }
}

}  // namespace cgnn

using namespace cgnn;

namespace cgnn {
int node_block_x3n16(const MlpDev& m, int precision, const cgnn_linear* w_x, const cgnn_linear* w_agg, const float* x,
                     const float* agg, int64_t n, float* x_out, int residual, int T, bool fuse,
                     const cgnn_linear* ws_next, const cgnn_linear* wd_next, void* ps_next, void* pd_next, int p_format,
                     hipStream_t st);   // node_block_n16.hip
}

template <int T, int PFMT>
static int launch_node_x3(const MlpDev& m, const X3Chunks& ch, const float* b1, const float* x, const float* agg,
                          int64_t n, float* x_out, int residual, const float* bd, void* ps, void* pd, hipStream_t st) {
    auto kern = node_block_x3_kernel<T, PFMT>;
    const int lds = 2 * CGNN_X3_CHUNK_BYTES;
    int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (size_t)(lds), "hipFuncSetAttribute(node_block_x3)");
    if (rc != CGNN_OK) return rc;
    const int grid = grid_for_tiles((n + 31) / 32, 1);
    typedef typename PFmt<PFMT>::elem E;
    kern<<<grid, CGNN_BLOCK, lds, st>>>(m, ch, b1, x, agg, n, x_out, residual, bd, (E*)ps, (E*)pd);
    return check_hip(hipGetLastError(), "cgnn_node_block(x3) launch");
}

extern "C" int cgnn_node_block(const cgnn_mlp* mlp, const cgnn_linear* w_x, const cgnn_linear* w_agg, const float* x,
                               const float* agg, int64_t n, float* x_out, int32_t residual, int32_t latent,
                               const cgnn_linear* ws_next, const cgnn_linear* wd_next, int32_t proj_precision,
                               void* ps_next, void* pd_next, int32_t p_format, void* stream) {
    if (!mlp || !w_x || !w_agg || !w_x->w || !w_agg->w) {
        set_error("cgnn_node_block: invalid argument");
        return CGNN_ERR_INVALID_ARG;
    }
    const bool want_proj = ws_next != nullptr;
    if (want_proj && (!wd_next || !ws_next->w || !wd_next->w || !ps_next || !pd_next)) {
        set_error("cgnn_node_block: the projection epilogue needs ws_next, wd_next, ps_next and pd_next");
        return CGNN_ERR_INVALID_ARG;
    }
    // layer[0] of `mlp` is ignored: fill it with the x-half so that validation passes.
    cgnn_mlp tmp = *mlp;
    tmp.layer[0] = *w_x;
    MlpDev m;
    int rc = make_mlp_dev(&tmp, &m, nullptr, "cgnn_node_block");
    if (rc != CGNN_OK) return rc;
    if (!x || !agg || !x_out || n < 0 || latent <= 0 || !m.gamma) {
        set_error("cgnn_node_block: invalid argument");
        return CGNN_ERR_INVALID_ARG;
    }
    const int hidden = w_x->out_dim;
    if (w_x->in_dim != latent || w_agg->in_dim != latent || w_agg->out_dim != hidden || m.out_dim[m.nh] != latent ||
        m.in_dim[m.nh] != hidden) {
        set_error("cgnn_node_block: layer shapes do not match latent=%d hidden=%d", latent, hidden);
        return CGNN_ERR_INVALID_ARG;
    }
    for (int l = 1; l < m.nh; ++l)
        if (m.in_dim[l] != hidden || m.out_dim[l] != hidden) {
            set_error("cgnn_node_block: hidden layer %d has the wrong shape", l);
            return CGNN_ERR_INVALID_ARG;
        }
    if (latent % 32 || hidden % 32) {
        set_error("cgnn_node_block: latent %d / hidden %d must be multiples of 32", latent, hidden);
        return CGNN_ERR_UNSUPPORTED;
    }
    if (mlp->precision == CGNN_BF16_N16) {
        set_error("cgnn_node_block: CGNN_BF16_N16 weights are for cgnn_edge_block only");
        return CGNN_ERR_UNSUPPORTED;
    }
    if (n == 0) return CGNN_OK;
    hipStream_t st = (hipStream_t)stream;
    const int HT = hidden / 32, DT = latent / 32, prec = mlp->precision;
    const float* b1 = w_x->b ? w_x->b : w_agg->b;

    // ---- 16-row F32X3 kernel (weights packed CGNN_F32X3_N16): two waves per SIMD ----
    if (prec == CGNN_F32X3_N16 || prec == CGNN_F16X2_N16) {
        if (HT != DT || !(DT == 1 || DT == 2 || DT == 4)) {
            set_error("cgnn_node_block: CGNN_F32X3_N16 / CGNN_F16X2_N16 need hidden == latent in {32, 64, 128}");
            return CGNN_ERR_UNSUPPORTED;
        }
        bool fuse = false;
        if (want_proj) {
            const bool shapes = ws_next->in_dim == latent && ws_next->out_dim == latent && wd_next->in_dim == latent &&
                                wd_next->out_dim == latent;
            fuse = shapes && proj_precision == CGNN_BF16_N16 &&
                   (p_format == CGNN_P_BF16_S32 || p_format == CGNN_P_BF16_S16 || (p_format == CGNN_P_F16_S32 && DT == 4));
            if (!fuse && proj_precision == CGNN_BF16_N16) {
                set_error("cgnn_node_block: CGNN_BF16_N16 projection weights can only be used fused (square, bf16 table)");
                return CGNN_ERR_UNSUPPORTED;
            }
        }
        rc = node_block_x3n16(m, prec, w_x, w_agg, x, agg, n, x_out, residual, DT, fuse, ws_next, wd_next, ps_next, pd_next,
                              p_format, st);
        if (rc != CGNN_OK || fuse || !want_proj) return rc;
        return cgnn_project_nodes(ws_next, wd_next, proj_precision, x_out, n, ps_next, pd_next, p_format, stream);
    }

    // ---- LDS-cycled F32X3 kernel (square layers up to 128), with the projection fused when it is bf16 ----
    if (prec == CGNN_F32X3 && HT == DT && (DT == 1 || DT == 2 || DT == 4) && n >= 2048) {
        const bool fuse = want_proj && proj_precision == CGNN_BF16 && ws_next->in_dim == latent &&
                          ws_next->out_dim == latent && wd_next->in_dim == latent && wd_next->out_dim == latent &&
                          (p_format == CGNN_P_BF16_S32 || p_format == CGNN_P_BF16_S16);
        X3Chunks ch;
        memset(&ch, 0, sizeof(ch));
        const int M = DT * DT * 2;
        const int CH = M < CGNN_X3_CHUNK_FRAGS ? M : CGNN_X3_CHUNK_FRAGS;
        auto add_layer = [&](const void* w) {
            for (int c = 0; c < M / CH; ++c) {
                ch.src[ch.count] = reinterpret_cast<const char*>(w) + (size_t)c * CH * 3072;
                ch.bytes[ch.count++] = (uint32_t)CH * 3072;
            }
        };
        add_layer(w_x->w);
        add_layer(w_agg->w);
        for (int l = 1; l <= m.nh; ++l) add_layer(m.w[l]);
        if (fuse) {
            ch.src[ch.count] = reinterpret_cast<const char*>(ws_next->w);
            ch.bytes[ch.count++] = (uint32_t)M * 1024;
            ch.src[ch.count] = reinterpret_cast<const char*>(wd_next->w);
            ch.bytes[ch.count++] = (uint32_t)M * 1024;
        }
        void* ps = fuse ? ps_next : nullptr;
        void* pd = fuse ? pd_next : nullptr;
        const float* bd = fuse ? wd_next->b : nullptr;
        const int fmt = fuse ? p_format : CGNN_P_BF16_S32;
#define CGNN_X3(Tt)                                                                                                 \
    if (DT == Tt) {                                                                                                  \
        rc = fmt == CGNN_P_BF16_S16                                                                                  \
                 ? launch_node_x3<Tt, CGNN_P_BF16_S16>(m, ch, b1, x, agg, n, x_out, residual, bd, ps, pd, st)         \
                 : launch_node_x3<Tt, CGNN_P_BF16_S32>(m, ch, b1, x, agg, n, x_out, residual, bd, ps, pd, st);        \
    }
        CGNN_X3(1) CGNN_X3(2) CGNN_X3(4)
#undef CGNN_X3
        if (rc != CGNN_OK || fuse || !want_proj) return rc;
        return cgnn_project_nodes(ws_next, wd_next, proj_precision, x_out, n, ps_next, pd_next, p_format, stream);
    }

    const int grid = grid_for_tiles((n + 31) / 32);
    bool launched = false;
#define CGNN_PAIR(Hh, Dd)                                                                                          \
    if (!launched && HT == Hh && DT == Dd) {                                                                        \
        if (prec == CGNN_F32)                                                                                       \
            node_block_kernel<CGNN_F32, Hh, Dd><<<grid, CGNN_BLOCK, 0, st>>>(m, w_x->w, w_agg->w, b1, x, agg, n,     \
                                                                            x_out, residual);                      \
        else if (prec == CGNN_F32X3)                                                                                \
            node_block_kernel<CGNN_F32X3, Hh, Dd><<<grid, CGNN_BLOCK, 0, st>>>(m, w_x->w, w_agg->w, b1, x, agg, n,   \
                                                                              x_out, residual);                    \
        else if (prec == CGNN_F16X2)                                                                                \
            node_block_kernel<CGNN_F16X2, Hh, Dd><<<grid, CGNN_BLOCK, 0, st>>>(m, w_x->w, w_agg->w, b1, x, agg, n,   \
                                                                              x_out, residual);                    \
        else                                                                                                        \
            node_block_kernel<CGNN_BF16, Hh, Dd><<<grid, CGNN_BLOCK, 0, st>>>(m, w_x->w, w_agg->w, b1, x, agg, n,    \
                                                                             x_out, residual);                     \
        launched = true;                                                                                            \
    }
    CGNN_FOR_EACH_PAIR(CGNN_PAIR)
#undef CGNN_PAIR
    if (!launched) {
        set_error("cgnn_node_block: no kernel for latent=%d hidden=%d", latent, hidden);
        return CGNN_ERR_UNSUPPORTED;
    }
    rc = check_hip(hipGetLastError(), "cgnn_node_block launch");
    if (rc != CGNN_OK || !want_proj) return rc;
    return cgnn_project_nodes(ws_next, wd_next, proj_precision, x_out, n, ps_next, pd_next, p_format, stream);
}
