// cgnn_node_block: the fused node update of one message-passing round
// (reference graph_network.py:94-96 + the residual at :181).
//
//   u = LayerNorm(W3 relu(W2 relu(Wx x + Wa agg + b1) + b2) + b3) ;  x_out = x + u
//
// [Wx | Wa] is the column split of the first Linear following cat([x, agg]) at :94.
#include <string.h>

#include "mlp_device.hpp"

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
#pragma unroll
                for (int t = 0; t < DT; ++t) out[t] += xv[t];
            }
            store_rows_full<DT>(out, x_out + row * D, h);
        }
    }
}

}  // namespace cgnn

using namespace cgnn;

extern "C" int cgnn_node_block(const cgnn_mlp* mlp, const cgnn_linear* w_x, const cgnn_linear* w_agg, const float* x,
                               const float* agg, int64_t n, float* x_out, int32_t residual, int32_t latent,
                               void* stream) {
    if (!mlp || !w_x || !w_agg || !w_x->w || !w_agg->w) {
        set_error("cgnn_node_block: invalid argument");
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
    const int grid = grid_for_tiles((n + 31) / 32);
    const float* b1 = w_x->b ? w_x->b : w_agg->b;
#define CGNN_PAIR(Hh, Dd)                                                                                          \
    if (HT == Hh && DT == Dd) {                                                                                     \
        if (prec == CGNN_F32)                                                                                       \
            node_block_kernel<CGNN_F32, Hh, Dd><<<grid, CGNN_BLOCK, 0, st>>>(m, w_x->w, w_agg->w, b1, x, agg, n,     \
                                                                            x_out, residual);                      \
        else if (prec == CGNN_F32X3)                                                                                \
            node_block_kernel<CGNN_F32X3, Hh, Dd><<<grid, CGNN_BLOCK, 0, st>>>(m, w_x->w, w_agg->w, b1, x, agg, n,   \
                                                                              x_out, residual);                    \
        else                                                                                                        \
            node_block_kernel<CGNN_BF16, Hh, Dd><<<grid, CGNN_BLOCK, 0, st>>>(m, w_x->w, w_agg->w, b1, x, agg, n,    \
                                                                             x_out, residual);                     \
        return check_hip(hipGetLastError(), "cgnn_node_block launch");                                              \
    }
    CGNN_FOR_EACH_PAIR(CGNN_PAIR)
#undef CGNN_PAIR
    set_error("cgnn_node_block: no kernel for latent=%d hidden=%d", latent, hidden);
    return CGNN_ERR_UNSUPPORTED;
}
