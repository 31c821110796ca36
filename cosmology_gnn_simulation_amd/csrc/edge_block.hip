// cgnn_edge_block: the fused edge update of one message-passing round
// (reference graph_network.py:89-90 + the residual at :182).
//
//   u = LayerNorm(W3 relu(W2 relu(Ps[src] + Pd[dst] + We e) + b2) + b3)
//
// Ps/Pd are the per-node halves of the first Linear (cgnn_project_nodes), so the
// E x 3D concatenation of the reference is never materialised.  One wave owns 32
// edges; the edge latent tile is loaded once, kept in registers for the residual,
// and written once.
#include <string.h>

#include "mlp_device.hpp"

namespace cgnn {

template <int PREC, bool WLDS, int HT, int DT>
__global__ __launch_bounds__(CGNN_BLOCK) void edge_block_kernel(MlpDev m, const float* __restrict__ ps,
                                                                const float* __restrict__ pd,
                                                                const int32_t* __restrict__ src,
                                                                const int32_t* __restrict__ dst, int64_t num_edges,
                                                                const float* e_in, float* e_out, float* e_upd,
                                                                int residual) {
    if (WLDS) stage_weights_to_lds(m, 0);
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int64_t tiles = (num_edges + 31) / 32;
    constexpr int D = 32 * DT, H = 32 * HT;
    const TileRange tr = tile_range(tiles);
    for (int64_t tile = tr.first; tile < tr.end; tile += tr.stride) {
        const int64_t e = tile * 32 + r;
        const int64_t ec = e < num_edges ? e : num_edges - 1;
        const int64_t s = src[ec], d = dst[ec];
        f32x16 ev[DT];
        load_rows_full<DT>(ev, e_in + ec * D, h);
        Operand<PREC, HT> oph;
        {
            f32x16 acc[HT];
            load_rows_full<HT>(acc, ps + s * H, h);
            add_rows_full<HT>(acc, pd + d * H, h);
            Operand<PREC, DT> op;
            op.template from_acc<false>(ev);
            dense<DT, HT>(acc, op, WSel<PREC, WLDS>::get(m, 0), lane);
            oph.template from_acc<true>(acc);
        }
        f32x16 out[DT];
        mlp_tail<PREC, WLDS, HT, DT>(m, oph, out, lane);
        layer_norm_rows<DT>(out, m.gamma, m.beta, h);
        if (e < num_edges) {
            if (e_upd != nullptr) store_rows_full<DT>(out, e_upd + e * D, h);
            if (residual) {
#pragma unroll
                for (int t = 0; t < DT; ++t) out[t] += ev[t];
            }
            store_rows_full<DT>(out, e_out + e * D, h);
        }
    }
}

template <int PREC, bool WLDS, int HT, int DT>
static int launch_edge(const MlpDev& m, size_t lds, const float* ps, const float* pd, const int32_t* src,
                       const int32_t* dst, int64_t num_edges, const float* e_in, float* e_out, float* e_upd,
                       int residual, hipStream_t st) {
    auto kern = edge_block_kernel<PREC, WLDS, HT, DT>;
    if (WLDS && lds > 48 * 1024) {
        int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                           "hipFuncSetAttribute(edge_block)");
        if (rc != CGNN_OK) return rc;
    }
    const int grid = grid_for_tiles((num_edges + 31) / 32, WLDS ? 1 : 2);
    kern<<<grid, CGNN_BLOCK, WLDS ? lds : 0, st>>>(m, ps, pd, src, dst, num_edges, e_in, e_out, e_upd, residual);
    return check_hip(hipGetLastError(), "cgnn_edge_block launch");
}

}  // namespace cgnn

using namespace cgnn;

extern "C" int cgnn_edge_block(const cgnn_mlp* mlp, const float* ps, const float* pd, const int32_t* src,
                               const int32_t* dst, int64_t num_edges, const float* e_in, float* e_out, float* e_upd,
                               int32_t residual, int32_t latent, void* stream) {
    MlpDev m;
    size_t lds = 0;
    int rc = make_mlp_dev(mlp, &m, &lds, "cgnn_edge_block");
    if (rc != CGNN_OK) return rc;
    if (!ps || !pd || !src || !dst || !e_in || !e_out || num_edges < 0 || latent <= 0) {
        set_error("cgnn_edge_block: invalid argument");
        return CGNN_ERR_INVALID_ARG;
    }
    if (!m.gamma) {
        set_error("cgnn_edge_block: the edge model needs its LayerNorm parameters");
        return CGNN_ERR_INVALID_ARG;
    }
    const int hidden = m.out_dim[0];
    if (m.in_dim[0] != latent || m.out_dim[m.nh] != latent || m.in_dim[m.nh] != hidden) {
        set_error("cgnn_edge_block: layer shapes do not match latent=%d hidden=%d", latent, hidden);
        return CGNN_ERR_INVALID_ARG;
    }
    for (int l = 1; l < m.nh; ++l)
        if (m.in_dim[l] != hidden || m.out_dim[l] != hidden) {
            set_error("cgnn_edge_block: hidden layer %d has the wrong shape", l);
            return CGNN_ERR_INVALID_ARG;
        }
    if (latent % 32 || hidden % 32) {
        set_error("cgnn_edge_block: latent %d / hidden %d must be multiples of 32", latent, hidden);
        return CGNN_ERR_UNSUPPORTED;
    }
    if (num_edges == 0) return CGNN_OK;
    hipStream_t st = (hipStream_t)stream;
    const int HT = hidden / 32, DT = latent / 32, prec = mlp->precision;
    const bool want_lds = prec == CGNN_BF16 && lds <= CGNN_LDS_WEIGHT_BUDGET && num_edges >= 4096;
#define CGNN_PAIR(Hh, Dd)                                                                                        \
    if (HT == Hh && DT == Dd) {                                                                                   \
        if (prec == CGNN_F32)                                                                                     \
            return launch_edge<CGNN_F32, false, Hh, Dd>(m, lds, ps, pd, src, dst, num_edges, e_in, e_out, e_upd,   \
                                                        residual, st);                                            \
        if ((Hh <= 4 && Dd <= 4) && want_lds)                                                                     \
            return launch_edge<CGNN_BF16, (Hh <= 4 && Dd <= 4), Hh, Dd>(m, lds, ps, pd, src, dst, num_edges, e_in, \
                                                                        e_out, e_upd, residual, st);              \
        return launch_edge<CGNN_BF16, false, Hh, Dd>(m, lds, ps, pd, src, dst, num_edges, e_in, e_out, e_upd,      \
                                                     residual, st);                                               \
    }
    CGNN_FOR_EACH_PAIR(CGNN_PAIR)
#undef CGNN_PAIR
    set_error("cgnn_edge_block: no kernel for latent=%d hidden=%d", latent, hidden);
    return CGNN_ERR_UNSUPPORTED;
}
