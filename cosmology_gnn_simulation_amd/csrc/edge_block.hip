// cgnn_edge_block: the fused edge update of one message-passing round
// (reference graph_network.py:89-90 + the residual at :182).
//
//   u = LayerNorm(W3 relu(W2 relu(Ps[src] + Pd[dst] + We e) + b2) + b3)
//
// Ps/Pd are the per-node halves of the first Linear (cgnn_project_nodes), so the
// E x 3D concatenation of the reference is never materialised.  One wave owns 32
// edges; the edge latent tile is loaded once, kept in registers for the residual,
// and written once.  The edge-latent tensors are in the TILED32 layout (include/cgnn.h): every tile moves
// with lane-linear, fully coalesced 16-byte accesses.
#include <stdlib.h>
#include <string.h>

#include "mlp_device.hpp"

#include "n16.hpp"

namespace cgnn {

template <int PREC, bool WLDS, int HT, int DT>
__global__ __launch_bounds__(CGNN_BLOCK) void edge_block_kernel(MlpDev m,
                                                                const typename PRow<PREC>::elem* __restrict__ ps,
                                                                const typename PRow<PREC>::elem* __restrict__ pd,
                                                                const int32_t* __restrict__ src,
                                                                const int32_t* __restrict__ dst, int64_t num_edges,
                                                                const float* e_in, float* e_out, float* e_upd,
                                                                int residual) {
    if (WLDS) stage_weights_to_lds(m, 0);
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int64_t tiles = (num_edges + 31) / 32;
    constexpr int D = 32 * DT;
    const TileRange tr = tile_range(tiles);
    for (int64_t tile = tr.first; tile < tr.end; tile += tr.stride) {
        const int64_t e = tile * 32 + r;
        const int64_t ec = e < num_edges ? e : num_edges - 1;
        const int64_t s = src[ec], d = dst[ec];
        f32x16 ev[DT];
        load_tile<DT>(ev, e_in + tile * (32 * D), lane);
        Operand<PREC, HT> oph;
        {
            f32x16 acc[HT];
            PRow<PREC>::template load<HT>(acc, ps, s, h);
            PRow<PREC>::template add<HT>(acc, pd, d, h);
            Operand<PREC, DT> op;
            op.template from_acc<false>(ev);
            dense<DT, HT>(acc, op, WSel<PREC, WLDS>::get(m, 0), lane);
            oph.template from_acc<true>(acc);
        }
        f32x16 out[DT];
        mlp_tail<PREC, WLDS, HT, DT>(m, oph, out, lane);
        layer_norm_rows<DT>(out, VecSel<WLDS>::gamma(m), VecSel<WLDS>::beta(m), h);
        // whole tiles are stored: rows past num_edges are padding of the TILED32 buffer
        if (e_upd != nullptr) store_tile<DT>(out, e_upd + tile * (32 * D), lane);
        if (residual) {
            // latent 256: the tile is not held through the MLP (128 registers: the kernel spilled 143-206 of them) but read
            // again here, from L2 (this wave is the only writer of the tile, and it writes below)
            if constexpr (DT >= 8) load_tile<DT>(ev, e_in + tile * (32 * D), lane);
#pragma unroll
            for (int t = 0; t < DT; ++t) out[t] += ev[t];
        }
        store_tile<DT>(out, e_out + tile * (32 * D), lane);
    }
}

// bf16 fast path: 512-thread workgroups (two waves per SIMD) sharing one LDS-resident copy of the packed
// weights, so that one wave's VALU phases (bf16 conversion, LayerNorm, address math) and memory waits overlap
// the other wave's MFMA phases.  To fit two waves per SIMD (<= 256 registers each) the f32 edge tile is not
// kept across the MLP: it is converted to the bf16 operand on arrival and re-read (an L2/MALL hit: the tile
// was streamed in a few microseconds earlier) for the f32 residual.
#define CGNN_EDGE_LDS_BLOCK 512
template <int HT, int DT>
__global__ __launch_bounds__(CGNN_EDGE_LDS_BLOCK) void edge_block_lds_kernel(
    MlpDev m, const __bf16* __restrict__ ps, const __bf16* __restrict__ pd, const int32_t* __restrict__ src,
    const int32_t* __restrict__ dst, int64_t num_edges, const float* e_in, float* e_out, float* e_upd, int residual) {
    stage_weights_to_lds(m, 0);
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int64_t tiles = (num_edges + 31) / 32;
    constexpr int D = 32 * DT;
    const TileRange tr = tile_range(tiles);
    for (int64_t tile = tr.first; tile < tr.end; tile += tr.stride) {
        const int64_t e = tile * 32 + r;
        const int64_t ec = e < num_edges ? e : num_edges - 1;
        const int64_t s = src[ec], d = dst[ec];
        const float* etile = e_in + tile * (32 * D);
        Operand<CGNN_BF16, HT> oph;
        {
            Operand<CGNN_BF16, DT> op;
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                f32x16 a[1];
                load_tile<1>(a, etile + t * 1024, lane);
#pragma unroll
                for (int sidx = 0; sidx < 2; ++sidx)
#pragma unroll
                    for (int j = 0; j < 8; ++j) op.v[2 * t + sidx][j] = (__bf16)a[0][8 * sidx + j];
            }
            __builtin_amdgcn_sched_barrier(0);   // do not let the P gathers pile on top of the e tile in registers
            f32x16 acc[HT];
            PRow<CGNN_BF16>::load<HT>(acc, ps, s, h);
            PRow<CGNN_BF16>::add<HT>(acc, pd, d, h);
            dense<DT, HT>(acc, op, WSel<CGNN_BF16, true>::get(m, 0), lane);
            oph.template from_acc<true>(acc);
        }
        f32x16 out[DT];
        mlp_tail<CGNN_BF16, true, HT, DT>(m, oph, out, lane);
        layer_norm_rows<DT>(out, VecSel<true>::gamma(m), VecSel<true>::beta(m), h);
        asm volatile("" ::: "memory");   // keep the residual re-read a separate load (no CSE with the first)
        if (e_upd != nullptr) store_tile<DT>(out, e_upd + tile * (32 * D), lane);
        if (residual) add_tile<DT>(out, etile, lane);
        store_tile<DT>(out, e_out + tile * (32 * D), lane);
    }
}

template <int HT, int DT>
static int launch_edge_lds(const MlpDev& m, size_t lds, const __bf16* ps, const __bf16* pd, const int32_t* src,
                           const int32_t* dst, int64_t num_edges, const float* e_in, float* e_out, float* e_upd,
                           int residual, hipStream_t st) {
    auto kern = edge_block_lds_kernel<HT, DT>;
    if (lds > 48 * 1024) {
        int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (size_t)((int)lds), "hipFuncSetAttribute(edge_block_lds)");
        if (rc != CGNN_OK) return rc;
    }
    const int grid = grid_for_tiles((num_edges + 31) / 32, 1, CGNN_EDGE_LDS_BLOCK / 64);
    kern<<<grid, CGNN_EDGE_LDS_BLOCK, lds, st>>>(m, ps, pd, src, dst, num_edges, e_in, e_out, e_upd, residual);
    return check_hip(hipGetLastError(), "cgnn_edge_block(lds) launch");
}

#ifndef CGNN_EDGE_N16_BLOCK
#define CGNN_EDGE_N16_BLOCK 512
#define CGNN_EDGE_N16_GS 4
#endif
// N16 variant (weights packed CGNN_BF16_N16, P tables CGNN_P_BF16_S16): 16 edges per wave, see n16.hpp.  The
// f32 tile is read once, kept in registers for the residual, and written once.
template <int HT, int DT, int BLOCK, bool AGG>
__global__ __launch_bounds__(BLOCK) void edge_block_n16_kernel(
    MlpDev m, const __bf16* __restrict__ ps, const __bf16* __restrict__ pd, const int32_t* __restrict__ src,
    const int32_t* __restrict__ dst, int64_t num_edges, const float* e_in, float* e_out, float* e_upd, int residual,
    const float* __restrict__ x_gather, float* __restrict__ agg_out, int seg_k) {
    stage_weights_to_lds(m, 0);
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    constexpr int D = 32 * DT, DO = 2 * DT, HO = 2 * HT;     // DO / HO: 16-feature tiles
    const int64_t tiles = (num_edges + 15) / 16;
    const TileRange tr = tile_range(tiles);
    const bf16x8 sel0 = p16_selector(lane, 0), sel1 = p16_selector(lane, 1);
    for (int64_t tile = tr.first; tile < tr.end; tile += tr.stride) {
        const int64_t e = tile * 16 + c;
        const int64_t ec = e < num_edges ? e : num_edges - 1;
        const int64_t s = src[ec], d = dst[ec];
        const int64_t tbase = (tile >> 1) * (32 * D) + n16_lane_offset(c, q, (int)(tile & 1));
        f32x4 ev[DO];
#pragma unroll
        for (int o = 0; o < DO; ++o)   // streamed once: non-temporal, so that the P-table rows keep their L2 lines
            ev[o] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(e_in + tbase + n16_tile_offset(o)));
        bf16x8 pso[HT], pdo[HT];
        load_p16_operand<HT>(pso, ps, s, q);
        load_p16_operand<HT>(pdo, pd, d, q);
        f32x4 out[DO];
        bf16x8 oph[HT];
        {
            bf16x8 op[DT];
            operand16<false, DT>(op, ev);
            f32x4 acc[HO];
            p16_accumulate<HT>(acc, pso, pdo, sel0, sel1);
            dense16<DT, HO, CGNN_EDGE_N16_GS>(acc, op, WSel<CGNN_BF16, true>::get(m, 0), lane);
            operand16<true, HT>(oph, acc);
        }
        for (int l = 1; l < m.nh; ++l) {
            f32x4 acc[HO];
            fill16<HO>(acc, VecSel<true>::bias(m, l), q);
            dense16<HT, HO, CGNN_EDGE_N16_GS>(acc, oph, WSel<CGNN_BF16, true>::get(m, l), lane);
            operand16<true, HT>(oph, acc);
        }
        fill16<DO>(out, VecSel<true>::bias(m, m.nh), q);
        dense16<HT, DO, CGNN_EDGE_N16_GS>(out, oph, WSel<CGNN_BF16, true>::get(m, m.nh), lane);
        layer_norm16<DO>(out, VecSel<true>::gamma(m), VecSel<true>::beta(m), q);
        if (e_upd != nullptr) {
#pragma unroll
            for (int o = 0; o < DO; ++o) *reinterpret_cast<f32x4*>(e_upd + tbase + n16_tile_offset(o)) = out[o];
        }
        if (AGG && agg_out != nullptr && x_gather == nullptr) {   // message_source "edge": aggregate the update itself
            const bool writer = (c & (seg_k - 1)) == 0 && e < num_edges;
#pragma unroll
            for (int o = 0; o < DO; ++o) {
                f32x4 g;
#pragma unroll
                for (int i = 0; i < 4; ++i) g[i] = seg_k == 16 ? segment_sum<16>(out[o][i]) : segment_sum<8>(out[o][i]);
                if (writer) *reinterpret_cast<f32x4*>(agg_out + d * D + 16 * o + 4 * q) = g;
            }
        }
#pragma unroll
        for (int o = 0; o < DO; ++o) {
            if (residual) out[o] += ev[o];
            __builtin_nontemporal_store(out[o], reinterpret_cast<f32x4*>(e_out + tbase + n16_tile_offset(o)));
        }
        if (AGG && agg_out != nullptr && x_gather != nullptr) {   // PyG default message: aggregate the sender node rows
            const bool writer = (c & (seg_k - 1)) == 0 && e < num_edges;
            const float* xr = x_gather + s * D + 4 * q;
#pragma unroll
            for (int o = 0; o < DO; ++o) {
                f32x4 g = *reinterpret_cast<const f32x4*>(xr + 16 * o);
#pragma unroll
                for (int i = 0; i < 4; ++i) g[i] = seg_k == 16 ? segment_sum<16>(g[i]) : segment_sum<8>(g[i]);
                if (writer) *reinterpret_cast<f32x4*>(agg_out + d * D + 16 * o + 4 * q) = g;
            }
        }
    }
}

template <int HT, int DT, int BLOCK, bool AGG>
static int launch_edge_n16_as(const MlpDev& m, size_t lds, const __bf16* ps, const __bf16* pd, const int32_t* src,
                              const int32_t* dst, int64_t num_edges, const float* e_in, float* e_out, float* e_upd,
                              int residual, const float* x_gather, float* agg_out, int seg_k, hipStream_t st) {
    auto kern = edge_block_n16_kernel<HT, DT, BLOCK, AGG>;
    if (lds > 48 * 1024) {
        int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (size_t)((int)lds), "hipFuncSetAttribute(edge_block_n16)");
        if (rc != CGNN_OK) return rc;
    }
    const int grid = grid_for_tiles((num_edges + 15) / 16, 1, BLOCK / 64);
    kern<<<grid, BLOCK, lds, st>>>(m, ps, pd, src, dst, num_edges, e_in, e_out, e_upd, residual, x_gather, agg_out,
                                   seg_k);
    return check_hip(hipGetLastError(), "cgnn_edge_block(n16) launch");
}

// The weights occupy most of the LDS, so one workgroup runs per CU and its size sets the occupancy: 512 threads
// (two waves per SIMD, 256 registers each) for the variant that also reduces the aggregate, 1024 threads (four waves
// per SIMD, 128 registers) for the plain edge update (measured 3.10 against 3.32 ms with 512 threads at cfg3).
template <int HT, int DT>
static int launch_edge_n16(const MlpDev& m, size_t lds, const __bf16* ps, const __bf16* pd, const int32_t* src,
                           const int32_t* dst, int64_t num_edges, const float* e_in, float* e_out, float* e_upd,
                           int residual, const float* x_gather, float* agg_out, int seg_k, hipStream_t st) {
    if (agg_out != nullptr)
        return launch_edge_n16_as<HT, DT, 512, true>(m, lds, ps, pd, src, dst, num_edges, e_in, e_out, e_upd, residual,
                                                     x_gather, agg_out, seg_k, st);
    return launch_edge_n16_as<HT, DT, 1024, false>(m, lds, ps, pd, src, dst, num_edges, e_in, e_out, e_upd, residual,
                                                   x_gather, agg_out, seg_k, st);
}

// Edge encoder in the N16 layout (reference graph_network.py:57: MLP + LayerNorm on the 4 edge features): narrow
// input (<= 32 features), weights packed CGNN_BF16_N16 and resident in LDS, output written straight into the
// TILED32 edge-latent buffer.  Same structure as the edge block: 16 edges per wave, two waves per SIMD.
#define CGNN_EDGE_ENC_BLOCK 1024   // ~90 registers per wave: four waves per SIMD share the LDS-resident weights
template <int HT, int DT>
__global__ __launch_bounds__(CGNN_EDGE_ENC_BLOCK) void edge_encode_n16_kernel(MlpDev m, const float* __restrict__ x,
                                                                             int64_t n, int ld_x, float* __restrict__ y) {
    stage_weights_to_lds(m, 0);
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    constexpr int D = 32 * DT, DO = 2 * DT, HO = 2 * HT;
    const int in_dim = m.in_dim[0];
    const int64_t tiles = (n + 15) / 16;
    const TileRange tr = tile_range(tiles);
    for (int64_t tile = tr.first; tile < tr.end; tile += tr.stride) {
        const int64_t e = tile * 16 + c;
        const int64_t ec = e < n ? e : n - 1;
        bf16x8 op[1];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int f = 16 * (j >> 2) + 4 * q + (j & 3);        // phi(0, q, j), n16.hpp
            op[0][j] = (__bf16)(f < in_dim ? x[ec * ld_x + f] : 0.f);
        }
        bf16x8 oph[HT];
        {
            f32x4 acc[HO];
            fill16<HO>(acc, VecSel<true>::bias(m, 0), q);
            dense16<1, HO>(acc, op, WSel<CGNN_BF16, true>::get(m, 0), lane);
            operand16<true, HT>(oph, acc);
        }
        for (int l = 1; l < m.nh; ++l) {
            f32x4 acc[HO];
            fill16<HO>(acc, VecSel<true>::bias(m, l), q);
            dense16<HT, HO>(acc, oph, WSel<CGNN_BF16, true>::get(m, l), lane);
            operand16<true, HT>(oph, acc);
        }
        f32x4 out[DO];
        fill16<DO>(out, VecSel<true>::bias(m, m.nh), q);
        dense16<HT, DO>(out, oph, WSel<CGNN_BF16, true>::get(m, m.nh), lane);
        layer_norm16<DO>(out, VecSel<true>::gamma(m), VecSel<true>::beta(m), q);
        const int64_t tbase = (tile >> 1) * (32 * D) + n16_lane_offset(c, q, (int)(tile & 1));
#pragma unroll
        for (int o = 0; o < DO; ++o)     // written once, read once by the edge stream: keep it out of the caches
            __builtin_nontemporal_store(out[o], reinterpret_cast<f32x4*>(y + tbase + n16_tile_offset(o)));
    }
}

template <int HT, int DT>
static int launch_edge_encode_n16(const MlpDev& m, size_t lds, const float* x, int64_t n, int ld_x, float* y,
                                  hipStream_t st) {
    auto kern = edge_encode_n16_kernel<HT, DT>;
    if (lds > 48 * 1024) {
        int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (size_t)((int)lds), "hipFuncSetAttribute(edge_encode_n16)");
        if (rc != CGNN_OK) return rc;
    }
    const int grid = grid_for_tiles((n + 15) / 16, 1, CGNN_EDGE_ENC_BLOCK / 64);
    kern<<<grid, CGNN_EDGE_ENC_BLOCK, lds, st>>>(m, x, n, ld_x, y);
    return check_hip(hipGetLastError(), "cgnn_mlp_rows(n16 encoder) launch");
}

int edge_encode_ring256(const MlpDev& m, const float* x, int64_t n, int ld_x, float* y, hipStream_t st,
                        int* handled);   // edge_block_ring256.hip

// Entry used by cgnn_mlp_rows (mlp_rows.hip) for CGNN_BF16_N16 weights.
int mlp_rows_n16_encoder(const MlpDev& m, size_t lds, const float* x, int64_t n, int ld_x, float* y, hipStream_t st) {
    const int hidden = m.out_dim[0], latent = m.out_dim[m.nh];
    if (hidden == 256 && latent == 256) {      // weights streamed through the LDS ring (they do not fit)
        int handled = 0;
        const int rc = edge_encode_ring256(m, x, n, ld_x, y, st, &handled);
        if (rc != CGNN_OK || handled) return rc;
        set_error("cgnn_mlp_rows: the 256-wide CGNN_BF16_N16 encoder takes <= 4 input features in rows of a multiple of 4 "
                  "floats, 16-byte aligned, 1..3 hidden layers with biases and LayerNorm (got in=%d, ld=%d)", m.in_dim[0], ld_x);
        return CGNN_ERR_UNSUPPORTED;
    }
    if (m.in_dim[0] > 32 || hidden % 32 || latent % 32 || lds > CGNN_LDS_WEIGHT_BUDGET || m.gamma == nullptr) {
        set_error("cgnn_mlp_rows: CGNN_BF16_N16 is the edge-encoder path (input <= 32 features, LayerNorm, weights "
                  "resident in LDS)");
        return CGNN_ERR_UNSUPPORTED;
    }
    const int HT = hidden / 32, DT = latent / 32;
#define CGNN_ENC(Hh, Dd) \
    if (HT == Hh && DT == Dd) return launch_edge_encode_n16<Hh, Dd>(m, lds, x, n, ld_x, y, st);
    CGNN_ENC(1, 1) CGNN_ENC(2, 2) CGNN_ENC(4, 4) CGNN_ENC(4, 2)
#undef CGNN_ENC
    set_error("cgnn_mlp_rows: no CGNN_BF16_N16 encoder kernel for hidden=%d latent=%d", hidden, latent);
    return CGNN_ERR_UNSUPPORTED;
}

template <int PREC, bool WLDS, int HT, int DT>
static int launch_edge(const MlpDev& m, size_t lds, const typename PRow<PREC>::elem* ps,
                       const typename PRow<PREC>::elem* pd, const int32_t* src,
                       const int32_t* dst, int64_t num_edges, const float* e_in, float* e_out, float* e_upd,
                       int residual, hipStream_t st) {
    auto kern = edge_block_kernel<PREC, WLDS, HT, DT>;
    if (WLDS && lds > 48 * 1024) {
        int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (size_t)((int)lds), "hipFuncSetAttribute(edge_block)");
        if (rc != CGNN_OK) return rc;
    }
    const int grid = grid_for_tiles((num_edges + 31) / 32, WLDS ? 1 : 2);
    kern<<<grid, CGNN_BLOCK, WLDS ? lds : 0, st>>>(m, ps, pd, src, dst, num_edges, e_in, e_out, e_upd, residual);
    return check_hip(hipGetLastError(), "cgnn_edge_block launch");
}

int edge_block_ring256(const MlpDev& m, const __bf16* ps, const __bf16* pd, const int32_t* src, const int32_t* dst,
                       int64_t num_edges, const float* e_in, float* e_out, float* e_upd, int residual,
                       hipStream_t st);   // edge_block_ring256.hip
int edge_block_f2(const MlpDev& m, const float* ps, const float* pd, const int32_t* src, const int32_t* dst,
                  int64_t num_edges, const float* e_in, float* e_out, float* e_upd, int residual,
                  hipStream_t st);   // edge_block_f2.hip

}  // namespace cgnn

using namespace cgnn;

extern "C" int cgnn_edge_block(const cgnn_mlp* mlp, const void* ps, const void* pd, const int32_t* src,
                               const int32_t* dst, int64_t num_edges, const float* e_in, float* e_out, float* e_upd,
                               int32_t residual, int32_t latent, const float* x_gather, float* agg_out,
                               int32_t seg_k, void* stream) {
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
    if (agg_out != nullptr) {
        if (prec != CGNN_BF16_N16) {
            set_error("cgnn_edge_block: fused aggregation needs CGNN_BF16_N16 weights (use cgnn_aggregate otherwise)");
            return CGNN_ERR_UNSUPPORTED;
        }
        if ((seg_k != 8 && seg_k != 16) || num_edges % seg_k != 0) {
            set_error("cgnn_edge_block: fused aggregation needs a fixed in-degree of 8 or 16 (seg_k=%d, E=%lld)", seg_k,
                      (long long)num_edges);
            return CGNN_ERR_UNSUPPORTED;
        }
    }
    if (prec == CGNN_F16X2_N16) {     // f32 accuracy on the fp16 matrix cores; Ps / Pd are CGNN_P_F32 tables
        if (HT != 4 || DT != 4) {
            set_error("cgnn_edge_block: CGNN_F16X2_N16 needs latent == hidden == 128 (got %d / %d)", latent, hidden);
            return CGNN_ERR_UNSUPPORTED;
        }
        return edge_block_f2(m, (const float*)ps, (const float*)pd, src, dst, num_edges, e_in, e_out, e_upd, residual, st);
    }
    if (prec == CGNN_BF16_N16 && HT == 8 && DT == 8) {   // latent = hidden = 256: weights streamed through an LDS ring
        if (agg_out != nullptr) {
            set_error("cgnn_edge_block: the 256-wide CGNN_BF16_N16 kernel has no fused aggregation (use cgnn_aggregate)");
            return CGNN_ERR_UNSUPPORTED;
        }
        return edge_block_ring256(m, (const __bf16*)ps, (const __bf16*)pd, src, dst, num_edges, e_in, e_out, e_upd, residual,
                                  st);
    }
    if (prec == CGNN_BF16_N16) {
        if (lds > CGNN_LDS_WEIGHT_BUDGET) {
            set_error("cgnn_edge_block: CGNN_BF16_N16 needs the weights (%zu bytes) resident in LDS", lds);
            return CGNN_ERR_UNSUPPORTED;
        }
#define CGNN_N16(Hh, Dd)      \
    if (HT == Hh && DT == Dd) \
        return launch_edge_n16<Hh, Dd>(m, lds, (const __bf16*)ps, (const __bf16*)pd, src, dst, num_edges, e_in, e_out, e_upd, residual, x_gather, agg_out, seg_k, st);
        CGNN_N16(1, 1) CGNN_N16(2, 2) CGNN_N16(4, 4) CGNN_N16(4, 2)
#undef CGNN_N16
        set_error("cgnn_edge_block: no CGNN_BF16_N16 kernel for latent=%d hidden=%d", latent, hidden);
        return CGNN_ERR_UNSUPPORTED;
    }
    const bool want_lds = prec == CGNN_BF16 && lds <= CGNN_LDS_WEIGHT_BUDGET && num_edges >= 4096;
#define CGNN_PAIR(Hh, Dd)                                                                                        \
    if (HT == Hh && DT == Dd) {                                                                                   \
        if (prec == CGNN_F32)                                                                                     \
            return launch_edge<CGNN_F32, false, Hh, Dd>(m, lds, (const float*)ps, (const float*)pd, src, dst,      \
                                                        num_edges, e_in, e_out, e_upd, residual, st);             \
        if ((Hh <= 4 && Dd <= 4) && want_lds)                                                                     \
            return launch_edge_lds<(Hh <= 4 ? Hh : 1), (Dd <= 4 ? Dd : 1)>(                                        \
                m, lds, (const __bf16*)ps, (const __bf16*)pd, src, dst, num_edges, e_in, e_out, e_upd, residual,  \
                st);                                                                                              \
        return launch_edge<CGNN_BF16, false, Hh, Dd>(m, lds, (const __bf16*)ps, (const __bf16*)pd, src, dst,       \
                                                     num_edges, e_in, e_out, e_upd, residual, st);                \
    }
    CGNN_FOR_EACH_PAIR(CGNN_PAIR)
#undef CGNN_PAIR
    set_error("cgnn_edge_block: no kernel for latent=%d hidden=%d", latent, hidden);
    return CGNN_ERR_UNSUPPORTED;
}
