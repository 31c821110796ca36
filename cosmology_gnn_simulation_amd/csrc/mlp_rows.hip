// cgnn_mlp_rows: row-wise MLP (+LayerNorm) -- the encoders (reference graph_network.py:54,57)
// and the decoders (:158-159) -- and cgnn_project_nodes (the sender/receiver halves of the edge
// model's first Linear, evaluated once per node instead of once per edge).
#include <string.h>

#include "mlp_device.hpp"

namespace cgnn {

template <int PREC, bool WLDS, int K0T, int HT, int OT>
__global__ __launch_bounds__(CGNN_BLOCK) void mlp_rows_kernel(MlpDev m, const float* __restrict__ x, int64_t n,
                                                              int ld_x, float* __restrict__ y, int ld_y,
                                                              int y_tiled) {
    if (WLDS) stage_weights_to_lds(m, 0);
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int64_t tiles = (n + 31) / 32;
    const int in_dim = m.in_dim[0], out_dim = m.out_dim[m.nh];
    const bool in_full = (in_dim == 32 * K0T) && (ld_x % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    const bool out_full = (out_dim == 32 * OT) && (ld_y % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
    const TileRange tr = tile_range(tiles);
    // The next tile's rows are requested as soon as this tile's have become the first operand: a workgroup with its weights in
    // LDS is alone on its CU (one wave per SIMD), and with load -> MLP -> store in sequence per tile nothing overlapped the
    // trip to memory.
    f32x16 a[K0T];
    auto load_tile_rows = [&](int64_t t) __attribute__((always_inline)) {
        const int64_t rw = t * 32 + r;
        const int64_t rc = rw < n ? rw : n - 1;
        if (in_full)
            load_rows_full<K0T>(a, x + rc * ld_x, h);
        else
            load_rows_ragged<K0T>(a, x + rc * ld_x, in_dim, h);
    };
    // (up to 128 wide: at 256 the second input tile does not fit next to the accumulators -- hundreds of spilled registers)
    constexpr bool AHEAD = K0T <= 4 && HT <= 4 && OT <= 4;
    if (AHEAD && tr.first < tr.end) load_tile_rows(tr.first);
    for (int64_t tile = tr.first; tile < tr.end; tile += tr.stride) {
        const int64_t row = tile * 32 + r;
        Operand<PREC, K0T> op0;
        if (!AHEAD) load_tile_rows(tile);
        op0.template from_acc<false>(a);
        if (AHEAD && tile + tr.stride < tr.end) load_tile_rows(tile + tr.stride);
        Operand<PREC, HT> oph;
        {
            f32x16 acc[HT];
            acc_fill_bias<HT>(acc, VecSel<WLDS>::bias(m, 0), m.out_dim[0], h);
            dense<K0T, HT>(acc, op0, WSel<PREC, WLDS>::get(m, 0), lane);
            oph.template from_acc<true>(acc);
        }
        f32x16 out[OT];
        mlp_tail<PREC, WLDS, HT, OT>(m, oph, out, lane);
        if (m.gamma != nullptr) layer_norm_rows<OT>(out, VecSel<WLDS>::gamma(m), VecSel<WLDS>::beta(m), h);
        if (y_tiled) {
            store_tile<OT>(out, y + tile * (1024 * OT), lane);
        } else if (row < n) {
            if (out_full)
                store_rows_full<OT>(out, y + row * ld_y, h);
            else
                store_rows_ragged<OT>(out, y + row * ld_y, out_dim, h);
        }
    }
}

// weight sources of the projection kernel: global (buffer loads, every wave streams both matrices from L2 per tile) or,
// WLDS, both matrices copied into LDS once per workgroup (bf16: 2 x 32 KiB at 128 x 128, two fp16 terms: 2 x 64 KiB)
template <int PREC, bool WLDS>
struct ProjW {
    typedef BufW<PREC> type;
    static __device__ __forceinline__ type get(const void* w, unsigned bytes, int) { return type(w, bytes); }
};
template <>
struct ProjW<CGNN_BF16, true> {
    typedef LdsW type;
    static __device__ __forceinline__ type get(const void*, unsigned bytes, int which) {
        return LdsW((LdsWeightPtr)(cgnn_smem + which * bytes));
    }
};
template <>
struct ProjW<CGNN_F16X2, true> {
    typedef LdsWf2g type;
    static __device__ __forceinline__ type get(const void*, unsigned bytes, int which) {
        return LdsWf2g((LdsWeightPtr)(cgnn_smem + which * bytes));
    }
};

template <int PREC, int PFMT, int DT, int HT, bool WLDS = false>
__global__ __launch_bounds__(CGNN_BLOCK) void project_kernel(const void* ws, const void* wd, const float* __restrict__ bd,
                                                             int hidden, const float* __restrict__ x, int64_t n,
                                                             typename PFmt<PFMT>::elem* __restrict__ ps,
                                                             typename PFmt<PFMT>::elem* __restrict__ pd) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int64_t tiles = (n + 31) / 32;
    constexpr unsigned wbytes = DT * HT * 1024 * (PREC == CGNN_BF16 ? 2 : (PREC == CGNN_F32X3 ? 6 : 4));
    if (WLDS) {
        typedef unsigned int u32x4p __attribute__((ext_vector_type(4)));
        u32x4p* dst = reinterpret_cast<u32x4p*>(cgnn_smem);
        const u32x4p* s0 = reinterpret_cast<const u32x4p*>(ws);
        const u32x4p* s1 = reinterpret_cast<const u32x4p*>(wd);
        for (unsigned i = threadIdx.x; i < wbytes / 16; i += blockDim.x) {
            if (ws != nullptr) dst[i] = s0[i];
            if (wd != nullptr) dst[wbytes / 16 + i] = s1[i];
        }
        __syncthreads();
    }
    const typename ProjW<PREC, WLDS>::type wsrc_s = ProjW<PREC, WLDS>::get(ws, wbytes, 0);
    const typename ProjW<PREC, WLDS>::type wsrc_d = ProjW<PREC, WLDS>::get(wd, wbytes, 1);
    const TileRange tr = tile_range(tiles);
    for (int64_t tile = tr.first; tile < tr.end; tile += tr.stride) {
        const int64_t row = tile * 32 + r;
        const int64_t rowc = row < n ? row : n - 1;
        Operand<PREC, DT> op;
        {
            f32x16 a[DT];
            load_rows_full<DT>(a, x + rowc * (32 * DT), h);
            op.template from_acc<false>(a);
        }
        if (ps != nullptr) {
            f32x16 acc[HT];
            acc_fill_bias<HT>(acc, (const float*)nullptr, hidden, h);
            dense<DT, HT>(acc, op, wsrc_s, lane);
            if (row < n) PFmt<PFMT>::template store<HT>(acc, ps, row, h);
        }
        if (pd != nullptr) {
            f32x16 acc[HT];
            acc_fill_bias<HT>(acc, bd, hidden, h);
            dense<DT, HT>(acc, op, wsrc_d, lane);
            if (row < n) PFmt<PFMT>::template store<HT>(acc, pd, row, h);
        }
    }
}

// both projection matrices resident in LDS when they fit and there are enough rows to pay for the copy (per workgroup)
template <int PREC, int PFMT, int DT, int HT>
static int launch_project(const void* ws, const void* wd, const float* bd, int hidden, const float* x, int64_t n,
                          void* ps, void* pd, hipStream_t st) {
    typedef typename PFmt<PFMT>::elem E;
    constexpr unsigned wbytes = DT * HT * 1024 * (PREC == CGNN_BF16 ? 2 : (PREC == CGNN_F32X3 ? 6 : 4));
    constexpr bool CAN = (PREC == CGNN_BF16 || PREC == CGNN_F16X2) && 2 * wbytes <= CGNN_LDS_WEIGHT_BUDGET;
    if (CAN && n >= 4096) {
        auto kern = project_kernel<PREC, PFMT, DT, HT, CAN>;
        if (2 * wbytes > 48 * 1024) {
            int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (size_t)((int)(2 * wbytes)), "hipFuncSetAttribute(project)");
            if (rc != CGNN_OK) return rc;
        }
        const int grid = grid_for_tiles((n + 31) / 32, 2 * wbytes > 76 * 1024 ? 1 : 2);
        kern<<<grid, CGNN_BLOCK, 2 * wbytes, st>>>(ws, wd, bd, hidden, x, n, (E*)ps, (E*)pd);
    } else {
        project_kernel<PREC, PFMT, DT, HT, false><<<grid_for_tiles((n + 31) / 32), CGNN_BLOCK, 0, st>>>(
            ws, wd, bd, hidden, x, n, (E*)ps, (E*)pd);
    }
    return check_hip(hipGetLastError(), "cgnn_project_nodes launch");
}

int mlp_rows_n16_encoder(const MlpDev& m, size_t lds, const float* x, int64_t n, int ld_x, float* y,
                         hipStream_t st);   // edge_block.hip

template <int PREC, bool WLDS, int K0T, int HT, int OT>
static int launch_mlp_rows(const MlpDev& m, size_t lds, const float* x, int64_t n, int ld_x, float* y, int ld_y,
                           int y_tiled, hipStream_t st) {
    auto kern = mlp_rows_kernel<PREC, WLDS, K0T, HT, OT>;
    if (WLDS && lds > 48 * 1024) {
        int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (size_t)((int)lds), "hipFuncSetAttribute(mlp_rows)");
        if (rc != CGNN_OK) return rc;
    }
    int grid = grid_for_tiles((n + 31) / 32, WLDS ? 1 : 2);
    kern<<<grid, CGNN_BLOCK, WLDS ? lds : 0, st>>>(m, x, n, ld_x, y, ld_y, y_tiled);
    return check_hip(hipGetLastError(), "cgnn_mlp_rows launch");
}

template <int PREC, int K0T, int HT, int OT>
static int dispatch_lds(const MlpDev& m, size_t lds, const float* x, int64_t n, int ld_x, float* y, int ld_y,
                        int y_tiled, hipStream_t st) {
    // weights resident in LDS when they fit (bf16 up to 128 wide; two fp16 terms, 64 KiB per 128 x 128 layer, with up to
    // two hidden layers): read from L2 by every wave they were 10 TB/s of L2 traffic for the decoders at 1 M rows
    constexpr bool CAN = (PREC == CGNN_BF16 || PREC == CGNN_F16X2) && HT <= 4 && OT <= 4 && K0T <= 4;
    if (CAN && lds <= CGNN_LDS_WEIGHT_BUDGET && n >= 4096)
        return launch_mlp_rows<PREC, CAN, K0T, HT, OT>(m, lds, x, n, ld_x, y, ld_y, y_tiled, st);
    return launch_mlp_rows<PREC, false, K0T, HT, OT>(m, lds, x, n, ld_x, y, ld_y, y_tiled, st);
}

}  // namespace cgnn

using namespace cgnn;

extern "C" {

int cgnn_mlp_rows(const cgnn_mlp* mlp, const float* x, int64_t n, int32_t ld_x, float* y, int32_t ld_y,
                  int32_t y_layout, void* stream) {
    MlpDev m;
    size_t lds = 0;
    int rc = make_mlp_dev(mlp, &m, &lds, "cgnn_mlp_rows");
    if (rc != CGNN_OK) return rc;
    if (!x || !y || n < 0 || ld_x < m.in_dim[0] || ld_y < m.out_dim[m.nh]) {
        set_error("cgnn_mlp_rows: invalid argument (n=%lld ld_x=%d in=%d ld_y=%d out=%d)", (long long)n, ld_x,
                  m.in_dim[0], ld_y, m.out_dim[m.nh]);
        return CGNN_ERR_INVALID_ARG;
    }
    if (mlp->precision == CGNN_F32X3_N16) {
        set_error("cgnn_mlp_rows: CGNN_F32X3_N16 weights are for cgnn_node_block only");
        return CGNN_ERR_UNSUPPORTED;
    }
    if (mlp->precision == CGNN_BF16_N16) {   // the edge encoder: narrow input -> TILED32 latents, 16 edges per wave
        if (y_layout != CGNN_TILED32) {
            set_error("cgnn_mlp_rows: CGNN_BF16_N16 weights write CGNN_TILED32 output only");
            return CGNN_ERR_UNSUPPORTED;
        }
        if (n == 0) return CGNN_OK;
        for (int l = 1; l < m.nh; ++l)
            if (m.in_dim[l] != m.out_dim[0] || m.out_dim[l] != m.out_dim[0]) {
                set_error("cgnn_mlp_rows: hidden layer %d has the wrong shape", l);
                return CGNN_ERR_INVALID_ARG;
            }
        return mlp_rows_n16_encoder(m, lds, x, n, ld_x, y, (hipStream_t)stream);
    }
    if (n == 0) return CGNN_OK;
    const int hidden = m.out_dim[0];
    for (int l = 1; l < m.nh; ++l)
        if (m.in_dim[l] != hidden || m.out_dim[l] != hidden) {
            set_error("cgnn_mlp_rows: hidden layer %d is %dx%d, expected %dx%d", l, m.out_dim[l], m.in_dim[l], hidden,
                      hidden);
            return CGNN_ERR_INVALID_ARG;
        }
    if (m.in_dim[m.nh] != hidden || hidden % 32 != 0) {
        set_error("cgnn_mlp_rows: hidden size %d must be a multiple of 32 and match the output layer", hidden);
        return CGNN_ERR_UNSUPPORTED;
    }
    const int K0T = (m.in_dim[0] + 31) / 32, HT = hidden / 32, OT = (m.out_dim[m.nh] + 31) / 32;
    if (m.gamma && m.out_dim[m.nh] != 32 * OT) {
        set_error("cgnn_mlp_rows: LayerNorm width %d must be a multiple of 32", m.out_dim[m.nh]);
        return CGNN_ERR_UNSUPPORTED;
    }
    if (y_layout != CGNN_ROWS && y_layout != CGNN_TILED32) {
        set_error("cgnn_mlp_rows: unknown y_layout %d", y_layout);
        return CGNN_ERR_INVALID_ARG;
    }
    const int y_tiled = y_layout == CGNN_TILED32;
    if (y_tiled && m.out_dim[m.nh] != 32 * OT) {
        set_error("cgnn_mlp_rows: CGNN_TILED32 output needs a width that is a multiple of 32 (got %d)",
                  m.out_dim[m.nh]);
        return CGNN_ERR_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    const int prec = mlp->precision;
#define CGNN_TRY(P, K, H, O)                                                              \
    if (prec == P && K0T == K && HT == H && OT == O)                                       \
        return dispatch_lds<P, K, H, O>(m, lds, x, n, ld_x, y, ld_y, y_tiled, st);
    // encoders: narrow input (<= 32 features) -> latent ; decoders: latent -> <= 32 outputs
#define CGNN_PAIR(H, D)               \
    CGNN_TRY(CGNN_F32, 1, H, D)       \
    CGNN_TRY(CGNN_BF16, 1, H, D)      \
    CGNN_TRY(CGNN_F32X3, 1, H, D)     \
    CGNN_TRY(CGNN_F16X2, 1, H, D)     \
    CGNN_TRY(CGNN_F32, D, H, 1)       \
    CGNN_TRY(CGNN_BF16, D, H, 1)      \
    CGNN_TRY(CGNN_F32X3, D, H, 1)     \
    CGNN_TRY(CGNN_F16X2, D, H, 1)
    CGNN_FOR_EACH_PAIR(CGNN_PAIR)
#undef CGNN_PAIR
    // node encoders with 33 .. 64 input features (reference config.py:18 --window_size > 8: data_utils.py:138-145 builds
    // 3 (W - 1) + W of them, 37 at W = 10), square models from 64 wide
#define CGNN_WIDE_ENC(T) \
    CGNN_TRY(CGNN_F32, 2, T, T) CGNN_TRY(CGNN_BF16, 2, T, T) CGNN_TRY(CGNN_F32X3, 2, T, T) CGNN_TRY(CGNN_F16X2, 2, T, T)
    CGNN_WIDE_ENC(2) CGNN_WIDE_ENC(4) CGNN_WIDE_ENC(8)
#undef CGNN_WIDE_ENC
#undef CGNN_TRY
    set_error("cgnn_mlp_rows: no kernel for in=%d hidden=%d out=%d (supported: input<=32 (<=64 where hidden == out >= 64) or output<=32 with "
              "(hidden,latent) in {(32,32),(64,64),(128,128),(256,256),(128,64),(128,256)})",
              m.in_dim[0], hidden, m.out_dim[m.nh]);
    return CGNN_ERR_UNSUPPORTED;
}

int cgnn_project_nodes(const cgnn_linear* ws, const cgnn_linear* wd, int32_t precision, const float* x, int64_t n,
                       void* ps, void* pd, int32_t p_format, void* stream) {
    if (!x || n < 0 || (!ps && !pd) || (ps && (!ws || !ws->w)) || (pd && (!wd || !wd->w))) {
        set_error("cgnn_project_nodes: invalid argument");
        return CGNN_ERR_INVALID_ARG;
    }
    const cgnn_linear* any = ps ? ws : wd;
    const int D = any->in_dim, H = any->out_dim;
    if ((ps && pd) && (ws->in_dim != wd->in_dim || ws->out_dim != wd->out_dim)) {
        set_error("cgnn_project_nodes: ws and wd shapes differ");
        return CGNN_ERR_INVALID_ARG;
    }
    if (D % 32 || H % 32) {
        set_error("cgnn_project_nodes: latent %d / hidden %d must be multiples of 32", D, H);
        return CGNN_ERR_UNSUPPORTED;
    }
    if (n == 0) return CGNN_OK;
    hipStream_t st = (hipStream_t)stream;
    const int DT = D / 32, HT = H / 32;
    const void* wsp = ps ? ws->w : nullptr;
    const void* wdp = pd ? wd->w : nullptr;
    const float* bd = pd ? wd->b : nullptr;
    const bool f32_like = precision == CGNN_F32 || precision == CGNN_F16X2;     // both write CGNN_P_F32 tables
    if (f32_like != (p_format == CGNN_P_F32) || (!f32_like && precision != CGNN_BF16) ||
        (p_format != CGNN_P_F32 && p_format != CGNN_P_BF16_S32 && p_format != CGNN_P_BF16_S16 && p_format != CGNN_P_F16_S32)) {
        set_error("cgnn_project_nodes: precision %d / p_format %d combination is not supported", precision, p_format);
        return CGNN_ERR_INVALID_ARG;
    }
    if (p_format == CGNN_P_F16_S32) {      // the table format of cgnn_edge_stream_run_w8: built where that kernel is
        if (HT != 4 || DT != 4) {
            set_error("cgnn_project_nodes: CGNN_P_F16_S32 tables are built for latent = hidden = 128 (got %d, %d)", D, H);
            return CGNN_ERR_UNSUPPORTED;
        }
        return launch_project<CGNN_BF16, CGNN_P_F16_S32, 4, 4>(wsp, wdp, bd, H, x, n, ps, pd, st);
    }
#define CGNN_PAIR(Hh, Dd)                                                                                          \
    if (HT == Hh && DT == Dd) {                                                                                     \
        if (p_format == CGNN_P_F32 && precision == CGNN_F16X2)                                                      \
            return launch_project<CGNN_F16X2, CGNN_P_F32, Dd, Hh>(wsp, wdp, bd, H, x, n, ps, pd, st);                \
        if (p_format == CGNN_P_F32)                                                                                 \
            return launch_project<CGNN_F32, CGNN_P_F32, Dd, Hh>(wsp, wdp, bd, H, x, n, ps, pd, st);                  \
        if (p_format == CGNN_P_BF16_S32)                                                                            \
            return launch_project<CGNN_BF16, CGNN_P_BF16_S32, Dd, Hh>(wsp, wdp, bd, H, x, n, ps, pd, st);            \
        return launch_project<CGNN_BF16, CGNN_P_BF16_S16, Dd, Hh>(wsp, wdp, bd, H, x, n, ps, pd, st);               \
    }
    CGNN_FOR_EACH_PAIR(CGNN_PAIR)
#undef CGNN_PAIR
    set_error("cgnn_project_nodes: no kernel for latent=%d hidden=%d", D, H);
    return CGNN_ERR_UNSUPPORTED;
}

}  // extern "C"
