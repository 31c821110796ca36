// Device view of a cgnn_mlp and the per-wave MLP chain used by every fused kernel.
#pragma once
#include <string.h>

#include "cgnn_common.hpp"

namespace cgnn {

#define CGNN_MAX_LAYERS (CGNN_MAX_HIDDEN_LAYERS + 1)
// dynamic LDS a kernel may claim for resident bf16 weights (160 KiB per CU)
#define CGNN_LDS_WEIGHT_BUDGET (152 * 1024)

struct MlpDev {
    const void* w[CGNN_MAX_LAYERS];
    const float* b[CGNN_MAX_LAYERS];
    int32_t in_dim[CGNN_MAX_LAYERS];
    int32_t out_dim[CGNN_MAX_LAYERS];
    uint32_t lds_off[CGNN_MAX_LAYERS];   // byte offset of the layer's packed weights in LDS (WLDS kernels)
    uint32_t bytes[CGNN_MAX_LAYERS];     // packed bytes of the layer
    uint32_t bias_off[CGNN_MAX_LAYERS];  // byte offset of the layer's bias vector in LDS (WLDS kernels)
    uint32_t gamma_off, beta_off;        // LayerNorm vectors in LDS (WLDS kernels)
    int32_t nh;                          // hidden layers; layers = nh + 1
    const float* gamma;
    const float* beta;
};

// Host: validate + flatten.  first_layer: 0 = use mlp->layer[0]; 1 = skip it (node block passes its split
// first layer separately).  Returns total packed bytes of the layers that would be LDS resident.
inline int make_mlp_dev(const cgnn_mlp* m, MlpDev* d, size_t* lds_bytes, const char* who) {
    if (!m) {
        set_error("%s: mlp is NULL", who);
        return CGNN_ERR_INVALID_ARG;
    }
    if (m->num_hidden_layers < 1 || m->num_hidden_layers > CGNN_MAX_HIDDEN_LAYERS) {
        set_error("%s: num_hidden_layers=%d outside [1,%d]", who, m->num_hidden_layers, CGNN_MAX_HIDDEN_LAYERS);
        return CGNN_ERR_UNSUPPORTED;
    }
    if (m->precision < CGNN_F32 || m->precision > CGNN_F16X2) {
        set_error("%s: unknown precision %d", who, m->precision);
        return CGNN_ERR_INVALID_ARG;
    }
    memset(d, 0, sizeof(*d));
    d->nh = m->num_hidden_layers;
    size_t off = 0;
    for (int l = 0; l <= d->nh; ++l) {
        const cgnn_linear& L = m->layer[l];
        if (!L.w || L.in_dim <= 0 || L.out_dim <= 0) {
            set_error("%s: layer %d is incomplete", who, l);
            return CGNN_ERR_INVALID_ARG;
        }
        d->w[l] = L.w;
        d->b[l] = L.b;
        d->in_dim[l] = L.in_dim;
        d->out_dim[l] = L.out_dim;
        const size_t nb = cgnn_packed_linear_bytes(L.out_dim, L.in_dim, m->precision);
        d->lds_off[l] = (uint32_t)off;
        d->bytes[l] = (uint32_t)nb;
        off += nb;
    }
    // small vectors (biases, LayerNorm affine) follow the weights in LDS: they are re-read for every tile, and
    // as global loads they were half of the kernel's vector-memory instructions
    for (int l = 0; l <= d->nh; ++l) {
        d->bias_off[l] = (uint32_t)off;
        off += ((size_t)d->out_dim[l] * 4 + 15) & ~(size_t)15;
    }
    d->gamma_off = (uint32_t)off;
    off += ((size_t)d->out_dim[d->nh] * 4 + 15) & ~(size_t)15;
    d->beta_off = (uint32_t)off;
    off += ((size_t)d->out_dim[d->nh] * 4 + 15) & ~(size_t)15;
    d->gamma = m->ln_gamma;
    d->beta = m->ln_beta;
    if ((m->ln_gamma == nullptr) != (m->ln_beta == nullptr)) {
        set_error("%s: ln_gamma and ln_beta must both be set or both be NULL", who);
        return CGNN_ERR_INVALID_ARG;
    }
    if (lds_bytes) *lds_bytes = off;
    return CGNN_OK;
}


extern __shared__ __attribute__((aligned(16))) char cgnn_smem[];

// Copy every layer's packed weights into LDS (all threads of the block), then barrier.
__device__ __forceinline__ void stage_weights_to_lds(const MlpDev& m, int first_layer) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    for (int l = first_layer; l <= m.nh; ++l) {
        const u32x4* src = reinterpret_cast<const u32x4*>(m.w[l]);
        u32x4* dst = reinterpret_cast<u32x4*>(cgnn_smem + m.lds_off[l]);
        const int n16 = (int)(m.bytes[l] >> 4);
        for (int i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
    }
    for (int l = first_layer; l <= m.nh; ++l) {
        float* dst = reinterpret_cast<float*>(cgnn_smem + m.bias_off[l]);
        for (int i = threadIdx.x; i < m.out_dim[l]; i += blockDim.x) dst[i] = m.b[l] ? m.b[l][i] : 0.f;
    }
    if (m.gamma != nullptr) {
        float* g = reinterpret_cast<float*>(cgnn_smem + m.gamma_off);
        float* b = reinterpret_cast<float*>(cgnn_smem + m.beta_off);
        for (int i = threadIdx.x; i < m.out_dim[m.nh]; i += blockDim.x) {
            g[i] = m.gamma[i];
            b[i] = m.beta[i];
        }
    }
    __syncthreads();
}

typedef const __attribute__((address_space(3))) float* LdsVecPtr;

template <bool WLDS>
struct VecSel {
    typedef const float* type;
    static __device__ __forceinline__ type bias(const MlpDev& m, int l) { return m.b[l]; }
    static __device__ __forceinline__ type gamma(const MlpDev& m) { return m.gamma; }
    static __device__ __forceinline__ type beta(const MlpDev& m) { return m.beta; }
};
template <>
struct VecSel<true> {
    typedef LdsVecPtr type;
    static __device__ __forceinline__ type bias(const MlpDev& m, int l) { return (LdsVecPtr)(cgnn_smem + m.bias_off[l]); }
    static __device__ __forceinline__ type gamma(const MlpDev& m) { return (LdsVecPtr)(cgnn_smem + m.gamma_off); }
    static __device__ __forceinline__ type beta(const MlpDev& m) { return (LdsVecPtr)(cgnn_smem + m.beta_off); }
};

template <int PREC, bool WLDS>
struct WSel {
    typedef BufW<PREC> type;
    static __device__ __forceinline__ type get(const MlpDev& m, int l) { return type(m.w[l], m.bytes[l]); }
};
template <>
struct WSel<CGNN_BF16, true> {
    typedef LdsW type;
    static __device__ __forceinline__ type get(const MlpDev& m, int l) {
        return LdsW((LdsWeightPtr)(cgnn_smem + m.lds_off[l]));
    }
};

// two-part (CGNN_F16X2) fragments resident in LDS: [m][part][lane][8 fp16]
struct LdsWf2g {
    LdsWeightPtr p;
    __device__ __forceinline__ explicit LdsWf2g(LdsWeightPtr q) : p(q) {}
    __device__ __forceinline__ f16x8x2 fetch(int m, int lane) const {
        f16x8x2 r;
        r.p[0] = __builtin_bit_cast(f16x8, p[(m * 2) * 64 + lane]);
        r.p[1] = __builtin_bit_cast(f16x8, p[(m * 2 + 1) * 64 + lane]);
        return r;
    }
};
template <>
struct WSel<CGNN_F16X2, true> {
    typedef LdsWf2g type;
    static __device__ __forceinline__ type get(const MlpDev& m, int l) {
        return LdsWf2g((LdsWeightPtr)(cgnn_smem + m.lds_off[l]));
    }
};

// Writers of the three cgnn_ptable formats from the 32-row act layout.
template <int PFMT>
struct PFmt;
template <>
struct PFmt<CGNN_P_F32> {
    typedef float elem;
    template <int HT>
    static __device__ __forceinline__ void store(const f32x16 (&a)[HT], float* b, int64_t row, int h) {
        PRow<CGNN_F32>::store<HT>(a, b, row, h);
    }
};
template <>
struct PFmt<CGNN_P_BF16_S32> {
    typedef __bf16 elem;
    template <int HT>
    static __device__ __forceinline__ void store(const f32x16 (&a)[HT], __bf16* b, int64_t row, int h) {
        PRow<CGNN_BF16>::store<HT>(a, b, row, h);
    }
};
template <>
struct PFmt<CGNN_P_F16_S32> {      // fp16; the two halves of CGNN_P_BF16_S32's row interleaved in 64-byte segments (cgnn.h)
    typedef _Float16 elem;
    template <int HT>
    static __device__ __forceinline__ void store(const f32x16 (&a)[HT], _Float16* b, int64_t row, int h) {
        typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
        f16x8v* p = reinterpret_cast<f16x8v*>(b + row * (32 * HT));      // 16-byte pieces of the row
#pragma unroll
        for (int j = 0; j < 2 * HT; ++j) {      // piece j of this lane's half: segment j >> 2, piece j & 3 within it
            f16x8v v;
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = (_Float16)a[j >> 1][8 * (j & 1) + c];
            p[(j >> 2) * 8 + h * 4 + (j & 3)] = v;
        }
    }
};
template <>
struct PFmt<CGNN_P_BF16_S16> {
    typedef __bf16 elem;
    template <int HT>
    static __device__ __forceinline__ void store(const f32x16 (&a)[HT], __bf16* b, int64_t row, int h) {
        store_prow_s16<HT>(a, b, row, h);
    }
};

// Hidden layers 1..nh-1 (Linear+ReLU, H->H) and the output layer (H->32*OT), starting from the ReLU'd
// first-layer activations already in `oph`.  Leaves the pre-LayerNorm output in `out`.
template <int PREC, bool WLDS, int HT, int OT>
__device__ __forceinline__ void mlp_tail(const MlpDev& m, Operand<PREC, HT>& oph, f32x16 (&out)[OT], int lane) {
    const int h = lane >> 5;
    for (int l = 1; l < m.nh; ++l) {
        f32x16 acc[HT];
        acc_fill_bias<HT>(acc, VecSel<WLDS>::bias(m, l), m.out_dim[l], h);
        dense<HT, HT>(acc, oph, WSel<PREC, WLDS>::get(m, l), lane);
        oph.template from_acc<true>(acc);
    }
    acc_fill_bias<OT>(out, VecSel<WLDS>::bias(m, m.nh), m.out_dim[m.nh], h);
    dense<HT, OT>(out, oph, WSel<PREC, WLDS>::get(m, m.nh), lane);
}

}  // namespace cgnn
