// 16-row ("N16") building blocks: one wave owns 16 edges and uses v_mfma_f32_16x16x32_bf16.
//
// Why a second tile shape: at two waves per SIMD a wave has 256 registers.  With 32-row tiles the f32 edge
// tile alone is 64 of them and does not fit next to accumulators, operands and weight fragments, so it had
// to be read twice from L2/HBM (once as the MFMA operand, once for the residual).  Halving the rows halves
// every per-wave array (tile 32, accumulators 32, operands 16+16 registers) and the tile stays in registers.
// The price is half the reuse of each LDS weight fragment, which the HBM-bound kernel can afford.
//
// N16 act layout: the edge index sits on the MFMA column c = lane & 15, features on the MFMA row; for the
// 16-feature tile O lane (c, q = lane >> 4) holds features 16 O + 4 q + i in register i (0..3): the C/D map of
// v_mfma_f32_16x16x32_bf16 (col = lane & 15, row = 4 (lane >> 4) + i).  Two consecutive accumulator tiles
// (2s, 2s+1) form the B operand of k-step s: element j of lane (c, q) is feature
//        phi(s, q, j) = 32 s + 16 (j >> 2) + 4 q + (j & 3),
// and the weights are packed (CGNN_BF16_N16) with exactly that k order.
#pragma once
#include "mlp_device.hpp"

namespace cgnn {

typedef const __attribute__((address_space(3))) f32x4* LdsVec4Ptr;

// out[O] += W[16 O .. 16 O + 15, :] . in    fragment m = O * KS + s at wp[m * 64 + lane]
template <int KS, int OT, int GSMAX = 4>
__device__ __forceinline__ void dense16(f32x4 (&out)[OT], const bf16x8 (&in)[KS], const LdsW& wp, int lane) {
    constexpr int M = OT * KS;
    constexpr int GS = (M < GSMAX) ? M : GSMAX;
    constexpr int NG = M / GS;
    static_assert(M % GS == 0, "group size must divide the MFMA count");
    bf16x8 buf[2][GS];
#pragma unroll
    for (int j = 0; j < GS; ++j) buf[0][j] = wp.fetch(j, lane);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) {
#pragma unroll
            for (int j = 0; j < GS; ++j) buf[(g + 1) & 1][j] = wp.fetch((g + 1) * GS + j, lane);
        }
#pragma unroll
        for (int j = 0; j < GS; ++j) {
            const int mm = g * GS + j;
            const int o = mm / KS, s = mm % KS;
            out[o] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(buf[g & 1][j], in[s], out[o], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// fragments [M0, M1) only, read from a source holding exactly that range
template <int KS, int OT, int M0, int M1>
__device__ __forceinline__ void dense16_part(f32x4 (&out)[OT], const bf16x8 (&in)[KS], const LdsW& wp, int lane) {
    constexpr int M = M1 - M0;
    constexpr int GS = (M < 4) ? M : 4;
    constexpr int NG = M / GS;
    static_assert(M % GS == 0 && M1 <= OT * KS, "bad fragment range");
    bf16x8 buf[2][GS];
#pragma unroll
    for (int j = 0; j < GS; ++j) buf[0][j] = wp.fetch(j, lane);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) {
#pragma unroll
            for (int j = 0; j < GS; ++j) buf[(g + 1) & 1][j] = wp.fetch((g + 1) * GS + j, lane);
        }
#pragma unroll
        for (int j = 0; j < GS; ++j) {
            const int mm = M0 + g * GS + j;
            const int o = mm / KS, s = mm % KS;
            out[o] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(buf[g & 1][j], in[s], out[o], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <bool RELU, int KS>
__device__ __forceinline__ void operand16(bf16x8 (&op)[KS], const f32x4 (&acc)[2 * KS]) {
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = acc[2 * s + (j >> 2)][j & 3];
            op[s][j] = (__bf16)(RELU ? fmaxf(x, 0.f) : x);
        }
}

template <int OT>
__device__ __forceinline__ void fill16(f32x4 (&acc)[OT], LdsVecPtr b, int q) {
#pragma unroll
    for (int o = 0; o < OT; ++o) acc[o] = *(LdsVec4Ptr)(b + 16 * o + 4 * q);
}

// P rows for the N16 kernel: bf16, H values per row stored [q][O][i]; lane (c, q) reads H/4 contiguous values.
template <int OT, bool ADD>
__device__ __forceinline__ void load_p16(f32x4 (&acc)[OT], const __bf16* __restrict__ base, int64_t row, int q) {
    const bf16x8* p = reinterpret_cast<const bf16x8*>(base + row * (16 * OT) + q * (4 * OT));
#pragma unroll
    for (int j = 0; j < OT / 2; ++j) {
        const bf16x8 v = p[j];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (ADD)
                acc[2 * j + (c >> 2)][c & 3] += (float)v[c];
            else
                acc[2 * j + (c >> 2)][c & 3] = (float)v[c];
        }
    }
}

// LayerNorm over the 16 OT features of each edge; an edge's features live on lanes c, c+16, c+32, c+48.
template <int OT>
__device__ __forceinline__ void layer_norm16(f32x4 (&a)[OT], LdsVecPtr gamma, LdsVecPtr beta, int q) {
    float s = 0.f;
#pragma unroll
    for (int o = 0; o < OT; ++o)
#pragma unroll
        for (int i = 0; i < 4; ++i) s += a[o][i];
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    const float mean = s * (1.0f / (16 * OT));
    float v = 0.f;
#pragma unroll
    for (int o = 0; o < OT; ++o)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float d = a[o][i] - mean;
            v += d * d;
        }
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    const float rstd = 1.0f / sqrtf(v * (1.0f / (16 * OT)) + 1e-5f);
#pragma unroll
    for (int o = 0; o < OT; ++o) {
        const f32x4 gm = *(LdsVec4Ptr)(gamma + 16 * o + 4 * q);
        const f32x4 bt = *(LdsVec4Ptr)(beta + 16 * o + 4 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[o][i] = (a[o][i] - mean) * rstd * gm[i] + bt[i];
    }
}

// Sum over the 8 or 16 consecutive lanes of a DPP row that share a receiver (fixed in-degree 8 or 16); every lane
// of the segment ends up with the total.  Butterfly: quad xor 1, quad xor 2, half-row mirror, row mirror.
template <int SEG>
__device__ __forceinline__ float segment_sum(float v) {
    static_assert(SEG == 8 || SEG == 16, "segments of 8 or 16 lanes");
    int t = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, t, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    t = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, t, 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    t = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, t, 0x141, 0xF, 0xF, true));   // row_half_mirror
    if (SEG == 16) {
        t = __builtin_bit_cast(int, v);
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, t, 0x140, 0xF, 0xF, true));  // row_mirror
    }
    return v;
}

// float offset, inside a TILED32 tile, of the 16-byte chunk lane (c, q) owns for 16-feature tile O, for the
// 16-row half `half` of the tile: feature 16 O + 4 q = 32 t + 8 g + 4 h with t = O >> 1, g = 2 (O & 1) + (q >> 1),
// h = q & 1; row r = 16 half + c.
__device__ __forceinline__ int n16_lane_offset(int c, int q, int half) {
    return ((q >> 1) * 64 + 32 * (q & 1) + 16 * half + c) * 4;
}
__device__ __forceinline__ constexpr int n16_tile_offset(int O) { return (4 * (O >> 1) + 2 * (O & 1)) * 64 * 4; }

}  // namespace cgnn

// ---------------------------------------------------------------------------------------------------------
// CGNN_F32X3 arithmetic in the N16 layout (node kernel): three bf16 terms per value, six MFMAs per fragment.
namespace cgnn {

template <bool RELU, int KS>
__device__ __forceinline__ void operand16x3(bf16x8 (&op)[3][KS], const f32x4 (&acc)[2 * KS]) {
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float x = acc[2 * s + (j >> 2)][j & 3];
            if (RELU) x = fmaxf(x, 0.f);
            const __bf16 x1 = (__bf16)x;
            const float r1 = x - (float)x1;
            const __bf16 x2 = (__bf16)r1;
            op[0][s][j] = x1;
            op[1][s][j] = x2;
            op[2][s][j] = (__bf16)(r1 - (float)x2);
        }
}

// fragments [M0, M1) of a layer (m = O * KS + s), read from an LDS chunk holding exactly that range
template <int KS, int OT, int M0, int M1>
__device__ __forceinline__ void dense16x3_part(f32x4 (&out)[OT], const bf16x8 (&in)[3][KS], const LdsWx3& wp, int lane) {
    constexpr int M = M1 - M0;
    constexpr int GS = (M < 2) ? M : 2;
    constexpr int NG = M / GS;
    static_assert(M % GS == 0 && M1 <= OT * KS, "bad fragment range");
    bf16x8x3 buf[2][GS];
#pragma unroll
    for (int j = 0; j < GS; ++j) buf[0][j] = wp.fetch(j, lane);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) {
#pragma unroll
            for (int j = 0; j < GS; ++j) buf[(g + 1) & 1][j] = wp.fetch((g + 1) * GS + j, lane);
        }
#pragma unroll
        for (int j = 0; j < GS; ++j) {
            const int mm = M0 + g * GS + j;
            const int o = mm / KS, s = mm % KS;
            const bf16x8x3& a = buf[g & 1][j];
            f32x4 c = out[o];
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[2], in[0][s], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], in[2][s], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[1], in[1][s], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[1], in[0][s], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], in[1][s], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], in[0][s], c, 0, 0, 0);
            out[o] = c;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// bias / LayerNorm vectors straight from global memory (the node kernel's LDS is taken by the weight ring)
template <int OT>
__device__ __forceinline__ void fill16_global(f32x4 (&acc)[OT], const float* __restrict__ b, int q) {
#pragma unroll
    for (int o = 0; o < OT; ++o) {
        if (b != nullptr)
            acc[o] = *reinterpret_cast<const f32x4*>(b + 16 * o + 4 * q);
        else
            acc[o] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

template <int OT>
__device__ __forceinline__ void layer_norm16_global(f32x4 (&a)[OT], const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, int q) {
    float s = 0.f;
#pragma unroll
    for (int o = 0; o < OT; ++o)
#pragma unroll
        for (int i = 0; i < 4; ++i) s += a[o][i];
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    const float mean = s * (1.0f / (16 * OT));
    float v = 0.f;
#pragma unroll
    for (int o = 0; o < OT; ++o)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float d = a[o][i] - mean;
            v += d * d;
        }
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    const float rstd = 1.0f / sqrtf(v * (1.0f / (16 * OT)) + 1e-5f);
#pragma unroll
    for (int o = 0; o < OT; ++o) {
        const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + 16 * o + 4 * q);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + 16 * o + 4 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[o][i] = (a[o][i] - mean) * rstd * gm[i] + bt[i];
    }
}

// P rows written from the N16 layout.  S16: lane (c, q) owns the contiguous run [q][O][i].  S32: feature
// 16 O + 4 q + i = 32 t + 8 g + 4 h + i with t = O >> 1, g = 2 (O & 1) + (q >> 1), h = q & 1.
template <int PFMT, int OT>
__device__ __forceinline__ void store_p16(const f32x4 (&acc)[OT], __bf16* __restrict__ base, int64_t row, int q) {
    __bf16* rp = base + row * (16 * OT);
    if (PFMT == CGNN_P_BF16_S16) {
        bf16x8* p = reinterpret_cast<bf16x8*>(rp + q * (4 * OT));
#pragma unroll
        for (int j = 0; j < OT / 2; ++j) {
            bf16x8 v;
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = (__bf16)acc[2 * j + (c >> 2)][c & 3];
            p[j] = v;
        }
    } else {
#pragma unroll
        for (int o = 0; o < OT; ++o) {
            const int t = o >> 1, g = 2 * (o & 1) + (q >> 1), hh = q & 1;
            bf16x4 v;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = (__bf16)acc[o][c];
            *reinterpret_cast<bf16x4*>(rp + hh * (8 * OT) + (4 * t + g) * 4) = v;
        }
    }
}

}  // namespace cgnn
