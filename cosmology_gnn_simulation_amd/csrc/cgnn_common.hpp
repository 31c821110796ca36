// Shared device-side building blocks for the gfx950 kernels.
//
// Activation layout ("act layout").  Every fused kernel keeps a tile of 32 rows
// (edges or nodes) per wave, TRANSPOSED: the row index sits on the MFMA column
// (lane & 31) and the feature index on the MFMA row, so that a layer's f32
// accumulator tile is directly the B operand of the next layer's MFMA and the
// whole MLP -> LayerNorm -> residual chain stays in registers (no LDS round trip
// between layers).  For feature tile t (32 features) lane l = r + 32 h holds, in
// register i of the tile (0..15),
//        feature  f(t,h,i) = 32 t + 8 (i >> 2) + 4 h + (i & 3)      of row r.
// This is the C/D map of v_mfma_f32_32x32x{2_f32,16_bf16}
// (row = (i&3) + 8 (i>>2) + 4 (lane>>5), col = lane & 31).
//
// Weights are pre-packed (pack.hip) so that the A operand of MFMA step
// (out tile o, k tile kt, step) is one contiguous lane-linear fragment whose k
// order matches f(t,h,i).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "cgnn.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define CGNN_WAVE 64
#define CGNN_ROWS_PER_WAVE 32
#define CGNN_BLOCK 256
#define CGNN_WAVES_PER_BLOCK (CGNN_BLOCK / CGNN_WAVE)

// (hidden/32, latent/32) pairs the fused kernels are compiled for.
#define CGNN_FOR_EACH_PAIR(X) X(1, 1) X(2, 2) X(4, 4) X(8, 8) X(4, 2) X(4, 8)

namespace cgnn {

void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);
int grid_for_tiles(int64_t tiles_of_32_rows, int blocks_per_cu = 2, int waves_per_block = CGNN_WAVES_PER_BLOCK);
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device): the attribute is sticky, and a driver call per
// launch is host time the launch-bound configurations and HIP-graph capture do not have
int ensure_dynamic_lds(const void* kernel, size_t bytes, const char* what);

// Persistent tile loop, XCD aware.  Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an L2),
// so XCD x = blockIdx % 8 sweeps the contiguous x-th eighth of the tile range and its workgroups advance
// through it together: rows gathered by neighbouring tiles (spatially sorted particles) are then fetched
// once per L2.  Placement only affects speed, never results.
struct TileRange {
    int64_t first, end, stride;
};
__device__ __forceinline__ TileRange tile_range(int64_t tiles) {
    // the wave index is wave-uniform; telling the compiler so keeps the whole tile loop (64-bit tile index,
    // tile base pointers) in scalar registers
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int waves = blockDim.x >> 6;
    const int nb = gridDim.x, b = blockIdx.x;
    TileRange r;
    if ((nb & 7) == 0) {
        const int xcd = b & 7, slot = b >> 3, per = nb >> 3;
        const int64_t t0 = tiles * xcd / 8;
        r.first = t0 + (int64_t)slot * waves + wave;
        r.end = tiles * (xcd + 1) / 8;
        r.stride = (int64_t)per * waves;
    } else {
        r.first = (int64_t)b * waves + wave;
        r.end = tiles;
        r.stride = (int64_t)nb * waves;
    }
    return r;
}

__device__ __forceinline__ int feat_of(int t, int h, int i) { return 32 * t + 8 * (i >> 2) + 4 * h + (i & 3); }

// ---------------------------------------------------------------------------
// MFMA operands
// ---------------------------------------------------------------------------
typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
template <int PREC, int T>
struct Operand;

template <int T>
struct Operand<CGNN_F32, T> {
    f32x16 v[T];
    template <bool RELU>
    __device__ __forceinline__ void from_acc(const f32x16 (&acc)[T]) {
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) v[t][i] = RELU ? fmaxf(acc[t][i], 0.f) : acc[t][i];
    }
};

template <int T>
struct Operand<CGNN_BF16, T> {
    bf16x8 v[2 * T];
    template <bool RELU>
    __device__ __forceinline__ void from_acc(const f32x16 (&acc)[T]) {
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float x = acc[t][8 * s + j];
                    v[2 * t + s][j] = (__bf16)(RELU ? fmaxf(x, 0.f) : x);
                }
    }
};

// f32 emulated with three bf16 terms (CGNN_F32X3): x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1),
// x3 = bf16(x - x1 - x2).
struct bf16x8x3 {
    bf16x8 p[3];
};

template <int T>
struct Operand<CGNN_F32X3, T> {
    bf16x8 v[3][2 * T];
    template <bool RELU>
    __device__ __forceinline__ void from_acc(const f32x16 (&acc)[T]) {
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float x = acc[t][8 * s + j];
                    if (RELU) x = fmaxf(x, 0.f);
                    const __bf16 x1 = (__bf16)x;
                    const float r1 = x - (float)x1;
                    const __bf16 x2 = (__bf16)r1;
                    const float r2 = r1 - (float)x2;
                    v[0][2 * t + s][j] = x1;
                    v[1][2 * t + s][j] = x2;
                    v[2][2 * t + s][j] = (__bf16)r2;
                }
    }
};

// f32 emulated with two fp16 terms (CGNN_F16X2; the arithmetic is described at CGNN_F16X2 in n16.hpp):
//   x = hi + lo / 2048,  hi = fp16(x),  lo = fp16((x - hi) * 2048);  three products per element, the two that carry one
// scaled term summed in a second accumulator that dense() folds in with weight 2^-11.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
struct f16x8x2 {
    f16x8 p[2];
};
#define CGNN_F16X2_SCALE 2048.0f
#define CGNN_F16X2_INV_SCALE (1.0f / 2048.0f)

__device__ __forceinline__ unsigned pack_f16(float a, float b) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_));
}

// two values -> (hi pair, lo pair)
__device__ __forceinline__ void split_f16x2(float a, float b, unsigned& hi, unsigned& lo) {
    typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
    hi = pack_f16(a, b);
    const f16x2_ h = __builtin_bit_cast(f16x2_, hi);
    lo = pack_f16((a - (float)h[0]) * CGNN_F16X2_SCALE, (b - (float)h[1]) * CGNN_F16X2_SCALE);
}


template <int T>
struct Operand<CGNN_F16X2, T> {
    f16x8 v[2][2 * T];
    template <bool RELU>
    __device__ __forceinline__ void from_acc(const f32x16 (&acc)[T]) {
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                u32x4_ hi, lo;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float a = acc[t][8 * s + 2 * j], b = acc[t][8 * s + 2 * j + 1];
                    if (RELU) {     // NaN stays NaN (fmaxf would turn it into 0 and hide an fp16 overflow)
                        a = a < 0.f ? 0.f : a;
                        b = b < 0.f ? 0.f : b;
                    }
                    unsigned h, l;
                    split_f16x2(a, b, h, l);
                    hi[j] = h;
                    lo[j] = l;
                }
                v[0][2 * t + s] = __builtin_bit_cast(f16x8, hi);
                v[1][2 * t + s] = __builtin_bit_cast(f16x8, lo);
            }
    }
};

// out[o] += W[o-tile, :] . in   for every out tile; wp = packed weights.
//
// The packed layout is flat in MFMA issue order: fragment m = (o*KT + kt)*S + s lives at wp[m*64 + lane]
// (S = 16 k-steps of 2 for f32, 2 k-steps of 16 for bf16).  The loop is fully unrolled (both operand arrays
// must be indexed statically to stay in registers) and software-pipelined by hand: the A fragments of group
// g+1 are fetched while group g's MFMAs issue, and a sched_barrier closes each group so that hipcc cannot
// hoist every load of the layer to the top (which spills).
template <int PREC>
struct Frag;
template <>
struct Frag<CGNN_F32> {
    typedef float type;
    static constexpr int S = 16;
    static constexpr int GS = 16;
    static constexpr int NB = 2;
};
template <>
struct Frag<CGNN_BF16> {
    typedef bf16x8 type;
    static constexpr int S = 2;
    static constexpr int GS = 4;
    static constexpr int NB = 2;
};

template <>
struct Frag<CGNN_F32X3> {
    typedef bf16x8x3 type;
    static constexpr int S = 2;
    static constexpr int GS = 2;
    static constexpr int NB = 3;    // weights stream from L2 at one wave per SIMD: keep two groups in flight
};

template <>
struct Frag<CGNN_F16X2> {
    typedef f16x8x2 type;
    static constexpr int S = 2;
    static constexpr int GS = 4;
    static constexpr int NB = 3;
};

// hi.hi into c0; hi.lo and lo.hi (both scaled by 2^11) into c1
template <int KT>
__device__ __forceinline__ void mfma_step2(const f16x8x2& a, const Operand<CGNN_F16X2, KT>& in, int kt, int s, f32x16& c0,
                                           f32x16& c1) {
    const int i = 2 * kt + s;
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.p[0], in.v[0][i], c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.p[0], in.v[1][i], c1, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.p[1], in.v[0][i], c1, 0, 0, 0);
}

template <int KT>
__device__ __forceinline__ f32x16 mfma_step(const bf16x8x3& a, const Operand<CGNN_F32X3, KT>& in, int kt, int s,
                                            f32x16 c) {
    const int i = 2 * kt + s;
    // smallest terms first
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[2], in.v[0][i], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], in.v[2][i], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[1], in.v[1][i], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[1], in.v[0][i], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], in.v[1][i], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], in.v[0][i], c, 0, 0, 0);
    return c;
}

template <int KT>
__device__ __forceinline__ f32x16 mfma_step(float a, const Operand<CGNN_F32, KT>& in, int kt, int s, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, in.v[kt][s], c, 0, 0, 0);
}
template <int KT>
__device__ __forceinline__ f32x16 mfma_step(bf16x8 a, const Operand<CGNN_BF16, KT>& in, int kt, int s, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, in.v[2 * kt + s], c, 0, 0, 0);
}

// Weight sources.  Global weights are read with buffer loads (uniform descriptor + one lane-offset VGPR +
// a scalar/immediate fragment offset) so that no per-fragment 64-bit address is ever materialised; flat
// global pointers made hipcc hoist one address pair per fragment out of the persistent tile loop and spill.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) bf16x8* LdsWeightPtr;

template <int PREC>
struct BufW;
template <>
struct BufW<CGNN_F32> {
    __amdgpu_buffer_rsrc_t rsrc;
    __device__ __forceinline__ BufW(const void* p, unsigned bytes)
        : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000)) {}
    __device__ __forceinline__ float fetch(int m, int lane) const {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 4, m * 256, 0));
    }
};
template <>
struct BufW<CGNN_BF16> {
    __amdgpu_buffer_rsrc_t rsrc;
    __device__ __forceinline__ BufW(const void* p, unsigned bytes)
        : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000)) {}
    __device__ __forceinline__ bf16x8 fetch(int m, int lane) const {
        return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, m * 1024, 0));
    }
};
template <>
struct BufW<CGNN_F32X3> {
    __amdgpu_buffer_rsrc_t rsrc;
    __device__ __forceinline__ BufW(const void* p, unsigned bytes)
        : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000)) {}
    __device__ __forceinline__ bf16x8x3 fetch(int m, int lane) const {
        bf16x8x3 r;
#pragma unroll
        for (int part = 0; part < 3; ++part)
            r.p[part] = __builtin_bit_cast(
                bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, (m * 3 + part) * 1024, 0));
        return r;
    }
};
template <>
struct BufW<CGNN_F16X2> {
    __amdgpu_buffer_rsrc_t rsrc;
    __device__ __forceinline__ BufW(const void* p, unsigned bytes)
        : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000)) {}
    __device__ __forceinline__ f16x8x2 fetch(int m, int lane) const {
        f16x8x2 r;
#pragma unroll
        for (int part = 0; part < 2; ++part)
            r.p[part] = __builtin_bit_cast(
                f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, (m * 2 + part) * 1024, 0));
        return r;
    }
};
struct LdsW {
    LdsWeightPtr p;
    __device__ __forceinline__ explicit LdsW(LdsWeightPtr q) : p(q) {}
    __device__ __forceinline__ bf16x8 fetch(int m, int lane) const { return p[m * 64 + lane]; }
};

template <int KT, int OT, int PREC, typename WSrc>
__device__ __forceinline__ void dense(f32x16 (&out)[OT], const Operand<PREC, KT>& in, const WSrc& wp, int lane) {
    typedef typename Frag<PREC>::type A;
    constexpr int S = Frag<PREC>::S;
    constexpr int M = OT * KT * S;
    constexpr int GS = (M < Frag<PREC>::GS) ? M : Frag<PREC>::GS;
    constexpr int NG = M / GS;
    constexpr int NB = Frag<PREC>::NB;      // ring depth: NB - 1 groups of fragments in flight ahead of the MFMAs
    static_assert(M % GS == 0, "group size must divide the MFMA count");
    A buf[NB][GS];
    // CGNN_F16X2: second accumulator for the products scaled by 2^11.  An output tile's MFMAs are consecutive, so two
    // of them (alternating by tile) are enough: zeroed at the tile's first fragment, folded in after its last.
    constexpr bool TWO = PREC == CGNN_F16X2;
    f32x16 c1[2];
#pragma unroll
    for (int p = 0; p < NB - 1; ++p)
        if (p < NG) {
#pragma unroll
            for (int j = 0; j < GS; ++j) buf[p][j] = wp.fetch(p * GS + j, lane);
        }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g + NB - 1 < NG) {
#pragma unroll
            for (int j = 0; j < GS; ++j) buf[(g + NB - 1) % NB][j] = wp.fetch((g + NB - 1) * GS + j, lane);
        }
#pragma unroll
        for (int j = 0; j < GS; ++j) {
            const int m = g * GS + j;
            const int o = m / (KT * S), kt = (m / S) % KT, s = m % S;
            if constexpr (TWO) {
                if (m % (KT * S) == 0) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) c1[o & 1][i] = 0.f;
                }
                mfma_step2<KT>(buf[g % NB][j], in, kt, s, out[o], c1[o & 1]);
                if (m % (KT * S) == KT * S - 1) out[o] += c1[o & 1] * CGNN_F16X2_INV_SCALE;
            } else {
                out[o] = mfma_step<KT>(buf[g % NB][j], in, kt, s, out[o]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Same, restricted to fragments [M0, M1) of the layer, read from a source that holds just that range
// (fragment m at local index m - M0): the LDS-cycled kernels stream a layer through LDS in chunks.
template <int KT, int OT, int M0, int M1, int PREC, typename WSrc>
__device__ __forceinline__ void dense_part(f32x16 (&out)[OT], const Operand<PREC, KT>& in, const WSrc& wp, int lane) {
    typedef typename Frag<PREC>::type A;
    constexpr int S = Frag<PREC>::S;
    constexpr int M = M1 - M0;
    constexpr int GS = (M < Frag<PREC>::GS) ? M : Frag<PREC>::GS;
    constexpr int NG = M / GS;
    static_assert(M % GS == 0 && M1 <= OT * KT * S, "bad fragment range");
    static_assert(PREC != CGNN_F16X2, "the two-accumulator form has no chunked variant in the 32-row layout");
    A buf[2][GS];
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
            const int m = M0 + g * GS + j;
            const int o = m / (KT * S), kt = (m / S) % KT, s = m % S;
            out[o] = mfma_step<KT>(buf[g & 1][j], in, kt, s, out[o]);
        }
#ifndef CGNN_NO_PART_BARRIER
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
}

// Three-part (CGNN_F32X3) fragments resident in LDS: [m][part][lane][8 bf16].
struct LdsWx3 {
    LdsWeightPtr p;
    __device__ __forceinline__ explicit LdsWx3(LdsWeightPtr q) : p(q) {}
    __device__ __forceinline__ bf16x8x3 fetch(int m, int lane) const {
        bf16x8x3 r;
#pragma unroll
        for (int part = 0; part < 3; ++part) r.p[part] = p[(m * 3 + part) * 64 + lane];
        return r;
    }
};

// ---------------------------------------------------------------------------
// act-layout tile helpers (T tiles of 32 features, one row per lane pair)
// ---------------------------------------------------------------------------
template <typename P>
struct VecOf4;
template <>
struct VecOf4<const float*> {
    typedef const f32x4* type;
};
template <>
struct VecOf4<const __attribute__((address_space(3))) float*> {
    typedef const __attribute__((address_space(3))) f32x4* type;
};

template <int T, typename P>
__device__ __forceinline__ void acc_fill_bias(f32x16 (&acc)[T], P b, int out_dim, int h) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int f = 32 * t + 8 * g + 4 * h;
            if (b != nullptr && f + 3 < out_dim) {
                const f32x4 v = *(typename VecOf4<P>::type)(b + f);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[t][4 * g + c] = v[c];
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[t][4 * g + c] = (b != nullptr && f + c < out_dim) ? b[f + c] : 0.f;
            }
        }
}

// Full-width (dim == 32 T, ld % 4 == 0) row tile: 16-byte loads.
template <int T>
__device__ __forceinline__ void load_rows_full(f32x16 (&a)[T], const float* __restrict__ rowp, int h) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(rowp + 32 * t + 8 * g + 4 * h);
#pragma unroll
            for (int c = 0; c < 4; ++c) a[t][4 * g + c] = v[c];
        }
}

template <int T>
__device__ __forceinline__ void add_rows_full(f32x16 (&a)[T], const float* __restrict__ rowp, int h) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(rowp + 32 * t + 8 * g + 4 * h);
#pragma unroll
            for (int c = 0; c < 4; ++c) a[t][4 * g + c] += v[c];
        }
}

template <int T>
__device__ __forceinline__ void store_rows_full(const f32x16 (&a)[T], float* __restrict__ rowp, int h) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = a[t][4 * g + c];
            *reinterpret_cast<f32x4*>(rowp + 32 * t + 8 * g + 4 * h) = v;
        }
}

// TILED32 layout (cgnn_layout in cgnn.h) of an [n, 32 T] float matrix: whole 32-row tiles, each stored in act
// order, so that a wave moves its tile with 4 T fully coalesced 1-KiB instructions (lane-linear 16 B):
//   element (row 32*tile + r, feature 32t + 8g + 4h + c)  at  tile*(1024 T) + ((4t + g)*64 + 32h + r)*4 + c
// Row-per-lane access to a row-major matrix costs ~3x (loads) to ~7x (stores) more texture-addresser cycles
// per instruction (32 partial lines instead of 8 full ones), which made the edge kernel address-bound.
template <int T>
__device__ __forceinline__ void load_tile(f32x16 (&a)[T], const float* __restrict__ tile_base, int lane) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(tile_base + ((4 * t + g) * 64 + lane) * 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) a[t][4 * g + c] = v[c];
        }
}

template <int T>
__device__ __forceinline__ void add_tile(f32x16 (&a)[T], const float* __restrict__ tile_base, int lane) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(tile_base + ((4 * t + g) * 64 + lane) * 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) a[t][4 * g + c] += v[c];
        }
}

template <int T>
__device__ __forceinline__ void store_tile(const f32x16 (&a)[T], float* __restrict__ tile_base, int lane) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = a[t][4 * g + c];
            *reinterpret_cast<f32x4*>(tile_base + ((4 * t + g) * 64 + lane) * 4) = v;
        }
}

// Per-node projection rows ("P rows": Ps = x Ws^T, Pd = x Wd^T + b1) as the edge kernel gathers them.
//  f32 precision : plain row-major f32 [n, H].
//  bf16 precision: bf16, H values per row stored half-split, [h][t][g][c]: lane (r, h) owns one contiguous
//                  run of H/2 bf16 (16 HT bytes * 2), read with HT*2 16-byte loads.  Halves the gather bytes
//                  and the table footprint (better L2 residency) at the precision the bf16 MLP has anyway.
template <int PREC>
struct PRow;
template <>
struct PRow<CGNN_F32> {
    typedef float elem;
    template <int HT>
    static __device__ __forceinline__ void load(f32x16 (&acc)[HT], const float* __restrict__ base, int64_t row, int h) {
        load_rows_full<HT>(acc, base + row * (32 * HT), h);
    }
    template <int HT>
    static __device__ __forceinline__ void add(f32x16 (&acc)[HT], const float* __restrict__ base, int64_t row, int h) {
        add_rows_full<HT>(acc, base + row * (32 * HT), h);
    }
    template <int HT>
    static __device__ __forceinline__ void store(const f32x16 (&acc)[HT], float* __restrict__ base, int64_t row, int h) {
        store_rows_full<HT>(acc, base + row * (32 * HT), h);
    }
};
template <>
struct PRow<CGNN_BF16> {
    typedef __bf16 elem;
    template <int HT>
    static __device__ __forceinline__ void load(f32x16 (&acc)[HT], const __bf16* __restrict__ base, int64_t row, int h) {
        const bf16x8* p = reinterpret_cast<const bf16x8*>(base + row * (32 * HT) + h * (16 * HT));
#pragma unroll
        for (int j = 0; j < 2 * HT; ++j) {
            const bf16x8 v = p[j];
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[j >> 1][8 * (j & 1) + c] = (float)v[c];
        }
    }
    template <int HT>
    static __device__ __forceinline__ void add(f32x16 (&acc)[HT], const __bf16* __restrict__ base, int64_t row, int h) {
        const bf16x8* p = reinterpret_cast<const bf16x8*>(base + row * (32 * HT) + h * (16 * HT));
#pragma unroll
        for (int j = 0; j < 2 * HT; ++j) {
            const bf16x8 v = p[j];
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[j >> 1][8 * (j & 1) + c] += (float)v[c];
        }
    }
    template <int HT>
    static __device__ __forceinline__ void store(const f32x16 (&acc)[HT], __bf16* __restrict__ base, int64_t row, int h) {
        bf16x8* p = reinterpret_cast<bf16x8*>(base + row * (32 * HT) + h * (16 * HT));
#pragma unroll
        for (int j = 0; j < 2 * HT; ++j) {
            bf16x8 v;
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = (__bf16)acc[j >> 1][8 * (j & 1) + c];
            p[j] = v;
        }
    }
};

// CGNN_P_BF16_S16 rows written from the 32-row act layout (the projection kernel keeps 32-row tiles): lane (r, h)
// holds features 32t + 8g + 4h + c, which is k-step s = t, quarter q = 2 (g & 1) + h, element j = 4 (g >> 1) + c of the
// N16 B-operand order: 16-byte element (4 s + q) of the row (n16.hpp, load_p16_operand).
template <int HT>
__device__ __forceinline__ void store_prow_s16(const f32x16 (&acc)[HT], __bf16* __restrict__ base, int64_t row, int h) {
    __bf16* rp = base + row * (32 * HT);
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int q = 2 * (g & 1) + h;
            bf16x4 v;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = (__bf16)acc[t][4 * g + c];
            *reinterpret_cast<bf16x4*>(rp + (4 * t + q) * 8 + 4 * (g >> 1)) = v;
        }
}

// Ragged width (dim not a multiple of 32, or unaligned rows): scalar, zero padded.
template <int T>
__device__ __forceinline__ void load_rows_ragged(f32x16 (&a)[T], const float* __restrict__ rowp, int dim, int h) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int f = feat_of(t, h, i);
            a[t][i] = f < dim ? rowp[f] : 0.f;
        }
}

template <int T>
__device__ __forceinline__ void store_rows_ragged(const f32x16 (&a)[T], float* __restrict__ rowp, int dim, int h) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int f = feat_of(t, h, i);
            if (f < dim) rowp[f] = a[t][i];
        }
}

// LayerNorm over the 32 T features of each row (eps 1e-5, biased variance; two
// pass: the values are in registers).  A row's features live on lanes r and r+32.
template <int T, typename P>
__device__ __forceinline__ void layer_norm_rows(f32x16 (&a)[T], P gamma, P beta, int h) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += a[t][i];
    s += __shfl_xor(s, 32);
    const float mean = s * (1.0f / (32 * T));
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float d = a[t][i] - mean;
            q += d * d;
        }
    q += __shfl_xor(q, 32);
    const float rstd = 1.0f / sqrtf(q * (1.0f / (32 * T)) + 1e-5f);
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int f = 32 * t + 8 * g + 4 * h;
            const f32x4 gm = *(typename VecOf4<P>::type)(gamma + f);
            const f32x4 bt = *(typename VecOf4<P>::type)(beta + f);
#pragma unroll
            for (int c = 0; c < 4; ++c) a[t][4 * g + c] = (a[t][4 * g + c] - mean) * rstd * gm[c] + bt[c];
        }
}

}  // namespace cgnn
