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
#include <type_traits>
#include <utility>

#include "mlp_device.hpp"

namespace cgnn {

typedef const __attribute__((address_space(3))) f32x4* LdsVec4Ptr;

template <class F, int... I>
__device__ __forceinline__ void static_for_each(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}

// out[O] += W[16 O .. 16 O + 15, :] . in    fragment m = O * KS + s at wp[m * 64 + lane]
// Issue order: blocks of OB = 4 output tiles, k-step by k-step, so that consecutive MFMAs write different
// accumulators and an accumulator recurs only every fourth MFMA (a chain of back-to-back MFMAs on one accumulator
// exposes the instruction's latency, about twice its issue time for 16x16x32).
template <int KS, int OT, int GSMAX = 4, int NB = 2>
__device__ __forceinline__ void dense16(f32x4 (&out)[OT], const bf16x8 (&in)[KS], const LdsW& wp, int lane) {
    constexpr int M = OT * KS;
    constexpr int OB = (OT % 4 == 0) ? 4 : ((OT % 2 == 0) ? 2 : 1);
    constexpr int GS = (OB < GSMAX && M % GSMAX == 0 && OB == 4) ? GSMAX : OB;   // one (block, k-step) per group
    constexpr int NG = M / GS;
    static_assert(M % GS == 0 && GS % OB == 0, "group size must divide the MFMA count");
    // issue index t -> (o, s): t = (ob * KS + s) * OB + oo
#define CGNN_D16_O(t) (((t) / (KS * OB)) * OB + (t) % OB)
#define CGNN_D16_S(t) (((t) / OB) % KS)
    // NB - 1 groups of fragments in flight ahead of the MFMAs (the LDS latency under load is several groups long)
    bf16x8 buf[NB][GS];
#pragma unroll
    for (int p = 0; p < NB - 1; ++p)
        if (p < NG) {
#pragma unroll
            for (int j = 0; j < GS; ++j) {
                const int t = p * GS + j;
                buf[p][j] = wp.fetch(CGNN_D16_O(t) * KS + CGNN_D16_S(t), lane);
            }
        }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g + NB - 1 < NG) {
#pragma unroll
            for (int j = 0; j < GS; ++j) {
                const int t = (g + NB - 1) * GS + j;
                buf[(g + NB - 1) % NB][j] = wp.fetch(CGNN_D16_O(t) * KS + CGNN_D16_S(t), lane);
            }
        }
#pragma unroll
        for (int j = 0; j < GS; ++j) {
            const int t = g * GS + j;
            out[CGNN_D16_O(t)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(buf[g % NB][j], in[CGNN_D16_S(t)],
                                                                         out[CGNN_D16_O(t)], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#undef CGNN_D16_O
#undef CGNN_D16_S
}

// dense16 with the LDS fragment pipeline written by hand (ds_read_b128 and counted s_waitcnt lgkmcnt in inline asm).
// hipcc's own waits in these loops are always lgkmcnt(0): every group of MFMAs then waits for the reads just issued for
// the groups after it, and the prefetch hides nothing (the kernel sat 40 % of its time in s_waitcnt).  Here NB - 1
// groups of GS fragments stay in flight: before group g is used the wave waits until at most GS * (groups issued
// after g) reads are outstanding -- LDS reads return in order.  The caller must not have scalar loads in flight
// (they share the counter and return out of order); compiler-tracked LDS reads issued earlier are only older entries.
template <int IMM>
__device__ __forceinline__ u32x4 lds_read_b128(unsigned addr) {
    u32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(IMM));
    return r;
}
template <int N>
__device__ __forceinline__ void lds_wait4(u32x4& a, u32x4& b, u32x4& c, u32x4& d) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}

struct NoBetween {
    template <class G>
    __device__ __forceinline__ void operator()(G) const {}
};
// `between(group)` runs after each group's MFMAs have been issued: the place for a caller's vector-memory
// instructions (one or two per group), whose slow issue then overlaps the matrix pipe instead of preceding it.
template <int KS, int OT, int NB = 3, class Between = NoBetween>
__device__ __forceinline__ void dense16_pipelined(f32x4 (&out)[OT], const bf16x8 (&in)[KS], const LdsW& wp, int lane,
                                                  Between&& between = Between{}) {
    constexpr int M = OT * KS, GS = 4, NG = M / GS;
    static_assert(OT % 4 == 0 && NB >= 2 && (NB - 1) * GS <= 15, "blocks of four output tiles; lgkmcnt is 4 bits");
    // issue index t -> (o, s): t = (ob * KS + s) * 4 + oo  (consecutive MFMAs write different accumulators)
#define CGNN_D16_O(t) (((t) / (KS * 4)) * 4 + (t) % 4)
#define CGNN_D16_S(t) (((t) / 4) % KS)
    const unsigned addr = (unsigned)(uintptr_t)wp.p + (unsigned)lane * 16u;
    u32x4 buf[NB][GS];
    static_for_each([&](auto pc) {
        constexpr int p = decltype(pc)::value;
        if constexpr (p < NG) {
            static_for_each([&](auto jc) {
                constexpr int t = p * GS + decltype(jc)::value;
                buf[p][decltype(jc)::value] = lds_read_b128<(CGNN_D16_O(t) * KS + CGNN_D16_S(t)) * 1024>(addr);
            }, std::make_integer_sequence<int, GS>{});
        }
    }, std::make_integer_sequence<int, NB - 1>{});
    static_for_each([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        if constexpr (g + NB - 1 < NG) {
            static_for_each([&](auto jc) {
                constexpr int t = (g + NB - 1) * GS + decltype(jc)::value;
                buf[(g + NB - 1) % NB][decltype(jc)::value] =
                    lds_read_b128<(CGNN_D16_O(t) * KS + CGNN_D16_S(t)) * 1024>(addr);
            }, std::make_integer_sequence<int, GS>{});
        }
        constexpr int newer = ((g + NB - 1 < NG ? g + NB - 1 : NG - 1) - g) * GS;     // reads issued after group g's
        lds_wait4<newer>(buf[g % NB][0], buf[g % NB][1], buf[g % NB][2], buf[g % NB][3]);
        static_for_each([&](auto jc) {
            constexpr int j = decltype(jc)::value, t = g * GS + j;
            out[CGNN_D16_O(t)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(bf16x8, buf[g % NB][j]), in[CGNN_D16_S(t)], out[CGNN_D16_O(t)], 0, 0, 0);
        }, std::make_integer_sequence<int, GS>{});
        between(gc);
        // no sched_barrier here: the asm statements keep the reads, waits and MFMA groups in order by themselves, and
        // the compiler may slide independent vector work between the groups (23.7 -> 23.2 ms for cgnn_edge_stream)
    }, std::make_integer_sequence<int, NG>{});
#undef CGNN_D16_O
#undef CGNN_D16_S
}

// `between` is called with integral constants 0 .. groups-1 (see dense16_pipelined); the fallback for narrow layers
// runs all of them after its MFMAs.
template <int KS, int OT, int NB = 3, class Between = NoBetween>
__device__ __forceinline__ void dense16_fast(f32x4 (&out)[OT], const bf16x8 (&in)[KS], const LdsW& wp, int lane,
                                             Between&& between = Between{}) {
    if constexpr (OT % 4 == 0) {
        dense16_pipelined<KS, OT, NB>(out, in, wp, lane, between);
    } else {
        dense16<KS, OT, 4, NB>(out, in, wp, lane);
        static_for_each(between, std::make_integer_sequence<int, (OT * KS + 3) / 4>{});
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

// ReLU on packed bf16: as 16-bit integers the negative values (sign bit set, -0 included) are the negative integers,
// so one v_pk_max_i16 per register does two values (the f32 form costs one v_max_f32 per value before the pack).
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 relu_bf16(bf16x8 v) {
    const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    return __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, v), z));
}

// two f32 -> one register of two bf16 (a single v_cvt_pk_bf16_f32; element-wise conversions made hipcc convert each
// value alone and merge the halves with v_perm_b32)
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

template <bool RELU, int KS>
__device__ __forceinline__ void operand16(bf16x8 (&op)[KS], const f32x4 (&acc)[2 * KS]) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        u32x4 v;
        v[0] = pack_bf16(acc[2 * s][0], acc[2 * s][1]);
        v[1] = pack_bf16(acc[2 * s][2], acc[2 * s][3]);
        v[2] = pack_bf16(acc[2 * s + 1][0], acc[2 * s + 1][1]);
        v[3] = pack_bf16(acc[2 * s + 1][2], acc[2 * s + 1][3]);
        const bf16x8 b = __builtin_bit_cast(bf16x8, v);
        op[s] = RELU ? relu_bf16(b) : b;
    }
}

template <int OT>
__device__ __forceinline__ void fill16(f32x4 (&acc)[OT], LdsVecPtr b, int q) {
#pragma unroll
    for (int o = 0; o < OT; ++o) acc[o] = *(LdsVec4Ptr)(b + 16 * o + 4 * q);
}

// P rows for the N16 kernel (CGNN_P_BF16_S16): bf16, stored in B-operand order -- for k-step s (features 32 s .. 32 s
// + 31) lane (c, q) owns the 16 bytes at (4 s + q) * 16, elements j = features phi(s, q, j).  The rows then enter the
// first layer's accumulators through the matrix pipe (two MFMAs per 16-feature tile against constant 0/1 selector
// fragments) instead of 2 x 32 bf16 -> f32 unpacks and 32 adds per lane on the vector pipe, which was the busier one.
template <int KS>
__device__ __forceinline__ void load_p16_operand(bf16x8 (&op)[KS], const __bf16* __restrict__ base, int64_t row, int q) {
    const bf16x8* p = reinterpret_cast<const bf16x8*>(base + row * (32 * KS)) + q;
#pragma unroll
    for (int s = 0; s < KS; ++s) op[s] = p[4 * s];
}
// The same loads issued from inline asm, for callers that do their own vmcnt accounting (cgnn_edge_stream): hipcc
// guards registers loaded across a loop's back edge with s_waitcnt vmcnt(0), which there also waits for the weight
// ring's LDS-DMA issued moments earlier.  The caller must wait (counted) before p16_ready() hands the registers on.
template <int KS>
__device__ __forceinline__ void load_p16_operand_untracked(bf16x8 (&op)[KS], const __bf16* __restrict__ base,
                                                           int64_t row, int q) {
    const __bf16* p = base + row * (32 * KS) + q * 8;
    static_for_each([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        u32x4 r;
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(r) : "v"(p), "n"(s * 64));
        op[s] = __builtin_bit_cast(bf16x8, r);
    }, std::make_integer_sequence<int, KS>{});
}
template <int S>
__device__ __forceinline__ bf16x8 load_p16_step_untracked(const __bf16* rowq) {   // rowq = base + row*(32 KS) + 8 q
    u32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(r) : "v"(rowq), "n"(S * 64));
    return __builtin_bit_cast(bf16x8, r);
}
template <int KS>
__device__ __forceinline__ void p16_ready(bf16x8 (&a)[KS], bf16x8 (&b)[KS]) {   // orders their uses after this point
    static_for_each([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        u32x4 x = __builtin_bit_cast(u32x4, a[s]), y = __builtin_bit_cast(u32x4, b[s]);
        asm volatile("" : "+v"(x), "+v"(y));
        a[s] = __builtin_bit_cast(bf16x8, x);
        b[s] = __builtin_bit_cast(bf16x8, y);
    }, std::make_integer_sequence<int, KS>{});
}

// A fragment that copies the 16 features {phi(s, q', j') : j' >> 2 == sub} of a k-step to the 16 rows of a C tile:
// row m = 4 q' + (j' & 3)  <-  k = 8 q' + j' = 8 (m >> 2) + 4 sub + (m & 3); lane (m, q) holds A[m][8 q + j].
__device__ __forceinline__ bf16x8 p16_selector(int lane, int sub) {
    const int m = lane & 15, q = lane >> 4;
    bf16x8 a;
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = (__bf16)((q == (m >> 2) && j == 4 * sub + (m & 3)) ? 1.0f : 0.0f);
    return a;
}
template <int KS>
__device__ __forceinline__ void p16_accumulate(f32x4 (&acc)[2 * KS], const bf16x8 (&ps)[KS], const bf16x8 (&pd)[KS],
                                               bf16x8 sel0, bf16x8 sel1) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int o = 0; o < 2 * KS; ++o)
        acc[o] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((o & 1) ? sel1 : sel0, ps[o >> 1], zero, 0, 0, 0);
#pragma unroll
    for (int o = 0; o < 2 * KS; ++o)
        acc[o] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((o & 1) ? sel1 : sel0, pd[o >> 1], acc[o], 0, 0, 0);
}

// The same rows through the vector pipe instead: acc (+)= f32(P row).  Element j of k-step s is tile 2 s + (j >> 2),
// register j & 3.  (The matrix-pipe form wins where four waves per SIMD keep that pipe fed; with two waves per SIMD
// and a compute-bound loop the unpack is cheaper than sixteen more MFMAs.)
template <int KS, bool ADD>
__device__ __forceinline__ void p16_unpack(f32x4 (&acc)[2 * KS], const bf16x8 (&p)[KS]) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const u32x4 w = __builtin_bit_cast(u32x4, p[s]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float lo = __builtin_bit_cast(float, w[t] << 16), hi = __builtin_bit_cast(float, w[t] & 0xffff0000u);
            f32x4& a = acc[2 * s + (t >> 1)];
            if (ADD) {
                a[2 * (t & 1)] += lo;
                a[2 * (t & 1) + 1] += hi;
            } else {
                a[2 * (t & 1)] = lo;
                a[2 * (t & 1) + 1] = hi;
            }
        }
    }
}

// Sum over the four 16-lane rows of a wave (lanes c, c+16, c+32, c+48), result in all of them: gfx950's
// v_permlane16_swap / v_permlane32_swap exchange rows between two registers on the vector pipe; __shfl_xor(.., 16 / 32)
// compiles to ds_bpermute_b32, an LDS round trip per step (four dependent ones per LayerNorm).  Same additions, same
// bits.  Inline asm: with the builtins hipcc (ROCm 7.2) merged the two results when both operands held the same value
// (scripts/dev/permlane_test.hip).
__device__ __forceinline__ float rows_sum(float s) {
    float a = s, b = s;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    a += b;
    b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}

// LayerNorm over the 16 OT features of each edge; an edge's features live on lanes c, c+16, c+32, c+48.  Two passes
// (mean, then centred squares) on whole f32x4 registers so that hipcc emits packed f32 instructions.
template <int OT, typename P = LdsVecPtr>
__device__ __forceinline__ void layer_norm16(f32x4 (&a)[OT], P gamma, P beta, int q) {
    f32x4 s4 = a[0];
#pragma unroll
    for (int o = 1; o < OT; ++o) s4 += a[o];
    const float s = rows_sum((s4[0] + s4[1]) + (s4[2] + s4[3]));
    const float mean = s * (1.0f / (16 * OT));
    f32x4 v4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int o = 0; o < OT; ++o) {
        a[o] -= mean;
        v4 += a[o] * a[o];
    }
    const float v = rows_sum((v4[0] + v4[1]) + (v4[2] + v4[3]));
    const float rstd = 1.0f / sqrtf(v * (1.0f / (16 * OT)) + 1e-5f);
#pragma unroll
    for (int o = 0; o < OT; ++o) {
        const f32x4 gm = *(typename VecOf4<P>::type)(gamma + 16 * o + 4 * q);
        const f32x4 bt = *(typename VecOf4<P>::type)(beta + 16 * o + 4 * q);
        a[o] = a[o] * (gm * rstd) + bt;
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

// fragments [M0, M1) of a layer (m = O * KS + s), read from an LDS chunk holding exactly that range.  (Alternating two
// accumulators term by term instead of six back-to-back MFMAs on one was measured: 0.95 vs 0.92 ms, no gain; so was a
// hand-written fragment pipeline with counted lgkmcnt waits, as in dense16_pipelined: 0.93 vs 0.94 ms.)
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

// ---------------------------------------------------------------------------------------------------------
// CGNN_F16X2 arithmetic in the N16 layout: f32 emulated with TWO fp16 terms per value and three MFMAs per fragment
// (v_mfma_f32_16x16x32_f16), half the matrix work of the three-bf16-term form for the same accuracy.
//
//   x = hi + lo / 2048,   hi = fp16(x),   lo = fp16((x - hi) * 2048)
//
// hi keeps 11 significand bits, the residual x - hi is exact in f32 and has at most 13 more, of which lo keeps 11:
// |x - (hi + lo/2048)| <= 2^-23 |x|.  The residual is scaled by 2^11 before the conversion so that it stays a normal
// fp16 number wherever hi is one (unscaled it would sink into fp16's subnormals for |x| < 2^-3 and lose bits); the
// products that carry one scaled term go to a second accumulator that is folded in with weight 2^-11:
//
//   sum x w = sum hi_x hi_w  +  2^-11 (sum hi_x lo_w + sum lo_x hi_w)  +  O(2^-22 |x| |w|)   [lo_x lo_w dropped]
//
// fp16 products are exact in the f32 accumulator, fp16 subnormal operands are honoured by the instruction (probe:
// scripts/dev/probe_f16_mfma.hip), so tiny values only lose the bits below 2^-35.  The one thing fp16 cannot hold is
// |x| >= 65520: hi becomes inf and the row's results are inf/NaN (visible, never silently wrong).  Latents behind a
// LayerNorm and sums of <= 64 of them are orders of magnitude below that; CGNN_F32X3 has the f32 range.
// (f16x8, f16x8x2, pack_f16, split_f16x2: cgnn_common.hpp, shared with the 32-row form CGNN_F16X2)
template <bool RELU, int KS>
__device__ __forceinline__ void operand16f2(f16x8 (&op)[2][KS], const f32x4 (&acc)[2 * KS]) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        u32x4 hi, lo;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float a = acc[2 * s + (t >> 1)][2 * (t & 1)], b = acc[2 * s + (t >> 1)][2 * (t & 1) + 1];
            if (RELU) {     // NaN stays NaN (fmaxf would turn it into 0 and hide an fp16 overflow, see above)
                a = a < 0.f ? 0.f : a;
                b = b < 0.f ? 0.f : b;
            }
            unsigned h, l;
            split_f16x2(a, b, h, l);
            hi[t] = h;
            lo[t] = l;
        }
        op[0][s] = __builtin_bit_cast(f16x8, hi);
        op[1][s] = __builtin_bit_cast(f16x8, lo);
    }
}

// the two accumulators of a layer folded into its value: c0 + c1 / 2048
template <int OT>
__device__ __forceinline__ void fold16f2(f32x4 (&c0)[OT], const f32x4 (&c1)[OT]) {
#pragma unroll
    for (int o = 0; o < OT; ++o) c0[o] += c1[o] * CGNN_F16X2_INV_SCALE;
}

// Two-part (CGNN_F16X2_N16) fragments in LDS: [m][part][lane][8 fp16]
struct LdsWf2 {
    LdsWeightPtr p;
    __device__ __forceinline__ explicit LdsWf2(LdsWeightPtr q) : p(q) {}
    __device__ __forceinline__ f16x8x2 fetch(int m, int lane) const {
        f16x8x2 r;
        r.p[0] = __builtin_bit_cast(f16x8, p[(m * 2) * 64 + lane]);
        r.p[1] = __builtin_bit_cast(f16x8, p[(m * 2 + 1) * 64 + lane]);
        return r;
    }
};

// fragments [M0, M1) of a layer (m = O * KS + s) from an LDS chunk holding exactly that range: c0 takes hi.hi, c1 the
// two cross products.  Groups of two fragments (same output tile, consecutive k-steps); c0 and c1 alternate.
template <int KS, int OT, int M0, int M1>
__device__ __forceinline__ void dense16f2_part(f32x4 (&c0)[OT], f32x4 (&c1)[OT], const f16x8 (&in)[2][KS],
                                               const LdsWf2& wp, int lane) {
    constexpr int M = M1 - M0;
    constexpr int GS = (M < 2) ? M : 2;
    constexpr int NG = M / GS;
    static_assert(M % GS == 0 && M1 <= OT * KS, "bad fragment range");
    f16x8x2 buf[2][GS];
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
            const f16x8x2& a = buf[g & 1][j];
            c0[o] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.p[0], in[0][s], c0[o], 0, 0, 0);
            c1[o] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.p[0], in[1][s], c1[o], 0, 0, 0);
            c1[o] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.p[1], in[0][s], c1[o], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The same arithmetic for the five-slot-ring node kernel (node_block_f2.hip): a ring chunk holds the 8 fragments of
// output tiles O0, O0 + 1 over the four k-steps ([tile][k-step][part][lane]); a GROUP is the two tiles at one k-step,
// six MFMAs on four different accumulators (a dependent pair is two instructions apart).  The fragment reads are inline
// asm with counted lgkmcnt waits and run two groups ahead of the MFMAs -- across chunk boundaries too (the kernel's
// barrier for chunk q also vouches for chunk q + 1), so only the first group of a step waits for a whole LDS round trip.
struct FragPipe16f2 {
    u32x4 buf[3][4];     // [group % 3][hi(O0), lo(O0), hi(O0+1), lo(O0+1)]
    // request group g (k-step) of the chunk at LDS byte address addr (+ lane * 16 already added)
    template <int SLOT, int G>
    __device__ __forceinline__ void request(unsigned addr) {
        buf[SLOT][0] = lds_read_b128<(0 * 4 + G) * 2048>(addr);
        buf[SLOT][1] = lds_read_b128<(0 * 4 + G) * 2048 + 1024>(addr);
        buf[SLOT][2] = lds_read_b128<(1 * 4 + G) * 2048>(addr);
        buf[SLOT][3] = lds_read_b128<(1 * 4 + G) * 2048 + 1024>(addr);
    }
    // group in SLOT is needed now; NEWER reads (of this pipe) were issued after it
    template <int SLOT, int NEWER, int OT, int KS>
    __device__ __forceinline__ void run(f32x4 (&c0)[OT], f32x4 (&c1)[OT], const f16x8 (&in)[2][KS], int o0, int s) {
        lds_wait4<NEWER>(buf[SLOT][0], buf[SLOT][1], buf[SLOT][2], buf[SLOT][3]);
        const f16x8 h0 = __builtin_bit_cast(f16x8, buf[SLOT][0]), l0 = __builtin_bit_cast(f16x8, buf[SLOT][1]);
        const f16x8 h1 = __builtin_bit_cast(f16x8, buf[SLOT][2]), l1 = __builtin_bit_cast(f16x8, buf[SLOT][3]);
        c0[o0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, in[0][s], c0[o0], 0, 0, 0);
        c0[o0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, in[0][s], c0[o0 + 1], 0, 0, 0);
        c1[o0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, in[1][s], c1[o0], 0, 0, 0);
        c1[o0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, in[1][s], c1[o0 + 1], 0, 0, 0);
        c1[o0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(l0, in[0][s], c1[o0], 0, 0, 0);
        c1[o0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(l1, in[0][s], c1[o0 + 1], 0, 0, 0);
    }
};

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
    s = rows_sum(s);
    const float mean = s * (1.0f / (16 * OT));
    float v = 0.f;
#pragma unroll
    for (int o = 0; o < OT; ++o)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float d = a[o][i] - mean;
            v += d * d;
        }
    v = rows_sum(v);
    const float rstd = 1.0f / sqrtf(v * (1.0f / (16 * OT)) + 1e-5f);
#pragma unroll
    for (int o = 0; o < OT; ++o) {
        const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + 16 * o + 4 * q);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + 16 * o + 4 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[o][i] = (a[o][i] - mean) * rstd * gm[i] + bt[i];
    }
}

// P rows written from the N16 layout.  S16: B-operand order, the pair of tiles (2 s, 2 s + 1) of lane (c, q) is the
// 16-byte element (4 s + q) of the row (see load_p16_operand).  S32: feature
// 16 O + 4 q + i = 32 t + 8 g + 4 h + i with t = O >> 1, g = 2 (O & 1) + (q >> 1), h = q & 1.
template <int PFMT, int OT>
__device__ __forceinline__ void store_p16(const f32x4 (&acc)[OT], __bf16* __restrict__ base, int64_t row, int q) {
    __bf16* rp = base + row * (16 * OT);
    if constexpr (PFMT == CGNN_P_F16_S32) {      // fp16, the halves interleaved in 64-byte segments (cgnn.h)
        typedef _Float16 f16x4v __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int o = 0; o < OT; ++o) {
            const int t = o >> 1, g = 2 * (o & 1) + (q >> 1), hh = q & 1;
            const int u = (4 * t + g) * 4;      // position within the half
            f16x4v v;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = (_Float16)acc[o][c];
            *reinterpret_cast<f16x4v*>(rp + (u >> 5) * 64 + hh * 32 + (u & 31)) = v;
        }
    } else if (PFMT == CGNN_P_BF16_S16) {
        bf16x8* p = reinterpret_cast<bf16x8*>(rp) + q;
#pragma unroll
        for (int j = 0; j < OT / 2; ++j) {
            bf16x8 v;
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = (__bf16)acc[2 * j + (c >> 2)][c & 3];
            p[4 * j] = v;
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
