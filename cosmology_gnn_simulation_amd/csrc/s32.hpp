// Shared pieces of the one-launch edge-stream kernels on 32-edge MFMA tiles (edge_stream32.hip: one wave per SIMD, two
// tiles per wave; edge_stream32w.hip: two waves per SIMD, one tile per wave): the layer-chunk image geometry, counted
// vector-memory waits, MFMA-slot fillers, the P-row selector MFMAs and small helpers.
#pragma once
#include <type_traits>

#include "n16.hpp"

namespace cgnn {

#define CGNN_S32_WAVES 4
#define CGNN_S32_BLOCK (CGNN_S32_WAVES * 64)
#define CGNN_S32_SLOTS 4
#ifndef CGNN_S32_PD
#define CGNN_S32_PD 2      // groups of LDS weight fragments in flight ahead of the MFMAs
#endif

template <int DT>
struct S32Geom {
    static constexpr int D = 32 * DT, KS = 2 * DT, NROW = DT;
    static constexpr unsigned W_BYTES = (unsigned)D * D * 2;          // one D x D bf16 layer, 1-KiB fragments (o, ks)
    static constexpr unsigned VEC_OFF = W_BYTES;                       // bias[D], gamma[D], beta[D] (f32)
    static constexpr unsigned RAW = W_BYTES + 3u * D * 4;
    static constexpr unsigned PIECE = CGNN_S32_WAVES * 1024u;          // one 1-KiB LDS-DMA instruction per wave
    static constexpr unsigned STRIDE = (RAW + PIECE - 1) / PIECE * PIECE;
    static constexpr int NP = (int)(STRIDE / PIECE);                   // pieces per wave and chunk
    static constexpr unsigned LDS = CGNN_S32_SLOTS * STRIDE;
};

static inline size_t s32_stride(int latent) {
    switch (latent) {
        case 32: return S32Geom<1>::STRIDE;
        case 64: return S32Geom<2>::STRIDE;
        case 128: return S32Geom<4>::STRIDE;
        default: return 0;
    }
}

struct S32Args {
    const char* image;       // chunk c at image + c * STRIDE, consumption order: [encoder layers] round 0 layers, round 1 ...
    int32_t rounds, nh;      // a round is nh + 1 chunks
    int32_t enc_in_dim;      // > 0: the first nh + 1 chunks are the edge encoder, fed from edge_attr
};

// ---- vector-memory waits -------------------------------------------------------------------------------------------
// A wave's vector-memory operations retire in issue order, so "X has landed" is a counted s_waitcnt: at most as many
// operations outstanding as the wave has issued since X.  The loop's operations come in a fixed order (see the
// schedule at the kernel), so the counts are constants; a smaller count than the true one only waits longer.
#define CGNN_S32_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
template <int N>
__device__ __forceinline__ void vm_wait_const() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

typedef __attribute__((address_space(3))) void* LdsVoidPtrG;
typedef const __attribute__((address_space(1))) void* GlobalVoidPtrG;

// A block's MFMAs are numbered 0 .. M-1 ("slots"); `fill(slot)` runs right behind MFMA `slot` and a scheduling barrier
// pins it there: the place for the OTHER tile's vector work and this wave's memory instructions, whose issue then
// overlaps the matrix pipe (left alone, hipcc gathers the independent vector work in front of the block).
struct NoFill32 {
    template <int Q>
    __device__ __forceinline__ void run() const {}
};
template <class F>
struct FnFill32 {
    F f;
    template <int Q>
    __device__ __forceinline__ void run() const {
        f(std::integral_constant<int, Q>{});
    }
};
template <class F>
__device__ __forceinline__ FnFill32<F> make_fill(F f) {
    return FnFill32<F>{f};
}

template <int NROW, int KS>
struct WBlock {
    static constexpr int M = NROW * KS, GS = M < 4 ? M : 4, NG = M / GS, PD = CGNN_S32_PD;
    // weight-fragment reads issued between MFMA `q` and its fill (fragment q + PD * GS, if the layer has one)
    static constexpr int fragtop(int q) { return (q + PD * GS < M) ? 1 : 0; }
    // ... in front of MFMAs (q0, q1]
    static constexpr int frags_between(int q0, int q1) {
        int n = 0;
        for (int x = q0 + 1; x <= q1; ++x) n += fragtop(x);
        return n;
    }
};

// P rows (CGNN_P_BF16_S32: lane (r, h) owns the 16-byte pieces 2 t + s of its half of the row = the B operand of k-step
// (t, s)) enter the accumulators through the matrix pipe: A = a constant 0/1 selector that copies k = 8 h' + j of
// k-step s to row 16 s + 8 (j >> 2) + 4 h' + (j & 3).  acc[t] = Ps[src] + Pd[dst] (exact products, f32 sums).
// 4 DT MFMAs = slots 0 .. 4 DT - 1.
__device__ __forceinline__ bf16x8 p32_selector(int lane, int s) {
    const int m = lane & 31, hh = lane >> 5;
    bf16x8 a;
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = (__bf16)((m == 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3)) ? 1.0f : 0.0f);
    return a;
}
template <int DT, class Fill>
__device__ __forceinline__ void selp32(f32x16 (&acc)[DT], const bf16x8 (&ps)[2 * DT], const bf16x8 (&pd)[2 * DT], bf16x8 sel0,
                                       bf16x8 sel1, const Fill& fill) {
    static_for_each([&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value;
        f32x16 c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sel0, ps[2 * t], c, 0, 0, 0);
        fill.template run<4 * t>();
        __builtin_amdgcn_sched_barrier(0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sel1, ps[2 * t + 1], c, 0, 0, 0);
        fill.template run<4 * t + 1>();
        __builtin_amdgcn_sched_barrier(0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sel0, pd[2 * t], c, 0, 0, 0);
        fill.template run<4 * t + 2>();
        __builtin_amdgcn_sched_barrier(0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sel1, pd[2 * t + 1], c, 0, 0, 0);
        acc[t] = c;
        fill.template run<4 * t + 3>();
        __builtin_amdgcn_sched_barrier(0);
    }, std::make_integer_sequence<int, DT>{});
}

__device__ __forceinline__ float half_swap_sum(float s) {     // s[lane] + s[lane ^ 32] on the vector pipe
    float a = s, b = s;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}

// share [q * n / nq, (q + 1) * n / nq) of n slices for slot q of nq
constexpr int share_lo(int n, int nq, int q) { return (int)((long)q * n / nq); }
constexpr int share_hi(int n, int nq, int q) { return (int)((long)(q + 1) * n / nq); }
// the last slot before q that ran a slice of an n-slice job (-1: none)
constexpr int prev_share_slot(int n, int nq, int q) {
    for (int x = q - 1; x >= 0; --x)
        if (share_hi(n, nq, x) > share_lo(n, nq, x)) return x;
    return -1;
}

}  // namespace cgnn
