// cgnn_edge_stream_run: all message-passing rounds of the EDGE stream in one launch (reference graph_network.py:89-90,182
// for round = 0 .. L-1, optionally the edge encoder :57 in front), second generation.
//
// Why a second kernel.  edge_stream.hip (16 edges per wave, two waves per SIMD, v_mfma_f32_16x16x32_bf16) issues one
// 1-KiB LDS weight fragment per 16-cycle MFMA: at full matrix rate that is the CU's whole 256 B/clk LDS port, its two
// waves per SIMD run in lockstep (their MFMA and vector phases add up instead of overlapping) and all eight meet at a
// barrier per layer; it reaches 28 % of the bf16 MFMA peak.  Here
//   * one wave per SIMD (256-thread workgroups, the whole 512-register file per wave) owns TWO tiles of 32 edges;
//     v_mfma_f32_32x32x16_bf16: a 1-KiB fragment feeds a 32-cycle MFMA (half the LDS bytes per flop, half the MFMA
//     issue slots), and the two tiles take turns on every layer: while one tile's 32 MFMAs of a layer issue, the
//     other tile's vector work (bf16 pack + ReLU, LayerNorm + residual) fills the issue slots between them;
//   * a ring step (one layer's weights in LDS) serves 64 edges per wave = 256 per CU: half the LDS-DMA bytes and
//     half the barriers per edge; the one barrier per step sits BETWEEN the two tiles' blocks and confirms the NEXT
//     layer, so no wave starts a layer by waiting;
//   * every layer travels as one self-contained chunk (packed weights + its bias / LayerNorm vectors) of a contiguous
//     image built once per model (cgnn_edge_stream_image_build): the ring's source address is base + step * stride.
// Register file.  A wave owns 512 registers, but vector instructions address only the 256 architectural ones; the
// other 256 (accumulation registers) serve MFMA operands / results and memory instructions.  What only the matrix pipe
// and loads touch is therefore placed there by hand (inline-asm "a" operands): the LDS weight fragments, the P rows
// and -- parked between its two uses per round, LayerNorm's residual add and the stores -- the f32 latent of both
// tiles (128 registers).  Accumulators and bf16 operands, which the vector pipe packs and normalises, stay
// architectural, so no layer output is copied between the two halves.
// Numerics: bf16 operands, f32 accumulation, f32 LayerNorm and residual, as cgnn_edge_block (CGNN_BF16).
#include <string.h>

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

static size_t s32_stride(int latent) {
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

// ---- vector-memory bookkeeping -------------------------------------------------------------------------------------
// A wave's vector-memory operations retire in issue order, so "X has landed" is a counted s_waitcnt: at most as many
// operations outstanding as the wave has issued since X.  The counters below hold that number for the things the loop
// waits on (wave-uniform scalars).  Counting too FEW later operations only waits longer; counting too many would be
// wrong, so operations whose number is not certain (compiler-issued loads and stores) are simply not counted.
struct VmTrack {
    int a1, a2;    // since the LDS-DMA pieces of the chunk one / two steps ahead
    int pa;        // since the last P-row load
    __device__ __forceinline__ void op(int n = 1) {
        a1 += n;
        a2 += n;
        pa += n;
    }
    __device__ __forceinline__ void piece() {
        a1 += 1;
        a2 = 0;
        pa += 1;
    }
};
#define CGNN_S32_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
__device__ __forceinline__ void vm_wait_at_most(int n) {
    n = __builtin_amdgcn_readfirstlane(n);
    if (n >= 48) CGNN_S32_VMCNT(48);
    else if (n >= 32) CGNN_S32_VMCNT(32);
    else if (n >= 24) CGNN_S32_VMCNT(24);
    else if (n >= 16) CGNN_S32_VMCNT(16);
    else if (n >= 12) CGNN_S32_VMCNT(12);
    else if (n >= 8) CGNN_S32_VMCNT(8);
    else if (n >= 6) CGNN_S32_VMCNT(6);
    else if (n >= 4) CGNN_S32_VMCNT(4);
    else if (n >= 3) CGNN_S32_VMCNT(3);
    else if (n >= 2) CGNN_S32_VMCNT(2);
    else if (n >= 1) CGNN_S32_VMCNT(1);
    else CGNN_S32_VMCNT(0);
}

typedef __attribute__((address_space(3))) void* LdsVoidPtrG;
typedef const __attribute__((address_space(1))) void* GlobalVoidPtrG;

// ---- the ring -------------------------------------------------------------------------------------------------------
// Four slots.  Step s computes out of slot s % 4.  Between the two tiles' blocks of step s every wave waits for its
// own pieces of chunk s + 1 and meets the others at a barrier: chunk s + 1 is then complete for everybody, and
// everybody has left step s - 1, whose slot is refilled with chunk s + 3 (pieces handed out between the MFMA groups
// that follow).  A chunk is requested two and a half steps before its first use.
template <class G>
struct Ring32 {
    const char* image;
    VmTrack& vm;
    int count;           // chunks per tile pair
    int wave, lane;
    int slot;            // of the current step
    int dma_chunk, dma_slot, dma_left;
    __device__ __forceinline__ Ring32(const char* img, VmTrack& v, int cnt, int w, int l, int total_steps)
        : image(img), vm(v), count(cnt), wave(w), lane(l), slot(0), dma_chunk(0), dma_slot(0), dma_left(total_steps) {}
    __device__ __forceinline__ unsigned base() const {
        return (unsigned)(uintptr_t)(LdsWeightPtr)(cgnn_smem) + (unsigned)slot * G::STRIDE;
    }
    __device__ __forceinline__ LdsVecPtr vec() const { return (LdsVecPtr)(cgnn_smem + (unsigned)slot * G::STRIDE + G::VEC_OFF); }
    __device__ __forceinline__ void piece(int i) {
        if (dma_left > 0) {
            const unsigned off = (unsigned)(wave + CGNN_S32_WAVES * i) * 1024u;
            const char* src = image + (size_t)dma_chunk * G::STRIDE + off + lane * 16;
            char* dst = cgnn_smem + (unsigned)dma_slot * G::STRIDE + off;
            asm volatile("" ::: "memory");
            __builtin_amdgcn_global_load_lds((GlobalVoidPtrG)src, (LdsVoidPtrG)dst, 16, 0, 0);
            asm volatile("" ::: "memory");
            vm.piece();
        }
    }
    __device__ __forceinline__ void dma_done() {
        if (dma_left > 0) {
            --dma_left;
            dma_chunk = dma_chunk + 1 == count ? 0 : dma_chunk + 1;
            dma_slot = (dma_slot + 1) & (CGNN_S32_SLOTS - 1);
        }
    }
    __device__ __forceinline__ void prime() {
        for (int c = 0; c < CGNN_S32_SLOTS - 1; ++c) {
            for (int i = 0; i < G::NP; ++i) piece(i);
            dma_done();
        }
        CGNN_S32_VMCNT(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        vm.a1 = vm.a2 = 0;
    }
    __device__ __forceinline__ void sync_next() {
        vm_wait_at_most(vm.a1);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        vm.a1 = vm.a2;
        vm.a2 = 0;
    }
    __device__ __forceinline__ void advance() { slot = (slot + 1) & (CGNN_S32_SLOTS - 1); }
};

// ---- per-tile registers ---------------------------------------------------------------------------------------------
// 32 edges in the act layout of cgnn_common.hpp: edge on the MFMA column (lane & 31), for 32-feature tile t register i
// of lane (r, h) holds feature 32 t + 8 (i >> 2) + 4 h + (i & 3).
// a value parked in an accumulation register: only acc_put / acc_get touch it
__device__ __forceinline__ unsigned acc_put(float v) {
    unsigned a;
    asm("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v));
    return a;
}
__device__ __forceinline__ float acc_get(unsigned a) {
    float v;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
    return v;
}
template <int DT>
struct Tile32 {
    unsigned evp[DT][16]; // f32 latent (the residual stream), parked in accumulation registers
    f32x16 acc[DT];       // accumulators of the layer in flight, then its output
    bf16x8 in[2 * DT];    // the layer's input operand: k-step 2 t + s = features 32 t + 16 s .. + 15 (fragment k order)
    float part[DT];       // LayerNorm partial sums
    float keep[4];        // new latent values of an even affine slice, packed with the odd one that follows
    float mean, rstd;
};

template <int IMM>
__device__ __forceinline__ u32x4 lds_read_b128_acc(unsigned addr) {
    u32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(r) : "v"(addr), "n"(IMM));
    return r;
}
template <int IMM>
__device__ __forceinline__ void lds_wait4i(u32x4& a, u32x4& b, u32x4& c, u32x4& d) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+a"(a), "+a"(b), "+a"(c), "+a"(d) : "n"(IMM));
}
template <int IMM>
__device__ __forceinline__ void lds_wait1i(u32x4& a) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+a"(a) : "n"(IMM));
}
template <int IMM>
__device__ __forceinline__ void lds_wait2i(u32x4& a, u32x4& b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+a"(a), "+a"(b) : "n"(IMM));
}

struct NoFill32 {
    template <int G>
    __device__ __forceinline__ void run() const {}
};
template <class F>
struct FnFill32 {
    F f;
    template <int G>
    __device__ __forceinline__ void run() const {
        f(std::integral_constant<int, G>{});
    }
};
template <class F>
__device__ __forceinline__ FnFill32<F> make_fill(F f) {
    return FnFill32<F>{f};
}

// acc[o] += W[32 o .. 32 o + 31, :] . in ; fragment m = o * KS + ks (1 KiB, lane-linear) at addr + m * 1024.  The LDS
// reads and their counted waits are written by hand (hipcc waits lgkmcnt(0) before every group otherwise); PD groups
// of four fragments are in flight ahead of the MFMAs; `fill.run<g>()` is called after each group's MFMAs have been
// issued: the place for the OTHER tile's vector work and this wave's memory instructions, whose issue then overlaps
// the matrix pipe.  Rows are finished one after the other (o outermost).
template <int NROW, int KS, class Fill>
__device__ __forceinline__ void wblock32(f32x16 (&acc)[NROW], const bf16x8 (&in)[KS], unsigned addr, const Fill& fill) {
    constexpr int M = NROW * KS, GS = M < 4 ? M : 4, NG = M / GS, PD = CGNN_S32_PD, NBUF = PD + 1;
    static_assert(M % GS == 0 && (GS == 4 || GS == 2 || GS == 1), "groups of four (two, one) fragments");
    const unsigned a = addr + (unsigned)(threadIdx.x & 63) * 16u;
    u32x4 buf[NBUF][GS];
    static_for_each([&](auto pc) __attribute__((always_inline)) {
        constexpr int p = decltype(pc)::value;
        if constexpr (p < NG) {
            static_for_each([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                buf[p][j] = lds_read_b128_acc<(p * GS + j) * 1024>(a);
            }, std::make_integer_sequence<int, GS>{});
        }
    }, std::make_integer_sequence<int, PD>{});
    static_for_each([&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value;
        if constexpr (g + PD < NG) {
            static_for_each([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                buf[(g + PD) % NBUF][j] = lds_read_b128_acc<((g + PD) * GS + j) * 1024>(a);
            }, std::make_integer_sequence<int, GS>{});
        }
        constexpr int newer = ((g + PD < NG ? g + PD : NG - 1) - g) * GS;     // fragment reads issued after group g's
        if constexpr (GS == 4)
            lds_wait4i<newer>(buf[g % NBUF][0], buf[g % NBUF][1], buf[g % NBUF][2], buf[g % NBUF][3]);
        else if constexpr (GS == 2)
            lds_wait2i<newer>(buf[g % NBUF][0], buf[g % NBUF][1]);
        else
            lds_wait1i<newer>(buf[g % NBUF][0]);
        static_for_each([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value, m = g * GS + j, o = m / KS, ks = m % KS;
            acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, buf[g % NBUF][j]), in[ks], acc[o], 0, 0,
                                                             0);
        }, std::make_integer_sequence<int, GS>{});
        fill.template run<g>();
    }, std::make_integer_sequence<int, NG>{});
}
template <int NROW, int KS>
struct WBlockGroups {
    static constexpr int M = NROW * KS, GS = M < 4 ? M : 4, NG = M / GS;
};

// P rows (CGNN_P_BF16_S32: lane (r, h) owns the 16-byte pieces 2 t + s of its half of the row = the B operand of k-step
// (t, s)) enter the accumulators through the matrix pipe: A = a constant 0/1 selector that copies k = 8 h' + j of
// k-step s to row 16 s + 8 (j >> 2) + 4 h' + (j & 3).  acc[t] = Ps[src] + Pd[dst] (exact products, f32 sums).
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
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sel1, ps[2 * t + 1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sel0, pd[2 * t], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sel1, pd[2 * t + 1], c, 0, 0, 0);
        acc[t] = c;
        fill.template run<t>();
    }, std::make_integer_sequence<int, DT>{});
}

// P-row loads from inline asm (the compiler would guard registers loaded across the ring's LDS-DMA with vmcnt(0)).
// The caller counts them (VmTrack) and calls p32_ready() behind its own wait.
template <int IDX>
__device__ __forceinline__ bf16x8 p32_load(const __bf16* rowh) {     // rowh = table + row * H + h * (H / 2)
    u32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=a"(r) : "v"(rowh), "n"(IDX * 16));
    return __builtin_bit_cast(bf16x8, r);
}
template <int DT>
__device__ __forceinline__ void p32_ready(bf16x8 (&a)[2 * DT], bf16x8 (&b)[2 * DT]) {
    static_for_each([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        u32x4 x = __builtin_bit_cast(u32x4, a[i]), y = __builtin_bit_cast(u32x4, b[i]);
        asm volatile("" : "+a"(x), "+a"(y));
        a[i] = __builtin_bit_cast(bf16x8, x);
        b[i] = __builtin_bit_cast(bf16x8, y);
    }, std::make_integer_sequence<int, 2 * DT>{});
}

// ---- vector jobs, cut into slices that fill the other tile's MFMA gaps ---------------------------------------------
// bf16 pack (+ ReLU) of a finished layer: slice u = (t, s) writes in[2 t + s] from acc[t][8 s .. 8 s + 7].
template <bool RELU, int DT, int U>
__device__ __forceinline__ void pack32_slice(Tile32<DT>& X) {
    constexpr int t = U >> 1, s = U & 1;
    u32x4 v;
    v[0] = pack_bf16(X.acc[t][8 * s + 0], X.acc[t][8 * s + 1]);
    v[1] = pack_bf16(X.acc[t][8 * s + 2], X.acc[t][8 * s + 3]);
    v[2] = pack_bf16(X.acc[t][8 * s + 4], X.acc[t][8 * s + 5]);
    v[3] = pack_bf16(X.acc[t][8 * s + 6], X.acc[t][8 * s + 7]);
    const bf16x8 b = __builtin_bit_cast(bf16x8, v);
    X.in[2 * t + s] = RELU ? relu_bf16(b) : b;
}
template <int DT>
struct Pack32 {
    static constexpr int NS = 2 * DT;
};
template <bool RELU, int DT, int S0, int S1>
__device__ __forceinline__ void pack32_run(Tile32<DT>& X) {
    static_for_each([&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value + S0;
        if constexpr (u < S1) pack32_slice<RELU, DT, u>(X);
    }, std::make_integer_sequence<int, (S1 > S0 ? S1 - S0 : 0)>{});
}

__device__ __forceinline__ float half_swap_sum(float s) {     // s[lane] + s[lane ^ 32] on the vector pipe
    float a = s, b = s;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}

// LayerNorm (eps 1e-5, biased variance, affine) of X.acc over the 32 DT features of each edge, then
//   RES:  ev += y (graph_network.py:182)      !RES: ev = y (the encoder, :57)
// and the bf16 pack of the new ev as the next layer-0 operand.  Slices:
//   [0, DT)            partial sums per feature tile                 DT                 total, mean
//   [DT + 1, 2 DT + 1) centre, partial sums of squares               2 DT + 1           variance, rstd
//   [2 DT + 2, 6 DT + 2) affine (+ residual) for (t, g): reads and rewrites the parked latent; every second one also
//                        packs the eight new values of (t, s = g >> 1) as the next layer-0 operand
template <int DT>
struct Ln32 {
    static constexpr int NS = 6 * DT + 2;
    static constexpr int AFF0 = 2 * DT + 2, AFF1 = 6 * DT + 2;      // the affine slices
};
// gamma / beta of the affine slices come through two register pairs filled by hand-issued LDS reads, one slice ahead
// (left to the compiler, all 8 DT reads of a LayerNorm are hoisted to its top: 128 registers at latent 128, spills).
struct LnVec32 {
    u32x4 g[2], b[2];
};
template <int IMM>
__device__ __forceinline__ void ln32_vec_read(u32x4& g, u32x4& b, unsigned ga, unsigned ba) {
    asm volatile("ds_read_b128 %0, %2 offset:%4\n\tds_read_b128 %1, %3 offset:%4" : "=&v"(g), "=&v"(b) : "v"(ga), "v"(ba), "n"(IMM));
}
template <int NEWER>
__device__ __forceinline__ void ln32_vec_wait(u32x4& g, u32x4& b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(g), "+v"(b) : "n"(NEWER));
}
// NEWER: LDS operations this wave has issued since the PREVIOUS slice ran (the weight-fragment reads at the top of the
// MFMA groups in between): the affine slice waits for its own vectors and leaves those in flight.
template <bool RES, int DT, int U, int NEWER>
__device__ __forceinline__ void ln32_slice(Tile32<DT>& X, LnVec32& V, unsigned ga, unsigned ba) {
    constexpr int D = 32 * DT;
    typedef Ln32<DT> LN;
    if constexpr (U < DT) {
        const f32x16& v = X.acc[U];
        X.part[U] = (((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]))) +
                    (((v[8] + v[9]) + (v[10] + v[11])) + ((v[12] + v[13]) + (v[14] + v[15])));
    } else if constexpr (U == DT) {
        float s = X.part[0];
#pragma unroll
        for (int t = 1; t < DT; ++t) s += X.part[t];
        X.mean = half_swap_sum(s) * (1.0f / D);
    } else if constexpr (U < 2 * DT + 1) {
        constexpr int t = U - DT - 1;
        f32x16& v = X.acc[t];
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            v[i] -= X.mean;
            q = fmaf(v[i], v[i], q);
        }
        X.part[t] = q;
    } else if constexpr (U == 2 * DT + 1) {
        float q = X.part[0];
#pragma unroll
        for (int t = 1; t < DT; ++t) q += X.part[t];
        X.rstd = 1.0f / sqrtf(half_swap_sum(q) * (1.0f / D) + 1e-5f);
        ln32_vec_read<0>(V.g[0], V.b[0], ga, ba);                      // vectors of the first affine slice
    } else if constexpr (U < LN::AFF1) {
        constexpr int k = U - LN::AFF0, t = k >> 2, g = k & 3;
        if constexpr (k + 1 < 4 * DT) {
            constexpr int t1 = (k + 1) >> 2, g1 = (k + 1) & 3;
            ln32_vec_read<(32 * t1 + 8 * g1) * 4>(V.g[(k + 1) & 1], V.b[(k + 1) & 1], ga, ba);
            ln32_vec_wait<NEWER + 2>(V.g[k & 1], V.b[k & 1]);
        } else {
            ln32_vec_wait<NEWER>(V.g[k & 1], V.b[k & 1]);
        }
        const f32x4 gm = __builtin_bit_cast(f32x4, V.g[k & 1]);
        const f32x4 bt = __builtin_bit_cast(f32x4, V.b[k & 1]);
        float e[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float w = gm[c] * X.rstd;
            if (RES)
                e[c] = fmaf(X.acc[t][4 * g + c], w, acc_get(X.evp[t][4 * g + c]) + bt[c]);
            else
                e[c] = fmaf(X.acc[t][4 * g + c], w, bt[c]);
            X.evp[t][4 * g + c] = acc_put(e[c]);
        }
        if constexpr ((g & 1) == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) X.keep[c] = e[c];
        } else {
            u32x4 v;
            v[0] = pack_bf16(X.keep[0], X.keep[1]);
            v[1] = pack_bf16(X.keep[2], X.keep[3]);
            v[2] = pack_bf16(e[0], e[1]);
            v[3] = pack_bf16(e[2], e[3]);
            X.in[2 * t + (g >> 1)] = __builtin_bit_cast(bf16x8, v);
        }
    }
}
// slices [S0, S1); NEWER0 applies to the first of them (see ln32_slice), the others follow it directly
template <bool RES, int DT, int S0, int S1, int NEWER0>
__device__ __forceinline__ void ln32_run(Tile32<DT>& X, LnVec32& V, unsigned ga, unsigned ba) {
    static_for_each([&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value + S0;
        if constexpr (u < S1) ln32_slice<RES, DT, u, (u == S0 ? NEWER0 : 0)>(X, V, ga, ba);
    }, std::make_integer_sequence<int, (S1 > S0 ? S1 - S0 : 0)>{});
}

template <int DT>
__device__ __forceinline__ void bias_fill32(f32x16 (&acc)[DT], LdsVecPtr b, int h) {
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = *(LdsVec4Ptr)(b + 32 * t + 8 * g + 4 * h);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[t][4 * g + c] = v[c];
        }
}

// share [g * n / ng, (g + 1) * n / ng) of n slices for group g of ng
constexpr int share_lo(int n, int ng, int g) { return g * n / ng; }
constexpr int share_hi(int n, int ng, int g) { return (g + 1) * n / ng; }
// weight-fragment reads wblock32 issues at the top of groups (g0, g1] (group x requests group x + PD)
constexpr int frags_between(int ng, int gs, int g0, int g1) {
    int n = 0;
    for (int x = g0 + 1; x <= g1; ++x)
        if (x + CGNN_S32_PD < ng) n += gs;
    return n;
}
// the last group before g that ran a slice of an n-slice job (-1: none)
constexpr int prev_share_group(int n, int ng, int g) {
    for (int x = g - 1; x >= 0; --x)
        if (share_hi(n, ng, x) > share_lo(n, ng, x)) return x;
    return -1;
}

// ---- the kernel -----------------------------------------------------------------------------------------------------
template <int DT, bool ENC>
__global__ __launch_bounds__(CGNN_S32_BLOCK, 1) void edge_stream32_kernel(
    S32Args a, const __bf16* __restrict__ ps_all, const __bf16* __restrict__ pd_all, int64_t round_stride,
    const int32_t* __restrict__ src, const int32_t* __restrict__ dst, int64_t num_edges, const float* e_in, float* e_out,
    const float* __restrict__ attr, int ld_attr) {
    typedef S32Geom<DT> G;
    constexpr int D = G::D, KS = G::KS;
    typedef WBlockGroups<DT, KS> WG;
    constexpr int NG = WG::NG;                 // MFMA groups of a full layer block
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int L = a.rounds, nh = a.nh;
    const int enc_steps = ENC ? nh + 1 : 0;
    const int steps_per_pair = enc_steps + L * (nh + 1);
    const int64_t tiles = (num_edges + 31) / 32;
    const int64_t pairs = (tiles + 1) / 2;
    const TileRange tr = tile_range(pairs);
    // the four waves share the ring's barriers: all run the iteration count of wave 0; a wave whose last pair falls
    // off the range recomputes its previous pair and skips the stores
    const int64_t first0 = tr.first - wave;
    const int iters = __builtin_amdgcn_readfirstlane(
        first0 < tr.end ? (int)((tr.end - first0 + tr.stride - 1) / tr.stride) : 0);
    if (iters == 0) return;
    VmTrack vm = {0, 0, 0};
    Ring32<G> ring(a.image, vm, steps_per_pair, wave, lane, iters * steps_per_pair);
    ring.prime();
    const bf16x8 sel0 = p32_selector(lane, 0), sel1 = p32_selector(lane, 1);

    Tile32<DT> A, B;
    bf16x8 ps[2 * DT], pd[2 * DT];
    int64_t pair = tr.first < tr.end ? tr.first : tr.end - 1;
    bool valid = tr.first < tr.end;

    auto p_issue = [&](const __bf16* tps, const __bf16* tpd, int32_t s, int32_t d, auto ic) __attribute__((always_inline)) {
        // load i (of 4 DT): table (i & 1), piece i >> 1
        constexpr int i = decltype(ic)::value, pc = i >> 1;
        if constexpr ((i & 1) == 0)
            ps[pc] = p32_load<pc>(tps + (int64_t)s * D + h * (D / 2));
        else
            pd[pc] = p32_load<pc>(tpd + (int64_t)d * D + h * (D / 2));
        vm.op();
        if constexpr (i == 4 * DT - 1) vm.pa = 0;
    };

    for (int it = 0; it < iters; ++it) {
        // ---- this pair's tiles, edges and inputs (compiler-tracked loads, uncounted: once per pair) ----------------
        const int64_t tA = 2 * pair, tB = (2 * pair + 1 < tiles) ? 2 * pair + 1 : 2 * pair;
        const bool validB = valid && (2 * pair + 1 < tiles);
        int32_t sA, dA, sB, dB;
        {
            const int64_t eA = tA * 32 + r, eB = tB * 32 + r;
            const int64_t ca = eA < num_edges ? eA : num_edges - 1, cb = eB < num_edges ? eB : num_edges - 1;
            sA = src[ca];
            dA = dst[ca];
            sB = src[cb];
            dB = dst[cb];
            if (ENC) {
                // lane (r, h), k-step 0, element j = edge feature 8 (j >> 2) + 4 h + (j & 3)
                const int fin = a.enc_in_dim;
                auto load_attr = [&](Tile32<DT>& X, int64_t e) __attribute__((always_inline)) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int f = 8 * (j >> 2) + 4 * h + (j & 3);
                        v[j] = f < fin ? attr[e * ld_attr + f] : 0.f;
                    }
                    u32x4 w;
                    w[0] = pack_bf16(v[0], v[1]);
                    w[1] = pack_bf16(v[2], v[3]);
                    w[2] = pack_bf16(v[4], v[5]);
                    w[3] = pack_bf16(v[6], v[7]);
                    X.in[0] = __builtin_bit_cast(bf16x8, w);
                    const u32x4 z = {0u, 0u, 0u, 0u};
                    X.in[1] = __builtin_bit_cast(bf16x8, z);
                };
                load_attr(A, ca);
                load_attr(B, cb);
            } else {
                auto load_latent = [&](Tile32<DT>& X, int64_t T) __attribute__((always_inline)) {
                    f32x16 v[DT];
                    load_tile<DT>(v, e_in + T * (32 * D), lane);
#pragma unroll
                    for (int t = 0; t < DT; ++t) {
#pragma unroll
                        for (int s = 0; s < 2; ++s) {
                            u32x4 w;
#pragma unroll
                            for (int x = 0; x < 4; ++x) w[x] = pack_bf16(v[t][8 * s + 2 * x], v[t][8 * s + 2 * x + 1]);
                            X.in[2 * t + s] = __builtin_bit_cast(bf16x8, w);
                        }
#pragma unroll
                        for (int i = 0; i < 16; ++i) X.evp[t][i] = acc_put(v[t][i]);
                    }
                };
                load_latent(A, tA);
                load_latent(B, tB);
            }
        }

        // ---- one MLP + LayerNorm pass over both tiles: nh + 1 ring steps ------------------------------------------
        // IS_ENC: the edge encoder (layer 0 = Linear of the edge features with bias, no P rows, no residual);
        // otherwise round rr (layer 0 = Ps[src] + Pd[dst] + We e).
        // `pa_pending`: the P rows of tile A for this round were already requested (by the previous round)
        auto pass = [&](auto enc_tag, int rr, bool pa_pending) __attribute__((always_inline)) {
            constexpr bool IS_ENC = decltype(enc_tag)::value;
            // the encoder's first Linear is packed with K padded to one 32-wide k tile = two k-steps, the second all zero
            constexpr int KS0 = IS_ENC ? 2 : KS;
            typedef WBlockGroups<DT, KS0> WG0;
            constexpr int NG0 = WG0::NG;
            const __bf16* tps = ps_all + (int64_t)rr * round_stride;
            const __bf16* tpd = pd_all + (int64_t)rr * round_stride;
            // ---------------- layer 0 ----------------
            {
                const unsigned base = ring.base();
                if constexpr (IS_ENC) {
                    bias_fill32<DT>(A.acc, ring.vec(), h);
                    const bf16x8 (&inA)[2] = reinterpret_cast<const bf16x8(&)[2]>(A.in[0]);
                    wblock32<DT, 2>(A.acc, inA, base, NoFill32{});
                } else {
                    if (!pa_pending)
                        static_for_each([&](auto ic) __attribute__((always_inline)) { p_issue(tps, tpd, sA, dA, ic); },
                                        std::make_integer_sequence<int, 4 * DT>{});
                    vm_wait_at_most(vm.pa);
                    p32_ready<DT>(ps, pd);
                    selp32<DT>(A.acc, ps, pd, sel0, sel1, NoFill32{});
                    static_for_each([&](auto ic) __attribute__((always_inline)) { p_issue(tps, tpd, sB, dB, ic); },
                                    std::make_integer_sequence<int, 4 * DT>{});
                    wblock32<DT, KS>(A.acc, A.in, base, NoFill32{});
                }
                ring.sync_next();
                if constexpr (IS_ENC) {
                    bias_fill32<DT>(B.acc, ring.vec(), h);
                    const bf16x8 (&inB)[2] = reinterpret_cast<const bf16x8(&)[2]>(B.in[0]);
                    wblock32<DT, 2>(B.acc, inB, base, make_fill([&](auto gc) __attribute__((always_inline)) {
                        constexpr int g = decltype(gc)::value;
                        pack32_run<true, DT, share_lo(2 * DT, NG0, g), share_hi(2 * DT, NG0, g)>(A);
                        for (int i = share_lo(G::NP, NG0, g); i < share_hi(G::NP, NG0, g); ++i) ring.piece(i);
                    }));
                } else {
                    vm_wait_at_most(vm.pa);
                    p32_ready<DT>(ps, pd);
                    selp32<DT>(B.acc, ps, pd, sel0, sel1, NoFill32{});
                    wblock32<DT, KS>(B.acc, B.in, base, make_fill([&](auto gc) __attribute__((always_inline)) {
                        constexpr int g = decltype(gc)::value;
                        pack32_run<true, DT, share_lo(2 * DT, NG0, g), share_hi(2 * DT, NG0, g)>(A);
                        for (int i = share_lo(G::NP, NG0, g); i < share_hi(G::NP, NG0, g); ++i) ring.piece(i);
                    }));
                }
                ring.dma_done();
                ring.advance();
            }
            // A.in = ReLU(layer 0 of A); B.acc = layer 0 of B, still to be packed
            // ---------------- hidden layers 1 .. nh - 1 ----------------
            for (int l = 1; l < nh; ++l) {
                const unsigned base = ring.base();
                bias_fill32<DT>(A.acc, ring.vec(), h);
                // B's pack reads B.acc and must be finished before B's block overwrites it: it fills A's block
                wblock32<DT, KS>(A.acc, A.in, base, make_fill([&](auto gc) __attribute__((always_inline)) {
                    constexpr int g = decltype(gc)::value;
                    pack32_run<true, DT, share_lo(2 * DT, NG, g), share_hi(2 * DT, NG, g)>(B);
                }));
                ring.sync_next();
                bias_fill32<DT>(B.acc, ring.vec(), h);
                wblock32<DT, KS>(B.acc, B.in, base, make_fill([&](auto gc) __attribute__((always_inline)) {
                    constexpr int g = decltype(gc)::value;
                    pack32_run<true, DT, share_lo(2 * DT, NG, g), share_hi(2 * DT, NG, g)>(A);
                    for (int i = share_lo(G::NP, NG, g); i < share_hi(G::NP, NG, g); ++i) ring.piece(i);
                }));
                ring.dma_done();
                ring.advance();
            }
            // ---------------- output layer + LayerNorm (+ residual) ----------------
            {
                const unsigned base = ring.base();
                const LdsVecPtr vec = ring.vec();
                const unsigned ga = base + G::VEC_OFF + (unsigned)D * 4u + 16u * (unsigned)h, ba = ga + (unsigned)D * 4u;
                LnVec32 V;
                bias_fill32<DT>(A.acc, vec, h);
                wblock32<DT, KS>(A.acc, A.in, base, make_fill([&](auto gc) __attribute__((always_inline)) {
                    constexpr int g = decltype(gc)::value;
                    pack32_run<true, DT, share_lo(2 * DT, NG, g), share_hi(2 * DT, NG, g)>(B);
                }));
                ring.sync_next();
                bias_fill32<DT>(B.acc, vec, h);
                constexpr int NLN = Ln32<DT>::NS;
                constexpr int COVER = (NLN * 2) / 5;      // slices of A's LayerNorm placed under B's MFMAs
                // the next round's P rows of tile A (the P registers are free since B's layer 0)
                const bool next_p = IS_ENC || rr + 1 < L;
                const __bf16* nps = IS_ENC ? ps_all : tps + round_stride;
                const __bf16* npd = IS_ENC ? pd_all : tpd + round_stride;
                wblock32<DT, KS>(B.acc, B.in, base, make_fill([&](auto gc) __attribute__((always_inline)) {
                    constexpr int g = decltype(gc)::value;
                    constexpr int gp = prev_share_group(COVER, NG, g);
                    ln32_run<!IS_ENC, DT, share_lo(COVER, NG, g), share_hi(COVER, NG, g),
                             frags_between(NG, WG::GS, gp < 0 ? g : gp, g)>(A, V, ga, ba);
                    for (int i = share_lo(G::NP, NG, g); i < share_hi(G::NP, NG, g); ++i) ring.piece(i);
                    if (next_p)
                        static_for_each([&](auto kc) __attribute__((always_inline)) {
                            constexpr int i = decltype(kc)::value + share_lo(4 * DT, NG, g);
                            if constexpr (i < share_hi(4 * DT, NG, g))
                                p_issue(nps, npd, sA, dA, std::integral_constant<int, i>{});
                        }, std::make_integer_sequence<int, (4 * DT + NG - 1) / NG + 1>{});
                }));
                ring.dma_done();
                ln32_run<!IS_ENC, DT, COVER, NLN, 0>(A, V, ga, ba);
                ln32_run<!IS_ENC, DT, 0, NLN, 0>(B, V, ga, ba);
                ring.advance();
            }
        };

        if (ENC) pass(std::true_type{}, 0, false);
        for (int rr = 0; rr < L; ++rr) pass(std::false_type{}, rr, ENC || rr > 0);

        auto store_latent = [&](const Tile32<DT>& X, int64_t T) __attribute__((always_inline)) {
            f32x16 v[DT];
#pragma unroll
            for (int t = 0; t < DT; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) v[t][i] = acc_get(X.evp[t][i]);
            store_tile<DT>(v, e_out + T * (32 * D), lane);
        };
        if (valid) store_latent(A, tA);
        if (validB) store_latent(B, tB);
        const int64_t pn = pair + tr.stride;
        if (pn < tr.end) {
            pair = pn;
        } else {
            valid = false;
        }
    }
}

template <int DT>
static int launch_stream32(const S32Args& a, const __bf16* ps, const __bf16* pd, int64_t round_stride, const int32_t* src,
                           const int32_t* dst, int64_t num_edges, const float* e_in, float* e_out, const float* attr,
                           int ld_attr, hipStream_t st) {
    typedef S32Geom<DT> G;
    const bool enc = a.enc_in_dim > 0;
    auto kern = enc ? edge_stream32_kernel<DT, true> : edge_stream32_kernel<DT, false>;
    if (G::LDS > 48 * 1024) {
        int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)G::LDS),
                           "hipFuncSetAttribute(edge_stream32)");
        if (rc != CGNN_OK) return rc;
    }
    const int64_t pairs = ((num_edges + 31) / 32 + 1) / 2;
    const int grid = grid_for_tiles(pairs, 1, CGNN_S32_WAVES);
    kern<<<grid, CGNN_S32_BLOCK, G::LDS, st>>>(a, ps, pd, round_stride, src, dst, num_edges, e_in, e_out, attr, ld_attr);
    return check_hip(hipGetLastError(), "cgnn_edge_stream_run launch");
}

}  // namespace cgnn

using namespace cgnn;

extern "C" size_t cgnn_edge_stream_image_bytes(int32_t latent, int32_t num_hidden_layers, int32_t num_rounds,
                                               int32_t with_encoder) {
    const size_t stride = s32_stride(latent);
    if (!stride || num_hidden_layers < 1 || num_rounds < 1) return 0;
    return stride * (size_t)(num_hidden_layers + 1) * (size_t)(num_rounds + (with_encoder ? 1 : 0));
}

// One chunk per layer: [packed weights | bias | gamma | beta | zero padding].  Device-to-device copies on `stream`.
static int s32_put_layer(char* chunk, size_t stride, int latent, const MlpDev& m, int l, bool with_bias, hipStream_t st) {
    const size_t wbytes = (size_t)latent * latent * 2;
    if (m.bytes[l] > wbytes) {
        set_error("cgnn_edge_stream_image_build: layer %d has %u packed bytes, a chunk holds %zu", l, m.bytes[l], wbytes);
        return CGNN_ERR_UNSUPPORTED;
    }
    int rc = check_hip(hipMemsetAsync(chunk, 0, stride, st), "hipMemsetAsync(image chunk)");
    if (rc != CGNN_OK) return rc;
    rc = check_hip(hipMemcpyAsync(chunk, m.w[l], m.bytes[l], hipMemcpyDeviceToDevice, st), "hipMemcpyAsync(weights)");
    if (rc != CGNN_OK) return rc;
    char* vec = chunk + wbytes;
    if (with_bias && m.b[l]) {
        rc = check_hip(hipMemcpyAsync(vec, m.b[l], (size_t)m.out_dim[l] * 4, hipMemcpyDeviceToDevice, st), "hipMemcpyAsync(bias)");
        if (rc != CGNN_OK) return rc;
    }
    if (l == m.nh) {
        rc = check_hip(hipMemcpyAsync(vec + (size_t)latent * 4, m.gamma, (size_t)latent * 4, hipMemcpyDeviceToDevice, st),
                       "hipMemcpyAsync(gamma)");
        if (rc != CGNN_OK) return rc;
        rc = check_hip(hipMemcpyAsync(vec + (size_t)latent * 8, m.beta, (size_t)latent * 4, hipMemcpyDeviceToDevice, st),
                       "hipMemcpyAsync(beta)");
        if (rc != CGNN_OK) return rc;
    }
    return CGNN_OK;
}

extern "C" int cgnn_edge_stream_image_build(const cgnn_mlp* rounds, int32_t num_rounds, const cgnn_mlp* encoder,
                                            int32_t latent, void* image, size_t image_bytes, void* stream) {
    if (!rounds || num_rounds < 1 || !image) {
        set_error("cgnn_edge_stream_image_build: invalid argument");
        return CGNN_ERR_INVALID_ARG;
    }
    const size_t stride = s32_stride(latent);
    if (!stride) {
        set_error("cgnn_edge_stream_image_build: latent %d not in {32, 64, 128}", latent);
        return CGNN_ERR_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    int nh = 0;
    size_t at = 0;
    for (int r = (encoder ? -1 : 0); r < num_rounds; ++r) {
        const cgnn_mlp* mm = r < 0 ? encoder : &rounds[r];
        MlpDev m;
        int rc = make_mlp_dev(mm, &m, nullptr, "cgnn_edge_stream_image_build");
        if (rc != CGNN_OK) return rc;
        if (r <= 0 && nh == 0) nh = m.nh;
        const int in0 = r < 0 ? m.in_dim[0] : latent;
        if (mm->precision != CGNN_BF16 || !m.gamma || m.nh != nh || m.out_dim[nh] != latent || (r < 0 && in0 > 16) ||
            (r >= 0 && m.in_dim[0] != latent)) {
            set_error("cgnn_edge_stream_image_build: %s must be CGNN_BF16 with LayerNorm, %d hidden layers, latent %d%s",
                      r < 0 ? "the encoder" : "every round", nh, latent, r < 0 ? " and at most 16 inputs" : "");
            return CGNN_ERR_UNSUPPORTED;
        }
        for (int l = 0; l <= nh; ++l) {
            if (m.out_dim[l] != latent || (l > 0 && m.in_dim[l] != latent)) {
                set_error("cgnn_edge_stream_image_build: hidden size must equal the latent size (%d)", latent);
                return CGNN_ERR_UNSUPPORTED;
            }
            if (at + stride > image_bytes) {
                set_error("cgnn_edge_stream_image_build: image too small");
                return CGNN_ERR_WORKSPACE;
            }
            // a round's layer-0 bias lives in its Pd table (cgnn_project_nodes); the encoder's is applied here
            rc = s32_put_layer((char*)image + at, stride, latent, m, l, r < 0 || l > 0, st);
            if (rc != CGNN_OK) return rc;
            at += stride;
        }
    }
    return CGNN_OK;
}

extern "C" int cgnn_edge_stream_run(const void* image, size_t image_bytes, int32_t latent, int32_t num_hidden_layers,
                                    int32_t num_rounds, int32_t enc_in_dim, const void* ps_all, const void* pd_all,
                                    int64_t round_stride, const int32_t* src, const int32_t* dst, int64_t num_edges,
                                    const float* e_in, float* e_out, const float* edge_attr, int32_t ld_attr, void* stream) {
    if (!image || !ps_all || !pd_all || !src || !dst || !e_out || num_edges < 0 || num_rounds < 1 || num_hidden_layers < 1 ||
        round_stride < 0 || (enc_in_dim > 0 ? (!edge_attr || ld_attr < enc_in_dim) : !e_in)) {
        set_error("cgnn_edge_stream_run: invalid argument");
        return CGNN_ERR_INVALID_ARG;
    }
    if (enc_in_dim > 16) {
        set_error("cgnn_edge_stream_run: the in-launch encoder takes at most 16 edge features (got %d)", enc_in_dim);
        return CGNN_ERR_UNSUPPORTED;
    }
    const size_t need = cgnn_edge_stream_image_bytes(latent, num_hidden_layers, num_rounds, enc_in_dim > 0);
    if (!need) {
        set_error("cgnn_edge_stream_run: no kernel for latent=%d (built for hidden == latent in {32,64,128})", latent);
        return CGNN_ERR_UNSUPPORTED;
    }
    if (image_bytes < need) {
        set_error("cgnn_edge_stream_run: image has %zu bytes, this model needs %zu", image_bytes, need);
        return CGNN_ERR_INVALID_ARG;
    }
    if (num_edges == 0) return CGNN_OK;
    S32Args a;
    a.image = (const char*)image;
    a.rounds = num_rounds;
    a.nh = num_hidden_layers;
    a.enc_in_dim = enc_in_dim > 0 ? enc_in_dim : 0;
    hipStream_t st = (hipStream_t)stream;
    switch (latent) {
        case 32:
            return launch_stream32<1>(a, (const __bf16*)ps_all, (const __bf16*)pd_all, round_stride, src, dst, num_edges, e_in,
                                      e_out, edge_attr, ld_attr, st);
        case 64:
            return launch_stream32<2>(a, (const __bf16*)ps_all, (const __bf16*)pd_all, round_stride, src, dst, num_edges, e_in,
                                      e_out, edge_attr, ld_attr, st);
        default:
            return launch_stream32<4>(a, (const __bf16*)ps_all, (const __bf16*)pd_all, round_stride, src, dst, num_edges, e_in,
                                      e_out, edge_attr, ld_attr, st);
    }
}
