// cgnn_edge_stream_run_w8: all message-passing rounds of the EDGE stream in one launch (reference graph_network.py:89-90,
// :182 for round = 0 .. L-1, optionally the edge encoder :57 in front), third generation: two waves per SIMD.
//
// Why a third kernel.  edge_stream32.hip runs ONE wave per SIMD with two 32-edge tiles and a hand-written interleave of
// one tile's vector work into the other tile's MFMA gaps.  Its counters (round 2) say the matrix time and the issue
// time of everything else ADD on that single wave (7.2 k + 8.4 k cycles per pass against 15.9 k measured): a lone wave
// issues at most one vector instruction per four cycles, every dependent pair and every hazard nop is exposed, and the
// f32 latents of both tiles have to be parked in accumulation registers (256 moves per pass).  Here
//   * 512-thread workgroups: TWO waves per SIMD, ONE 32-edge tile each, the same v_mfma_f32_32x32x16_bf16 economy (a
//     1-KiB LDS fragment feeds a 32-cycle MFMA, the ring step serves 256 edges per CU).  The hardware interleaves the
//     two waves: a wave that only issues vector instructions runs beside a wave that only issues MFMAs at full matrix
//     rate and 5.2 cycles per vector instruction (scripts/dev/probe_mfma_valu.hip);
//   * 256 registers per wave, all architectural: f32 latent 64, accumulators 64, bf16 operand 32 (+ the next operand,
//     packed row tile by row tile as accumulators die), weight fragments, P rows.  Nothing is parked; every memory
//     operation except the fragment reads is visible to the compiler (no hand-counted register loads);
//   * LAG 1: waves 4-7 (the second wave of every SIMD) run one ring step behind waves 0-3, so that one wave's
//     LayerNorm (vector pipe only) meets its partner's matrix work instead of its partner's LayerNorm;
//   * P rows through LDS.  A lane's half of a sender row is one 128-byte line; gathering it as eight 16-byte pieces costs
//     eight load instructions that each touch 64 lines (the texture addresser takes a line per cycle: 4 k cycles per
//     pass and CU, and the issuing waves queue behind it).  Here four lanes fetch 64 contiguous bytes of a line by LDS-DMA
//     (16 lines per instruction) into a 4-KiB staging area, XOR-swizzled so that the B-operand pieces come back with
//     conflict-free ds_read_b128; the receiver rows of a tile (fixed in-degree k, receiver-sorted: at most four) are
//     fetched by ONE load and broadcast through LDS.  The kernel therefore takes fixed-k graphs (8 <= k, k | 32 or 32 | k:
//     what data_utils.preprocess emits); other edge lists run cgnn_edge_stream_run.
// Same image as cgnn_edge_stream_run (cgnn_edge_stream_image_build), same P tables (CGNN_P_BF16_S32), same numerics
// (bf16 operands, f32 accumulation, f32 LayerNorm and residual; LayerNorm's variance as E[x^2] - mean^2).
#include <string.h>

#include <type_traits>

#include "s32.hpp"

namespace cgnn {

#define CGNN_W8_WAVES 8
#define CGNN_W8_BLOCK (CGNN_W8_WAVES * 64)
#define CGNN_W8_SLOTS 3
#define CGNN_IC(x) std::integral_constant<int, (x)> {}
#include "w8_dev.hpp"      // the schedule constants at their shipped values; cycle stamps and timing-only ablations of developer builds

template <int DT>
struct W8Geom {
    typedef S32Geom<DT> G;
    static constexpr int D = G::D, KS = G::KS;
    static constexpr unsigned STRIDE = G::STRIDE, VEC_OFF = G::VEC_OFF;
    static constexpr int PIECES = (int)(STRIDE / 1024u);                           // 1-KiB LDS-DMA instructions per chunk
    static constexpr int NP = (PIECES + CGNN_W8_WAVES - 1) / CGNN_W8_WAVES;        // ... per wave
    static constexpr unsigned LNBUF_OFF = CGNN_W8_SLOTS * STRIDE;                  // two (gamma | beta) buffers behind the ring
    static constexpr unsigned LNBUF_BYTES = 2u * D * 4u;
    static constexpr unsigned PDST_OFF = LNBUF_OFF + 2u * LNBUF_BYTES;           // per wave: the tile's receiver P rows (<= 4 rows)
    static constexpr unsigned PDST_ROW = 256u + 16u;                             // a receiver's row, padded by one 16-byte bank group:
                                                                                 // the (<= 4) rows a read touches start in different ones
    static constexpr unsigned PDST_BYTES = 4u * PDST_ROW;
    static constexpr unsigned PSST_OFF = PDST_OFF + CGNN_W8_WAVES * PDST_BYTES;  // per wave: half of every sender half-row of the tile
    static constexpr unsigned PSST_BYTES = 64u * (unsigned)D / 2u;               // 64 lines x D / 2 bytes (latent 128: 4 KiB)
    static constexpr unsigned LDS = PSST_OFF + CGNN_W8_WAVES * PSST_BYTES;
    static_assert(LDS <= 160u * 1024u, "LDS budget");
};

// ---- the ring -------------------------------------------------------------------------------------------------------
// Three slots, one barrier per ring step ("interval").  In interval g waves 0-3 compute chunk g out of slot g % 3 and
// waves 4-7 chunk g - LAG.
//   LAG 0: every wave issues its pieces of chunk g + 2 into the slot everybody left at the last barrier, and waits --
//          counted -- for its own pieces of chunk g + 1 (issued an interval ago) before the barrier;
//   LAG 1: slots g - 1 and g are being read, so the pieces of chunk g + 1 go into the third one in the FIRST MFMA slots
//          of the interval and are awaited at its end (vmcnt(0): an interval is 4-5 k cycles, an L2-warm piece lands in
//          a few hundred).
// Every wave issues NP pieces per chunk; where 8 NP exceeds the chunk's pieces the surplus ones repeat an earlier
// piece (same bytes to the same place), which keeps every wave's operation count, and so every wait count, the same.
// (Measured and dropped: skipping the surplus pieces with a wave-uniform branch -- control flow inside the MFMA blocks
// cost 173 spilled registers and a 2.7 x slower kernel.)
template <class W, int LAG>
struct RingW {
    const char* image;
    int count;           // chunks per tile
    int wave, lane;
    int slot;            // of this wave's current step
    int dma_chunk, dma_slot;
    CGNN_W8_STAMP_MEMBER
    __device__ __forceinline__ RingW(const char* img, int cnt, int w, int l)
        : image(img), count(cnt), wave(w), lane(l), slot(0), dma_chunk(0), dma_slot(0) {}
    __device__ __forceinline__ unsigned lds0() const { return (unsigned)(uintptr_t)(LdsWeightPtr)(cgnn_smem); }
    __device__ __forceinline__ unsigned base() const { return lds0() + (unsigned)slot * W::STRIDE; }
    __device__ __forceinline__ unsigned base_next() const {
        return lds0() + (unsigned)(slot + 1 == CGNN_W8_SLOTS ? 0 : slot + 1) * W::STRIDE;
    }
    __device__ __forceinline__ unsigned vec_addr() const { return base() + W::VEC_OFF; }
    __device__ __forceinline__ unsigned lnbuf(int which) const { return lds0() + W::LNBUF_OFF + (unsigned)which * W::LNBUF_BYTES; }
    __device__ __forceinline__ void piece(int i) {
        asm volatile("" ::: "memory");
#if defined(__HIP_DEVICE_COMPILE__)      // (the host pass of hipcc does not know the buffer builtins)
        int idx = wave + CGNN_W8_WAVES * i;
        if (idx >= W::PIECES) idx -= CGNN_W8_WAVES;
        const unsigned off = (unsigned)idx * 1024u;
        char* dst = cgnn_smem + (unsigned)dma_slot * W::STRIDE + off;
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(image), 0, (int)((unsigned)count * W::STRIDE), 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LdsVoidPtrG)dst, 16, (unsigned)lane * 16u,
                                                 (unsigned)dma_chunk * W::STRIDE + off, 0, 0);
#endif
        asm volatile("" ::: "memory");
    }
    __device__ __forceinline__ void dma_done() {
        dma_chunk = dma_chunk + 1 == count ? 0 : dma_chunk + 1;
        dma_slot = dma_slot + 1 == CGNN_W8_SLOTS ? 0 : dma_slot + 1;
    }
    __device__ __forceinline__ void prime() {
        for (int c = 0; c < CGNN_W8_SLOTS - 1 - LAG; ++c) {
            for (int i = 0; i < W::NP; ++i) piece(i);
            dma_done();
        }
        CGNN_S32_VMCNT(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    // End of an interval.  EXTRA (LAG 0) = vector-memory operations this wave has issued in this interval BEHIND its ring
    // pieces (they are newer than the pieces waited for; a smaller count than the true one only waits longer).
    // K0: first of this call's three timer slots (developer builds).
    template <int EXTRA, int K0 = 17>
    __device__ __forceinline__ void interval_end() {
        CGNN_W8_STAMP(K0);
        if constexpr (w8dev::SLEEP > 0) __builtin_amdgcn_s_sleep(w8dev::SLEEP);
        vm_wait_const<(LAG ? 0 : (CGNN_W8_CARRY ? EXTRA : W::NP + EXTRA))>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        CGNN_W8_STAMP(K0 + 1);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        CGNN_W8_STAMP(K0 + 2);
    }
    __device__ __forceinline__ void advance() { slot = slot + 1 == CGNN_W8_SLOTS ? 0 : slot + 1; }
    // an interval in which this wave computes nothing (the lagging waves' first, the leading waves' last)
    __device__ __forceinline__ void idle_interval() {
        for (int i = 0; i < W::NP; ++i) piece(i);
        dma_done();
        interval_end<0>();
    }
};

template <int N>
__device__ __forceinline__ void lds_wait2(u32x4& a, u32x4& b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}

// acc[o] += W[32 o .. 32 o + 31, :] . in ; fragment m = o * KS + ks (1 KiB, lane-linear) at addr + m * 1024, rows one
// after the other.  Fragment reads and their counted waits by hand (hipcc waits lgkmcnt(0) before every group
// otherwise); PD groups of GS fragments in flight ahead of the MFMAs.  `fill.run<m>()` runs right behind MFMA m.
// The fragment pipeline runs ACROSS blocks (round 4): the buffers live at kernel scope (FragBuf), group g of a block
// sits in buffer (ROT + g) % NBUF, and a block with COUT requests the first PD groups of the NEXT chunk (addr_next)
// behind its own last MFMAs -- the ring's barrier has vouched for that chunk one interval early (RingW, LAG 0) -- so that
// the next block (CIN) starts on fragments that landed under this block's tail and the barrier instead of opening with
// an exposed LDS round trip.
struct FragBuf {
    static constexpr int PD = CGNN_W8_PD, NBUF = PD + 1;
    u32x4 b[NBUF][CGNN_W8_GS];
};
template <int M>
struct WBlockGeom {
    static constexpr int GS = CGNN_W8_GS, NG = M / GS;
    static_assert(M % GS == 0 && (GS == 4 || GS == 2), "groups of four or two fragments");
    static constexpr int rot_after(int rot) { return (rot + NG) % FragBuf::NBUF; }
};
// the first PD groups of a block, requested ahead of it (before the selector MFMAs of a round's first step)
template <int M, int ROT>
__device__ __forceinline__ void wblock_request(FragBuf& fb, unsigned addr) {
    constexpr int GS = WBlockGeom<M>::GS, NG = WBlockGeom<M>::NG, PD = FragBuf::PD, NBUF = FragBuf::NBUF;
    const unsigned a = addr + (unsigned)(threadIdx.x & 63) * 16u;
    static_for_each([&](auto pc) __attribute__((always_inline)) {
        constexpr int p = decltype(pc)::value;
        if constexpr (p < NG) {
            static_for_each([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                fb.b[(ROT + p) % NBUF][j] = lds_read_b128<(p * GS + j) * 1024>(a);
            }, std::make_integer_sequence<int, GS>{});
        }
    }, std::make_integer_sequence<int, PD>{});
}
template <int NROW, int KS, int ROT, bool CIN, bool COUT, class Fill>
__device__ __forceinline__ void wblockw(f32x16 (&acc)[NROW], const bf16x8 (&in)[KS], FragBuf& fb, unsigned addr, unsigned addr_next,
                                        const Fill& fill) {
    constexpr int M = NROW * KS, GS = WBlockGeom<M>::GS, NG = WBlockGeom<M>::NG, PD = FragBuf::PD, NBUF = FragBuf::NBUF;
    static_assert(!COUT || NG >= PD, "a block that requests the next one's fragments is at least PD groups long");
    const unsigned a = addr + (unsigned)(threadIdx.x & 63) * 16u;
    const unsigned an = addr_next + (unsigned)(threadIdx.x & 63) * 16u;
    (void)an;
    if constexpr (!CIN) wblock_request<M, ROT>(fb, addr);
    static_for_each([&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value;
        // fragment reads issued after group g's (and before this wait)
        constexpr int newer = COUT ? (PD - 1) * GS : ((g + PD - 1 < NG ? g + PD - 1 : NG - 1) - g) * GS;
        constexpr int bi = (ROT + g) % NBUF;
        if constexpr (GS == 4)
            lds_wait4<newer>(fb.b[bi][0], fb.b[bi][1], fb.b[bi][2], fb.b[bi][3]);
        else
            lds_wait2<newer>(fb.b[bi][0], fb.b[bi][1]);
        static_for_each([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value, m = g * GS + j, o = m / KS, ks = m % KS;
            acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb.b[bi][j]), in[ks], acc[o], 0, 0, 0);
            if constexpr (g + PD < NG)
                fb.b[(ROT + g + PD) % NBUF][j] = lds_read_b128<((g + PD) * GS + j) * 1024>(a);
            else if constexpr (COUT)
                fb.b[(ROT + g + PD) % NBUF][j] = lds_read_b128<((g + PD - NG) * GS + j) * 1024>(an);
            fill.template run<m>();
            __builtin_amdgcn_sched_barrier(0);
        }, std::make_integer_sequence<int, GS>{});
    }, std::make_integer_sequence<int, NG>{});
}

// bf16 pack (+ ReLU) of a finished row tile: out[2 t + s] from acc[t][8 s .. 8 s + 7]
template <bool RELU, int DT, int T, int S>
__device__ __forceinline__ void packw_slice(bf16x8 (&out)[2 * DT], const f32x16 (&acc)[DT]) {
    u32x4 v;
    v[0] = pack_bf16(acc[T][8 * S + 0], acc[T][8 * S + 1]);
    v[1] = pack_bf16(acc[T][8 * S + 2], acc[T][8 * S + 3]);
    v[2] = pack_bf16(acc[T][8 * S + 4], acc[T][8 * S + 5]);
    v[3] = pack_bf16(acc[T][8 * S + 6], acc[T][8 * S + 7]);
    const bf16x8 b = __builtin_bit_cast(bf16x8, v);
    out[2 * T + S] = RELU ? relu_bf16(b) : b;
}

// acc[t] = bias rows, t in [T0, T1) (plain LDS loads from a chunk's vector block)
template <int DT, int T0, int T1>
__device__ __forceinline__ void bias_rowsw(f32x16 (&acc)[DT], unsigned vec_addr, int h) {
    if constexpr (w8dev::ABL_BIAS) return;
    const LdsVecPtr b = (LdsVecPtr)(uintptr_t)vec_addr;
#pragma unroll
    for (int t = T0; t < T1; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = *(LdsVec4Ptr)(b + 32 * t + 8 * g + 4 * h);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[t][4 * g + c] = v[c];
        }
}

// LayerNorm (eps 1e-5, biased variance, affine) of acc over the 32 DT features of each edge (edge = lane & 31, the
// other half of its features in lane ^ 32), in two parts so that a ring barrier can sit between them:
//   ln_sums_*    one pass over the accumulator: sum and sum of squares (var = E[x^2] - mean^2 in f32: exact to ~1e-7
//                (1 + mean^2 / var), far below the bf16 operands of this path while |mean| stays within ~100 standard
//                deviations; the centred two-pass form costs 64 more vector instructions per tile and round);
//   ln_affine_w  slices [K0, K1) of 2 DT: eight values each (registers 8 s .. 8 s + 7 of row tile t, k = 2 t + s):
//                RES: ev += y (graph_network.py:182)   !RES: ev = y (the encoder, :57)   and the bf16 pack of the new ev
//                as k-step k of the next layer-0 operand.  gamma at lnv, beta at lnv + 4 D (LDS byte address).
struct LnStats {
    float rstd, nmr;      // normalised value = x * rstd + nmr
};
struct LnSums {           // partial sums of a row's values and squares (four independent chains each)
    float s[4], q[4];
};
__device__ __forceinline__ void ln_sums_clear(LnSums& a) {
#pragma unroll
    for (int i = 0; i < 4; ++i) a.s[i] = a.q[i] = 0.f;
}
// registers [8 S, 8 S + 8) of row tile T join the sums: 16 vector instructions, placed by the caller (behind the MFMAs
// of the NEXT row tile, whose matrix time covers them and the wait for this row tile's last MFMA)
template <bool ZM, int DT, int T, int S>
__device__ __forceinline__ void ln_sums_add(LnSums& a, const f32x16 (&acc)[DT]) {
#pragma unroll
    for (int i = 8 * S; i < 8 * S + 8; i += 4) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if constexpr (!ZM) a.s[c] += acc[T][i + c];
            a.q[c] = __builtin_fmaf(acc[T][i + c], acc[T][i + c], a.q[c]);
        }
    }
}
template <bool ZM, int DT>
__device__ __forceinline__ LnStats ln_stats_finish(const LnSums& a) {
    constexpr int D = 32 * DT;
    const float ex2 = half_swap_sum((a.q[0] + a.q[1]) + (a.q[2] + a.q[3])) * (1.0f / D);
    LnStats st;
    if constexpr (ZM) {
        st.rstd = __builtin_amdgcn_rsqf(ex2 + 1e-5f);
        st.nmr = 0.f;
    } else {
        const float mean = half_swap_sum((a.s[0] + a.s[1]) + (a.s[2] + a.s[3])) * (1.0f / D);
        st.rstd = __builtin_amdgcn_rsqf(__builtin_fmaxf(__builtin_fmaf(-mean, mean, ex2), 0.f) + 1e-5f);
        st.nmr = -mean * st.rstd;
    }
    return st;
}
// gamma / beta of a slice (eight features: two 16-byte pieces of each vector) are read one slice ahead (plain LDS loads:
// hand-issued asm reads with early-clobber register quads cost this kernel 60-110 spilled registers); the scheduling
// barrier per slice keeps the compiler from hoisting every slice's reads to the top (128 registers at latent 128).
template <bool RES, bool BETA, bool ZM, int DT, int K0, int K1>
__device__ __forceinline__ void ln_affine_w(const f32x16 (&acc)[DT], f32x16 (&ev)[DT], bf16x8 (&in)[2 * DT], unsigned lnv, int h,
                                            LnStats st) {
    if constexpr (K1 > K0) {
        constexpr int D = 32 * DT;
        const LdsVecPtr gp = (LdsVecPtr)(uintptr_t)lnv + 4 * h;
        f32x4 gm[2][2], bt[2][2];
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
            gm[K0 & 1][gg] = *(LdsVec4Ptr)(gp + 32 * (K0 >> 1) + 8 * (2 * (K0 & 1) + gg));
            if constexpr (BETA) bt[K0 & 1][gg] = *(LdsVec4Ptr)(gp + D + 32 * (K0 >> 1) + 8 * (2 * (K0 & 1) + gg));
        }
        __builtin_amdgcn_sched_barrier(0);
        static_for_each([&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value + K0, t = k >> 1, s = k & 1, cur = k & 1, nxt = cur ^ 1;
            if constexpr (k + 1 < K1) {
                constexpr int t1 = (k + 1) >> 1, s1 = (k + 1) & 1;
#pragma unroll
                for (int gg = 0; gg < 2; ++gg) {
                    gm[nxt][gg] = *(LdsVec4Ptr)(gp + 32 * t1 + 8 * (2 * s1 + gg));
                    if constexpr (BETA) bt[nxt][gg] = *(LdsVec4Ptr)(gp + D + 32 * t1 + 8 * (2 * s1 + gg));
                }
            }
            if constexpr (w8dev::ABL_LN) {      // (timing-only: pack and add, keeps the MFMAs alive)
                packw_slice<false, DT, t, s>(in, acc);
#pragma unroll
                for (int i = 0; i < 8; ++i) ev[t][8 * s + i] += acc[t][8 * s + i];
            } else {
                // the slice's eight values phase by phase (normalise | add beta and the residual | scale by gamma | pack), a
                // scheduling barrier between the phases: every instruction's inputs are eight instructions old (as one chain
                // per value hipcc reuses ONE temporary and each fmac waits for the add right in front of it)
                float nrm[8], bs[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) nrm[i] = ZM ? acc[t][8 * s + i] * st.rstd : __builtin_fmaf(acc[t][8 * s + i], st.rstd, st.nmr);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (BETA) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) bs[i] = RES ? bt[cur][i >> 2][i & 3] + ev[t][8 * s + i] : bt[cur][i >> 2][i & 3];
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    static_assert(BETA || RES, "a pass without beta adds into the residual");
#pragma unroll
                    for (int i = 0; i < 8; ++i) bs[i] = ev[t][8 * s + i];
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) ev[t][8 * s + i] = __builtin_fmaf(nrm[i], gm[cur][i >> 2][i & 3], bs[i]);
                __builtin_amdgcn_sched_barrier(0);
                u32x4 v;
#pragma unroll
                for (int x = 0; x < 4; ++x) v[x] = pack_bf16(ev[t][8 * s + 2 * x], ev[t][8 * s + 2 * x + 1]);
                in[2 * t + s] = __builtin_bit_cast(bf16x8, v);
            }
            __builtin_amdgcn_sched_barrier(0);
        }, std::make_integer_sequence<int, K1 - K0>{});
    }
}

// selector MFMAs of row tiles [T0, T1): acc[t] = Ps[src] + Pd[dst] (see selp32 in s32.hpp)
template <int DT, int T0, int T1>
__device__ __forceinline__ void selp_rows(f32x16 (&acc)[DT], const bf16x8 (&ps)[2 * DT], const bf16x8 (&pd)[2 * DT], bf16x8 sel0,
                                          bf16x8 sel1) {
    if constexpr (w8dev::ABL_SEL) return;
    static_for_each([&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value + T0;
        f32x16 c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sel0, ps[2 * t], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sel1, ps[2 * t + 1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sel0, pd[2 * t], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sel1, pd[2 * t + 1], c, 0, 0, 0);
        acc[t] = c;
    }, std::make_integer_sequence<int, T1 - T0>{});
}

// fp16 P tables (CGNN_P_F16_S32): acc[t][8 s + j] = Ps[src] + Pd[dst] on the vector pipe, ONE instruction per value --
// v_fma_mix_f32 widens both fp16 halves on the way in (f32 sum of two fp16 numbers: exact up to the final f32 rounding) --
// instead of four selector MFMAs per row tile (16 of the 112 MFMAs of a pass: with them ablated the kernel took 9.3 %
// less time, with these 64 vector instructions in their place ... see DESIGN section 6).
__device__ __forceinline__ float mix_add_lo(unsigned a, unsigned b) {
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float mix_add_hi(unsigned a, unsigned b) {
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <int DT, int T, int S>
__device__ __forceinline__ void addp_half(f32x16 (&acc)[DT], const bf16x8 (&ps)[2 * DT], const bf16x8 (&pd)[2 * DT]) {
    const u32x4 a = __builtin_bit_cast(u32x4, ps[2 * T + S]), b = __builtin_bit_cast(u32x4, pd[2 * T + S]);
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        acc[T][8 * S + 2 * x] = mix_add_lo(a[x], b[x]);
        acc[T][8 * S + 2 * x + 1] = mix_add_hi(a[x], b[x]);
    }
}
template <int DT, int T0, int T1>
__device__ __forceinline__ void addp_rows(f32x16 (&acc)[DT], const bf16x8 (&ps)[2 * DT], const bf16x8 (&pd)[2 * DT]) {
    static_for_each([&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value + T0;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const u32x4 a = __builtin_bit_cast(u32x4, ps[2 * t + s]), b = __builtin_bit_cast(u32x4, pd[2 * t + s]);
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                acc[t][8 * s + 2 * x] = mix_add_lo(a[x], b[x]);
                acc[t][8 * s + 2 * x + 1] = mix_add_hi(a[x], b[x]);
            }
        }
    }, std::make_integer_sequence<int, T1 - T0>{});
}

// ---- the kernel -----------------------------------------------------------------------------------------------------
// One wave, one tile, per pass (= the encoder, or one round), NH hidden layers:
//   first step   the previous pass's LayerNorm affine part (its vectors were copied aside before the barrier) with the
//                second half of the sender rows in flight; selector MFMAs (Ps[src] + Pd[dst] -> accumulators; encoder:
//                bias); layer 0: 32 MFMAs, each finished row tile packed (ReLU) into the next operand under the next
//                rows' MFMAs
//   hidden steps bias, MFMAs, pack
//   last step    bias, MFMAs (first half of the next pass's sender rows and the receiver rows requested from their first
//                slots), LayerNorm statistics, (waves 0-3: LayerNorm vectors -> side buffer)
// Ring pieces go out in the first MFMA slots of every step; every step ends with RingW::interval_end.
template <int DT, int NH, bool ENC, int LAG, bool PF16, bool FOLD>
__global__ __launch_bounds__(CGNN_W8_BLOCK, 2) void edge_stream32w_kernel(
    S32Args a, const __bf16* __restrict__ ps_all, const __bf16* __restrict__ pd_all, int64_t round_stride,
    const int32_t* __restrict__ src, const int32_t* __restrict__ dst, int64_t num_edges, const float* e_in, float* e_out,
    const float* __restrict__ attr, int ld_attr, int seg_k) {
    typedef W8Geom<DT> W;
    static_assert(DT == 4, "the sender-row staging is laid out for latent 128 (four lanes per 64-byte half line)");
    constexpr int D = W::D, KS = W::KS, NP = W::NP;
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool lagging = LAG != 0 && wave >= CGNN_W8_WAVES / 2;
    const int L = a.rounds;
    const int steps_per_tile = (ENC ? NH + 1 : 0) + L * (NH + 1);
    const int64_t tiles = (num_edges + 31) / 32;
    const TileRange tr = tile_range(tiles);
    // the eight waves share the ring's barriers: all run the iteration count of wave 0; a wave whose last tile falls
    // off the range recomputes its previous tile and skips the stores
    const int64_t first0 = tr.first - wave;
    const int iters = __builtin_amdgcn_readfirstlane(
        first0 < tr.end ? (int)((tr.end - first0 + tr.stride - 1) / tr.stride) : 0);
    if (iters == 0) return;
    RingW<W, LAG> ring(a.image, steps_per_tile, wave, lane);
    CGNN_W8_STAMP_BEGIN(ring)
    ring.prime();
    if (lagging) ring.idle_interval();
    const bf16x8 sel0 = p32_selector(lane, 0), sel1 = p32_selector(lane, 1);

    f32x16 ev[DT], acc[DT];
    FragBuf fb;
    constexpr bool CARRY = CGNN_W8_CARRY && LAG == 0;
    // buffer rotation of the fragment pipeline at layer l of a pass (ENC pass: the first Linear has two k-steps)
    auto rot_at = [](bool enc_first, int l) constexpr -> int {
        int r = 0;
        if (l >= 1) r = enc_first ? WBlockGeom<DT * 2>::rot_after(0) : WBlockGeom<DT * KS>::rot_after(0);
        for (int i = 1; i < l; ++i) r = WBlockGeom<DT * KS>::rot_after(r);
        return r;
    };
    (void)rot_at;
    bf16x8 inb[2][2 * DT];      // layer l reads inb[l & 1] and packs into inb[(l + 1) & 1]; LayerNorm writes inb[0]
    bf16x8 ps[2 * DT], pd[2 * DT];
    int lnpar = 0;              // which side buffer holds the pending LayerNorm's vectors
    LnStats lnst = {0.f, 0.f};  // ... and its statistics
    constexpr int KA = CGNN_W8_LN_EARLY < 2 * DT ? CGNN_W8_LN_EARLY : 2 * DT;
    int64_t tile = tr.first < tr.end ? tr.first : tr.end - 1;
    bool valid = tr.first < tr.end;

    // ---- receiver rows: one 16-byte chunk per lane of the tile's (<= 4) receiver rows -> LDS -> the eight pieces of this
    // lane's half row (broadcast reads)
    const unsigned pdst = ring.lds0() + W::PDST_OFF + (unsigned)wave * W::PDST_BYTES;
    const int kk = seg_k < 32 ? seg_k : 32;                // edges of a tile per receiver
    auto load_pd_chunk = [&](const __bf16* table, unsigned off) __attribute__((always_inline)) -> u32x4 {
#if defined(__HIP_DEVICE_COMPILE__)
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(table), 0, 0x7fffffff, 0x00020000);
        return __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0);
#else
        return u32x4{0u, 0u, 0u, 0u};
#endif
    };
    auto park_pd = [&](u32x4 chunk) __attribute__((always_inline)) {
        *(__attribute__((address_space(3))) u32x4*)(uintptr_t)(pdst + (unsigned)(lane >> 4) * W::PDST_ROW + (unsigned)(lane & 15) * 16u) = chunk;
    };
    // piece i (0 .. 7) of this lane's half of its receiver's row.  bf16 tables: the half is contiguous (chunk 8 h + i); fp16
    // tables (CGNN_P_F16_S32) interleave the halves line by line: chunk 8 (i >> 2) + 4 h + (i & 3)
    const unsigned pd_read = pdst + (unsigned)(r / kk) * W::PDST_ROW + (PF16 ? 4u : 8u) * (unsigned)h * 16u;
    auto pd_piece = [](int i) constexpr -> unsigned { return PF16 ? (unsigned)(((i >> 2) * 8 + (i & 3)) * 16) : (unsigned)(16 * i); };
    auto read_pd = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2 * DT; ++i)
            pd[i] = __builtin_bit_cast(bf16x8, *(const __attribute__((address_space(3))) u32x4*)(uintptr_t)(pd_read + pd_piece(i)));
    };
    auto read_pd_row = [&](auto tc) __attribute__((always_inline)) {      // the two pieces of row tile t
        constexpr int t = decltype(tc)::value;
#pragma unroll
        for (int i = 2 * t; i < 2 * t + 2; ++i)
            pd[i] = __builtin_bit_cast(bf16x8, *(const __attribute__((address_space(3))) u32x4*)(uintptr_t)(pd_read + pd_piece(i)));
    };
    u32x4 pdchunk = {0u, 0u, 0u, 0u};

    // ---- sender rows through the staging area.  A lane's half row is one 128-byte line = pieces 0 .. 7; sub-gather SUB
    // moves pieces 4 SUB .. 4 SUB + 3 (64 bytes) of all 64 lines: DMA instruction i, lane 4 a + j -> line 16 i + a (edge
    // (16 i + a) % 32, half (16 i + a) / 32), piece 4 SUB + (j ^ x) with x = (a >> 2) & 3; it lands at line * 64 + j * 16,
    // i.e. piece c of a line sits in slot c ^ x: the sixteen lanes of a ds_read_b128 group then hit sixteen different
    // 16-byte bank groups (lines n + 32 h with n % 4 giving four 64-byte positions of a 256-byte bank row, x the slot).
    const unsigned psst = ring.lds0() + W::PSST_OFF + (unsigned)wave * W::PSST_BYTES;
    const unsigned ps_cq = (unsigned)((lane & 3) ^ ((lane >> 4) & 3)) * 16u;                       // this lane's piece within the quad
    const unsigned ps_read = psst + (unsigned)(r + 32 * h) * 64u + (unsigned)((r >> 2) & 3) * 16u; // slot of piece c: ps_read ^ (c << 4)
    unsigned so_lo = 0, so_hi = 0;      // byte offsets of the rows of edges (lane >> 2) and 16 + (lane >> 2) of the tile
    auto stage_ps = [&](auto sub_c, auto i_c, const __bf16* table) __attribute__((always_inline)) {
        constexpr int SUB = decltype(sub_c)::value, i = decltype(i_c)::value;
        (void)SUB, (void)i, (void)ps_cq;      // (used in the device pass only)
        asm volatile("" ::: "memory");
#if defined(__HIP_DEVICE_COMPILE__)
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(table), 0, 0x7fffffff, 0x00020000);
        char* dstp = cgnn_smem + W::PSST_OFF + (unsigned)wave * W::PSST_BYTES + (unsigned)i * 1024u;
        // (the constant part of the source offset travels as the scalar offset: the instruction's immediate offset is
        // added to the LDS address as well)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LdsVoidPtrG)dstp, 16, ((i & 1) ? so_hi : so_lo) + ps_cq,
                                                 PF16 ? (i >> 1) * 64 + SUB * 128 : (i >> 1) * D + SUB * 64, 0, 0);
#endif
        asm volatile("" ::: "memory");
    };
    auto read_ps = [&](auto sub_c) __attribute__((always_inline)) {
        constexpr int SUB = decltype(sub_c)::value;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            ps[4 * SUB + c] = __builtin_bit_cast(
                bf16x8, *(const __attribute__((address_space(3))) u32x4*)(uintptr_t)(ps_read ^ ((unsigned)c << 4)));
    };
    // ring pieces of this interval: NP consecutive MFMA slots of the block
    auto pieces_at = [&](auto qc, auto mc) __attribute__((always_inline)) {      // blocks of M MFMAs: slots P0 .. P0 + NP - 1
        constexpr int q = decltype(qc)::value, M = decltype(mc)::value;
        constexpr int P0 = CGNN_W8_PIECE_SLOT + NP <= M ? CGNN_W8_PIECE_SLOT : M - NP;
        if constexpr (q >= P0 && q < P0 + NP) ring.piece(q - P0);
    };

    // A tile's per-lane inputs: the sender rows' byte offsets for the staging gathers, the receiver chunk's offset and (ENC)
    // the lane's eight edge features (k-step 0 of the encoder's operand: element j = feature 8 (j >> 2) + 4 h + (j & 3)).
    // Loaded a whole tile ahead (at the top of the current tile, seven registers): their latency is a trip to HBM, and every
    // wave of the workgroup would otherwise sit in it at the same time.
    struct TileIn {      // raw loaded values: nothing is computed from them before the tile starts (a use would wait for the load)
        int src_lo, src_hi, dst_row;
        f32x4 q;          // ENC: the edge's (<= 4) features: one aligned 16-byte load (the launcher checks enc_in <= 4, ld_attr % 4 == 0)
    };
    auto load_tile_in = [&](int64_t T, TileIn& ti) __attribute__((always_inline)) {
        const int64_t e = T * 32 + r;
        const int64_t ce = e < num_edges ? e : num_edges - 1;
        const int64_t e_lo = T * 32 + (lane >> 2), e_hi = e_lo + 16;
        ti.src_lo = src[e_lo < num_edges ? e_lo : num_edges - 1];
        ti.src_hi = src[e_hi < num_edges ? e_hi : num_edges - 1];
        // lane l: chunk l & 15 of the row of receiver group l >> 4 (groups beyond the tile's last repeat it)
        const int groups = 32 / kk;
        const int j = (lane >> 4) < groups ? (lane >> 4) : groups - 1;
        const int64_t eg = T * 32 + (int64_t)j * kk;
        ti.dst_row = dst[eg < num_edges ? eg : num_edges - 1];
        if (ENC) ti.q = *reinterpret_cast<const f32x4*>(attr + ce * ld_attr);
    };
    TileIn nxt;
    load_tile_in(tile, nxt);

    for (int it = 0; it < iters; ++it) {
        // ---- this tile's edges and inputs (requested during the previous tile's last block) ------------------------
        // byte offset of the one chunk this lane fetches of the tile's receiver rows
        const unsigned dchunk = (unsigned)nxt.dst_row * (unsigned)D * 2u + (unsigned)(lane & 15) * 16u;
        const int64_t tn = tile + tr.stride;
        const int64_t tile_next = tn < tr.end ? tn : tile;      // (a wave past its range recomputes its last tile)
        {
            so_lo = (unsigned)nxt.src_lo * (unsigned)D * 2u;
            so_hi = (unsigned)nxt.src_hi * (unsigned)D * 2u;
            const f32x4 q_now = nxt.q;
            load_tile_in(tile_next, nxt);
            if (ENC) {
                // lane (r, h), k-step 0, element j = edge feature 8 (j >> 2) + 4 h + (j & 3): features 0 .. 3 sit in the h = 0 lanes
                const int fin = a.enc_in_dim;
                float v[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) v[jj] = (jj < 4 && h == 0 && jj < fin) ? q_now[jj] : 0.f;
                u32x4 w;
                w[0] = pack_bf16(v[0], v[1]);
                w[1] = pack_bf16(v[2], v[3]);
                w[2] = pack_bf16(v[4], v[5]);
                w[3] = pack_bf16(v[6], v[7]);
                inb[0][0] = __builtin_bit_cast(bf16x8, w);
                const u32x4 z = {0u, 0u, 0u, 0u};
                inb[0][1] = __builtin_bit_cast(bf16x8, z);
            } else {
                load_tile<DT>(ev, e_in + tile * (32 * D), lane);
#pragma unroll
                for (int t = 0; t < DT; ++t)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        u32x4 w;
#pragma unroll
                        for (int x = 0; x < 4; ++x) w[x] = pack_bf16(ev[t][8 * s + 2 * x], ev[t][8 * s + 2 * x + 1]);
                        inb[0][2 * t + s] = __builtin_bit_cast(bf16x8, w);
                    }
                // round 0's P rows, synchronously (once per tile)
                park_pd(load_pd_chunk(pd_all, dchunk));
                static_for_each([&](auto ic) __attribute__((always_inline)) { stage_ps(CGNN_IC(0), ic, ps_all); },
                                std::make_integer_sequence<int, 4>{});
                CGNN_S32_VMCNT(0);
                read_ps(CGNN_IC(0));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                static_for_each([&](auto ic) __attribute__((always_inline)) { stage_ps(CGNN_IC(1), ic, ps_all); },
                                std::make_integer_sequence<int, 4>{});
                CGNN_S32_VMCNT(0);
                read_ps(CGNN_IC(1));
                read_pd();
            }
        }

        // ---- the steps ---------------------------------------------------------------------------------------------
        // first step of a pass.  IS_ENC: the edge encoder (layer 0 = Linear of the edge features with bias, no P rows);
        // otherwise round rr (layer 0 = Ps[src] + Pd[dst] + We e).  PEND: LayerNorm of the previous pass still to do
        // (0 none: the tile's first pass, its P rows are in registers; 1 the encoder's, 2 a round's).  With PEND the first
        // half of the sender rows sits in the staging area and the receiver chunk in `pdchunk`, both requested in the
        // previous pass's last step.
        auto step_first = [&](auto enc_tag, auto pend_tag, int rr) __attribute__((always_inline)) {
            constexpr bool IS_ENC = decltype(enc_tag)::value;
            constexpr int PEND = decltype(pend_tag)::value;
            constexpr int KS0 = IS_ENC ? 2 : KS;     // the encoder's first Linear: K padded to one 32-wide k tile
            // fp16 P tables: the sums Ps[src] + Pd[dst] of row tiles 1 .. DT - 1 ride in the MFMA slots of the row tile before
            // (three hidden layers: the pieces read just in time inside the block cost four spilled registers; there the sums
            // stay in front of the block)
            constexpr bool INBLK = PF16 && CGNN_W8_PF16_INBLK && NH <= 2 && !IS_ENC && PEND != 0;
            CGNN_W8_STAMP(8);
            if constexpr (PEND != 0) {
                if constexpr (!IS_ENC) {
                    // the first half of the sender rows was requested behind the ring pieces of the last interval; its
                    // LDS-DMA is invisible to the compiler's own waits (LAG 1: interval_end has already waited for it)
                    if constexpr (LAG == 0) CGNN_S32_VMCNT(0);
                    park_pd(pdchunk);
                    read_ps(CGNN_IC(0));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the staging area is free again
                    const __bf16* tps = ps_all + (int64_t)rr * round_stride;
                    static_for_each([&](auto ic) __attribute__((always_inline)) { stage_ps(CGNN_IC(1), ic, tps); },
                                    std::make_integer_sequence<int, 4>{});
                }
                // (measured and dropped: the selector MFMAs of a row tile right behind its two LayerNorm slices, so that they
                // run under the next slices' vector work: no gain, four spilled registers)
                ln_affine_w<PEND == 2, !(FOLD && PEND == 2), FOLD, DT, KA, 2 * DT>(acc, ev, inb[0], ring.lnbuf(lnpar ^ 1), h, lnst);
                if constexpr (!IS_ENC) {
                    if constexpr (INBLK) read_pd_row(CGNN_IC(0)); else read_pd();
                }
            }
            const unsigned base = ring.base();
            CGNN_W8_STAMP(9);
            // layer 0's first fragment groups land under the selector MFMAs
            constexpr bool REQ = CARRY && !IS_ENC && CGNN_W8_REQ_EARLY;
            if constexpr (REQ) wblock_request<DT * KS0, 0>(fb, base);
            if constexpr (IS_ENC) {
                bias_rowsw<DT, 0, DT>(acc, ring.vec_addr() + (unsigned)D * 4u, h);      // the encoder's first bias: vector 1 of its own chunk
            } else if constexpr (PEND != 0) {
                if constexpr (INBLK) {
                    addp_rows<DT, 0, 1>(acc, ps, pd);      // row tile 0 here, the others from the MFMA slots of the block below
                } else if constexpr (PF16) {
                    addp_rows<DT, 0, DT / 2>(acc, ps, pd);
                    CGNN_S32_VMCNT(0);      // the second half of the sender rows (requested a LayerNorm ago; nothing newer in flight)
                    read_ps(CGNN_IC(1));
                    addp_rows<DT, DT / 2, DT>(acc, ps, pd);
                } else {
                    selp_rows<DT, 0, DT / 2>(acc, ps, pd, sel0, sel1);
                    CGNN_S32_VMCNT(0);      // the second half of the sender rows (requested a LayerNorm ago; nothing newer in flight)
                    read_ps(CGNN_IC(1));
                    selp_rows<DT, DT / 2, DT>(acc, ps, pd, sel0, sel1);
                }
            } else {
                if constexpr (PF16) addp_rows<DT, 0, DT>(acc, ps, pd);
                else selp_rows<DT, 0, DT>(acc, ps, pd, sel0, sel1);
            }
            CGNN_W8_STAMP(10);
            const bf16x8 (&in0)[KS0] = reinterpret_cast<const bf16x8(&)[KS0]>(inb[0][0]);
            // each finished row tile is packed under the next one's MFMAs, and its accumulators take the NEXT layer's bias
            // (vector 0 of this chunk: the image stores every bias one chunk early, so that these reads are over before the
            // barrier instead of opening the next step with an LDS round trip)
            wblockw<DT, KS0, 0, REQ, CARRY>(acc, in0, fb, base, ring.base_next(), make_fill([&](auto qc) __attribute__((always_inline)) {
                constexpr int q = decltype(qc)::value, t = q / KS0 - 1, w = q % KS0;
                pieces_at(qc, CGNN_IC(DT * KS0));
                if constexpr (INBLK && t + 2 < DT) {      // row tile tn = t + 2 while row tile t + 1 runs its MFMAs
                    constexpr int tn = t + 2 < DT ? t + 2 : 0;
                    if constexpr (w == 0) {
                        if constexpr (tn == DT / 2) {
                            CGNN_S32_VMCNT(0);      // the second half of the sender rows (requested a LayerNorm ago)
                            read_ps(CGNN_IC(1));
                        }
                        read_pd_row(CGNN_IC(tn));
                    }
                    if constexpr (w == CGNN_W8_ADDP0) addp_half<DT, tn, 0>(acc, ps, pd);
                    if constexpr (w == CGNN_W8_ADDP1) addp_half<DT, tn, 1>(acc, ps, pd);
                }
                if constexpr (t >= 0 && w == (KS0 > 4 ? CGNN_W8_PK0 : (KS0 > 2 ? 1 : KS0 - 1))) packw_slice<true, DT, (t < 0 ? 0 : t), 0>(inb[1], acc);
                if constexpr (t >= 0 && w == (KS0 > 4 ? CGNN_W8_PK1 : KS0 - 1)) {
                    packw_slice<true, DT, (t < 0 ? 0 : t), 1>(inb[1], acc);
                    bias_rowsw<DT, (t < 0 ? 0 : t), (t < 0 ? 0 : t) + 1>(acc, ring.vec_addr(), h);
                }
            }));
            packw_slice<true, DT, DT - 1, 0>(inb[1], acc);
            packw_slice<true, DT, DT - 1, 1>(inb[1], acc);
            bias_rowsw<DT, DT - 1, DT>(acc, ring.vec_addr(), h);
            ring.dma_done();
            ring.template interval_end<0, 11>();
            ring.advance();
        };
        // hidden layer l (1 .. NH - 1)
        auto step_hidden = [&](auto enc_tag, auto lc) __attribute__((always_inline)) {
            constexpr int l = decltype(lc)::value;
            constexpr bool IS_ENC = decltype(enc_tag)::value;
            CGNN_W8_STAMP(14);
            wblockw<DT, KS, rot_at(IS_ENC, l), CARRY, CARRY>(acc, inb[l & 1], fb, ring.base(), ring.base_next(), make_fill([&](auto qc) __attribute__((always_inline)) {
                constexpr int q = decltype(qc)::value, t = q / KS - 1, w = q % KS;
                pieces_at(qc, CGNN_IC(DT * KS));
                if constexpr (t >= 0 && w == (KS > 4 ? CGNN_W8_PK0 : (KS > 2 ? 1 : KS - 1))) packw_slice<true, DT, (t < 0 ? 0 : t), 0>(inb[(l + 1) & 1], acc);
                if constexpr (t >= 0 && w == (KS > 4 ? CGNN_W8_PK1 : KS - 1)) {
                    packw_slice<true, DT, (t < 0 ? 0 : t), 1>(inb[(l + 1) & 1], acc);
                    bias_rowsw<DT, (t < 0 ? 0 : t), (t < 0 ? 0 : t) + 1>(acc, ring.vec_addr(), h);
                }
            }));
            packw_slice<true, DT, DT - 1, 0>(inb[(l + 1) & 1], acc);
            packw_slice<true, DT, DT - 1, 1>(inb[(l + 1) & 1], acc);
            bias_rowsw<DT, DT - 1, DT>(acc, ring.vec_addr(), h);
            ring.dma_done();
            ring.template interval_end<0, 15>();
            ring.advance();
        };
        // output layer; WITH_P: request the next pass's P rows (tables nps / npd) from the slots behind the ring pieces
        auto step_last = [&](auto enc_tag, auto p_tag, auto res_tag, const __bf16* nps, const __bf16* npd) __attribute__((always_inline)) {
            constexpr bool WITH_P = decltype(p_tag)::value, RES = decltype(res_tag)::value, IS_ENC = decltype(enc_tag)::value;
            CGNN_W8_STAMP(0);
            CGNN_W8_STAMP(1);
            LnSums sums;
            ln_sums_clear(sums);
            wblockw<DT, KS, rot_at(IS_ENC, NH), CARRY, false>(acc, inb[NH & 1], fb, ring.base(), 0u, make_fill([&](auto qc) __attribute__((always_inline)) {
                constexpr int q = decltype(qc)::value, t = q / KS - 1, w = q % KS;
                pieces_at(qc, CGNN_IC(DT * KS));
                constexpr int PQ = CGNN_W8_PIECE_SLOT == 0 ? NP : 0;      // the P requests: behind the pieces, or in the first slots
                if constexpr (WITH_P && q >= PQ && q < PQ + 4) stage_ps(CGNN_IC(0), CGNN_IC(q - PQ), nps);
                if constexpr (WITH_P && q == PQ + 4) pdchunk = load_pd_chunk(npd, dchunk);
                // LayerNorm's sums of a finished row tile, under the next row tile's MFMAs
                if constexpr (t >= 0 && w == (KS > 4 ? CGNN_W8_SM0 : (KS > 2 ? 2 : KS - 1))) ln_sums_add<FOLD, DT, (t < 0 ? 0 : t), 0>(sums, acc);
                if constexpr (t >= 0 && w == (KS > 4 ? CGNN_W8_SM1 : KS - 1)) ln_sums_add<FOLD, DT, (t < 0 ? 0 : t), 1>(sums, acc);
            }));
            CGNN_W8_STAMP(2);
            // LayerNorm: the statistics and the first KA affine slices here (vectors straight from this chunk), the rest behind
            // the barrier
            ln_sums_add<FOLD, DT, DT - 1, 0>(sums, acc);
            ln_sums_add<FOLD, DT, DT - 1, 1>(sums, acc);
            lnst = ln_stats_finish<FOLD, DT>(sums);
            // FOLD: a round that has a successor (WITH_P) adds no beta (the caller folded it forward, include/cgnn.h)
            ln_affine_w<RES, !(FOLD && RES && WITH_P), FOLD, DT, 0, KA>(acc, ev, inb[0], ring.vec_addr() + (unsigned)D * 4u, h, lnst);
            // the LayerNorm vectors of this chunk, for the affine part that runs in the next interval (when this slot may
            // already be refilled): waves 0-3 copy gamma | beta (2 D floats) aside
            if (wave < CGNN_W8_WAVES / 2) {
                const int idx = wave * 64 + lane;
                if (idx < 2 * D) {
                    const LdsVecPtr sp = (LdsVecPtr)(uintptr_t)(ring.vec_addr() + (unsigned)D * 4u);
                    const float v = sp[idx];
                    ((__attribute__((address_space(3))) float*)(uintptr_t)ring.lnbuf(lnpar))[idx] = v;
                }
            }
            lnpar ^= 1;
            ring.dma_done();
            ring.template interval_end<(WITH_P && (!CARRY || CGNN_W8_PIECE_SLOT == 0) ? 5 : 0), 3>();
            ring.advance();
        };
        auto hidden_steps = [&](auto enc_tag) __attribute__((always_inline)) {
            static_for_each([&](auto lc) __attribute__((always_inline)) {
                step_hidden(enc_tag, std::integral_constant<int, decltype(lc)::value + 1>{});
            }, std::make_integer_sequence<int, NH - 1>{});
        };
        constexpr std::true_type ENC_PASS{};
        constexpr std::false_type ROUND_PASS{};

        if (ENC) {
            step_first(ENC_PASS, CGNN_IC(0), 0);
            hidden_steps(ENC_PASS);
            step_last(ENC_PASS, std::true_type{}, std::false_type{}, ps_all, pd_all);
            step_first(ROUND_PASS, CGNN_IC(1), 0);
            hidden_steps(ROUND_PASS);
        } else {
            step_first(ROUND_PASS, CGNN_IC(0), 0);
            hidden_steps(ROUND_PASS);
        }
        for (int rr = 1; rr < L; ++rr) {
            step_last(ROUND_PASS, std::true_type{}, std::true_type{}, ps_all + (int64_t)rr * round_stride, pd_all + (int64_t)rr * round_stride);
            step_first(ROUND_PASS, CGNN_IC(2), rr);
            hidden_steps(ROUND_PASS);
        }
        step_last(ROUND_PASS, std::false_type{}, std::true_type{}, ps_all, pd_all);
        ln_affine_w<true, true, FOLD, DT, KA, 2 * DT>(acc, ev, inb[0], ring.lnbuf(lnpar ^ 1), h, lnst);

        if (valid) {
            // the final latents go past the caches (written once, not read again in this launch: as plain stores they
            // push the P rows out of L2; -0.6 % in a same-box A/B)
            float* const tb = e_out + tile * (32 * D);
#pragma unroll
            for (int t = 0; t < DT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = {ev[t][4 * g], ev[t][4 * g + 1], ev[t][4 * g + 2], ev[t][4 * g + 3]};
                    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(tb + ((4 * t + g) * 64 + lane) * 4));
                }
        }
        if (tn < tr.end) {
            tile = tn;
        } else {
            valid = false;
        }
    }
    if (LAG != 0 && !lagging) ring.idle_interval();
    CGNN_W8_STAMP_END(wave, lane)
    CGNN_S32_VMCNT(0);      // the ring's last refills (unread) must have landed before the workgroup's LDS is released
    __builtin_amdgcn_s_barrier();
}

template <int DT, int NH, int LAG, bool PF16, bool FOLD>
static int launch_w8(const S32Args& a, const __bf16* ps, const __bf16* pd, int64_t round_stride, const int32_t* src,
                     const int32_t* dst, int64_t num_edges, const float* e_in, float* e_out, const float* attr, int ld_attr,
                     int seg_k, hipStream_t st) {
    typedef W8Geom<DT> W;
    const bool enc = a.enc_in_dim > 0;
    auto kern = enc ? edge_stream32w_kernel<DT, NH, true, LAG, PF16, FOLD> : edge_stream32w_kernel<DT, NH, false, LAG, PF16, FOLD>;
    if (W::LDS > 48 * 1024) {
        int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), W::LDS, "hipFuncSetAttribute(edge_stream32w)");
        if (rc != CGNN_OK) return rc;
    }
    const int64_t tiles = (num_edges + 31) / 32;
    const int grid = grid_for_tiles(tiles, 1, CGNN_W8_WAVES);
    kern<<<grid, CGNN_W8_BLOCK, W::LDS, st>>>(a, ps, pd, round_stride, src, dst, num_edges, e_in, e_out, attr, ld_attr, seg_k);
    CGNN_W8_STAMP_REPORT(st, num_edges, grid, a.rounds + (enc ? 1 : 0), LAG)
    return check_hip(hipGetLastError(), "cgnn_edge_stream_run_w8 launch");
}

}  // namespace cgnn

using namespace cgnn;

// The image of cgnn_edge_stream_image_build with every bias moved ONE CHUNK EARLY: vector 0 of chunk c = the bias of chunk
// c + 1's layer (zeros where that layer has none: a round's first Linear, whose bias lives in its Pd table), and vector 1
// of chunk 0 = that chunk's own bias (the encoder's first Linear; first-layer chunks carry no LayerNorm vectors there).
extern "C" int cgnn_edge_stream_image_build_w8(const cgnn_mlp* rounds, int32_t num_rounds, const cgnn_mlp* encoder,
                                               int32_t latent, void* image, size_t image_bytes, void* stream) {
    int rc = cgnn_edge_stream_image_build(rounds, num_rounds, encoder, latent, image, image_bytes, stream);
    if (rc != CGNN_OK) return rc;
    const size_t stride = s32_stride(latent);
    const int nh = rounds[0].num_hidden_layers;
    const int count = (num_rounds + (encoder ? 1 : 0)) * (nh + 1);
    hipStream_t st = (hipStream_t)stream;
    const size_t vbytes = (size_t)latent * 4, voff = (size_t)latent * latent * 2;
    char* img = (char*)image;
    rc = check_hip(hipMemcpyAsync(img + voff + vbytes, img + voff, vbytes, hipMemcpyDeviceToDevice, st), "hipMemcpyAsync(own bias)");
    for (int c = 0; c + 1 < count && rc == CGNN_OK; ++c)
        rc = check_hip(hipMemcpyAsync(img + (size_t)c * stride + voff, img + (size_t)(c + 1) * stride + voff, vbytes,
                                      hipMemcpyDeviceToDevice, st), "hipMemcpyAsync(bias, one chunk early)");
    if (rc == CGNN_OK) rc = check_hip(hipMemsetAsync(img + (size_t)(count - 1) * stride + voff, 0, vbytes, st), "hipMemsetAsync(last bias)");
    return rc;
}

extern "C" int cgnn_edge_stream_w8_supported(int32_t latent, int32_t num_hidden_layers, int32_t fixed_k) {
    const bool k_ok = fixed_k >= 8 && (fixed_k <= 32 ? 32 % fixed_k == 0 : fixed_k % 32 == 0);
    return (latent == 128 && num_hidden_layers >= 1 && num_hidden_layers <= 3 && k_ok) ? 1 : 0;
}

extern "C" int cgnn_edge_stream_run_w8(const void* image, size_t image_bytes, int32_t latent, int32_t num_hidden_layers,
                                       int32_t num_rounds, int32_t enc_in_dim, const void* ps_all, const void* pd_all,
                                       int64_t round_stride, const int32_t* src, const int32_t* dst, int64_t num_edges,
                                       const float* e_in, float* e_out, const float* edge_attr, int32_t ld_attr, int32_t lag,
                                       int32_t fixed_k, int32_t p_format, int32_t flags, void* stream) {
    if (!image || !ps_all || !pd_all || !src || !dst || !e_out || num_edges < 0 || num_rounds < 1 || num_hidden_layers < 1 ||
        round_stride < 0 || (enc_in_dim > 0 ? (!edge_attr || ld_attr < enc_in_dim) : !e_in) || lag < 0 || lag > 1 || fixed_k < 0) {
        set_error("cgnn_edge_stream_run_w8: invalid argument");
        return CGNN_ERR_INVALID_ARG;
    }
    if (p_format != CGNN_P_BF16_S32 && p_format != CGNN_P_F16_S32) {
        set_error("cgnn_edge_stream_run_w8: ps_all / pd_all must be CGNN_P_F16_S32 or CGNN_P_BF16_S32 tables (got p_format %d)",
                  p_format);
        return CGNN_ERR_INVALID_ARG;
    }
    if ((flags & ~CGNN_STREAM_FOLDED) != 0 || ((flags & CGNN_STREAM_FOLDED) && p_format != CGNN_P_F16_S32)) {
        set_error("cgnn_edge_stream_run_w8: flags = %d (known: CGNN_STREAM_FOLDED, with CGNN_P_F16_S32 tables only)", flags);
        return CGNN_ERR_INVALID_ARG;
    }
    if (p_format == CGNN_P_F16_S32 && lag != 0) {
        set_error("cgnn_edge_stream_run_w8: fp16 tables (CGNN_P_F16_S32) run with lag = 0 (lag = 1 is kept for bf16 tables)");
        return CGNN_ERR_UNSUPPORTED;
    }
    if (enc_in_dim > 4 || (enc_in_dim > 0 && ((ld_attr & 3) != 0 || ((uintptr_t)edge_attr & 15) != 0))) {
        set_error("cgnn_edge_stream_run_w8: the in-launch encoder reads an edge's features with one aligned 16-byte load: at most "
                  "4 features (got %d), ld_attr a multiple of 4 (got %d), edge_attr 16-byte aligned; other layouts: "
                  "cgnn_edge_stream_run", enc_in_dim, ld_attr);
        return CGNN_ERR_UNSUPPORTED;
    }
    if (!cgnn_edge_stream_w8_supported(latent, num_hidden_layers, fixed_k) || num_edges % fixed_k != 0) {
        set_error("cgnn_edge_stream_run_w8: no kernel for latent=%d, %d hidden layers, fixed_k=%d on %lld edges (built for latent "
                  "128, 1..3 hidden layers of the same width and receiver-sorted edge lists of fixed in-degree 8, 16, 32, 64, ...; "
                  "other shapes: cgnn_edge_stream_run)", latent, num_hidden_layers, fixed_k, (long long)num_edges);
        return CGNN_ERR_UNSUPPORTED;
    }
    const size_t need = cgnn_edge_stream_image_bytes(latent, num_hidden_layers, num_rounds, enc_in_dim > 0);
    if (image_bytes < need) {
        set_error("cgnn_edge_stream_run_w8: image has %zu bytes, this model needs %zu", image_bytes, need);
        return CGNN_ERR_INVALID_ARG;
    }
    if (num_edges == 0) return CGNN_OK;
    S32Args a;
    a.image = (const char*)image;
    a.rounds = num_rounds;
    a.nh = num_hidden_layers;
    a.enc_in_dim = enc_in_dim > 0 ? enc_in_dim : 0;
    hipStream_t st = (hipStream_t)stream;
#define CGNN_W8_GO(NHx, LAGx, PFx, FOLDx)                                                                                    \
    return launch_w8<4, NHx, LAGx, PFx, FOLDx>(a, (const __bf16*)ps_all, (const __bf16*)pd_all, round_stride, src, dst, num_edges, \
                                               e_in, e_out, edge_attr, ld_attr, fixed_k, st)
    if (flags & CGNN_STREAM_FOLDED) {
        switch (num_hidden_layers) {
            case 1: CGNN_W8_GO(1, 0, true, true);
            case 2: CGNN_W8_GO(2, 0, true, true);
            default: CGNN_W8_GO(3, 0, true, true);
        }
    }
    if (p_format == CGNN_P_F16_S32) {
        switch (num_hidden_layers) {
            case 1: CGNN_W8_GO(1, 0, true, false);
            case 2: CGNN_W8_GO(2, 0, true, false);
            default: CGNN_W8_GO(3, 0, true, false);
        }
    }
    if (lag) {
        switch (num_hidden_layers) {
            case 1: CGNN_W8_GO(1, 1, false, false);
            case 2: CGNN_W8_GO(2, 1, false, false);
            default: CGNN_W8_GO(3, 1, false, false);
        }
    }
    switch (num_hidden_layers) {
        case 1: CGNN_W8_GO(1, 0, false, false);
        case 2: CGNN_W8_GO(2, 0, false, false);
        default: CGNN_W8_GO(3, 0, false, false);
    }
#undef CGNN_W8_GO
}
