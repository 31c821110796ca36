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
//     two waves: one wave's bf16 pack, LayerNorm and waits run under the other's MFMAs, and the SIMD issues two vector
//     instructions in the time a lone wave issues one;
//   * 256 registers per wave, all architectural: f32 latent 64, accumulators 64, bf16 operand 32 (+ the next operand,
//     packed row tile by row tile as accumulators die), weight fragments, P rows.  Nothing is parked; every memory
//     operation except the fragment reads is visible to the compiler (no hand-counted register loads);
//   * optional LAG: waves 4-7 (the second wave of every SIMD) run one ring step behind waves 0-3, so that one wave's
//     LayerNorm (vector pipe only) coincides with its partner's matrix work instead of its partner's LayerNorm.
// Same image as cgnn_edge_stream_run (cgnn_edge_stream_image_build), same P tables (CGNN_P_BF16_S32), same numerics
// (bf16 operands, f32 accumulation, f32 LayerNorm and residual).
#include <string.h>

#include <type_traits>

#include "s32.hpp"

namespace cgnn {

#define CGNN_W8_WAVES 8
#define CGNN_W8_BLOCK (CGNN_W8_WAVES * 64)
#define CGNN_W8_SLOTS 4
#define CGNN_IC(x) std::integral_constant<int, (x)> {}
#ifndef CGNN_W8_GS
#define CGNN_W8_GS 4       // LDS weight fragments per group
#endif
#ifndef CGNN_W8_LN_EARLY
#define CGNN_W8_LN_EARLY 0    // LayerNorm slices (of 2 * latent / 32) done in the output layer's own step, before the barrier
#endif
#ifndef CGNN_W8_PD
#define CGNN_W8_PD 1       // groups in flight ahead of the MFMAs
#endif

#ifdef CGNN_W8_STAMPS   // developer build: per-phase cycle sums (s_memtime into scalar registers, no memory traffic inside the
                        // loop) of one workgroup's waves, printed by the launcher
__device__ unsigned long long cgnn_w8_stamps[8 * 32];
struct W8Timer {
    unsigned long long sum[20], prev;
};
#define CGNN_W8_STAMP(k)                                                                 \
    {                                                                                    \
        unsigned long long t_;                                                           \
        __builtin_amdgcn_sched_barrier(0);                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");       \
        __builtin_amdgcn_sched_barrier(0);                                               \
        tm.sum[k] += t_ - tm.prev;                                                       \
        tm.prev = t_;                                                                    \
    }
#else
#define CGNN_W8_STAMP(k)
#endif

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
    static constexpr unsigned PDST_BYTES = 64u * 16u;
    static constexpr unsigned LDS = PDST_OFF + CGNN_W8_WAVES * PDST_BYTES;
};

// ---- the ring -------------------------------------------------------------------------------------------------------
// Four slots, one barrier per ring step ("interval").  In interval g waves 0-3 compute chunk g out of slot g % 4 and
// waves 4-7 chunk g - LAG; every wave issues its pieces of chunk g + 3 - LAG into the one slot nobody reads or awaits
// (LAG 1: slots g - 1 and g are being read, g + 1 must be complete at the interval's end; LAG 0: g, then g + 1 and
// g + 2 in flight), waits -- counted -- for its own pieces of chunk g + 1 and meets the others at the barrier.
// Every wave issues NP pieces per chunk; where 8 NP exceeds the chunk's pieces the surplus ones repeat an earlier
// piece (same bytes to the same place), which keeps every wave's operation count, and so every wait count, the same.
template <class W, int LAG>
struct RingW {
    const char* image;
    int count;           // chunks per tile
    int wave, lane;
    int slot;            // of this wave's current step
    int dma_chunk, dma_slot;
    __device__ __forceinline__ RingW(const char* img, int cnt, int w, int l)
        : image(img), count(cnt), wave(w), lane(l), slot(0), dma_chunk(0), dma_slot(0) {}
    __device__ __forceinline__ unsigned lds0() const { return (unsigned)(uintptr_t)(LdsWeightPtr)(cgnn_smem); }
    __device__ __forceinline__ unsigned base() const { return lds0() + (unsigned)slot * W::STRIDE; }
    __device__ __forceinline__ unsigned vec_addr() const { return base() + W::VEC_OFF; }
    __device__ __forceinline__ unsigned lnbuf(int which) const { return lds0() + W::LNBUF_OFF + (unsigned)which * W::LNBUF_BYTES; }
    bool primed = false;
    __device__ __forceinline__ void piece(int i) {
#ifdef CGNN_W8_ABL_DMA
        if (primed) return;
#endif
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
        dma_slot = (dma_slot + 1) & (CGNN_W8_SLOTS - 1);
    }
    __device__ __forceinline__ void prime() {
        for (int c = 0; c < CGNN_W8_SLOTS - 1 - LAG; ++c) {
            for (int i = 0; i < W::NP; ++i) piece(i);
            dma_done();
        }
        CGNN_S32_VMCNT(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        primed = true;
    }
    // End of an interval.  EXTRA = vector-memory operations other than ring pieces this wave has issued in this interval
    // and that may still be in flight (they are newer than the pieces waited for).  A smaller count than the true one
    // only waits longer.
#ifdef CGNN_W8_STAMPS
    W8Timer tm;
#endif
    template <int EXTRA, int K0 = 17>      // (K0: first of this call's three timer slots, developer builds)
    __device__ __forceinline__ void interval_end() {
        CGNN_W8_STAMP(K0);
#ifndef CGNN_W8_ABL_VMWAIT
        vm_wait_const<(2 - LAG) * W::NP + EXTRA>();
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        CGNN_W8_STAMP(K0 + 1);
#ifndef CGNN_W8_ABL_BARRIER     // (developer timing builds CGNN_W8_ABL_*: wrong results)
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
        CGNN_W8_STAMP(K0 + 2);
    }
    __device__ __forceinline__ void advance() { slot = (slot + 1) & (CGNN_W8_SLOTS - 1); }
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
template <int NROW, int KS, class Fill>
__device__ __forceinline__ void wblockw(f32x16 (&acc)[NROW], const bf16x8 (&in)[KS], unsigned addr, const Fill& fill) {
    constexpr int M = NROW * KS, GS = (M % CGNN_W8_GS == 0) ? CGNN_W8_GS : 2, NG = M / GS, PD = CGNN_W8_PD, NBUF = PD + 1;
    static_assert(M % GS == 0 && (GS == 4 || GS == 2), "groups of four or two fragments");
    const unsigned a = addr + (unsigned)(threadIdx.x & 63) * 16u;
    u32x4 buf[NBUF][GS];
    static_for_each([&](auto pc) __attribute__((always_inline)) {
        constexpr int p = decltype(pc)::value;
        if constexpr (p < NG) {
            static_for_each([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                buf[p][j] = lds_read_b128<(p * GS + j) * 1024>(a);
            }, std::make_integer_sequence<int, GS>{});
        }
    }, std::make_integer_sequence<int, PD>{});
    static_for_each([&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value;
        constexpr int newer = ((g + PD - 1 < NG ? g + PD - 1 : NG - 1) - g) * GS;     // fragment reads issued after group g's
        if constexpr (GS == 4)
            lds_wait4<newer>(buf[g % NBUF][0], buf[g % NBUF][1], buf[g % NBUF][2], buf[g % NBUF][3]);
        else
            lds_wait2<newer>(buf[g % NBUF][0], buf[g % NBUF][1]);
        static_for_each([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value, m = g * GS + j, o = m / KS, ks = m % KS;
            acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, buf[g % NBUF][j]), in[ks], acc[o], 0, 0,
                                                             0);
            if constexpr (g + PD < NG) buf[(g + PD) % NBUF][j] = lds_read_b128<((g + PD) * GS + j) * 1024>(a);
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

// acc[t] = bias rows (plain LDS loads from the chunk's vector block)
template <int DT>
__device__ __forceinline__ void bias_rowsw(f32x16 (&acc)[DT], unsigned vec_addr, int h) {
    const LdsVecPtr b = (LdsVecPtr)(uintptr_t)vec_addr;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = *(LdsVec4Ptr)(b + 32 * t + 8 * g + 4 * h);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[t][4 * g + c] = v[c];
        }
}

// LayerNorm (eps 1e-5, biased variance, affine) of acc over the 32 DT features of each edge (edge = lane & 31, the
// other half of its features in lane ^ 32), in two parts so that a ring barrier can sit between them:
//   ln_stats_w   one pass over the accumulator: sum and sum of squares (var = E[x^2] - mean^2 in f32: exact to ~1e-7
//                (1 + mean^2 / var), far below the bf16 operands of this path while |mean| stays within ~100 standard
//                deviations; the centred two-pass form costs 64 more vector instructions per tile and round);
//   ln_affine_w  slices [K0, K1) of 2 DT: eight values each (registers 8 s .. 8 s + 7 of row tile t, k = 2 t + s):
//                RES: ev += y (graph_network.py:182)   !RES: ev = y (the encoder, :57)   and the bf16 pack of the new ev
//                as k-step k of the next layer-0 operand.  gamma at lnv, beta at lnv + 4 D (LDS byte address).
struct LnStats {
    float rstd, nmr;      // normalised value = x * rstd + nmr
};
template <int DT>
__device__ __forceinline__ LnStats ln_stats_w(const f32x16 (&acc)[DT]) {
    constexpr int D = 32 * DT;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int i = 0; i < 16; i += 4) {
            s0 += acc[t][i];
            s1 += acc[t][i + 1];
            s2 += acc[t][i + 2];
            s3 += acc[t][i + 3];
            q0 = __builtin_fmaf(acc[t][i], acc[t][i], q0);
            q1 = __builtin_fmaf(acc[t][i + 1], acc[t][i + 1], q1);
            q2 = __builtin_fmaf(acc[t][i + 2], acc[t][i + 2], q2);
            q3 = __builtin_fmaf(acc[t][i + 3], acc[t][i + 3], q3);
        }
    const float mean = half_swap_sum((s0 + s1) + (s2 + s3)) * (1.0f / D);
    const float ex2 = half_swap_sum((q0 + q1) + (q2 + q3)) * (1.0f / D);
    LnStats st;
    st.rstd = __builtin_amdgcn_rsqf(__builtin_fmaxf(__builtin_fmaf(-mean, mean, ex2), 0.f) + 1e-5f);
    st.nmr = -mean * st.rstd;
    return st;
}
template <bool RES, int DT, int K0, int K1>
__device__ __forceinline__ void ln_affine_w(const f32x16 (&acc)[DT], f32x16 (&ev)[DT], bf16x8 (&in)[2 * DT], unsigned lnv, int h,
                                            LnStats st) {
    constexpr int D = 32 * DT;
    const LdsVecPtr gp = (LdsVecPtr)(uintptr_t)lnv;
    __builtin_amdgcn_sched_barrier(0);
    static_for_each([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value + K0, t = k >> 1, s = k & 1;
#ifdef CGNN_W8_ABL_LN       // (developer timing build, wrong results: keeps the MFMAs alive)
        packw_slice<false, DT, t, s>(in, acc);
#pragma unroll
        for (int i = 0; i < 8; ++i) ev[t][8 * s + i] += acc[t][8 * s + i];
#else
        u32x4 v;
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
            const int g = 2 * s + gg;
            const f32x4 gm = *(LdsVec4Ptr)(gp + 32 * t + 8 * g + 4 * h);
            const f32x4 bt = *(LdsVec4Ptr)(gp + D + 32 * t + 8 * g + 4 * h);
            float e[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float nrm = __builtin_fmaf(acc[t][4 * g + c], st.rstd, st.nmr);
                const float base = RES ? bt[c] + ev[t][4 * g + c] : bt[c];
                e[c] = __builtin_fmaf(nrm, gm[c], base);
                ev[t][4 * g + c] = e[c];
            }
            v[2 * gg] = pack_bf16(e[0], e[1]);
            v[2 * gg + 1] = pack_bf16(e[2], e[3]);
        }
        in[2 * t + s] = __builtin_bit_cast(bf16x8, v);
#endif
        __builtin_amdgcn_sched_barrier(0);      // (the scheduler otherwise hoists every slice's LDS reads to the top: 128 registers)
    }, std::make_integer_sequence<int, (K1 > K0 ? K1 - K0 : 0)>{});
}

// ---- the kernel -----------------------------------------------------------------------------------------------------
// One wave, one tile, per pass (= the encoder, or one round), NH hidden layers:
//   first step   [pd rows requested] LayerNorm of the PREVIOUS pass (its vectors were copied aside before the barrier)
//                selector MFMAs (Ps[src] + Pd[dst] -> accumulators) | encoder: bias
//                layer 0: 32 MFMAs; each finished row tile is packed (ReLU) into the next operand under the next rows' MFMAs
//   hidden steps bias, MFMAs, pack
//   last step    bias, [ps rows of the next pass requested], MFMAs, (waves 0-3: LayerNorm vectors -> side buffer)
// Ring pieces are spread over every step's MFMA slots; every step ends with RingW::interval_end.
// PDB ("receiver rows broadcast"): the edge list is receiver-sorted with a fixed in-degree seg_k that divides 32 or is a
// multiple of it (what data_utils.preprocess emits), so a tile's 32 edges have at most four receivers: their
// Pd rows are fetched by ONE load instruction per round (every lane one 16-byte chunk), parked in LDS and read back as
// the eight B-operand pieces with broadcast reads, instead of eight gather instructions in which sixteen lanes fetch the
// same bytes.  Vector-memory instructions are what this kernel's waves wait on most (their issue blocks the wave).
template <int DT, int NH, bool ENC, int LAG, bool PDB>
__global__ __launch_bounds__(CGNN_W8_BLOCK, 2) void edge_stream32w_kernel(
    S32Args a, const __bf16* __restrict__ ps_all, const __bf16* __restrict__ pd_all, int64_t round_stride,
    const int32_t* __restrict__ src, const int32_t* __restrict__ dst, int64_t num_edges, const float* e_in, float* e_out,
    const float* __restrict__ attr, int ld_attr, int seg_k) {
    typedef W8Geom<DT> W;
    constexpr int D = W::D, KS = W::KS, NP = W::NP;
    constexpr int MQ = DT * KS;                 // MFMAs (slots) of a full layer block
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
#ifdef CGNN_W8_STAMPS
    auto& tm = ring.tm;
    for (int k = 0; k < 20; ++k) tm.sum[k] = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm.prev)::"memory");
#endif
    ring.prime();
    if (lagging) ring.idle_interval();
    const bf16x8 sel0 = p32_selector(lane, 0), sel1 = p32_selector(lane, 1);

    f32x16 ev[DT], acc[DT];
    bf16x8 inb[2][2 * DT];      // layer l reads inb[l & 1] and packs into inb[(l + 1) & 1]; LayerNorm writes inb[0]
    bf16x8 ps[2 * DT], pd[2 * DT];
    int lnpar = 0;              // which side buffer holds the pending LayerNorm's vectors
    LnStats lnst = {0.f, 0.f};  // ... and its statistics
    constexpr int KA = CGNN_W8_LN_EARLY < 2 * DT ? CGNN_W8_LN_EARLY : 2 * DT;
    int64_t tile = tr.first < tr.end ? tr.first : tr.end - 1;
    bool valid = tr.first < tr.end;

    // P rows: lane (r, h) owns the 16-byte pieces of its half of the row; piece pc = the B operand of k-step pc
    auto load_p = [&](bf16x8 (&p)[2 * DT], const __bf16* table, unsigned off) __attribute__((always_inline)) {
#ifdef CGNN_W8_ABL_P
        const u32x4 z = {off, off, off, off};
#pragma unroll
        for (int i = 0; i < 2 * DT; ++i) p[i] = __builtin_bit_cast(bf16x8, z);
        return;
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CGNN_W8_P_GLOBAL)
        // buffer form: the round's table as the descriptor (scalar), the lane's row offset as a 32-bit vector offset, the
        // piece as the instruction's immediate: no 64-bit vector address per lane
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(table), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int i = 0; i < 2 * DT; ++i)
            p[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(off + 16u * (unsigned)i), 0, 0));
#else
        const bf16x8* q = reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(table) + off);
#pragma unroll
        for (int i = 0; i < 2 * DT; ++i) p[i] = q[i];
#endif
    };
    auto load_p_piece = [&](const __bf16* table, unsigned off) __attribute__((always_inline)) -> bf16x8 {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CGNN_W8_ABL_P)
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(table), 0, 0x7fffffff, 0x00020000);
        return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0));
#else
        const u32x4 z = {off, off, off, off};
        return __builtin_bit_cast(bf16x8, z);
#endif
    };
    // PDB: one 16-byte chunk per lane of the tile's receiver rows -> LDS -> the eight pieces of this lane's half row
    const unsigned pdst = ring.lds0() + W::PDST_OFF + (unsigned)wave * W::PDST_BYTES;
    const int kk = PDB ? (seg_k < 32 ? seg_k : 32) : 32;                // edges of a tile per receiver
    auto load_pd_chunk = [&](const __bf16* table, unsigned off) __attribute__((always_inline)) -> u32x4 {
#if defined(__HIP_DEVICE_COMPILE__)
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(table), 0, 0x7fffffff, 0x00020000);
        return __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0);
#else
        return u32x4{0u, 0u, 0u, 0u};
#endif
    };
    auto park_pd = [&](u32x4 chunk) __attribute__((always_inline)) {
        *(__attribute__((address_space(3))) u32x4*)(uintptr_t)(pdst + (unsigned)lane * 16u) = chunk;
    };
    auto read_pd = [&](bf16x8 (&p)[2 * DT]) __attribute__((always_inline)) {
        const unsigned base = pdst + ((unsigned)(r / kk) * 16u + 8u * (unsigned)h) * 16u;
#pragma unroll
        for (int i = 0; i < 2 * DT; ++i)
            p[i] = __builtin_bit_cast(bf16x8, *(const __attribute__((address_space(3))) u32x4*)(uintptr_t)(base + 16u * (unsigned)i));
    };
    u32x4 pdchunk = {0u, 0u, 0u, 0u};
    // ring pieces of this interval, spread over the M slots of a block
    auto pieces_at = [&](auto mc, auto qc) __attribute__((always_inline)) {
        constexpr int M = decltype(mc)::value, q = decltype(qc)::value;
        for (int i = share_lo(NP, M, q); i < share_hi(NP, M, q); ++i) ring.piece(i);
    };

    for (int it = 0; it < iters; ++it) {
        // ---- this tile's edges and inputs (compiler-tracked loads: once per tile) ----------------------------------
        unsigned so, dof;       // byte offsets of this lane's halves of the sender / receiver P rows
        unsigned dchunk = 0;    // PDB: byte offset of the one chunk this lane fetches of the tile's receiver rows
        {
            const int64_t e = tile * 32 + r;
            const int64_t ce = e < num_edges ? e : num_edges - 1;
            so = ((unsigned)src[ce] * (unsigned)D + (unsigned)h * (D / 2)) * 2u;
            dof = ((unsigned)dst[ce] * (unsigned)D + (unsigned)h * (D / 2)) * 2u;
            if constexpr (PDB) {
                // lane l: chunk l & 15 of the row of receiver group l >> 4 (groups beyond the tile's last repeat it)
                const int groups = 32 / kk;
                const int j = (lane >> 4) < groups ? (lane >> 4) : groups - 1;
                const int64_t eg = tile * 32 + (int64_t)j * kk;
                dchunk = (unsigned)dst[eg < num_edges ? eg : num_edges - 1] * (unsigned)D * 2u + (unsigned)(lane & 15) * 16u;
            }
#ifdef CGNN_W8_ABL_PSAME      // every lane gathers from the same two rows: two or four cache lines per load instruction
            so = ((unsigned)src[tile * 32 < num_edges ? tile * 32 : 0] * (unsigned)D + (unsigned)h * (D / 2)) * 2u;
#endif
            if (ENC) {
                // lane (r, h), k-step 0, element j = edge feature 8 (j >> 2) + 4 h + (j & 3)
                const int fin = a.enc_in_dim;
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int f = 8 * (j >> 2) + 4 * h + (j & 3);
                    v[j] = f < fin ? attr[ce * ld_attr + f] : 0.f;
                }
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
                load_p(ps, ps_all, so);
                if constexpr (PDB) {
                    park_pd(load_pd_chunk(pd_all, dchunk));
                    read_pd(pd);
                } else {
                    load_p(pd, pd_all, dof);
                }
            }
        }

        // ---- the steps ---------------------------------------------------------------------------------------------
        // first step of a pass.  IS_ENC: the edge encoder (layer 0 = Linear of the edge features with bias, no P rows);
        // otherwise round rr (layer 0 = Ps[src] + Pd[dst] + We e).  PEND: LayerNorm of the previous pass still to do
        // (0 none, 1 the encoder's, 2 a round's).  The sender rows `ps` were requested in the previous pass's last step.
        auto step_first = [&](auto enc_tag, auto pend_tag, int rr) __attribute__((always_inline)) {
            constexpr bool IS_ENC = decltype(enc_tag)::value;
            constexpr int PEND = decltype(pend_tag)::value;
            constexpr int KS0 = IS_ENC ? 2 : KS;     // the encoder's first Linear: K padded to one 32-wide k tile
            constexpr int M0 = DT * KS0;
            CGNN_W8_STAMP(8);
            if constexpr (PEND != 0) {
                // the rest of the previous pass's LayerNorm (its first slices ran before the barrier, in that pass's last step)
                if constexpr (!IS_ENC && PDB) park_pd(pdchunk);      // requested in the previous pass's last step
                if constexpr (!IS_ENC && !PDB) load_p(pd, pd_all + (int64_t)rr * round_stride, dof);
                ln_affine_w<PEND == 2, DT, KA, 2 * DT>(acc, ev, inb[0], ring.lnbuf(lnpar ^ 1), h, lnst);
                if constexpr (!IS_ENC && PDB) read_pd(pd);
            }
            const unsigned base = ring.base();
            CGNN_W8_STAMP(9);
            if constexpr (IS_ENC) {
                bias_rowsw<DT>(acc, ring.vec_addr(), h);
            } else {
                selp32<DT>(acc, ps, pd, sel0, sel1, NoFill32{});
            }
            CGNN_W8_STAMP(10);
            const bf16x8 (&in0)[KS0] = reinterpret_cast<const bf16x8(&)[KS0]>(inb[0][0]);
            wblockw<DT, KS0>(acc, in0, base, make_fill([&](auto qc) __attribute__((always_inline)) {
                constexpr int q = decltype(qc)::value, t = q / KS0 - 1, w = q % KS0;
                pieces_at(CGNN_IC(M0), qc);
                if constexpr (t >= 0 && w == (KS0 > 2 ? 1 : KS0 - 1)) packw_slice<true, DT, (t < 0 ? 0 : t), 0>(inb[1], acc);
                if constexpr (t >= 0 && w == (KS0 > 4 ? 3 : KS0 - 1)) packw_slice<true, DT, (t < 0 ? 0 : t), 1>(inb[1], acc);
            }));
            packw_slice<true, DT, DT - 1, 0>(inb[1], acc);
            packw_slice<true, DT, DT - 1, 1>(inb[1], acc);
            ring.dma_done();
            ring.template interval_end<0, 11>();
            ring.advance();
        };
        // hidden layer l (1 .. NH - 1)
        auto step_hidden = [&](auto lc, int rr) __attribute__((always_inline)) {
            constexpr int l = decltype(lc)::value;
            CGNN_W8_STAMP(14);
            bias_rowsw<DT>(acc, ring.vec_addr(), h);
            wblockw<DT, KS>(acc, inb[l & 1], ring.base(), make_fill([&](auto qc) __attribute__((always_inline)) {
                constexpr int q = decltype(qc)::value, t = q / KS - 1, w = q % KS;
                pieces_at(CGNN_IC(MQ), qc);
                if constexpr (t >= 0 && w == (KS > 2 ? 1 : KS - 1)) packw_slice<true, DT, (t < 0 ? 0 : t), 0>(inb[(l + 1) & 1], acc);
                if constexpr (t >= 0 && w == (KS > 4 ? 3 : KS - 1)) packw_slice<true, DT, (t < 0 ? 0 : t), 1>(inb[(l + 1) & 1], acc);
            }));
            packw_slice<true, DT, DT - 1, 0>(inb[(l + 1) & 1], acc);
            packw_slice<true, DT, DT - 1, 1>(inb[(l + 1) & 1], acc);
            ring.dma_done();
            ring.template interval_end<0, 15>();
            ring.advance();
        };
        // output layer; WITH_PS: request the next pass's sender rows (table `nps`) first
        auto step_last = [&](auto ps_tag, auto res_tag, const __bf16* nps, const __bf16* npd, int rr) __attribute__((always_inline)) {
            constexpr bool WITH_PS = decltype(ps_tag)::value, RES = decltype(res_tag)::value;
            CGNN_W8_STAMP(0);
            bias_rowsw<DT>(acc, ring.vec_addr(), h);
            CGNN_W8_STAMP(1);
            // the next pass's P rows are requested one instruction per MFMA slot (a burst of nine loads per wave fills the
            // memory pipeline's queues: the waves dispatched second then wait thousands of cycles to issue theirs)
            wblockw<DT, KS>(acc, inb[NH & 1], ring.base(), make_fill([&](auto qc) __attribute__((always_inline)) {
                constexpr int q = decltype(qc)::value;
                if constexpr (WITH_PS && q < 2 * DT) ps[q] = load_p_piece(nps, so + 16u * (unsigned)q);
                if constexpr (WITH_PS && PDB && q == 2 * DT) pdchunk = load_pd_chunk(npd, dchunk);
                if constexpr (q > 2 * DT) pieces_at(CGNN_IC(MQ - 2 * DT - 1), CGNN_IC(q - 2 * DT - 1));
            }));
            CGNN_W8_STAMP(2);
            // LayerNorm: statistics and the first KA slices here, under the partner wave's matrix work of this interval; the
            // rest behind the barrier (balances the two intervals' vector work when the partner runs a layer behind)
            lnst = ln_stats_w<DT>(acc);
            ln_affine_w<RES, DT, 0, KA>(acc, ev, inb[0], ring.vec_addr() + (unsigned)D * 4u, h, lnst);
            // the LayerNorm vectors of this chunk, for the slices that run in the next interval (when this slot may
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
            ring.template interval_end<(WITH_PS ? 2 * DT + (PDB ? 1 : 0) : 0), 3>();
            ring.advance();
        };
        auto hidden_steps = [&](int rr) __attribute__((always_inline)) {
            static_for_each([&](auto lc) __attribute__((always_inline)) {
                step_hidden(std::integral_constant<int, decltype(lc)::value + 1>{}, rr);
            }, std::make_integer_sequence<int, NH - 1>{});
        };

        if (ENC) {
            step_first(std::true_type{}, CGNN_IC(0), 0);
            hidden_steps(-1);
            step_last(std::true_type{}, std::false_type{}, ps_all, pd_all, -1);
            step_first(std::false_type{}, CGNN_IC(1), 0);
            hidden_steps(0);
        } else {
            step_first(std::false_type{}, CGNN_IC(0), 0);
            hidden_steps(0);
        }
        for (int rr = 1; rr < L; ++rr) {
            step_last(std::true_type{}, std::true_type{}, ps_all + (int64_t)rr * round_stride, pd_all + (int64_t)rr * round_stride, rr);
            step_first(std::false_type{}, CGNN_IC(2), rr);
            hidden_steps(rr);
        }
        step_last(std::false_type{}, std::true_type{}, ps_all, pd_all, -1);
        ln_affine_w<true, DT, KA, 2 * DT>(acc, ev, inb[0], ring.lnbuf(lnpar ^ 1), h, lnst);

        if (valid) store_tile<DT>(ev, e_out + tile * (32 * D), lane);
        const int64_t tn = tile + tr.stride;
        if (tn < tr.end) {
            tile = tn;
        } else {
            valid = false;
        }
    }
    if (LAG != 0 && !lagging) ring.idle_interval();
#ifdef CGNN_W8_STAMPS
    if (blockIdx.x == 8 && lane == 0)
        for (int k = 0; k < 20; ++k) cgnn_w8_stamps[wave * 32 + k] = tm.sum[k];
#endif
    CGNN_S32_VMCNT(0);      // the ring's last refills (unread) must have landed before the workgroup's LDS is released
    __builtin_amdgcn_s_barrier();
}

template <int DT, int NH, int LAG, bool PDB>
static int launch_w8(const S32Args& a, const __bf16* ps, const __bf16* pd, int64_t round_stride, const int32_t* src,
                     const int32_t* dst, int64_t num_edges, const float* e_in, float* e_out, const float* attr, int ld_attr,
                     int seg_k, hipStream_t st) {
    typedef W8Geom<DT> W;
    const bool enc = a.enc_in_dim > 0;
    auto kern = enc ? edge_stream32w_kernel<DT, NH, true, LAG, PDB> : edge_stream32w_kernel<DT, NH, false, LAG, PDB>;
    static bool attr_set[2][16] = {};      // per (kernel, device): the attribute is sticky, setting it costs a driver call per launch
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (W::LDS > 48 * 1024 && !attr_set[enc][dev]) {
        int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)W::LDS),
                           "hipFuncSetAttribute(edge_stream32w)");
        if (rc != CGNN_OK) return rc;
        attr_set[enc][dev] = true;
    }
    const int64_t tiles = (num_edges + 31) / 32;
    const int grid = grid_for_tiles(tiles, 1, CGNN_W8_WAVES);
    kern<<<grid, CGNN_W8_BLOCK, W::LDS, st>>>(a, ps, pd, round_stride, src, dst, num_edges, e_in, e_out, attr, ld_attr, seg_k);
#ifdef CGNN_W8_STAMPS
    {
        static int printed = 0;
        hipStreamSynchronize(st);
        if (printed++ == 2) {
            static unsigned long long hs[8 * 32];
            hipMemcpyFromSymbol(hs, HIP_SYMBOL(cgnn_w8_stamps), sizeof(hs));
            // a stamp adds the cycles since the previous stamp to its slot: slot k = the phase that ENDS at stamp k
            const char* names[20] = {"(gap)", "last: bias, P requests", "last: MFMAs", "last: LayerNorm part, copy", "last: vmcnt wait",
                                     "last: barrier", "", "", "(gap; tile ends)", "first: LayerNorm rest, Pd", "first: selector MFMAs",
                                     "first: L0 + pack", "first: vmcnt wait", "first: barrier", "(gap)", "hidden: bias, MFMAs, pack",
                                     "hidden: vmcnt wait", "hidden: barrier / idle interval", "idle: vmcnt", "idle: barrier"};
            const double passes = (double)((num_edges + 31) / 32) / (grid * 8.0) * (a.rounds + (enc ? 1 : 0));
            printf("lag %d: cycles per pass and wave, by phase (sums over the wave's whole run / %.0f passes)\n", LAG, passes);
            double tot[8] = {0};
            for (int k = 0; k < 20; ++k) {
                bool any = false;
                for (int w = 0; w < 8; ++w) any |= hs[w * 32 + k] != 0;
                if (!any) continue;
                printf("  %2d %-28s", k, names[k]);
                for (int w = 0; w < 8; ++w) {
                    printf(" %7.0f", hs[w * 32 + k] / passes);
                    tot[w] += hs[w * 32 + k] / passes;
                }
                printf("\n");
            }
            printf("     %-28s", "total");
            for (int w = 0; w < 8; ++w) printf(" %7.0f", tot[w]);
            printf("\n");
        }
    }
#endif
    return check_hip(hipGetLastError(), "cgnn_edge_stream_run_w8 launch");
}

}  // namespace cgnn

using namespace cgnn;

extern "C" int cgnn_edge_stream_w8_supported(int32_t latent, int32_t num_hidden_layers) {
    return (latent == 128 && num_hidden_layers >= 1 && num_hidden_layers <= 3) ? 1 : 0;
}

extern "C" int cgnn_edge_stream_run_w8(const void* image, size_t image_bytes, int32_t latent, int32_t num_hidden_layers,
                                       int32_t num_rounds, int32_t enc_in_dim, const void* ps_all, const void* pd_all,
                                       int64_t round_stride, const int32_t* src, const int32_t* dst, int64_t num_edges,
                                       const float* e_in, float* e_out, const float* edge_attr, int32_t ld_attr, int32_t lag,
                                       int32_t fixed_k, void* stream) {
    if (!image || !ps_all || !pd_all || !src || !dst || !e_out || num_edges < 0 || num_rounds < 1 || num_hidden_layers < 1 ||
        round_stride < 0 || (enc_in_dim > 0 ? (!edge_attr || ld_attr < enc_in_dim) : !e_in) || lag < 0 || lag > 1 || fixed_k < 0) {
        set_error("cgnn_edge_stream_run_w8: invalid argument");
        return CGNN_ERR_INVALID_ARG;
    }
    if (enc_in_dim > 16) {
        set_error("cgnn_edge_stream_run_w8: the in-launch encoder takes at most 16 edge features (got %d)", enc_in_dim);
        return CGNN_ERR_UNSUPPORTED;
    }
    if (!cgnn_edge_stream_w8_supported(latent, num_hidden_layers)) {
        set_error("cgnn_edge_stream_run_w8: no kernel for latent=%d with %d hidden layers (built for latent 128, 1..3 hidden "
                  "layers of the same width)", latent, num_hidden_layers);
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
    // receiver rows by broadcast where a tile of 32 edges holds whole receivers (or one receiver holds whole tiles)
    const bool pdb = fixed_k > 0 && num_edges % fixed_k == 0 && (fixed_k <= 32 ? (32 % fixed_k == 0 && fixed_k >= 8) : fixed_k % 32 == 0);
    const int seg_k = pdb ? fixed_k : 0;
#define CGNN_W8_GO(NHx, LAGx)                                                                                              \
    return pdb ? launch_w8<4, NHx, LAGx, true>(a, (const __bf16*)ps_all, (const __bf16*)pd_all, round_stride, src, dst,     \
                                               num_edges, e_in, e_out, edge_attr, ld_attr, seg_k, st)                      \
               : launch_w8<4, NHx, LAGx, false>(a, (const __bf16*)ps_all, (const __bf16*)pd_all, round_stride, src, dst,    \
                                                num_edges, e_in, e_out, edge_attr, ld_attr, seg_k, st)
    if (lag) {
        switch (num_hidden_layers) {
            case 1: CGNN_W8_GO(1, 1);
            case 2: CGNN_W8_GO(2, 1);
            default: CGNN_W8_GO(3, 1);
        }
    }
    switch (num_hidden_layers) {
        case 1: CGNN_W8_GO(1, 0);
        case 2: CGNN_W8_GO(2, 0);
        default: CGNN_W8_GO(3, 0);
    }
#undef CGNN_W8_GO
}
