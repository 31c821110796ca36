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

#include "s32.hpp"

namespace cgnn {

// ---- the ring -------------------------------------------------------------------------------------------------------
// Four slots.  Step s computes out of slot s % 4.  Between the two tiles' blocks of step s every wave waits for its
// own pieces of chunk s + 1 and meets the others at a barrier: chunk s + 1 is then complete for everybody, and
// everybody has left step s - 1, whose slot is refilled with chunk s + 3 (pieces handed out between the MFMAs that
// follow).  A chunk is requested two and a half steps before its first use.
template <class G>
struct Ring32 {
    const char* image;
    int count;           // chunks per tile pair
    int wave, lane;
    int slot;            // of the current step
    int dma_chunk, dma_slot;
    bool primed = false;
    __device__ __forceinline__ Ring32(const char* img, int cnt, int w, int l)
        : image(img), count(cnt), wave(w), lane(l), slot(0), dma_chunk(0), dma_slot(0) {}
    __device__ __forceinline__ unsigned lds0() const { return (unsigned)(uintptr_t)(LdsWeightPtr)(cgnn_smem); }
    __device__ __forceinline__ unsigned base() const { return lds0() + (unsigned)slot * G::STRIDE; }
    __device__ __forceinline__ unsigned base_next() const {
        return lds0() + (unsigned)((slot + 1) & (CGNN_S32_SLOTS - 1)) * G::STRIDE;
    }
    // LDS byte address of the vector block (bias, gamma, beta) of the chunk `ahead` steps ahead of the current one
    __device__ __forceinline__ unsigned vec_addr(int ahead) const {
        return lds0() + (unsigned)((slot + ahead) & (CGNN_S32_SLOTS - 1)) * G::STRIDE + G::VEC_OFF;
    }
    // The refill never stops: past the last step it copies chunks nobody will read into slots nobody reads any more
    // (always the slot of the step just left), which keeps every wait count a constant and the loop free of branches;
    // the kernel drains them before it ends.
    __device__ __forceinline__ void piece(int i) {
        const unsigned off = (unsigned)(wave + CGNN_S32_WAVES * i) * 1024u;
        char* dst = cgnn_smem + (unsigned)dma_slot * G::STRIDE + off;
        asm volatile("" ::: "memory");
#if defined(__HIP_DEVICE_COMPILE__)
        // Buffer form of the LDS-DMA instruction (buffer_load_dwordx4 ... offen lds): descriptor of the image + the
        // piece's byte offset as a SCALAR + one constant lane offset, so a piece costs no vector instruction.  The
        // global form needs a 64-bit vector add per piece (27 pieces per pass): 0.3 ms of this kernel (A/B on one box:
        // 20.71 -> 20.39 ms).  An earlier attempt at a scalar base through inline asm (s_mov m0 + global_load_lds with an
        // SGPR pair) was 0.1-0.5 ms SLOWER: an asm statement fences the scheduler, this builtin does not.
        // (The host pass of hipcc does not know the buffer builtin: it sees the global form below, which never runs.)
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(image), 0, (int)((unsigned)count * G::STRIDE), 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LdsVoidPtrG)dst, 16, (unsigned)lane * 16u,
                                                 (unsigned)dma_chunk * G::STRIDE + off, 0, 0);
#else
        const char* src = image + (size_t)dma_chunk * G::STRIDE + off + lane * 16;
        __builtin_amdgcn_global_load_lds((GlobalVoidPtrG)src, (LdsVoidPtrG)dst, 16, 0, 0);
#endif
        asm volatile("" ::: "memory");
    }
    __device__ __forceinline__ void dma_done() {
        dma_chunk = dma_chunk + 1 == count ? 0 : dma_chunk + 1;
        dma_slot = (dma_slot + 1) & (CGNN_S32_SLOTS - 1);
    }
    __device__ __forceinline__ void prime() {
        for (int c = 0; c < CGNN_S32_SLOTS - 1; ++c) {
            for (int i = 0; i < G::NP; ++i) piece(i);
            dma_done();
        }
        CGNN_S32_VMCNT(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        primed = true;
    }
    // `wait for everything but the newest N operations`, N = the pieces of the chunk two steps ahead, which were issued
    // after the ones we need (N more if the caller has issued EXTRA operations of its own since)
    template <int EXTRA>
    __device__ __forceinline__ void wait_next_chunk() const {
        vm_wait_const<G::NP + EXTRA>();
    }
    template <int EXTRA>
    __device__ __forceinline__ void sync_next() {
        wait_next_chunk<EXTRA>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    __device__ __forceinline__ void advance() { slot = (slot + 1) & (CGNN_S32_SLOTS - 1); }
};

// ---- per-tile registers ---------------------------------------------------------------------------------------------
// 32 edges in the act layout of cgnn_common.hpp: edge on the MFMA column (lane & 31), for 32-feature tile t register i
// of lane (r, h) holds feature 32 t + 8 (i >> 2) + 4 h + (i & 3).
// Values parked in accumulation registers: only these helpers touch them.  Four at a time, because hipcc pads every
// inline-asm statement whose outputs a vector instruction reads next with a wait state (one s_nop per statement).
// (Leaving the moves to the compiler -- an empty asm with a "+a" operand -- does not work under pressure: the register
// allocator then keeps the values architectural and spills them.  And do not hand a vector ELEMENT straight to
// __builtin_bit_cast: clang reads element 0.)
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
__device__ __forceinline__ void acc_put4(unsigned& a0, unsigned& a1, unsigned& a2, unsigned& a3, float v0, float v1, float v2,
                                         float v3) {
    asm("v_accvgpr_write_b32 %0, %4\n\tv_accvgpr_write_b32 %1, %5\n\tv_accvgpr_write_b32 %2, %6\n\tv_accvgpr_write_b32 %3, %7"
        : "=a"(a0), "=a"(a1), "=a"(a2), "=a"(a3)
        : "v"(v0), "v"(v1), "v"(v2), "v"(v3));
}
__device__ __forceinline__ void acc_get4(float& v0, float& v1, float& v2, float& v3, unsigned a0, unsigned a1, unsigned a2,
                                         unsigned a3) {
    asm("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\tv_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7"
        : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3)
        : "a"(a0), "a"(a1), "a"(a2), "a"(a3));
}
template <int DT>
struct Tile32 {
    unsigned evp[DT][16]; // f32 latent (the residual stream), parked in accumulation registers
    f32x16 acc[DT];       // accumulators of the layer in flight, then its output
    bf16x8 in[2 * DT];    // the layer's input operand: k-step 2 t + s = features 32 t + 16 s .. + 15 (fragment k order)
    float part[DT];       // LayerNorm partial sums
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

// acc[o] += W[32 o .. 32 o + 31, :] . in ; fragment m = o * KS + ks (1 KiB, lane-linear) at addr + m * 1024.  The LDS
// reads and their counted waits are written by hand (hipcc waits lgkmcnt(0) before every group otherwise); PD groups
// of four fragments are in flight ahead of the MFMAs.  Rows are finished one after the other (o outermost).
template <int NROW, int KS, class Fill>
__device__ __forceinline__ void wblock32(f32x16 (&acc)[NROW], const bf16x8 (&in)[KS], unsigned addr, const Fill& fill) {
    typedef WBlock<NROW, KS> WB;
    constexpr int M = WB::M, GS = WB::GS, NG = WB::NG, PD = WB::PD, NBUF = PD + 1;
    static_assert(M % GS == 0 && (GS == 4 || GS == 2 || GS == 1), "groups of four (two, one) fragments");
    // issue index t -> fragment: rows in pairs, the two rows of a pair alternating (consecutive MFMAs then write
    // different accumulators), or plainly row after row
#define CGNN_S32_FRAG(t) (t)
    const unsigned a = addr + (unsigned)(threadIdx.x & 63) * 16u;
    u32x4 buf[NBUF][GS];
    static_for_each([&](auto pc) __attribute__((always_inline)) {
        constexpr int p = decltype(pc)::value;
        if constexpr (p < NG) {
            static_for_each([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                buf[p][j] = lds_read_b128_acc<CGNN_S32_FRAG(p * GS + j) * 1024>(a);
            }, std::make_integer_sequence<int, GS>{});
        }
    }, std::make_integer_sequence<int, PD>{});
    static_for_each([&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value;
        // fragments of groups <= g + PD - 1 have been requested (one read behind every MFMA: a burst of four at the
        // top of a group queues behind the other waves' bursts at the LDS)
        constexpr int newer = ((g + PD - 1 < NG ? g + PD - 1 : NG - 1) - g) * GS;     // fragment reads issued after group g's
        if constexpr (GS == 4)
            lds_wait4i<newer>(buf[g % NBUF][0], buf[g % NBUF][1], buf[g % NBUF][2], buf[g % NBUF][3]);
        else if constexpr (GS == 2)
            lds_wait2i<newer>(buf[g % NBUF][0], buf[g % NBUF][1]);
        else
            lds_wait1i<newer>(buf[g % NBUF][0]);
        static_for_each([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value, m = g * GS + j, fm = CGNN_S32_FRAG(m), o = fm / KS, ks = fm % KS;
            acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, buf[g % NBUF][j]), in[ks], acc[o], 0, 0,
                                                             0);
            if constexpr (g + PD < NG) buf[(g + PD) % NBUF][j] = lds_read_b128_acc<CGNN_S32_FRAG((g + PD) * GS + j) * 1024>(a);
            fill.template run<m>();
            __builtin_amdgcn_sched_barrier(0);
        }, std::make_integer_sequence<int, GS>{});
    }, std::make_integer_sequence<int, NG>{});
#undef CGNN_S32_FRAG
}

// The same block with its fragment pipeline running ACROSS block boundaries.  wblock32 starts every block by requesting
// its first PD groups and waiting for the first of them: an LDS round trip with nothing in flight, six times per pass.
// Here the caller owns the fragment buffers; a block finds its first PD groups already requested (by the block before
// it, or wprefetch32 at the start of the kernel) and requests the first PD groups of the NEXT block behind its own last
// MFMAs -- the next block reads either the same chunk (the other tile) or the next one, which the step's barrier has
// already confirmed.  One fragment read behind every MFMA, always; every wait leaves (PD - 1) groups in flight.
// Group p of a block lives in buf[p % (PD + 1)]: the next block's group g + PD - NG lands in the slot group g has just
// been read from, which is the right one when (NG - PD) is a multiple of PD + 1 (32 / 4 = 8 groups and the encoder's 2).
typedef u32x4 WBuf32[CGNN_S32_PD + 1][4];
__device__ __forceinline__ void wprefetch32(WBuf32& buf, unsigned addr) {
    const unsigned a = addr + (unsigned)(threadIdx.x & 63) * 16u;
    static_for_each([&](auto pc) __attribute__((always_inline)) {
        constexpr int p = decltype(pc)::value;
        static_for_each([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            buf[p][j] = lds_read_b128_acc<(p * 4 + j) * 1024>(a);
        }, std::make_integer_sequence<int, 4>{});
    }, std::make_integer_sequence<int, CGNN_S32_PD>{});
}
template <int NROW, int KS, class Fill>
__device__ __forceinline__ void wblock32p(f32x16 (&acc)[NROW], const bf16x8 (&in)[KS], unsigned addr, unsigned next_addr,
                                          WBuf32& buf, const Fill& fill) {
    constexpr int M = NROW * KS, GS = 4, NG = M / GS, PD = CGNN_S32_PD, NBUF = PD + 1;
    static_assert(M % GS == 0 && NG >= PD && (NG - PD) % NBUF == 0, "block shape does not close the buffer rotation");
    const unsigned a = addr + (unsigned)(threadIdx.x & 63) * 16u, an = next_addr + (unsigned)(threadIdx.x & 63) * 16u;
    static_for_each([&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value;
        lds_wait4i<(PD - 1) * GS>(buf[g % NBUF][0], buf[g % NBUF][1], buf[g % NBUF][2], buf[g % NBUF][3]);
        static_for_each([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value, m = g * GS + j, o = m / KS, ks = m % KS;
            acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, buf[g % NBUF][j]), in[ks], acc[o], 0, 0,
                                                             0);
            if constexpr (g + PD < NG)
                buf[(g + PD) % NBUF][j] = lds_read_b128_acc<((g + PD) * GS + j) * 1024>(a);
            else
                buf[(g + PD - NG) % NBUF][j] = lds_read_b128_acc<((g + PD - NG) * GS + j) * 1024>(an);
            fill.template run<m>();
            __builtin_amdgcn_sched_barrier(0);
        }, std::make_integer_sequence<int, GS>{});
    }, std::make_integer_sequence<int, NG>{});
}

// P-row loads from inline asm (the compiler would guard registers loaded across the ring's LDS-DMA with vmcnt(0)),
// straight into accumulation registers; 32-bit lane offset + uniform 64-bit table base.  The caller waits (counted)
// and then calls p32_ready().
template <int IDX>
__device__ __forceinline__ bf16x8 p32_load(unsigned off, const __bf16* table) {     // off = (row * H + h * (H / 2)) * 2
    u32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=a"(r) : "v"(off), "s"(table), "n"(IDX * 16));
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
template <bool RELU, int DT, int S0, int S1>
__device__ __forceinline__ void pack32_run(Tile32<DT>& X) {
    static_for_each([&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value + S0;
        if constexpr (u < S1) pack32_slice<RELU, DT, u>(X);
    }, std::make_integer_sequence<int, (S1 > S0 ? S1 - S0 : 0)>{});
}

// LayerNorm (eps 1e-5, biased variance, affine) of X.acc over the 32 DT features of each edge, then
//   RES:  ev += y (graph_network.py:182)      !RES: ev = y (the encoder, :57)
// and the bf16 pack of the new ev as the next layer-0 operand.  Slices:
//   [0, DT)            partial sums per feature tile                 DT                 total, mean
//   [DT + 1, 2 DT + 1) centre, partial sums of squares               2 DT + 1           variance, rstd
//   [2 DT + 2, 4 DT + 2) affine (+ residual) for eight values (t, s): reads and rewrites the parked latent and packs
//                        the new values as k-step (t, s) of the next layer-0 operand
// The arithmetic runs on register pairs (v_pk_add / v_pk_mul / v_pk_fma_f32) in short independent chains: with one
// wave per SIMD every instruction costs an issue slot of four cycles and a dependent one waits for its producer.
template <int DT>
struct Ln32 {
    static constexpr int NS = 4 * DT + 2;
    static constexpr int AFF0 = 2 * DT + 2;      // first affine slice
    static constexpr int NAFF = 2 * DT;
};
// gamma / beta of the affine slices come through two register sets filled by hand-issued LDS reads, one slice ahead
// (left to the compiler, all reads of a LayerNorm are hoisted to its top: 128 registers at latent 128, spills).
struct LnVec32 {
    u32x4 g[2][2], b[2][2];      // [set][feature group 2 s, 2 s + 1]
};
template <int IMM>
__device__ __forceinline__ void ln32_vec_read(u32x4& g0, u32x4& g1, u32x4& b0, u32x4& b1, unsigned ga, unsigned ba) {
    asm volatile("ds_read_b128 %0, %4 offset:%6\n\tds_read_b128 %1, %4 offset:%7\n\t"
                 "ds_read_b128 %2, %5 offset:%6\n\tds_read_b128 %3, %5 offset:%7"
                 : "=&v"(g0), "=&v"(g1), "=&v"(b0), "=&v"(b1)
                 : "v"(ga), "v"(ba), "n"(IMM), "n"(IMM + 32));
}
template <int NEWER>
__device__ __forceinline__ void ln32_vec_wait(u32x4& g0, u32x4& g1, u32x4& b0, u32x4& b1) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(g0), "+v"(g1), "+v"(b0), "+v"(b1) : "n"(NEWER > 11 ? 11 : NEWER));
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
// NEWER: LDS operations this wave has issued since the PREVIOUS slice ran (the weight-fragment reads behind the MFMAs
// in between): the affine slice waits for its own vectors and leaves those in flight.  A smaller number than the true
// one only waits longer.
template <bool RES, int DT, int U, int NEWER>
__device__ __forceinline__ void ln32_slice(Tile32<DT>& X, LnVec32& V, unsigned ga, unsigned ba) {
    constexpr int D = 32 * DT;
    typedef Ln32<DT> LN;
    if constexpr (U < DT) {
        const f32x16& v = X.acc[U];
        const f32x2 a0 = f32x2{v[0], v[1]} + f32x2{v[2], v[3]}, a1 = f32x2{v[4], v[5]} + f32x2{v[6], v[7]};
        const f32x2 a2 = f32x2{v[8], v[9]} + f32x2{v[10], v[11]}, a3 = f32x2{v[12], v[13]} + f32x2{v[14], v[15]};
        const f32x2 s = (a0 + a1) + (a2 + a3);
        X.part[U] = s[0] + s[1];
    } else if constexpr (U == DT) {
        float s = X.part[0];
#pragma unroll
        for (int t = 1; t < DT; ++t) s += X.part[t];
        X.mean = half_swap_sum(s) * (1.0f / D);
    } else if constexpr (U < 2 * DT + 1) {
        constexpr int t = U - DT - 1;
        f32x16& v = X.acc[t];
        const f32x2 m2 = {X.mean, X.mean};
        f32x2 q[4] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            const f32x2 d = f32x2{v[i], v[i + 1]} - m2;
            v[i] = d[0];
            v[i + 1] = d[1];
            q[(i >> 1) & 3] = __builtin_elementwise_fma(d, d, q[(i >> 1) & 3]);
        }
        const f32x2 qq = (q[0] + q[1]) + (q[2] + q[3]);
        X.part[t] = qq[0] + qq[1];
    } else if constexpr (U == 2 * DT + 1) {
        float q = X.part[0];
#pragma unroll
        for (int t = 1; t < DT; ++t) q += X.part[t];
        X.rstd = __builtin_amdgcn_rsqf(half_swap_sum(q) * (1.0f / D) + 1e-5f);
        ln32_vec_read<0>(V.g[0][0], V.g[0][1], V.b[0][0], V.b[0][1], ga, ba);      // vectors of the first affine slice
    } else {
        constexpr int k = U - LN::AFF0, t = k >> 1, sx = k & 1, cur = k & 1, nxt = cur ^ 1;
        if constexpr (k + 1 < LN::NAFF) {
            constexpr int t1 = (k + 1) >> 1, s1 = (k + 1) & 1;
            ln32_vec_read<(32 * t1 + 16 * s1) * 4>(V.g[nxt][0], V.g[nxt][1], V.b[nxt][0], V.b[nxt][1], ga, ba);
            ln32_vec_wait<NEWER + 4>(V.g[cur][0], V.g[cur][1], V.b[cur][0], V.b[cur][1]);
        } else {
            ln32_vec_wait<NEWER>(V.g[cur][0], V.g[cur][1], V.b[cur][0], V.b[cur][1]);
        }
        const f32x2 r2 = {X.rstd, X.rstd};
        f32x2 e[4];
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
            const f32x4 gm = __builtin_bit_cast(f32x4, V.g[cur][gg]);
            const f32x4 bt = __builtin_bit_cast(f32x4, V.b[cur][gg]);
            constexpr int i0 = 8 * sx;
            const int i = i0 + 4 * gg;
            f32x2 base[2] = {f32x2{bt[0], bt[1]}, f32x2{bt[2], bt[3]}};
            if (RES) {
                float p0, p1, p2, p3;
                acc_get4(p0, p1, p2, p3, X.evp[t][i], X.evp[t][i + 1], X.evp[t][i + 2], X.evp[t][i + 3]);
                base[0] += f32x2{p0, p1};
                base[1] += f32x2{p2, p3};
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const f32x2 w = f32x2{gm[2 * c], gm[2 * c + 1]} * r2;
                const f32x2 d = {X.acc[t][i + 2 * c], X.acc[t][i + 2 * c + 1]};
                e[2 * gg + c] = __builtin_elementwise_fma(d, w, base[c]);
            }
            acc_put4(X.evp[t][i], X.evp[t][i + 1], X.evp[t][i + 2], X.evp[t][i + 3], e[2 * gg][0], e[2 * gg][1],
                     e[2 * gg + 1][0], e[2 * gg + 1][1]);
        }
        u32x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = pack_bf16(e[c][0], e[c][1]);
        X.in[2 * t + sx] = __builtin_bit_cast(bf16x8, v);
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

// acc = bias (rows [T0, T1) of the tile); plain LDS loads, placed by the caller where their latency is covered
template <int DT, int T0, int T1>
__device__ __forceinline__ void bias_rows32(f32x16 (&acc)[DT], unsigned vec_addr, int h) {
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

#ifdef CGNN_S32_STAMPS   // developer build: s_memtime stamps of one workgroup's waves over one pass (printed by the launcher)
__device__ unsigned long long cgnn_s32_stamps[4 * 64];
#define CGNN_S32_STAMP(k)                                                                                  \
    if (stamp_on && lane == 0) {                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        cgnn_s32_stamps[wave * 64 + (k)] = __builtin_readcyclecounter();                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
    }
#else
#define CGNN_S32_STAMP(k)
#endif

// ---- the kernel -----------------------------------------------------------------------------------------------------
// Schedule of one pass (= the encoder, or one round) over the wave's tile pair (A, B), nh = 2; "X.Ll" = the MFMAs of
// layer l for tile X, "| ..." = what fills their gaps:
//   step 0   A.P (selector MFMAs)  | LayerNorm(B) of the PREVIOUS pass, first part
//            A.L0                  | LayerNorm(B) rest; requests for B's P rows
//            -- barrier (next layer complete; refill of the slot the last step left) --
//            B.P, B.L0             | refill pieces; pack(A.L0); bias of A.L1
//   step 1   A.L1                  | pack(B.L0); bias of B.L1
//            -- barrier --
//            B.L1                  | refill pieces; pack(A.L1); bias of A.L2
//   step 2   A.L2                  | pack(B.L1); bias of B.L2; requests for A's P rows of the next round
//            -- barrier --
//            B.L2                  | refill pieces; LayerNorm(A)
// so every vector job runs under the other tile's MFMAs, and B's LayerNorm under the next pass's first blocks.
template <int DT, bool ENC>
__global__ __launch_bounds__(CGNN_S32_BLOCK, 1) void edge_stream32_kernel(
    S32Args a, const __bf16* __restrict__ ps_all, const __bf16* __restrict__ pd_all, int64_t round_stride,
    const int32_t* __restrict__ src, const int32_t* __restrict__ dst, int64_t num_edges, const float* e_in, float* e_out,
    const float* __restrict__ attr, int ld_attr) {
    typedef S32Geom<DT> G;
    constexpr int D = G::D, KS = G::KS;
    typedef WBlock<DT, KS> WB;
    constexpr int MQ = WB::M;                  // MFMAs (slots) of a full layer block
    constexpr int PQ = 4 * DT;                 // MFMAs (slots) of a P block
    constexpr int NLN = Ln32<DT>::NS, NPACK = 2 * DT, NPL = 4 * DT;
    // a block's pack of the other tile ends at slot QP; the other tile's next bias is requested at slot QB, several
    // MFMAs before that tile's block begins
    constexpr int QP = MQ >= 16 ? MQ - 12 : (MQ > 4 ? MQ - 4 : MQ), QB = MQ >= 16 ? MQ - 10 : MQ - 1;
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
    Ring32<G> ring(a.image, steps_per_pair, wave, lane);
    ring.prime();
    // latent 128: the LDS fragment pipeline runs across block boundaries (wblock32p); its buffers live here
    constexpr bool XBLOCK = DT == 4;
    WBuf32 wbuf;
    if constexpr (XBLOCK) wprefetch32(wbuf, ring.base());
#define CGNN_S32_WB(KSx, ACC, IN, BASE, NEXT, ...)                                  \
    if constexpr (XBLOCK)                                                           \
        wblock32p<DT, KSx>(ACC, IN, BASE, NEXT, wbuf, __VA_ARGS__);                 \
    else                                                                            \
        wblock32<DT, KSx>(ACC, IN, BASE, __VA_ARGS__)
    const bf16x8 sel0 = p32_selector(lane, 0), sel1 = p32_selector(lane, 1);

    Tile32<DT> A, B;
    bf16x8 ps[2 * DT], pd[2 * DT];
    LnVec32 V;
    int64_t pair = tr.first < tr.end ? tr.first : tr.end - 1;
    bool valid = tr.first < tr.end;

    // P-row request i (of 4 DT): table (i & 1), 16-byte piece i >> 1; `so` / `dof` = byte offsets of the lane's row halves
    auto p_issue = [&](const __bf16* tps, const __bf16* tpd, unsigned so, unsigned dof, auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value, pc = i >> 1;
        if constexpr ((i & 1) == 0)
            ps[pc] = p32_load<pc>(so, tps);
        else
            pd[pc] = p32_load<pc>(dof, tpd);
    };
    auto p_issue_range = [&](const __bf16* tps, const __bf16* tpd, unsigned so, unsigned dof, auto lo_c, auto hi_c)
                             __attribute__((always_inline)) {
        constexpr int lo = decltype(lo_c)::value, hi = decltype(hi_c)::value;
        static_for_each([&](auto kc) __attribute__((always_inline)) {
            constexpr int i = decltype(kc)::value + lo;
            if constexpr (i < hi) p_issue(tps, tpd, so, dof, std::integral_constant<int, i>{});
        }, std::make_integer_sequence<int, (hi > lo ? hi - lo : 0)>{});
    };
#define CGNN_IC(x) std::integral_constant<int, (x)> {}

    for (int it = 0; it < iters; ++it) {
        // ---- this pair's tiles, edges and inputs (compiler-tracked loads: once per pair) ---------------------------
        const int64_t tA = 2 * pair, tB = (2 * pair + 1 < tiles) ? 2 * pair + 1 : 2 * pair;
        const bool validB = valid && (2 * pair + 1 < tiles);
        unsigned soA, doA, soB, doB;      // byte offsets of this lane's halves of the sender / receiver P rows
        {
            const int64_t eA = tA * 32 + r, eB = tB * 32 + r;
            const int64_t ca = eA < num_edges ? eA : num_edges - 1, cb = eB < num_edges ? eB : num_edges - 1;
            soA = ((unsigned)src[ca] * (unsigned)D + (unsigned)h * (D / 2)) * 2u;
            doA = ((unsigned)dst[ca] * (unsigned)D + (unsigned)h * (D / 2)) * 2u;
            soB = ((unsigned)src[cb] * (unsigned)D + (unsigned)h * (D / 2)) * 2u;
            doB = ((unsigned)dst[cb] * (unsigned)D + (unsigned)h * (D / 2)) * 2u;
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
        // PEND: tile B's LayerNorm of the previous pass is still to do (0 none, 1 the encoder's, 2 a round's); its
        // vectors sit in the ring slot of the previous step.  `pa_pending`: A's P rows were requested by that pass.
        auto pass = [&](auto enc_tag, auto pend_tag, int rr, bool pa_pending) __attribute__((always_inline)) {
            constexpr bool IS_ENC = decltype(enc_tag)::value;
#ifdef CGNN_S32_STAMPS
            const bool stamp_on = blockIdx.x == 8 && it == 2 && rr == 3 && !IS_ENC;
#endif
            CGNN_S32_STAMP(0);
            constexpr int PEND = decltype(pend_tag)::value;
            constexpr int KS0 = IS_ENC ? 2 : KS;     // the encoder's first Linear: K padded to one 32-wide k tile
            typedef WBlock<DT, KS0> WB0;
            constexpr int MQ0 = WB0::M;
            const __bf16* tps = ps_all + (int64_t)rr * round_stride;
            const __bf16* tpd = pd_all + (int64_t)rr * round_stride;
            const unsigned pga = ring.vec_addr(-1) + (unsigned)D * 4u + 16u * (unsigned)h, pba = pga + (unsigned)D * 4u;
            // ---------------- layer 0 ----------------
            {
                const unsigned base = ring.base();
                if constexpr (IS_ENC) {
                    bias_rows32<DT, 0, DT>(A.acc, ring.vec_addr(0), h);
                    const bf16x8 (&inA)[2] = reinterpret_cast<const bf16x8(&)[2]>(A.in[0]);
                    CGNN_S32_WB(2, A.acc, inA, base, base, make_fill([&](auto qc) __attribute__((always_inline)) {
                        constexpr int q = decltype(qc)::value;
                        if constexpr (q == MQ0 - 1) bias_rows32<DT, 0, DT>(B.acc, ring.vec_addr(0), h);
                    }));
                } else {
                    // A's P rows were requested in the previous pass, before its last refill pieces; the first pass
                    // of a pair asks for its own (the previous pair's request was for other edges)
                    if (pa_pending) {
                        ring.template wait_next_chunk<0>();
                    } else {
                        p_issue_range(tps, tpd, soA, doA, CGNN_IC(0), CGNN_IC(NPL));
                        CGNN_S32_VMCNT(0);
                    }
                    p32_ready<DT>(ps, pd);
                    constexpr int S1 = PEND ? NLN * PQ / (PQ + MQ) : 0;     // LayerNorm slices under the P block
                    // B's P rows are requested as soon as A's selector MFMAs of a feature tile have read the registers
                    // (row t's four pieces behind its fourth MFMA): a whole block ahead of their use
                    selp32<DT>(A.acc, ps, pd, sel0, sel1, make_fill([&](auto qc) __attribute__((always_inline)) {
                        constexpr int q = decltype(qc)::value;
                        if constexpr (PEND != 0)
                            ln32_run<PEND == 2, DT, share_lo(S1, PQ, q), share_hi(S1, PQ, q), 0>(B, V, pga, pba);
                        if constexpr ((q & 3) == 3) p_issue_range(tps, tpd, soB, doB, CGNN_IC(q - 3), CGNN_IC(q + 1));
                    }));
                    CGNN_S32_STAMP(1);
                    CGNN_S32_WB(KS, A.acc, A.in, base, base, make_fill([&](auto qc) __attribute__((always_inline)) {
                        constexpr int q = decltype(qc)::value;
                        if constexpr (PEND != 0) {
                            constexpr int lo = S1 + share_lo(NLN - S1, MQ, q), hi = S1 + share_hi(NLN - S1, MQ, q);
                            constexpr int qp = prev_share_slot(NLN - S1, MQ, q);
                            ln32_run<PEND == 2, DT, lo, hi, (qp < 0 ? 0 : WB::frags_between(qp, q))>(B, V, pga, pba);
                        }
                    }));
                }
                CGNN_S32_STAMP(2);
                ring.template sync_next<IS_ENC ? 0 : NPL>();
                CGNN_S32_STAMP(3);
                if constexpr (IS_ENC) {
                    const bf16x8 (&inB)[2] = reinterpret_cast<const bf16x8(&)[2]>(B.in[0]);
                    CGNN_S32_WB(2, B.acc, inB, base, ring.base_next(), make_fill([&](auto qc) __attribute__((always_inline)) {
                        constexpr int q = decltype(qc)::value;
                        pack32_run<true, DT, share_lo(NPACK, MQ0, q), share_hi(NPACK, MQ0, q)>(A);
                        for (int i = share_lo(G::NP, MQ0, q); i < share_hi(G::NP, MQ0, q); ++i) ring.piece(i);
                        if constexpr (q == MQ0 - 1) bias_rows32<DT, 0, DT>(A.acc, ring.vec_addr(1), h);
                    }));
                } else {
                    CGNN_S32_VMCNT(0);          // B's P rows (requested a block ago; nothing newer is in flight)
                    p32_ready<DT>(ps, pd);
                    CGNN_S32_STAMP(4);
                    // pieces first (the P block has no LDS traffic of its own), pack(A) spread over both blocks
                    selp32<DT>(B.acc, ps, pd, sel0, sel1, make_fill([&](auto qc) __attribute__((always_inline)) {
                        constexpr int q = decltype(qc)::value;
                        for (int i = share_lo(G::NP, PQ, q); i < share_hi(G::NP, PQ, q); ++i) ring.piece(i);
                    }));
                    CGNN_S32_STAMP(5);
                    CGNN_S32_WB(KS, B.acc, B.in, base, ring.base_next(), make_fill([&](auto qc) __attribute__((always_inline)) {
                        constexpr int q = decltype(qc)::value;
                        if constexpr (q < QP) pack32_run<true, DT, share_lo(NPACK, QP, q), share_hi(NPACK, QP, q)>(A);
                        if constexpr (q == QB) bias_rows32<DT, 0, DT>(A.acc, ring.vec_addr(1), h);
                    }));
                    CGNN_S32_STAMP(6);
                }
                ring.dma_done();
                ring.advance();
            }
            // A.in = ReLU(layer 0 of A), A.acc = bias of A's next layer; B.acc = layer 0 of B, still to be packed
            // ---------------- hidden layers 1 .. nh - 1 ----------------
            for (int l = 1; l < nh; ++l) {
                const unsigned base = ring.base();
                CGNN_S32_WB(KS, A.acc, A.in, base, base, make_fill([&](auto qc) __attribute__((always_inline)) {
                    constexpr int q = decltype(qc)::value;
                    if constexpr (q < QP) pack32_run<true, DT, share_lo(NPACK, QP, q), share_hi(NPACK, QP, q)>(B);
                    if constexpr (q == QB) bias_rows32<DT, 0, DT>(B.acc, ring.vec_addr(0), h);
                }));
                CGNN_S32_STAMP(7);
                ring.template sync_next<0>();
                CGNN_S32_STAMP(8);
                CGNN_S32_WB(KS, B.acc, B.in, base, ring.base_next(), make_fill([&](auto qc) __attribute__((always_inline)) {
                    constexpr int q = decltype(qc)::value;
                    for (int i = share_lo(G::NP, MQ, q); i < share_hi(G::NP, MQ, q); ++i) ring.piece(i);
                    if constexpr (q < QP) pack32_run<true, DT, share_lo(NPACK, QP, q), share_hi(NPACK, QP, q)>(A);
                    if constexpr (q == QB) bias_rows32<DT, 0, DT>(A.acc, ring.vec_addr(1), h);
                }));
                CGNN_S32_STAMP(9);
                ring.dma_done();
                ring.advance();
            }
            // ---------------- output layer + LayerNorm (+ residual) ----------------
            {
                const unsigned base = ring.base();
                const unsigned ga = ring.vec_addr(0) + (unsigned)D * 4u + 16u * (unsigned)h, ba = ga + (unsigned)D * 4u;
                // the next round's P rows of tile A (the P registers are free since B's layer 0)
                // (the last round requests round 0's rows again: nobody reads them, but the loop stays branch-free and
                // the wait counts constant)
                const bool wrap = !IS_ENC && rr + 1 == L;
                const __bf16* nps = (IS_ENC || wrap) ? ps_all : tps + round_stride;
                const __bf16* npd = (IS_ENC || wrap) ? pd_all : tpd + round_stride;
                CGNN_S32_WB(KS, A.acc, A.in, base, base, make_fill([&](auto qc) __attribute__((always_inline)) {
                    constexpr int q = decltype(qc)::value;
                    if constexpr (q < QP) pack32_run<true, DT, share_lo(NPACK, QP, q), share_hi(NPACK, QP, q)>(B);
                    p_issue_range(nps, npd, soA, doA, CGNN_IC(share_lo(NPL, MQ, q)), CGNN_IC(share_hi(NPL, MQ, q)));
                    if constexpr (q == QB) bias_rows32<DT, 0, DT>(B.acc, ring.vec_addr(0), h);
                }));
                CGNN_S32_STAMP(10);
                ring.template sync_next<NPL>();
                CGNN_S32_STAMP(11);
                // pieces in the first slots, A's LayerNorm over all of them
                CGNN_S32_WB(KS, B.acc, B.in, base, ring.base_next(), make_fill([&](auto qc) __attribute__((always_inline)) {
                    constexpr int q = decltype(qc)::value;
                    constexpr int qp = prev_share_slot(NLN, MQ, q);
                    ln32_run<!IS_ENC, DT, share_lo(NLN, MQ, q), share_hi(NLN, MQ, q), (qp < 0 ? 0 : WB::frags_between(qp, q))>(
                        A, V, ga, ba);
                    for (int i = share_lo(G::NP, MQ, q); i < share_hi(G::NP, MQ, q); ++i) ring.piece(i);
                }));
                CGNN_S32_STAMP(12);
                ring.dma_done();
                ring.advance();
            }
            // pending: LayerNorm of B (its output is in B.acc, the vectors in the slot just left)
        };
        auto flush_ln_b = [&](auto res_tag) __attribute__((always_inline)) {
            const unsigned pga = ring.vec_addr(-1) + (unsigned)D * 4u + 16u * (unsigned)h, pba = pga + (unsigned)D * 4u;
            ln32_run<decltype(res_tag)::value, DT, 0, NLN, 0>(B, V, pga, pba);
        };

        if (ENC) {
            pass(std::true_type{}, CGNN_IC(0), 0, false);
            pass(std::false_type{}, CGNN_IC(1), 0, true);
            for (int rr = 1; rr < L; ++rr) pass(std::false_type{}, CGNN_IC(2), rr, true);
        } else {
            pass(std::false_type{}, CGNN_IC(0), 0, false);
            for (int rr = 1; rr < L; ++rr) pass(std::false_type{}, CGNN_IC(2), rr, true);
        }
        flush_ln_b(std::true_type{});

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
    CGNN_S32_VMCNT(0);      // the ring's last refills (unread) must have landed before the workgroup's LDS is released
    __builtin_amdgcn_s_barrier();
#undef CGNN_IC
#undef CGNN_S32_WB
}

template <int DT>
static int launch_stream32(const S32Args& a, const __bf16* ps, const __bf16* pd, int64_t round_stride, const int32_t* src,
                           const int32_t* dst, int64_t num_edges, const float* e_in, float* e_out, const float* attr,
                           int ld_attr, hipStream_t st) {
    typedef S32Geom<DT> G;
    const bool enc = a.enc_in_dim > 0;
    auto kern = enc ? edge_stream32_kernel<DT, true> : edge_stream32_kernel<DT, false>;
    if (G::LDS > 48 * 1024) {
        int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (size_t)((int)G::LDS), "hipFuncSetAttribute(edge_stream32)");
        if (rc != CGNN_OK) return rc;
    }
    const int64_t pairs = ((num_edges + 31) / 32 + 1) / 2;
    const int grid = grid_for_tiles(pairs, 1, CGNN_S32_WAVES);
    kern<<<grid, CGNN_S32_BLOCK, G::LDS, st>>>(a, ps, pd, round_stride, src, dst, num_edges, e_in, e_out, attr, ld_attr);
#ifdef CGNN_S32_STAMPS
    {
        static int printed = 0;
        hipStreamSynchronize(st);
        if (printed++ == 2) {
            static unsigned long long hs[4 * 64];
            hipMemcpyFromSymbol(hs, HIP_SYMBOL(cgnn_s32_stamps), sizeof(hs));
            const char* names[13] = {"start", "A.P done", "A.L0 done", "barrier0", "B rows ready", "B.P done", "B.L0 done",
                                     "A.L1 done", "barrier1", "B.L1 done", "A.L2 done", "barrier2", "B.L2 done"};
            for (int k = 0; k < 13; ++k) {
                printf("stamp %2d %-14s", k, names[k]);
                for (int w = 0; w < 4; ++w) printf(" %7lld (+%5lld)", (long long)(hs[w * 64 + k] - hs[0]),
                                                   k ? (long long)(hs[w * 64 + k] - hs[w * 64 + k - 1]) : 0LL);
                printf("\n");
            }
        }
    }
#endif
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
