// Schedule constants of edge_stream32w.hip at their shipped values, and what only developer builds carry: in-kernel cycle
// stamps and timing-only ablations (wrong results).  Product builds define none of the CGNN_W8_* macros, so every constant
// below has its default and every `if constexpr (w8dev::ABL_...)` branch is compiled out; scripts/ab/build_variants.sh builds
// side-by-side variants with -DCGNN_W8_...=... (DESIGN.md section 6 quotes what they measured).
#pragma once

#ifndef CGNN_W8_GS
#define CGNN_W8_GS 2       // LDS weight fragments per group (2 with two groups in flight: 17.78 against 17.98 ms with 4 / 1,
                           // same box; 4 / 2 and an issue priority around every MFMA measured no better)
#endif
#ifndef CGNN_W8_PD
#define CGNN_W8_PD 2       // groups in flight ahead of the MFMAs
#endif
#ifndef CGNN_W8_LN_EARLY
#define CGNN_W8_LN_EARLY 0    // LayerNorm affine slices (of 2 * latent / 32) done in the output layer's own step, before the barrier
#endif
// MFMA slots (within the next row tile's eight) that carry a finished row tile's pack halves / LayerNorm sums.  One slot
// later than "as early as possible" (1, 3 / 2, 4): the vector instructions then read accumulators whose last MFMA is one
// more MFMA old and issue without the wait (hipcc had padded them with s_nop 8 / 11): -1.5 % in same-box A/Bs.  (Holding the
// second-to-last row tile's share back for the end of the block, to have work under the last MFMA's latency: +0.3 .. 1.2 %.)
#ifndef CGNN_W8_PK0
#define CGNN_W8_PK0 2
#define CGNN_W8_PK1 4
#endif
#ifndef CGNN_W8_SM0
#define CGNN_W8_SM0 4
#define CGNN_W8_SM1 6
#endif
#ifndef CGNN_W8_PIECE_SLOT
#define CGNN_W8_PIECE_SLOT 16     // first MFMA slot of a 32-MFMA block that issues a ring piece (a piece has the rest of this
                                  // interval to land; from slot 0: +0.7 % in a same-box A/B, 4 / 8 / 24: the same as 16 within
                                  // 0.5 % -- the head of a block carries the fragment pipeline's start)
#endif
#ifndef CGNN_W8_CARRY
#define CGNN_W8_CARRY 1    // LAG 0: the ring's barrier vouches for chunk g + 2 (not g + 1), the LDS fragment pipeline runs across steps
#endif
#ifndef CGNN_W8_REQ_EARLY
#define CGNN_W8_REQ_EARLY 1   // a round's layer-0 fragments requested before its P sums / selector MFMAs
#endif
#ifndef CGNN_W8_PF16_INBLK
#define CGNN_W8_PF16_INBLK 1  // fp16 tables: the P sums of row tiles 1 .. 3 in the MFMA slots of the row tile before (17.03 against
                              // 17.16 ms with all 64 in front of the block)
#endif
#ifndef CGNN_W8_ADDP0     // slots (within a row tile's eight) that carry the next row tile's P sums, eight values each
#define CGNN_W8_ADDP0 5
#define CGNN_W8_ADDP1 6
#endif

// (included inside namespace cgnn)
namespace w8dev {      // timing-only ablations (results are wrong): what a phase costs, measured by leaving it out
#ifdef CGNN_W8_ABL_LN
constexpr bool ABL_LN = true;        // LayerNorm's affine part -> pack + add
#else
constexpr bool ABL_LN = false;
#endif
#ifdef CGNN_W8_ABL_SEL
constexpr bool ABL_SEL = true;       // no selector MFMAs (bf16 tables)
#else
constexpr bool ABL_SEL = false;
#endif
#ifdef CGNN_W8_ABL_BIAS
constexpr bool ABL_BIAS = true;      // no bias reads
#else
constexpr bool ABL_BIAS = false;
#endif
#ifdef CGNN_W8_SLEEP
constexpr int SLEEP = CGNN_W8_SLEEP; // idle cycles (x 64) per interval: is the kernel bound by cycles or by the clock it is given?
#else
constexpr int SLEEP = 0;
#endif
}  // namespace w8dev

#ifdef CGNN_W8_STAMPS   // per-phase cycle sums (s_memtime into scalar registers, no memory traffic inside the loop) of one
                        // workgroup's waves, printed by the launcher
#include <stdio.h>
__device__ unsigned long long cgnn_w8_stamps[8 * 32];
struct W8Timer {
    unsigned long long sum[20], prev;
};
#define CGNN_W8_STAMP_MEMBER W8Timer tm;
#define CGNN_W8_STAMP(k)                                                                 \
    {                                                                                    \
        unsigned long long t_;                                                           \
        __builtin_amdgcn_sched_barrier(0);                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");       \
        __builtin_amdgcn_sched_barrier(0);                                               \
        tm.sum[k] += t_ - tm.prev;                                                       \
        tm.prev = t_;                                                                    \
    }
#define CGNN_W8_STAMP_BEGIN(ring)                                                            \
    auto& tm = ring.tm;                                                                      \
    for (int k = 0; k < 20; ++k) tm.sum[k] = 0;                                              \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm.prev)::"memory");
#define CGNN_W8_STAMP_END(wave, lane)  \
    if (blockIdx.x == 8 && lane == 0)  \
        for (int k = 0; k < 20; ++k) cgnn_w8_stamps[wave * 32 + k] = tm.sum[k];
// a stamp adds the cycles since the previous stamp to its slot: slot k = the phase that ENDS at stamp k
#define CGNN_W8_STAMP_REPORT(st, num_edges, grid, passes_per_tile, lag)                                                           \
    {                                                                                                                             \
        static int printed = 0;                                                                                                   \
        (void)hipStreamSynchronize(st);                                                                                           \
        if (printed++ == 2) {                                                                                                     \
            static unsigned long long hs[8 * 32];                                                                                 \
            (void)hipMemcpyFromSymbol(hs, HIP_SYMBOL(cgnn_w8_stamps), sizeof(hs));                                                \
            const char* names[20] = {"(gap)", "last: bias, P requests", "last: MFMAs", "last: LayerNorm part, copy",              \
                                     "last: vmcnt wait", "last: barrier", "", "", "(gap; tile ends)", "first: LayerNorm rest, Pd", \
                                     "first: P sums", "first: L0 + pack", "first: vmcnt wait", "first: barrier", "(gap)",          \
                                     "hidden: bias, MFMAs, pack", "hidden: vmcnt wait", "hidden: barrier / idle interval",         \
                                     "idle: vmcnt", "idle: barrier"};                                                              \
            const double passes = (double)(((num_edges) + 31) / 32) / ((grid) * 8.0) * (passes_per_tile);                          \
            printf("lag %d: cycles per pass and wave, by phase (sums over the wave's whole run / %.0f passes)\n", lag, passes);    \
            double tot[8] = {0};                                                                                                  \
            for (int k = 0; k < 20; ++k) {                                                                                        \
                bool any = false;                                                                                                 \
                for (int w = 0; w < 8; ++w) any |= hs[w * 32 + k] != 0;                                                           \
                if (!any) continue;                                                                                               \
                printf("  %2d %-28s", k, names[k]);                                                                               \
                for (int w = 0; w < 8; ++w) {                                                                                     \
                    printf(" %7.0f", hs[w * 32 + k] / passes);                                                                    \
                    tot[w] += hs[w * 32 + k] / passes;                                                                            \
                }                                                                                                                 \
                printf("\n");                                                                                                     \
            }                                                                                                                     \
            printf("     %-28s", "total");                                                                                        \
            for (int w = 0; w < 8; ++w) printf(" %7.0f", tot[w]);                                                                 \
            printf("\n");                                                                                                         \
        }                                                                                                                         \
    }
#else
#define CGNN_W8_STAMP_MEMBER
#define CGNN_W8_STAMP(k)
#define CGNN_W8_STAMP_BEGIN(ring)
#define CGNN_W8_STAMP_END(wave, lane)
#define CGNN_W8_STAMP_REPORT(st, num_edges, grid, passes_per_tile, lag)
#endif
