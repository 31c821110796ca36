// cgnn_edge_stream: all message-passing rounds of the EDGE stream in one launch (reference graph_network.py:89-90,182
// for round = 0 .. L-1), for the reference's actual data flow.
//
// Under the reference's aggregation (PyG's default message: sender NODE latents) the node stream never
// reads the edge stream, so the per-node halves Ps_r / Pd_r of every round's first edge Linear are known before any
// edge is touched.  An edge's latent tile can then stay in registers through all L updates
//     e <- e + LN_r(MLP_r(Ps_r[src] + Pd_r[dst] + We_r e))          r = 0 .. L-1
// and the E x D f32 stream crosses HBM once instead of L times; with the edge encoder run on the tile in the same launch
// (ENC) it is only ever written (cfg3, PMC: 16 GB of HBM traffic instead of 182 GB).  What no longer
// fits is the weights: one round's three 128x128 bf16 layers fill the LDS, so the layers of all rounds cycle through
// a three-slot LDS ring, copied by LDS-DMA two layers ahead of their use; the eight waves of a workgroup meet at one
// barrier per layer.  Arithmetic (MFMA shapes, operand order, f32 LayerNorm and residual) is that of
// edge_block_n16_kernel, so the result is bit-identical to L per-round launches.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "n16.hpp"

namespace cgnn {

#ifndef CGNN_STREAM_BLOCK
#define CGNN_STREAM_BLOCK 512
#endif
#define CGNN_STREAM_WAVES (CGNN_STREAM_BLOCK / 64)
#define CGNN_STREAM_SLOTS 3
#define CGNN_STREAM_MAX_CHUNKS 64
#define CGNN_STREAM_MAX_ROUNDS 32
#ifndef CGNN_STREAM_NB
#define CGNN_STREAM_NB 2     // buffered LDS fragment groups per wave (one in flight ahead; 3 and 4 measured 1.5 % and 4 % slower)
#endif

struct StreamArgs {
    const char* w[CGNN_STREAM_MAX_CHUNKS];       // packed CGNN_BF16_N16 layers in consumption order (round-major)
    const float* bias[CGNN_STREAM_MAX_CHUNKS];   // bias of the same layer (entry of a round's layer 0 unused: it is in Pd)
    const float* gamma[CGNN_STREAM_MAX_ROUNDS];
    const float* beta[CGNN_STREAM_MAX_ROUNDS];
    uint32_t chunk_bytes[CGNN_STREAM_MAX_CHUNKS];
    int32_t rounds, nh;
    // optional edge encoder (graph_network.py:57) run on each tile before round 0: its nh + 1 layers are entries
    // 0 .. enc_layers-1 of w / bias / chunk_bytes, the rounds' layers follow
    int32_t enc_layers, enc_in_dim;
    const float* enc_gamma;
    const float* enc_beta;
};

#ifdef CGNN_STREAM_STAMPS   // developer build: s_memtime stamps of one workgroup's waves over their second tile
__device__ unsigned long long cgnn_stream_stamps[8 * 256];
#define CGNN_STAMP(k)                                                                        \
    if (stamp_on && lane == 0 && (k) < 256) cgnn_stream_stamps[wave * 256 + (k)] = __builtin_readcyclecounter()
#else
#define CGNN_STAMP(k)
#endif

typedef __attribute__((address_space(3))) void* LdsVoidPtrS;
typedef const __attribute__((address_space(1))) void* GlobalVoidPtrS;

#define CGNN_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// The ring.  Every wave copies its share (1-KiB pieces wave, wave + 8, ...) of a layer into a slot with LDS-DMA.
// The wave's vector-memory operations retire in issue order, so "the pieces of layer c have landed" is a counted
// s_waitcnt: at most as many operations outstanding as this wave has issued since those pieces.  after[] tracks that
// number for the two layers in flight (wave-uniform scalars); other loads and stores of the kernel report themselves
// with note().
template <uint32_t SLOT_BYTES>
struct LayerRing {
    const StreamArgs& a;
    int wave, lane;
    int count;           // layers per tile = rounds * (nh + 1)
    int chunk;           // layer the next acquire() returns
    int slot;            // its slot
    int left;            // acquires still to come (over all tiles of this workgroup)
    int after0, after1;  // operations issued after the pieces of `chunk` / of the layer after it
    const __attribute__((address_space(3))) unsigned long long* tab_ptr;
    const __attribute__((address_space(3))) uint32_t* tab_bytes;
    __device__ __forceinline__ LayerRing(const StreamArgs& aa, int w, int l, int steps, char* table)
        : a(aa), wave(w), lane(l), count(aa.enc_layers + aa.rounds * (aa.nh + 1)), chunk(0), slot(0), left(steps), after0(0),
          after1(0),
          tab_ptr((const __attribute__((address_space(3))) unsigned long long*)table),
          tab_bytes((const __attribute__((address_space(3))) uint32_t*)(table + 8 * CGNN_STREAM_MAX_CHUNKS)) {}

    // The layer table (source pointer, bytes) is read from an LDS copy, not from the kernel arguments: a scalar load in
    // the loop makes the compiler wait for ALL outstanding LDS reads (s_waitcnt lgkmcnt(0): scalar loads share that
    // counter and return out of order), which serialised the weight-fragment prefetch of every MFMA group.
    __device__ __forceinline__ int issue(int c, int s) const {
        asm volatile("" ::: "memory");
        const unsigned long long pv = tab_ptr[c];
        const char* src = reinterpret_cast<const char*>(
            ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(pv >> 32)) << 32) |
            (unsigned)__builtin_amdgcn_readfirstlane((int)(pv & 0xffffffffull)));
        const uint32_t nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)tab_bytes[c]);
        char* dst = cgnn_smem + s * SLOT_BYTES;
        int n = 0;
        for (uint32_t off = wave * 1024u; off < nb; off += CGNN_STREAM_WAVES * 1024u) {
            __builtin_amdgcn_global_load_lds((GlobalVoidPtrS)(src + off + lane * 16), (LdsVoidPtrS)(dst + off), 16, 0, 0);
            ++n;
        }
        asm volatile("" ::: "memory");
        return n;
    }
    __device__ __forceinline__ void note(int ops) {
        after0 += ops;
        after1 += ops;
    }
    __device__ __forceinline__ void prime() {    // layers 0 and 1 of the first tile
        const int n0 = issue(0, 0);
        (void)n0;
        after0 = 0;
        if (left > 1) {
            const int n1 = issue(count > 1 ? 1 : 0, 1);
            after0 = n1;
        }
        after1 = 0;
    }
    // Layer `chunk` readable by every wave; returns its LDS image.  start_next() then starts the copy of the layer two
    // ahead (into the slot everybody has just left).  They are separate so that a caller can first consume registers
    // whose loads the compiler guards with a full vmcnt(0) -- it cannot count across the loop's back edge -- while
    // only old operations are outstanding.
    __device__ __forceinline__ LdsW wait_ready() {
        const int n = __builtin_amdgcn_readfirstlane(after0);
        if (n >= 12)
            CGNN_VMCNT(12);
        else if (n >= 4)
            CGNN_VMCNT(4);
        else if (n >= 1)
            CGNN_VMCNT(1);
        else
            CGNN_VMCNT(0);
        __builtin_amdgcn_s_barrier();   // everybody's pieces have landed, and everybody is done with the previous layer's slot
        asm volatile("" ::: "memory");
        return LdsW((LdsWeightPtr)(cgnn_smem + slot * SLOT_BYTES));
    }
    // begin_next() does the bookkeeping of the copy two layers ahead and returns this wave's number of 1-KiB pieces;
    // piece(i) issues one of them (the caller spreads them between its MFMA groups: a vector-memory instruction takes
    // ~100+ cycles to issue when all eight waves issue theirs at once right after the barrier).
    const char* nsrc;
    char* ndst;
    uint32_t nbytes;
    __device__ __forceinline__ int begin_next() {
        --left;
        int c2 = chunk + 2;
        if (c2 >= count) c2 -= count;
        if (c2 >= count) c2 -= count;          // count == 1
        const int s2 = slot == 0 ? 2 : slot - 1;   // (slot + 2) % 3
        after0 = after1;
        after1 = 0;
        int n = 0;
        if (left > 1) {
            const unsigned long long pv = tab_ptr[c2];
            nsrc = reinterpret_cast<const char*>(
                ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(pv >> 32)) << 32) |
                (unsigned)__builtin_amdgcn_readfirstlane((int)(pv & 0xffffffffull)));
            nbytes = (uint32_t)__builtin_amdgcn_readfirstlane((int)tab_bytes[c2]);
            ndst = cgnn_smem + s2 * SLOT_BYTES;
            const uint32_t first = wave * 1024u;
            n = nbytes > first ? (int)((nbytes - first + CGNN_STREAM_WAVES * 1024u - 1) / (CGNN_STREAM_WAVES * 1024u)) : 0;
            after0 += n;
        }
        chunk = chunk + 1 == count ? 0 : chunk + 1;
        slot = slot == 2 ? 0 : slot + 1;
        return n;
    }
    __device__ __forceinline__ void piece(int i, int n) const {
        if (i < n) {
            const uint32_t off = (wave + CGNN_STREAM_WAVES * i) * 1024u;
            asm volatile("" ::: "memory");
            __builtin_amdgcn_global_load_lds((GlobalVoidPtrS)(nsrc + off + lane * 16), (LdsVoidPtrS)(ndst + off), 16, 0, 0);
            asm volatile("" ::: "memory");
        }
    }
    __device__ __forceinline__ void start_next() {
        const int n = begin_next();
        for (int i = 0; i < n; ++i) piece(i, n);
    }
    __device__ __forceinline__ LdsW acquire() {
        const LdsW w = wait_ready();
        start_next();
        return w;
    }
};

// ENC: the tile's initial latents come from the edge encoder run in-kernel on `attr` (edge features, <= 32 per edge)
// instead of being read from e_in: the encoder's E x D f32 output is then never written or read.
template <int HT, int DT, bool PMFMA, bool ENC>
__global__ __launch_bounds__(CGNN_STREAM_BLOCK) void edge_stream_n16_kernel(
    StreamArgs a, const __bf16* __restrict__ ps_all, const __bf16* __restrict__ pd_all, int64_t round_stride,
    const int32_t* __restrict__ src, const int32_t* __restrict__ dst, int64_t num_edges, const float* e_in, float* e_out,
    const float* __restrict__ attr, int ld_attr) {
    constexpr int D = 32 * DT, H = 32 * HT, DO = 2 * DT, HO = 2 * HT;
    constexpr int W = H > D ? H : D;
    constexpr uint32_t SLOT = 2u * (uint32_t)(H * (H > D ? H : D));
    constexpr int PMAX = (int)((SLOT + CGNN_STREAM_WAVES * 1024u - 1) / (CGNN_STREAM_WAVES * 1024u));   // pieces per wave and layer (at most)
    const int L = a.rounds, nh = a.nh, NV = nh + 2;      // vectors per round: biases of layers 1..nh, gamma, beta
    float* vecs = reinterpret_cast<float*>(cgnn_smem + CGNN_STREAM_SLOTS * SLOT);
    for (int idx = threadIdx.x; idx < L * NV * W; idx += blockDim.x) {
        const int r = idx / (NV * W), v = (idx / W) % NV, i = idx % W;
        const float* p = v < nh ? a.bias[a.enc_layers + r * (nh + 1) + v + 1] : (v == nh ? a.gamma[r] : a.beta[r]);
        const int dim = v < nh - 1 ? H : D;
        vecs[idx] = (p != nullptr && i < dim) ? p[i] : 0.f;
    }
    // encoder vectors: biases of its layers 0 .. nh, gamma, beta
    constexpr int ENC_EXTRA = ENC ? 1 : 0;
    float* evecs = vecs + L * NV * W;
    if (ENC) {
        for (int idx = threadIdx.x; idx < (nh + 3) * W; idx += blockDim.x) {
            const int v = idx / W, i = idx % W;
            const float* p = v <= nh ? a.bias[v] : (v == nh + 1 ? a.enc_gamma : a.enc_beta);
            const int dim = v < nh ? H : D;
            evecs[idx] = (p != nullptr && i < dim) ? p[i] : 0.f;
        }
    }
    char* table = cgnn_smem + CGNN_STREAM_SLOTS * SLOT + ((size_t)L * NV + ENC_EXTRA * (nh + 3)) * W * 4;   // layer table
    if ((int)threadIdx.x < a.enc_layers + L * (nh + 1)) {
        reinterpret_cast<unsigned long long*>(table)[threadIdx.x] = (unsigned long long)a.w[threadIdx.x];
        reinterpret_cast<uint32_t*>(table + 8 * CGNN_STREAM_MAX_CHUNKS)[threadIdx.x] = a.chunk_bytes[threadIdx.x];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t tiles = (num_edges + 15) / 16;
    const TileRange tr = tile_range(tiles);
    // every wave of the workgroup runs the same number of tile iterations (they share the ring's barriers): the count
    // of wave 0; a wave whose last tile falls off the range recomputes its previous tile and skips the store
    const int64_t first0 = tr.first - wave;
    // (the 64-bit division runs on the vector unit: tell the compiler its result is wave-uniform)
    const int iters = __builtin_amdgcn_readfirstlane(
        first0 < tr.end ? (int)((tr.end - first0 + tr.stride - 1) / tr.stride) : 0);
    if (iters == 0) return;
    LayerRing<SLOT> ring(a, wave, lane, iters * (a.enc_layers + L * (nh + 1)), table);
    ring.prime();
    const bf16x8 sel0 = p16_selector(lane, 0), sel1 = p16_selector(lane, 1);
    const LdsVecPtr vbase = (LdsVecPtr)(cgnn_smem + CGNN_STREAM_SLOTS * SLOT);

    int64_t tile = tr.first < tr.end ? tr.first : tr.end - 1;
    bool valid = tr.first < tr.end;
    int64_t tbase = (tile >> 1) * (32 * D) + n16_lane_offset(c, q, (int)(tile & 1));
    int64_t s, d;
    {
        const int64_t e = tile * 16 + c;
        const int64_t ec = e < num_edges ? e : num_edges - 1;
        s = src[ec];
        d = dst[ec];
    }
    f32x4 ev[DO];
    float at[8];          // ENC: this lane's edge features phi(0, q, j) of edge c (zero beyond the encoder's fan-in)
    const int enc_in = a.enc_in_dim;
    auto load_attr = [&](float (&dstv)[8], int64_t t) __attribute__((always_inline)) {
        const int64_t e = t * 16 + c;
        const int64_t ec = e < num_edges ? e : num_edges - 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int f = 16 * (j >> 2) + 4 * q + (j & 3);
            dstv[j] = f < enc_in ? attr[ec * ld_attr + f] : 0.f;
        }
    };
    if (ENC) {
        load_attr(at, tile);
        ring.note(8);
    } else {
#pragma unroll
        for (int o = 0; o < DO; ++o)
            ev[o] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(e_in + tbase + n16_tile_offset(o)));
        ring.note(DO);
    }
    bf16x8 pso[HT], pdo[HT];
    load_p16_operand<HT>(pso, ps_all, s, q);
    load_p16_operand<HT>(pdo, pd_all, d, q);
    ring.note(2 + 2 * HT);

    // (Tried and dropped: TWO tiles per wave per ring step, the second tile's f32 latents parked in LDS behind a
    // two-slot ring -- half the barriers and LDS-DMA pieces per edge, bit-identical results, but 256 registers with
    // spills and an exposed tile switch: 22.7 ms against 22.1 ms for this kernel on the same GPU.)
    // (Tried and dropped: running the two waves of a SIMD half a step apart -- one post-processing the previous layer
    // on the vector pipe while the other issues this layer's MFMAs -- measured 25.2 ms against 23.5 ms in lockstep at
    // cfg3; the longer live ranges cost more than the pipes' overlap gained.)
    for (int it = 0; it < iters; ++it) {
#ifdef CGNN_STREAM_STAMPS
        const bool stamp_on = blockIdx.x == 8 && it == 1;
        int sk = 0;
#endif
        const bool more = it + 1 < iters;
        int64_t tile_n = tile + tr.stride;
        const bool valid_n = tile_n < tr.end;
        if (!valid_n) tile_n = tile;
        const int64_t tbase_n = (tile_n >> 1) * (32 * D) + n16_lane_offset(c, q, (int)(tile_n & 1));
        int32_t s_n = (int32_t)s, d_n = (int32_t)d;     // widened only where they are used, a tile later
        f32x4 ev_n[DO];
        float at_n[8];
        // One round.  The last round of a tile is its own instantiation: only there are the next tile's latents and
        // round-0 rows fetched, so ev_n is not carried (and copied) through the other rounds.
        auto round = [&](int r, auto last_tag) __attribute__((always_inline)) {
            constexpr bool LAST = decltype(last_tag)::value;
            const LdsVecPtr vr = vbase + r * NV * W;
            bf16x8 oph[HT];
            {
                CGNN_STAMP(sk++);      // 0: arriving at the layer-0 barrier
                const LdsW w0 = ring.wait_ready();
                CGNN_STAMP(sk++);      // 1: released
                f32x4 acc[HO];
                // The P rows were requested a round ago.  With nh >= 2 that is before the pieces wait_ready() has just
                // waited for (counted vmcnt retires in order).  With ONE hidden layer a round is two ring steps, the
                // pieces of this layer were issued in the same step as the rows and BEFORE them, so the counted wait
                // does not cover the rows: drain (found by test_all_rounds_in_one_launch_random_shapes[nh=1], a rare
                // wrong tile when a row's load lost the race).
                if (nh == 1) CGNN_VMCNT(0);
                p16_ready<HT>(pso, pdo);
                if (PMFMA) {
                    p16_accumulate<HT>(acc, pso, pdo, sel0, sel1);
                } else {
                    p16_unpack<HT, false>(acc, pso);
                    p16_unpack<HT, true>(acc, pdo);
                }
                if (r == 0 && more) {      // the next tile's edge list entries, a whole tile ahead of their use
                    const int64_t e = tile_n * 16 + c;
                    const int64_t ec = e < num_edges ? e : num_edges - 1;
                    s_n = src[ec];
                    d_n = dst[ec];
                    ring.note(2);
                }
                if (LAST && more) {
                    if (ENC) {
                        load_attr(at_n, tile_n);
                        ring.note(8);
                    } else {
#pragma unroll
                        for (int o = 0; o < DO; ++o)
                            ev_n[o] = __builtin_nontemporal_load(
                                reinterpret_cast<const f32x4*>(e_in + tbase_n + n16_tile_offset(o)));
                        ring.note(DO);
                    }
                }
                CGNN_STAMP(sk++);      // 2: P MFMAs issued
                // The P registers are free again: the rows of the next round (or of the next tile's round 0) and this
                // wave's pieces of the layer two ahead are requested BETWEEN the MFMA groups below, pieces first.
                const bool pref = !LAST || more;
                const __bf16* prs = (LAST ? ps_all : ps_all + (r + 1) * round_stride) +
                                    (LAST ? (int64_t)s_n : s) * H + 8 * q;
                const __bf16* prd = (LAST ? pd_all : pd_all + (r + 1) * round_stride) +
                                    (LAST ? (int64_t)d_n : d) * H + 8 * q;
                const int np = ring.begin_next();
                CGNN_STAMP(sk++);      // 3: ring bookkeeping done
                bf16x8 op[DT];
                operand16<false, DT>(op, ev);
                constexpr int NG0 = (HO * DT + 3) / 4;
                dense16_fast<DT, HO, CGNN_STREAM_NB>(acc, op, w0, lane, [&](auto gc) __attribute__((always_inline)) {
                    constexpr int g = decltype(gc)::value;
                    if constexpr (g < PMAX) ring.piece(g, np);
                    static_for_each([&](auto sc) __attribute__((always_inline)) {
                        constexpr int ks = decltype(sc)::value;
                        if constexpr ((PMAX + ks < NG0 - 1 ? PMAX + ks : NG0 - 1) == g) {
                            if (pref) {
                                pso[ks] = load_p16_step_untracked<ks>(prs);
                                pdo[ks] = load_p16_step_untracked<ks>(prd);
                            }
                        }
                    }, std::make_integer_sequence<int, HT>{});
                });
                if (pref) ring.note(2 * HT);
                CGNN_STAMP(sk++);      // 4: layer-0 MFMAs issued
                operand16<true, HT>(oph, acc);
            }
            auto pieces_only = [&](int np) __attribute__((always_inline)) {
                return [&ring, np](auto gc) __attribute__((always_inline)) {
                    constexpr int g = decltype(gc)::value;
                    if constexpr (g < PMAX) ring.piece(g, np);
                };
            };
            for (int l = 1; l < nh; ++l) {
                CGNN_STAMP(sk++);      // 5: packed, arriving at the hidden-layer barrier
                const LdsW wl = ring.wait_ready();
                CGNN_STAMP(sk++);      // 6: released
                const int np = ring.begin_next();
                CGNN_STAMP(sk++);      // 7: ring bookkeeping done
                f32x4 acc[HO];
                fill16<HO>(acc, vr + (l - 1) * W, q);
                dense16_fast<HT, HO, CGNN_STREAM_NB>(acc, oph, wl, lane, pieces_only(np));
                CGNN_STAMP(sk++);      // 8: MFMAs issued
                operand16<true, HT>(oph, acc);
            }
            CGNN_STAMP(sk++);          // 9: arriving at the output-layer barrier
            const LdsW wo = ring.wait_ready();
            CGNN_STAMP(sk++);          // 10: released
            const int npo = ring.begin_next();
            CGNN_STAMP(sk++);          // 11: ring bookkeeping done
            f32x4 out[DO];
            fill16<DO>(out, vr + (nh - 1) * W, q);
            dense16_fast<HT, DO, CGNN_STREAM_NB>(out, oph, wo, lane, pieces_only(npo));
            CGNN_STAMP(sk++);          // 12: MFMAs issued
            layer_norm16<DO>(out, vr + nh * W, vr + (nh + 1) * W, q);
#pragma unroll
            for (int o = 0; o < DO; ++o) ev[o] += out[o];
            CGNN_STAMP(sk++);          // 13: LayerNorm + residual done
        };
        if (ENC) {     // edge encoder: MLP + LayerNorm of the edge features, no residual (its layers lead the ring)
            const LdsVecPtr ve = vbase + L * NV * W;
            bf16x8 oph[HT];
            {
                const LdsW w0 = ring.acquire();
                bf16x8 op[1];
                {
                    u32x4 v;
                    v[0] = pack_bf16(at[0], at[1]);
                    v[1] = pack_bf16(at[2], at[3]);
                    v[2] = pack_bf16(at[4], at[5]);
                    v[3] = pack_bf16(at[6], at[7]);
                    op[0] = __builtin_bit_cast(bf16x8, v);
                }
                f32x4 acc[HO];
                fill16<HO>(acc, ve, q);
                dense16_fast<1, HO, CGNN_STREAM_NB>(acc, op, w0, lane);
                operand16<true, HT>(oph, acc);
            }
            for (int l = 1; l < nh; ++l) {
                const LdsW wl = ring.acquire();
                f32x4 acc[HO];
                fill16<HO>(acc, ve + l * W, q);
                dense16_fast<HT, HO, CGNN_STREAM_NB>(acc, oph, wl, lane);
                operand16<true, HT>(oph, acc);
            }
            const LdsW wo = ring.acquire();
            fill16<DO>(ev, ve + nh * W, q);
            dense16_fast<HT, DO, CGNN_STREAM_NB>(ev, oph, wo, lane);
            layer_norm16<DO>(ev, ve + (nh + 1) * W, ve + (nh + 2) * W, q);
        }
        for (int r = 0; r + 1 < L; ++r) round(r, std::false_type{});
        round(L - 1, std::true_type{});
        if (valid) {
#pragma unroll
            for (int o = 0; o < DO; ++o)
                __builtin_nontemporal_store(ev[o], reinterpret_cast<f32x4*>(e_out + tbase + n16_tile_offset(o)));
            ring.note(DO);
        }
        if (more) {
            if (ENC) {
#pragma unroll
                for (int j = 0; j < 8; ++j) at[j] = at_n[j];
            } else {
#pragma unroll
                for (int o = 0; o < DO; ++o) ev[o] = ev_n[o];
            }
            tile = tile_n;
            tbase = tbase_n;
            valid = valid_n;
            s = s_n;
            d = d_n;
        }
    }
}

template <int HT, int DT>
static int launch_stream(const StreamArgs& a, size_t lds, const __bf16* ps, const __bf16* pd, int64_t round_stride,
                         const int32_t* src, const int32_t* dst, int64_t num_edges, const float* e_in, float* e_out,
                         const float* attr, int ld_attr, hipStream_t st) {
    // P rows through the matrix pipe (measured 24.2 ms against 28.1 ms through the vector pipe at cfg3: the loop is
    // bound by vector issue)
    constexpr bool pmfma = true;
    const bool enc = a.enc_layers > 0;
    auto kern = enc ? edge_stream_n16_kernel<HT, DT, pmfma, true> : edge_stream_n16_kernel<HT, DT, pmfma, false>;
    if (lds > 48 * 1024) {
        int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (size_t)((int)lds), "hipFuncSetAttribute(edge_stream)");
        if (rc != CGNN_OK) return rc;
    }
    const int grid = grid_for_tiles((num_edges + 15) / 16, 1, CGNN_STREAM_WAVES);
    kern<<<grid, CGNN_STREAM_BLOCK, lds, st>>>(a, ps, pd, round_stride, src, dst, num_edges, e_in, e_out, attr, ld_attr);
#ifdef CGNN_STREAM_STAMPS
    {
        static int printed = 0;
        hipStreamSynchronize(st);
        if (printed++ == 3) {
            static unsigned long long h[8 * 256];
            hipMemcpyFromSymbol(h, HIP_SYMBOL(cgnn_stream_stamps), sizeof(h));
            for (int k = 0; k < 14 * 3; ++k) {
                printf("stamp %3d:", k);
                for (int w = 0; w < 8; ++w) printf(" %8lld", (long long)(h[w * 256 + k] - h[0]));
                printf("\n");
            }
        }
    }
#endif
    return check_hip(hipGetLastError(), "cgnn_edge_stream launch");
}

}  // namespace cgnn

using namespace cgnn;

extern "C" int cgnn_edge_stream(const cgnn_mlp* rounds, int32_t num_rounds, const void* ps_all, const void* pd_all,
                                int64_t round_stride, const int32_t* src, const int32_t* dst, int64_t num_edges,
                                const float* e_in, float* e_out, int32_t latent, const cgnn_mlp* encoder,
                                const float* edge_attr, int32_t ld_attr, void* stream) {
    if (!rounds || num_rounds < 1 || !ps_all || !pd_all || !src || !dst || !e_out || num_edges < 0 || latent <= 0 ||
        round_stride < 0 || (encoder ? (!edge_attr || ld_attr <= 0) : !e_in)) {
        set_error("cgnn_edge_stream: invalid argument");
        return CGNN_ERR_INVALID_ARG;
    }
    if (num_rounds > CGNN_STREAM_MAX_ROUNDS) {
        set_error("cgnn_edge_stream: %d rounds, at most %d", num_rounds, CGNN_STREAM_MAX_ROUNDS);
        return CGNN_ERR_UNSUPPORTED;
    }
    StreamArgs a;
    memset(&a, 0, sizeof(a));
    a.rounds = num_rounds;
    int hidden = 0, nh = 0;
    {
        MlpDev m0;
        int rc = make_mlp_dev(&rounds[0], &m0, nullptr, "cgnn_edge_stream");
        if (rc != CGNN_OK) return rc;
        hidden = m0.out_dim[0];
        nh = m0.nh;
    }
    const int enc_n = encoder ? nh + 1 : 0;
    if ((int64_t)num_rounds * (nh + 1) + enc_n > CGNN_STREAM_MAX_CHUNKS) {
        set_error("cgnn_edge_stream: %d rounds x %d layers exceed %d ring entries", num_rounds, nh + 1,
                  CGNN_STREAM_MAX_CHUNKS);
        return CGNN_ERR_UNSUPPORTED;
    }
    if (encoder) {
        MlpDev m;
        int rc = make_mlp_dev(encoder, &m, nullptr, "cgnn_edge_stream(encoder)");
        if (rc != CGNN_OK) return rc;
        if (encoder->precision != CGNN_BF16_N16 || !m.gamma || m.nh != nh || m.in_dim[0] > 32 || ld_attr < m.in_dim[0] ||
            m.out_dim[0] != hidden || m.out_dim[nh] != latent) {
            set_error("cgnn_edge_stream: the encoder must be CGNN_BF16_N16 with LayerNorm, <= 32 inputs and the rounds' "
                      "hidden / latent sizes and depth");
            return CGNN_ERR_UNSUPPORTED;
        }
        for (int l = 0; l <= nh; ++l) {
            a.w[l] = (const char*)m.w[l];
            a.bias[l] = m.b[l];
            a.chunk_bytes[l] = m.bytes[l];
        }
        a.enc_layers = enc_n;
        a.enc_in_dim = m.in_dim[0];
        a.enc_gamma = m.gamma;
        a.enc_beta = m.beta;
    }
    for (int r = 0; r < num_rounds; ++r) {
        MlpDev m;
        int rc = make_mlp_dev(&rounds[r], &m, nullptr, "cgnn_edge_stream");
        if (rc != CGNN_OK) return rc;
        if (rounds[r].precision != CGNN_BF16_N16 || !m.gamma) {
            set_error("cgnn_edge_stream: round %d needs CGNN_BF16_N16 weights and LayerNorm parameters", r);
            return CGNN_ERR_UNSUPPORTED;
        }
        if (m.nh != nh || m.in_dim[0] != latent || m.out_dim[0] != hidden || m.out_dim[nh] != latent ||
            m.in_dim[nh] != hidden) {
            set_error("cgnn_edge_stream: round %d does not have the shape of round 0 (latent %d, hidden %d, %d hidden "
                      "layers)", r, latent, hidden, nh);
            return CGNN_ERR_INVALID_ARG;
        }
        for (int l = 0; l <= nh; ++l) {
            if (l >= 1 && l < nh && (m.in_dim[l] != hidden || m.out_dim[l] != hidden)) {
                set_error("cgnn_edge_stream: round %d hidden layer %d has the wrong shape", r, l);
                return CGNN_ERR_INVALID_ARG;
            }
            const int ci = enc_n + r * (nh + 1) + l;
            a.w[ci] = (const char*)m.w[l];
            a.bias[ci] = m.b[l];
            a.chunk_bytes[ci] = m.bytes[l];
        }
        a.gamma[r] = m.gamma;
        a.beta[r] = m.beta;
    }
    a.nh = nh;
    if (latent % 32 || hidden % 32) {
        set_error("cgnn_edge_stream: latent %d / hidden %d must be multiples of 32", latent, hidden);
        return CGNN_ERR_UNSUPPORTED;
    }
    if (num_edges == 0) return CGNN_OK;
    const int HT = hidden / 32, DT = latent / 32;
    const int wmax = hidden > latent ? hidden : latent;
    const size_t slot = 2u * (size_t)hidden * wmax;
    const size_t lds = CGNN_STREAM_SLOTS * slot + ((size_t)num_rounds * (nh + 2) + (encoder ? nh + 3 : 0)) * wmax * 4 +
                       12 * CGNN_STREAM_MAX_CHUNKS;
    if (lds > 160 * 1024) {
        set_error("cgnn_edge_stream: needs %zu bytes of LDS (three %zu-byte layer slots + %d rounds of vectors)", lds, slot,
                  num_rounds);
        return CGNN_ERR_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
#define CGNN_STREAM(Hh, Dd)   \
    if (HT == Hh && DT == Dd) \
        return launch_stream<Hh, Dd>(a, lds, (const __bf16*)ps_all, (const __bf16*)pd_all, round_stride, src, dst, num_edges, e_in, e_out, edge_attr, ld_attr, st);
    CGNN_STREAM(1, 1) CGNN_STREAM(2, 2) CGNN_STREAM(4, 4)
#undef CGNN_STREAM
    set_error("cgnn_edge_stream: no kernel for latent=%d hidden=%d (built for hidden == latent in {32,64,128})", latent,
              hidden);
    return CGNN_ERR_UNSUPPORTED;
}
