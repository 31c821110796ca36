// cgnn_edge_block, CGNN_F16X2_N16 weights, latent = hidden = 128: one round's edge update (reference graph_network.py:89-90,
// :182) at f32 accuracy on the fp16 matrix cores -- the arithmetic of n16.hpp (two fp16 terms per value, three
// v_mfma_f32_16x16x32_f16 per fragment), the LDS weight ring of f2_ring.hpp / node_block_f2.hip.
//
//   e' = e + LayerNorm(W3 relu(W2 relu(Ps[src] + Pd[dst] + We e) + b2) + b3)
//
// The first Linear acts on cat([x_src, x_dst, e]) (:89); its two node thirds arrive as the f32 tables Ps = x Ws^T and
// Pd = x Wd^T + b1 (cgnn_project_nodes, CGNN_P_F32), gathered per edge straight into the accumulator; `units` = the e
// third, the hidden Linears, the output Linear, 64 KiB each, stream through the ring (eight slots here: no resident
// projection matrices).  A wave owns 16 edges (half a CGNN_TILED32 tile: its loads and stores are 256-byte runs), a
// workgroup step 128.  The next tile's latents, its two P rows and the indices of the tile after it are requested at the
// start of the tail of the current one (LayerNorm, residual, stores).
//
// Two instantiations: full steps (every wave stores, the tail's counted wait leaves the stores in flight) and RAGGED (one
// last partial step: waves past the end only keep the barriers and the ring going, the tail drains everything).
#include <string.h>

#include "f2_ring.hpp"

namespace cgnn {

int num_compute_units();   // runtime.hip

#define CGNN_F2E_MAX_UNITS 4      // hidden layers <= 3

struct F2EdgeArgs {
    const char* unit[CGNN_F2E_MAX_UNITS];   // packed CGNN_F16X2_N16: We (the e third of Linear 0), hidden..., output
    const float* bias[CGNN_F2E_MAX_UNITS];  // bias[l] of Linear l = 1 .. nh (Linear 0's lives in Pd)
    const float* gamma;
    const float* beta;
    const float* ps;                         // [n, 128] f32
    const float* pd;
    const int32_t* src;
    const int32_t* dst;
    const float* e_in;                       // CGNN_TILED32
    float* e_out;
    float* e_upd;                            // optional: the update before the residual (message_source = "edge")
    int64_t num_edges;
    int64_t first_half_tile;                 // 16-row half tiles [first, first + 8 * steps) belong to this launch
    int64_t half_tiles;                      // ... clipped to this many in all (RAGGED)
    int64_t steps;
    int32_t residual;
};

namespace f2r_edge {
constexpr int NS = 8, PD = NS - 1;
constexpr int RING_OFF = f2r::VEC_BYTES;
constexpr int LDS_BYTES = RING_OFF + NS * f2r::CHUNK;    // 132 KiB

__device__ __forceinline__ int idx_load(const int32_t* p) {
    int r;
    asm volatile("global_load_dword %0, %1, off" : "=v"(r) : "v"(p) : "memory");
    return r;
}
template <int N>
__device__ __forceinline__ void tile_ready(f32x4 (&a)[f2r::OT], f32x4 (&b)[f2r::OT], f32x4 (&c)[f2r::OT], int& s, int& d) {
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                 : "n"(N)
                 : "memory");
    asm volatile("" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
    asm volatile(""
                 : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(s),
                   "+v"(d));
}
}  // namespace f2r_edge

template <int NH, bool RAGGED>
__global__ __launch_bounds__(CGNN_F2R_BLOCK) void edge_block_f2ring_kernel(F2EdgeArgs a) {
    using namespace f2r;
    using namespace f2r_edge;
    constexpr int NU = NH + 1, NC = NU * UNIT_CHUNKS;
    static_assert(NU <= CGNN_F2E_MAX_UNITS, "too many layers");
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    {   // resident: bias of Linear 1 .. NH at vec[l], LayerNorm vectors behind them
        float* vec = reinterpret_cast<float*>(cgnn_smem);
        for (int i = threadIdx.x; i < D; i += blockDim.x) {
#pragma unroll
            for (int l = 1; l <= NH; ++l) vec[l * D + i] = a.bias[l][i];
            vec[(NH + 1) * D + i] = a.gamma[i];
            vec[(NH + 2) * D + i] = a.beta[i];
        }
    }
    __syncthreads();
    const LdsVecPtr vec = (LdsVecPtr)cgnn_smem;
    const unsigned ring_lds = (unsigned)(uintptr_t)(cgnn_smem + RING_OFF);

    const unsigned voff = (unsigned)wave * 1024u + (unsigned)lane * 16u;
    int slot = 0;
    auto issue = [&](int chunk, int into_slot) {
        const char* src = a.unit[chunk / UNIT_CHUNKS] + (chunk % UNIT_CHUNKS) * CHUNK;
#pragma unroll
        for (int i = 0; i < PC; ++i)
            dma_piece(src + i * (WAVES * 1024), voff, ring_lds + into_slot * CHUNK + (wave + WAVES * i) * 1024);
    };
#pragma unroll
    for (int i = 0; i < PD; ++i) issue(i, i);

    // half tile ht (16 edges) of this launch -> offsets; a half tile past the end is replaced by the last one for the loads
    const int64_t last_ht = a.half_tiles - 1;
    auto edge_of = [&](int64_t ht) {
        const int64_t e = ht * 16 + c;
        return e < a.num_edges ? e : a.num_edges - 1;
    };
    auto tile_offset = [&](int64_t ht) { return (ht >> 1) * (32 * D) + n16_lane_offset(c, q, (int)(ht & 1)); };
    auto clip = [&](int64_t ht) { return RAGGED && ht > last_ht ? last_ht : ht; };

    const int nb = gridDim.x;
    int64_t step = blockIdx.x;
    f32x4 en[OT], psn[OT], pdn[OT];
    int sn, dn;      // indices of the tile after the prefetched one
    {
        const int64_t ht = clip(a.first_half_tile + step * WAVES + wave);
        const int64_t ec = edge_of(ht);
        int s0 = idx_load(a.src + ec), d0 = idx_load(a.dst + ec);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(s0), "+v"(d0)::"memory");
        const float* ep = a.e_in + tile_offset(ht);
        const float* pp = a.ps + (int64_t)s0 * D + 4 * q;
        const float* dp = a.pd + (int64_t)d0 * D + 4 * q;
        static_for_each([&](auto oc) {
            constexpr int o = decltype(oc)::value;
            en[o] = row_load<0>(ep + n16_tile_offset(o));     // (beyond the instruction's 12-bit offset)
        }, std::make_integer_sequence<int, OT>{});
        static_for_each([&](auto oc) { psn[decltype(oc)::value] = row_load<decltype(oc)::value * 64>(pp); },
                        std::make_integer_sequence<int, OT>{});
        static_for_each([&](auto oc) { pdn[decltype(oc)::value] = row_load<decltype(oc)::value * 64>(dp); },
                        std::make_integer_sequence<int, OT>{});
        const int64_t nstep = step + nb < a.steps ? step + nb : step;
        const int64_t ec1 = edge_of(clip(a.first_half_tile + nstep * WAVES + wave));
        sn = idx_load(a.src + ec1);
        dn = idx_load(a.dst + ec1);
        tile_ready<0>(en, psn, pdn, sn, dn);
    }

    for (; step < a.steps; step += nb) {
        const int64_t ht_raw = a.first_half_tile + step * WAVES + wave;
        const bool valid = !RAGGED || ht_raw <= last_ht;           // wave-uniform
        const int64_t ht = clip(ht_raw);
        const int64_t next_step = step + nb < a.steps ? step + nb : step;
        const int64_t next2_step = next_step + nb < a.steps ? next_step + nb : next_step;
        const int64_t ht1 = clip(a.first_half_tile + next_step * WAVES + wave);
        const int64_t ec2 = edge_of(clip(a.first_half_tile + next2_step * WAVES + wave));

#ifdef CGNN_F2R_STAMPS
        const bool stamp_on = blockIdx.x == 9 && step == blockIdx.x + 3 * (int64_t)nb;
#endif
        F2R_STAMP(0);
        FragPipe16f2 pipe;
        f32x4 ev[OT], c0[OT], c1[OT];
        f16x8 op[2][KS];
#pragma unroll
        for (int o = 0; o < OT; ++o) {
            ev[o] = en[o];
            c0[o] = psn[o] + pdn[o];       // Linear 0's node thirds (+ b1): the accumulator starts from them
        }
        operand16f2<false, KS>(op, ev);
        fill16_global<OT>(c1, nullptr, q);
        F2R_STAMP(1);
        CGNN_F2R_UNIT(0, c0, c1, op)
        F2R_STAMP(2);
        fold16f2<OT>(c0, c1);
        F2R_SPLIT(true, op, c0);
        if constexpr (NH >= 2) {
            fill16<OT>(c0, vec + 1 * D, q);
            fill16_global<OT>(c1, nullptr, q);
            CGNN_F2R_UNIT(1, c0, c1, op)
            fold16f2<OT>(c0, c1);
            F2R_SPLIT(true, op, c0);
        }
        if constexpr (NH >= 3) {
            fill16<OT>(c0, vec + 2 * D, q);
            fill16_global<OT>(c1, nullptr, q);
            CGNN_F2R_UNIT(2, c0, c1, op)
            fold16f2<OT>(c0, c1);
            F2R_SPLIT(true, op, c0);
        }
        F2R_STAMP(3);
        fill16<OT>(c0, vec + NH * D, q);
        fill16_global<OT>(c1, nullptr, q);
        CGNN_F2R_UNIT(NU - 1, c0, c1, op)
        F2R_STAMP(4);

        // ---- tail: requests for the next tile (its indices arrived a step ago), then LayerNorm, residual, stores ----
        {
            const float* ep = a.e_in + tile_offset(ht1);
            const float* pp = a.ps + (int64_t)sn * D + 4 * q;
            const float* dp = a.pd + (int64_t)dn * D + 4 * q;
            static_for_each([&](auto oc) {
                constexpr int o = decltype(oc)::value;
                en[o] = row_load<0>(ep + n16_tile_offset(o));     // (beyond the instruction's 12-bit offset)
            }, std::make_integer_sequence<int, OT>{});
            static_for_each([&](auto oc) { psn[decltype(oc)::value] = row_load<decltype(oc)::value * 64>(pp); },
                            std::make_integer_sequence<int, OT>{});
            static_for_each([&](auto oc) { pdn[decltype(oc)::value] = row_load<decltype(oc)::value * 64>(dp); },
                            std::make_integer_sequence<int, OT>{});
            sn = idx_load(a.src + ec2);
            dn = idx_load(a.dst + ec2);
        }
        F2R_STAMP(5);
        fold16f2<OT>(c0, c1);
        layer_norm16<OT>(c0, vec + (NH + 1) * D, vec + (NH + 2) * D, q);
        F2R_STAMP(6);
        if (valid) {   // always true in the full-step instantiation
            const int64_t tb = tile_offset(ht);
            if (a.e_upd != nullptr) {   // block-uniform
#pragma unroll
                for (int o = 0; o < OT; ++o) *reinterpret_cast<f32x4*>(a.e_upd + tb + n16_tile_offset(o)) = c0[o];
            }
#pragma unroll
            for (int o = 0; o < OT; ++o) {
                if (a.residual) c0[o] += ev[o];
                *reinterpret_cast<f32x4*>(a.e_out + tb + n16_tile_offset(o)) = c0[o];
            }
        }
        F2R_STAMP(7);
        // younger than the requests: the 8 (or 16) stores of this tile -- unless the wave is past the end (RAGGED)
        if (RAGGED)
            tile_ready<0>(en, psn, pdn, sn, dn);
        else
            tile_ready<8>(en, psn, pdn, sn, dn);
        F2R_STAMP(8);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

template <int NH, bool RAGGED>
static int launch_f2edge(const F2EdgeArgs& a, hipStream_t st) {
    auto kern = edge_block_f2ring_kernel<NH, RAGGED>;
    int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (size_t)(f2r_edge::LDS_BYTES), "hipFuncSetAttribute(edge_block_f2ring)");
    if (rc != CGNN_OK) return rc;
    const int grid = (int)(a.steps < (int64_t)num_compute_units() ? a.steps : (int64_t)num_compute_units());
    kern<<<grid, CGNN_F2R_BLOCK, f2r_edge::LDS_BYTES, st>>>(a);
#ifdef CGNN_F2R_STAMPS
    {
        static int printed = 0;
        (void)hipStreamSynchronize(st);
        if (!RAGGED && printed++ == 3) {
            static unsigned long long hs[8 * 64];
            (void)hipMemcpyFromSymbol(hs, HIP_SYMBOL(cgnn_f2r_stamps), sizeof(hs));
            const char* names[9] = {"start", "e split", "unit We", "hidden", "unit out", "requests", "fold+LN", "stores", "ready"};
            for (int k = 0; k < 9; ++k) {
                printf("stamp %2d %-10s", k, names[k]);
                for (int w = 0; w < 8; ++w)
                    printf(" %6lld(+%5lld)", (long long)(hs[w * 64 + k] - hs[0]),
                           k ? (long long)(hs[w * 64 + k] - hs[w * 64 + k - 1]) : 0LL);
                printf("\n");
            }
        }
    }
#endif
    return check_hip(hipGetLastError(), "cgnn_edge_block(f16x2 ring) launch");
}

// Called by cgnn_edge_block (edge_block.hip) for CGNN_F16X2_N16 models after argument validation.
int edge_block_f2(const MlpDev& m, const float* ps, const float* pd, const int32_t* src, const int32_t* dst,
                  int64_t num_edges, const float* e_in, float* e_out, float* e_upd, int residual, hipStream_t st) {
    if (m.nh < 1 || m.nh > 3 || !m.gamma || !m.beta) {
        set_error("cgnn_edge_block: CGNN_F16X2_N16 needs 1..3 hidden layers and LayerNorm");
        return CGNN_ERR_UNSUPPORTED;
    }
    F2EdgeArgs a;
    memset(&a, 0, sizeof(a));
    for (int l = 0; l <= m.nh; ++l) {
        a.unit[l] = reinterpret_cast<const char*>(m.w[l]);
        a.bias[l] = m.b[l];
        if (l >= 1 && !m.b[l]) {
            set_error("cgnn_edge_block: CGNN_F16X2_N16 needs a bias on every Linear");
            return CGNN_ERR_UNSUPPORTED;
        }
    }
    a.gamma = m.gamma;
    a.beta = m.beta;
    a.ps = ps;
    a.pd = pd;
    a.src = src;
    a.dst = dst;
    a.e_in = e_in;
    a.e_out = e_out;
    a.e_upd = e_upd;
    a.num_edges = num_edges;
    a.residual = residual;
    a.half_tiles = 2 * ((num_edges + 31) / 32);       // the TILED32 buffers hold whole 32-row tiles
    const int64_t full = a.half_tiles / 8;
    int rc = CGNN_OK;
#define CGNN_GO(NHh, RAG)                      \
    if (rc == CGNN_OK && m.nh == NHh) rc = launch_f2edge<NHh, RAG>(a, st);
    if (full > 0) {
        a.first_half_tile = 0;
        a.steps = full;
        CGNN_GO(1, false) CGNN_GO(2, false) CGNN_GO(3, false)
    }
    if (rc == CGNN_OK && a.half_tiles % 8 != 0) {
        a.first_half_tile = full * 8;
        a.steps = 1;
        CGNN_GO(1, true) CGNN_GO(2, true) CGNN_GO(3, true)
    }
#undef CGNN_GO
    return rc;
}

}  // namespace cgnn
