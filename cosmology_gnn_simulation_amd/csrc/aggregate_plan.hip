// cgnn_aggregate for a graph that is used more than once (every message-passing round of a forward reads the same
// receiver-sorted fixed-k edge list): a per-graph PLAN removes the repeated sender rows from the gather.
//
// reference graph_network.py:92 (PyG propagate, aggr='add', default message = sender node rows):
//     out[i] = sum_{j < k} table[gather[i k + j]]
//
// cgnn_aggregate's fixed-k kernel reads k rows per receiver from L2: 16 M row gathers (8.2 GB) per round at cfg3, 13.7
// TB/s of L2 -> CU traffic and 0.60 ms, although the receivers are in spatial (cell) order and neighbouring receivers
// share most of their senders.  The plan cuts the receivers into blocks of 64 (k = 8, 16; else 32) consecutive ones and stores, per block,
// the list of DISTINCT sender rows (about 300 of the 1024 references at k = 16 on a uniform box) and, per edge, the
// position of its sender in that list (uint16).  The kernel copies a block's distinct rows into LDS once (a 32-feature
// slice at a time: up to 352 rows x 128 B) and every receiver sums its k neighbours from LDS, in the same order as
// cgnn_aggregate (balanced tree for k = 8 / 16, left to right otherwise): results are bit-identical.
//
// A block with more than 352 distinct senders (strongly clustered particles, random graphs) gathers straight from memory
// like the plain kernel.
#include <type_traits>
#include <utility>

#include "cgnn_common.hpp"

namespace cgnn {

template <class F, int... I>
__device__ __forceinline__ void ap_static_for(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}

// receivers per block: 64 for the unrolled k = 8 / 16 kernels (two receivers per thread), 32 for the runtime-k kernel
// (one per thread; up to k = 32 a block's distinct senders then still fit the staging area)
__host__ __device__ inline int ap_block_rows(int k) { return (k == 8 || k == 16) ? 64 : 32; }
#define CGNN_AP_MAX_UNIQUE 512         // distinct senders a block's list can hold
#define CGNN_AP_STAGE_ROWS 352         // ... of which the kernel stages this many (two slices of them in flight in registers: 2 workgroups per CU); more: direct gather
#define CGNN_AP_ROW_F4 9               // LDS row pitch in 16-byte units: 128 bytes of data + 16 of padding
#define CGNN_AP_HASH 4096              // open-addressing table (>= 2 x the most keys a block can hold ... see build)
#define CGNN_AP_THREADS 256
#define CGNN_AP_MAX_K 32               // 64 x 32 = 2048 references per block
// buffer offset of "no row" in the branch-free loads / stores: past the end of any table the path is used for (< 4 GiB - 2 KiB)
// and, with the largest slice offset (7 x 128 bytes) added, still below 2^32 -- the range check is made on the 32-bit sum
#define CGNN_AP_NO_ROW 0xfffffbf0u

// plan blob: [count: nblocks x int32, padded to 256 B][unique: nblocks x 512 x int32][local: num_edges x uint16]
struct PlanView {
    int32_t* count;
    int32_t* unique;
    uint16_t* local;
};
__host__ __device__ inline size_t plan_count_bytes(int64_t nblocks) { return (size_t)((nblocks * 4 + 255) / 256) * 256; }
__host__ __device__ inline PlanView plan_view(void* blob, int64_t nblocks) {
    PlanView v;
    char* p = reinterpret_cast<char*>(blob);
    v.count = reinterpret_cast<int32_t*>(p);
    v.unique = reinterpret_cast<int32_t*>(p + plan_count_bytes(nblocks));
    v.local = reinterpret_cast<uint16_t*>(p + plan_count_bytes(nblocks) + (size_t)nblocks * CGNN_AP_MAX_UNIQUE * 4);
    return v;
}

__device__ __forceinline__ unsigned ap_hash(int32_t id) { return ((unsigned)id * 2654435761u) >> 20; }   // 12 bits

// One workgroup per block: hash-deduplicate the block's sender ids, number the distinct ones by table position.
__global__ __launch_bounds__(CGNN_AP_THREADS) void aggregate_plan_kernel(const int32_t* __restrict__ gather, int64_t num_nodes,
                                                                         int k, PlanView plan) {
    __shared__ int32_t keys[CGNN_AP_HASH];
    __shared__ int32_t rank[CGNN_AP_HASH];
    __shared__ int32_t wave_total[CGNN_AP_THREADS / 64];
    const int64_t b = blockIdx.x;
    const int block_rows = ap_block_rows(k);
    const int64_t row0 = b * block_rows;
    const int rows = (int)((num_nodes - row0) < block_rows ? (num_nodes - row0) : block_rows);
    const int refs = rows * k;
    const int32_t* g = gather + row0 * k;
    for (int i = threadIdx.x; i < CGNN_AP_HASH; i += CGNN_AP_THREADS) keys[i] = -1;
    __syncthreads();
    for (int e = threadIdx.x; e < refs; e += CGNN_AP_THREADS) {
        const int32_t id = g[e];
        unsigned s = ap_hash(id);
        for (;;) {
            const int32_t old = atomicCAS(&keys[s], -1, id);
            if (old == -1 || old == id) break;
            s = (s + 1) & (CGNN_AP_HASH - 1);
        }
    }
    __syncthreads();
    // exclusive scan of the occupied slots: 16 consecutive slots per thread
    constexpr int PER = CGNN_AP_HASH / CGNN_AP_THREADS;
    int mine = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) mine += keys[threadIdx.x * PER + i] >= 0;
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d);
        if ((int)(threadIdx.x & 63) >= d) incl += v;
    }
    if ((threadIdx.x & 63) == 63) wave_total[threadIdx.x >> 6] = incl;
    __syncthreads();
    int base = incl - mine;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wave_total[w];
    int total = 0;
    for (int w = 0; w < CGNN_AP_THREADS / 64; ++w) total += wave_total[w];
    const bool fits = total <= CGNN_AP_MAX_UNIQUE;
    int r = base;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int s = threadIdx.x * PER + i;
        const int32_t id = keys[s];
        if (id >= 0) {
            rank[s] = r;
            if (fits) plan.unique[b * CGNN_AP_MAX_UNIQUE + r] = id;
            ++r;
        }
    }
    if (threadIdx.x == 0) plan.count[b] = fits ? total : -1;
    __syncthreads();
    if (!fits) return;
    for (int e = threadIdx.x; e < refs; e += CGNN_AP_THREADS) {
        const int32_t id = g[e];
        unsigned s = ap_hash(id);
        while (keys[s] != id) s = (s + 1) & (CGNN_AP_HASH - 1);
        plan.local[row0 * k + e] = (uint16_t)rank[s];
    }
}

// One workgroup per block of 64 receivers; a 32-feature slice at a time through LDS.
// SL: width / 32 when the slice loop is unrolled (0: runtime count) AND the table is below 4 GiB: the row loads then are
// buffer loads without a branch (a lane without a row gets an offset past the end: no memory access, zeros) in straight-line
// code, so that hipcc's own waits are exact counts -- with `if (row) load` in a runtime loop it waited vmcnt(0) before
// nearly every use and the "two slices in flight" were one.
// (three waves per SIMD asked of the branch-free k = 8 / 16 forms: 168 registers, three workgroups per CU -- they compile to
// 168-170 without the hint)
template <int K, int SL>   // K = 8, 16: unrolled balanced tree (cgnn_aggregate's order); 0: runtime k, left to right
__global__ __launch_bounds__(CGNN_AP_THREADS, (K != 0 && SL != 0) ? 3 : 1) void aggregate_planned_kernel(const float* __restrict__ table,
                                                                            const int32_t* __restrict__ gather, PlanView plan,
                                                                            int krt, int64_t num_nodes, int width,
                                                                            float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char ap_smem[];
    typedef __attribute__((address_space(3))) f32x4* LdsF4;
    const LdsF4 stage = (LdsF4)ap_smem;          // [unique][8] f32x4: 128 B per row
    const int k = K ? K : krt;
    // Workgroups are dealt round-robin over the 8 XCDs (each with its own L2): XCD x takes the x-th contiguous eighth of
    // the blocks, so that neighbouring blocks -- which share senders -- meet in one L2 (placement only affects speed).
    const int64_t nblk = gridDim.x, per = nblk >> 3, rem = nblk & 7;
    const int64_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int64_t b = xcd * per + (xcd < rem ? xcd : rem) + slot;
    constexpr int HALVES = (K == 8 || K == 16) ? 2 : 1;     // 64-receiver blocks: two receivers per thread; else one
    const int64_t row0 = b * (32 * HALVES);
    const int U = plan.count[b];
    const int chunk = threadIdx.x & 7;            // 16-byte piece of the 128-byte slice
    const int r_lo = threadIdx.x >> 3;            // receivers r_lo and r_lo + 32 of the block
    const int slices = SL ? SL : width / 32;
    if (U < 0 || U > CGNN_AP_STAGE_ROWS) {   // too many distinct senders: the plain gather (same summation order)
        for (int s = 0; s < slices; ++s)
#pragma unroll
            for (int half = 0; half < HALVES; ++half) {
                const int64_t row = row0 + r_lo + 32 * half;
                if (row >= num_nodes) continue;
                const int32_t* g = gather + row * k;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                if (K == 8 || K == 16) {
                    f32x4 v[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j)
                        if (j < K) v[j] = *reinterpret_cast<const f32x4*>(table + (int64_t)g[j] * width + s * 32 + chunk * 4);
                    const f32x4 h0 = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                    acc = K == 16 ? h0 + (((v[8] + v[9]) + (v[10] + v[11])) + ((v[12] + v[13]) + (v[14] + v[15]))) : h0;
                } else {
                    for (int j = 0; j < k; ++j)
                        acc += *reinterpret_cast<const f32x4*>(table + (int64_t)g[j] * width + s * 32 + chunk * 4);
                }
                __builtin_nontemporal_store(acc, reinterpret_cast<f32x4*>(out + row * width + s * 32 + chunk * 4));
            }
        return;
    }
    // positions of this thread's two receivers' senders in the block's list (read once, used for every slice)
    constexpr int KR = K ? K : CGNN_AP_MAX_K;
    uint16_t li[HALVES][KR];
#pragma unroll
    for (int half = 0; half < HALVES; ++half) {
        const int64_t row = row0 + r_lo + 32 * half;
        const uint16_t* lp = plan.local + (row < num_nodes ? row : num_nodes - 1) * k;
        if (K) {
            const u32x4* lp4 = reinterpret_cast<const u32x4*>(lp);        // K * 2 bytes, 16-byte aligned (K = 8, 16)
#pragma unroll
            for (int w = 0; w < K / 8; ++w) {
                const u32x4 v = lp4[w];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    li[half][8 * w + 2 * i] = (uint16_t)(v[i] & 0xffffu);
                    li[half][8 * w + 2 * i + 1] = (uint16_t)(v[i] >> 16);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < KR; ++j) li[half][j] = j < k ? lp[j] : (uint16_t)0;
        }
    }
    // The distinct rows of slice s + 1 travel to registers while slice s is summed from LDS: thread (r_lo, chunk) owns the
    // 16-byte piece `chunk` of rows r_lo, r_lo + 32, ...  LDS rows are padded to 144 bytes so that the eight rows a wave
    // reads at once do not all start on the same two banks.
    constexpr int PASSES = CGNN_AP_STAGE_ROWS / 32;
    const int32_t* uq = plan.unique + b * CGNN_AP_MAX_UNIQUE;
    int32_t mine[PASSES];      // row ids (-1: none)
#pragma unroll
    for (int i = 0; i < PASSES; ++i) mine[i] = r_lo + 32 * i < U ? uq[r_lo + 32 * i] : -1;
    const float* const tcol = table + chunk * 4;
    // NPRE slices in flight: slice s + NPRE is requested when slice s has been copied to LDS (three measured slower: 232
    // registers, two workgroups per CU)
    constexpr int NPRE = 2;
    f32x4 pre[NPRE][PASSES];
    unsigned off[PASSES];      // SL: byte offset of the lane's piece of row mine[i] (past the end: no row)
    (void)off;
    auto fetch = [&](f32x4 (&dstp)[PASSES], int sl) __attribute__((always_inline)) {
        if constexpr (SL != 0) {
#if defined(__HIP_DEVICE_COMPILE__)
            // (every row offset below "no row" is in range: the table may be longer than the receivers' part of it)
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(table), 0, (int)CGNN_AP_NO_ROW, 0x00020000);
#pragma unroll
            for (int i = 0; i < PASSES; ++i)
                dstp[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off[i], sl * 128, 0));
#endif
        } else {
#pragma unroll
            for (int i = 0; i < PASSES; ++i)
                if (mine[i] >= 0) dstp[i] = *reinterpret_cast<const f32x4*>(tcol + (int64_t)mine[i] * width + sl * 32);
        }
    };
    if constexpr (SL != 0) {
#pragma unroll
        for (int i = 0; i < PASSES; ++i)
            off[i] = mine[i] >= 0 ? (unsigned)mine[i] * (unsigned)width * 4u + (unsigned)chunk * 16u : CGNN_AP_NO_ROW;
    }
#pragma unroll
    for (int p = 0; p < NPRE; ++p)
        if (p < slices) fetch(pre[p], p);
    auto one_slice = [&](int s, auto parity) __attribute__((always_inline)) {
        constexpr int P = decltype(parity)::value;
#pragma unroll
        for (int i = 0; i < PASSES; ++i)
            if (SL != 0 || mine[i] >= 0) stage[(r_lo + 32 * i) * CGNN_AP_ROW_F4 + chunk] = pre[P][i];      // (SL: zeros for "no row")
        __syncthreads();
        if (s + NPRE < slices) fetch(pre[P], s + NPRE);
#pragma unroll
        for (int half = 0; half < HALVES; ++half) {
            const int64_t row = row0 + r_lo + 32 * half;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (K == 8 || K == 16) {
                f32x4 v[16];
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (j < K) v[j] = stage[(int)li[half][j] * CGNN_AP_ROW_F4 + chunk];
                const f32x4 h0 = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                acc = K == 16 ? h0 + (((v[8] + v[9]) + (v[10] + v[11])) + ((v[12] + v[13]) + (v[14] + v[15]))) : h0;
            } else {
#pragma unroll
                for (int j = 0; j < KR; ++j)
                    if (j < k) acc += stage[(int)li[half][j] * CGNN_AP_ROW_F4 + chunk];
            }
            if constexpr (SL != 0 && K != 0) {
#if defined(__HIP_DEVICE_COMPILE__)
                // (no branch either: a row past the end gets an offset past the end of `out`, the store is dropped; with it
                // the k = 16 kernel needs 168 registers instead of 204: three workgroups per CU.  The runtime-k kernel
                // measured 2 % slower this way and keeps the predicated store)
                const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
                    out, 0, (int)((unsigned)num_nodes * (unsigned)width * 4u), 0x00020000);
                const unsigned ooff = row < num_nodes ? (unsigned)row * (unsigned)width * 4u + (unsigned)chunk * 16u : CGNN_AP_NO_ROW;
                // The slice offset travels in the VECTOR offset, the scalar offset stays the constant 0.  With a register in
                // the scalar-offset field hipcc assumes that a 16-byte buffer store needs no wait state before a vector
                // instruction overwrites its data registers (GCNHazardRecognizer: "this hazard only exists if the
                // instruction is not using a register in the soffset field") -- on gfx950 it does: the next slice's
                // v_pk_add_f32 into the same registers, issued right behind the store, changed element 1 of the stored
                // vector in lanes 12-15 of every 16 (seen as wrong sums in odd rows; scripts/dev/dbg_agg.py).
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc), orsrc, (int)(ooff + (unsigned)s * 128u), 0, 2 /* nt */);
#endif
            } else if (row < num_nodes) {
                __builtin_nontemporal_store(acc, reinterpret_cast<f32x4*>(out + row * width + s * 32 + chunk * 4));
            }
        }
        __syncthreads();                            // everybody is done reading this slice
    };
    if constexpr (SL != 0) {
        ap_static_for([&](auto sc) __attribute__((always_inline)) {
            one_slice(decltype(sc)::value, std::integral_constant<int, decltype(sc)::value % NPRE>{});
        }, std::make_integer_sequence<int, SL>{});
    } else {
        for (int s = 0; s < slices; s += 2) {
            one_slice(s, std::integral_constant<int, 0>{});
            if (s + 1 < slices) one_slice(s + 1, std::integral_constant<int, 1>{});
        }
    }
}

}  // namespace cgnn

using namespace cgnn;

extern "C" {

size_t cgnn_aggregate_plan_bytes(int64_t num_nodes, int32_t fixed_k) {
    if (num_nodes <= 0 || fixed_k <= 0 || fixed_k > CGNN_AP_MAX_K) return 0;
    const int64_t nblocks = (num_nodes + ap_block_rows(fixed_k) - 1) / ap_block_rows(fixed_k);
    return plan_count_bytes(nblocks) + (size_t)nblocks * CGNN_AP_MAX_UNIQUE * 4 + (size_t)num_nodes * fixed_k * 2;
}

int cgnn_aggregate_plan_build(const int32_t* gather, int64_t num_nodes, int32_t fixed_k, void* plan, void* stream) {
    if (!gather || !plan || num_nodes <= 0 || fixed_k <= 0) {
        set_error("cgnn_aggregate_plan_build: invalid argument");
        return CGNN_ERR_INVALID_ARG;
    }
    if (fixed_k > CGNN_AP_MAX_K) {
        set_error("cgnn_aggregate_plan_build: fixed_k=%d above %d", fixed_k, CGNN_AP_MAX_K);
        return CGNN_ERR_UNSUPPORTED;
    }
    const int64_t nblocks = (num_nodes + ap_block_rows(fixed_k) - 1) / ap_block_rows(fixed_k);
    aggregate_plan_kernel<<<(unsigned)nblocks, CGNN_AP_THREADS, 0, (hipStream_t)stream>>>(gather, num_nodes, fixed_k,
                                                                                         plan_view(plan, nblocks));
    return check_hip(hipGetLastError(), "cgnn_aggregate_plan_build launch");
}

int cgnn_aggregate_planned(const float* table, const int32_t* gather, const void* plan, int64_t num_nodes, int32_t fixed_k,
                           int32_t width, float* out, void* stream) {
    return cgnn_aggregate_planned_rows(table, 0, gather, plan, num_nodes, fixed_k, width, out, stream);
}

int cgnn_aggregate_planned_rows(const float* table, int64_t table_rows, const int32_t* gather, const void* plan,
                                int64_t num_nodes, int32_t fixed_k, int32_t width, float* out, void* stream) {
    if (!table || !gather || !plan || !out || num_nodes < 0 || fixed_k <= 0 || width <= 0 || table_rows < 0) {
        set_error("cgnn_aggregate_planned: invalid argument");
        return CGNN_ERR_INVALID_ARG;
    }
    if (width % 32 != 0 || fixed_k > CGNN_AP_MAX_K) {
        set_error("cgnn_aggregate_planned: width %d must be a multiple of 32 and fixed_k %d <= %d", width, fixed_k,
                  CGNN_AP_MAX_K);
        return CGNN_ERR_UNSUPPORTED;
    }
    if (num_nodes == 0) return CGNN_OK;
    const int64_t nblocks = (num_nodes + ap_block_rows(fixed_k) - 1) / ap_block_rows(fixed_k);
    const PlanView pv = plan_view(const_cast<void*>(plan), nblocks);
    const int lds = CGNN_AP_STAGE_ROWS * CGNN_AP_ROW_F4 * 16;
    hipStream_t st = (hipStream_t)stream;
    // unrolled slice loop + branch-free buffer loads / stores: latent 128 / 256 with table and output below 4 GiB (32-bit
    // row offsets).  The table may hold more rows than there are receivers (ghost rows of a spatial shard): its row count
    // must be known (cgnn_aggregate_planned_rows), else the general kernel runs
    const int64_t most_rows = table_rows > num_nodes ? table_rows : num_nodes;
    const int sl = (table_rows > 0 && most_rows * width * 4 <= (int64_t)CGNN_AP_NO_ROW && (width == 128 || width == 256)) ? width / 32 : 0;
#define CGNN_AP_GO(Kk)                                                                                              \
    if (sl == 4) CGNN_AP_GO2(Kk, 4) else if (sl == 8) CGNN_AP_GO2(Kk, 8) else CGNN_AP_GO2(Kk, 0)
#define CGNN_AP_GO2(Kk, SLl)                                                                                        \
    {                                                                                                               \
        auto kern = aggregate_planned_kernel<Kk, SLl>;                                                              \
        int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (size_t)(lds), "hipFuncSetAttribute(aggregate_planned)");                                               \
        if (rc != CGNN_OK) return rc;                                                                               \
        kern<<<(unsigned)nblocks, CGNN_AP_THREADS, lds, st>>>(table, gather, pv, fixed_k, num_nodes, width, out);    \
    }
    if (fixed_k == 16) CGNN_AP_GO(16)
    else if (fixed_k == 8) CGNN_AP_GO(8)
    else CGNN_AP_GO(0)
#undef CGNN_AP_GO
#undef CGNN_AP_GO2
    return check_hip(hipGetLastError(), "cgnn_aggregate_planned launch");
}

}  // extern "C"
