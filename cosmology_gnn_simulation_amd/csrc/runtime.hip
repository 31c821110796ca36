// Library queries, error plumbing, weight packing and the bandwidth-bound
// utility kernels (aggregation, row gather/scatter, segmented column sum).
#include <stdarg.h>
#include <string.h>

#include <map>
#include <mutex>
#include <utility>

#include "cgnn_common.hpp"

namespace cgnn {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return CGNN_OK;
    set_error("%s: %s", what, hipGetErrorString(e));
    return CGNN_ERR_HIP;
}

// compute units of the CURRENT device (the one the caller's stream belongs to), cached per device
static int g_num_cu[64] = {0};

static int num_cus() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (g_num_cu[dev] == 0) {
        hipDeviceProp_t p;
        int n = 0;
        if (hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        g_num_cu[dev] = n > 0 ? n : 256;
    }
    return g_num_cu[dev];
}

// One wave owns one 32-row tile at a time; blocks are persistent (grid-stride).
int grid_for_tiles(int64_t tiles, int blocks_per_cu, int waves_per_block) {
    int64_t blocks = (tiles + waves_per_block - 1) / waves_per_block;
    const int64_t cap = (int64_t)num_cus() * blocks_per_cu;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    if (blocks >= 8) blocks = (blocks + 7) & ~(int64_t)7;   // multiple of 8: enables the XCD-aware tile map
    return (int)blocks;
}

int num_compute_units() { return num_cus(); }

int ensure_dynamic_lds(const void* kernel, size_t bytes, const char* what) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, size_t> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::lock_guard<std::mutex> lock(mu);
    auto it = done.find({kernel, dev});
    if (it != done.end() && it->second >= bytes) return CGNN_OK;
    const int rc = check_hip(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes), what);
    if (rc == CGNN_OK) done[{kernel, dev}] = bytes;
    return rc;
}

// ------------------------------------------------------------------ packing
__global__ void pack_f32_kernel(const float* __restrict__ w, int out_dim, int ld, int col0, int ncols, int KT,
                                int64_t total, float* __restrict__ wp) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int l = (int)(idx & 63);
    const int i = (int)((idx >> 6) & 15);
    const int64_t okt = idx >> 10;
    const int kt = (int)(okt % KT);
    const int o = (int)(okt / KT);
    const int row = 32 * o + (l & 31);
    const int k = 32 * kt + 8 * (i >> 2) + 4 * (l >> 5) + (i & 3);
    wp[idx] = (row < out_dim && k < ncols) ? w[(int64_t)row * ld + col0 + k] : 0.f;
}

__global__ void pack_bf16_kernel(const float* __restrict__ w, int out_dim, int ld, int col0, int ncols, int KT,
                                 int64_t total, __bf16* __restrict__ wp) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int j = (int)(idx & 7);
    const int l = (int)((idx >> 3) & 63);
    const int s = (int)((idx >> 9) & 1);
    const int64_t okt = idx >> 10;
    const int kt = (int)(okt % KT);
    const int o = (int)(okt / KT);
    const int row = 32 * o + (l & 31);
    const int k = 32 * kt + 16 * s + 8 * (j >> 2) + 4 * (l >> 5) + (j & 3);
    const float v = (row < out_dim && k < ncols) ? w[(int64_t)row * ld + col0 + k] : 0.f;
    wp[idx] = (__bf16)v;
}

// CGNN_F32X3: the bf16 fragment layout with three parts per fragment, [m][part][lane][j]; part p holds the p-th
// bf16 term of the weight (w = w1 + w2 + w3).
__global__ void pack_f32x3_kernel(const float* __restrict__ w, int out_dim, int ld, int col0, int ncols, int KT,
                                  int64_t total, __bf16* __restrict__ wp) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int j = (int)(idx & 7);
    const int l = (int)((idx >> 3) & 63);
    const int64_t mp = idx >> 9;
    const int part = (int)(mp % 3);
    const int64_t m = mp / 3;
    const int s = (int)(m & 1);
    const int64_t okt = m >> 1;
    const int kt = (int)(okt % KT);
    const int o = (int)(okt / KT);
    const int row = 32 * o + (l & 31);
    const int k = 32 * kt + 16 * s + 8 * (j >> 2) + 4 * (l >> 5) + (j & 3);
    const float v = (row < out_dim && k < ncols) ? w[(int64_t)row * ld + col0 + k] : 0.f;
    const __bf16 w1 = (__bf16)v;
    const float r1 = v - (float)w1;
    const __bf16 w2 = (__bf16)r1;
    const __bf16 w3 = (__bf16)(r1 - (float)w2);
    wp[idx] = part == 0 ? w1 : (part == 1 ? w2 : w3);
}

// CGNN_F16X2: the same fragment layout with two fp16 parts, [m][part][lane][j]: part 0 = fp16(w), part 1 =
// fp16((w - part0) * 2048)  (cgnn_common.hpp)
__global__ void pack_f16x2_kernel(const float* __restrict__ w, int out_dim, int ld, int col0, int ncols, int KT,
                                  int64_t total, _Float16* __restrict__ wp) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int j = (int)(idx & 7);
    const int l = (int)((idx >> 3) & 63);
    const int64_t mp = idx >> 9;
    const int part = (int)(mp & 1);
    const int64_t m = mp >> 1;
    const int s = (int)(m & 1);
    const int64_t okt = m >> 1;
    const int kt = (int)(okt % KT);
    const int o = (int)(okt / KT);
    const int row = 32 * o + (l & 31);
    const int k = 32 * kt + 16 * s + 8 * (j >> 2) + 4 * (l >> 5) + (j & 3);
    const float v = (row < out_dim && k < ncols) ? w[(int64_t)row * ld + col0 + k] : 0.f;
    const _Float16 hi = (_Float16)v;
    wp[idx] = part == 0 ? hi : (_Float16)((v - (float)hi) * 2048.0f);
}

// CGNN_BF16_N16: fragment m = O * KS + s (16-feature out tile O, 32-wide k-step s), A[row][k] of
// v_mfma_f32_16x16x32_bf16: lane l holds row l & 15, k = 8 (l >> 4) + j  <->  feature phi(s, l >> 4, j) (n16.hpp)
__global__ void pack_bf16_n16_kernel(const float* __restrict__ w, int out_dim, int ld, int col0, int ncols, int KS,
                                     int64_t total, __bf16* __restrict__ wp) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int j = (int)(idx & 7);
    const int l = (int)((idx >> 3) & 63);
    const int64_t m = idx >> 9;
    const int s = (int)(m % KS);
    const int o = (int)(m / KS);
    const int row = 16 * o + (l & 15);
    const int k = 32 * s + 16 * (j >> 2) + 4 * (l >> 4) + (j & 3);
    const float v = (row < out_dim && k < ncols) ? w[(int64_t)row * ld + col0 + k] : 0.f;
    wp[idx] = (__bf16)v;
}

// CGNN_F32X3_N16: the N16 fragment order with three bf16 terms per weight, [m][part][lane][j].
__global__ void pack_f32x3_n16_kernel(const float* __restrict__ w, int out_dim, int ld, int col0, int ncols, int KS,
                                      int64_t total, __bf16* __restrict__ wp) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int j = (int)(idx & 7);
    const int l = (int)((idx >> 3) & 63);
    const int64_t mp = idx >> 9;
    const int part = (int)(mp % 3);
    const int64_t m = mp / 3;
    const int s = (int)(m % KS);
    const int o = (int)(m / KS);
    const int row = 16 * o + (l & 15);
    const int k = 32 * s + 16 * (j >> 2) + 4 * (l >> 4) + (j & 3);
    const float v = (row < out_dim && k < ncols) ? w[(int64_t)row * ld + col0 + k] : 0.f;
    const __bf16 w1 = (__bf16)v;
    const float r1 = v - (float)w1;
    const __bf16 w2 = (__bf16)r1;
    const __bf16 w3 = (__bf16)(r1 - (float)w2);
    wp[idx] = part == 0 ? w1 : (part == 1 ? w2 : w3);
}

// CGNN_F16X2_N16: the N16 fragment order with two fp16 terms per weight, [m][part][lane][j]: part 0 = fp16(w), part 1 =
// fp16((w - part0) * 2048)  (n16.hpp, CGNN_F16X2 arithmetic).
__global__ void pack_f16x2_n16_kernel(const float* __restrict__ w, int out_dim, int ld, int col0, int ncols, int KS,
                                      int64_t total, _Float16* __restrict__ wp) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int j = (int)(idx & 7);
    const int l = (int)((idx >> 3) & 63);
    const int64_t mp = idx >> 9;
    const int part = (int)(mp & 1);
    const int64_t m = mp >> 1;
    const int s = (int)(m % KS);
    const int o = (int)(m / KS);
    const int row = 16 * o + (l & 15);
    const int k = 32 * s + 16 * (j >> 2) + 4 * (l >> 4) + (j & 3);
    const float v = (row < out_dim && k < ncols) ? w[(int64_t)row * ld + col0 + k] : 0.f;
    const _Float16 hi = (_Float16)v;
    wp[idx] = part == 0 ? hi : (_Float16)((v - (float)hi) * 2048.0f);
}

// ------------------------------------------------------------------ aggregation
// Fixed in-degree, receiver-sorted: one (row, 16-byte chunk) per thread; the k
// neighbour rows are read with 16 B per lane, a row's chunks on adjacent lanes.
// XCD-aware work split for the gather kernels: workgroup b runs on XCD b % 8 (round-robin dispatch), so the
// workgroups of one XCD take one contiguous eighth of the (spatially ordered) receivers and the sender rows they
// share are fetched into that XCD's L2 once instead of into all eight.
struct WorkRange {
    int64_t first, end, stride;
};
__device__ __forceinline__ WorkRange xcd_work_range(int64_t total) {
    const int nb = gridDim.x, b = blockIdx.x;
    WorkRange r;
    if ((nb & 7) == 0) {
        const int xcd = b & 7, slot = b >> 3, per = nb >> 3;
        r.first = total * xcd / 8 + (int64_t)slot * blockDim.x + threadIdx.x;
        r.end = total * (xcd + 1) / 8;
        r.stride = (int64_t)per * blockDim.x;
    } else {
        r.first = (int64_t)b * blockDim.x + threadIdx.x;
        r.end = total;
        r.stride = (int64_t)nb * blockDim.x;
    }
    return r;
}

__global__ void aggregate_fixedk_kernel(const float* __restrict__ table, const int32_t* __restrict__ gather,
                                        int k, int64_t num_nodes, int chunks, float* __restrict__ out) {
    const WorkRange wr = xcd_work_range(num_nodes * chunks);
    for (int64_t gid = wr.first; gid < wr.end; gid += wr.stride) {
        const int64_t row = gid / chunks;
        const int c = (int)(gid - row * chunks);
        const int64_t e0 = row * k;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (k == 16 || k == 8) {
            // balanced pairwise tree: the order of the cross-lane reduction in the fused edge kernel, so that the
            // fused and the stand-alone aggregation are bit-identical.  All k index loads, then all k row loads, are
            // issued before the first add (16 gathers in flight per lane).
            int32_t idx[16];
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (j < k) idx[j] = gather ? gather[e0 + j] : 0;
            f32x4 v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (j < k) {
                    const int64_t r = gather ? (int64_t)idx[j] : (e0 + j);
                    v[j] = *reinterpret_cast<const f32x4*>(table + (r * chunks + c) * 4);
                }
            const f32x4 h0 = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            if (k == 16)
                acc = h0 + (((v[8] + v[9]) + (v[10] + v[11])) + ((v[12] + v[13]) + (v[14] + v[15])));
            else
                acc = h0;
        } else {
#pragma unroll 8
            for (int j = 0; j < k; ++j) {
                const int64_t idx = gather ? (int64_t)gather[e0 + j] : (e0 + j);
                const f32x4 v = *reinterpret_cast<const f32x4*>(table + (idx * chunks + c) * 4);
                acc += v;
            }
        }
        // written once and read back once by the node kernel: non-temporal, so the sender rows keep their L2 lines
        __builtin_nontemporal_store(acc, reinterpret_cast<f32x4*>(out + gid * 4));
    }
}

// float offset of the 16-byte chunk c (features 4c..4c+3) of row e in a TILED32 matrix of `chunks` chunks/row
__device__ __forceinline__ int64_t tiled_chunk_offset(int64_t e, int c, int chunks) {
    const int q = c >> 1, hh = c & 1;   // q = 4t + g
    return (e >> 5) * ((int64_t)chunks * 128) + (((int64_t)q * 64 + 32 * hh + (e & 31)) << 2);
}

// Per-edge messages in TILED32 layout, fixed in-degree: the k chunks a thread sums are contiguous 16-byte
// pieces (consecutive edges of a receiver sit on consecutive lanes' slots of the tile).
__global__ void aggregate_fixedk_tiled_kernel(const float* __restrict__ table, int k, int64_t num_nodes, int chunks,
                                              float* __restrict__ out) {
    const int64_t total = num_nodes * chunks;
    for (int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gid < total;
         gid += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = gid / chunks;
        const int c = (int)(gid - row * chunks);
        const int64_t e0 = row * k;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (k == 16 || k == 8) {   // same balanced tree as above
            f32x4 half[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                if (8 * hh >= k) break;
                f32x4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    v[j] = *reinterpret_cast<const f32x4*>(table + tiled_chunk_offset(e0 + 8 * hh + j, c, chunks));
                half[hh] = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            }
            acc = k == 16 ? half[0] + half[1] : half[0];
        } else {
#pragma unroll 8
            for (int j = 0; j < k; ++j)
                acc += *reinterpret_cast<const f32x4*>(table + tiled_chunk_offset(e0 + j, c, chunks));
        }
        *reinterpret_cast<f32x4*>(out + gid * 4) = acc;
    }
}

__global__ void relayout_kernel(const float* __restrict__ src, int from_tiled, float* __restrict__ dst, int to_tiled,
                                int64_t n, int64_t n_pad, int chunks) {
    const int64_t total = n_pad * chunks;
    for (int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gid < total;
         gid += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = gid / chunks;
        const int c = (int)(gid - row * chunks);
        const int64_t so = from_tiled ? tiled_chunk_offset(row, c, chunks) : gid * 4;
        const int64_t dt = to_tiled ? tiled_chunk_offset(row, c, chunks) : gid * 4;
        if (row < n)
            *reinterpret_cast<f32x4*>(dst + dt) = *reinterpret_cast<const f32x4*>(src + so);
        else if (to_tiled)
            *reinterpret_cast<f32x4*>(dst + dt) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// General edge list: each group of `chunks` lanes walks a contiguous run of edges
// and flushes one atomic row per destination change.
__global__ void aggregate_atomic_kernel(const float* __restrict__ table, int table_tiled,
                                        const int32_t* __restrict__ gather, const int32_t* __restrict__ dst,
                                        int64_t num_edges, int chunks, int edges_per_group,
                                        float* __restrict__ out) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t group = gid / chunks;
    const int c = (int)(gid - group * chunks);
    const int64_t e0 = group * edges_per_group;
    if (e0 >= num_edges) return;
    const int64_t e1 = (e0 + edges_per_group < num_edges) ? e0 + edges_per_group : num_edges;
    int cur = dst[e0];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int64_t e = e0; e < e1; ++e) {
        const int d = dst[e];
        if (d != cur) {
            float* o = out + ((int64_t)cur * chunks + c) * 4;
            atomicAdd(o + 0, acc[0]);
            atomicAdd(o + 1, acc[1]);
            atomicAdd(o + 2, acc[2]);
            atomicAdd(o + 3, acc[3]);
            acc = f32x4{0.f, 0.f, 0.f, 0.f};
            cur = d;
        }
        const int64_t idx = gather ? (int64_t)gather[e] : e;
        acc += *reinterpret_cast<const f32x4*>(table + (table_tiled ? tiled_chunk_offset(idx, c, chunks)
                                                                    : (idx * chunks + c) * 4));
    }
    float* o = out + ((int64_t)cur * chunks + c) * 4;
    atomicAdd(o + 0, acc[0]);
    atomicAdd(o + 1, acc[1]);
    atomicAdd(o + 2, acc[2]);
    atomicAdd(o + 3, acc[3]);
}


// General edge list, the shape the float-atomic unit wants (MI355X: about 1.3 TB/s of added bytes chip-wide, reached
// only when ONE wave instruction adds 256 contiguous bytes, or two 128-byte segments; a 16-byte lane stride is several
// times slower).  LPR = width / 4 lanes share a row (16 bytes per lane: the efficient load shape), so a wave walks
// 64 / LPR contiguous edge ranges at once.
//  1. run-length reduction in registers: consecutive edges with the same destination (every edge of a receiver in a
//     receiver-sorted list) are summed before anything is added to memory;
//  2. a flush goes through a 1-KiB LDS block per wave, written row by row (16 bytes per lane) and read back dword by
//     dword, so that every atomic wave instruction covers 64 consecutive floats of one row (32 consecutive floats of
//     two rows at width 32): the full-rate shape;
//  3. four edges per step: their indices and rows are requested before the first add.
// Sum order inside a run is the list order; across runs it is the atomics' arrival order (not reproducible).
template <int LPR>
__global__ __launch_bounds__(CGNN_BLOCK) void aggregate_scatter_kernel(const float* __restrict__ table, int table_tiled,
                                                                       const int32_t* __restrict__ gather,
                                                                       const int32_t* __restrict__ dst, int64_t num_edges,
                                                                       int edges_per_group, float* __restrict__ out) {
    constexpr int GPW = 64 / LPR;          // groups (= edge ranges) per wave
    constexpr int W = 4 * LPR;             // row width in floats
    constexpr int U = 4;
    __shared__ __attribute__((aligned(16))) float stage[CGNN_WAVES_PER_BLOCK][256];
    __shared__ int stage_dst[CGNN_WAVES_PER_BLOCK][GPW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane / LPR, c = lane % LPR;
    const int64_t group = ((int64_t)blockIdx.x * CGNN_WAVES_PER_BLOCK + wave) * GPW + g;
    const int64_t e0 = group * edges_per_group;
    int64_t e1 = e0 + edges_per_group < num_edges ? e0 + edges_per_group : num_edges;
    if (e1 < e0) e1 = e0;
    // every group of the wave runs the same number of steps (the flush is a wave-wide exchange)
    const int steps = (edges_per_group + U - 1) / U;
    int cur = e0 < num_edges ? dst[e0] : -1;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    auto flush = [&](bool mine) {      // `mine`: this lane's group hands its sum over (wave-uniform call)
        *reinterpret_cast<f32x4*>(&stage[wave][lane * 4]) = mine ? acc : f32x4{0.f, 0.f, 0.f, 0.f};
        if (c == 0) stage_dst[wave][g] = mine ? cur : -1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = j * 64 + lane;                 // dword x of the block = row x / W, column x % W
            const int d = stage_dst[wave][x / W];
            if (d >= 0) atomicAdd(out + (int64_t)d * W + (x % W), stage[wave][x]);
        }
        __builtin_amdgcn_wave_barrier();
        if (mine) acc = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    for (int s = 0; s < steps; ++s) {
        const int64_t eb = e0 + (int64_t)s * U;
        int d[U];
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t e = eb + u;
            d[u] = e < e1 ? dst[e] : -2;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t e = eb + u;
            if (e < e1) {
                const int64_t idx = gather ? (int64_t)gather[e] : e;
                v[u] = *reinterpret_cast<const f32x4*>(table + (table_tiled ? tiled_chunk_offset(idx, c, LPR)
                                                                            : (idx * LPR + c) * 4));
            } else {
                v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool change = d[u] != -2 && d[u] != cur;
            if (__any(change)) flush(change);
            if (change) cur = d[u];
            acc += v[u];
        }
    }
    if (__any(cur >= 0)) flush(cur >= 0);
}

// ------------------------------------------------------------------ rows
__global__ void gather_rows_kernel(const float* __restrict__ table, const int32_t* __restrict__ idx, int64_t n_idx,
                                   int chunks, float* __restrict__ out) {
    const int64_t total = n_idx * chunks;
    for (int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gid < total;
         gid += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = gid / chunks;
        const int c = (int)(gid - row * chunks);
        *reinterpret_cast<f32x4*>(out + gid * 4) =
            *reinterpret_cast<const f32x4*>(table + ((int64_t)idx[row] * chunks + c) * 4);
    }
}

__global__ void scatter_rows_kernel(const float* __restrict__ rows, const int32_t* __restrict__ idx, int64_t n_idx,
                                    int chunks, float* __restrict__ table) {
    const int64_t total = n_idx * chunks;
    for (int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gid < total;
         gid += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = gid / chunks;
        const int c = (int)(gid - row * chunks);
        *reinterpret_cast<f32x4*>(table + ((int64_t)idx[row] * chunks + c) * 4) =
            *reinterpret_cast<const f32x4*>(rows + gid * 4);
    }
}

// Scalar-width variants (width not a multiple of 4).
__global__ void gather_rows_scalar_kernel(const float* __restrict__ table, const int32_t* __restrict__ idx,
                                          int64_t n_idx, int width, float* __restrict__ out) {
    const int64_t total = n_idx * width;
    for (int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gid < total;
         gid += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = gid / width;
        out[gid] = table[(int64_t)idx[row] * width + (gid - row * width)];
    }
}

__global__ void scatter_rows_scalar_kernel(const float* __restrict__ rows, const int32_t* __restrict__ idx,
                                           int64_t n_idx, int width, float* __restrict__ table) {
    const int64_t total = n_idx * width;
    for (int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gid < total;
         gid += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = gid / width;
        table[(int64_t)idx[row] * width + (gid - row * width)] = rows[gid];
    }
}

// ------------------------------------------------------------------ window features
__device__ __forceinline__ float torch_remainder(float a, float b) {   // torch.remainder for float32
    float r = fmodf(a, b);
    if (r != 0.f && ((r < 0.f) != (b < 0.f))) r = __fadd_rn(r, b);
    return r;
}

__global__ void window_features_kernel(const float* __restrict__ pos_seq, const float* __restrict__ temp_seq,
                                       const float* __restrict__ pos_noise, const float* __restrict__ temp_noise,
                                       int W, int64_t n, float box, float dt, float vel_mean, float vel_std,
                                       float temp_mean, float temp_std, float* __restrict__ x,
                                       float* __restrict__ recent_pos) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int F = 3 * (W - 1) + W;
    const float half = box * 0.5f, nhalf = -half;
    float* xr = x + i * F;
    float prev[3];
    for (int t = 0; t < W; ++t) {
        float cur[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float p = pos_seq[((int64_t)t * n + i) * 3 + c];
            if (pos_noise != nullptr) p = __fadd_rn(p, pos_noise[(i * W + t) * 3 + c]);
            cur[c] = torch_remainder(p, box);
        }
        if (t > 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float d = __fsub_rn(cur[c], prev[c]);
                if (d < nhalf) d = __fadd_rn(d, box);
                if (d > half) d = __fsub_rn(d, box);
                const float v = __fdiv_rn(d, dt);
                xr[3 * (t - 1) + c] = __fdiv_rn(__fsub_rn(v, vel_mean), vel_std);
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) prev[c] = cur[c];
        float T = temp_seq[(int64_t)t * n + i];
        if (temp_noise != nullptr) T = __fadd_rn(T, temp_noise[i * W + t]);
        xr[3 * (W - 1) + t] = __fdiv_rn(__fsub_rn(T, temp_mean), temp_std);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) recent_pos[i * 3 + c] = prev[c];
}

// ------------------------------------------------------------------ momentum
// Each block owns a contiguous run of rows.  Runs inside one graph are reduced in
// double precision in LDS and flushed with one atomic per column.
#define CGNN_COLSUM_ROWS 2048
__global__ void segment_colsum_kernel(const float* __restrict__ acc, const int32_t* __restrict__ batch, int64_t n,
                                      int width, double* __restrict__ sums) {
    __shared__ double red[CGNN_BLOCK];
    const int64_t r0 = (int64_t)blockIdx.x * CGNN_COLSUM_ROWS;
    const int64_t r1 = (r0 + CGNN_COLSUM_ROWS < n) ? r0 + CGNN_COLSUM_ROWS : n;
    const int g0 = batch ? batch[r0] : 0;
    const int g1 = batch ? batch[r1 - 1] : 0;
    if (g0 == g1) {
        for (int c = 0; c < width; ++c) {
            double s = 0.0;
            for (int64_t r = r0 + threadIdx.x; r < r1; r += blockDim.x) s += (double)acc[r * width + c];
            red[threadIdx.x] = s;
            __syncthreads();
            for (int off = CGNN_BLOCK / 2; off > 0; off >>= 1) {
                if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
                __syncthreads();
            }
            if (threadIdx.x == 0) atomicAdd(&sums[(int64_t)g0 * width + c], red[0]);
            __syncthreads();
        }
    } else {
        for (int64_t r = r0 + threadIdx.x; r < r1; r += blockDim.x) {
            const int g = batch[r];
            for (int c = 0; c < width; ++c) atomicAdd(&sums[(int64_t)g * width + c], (double)acc[r * width + c]);
        }
    }
}

static inline int blocks_for(int64_t total, int cap_mult = 16) {
    int64_t b = (total + CGNN_BLOCK - 1) / CGNN_BLOCK;
    const int64_t cap = (int64_t)num_cus() * cap_mult;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

// grid for the XCD-aware kernels: a multiple of 8 that still covers `total` when split into eighths
static inline int blocks_for_xcd(int64_t total, int cap_mult) {
    int64_t per = ((total + 7) / 8 + CGNN_BLOCK - 1) / CGNN_BLOCK;     // blocks needed by the largest eighth
    const int64_t cap = (int64_t)num_cus() * cap_mult / 8;
    if (per > cap) per = cap;
    if (per < 1) per = 1;
    return (int)(per * 8);
}

}  // namespace cgnn

using namespace cgnn;

extern "C" {

int cgnn_version(void) { return CGNN_VERSION; }
const char* cgnn_arch(void) { return "gfx950"; }
const char* cgnn_last_error(void) { return g_err; }

size_t cgnn_packed_linear_bytes(int32_t out_dim, int32_t ncols, int32_t precision) {
    if (out_dim <= 0 || ncols <= 0) return 0;
    const size_t kt = (size_t)(ncols + 31) / 32;
    if (precision == CGNN_BF16_N16) return (size_t)((out_dim + 15) / 16) * kt * 1024;
    if (precision == CGNN_F32X3_N16) return (size_t)((out_dim + 15) / 16) * kt * 3072;
    if (precision == CGNN_F16X2_N16) return (size_t)((out_dim + 15) / 16) * kt * 2048;
    const size_t ot = (size_t)(out_dim + 31) / 32;
    return ot * kt * 1024 * (precision == CGNN_BF16 ? 2 : (precision == CGNN_F32X3 ? 6 : 4));
}

int cgnn_pack_linear(const float* w, int32_t out_dim, int32_t ld, int32_t col0, int32_t ncols, int32_t precision,
                     void* packed, void* stream) {
    if (!w || !packed || out_dim <= 0 || ncols <= 0 || ld < col0 + ncols || col0 < 0) {
        set_error("cgnn_pack_linear: invalid argument");
        return CGNN_ERR_INVALID_ARG;
    }
    const int OT = (out_dim + 31) / 32, KT = (ncols + 31) / 32;
    hipStream_t st = (hipStream_t)stream;
    if (precision == CGNN_F32X3_N16) {
        const int64_t tot = (int64_t)((out_dim + 15) / 16) * KT * 512 * 3;
        pack_f32x3_n16_kernel<<<(unsigned)((tot + CGNN_BLOCK - 1) / CGNN_BLOCK), CGNN_BLOCK, 0, st>>>(
            w, out_dim, ld, col0, ncols, KT, tot, (__bf16*)packed);
        return check_hip(hipGetLastError(), "cgnn_pack_linear launch");
    }
    if (precision == CGNN_F16X2_N16) {
        const int64_t tot = (int64_t)((out_dim + 15) / 16) * KT * 512 * 2;
        pack_f16x2_n16_kernel<<<(unsigned)((tot + CGNN_BLOCK - 1) / CGNN_BLOCK), CGNN_BLOCK, 0, st>>>(
            w, out_dim, ld, col0, ncols, KT, tot, (_Float16*)packed);
        return check_hip(hipGetLastError(), "cgnn_pack_linear launch");
    }
    if (precision == CGNN_BF16_N16) {
        const int64_t tot16 = (int64_t)((out_dim + 15) / 16) * KT * 512;
        pack_bf16_n16_kernel<<<(unsigned)((tot16 + CGNN_BLOCK - 1) / CGNN_BLOCK), CGNN_BLOCK, 0, st>>>(
            w, out_dim, ld, col0, ncols, KT, tot16, (__bf16*)packed);
        return check_hip(hipGetLastError(), "cgnn_pack_linear launch");
    }
    if (precision == CGNN_F16X2) {
        const int64_t tot2 = (int64_t)OT * KT * 1024 * 2;
        pack_f16x2_kernel<<<(unsigned)((tot2 + CGNN_BLOCK - 1) / CGNN_BLOCK), CGNN_BLOCK, 0, st>>>(
            w, out_dim, ld, col0, ncols, KT, tot2, (_Float16*)packed);
        return check_hip(hipGetLastError(), "cgnn_pack_linear launch");
    }
    if (precision == CGNN_F32X3) {
        const int64_t tot3 = (int64_t)OT * KT * 1024 * 3;
        pack_f32x3_kernel<<<(unsigned)((tot3 + CGNN_BLOCK - 1) / CGNN_BLOCK), CGNN_BLOCK, 0, st>>>(
            w, out_dim, ld, col0, ncols, KT, tot3, (__bf16*)packed);
        return check_hip(hipGetLastError(), "cgnn_pack_linear launch");
    }
    const int64_t total = (int64_t)OT * KT * 1024;
    const int blocks = (int)((total + CGNN_BLOCK - 1) / CGNN_BLOCK);
    if (precision == CGNN_F32)
        pack_f32_kernel<<<blocks, CGNN_BLOCK, 0, st>>>(w, out_dim, ld, col0, ncols, KT, total, (float*)packed);
    else if (precision == CGNN_BF16)
        pack_bf16_kernel<<<blocks, CGNN_BLOCK, 0, st>>>(w, out_dim, ld, col0, ncols, KT, total, (__bf16*)packed);
    else {
        set_error("cgnn_pack_linear: unknown precision %d", precision);
        return CGNN_ERR_INVALID_ARG;
    }
    return check_hip(hipGetLastError(), "cgnn_pack_linear launch");
}

int64_t cgnn_tiled_rows(int64_t n) { return n <= 0 ? 0 : ((n + 31) / 32) * 32; }

int cgnn_relayout(const float* src, int32_t from, float* dst, int32_t to, int64_t n, int32_t width, void* stream) {
    if (!src || !dst || n < 0 || width <= 0 || (from != CGNN_ROWS && from != CGNN_TILED32) ||
        (to != CGNN_ROWS && to != CGNN_TILED32)) {
        set_error("cgnn_relayout: invalid argument");
        return CGNN_ERR_INVALID_ARG;
    }
    if ((from == CGNN_TILED32 || to == CGNN_TILED32) && width % 32 != 0) {
        set_error("cgnn_relayout: CGNN_TILED32 needs width %% 32 == 0 (got %d)", width);
        return CGNN_ERR_UNSUPPORTED;
    }
    if (width % 4 != 0) {
        set_error("cgnn_relayout: width %d is not a multiple of 4", width);
        return CGNN_ERR_UNSUPPORTED;
    }
    if (n == 0) return CGNN_OK;
    const int64_t n_pad = to == CGNN_TILED32 ? cgnn_tiled_rows(n) : n;
    relayout_kernel<<<blocks_for(n_pad * (width / 4), 32), CGNN_BLOCK, 0, (hipStream_t)stream>>>(
        src, from == CGNN_TILED32, dst, to == CGNN_TILED32, n, n_pad, width / 4);
    return check_hip(hipGetLastError(), "cgnn_relayout launch");
}

int cgnn_aggregate(const float* table, int32_t table_layout, const int32_t* gather, const int32_t* dst,
                   int64_t num_edges, int32_t fixed_k, int64_t num_nodes, int32_t width, float* out, void* stream) {
    if (!table || !out || num_edges < 0 || num_nodes < 0 || width <= 0 || fixed_k < 0) {
        set_error("cgnn_aggregate: invalid argument");
        return CGNN_ERR_INVALID_ARG;
    }
    if (width % 4 != 0) {
        set_error("cgnn_aggregate: width %d is not a multiple of 4", width);
        return CGNN_ERR_UNSUPPORTED;
    }
    const int tiled = table_layout == CGNN_TILED32;
    if (table_layout != CGNN_ROWS && !tiled) {
        set_error("cgnn_aggregate: unknown table_layout %d", table_layout);
        return CGNN_ERR_INVALID_ARG;
    }
    if (tiled && (gather != nullptr || width % 32 != 0)) {
        set_error("cgnn_aggregate: CGNN_TILED32 tables are per-edge messages (gather == NULL, width %% 32 == 0)");
        return CGNN_ERR_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    const int chunks = width / 4;
    if (num_nodes == 0) return CGNN_OK;
    if (fixed_k > 0) {
        if (num_edges != num_nodes * fixed_k) {
            set_error("cgnn_aggregate: fixed_k=%d needs num_edges == num_nodes*k (%lld vs %lld)", fixed_k,
                      (long long)num_edges, (long long)(num_nodes * fixed_k));
            return CGNN_ERR_INVALID_ARG;
        }
        if (tiled)
            aggregate_fixedk_tiled_kernel<<<blocks_for(num_nodes * chunks, 64), CGNN_BLOCK, 0, st>>>(
                table, fixed_k, num_nodes, chunks, out);
        else
            aggregate_fixedk_kernel<<<blocks_for_xcd(num_nodes * chunks, 64), CGNN_BLOCK, 0, st>>>(
                table, gather, fixed_k, num_nodes, chunks, out);
        return check_hip(hipGetLastError(), "cgnn_aggregate(fixed_k) launch");
    }
    if (!dst && num_edges > 0) {
        set_error("cgnn_aggregate: dst is required when fixed_k == 0");
        return CGNN_ERR_INVALID_ARG;
    }
    int rc = check_hip(hipMemsetAsync(out, 0, (size_t)num_nodes * width * sizeof(float), st), "cgnn_aggregate memset");
    if (rc != CGNN_OK || num_edges == 0) return rc;
    if (chunks == 8 || chunks == 16 || chunks == 32 || chunks == 64) {
        // widths 32 / 64 / 128 / 256: contiguous 256-byte atomic instructions (aggregate_scatter_kernel)
        const int epg = 32;
        const int gpw = 64 / chunks;
        const int64_t groups = (num_edges + epg - 1) / epg;
        const int64_t waves = (groups + gpw - 1) / gpw;
        const int64_t blocks = (waves + CGNN_WAVES_PER_BLOCK - 1) / CGNN_WAVES_PER_BLOCK;
#define CGNN_SCATTER(LPR) \
    aggregate_scatter_kernel<LPR><<<(unsigned)blocks, CGNN_BLOCK, 0, st>>>(table, tiled, gather, dst, num_edges, epg, out)
        if (chunks == 8) CGNN_SCATTER(8);
        else if (chunks == 16) CGNN_SCATTER(16);
        else if (chunks == 32) CGNN_SCATTER(32);
        else CGNN_SCATTER(64);
#undef CGNN_SCATTER
        return check_hip(hipGetLastError(), "cgnn_aggregate(scatter) launch");
    }
    const int epg = 16;
    const int64_t groups = (num_edges + epg - 1) / epg;
    const int64_t threads = groups * chunks;
    const int64_t blocks = (threads + CGNN_BLOCK - 1) / CGNN_BLOCK;
    aggregate_atomic_kernel<<<(unsigned)blocks, CGNN_BLOCK, 0, st>>>(table, tiled, gather, dst, num_edges, chunks, epg,
                                                                     out);
    return check_hip(hipGetLastError(), "cgnn_aggregate(atomic) launch");
}

int cgnn_gather_rows(const float* table, const int32_t* idx, int64_t n_idx, int32_t width, float* out, void* stream) {
    if (!table || !idx || !out || n_idx < 0 || width <= 0) {
        set_error("cgnn_gather_rows: invalid argument");
        return CGNN_ERR_INVALID_ARG;
    }
    if (n_idx == 0) return CGNN_OK;
    hipStream_t st = (hipStream_t)stream;
    if (width % 4 == 0)
        gather_rows_kernel<<<blocks_for(n_idx * (width / 4), 32), CGNN_BLOCK, 0, st>>>(table, idx, n_idx, width / 4, out);
    else
        gather_rows_scalar_kernel<<<blocks_for(n_idx * width, 32), CGNN_BLOCK, 0, st>>>(table, idx, n_idx, width, out);
    return check_hip(hipGetLastError(), "cgnn_gather_rows launch");
}

int cgnn_scatter_rows(const float* rows, const int32_t* idx, int64_t n_idx, int32_t width, float* table,
                      void* stream) {
    if (!table || !idx || !rows || n_idx < 0 || width <= 0) {
        set_error("cgnn_scatter_rows: invalid argument");
        return CGNN_ERR_INVALID_ARG;
    }
    if (n_idx == 0) return CGNN_OK;
    hipStream_t st = (hipStream_t)stream;
    if (width % 4 == 0)
        scatter_rows_kernel<<<blocks_for(n_idx * (width / 4), 32), CGNN_BLOCK, 0, st>>>(rows, idx, n_idx, width / 4,
                                                                                   table);
    else
        scatter_rows_scalar_kernel<<<blocks_for(n_idx * width, 32), CGNN_BLOCK, 0, st>>>(rows, idx, n_idx, width,
                                                                                     table);
    return check_hip(hipGetLastError(), "cgnn_scatter_rows launch");
}

int cgnn_window_features(const float* pos_seq, const float* temp_seq, const float* pos_noise, const float* temp_noise,
                         int32_t window, int64_t n, float box_size, float dt, float vel_mean, float vel_std,
                         float temp_mean, float temp_std, float* x, float* recent_pos, void* stream) {
    if (!pos_seq || !temp_seq || !x || !recent_pos || window < 2 || n < 0 || !(box_size > 0.f) || dt == 0.f ||
        vel_std == 0.f || temp_std == 0.f) {
        set_error("cgnn_window_features: invalid argument");
        return CGNN_ERR_INVALID_ARG;
    }
    if (n == 0) return CGNN_OK;
    window_features_kernel<<<(unsigned)((n + CGNN_BLOCK - 1) / CGNN_BLOCK), CGNN_BLOCK, 0, (hipStream_t)stream>>>(
        pos_seq, temp_seq, pos_noise, temp_noise, window, n, box_size, dt, vel_mean, vel_std, temp_mean, temp_std, x,
        recent_pos);
    return check_hip(hipGetLastError(), "cgnn_window_features launch");
}

int cgnn_segment_colsum(const float* acc, const int32_t* batch, int64_t n, int32_t width, int32_t num_graphs,
                        double* sums, void* stream) {
    if (!acc || !sums || n < 0 || width <= 0 || num_graphs <= 0) {
        set_error("cgnn_segment_colsum: invalid argument");
        return CGNN_ERR_INVALID_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    int rc = check_hip(hipMemsetAsync(sums, 0, (size_t)num_graphs * width * sizeof(double), st),
                       "cgnn_segment_colsum memset");
    if (rc != CGNN_OK || n == 0) return rc;
    const int64_t blocks = (n + CGNN_COLSUM_ROWS - 1) / CGNN_COLSUM_ROWS;
    segment_colsum_kernel<<<(unsigned)blocks, CGNN_BLOCK, 0, st>>>(acc, batch, n, width, sums);
    return check_hip(hipGetLastError(), "cgnn_segment_colsum launch");
}

}  // extern "C"
