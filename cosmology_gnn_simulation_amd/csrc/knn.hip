// cgnn_knn_periodic: exact periodic k-nearest-neighbour graph on the device.
//
// Replaces reference data_utils.py:9-33 (27 ghost copies), :148-152
// (torch_cluster.knn on the 27N-point set + index swap + mapping) and :162-164
// (edge features).  The 27N extended set is never built: particles are binned
// into a uniform cell grid over [0, L)^3 and a query walks cubic shells of cells
// around its own cell, wrapping cell coordinates periodically; the wrap count per
// axis (-1, 0, +1) is exactly the reference's shift, so each candidate is the
// image  fl32(pos + shift)  the reference would have put in its extended array.
//
// Ordering contract (also oracle/cpu_ref.py:knn_extended): neighbours ascend by
// (d2, image index) where d2 is the float32 squared distance with one rounding
// per operation, summed x, y, z (nanoflann's L2 adaptor for dim 3), and
// image index = shift_id * N + particle, shift_id in cartesian_prod order
// (x slowest; the centre is 13).  The query itself therefore comes first.
#include "cgnn_common.hpp"
#include "scan.hpp"

namespace cgnn {

struct KnnLayout {
    int G;                // cells per axis
    int64_t cells;        // Gp^3 cell slots, Gp = G rounded up to a power of two (Morton-coded ids)
    size_t off_count, off_start, off_cursor, off_bsum, off_cellof, off_sorted, total;
};

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }


static KnnLayout knn_layout(int64_t n) {
    KnnLayout L;
    int G = (int)floor(cbrt((double)n / 2.0));
    if (G < 1) G = 1;
    if (G > 256) G = 256;
    L.G = G;
    int Gp = 1;
    while (Gp < G) Gp <<= 1;
    L.cells = (int64_t)Gp * Gp * Gp;
    size_t off = 0;
    L.off_count = off;  off = align256(off + (size_t)(L.cells + 1) * 4);
    L.off_start = off;  off = align256(off + (size_t)(L.cells + 1) * 4);
    L.off_cursor = off; off = align256(off + (size_t)(L.cells + 1) * 4);
    const size_t nblk = (size_t)((L.cells + 1 + CGNN_SCAN_ITEMS - 1) / CGNN_SCAN_ITEMS);
    L.off_bsum = off;   off = align256(off + (nblk + 1) * 4);
    L.off_cellof = off; off = align256(off + (size_t)n * 4);
    L.off_sorted = off; off = align256(off + (size_t)n * 16);
    L.total = off;
    return L;
}

// Cells are numbered along a Z-order (Morton) curve, so the cell-sorted particle order -- which the engine
// adopts as its node numbering -- keeps 3-D neighbours close in memory in all three directions (a row-major
// cell order leaves x-neighbours a whole slab apart, and the sender gathers then miss L2).
__device__ __forceinline__ unsigned spread3(unsigned v) {   // 10 bits -> every third bit
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__device__ __forceinline__ int morton3(int x, int y, int z) {
    return (int)((spread3((unsigned)x) << 2) | (spread3((unsigned)y) << 1) | spread3((unsigned)z));
}

__device__ __forceinline__ int cell_coord(float p, float inv_h, int G) {
    int c = (int)floorf(p * inv_h);
    c = c < 0 ? 0 : c;
    return c >= G ? G - 1 : c;
}

__global__ void knn_count_kernel(const float* __restrict__ pos, int64_t n, float inv_h, int G,
                                 int32_t* __restrict__ cell_of, int32_t* __restrict__ count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int cx = cell_coord(pos[3 * i + 0], inv_h, G), cy = cell_coord(pos[3 * i + 1], inv_h, G),
              cz = cell_coord(pos[3 * i + 2], inv_h, G);
    const int cell = morton3(cx, cy, cz);
    cell_of[i] = cell;
    atomicAdd(&count[cell], 1);
}

__global__ void knn_fill_kernel(const float* __restrict__ pos, int64_t n, const int32_t* __restrict__ cell_of,
                                const int32_t* __restrict__ start, int32_t* __restrict__ cursor,
                                float4* __restrict__ sorted) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int cell = cell_of[i];
    const int slot = start[cell] + atomicAdd(&cursor[cell], 1);
    sorted[slot] = make_float4(pos[3 * i + 0], pos[3 * i + 1], pos[3 * i + 2], __int_as_float((int)i));
}

#define CGNN_KNN_IDX_BITS 27
#define CGNN_KNN_IDX_MASK ((1u << CGNN_KNN_IDX_BITS) - 1u)

template <int K>
__global__ __launch_bounds__(CGNN_BLOCK) void knn_search_kernel(const float* __restrict__ pos, int64_t n, float box,
                                                                float h, float inv_h, int G,
                                                                const int32_t* __restrict__ start,
                                                                const int32_t* __restrict__ cell_of,
                                                                const float4* __restrict__ sorted,
                                                                const int32_t* __restrict__ query_ids, int64_t nq,
                                                                int k, int32_t* __restrict__ senders,
                                                                float* __restrict__ edge_attr) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nq) return;
    float qx, qy, qz;
    int64_t out_row;  // row of the outputs this query fills
    if (query_ids != nullptr) {
        const int q = query_ids[t];
        qx = pos[3 * (int64_t)q + 0];
        qy = pos[3 * (int64_t)q + 1];
        qz = pos[3 * (int64_t)q + 2];
        out_row = t;
    } else {
        const float4 s = sorted[t];  // walk queries in cell order: neighbouring lanes touch the same cells
        qx = s.x;
        qy = s.y;
        qz = s.z;
        out_row = __float_as_int(s.w);
    }
    const int cx = cell_coord(qx, inv_h, G), cy = cell_coord(qy, inv_h, G), cz = cell_coord(qz, inv_h, G);

    float bd[K];
    unsigned bi[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        bd[j] = __builtin_inff();
        bi[j] = 0xFFFFFFFFu;
    }
    const int rmax = 2 * G - 1;  // beyond this every one of the 27 images has been visited
    for (int r = 0; r <= rmax; ++r) {
        for (int dx = -r; dx <= r; ++dx) {
            const int ux = cx + dx;
            if (ux < -G || ux >= 2 * G) continue;
            const int sx = ux < 0 ? -1 : (ux >= G ? 1 : 0);
            const int wx = ux - sx * G;
            const float shx = (float)sx * box;
            const bool edge_x = (dx == -r) || (dx == r);
            for (int dy = -r; dy <= r; ++dy) {
                const int uy = cy + dy;
                if (uy < -G || uy >= 2 * G) continue;
                const int sy = uy < 0 ? -1 : (uy >= G ? 1 : 0);
                const int wy = uy - sy * G;
                const float shy = (float)sy * box;
                const bool edge_xy = edge_x || (dy == -r) || (dy == r);
                const int zstep = edge_xy ? 1 : (2 * r > 0 ? 2 * r : 1);  // interior columns: only the two end caps
                for (int dz = -r; dz <= r; dz += zstep) {
                    const int uz = cz + dz;
                    if (uz < -G || uz >= 2 * G) continue;
                    const int sz = uz < 0 ? -1 : (uz >= G ? 1 : 0);
                    const int wz = uz - sz * G;
                    const float shz = (float)sz * box;
                    const unsigned shift_id = (unsigned)((sx + 1) * 9 + (sy + 1) * 3 + (sz + 1));
                    const int cell = morton3(wx, wy, wz);
                    const int p0 = start[cell], p1 = start[cell + 1];
                    for (int p = p0; p < p1; ++p) {
                        const float4 c = sorted[p];
                        // image position exactly as the reference builds it: fl32(pos + shift)
                        const float ex = __fadd_rn(c.x, shx), ey = __fadd_rn(c.y, shy), ez = __fadd_rn(c.z, shz);
                        const float ddx = __fsub_rn(ex, qx), ddy = __fsub_rn(ey, qy), ddz = __fsub_rn(ez, qz);
                        const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(ddx, ddx), __fmul_rn(ddy, ddy)),
                                                   __fmul_rn(ddz, ddz));
                        const unsigned key = (shift_id << CGNN_KNN_IDX_BITS) | (unsigned)__float_as_int(c.w);
                        if (d2 < bd[K - 1] || (d2 == bd[K - 1] && key < bi[K - 1])) {
                            bd[K - 1] = d2;
                            bi[K - 1] = key;
#pragma unroll
                            for (int j = K - 1; j > 0; --j) {
                                const bool sw = bd[j] < bd[j - 1] || (bd[j] == bd[j - 1] && bi[j] < bi[j - 1]);
                                const float td = bd[j];
                                const unsigned ti = bi[j];
                                bd[j] = sw ? bd[j - 1] : td;
                                bi[j] = sw ? bi[j - 1] : ti;
                                bd[j - 1] = sw ? td : bd[j - 1];
                                bi[j - 1] = sw ? ti : bi[j - 1];
                            }
                        }
                    }
                }
            }
        }
        // Everything not yet visited lies outside the cube of cells [c-r, c+r]^3, i.e. at least r*h away
        // (minus float32 slack in the cell assignment and in fl32(pos + shift)).
        const float bound = (float)r * h * (1.0f - 1e-5f) - 1e-5f * box;
        float kth = bd[K - 1];
#pragma unroll
        for (int j = 0; j < K; ++j) kth = (j == k - 1) ? bd[j] : kth;  // static indices: bd stays in registers
        if (bound > 0.f && kth <= bound * bound) break;
    }
    const float px = qx, py = qy, pz = qz;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        if (j < k) {
            const int snd = (int)(bi[j] & CGNN_KNN_IDX_MASK);
            senders[out_row * k + j] = snd;
            if (edge_attr != nullptr) {
                // reference data_utils.py:162-164: mapped (un-shifted) sender minus receiver
                const float ax = __fsub_rn(pos[3 * (int64_t)snd + 0], px);
                const float ay = __fsub_rn(pos[3 * (int64_t)snd + 1], py);
                const float az = __fsub_rn(pos[3 * (int64_t)snd + 2], pz);
                const float nn = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(ax, ax), __fmul_rn(ay, ay)), __fmul_rn(az, az)));
                *reinterpret_cast<float4*>(edge_attr + (out_row * k + j) * 4) = make_float4(ax, ay, az, nn);
            }
        }
    }
}

__global__ void knn_perm_kernel(const float4* __restrict__ sorted, int64_t n, int32_t* __restrict__ perm) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) perm[i] = __float_as_int(sorted[i].w);
}

}  // namespace cgnn

using namespace cgnn;

extern "C" {

size_t cgnn_knn_workspace_bytes(int64_t n, int32_t k) {
    (void)k;
    if (n <= 0) return 256;
    return knn_layout(n).total;
}

int cgnn_knn_periodic(const float* pos, int64_t n, float box_size, int32_t k, const int32_t* query_ids, int64_t nq,
                      int32_t* senders, float* edge_attr, void* workspace, size_t workspace_bytes, void* stream) {
    if (!pos || !senders || !workspace || n <= 0 || k <= 0 || !(box_size > 0.f)) {
        set_error("cgnn_knn_periodic: invalid argument");
        return CGNN_ERR_INVALID_ARG;
    }
    if (k > 64) {
        set_error("cgnn_knn_periodic: k=%d > 64 is not compiled", k);
        return CGNN_ERR_UNSUPPORTED;
    }
    if ((int64_t)k > 27 * n) {
        set_error("cgnn_knn_periodic: k=%d exceeds the 27*n=%lld periodic images", k, (long long)(27 * n));
        return CGNN_ERR_INVALID_ARG;
    }
    if (n >= ((int64_t)1 << CGNN_KNN_IDX_BITS)) {
        set_error("cgnn_knn_periodic: n=%lld >= 2^%d particles per call is not supported", (long long)n,
                  CGNN_KNN_IDX_BITS);
        return CGNN_ERR_UNSUPPORTED;
    }
    if ((reinterpret_cast<uintptr_t>(workspace) & 15) != 0) {
        set_error("cgnn_knn_periodic: workspace must be 16-byte aligned");
        return CGNN_ERR_INVALID_ARG;
    }
    const KnnLayout L = knn_layout(n);
    if (workspace_bytes < L.total) {
        set_error("cgnn_knn_periodic: workspace %zu < required %zu bytes", workspace_bytes, L.total);
        return CGNN_ERR_WORKSPACE;
    }
    if (query_ids == nullptr) nq = n;
    if (nq < 0) {
        set_error("cgnn_knn_periodic: negative query count");
        return CGNN_ERR_INVALID_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    char* ws = reinterpret_cast<char*>(workspace);
    int32_t* count = reinterpret_cast<int32_t*>(ws + L.off_count);
    int32_t* start = reinterpret_cast<int32_t*>(ws + L.off_start);
    int32_t* cursor = reinterpret_cast<int32_t*>(ws + L.off_cursor);
    int32_t* bsum = reinterpret_cast<int32_t*>(ws + L.off_bsum);
    int32_t* cell_of = reinterpret_cast<int32_t*>(ws + L.off_cellof);
    float4* sorted = reinterpret_cast<float4*>(ws + L.off_sorted);
    const int G = L.G;
    const float h = box_size / (float)G;
    const float inv_h = (float)G / box_size;
    const int64_t m = L.cells + 1;  // count[cells] = 0 so that start[cells] = n
    int rc = check_hip(hipMemsetAsync(count, 0, (size_t)m * 4, st), "knn memset count");
    if (rc) return rc;
    rc = check_hip(hipMemsetAsync(cursor, 0, (size_t)m * 4, st), "knn memset cursor");
    if (rc) return rc;
    const unsigned nb = (unsigned)((n + CGNN_BLOCK - 1) / CGNN_BLOCK);
    knn_count_kernel<<<nb, CGNN_BLOCK, 0, st>>>(pos, n, inv_h, G, cell_of, count);
    exclusive_scan_i32(count, m, bsum, start, st);
    knn_fill_kernel<<<nb, CGNN_BLOCK, 0, st>>>(pos, n, cell_of, start, cursor, sorted);
    rc = check_hip(hipGetLastError(), "cgnn_knn_periodic build launches");
    if (rc) return rc;
    if (nq == 0) return CGNN_OK;
    const unsigned qb = (unsigned)((nq + CGNN_BLOCK - 1) / CGNN_BLOCK);
#define CGNN_KNN_LAUNCH(KK)                                                                                       \
    knn_search_kernel<KK><<<qb, CGNN_BLOCK, 0, st>>>(pos, n, box_size, h, inv_h, G, start, cell_of, sorted, query_ids, nq, \
                                                     k, senders, edge_attr)
    if (k <= 8)
        CGNN_KNN_LAUNCH(8);
    else if (k <= 16)
        CGNN_KNN_LAUNCH(16);
    else if (k <= 32)
        CGNN_KNN_LAUNCH(32);
    else
        CGNN_KNN_LAUNCH(64);
#undef CGNN_KNN_LAUNCH
    return check_hip(hipGetLastError(), "cgnn_knn_periodic search launch");
}

int cgnn_knn_sorted_order(const void* workspace, int64_t n, int32_t* perm, void* stream) {
    if (!workspace || !perm || n <= 0) {
        set_error("cgnn_knn_sorted_order: invalid argument");
        return CGNN_ERR_INVALID_ARG;
    }
    const KnnLayout L = knn_layout(n);
    const float4* sorted = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(workspace) + L.off_sorted);
    knn_perm_kernel<<<(unsigned)((n + CGNN_BLOCK - 1) / CGNN_BLOCK), CGNN_BLOCK, 0, (hipStream_t)stream>>>(sorted, n,
                                                                                                        perm);
    return check_hip(hipGetLastError(), "cgnn_knn_sorted_order launch");
}

}  // extern "C"
