// Grouped (CSR) form of an edge list and the gather-sum over it: cgnn_csr_build, cgnn_aggregate_csr.
//
// The forward aggregation reads a receiver-sorted, fixed in-degree list (cgnn_aggregate, fixed_k).  Its transpose
// -- the backward of `propagate` (reference graph_network.py:92 under autograd), which sums the receivers'
// gradients at every *sender* -- has a variable degree per row; scattering it with float atomics cost 27 ms at
// 10^6 particles, k = 16, against 0.7 ms for the forward.  Grouping the edges by sender once per graph makes the
// transpose the same kind of atomic-free gather as the forward, and its summation order deterministic.
#include "scan.hpp"

namespace cgnn {

struct CsrLayout {
    size_t off_count, off_cursor, off_bsum, total;
};
static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
static CsrLayout csr_layout(int64_t rows) {
    CsrLayout L;
    size_t off = 0;
    L.off_count = off;  off = align256(off + (size_t)(rows + 1) * 4);
    L.off_cursor = off; off = align256(off + (size_t)(rows + 1) * 4);
    L.off_bsum = off;   off = align256(off + (size_t)(scan_blocks(rows + 1) + 1) * 4);
    L.total = off;
    return L;
}

__global__ void csr_count_kernel(const int32_t* __restrict__ key, int64_t ne, int64_t rows, int32_t* __restrict__ count,
                                 int32_t* __restrict__ bad) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ne) return;
    const int32_t k = key[e];
    if (k < 0 || k >= rows) {
        *bad = 1;
        return;
    }
    atomicAdd(&count[k], 1);
}

__global__ void csr_fill_kernel(const int32_t* __restrict__ key, const int32_t* __restrict__ val, int64_t ne,
                                int64_t rows, const int32_t* __restrict__ row_ptr, int32_t* __restrict__ cursor,
                                int32_t* __restrict__ col) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ne) return;
    const int32_t k = key[e];
    if (k < 0 || k >= rows) return;
    col[row_ptr[k] + atomicAdd(&cursor[k], 1)] = val ? val[e] : (int32_t)e;
}

// The fill order within a row depends on atomic timing; sorting each row (ascending value) makes the layout --
// and with it the summation order of cgnn_aggregate_csr -- reproducible.  Rows are short (the mean is k).
__device__ __forceinline__ void sift_down(int32_t* a, int start, int end) {
    int root = start;
    while (2 * root + 1 <= end) {
        int child = 2 * root + 1;
        if (child + 1 <= end && a[child] < a[child + 1]) ++child;
        if (a[root] >= a[child]) return;
        const int32_t t = a[root];
        a[root] = a[child];
        a[child] = t;
        root = child;
    }
}
__global__ void csr_sort_rows_kernel(const int32_t* __restrict__ row_ptr, int64_t rows, int32_t* __restrict__ col) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    int32_t* a = col + row_ptr[i];
    const int len = row_ptr[i + 1] - row_ptr[i];
    if (len <= 48) {                       // insertion sort
        for (int j = 1; j < len; ++j) {
            const int32_t v = a[j];
            int p = j - 1;
            while (p >= 0 && a[p] > v) {
                a[p + 1] = a[p];
                --p;
            }
            a[p + 1] = v;
        }
    } else {                               // heap sort: O(len log len) for the rare hub row
        for (int s = (len - 2) / 2; s >= 0; --s) sift_down(a, s, len - 1);
        for (int end = len - 1; end > 0; --end) {
            const int32_t t = a[end];
            a[end] = a[0];
            a[0] = t;
            sift_down(a, 0, end - 1);
        }
    }
}

// out[i] = (add1 ? add1[i] : 0) + (add2 ? add2[i] : 0) + sum_{p in [row_ptr[i], row_ptr[i+1])} table[col[p]]: one thread
// per (row, 16-byte chunk), ascending p.  The addends let the backward of a residual round write
// dx_i = dx_{i+1} + du1 + A^T du2 in one pass (as separate elementwise adds they were 3 GB of traffic per round at 1 M
// particles); out may alias add1 or add2 (each element is read, then written, by the one thread that owns it).
__global__ __launch_bounds__(CGNN_BLOCK) void aggregate_csr_kernel(const float* __restrict__ table,
                                                                   const int32_t* __restrict__ row_ptr,
                                                                   const int32_t* __restrict__ col, int64_t rows,
                                                                   int chunks, const float* add1, const float* add2,
                                                                   float* out) {
    const int64_t total = rows * chunks;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = w / chunks;
        const int c = (int)(w - i * chunks);
        const int p0 = row_ptr[i], p1 = row_ptr[i + 1];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        int p = p0;
        for (; p + 3 < p1; p += 4) {       // four independent gathers in flight
            const int32_t j0 = col[p], j1 = col[p + 1], j2 = col[p + 2], j3 = col[p + 3];
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(table + ((int64_t)j0 * chunks + c) * 4);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(table + ((int64_t)j1 * chunks + c) * 4);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(table + ((int64_t)j2 * chunks + c) * 4);
            const f32x4 v3 = *reinterpret_cast<const f32x4*>(table + ((int64_t)j3 * chunks + c) * 4);
            acc += v0;
            acc += v1;
            acc += v2;
            acc += v3;
        }
        for (; p < p1; ++p) acc += *reinterpret_cast<const f32x4*>(table + ((int64_t)col[p] * chunks + c) * 4);
        f32x4 base = {0.f, 0.f, 0.f, 0.f};      // (dx_{i+1} + du1) first, then the gathered sum: the order of the separate adds
        if (add1 != nullptr) base = *reinterpret_cast<const f32x4*>(add1 + (i * chunks + c) * 4);
        if (add2 != nullptr) base += *reinterpret_cast<const f32x4*>(add2 + (i * chunks + c) * 4);
        if (add1 != nullptr || add2 != nullptr) acc = base + acc;
        *reinterpret_cast<f32x4*>(out + (i * chunks + c) * 4) = acc;
    }
}

}  // namespace cgnn

using namespace cgnn;

extern "C" {

size_t cgnn_csr_workspace_bytes(int64_t num_rows) { return num_rows < 0 ? 0 : csr_layout(num_rows).total + 256; }

int cgnn_csr_build(const int32_t* key, const int32_t* val, int64_t num_edges, int64_t num_rows, int32_t* row_ptr,
                   int32_t* col, void* workspace, size_t workspace_bytes, void* stream) {
    if (num_edges < 0 || num_rows < 0 || !row_ptr || (num_edges > 0 && (!key || !col)) || !workspace ||
        num_edges > 0x7fffffffLL) {
        set_error("cgnn_csr_build: invalid argument");
        return CGNN_ERR_INVALID_ARG;
    }
    const CsrLayout L = csr_layout(num_rows);
    if (workspace_bytes < L.total + 256) {
        set_error("cgnn_csr_build: workspace of %zu bytes, need %zu", workspace_bytes, L.total + 256);
        return CGNN_ERR_INVALID_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    int32_t* count = (int32_t*)(ws + L.off_count);
    int32_t* cursor = (int32_t*)(ws + L.off_cursor);
    int32_t* bsum = (int32_t*)(ws + L.off_bsum);
    int32_t* bad = (int32_t*)(ws + L.total);
    const int64_t m = num_rows + 1;       // count[num_rows] = 0 so that row_ptr[num_rows] = num_edges
    int rc = check_hip(hipMemsetAsync(ws, 0, L.total + 256, st), "cgnn_csr_build memset");
    if (rc) return rc;
    const unsigned eb = (unsigned)((num_edges + CGNN_BLOCK - 1) / CGNN_BLOCK);
    if (num_edges > 0) csr_count_kernel<<<eb, CGNN_BLOCK, 0, st>>>(key, num_edges, num_rows, count, bad);
    exclusive_scan_i32(count, m, bsum, row_ptr, st);
    if (num_edges > 0) {
        csr_fill_kernel<<<eb, CGNN_BLOCK, 0, st>>>(key, val, num_edges, num_rows, row_ptr, cursor, col);
        if (num_rows > 0)
            csr_sort_rows_kernel<<<(unsigned)((num_rows + CGNN_BLOCK - 1) / CGNN_BLOCK), CGNN_BLOCK, 0, st>>>(
                row_ptr, num_rows, col);
    }
    rc = check_hip(hipGetLastError(), "cgnn_csr_build launches");
    if (rc) return rc;
    int32_t bad_host = 0;
    rc = check_hip(hipMemcpyAsync(&bad_host, bad, 4, hipMemcpyDeviceToHost, st), "cgnn_csr_build readback");
    if (rc) return rc;
    rc = check_hip(hipStreamSynchronize(st), "cgnn_csr_build sync");
    if (rc) return rc;
    if (bad_host) {
        set_error("cgnn_csr_build: a key lies outside [0, %lld)", (long long)num_rows);
        return CGNN_ERR_INVALID_ARG;
    }
    return CGNN_OK;
}

int cgnn_aggregate_csr_add(const float* table, const int32_t* row_ptr, const int32_t* col, int64_t num_rows,
                           int32_t width, const float* add1, const float* add2, float* out, void* stream);

int cgnn_aggregate_csr(const float* table, const int32_t* row_ptr, const int32_t* col, int64_t num_rows, int32_t width,
                       float* out, void* stream) {
    return cgnn_aggregate_csr_add(table, row_ptr, col, num_rows, width, nullptr, nullptr, out, stream);
}

int cgnn_aggregate_csr_add(const float* table, const int32_t* row_ptr, const int32_t* col, int64_t num_rows,
                           int32_t width, const float* add1, const float* add2, float* out, void* stream) {
    if (!table || !row_ptr || !out || num_rows < 0 || width <= 0) {
        set_error("cgnn_aggregate_csr: invalid argument");
        return CGNN_ERR_INVALID_ARG;
    }
    if (width % 4 != 0) {
        set_error("cgnn_aggregate_csr: width %d is not a multiple of 4", width);
        return CGNN_ERR_UNSUPPORTED;
    }
    if (num_rows == 0) return CGNN_OK;
    const int chunks = width / 4;
    const int64_t total = num_rows * chunks;
    int64_t blocks = (total + CGNN_BLOCK - 1) / CGNN_BLOCK;
    if (blocks > (1 << 20)) blocks = 1 << 20;
    aggregate_csr_kernel<<<(unsigned)blocks, CGNN_BLOCK, 0, (hipStream_t)stream>>>(table, row_ptr, col, num_rows, chunks,
                                                                                   add1, add2, out);
    return check_hip(hipGetLastError(), "cgnn_aggregate_csr launch");
}

}  // extern "C"
