// Exclusive scan of int32 counters (three small kernels), shared by the k-NN cell grid and the CSR builder.
#pragma once
#include "cgnn_common.hpp"

#define CGNN_SCAN_ITEMS 2048  // per block (256 threads x 8)

namespace cgnn {

// ---- exclusive scan of `count[0..m)` into `start[0..m)` (three small kernels) ----
static __global__ void scan_block_sums_kernel(const int32_t* __restrict__ in, int64_t m, int32_t* __restrict__ bsum) {
    __shared__ int red[CGNN_BLOCK];
    const int64_t base = (int64_t)blockIdx.x * CGNN_SCAN_ITEMS;
    int s = 0;
    for (int j = 0; j < CGNN_SCAN_ITEMS / CGNN_BLOCK; ++j) {
        const int64_t i = base + j * CGNN_BLOCK + threadIdx.x;
        if (i < m) s += in[i];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = CGNN_BLOCK / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) bsum[blockIdx.x] = red[0];
}

static __global__ void scan_block_offsets_kernel(int32_t* __restrict__ bsum, int nblk) {
    // single thread block, serial over <= 8193 entries: negligible
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < nblk; ++i) {
            const int v = bsum[i];
            bsum[i] = run;
            run += v;
        }
    }
}

static __global__ void scan_apply_kernel(const int32_t* __restrict__ in, int64_t m, const int32_t* __restrict__ bsum,
                                  int32_t* __restrict__ out) {
    __shared__ int part[CGNN_BLOCK];
    const int64_t base = (int64_t)blockIdx.x * CGNN_SCAN_ITEMS + (int64_t)threadIdx.x * (CGNN_SCAN_ITEMS / CGNN_BLOCK);
    int v[CGNN_SCAN_ITEMS / CGNN_BLOCK];
    int s = 0;
#pragma unroll
    for (int j = 0; j < CGNN_SCAN_ITEMS / CGNN_BLOCK; ++j) {
        v[j] = (base + j < m) ? in[base + j] : 0;
        s += v[j];
    }
    part[threadIdx.x] = s;
    __syncthreads();
    // Hillis-Steele inclusive scan of the 256 per-thread totals
    for (int off = 1; off < CGNN_BLOCK; off <<= 1) {
        int t = ((int)threadIdx.x >= off) ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    int run = bsum[blockIdx.x] + part[threadIdx.x] - s;
#pragma unroll
    for (int j = 0; j < CGNN_SCAN_ITEMS / CGNN_BLOCK; ++j) {
        if (base + j < m) out[base + j] = run;
        run += v[j];
    }
}

// out[i] = sum_{j<i} in[j] for i in [0, m); bsum: scratch of scan_blocks(m) + 1 ints.
static inline int64_t scan_blocks(int64_t m) { return (m + CGNN_SCAN_ITEMS - 1) / CGNN_SCAN_ITEMS; }
static inline void exclusive_scan_i32(const int32_t* in, int64_t m, int32_t* bsum, int32_t* out, hipStream_t st) {
    const int sblk = (int)scan_blocks(m);
    scan_block_sums_kernel<<<sblk, CGNN_BLOCK, 0, st>>>(in, m, bsum);
    scan_block_offsets_kernel<<<1, 64, 0, st>>>(bsum, sblk);
    scan_apply_kernel<<<sblk, CGNN_BLOCK, 0, st>>>(in, m, bsum, out);
}

}  // namespace cgnn
