// cgnn_weight_grad_x3: the parameter-gradient reduction dW = g^T a of a 128 x 128 Linear (reference train.py:263-265,
// what autograd computes for every nn.Linear of graph_network.py:15-32) on the bf16 matrix cores, deterministic.
//
// cgnn_weight_grad (backward.hip) runs v_mfma_f32_32x32x2_f32 and adds row chunks with float atomics: 20 ms of an 80 ms
// training step at 1 M particles, and a sum whose order changes from run to run.  Here
//   * both operands are split into three bf16 terms in registers (x = x1 + x2 + x3, cgnn_common.hpp) and the six
//     products of weight >= 2^-16 go through v_mfma_f32_32x32x16_bf16 with f32 accumulation: f32-level error at the
//     bf16 rate (gradients of 1e-8 keep their bits: bf16 has the f32 exponent range, which is why the two-fp16-term
//     form is not used here);
//   * one wave owns the WHOLE 128 x 128 product of its row range in 256 accumulation registers, so every g / a value is
//     loaded and split once per wave (the f32 kernel splits the output into 32 x 128 items: four loads of each a row);
//     a lane loads 16 bytes of a row -- columns 4 i .. 4 i + 3 -- so an instruction reads two whole 512-byte rows, and
//     the four components are the lane's entries of four column-interleaved 32-wide MFMA tiles (tile t = columns
//     {4 i + t});
//   * the four waves of a workgroup add their partial products through LDS (wave 1, 2, 3 onto wave 0, in that order),
//     every workgroup writes one partial to a workspace and a second kernel adds the 256 partials in a fixed order:
//     the result does not depend on scheduling (same bits every run).
#include <string.h>

#include "n16.hpp"

namespace cgnn {

#define CGNN_WGX3_WAVES 1024                      // waves = row ranges (256 workgroups of four)
#define CGNN_WGX3_PARTS 256                       // partial products in the workspace: one per workgroup
#define CGNN_WGX3_PART_FLOATS (16 * 16 * 64 + 128)   // 16 tiles x 16 registers x 64 lanes, then the 128 column sums of g
#define CGNN_WGX3_GROUPS 4                        // second stage: partials [64 q, 64 q + 64) per thread of group q

typedef float f32x4w __attribute__((ext_vector_type(4)));

// three bf16 terms of eight values v[j][t], j = 0..7 (the k index of the lane), as MFMA operands
__device__ __forceinline__ void split8_x3(const f32x4w (&v)[8], int t, bf16x8 (&out)[3]) {
    u32x4 p[3];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const float a = v[2 * h][t], b = v[2 * h + 1][t];
        const unsigned w1 = pack_bf16(a, b);
        const float ra = a - __builtin_bit_cast(float, w1 << 16), rb = b - __builtin_bit_cast(float, w1 & 0xffff0000u);
        const unsigned w2 = pack_bf16(ra, rb);
        const float sa = ra - __builtin_bit_cast(float, w2 << 16), sb = rb - __builtin_bit_cast(float, w2 & 0xffff0000u);
        p[0][h] = w1;
        p[1][h] = w2;
        p[2][h] = pack_bf16(sa, sb);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) out[q] = __builtin_bit_cast(bf16x8, p[q]);
}

__global__ __launch_bounds__(256) void weight_grad_x3_kernel(const float* __restrict__ g, int ld_g,
                                                             const float* __restrict__ a, int ld_a, int64_t n,
                                                             int64_t rows_per_wave, float* __restrict__ part) {
    const int lane = threadIdx.x & 63, i = lane & 31, kk = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t W = (int64_t)blockIdx.x * 4 + wave;
    const int64_t r0 = W * rows_per_wave;
    const int64_t r1 = r0 + rows_per_wave < n ? r0 + rows_per_wave : n;
    f32x16 acc[4][4];
#pragma unroll
    for (int tg = 0; tg < 4; ++tg)
#pragma unroll
        for (int ta = 0; ta < 4; ++ta)
#pragma unroll
            for (int x = 0; x < 16; ++x) acc[tg][ta][x] = 0.f;
    f32x4w cs = {0.f, 0.f, 0.f, 0.f};
    const float* gp = g + 4 * i;
    const float* ap = a + 4 * i;
    const f32x4w zero = {0.f, 0.f, 0.f, 0.f};
    // Full 16-row steps load unconditionally (predicated loads made hipcc branch around every row and wait inside the
    // load sequence); the one partial step a row range can end with clamps its addresses and zeroes what it must not see.
    auto load_full = [&](int64_t rr, f32x4w (&gv)[8], f32x4w (&av)[8]) __attribute__((always_inline)) {
        const float* gq = gp + (rr + 8 * kk) * ld_g;       // k index 8 kk + j of the 16-row MFMA step
        const float* aq = ap + (rr + 8 * kk) * ld_a;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            gv[j] = *reinterpret_cast<const f32x4w*>(gq + (int64_t)j * ld_g);
            av[j] = *reinterpret_cast<const f32x4w*>(aq + (int64_t)j * ld_a);
        }
    };
    auto load_tail = [&](int64_t rr, f32x4w (&gv)[8], f32x4w (&av)[8]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t row = rr + 8 * kk + j;
            const bool ok = row < r1;
            const int64_t rc = ok ? row : r1 - 1;
            const f32x4w x = *reinterpret_cast<const f32x4w*>(gp + rc * ld_g);
            const f32x4w y = *reinterpret_cast<const f32x4w*>(ap + rc * ld_a);
            gv[j] = ok ? x : zero;
            av[j] = ok ? y : zero;
        }
    };
    auto step = [&](const f32x4w (&gv)[8], const bf16x8 (&A)[4][3], const bf16x8 (&B)[4][3]) __attribute__((always_inline)) {
#pragma unroll
        for (int tg = 0; tg < 4; ++tg)
#pragma unroll
            for (int ta = 0; ta < 4; ++ta) {
                f32x16 c = acc[tg][ta];                 // smallest terms first
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[tg][2], B[ta][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[tg][0], B[ta][2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[tg][1], B[ta][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[tg][1], B[ta][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[tg][0], B[ta][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[tg][0], B[ta][0], c, 0, 0, 0);
                acc[tg][ta] = c;
            }
    };
    if (r0 < r1) {       // (waves past the last row only write their zero partial)
        const int64_t full_end = r0 + (r1 - r0) / 16 * 16;
        f32x4w gv[8], av[8];
        if (r0 < full_end)
            load_full(r0, gv, av);
        else
            load_tail(r0, gv, av);
        for (int64_t rr = r0; rr < r1; rr += 16) {
            bf16x8 A[4][3], B[4][3];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                split8_x3(gv, t, A[t]);
                split8_x3(av, t, B[t]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) cs += gv[j];
            // the next step's rows travel under this step's 96 MFMAs
            if (rr + 32 <= full_end)
                load_full(rr + 16, gv, av);
            else if (rr + 16 < r1)
                load_tail(rr + 16, gv, av);
            step(gv, A, B);
        }
    }
    // the workgroup's four partial products meet in LDS: wave w = 1, 2, 3 in turn lays its accumulators down, wave 0 adds
    float* const red = reinterpret_cast<float*>(cgnn_smem);        // 16 x 16 x 64 floats + 4 x 64 column sums
#pragma unroll 1
    for (int w = 1; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int tg = 0; tg < 4; ++tg)
#pragma unroll
                for (int ta = 0; ta < 4; ++ta)
#pragma unroll
                    for (int x = 0; x < 16; ++x) red[((tg * 4 + ta) * 16 + x) * 64 + lane] = acc[tg][ta][x];
#pragma unroll
            for (int c = 0; c < 4; ++c) red[16 * 16 * 64 + c * 64 + lane] = cs[c];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int tg = 0; tg < 4; ++tg)
#pragma unroll
                for (int ta = 0; ta < 4; ++ta)
#pragma unroll
                    for (int x = 0; x < 16; ++x) acc[tg][ta][x] += red[((tg * 4 + ta) * 16 + x) * 64 + lane];
#pragma unroll
            for (int c = 0; c < 4; ++c) cs[c] += red[16 * 16 * 64 + c * 64 + lane];
        }
    }
    if (wave != 0) return;
    float* out = part + (int64_t)blockIdx.x * CGNN_WGX3_PART_FLOATS;
#pragma unroll
    for (int tg = 0; tg < 4; ++tg)
#pragma unroll
        for (int ta = 0; ta < 4; ++ta)
#pragma unroll
            for (int x = 0; x < 16; ++x) out[((tg * 4 + ta) * 16 + x) * 64 + lane] = acc[tg][ta][x];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float s = cs[c] + __shfl_xor(cs[c], 32);
        if (kk == 0) out[16 * 16 * 64 + c * 32 + i] = s;
    }
}

// element e of a partial: e < 16384: ((tg * 4 + ta) * 16 + x) * 64 + lane  ->  dw[4 m + tg][4 (lane & 31) + ta],
// m = (x & 3) + 8 (x >> 2) + 4 (lane >> 5)  (the 32 x 32 MFMA result layout);  e >= 16384: column sum c * 32 + i -> db[4 i + c]
__global__ __launch_bounds__(256) void weight_grad_x3_reduce_kernel(const float* __restrict__ part,
                                                                    float* __restrict__ dw, int ld_dw, int col0,
                                                                    float* __restrict__ db) {
    __shared__ float red[CGNN_WGX3_GROUPS][64];
    const int el = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + el;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (e < CGNN_WGX3_PART_FLOATS) {
        constexpr int PER = CGNN_WGX3_PARTS / CGNN_WGX3_GROUPS;
        const float* p = part + (int64_t)q * PER * CGNN_WGX3_PART_FLOATS + e;
#pragma unroll 4
        for (int k = 0; k < PER; k += 4)
#pragma unroll
            for (int u = 0; u < 4; ++u) s[u] += p[(int64_t)(k + u) * CGNN_WGX3_PART_FLOATS];
    }
    red[q][el] = (s[0] + s[1]) + (s[2] + s[3]);
    __syncthreads();
    if (q != 0 || e >= CGNN_WGX3_PART_FLOATS) return;
    const float total = (red[0][el] + red[1][el]) + (red[2][el] + red[3][el]);
    if (e < 16 * 16 * 64) {
        const int lane = e & 63, x = (e >> 6) & 15, ta = (e >> 10) & 3, tg = e >> 12;
        const int m = (x & 3) + 8 * (x >> 2) + 4 * (lane >> 5);
        dw[(int64_t)(4 * m + tg) * ld_dw + col0 + 4 * (lane & 31) + ta] += total;
    } else if (db != nullptr) {
        const int r = e - 16 * 16 * 64;
        db[4 * (r & 31) + (r >> 5)] += total;
    }
}

}  // namespace cgnn

using namespace cgnn;

extern "C" {

size_t cgnn_weight_grad_x3_workspace_bytes(void) {
    return (size_t)CGNN_WGX3_PARTS * CGNN_WGX3_PART_FLOATS * sizeof(float);
}

int cgnn_weight_grad_x3(const float* g, int32_t ld_g, const float* a, int32_t ld_a, int64_t n, float* dw, int32_t ld_dw,
                        int32_t col0, float* db, void* workspace, size_t workspace_bytes, void* stream) {
    if (!g || !a || !dw || !workspace || n < 0 || ld_g < 128 || ld_a < 128 || col0 < 0 || ld_dw < col0 + 128) {
        set_error("cgnn_weight_grad_x3: invalid argument");
        return CGNN_ERR_INVALID_ARG;
    }
    if ((ld_g & 3) || (ld_a & 3) || (reinterpret_cast<uintptr_t>(g) & 15) || (reinterpret_cast<uintptr_t>(a) & 15)) {
        set_error("cgnn_weight_grad_x3: g and a must be 16-byte aligned with leading dimensions that are multiples of 4 "
                  "(ld_g=%d ld_a=%d); use cgnn_weight_grad", ld_g, ld_a);
        return CGNN_ERR_UNSUPPORTED;
    }
    if (workspace_bytes < cgnn_weight_grad_x3_workspace_bytes()) {
        set_error("cgnn_weight_grad_x3: workspace of %zu bytes, need %zu", workspace_bytes,
                  cgnn_weight_grad_x3_workspace_bytes());
        return CGNN_ERR_INVALID_ARG;
    }
    if (n == 0) return CGNN_OK;
    hipStream_t st = (hipStream_t)stream;
    int64_t rows_per_wave = (n + CGNN_WGX3_WAVES - 1) / CGNN_WGX3_WAVES;
    rows_per_wave = (rows_per_wave + 15) / 16 * 16;
    float* part = reinterpret_cast<float*>(workspace);
    constexpr int lds = (16 * 16 * 64 + 4 * 64) * (int)sizeof(float);       // 65 KiB
    int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(weight_grad_x3_kernel), (size_t)(lds), "hipFuncSetAttribute(weight_grad_x3)");
    if (rc != CGNN_OK) return rc;
    weight_grad_x3_kernel<<<CGNN_WGX3_WAVES / 4, 256, lds, st>>>(g, ld_g, a, ld_a, n, rows_per_wave, part);
    rc = check_hip(hipGetLastError(), "cgnn_weight_grad_x3 launch");
    if (rc != CGNN_OK) return rc;
    weight_grad_x3_reduce_kernel<<<(CGNN_WGX3_PART_FLOATS + 63) / 64, 256, 0, st>>>(part, dw, ld_dw, col0, db);
    return check_hip(hipGetLastError(), "cgnn_weight_grad_x3 reduce launch");
}

}  // extern "C"
