// Backward of the node stream (reference train.py:263 `combined_loss.backward()` through
// graph_network.py:94-96,154-183): cgnn_mlp_backward, cgnn_weight_grad, cgnn_col_dot.  Exact f32 MFMA.
//
// In reference-faithful mode (PyG's default message) only the node encoder, the node models and the decoders
// receive gradient, all of them row-wise MLPs (+LayerNorm); the aggregation's transpose is
// cgnn_aggregate with src/dst swapped.  The data-gradient kernel below recomputes the forward of its 32-row tile
// instead of reading saved activations, and leaves the per-layer activations and pre-activation gradients in
// scratch matrices from which the weight gradients are plain row reductions.
#include <string.h>

#include "mlp_device.hpp"

namespace cgnn {

struct BwdBufs {
    float* h[CGNN_MAX_HIDDEN_LAYERS];
    float* g_a[CGNN_MAX_HIDDEN_LAYERS];
    float* g_o;
    float* zhat;
};

template <int T>
__device__ __forceinline__ void zero_tiles(f32x16 (&a)[T]) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) a[t][i] = 0.f;
}

// K1T / K2T: 32-feature tiles of the two input parts (K2T = 0: single input); HT hidden tiles; OT output tiles.
// PF / PB: arithmetic of the recomputed forward / of the gradient chain.  (F32, F32) exact; (F32X3, F32X3) three bf16
// terms; (F16X2, F32X3): the forward recomputation -- activations of O(1) behind a LayerNorm -- on two fp16 terms (half
// the matrix work), the gradients, which can be 1e-8, on three bf16 terms (f32 exponent range).
// CGNN_F32 (v_mfma_f32_32x32x2_f32, exact) or CGNN_F32X3 (f32 emulated by three bf16 terms, six bf16 MFMAs per
// product block, f32-level error: the activations and gradients stay f32 tiles, only the MFMA operands are split).
template <int PF, int PB, int K1T, int K2T, int HT, int OT, bool LN>
__global__ __launch_bounds__(CGNN_BLOCK) void mlp_backward_kernel(
    MlpDev f, MlpDev b, const void* f_w2, const void* b_w2, const float* __restrict__ u1, int ld1,
    const float* __restrict__ u2, int ld2, const float* __restrict__ dy, int ld_dy, int64_t n, BwdBufs buf,
    float* __restrict__ du1, int ld_du1, float* __restrict__ du2, int ld_du2) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int64_t tiles = (n + 31) / 32;
    constexpr int H = 32 * HT, OW = 32 * OT;
    constexpr int K2 = K2T > 0 ? K2T : 1;
    const int in1 = f.in_dim[0], out_dim = f.out_dim[f.nh];
    const bool in1_full = (in1 == 32 * K1T) && (ld1 % 4 == 0) && ((reinterpret_cast<uintptr_t>(u1) & 15) == 0);
    constexpr unsigned TILE_F = PF == CGNN_F32X3 ? 6144u : 4096u, TILE_B = PB == CGNN_F32X3 ? 6144u : 4096u;   // one packed 32 x 32 tile
    const BufW<PF> fw2(f_w2, K2 * HT * TILE_F);
    const BufW<PB> bw2(b_w2, K2 * HT * TILE_B);
    const TileRange tr = tile_range(tiles);
    for (int64_t tile = tr.first; tile < tr.end; tile += tr.stride) {
        const int64_t row = tile * 32 + r;
        const bool live = row < n;
        const int64_t rowc = live ? row : n - 1;
        // ------------------------------------------------------------ forward, recomputed
        Operand<PF, HT> oph;
        auto relu_store = [&](f32x16 (&acc)[HT], float* dst) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < HT; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][i] = acc[t][i] < 0.f ? 0.f : acc[t][i];     // NaN stays NaN (an fp16 overflow must show)
            if (live) store_rows_full<HT>(acc, dst + row * H, h);
            oph.template from_acc<false>(acc);
        };
        {
            f32x16 acc[HT];
            acc_fill_bias<HT>(acc, f.b[0], H, h);
            {
                f32x16 t1[K1T];
                if (in1_full)
                    load_rows_full<K1T>(t1, u1 + rowc * ld1, h);
                else
                    load_rows_ragged<K1T>(t1, u1 + rowc * ld1, in1, h);
                Operand<PF, K1T> op;
                op.template from_acc<false>(t1);
                dense<K1T, HT>(acc, op, WSel<PF, false>::get(f, 0), lane);
            }
            if (K2T > 0) {
                f32x16 t2[K2];
                load_rows_full<K2>(t2, u2 + rowc * ld2, h);
                Operand<PF, K2> op;
                op.template from_acc<false>(t2);
                dense<K2, HT>(acc, op, fw2, lane);
            }
            relu_store(acc, buf.h[0]);
        }
        for (int l = 1; l < f.nh; ++l) {
            f32x16 acc[HT];
            acc_fill_bias<HT>(acc, f.b[l], H, h);
            dense<HT, HT>(acc, oph, WSel<PF, false>::get(f, l), lane);
            relu_store(acc, buf.h[l]);
        }
        f32x16 g[OT];      // becomes dL/d(pre-LayerNorm output)
        {
            f32x16 out[OT];
            acc_fill_bias<OT>(out, f.b[f.nh], out_dim, h);
            dense<HT, OT>(out, oph, WSel<PF, false>::get(f, f.nh), lane);
            // -------------------------------------------------------- output gradient through LayerNorm
            if (out_dim == OW && ld_dy % 4 == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0)
                load_rows_full<OT>(g, dy + rowc * ld_dy, h);
            else
                load_rows_ragged<OT>(g, dy + rowc * ld_dy, out_dim, h);
            if (LN) {
                float s = 0.f;
#pragma unroll
                for (int t = 0; t < OT; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) s += out[t][i];
                s += __shfl_xor(s, 32);
                const float mean = s * (1.0f / OW);
                float q = 0.f;
#pragma unroll
                for (int t = 0; t < OT; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float d = out[t][i] - mean;
                        q += d * d;
                    }
                q += __shfl_xor(q, 32);
                const float rstd = 1.0f / sqrtf(q * (1.0f / OW) + 1e-5f);
                float m1 = 0.f, m2 = 0.f;
#pragma unroll
                for (int t = 0; t < OT; ++t)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const f32x4 gm = *reinterpret_cast<const f32x4*>(f.gamma + 32 * t + 8 * gq + 4 * h);
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const int i = 4 * gq + c;
                            const float z = (out[t][i] - mean) * rstd;
                            out[t][i] = z;                       // out now holds zhat
                            const float gz = g[t][i] * gm[c];
                            g[t][i] = gz;                        // g now holds dL/dzhat
                            m1 += gz;
                            m2 += gz * z;
                        }
                    }
                m1 += __shfl_xor(m1, 32);
                m2 += __shfl_xor(m2, 32);
                m1 *= (1.0f / OW);
                m2 *= (1.0f / OW);
                if (live) store_rows_full<OT>(out, buf.zhat + row * OW, h);
#pragma unroll
                for (int t = 0; t < OT; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) g[t][i] = rstd * (g[t][i] - m1 - out[t][i] * m2);
            }
        }
        if (live) store_rows_full<OT>(g, buf.g_o + row * OW, h);
        // ------------------------------------------------------------ backward through the hidden layers
        Operand<PB, HT> og;     // dL/d(pre-activation) of the layer being left
        // gh = W^T g of the layer above; mask by the (stored) activation's sign, keep as g_a[l], make it the next operand
        auto relu_backward = [&](f32x16 (&gh)[HT], int l) __attribute__((always_inline)) {
            f32x16 hv[HT];
            load_rows_full<HT>(hv, buf.h[l] + rowc * H, h);
#pragma unroll
            for (int t = 0; t < HT; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) gh[t][i] = hv[t][i] > 0.f ? gh[t][i] : 0.f;
            if (live) store_rows_full<HT>(gh, buf.g_a[l] + row * H, h);
            og.template from_acc<false>(gh);
        };
        {
            Operand<PB, OT> go;
            go.template from_acc<false>(g);
            f32x16 gh[HT];
            zero_tiles<HT>(gh);
            dense<OT, HT>(gh, go, WSel<PB, false>::get(b, f.nh), lane);     // W_nh^T
            relu_backward(gh, f.nh - 1);
        }
        for (int l = f.nh - 1; l >= 1; --l) {
            f32x16 gh[HT];
            zero_tiles<HT>(gh);
            dense<HT, HT>(gh, og, WSel<PB, false>::get(b, l), lane);        // W_l^T
            relu_backward(gh, l - 1);
        }
        // ------------------------------------------------------------ input gradients
        if (du1 != nullptr) {
            f32x16 gx[K1T];
            zero_tiles<K1T>(gx);
            dense<HT, K1T>(gx, og, WSel<PB, false>::get(b, 0), lane);       // W_0a^T
            if (live) {
                if (in1_full && ld_du1 % 4 == 0)
                    store_rows_full<K1T>(gx, du1 + row * ld_du1, h);
                else
                    store_rows_ragged<K1T>(gx, du1 + row * ld_du1, in1, h);
            }
        }
        if (K2T > 0 && du2 != nullptr) {
            f32x16 gx[K2];
            zero_tiles<K2>(gx);
            dense<HT, K2>(gx, og, bw2, lane);                                     // W_0b^T
            if (live) store_rows_full<K2>(gx, du2 + row * ld_du2, h);
        }
    }
}

// dw[o, col0+i] += sum_r g[r,o] a[r,i] (and db[o] += sum_r g[r,o]): per workgroup one chunk of rows; each wave
// owns one 32-wide slab of g's columns (an output tile) against G 32-wide slabs of a at once, so g is read once
// per G input tiles and the column sums of g -- the bias gradient -- fall out of the A operands already loaded.
// PART (cgnn_weight_grad_ordered): no atomics -- row chunk c writes its products to part_w[c][32 OT][32 IT] (dw's tile
// grid, padded) and its column sums to part_b[c][32 OT]; weight_grad_reduce_kernel adds the chunks in a fixed order.
#define CGNN_WGRAD_ROWS 1024
#define CGNN_WGRAD_MAX_CHUNKS 256      // ordered form: the workspace holds one dw per row chunk: see wgrad_ordered_shape
template <int G, bool PART = false>
__global__ __launch_bounds__(CGNN_BLOCK) void weight_grad_kernel(const float* __restrict__ g, int ld_g, int out_dim,
                                                                 const float* __restrict__ a, int ld_a, int in_dim,
                                                                 int64_t n, int it_groups, int n_items,
                                                                 float* __restrict__ dw, int ld_dw, int col0,
                                                                 float* __restrict__ db, int64_t chunk_rows = CGNN_WGRAD_ROWS) {
    const int lane = threadIdx.x & 63, i = lane & 31, kk = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = blockIdx.y * CGNN_WAVES_PER_BLOCK + wave;
    if (item >= n_items) return;
    const int ot = item / it_groups, ig = item % it_groups;
    const int64_t r0 = (int64_t)blockIdx.x * chunk_rows;
    const int64_t r1 = r0 + chunk_rows < n ? r0 + chunk_rows : n;
    const int oc = 32 * ot + i;
    const bool o_ok = oc < out_dim;
    bool i_ok[G];
#pragma unroll
    for (int q = 0; q < G; ++q) i_ok[q] = 32 * (ig * G + q) + i < in_dim;
    const float* gp = g + oc;
    const float* ap = a + 32 * ig * G + i;
    f32x16 acc[G];
#pragma unroll
    for (int q = 0; q < G; ++q)
#pragma unroll
        for (int x = 0; x < 16; ++x) acc[q][x] = 0.f;
    float colsum = 0.f;
    for (int64_t rr = r0; rr < r1; rr += 8) {
        float av[4], bv[G][4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int64_t row = rr + 2 * s + kk;
            const bool ok = row < r1;
            av[s] = (ok && o_ok) ? gp[row * ld_g] : 0.f;
#pragma unroll
            for (int q = 0; q < G; ++q) bv[q][s] = (ok && i_ok[q]) ? ap[row * ld_a + 32 * q] : 0.f;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            colsum += av[s];
#pragma unroll
            for (int q = 0; q < G; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[q][s], acc[q], 0, 0, 0);
        }
    }
    // D[row][col]: col = lane & 31 (input column), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (output row)
#pragma unroll
    for (int q = 0; q < G; ++q)
#pragma unroll
        for (int x = 0; x < 16; ++x) {
            const int o = 32 * ot + (x & 3) + 8 * (x >> 2) + 4 * kk;
            if (PART) {      // dw = part_w, ld_dw = 32 IT (padded in_dim), col0 = 32 OT (padded out_dim): every slot is written
                if (32 * (ig * G + q) < ld_dw)
                    dw[((int64_t)blockIdx.x * col0 + o) * ld_dw + 32 * (ig * G + q) + i] = acc[q][x];
            } else if (o < out_dim && i_ok[q]) {
                atomicAdd(dw + (int64_t)o * ld_dw + col0 + 32 * (ig * G + q) + i, acc[q][x]);
            }
        }
    if (db != nullptr && ig == 0) {
        colsum += __shfl_xor(colsum, 32);
        if (PART) {
            if (kk == 0) db[(int64_t)blockIdx.x * col0 + oc] = colsum;
        } else if (kk == 0 && o_ok) {
            atomicAdd(db + oc, colsum);
        }
    }
}

// dw[o, col0 + i] += sum over chunks of part_w[c][o][i], db[o] += sum of part_b[c][o], in a fixed order: a workgroup takes
// 32 consecutive elements, eight lanes per element add every eighth chunk each (c = j, j + 8, ...), the eight meet left to
// right.  (One thread per element walking all chunks left a 128 x 128 reduction with 64 workgroups: 18 ms per step.)
__global__ __launch_bounds__(CGNN_BLOCK) void weight_grad_reduce_kernel(const float* __restrict__ part_w,
                                                                       const float* __restrict__ part_b, int chunks, int po,
                                                                       int pi, int out_dim, int in_dim, float* __restrict__ dw,
                                                                       int ld_dw, int col0, float* __restrict__ db) {
    __shared__ float red[8][33];
    const int el = threadIdx.x & 31, j = threadIdx.x >> 5;
    const int64_t n_w = (int64_t)po * pi;
    const int64_t e = (int64_t)blockIdx.x * 32 + el;      // elements of dw first, then (padded) those of db
    const bool is_w = e < n_w;
    const int64_t eb = e - n_w;
    const bool is_b = !is_w && db != nullptr && eb < po;
    float sum = 0.f;
    if (is_w) {
        for (int c = j; c < chunks; c += 8) sum += part_w[(int64_t)c * n_w + e];
    } else if (is_b) {
        for (int c = j; c < chunks; c += 8) sum += part_b[(int64_t)c * po + eb];
    }
    red[j][el] = sum;
    __syncthreads();
    if (j == 0) {
        float t = red[0][el];
#pragma unroll
        for (int x = 1; x < 8; ++x) t += red[x][el];
        if (is_w) {
            const int o = (int)(e / pi), i = (int)(e % pi);
            if (o < out_dim && i < in_dim) dw[(int64_t)o * ld_dw + col0 + i] += t;
        } else if (is_b && eb < out_dim) {
            db[eb] += t;
        }
    }
}

// out[c] += sum_r a[r,c] (* b[r,c]): lanes take 4 columns x 1 row each (16-byte loads when aligned), the row lanes
// of a workgroup meet in LDS, one atomic per column per workgroup.
#define CGNN_COLDOT_ROWS 512
// PART: instead of the atomics every workgroup writes its sums to part[blockIdx.x][0 / 1][width] (col_dot_reduce_kernel adds
// them in a fixed order: the result has the same bits on every run).
template <bool BOTH, bool PART = false>      // BOTH: out2[c] += sum_r a[r,c] as well (a is read once for the two sums)
__global__ __launch_bounds__(CGNN_BLOCK) void col_dot_kernel(const float* __restrict__ a, int ld_a,
                                                             const float* __restrict__ b, int ld_b, int64_t n,
                                                             int width, float* __restrict__ out,
                                                             float* __restrict__ out2) {
    __shared__ float red[CGNN_BLOCK][BOTH ? 8 : 4];
    const int c4n = (width + 3) / 4;                 // 4-column groups (<= 64 handled per pass)
    const int64_t r0 = (int64_t)blockIdx.x * CGNN_COLDOT_ROWS;
    const int64_t r1 = r0 + CGNN_COLDOT_ROWS < n ? r0 + CGNN_COLDOT_ROWS : n;
    const bool vec = (width % 4 == 0) && (ld_a % 4 == 0) && ((reinterpret_cast<uintptr_t>(a) & 15) == 0) &&
                     (b == nullptr || ((ld_b % 4 == 0) && ((reinterpret_cast<uintptr_t>(b) & 15) == 0)));
    for (int c0 = 0; c0 < c4n; c0 += 64) {
        const int groups = (c4n - c0) < 64 ? (c4n - c0) : 64;      // column groups this pass
        int cols = 1;
        while (cols < groups) cols <<= 1;                          // lanes per row (power of two <= 64)
        const int rl = CGNN_BLOCK / cols;                          // row lanes
        const int cg = threadIdx.x % cols, rlane = threadIdx.x / cols;
        const int c = 4 * (c0 + cg);
        float s[4] = {0.f, 0.f, 0.f, 0.f}, t[4] = {0.f, 0.f, 0.f, 0.f};
        if (cg < groups) {
            for (int64_t r = r0 + rlane; r < r1; r += rl) {
                float va[4], vb[4] = {1.f, 1.f, 1.f, 1.f};
                if (vec) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(a + r * ld_a + c);
#pragma unroll
                    for (int j = 0; j < 4; ++j) va[j] = t[j];
                    if (b != nullptr) {
                        const f32x4 u = *reinterpret_cast<const f32x4*>(b + r * ld_b + c);
#pragma unroll
                        for (int j = 0; j < 4; ++j) vb[j] = u[j];
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        va[j] = c + j < width ? a[r * ld_a + c + j] : 0.f;
                        if (b != nullptr) vb[j] = c + j < width ? b[r * ld_b + c + j] : 0.f;
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s[j] += va[j] * vb[j];
                    if (BOTH) t[j] += va[j];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            red[threadIdx.x][j] = s[j];
            if (BOTH) red[threadIdx.x][4 + j] = t[j];
        }
        __syncthreads();
        for (int off = rl / 2; off > 0; off >>= 1) {
            if (rlane < off)
#pragma unroll
                for (int j = 0; j < (BOTH ? 8 : 4); ++j) red[threadIdx.x][j] += red[threadIdx.x + off * cols][j];
            __syncthreads();
        }
        if (rlane == 0 && cg < groups)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c + j < width) {
                    if (PART) {      // out = the partial-sum workspace
                        float* const mine = out + (int64_t)blockIdx.x * (BOTH ? 2 : 1) * width;
                        mine[c + j] = red[threadIdx.x][j];
                        if (BOTH) mine[width + c + j] = red[threadIdx.x][4 + j];
                    } else {
                        atomicAdd(out + c + j, red[threadIdx.x][j]);
                        if (BOTH) atomicAdd(out2 + c + j, red[threadIdx.x][4 + j]);
                    }
                }
        __syncthreads();
    }
}

// out[c] += the partial sums of column c over all workgroups, in a fixed order: four row lanes take every fourth partial
// each, left to right, and meet left to right.  One workgroup per 64 columns.
__global__ __launch_bounds__(CGNN_BLOCK) void col_dot_reduce_kernel(const float* __restrict__ part, int64_t nblocks, int nsums,
                                                                   int width, float* __restrict__ out,
                                                                   float* __restrict__ out2) {
    __shared__ float red[CGNN_BLOCK][2];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    float s0 = 0.f, s1 = 0.f;
    if (c < width)
        for (int64_t blk = rl; blk < nblocks; blk += CGNN_BLOCK / 64) {
            s0 += part[blk * nsums * width + c];
            if (nsums == 2) s1 += part[blk * nsums * width + width + c];
        }
    red[threadIdx.x][0] = s0;
    red[threadIdx.x][1] = s1;
    __syncthreads();
    if (rl == 0 && c < width) {
        const int t = threadIdx.x;
        out[c] += ((red[t][0] + red[t + 64][0]) + red[t + 128][0]) + red[t + 192][0];
        if (nsums == 2) out2[c] += ((red[t][1] + red[t + 64][1]) + red[t + 128][1]) + red[t + 192][1];
    }
}

template <int PF, int PB, int K1T, int K2T, int HT, int OT>
static int launch_bwd(bool ln, const MlpDev& f, const MlpDev& b, const void* fw2, const void* bw2, const float* u1, int ld1,
                      const float* u2, int ld2, const float* dy, int ld_dy, int64_t n, const BwdBufs& buf, float* du1,
                      int ld_du1, float* du2, int ld_du2, hipStream_t st) {
    const int grid = grid_for_tiles((n + 31) / 32);
    if (ln)
        mlp_backward_kernel<PF, PB, K1T, K2T, HT, OT, true><<<grid, CGNN_BLOCK, 0, st>>>(f, b, fw2, bw2, u1, ld1, u2, ld2, dy, ld_dy,
                                                                               n, buf, du1, ld_du1, du2, ld_du2);
    else
        mlp_backward_kernel<PF, PB, K1T, K2T, HT, OT, false><<<grid, CGNN_BLOCK, 0, st>>>(f, b, fw2, bw2, u1, ld1, u2, ld2, dy,
                                                                                ld_dy, n, buf, du1, ld_du1, du2, ld_du2);
    return check_hip(hipGetLastError(), "cgnn_mlp_backward launch");
}

}  // namespace cgnn

using namespace cgnn;

extern "C" {

int cgnn_mlp_backward(const cgnn_mlp* fwd, const cgnn_linear* fwd_part2, const cgnn_mlp* bwd,
                      const cgnn_linear* bwd_part2, const float* u1, int32_t ld1, const float* u2, int32_t ld2,
                      const float* dy, int32_t ld_dy, int64_t n, const cgnn_mlp_bwd_buffers* buf, float* du1,
                      int32_t ld_du1, float* du2, int32_t ld_du2, void* stream) {
    MlpDev f, b;
    int rc = make_mlp_dev(fwd, &f, nullptr, "cgnn_mlp_backward(fwd)");
    if (rc != CGNN_OK) return rc;
    rc = make_mlp_dev(bwd, &b, nullptr, "cgnn_mlp_backward(bwd)");
    if (rc != CGNN_OK) return rc;
    const bool mixed = fwd->precision == CGNN_F16X2 && bwd->precision == CGNN_F32X3;
    if (!mixed && ((fwd->precision != CGNN_F32 && fwd->precision != CGNN_F32X3) || bwd->precision != fwd->precision)) {
        set_error("cgnn_mlp_backward: (fwd, bwd) weights must be (CGNN_F32, CGNN_F32), (CGNN_F32X3, CGNN_F32X3) or "
                  "(CGNN_F16X2, CGNN_F32X3)");
        return CGNN_ERR_UNSUPPORTED;
    }
    // (fwd_part2 / bwd_part2 carry no precision of their own: they must be packed like fwd / bwd)
    if (!u1 || !dy || !buf || n < 0 || f.nh != b.nh || !buf->g_o || (fwd_part2 != nullptr) != (bwd_part2 != nullptr) ||
        (fwd_part2 && !u2)) {
        set_error("cgnn_mlp_backward: invalid argument");
        return CGNN_ERR_INVALID_ARG;
    }
    const int hidden = f.out_dim[0], out_dim = f.out_dim[f.nh], in1 = f.in_dim[0];
    for (int l = 0; l <= f.nh; ++l) {
        const int want_in = l == 0 ? in1 : hidden, want_out = l == f.nh ? out_dim : hidden;
        if (f.in_dim[l] != want_in || f.out_dim[l] != want_out || b.in_dim[l] != want_out || b.out_dim[l] != want_in) {
            set_error("cgnn_mlp_backward: layer %d shapes are inconsistent (bwd must hold the transposed weights)", l);
            return CGNN_ERR_INVALID_ARG;
        }
        if (l < f.nh && (!buf->h[l] || !buf->g_a[l])) {
            set_error("cgnn_mlp_backward: scratch buffers for hidden layer %d are missing", l);
            return CGNN_ERR_INVALID_ARG;
        }
    }
    const bool ln = f.gamma != nullptr;
    if (ln && !buf->zhat) {
        set_error("cgnn_mlp_backward: zhat scratch is required with LayerNorm");
        return CGNN_ERR_INVALID_ARG;
    }
    int in2 = 0;
    if (fwd_part2) {
        in2 = fwd_part2->in_dim;
        if (fwd_part2->out_dim != hidden || bwd_part2->in_dim != hidden || bwd_part2->out_dim != in2 || in2 % 32 ||
            ld2 < in2 || (du2 && ld_du2 < in2)) {
            set_error("cgnn_mlp_backward: second input part has inconsistent shapes");
            return CGNN_ERR_INVALID_ARG;
        }
    }
    if (hidden % 32 || (ln && out_dim % 32) || ld1 < in1 || ld_dy < out_dim || (du1 && ld_du1 < in1)) {
        set_error("cgnn_mlp_backward: unsupported shape (hidden %d, out %d)", hidden, out_dim);
        return CGNN_ERR_UNSUPPORTED;
    }
    if (n == 0) return CGNN_OK;
    BwdBufs bb;
    memset(&bb, 0, sizeof(bb));
    for (int l = 0; l < f.nh; ++l) {
        bb.h[l] = buf->h[l];
        bb.g_a[l] = buf->g_a[l];
    }
    bb.g_o = buf->g_o;
    bb.zhat = buf->zhat;
    hipStream_t st = (hipStream_t)stream;
    const int K1T = (in1 + 31) / 32, K2T = in2 / 32, HT = hidden / 32, OT = (out_dim + 31) / 32;
    const void* fw2 = fwd_part2 ? fwd_part2->w : nullptr;
    const void* bw2 = bwd_part2 ? bwd_part2->w : nullptr;
    const bool x3 = fwd->precision == CGNN_F32X3;
#define CGNN_BWD(K1, K2, Hh, Oo)                                                                                        \
    if (K1T == K1 && K2T == K2 && HT == Hh && OT == Oo)                                                                  \
        return mixed ? launch_bwd<CGNN_F16X2, CGNN_F32X3, K1, K2, Hh, Oo>(ln, f, b, fw2, bw2, u1, ld1, u2, ld2, dy, ld_dy, \
                                                                         n, bb, du1, ld_du1, du2, ld_du2, st)            \
               : x3  ? launch_bwd<CGNN_F32X3, CGNN_F32X3, K1, K2, Hh, Oo>(ln, f, b, fw2, bw2, u1, ld1, u2, ld2, dy, ld_dy, \
                                                                         n, bb, du1, ld_du1, du2, ld_du2, st)            \
                     : launch_bwd<CGNN_F32, CGNN_F32, K1, K2, Hh, Oo>(ln, f, b, fw2, bw2, u1, ld1, u2, ld2, dy, ld_dy, n,  \
                                                                     bb, du1, ld_du1, du2, ld_du2, st);
    // square models hidden == latent in {32, 64, 128, 256}: node block (two inputs), encoder (narrow input), decoder
#define CGNN_BWD_T(Tt) CGNN_BWD(Tt, Tt, Tt, Tt) CGNN_BWD(1, 0, Tt, Tt) CGNN_BWD(Tt, 0, Tt, 1)
    CGNN_BWD_T(1) CGNN_BWD_T(2) CGNN_BWD_T(4) CGNN_BWD_T(8)
    // encoders with 33 .. 64 input features (reference config.py:18 --window_size > 8: 3 (W - 1) + W node features, 37 at W = 10)
    CGNN_BWD(2, 0, 2, 2) CGNN_BWD(2, 0, 4, 4) CGNN_BWD(2, 0, 8, 8)
    // mlp_hidden_size != latent_size (reference config.py:19-20 and train.py:165-171 pass them independently): the pairs the
    // forward kernels are compiled for (CGNN_FOR_EACH_PAIR): hidden 128 with latent 64 or 256
#define CGNN_BWD_M(Lt) CGNN_BWD(Lt, Lt, 4, Lt) CGNN_BWD(1, 0, 4, Lt) CGNN_BWD(Lt, 0, 4, 1)
    CGNN_BWD_M(2) CGNN_BWD_M(8)
#undef CGNN_BWD_M
#undef CGNN_BWD_T
#undef CGNN_BWD
    set_error("cgnn_mlp_backward: no kernel for in=(%d,%d) hidden=%d out=%d (training is built for hidden == latent in "
              "{32,64,128,256} and for hidden 128 with latent 64 or 256)", in1, in2, hidden, out_dim);
    return CGNN_ERR_UNSUPPORTED;
}

int cgnn_weight_grad(const float* g, int32_t ld_g, int32_t out_dim, const float* a, int32_t ld_a, int32_t in_dim,
                     int64_t n, float* dw, int32_t ld_dw, int32_t col0, float* db, void* stream) {
    if (!g || !a || !dw || out_dim <= 0 || in_dim <= 0 || n < 0 || ld_g < out_dim || ld_a < in_dim ||
        ld_dw < col0 + in_dim || col0 < 0) {
        set_error("cgnn_weight_grad: invalid argument");
        return CGNN_ERR_INVALID_ARG;
    }
    if (n == 0) return CGNN_OK;
    const int ot = (out_dim + 31) / 32, it = (in_dim + 31) / 32;
    const int G = it >= 4 ? 4 : (it >= 2 ? 2 : 1);
    const int it_groups = (it + G - 1) / G;
    const int n_items = ot * it_groups;
    dim3 grid((unsigned)((n + CGNN_WGRAD_ROWS - 1) / CGNN_WGRAD_ROWS),
              (unsigned)((n_items + CGNN_WAVES_PER_BLOCK - 1) / CGNN_WAVES_PER_BLOCK));
    hipStream_t st = (hipStream_t)stream;
    if (G == 4)
        weight_grad_kernel<4><<<grid, CGNN_BLOCK, 0, st>>>(g, ld_g, out_dim, a, ld_a, in_dim, n, it_groups, n_items, dw,
                                                           ld_dw, col0, db);
    else if (G == 2)
        weight_grad_kernel<2><<<grid, CGNN_BLOCK, 0, st>>>(g, ld_g, out_dim, a, ld_a, in_dim, n, it_groups, n_items, dw,
                                                           ld_dw, col0, db);
    else
        weight_grad_kernel<1><<<grid, CGNN_BLOCK, 0, st>>>(g, ld_g, out_dim, a, ld_a, in_dim, n, it_groups, n_items, dw,
                                                           ld_dw, col0, db);
    return check_hip(hipGetLastError(), "cgnn_weight_grad launch");
}

static void wgrad_ordered_shape(int64_t n, int out_dim, int in_dim, int* chunks, int64_t* chunk_rows, int* po, int* pi) {
    // as many row chunks as the atomic form (1024 rows each) while the partial products stay within 64 MiB, never fewer than
    // CGNN_WGRAD_MAX_CHUNKS allows (256 x 256: 256 chunks of 256 KiB)
    *po = (out_dim + 31) / 32 * 32;
    *pi = (in_dim + 31) / 32 * 32;
    int64_t c = (n + CGNN_WGRAD_ROWS - 1) / CGNN_WGRAD_ROWS;
    int64_t cap = ((int64_t)64 << 20) / ((int64_t)*po * *pi * 4);
    if (cap < CGNN_WGRAD_MAX_CHUNKS) cap = CGNN_WGRAD_MAX_CHUNKS;
    if (c > cap) c = cap;
    if (c < 1) c = 1;
    int64_t rows = ((n + c - 1) / c + 7) / 8 * 8;
    if (rows < 8) rows = 8;
    *chunk_rows = rows;
    *chunks = (int)((n + rows - 1) / rows);
}

size_t cgnn_weight_grad_workspace_bytes(int64_t n, int32_t out_dim, int32_t in_dim) {
    if (n <= 0 || out_dim <= 0 || in_dim <= 0) return 0;
    int chunks, po, pi;
    int64_t rows;
    wgrad_ordered_shape(n, out_dim, in_dim, &chunks, &rows, &po, &pi);
    return (size_t)chunks * ((size_t)po * pi + po) * sizeof(float);
}

// cgnn_weight_grad with the same bits on every run: the row chunks' products go to `workspace` and are added in a fixed
// order by a second kernel (any shape; the 128 x 128 Linears have cgnn_weight_grad_x3).
int cgnn_weight_grad_ordered(const float* g, int32_t ld_g, int32_t out_dim, const float* a, int32_t ld_a, int32_t in_dim,
                             int64_t n, float* dw, int32_t ld_dw, int32_t col0, float* db, void* workspace,
                             size_t workspace_bytes, void* stream) {
    if (!g || !a || !dw || !workspace || out_dim <= 0 || in_dim <= 0 || n < 0 || ld_g < out_dim || ld_a < in_dim ||
        ld_dw < col0 + in_dim || col0 < 0) {
        set_error("cgnn_weight_grad_ordered: invalid argument");
        return CGNN_ERR_INVALID_ARG;
    }
    if (workspace_bytes < cgnn_weight_grad_workspace_bytes(n, out_dim, in_dim)) {
        set_error("cgnn_weight_grad_ordered: workspace has %zu bytes, needs %zu", workspace_bytes,
                  cgnn_weight_grad_workspace_bytes(n, out_dim, in_dim));
        return CGNN_ERR_INVALID_ARG;
    }
    if (n == 0) return CGNN_OK;
    int chunks, po, pi;
    int64_t rows;
    wgrad_ordered_shape(n, out_dim, in_dim, &chunks, &rows, &po, &pi);
    float* part_w = reinterpret_cast<float*>(workspace);
    float* part_b = part_w + (size_t)chunks * po * pi;
    const int ot = po / 32, it = pi / 32;
    const int G = it >= 4 ? 4 : (it >= 2 ? 2 : 1);
    const int it_groups = (it + G - 1) / G;
    const int n_items = ot * it_groups;
    dim3 grid((unsigned)chunks, (unsigned)((n_items + CGNN_WAVES_PER_BLOCK - 1) / CGNN_WAVES_PER_BLOCK));
    hipStream_t st = (hipStream_t)stream;
    // (PART: dw = part_w, ld_dw = padded in_dim, col0 = padded out_dim, db = part_b: always written, read only if db != NULL)
    if (G == 4)
        weight_grad_kernel<4, true><<<grid, CGNN_BLOCK, 0, st>>>(g, ld_g, out_dim, a, ld_a, in_dim, n, it_groups, n_items,
                                                                 part_w, pi, po, part_b, rows);
    else if (G == 2)
        weight_grad_kernel<2, true><<<grid, CGNN_BLOCK, 0, st>>>(g, ld_g, out_dim, a, ld_a, in_dim, n, it_groups, n_items,
                                                                 part_w, pi, po, part_b, rows);
    else
        weight_grad_kernel<1, true><<<grid, CGNN_BLOCK, 0, st>>>(g, ld_g, out_dim, a, ld_a, in_dim, n, it_groups, n_items,
                                                                 part_w, pi, po, part_b, rows);
    const int64_t elems = (int64_t)po * pi + po;      // dw's, then db's
    static_assert(CGNN_BLOCK == 256, "weight_grad_reduce_kernel: 32 elements x 8 lanes");
    weight_grad_reduce_kernel<<<(unsigned)((elems + 31) / 32), CGNN_BLOCK, 0, st>>>(
        part_w, part_b, chunks, po, pi, out_dim, in_dim, dw, ld_dw, col0, db);
    return check_hip(hipGetLastError(), "cgnn_weight_grad_ordered launch");
}

int cgnn_col_dot(const float* a, int32_t ld_a, const float* b, int32_t ld_b, int64_t n, int32_t width, float* out,
                 void* stream) {
    if (!a || !out || width <= 0 || n < 0 || ld_a < width || (b && ld_b < width)) {
        set_error("cgnn_col_dot: invalid argument");
        return CGNN_ERR_INVALID_ARG;
    }
    if (n == 0) return CGNN_OK;
    col_dot_kernel<false><<<(unsigned)((n + CGNN_COLDOT_ROWS - 1) / CGNN_COLDOT_ROWS), CGNN_BLOCK, 0,
                            (hipStream_t)stream>>>(a, ld_a, b, ld_b, n, width, out, nullptr);
    return check_hip(hipGetLastError(), "cgnn_col_dot launch");
}

size_t cgnn_col_dot_workspace_bytes(int64_t n, int32_t width) {
    if (n <= 0 || width <= 0) return 0;
    return (size_t)((n + CGNN_COLDOT_ROWS - 1) / CGNN_COLDOT_ROWS) * 2 * (size_t)width * sizeof(float);
}

// The same sums with the same bits on every run: per-workgroup partial sums in `workspace`, added in a fixed order by a
// second kernel (cgnn_col_dot / cgnn_col_dot2 meet in float atomics).  out_a may be NULL (then b may be NULL as well).
int cgnn_col_dot_ordered(const float* a, int32_t ld_a, const float* b, int32_t ld_b, int64_t n, int32_t width, float* out_ab,
                         float* out_a, void* workspace, size_t workspace_bytes, void* stream) {
    if (!a || !out_ab || width <= 0 || n < 0 || ld_a < width || (b && ld_b < width) || (out_a && !b) || !workspace) {
        set_error("cgnn_col_dot_ordered: invalid argument");
        return CGNN_ERR_INVALID_ARG;
    }
    if (workspace_bytes < cgnn_col_dot_workspace_bytes(n, width)) {
        set_error("cgnn_col_dot_ordered: workspace has %zu bytes, needs %zu", workspace_bytes,
                  cgnn_col_dot_workspace_bytes(n, width));
        return CGNN_ERR_INVALID_ARG;
    }
    if (n == 0) return CGNN_OK;
    static_assert(CGNN_BLOCK == 256, "col_dot_reduce_kernel adds four row lanes");
    const unsigned nblocks = (unsigned)((n + CGNN_COLDOT_ROWS - 1) / CGNN_COLDOT_ROWS);
    float* part = reinterpret_cast<float*>(workspace);
    hipStream_t st = (hipStream_t)stream;
    if (out_a)
        col_dot_kernel<true, true><<<nblocks, CGNN_BLOCK, 0, st>>>(a, ld_a, b, ld_b, n, width, part, nullptr);
    else
        col_dot_kernel<false, true><<<nblocks, CGNN_BLOCK, 0, st>>>(a, ld_a, b, ld_b, n, width, part, nullptr);
    col_dot_reduce_kernel<<<(unsigned)((width + 63) / 64), CGNN_BLOCK, 0, st>>>(part, nblocks, out_a ? 2 : 1, width, out_ab, out_a);
    return check_hip(hipGetLastError(), "cgnn_col_dot_ordered launch");
}

int cgnn_col_dot2(const float* a, int32_t ld_a, const float* b, int32_t ld_b, int64_t n, int32_t width, float* out_ab,
                  float* out_a, void* stream) {
    if (!a || !b || !out_ab || !out_a || width <= 0 || n < 0 || ld_a < width || ld_b < width) {
        set_error("cgnn_col_dot2: invalid argument");
        return CGNN_ERR_INVALID_ARG;
    }
    if (n == 0) return CGNN_OK;
    col_dot_kernel<true><<<(unsigned)((n + CGNN_COLDOT_ROWS - 1) / CGNN_COLDOT_ROWS), CGNN_BLOCK, 0,
                           (hipStream_t)stream>>>(a, ld_a, b, ld_b, n, width, out_ab, out_a);
    return check_hip(hipGetLastError(), "cgnn_col_dot2 launch");
}

}  // extern "C"
