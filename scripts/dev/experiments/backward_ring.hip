// EXPERIMENT RECORD (round 4) -- not part of the library.  Measured on one MI355X, training step at 1 M particles, fp32x3
// (gpurun_out/r04_k_time_train.log, r04_l_prof): the 60 node-model calls of three steps took 1.976 ms each through this kernel
// against 1.96 ms through mlp_backward_kernel -- no gain, all 104 tests of tests/test_gpu_training.py green with it.  The ring
// removes three quarters of the L2 weight traffic (20 -> 5 GB per call), but its 96 KiB of LDS leave room for ONE workgroup
// per CU: one wave per SIMD, where the stock kernel runs two whose loads, splits and stores overlap the other's MFMAs.
// L2 bandwidth was not what bounded the stock kernel.  To build it again: add this file to csrc/Makefile's SRCS, declare
// mlp_backward_ring in backward.hip and call it ahead of the CGNN_BWD dispatch when bwd->precision == CGNN_F32X3.
//
// cgnn_mlp_backward for the node models of a 128-wide network (two 128-feature inputs, hidden = out = 128, LayerNorm):
// the data-gradient kernel of backward.hip with its weights streamed through LDS (reference train.py:263 through
// graph_network.py:94-96).
//
// Why.  mlp_backward_kernel lets every wave fetch every weight fragment of its 32-row tile from L2: the recomputed
// forward (two fp16 terms: 64 KiB per layer) and the gradient chain (three bf16 terms: 96 KiB per layer) are 640 KiB per
// tile, 20 GB per call at 1 M rows -- 10 TB/s of L2 traffic for the 2.0 ms the call takes, which is what bounds it (23 of
// a 57 ms training step).  Here the four waves of a workgroup share each fragment: the layers are cut
// into chunks of 16 fragments (32 / 48 KiB) that cycle through the two-slot LDS ring of weight_ring.hpp (LDS-DMA of chunk
// i + 1 under the MFMAs of chunk i, one barrier per chunk; a chunk is 96 MFMAs of 32 cycles per wave, longer than an L2
// round trip).  Same arithmetic, same fragment order per output tile: results are bit-identical to mlp_backward_kernel.
#include <string.h>

#include "mlp_device.hpp"
#include "weight_ring.hpp"

namespace cgnn {

struct BwdBufs {      // (as in backward.hip)
    float* h[CGNN_MAX_HIDDEN_LAYERS];
    float* g_a[CGNN_MAX_HIDDEN_LAYERS];
    float* g_o;
    float* zhat;
};

namespace bwr {
constexpr int T = 4, W = 32 * T;                 // 128-wide layers
constexpr int M = T * T * 2, CH = 16;            // MFMA-step fragments per layer, per chunk (two 32-row output tiles)
constexpr int SLOT = CH * 3072;                  // a three-bf16-term chunk (48 KiB); a two-fp16-term chunk is 32 KiB
typedef WeightRingT<SLOT, 2> Ring;

// dense_part (cgnn_common.hpp) for the two-accumulator fp16 form: the chunk holds whole output tiles (8 fragments each),
// so the second accumulator is zeroed and folded inside the chunk exactly as dense() does it
template <int M0, int M1>
__device__ __forceinline__ void dense_part_f2(f32x16 (&out)[T], const Operand<CGNN_F16X2, T>& in, const LdsWf2g& wp, int lane) {
    constexpr int S = 2, MM = M1 - M0, GS = 4, NG = MM / GS, PER = T * S;
    static_assert(MM % GS == 0 && M0 % PER == 0 && M1 % PER == 0, "chunks of whole output tiles");
    f16x8x2 buf[2][GS];
    f32x16 c1[2];
#pragma unroll
    for (int j = 0; j < GS; ++j) buf[0][j] = wp.fetch(j, lane);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) {
#pragma unroll
            for (int j = 0; j < GS; ++j) buf[(g + 1) & 1][j] = wp.fetch((g + 1) * GS + j, lane);
        }
#pragma unroll
        for (int j = 0; j < GS; ++j) {
            const int m = M0 + g * GS + j;
            const int o = m / PER, kt = (m / S) % T, s = m % S;
            if (m % PER == 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) c1[o & 1][i] = 0.f;
            }
            mfma_step2<T>(buf[g & 1][j], in, kt, s, out[o], c1[o & 1]);
            if (m % PER == PER - 1) out[o] += c1[o & 1] * CGNN_F16X2_INV_SCALE;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int PREC>
struct Layer;      // one 128 x 128 layer = the ring's next two chunks
template <>
struct Layer<CGNN_F16X2> {
    static __device__ __forceinline__ void run(f32x16 (&acc)[T], const Operand<CGNN_F16X2, T>& op, Ring& ring, bool more, int lane) {
        const LdsWf2g w0(ring.acquire(more));
        dense_part_f2<0, CH>(acc, op, w0, lane);
        const LdsWf2g w1(ring.acquire(more));
        dense_part_f2<CH, 2 * CH>(acc, op, w1, lane);
    }
};
template <>
struct Layer<CGNN_F32X3> {
    static __device__ __forceinline__ void run(f32x16 (&acc)[T], const Operand<CGNN_F32X3, T>& op, Ring& ring, bool more, int lane) {
        const LdsWx3 w0(ring.acquire(more));
        dense_part<T, T, 0, CH>(acc, op, w0, lane);
        const LdsWx3 w1(ring.acquire(more));
        dense_part<T, T, CH, 2 * CH>(acc, op, w1, lane);
    }
};
}  // namespace bwr

// PF: arithmetic of the recomputed forward (CGNN_F16X2 or CGNN_F32X3); the gradient chain runs on three bf16 terms.
// chunks: forward W0a, W0b, W1 .. W_nh, then the transposed W_nh .. W1, W0a, W0b, two chunks each.  Rows are 128 floats
// (ld = 128, 16-byte aligned: the launcher checks).
template <int PF, bool LN>
__global__ __launch_bounds__(CGNN_BLOCK) void mlp_backward_ring_kernel(MlpDev f, X3Chunks chunks, const float* __restrict__ u1,
                                                                       const float* __restrict__ u2,
                                                                       const float* __restrict__ dy, int64_t n, BwdBufs buf,
                                                                       float* __restrict__ du1, float* __restrict__ du2) {
    using namespace bwr;
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t tiles = (n + 31) / 32;
    // block-uniform tile loop (every wave must reach every barrier): wave w of the block takes tile bt + w
    const int nb = gridDim.x, b = blockIdx.x;
    int64_t bt, bend, bstride;
    if ((nb & 7) == 0) {
        const int xcd = b & 7, slot = b >> 3, per = nb >> 3;
        bt = tiles * xcd / 8 + (int64_t)slot * CGNN_WAVES_PER_BLOCK;
        bend = tiles * (xcd + 1) / 8;
        bstride = (int64_t)per * CGNN_WAVES_PER_BLOCK;
    } else {
        bt = (int64_t)b * CGNN_WAVES_PER_BLOCK;
        bend = tiles;
        bstride = (int64_t)nb * CGNN_WAVES_PER_BLOCK;
    }
    Ring ring(chunks, wave, lane);
    if (bt < bend) ring.issue(0);
    for (; bt < bend; bt += bstride) {
        const bool more = bt + bstride < bend;
        const int64_t tile = bt + wave;
        const int64_t row = tile * 32 + r;
        const bool live = tile < bend && row < n;
        const int64_t rowc = live ? row : n - 1;
        // ------------------------------------------------------------ forward, recomputed
        Operand<PF, T> oph;
        auto relu_store = [&](f32x16 (&acc)[T], float* dst) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][i] = acc[t][i] < 0.f ? 0.f : acc[t][i];     // NaN stays NaN (an fp16 overflow must show)
            if (live) store_rows_full<T>(acc, dst + row * W, h);
            oph.template from_acc<false>(acc);
        };
        {
            f32x16 acc[T];
            acc_fill_bias<T>(acc, f.b[0], W, h);
            {
                f32x16 t1[T];
                load_rows_full<T>(t1, u1 + rowc * W, h);
                Operand<PF, T> op;
                op.template from_acc<false>(t1);
                Layer<PF>::run(acc, op, ring, more, lane);
            }
            {
                f32x16 t2[T];
                load_rows_full<T>(t2, u2 + rowc * W, h);
                Operand<PF, T> op;
                op.template from_acc<false>(t2);
                Layer<PF>::run(acc, op, ring, more, lane);
            }
            relu_store(acc, buf.h[0]);
        }
        for (int l = 1; l < f.nh; ++l) {
            f32x16 acc[T];
            acc_fill_bias<T>(acc, f.b[l], W, h);
            Layer<PF>::run(acc, oph, ring, more, lane);
            relu_store(acc, buf.h[l]);
        }
        f32x16 g[T];      // becomes dL/d(pre-LayerNorm output)
        {
            f32x16 out[T];
            acc_fill_bias<T>(out, f.b[f.nh], W, h);
            Layer<PF>::run(out, oph, ring, more, lane);
            // -------------------------------------------------------- output gradient through LayerNorm
            load_rows_full<T>(g, dy + rowc * W, h);
            if (LN) {
                float s = 0.f;
#pragma unroll
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) s += out[t][i];
                s += __shfl_xor(s, 32);
                const float mean = s * (1.0f / W);
                float q = 0.f;
#pragma unroll
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float d = out[t][i] - mean;
                        q += d * d;
                    }
                q += __shfl_xor(q, 32);
                const float rstd = 1.0f / sqrtf(q * (1.0f / W) + 1e-5f);
                float m1 = 0.f, m2 = 0.f;
#pragma unroll
                for (int t = 0; t < T; ++t)
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
                m1 *= (1.0f / W);
                m2 *= (1.0f / W);
                if (live) store_rows_full<T>(out, buf.zhat + row * W, h);
#pragma unroll
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) g[t][i] = rstd * (g[t][i] - m1 - out[t][i] * m2);
            }
        }
        if (live) store_rows_full<T>(g, buf.g_o + row * W, h);
        // ------------------------------------------------------------ backward through the hidden layers
        Operand<CGNN_F32X3, T> og;     // dL/d(pre-activation) of the layer being left
        auto relu_backward = [&](f32x16 (&gh)[T], int l) __attribute__((always_inline)) {
            f32x16 hv[T];
            load_rows_full<T>(hv, buf.h[l] + rowc * W, h);
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) gh[t][i] = hv[t][i] > 0.f ? gh[t][i] : 0.f;
            if (live) store_rows_full<T>(gh, buf.g_a[l] + row * W, h);
            og.template from_acc<false>(gh);
        };
        auto zero = [&](f32x16 (&a)[T]) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) a[t][i] = 0.f;
        };
        {
            Operand<CGNN_F32X3, T> go;
            go.template from_acc<false>(g);
            f32x16 gh[T];
            zero(gh);
            Layer<CGNN_F32X3>::run(gh, go, ring, more, lane);     // W_nh^T
            relu_backward(gh, f.nh - 1);
        }
        for (int l = f.nh - 1; l >= 1; --l) {
            f32x16 gh[T];
            zero(gh);
            Layer<CGNN_F32X3>::run(gh, og, ring, more, lane);     // W_l^T
            relu_backward(gh, l - 1);
        }
        // ------------------------------------------------------------ input gradients (both are always produced here)
        {
            f32x16 gx[T];
            zero(gx);
            Layer<CGNN_F32X3>::run(gx, og, ring, more, lane);     // W_0a^T
            if (live) store_rows_full<T>(gx, du1 + row * W, h);
        }
        {
            f32x16 gx[T];
            zero(gx);
            Layer<CGNN_F32X3>::run(gx, og, ring, more, lane);     // W_0b^T
            if (live) store_rows_full<T>(gx, du2 + row * W, h);
        }
    }
}

// Returns CGNN_OK and *taken = 1 when the ring kernel covers the call (and has been launched), *taken = 0 otherwise.
// fwd / bwd: validated device views; fw2 / bw2: the second input part's packed weights.
int mlp_backward_ring(int pf, bool ln, const MlpDev& f, const MlpDev& b, const void* fw2, const void* bw2, int in1, int in2,
                      int hidden, int out_dim, const float* u1, int ld1, const float* u2, int ld2, const float* dy, int ld_dy,
                      int64_t n, const BwdBufs& buf, float* du1, int ld_du1, float* du2, int ld_du2, hipStream_t st, int* taken) {
    using namespace bwr;
    *taken = 0;
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!(pf == CGNN_F16X2 || pf == CGNN_F32X3) || in1 != W || in2 != W || hidden != W || out_dim != W || !fw2 || !bw2 || !u2 ||
        !du1 || !du2 || ld1 != W || ld2 != W || ld_dy != W || ld_du1 != W || ld_du2 != W || !al(u1) || !al(u2) || !al(dy) ||
        !al(du1) || !al(du2) || f.nh < 1 || f.nh > 3 || n < 4096)
        return CGNN_OK;
    for (int l = 0; l <= f.nh; ++l)
        if (!f.b[l]) return CGNN_OK;
    X3Chunks ch;
    memset(&ch, 0, sizeof(ch));
    const size_t fb = pf == CGNN_F16X2 ? 2048 : 3072;
    auto add = [&](const void* w, size_t frag_bytes) {
        for (int c = 0; c < M / CH; ++c) {
            ch.src[ch.count] = reinterpret_cast<const char*>(w) + (size_t)c * CH * frag_bytes;
            ch.bytes[ch.count++] = (uint32_t)(CH * frag_bytes);
        }
    };
    add(f.w[0], fb);
    add(fw2, fb);
    for (int l = 1; l <= f.nh; ++l) add(f.w[l], fb);
    for (int l = f.nh; l >= 1; --l) add(b.w[l], 3072);
    add(b.w[0], 3072);
    add(bw2, 3072);
    const int lds = 2 * SLOT;
#define CGNN_GO(PFx, LNx)                                                                                              \
    {                                                                                                                   \
        auto kern = mlp_backward_ring_kernel<PFx, LNx>;                                                                 \
        int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (size_t)lds, "hipFuncSetAttribute(mlp_backward_ring)"); \
        if (rc != CGNN_OK) return rc;                                                                                   \
        const int grid = grid_for_tiles((n + 31) / 32, 1);                                                              \
        kern<<<grid, CGNN_BLOCK, lds, st>>>(f, ch, u1, u2, dy, n, buf, du1, du2);                                       \
    }
    if (pf == CGNN_F16X2) {
        if (ln) CGNN_GO(CGNN_F16X2, true) else CGNN_GO(CGNN_F16X2, false)
    } else {
        if (ln) CGNN_GO(CGNN_F32X3, true) else CGNN_GO(CGNN_F32X3, false)
    }
#undef CGNN_GO
    *taken = 1;
    return check_hip(hipGetLastError(), "cgnn_mlp_backward(ring) launch");
}

}  // namespace cgnn
