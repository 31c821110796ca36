// Developer probe (round 4): v_mfma_f32_32x32x16_bf16 against v_mfma_f32_16x16x32_bf16 in a loop shaped like the edge stream's
// MFMA blocks -- two waves per SIMD, every A fragment (1 KiB) read from LDS, F independent vector instructions per 32 cycles
// of matrix work, random operands (the clock the chip holds depends on the data and on the MFMA shape:
// MI355X_MICROARCH.md, "DVFS give-back" (7)).  Same flops, same LDS bytes per flop in both shapes: a 16x16x32 fragment feeds
// two MFMAs (two 16-column tiles), a 32x32x16 fragment one.
//   hipcc --offload-arch=gfx950 -O3 scripts/dev/probe_mfma_shape.hip -o scripts/dev/build/probe_mfma_shape
//   scripts/dev/build/probe_mfma_shape          (on the GPU box)
// Prints, per shape and F: ms per launch, TFLOP/s, the in-kernel clock (s_memtime / s_memrealtime) and cycles per 32 matrix cycles.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

extern __shared__ char smem[];

#define ITER 3000      // x 32 fragments per wave and launch

template <int F>
__device__ __forceinline__ void fillers(float (&v)[8], float a, float b) {
#pragma unroll
    for (int i = 0; i < F; ++i) v[i & 7] = __builtin_fmaf(v[i & 7], a, b);
}

// SHAPE 32: 32 fragments x one 32x32x16 MFMA (4 accumulators of 16 registers, 8 k-steps each)
// SHAPE 16: 32 fragments x two 16x16x32 MFMAs (2 column tiles x 8 row tiles of 4 registers, 4 k-steps each)
template <int SHAPE, int F>
__global__ __launch_bounds__(512) void probe(float* out, unsigned long long* clk, unsigned seed) {
    const int lane = threadIdx.x & 63;
    // random bf16 weights in LDS (values in (-1, 1): exponent field of 0.5 .. 1, random mantissa and sign)
    {
        unsigned* w = reinterpret_cast<unsigned*>(smem);
        unsigned x = seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x * 9973u + 1u;
        for (int i = threadIdx.x; i < 32 * 1024 / 4; i += blockDim.x) {
            x = x * 1664525u + 1013904223u;
            const unsigned lo = 0x3f00u | ((x >> 8) & 0x807fu), hi = 0x3f00u | ((x >> 20) & 0x807fu);
            w[i] = lo | (hi << 16);
        }
    }
    __syncthreads();
    bf16x8 bin[8];
    {
        unsigned x = seed + lane * 7919u + (threadIdx.x >> 6) * 104729u;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            u32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                x = x * 1664525u + 1013904223u;
                v[j] = (0x3f00u | ((x >> 8) & 0x807fu)) | ((0x3f00u | ((x >> 20) & 0x807fu)) << 16);
            }
            bin[k] = __builtin_bit_cast(bf16x8, v);
        }
    }
    float fv[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
    const float fa = 0.999f + 1e-6f * lane, fb = 1e-3f;
    f32x16 a32[4];
    f32x4 a16[2][8];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) a32[t][i] = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < 8; ++t) a16[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const __attribute__((address_space(3))) bf16x8* frag =
        reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem) + lane;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            const bf16x8 a = frag[m * 64];
            if constexpr (SHAPE == 32) {
                a32[m >> 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bin[m & 7], a32[m >> 3], 0, 0, 0);
                fillers<F>(fv, fa, fb);
            } else {
                a16[0][m >> 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bin[m & 3], a16[0][m >> 2], 0, 0, 0);
                fillers<F / 2>(fv, fa, fb);
                a16[1][m >> 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bin[4 + (m & 3)], a16[1][m >> 2], 0, 0, 0);
                fillers<F - F / 2>(fv, fa, fb);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // keep the accumulators bounded (a bf16 pack + ReLU like the kernel's would; here: scale)
        if ((it & 63) == 63) {
            if constexpr (SHAPE == 32) {
#pragma unroll
                for (int t = 0; t < 4; ++t) a32[t] *= 1e-3f;
            } else {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int t = 0; t < 8; ++t) a16[c][t] *= 1e-3f;
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    if constexpr (SHAPE == 32) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) s += a32[t][i];
    } else {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < 8; ++t) s += a16[c][t][0] + a16[c][t][1] + a16[c][t][2] + a16[c][t][3];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s += fv[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = t1 - t0;
        clk[2 * blockIdx.x + 1] = r1 - r0;
    }
}

template <int SHAPE, int F>
static void run(float* out, unsigned long long* clk, int blocks, const char* name) {
    const size_t lds = 100 * 1024;      // one workgroup per CU: two waves per SIMD
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<SHAPE, F>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) probe<SHAPE, F><<<blocks, 512, lds>>>(out, clk, 17u + w);
    hipDeviceSynchronize();
    const int reps = 12;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) probe<SHAPE, F><<<blocks, 512, lds>>>(out, clk, 99u + r);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), clk, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double cyc = 0, real = 0;
    for (int b = 0; b < blocks; ++b) {
        cyc += (double)h[2 * b];
        real += (double)h[2 * b + 1];
    }
    const double ghz = cyc / real * 0.1;      // s_memrealtime ticks at 100 MHz
    const double flops = (double)blocks * 8 * ITER * 32 * 32768.0;
    const double per32 = cyc / blocks / (ITER * 32.0) / 2.0;      // wave cycles per fragment, two waves share a SIMD's pipe
    printf("%-10s F=%d  %8.3f ms  %7.1f TFLOP/s  clock %.3f GHz  %5.1f cycles per 32 matrix cycles per SIMD\n", name, F, ms,
           flops / ms / 1e9, ghz, per32);
    fflush(stdout);
}

int main() {
    int cus = 256;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) == hipSuccess) cus = p.multiProcessorCount;
    const int blocks = cus;
    float* out;
    unsigned long long* clk;
    hipMalloc(&out, (size_t)blocks * 512 * sizeof(float));
    hipMalloc(&clk, (size_t)blocks * 2 * sizeof(unsigned long long));
    for (int round = 0; round < 2; ++round) {
        run<32, 0>(out, clk, blocks, "32x32x16");
        run<16, 0>(out, clk, blocks, "16x16x32");
        run<32, 4>(out, clk, blocks, "32x32x16");
        run<16, 4>(out, clk, blocks, "16x16x32");
        run<32, 6>(out, clk, blocks, "32x32x16");
        run<16, 6>(out, clk, blocks, "16x16x32");
        run<32, 8>(out, clk, blocks, "32x32x16");
        run<16, 8>(out, clk, blocks, "16x16x32");
    }
    return 0;
}
