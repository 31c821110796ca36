// Developer probe: how many independent vector instructions hide behind a v_mfma_f32_32x32x16_bf16 of the SAME wave (and of
// a co-resident wave) when the MFMA's accumulator lives in architectural registers (v) versus accumulation registers (a)?
//   hipcc --offload-arch=gfx950 -O2 scripts/dev/probe_mfma_valu.hip -o /tmp/probe_mfma_valu && /tmp/probe_mfma_valu
// Prints cycles per MFMA (s_memtime) for F = 0..10 fillers per MFMA, accumulators in v / a, one and two waves per SIMD.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define STR2(x) #x
#define STR(x) STR2(x)

// one MFMA + F fillers, four times (four accumulators), ITER times
#define FILL_0
#define FILL_1 "v_fma_f32 v100, v101, v102, v100\n\t"
#define FILL_2 FILL_1 "v_fma_f32 v103, v101, v102, v103\n\t"
#define FILL_3 FILL_2 "v_fma_f32 v104, v101, v102, v104\n\t"
#define FILL_4 FILL_3 "v_fma_f32 v105, v101, v102, v105\n\t"
#define FILL_5 FILL_4 "v_fma_f32 v106, v101, v102, v106\n\t"
#define FILL_6 FILL_5 "v_fma_f32 v107, v101, v102, v107\n\t"
#define FILL_8 FILL_6 "v_fma_f32 v108, v101, v102, v108\n\tv_fma_f32 v109, v101, v102, v109\n\t"
#define FILL_10 FILL_8 "v_fma_f32 v110, v101, v102, v110\n\tv_fma_f32 v111, v101, v102, v111\n\t"

#define BODY_V(F)                                                     \
    "v_mfma_f32_32x32x16_bf16 v[0:15], v[64:67], v[68:71], v[0:15]\n\t" F   \
    "v_mfma_f32_32x32x16_bf16 v[16:31], v[64:67], v[68:71], v[16:31]\n\t" F \
    "v_mfma_f32_32x32x16_bf16 v[32:47], v[64:67], v[68:71], v[32:47]\n\t" F \
    "v_mfma_f32_32x32x16_bf16 v[48:63], v[64:67], v[68:71], v[48:63]\n\t" F
#define BODY_A(F)                                                     \
    "v_mfma_f32_32x32x16_bf16 a[0:15], v[64:67], v[68:71], a[0:15]\n\t" F   \
    "v_mfma_f32_32x32x16_bf16 a[16:31], v[64:67], v[68:71], a[16:31]\n\t" F \
    "v_mfma_f32_32x32x16_bf16 a[32:47], v[64:67], v[68:71], a[32:47]\n\t" F \
    "v_mfma_f32_32x32x16_bf16 a[48:63], v[64:67], v[68:71], a[48:63]\n\t" F

#define CLOB_V                                                                                                             \
    "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18",   \
        "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35",  \
        "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52",  \
        "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69",  \
        "v70", "v71", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111"
#define CLOB_A                                                                                                             \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18",   \
        "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35",  \
        "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52",  \
        "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63"

#define ITER 2000

template <int F, bool ACC_A>
__global__ __launch_bounds__(512) void probe(unsigned long long* out, float seed) {
    unsigned long long t0, t1;
    // operands: random-ish bf16 bit patterns, fillers on small floats
    asm volatile(
        "v_mov_b32 v64, 0x3f803f80\n\tv_mov_b32 v65, 0x3f003f80\n\tv_mov_b32 v66, 0xbf803e80\n\tv_mov_b32 v67, 0x3e803f00\n\t"
        "v_mov_b32 v68, 0x3f803d80\n\tv_mov_b32 v69, 0xbe003f80\n\tv_mov_b32 v70, 0x3f803f80\n\tv_mov_b32 v71, 0x3d803f80\n\t"
        "v_mov_b32 v101, 0x3f000000\n\tv_mov_b32 v102, 0x3e800000\n\t" ::
            : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v101", "v102");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < ITER; ++it) {
        if constexpr (ACC_A) {
            if constexpr (F == 0) asm volatile(BODY_A(FILL_0)::: CLOB_V, CLOB_A);
            if constexpr (F == 2) asm volatile(BODY_A(FILL_2)::: CLOB_V, CLOB_A);
            if constexpr (F == 4) asm volatile(BODY_A(FILL_4)::: CLOB_V, CLOB_A);
            if constexpr (F == 5) asm volatile(BODY_A(FILL_5)::: CLOB_V, CLOB_A);
            if constexpr (F == 6) asm volatile(BODY_A(FILL_6)::: CLOB_V, CLOB_A);
            if constexpr (F == 8) asm volatile(BODY_A(FILL_8)::: CLOB_V, CLOB_A);
            if constexpr (F == 10) asm volatile(BODY_A(FILL_10)::: CLOB_V, CLOB_A);
        } else {
            if constexpr (F == 0) asm volatile(BODY_V(FILL_0)::: CLOB_V);
            if constexpr (F == 2) asm volatile(BODY_V(FILL_2)::: CLOB_V);
            if constexpr (F == 4) asm volatile(BODY_V(FILL_4)::: CLOB_V);
            if constexpr (F == 5) asm volatile(BODY_V(FILL_5)::: CLOB_V);
            if constexpr (F == 6) asm volatile(BODY_V(FILL_6)::: CLOB_V);
            if constexpr (F == 8) asm volatile(BODY_V(FILL_8)::: CLOB_V);
            if constexpr (F == 10) asm volatile(BODY_V(FILL_10)::: CLOB_V);
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

// a VALU-only partner wave and an MFMA-only wave on the same SIMD: kernel with 512 threads, waves 0-3 MFMA, 4-7 VALU
template <bool ACC_A>
__global__ __launch_bounds__(512) void probe_split(unsigned long long* out) {
    unsigned long long t0, t1;
    asm volatile(
        "v_mov_b32 v64, 0x3f803f80\n\tv_mov_b32 v65, 0x3f003f80\n\tv_mov_b32 v66, 0xbf803e80\n\tv_mov_b32 v67, 0x3e803f00\n\t"
        "v_mov_b32 v68, 0x3f803d80\n\tv_mov_b32 v69, 0xbe003f80\n\tv_mov_b32 v70, 0x3f803f80\n\tv_mov_b32 v71, 0x3d803f80\n\t"
        "v_mov_b32 v101, 0x3f000000\n\tv_mov_b32 v102, 0x3e800000\n\t" ::
            : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v101", "v102");
    const bool mfma_wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) < 4;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    if (mfma_wave) {
        for (int it = 0; it < ITER; ++it) {
            if constexpr (ACC_A)
                asm volatile(BODY_A(FILL_0)::: CLOB_V, CLOB_A);
            else
                asm volatile(BODY_V(FILL_0)::: CLOB_V);
        }
    } else {
        for (int it = 0; it < ITER; ++it) asm volatile(FILL_10 FILL_10 FILL_10 FILL_10::: CLOB_V);      // 40 VALU per iteration
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

// bare MFMAs, 72 registers: up to four waves per SIMD (1024 threads); host-timed as well (is the matrix pipe really
// faster than one MFMA per 32 cycles when several waves feed it?)
__global__ __launch_bounds__(1024) void probe_lean(unsigned long long* out) {
    unsigned long long t0, t1;
    asm volatile(
        "v_mov_b32 v64, 0x3f803f80\n\tv_mov_b32 v65, 0x3f003f80\n\tv_mov_b32 v66, 0xbf803e80\n\tv_mov_b32 v67, 0x3e803f00\n\t"
        "v_mov_b32 v68, 0x3f803d80\n\tv_mov_b32 v69, 0xbe003f80\n\tv_mov_b32 v70, 0x3f803f80\n\tv_mov_b32 v71, 0x3d803f80\n\t" ::
            : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < ITER; ++it) asm volatile(BODY_V(FILL_0)::: "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}
static void run_lean(int threads) {
    unsigned long long* d;
    const int blocks = 256, waves = threads / 64;
    hipMalloc(&d, sizeof(unsigned long long) * blocks * waves);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) probe_lean<<<blocks, threads>>>(d);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < 10; ++rep) probe_lean<<<blocks, threads>>>(d);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * waves);
    hipMemcpy(h.data(), d, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    s /= h.size();
    const double mfmas = 4.0 * ITER * waves * blocks * 10;      // per timed region
    printf("bare MFMAs, %d waves per SIMD: %6.1f cycles per MFMA per wave (s_memtime); host: %.3f ms for 10 launches = %.1f TFLOP/s "
           "= one MFMA per %.1f ns per SIMD\n", waves / 4, s / (4.0 * ITER), ms, mfmas * 32768.0 / (ms * 1e-3) / 1e12,
           ms * 1e6 / (mfmas / 1024.0));
    hipFree(d);
}

// An MFMA-only wave beside a wave that only issues vector-memory loads (waves 0-3 / 4-7 of a 512-thread workgroup):
// MODE 0: 16 bytes per lane from 64 different rows (the P-row gather of the edge-stream kernels), 1: 1 KiB contiguous per
// instruction, 2: LDS-DMA (buffer_load ... lds) of 1 KiB contiguous.  Eight loads in flight, then a counted wait.
template <int MODE>
__global__ __launch_bounds__(512) void probe_vmem(unsigned long long* out, const char* table, unsigned rows) {
    __shared__ __attribute__((aligned(16))) char lds[8 * 1024];
    unsigned long long t0, t1;
    asm volatile(
        "v_mov_b32 v64, 0x3f803f80\n\tv_mov_b32 v65, 0x3f003f80\n\tv_mov_b32 v66, 0xbf803e80\n\tv_mov_b32 v67, 0x3e803f00\n\t"
        "v_mov_b32 v68, 0x3f803d80\n\tv_mov_b32 v69, 0xbe003f80\n\tv_mov_b32 v70, 0x3f803f80\n\tv_mov_b32 v71, 0x3d803f80\n\t" ::
            : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71");
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const bool mfma_wave = wave < 4;
    unsigned keep = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    if (mfma_wave) {
        for (int it = 0; it < ITER; ++it) asm volatile(BODY_V(FILL_0)::: CLOB_V);
    } else {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(table), 0, 0x7fffffff, 0x00020000);
        unsigned x = (blockIdx.x * 8 + wave) * 2654435761u + lane * 40503u;
        for (int it = 0; it < ITER / 4; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                x = x * 1664525u + 1013904223u;
                unsigned off;
                if (MODE == 0)
                    off = ((x >> 8) % rows) * 256u + (lane >> 5) * 128u + j * 16u;         // own row, piece j (rows re-drawn per j: no reuse)
                else
                    off = ((__builtin_amdgcn_readfirstlane(x) >> 8) % (rows / 4)) * 1024u + lane * 16u;
                if (MODE == 2) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + (wave - 4) * 1024 * 0 + j * 1024), 16, off, 0, 0, 0);
                } else {
                    const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0);
                    keep += v[0];
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0 + (keep == 0x12345678u ? 1 : 0);
}
template <int MODE>
static void run_vmem(const char* what) {
    unsigned long long* d;
    char* table;
    const unsigned rows = 1u << 20;      // 256 MB of 256-byte rows
    hipMalloc(&d, sizeof(unsigned long long) * 256 * 8);
    hipMalloc(&table, (size_t)rows * 256);
    hipMemset(table, 1, (size_t)rows * 256);
    for (int rep = 0; rep < 3; ++rep) probe_vmem<MODE><<<256, 512>>>(d, table, rows);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), d, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
    double sm = 0, sv = 0;
    for (int b = 0; b < 256; ++b)
        for (int w = 0; w < 8; ++w) (w < 4 ? sm : sv) += (double)h[b * 8 + w];
    sm /= 256 * 4;
    sv /= 256 * 4;
    printf("MFMA wave beside a loading wave (%s): %6.1f cycles per MFMA; loading wave %7.1f cycles per load instruction\n", what,
           sm / (4.0 * ITER), sv / (2.0 * ITER));
    hipFree(d);
    hipFree(table);
}

template <int F, bool ACC_A>
static void run(int threads, const char* what) {
    unsigned long long* d;
    const int blocks = 256, waves = threads / 64;
    hipMalloc(&d, sizeof(unsigned long long) * blocks * waves);
    for (int rep = 0; rep < 3; ++rep) probe<F, ACC_A><<<blocks, threads>>>(d, 1.0f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * waves);
    hipMemcpy(h.data(), d, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    s /= h.size();
    printf("%-22s acc in %s, %2d fillers per MFMA: %7.1f cycles per MFMA (per wave), %6.1f per MFMA per SIMD\n", what,
           ACC_A ? "a" : "v", F, s / (4.0 * ITER), s / (4.0 * ITER) / (threads / 256));
    hipFree(d);
}

template <bool ACC_A>
static void run_split() {
    unsigned long long* d;
    const int blocks = 256, waves = 8;
    hipMalloc(&d, sizeof(unsigned long long) * blocks * waves);
    for (int rep = 0; rep < 3; ++rep) probe_split<ACC_A><<<blocks, 512>>>(d);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * waves);
    hipMemcpy(h.data(), d, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
    double sm = 0, sv = 0;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < 8; ++w) (w < 4 ? sm : sv) += (double)h[b * 8 + w];
    sm /= blocks * 4;
    sv /= blocks * 4;
    printf("split roles, acc in %s: MFMA wave %7.1f cycles per MFMA; VALU wave %6.2f cycles per vector instruction\n", ACC_A ? "a" : "v",
           sm / (4.0 * ITER), sv / (40.0 * ITER));
    hipFree(d);
}

int main() {
    run_vmem<0>("16 B per lane from 64 rows");
    run_vmem<1>("1 KiB contiguous");
    run_vmem<2>("LDS-DMA, 1 KiB contiguous");
    for (int t = 256; t <= 1024; t += 256) run_lean(t);
#define RUNF(F)                                       \
    run<F, false>(256, "one wave per SIMD");         \
    run<F, true>(256, "one wave per SIMD");          \
    run<F, false>(512, "two waves per SIMD");        \
    run<F, true>(512, "two waves per SIMD");
    RUNF(0) RUNF(2) RUNF(4) RUNF(5) RUNF(6) RUNF(8) RUNF(10)
    run_split<false>();
    run_split<true>();
    return 0;
}
