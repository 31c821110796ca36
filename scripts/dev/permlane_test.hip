// Developer check: gfx950 v_permlane16_swap / v_permlane32_swap as cross-row sums (vs ds_bpermute based __shfl_xor).
#include <hip/hip_runtime.h>
#include <stdio.h>
// Inline asm: with the builtin, hipcc (ROCm 7.2) merged the two results when both operands held the same value.
__device__ float xsum16(float s) {
    float a = s, b = s;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ float xsum32(float s) {
    float a = s, b = s;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
__global__ void k(const float* in, float* a, float* b) {
    const float s = in[threadIdx.x];
    a[threadIdx.x] = xsum32(xsum16(s));
    float t = s + __shfl_xor(s, 16);
    t += __shfl_xor(t, 32);
    b[threadIdx.x] = t;
}
int main() {
    float h[64], ha[64], hb[64], *d, *a, *b;
    for (int i = 0; i < 64; ++i) h[i] = (float)(i * i % 37) + 0.25f * i;
    hipMalloc(&d, 256); hipMalloc(&a, 256); hipMalloc(&b, 256);
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, a, b);
    hipMemcpy(ha, a, 256, hipMemcpyDeviceToHost); hipMemcpy(hb, b, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) if (ha[i] != hb[i]) { ++bad; if (bad < 5) printf("lane %d: %f vs %f\n", i, ha[i], hb[i]); }
    printf("permlane swap sums: %s (%d mismatches)\n", bad ? "DIFFER" : "match __shfl_xor", bad);
    return bad != 0;
}
