// Developer probe: does v_mfma_f32_16x16x32_f16 honour fp16 subnormal inputs, and what do inf inputs give?
//   hipcc --offload-arch=gfx950 -O2 scripts/dev/probe_f16_mfma.hip -o /tmp/probe_f16 && /tmp/probe_f16
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float a_val, float b_val, float* out) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
    const int lane = threadIdx.x;
    // A row (lane & 15), k block (lane >> 4) * 8 + i ; B column (lane & 15), same k
    if ((lane >> 4) == 0) { a[0] = (_Float16)a_val; b[0] = (_Float16)b_val; }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    if (lane == 0) { out[0] = acc[0]; out[1] = (float)a[0]; out[2] = (float)b[0]; }
}
int main() {
    float* d; hipMalloc(&d, 16);
    const float cases[][2] = {{1.f, 1.f}, {9.5367431640625e-7f /*2^-20*/, 1024.f}, {5.9604644775390625e-8f /*2^-24*/, 16384.f},
                              {3.0e-5f, 3.0e-5f}, {70000.f, 1.f}, {60000.f, 60000.f}, {6.1e-5f, 1.0f}, {1.0e-6f, 1.0f}};
    for (auto& c : cases) {
        probe<<<1, 64>>>(c[0], c[1], d);
        float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
        printf("a=%g (f16 %g)  b=%g (f16 %g)  mfma=%.9g  exact=%.9g\n", c[0], h[1], c[1], h[2], h[0], (double)h[1] * h[2]);
    }
    return 0;
}
