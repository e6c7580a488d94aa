// Measures the sustained v_mfma_f32_32x32x16_bf16 rate of this device (random operands), 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ void k(const float* x, float* out, int iters) {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)x[(threadIdx.x * 8 + j) & 1023]; b[j] = (__bf16)x[(threadIdx.x * 8 + j + 77) & 1023]; }
    f32x16 c0, c1, c2, c3;
    for (int i = 0; i < 16; ++i) { c0[i] = 0; c1[i] = 0; c2[i] = 0; c3[i] = 0; }
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float *x, *o; hipMalloc(&x, 4096); hipMalloc(&o, 256 * 8 * 512 * 4);
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
    hipMemcpy(x, h, 4096, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps = 1; wps <= 2; ++wps) {
        int threads = 256 * wps, blocks = 256, iters = 20000;
        k<<<blocks, threads>>>(x, o, 100);
        hipEventRecord(e0); k<<<blocks, threads>>>(x, o, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double fl = (double)blocks * (threads / 64) * iters * 4.0 * 32 * 32 * 16 * 2;
        printf("waves/SIMD %d: %.3f ms  %.1f TFLOP/s (bf16 dense MFMA)\n", wps, ms, fl / ms / 1e9);
    }
    return 0;
}
