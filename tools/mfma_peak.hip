// Pure-MFMA ceiling probe for v_mfma_f32_32x32x2_f32 on the box at hand (no memory traffic):
// prints achieved TFLOP/s so kernel efficiencies can be quoted against what the silicon sustains.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[3], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main(int argc, char** argv) {
    int blocks_per_cu = argc > 1 ? atoi(argv[1]) : 1;
    int blocks = 256 * blocks_per_cu, iters = 20000;
    float* out; hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_loop<<<blocks, 256>>>(out, 100, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mfma_loop<<<blocks, 256>>>(out, iters, 0.5f, 0.25f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 /*waves*/ * iters * 32.0 * (2.0 * 32 * 32 * 2);
    printf("blocks/CU %d: %.1f TFLOP/s (%.2f ms) -> implied clock %.2f GHz\n", blocks_per_cu, flops / ms / 1e9, ms,
           flops / ms / 1e9 / 157.3 * 2.4);
    return 0;
}
