// Pure-MFMA ceiling probe for v_mfma_f32_32x32x16_bf16 on the box at hand (no memory traffic): the 24-MFMA pattern of one
// K-step of the split-bf16 kernels (2x2 accumulators x 6 plane pairs) on register-resident operands, random or zero data.
// Says what the matrix pipe sustains under the power limit, so kernel efficiencies can be quoted against the silicon.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak_bf16.hip -o tools/mfma_peak_bf16 && tools/mfma_peak_bf16 [waves/SIMD] [ms]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, unsigned mask) {
    f32x16 acc[2][2];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i >> 1][i & 1][r] = 0.f;
    bf16x8 fa[2][3], fb[2][3];
    for (int i = 0; i < 6; ++i) {
        u32x4 ua, ub;
        for (int e = 0; e < 4; ++e) {   // random mantissas, exponents near 1.0 (bf16 = high half of an f32)
            const unsigned ra = hash(threadIdx.x * 97u + i * 13u + e), rb = hash(threadIdx.x * 89u + i * 17u + e + 1000u);
            ua[e] = ((ra & 0x807f807fu) | 0x3f003f00u) & mask;
            ub[e] = ((rb & 0x807f807fu) | 0x3f003f00u) & mask;
        }
        fa[i / 3][i % 3] = __builtin_bit_cast(bf16x8, ua);
        fb[i / 3][i % 3] = __builtin_bit_cast(bf16x8, ub);
    }
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][PA[q]], fb[b][PB[q]], acc[a][b], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i >> 1][i & 1][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 2;          // waves per SIMD (blocks of 4 waves per CU)
    const double target_ms = argc > 2 ? atof(argv[2]) : 60.0;
    const int blocks = 256 * wps;
    float* out; hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (unsigned mask : {0xffffffffu, 0u}) {
        int iters = 2000;
        for (int pass = 0; pass < 2; ++pass) {
            hipEventRecord(e0);
            mfma_loop<<<blocks, 256>>>(out, iters, mask);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)blocks * 4 * iters * 24.0 * (2.0 * 32 * 32 * 16);
            if (pass == 1)
                printf("%s data, %d waves/SIMD: %.0f TFLOP/s bf16 executed over %.1f ms (= %.1f f32-equivalent at 6 products) -> implied clock %.2f GHz\n",
                       mask ? "random" : "zero", wps, flops / ms / 1e9, ms, flops / ms / 1e9 / 6, flops / ms / 1e9 / 2516.6 * 2.4);
            iters = (int)(iters * target_ms / ms);
        }
    }
    return 0;
}
