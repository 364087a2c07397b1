// Shared helpers for the gfx950 kernels of libideas_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ideas_hip.h"

#define IDEAS_WAVE 64

static inline int ideas_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? IDEAS_OK : (int)e;
}

static inline bool ideas_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static inline int64_t ideas_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Workgroup id -> tile id so that each XCD (block b runs on XCD b % 8, private 4 MiB L2) owns a contiguous band of tiles.
// Bijective for any grid size (the q/rem form of cdna_hip_programming.md T1).
__device__ __forceinline__ int xcd_swizzle(int bid, int nblk) {
    const int q = nblk >> 3, rem = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
}

// mirror an out-of-range coordinate back into [0,n) (ReflectionPad2d semantics, no edge repeat)
__device__ __forceinline__ int reflect_coord(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

// a*b then +c with TWO roundings.  hipcc contracts `a*b + c` into one FMA by default (and __fmul_rn/__fadd_rn are
// plain operators in HIP), which would make a fused conv epilogue differ by 1 ulp from conv -> fused_bias_act.
__device__ __forceinline__ float mul_then_add(float a, float b, float c) {
#pragma clang fp contract(off)
    const float m = a * b;
    return m + c;
}
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}

// conv_b3.hip: forward-family implicit GEMM with the bf16x3 split contraction (arguments already validated;
// `wplanes` from ideas_b3_split_weights)
int ideas_b3_fwd(void* y, const void* x, const void* wplanes, const float* in_scale, const float* out_scale,
                 const float* bias, const void* resid, const ideas_conv_params* p, hipStream_t stream);
// conv_b3_wgrad.hip: weight gradient with the split contraction (arguments validated, ideas_b3_wgrad_supported)
int ideas_b3_wgrad(float* gw, const void* gy, const void* x, const float* in_scale, const float* out_scale,
                   const ideas_conv_params* p, hipStream_t stream);
// conv_b3_wino.hip: 3x3/s1/p1 Winograd F(2,3) with the split contraction (uplanes from ideas_b3_wino_split_weights)
int ideas_b3_wino_fwd(void* y, const void* x, const void* uplanes, const float* in_scale, const float* out_scale,
                      const float* bias, const void* resid, const ideas_conv_params* p, hipStream_t stream);
