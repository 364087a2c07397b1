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

// Split-K grids of the weight-gradient kernels: every tile of one K split reads the same pixels of gy / x, so the tiles of a
// split should meet in ONE XCD's L2 instead of being dealt round-robin over the eight (each XCD would then fetch the split's
// pixels from HBM itself).  1-D grid of tiles * splits blocks in split-major order, cut into eight contiguous bands by
// xcd_swizzle: XCD k runs splits [k S/8, (k+1) S/8) with all their tiles, dispatched back to back; every XCD gets the same
// number of blocks (+-1) whatever tiles and splits are.  Measured (tools/bench_igemm.py, B = 32): bf16 weight gradient of the
// 128-channel layers at 256x256 384 -> 592 TFLOP/s, 64->128 @256x256 450 -> 647; split-bf16 kernels +2-5 %.
static inline unsigned splitk_grid(int64_t tiles, int64_t splits) { return (unsigned)(tiles * splits); }
__device__ __forceinline__ void splitk_xcd_map(int L, int tiles, int splits, int& tile, int& split) {
    const int l = xcd_swizzle(L, tiles * splits);
    split = l / tiles;
    tile = l - split * tiles;
}

// ---- activation element access for the two storage dtypes: f32 arithmetic either way --------------------------------
typedef unsigned short ideas_bf16;                 // one bf16 element in HBM
struct ideas_bf16x4 { uint2 v; };                  // four consecutive bf16 (8 bytes)
__device__ __forceinline__ unsigned ideas_pk_bf16(float a, float b) {          // two f32 -> two RNE bf16 (v_cvt_pk_bf16_f32)
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ t = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2_));
}
__device__ __forceinline__ float4 to_f4(float4 v) { return v; }
__device__ __forceinline__ float4 to_f4(ideas_bf16x4 p) {
    return make_float4(__builtin_bit_cast(float, p.v.x << 16), __builtin_bit_cast(float, p.v.x & 0xffff0000u),
                       __builtin_bit_cast(float, p.v.y << 16), __builtin_bit_cast(float, p.v.y & 0xffff0000u));
}
template <typename V> __device__ __forceinline__ V from_f4(float4 v);
template <> __device__ __forceinline__ float4 from_f4<float4>(float4 v) { return v; }
template <> __device__ __forceinline__ ideas_bf16x4 from_f4<ideas_bf16x4>(float4 v) {
    ideas_bf16x4 r;
    r.v = make_uint2(ideas_pk_bf16(v.x, v.y), ideas_pk_bf16(v.z, v.w));
    return r;
}
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const ideas_bf16* p) { return __builtin_bit_cast(float, (unsigned)(*p) << 16); }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(ideas_bf16* p, float v) { *p = (ideas_bf16)(ideas_pk_bf16(v, 0.f) & 0xffffu); }

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

// ---- batched weight preparation (ideas_weight_prep_batched): one launch per derived form over a table of parameters -----------
// The derived weights (split-bf16 planes, Winograd planes, bf16 packs) are remade after every optimiser step, one small launch
// per parameter and form: ~390 launches of 5-25 us per iteration, most of them too small to fill the chip.  The table-driven
// kernels run the same per-element bodies as the single-tensor kernels (bitwise the same results) over all of a group's parameters.
__device__ __forceinline__ const ideas_prep_desc* prep_lookup(const ideas_prep_desc* tbl, int n, int& local, int& nblk) {
    int lo = 0, hi = n - 1;                        // block-uniform binary search over the first-block prefix
    const int b = (int)blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tbl[mid].block0 <= b) lo = mid; else hi = mid - 1;
    }
    local = b - tbl[lo].block0;
    nblk = tbl[lo].nblocks;
    return tbl + lo;
}
// per-form launchers (conv_b3.hip / conv_b3_wino.hip / conv_bf16.hip)
void ideas_b3_split_batched(const ideas_prep_desc* tbl, int n, int blocks, hipStream_t stream);
void ideas_b3_wino_split_batched(const ideas_prep_desc* tbl, int n, int blocks, hipStream_t stream);
void ideas_bf16_pack_batched(const ideas_prep_desc* tbl, int n, int blocks, hipStream_t stream);

// conv_b3.hip: forward-family implicit GEMM with the bf16x3 split contraction (arguments already validated;
// `wplanes` from ideas_b3_split_weights)
int ideas_b3_fwd(void* y, const void* x, const void* wplanes, const float* in_scale, const float* out_scale,
                 const float* bias, const void* resid, const ideas_conv_params* p, hipStream_t stream);
int ideas_b3_fwd_multi(int n, void* y, const void* x, const void* const* wplanes, const float* in_scale, const float* out_scale,
                       const ideas_conv_params* ps, hipStream_t stream);
// conv_b3_pw.hip: 1x1 / stride-1 layers with Cin <= 128 as a flat HBM-bound GEMM (ideas_b3_pw_ok decides; same results as ideas_b3_fwd)
int ideas_b3_pw_ok(const ideas_conv_params* p, const float* in_scale, const float* out_scale, const void* resid);
int ideas_b3_pw_fwd(void* y, const void* x, const void* wplanes, const float* bias, const void* resid, const ideas_conv_params* p,
                    hipStream_t stream);
int ideas_b3_pw_wgrad_ok(const ideas_conv_params* p, const float* in_scale, const float* out_scale);
int ideas_b3_pw_wgrad(float* gw, const void* gy, const void* x, const ideas_conv_params* p, hipStream_t stream);
// conv_b3_s2fir.hip, MODE 1: plain 3x3 / stride-2 / unpadded convs (optionally modulated) on the fused kernel's LDS image, no FIR
int ideas_b3_s2img_ok(const ideas_conv_params* p, const float* in_scale);
int ideas_b3_s2img_fwd(void* y, const void* x, const void* wplanes, const float* in_scale, const float* out_scale, const float* bias,
                       const void* resid, const ideas_conv_params* p, hipStream_t stream);
// conv_b3_tphase.hip: the four phases of a 3x3 / stride-2 / pad-0 transposed conv in one pass over a shared LDS image of the input
// (-1: the launches are not that geometry -> conv_b3_multi_kernel)
int ideas_b3_fwd_tphase(int n, void* y, const void* x, const void* const* wplanes, const float* in_scale, const float* out_scale,
                        const ideas_conv_params* ps, hipStream_t stream, ideas_conv_params* strips, const void** strip_w, int* nstrips);
// conv_b3_wgrad.hip: weight gradient with the split contraction (arguments validated, ideas_b3_wgrad_supported)
int ideas_b3_wgrad(float* gw, const void* gy, const void* x, const float* in_scale, const float* out_scale,
                   const ideas_conv_params* p, hipStream_t stream);
// conv_b3_wgrad3.hip: the same for 3x3 kernels, tap-fused with a rolling window (arguments validated, ideas_b3_wgrad3_supported)
int ideas_b3_wgrad3(float* gw, const void* gy, const void* x, const float* in_scale, const float* out_scale,
                    const ideas_conv_params* p, hipStream_t stream);
// IDEAS_B3_WGRAD3=0 in the environment keeps every weight gradient on conv_b3_wgrad.hip (A/B measurements)
bool ideas_b3_wgrad3_enabled();
// conv_b3_wino.hip: 3x3/s1/p1 Winograd F(2,3) with the split contraction (uplanes from ideas_b3_wino_split_weights)
int ideas_b3_wino_fwd(void* y, const void* x, const void* uplanes, const float* in_scale, const float* out_scale,
                      const float* bias, const void* resid, const ideas_conv_params* p, hipStream_t stream);

// conv_bf16.hip: bf16 mixed-precision family (dtype IDEAS_BF16): bf16 activations, packed bf16 weights (ideas_bf16_pack_weights)
int ideas_bf16_fwd(void* y, const void* x, const void* wpack, int per_image, const float* out_scale, const float* bias,
                   const void* resid, const ideas_conv_params* p, hipStream_t stream);
// conv_bf16_pw.hip: 1x1 / stride-1 layers with 32 / 64 / 128 input channels as a flat HBM-bound GEMM (ideas_bf16_pw_ok decides; same results)
int ideas_bf16_pw_ok(const ideas_conv_params* p, int per_image, const float* out_scale, const void* y, const void* resid);
int ideas_bf16_pw_fwd(void* y, const void* x, const void* wpack, const float* bias, const void* resid, const ideas_conv_params* p,
                      hipStream_t stream);
int ideas_bf16_fwd_multi(int n, void* y, const void* x, const void* const* wpack, int per_image, const float* out_scale,
                         const ideas_conv_params* ps, hipStream_t stream);
int ideas_bf16_wgrad(float* gw, const void* gy, const void* x, const float* in_scale, const float* out_scale,
                     const ideas_conv_params* p, hipStream_t stream);
