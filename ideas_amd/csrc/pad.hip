// Adjoint of ReflectionPad2d(p) (models.py:102-106) in NHWC: folds the gradient of the padded tensor
// [B, H+2p, W+2p, C] back onto [B, H, W, C] in one pass.  Row y of the input appears in the padded tensor at
// y+p, and additionally at p-y (for 1 <= y <= p) and at 2(H-1)+p-y (for H-1-p <= y <= H-2); same for columns.
#include "common.hpp"

namespace {

__device__ __forceinline__ void acc_add(float4& a, const float4& v) { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
__device__ __forceinline__ void acc_add(float4& a, const ideas_bf16x4& v) { acc_add(a, to_f4(v)); }
__device__ __forceinline__ void acc_add(float& a, const float& v) { a += v; }
__device__ __forceinline__ void acc_zero(float4& a) { a = make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void acc_zero(float& a) { a = 0.f; }
template <typename V> struct AccOf { typedef V type; };
template <> struct AccOf<ideas_bf16x4> { typedef float4 type; };
__device__ __forceinline__ float4 acc_out(float4 a, float4*) { return a; }
__device__ __forceinline__ float acc_out(float a, float*) { return a; }
__device__ __forceinline__ ideas_bf16x4 acc_out(float4 a, ideas_bf16x4*) { return from_f4<ideas_bf16x4>(a); }

// V = float4 / ideas_bf16x4 (C % 4 == 0, C4 = C / 4) or float (any C, C4 = C)
template <typename V>
__global__ __launch_bounds__(256) void reflect_fold_kernel(V* __restrict__ gx, const V* __restrict__ gp, int B,
                                                           int H, int W, int C4, int pad) {
    const int64_t total = (int64_t)B * H * W * C4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int64_t r = i;
        const int c = (int)(r % C4); r /= C4;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H);
        const int b = (int)(r / H);
        int ys[3], xs[3], ny = 0, nx = 0;
        ys[ny++] = y + pad;
        if (y >= 1 && y <= pad) ys[ny++] = pad - y;
        if (y >= H - 1 - pad && y <= H - 2) ys[ny++] = 2 * (H - 1) + pad - y;
        xs[nx++] = x + pad;
        if (x >= 1 && x <= pad) xs[nx++] = pad - x;
        if (x >= W - 1 - pad && x <= W - 2) xs[nx++] = 2 * (W - 1) + pad - x;
        typename AccOf<V>::type acc;
        acc_zero(acc);
        for (int a = 0; a < ny; ++a)
            for (int e = 0; e < nx; ++e) acc_add(acc, gp[(((int64_t)b * Hp + ys[a]) * Wp + xs[e]) * C4 + c]);
        gx[i] = acc_out(acc, (V*)nullptr);
    }
}

}  // namespace

extern "C" int ideas_reflect_fold(void* gx, const void* gpadded, int B, int H, int W, int C, int pad, int dtype,
                                  void* stream) {
    if (dtype != IDEAS_F32 && dtype != IDEAS_BF16) return IDEAS_E_UNSUPPORTED;
    if (!gx || !gpadded) return IDEAS_E_NULL;
    if (dtype == IDEAS_BF16 && (C & 3)) return IDEAS_E_ALIGN;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || pad <= 0 || pad >= H || pad >= W) return IDEAS_E_SHAPE;
    const bool vec = !(C & 3) && ideas_aligned16(gx) && ideas_aligned16(gpadded);
    const int C4 = vec ? C / 4 : C;
    const int64_t total = (int64_t)B * H * W * C4;
    int64_t grid = ideas_cdiv(total, 256);
    if (grid > 16384) grid = 16384;
    if (dtype == IDEAS_BF16)
        hipLaunchKernelGGL(reflect_fold_kernel<ideas_bf16x4>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream,
                           (ideas_bf16x4*)gx, (const ideas_bf16x4*)gpadded, B, H, W, C4, pad);
    else if (vec)
        hipLaunchKernelGGL(reflect_fold_kernel<float4>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (float4*)gx,
                           (const float4*)gpadded, B, H, W, C4, pad);
    else   // channel counts that are not a multiple of 4 (the N-channel / RGB ends of Gstru, Ex, E): scalar path
        hipLaunchKernelGGL(reflect_fold_kernel<float>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (float*)gx,
                           (const float*)gpadded, B, H, W, C4, pad);
    return ideas_launch_status();
}
