// patchify_image (utils.py:127-149), device side: the n_crop boxes of one call are cropped out of every image of the batch and
// resized bilinearly (F.interpolate(mode="bilinear", align_corners=False): the area_pixel source index scale*(dst+0.5)-0.5 clamped
// at 0, neighbour clamped at the crop's last row / column) to out_h x out_w, stacked image-major: [B * n_crop, C, out_h, out_w].
// The reference issues one interpolate per box over a strided NCHW view plus a stack; on channels_last images with C = 3 that was
// 88 launches of ~59 us per iteration (5.2 ms) for 1.5 MB of output each.  Here: ONE launch per call, one thread per output
// pixel (all channels), NHWC in and out; the boxes travel as a by-value kernel argument.  The gradient scatters with f32 atomics
// into a zeroed image (crops overlap), as torch's own backward does.
#include "common.hpp"

namespace {

constexpr int MAX_BOXES = 64;
struct Boxes { int v[MAX_BOXES][4]; };    // (y, x, h, w) of every crop, in source pixels

struct Tap { int i0, i1; float l0, l1; };
__device__ __forceinline__ Tap bilinear_tap(int dst, int in, int out) {
    const float scale = (float)in / (float)out;
    float src;
    {
#pragma clang fp contract(off)
        src = scale * ((float)dst + 0.5f) - 0.5f;
    }
    if (src < 0.f) src = 0.f;
    Tap t;
    t.i0 = (int)src;
    if (t.i0 > in - 1) t.i0 = in - 1;
    t.i1 = t.i0 + (t.i0 < in - 1 ? 1 : 0);
    t.l1 = src - (float)t.i0;
    t.l0 = 1.f - t.l1;
    return t;
}

template <typename T, int C>
__global__ __launch_bounds__(256) void patch_resize_kernel(T* __restrict__ y, const T* __restrict__ x, Boxes boxes, int B, int H,
                                                          int W, int n_crop, int OH, int OW) {
    const int64_t n = (int64_t)B * n_crop * OH * OW;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int ox = (int)(i % OW);
    int64_t r = i / OW;
    const int oy = (int)(r % OH);
    r /= OH;
    const int k = (int)(r % n_crop);
    const int b = (int)(r / n_crop);
    const int by = boxes.v[k][0], bx = boxes.v[k][1], bh = boxes.v[k][2], bw = boxes.v[k][3];
    const Tap ty = bilinear_tap(oy, bh, OH), tx = bilinear_tap(ox, bw, OW);
    const T* r0 = x + (((int64_t)b * H + by + ty.i0) * W + bx) * C;
    const T* r1 = x + (((int64_t)b * H + by + ty.i1) * W + bx) * C;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float a = ld1(r0 + tx.i0 * C + c), bb = ld1(r0 + tx.i1 * C + c);
        const float cc = ld1(r1 + tx.i0 * C + c), d = ld1(r1 + tx.i1 * C + c);
        st1(y + i * C + c, ty.l0 * (tx.l0 * a + tx.l1 * bb) + ty.l1 * (tx.l0 * cc + tx.l1 * d));
    }
}

template <typename T, int C>
__global__ __launch_bounds__(256) void patch_resize_bwd_kernel(float* __restrict__ gx, const T* __restrict__ gy, Boxes boxes, int B,
                                                              int H, int W, int n_crop, int OH, int OW) {
    const int64_t n = (int64_t)B * n_crop * OH * OW;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int ox = (int)(i % OW);
    int64_t r = i / OW;
    const int oy = (int)(r % OH);
    r /= OH;
    const int k = (int)(r % n_crop);
    const int b = (int)(r / n_crop);
    const int by = boxes.v[k][0], bx = boxes.v[k][1], bh = boxes.v[k][2], bw = boxes.v[k][3];
    const Tap ty = bilinear_tap(oy, bh, OH), tx = bilinear_tap(ox, bw, OW);
    float* r0 = gx + (((int64_t)b * H + by + ty.i0) * W + bx) * C;
    float* r1 = gx + (((int64_t)b * H + by + ty.i1) * W + bx) * C;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float g = ld1(gy + i * C + c);
        atomicAdd(r0 + tx.i0 * C + c, ty.l0 * tx.l0 * g);
        atomicAdd(r0 + tx.i1 * C + c, ty.l0 * tx.l1 * g);
        atomicAdd(r1 + tx.i0 * C + c, ty.l1 * tx.l0 * g);
        atomicAdd(r1 + tx.i1 * C + c, ty.l1 * tx.l1 * g);
    }
}

int check_boxes(const int* boxes, int n_crop, int H, int W, Boxes& out) {
    if (n_crop <= 0 || n_crop > MAX_BOXES) return IDEAS_E_SHAPE;
    for (int k = 0; k < n_crop; ++k) {
        const int y = boxes[4 * k], x = boxes[4 * k + 1], h = boxes[4 * k + 2], w = boxes[4 * k + 3];
        if (y < 0 || x < 0 || h <= 0 || w <= 0 || y + h > H || x + w > W) return IDEAS_E_SHAPE;
        out.v[k][0] = y; out.v[k][1] = x; out.v[k][2] = h; out.v[k][3] = w;
    }
    return 0;
}

}  // namespace

extern "C" int ideas_patch_resize(void* y, const void* x, const int* boxes, int n_crop, int B, int C, int H, int W, int out_h,
                                  int out_w, int dtype, void* stream_) {
    if (!y || !x || !boxes) return IDEAS_E_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || out_h <= 0 || out_w <= 0) return IDEAS_E_SHAPE;
    if (C != 3 && C != 1) return IDEAS_E_UNSUPPORTED;
    if (dtype != IDEAS_F32 && dtype != IDEAS_BF16) return IDEAS_E_UNSUPPORTED;
    Boxes bx;
    if (const int rc = check_boxes(boxes, n_crop, H, W, bx)) return rc;
    const int64_t n = (int64_t)B * n_crop * out_h * out_w;
    const dim3 grid((unsigned)ideas_cdiv(n, 256)), block(256);
    hipStream_t s = (hipStream_t)stream_;
#define GO(T, CC) hipLaunchKernelGGL((patch_resize_kernel<T, CC>), grid, block, 0, s, (T*)y, (const T*)x, bx, B, H, W, n_crop, out_h, out_w)
    if (dtype == IDEAS_F32) { if (C == 3) GO(float, 3); else GO(float, 1); }
    else { if (C == 3) GO(ideas_bf16, 3); else GO(ideas_bf16, 1); }
#undef GO
    return ideas_launch_status();
}

extern "C" int ideas_patch_resize_bwd(float* gx, const void* gy, const int* boxes, int n_crop, int B, int C, int H, int W, int out_h,
                                      int out_w, int clear, int dtype, void* stream_) {
    if (!gx || !gy || !boxes) return IDEAS_E_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || out_h <= 0 || out_w <= 0) return IDEAS_E_SHAPE;
    if (C != 3 && C != 1) return IDEAS_E_UNSUPPORTED;
    if (dtype != IDEAS_F32 && dtype != IDEAS_BF16) return IDEAS_E_UNSUPPORTED;
    Boxes bx;
    if (const int rc = check_boxes(boxes, n_crop, H, W, bx)) return rc;
    hipStream_t s = (hipStream_t)stream_;
    if (clear) {
        const hipError_t e = hipMemsetAsync(gx, 0, (size_t)B * H * W * C * sizeof(float), s);
        if (e != hipSuccess) return (int)e;
    }
    const int64_t n = (int64_t)B * n_crop * out_h * out_w;
    const dim3 grid((unsigned)ideas_cdiv(n, 256)), block(256);
#define GO(T, CC) hipLaunchKernelGGL((patch_resize_bwd_kernel<T, CC>), grid, block, 0, s, gx, (const T*)gy, bx, B, H, W, n_crop, out_h, out_w)
    if (dtype == IDEAS_F32) { if (C == 3) GO(float, 3); else GO(float, 1); }
    else { if (C == 3) GO(ideas_bf16, 3); else GO(ideas_bf16, 1); }
#undef GO
    return ideas_launch_status();
}
