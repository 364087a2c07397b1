// Input pipeline, device side: decoded images arrive as uint8 HWC (what PIL hands out); one kernel
// does torchvision's ToTensor + Normalize((0.5,)*3, (0.5,)*3) + the per-sample RandomHorizontalFlip of train.py:443-449 and
// writes the f32 NHWC activation layout of the networks.  Same operation order as the reference's transforms
// (v/255, then (v - 0.5) / 0.5), so the result is bit-identical to the CPU pipeline.
#include "common.hpp"

namespace {

__global__ __launch_bounds__(256) void image_u8_to_f32_kernel(float* __restrict__ y, const unsigned char* __restrict__ x,
                                                              const unsigned char* __restrict__ flip, int B, int H, int W, int C,
                                                              float mean, float stdv) {
    const int64_t n = (int64_t)B * H * W * C;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int c = (int)(i % C);
        int64_t r = i / C;
        const int w = (int)(r % W);
        r /= W;
        const int h = (int)(r % H);
        const int b = (int)(r / H);
        const int ws = (flip && flip[b]) ? W - 1 - w : w;
        const float v = (float)x[(((int64_t)b * H + h) * W + ws) * C + c];
        float t;
        {
#pragma clang fp contract(off)
            t = v / 255.0f;
            t = (t - mean) / stdv;
        }
        y[i] = t;
    }
}

}  // namespace

extern "C" int ideas_image_u8_to_f32(float* y, const void* x_u8, const void* flip_u8, int B, int H, int W, int C, float mean,
                                     float stdv, void* stream_) {
    if (!y || !x_u8) return IDEAS_E_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || stdv == 0.0f) return IDEAS_E_SHAPE;
    const int64_t n = (int64_t)B * H * W * C;
    int64_t grid = ideas_cdiv(n, 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(image_u8_to_f32_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream_, y,
                       (const unsigned char*)x_u8, (const unsigned char*)flip_u8, B, H, W, C, mean, stdv);
    return ideas_launch_status();
}
