// Fused Adam (beta1 = 0) + EMA over one flat parameter group.
//
// The reference steps three torch Adams with betas (0, 0.99) (train.py:417-432; the D group uses 0**r = 0 as well) and
// then updates the EMA copies of E/G/Gstru/Ex with two small kernels per parameter tensor (utils.py:55-60: 286
// launches per iteration).  With beta1 = 0 the first moment IS the gradient, so one pass suffices:
//     v   = beta2 * v + (1 - beta2) * g * g
//     p  -= lr * g / (sqrt(v) / sqrt(1 - beta2^t) + eps)
//     ema = decay * ema + (1 - decay) * p                     (optional)
// over a flat f32 buffer that aliases every parameter of the group (20-28 B per element, one launch).
#include "common.hpp"

namespace {

template <bool EMA>
__global__ __launch_bounds__(256) void adam_ema_kernel(float4* __restrict__ p, const float4* __restrict__ g,
                                                       float4* __restrict__ v, float4* __restrict__ ema, int64_t n4,
                                                       float lr, float beta2, float eps, float sqrt_bc2, float decay) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const float omb = 1.0f - beta2, omd = 1.0f - decay;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pp = p[i], vv = v[i];
        const float4 gg = g[i];
        float4 ee;
        if (EMA) ee = ema[i];
#define ONE(f)                                                         \
    {                                                                  \
        vv.f = vv.f * beta2 + omb * gg.f * gg.f;                       \
        const float denom = sqrtf(vv.f) / sqrt_bc2 + eps;              \
        pp.f = pp.f - lr * (gg.f / denom);                             \
        if (EMA) ee.f = ee.f * decay + omd * pp.f;                     \
    }
        ONE(x) ONE(y) ONE(z) ONE(w)
#undef ONE
        p[i] = pp;
        v[i] = vv;
        if (EMA) ema[i] = ee;
    }
}

}  // namespace

extern "C" int ideas_adam_ema(float* p, const float* g, float* v, float* ema, int64_t n, float lr, float beta2, float eps,
                              float bias_correction2, float ema_decay, void* stream) {
    if (!p || !g || !v) return IDEAS_E_NULL;
    if (n <= 0 || (n & 3)) return IDEAS_E_SHAPE;          // the flat group is padded to a multiple of 4
    if (!ideas_aligned16(p) || !ideas_aligned16(g) || !ideas_aligned16(v) || (ema && !ideas_aligned16(ema))) return IDEAS_E_ALIGN;
    if (!(bias_correction2 > 0.f)) return IDEAS_E_SHAPE;
    const int64_t n4 = n >> 2;
    int64_t grid = ideas_cdiv(n4, 256);
    if (grid > 8192) grid = 8192;
    const float sqrt_bc2 = sqrtf(bias_correction2);
    if (ema)
        hipLaunchKernelGGL(adam_ema_kernel<true>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (float4*)p,
                           (const float4*)g, (float4*)v, (float4*)ema, n4, lr, beta2, eps, sqrt_bc2, ema_decay);
    else
        hipLaunchKernelGGL(adam_ema_kernel<false>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (float4*)p,
                           (const float4*)g, (float4*)v, (float4*)nullptr, n4, lr, beta2, eps, sqrt_bc2, ema_decay);
    return ideas_launch_status();
}
