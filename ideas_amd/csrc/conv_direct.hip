// Direct (VALU) convolution + weight gradient with the ideas_conv_params parameterisation, NHWC, f32.
//
// Only the handful of tiny-K layers go here: the from-RGB 1x1 convs (Cin = 3), Gstru's first conv (Cin = N),
// to_rgb's input gradient (3 channels in) and Ex's last layer (N channels out).  They are HBM-bound (a few
// MACs per byte), so a thread per (pixel, output channel) with coalesced stores is enough; the MFMA tile
// would be >90 % padding.  It doubles as the on-device cross-check for the MFMA kernel in tests.
#include "common.hpp"

namespace {

__device__ __forceinline__ bool in_coord(int& i, int n, int reflect) {
    if (reflect) { i = reflect_coord(i, n); return true; }
    return i >= 0 && i < n;
}

__global__ __launch_bounds__(256) void conv_direct_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                          const float* __restrict__ w, const float* __restrict__ in_scale,
                                                          const float* __restrict__ out_scale, const float* __restrict__ bias,
                                                          const float* __restrict__ resid, ideas_conv_params p) {
    const int64_t total = (int64_t)p.B * p.OH * p.OW * p.Cout;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int K = p.TY * p.TX * p.Cin;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int64_t r = i;
        const int o = (int)(r % p.Cout); r /= p.Cout;
        const int ox = (int)(r % p.OW); r /= p.OW;
        const int oy = (int)(r % p.OH);
        const int b = (int)(r / p.OH);
        float acc = 0.f;
        const float* wr = w + (int64_t)o * K;
        for (int ty = 0; ty < p.TY; ++ty) {
            int iy = oy * p.sy + ty * p.dy + p.offy;
            if (!in_coord(iy, p.IH, p.reflect)) continue;
            for (int tx = 0; tx < p.TX; ++tx) {
                int ix = ox * p.sx + tx * p.dx + p.offx;
                if (!in_coord(ix, p.IW, p.reflect)) continue;
                const float* xp = x + (((int64_t)b * p.IH + iy) * p.IW + ix) * p.Cin;
                const float* wp = wr + (ty * p.TX + tx) * p.Cin;
                if (in_scale) {
                    const float* sp = in_scale + (int64_t)b * p.Cin;
                    for (int ci = 0; ci < p.Cin; ++ci) acc = fmaf(xp[ci] * sp[ci], wp[ci], acc);
                } else {
                    for (int ci = 0; ci < p.Cin; ++ci) acc = fmaf(xp[ci], wp[ci], acc);
                }
            }
        }
        float v = acc * p.gain;
        if (out_scale) v *= out_scale[(int64_t)b * p.Cout + o];
        if (bias) v += bias[o];
        if (p.act) v = (v > 0.f ? v : v * p.alpha) * p.act_gain;
        const int64_t yi = (((int64_t)b * p.YH + (oy * p.osy + p.ooy)) * p.YW + (ox * p.osx + p.oox)) * p.Cout + o;
        if (resid) v = (v + resid[yi]) * p.resid_gain;
        if (p.accumulate) y[yi] += v; else y[yi] = v;
    }
}

// gw[o][k] += sum_p G(p,o) * X(p,k): each block reduces a chunk of pixels for every (o,k) pair.
__global__ __launch_bounds__(256) void wgrad_direct_kernel(float* __restrict__ gw, const float* __restrict__ gy,
                                                           const float* __restrict__ x, const float* __restrict__ in_scale,
                                                           const float* __restrict__ out_scale, ideas_conv_params p,
                                                           int64_t pix_per_block) {
    const int K = p.TY * p.TX * p.Cin;
    const int E = p.Cout * K;
    const int64_t P = (int64_t)p.B * p.OH * p.OW;
    const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
    const int64_t p1 = (p0 + pix_per_block < P) ? p0 + pix_per_block : P;
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        // o fastest across lanes -> coalesced gy reads, broadcast x reads
        const int o = e % p.Cout;
        const int k = e / p.Cout;
        const int ci = k % p.Cin;
        const int t = k / p.Cin;
        const int tx = t % p.TX, ty = t / p.TX;
        float acc = 0.f;
        for (int64_t pp = p0; pp < p1; ++pp) {
            int64_t r = pp;
            const int ox = (int)(r % p.OW); r /= p.OW;
            const int oy = (int)(r % p.OH);
            const int b = (int)(r / p.OH);
            int iy = oy * p.sy + ty * p.dy + p.offy;
            int ix = ox * p.sx + tx * p.dx + p.offx;
            if (!in_coord(iy, p.IH, p.reflect) || !in_coord(ix, p.IW, p.reflect)) continue;
            float xv = x[(((int64_t)b * p.IH + iy) * p.IW + ix) * p.Cin + ci];
            if (in_scale) xv *= in_scale[(int64_t)b * p.Cin + ci];
            float g = gy[(((int64_t)b * p.YH + (oy * p.osy + p.ooy)) * p.YW + (ox * p.osx + p.oox)) * p.Cout + o];
            if (out_scale) g *= out_scale[(int64_t)b * p.Cout + o];
            acc = fmaf(g, xv, acc);
        }
        atomicAdd(&gw[(int64_t)o * K + k], acc * p.gain);
    }
}

int check_conv(const ideas_conv_params* p) {
    if (!p) return IDEAS_E_NULL;
    if (p->B <= 0 || p->IH <= 0 || p->IW <= 0 || p->Cin <= 0 || p->YH <= 0 || p->YW <= 0 || p->Cout <= 0) return IDEAS_E_SHAPE;
    if (p->OH <= 0 || p->OW <= 0 || p->TY <= 0 || p->TX <= 0 || p->osy <= 0 || p->osx <= 0) return IDEAS_E_SHAPE;
    if ((p->OH - 1) * p->osy + p->ooy >= p->YH || (p->OW - 1) * p->osx + p->oox >= p->YW || p->ooy < 0 || p->oox < 0)
        return IDEAS_E_SHAPE;
    return IDEAS_OK;
}

}  // namespace

extern "C" int ideas_conv_check_params(const ideas_conv_params* p) { return check_conv(p); }

extern "C" int ideas_conv_direct(void* y, const void* x, const void* wmat, const float* in_scale, const float* out_scale,
                                 const float* bias, const void* resid, const ideas_conv_params* p, int dtype,
                                 void* stream) {
    if (dtype != IDEAS_F32) return IDEAS_E_UNSUPPORTED;
    if (!y || !x || !wmat) return IDEAS_E_NULL;
    int rc = check_conv(p);
    if (rc) return rc;
    const int64_t total = (int64_t)p->B * p->OH * p->OW * p->Cout;
    int64_t grid = ideas_cdiv(total, 256);
    if (grid > 65536) grid = 65536;
    hipLaunchKernelGGL(conv_direct_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (float*)y,
                       (const float*)x, (const float*)wmat, in_scale, out_scale, bias, (const float*)resid, *p);
    return ideas_launch_status();
}

extern "C" int ideas_conv_wgrad_direct(float* gw, const void* gy, const void* x, const float* in_scale,
                                       const float* out_scale, const ideas_conv_params* p, int dtype, void* stream) {
    if (dtype != IDEAS_F32) return IDEAS_E_UNSUPPORTED;
    if (!gw || !gy || !x) return IDEAS_E_NULL;
    int rc = check_conv(p);
    if (rc) return rc;
    const int64_t P = (int64_t)p->B * p->OH * p->OW;
    int64_t blocks = ideas_cdiv(P, 256);
    if (blocks > 2048) blocks = 2048;
    const int64_t per = ideas_cdiv(P, blocks);
    blocks = ideas_cdiv(P, per);
    hipLaunchKernelGGL(wgrad_direct_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, gw,
                       (const float*)gy, (const float*)x, in_scale, out_scale, *p, per);
    return ideas_launch_status();
}
