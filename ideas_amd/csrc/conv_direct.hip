// Direct (VALU) convolution + weight gradient with the ideas_conv_params parameterisation, NHWC, f32.
//
// Only the handful of tiny-K layers go here: the from-RGB 1x1 convs (Cin = 3), Gstru's first conv (Cin = N),
// to_rgb's input gradient (3 channels in) and Ex's last layer (N channels out).  They are HBM-bound (a few
// MACs per byte), so a thread per (pixel, output channel) with coalesced stores is enough; the MFMA tile
// would be >90 % padding.  It doubles as the on-device cross-check for the MFMA kernel in tests.
#include "common.hpp"

namespace {

// element access for the two activation dtypes (float, or bf16 stored as unsigned short): f32 arithmetic either way
typedef unsigned short bf16_t;
__device__ __forceinline__ float ldv(const float* p) { return *p; }
__device__ __forceinline__ float ldv(const bf16_t* p) { return __builtin_bit_cast(float, (unsigned)(*p) << 16); }
__device__ __forceinline__ void stv(float* p, float v) { *p = v; }
__device__ __forceinline__ void stv(bf16_t* p, float v) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const f32x2 t = {v, 0.f};
    *p = (bf16_t)(__builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2)) & 0xffffu);   // RNE
}

__device__ __forceinline__ bool in_coord(int& i, int n, int reflect) {
    if (reflect) { i = reflect_coord(i, n); return true; }
    return i >= 0 && i < n;
}

template <typename T>
__global__ __launch_bounds__(256) void conv_direct_kernel(T* __restrict__ y, const T* __restrict__ x,
                                                          const float* __restrict__ w, const float* __restrict__ in_scale,
                                                          const float* __restrict__ out_scale, const float* __restrict__ bias,
                                                          const T* __restrict__ resid, ideas_conv_params p) {
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
                const T* xp = x + (((int64_t)b * p.IH + iy) * p.IW + ix) * p.Cin;
                const float* wp = wr + (ty * p.TX + tx) * p.Cin;
                if (in_scale) {
                    const float* sp = in_scale + (int64_t)b * p.Cin;
                    for (int ci = 0; ci < p.Cin; ++ci) acc = fmaf(ldv(xp + ci) * sp[ci], wp[ci], acc);
                } else {
                    for (int ci = 0; ci < p.Cin; ++ci) acc = fmaf(ldv(xp + ci), wp[ci], acc);
                }
            }
        }
        float v = mul_rn(acc, p.gain);
        if (out_scale) v = mul_rn(v, out_scale[(int64_t)b * p.Cout + o]);
        v = mul_then_add(v, 1.0f, bias ? bias[o] : 0.f);
        if (p.act) v = (v > 0.f ? v : v * p.alpha) * p.act_gain;
        const int64_t yi = (((int64_t)b * p.YH + (oy * p.osy + p.ooy)) * p.YW + (ox * p.osx + p.oox)) * p.Cout + o;
        if (resid) v = (v + ldv(resid + yi)) * p.resid_gain;
        if (p.accumulate) v += ldv(y + yi);
        stv(y + yi, v);
    }
}

// gw[o][k] += sum_p G(p,o) * X(p,k): each block reduces a chunk of pixels for every (o,k) pair.
template <typename T>
__global__ __launch_bounds__(256) void wgrad_direct_kernel(float* __restrict__ gw, const T* __restrict__ gy,
                                                           const T* __restrict__ x, const float* __restrict__ in_scale,
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
            float xv = ldv(x + (((int64_t)b * p.IH + iy) * p.IW + ix) * p.Cin + ci);
            if (in_scale) xv *= in_scale[(int64_t)b * p.Cin + ci];
            float g = ldv(gy + (((int64_t)b * p.YH + (oy * p.osy + p.ooy)) * p.YW + (ox * p.osx + p.oox)) * p.Cout + o);
            if (out_scale) g *= out_scale[(int64_t)b * p.Cout + o];
            acc = fmaf(g, xv, acc);
        }
        atomicAdd(&gw[(int64_t)o * K + k], acc * p.gain);
    }
}

// ---- pointwise (1x1, stride 1, no padding) fast paths -------------------------------------------------
// The tiny-K layers on the path are all pointwise: from-RGB (3 -> 32/64), Gstru's N -> 32, to_rgb's input
// gradient (3 -> 128), Ex's last layer.  A thread owns ONE output channel (weights in registers), walks pixels
// with a fixed stride and writes coalesced along channels; x is a broadcast load.  Pure HBM streaming.
template <int KMAX, typename T>
__global__ __launch_bounds__(256) void pointwise_smallk_kernel(T* __restrict__ y, const T* __restrict__ x,
                                                               const float* __restrict__ w, const float* __restrict__ bias,
                                                               const T* __restrict__ resid, int64_t P, int Cin, int Cout,
                                                               float gain, int act, float alpha, float act_gain,
                                                               float resid_gain, int accumulate) {
    const int groups = blockDim.x / Cout;            // pixel lanes per block
    const int o = threadIdx.x % Cout, grp = threadIdx.x / Cout;
    if (grp >= groups) return;
    float wr[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) wr[k] = k < Cin ? w[(int64_t)o * Cin + k] : 0.f;
    const float bv = bias ? bias[o] : 0.f;
    const int64_t stride = (int64_t)gridDim.x * groups;
    for (int64_t pp = (int64_t)blockIdx.x * groups + grp; pp < P; pp += stride) {
        const T* xp = x + pp * Cin;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < Cin) acc = fmaf(ldv(xp + k), wr[k], acc);
        float v = mul_then_add(acc, gain, bv);   // no FMA contraction: bitwise the unfused conv -> bias_act
        if (act) v = (v > 0.f ? v : v * alpha) * act_gain;
        const int64_t yi = pp * Cout + o;
        if (resid) v = (v + ldv(resid + yi)) * resid_gain;
        if (accumulate) v += ldv(y + yi);
        stv(y + yi, v);
    }
}

// gw[o][ci] += gain * sum_p gy[p][o] * x[p][ci] with min(Cin, Cout) <= 8.  WIDE_OUT: threads span o (gy coalesced,
// x broadcast, Cin accumulators); otherwise threads span ci (x coalesced, gy broadcast, Cout accumulators).
template <int SMALL, bool WIDE_OUT, typename T>
__global__ __launch_bounds__(256) void pointwise_small_wgrad_kernel(float* __restrict__ gw, const T* __restrict__ gy,
                                                                    const T* __restrict__ x, int64_t P, int Cin, int Cout,
                                                                    float gain, int64_t pix_per_block) {
    __shared__ float s_red[256 * 8];   // [wide][SMALL] block-level partial sums
    const int wide = WIDE_OUT ? Cout : Cin, small = WIDE_OUT ? Cin : Cout;
    const int groups = blockDim.x / wide;
    const int c = threadIdx.x % wide, grp = threadIdx.x / wide;
    for (int i = threadIdx.x; i < wide * SMALL; i += blockDim.x) s_red[i] = 0.f;
    __syncthreads();
    if (grp < groups) {
        const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
        const int64_t p1 = (p0 + pix_per_block < P) ? p0 + pix_per_block : P;
        float acc[SMALL];
#pragma unroll
        for (int k = 0; k < SMALL; ++k) acc[k] = 0.f;
        for (int64_t pp = p0 + grp; pp < p1; pp += groups) {
            const float a = WIDE_OUT ? ldv(gy + pp * Cout + c) : ldv(x + pp * Cin + c);
            const T* bp = WIDE_OUT ? x + pp * Cin : gy + pp * Cout;
#pragma unroll
            for (int k = 0; k < SMALL; ++k)
                if (k < small) acc[k] = fmaf(a, ldv(bp + k), acc[k]);
        }
#pragma unroll
        for (int k = 0; k < SMALL; ++k)
            if (k < small) atomicAdd(&s_red[c * SMALL + k], acc[k]);   // LDS: fold the pixel groups of this block
    }
    __syncthreads();
    // one global atomic per (block, output element): few blocks x few outputs, so no hot-address serialisation
    for (int i = threadIdx.x; i < wide * SMALL; i += blockDim.x) {
        const int cc = i / SMALL, k = i % SMALL;
        if (k < small) {
            const int o = WIDE_OUT ? cc : k, ci = WIDE_OUT ? k : cc;
            atomicAdd(&gw[(int64_t)o * Cin + ci], s_red[i] * gain);
        }
    }
}

bool is_pointwise(const ideas_conv_params* p) {
    return p->TY == 1 && p->TX == 1 && p->sy == 1 && p->sx == 1 && p->offy == 0 && p->offx == 0 && p->osy == 1 &&
           p->osx == 1 && p->ooy == 0 && p->oox == 0 && p->IH == p->OH && p->IW == p->OW && p->YH == p->OH && p->YW == p->OW;
}

int check_conv(const ideas_conv_params* p) {
    if (!p) return IDEAS_E_NULL;
    if (p->B <= 0 || p->IH <= 0 || p->IW <= 0 || p->Cin <= 0 || p->YH <= 0 || p->YW <= 0 || p->Cout <= 0) return IDEAS_E_SHAPE;
    if (p->OH <= 0 || p->OW <= 0 || p->TY <= 0 || p->TX <= 0 || p->osy <= 0 || p->osx <= 0) return IDEAS_E_SHAPE;
    if ((p->OH - 1) * p->osy + p->ooy >= p->YH || (p->OW - 1) * p->osx + p->oox >= p->YW || p->ooy < 0 || p->oox < 0)
        return IDEAS_E_SHAPE;
    return IDEAS_OK;
}

template <typename T>
int conv_direct_impl(void* y, const void* x, const void* wmat, const float* in_scale, const float* out_scale, const float* bias,
                     const void* resid, const ideas_conv_params* p, void* stream) {
    if (is_pointwise(p) && !in_scale && !out_scale && p->Cin <= 8 && p->Cout <= 256) {
        const int64_t P = (int64_t)p->B * p->OH * p->OW;
        const int groups = 256 / p->Cout;
        int64_t grid = ideas_cdiv(P, (int64_t)groups * 8);
        if (grid > 8192) grid = 8192;
        if (grid < 1) grid = 1;
        hipLaunchKernelGGL((pointwise_smallk_kernel<8, T>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (T*)y,
                           (const T*)x, (const float*)wmat, bias, (const T*)resid, P, p->Cin, p->Cout, p->gain, p->act, p->alpha,
                           p->act_gain, p->resid_gain, p->accumulate);
        return ideas_launch_status();
    }
    const int64_t total = (int64_t)p->B * p->OH * p->OW * p->Cout;
    int64_t grid = ideas_cdiv(total, 256);
    if (grid > 65536) grid = 65536;
    hipLaunchKernelGGL(conv_direct_kernel<T>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (T*)y, (const T*)x,
                       (const float*)wmat, in_scale, out_scale, bias, (const T*)resid, *p);
    return ideas_launch_status();
}

template <typename T>
int wgrad_direct_impl(float* gw, const void* gy, const void* x, const float* in_scale, const float* out_scale,
                      const ideas_conv_params* p, void* stream) {
    const int64_t P = (int64_t)p->B * p->OH * p->OW;
    if (is_pointwise(p) && !in_scale && !out_scale && (p->Cin <= 8 || p->Cout <= 8) && p->Cin <= 256 && p->Cout <= 256) {
        int64_t blocks = ideas_cdiv(P, 512);
        if (blocks > 1024) blocks = 1024;
        const int64_t per = ideas_cdiv(P, blocks);
        blocks = ideas_cdiv(P, per);
        if (p->Cin <= 8 && p->Cin <= p->Cout)
            hipLaunchKernelGGL((pointwise_small_wgrad_kernel<8, true, T>), dim3((unsigned)blocks), dim3(256), 0,
                               (hipStream_t)stream, gw, (const T*)gy, (const T*)x, P, p->Cin, p->Cout, p->gain, per);
        else
            hipLaunchKernelGGL((pointwise_small_wgrad_kernel<8, false, T>), dim3((unsigned)blocks), dim3(256), 0,
                               (hipStream_t)stream, gw, (const T*)gy, (const T*)x, P, p->Cin, p->Cout, p->gain, per);
        return ideas_launch_status();
    }
    int64_t blocks = ideas_cdiv(P, 256);
    if (blocks > 2048) blocks = 2048;
    const int64_t per = ideas_cdiv(P, blocks);
    blocks = ideas_cdiv(P, per);
    hipLaunchKernelGGL(wgrad_direct_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, gw, (const T*)gy,
                       (const T*)x, in_scale, out_scale, *p, per);
    return ideas_launch_status();
}

}  // namespace

extern "C" int ideas_conv_check_params(const ideas_conv_params* p) { return check_conv(p); }

/* 1 if the VALU kernels are the intended bf16 path for this geometry (the HBM-bound tiny-K / tiny-N layers: from-RGB, to-RGB
 * and their gradients); everything else without an MFMA bf16 kernel is computed in f32 by the caller. */
extern "C" int ideas_bf16_direct_supported(const ideas_conv_params* p) {
    if (!p || check_conv(p)) return 0;
    return is_pointwise(p) && (p->Cin <= 8 || p->Cout <= 8) && p->Cin <= 256 && p->Cout <= 256;
}

extern "C" int ideas_conv_direct(void* y, const void* x, const void* wmat, const float* in_scale, const float* out_scale,
                                 const float* bias, const void* resid, const ideas_conv_params* p, int dtype,
                                 void* stream) {
    if (dtype != IDEAS_F32 && dtype != IDEAS_BF16) return IDEAS_E_UNSUPPORTED;
    if (!y || !x || !wmat) return IDEAS_E_NULL;
    int rc = check_conv(p);
    if (rc) return rc;
    if (dtype == IDEAS_BF16) return conv_direct_impl<bf16_t>(y, x, wmat, in_scale, out_scale, bias, resid, p, stream);
    return conv_direct_impl<float>(y, x, wmat, in_scale, out_scale, bias, resid, p, stream);
}

extern "C" int ideas_conv_wgrad_direct(float* gw, const void* gy, const void* x, const float* in_scale,
                                       const float* out_scale, const ideas_conv_params* p, int dtype, void* stream) {
    if (dtype != IDEAS_F32 && dtype != IDEAS_BF16) return IDEAS_E_UNSUPPORTED;
    if (!gw || !gy || !x) return IDEAS_E_NULL;
    int rc = check_conv(p);
    if (rc) return rc;
    if (dtype == IDEAS_BF16) return wgrad_direct_impl<bf16_t>(gw, gy, x, in_scale, out_scale, p, stream);
    return wgrad_direct_impl<float>(gw, gy, x, in_scale, out_scale, p, stream);
}
