// Small reductions around the modulated convolution (stylegan2/model.py:239-244).
//
//   ideas_demod      d[b,o] = rsqrt(sum_i s[b,i]^2 * wsq[o,i] + eps): one wavefront per (b,o), 64-lane shuffle
//                    reduction over Cin.  This is the reference's weight.pow(2).sum([2,3,4]) without the
//                    [B,Cout,Cin,3,3] per-sample weight tensor: sum_{i,k}(scale*W[o,i,k]*s[b,i])^2
//                    = sum_i s[b,i]^2 * (scale^2 * sum_k W[o,i,k]^2).
//   ideas_pixel_dot  out[b,c] += sum_p a[b,p,c] * g[b,p,c] (NHWC): the two per-sample reductions the modconv
//                    backward needs (d style = <x, dx/s>, d demod = <dy, y/d>).
//
// Accuracy (round 3).  The style gradient gs = <x, gx>/s + 2 s sum_o gq wsq is the sum of two terms that largely CANCEL (the
// demodulated output does not change when s is rescaled), so a relative error eps on either term becomes eps |term| / |gs| on the
// result -- measured 4x the f32 CPU reference's error on G's modulation weights at 256x256 when the 65 536-pixel dot products
// and the demodulation algebra ran in f32.  These kernels are HBM-bound (pixel_dot, act_bwd_dot) or tiny (demod_bwd), so wider
// arithmetic is free: products and sums of the dot products are DOUBLE from the first multiply to the global accumulator
// (ds_add_f64 / global_atomic_add_f64), and demod_bwd evaluates q = sum_i s^2 wsq, d = (q + eps)^-1/2 and both terms of gs in
// double from the f32 styles and a double wsq; only gs / gq are rounded to f32, once.
#include "common.hpp"

namespace {

__global__ __launch_bounds__(256) void demod_kernel(float* __restrict__ d, const float* __restrict__ s,
                                                    const float* __restrict__ wsq, int B, int Cin, int Cout, float eps) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t idx = (int64_t)blockIdx.x * 4 + wave;
    if (idx >= (int64_t)B * Cout) return;
    const int b = (int)(idx / Cout), o = (int)(idx % Cout);
    const float* sp = s + (int64_t)b * Cin;
    const float* wp = wsq + (int64_t)o * Cin;
    float acc = 0.f;
    for (int i = lane; i < Cin; i += 64) { const float sv = sp[i]; acc = fmaf(sv * sv, wp[i], acc); }
    acc = wave_sum(acc);
    if (lane == 0) d[idx] = rsqrtf(acc + eps);
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// wsq[o][i] = scale2 * sum_k W[o][i][k]^2 of a 4-D weight of any strides (one thread per (o, i); the fastest-varying thread
// index follows the smaller of the two channel strides so the reads coalesce in either memory format).
template <typename ACC>
__global__ __launch_bounds__(256) void weight_sqsum_kernel(ACC* __restrict__ wsq, const float* __restrict__ w, int Cout, int Cin,
                                                           int KH, int KW, int64_t so, int64_t si, int64_t sky, int64_t skx,
                                                           ACC scale2) {
    const int64_t n = (int64_t)Cout * Cin;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    int o, i;
    if (so < si) { o = (int)(t % Cout); i = (int)(t / Cout); } else { i = (int)(t % Cin); o = (int)(t / Cin); }
    const float* p = w + o * so + i * si;
    ACC acc = 0;
    for (int ky = 0; ky < KH; ++ky)
        for (int kx = 0; kx < KW; ++kx) { const ACC v = p[ky * sky + kx * skx]; acc = v * v + acc; }
    wsq[(int64_t)o * Cin + i] = acc * scale2;
}

// Backward of the demodulation folded into the style gradient, in double.  Inputs: the two per-sample reductions of the conv
// backward (double accumulators),
//   dot_s[b,i] = <x, gx>[b,i]  (gx = s * dL/d(s x))        dot_d[b,o] = <gy, y>[b,o]  (y = d32 * conv, d32 = the f32 factor the
//                                                                                       forward multiplied by)
// Outputs: gq[b,o] = dL/dq of d = (q + eps)^(-1/2), q[b,o] = sum_i s^2 wsq:  dL/dd = dot_d / d32 (exact for the y that was
//          computed), dd/dq = -0.5 d^3 with d re-evaluated in double  =>  gq = -0.5 * (dot_d / d32) * d^3;
//          gs[b,i] = (s != 0 ? dot_s / s : 0) + 2 s[b,i] * sum_o gq[b,o] * wsq[o,i]      (d32, dot_d == NULL: first term only).
// Two launches: demod_gq_kernel, one wave per (b, o), leaves gq in float (for ideas_demod_wgrad) and in double IN PLACE of dot_d;
// demod_gs_kernel, 64 input channels x 4 slices of the o sum per block, reads the doubles back.  (One kernel with a block per 256
// input channels that recomputed gq[b, :] itself ran 64 blocks of 1024 serial double steps: 218 us per call, 10.5 ms per iteration.)
__global__ __launch_bounds__(256) void demod_gq_kernel(float* __restrict__ gq, double* __restrict__ dot_d, const float* __restrict__ d32,
                                                       const float* __restrict__ s, const double* __restrict__ wsq, int Cin, int Cout,
                                                       float eps) {
    const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + wave;
    if (o >= Cout) return;
    const float* sp = s + (int64_t)b * Cin;
    const double* wp = wsq + (int64_t)o * Cin;
    double q = 0.0;
    for (int i = lane; i < Cin; i += 64) { const double sv = sp[i]; q = fma(sv * sv, wp[i], q); }
    q = wave_sum_f64(q);
    if (lane == 0) {
        const double d2 = 1.0 / (q + (double)eps);
        const double g = -0.5 * (dot_d[(int64_t)b * Cout + o] / (double)d32[(int64_t)b * Cout + o]) * d2 * sqrt(d2);
        dot_d[(int64_t)b * Cout + o] = g;
        gq[(int64_t)b * Cout + o] = (float)g;
    }
}

__global__ __launch_bounds__(256) void demod_gs_kernel(float* __restrict__ gs, const double* __restrict__ dot_s,
                                                       const double* __restrict__ gq64, const float* __restrict__ s,
                                                       const double* __restrict__ wsq, int Cin, int Cout) {
    __shared__ double part[4][64];
    const int b = blockIdx.y, il = threadIdx.x & 63, og = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + il;
    double acc = 0.0;
    if (gq64 && i < Cin) {
        const double* gp = gq64 + (int64_t)b * Cout;
        for (int o = og; o < Cout; o += 4) acc = fma(gp[o], wsq[(int64_t)o * Cin + i], acc);
    }
    part[og][il] = acc;
    __syncthreads();
    if (og != 0 || i >= Cin) return;
    const double sv = s[(int64_t)b * Cin + i];
    double g = sv != 0.0 ? dot_s[(int64_t)b * Cin + i] / sv : 0.0;
    if (gq64) g = fma(2.0 * sv, (part[0][il] + part[1][il]) + (part[2][il] + part[3][il]), g);
    gs[(int64_t)b * Cin + i] = (float)g;
}

// Weight gradient through the demodulation:  gw[o][i][k] += coef * W[o][i][k] * sum_b gq[b,o] * s[b,i]^2   (coef = 2 scale^2),
// added in place to a gradient tensor of strides (go, gi, gky, gkx) -- the parameter's .grad or a fresh buffer.
__global__ __launch_bounds__(256) void demod_wgrad_kernel(float* __restrict__ gw, const float* __restrict__ w, const float* __restrict__ gq,
                                                          const float* __restrict__ s, int B, int Cout, int Cin, int KH, int KW,
                                                          int64_t so, int64_t si, int64_t sky, int64_t skx, int64_t go, int64_t gi,
                                                          int64_t gky, int64_t gkx, float coef) {
    const int64_t n = (int64_t)Cout * Cin;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    int o, i;
    if (so < si) { o = (int)(t % Cout); i = (int)(t / Cout); } else { i = (int)(t % Cin); o = (int)(t / Cin); }
    float acc = 0.f;
    for (int b = 0; b < B; ++b) { const float sv = s[(int64_t)b * Cin + i]; acc = fmaf(gq[(int64_t)b * Cout + o], sv * sv, acc); }
    acc *= coef;
    const float* wp = w + o * so + i * si;
    float* gp = gw + o * go + i * gi;
    for (int ky = 0; ky < KH; ++ky)
        for (int kx = 0; kx < KW; ++kx) gp[ky * gky + kx * gkx] = fmaf(acc, wp[ky * sky + kx * skx], gp[ky * gky + kx * gkx]);
}

template <typename T, typename V>      // T = element (float / bf16), V = four consecutive elements
__global__ __launch_bounds__(256) void pixel_dot_kernel(double* __restrict__ out, const T* __restrict__ a,
                                                        const T* __restrict__ g, int64_t P, int C,
                                                        int64_t pix_per_block) {
    extern __shared__ double s_acc[];
    const int b = blockIdx.y;
    const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
    const int64_t p1 = (p0 + pix_per_block < P) ? p0 + pix_per_block : P;
    for (int c = threadIdx.x; c < C; c += blockDim.x) s_acc[c] = 0.0;
    __syncthreads();
    const T* ab = a + (int64_t)b * P * C;
    const T* gb = g + (int64_t)b * P * C;
    if ((C & 3) == 0) {
        const int C4 = C >> 2;
        auto dot_rows = [&](int c4, int pr, int R) {
            double ax = 0.0, ay = 0.0, az = 0.0, aw = 0.0;
            for (int64_t pp = p0 + pr; pp < p1; pp += R) {
                const float4 av = to_f4(*reinterpret_cast<const V*>(ab + pp * C + c4 * 4));
                const float4 gv = to_f4(*reinterpret_cast<const V*>(gb + pp * C + c4 * 4));
                ax = fma((double)av.x, (double)gv.x, ax); ay = fma((double)av.y, (double)gv.y, ay);
                az = fma((double)av.z, (double)gv.z, az); aw = fma((double)av.w, (double)gv.w, aw);
            }
            atomicAdd(&s_acc[c4 * 4 + 0], ax); atomicAdd(&s_acc[c4 * 4 + 1], ay);
            atomicAdd(&s_acc[c4 * 4 + 2], az); atomicAdd(&s_acc[c4 * 4 + 3], aw);
        };
        if (C4 <= 256) {
            // thread = (pixel row pr, channel quad c4): fixed channels per thread, coalesced along C
            const int R = 256 / C4;
            const int c4 = (int)threadIdx.x % C4, pr = (int)threadIdx.x / C4;
            if (pr < R) dot_rows(c4, pr, R);
        } else {
            for (int c4 = threadIdx.x; c4 < C4; c4 += 256) dot_rows(c4, 0, 1);
        }
    } else {
        for (int64_t i = p0 * C + threadIdx.x; i < p1 * C; i += blockDim.x)
            atomicAdd(&s_acc[(int)(i % C)], (double)ld1(ab + i) * (double)ld1(gb + i));
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) atomicAdd(&out[(int64_t)b * C + c], s_acc[c]);
}

// Backward prologue of a fused (modulated conv + bias + leaky-ReLU): one pass over (gy, out) that produces
//   g_pre = (out > 0 ? gy : gy*alpha) * act_gain        (the gradient w.r.t. the pre-activation)
//   bgrad[c]   += sum_{b,p} g_pre
//   dot[b,c]   += sum_p g_pre * pre,   pre = inverse_act(out) - bias[c]   (= demodulated conv output)
// so the pre-activation tensor never has to be kept for the backward (d(demod) = dot / demod).
template <typename T, typename V>
__global__ __launch_bounds__(256) void act_bwd_dot_kernel(T* __restrict__ gpre, float* __restrict__ bgrad,
                                                          double* __restrict__ dot, const T* __restrict__ gy,
                                                          const T* __restrict__ out, const float* __restrict__ bias,
                                                          const float* __restrict__ gscale, int64_t P, int C,
                                                          int64_t pix_per_block, float alpha, float act_gain) {
    extern __shared__ double s_acc[];   // [2][C]: dot partials, bias-grad partials (double: see the header of this file)
    const int b = blockIdx.y;
    const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
    const int64_t p1 = (p0 + pix_per_block < P) ? p0 + pix_per_block : P;
    for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) s_acc[c] = 0.0;
    __syncthreads();
    const int64_t base = (int64_t)b * P * C;
    const int C4 = C >> 2;
    const float inv_gain = 1.0f / act_gain, inv_alpha = 1.0f / alpha;
    auto rows = [&](int c4, int pr, int R) {
        struct { double x, y, z, w; } accd = {0.0, 0.0, 0.0, 0.0}, accb = {0.0, 0.0, 0.0, 0.0};
        const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        // optional per-(b, c) factor on the stored gradient only (bf16 path: the demodulation d[b,c], so that the input- and
        // weight-gradient kernels need no per-sample input scale); bias_grad / dot are taken before it
        const float4 gs = gscale ? *reinterpret_cast<const float4*>(gscale + (int64_t)b * C + c4 * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
        for (int64_t pp = p0 + pr; pp < p1; pp += R) {
            const int64_t off = base + pp * C + c4 * 4;
            const float4 g = to_f4(*reinterpret_cast<const V*>(gy + off));
            const float4 o = to_f4(*reinterpret_cast<const V*>(out + off));
            float4 gp;
#define ONE(f)                                                                    \
    {                                                                             \
        const float gsel = (o.f > 0.f) ? g.f : g.f * alpha;                       \
        gp.f = gsel * act_gain;                                                   \
        const float t = o.f * inv_gain;                                           \
        const float pre = ((t > 0.f) ? t : t * inv_alpha) - bv.f;                 \
        accd.f = fma((double)gp.f, (double)pre, accd.f);                          \
        accb.f += (double)gp.f;                                                   \
    }
            ONE(x) ONE(y) ONE(z) ONE(w)
#undef ONE
            *reinterpret_cast<V*>(gpre + off) = from_f4<V>(make_float4(gp.x * gs.x, gp.y * gs.y, gp.z * gs.z, gp.w * gs.w));
        }
        atomicAdd(&s_acc[c4 * 4 + 0], accd.x); atomicAdd(&s_acc[c4 * 4 + 1], accd.y);
        atomicAdd(&s_acc[c4 * 4 + 2], accd.z); atomicAdd(&s_acc[c4 * 4 + 3], accd.w);
        atomicAdd(&s_acc[C + c4 * 4 + 0], accb.x); atomicAdd(&s_acc[C + c4 * 4 + 1], accb.y);
        atomicAdd(&s_acc[C + c4 * 4 + 2], accb.z); atomicAdd(&s_acc[C + c4 * 4 + 3], accb.w);
    };
    if (C4 <= 256) {
        const int R = 256 / C4;
        const int c4 = (int)threadIdx.x % C4, pr = (int)threadIdx.x / C4;
        if (pr < R) rows(c4, pr, R);
    } else {
        for (int c4 = threadIdx.x; c4 < C4; c4 += 256) rows(c4, 0, 1);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        atomicAdd(&dot[(int64_t)b * C + c], s_acc[c]);
        atomicAdd(&bgrad[c], (float)s_acc[C + c]);
    }
}

}  // namespace

extern "C" int ideas_demod(float* d, const float* s, const float* wsq, int B, int Cin, int Cout, float eps,
                           void* stream) {
    if (!d || !s || !wsq) return IDEAS_E_NULL;
    if (B <= 0 || Cin <= 0 || Cout <= 0) return IDEAS_E_SHAPE;
    const int64_t waves = (int64_t)B * Cout;
    hipLaunchKernelGGL(demod_kernel, dim3((unsigned)ideas_cdiv(waves, 4)), dim3(256), 0, (hipStream_t)stream, d, s, wsq,
                       B, Cin, Cout, eps);
    return ideas_launch_status();
}

extern "C" int ideas_weight_sqsum(float* wsq, const float* w, int Cout, int Cin, int KH, int KW, int64_t so, int64_t si, int64_t sky,
                                  int64_t skx, float scale2, void* stream) {
    if (!wsq || !w) return IDEAS_E_NULL;
    if (Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0) return IDEAS_E_SHAPE;
    const int64_t n = (int64_t)Cout * Cin;
    hipLaunchKernelGGL(weight_sqsum_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, wsq, w, Cout, Cin, KH,
                       KW, so, si, sky, skx, scale2);
    return ideas_launch_status();
}

extern "C" int ideas_weight_sqsum_f64(double* wsq, const float* w, int Cout, int Cin, int KH, int KW, int64_t so, int64_t si, int64_t sky,
                                      int64_t skx, double scale2, void* stream) {
    if (!wsq || !w) return IDEAS_E_NULL;
    if (Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0) return IDEAS_E_SHAPE;
    const int64_t n = (int64_t)Cout * Cin;
    hipLaunchKernelGGL(weight_sqsum_kernel<double>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, wsq, w, Cout, Cin,
                       KH, KW, so, si, sky, skx, scale2);
    return ideas_launch_status();
}

extern "C" int ideas_demod_bwd(float* gs, float* gq, const double* dot_s, double* dot_d, const float* d, const float* s,
                               const double* wsq, int B, int Cin, int Cout, float eps, void* stream) {
    if (!gs || !dot_s || !s) return IDEAS_E_NULL;
    if (d && (!gq || !dot_d || !wsq)) return IDEAS_E_NULL;
    if (B <= 0 || Cin <= 0 || Cout <= 0 || B > 65535) return IDEAS_E_SHAPE;
    if (d)
        hipLaunchKernelGGL(demod_gq_kernel, dim3((Cout + 3) / 4, B), dim3(256), 0, (hipStream_t)stream, gq, dot_d, d, s, wsq, Cin, Cout, eps);
    hipLaunchKernelGGL(demod_gs_kernel, dim3((Cin + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, gs, dot_s,
                       d ? (const double*)dot_d : (const double*)nullptr, s, wsq, Cin, Cout);
    return ideas_launch_status();
}

extern "C" int ideas_demod_wgrad(float* gw, const float* w, const float* gq, const float* s, int B, int Cout, int Cin, int KH, int KW,
                                 int64_t so, int64_t si, int64_t sky, int64_t skx, int64_t go, int64_t gi, int64_t gky, int64_t gkx,
                                 float coef, void* stream) {
    if (!gw || !w || !gq || !s) return IDEAS_E_NULL;
    if (B <= 0 || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0) return IDEAS_E_SHAPE;
    const int64_t n = (int64_t)Cout * Cin;
    hipLaunchKernelGGL(demod_wgrad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gw, w, gq, s, B, Cout,
                       Cin, KH, KW, so, si, sky, skx, go, gi, gky, gkx, coef);
    return ideas_launch_status();
}

extern "C" int ideas_pixel_dot(double* out, const void* a, const void* g, int B, int64_t P, int C, int dtype,
                               void* stream) {
    if (dtype != IDEAS_F32 && dtype != IDEAS_BF16) return IDEAS_E_UNSUPPORTED;
    if (!out || !a || !g) return IDEAS_E_NULL;
    if (B <= 0 || P <= 0 || C <= 0 || C > 6144 || B > 65535) return IDEAS_E_SHAPE;
    if ((C & 3) == 0 && (!ideas_aligned16(a) || !ideas_aligned16(g))) return IDEAS_E_ALIGN;
    if (dtype == IDEAS_BF16 && (C & 3)) return IDEAS_E_ALIGN;
    int64_t chunks = ideas_cdiv(2048, B);
    const int64_t max_chunks = ideas_cdiv(P, 64);
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks < 1) chunks = 1;
    const int64_t per = ideas_cdiv(P, chunks);
    chunks = ideas_cdiv(P, per);
    if (dtype == IDEAS_BF16)
        hipLaunchKernelGGL((pixel_dot_kernel<ideas_bf16, ideas_bf16x4>), dim3((unsigned)chunks, (unsigned)B), dim3(256),
                           (size_t)C * sizeof(double), (hipStream_t)stream, out, (const ideas_bf16*)a, (const ideas_bf16*)g, P, C, per);
    else
        hipLaunchKernelGGL((pixel_dot_kernel<float, float4>), dim3((unsigned)chunks, (unsigned)B), dim3(256),
                           (size_t)C * sizeof(double), (hipStream_t)stream, out, (const float*)a, (const float*)g, P, C, per);
    return ideas_launch_status();
}

extern "C" int ideas_act_bwd_dot(void* gpre, float* bias_grad, double* dot, const void* gy, const void* out,
                                 const float* bias, const float* gpre_scale, int B, int64_t P, int C, float alpha,
                                 float act_gain, int dtype, void* stream) {
    if (dtype != IDEAS_F32 && dtype != IDEAS_BF16) return IDEAS_E_UNSUPPORTED;
    if (!gpre || !bias_grad || !dot || !gy || !out) return IDEAS_E_NULL;
    if (B <= 0 || P <= 0 || C <= 0 || C > 3072 || B > 65535) return IDEAS_E_SHAPE;
    if (C & 3) return IDEAS_E_ALIGN;
    if (!ideas_aligned16(gpre) || !ideas_aligned16(gy) || !ideas_aligned16(out) || (bias && !ideas_aligned16(bias)))
        return IDEAS_E_ALIGN;
    int64_t chunks = ideas_cdiv(2048, B);
    const int64_t max_chunks = ideas_cdiv(P, 64);
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks < 1) chunks = 1;
    const int64_t per = ideas_cdiv(P, chunks);
    chunks = ideas_cdiv(P, per);
    if (dtype == IDEAS_BF16)
        hipLaunchKernelGGL((act_bwd_dot_kernel<ideas_bf16, ideas_bf16x4>), dim3((unsigned)chunks, (unsigned)B), dim3(256),
                           (size_t)2 * C * sizeof(double), (hipStream_t)stream, (ideas_bf16*)gpre, bias_grad, dot,
                           (const ideas_bf16*)gy, (const ideas_bf16*)out, bias, gpre_scale, P, C, per, alpha, act_gain);
    else
        hipLaunchKernelGGL((act_bwd_dot_kernel<float, float4>), dim3((unsigned)chunks, (unsigned)B), dim3(256),
                           (size_t)2 * C * sizeof(double), (hipStream_t)stream, (float*)gpre, bias_grad, dot, (const float*)gy,
                           (const float*)out, bias, gpre_scale, P, C, per, alpha, act_gain);
    return ideas_launch_status();
}
