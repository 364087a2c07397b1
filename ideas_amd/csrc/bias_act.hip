// fused bias + leaky-ReLU (+ fused bias-gradient reduction) for gfx950.
//
// Replaces fused_bias_act_kernel (stylegan2/op/fused_bias_act_kernel.cu:18-49).  HBM-bound: every element
// is read once and written once (8 B/elem forward, 12 B/elem backward in f32), so the kernel is all about
// 16-byte coalesced traffic.  NHWC keeps a thread's four lanes on four fixed channels for the whole
// grid-stride loop, which lets the bias gradient be reduced in registers -> LDS -> one atomic per channel per
// block instead of the reference's second full pass (fused_act.py:33-38).
#include "common.hpp"

namespace {

struct ActArgs {
    float alpha, scale;
    int act, grad;
};

__device__ __forceinline__ float act_one(float v, float r, const ActArgs& a) {
    // same operation order as the reference: add (done by caller), select-multiply, multiply
    float y;
    if (a.grad == 2) {
        y = 0.0f;
    } else if (a.act == 3) {
        float sel = (a.grad == 0) ? v : r;
        y = (sel > 0.0f) ? v : v * a.alpha;
    } else {
        y = v;
    }
    return y * a.scale;
}

// ---- NHWC / [B,C]: channel = i % C, vectorised by 4 along C ------------------------------------
template <typename V, bool HAS_B, bool HAS_REF, bool BGRAD, bool TILED>
__global__ __launch_bounds__(256) void bias_act_nhwc_v4(V* __restrict__ y, const V* __restrict__ x,
                                                        const float* __restrict__ b, const V* __restrict__ ref,
                                                        float* __restrict__ bgrad, int64_t n4, int C, ActArgs a) {
    extern __shared__ float s_bg[];
    constexpr int U = 4;
    // TILED (1024 % C == 0): a block's U loads of one trip are ADJACENT 4 KB rows (one contiguous 16 KB tile), and the
    // resident blocks together sweep one contiguous window of the tensor -- the far-strided variant below keeps 4 (8 with
    // ref, 12 with the stores) streams 16 MB apart in flight and ran at 4.3-4.9 TB/s against 6.2 for torch's elementwise
    // kernels.  The channels a thread owns are the same in every row because a row (1024 floats) is a multiple of C.
    const int64_t tid = TILED ? (int64_t)blockIdx.x * (blockDim.x * U) + threadIdx.x : (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = TILED ? blockDim.x : (int64_t)gridDim.x * blockDim.x;  // host guarantees (4*stride) % C == 0 when BGRAD
    const int64_t trip = TILED ? (int64_t)gridDim.x * blockDim.x * U : U * stride;
    const int c0 = (int)((tid * 4) % C);
    float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (HAS_B && tid < n4) bb = *reinterpret_cast<const float4*>(b + c0);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (BGRAD) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) s_bg[c] = 0.f;
        __syncthreads();
    }
    // 4 independent 16-byte loads (8 with ref) in flight per thread per trip: one load per trip leaves the kernel
    // latency-bound at ~4.6 TB/s; the channel a thread owns is unchanged because every offset is a multiple of `stride`
    for (int64_t i0 = tid; i0 < n4; i0 += trip) {
        float4 v[U], r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            if (i < n4) {
                v[u] = to_f4(x[i]);
                if (HAS_REF) r[u] = to_f4(ref[i]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            if (i >= n4) break;
            float4 vv = v[u];
            float4 rr = HAS_REF ? r[u] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (HAS_B) { vv.x += bb.x; vv.y += bb.y; vv.z += bb.z; vv.w += bb.w; }
            float4 o;
            o.x = act_one(vv.x, rr.x, a); o.y = act_one(vv.y, rr.y, a);
            o.z = act_one(vv.z, rr.z, a); o.w = act_one(vv.w, rr.w, a);
            y[i] = from_f4<V>(o);
            if (BGRAD) { acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
        }
    }
    if (BGRAD) {
        if (tid < n4 || TILED) {
            atomicAdd(&s_bg[c0 + 0], acc.x); atomicAdd(&s_bg[c0 + 1], acc.y);
            atomicAdd(&s_bg[c0 + 2], acc.z); atomicAdd(&s_bg[c0 + 3], acc.w);
        }
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            float v = s_bg[c];
            if (v != 0.f) atomicAdd(&bgrad[c], v);
        }
    }
}

// scalar NHWC fallback (C % 4 != 0 or unaligned): one element per thread-iteration
template <typename T, bool BGRAD>
__global__ __launch_bounds__(256) void bias_act_nhwc_s(T* __restrict__ y, const T* __restrict__ x,
                                                       const float* __restrict__ b, const T* __restrict__ ref,
                                                       float* __restrict__ bgrad, int64_t n, int C, ActArgs a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int c = (int)(i % C);
        float v = ld1(x + i);
        if (b) v += b[c];
        float o = act_one(v, ref ? ld1(ref + i) : 0.f, a);
        st1(y + i, o);
        if (BGRAD) atomicAdd(&bgrad[c], o);
    }
}

// Per-channel sum of a channels-innermost tensor for ANY C (the gradient of a plain conv bias: G.to_rgb, C = 3, models.py:120,
// stylegan2/model.py:94-123).  torch's sum over (0, 2, 3) of a channels_last [32, 3, 256, 256] tensor ran as one block: 1.0 ms,
// three times per iteration.  The grid stride is a multiple of C, so a thread stays on one channel; block partials meet in LDS.
template <typename T>
__global__ __launch_bounds__(256) void channel_sum_nhwc_kernel(float* __restrict__ out, const T* __restrict__ x, int64_t n, int C) {
    extern __shared__ float s_ch[];
    for (int c = threadIdx.x; c < C; c += 256) s_ch[c] = 0.f;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    for (int64_t i = i0; i < n; i += stride) acc += ld1(x + i);
    atomicAdd(&s_ch[(int)(i0 % C)], acc);
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) atomicAdd(&out[c], s_ch[c]);
}

// ---- NCHW: one block works inside ONE (b,c) plane, so the channel is block-uniform ---------------
template <bool BGRAD>
__global__ __launch_bounds__(256) void bias_act_nchw(float* __restrict__ y, const float* __restrict__ x,
                                                     const float* __restrict__ b, const float* __restrict__ ref,
                                                     float* __restrict__ bgrad, int64_t inner, int C, int chunks,
                                                     int vec_ok, ActArgs a) {
    __shared__ float s_part[4];
    const int64_t plane = blockIdx.x / chunks;
    const int chunk = blockIdx.x % chunks;
    const int c = (int)(plane % C);
    const float bv = b ? b[c] : 0.f;
    const int64_t base = plane * inner;
    const int64_t per = (inner + chunks - 1) / chunks;
    const int64_t lo = chunk * per;
    const int64_t hi = (lo + per < inner) ? lo + per : inner;
    float acc = 0.f;
    const bool vec = vec_ok && ((inner & 3) == 0) && ((per & 3) == 0);
    if (vec) {
        const float4* x4 = reinterpret_cast<const float4*>(x + base);
        const float4* r4 = ref ? reinterpret_cast<const float4*>(ref + base) : nullptr;
        float4* y4 = reinterpret_cast<float4*>(y + base);
        for (int64_t i = lo / 4 + threadIdx.x; i < hi / 4; i += blockDim.x) {
            float4 v = x4[i];
            float4 r = r4 ? r4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 o;
            o.x = act_one(v.x + bv, r.x, a); o.y = act_one(v.y + bv, r.y, a);
            o.z = act_one(v.z + bv, r.z, a); o.w = act_one(v.w + bv, r.w, a);
            y4[i] = o;
            if (BGRAD) acc += (o.x + o.y) + (o.z + o.w);
        }
    } else {
        for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
            float o = act_one(x[base + i] + bv, ref ? ref[base + i] : 0.f, a);
            y[base + i] = o;
            if (BGRAD) acc += o;
        }
    }
    if (BGRAD) {
        acc = wave_sum(acc);
        if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.f;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += s_part[w];
            atomicAdd(&bgrad[c], t);
        }
    }
}

int gcd_i(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

}  // namespace

extern "C" int ideas_fused_bias_act(void* y, const void* x, const void* b, const void* ref, float* bias_grad,
                                    int64_t n, int C, int64_t inner, int layout, int act, int grad, float alpha,
                                    float scale, int dtype, void* stream_) {
    if (dtype != IDEAS_F32 && dtype != IDEAS_BF16) return IDEAS_E_UNSUPPORTED;
    if (dtype == IDEAS_BF16 && layout != IDEAS_NHWC && inner != 1) return IDEAS_E_UNSUPPORTED;   // bf16: NHWC / [B,C] only
    if (n == 0) return IDEAS_OK;
    if (!y || !x) return IDEAS_E_NULL;
    if (n < 0 || C <= 0 || inner <= 0) return IDEAS_E_SHAPE;
    if (act != 1 && act != 3) return IDEAS_E_UNSUPPORTED;
    if (grad < 0 || grad > 2) return IDEAS_E_UNSUPPORTED;
    if (grad == 1 && act == 3 && !ref) return IDEAS_E_NULL;
    if (bias_grad && grad != 1) return IDEAS_E_UNSUPPORTED;
    hipStream_t stream = (hipStream_t)stream_;
    ActArgs a{alpha, scale, act, grad};
    const float* xf = (const float*)x;
    const float* bf = (const float*)b;
    const float* rf = (grad == 1) ? (const float*)ref : nullptr;
    float* yf = (float*)y;

    if (layout == IDEAS_NHWC || inner == 1) {
        if (n % C != 0) return IDEAS_E_SHAPE;
        const bool vec = (C % 4 == 0) && ideas_aligned16(x) && ideas_aligned16(y) && (!rf || ideas_aligned16(ref)) &&
                         (!bf || ideas_aligned16(bf)) && C <= 8192;
        if (vec) {
            const int64_t n4 = n / 4;
            const int c4 = C / 4;
            int64_t grid = ideas_cdiv(n4, 256);
            if (grid > 4096) grid = 4096;
            // keep each thread on fixed channels: 256*grid must be a multiple of C/4
            const int m = c4 / gcd_i(c4, 256);
            grid = ideas_cdiv(grid, m) * m;
            const size_t lds = bias_grad ? (size_t)C * sizeof(float) : 0;
            const bool tiled = (1024 % C == 0);
            if (tiled) {
                grid = ideas_cdiv(n4, 1024);
                if (grid > 4096) grid = 4096;
            }
#define LAUNCH_V4T(V, HB, HR, BG, TL)                                                                                    \
    hipLaunchKernelGGL((bias_act_nhwc_v4<V, HB, HR, BG, TL>), dim3((unsigned)grid), dim3(256), lds, stream, (V*)y,      \
                       (const V*)x, bf, (const V*)((grad == 1) ? ref : nullptr), bias_grad, n4, C, a)
#define LAUNCH_V4(HB, HR, BG)                                                                                            \
    do {                                                                                                                 \
        if (dtype == IDEAS_BF16) { if (tiled) LAUNCH_V4T(ideas_bf16x4, HB, HR, BG, true); else LAUNCH_V4T(ideas_bf16x4, HB, HR, BG, false); } \
        else { if (tiled) LAUNCH_V4T(float4, HB, HR, BG, true); else LAUNCH_V4T(float4, HB, HR, BG, false); }           \
    } while (0)
            if (bias_grad) {
                if (bf && rf) LAUNCH_V4(true, true, true);
                else if (bf) LAUNCH_V4(true, false, true);
                else if (rf) LAUNCH_V4(false, true, true);
                else LAUNCH_V4(false, false, true);
            }
            else if (bf && rf) LAUNCH_V4(true, true, false);
            else if (bf) LAUNCH_V4(true, false, false);
            else if (rf) LAUNCH_V4(false, true, false);
            else LAUNCH_V4(false, false, false);
#undef LAUNCH_V4
#undef LAUNCH_V4T
        } else {
            int64_t grid = ideas_cdiv(n, 256);
            if (grid > 8192) grid = 8192;
#define LAUNCH_S(T, BG)                                                                                                  \
    hipLaunchKernelGGL((bias_act_nhwc_s<T, BG>), dim3((unsigned)grid), dim3(256), 0, stream, (T*)y, (const T*)x, bf,    \
                       (const T*)((grad == 1) ? ref : nullptr), bias_grad, n, C, a)
            if (dtype == IDEAS_BF16) { if (bias_grad) LAUNCH_S(ideas_bf16, true); else LAUNCH_S(ideas_bf16, false); }
            else { if (bias_grad) LAUNCH_S(float, true); else LAUNCH_S(float, false); }
#undef LAUNCH_S
        }
        return ideas_launch_status();
    }
    if (layout != IDEAS_NCHW) return IDEAS_E_UNSUPPORTED;
    if (n % inner != 0) return IDEAS_E_SHAPE;
    const int64_t planes = n / inner;
    int chunks = (int)ideas_cdiv(inner, 4096);
    chunks = chunks < 1 ? 1 : chunks;
    // round the per-chunk span to a multiple of 4 so the float4 path stays aligned
    while (chunks > 1 && (ideas_cdiv(inner, chunks) & 3)) --chunks;
    const int64_t grid = planes * chunks;
    if (grid > 0x7fffffffLL) return IDEAS_E_SHAPE;
    const int vec_ok = ideas_aligned16(x) && ideas_aligned16(y) && (!rf || ideas_aligned16(rf));
    if (bias_grad)
        hipLaunchKernelGGL((bias_act_nchw<true>), dim3((unsigned)grid), dim3(256), 0, stream, yf, xf, bf, rf, bias_grad,
                           inner, C, chunks, vec_ok, a);
    else
        hipLaunchKernelGGL((bias_act_nchw<false>), dim3((unsigned)grid), dim3(256), 0, stream, yf, xf, bf, rf,
                           bias_grad, inner, C, chunks, vec_ok, a);
    return ideas_launch_status();
}

extern "C" int ideas_channel_sum(float* out, const void* x, int64_t n, int C, int clear, int dtype, void* stream_) {
    if (!out || !x) return IDEAS_E_NULL;
    if (n <= 0 || C <= 0 || C > 8192 || n % C != 0) return IDEAS_E_SHAPE;
    if (dtype != IDEAS_F32 && dtype != IDEAS_BF16) return IDEAS_E_UNSUPPORTED;
    hipStream_t stream = (hipStream_t)stream_;
    if (clear) {
        const hipError_t e = hipMemsetAsync(out, 0, (size_t)C * sizeof(float), stream);
        if (e != hipSuccess) return (int)e;
    }
    int64_t grid = ideas_cdiv(n, 256 * 16);                  // ~16 elements per thread
    if (grid > 2048) grid = 2048;
    const int m = C / gcd_i(C, 256);                         // 256 * grid must be a multiple of C
    grid = ideas_cdiv(grid, m) * m;
    const size_t lds = (size_t)C * sizeof(float);
    if (dtype == IDEAS_BF16)
        hipLaunchKernelGGL(channel_sum_nhwc_kernel<ideas_bf16>, dim3((unsigned)grid), dim3(256), lds, stream, out, (const ideas_bf16*)x, n, C);
    else
        hipLaunchKernelGGL(channel_sum_nhwc_kernel<float>, dim3((unsigned)grid), dim3(256), lds, stream, out, (const float*)x, n, C);
    return ideas_launch_status();
}
