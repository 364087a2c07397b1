// upfirdn2d for gfx950: zero-stuff / pad / FIR (flipped) / decimate.
//
// Replaces upfirdn2d_kernel + upfirdn2d_kernel_large (stylegan2/op/upfirdn2d_kernel.cu:49-207).  On the IDEAS
// path this op is only ever a 4x4 blur with up = down = 1, pure HBM traffic: read the plane
// once, write it once.  Three kernels:
//   * blur4_nhwc  — the fast path.  Channels are innermost, so a thread owns 4 channels of one output
//                   column and marches down the rows with a 4x4 register window: 4 x 16-byte loads and one
//                   16-byte store per output, every access coalesced along C, no LDS needed.
//   * blur_nchw_tile — NCHW (the reference's layout): LDS-staged (TH+kh-1) x (TW+kw-1) input tile per plane.
//   * generic     — any up/down/kernel <= 8x8, either layout, one output per thread (the op's full contract).
#include "common.hpp"

namespace {

struct FirParams {
    int B, C, in_h, in_w, out_h, out_w, kh, kw;
    int up_x, up_y, down_x, down_y, pad_x0, pad_y0;
    float gain;
    int flip;
    int seg_rows;   // blur4_nhwc: output rows marched per thread
};

// ---------------- generic: out[oy,ox] = sum_k U[oy*down + ky - pad0] * Kf[ky] ----------------------
template <bool NHWC, typename T>
__global__ __launch_bounds__(256) void upfirdn2d_generic(T* __restrict__ y, const T* __restrict__ x,
                                                         const float* __restrict__ fir, FirParams p) {
    __shared__ float sk[64];
    for (int t = threadIdx.x; t < p.kh * p.kw; t += blockDim.x) {
        int ky = t / p.kw, kx = t % p.kw;
        int src = p.flip ? (p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx) : t;
        sk[t] = fir[src] * p.gain;
    }
    __syncthreads();
    const int64_t total = (int64_t)p.B * p.C * p.out_h * p.out_w;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int c, ox, oy, b;
        int64_t r = i;
        if (NHWC) {
            c = (int)(r % p.C); r /= p.C;
            ox = (int)(r % p.out_w); r /= p.out_w;
            oy = (int)(r % p.out_h); b = (int)(r / p.out_h);
        } else {
            ox = (int)(r % p.out_w); r /= p.out_w;
            oy = (int)(r % p.out_h); r /= p.out_h;
            c = (int)(r % p.C); b = (int)(r / p.C);
        }
        float acc = 0.f;
        for (int ky = 0; ky < p.kh; ++ky) {
            int uy = oy * p.down_y + ky - p.pad_y0;
            if (uy < 0 || uy % p.up_y) continue;
            int iy = uy / p.up_y;
            if (iy >= p.in_h) continue;
            for (int kx = 0; kx < p.kw; ++kx) {
                int ux = ox * p.down_x + kx - p.pad_x0;
                if (ux < 0 || ux % p.up_x) continue;
                int ix = ux / p.up_x;
                if (ix >= p.in_w) continue;
                int64_t src = NHWC ? (((int64_t)b * p.in_h + iy) * p.in_w + ix) * p.C + c
                                   : (((int64_t)b * p.C + c) * p.in_h + iy) * p.in_w + ix;
                acc += ld1(x + src) * sk[ky * p.kw + kx];
            }
        }
        st1(y + i, acc);
    }
}

// ---------------- NHWC 4x4 blur, up = down = 1 ---------------------------------------------------
// thread = (b, row-segment, ox, c4); window w[r][t] holds input rows iy0..iy0+3 at columns ix0..ix0+3
// EPI: a fused elementwise stage on the blurred value before it is stored (ideas_blur_fused):
//   EPI_ACT_BWD   y = (ref > 0 ? v : v * alpha) * scale, bias_grad[c] += sum y    -- blur^T followed by the leaky-ReLU backward
//                                                                                   of the layer below (ref = its saved output)
//   EPI_BIAS_ACT  y = lrelu(v + bias[c], alpha) * scale                           -- the blur of an upsampling conv + FusedLeakyReLU
// Same operation order as bias_act.hip's act_one, so f32 results are bitwise those of blur -> fused_bias_act.
enum { EPI_NONE = 0, EPI_ACT_BWD = 1, EPI_BIAS_ACT = 2 };
struct FirEpi {
    const void* ref;
    const float* bias;
    float* bgrad;
    float alpha, scale;
};
__device__ __forceinline__ float epi_act(float v, float sel, float alpha, float scale) { return (sel > 0.f ? v : v * alpha) * scale; }

// per-thread partial bias gradients -> LDS (one slot per channel) -> one global atomic per channel and block
__device__ __forceinline__ void bgrad_flush(float* s_bg, float* __restrict__ bgrad, int C, bool active, int c0, const float* part, int n) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) s_bg[c] = 0.f;
    __syncthreads();
    if (active)
        for (int e = 0; e < n; ++e) atomicAdd(&s_bg[c0 + e], part[e]);
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float v = s_bg[c];
        if (v != 0.f) atomicAdd(&bgrad[c], v);
    }
}

template <typename V, int EPI>
__global__ __launch_bounds__(256) void blur4_nhwc(V* __restrict__ y, const V* __restrict__ x,
                                                  const float* __restrict__ fir, FirParams p, FirEpi ep) {
    __shared__ float sk[16];
    extern __shared__ float s_bg[];
    if (threadIdx.x < 16) {
        int t = threadIdx.x;
        int src = p.flip ? 15 - t : t;
        sk[t] = fir[src] * p.gain;
    }
    __syncthreads();
    const int C4 = p.C >> 2;
    const int segs = (p.out_h + p.seg_rows - 1) / p.seg_rows;
    const int64_t total = (int64_t)p.B * segs * p.out_w * C4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i < total;
    if (EPI != EPI_ACT_BWD && !active) return;
    // EPI_ACT_BWD ends in block barriers (bgrad_flush), which every lane of every wave has to reach along the SAME path: threads
    // past the end redo the last thread's work with their stores and their share of the bias gradient masked off
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    int64_t r = active ? i : total - 1;
    const int c4 = (int)(r % C4); r /= C4;
    const int ox = (int)(r % p.out_w); r /= p.out_w;
    const int seg = (int)(r % segs);
    const int b = (int)(r / segs);
    const int oy0 = seg * p.seg_rows;
    const int oy1 = (oy0 + p.seg_rows < p.out_h) ? oy0 + p.seg_rows : p.out_h;
    const int ix0 = ox - p.pad_x0;
    float k[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) k[t] = sk[t];

    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const V* xb = x + (int64_t)b * p.in_h * p.in_w * C4 + c4;
    unsigned colmask = 0;   // bit t: column ix0+t is inside the image (a bool[4] ends up in scratch memory)
#pragma unroll
    for (int t = 0; t < 4; ++t) colmask |= ((ix0 + t >= 0) && (ix0 + t < p.in_w)) ? (1u << t) : 0u;

    float4 w[4][4];
    auto load_row = [&](int iy, float4 (&dst)[4]) {
        const bool rowok = (iy >= 0) && (iy < p.in_h);
        const V* xr = xb + ((int64_t)iy * p.in_w + ix0) * C4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float4 v = zero;
            if (rowok && ((colmask >> t) & 1u)) v = to_f4(xr[(int64_t)t * C4]);
            dst[t] = v;
        }
    };
    const int iy0 = oy0 - p.pad_y0;
    load_row(iy0 + 0, w[0]);
    load_row(iy0 + 1, w[1]);
    load_row(iy0 + 2, w[2]);
    V* yb = y + (((int64_t)b * p.out_h) * p.out_w + ox) * C4 + c4;
    const V* rb = reinterpret_cast<const V*>(ep.ref) + (((int64_t)b * p.out_h) * p.out_w + ox) * C4 + c4;
    float4 bb = zero;
    if (EPI == EPI_BIAS_ACT) bb = *reinterpret_cast<const float4*>(ep.bias + 4 * c4);
    auto emit = [&](int oy, const float4 (&r0)[4], const float4 (&r1)[4], const float4 (&r2)[4], const float4 (&r3)[4]) {
        float4 acc = zero;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float k0 = k[t], k1 = k[4 + t], k2 = k[8 + t], k3 = k[12 + t];
            acc.x += r0[t].x * k0 + r1[t].x * k1 + r2[t].x * k2 + r3[t].x * k3;
            acc.y += r0[t].y * k0 + r1[t].y * k1 + r2[t].y * k2 + r3[t].y * k3;
            acc.z += r0[t].z * k0 + r1[t].z * k1 + r2[t].z * k2 + r3[t].z * k3;
            acc.w += r0[t].w * k0 + r1[t].w * k1 + r2[t].w * k2 + r3[t].w * k3;
        }
        if (EPI == EPI_ACT_BWD) {
            const float4 rf = to_f4(rb[(int64_t)oy * p.out_w * C4]);
            if (sizeof(V) != sizeof(float4)) acc = to_f4(from_f4<V>(acc));       // bf16: the unfused path stores the blur first
            acc = make_float4(epi_act(acc.x, rf.x, ep.alpha, ep.scale), epi_act(acc.y, rf.y, ep.alpha, ep.scale),
                              epi_act(acc.z, rf.z, ep.alpha, ep.scale), epi_act(acc.w, rf.w, ep.alpha, ep.scale));
            bsum.x += acc.x; bsum.y += acc.y; bsum.z += acc.z; bsum.w += acc.w;
            if (!active) return;
        }
        if (EPI == EPI_BIAS_ACT) {
            if (sizeof(V) != sizeof(float4)) acc = to_f4(from_f4<V>(acc));
            const float4 v = make_float4(acc.x + bb.x, acc.y + bb.y, acc.z + bb.z, acc.w + bb.w);
            acc = make_float4(epi_act(v.x, v.x, ep.alpha, ep.scale), epi_act(v.y, v.y, ep.alpha, ep.scale),
                              epi_act(v.z, v.z, ep.alpha, ep.scale), epi_act(v.w, v.w, ep.alpha, ep.scale));
        }
        yb[(int64_t)oy * p.out_w * C4] = from_f4<V>(acc);
    };
    // two output rows per trip: 8 independent 16-byte loads in flight instead of 4 (the window shift is a
    // dependent chain, so one row per trip is latency-bound)
    float4 w4[4];
    int oy = oy0;
    for (; oy + 1 < oy1; oy += 2) {
        load_row(oy - p.pad_y0 + 3, w[3]);
        load_row(oy - p.pad_y0 + 4, w4);
        emit(oy, w[0], w[1], w[2], w[3]);
        emit(oy + 1, w[1], w[2], w[3], w4);
#pragma unroll
        for (int t = 0; t < 4; ++t) { w[0][t] = w[2][t]; w[1][t] = w[3][t]; w[2][t] = w4[t]; }
    }
    if (oy < oy1) {
        load_row(oy - p.pad_y0 + 3, w[3]);
        emit(oy, w[0], w[1], w[2], w[3]);
    }
    if (EPI == EPI_ACT_BWD) {
        const float part[4] = {bsum.x, bsum.y, bsum.z, bsum.w};
        bgrad_flush(s_bg, ep.bgrad, p.C, active, 4 * c4, part, 4);
    }
}

// ---------------- NHWC 4x4 FIR with decimation by 2 (up = 1, down = 2) ----------------------------------------------------
// The blur in front of a 1x1 stride-2 skip conv only feeds every second pixel of every second row into the conv
// (models.py:78-95): computing it at the OUTPUT resolution writes N/4 instead of N and lets the conv run unstrided.
// Same marching scheme as blur4_nhwc, the window advances two input rows per output row (8 loads, 1 store per output).
template <typename V>
__global__ __launch_bounds__(256) void fir4_down2_nhwc(V* __restrict__ y, const V* __restrict__ x, const float* __restrict__ fir,
                                                       FirParams p) {
    __shared__ float sk[16];
    if (threadIdx.x < 16) {
        const int t = threadIdx.x;
        sk[t] = fir[p.flip ? 15 - t : t] * p.gain;
    }
    __syncthreads();
    const int C4 = p.C >> 2;
    const int segs = (p.out_h + p.seg_rows - 1) / p.seg_rows;
    const int64_t total = (int64_t)p.B * segs * p.out_w * C4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int64_t r = i;
    const int c4 = (int)(r % C4); r /= C4;
    const int ox = (int)(r % p.out_w); r /= p.out_w;
    const int seg = (int)(r % segs);
    const int b = (int)(r / segs);
    const int oy0 = seg * p.seg_rows;
    const int oy1 = (oy0 + p.seg_rows < p.out_h) ? oy0 + p.seg_rows : p.out_h;
    const int ix0 = 2 * ox - p.pad_x0;
    float k[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) k[t] = sk[t];
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const V* xb = x + (int64_t)b * p.in_h * p.in_w * C4 + c4;
    unsigned colmask = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) colmask |= ((ix0 + t >= 0) && (ix0 + t < p.in_w)) ? (1u << t) : 0u;
    auto load_row = [&](int iy, float4 (&dst)[4]) {
        const bool rowok = (iy >= 0) && (iy < p.in_h);
        const V* xr = xb + ((int64_t)iy * p.in_w + ix0) * C4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float4 v = zero;
            if (rowok && ((colmask >> t) & 1u)) v = to_f4(xr[(int64_t)t * C4]);
            dst[t] = v;
        }
    };
    float4 w[4][4];
    load_row(2 * oy0 - p.pad_y0 + 0, w[0]);
    load_row(2 * oy0 - p.pad_y0 + 1, w[1]);
    V* yb = y + (((int64_t)b * p.out_h) * p.out_w + ox) * C4 + c4;
    for (int oy = oy0; oy < oy1; ++oy) {
        load_row(2 * oy - p.pad_y0 + 2, w[2]);
        load_row(2 * oy - p.pad_y0 + 3, w[3]);
        float4 acc = zero;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float kk = k[4 * j + t];
                acc.x = fmaf(w[j][t].x, kk, acc.x); acc.y = fmaf(w[j][t].y, kk, acc.y);
                acc.z = fmaf(w[j][t].z, kk, acc.z); acc.w = fmaf(w[j][t].w, kk, acc.w);
            }
        yb[(int64_t)oy * p.out_w * C4] = from_f4<V>(acc);
#pragma unroll
        for (int t = 0; t < 4; ++t) { w[0][t] = w[2][t]; w[1][t] = w[3][t]; }
    }
}

// ---------------- NHWC 4x4 FIR after zero-stuffing by 2 (up = 2, down = 1) ---------------------------------------------
// The adjoint of the kernel above (its backward), and the blur behind a 1x1 stride-2 TRANSPOSED skip conv (three of four
// pixels of that conv's output are structural zeros; here they are never written or read).  Per axis, output o = 2i + a
// (a = 0, 1) sees the two taps k = k0(a), k0(a) + 2 with k0(a) = (pad0 - a) & 1, at inputs i + d(a), i + d(a) + 1,
// d(a) = (a + k0(a) - pad0) / 2 -- so a 2 x 2 output block needs a 3 x 3 input neighbourhood.  A thread owns the output
// column pair (2j, 2j + 1) of 4 channels and marches down block rows: 3 loads and 4 stores per block row.
// `resid` (optional, same shape as y): y = fir(x) + resid -- the residual merge behind an upsampling skip, and (as the adjoint of
// fir4_down2) the sum of the two input gradients of a downsampling ResBlock, without a separate add pass (ideas_fir_up2_add).
template <typename V>
__global__ __launch_bounds__(256) void fir4_up2_nhwc(V* __restrict__ y, const V* __restrict__ x, const float* __restrict__ fir,
                                                     FirParams p, const V* __restrict__ resid) {
    __shared__ float sk[16];
    if (threadIdx.x < 16) {
        const int t = threadIdx.x;
        sk[t] = fir[p.flip ? 15 - t : t] * p.gain;
    }
    __syncthreads();
    const int C4 = p.C >> 2;
    const int bh = (p.out_h + 1) >> 1, bw = (p.out_w + 1) >> 1;          // 2 x 2 output blocks
    const int segs = (bh + p.seg_rows - 1) / p.seg_rows;
    const int64_t total = (int64_t)p.B * segs * bw * C4;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int64_t r = idx;
    const int c4 = (int)(r % C4); r /= C4;
    const int j = (int)(r % bw); r /= bw;
    const int seg = (int)(r % segs);
    const int b = (int)(r / segs);
    const int i0 = seg * p.seg_rows;
    const int i1 = (i0 + p.seg_rows < bh) ? i0 + p.seg_rows : bh;
    // per-axis tap / offset tables (a = output parity)
    const int ky0[2] = {p.pad_y0 & 1, (p.pad_y0 - 1) & 1}, kx0[2] = {p.pad_x0 & 1, (p.pad_x0 - 1) & 1};
    const int dy[2] = {(ky0[0] - p.pad_y0) >> 1, (1 + ky0[1] - p.pad_y0) >> 1};
    const int dx[2] = {(kx0[0] - p.pad_x0) >> 1, (1 + kx0[1] - p.pad_x0) >> 1};
    const int ey = dy[1] - dy[0], ex = dx[1] - dx[0];                       // 0 or 1: which 2 of the 3 rows / columns parity 1 uses
    float wq[2][2][2][2];                                                  // [a][bb][u][v]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int v = 0; v < 2; ++v) wq[a][bb][u][v] = sk[(ky0[a] + 2 * u) * 4 + kx0[bb] + 2 * v];
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const V* xb = x + (int64_t)b * p.in_h * p.in_w * C4 + c4;
    const int cx0 = j + dx[0];
    unsigned colmask = 0;
#pragma unroll
    for (int t = 0; t < 3; ++t) colmask |= ((cx0 + t >= 0) && (cx0 + t < p.in_w)) ? (1u << t) : 0u;
    auto load_row = [&](int iy, float4 (&dst)[3]) {
        const bool rowok = (iy >= 0) && (iy < p.in_h);
        const V* xr = xb + ((int64_t)iy * p.in_w + cx0) * C4;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            float4 v = zero;
            if (rowok && ((colmask >> t) & 1u)) v = to_f4(xr[(int64_t)t * C4]);
            dst[t] = v;
        }
    };
    float4 R[3][3];
    load_row(i0 + dy[0] + 0, R[0]);
    load_row(i0 + dy[0] + 1, R[1]);
    const bool colB = 2 * j + 1 < p.out_w;
    for (int i = i0; i < i1; ++i) {
        load_row(i + dy[0] + 2, R[2]);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int oy = 2 * i + a;
            if (oy >= p.out_h) break;
            // rows of this parity: R[ra], R[ra + 1] with ra = a ? ey : 0 (selected without dynamic register indexing)
            float4 top[3], bot[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const bool hi = (a == 1) && (ey == 1);
                top[t] = hi ? R[1][t] : R[0][t];
                bot[t] = hi ? R[2][t] : R[1][t];
            }
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                if (bb == 1 && !colB) break;
                const bool hx = (bb == 1) && (ex == 1);
                const float4 t0 = hx ? top[1] : top[0], t1 = hx ? top[2] : top[1];
                const float4 b0 = hx ? bot[1] : bot[0], b1 = hx ? bot[2] : bot[1];
                const float w00 = wq[a][bb][0][0], w01 = wq[a][bb][0][1], w10 = wq[a][bb][1][0], w11 = wq[a][bb][1][1];
                float4 acc;
                acc.x = t0.x * w00 + t1.x * w01 + b0.x * w10 + b1.x * w11;
                acc.y = t0.y * w00 + t1.y * w01 + b0.y * w10 + b1.y * w11;
                acc.z = t0.z * w00 + t1.z * w01 + b0.z * w10 + b1.z * w11;
                acc.w = t0.w * w00 + t1.w * w01 + b0.w * w10 + b1.w * w11;
                const int64_t yi = (((int64_t)b * p.out_h + oy) * p.out_w + 2 * j + bb) * C4 + c4;
                if (resid) {
                    if (sizeof(V) != sizeof(float4)) acc = to_f4(from_f4<V>(acc));     // bf16: the two-kernel chain stores the FIR first
                    const float4 rv = to_f4(resid[yi]);
                    acc = make_float4(acc.x + rv.x, acc.y + rv.y, acc.z + rv.z, acc.w + rv.w);
                }
                y[yi] = from_f4<V>(acc);
            }
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) { R[0][t] = R[1][t]; R[1][t] = R[2][t]; }
    }
}

// ---------------- f32 NHWC 4x4 blur, two output columns per thread (separable FIR) ---------------------------------------------
// blur4_nhwc issues four 16-byte loads per 16-byte output and is bound by load instructions, not by HBM (4.5-5.0 TB/s against the
// 5.5-6 of a plain streaming kernel).  A thread that owns the column pair (2q, 2q+1) loads five vectors per row for two outputs and
// -- every FIR of the path being an outer product kv (x) kh (make_kernel, stylegan2/model.py:22-30) -- keeps two horizontally
// filtered values per row instead of a 4 x 5 window.  The factorisation is checked on the device; a rank > 1 table takes a direct
// loop.  (Summation order differs from blur4_nhwc in the last bit; fused and unfused stages share this kernel, so they still agree
// bitwise with each other.)
template <int EPI>
__global__ __launch_bounds__(256) void blur4_f32_c2(float4* __restrict__ y, const float4* __restrict__ x,
                                                    const float* __restrict__ fir, FirParams p, FirEpi ep) {
    __shared__ float sk[16];
    extern __shared__ float s_bg[];
    if (threadIdx.x < 16) {
        const int t = threadIdx.x;
        sk[t] = fir[p.flip ? 15 - t : t] * p.gain;
    }
    __syncthreads();
    float kh[4], kv[4];
    float kmax = 0.f, res = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) { kh[t] = sk[t]; kv[t] = sk[0] != 0.f ? sk[4 * t] / sk[0] : 0.f; }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t) { kmax = fmaxf(kmax, fabsf(sk[4 * j + t])); res = fmaxf(res, fabsf(sk[4 * j + t] - kv[j] * kh[t])); }
    const bool sep = res <= 1e-6f * kmax;           // block-uniform
    const int C4 = p.C >> 2;
    const int pw = (p.out_w + 1) >> 1;
    const int segs = (p.out_h + p.seg_rows - 1) / p.seg_rows;
    const int64_t total = (int64_t)p.B * segs * pw * C4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i < total;
    if (EPI != EPI_ACT_BWD && !active) return;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    int64_t r = active ? i : total - 1;      // (see blur4_nhwc: one path to the barriers of bgrad_flush)
    const int c4 = (int)(r % C4); r /= C4;
    const int q = (int)(r % pw); r /= pw;
    const int seg = (int)(r % segs);
    const int b = (int)(r / segs);
    const int ox = 2 * q;
    const bool colB = ox + 1 < p.out_w;
    const int oy0 = seg * p.seg_rows;
    const int oy1 = (oy0 + p.seg_rows < p.out_h) ? oy0 + p.seg_rows : p.out_h;
    const int ix0 = ox - p.pad_x0;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* xb = x + (int64_t)b * p.in_h * p.in_w * C4 + c4;
    float4* yb = y + (((int64_t)b * p.out_h) * p.out_w + ox) * C4 + c4;
    const float4* rb = reinterpret_cast<const float4*>(ep.ref) + (((int64_t)b * p.out_h) * p.out_w + ox) * C4 + c4;
    float4 bb = zero;
    if (EPI == EPI_BIAS_ACT) bb = *reinterpret_cast<const float4*>(ep.bias + 4 * c4);
    auto finish = [&](int oy, int col, float4 acc) {
        const int64_t off = (int64_t)oy * p.out_w * C4 + (int64_t)col * C4;
        if (EPI == EPI_ACT_BWD) {
            const float4 rf = rb[off];
            acc = make_float4(epi_act(acc.x, rf.x, ep.alpha, ep.scale), epi_act(acc.y, rf.y, ep.alpha, ep.scale),
                              epi_act(acc.z, rf.z, ep.alpha, ep.scale), epi_act(acc.w, rf.w, ep.alpha, ep.scale));
            bsum.x += acc.x; bsum.y += acc.y; bsum.z += acc.z; bsum.w += acc.w;
            if (!active) return;
        }
        if (EPI == EPI_BIAS_ACT) {
            const float4 v = make_float4(acc.x + bb.x, acc.y + bb.y, acc.z + bb.z, acc.w + bb.w);
            acc = make_float4(epi_act(v.x, v.x, ep.alpha, ep.scale), epi_act(v.y, v.y, ep.alpha, ep.scale),
                              epi_act(v.z, v.z, ep.alpha, ep.scale), epi_act(v.w, v.w, ep.alpha, ep.scale));
        }
        yb[off] = acc;
    };
    unsigned colmask = 0;
#pragma unroll
    for (int t = 0; t < 5; ++t) colmask |= ((ix0 + t >= 0) && (ix0 + t < p.in_w)) ? (1u << t) : 0u;
    if (!sep) {      // direct 16-tap form, one output at a time: correct for any FIR, not tuned
#pragma unroll 1
        for (int oy = oy0; oy < oy1; ++oy)
#pragma unroll 1
            for (int col = 0; col < (colB ? 2 : 1); ++col) {
                float4 acc = zero;
#pragma unroll 1
                for (int j = 0; j < 4; ++j) {
                    const int iy = oy - p.pad_y0 + j;
                    if (iy < 0 || iy >= p.in_h) continue;
#pragma unroll 1
                    for (int t = 0; t < 4; ++t) {
                        const int ix = ix0 + col + t;
                        if (ix < 0 || ix >= p.in_w) continue;
                        const float4 v = xb[((int64_t)iy * p.in_w + ix) * C4];
                        const float kk = sk[4 * j + t];
                        acc.x = fmaf(v.x, kk, acc.x); acc.y = fmaf(v.y, kk, acc.y); acc.z = fmaf(v.z, kk, acc.z); acc.w = fmaf(v.w, kk, acc.w);
                    }
                }
                finish(oy, col, acc);
            }
        if (EPI == EPI_ACT_BWD) {
            const float part[4] = {bsum.x, bsum.y, bsum.z, bsum.w};
            bgrad_flush(s_bg, ep.bgrad, p.C, active, 4 * c4, part, 4);
        }
        return;
    }
    // one input row -> its two horizontally filtered values (columns ox and ox + 1); loads are clamped + masked by value
    // (component-wise: a ternary on the float4 aggregate becomes a select between ADDRESSES and spills to scratch)
    auto hrow = [&](int iy, float4& ha, float4& hb) {
        const bool rowok = (iy >= 0) && (iy < p.in_h);
        const float4* xr = xb + (int64_t)(rowok ? iy : 0) * p.in_w * C4;
        float4 v[5];
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const bool ok = rowok && ((colmask >> t) & 1u);
            const float4 w = xr[(int64_t)(ok ? ix0 + t : 0) * C4];
            v[t] = make_float4(ok ? w.x : 0.f, ok ? w.y : 0.f, ok ? w.z : 0.f, ok ? w.w : 0.f);
        }
        ha = zero; hb = zero;
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            if (t < 4) { ha.x = fmaf(v[t].x, kh[t], ha.x); ha.y = fmaf(v[t].y, kh[t], ha.y); ha.z = fmaf(v[t].z, kh[t], ha.z); ha.w = fmaf(v[t].w, kh[t], ha.w); }
            if (t > 0) { hb.x = fmaf(v[t].x, kh[t - 1], hb.x); hb.y = fmaf(v[t].y, kh[t - 1], hb.y); hb.z = fmaf(v[t].z, kh[t - 1], hb.z); hb.w = fmaf(v[t].w, kh[t - 1], hb.w); }
        }
    };
    auto vsum = [&](const float4& r0, const float4& r1, const float4& r2, const float4& r3) {
        return make_float4(r0.x * kv[0] + r1.x * kv[1] + r2.x * kv[2] + r3.x * kv[3], r0.y * kv[0] + r1.y * kv[1] + r2.y * kv[2] + r3.y * kv[3],
                           r0.z * kv[0] + r1.z * kv[1] + r2.z * kv[2] + r3.z * kv[3], r0.w * kv[0] + r1.w * kv[1] + r2.w * kv[2] + r3.w * kv[3]);
    };
    float4 a0, a1, a2, a3, a4, b0, b1, b2, b3, b4;
    {
        const int iy0 = oy0 - p.pad_y0;
        hrow(iy0 + 0, a0, b0); hrow(iy0 + 1, a1, b1); hrow(iy0 + 2, a2, b2);
    }
    int oy = oy0;
#pragma unroll 1
    for (; oy + 1 < oy1; oy += 2) {      // two output rows per trip: ten 16-byte loads in flight
        hrow(oy - p.pad_y0 + 3, a3, b3);
        hrow(oy - p.pad_y0 + 4, a4, b4);
        finish(oy, 0, vsum(a0, a1, a2, a3));
        if (colB) finish(oy, 1, vsum(b0, b1, b2, b3));
        finish(oy + 1, 0, vsum(a1, a2, a3, a4));
        if (colB) finish(oy + 1, 1, vsum(b1, b2, b3, b4));
        a0 = a2; a1 = a3; a2 = a4; b0 = b2; b1 = b3; b2 = b4;
    }
    if (oy < oy1) {
        hrow(oy - p.pad_y0 + 3, a3, b3);
        finish(oy, 0, vsum(a0, a1, a2, a3));
        if (colB) finish(oy, 1, vsum(b0, b1, b2, b3));
    }
    if (EPI == EPI_ACT_BWD) {
        const float part[4] = {bsum.x, bsum.y, bsum.z, bsum.w};
        bgrad_flush(s_bg, ep.bgrad, p.C, active, 4 * c4, part, 4);
    }
}

// ---------------- bf16 NHWC 4x4 blur, 8 channels (16 bytes) per thread -------------------------------------------------
// With 2-byte elements the 4-channel window kernel above moves half the bytes per instruction and is latency-bound
// (1.45 TB/s measured).  Here a thread owns 8 channels of one output column.  A 4x4 window of 8-channel inputs would be 128
// registers, so the kernel uses that every FIR on the path is an outer product kv (x) kh (make_kernel, stylegan2/model.py:22-30):
// each input row is reduced to ONE horizontally filtered row of 8 floats as it arrives, and the window is four of those.
// The factorisation is checked on the device from the 16 taps; a FIR that is not rank 1 takes the direct 16-tap loop.
struct F8 { float v[8]; };
__device__ __forceinline__ void unpack8(const uint4& u, float (&f)[8]) {
    f[0] = __builtin_bit_cast(float, u.x << 16); f[1] = __builtin_bit_cast(float, u.x & 0xffff0000u);
    f[2] = __builtin_bit_cast(float, u.y << 16); f[3] = __builtin_bit_cast(float, u.y & 0xffff0000u);
    f[4] = __builtin_bit_cast(float, u.z << 16); f[5] = __builtin_bit_cast(float, u.z & 0xffff0000u);
    f[6] = __builtin_bit_cast(float, u.w << 16); f[7] = __builtin_bit_cast(float, u.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    return make_uint4(ideas_pk_bf16(f[0], f[1]), ideas_pk_bf16(f[2], f[3]), ideas_pk_bf16(f[4], f[5]), ideas_pk_bf16(f[6], f[7]));
}

template <int EPI>
__global__ __launch_bounds__(256) void blur4_bf16x8(uint4* __restrict__ y, const uint4* __restrict__ x,
                                                    const float* __restrict__ fir, FirParams p, FirEpi ep) {
    __shared__ float sk[16];
    extern __shared__ float s_bg[];
    if (threadIdx.x < 16) {
        const int t = threadIdx.x;
        sk[t] = fir[p.flip ? 15 - t : t] * p.gain;
    }
    __syncthreads();
    float kh[4], kv[4];
    float kmax = 0.f, res = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) { kh[t] = sk[t]; kv[t] = sk[0] != 0.f ? sk[4 * t] / sk[0] : 0.f; }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t) { kmax = fmaxf(kmax, fabsf(sk[4 * j + t])); res = fmaxf(res, fabsf(sk[4 * j + t] - kv[j] * kh[t])); }
    const bool sep = res <= 1e-6f * kmax;           // block-uniform

    const int C8 = p.C >> 3;
    const int segs = (p.out_h + p.seg_rows - 1) / p.seg_rows;
    const int64_t total = (int64_t)p.B * segs * p.out_w * C8;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i < total;
    if (EPI != EPI_ACT_BWD && !active) return;
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int64_t r = active ? i : total - 1;      // (see blur4_nhwc: one path to the barriers of bgrad_flush)
    const int c8 = (int)(r % C8); r /= C8;
    const int ox = (int)(r % p.out_w); r /= p.out_w;
    const int seg = (int)(r % segs);
    const int b = (int)(r / segs);
    const int oy0 = seg * p.seg_rows;
    const int oy1 = (oy0 + p.seg_rows < p.out_h) ? oy0 + p.seg_rows : p.out_h;
    const int ix0 = ox - p.pad_x0;
    const uint4* xb = x + (int64_t)b * p.in_h * p.in_w * C8 + c8;
    uint4* yb = y + (((int64_t)b * p.out_h) * p.out_w + ox) * C8 + c8;
    const uint4* rb = reinterpret_cast<const uint4*>(ep.ref) + (((int64_t)b * p.out_h) * p.out_w + ox) * C8 + c8;
    float bb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (EPI == EPI_BIAS_ACT) {
#pragma unroll
        for (int e = 0; e < 8; ++e) bb[e] = ep.bias[8 * c8 + e];
    }
    // the fused stage of one output vector (the blur is rounded to bf16 first, as the unfused blur -> bias_act path stores it)
    auto finish = [&](int oy, float (&o)[8]) {
        if (EPI != EPI_NONE) {
            const uint4 q = pack8(o);
            unpack8(q, o);
        }
        if (EPI == EPI_ACT_BWD) {
            float rf[8];
            unpack8(rb[(int64_t)oy * p.out_w * C8], rf);
#pragma unroll
            for (int e = 0; e < 8; ++e) { o[e] = epi_act(o[e], rf[e], ep.alpha, ep.scale); bsum[e] += o[e]; }
            if (!active) return;
        }
        if (EPI == EPI_BIAS_ACT) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float v = o[e] + bb[e]; o[e] = epi_act(v, v, ep.alpha, ep.scale); }
        }
        yb[(int64_t)oy * p.out_w * C8] = pack8(o);
    };
    unsigned colmask = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) colmask |= ((ix0 + t >= 0) && (ix0 + t < p.in_w)) ? (1u << t) : 0u;
    // (component-wise masking: a ternary on the uint4 aggregate becomes a select between ADDRESSES and spills to scratch)
    auto load_row = [&](int iy, uint4 (&dst)[4]) {
        const bool rowok = (iy >= 0) && (iy < p.in_h);
        const int iyc = rowok ? iy : 0;
        const uint4* xr = xb + (int64_t)iyc * p.in_w * C8;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bool ok = rowok && ((colmask >> t) & 1u);
            const int ixc = ok ? ix0 + t : 0;                      // clamped: always a valid address
            const uint4 v = xr[(int64_t)ixc * C8];
            const unsigned m = ok ? 0xffffffffu : 0u;
            dst[t] = make_uint4(v.x & m, v.y & m, v.z & m, v.w & m);
        }
    };
    if (!sep) {      // direct 16-tap form (no window): correct for any FIR, not tuned
#pragma unroll 1
        for (int oy = oy0; oy < oy1; ++oy) {
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int j = 0; j < 4; ++j) {
                uint4 row[4];
                load_row(oy - p.pad_y0 + j, row);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float f[8];
                    unpack8(row[t], f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] = fmaf(f[e], sk[4 * j + t], acc[e]);
                }
            }
            finish(oy, acc);
        }
        if (EPI == EPI_ACT_BWD) bgrad_flush(s_bg, ep.bgrad, p.C, active, 8 * c8, bsum, 8);
        return;
    }
    auto hfilter = [&](const uint4 (&row)[4], F8& h) {
#pragma unroll
        for (int e = 0; e < 8; ++e) h.v[e] = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float f[8];
            unpack8(row[t], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) h.v[e] = fmaf(f[e], kh[t], h.v[e]);
        }
    };
    auto emit = [&](int oy, const F8& a0, const F8& a1, const F8& a2, const F8& a3) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = a0.v[e] * kv[0] + a1.v[e] * kv[1] + a2.v[e] * kv[2] + a3.v[e] * kv[3];
        finish(oy, o);
    };
    F8 h0, h1, h2, h3, h4;
    {
        uint4 r0[4], r1[4], r2[4];
        const int iy0 = oy0 - p.pad_y0;
        load_row(iy0 + 0, r0); load_row(iy0 + 1, r1); load_row(iy0 + 2, r2);
        hfilter(r0, h0); hfilter(r1, h1); hfilter(r2, h2);
    }
    int oy = oy0;
#pragma unroll 1
    for (; oy + 1 < oy1; oy += 2) {      // two output rows per trip: eight 16-byte loads in flight
        uint4 ra[4], rb[4];
        load_row(oy - p.pad_y0 + 3, ra);
        load_row(oy - p.pad_y0 + 4, rb);
        hfilter(ra, h3); hfilter(rb, h4);
        emit(oy, h0, h1, h2, h3);
        emit(oy + 1, h1, h2, h3, h4);
        h0 = h2; h1 = h3; h2 = h4;
    }
    if (oy < oy1) {
        uint4 ra[4];
        load_row(oy - p.pad_y0 + 3, ra);
        hfilter(ra, h3);
        emit(oy, h0, h1, h2, h3);
    }
    if (EPI == EPI_ACT_BWD) bgrad_flush(s_bg, ep.bgrad, p.C, active, 8 * c8, bsum, 8);
}

// ---------------- the same, TWO output columns per thread -----------------------------------------------------------------------
// blur4_bf16x8 issues four 16-byte loads per 16-byte output; a thread that owns the column pair (2q, 2q+1) needs five per row for
// two outputs (the windows overlap in three columns), i.e. 2.5 loads per output, and keeps two horizontally filtered values per
// row.  Only for separable FIRs (every FIR of the path; others stay on the one-column kernel).
template <int EPI>
__global__ __launch_bounds__(256) void blur4_bf16x8_c2(uint4* __restrict__ y, const uint4* __restrict__ x,
                                                       const float* __restrict__ fir, FirParams p, FirEpi ep) {
    __shared__ float sk[16];
    extern __shared__ float s_bg[];
    if (threadIdx.x < 16) {
        const int t = threadIdx.x;
        sk[t] = fir[p.flip ? 15 - t : t] * p.gain;
    }
    __syncthreads();
    float kh[4], kv[4];
    float kmax = 0.f, res = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) { kh[t] = sk[t]; kv[t] = sk[0] != 0.f ? sk[4 * t] / sk[0] : 0.f; }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t) { kmax = fmaxf(kmax, fabsf(sk[4 * j + t])); res = fmaxf(res, fabsf(sk[4 * j + t] - kv[j] * kh[t])); }
    const bool sep = res <= 1e-6f * kmax;           // block-uniform; a rank > 1 table takes the direct loop below
    const int C8 = p.C >> 3;
    const int pw = (p.out_w + 1) >> 1;                       // column pairs
    const int segs = (p.out_h + p.seg_rows - 1) / p.seg_rows;
    const int64_t total = (int64_t)p.B * segs * pw * C8;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i < total;
    if (EPI != EPI_ACT_BWD && !active) return;
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int64_t r = active ? i : total - 1;      // (see blur4_nhwc: one path to the barriers of bgrad_flush)
    const int c8 = (int)(r % C8); r /= C8;
    const int q = (int)(r % pw); r /= pw;
    const int seg = (int)(r % segs);
    const int b = (int)(r / segs);
    const int ox = 2 * q;
    const bool colB = ox + 1 < p.out_w;
    const int oy0 = seg * p.seg_rows;
    const int oy1 = (oy0 + p.seg_rows < p.out_h) ? oy0 + p.seg_rows : p.out_h;
    const int ix0 = ox - p.pad_x0;
    const uint4* xb = x + (int64_t)b * p.in_h * p.in_w * C8 + c8;
    uint4* yb = y + (((int64_t)b * p.out_h) * p.out_w + ox) * C8 + c8;
    const uint4* rb = reinterpret_cast<const uint4*>(ep.ref) + (((int64_t)b * p.out_h) * p.out_w + ox) * C8 + c8;
    float bb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (EPI == EPI_BIAS_ACT) {
#pragma unroll
        for (int e = 0; e < 8; ++e) bb[e] = ep.bias[8 * c8 + e];
    }
    auto finish = [&](int oy, int col, float (&o)[8]) {
        const int64_t off = (int64_t)oy * p.out_w * C8 + (int64_t)col * C8;
        if (EPI != EPI_NONE) {
            const uint4 qv = pack8(o);
            unpack8(qv, o);
        }
        if (EPI == EPI_ACT_BWD) {
            float rf[8];
            unpack8(rb[off], rf);
#pragma unroll
            for (int e = 0; e < 8; ++e) { o[e] = epi_act(o[e], rf[e], ep.alpha, ep.scale); bsum[e] += o[e]; }
            if (!active) return;
        }
        if (EPI == EPI_BIAS_ACT) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float v = o[e] + bb[e]; o[e] = epi_act(v, v, ep.alpha, ep.scale); }
        }
        yb[off] = pack8(o);
    };
    unsigned colmask = 0;
#pragma unroll
    for (int t = 0; t < 5; ++t) colmask |= ((ix0 + t >= 0) && (ix0 + t < p.in_w)) ? (1u << t) : 0u;
    if (!sep) {      // direct 16-tap form, one output at a time: correct for any FIR, not tuned
#pragma unroll 1
        for (int oy = oy0; oy < oy1; ++oy)
#pragma unroll 1
            for (int col = 0; col < (colB ? 2 : 1); ++col) {
                float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
                for (int j = 0; j < 4; ++j) {
                    const int iy = oy - p.pad_y0 + j;
                    if (iy < 0 || iy >= p.in_h) continue;
#pragma unroll 1
                    for (int t = 0; t < 4; ++t) {
                        const int ix = ix0 + col + t;
                        if (ix < 0 || ix >= p.in_w) continue;
                        float f[8];
                        unpack8(xb[((int64_t)iy * p.in_w + ix) * C8], f);
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[e] = fmaf(f[e], sk[4 * j + t], acc[e]);
                    }
                }
                finish(oy, col, acc);
            }
        if (EPI == EPI_ACT_BWD) bgrad_flush(s_bg, ep.bgrad, p.C, active, 8 * c8, bsum, 8);
        return;
    }
    // one input row -> its two horizontally filtered values (columns ox and ox + 1)
    auto hrow = [&](int iy, F8& ha, F8& hb) {
        const bool rowok = (iy >= 0) && (iy < p.in_h);
        const uint4* xr = xb + (int64_t)(rowok ? iy : 0) * p.in_w * C8;
        uint4 v[5];
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const bool ok = rowok && ((colmask >> t) & 1u);
            const uint4 w = xr[(int64_t)(ok ? ix0 + t : 0) * C8];
            const unsigned m = ok ? 0xffffffffu : 0u;
            v[t] = make_uint4(w.x & m, w.y & m, w.z & m, w.w & m);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { ha.v[e] = 0.f; hb.v[e] = 0.f; }
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            float f[8];
            unpack8(v[t], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (t < 4) ha.v[e] = fmaf(f[e], kh[t], ha.v[e]);
                if (t > 0) hb.v[e] = fmaf(f[e], kh[t - 1], hb.v[e]);
            }
        }
    };
    auto emit = [&](int oy, const F8& a0, const F8& a1, const F8& a2, const F8& a3, const F8& b0, const F8& b1, const F8& b2, const F8& b3) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = a0.v[e] * kv[0] + a1.v[e] * kv[1] + a2.v[e] * kv[2] + a3.v[e] * kv[3];
        finish(oy, 0, o);
        if (colB) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = b0.v[e] * kv[0] + b1.v[e] * kv[1] + b2.v[e] * kv[2] + b3.v[e] * kv[3];
            finish(oy, 1, o);
        }
    };
    F8 a0, a1, a2, a3, a4, b0, b1, b2, b3, b4;
    {
        const int iy0 = oy0 - p.pad_y0;
        hrow(iy0 + 0, a0, b0); hrow(iy0 + 1, a1, b1); hrow(iy0 + 2, a2, b2);
    }
    int oy = oy0;
#pragma unroll 1
    for (; oy + 1 < oy1; oy += 2) {      // two output rows per trip: ten 16-byte loads in flight
        hrow(oy - p.pad_y0 + 3, a3, b3);
        hrow(oy - p.pad_y0 + 4, a4, b4);
        emit(oy, a0, a1, a2, a3, b0, b1, b2, b3);
        emit(oy + 1, a1, a2, a3, a4, b1, b2, b3, b4);
        a0 = a2; a1 = a3; a2 = a4; b0 = b2; b1 = b3; b2 = b4;
    }
    if (oy < oy1) {
        hrow(oy - p.pad_y0 + 3, a3, b3);
        emit(oy, a0, a1, a2, a3, b0, b1, b2, b3);
    }
    if (EPI == EPI_ACT_BWD) bgrad_flush(s_bg, ep.bgrad, p.C, active, 8 * c8, bsum, 8);
}

// ---------------- bf16 NHWC decimating / zero-stuffing 4x4 FIRs, 8 channels (16 bytes) per thread ------------------------------
// Same reason as blur4_bf16x8: with 2-byte elements the 4-channel kernels above move 8 bytes per lane and instruction and are
// latency-bound (fir4_up2 / fir4_down2 on bf16: ~1.5 TB/s).  down2 keeps a window of four horizontally filtered rows (the FIR is an
// outer product; checked on the device, a rank > 1 table takes the direct loop) and advances two input rows per output row; up2
// keeps the 3 x 3 input neighbourhood of a 2 x 2 output block in registers (72 floats).
__global__ __launch_bounds__(256) void fir4_down2_bf16x8(uint4* __restrict__ y, const uint4* __restrict__ x,
                                                         const float* __restrict__ fir, FirParams p) {
    __shared__ float sk[16];
    if (threadIdx.x < 16) {
        const int t = threadIdx.x;
        sk[t] = fir[p.flip ? 15 - t : t] * p.gain;
    }
    __syncthreads();
    float kh[4], kv[4];
    float kmax = 0.f, res = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) { kh[t] = sk[t]; kv[t] = sk[0] != 0.f ? sk[4 * t] / sk[0] : 0.f; }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t) { kmax = fmaxf(kmax, fabsf(sk[4 * j + t])); res = fmaxf(res, fabsf(sk[4 * j + t] - kv[j] * kh[t])); }
    const bool sep = res <= 1e-6f * kmax;           // block-uniform
    const int C8 = p.C >> 3;
    const int segs = (p.out_h + p.seg_rows - 1) / p.seg_rows;
    const int64_t total = (int64_t)p.B * segs * p.out_w * C8;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int64_t r = i;
    const int c8 = (int)(r % C8); r /= C8;
    const int ox = (int)(r % p.out_w); r /= p.out_w;
    const int seg = (int)(r % segs);
    const int b = (int)(r / segs);
    const int oy0 = seg * p.seg_rows;
    const int oy1 = (oy0 + p.seg_rows < p.out_h) ? oy0 + p.seg_rows : p.out_h;
    const int ix0 = 2 * ox - p.pad_x0;
    const uint4* xb = x + (int64_t)b * p.in_h * p.in_w * C8 + c8;
    uint4* yb = y + (((int64_t)b * p.out_h) * p.out_w + ox) * C8 + c8;
    unsigned colmask = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) colmask |= ((ix0 + t >= 0) && (ix0 + t < p.in_w)) ? (1u << t) : 0u;
    auto load_row = [&](int iy, uint4 (&dst)[4]) {
        const bool rowok = (iy >= 0) && (iy < p.in_h);
        const uint4* xr = xb + (int64_t)(rowok ? iy : 0) * p.in_w * C8;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bool ok = rowok && ((colmask >> t) & 1u);
            const uint4 v = xr[(int64_t)(ok ? ix0 + t : 0) * C8];
            const unsigned m = ok ? 0xffffffffu : 0u;
            dst[t] = make_uint4(v.x & m, v.y & m, v.z & m, v.w & m);
        }
    };
    if (!sep) {      // direct 16-tap form: correct for any FIR, not tuned
#pragma unroll 1
        for (int oy = oy0; oy < oy1; ++oy) {
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int j = 0; j < 4; ++j) {
                uint4 row[4];
                load_row(2 * oy - p.pad_y0 + j, row);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float f[8];
                    unpack8(row[t], f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] = fmaf(f[e], sk[4 * j + t], acc[e]);
                }
            }
            yb[(int64_t)oy * p.out_w * C8] = pack8(acc);
        }
        return;
    }
    auto hfilter = [&](const uint4 (&row)[4], F8& h) {
#pragma unroll
        for (int e = 0; e < 8; ++e) h.v[e] = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float f[8];
            unpack8(row[t], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) h.v[e] = fmaf(f[e], kh[t], h.v[e]);
        }
    };
    F8 h0, h1, h2, h3;
    {
        uint4 r0[4], r1[4];
        load_row(2 * oy0 - p.pad_y0 + 0, r0);
        load_row(2 * oy0 - p.pad_y0 + 1, r1);
        hfilter(r0, h0); hfilter(r1, h1);
    }
#pragma unroll 1
    for (int oy = oy0; oy < oy1; ++oy) {
        uint4 ra[4], rb[4];
        load_row(2 * oy - p.pad_y0 + 2, ra);
        load_row(2 * oy - p.pad_y0 + 3, rb);
        hfilter(ra, h2); hfilter(rb, h3);
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = h0.v[e] * kv[0] + h1.v[e] * kv[1] + h2.v[e] * kv[2] + h3.v[e] * kv[3];
        yb[(int64_t)oy * p.out_w * C8] = pack8(o);
        h0 = h2; h1 = h3;
    }
}

__global__ __launch_bounds__(256) void fir4_up2_bf16x8(uint4* __restrict__ y, const uint4* __restrict__ x, const float* __restrict__ fir,
                                                       FirParams p, const uint4* __restrict__ resid) {
    __shared__ float sk[16];
    if (threadIdx.x < 16) {
        const int t = threadIdx.x;
        sk[t] = fir[p.flip ? 15 - t : t] * p.gain;
    }
    __syncthreads();
    const int C8 = p.C >> 3;
    const int bh = (p.out_h + 1) >> 1, bw = (p.out_w + 1) >> 1;          // 2 x 2 output blocks
    const int segs = (bh + p.seg_rows - 1) / p.seg_rows;
    const int64_t total = (int64_t)p.B * segs * bw * C8;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int64_t r = idx;
    const int c8 = (int)(r % C8); r /= C8;
    const int j = (int)(r % bw); r /= bw;
    const int seg = (int)(r % segs);
    const int b = (int)(r / segs);
    const int i0 = seg * p.seg_rows;
    const int i1 = (i0 + p.seg_rows < bh) ? i0 + p.seg_rows : bh;
    // per-axis tap / offset tables (a = output parity), as in fir4_up2_nhwc
    const int ky0[2] = {p.pad_y0 & 1, (p.pad_y0 - 1) & 1}, kx0[2] = {p.pad_x0 & 1, (p.pad_x0 - 1) & 1};
    const int dy[2] = {(ky0[0] - p.pad_y0) >> 1, (1 + ky0[1] - p.pad_y0) >> 1};
    const int dx[2] = {(kx0[0] - p.pad_x0) >> 1, (1 + kx0[1] - p.pad_x0) >> 1};
    const int ey = dy[1] - dy[0], ex = dx[1] - dx[0];
    float wq[2][2][2][2];                                                  // [a][bb][u][v]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int v = 0; v < 2; ++v) wq[a][bb][u][v] = sk[(ky0[a] + 2 * u) * 4 + kx0[bb] + 2 * v];
    const uint4* xb = x + (int64_t)b * p.in_h * p.in_w * C8 + c8;
    const int cx0 = j + dx[0];
    unsigned colmask = 0;
#pragma unroll
    for (int t = 0; t < 3; ++t) colmask |= ((cx0 + t >= 0) && (cx0 + t < p.in_w)) ? (1u << t) : 0u;
    auto load_row = [&](int iy, F8 (&dst)[3]) {
        const bool rowok = (iy >= 0) && (iy < p.in_h);
        const uint4* xr = xb + (int64_t)(rowok ? iy : 0) * p.in_w * C8;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const bool ok = rowok && ((colmask >> t) & 1u);
            const uint4 v = xr[(int64_t)(ok ? cx0 + t : 0) * C8];
            const unsigned m = ok ? 0xffffffffu : 0u;
            unpack8(make_uint4(v.x & m, v.y & m, v.z & m, v.w & m), dst[t].v);
        }
    };
    F8 R0[3], R1[3], R2[3];
    load_row(i0 + dy[0] + 0, R0);
    load_row(i0 + dy[0] + 1, R1);
    const bool colB = 2 * j + 1 < p.out_w;
#pragma unroll 1
    for (int i = i0; i < i1; ++i) {
        load_row(i + dy[0] + 2, R2);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int oy = 2 * i + a;
            if (oy >= p.out_h) break;
            const bool hi = (a == 1) && (ey == 1);
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                if (bb == 1 && !colB) break;
                const bool hx = (bb == 1) && (ex == 1);
                const float w00 = wq[a][bb][0][0], w01 = wq[a][bb][0][1], w10 = wq[a][bb][1][0], w11 = wq[a][bb][1][1];
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float top0 = hi ? R1[0].v[e] : R0[0].v[e], top1 = hi ? R1[1].v[e] : R0[1].v[e], top2 = hi ? R1[2].v[e] : R0[2].v[e];
                    const float bot0 = hi ? R2[0].v[e] : R1[0].v[e], bot1 = hi ? R2[1].v[e] : R1[1].v[e], bot2 = hi ? R2[2].v[e] : R1[2].v[e];
                    const float t0 = hx ? top1 : top0, t1 = hx ? top2 : top1, b0 = hx ? bot1 : bot0, b1 = hx ? bot2 : bot1;
                    o[e] = t0 * w00 + t1 * w01 + b0 * w10 + b1 * w11;
                }
                const int64_t yi = (((int64_t)b * p.out_h + oy) * p.out_w + 2 * j + bb) * C8 + c8;
                if (resid) {
                    const uint4 q = pack8(o);          // the two-kernel chain stores the FIR (bf16) first
                    unpack8(q, o);
                    float rf[8];
                    unpack8(resid[yi], rf);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] += rf[e];
                }
                y[yi] = pack8(o);
            }
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) { R0[t] = R1[t]; R1[t] = R2[t]; }
    }
}

// ---------------- NCHW tiled blur, up = down = 1, k <= 4 -------------------------------------------
#define TNH 16
#define TNW 64
__global__ __launch_bounds__(256) void blur_nchw_tile(float* __restrict__ y, const float* __restrict__ x,
                                                      const float* __restrict__ fir, FirParams p) {
    __shared__ float sk[16];
    __shared__ float sx[TNH + 3][TNW + 3 + 1];
    if (threadIdx.x < 16) {
        int t = threadIdx.x;
        int ky = t >> 2, kx = t & 3;
        float v = 0.f;
        if (ky < p.kh && kx < p.kw) {
            int src = p.flip ? (p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx) : ky * p.kw + kx;
            v = fir[src] * p.gain;
        }
        sk[t] = v;
    }
    const int tiles_x = (p.out_w + TNW - 1) / TNW;
    const int tiles_y = (p.out_h + TNH - 1) / TNH;
    int64_t bid = blockIdx.x;
    const int tx = (int)(bid % tiles_x); bid /= tiles_x;
    const int ty = (int)(bid % tiles_y);
    const int64_t plane = bid / tiles_y;
    const int oy0 = ty * TNH, ox0 = tx * TNW;
    const float* xp = x + plane * p.in_h * p.in_w;
    for (int t = threadIdx.x; t < (TNH + 3) * (TNW + 3); t += blockDim.x) {
        int ry = t / (TNW + 3), rx = t % (TNW + 3);
        int iy = oy0 + ry - p.pad_y0, ix = ox0 + rx - p.pad_x0;
        float v = 0.f;
        if (iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w) v = xp[(int64_t)iy * p.in_w + ix];
        sx[ry][rx] = v;
    }
    __syncthreads();
    float* yp = y + plane * p.out_h * p.out_w;
    for (int t = threadIdx.x; t < TNH * TNW; t += blockDim.x) {
        int ry = t / TNW, rx = t % TNW;
        int oy = oy0 + ry, ox = ox0 + rx;
        if (oy >= p.out_h || ox >= p.out_w) continue;
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < 4; ++ky)
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) acc += sx[ry + ky][rx + kx] * sk[ky * 4 + kx];
        yp[(int64_t)oy * p.out_w + ox] = acc;
    }
}

}  // namespace

// the unit-stride 4x4 NHWC blur (+ fused stage): shared by ideas_upfirdn2d (EPI_NONE) and ideas_blur_fused
template <int EPI>
static int launch_blur4(void* y, const void* x, const float* fir, FirParams p, FirEpi ep, int dtype, hipStream_t stream) {
    // rows marched per thread: each segment re-reads 3 halo rows (19/16 of the input).  With the two-column kernels (half the
    // threads per row) 16 beats 8 / 24 / 32 / 64 at every size: 5.3-5.5 TB/s against 4.9-5.0 (32) in f32 at 256x256
    p.seg_rows = 16;
    const int segs = (p.out_h + p.seg_rows - 1) / p.seg_rows;
    const int64_t total = (int64_t)p.B * segs * p.out_w * (p.C / 4);
    const int64_t grid = ideas_cdiv(total, 256);
    if (grid > 0x7fffffffLL) return IDEAS_E_SHAPE;
    const size_t lds = EPI == EPI_ACT_BWD ? (size_t)p.C * sizeof(float) : 0;
#ifndef IDEAS_BLUR_C2
#define IDEAS_BLUR_C2 1
#endif
    if (IDEAS_BLUR_C2 && dtype == IDEAS_BF16 && p.C % 8 == 0) {
        const int64_t total8 = (int64_t)p.B * segs * ((p.out_w + 1) / 2) * (p.C / 8);
        hipLaunchKernelGGL(blur4_bf16x8_c2<EPI>, dim3((unsigned)ideas_cdiv(total8, 256)), dim3(256), lds, stream, (uint4*)y,
                           (const uint4*)x, fir, p, ep);
        return ideas_launch_status();
    }
    if (dtype == IDEAS_BF16 && p.C % 8 == 0) {
        const int64_t total8 = (int64_t)p.B * segs * p.out_w * (p.C / 8);
        hipLaunchKernelGGL(blur4_bf16x8<EPI>, dim3((unsigned)ideas_cdiv(total8, 256)), dim3(256), lds, stream, (uint4*)y,
                           (const uint4*)x, fir, p, ep);
    } else if (dtype == IDEAS_BF16)
        hipLaunchKernelGGL((blur4_nhwc<ideas_bf16x4, EPI>), dim3((unsigned)grid), dim3(256), lds, stream, (ideas_bf16x4*)y,
                           (const ideas_bf16x4*)x, fir, p, ep);
    else if (IDEAS_BLUR_C2) {
        const int64_t total2 = (int64_t)p.B * segs * ((p.out_w + 1) / 2) * (p.C / 4);
        hipLaunchKernelGGL(blur4_f32_c2<EPI>, dim3((unsigned)ideas_cdiv(total2, 256)), dim3(256), lds, stream, (float4*)y,
                           (const float4*)x, fir, p, ep);
    } else
        hipLaunchKernelGGL((blur4_nhwc<float4, EPI>), dim3((unsigned)grid), dim3(256), lds, stream, (float4*)y, (const float4*)x, fir,
                           p, ep);
    return ideas_launch_status();
}

extern "C" int ideas_blur_fused(void* y, const void* x, const float* fir, int B, int C, int in_h, int in_w, int out_h, int out_w,
                                int pad_x0, int pad_y0, float gain, int flip, int mode, const void* ref, const float* bias,
                                float* bias_grad, float alpha, float scale, int dtype, void* stream_) {
    if (dtype != IDEAS_F32 && dtype != IDEAS_BF16) return IDEAS_E_UNSUPPORTED;
    if (!y || !x || !fir) return IDEAS_E_NULL;
    if (B <= 0 || C <= 0 || in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0) return IDEAS_E_SHAPE;
    if (C % 4 || C > 8192) return IDEAS_E_UNSUPPORTED;
    if (!ideas_aligned16(x) || !ideas_aligned16(y)) return IDEAS_E_ALIGN;
    FirParams p{B, C, in_h, in_w, out_h, out_w, 4, 4, 1, 1, 1, 1, pad_x0, pad_y0, gain, flip, 16};
    FirEpi ep{ref, bias, bias_grad, alpha, scale};
    hipStream_t stream = (hipStream_t)stream_;
    if (mode == EPI_ACT_BWD) {
        if (!ref || !bias_grad) return IDEAS_E_NULL;
        if (!ideas_aligned16(ref)) return IDEAS_E_ALIGN;
        return launch_blur4<EPI_ACT_BWD>(y, x, fir, p, ep, dtype, stream);
    }
    if (mode == EPI_BIAS_ACT) {
        if (!bias) return IDEAS_E_NULL;
        if (!ideas_aligned16(bias)) return IDEAS_E_ALIGN;
        return launch_blur4<EPI_BIAS_ACT>(y, x, fir, p, ep, dtype, stream);
    }
    return IDEAS_E_UNSUPPORTED;
}

static int upfirdn2d_impl(void* y, const void* x, const float* fir, int B, int C, int in_h, int in_w, int out_h,
                          int out_w, int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0,
                          int pad_y0, float gain, int flip, int layout, int dtype, void* stream_, const void* resid) {
    if (dtype != IDEAS_F32 && dtype != IDEAS_BF16) return IDEAS_E_UNSUPPORTED;
    if (dtype == IDEAS_BF16 && layout != IDEAS_NHWC) return IDEAS_E_UNSUPPORTED;
    if (!y || !x || !fir) return IDEAS_E_NULL;
    if (B <= 0 || C <= 0 || in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0) return IDEAS_E_SHAPE;
    if (kh <= 0 || kw <= 0 || kh > 8 || kw > 8) return IDEAS_E_UNSUPPORTED;
    if (up_x <= 0 || up_y <= 0 || down_x <= 0 || down_y <= 0) return IDEAS_E_SHAPE;
    if (layout != IDEAS_NCHW && layout != IDEAS_NHWC) return IDEAS_E_UNSUPPORTED;
    hipStream_t stream = (hipStream_t)stream_;
    FirParams p{B, C, in_h, in_w, out_h, out_w, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0, gain, flip, 16};
    const bool unit = up_x == 1 && up_y == 1 && down_x == 1 && down_y == 1;
    if (unit && layout == IDEAS_NHWC && kh == 4 && kw == 4 && (C % 4 == 0) && ideas_aligned16(x) && ideas_aligned16(y)) {
        return launch_blur4<EPI_NONE>(y, x, fir, p, FirEpi{nullptr, nullptr, nullptr, 0.f, 1.f}, dtype, stream);
    }
    const bool vec4 = layout == IDEAS_NHWC && kh == 4 && kw == 4 && (C % 4 == 0) && ideas_aligned16(x) && ideas_aligned16(y);
    if (vec4 && up_x == 1 && up_y == 1 && down_x == 2 && down_y == 2) {
        p.seg_rows = out_h >= 64 ? 16 : 8;
        const int segs = (out_h + p.seg_rows - 1) / p.seg_rows;
        const int64_t grid = ideas_cdiv((int64_t)B * segs * out_w * (C / 4), 256);
        if (grid > 0x7fffffffLL) return IDEAS_E_SHAPE;
        if (dtype == IDEAS_BF16 && C % 8 == 0)
            hipLaunchKernelGGL(fir4_down2_bf16x8, dim3((unsigned)ideas_cdiv((int64_t)B * segs * out_w * (C / 8), 256)), dim3(256), 0, stream,
                               (uint4*)y, (const uint4*)x, fir, p);
        else if (dtype == IDEAS_BF16)
            hipLaunchKernelGGL(fir4_down2_nhwc<ideas_bf16x4>, dim3((unsigned)grid), dim3(256), 0, stream, (ideas_bf16x4*)y,
                               (const ideas_bf16x4*)x, fir, p);
        else
            hipLaunchKernelGGL(fir4_down2_nhwc<float4>, dim3((unsigned)grid), dim3(256), 0, stream, (float4*)y, (const float4*)x, fir, p);
        return ideas_launch_status();
    }
    if (resid && !(vec4 && up_x == 2 && up_y == 2 && down_x == 1 && down_y == 1 && ideas_aligned16(resid))) return IDEAS_E_UNSUPPORTED;
    if (vec4 && up_x == 2 && up_y == 2 && down_x == 1 && down_y == 1) {
        const int bh = (out_h + 1) / 2, bw = (out_w + 1) / 2;
        p.seg_rows = bh >= 64 ? 16 : 8;
        const int segs = (bh + p.seg_rows - 1) / p.seg_rows;
        const int64_t grid = ideas_cdiv((int64_t)B * segs * bw * (C / 4), 256);
        if (grid > 0x7fffffffLL) return IDEAS_E_SHAPE;
        if (dtype == IDEAS_BF16 && C % 8 == 0)
            hipLaunchKernelGGL(fir4_up2_bf16x8, dim3((unsigned)ideas_cdiv((int64_t)B * segs * bw * (C / 8), 256)), dim3(256), 0, stream,
                               (uint4*)y, (const uint4*)x, fir, p, (const uint4*)resid);
        else if (dtype == IDEAS_BF16)
            hipLaunchKernelGGL(fir4_up2_nhwc<ideas_bf16x4>, dim3((unsigned)grid), dim3(256), 0, stream, (ideas_bf16x4*)y,
                               (const ideas_bf16x4*)x, fir, p, (const ideas_bf16x4*)resid);
        else
            hipLaunchKernelGGL(fir4_up2_nhwc<float4>, dim3((unsigned)grid), dim3(256), 0, stream, (float4*)y, (const float4*)x, fir, p,
                               (const float4*)resid);
        return ideas_launch_status();
    }
    if (unit && layout == IDEAS_NCHW && kh <= 4 && kw <= 4) {
        const int64_t grid = (int64_t)B * C * ideas_cdiv(out_h, TNH) * ideas_cdiv(out_w, TNW);
        if (grid > 0x7fffffffLL) return IDEAS_E_SHAPE;
        hipLaunchKernelGGL(blur_nchw_tile, dim3((unsigned)grid), dim3(256), 0, stream, (float*)y, (const float*)x, fir, p);
        return ideas_launch_status();
    }
    const int64_t total = (int64_t)B * C * out_h * out_w;
    int64_t grid = ideas_cdiv(total, 256);
    if (grid > 65536) grid = 65536;
    if (dtype == IDEAS_BF16)
        hipLaunchKernelGGL((upfirdn2d_generic<true, ideas_bf16>), dim3((unsigned)grid), dim3(256), 0, stream, (ideas_bf16*)y,
                           (const ideas_bf16*)x, fir, p);
    else if (layout == IDEAS_NHWC)
        hipLaunchKernelGGL((upfirdn2d_generic<true, float>), dim3((unsigned)grid), dim3(256), 0, stream, (float*)y,
                           (const float*)x, fir, p);
    else
        hipLaunchKernelGGL((upfirdn2d_generic<false, float>), dim3((unsigned)grid), dim3(256), 0, stream, (float*)y,
                           (const float*)x, fir, p);
    return ideas_launch_status();
}

extern "C" int ideas_upfirdn2d(void* y, const void* x, const float* fir, int B, int C, int in_h, int in_w, int out_h,
                               int out_w, int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0,
                               int pad_y0, float gain, int flip, int layout, int dtype, void* stream_) {
    return upfirdn2d_impl(y, x, fir, B, C, in_h, in_w, out_h, out_w, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0, gain, flip,
                          layout, dtype, stream_, nullptr);
}

extern "C" int ideas_fir_up2_add(void* y, const void* x, const float* fir, const void* resid, int B, int C, int in_h, int in_w,
                                 int out_h, int out_w, int pad_x0, int pad_y0, float gain, int flip, int dtype, void* stream_) {
    if (!resid) return IDEAS_E_NULL;
    return upfirdn2d_impl(y, x, fir, B, C, in_h, in_w, out_h, out_w, 4, 4, 2, 2, 1, 1, pad_x0, pad_y0, gain, flip, IDEAS_NHWC, dtype,
                          stream_, resid);
}
