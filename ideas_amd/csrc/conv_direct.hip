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

// The same with FOUR consecutive output channels per thread (Cout % 4 == 0): one 16-byte (f32) / 8-byte (bf16) store per lane
// instead of four scalar ones -- the scalar kernel above is bound by store instructions (1.3-1.9 TB/s on from-RGB 3 -> 64 and on
// to-RGB's input gradient 3 -> 128 at 256x256).  Same per-element operation order.
template <typename T> struct Vec4Of;
template <> struct Vec4Of<float> { typedef float4 type; };
template <> struct Vec4Of<bf16_t> { typedef ideas_bf16x4 type; };
template <int KMAX, typename T>
__global__ __launch_bounds__(256) void pointwise_smallk_vec_kernel(T* __restrict__ y, const T* __restrict__ x,
                                                                   const float* __restrict__ w, const float* __restrict__ bias,
                                                                   const T* __restrict__ resid, int64_t P, int Cin, int Cout,
                                                                   float gain, int act, float alpha, float act_gain,
                                                                   float resid_gain, int accumulate) {
    typedef typename Vec4Of<T>::type V;
    const int C4 = Cout >> 2;
    const int groups = blockDim.x / C4;              // pixel lanes per block
    const int o4 = threadIdx.x % C4, grp = threadIdx.x / C4;
    if (grp >= groups) return;
    float wr[4][KMAX];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < KMAX; ++k) wr[e][k] = k < Cin ? w[(int64_t)(4 * o4 + e) * Cin + k] : 0.f;
    float bv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[e] = bias ? bias[4 * o4 + e] : 0.f;
    const int64_t stride = (int64_t)gridDim.x * groups;
    V* yv = reinterpret_cast<V*>(y);
    const V* rv = reinterpret_cast<const V*>(resid);
    for (int64_t pp = (int64_t)blockIdx.x * groups + grp; pp < P; pp += stride) {
        const T* xp = x + pp * Cin;
        float xs[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) xs[k] = k < Cin ? ldv(xp + k) : 0.f;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < Cin) acc = fmaf(xs[k], wr[e][k], acc);
            float t = mul_then_add(acc, gain, bv[e]);
            if (act) t = (t > 0.f ? t : t * alpha) * act_gain;
            v[e] = t;
        }
        const int64_t yi = pp * C4 + o4;
        if (resid) {
            const float4 r4 = to_f4(rv[yi]);
            v[0] = (v[0] + r4.x) * resid_gain; v[1] = (v[1] + r4.y) * resid_gain; v[2] = (v[2] + r4.z) * resid_gain; v[3] = (v[3] + r4.w) * resid_gain;
        }
        if (accumulate) {
            const float4 a4 = to_f4(yv[yi]);
            v[0] += a4.x; v[1] += a4.y; v[2] += a4.z; v[3] += a4.w;
        }
        yv[yi] = from_f4<V>(make_float4(v[0], v[1], v[2], v[3]));
    }
}

// gw[o][ci] += gain * sum_p gy[p][o] * x[p][ci] with min(Cin, Cout) <= 8.  WIDE_OUT: threads span o (gy coalesced,
// x broadcast, Cin accumulators); otherwise threads span ci (x coalesced, gy broadcast, Cout accumulators).
template <int SMALL, bool WIDE_OUT, typename T>
__global__ __launch_bounds__(256) void pointwise_small_wgrad_kernel(float* __restrict__ gw, const T* __restrict__ gy,
                                                                    const T* __restrict__ x, int64_t P, int Cin, int Cout,
                                                                    float gain, int64_t pix_per_block) {
    __shared__ float s_red[512 * 8];   // [wide][SMALL] block-level partial sums (wide <= 512)
    const int wide = WIDE_OUT ? Cout : Cin, small = WIDE_OUT ? Cin : Cout;
    const int lanes = wide < (int)blockDim.x ? wide : (int)blockDim.x;      // > 256 wide channels: each thread takes two
    const int groups = blockDim.x / lanes;
    const int c0 = threadIdx.x % lanes, grp = threadIdx.x / lanes;
    for (int i = threadIdx.x; i < wide * SMALL; i += blockDim.x) s_red[i] = 0.f;
    __syncthreads();
    if (grp < groups) {
        const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
        const int64_t p1 = (p0 + pix_per_block < P) ? p0 + pix_per_block : P;
        for (int c = c0; c < wide; c += lanes) {
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

// ---- pointwise layers with one WIDE and one tiny channel count: 16-byte vector access along the wide axis ---------------
// to_rgb (128 -> 3), the input gradient of the from-RGB convs (64 / 32 -> 3), the 8- / N-channel heads, and their weight
// gradients.  They are pure streams over the wide tensor (1.6 GFLOP against 0.5 GB at B = 32), but the one-thread-per-output
// kernels above read it with 2- / 4-byte accesses and ran at 0.2-0.5 TB/s (to_rgb forward: 2.4 ms in bf16).  Here a group of
// L lanes owns one pixel, each lane VEC = 16 B / sizeof(T) consecutive channels (coalesced 16-byte loads, L * 16 contiguous
// bytes per pixel), the tiny axis lives in registers, and the L partial sums meet through wave shuffles.
template <typename T> struct VecOf;
template <> struct VecOf<float> { static constexpr int N = 4; };
template <> struct VecOf<bf16_t> { static constexpr int N = 8; };
__device__ __forceinline__ void ldvec(const float* p, float (&f)[4]) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
}
__device__ __forceinline__ void ldvec(const bf16_t* p, float (&f)[8]) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    f[0] = __builtin_bit_cast(float, u.x << 16); f[1] = __builtin_bit_cast(float, u.x & 0xffff0000u);
    f[2] = __builtin_bit_cast(float, u.y << 16); f[3] = __builtin_bit_cast(float, u.y & 0xffff0000u);
    f[4] = __builtin_bit_cast(float, u.z << 16); f[5] = __builtin_bit_cast(float, u.z & 0xffff0000u);
    f[6] = __builtin_bit_cast(float, u.w << 16); f[7] = __builtin_bit_cast(float, u.w & 0xffff0000u);
}

// y[p][o] = epilogue(sum_c x[p][c] w[o][c]),  Cout <= NS, Cin = L * VEC * IT
template <typename T, int NS, int IT>
__global__ __launch_bounds__(256) void pointwise_rowdot_kernel(T* __restrict__ y, const T* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, const T* __restrict__ resid, int64_t P,
                                                               int Cin, int Cout, int L, float gain, int act, float alpha,
                                                               float act_gain, float resid_gain, int accumulate) {
    constexpr int VEC = VecOf<T>::N;
    const int lane = threadIdx.x & 63;
    const int sub = lane & (L - 1), grp = lane / L, G = 64 / L;        // lane within its pixel group, group within the wave
    float wr[NS][IT][VEC];
#pragma unroll
    for (int o = 0; o < NS; ++o)
#pragma unroll
        for (int it = 0; it < IT; ++it)
#pragma unroll
            for (int e = 0; e < VEC; ++e) wr[o][it][e] = o < Cout ? w[(int64_t)o * Cin + (it * L + sub) * VEC + e] : 0.f;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t p0 = wave * G; p0 < P; p0 += nwaves * G) {
        const int64_t pp = p0 + grp;
        const bool ok = pp < P;
        const T* xp = x + (ok ? pp : P - 1) * Cin;
        float acc[NS];
#pragma unroll
        for (int o = 0; o < NS; ++o) acc[o] = 0.f;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            float f[VEC];
            ldvec(xp + (it * L + sub) * VEC, f);
#pragma unroll
            for (int o = 0; o < NS; ++o)
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[o] = fmaf(f[e], wr[o][it][e], acc[o]);
        }
        for (int m = 1; m < L; m <<= 1)
#pragma unroll
            for (int o = 0; o < NS; ++o) acc[o] += __shfl_xor(acc[o], m, 64);
        if (ok && sub < Cout) {                       // lane `sub` of the group writes output channel `sub`
            float a = acc[0];
#pragma unroll
            for (int o = 1; o < NS; ++o) a = sub == o ? acc[o] : a;
            float v = mul_then_add(a, gain, bias ? bias[sub] : 0.f);
            if (act) v = (v > 0.f ? v : v * alpha) * act_gain;
            const int64_t yi = pp * Cout + sub;
            if (resid) v = (v + ldv(resid + yi)) * resid_gain;
            if (accumulate) v += ldv(y + yi);
            stv(y + yi, v);
        }
    }
}

// gw += gain * sum_p S[p][s] * W[p][c]:  W = the wide tensor [P][Cw] (16-byte loads), S = the tiny one [P][Cs], Cs <= NS.
// SMALL_IS_OUT: S = gy (Cs = Cout), W = x (Cw = Cin) -> gw[s][c];  otherwise S = x (Cs = Cin), W = gy (Cw = Cout) -> gw[c][s].
template <typename T, int NS, int IT, bool SMALL_IS_OUT>
__global__ __launch_bounds__(256) void pointwise_wgrad_vec_kernel(float* __restrict__ gw, const T* __restrict__ S, const T* __restrict__ W,
                                                                  int64_t P, int Cs, int Cw, int L, float gain, int64_t pix_per_block) {
    constexpr int VEC = VecOf<T>::N;
    extern __shared__ float s_red[];                 // [Cs][Cw] block partial sums
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int sub = lane & (L - 1), grp = lane / L, G = 64 / L;
    for (int i = threadIdx.x; i < Cs * Cw; i += blockDim.x) s_red[i] = 0.f;
    __syncthreads();
    float acc[NS][IT][VEC];
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_)
#pragma unroll
        for (int it = 0; it < IT; ++it)
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[s_][it][e] = 0.f;
    const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
    const int64_t p1 = (p0 + pix_per_block < P) ? p0 + pix_per_block : P;
    for (int64_t pp = p0 + wv * G + grp; pp < p1; pp += 4 * G) {
        float sv[NS];
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) sv[s_] = s_ < Cs ? ldv(S + pp * Cs + s_) : 0.f;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            float f[VEC];
            ldvec(W + pp * Cw + (it * L + sub) * VEC, f);
#pragma unroll
            for (int s_ = 0; s_ < NS; ++s_)
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[s_][it][e] = fmaf(sv[s_], f[e], acc[s_][it][e]);
        }
    }
    // fold the pixel groups of the wave (lanes with equal `sub`), then the waves through LDS, then one atomic per element
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_)
#pragma unroll
        for (int it = 0; it < IT; ++it)
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float v = acc[s_][it][e];
                for (int m = L; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
                if (grp == 0 && s_ < Cs) atomicAdd(&s_red[s_ * Cw + (it * L + sub) * VEC + e], v);
            }
    __syncthreads();
    for (int i = threadIdx.x; i < Cs * Cw; i += blockDim.x) {
        const int s_ = i / Cw, c = i % Cw;
        const int64_t dst = SMALL_IS_OUT ? (int64_t)s_ * Cw + c : (int64_t)c * Cs + s_;
        atomicAdd(&gw[dst], s_red[i] * gain);
    }
}

// (L, IT) with wide = L * VEC * IT, L a power of two <= 64, IT <= 4; false if the wide axis does not decompose that way
template <typename T>
bool vec_split(int wide, int& L, int& IT) {
    constexpr int VEC = VecOf<T>::N;
    if (wide % VEC) return false;
    const int chunks = wide / VEC;
    for (IT = 1; IT <= 4; ++IT) {
        if (chunks % IT) continue;
        L = chunks / IT;
        if (L <= 64 && L >= 1 && (L & (L - 1)) == 0) return true;
    }
    return false;
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
    int L = 0, IT = 0;
    if (is_pointwise(p) && !in_scale && !out_scale && p->Cout <= 8 && p->Cin >= 32 && vec_split<T>(p->Cin, L, IT) && p->Cout <= L &&
        ideas_aligned16(x)) {
        const int64_t P = (int64_t)p->B * p->OH * p->OW;
        int64_t grid = ideas_cdiv(P, (int64_t)(64 / L) * 4 * 4);
        if (grid > 4096) grid = 4096;
        if (grid < 1) grid = 1;
#define ROWDOT(NS_, IT_)                                                                                                              \
    hipLaunchKernelGGL((pointwise_rowdot_kernel<T, NS_, IT_>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (T*)y,       \
                       (const T*)x, (const float*)wmat, bias, (const T*)resid, P, p->Cin, p->Cout, L, p->gain, p->act, p->alpha,      \
                       p->act_gain, p->resid_gain, p->accumulate)
        if (p->Cout <= 4) { if (IT == 1) ROWDOT(4, 1); else if (IT == 2) ROWDOT(4, 2); else if (IT == 3) ROWDOT(4, 3); else ROWDOT(4, 4); }
        else { if (IT == 1) ROWDOT(8, 1); else if (IT == 2) ROWDOT(8, 2); else if (IT == 3) ROWDOT(8, 3); else ROWDOT(8, 4); }
#undef ROWDOT
        return ideas_launch_status();
    }
    if (is_pointwise(p) && !in_scale && !out_scale && p->Cin <= 4 && p->Cout % 4 == 0 && p->Cout >= 16 && p->Cout <= 1024 &&
        ideas_aligned16(y) && (!resid || ideas_aligned16(resid))) {
        const int64_t P = (int64_t)p->B * p->OH * p->OW;
        const int groups = 256 / (p->Cout / 4);
        int64_t grid = ideas_cdiv(P, (int64_t)groups * 8);
        if (grid > 8192) grid = 8192;
        if (grid < 1) grid = 1;
        hipLaunchKernelGGL((pointwise_smallk_vec_kernel<4, T>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (T*)y,
                           (const T*)x, (const float*)wmat, bias, (const T*)resid, P, p->Cin, p->Cout, p->gain, p->act, p->alpha,
                           p->act_gain, p->resid_gain, p->accumulate);
        return ideas_launch_status();
    }
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
    {
        // wide/tiny pointwise layers: vector kernel (tiny axis <= 4, wide axis >= 32 and decomposable)
        const bool small_out = p->Cout <= p->Cin;
        const int Cs = small_out ? p->Cout : p->Cin, Cw = small_out ? p->Cin : p->Cout;
        int L = 0, IT = 0;
        if (is_pointwise(p) && !in_scale && !out_scale && Cs <= 4 && Cw >= 32 && Cw <= 512 && vec_split<T>(Cw, L, IT) && IT <= 2 &&
            ideas_aligned16(small_out ? x : gy)) {
            int64_t blocks = ideas_cdiv(P, 2048);
            if (blocks > 2048) blocks = 2048;
            const int64_t per = ideas_cdiv(P, blocks);
            blocks = ideas_cdiv(P, per);
            const size_t lds = (size_t)Cs * Cw * sizeof(float);
#define WGV(IT_, SO_)                                                                                                                 \
    hipLaunchKernelGGL((pointwise_wgrad_vec_kernel<T, 4, IT_, SO_>), dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, gw, \
                       (const T*)(SO_ ? gy : x), (const T*)(SO_ ? x : gy), P, Cs, Cw, L, p->gain, per)
            if (small_out) { if (IT == 1) WGV(1, true); else WGV(2, true); }
            else { if (IT == 1) WGV(1, false); else WGV(2, false); }
#undef WGV
            return ideas_launch_status();
        }
    }
    if (is_pointwise(p) && !in_scale && !out_scale && (p->Cin <= 8 || p->Cout <= 8) && p->Cin <= 512 && p->Cout <= 512) {
        // 64 pixels per block (up to 1024 blocks): the layers that land here are the 16x16 ends of E / Gstru / Ex (P = 8192 at
        // B = 32), and 512 pixels per block left 16 blocks walking 1024 serial iterations each (0.56 ms for the 512 -> 8 layer)
        int64_t blocks = ideas_cdiv(P, 64);
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
    return is_pointwise(p) && (p->Cin <= 8 || p->Cout <= 8) && p->Cin <= 512 && p->Cout <= 512;
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
