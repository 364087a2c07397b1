// 3x3 / stride-1 / pad-1 convolution: 1-D Winograd F(2,3) along x (conv_wino.hip's algebra) contracted with the
// exact 3-way bf16 split of b3.hpp (conv_b3.hip's arithmetic), NHWC, f32 in/out.
//
//     out[y, 2t  ] = M0 + M1 + M2          M_v[y,t,o] = sum_{ky,ci} U_v[o][ky][ci] * V_v[y+ky-1, t, ci]
//     out[y, 2t+1] = M1 - M2 - M3          V = (d0-d2, d1+d2, d2-d1, d1-d3),  d_j = x[., 2t-1+j, ci]
//                                          U = (w0, (w0+w1+w2)/2, (w0-w1+w2)/2, w2)
//
// The b3 kernels are power-limited (the chip clocks down under the bf16 MFMA stream), so the lever that is left is
// issuing fewer products: F(2,3) needs 2/3 of the direct kernel's MFMAs AND 2/3 of its operand splits.
//
// Block = 256 threads, tile 64 column pairs (128 output pixels) x 64 output channels; wave v owns Winograd component v
// (its own A planes V_v, its own B planes U_v, four 32x32 accumulators), so the K loop is conv_b3's with 12 operand
// fetches and 24 MFMAs per wave and step.  K-steps run over (16-channel chunk, ky) with ky innermost.
//   A: thread (row r, quad kq) fetches the four window columns d0..d3 (buffer loads, padding -> out-of-range offset ->
//      hardware zeros), forms V0..V3 in f32, splits the 16 values and writes 12 eight-byte groups;
//   B: the transformed weights arrive pre-split from ideas_b3_wino_split_weights as [4 v][3 planes][step][Cout][16] bf16,
//      which IS the MFMA operand layout (row = channel, 16 K values contiguous): every wave fetches its six operand
//      registers for the next step straight from global memory (one coalesced 1 KB buffer load each) -- the weights never
//      pass through LDS, which halves the LDS write traffic and the operand ds_reads of the first version.
// LDS: two pipeline buffers of the 12 A planes (2 x 24 KB), one barrier per step.
// Epilogue: the four components of an output pair live in four different waves; they meet in LDS (two 32-channel halves),
// then out = inverse transform -> gain / demod / bias / act / residual exactly as conv_wino.hip.
#include "b3.hpp"
#ifndef WINO_ABL
#define WINO_ABL 0
#endif
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int WP = 64;            // column pairs per block
constexpr int WN = 64;            // output channels per block
constexpr int XROW = 36;          // floats per row of the exchange buffer (32 + pad: conflict-free float4 reads)

// ---------------------------------------------------------------------------------------------------------------
// weights: element (n, ky, kx, c) at w[base + n*sn + ky*sky + kx*skx + c*sc]  ->  U planes [4][3][3*C/16][N][16] bf16
//   forward : n = o, c = i on the OHWI parameter          (sn = 9I, sky = 3I, skx = I, sc = 1, base = 0)
//   dgrad   : n = i, c = o, taps flipped, same parameter  (sn = 1, sky = -3I, skx = -I, sc = 9I, base = 8I)
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wino_split_weights_body(uint2* __restrict__ dst, const float* __restrict__ w, int N, int C, int64_t sn,
                                                        int64_t sky, int64_t skx, int64_t sc, int64_t base, int64_t bid, int64_t nblk) {
    const int64_t total = (int64_t)N * 3 * (C / 4);
    const int nsteps = 3 * (C / 16);
    const int64_t plane = (int64_t)nsteps * N * 4;        // uint2 units per (v, plane)
    for (int64_t i = bid * 256 + threadIdx.x; i < total; i += nblk * 256) {
        const int c = (int)(i % (C / 4)) * 4;
        const int ky = (int)((i / (C / 4)) % 3);
        const int n = (int)(i / (3 * (C / 4)));
        float wk[3][4];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int e = 0; e < 4; ++e) wk[kx][e] = w[base + n * sn + ky * sky + kx * skx + (c + e) * sc];
        // The transformed weights are the same for every pixel of every sample, so an error in them is a COHERENT error of the
        // outputs: it does not average out in the backward's sums over 65 536 pixels (weight gradient, style / demodulation dot
        // products) the way per-pixel rounding does.  Rounding U twice in f32 ((w0 + w2) + w1, 1.2e-7 relative) made G's late-layer
        // gradients sit 4x further from f64 than stock f32 arithmetic (tests/test_nets_gpu.py::test_full_width_gradients_vs_oracle);
        // the sums are exact in double, and the three planes are cut from the double: hi + mid + lo = U to ~2^-26.
        double u[4][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const double s = (double)wk[0][e] + (double)wk[2][e];
            u[0][e] = wk[0][e];
            u[1][e] = (s + (double)wk[1][e]) * 0.5;
            u[2][e] = (s - (double)wk[1][e]) * 0.5;
            u[3][e] = wk[2][e];
        }
        const int step = (c >> 4) * 3 + ky;
        const int64_t o = ((int64_t)step * N + n) * 4 + ((c & 15) >> 2);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            Split4 s;
            split2d(u[v][0], u[v][1], s.p[0].x, s.p[1].x, s.p[2].x);
            split2d(u[v][2], u[v][3], s.p[0].y, s.p[1].y, s.p[2].y);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) dst[(v * 3 + pl) * plane + o] = s.p[pl];
        }
    }
}

__global__ __launch_bounds__(256) void wino_split_weights_kernel(uint2* __restrict__ dst, const float* __restrict__ w, int N, int C,
                                                                 int64_t sn, int64_t sky, int64_t skx, int64_t sc, int64_t base) {
    wino_split_weights_body(dst, w, N, C, sn, sky, skx, sc, base, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(256) void wino_split_weights_batched_kernel(const ideas_prep_desc* __restrict__ tbl, int n) {
    int local, nblk;
    const ideas_prep_desc* d = prep_lookup(tbl, n, local, nblk);
    wino_split_weights_body((uint2*)d->dst, d->w, d->a[0], d->a[1], d->s[0], d->s[1], d->s[2], d->s[3], d->s[4], local, nblk);
}

// NH = 1: 256 threads, tile 64 pairs x  64 channels, K-step 16 channels (4 waves  = the 4 Winograd components)
// NH = 2: 512 threads, tile 64 pairs x 128 channels, K-step 32 channels (8 waves  = 4 components x 2 channel halves):
//         the same window transform + split now feeds twice the products (3.3 instead of 6.6 vector instructions per
//         MFMA) and there is one barrier per 48 MFMAs; one block per CU (96 KB LDS), two waves per SIMD.
// Epilogue tail shared by the Winograd kernels: four consecutive channels n..n+3 of one pair row from the four component values
// (inverse transform, gain / demodulation / bias / activation / residual in the op order of the unfused kernels), stored as ONE
// 16-byte store per pixel when the row is 16-byte addressable (`vec`: Cout % 4 == 0 and aligned pointers) -- the element-wise form
// was 32 global_store_dword per thread whose lanes hit 4-byte pieces 32 bytes apart (12 us of a 40 us tile at 256x256).
__device__ __forceinline__ void wino_finish4(float* __restrict__ y, const float* __restrict__ resid, const float* __restrict__ out_scale,
                                             const float* __restrict__ bias, const ideas_conv_params& p, const float (&mm)[4][4],
                                             int64_t opix, int pb, int n, bool vec) {
    if (vec) {
        float osv[4] = {1.f, 1.f, 1.f, 1.f}, bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (out_scale) {
            const float4 t4 = *reinterpret_cast<const float4*>(out_scale + (int64_t)pb * p.Cout + n);
            osv[0] = t4.x; osv[1] = t4.y; osv[2] = t4.z; osv[3] = t4.w;
        }
        if (bias) {
            const float4 t4 = *reinterpret_cast<const float4*>(bias + n);
            bv[0] = t4.x; bv[1] = t4.y; bv[2] = t4.z; bv[3] = t4.w;
        }
#pragma unroll
        for (int px = 0; px < 2; ++px) {
            const int64_t yi = opix + (int64_t)px * p.Cout + n;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float o2 = px == 0 ? (mm[0][e] + mm[1][e]) + mm[2][e] : (mm[1][e] - mm[2][e]) - mm[3][e];
                float t = mul_rn(o2, p.gain);
                if (out_scale) t = mul_rn(t, osv[e]);
                t = mul_then_add(t, 1.0f, bv[e]);
                if (p.act) t = (t > 0.f ? t : t * p.alpha) * p.act_gain;
                v[e] = t;
            }
            if (resid) {
                const float4 r4 = *reinterpret_cast<const float4*>(resid + yi);
                v[0] = (v[0] + r4.x) * p.resid_gain; v[1] = (v[1] + r4.y) * p.resid_gain;
                v[2] = (v[2] + r4.z) * p.resid_gain; v[3] = (v[3] + r4.w) * p.resid_gain;
            }
            if (p.accumulate) {
                const float4 o4 = *reinterpret_cast<const float4*>(y + yi);
                v[0] += o4.x; v[1] += o4.y; v[2] += o4.z; v[3] += o4.w;
            }
            *reinterpret_cast<float4*>(y + yi) = make_float4(v[0], v[1], v[2], v[3]);
        }
        return;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (n + e >= p.Cout) continue;
        const float o2[2] = {(mm[0][e] + mm[1][e]) + mm[2][e], (mm[1][e] - mm[2][e]) - mm[3][e]};
        const float os = out_scale ? out_scale[(int64_t)pb * p.Cout + n + e] : 1.f;
        const float bvv = bias ? bias[n + e] : 0.f;
#pragma unroll
        for (int px = 0; px < 2; ++px) {
            float v = mul_rn(o2[px], p.gain);
            if (out_scale) v = mul_rn(v, os);
            v = mul_then_add(v, 1.0f, bvv);
            if (p.act) v = (v > 0.f ? v : v * p.alpha) * p.act_gain;
            const int64_t yi = opix + (int64_t)px * p.Cout + n + e;
            if (resid) v = (v + resid[yi]) * p.resid_gain;
            if (p.accumulate) y[yi] += v; else y[yi] = v;
        }
    }
}
__device__ __forceinline__ bool wino_vec_ok(const float* y, const float* resid, const float* out_scale, const float* bias, int Cout) {
    return (Cout & 3) == 0 && (((uintptr_t)y | (uintptr_t)resid | (uintptr_t)out_scale | (uintptr_t)bias) & 15) == 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Branch-free tails for the epilogue configurations the training step actually launches (round 5).  wino_finish4 tests five run-time
// flags per channel quad (out_scale / bias / act / resid / accumulate): uniform branches, but each one ends a basic block, so its
// loads are waited for one by one and nothing is packed -- ~600 of a wave's ~1550 vector instructions per 8-chunk tile, executed
// with the matrix pipe idle.  Here the configuration is a template parameter chosen by ONE uniform switch per
// half tile, a thread finishes its 8 channels x 2 pixels in straight-line code on 4-wide vectors (v_pk_add / v_pk_mul; no
// contraction: the op order and roundings are wino_finish4's, the results bitwise equal), the leaky-ReLU is max(t, alpha t)
// (equal to the select for 0 <= alpha <= 1, the only slopes admitted), and the four stores take 32-bit buffer offsets.
//   OS: per-(sample, channel) output scale   BA: bias + leaky-ReLU * act_gain   RS: (v + resid) * resid_gain
// ---------------------------------------------------------------------------------------------------------------
enum { EPI_GENERIC = 0, EPI_PLAIN = 1, EPI_OS = 2, EPI_BA = 3, EPI_OS_BA = 4, EPI_OS_BA_RS = 5 };
typedef float v4f __attribute__((ext_vector_type(4)));

template <bool OS, bool BA, bool RS>
__device__ __forceinline__ void wino_finish_fast(__amdgpu_buffer_rsrc_t ry, unsigned yoff, unsigned px_step, const float* __restrict__ resid,
                                                 const float* __restrict__ os_p, const float* __restrict__ bias_p,
                                                 const float* __restrict__ ex, bool two, float gain, float alpha, float act_gain,
                                                 float resid_gain) {
#pragma clang fp contract(off)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        if (g == 1 && !two) break;
        const v4f m0 = *reinterpret_cast<const v4f*>(ex + g * 4 + 0 * WP * XROW);
        const v4f m1 = *reinterpret_cast<const v4f*>(ex + g * 4 + 1 * WP * XROW);
        const v4f m2 = *reinterpret_cast<const v4f*>(ex + g * 4 + 2 * WP * XROW);
        const v4f m3 = *reinterpret_cast<const v4f*>(ex + g * 4 + 3 * WP * XROW);
        v4f osv, bv, r0, r1;
        if (OS) osv = *reinterpret_cast<const v4f*>(os_p + g * 4);
        if (BA) bv = *reinterpret_cast<const v4f*>(bias_p + g * 4);
        if (RS) {
            r0 = *reinterpret_cast<const v4f*>(resid + (yoff >> 2) + g * 4);
            r1 = *reinterpret_cast<const v4f*>(resid + ((yoff + px_step) >> 2) + g * 4);
        }
        v4f t0 = (m0 + m1) + m2, t1 = (m1 - m2) - m3;
        t0 = t0 * gain; t1 = t1 * gain;
        if (OS) { t0 = t0 * osv; t1 = t1 * osv; }
        if (BA) {
            t0 = t0 + bv; t1 = t1 + bv;
            const v4f a0 = t0 * alpha, a1 = t1 * alpha;
            t0 = __builtin_elementwise_max(t0, a0) * act_gain;
            t1 = __builtin_elementwise_max(t1, a1) * act_gain;
        }
        if (RS) { t0 = (t0 + r0) * resid_gain; t1 = (t1 + r1) * resid_gain; }
        // (no SGPR offset on the stores: conv_b3_s2fir.hip records the data hazard of the `soffset` form on this part)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, t0), ry,
                                               (int)(yoff + g * 16), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, t1), ry,
                                               (int)(yoff + px_step + g * 16), 0, 0);
    }
}

template <bool SCALE, bool REFLECT, int NH>
__global__ __launch_bounds__(256 * NH, NH == 1 ? 3 : 1) void conv_b3_wino_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                              const void* __restrict__ uplanes,
                                                              const float* __restrict__ in_scale,
                                                              const float* __restrict__ out_scale,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ resid, ideas_conv_params p,
                                                              int tiles_n, unsigned x_bytes, unsigned plane_bytes, int epi) {
    constexpr int KS = BK * NH;                 // channels per K-step
    constexpr int KQ = 4 * NH;                  // float4 quads per row and step
    constexpr int RB = ROWB * NH;               // bytes per LDS row
    constexpr int PL = WP * RB;                 // bytes per plane
    constexpr int BN = WN * NH;                 // channels per block
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 12 * PL];   // two buffers of the A planes [v*3+pl]
    static_assert(2 * 12 * PL >= NH * 4 * WP * XROW * 4, "exchange buffer must fit");

    const int t = threadIdx.x;
    const int H = p.IH, W = p.IW, W2 = W >> 1;
    const int64_t M = (int64_t)p.B * H * W2;
    const int swz = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tile_n = swz % tiles_n, tile_m = swz / tiles_n;
    const int64_t m0 = (int64_t)tile_m * WP;
    const int n0 = tile_n * BN;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)x_bytes, (int)RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc((void*)uplanes, 0, (int)(12u * plane_bytes), (int)RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc((void*)in_scale, 0, SCALE ? p.B * p.Cin * 4 : 0, (int)RSRC_FLAGS);

    // ---- staging: thread = (pair row r, channel quad kq of the K-step) --------------------------------------------------
    const int r = t / KQ, kq = t % KQ;
    unsigned colo[4], rowb[3], inv = 0, sbase;          // byte offsets; bit (ky*4+j) of inv = tap in the zero padding
    {
        int64_t m = m0 + r;
        m = m < M ? m : M - 1;                           // rows past M repeat the last one; nothing of them is stored
        const int ptx = (int)(m % W2);
        const int64_t q = m / W2;
        const int py = (int)(q % H), pb = (int)(q / H);
        bool cok[4], rok[3];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int ix = 2 * ptx - 1 + j;
            if (REFLECT) { ix = reflect_coord(ix, W); cok[j] = true; }
            else cok[j] = ix >= 0 && ix < W;
            colo[j] = (unsigned)(ix * p.Cin) * 4u;
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            int iy = py + ky - 1;
            if (REFLECT) { iy = reflect_coord(iy, H); rok[ky] = true; }
            else rok[ky] = iy >= 0 && iy < H;
            rowb[ky] = (unsigned)(((pb * H + iy) * W) * p.Cin + kq * 4) * 4u;
#pragma unroll
            for (int j = 0; j < 4; ++j) inv |= ((rok[ky] && cok[j]) ? 0u : 1u) << (ky * 4 + j);
        }
        sbase = (unsigned)(pb * p.Cin + kq * 4) * 4u;
    }
    // LDS position of this thread's 8-byte group: 16-byte chunks of a row are XOR-swizzled so that the ds_read_b128 operand
    // fetch is conflict-free (32-byte rows: chunk ^= row bit 3;  64-byte rows: chunk ^= row bits 2-3)
    const int a_lds = NH == 1 ? r * RB + ((kq * 8) ^ (((r >> 3) & 1) << 4))
                              : r * RB + ((((kq >> 1) ^ ((r >> 2) & 3)) << 4) | ((kq & 1) << 3));

    int k_ky = 0, k_ci = 0;                              // block-uniform walk, ky innermost
    struct Stage { float4 d[4], s; };
    Stage st0, st1;
    auto gloadA = [&](Stage& st) {
        // rowb[k_ky] by scalar masks (a select on a uniform condition comes back from hipcc as a branch)
        const unsigned e0 = (unsigned)(k_ky - 1) >> 31, e2 = (unsigned)k_ky >> 1, e1 = 1u - e0 - e2;
        const unsigned base = ((rowb[0] & (0u - e0)) | (rowb[1] & (0u - e1)) | (rowb[2] & (0u - e2))) + (unsigned)k_ci * 4u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned off = (base + colo[j]) | (unsigned)__builtin_amdgcn_sbfe(inv, k_ky * 4 + j, 1);
#if WINO_ABL == 2
            st.d[j] = make_float4(__builtin_bit_cast(float, off), 1.f, 2.f, 3.f);
#else
            st.d[j] = buffer_load4(rx, off, 0);
#endif
        }
        if (SCALE) st.s = buffer_load4(rs_, sbase, (unsigned)k_ci * 4u);
        // (pure arithmetic: hipcc turns uniform selects back into scalar BRANCHES, which cut the K loop's scheduling region)
        const int wrap = (k_ky + 1) / 3;                  // k_ky in 0..2
        k_ky = k_ky + 1 - 3 * wrap;
        k_ci += wrap * KS;
    };
    struct Planes { uint2 q[4][3]; };
    auto transform_split = [&](const Stage& st) {
        float4 d0 = st.d[0], d1 = st.d[1], d2 = st.d[2], d3 = st.d[3];
#if WINO_ABL == 1
        {
            Planes pl;
            const float4 dd[4] = {d0, d1, d2, d3};
            for (int c = 0; c < 4; ++c) {
                pl.q[c][0] = make_uint2(__builtin_bit_cast(unsigned, dd[c].x), __builtin_bit_cast(unsigned, dd[c].y));
                pl.q[c][1] = make_uint2(__builtin_bit_cast(unsigned, dd[c].z), __builtin_bit_cast(unsigned, dd[c].w));
                pl.q[c][2] = make_uint2(__builtin_bit_cast(unsigned, dd[c].y), __builtin_bit_cast(unsigned, dd[c].w));
            }
            return pl;
        }
#endif
        float4 v[4];
        v[0] = make_float4(d0.x - d2.x, d0.y - d2.y, d0.z - d2.z, d0.w - d2.w);
        v[1] = make_float4(d1.x + d2.x, d1.y + d2.y, d1.z + d2.z, d1.w + d2.w);
        v[2] = make_float4(d2.x - d1.x, d2.y - d1.y, d2.z - d1.z, d2.w - d1.w);
        v[3] = make_float4(d1.x - d3.x, d1.y - d3.y, d1.z - d3.z, d1.w - d3.w);
        Planes pl;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float4 e = v[c];
            if (SCALE) e = make_float4(mul_rn(e.x, st.s.x), mul_rn(e.y, st.s.y), mul_rn(e.z, st.s.z), mul_rn(e.w, st.s.w));
            const Split4 s = split4(e);
#pragma unroll
            for (int q = 0; q < 3; ++q) pl.q[c][q] = s.p[q];
        }
        return pl;
    };
    auto lstoreA = [&](int buf, const Planes& pl) {
#if WINO_ABL == 3
        unsigned acc_ = 0;
        for (int c = 0; c < 4; ++c) for (int q = 0; q < 3; ++q) acc_ ^= pl.q[c][q].x ^ pl.q[c][q].y;
        if (acc_ == 0x12345u) *reinterpret_cast<unsigned*>(smem + buf * 12 * PL + a_lds) = acc_;
        return;
#endif
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int q = 0; q < 3; ++q) *reinterpret_cast<uint2*>(smem + buf * 12 * PL + (c * 3 + q) * PL + a_lds) = pl.q[c][q];
    };

    const int lane = t & 63, wave = t >> 6;
    const int wv = wave & 3, wh = wave >> 2;             // Winograd component, channel half of this wave
    const int li = lane & 31, lh = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    const int f_base = (wv * 3) * PL + li * RB;
    auto f_chunk = [&](int kh) { return NH == 1 ? ((lh ^ ((li >> 3) & 1)) << 4) : (((kh * 2 + lh) ^ ((li >> 2) & 3)) << 4); };
    // B operand registers of this wave: planes (wv, pl), channel rows n0 + wh*64 + b*32 + li, K half lh of a 16-channel step
    const unsigned fb_voff = (unsigned)((n0 + wh * 64 + li) * 32 + lh * 16) + (unsigned)(wv * 3) * plane_bytes;
    struct BFrag { bf16x8 f[2][3]; };
    int b_ky = 0, b_kh = 0, b_c = 0;                     // uniform walk over the 16-channel weight steps in consumption order
    auto gloadB = [&](BFrag& fb) {
        const unsigned soff = (unsigned)(((NH * b_c + b_kh) * 3 + b_ky) * p.Cout) * 32u;
#if WINO_ABL == 4
        for (int b = 0; b < 2; ++b) for (int pl = 0; pl < 3; ++pl) { typedef unsigned u4 __attribute__((ext_vector_type(4))); u4 u = {soff + fb_voff, soff * 3u, fb_voff, soff ^ 0x3f803f80u}; fb.f[b][pl] = __builtin_bit_cast(bf16x8, u); }
        if (false)
#endif
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#if WINO_ABL == 7      // every B load hits the same few lines (L1 resident): same instruction stream and waits, no L2 traffic
                fb.f[b][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                    ru, (int)((unsigned)(lane * 16) + (unsigned)(b * 32 * 32)), (int)(soff & 0x400u), 0));
#else
                fb.f[b][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                    ru, (int)(fb_voff + (unsigned)pl * plane_bytes + (unsigned)(b * 32 * 32)), (int)soff, 0));
#endif
        const int ch = (b_kh + 1) / NH;                  // carries, pure arithmetic
        b_kh = b_kh + 1 - NH * ch;
        const int cy = (b_ky + ch) / 3;
        b_ky = b_ky + ch - 3 * cy;
        b_c += cy;
    };
    auto mfmas = [&](const bf16x8 (&fa)[2][3], const BFrag& fb) {
#if WINO_ABL == 5
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int pl = 0; pl < 3; ++pl) for (int e = 0; e < 8; ++e) acc[a][b][e] += (float)fa[a][pl][e] + (float)fb.f[b][pl][e];
        return;
#endif
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][PA[q]], fb.f[b][PB[q]], acc[a][b], 0, 0, 0);
    };
    auto afrags = [&](const unsigned char* base, int kh, bf16x8 (&fa)[2][3]) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                fa[a][pl] = *reinterpret_cast<const bf16x8*>(base + f_base + pl * PL + a * 32 * RB + f_chunk(kh));
    };

    // step t: LDS[t&1] holds tile t's A planes, `fbc` the weights of its first 16 channels; tile t+1's window is in `stg`
    // (transformed + split into LDS[(t+1)&1] during this step); tile t+2's window is fetched into `ld`; the weights of
    // every 16-channel half-step are fetched one half-step ahead
    auto step = [&](int tix, Stage& ld, const Stage& stg, BFrag& fbc, BFrag& fbn) {
        const unsigned char* base = smem + (tix & 1) * 12 * PL;
        if (NH == 2) {                       // vmcnt retires in order: the weights wanted in half a step go first, the window of tile t+2 behind them
            gloadB(fbn);
            gloadA(ld);
        } else {
            gloadA(ld);
            gloadB(fbn);
        }
        if (NH == 2) __builtin_amdgcn_sched_barrier(0);   // the loads are ISSUED here: hipcc otherwise sinks them to just before their use
        bf16x8 fa[2][3];
        afrags(base, 0, fa);
        lstoreA((tix & 1) ^ 1, transform_split(stg));
        mfmas(fa, fbc);
        if (NH == 2) {
            __builtin_amdgcn_sched_barrier(0);
            gloadB(fbc);
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 fa1[2][3];
            afrags(base, 1, fa1);
            mfmas(fa1, fbn);
        }
        __syncthreads();
    };
    const int nk = 3 * (p.Cin / KS);
    BFrag fb0, fb1;
    gloadA(st0);
    gloadB(fb0);
    lstoreA(0, transform_split(st0));
    gloadA(st1);
    __syncthreads();
    int kt = 0;
    if (NH == 2) {                                       // both weight buffers return to their roles every step
        for (; kt + 1 < nk; kt += 2) {
            step(kt, st0, st1, fb0, fb1);
            step(kt + 1, st1, st0, fb0, fb1);
        }
        if (kt < nk) step(kt, st0, st1, fb0, fb1);
    } else {
        for (; kt + 1 < nk; kt += 2) {
            step(kt, st0, st1, fb0, fb1);
            step(kt + 1, st1, st0, fb1, fb0);
        }
        if (kt < nk) step(kt, st0, st1, fb0, fb1);
    }

    // ---- epilogue: the four components meet in LDS, inverse transform, fused gain / demod / bias / act / residual -----
    float* exch = reinterpret_cast<float*>(smem);       // [NH halves][4 v][64 rows][XROW]
    const bool vec = wino_vec_ok(y, resid, out_scale, bias, p.Cout);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
        (void*)y, 0, epi != EPI_GENERIC ? (int)((unsigned)p.B * (unsigned)p.IH * (unsigned)p.IW * (unsigned)p.Cout * 4u) : 0, (int)RSRC_FLAGS);
    const int er = (t >> 2) & 63, cg = t & 3, eh = t >> 8;   // this thread finishes 8 channels of pair row er, half eh
    const bool row_live = m0 + er < M;
    int pb = 0;
    int64_t opix = 0;
    if (row_live) {
        const int64_t m = m0 + er;
        const int ptx = (int)(m % W2);
        const int64_t q = m / W2;
        pb = (int)(q / H);
        opix = ((q * W) + 2 * ptx) * p.Cout;             // q = b*H + y
    }
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                exch[((wh * 4 + wv) * WP + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * XROW + li] = acc[a][hb][e];
        __syncthreads();
        if (row_live && epi != EPI_GENERIC) {            // (uniform switch; see wino_finish_fast)
            const int n = n0 + eh * 64 + hb * 32 + cg * 8;
            if (n < p.Cout) {
                const float* ex = exch + (eh * 4 * WP + er) * XROW + cg * 8;
                const unsigned yoff = (unsigned)(opix + n) * 4u, pxs = (unsigned)p.Cout * 4u;
                const float* osp = out_scale + (int64_t)pb * p.Cout + n;
                const float* bp = bias + n;
                const bool two = n + 4 < p.Cout;
                switch (epi) {
                    case EPI_PLAIN: wino_finish_fast<false, false, false>(ry, yoff, pxs, resid, osp, bp, ex, two, p.gain, p.alpha, p.act_gain, p.resid_gain); break;
                    case EPI_OS: wino_finish_fast<true, false, false>(ry, yoff, pxs, resid, osp, bp, ex, two, p.gain, p.alpha, p.act_gain, p.resid_gain); break;
                    case EPI_BA: wino_finish_fast<false, true, false>(ry, yoff, pxs, resid, osp, bp, ex, two, p.gain, p.alpha, p.act_gain, p.resid_gain); break;
                    case EPI_OS_BA: wino_finish_fast<true, true, false>(ry, yoff, pxs, resid, osp, bp, ex, two, p.gain, p.alpha, p.act_gain, p.resid_gain); break;
                    default: wino_finish_fast<true, true, true>(ry, yoff, pxs, resid, osp, bp, ex, two, p.gain, p.alpha, p.act_gain, p.resid_gain); break;
                }
            }
        } else if (row_live) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int cl = cg * 8 + g * 4;           // channel offset inside the 32-channel slice
                const int n = n0 + eh * 64 + hb * 32 + cl;
                if (n < p.Cout) {
                    const float* ex = exch + (eh * 4 * WP + er) * XROW + cl;
                    const float4 m0v = *reinterpret_cast<const float4*>(ex + 0 * WP * XROW);
                    const float4 m1v = *reinterpret_cast<const float4*>(ex + 1 * WP * XROW);
                    const float4 m2v = *reinterpret_cast<const float4*>(ex + 2 * WP * XROW);
                    const float4 m3v = *reinterpret_cast<const float4*>(ex + 3 * WP * XROW);
                    const float mm[4][4] = {{m0v.x, m0v.y, m0v.z, m0v.w}, {m1v.x, m1v.y, m1v.z, m1v.w},
                                            {m2v.x, m2v.y, m2v.z, m2v.w}, {m3v.x, m3v.y, m3v.z, m3v.w}};
                    wino_finish4(y, resid, out_scale, bias, p, mm, opix, pb, n, vec);
                }
            }
        }
        __syncthreads();
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Row-sharing variant of the 8-wave tile.  The kernel above walks K as (chunk, ky) and stages one input row per step: every
// input row is fetched, transformed, split and written to LDS three times (once per ky).  Here the block's 64 pair-rows are a
// 2-D patch of TR output rows x TP column pairs (TR * TP = 64); a K-step is one 16-channel chunk: the TR + 2 input rows of the
// patch are staged ONCE and the three ky sub-steps read them at row offsets 0, TP, 2 TP (still 32 consecutive LDS rows per MFMA
// operand, so the conflict-free row swizzle is unchanged).  Per 72 MFMAs of a wave: (TR + 2) / TR instead of 3 stagings per
// output row (TP = 32: 4 instead of 6 rows per 2 output rows), one barrier instead of 1.5.  Ablations on the kernel above
// (WINO_ABL): without the window loads +17 %, without transform + split +11 %, without the LDS stores +13 %.
// Needs H % TR == 0 and (W / 2) % TP == 0; everything else (weights layout, epilogue, wave roles) is the kernel above.
// (Tried for the 64-channel N tile as well -- 8 waves, one accumulator column each: 10-20 % SLOWER than the 4-wave kernel above at
// three blocks per CU (Dreal.1.conv1's input gradient 236 -> 215 TFLOP/s, E.1.conv1 145 -> 117): with half the MFMAs per staged row
// the single block per CU is latency-bound.  32 < Cout <= 64 stays on the 4-wave kernel.)
// (Tried: THREE weight register sets, set ky refilled for (chunk + 1, ky) right behind the MFMAs that read it, so every weight load
// is in flight for a whole chunk instead of one sub-step -- 242 registers, no spill, counted waits of vmcnt(17..18); 1.5-3 % SLOWER
// on the 128..512-channel layers (same box: 227.7 -> 224.4, 273.9 -> 266.3, 299.7 -> 289.7 TFLOP/s).  The weight loads cost issue
// slots and L2 bandwidth, not latency: one sub-step of prefetch distance already covers it.)
// ---------------------------------------------------------------------------------------------------------------
template <bool SCALE, bool REFLECT, int TP>
__global__ __launch_bounds__(512, 1) void conv_b3_wino2d_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                                const void* __restrict__ uplanes,
                                                                const float* __restrict__ in_scale,
                                                                const float* __restrict__ out_scale,
                                                                const float* __restrict__ bias,
                                                                const float* __restrict__ resid, ideas_conv_params p,
                                                                int tiles_n, unsigned x_bytes, unsigned plane_bytes, int ntiles, int epi) {
    constexpr int TR = WP / TP;                 // output rows of the patch
    constexpr int SR = (TR + 2) * TP;           // staged pair-rows per chunk
    constexpr int PL = SR * ROWB;               // bytes per plane
    constexpr int BN = 2 * WN;                  // channels per block
    constexpr int BUFB = 12 * PL;
    constexpr int XB = 2 * 4 * WP * XROW * 4;   // exchange buffer of the epilogue
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUFB > XB ? 2 * BUFB : XB];

    const int t = threadIdx.x;
    const int H = p.IH, W = p.IW, W2 = W >> 1;
    const int tpr = W2 / TP, tpi = (H / TR) * tpr;          // patches per row block, per image
    // PERSISTENT blocks: block b runs tiles b, b + G, b + 2G, ... of the XCD-banded order (G = gridDim.x, a multiple of 8, so all
    // of a block's tiles lie in its own XCD's band); the first window and weight loads of the next tile are issued BEFORE the
    // epilogue of the current one, so their latency (and the block launch) no longer sits in front of every tile's K loop -- the
    // 8-chunk layers at 256x256 spend 40 us per tile, ~3 of them in that prologue.
    int pb, y0, px0, n0;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)x_bytes, (int)RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc((void*)uplanes, 0, (int)(12u * plane_bytes), (int)RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc((void*)in_scale, 0, SCALE ? p.B * p.Cin * 4 : 0, (int)RSRC_FLAGS);
    // (the fast epilogues store through a buffer resource with 32-bit offsets; num_records = 0 when they are not in use)
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
        (void*)y, 0, epi != EPI_GENERIC ? (int)((unsigned)p.B * (unsigned)H * (unsigned)W * (unsigned)p.Cout * 4u) : 0, (int)RSRC_FLAGS);

    // ---- staging: thread = (staged pair-row r, channel quad kq of the chunk); threads past SR * 4 idle in the staging parts ----
    const int r = t >> 2, kq = t & 3;
    const bool stager = r < SR;
    unsigned coff[4], cmask[4], sbase;           // byte offsets of the four window columns; mask = 0xffffffff in the zero padding
    auto set_tile = [&](int d) {
        const int swz = xcd_swizzle(d, ntiles);
        const int tile_n = swz % tiles_n, tile_m = swz / tiles_n;
        pb = tile_m / tpi;
        const int prem = tile_m - pb * tpi;
        y0 = (prem / tpr) * TR;
        px0 = (prem % tpr) * TP;
        n0 = tile_n * BN;
        const int ir = stager ? r / TP : 0, ptx = r % TP;
        int iy = y0 - 1 + ir;
        bool rok = true;
        if (REFLECT) iy = reflect_coord(iy, H); else rok = iy >= 0 && iy < H;
        const unsigned rowb = (unsigned)(((pb * H + (rok ? iy : 0)) * W) * p.Cin + kq * 4) * 4u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int ix = 2 * (px0 + ptx) - 1 + j;
            bool cok = true;
            if (REFLECT) ix = reflect_coord(ix, W); else cok = ix >= 0 && ix < W;
            coff[j] = rowb + (unsigned)((cok ? ix : 0) * p.Cin) * 4u;
            cmask[j] = (rok && cok) ? 0u : 0xffffffffu;
        }
        sbase = (unsigned)(pb * p.Cin + kq * 4) * 4u;
    };
    const int a_lds = r * ROWB + ((kq * 8) ^ (((r >> 3) & 1) << 4));

    struct Stage { float4 d[4], s; };
    Stage st0, st1;
    int k_ci = 0;
    auto gloadA = [&](Stage& st) {
        const unsigned so = (unsigned)k_ci * 4u;
#pragma unroll
        for (int j = 0; j < 4; ++j) st.d[j] = buffer_load4(rx, (coff[j] + so) | cmask[j], 0);   // padding: out of range -> zeros
        if (SCALE) st.s = buffer_load4(rs_, sbase, so);
        k_ci += BK;
    };
    struct Planes { uint2 q[4][3]; };
    auto transform_split = [&](const Stage& st) {
        const float4 d0 = st.d[0], d1 = st.d[1], d2 = st.d[2], d3 = st.d[3];
        float4 v[4];
        v[0] = make_float4(d0.x - d2.x, d0.y - d2.y, d0.z - d2.z, d0.w - d2.w);
        v[1] = make_float4(d1.x + d2.x, d1.y + d2.y, d1.z + d2.z, d1.w + d2.w);
        v[2] = make_float4(d2.x - d1.x, d2.y - d1.y, d2.z - d1.z, d2.w - d1.w);
        v[3] = make_float4(d1.x - d3.x, d1.y - d3.y, d1.z - d3.z, d1.w - d3.w);
        Planes pl;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float4 e = v[c];
            if (SCALE) e = make_float4(mul_rn(e.x, st.s.x), mul_rn(e.y, st.s.y), mul_rn(e.z, st.s.z), mul_rn(e.w, st.s.w));
            const Split4 s = split4(e);
#pragma unroll
            for (int q = 0; q < 3; ++q) pl.q[c][q] = s.p[q];
        }
        return pl;
    };
    auto lstoreA = [&](int buf, const Planes& pl) {
        if (SR * 4 < 512 && !stager) return;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int q = 0; q < 3; ++q) *reinterpret_cast<uint2*>(smem + buf * BUFB + (c * 3 + q) * PL + a_lds) = pl.q[c][q];
    };

    const int lane = t & 63, wave = t >> 6;
    const int wv = wave & 3, wh = wave >> 2;
    const int li = lane & 31, lh = lane >> 5;
    f32x16 acc[2][2];
    // operand rows of sub-step ky: a*32 + ky*TP + li; the swizzle bit is bit 3 of that row (TP is a multiple of 8)
    int f_off[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int row = ky * TP + li;
        f_off[ky] = (wv * 3) * PL + row * ROWB + ((lh ^ ((row >> 3) & 1)) << 4);
    }
    unsigned fb_voff = 0;                                 // (set per tile: depends on the tile's first channel)
    struct BFrag { bf16x8 f[2][3]; };
    int b_step = 0;                                       // (chunk * 3 + ky), consumption order
    auto gloadB = [&](BFrag& fb) {
        const unsigned soff = (unsigned)(b_step * p.Cout) * 32u;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                fb.f[b][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                    ru, (int)(fb_voff + (unsigned)pl * plane_bytes + (unsigned)(b * 32 * 32)), (int)soff, 0));
        ++b_step;
    };
    auto mfmas = [&](const bf16x8 (&fa)[2][3], const BFrag& fb) {
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][PA[q]], fb.f[b][PB[q]], acc[a][b], 0, 0, 0);
    };
    auto afrags = [&](const unsigned char* base, int ky, bf16x8 (&fa)[2][3]) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                fa[a][pl] = *reinterpret_cast<const bf16x8*>(base + f_off[ky] + pl * PL + a * 32 * ROWB);
    };

    // chunk c: LDS[c&1] holds its planes, `fbA` the weights of (c, ky 0); chunk c+1's window is in `stg` (transformed + split into
    // LDS[(c+1)&1] under the first MFMAs), chunk c+2's window is fetched into `ld`; the weights of every sub-step are fetched one
    // sub-step ahead.  The sets swap roles every chunk (3 sub-steps), so the loop body is two chunks.
    auto step_plain = [&](int c, Stage& ld, const Stage& stg, BFrag& fbA, BFrag& fbB) {      // TP < 32 (234-238 registers already)
        const unsigned char* base = smem + (c & 1) * BUFB;
        gloadB(fbB);                                       // (c, ky 1): first in the vmcnt queue, wanted soonest
        gloadA(ld);
        __builtin_amdgcn_sched_barrier(0);
        {
            bf16x8 fa[2][3];
            afrags(base, 0, fa);
            lstoreA((c & 1) ^ 1, transform_split(stg));
            mfmas(fa, fbA);
        }
        __builtin_amdgcn_sched_barrier(0);
        gloadB(fbA);                                       // (c, ky 2)
        __builtin_amdgcn_sched_barrier(0);
        {
            bf16x8 fa[2][3];
            afrags(base, 1, fa);
            mfmas(fa, fbB);
        }
        __builtin_amdgcn_sched_barrier(0);
        gloadB(fbB);                                       // (c + 1, ky 0)
        __builtin_amdgcn_sched_barrier(0);
        {
            bf16x8 fa[2][3];
            afrags(base, 2, fa);
            mfmas(fa, fbA);
        }
        __syncthreads();
    };
    // TP = 32 (207 registers before, 235-245 with this; the narrower patches would spill):
    // The operand reads run one sub-step ahead of the MFMAs that use them (two register sets, `faX` = the fragments of (c, ky 0),
    // already loaded): a sub-step no longer starts with six ds_read_b128 and a wait for them.  The chunk's ONE barrier sits in front
    // of the last sub-step's MFMAs -- by then every wave has finished reading LDS[c&1] (its ky 2 fragments have arrived) and written
    // its part of LDS[(c+1)&1] -- so the first fragments of chunk c+1 are read under those MFMAs as well.  Same box: 263 -> 269,
    // 304 -> 312, 318 -> 328 TFLOP/s on the 128 / 256 / 512-channel layers.
    auto step = [&](int c, Stage& ld, const Stage& stg, BFrag& fbA, BFrag& fbB, bf16x8 (&faX)[2][3], bf16x8 (&faY)[2][3], bool more_chunks) {
        const unsigned char* base = smem + (c & 1) * BUFB;
        gloadB(fbB);                                       // (c, ky 1): first in the vmcnt queue, wanted soonest
        gloadA(ld);
        __builtin_amdgcn_sched_barrier(0);
        afrags(base, 1, faY);
        __builtin_amdgcn_sched_barrier(0);
        lstoreA((c & 1) ^ 1, transform_split(stg));
        mfmas(faX, fbA);
        __builtin_amdgcn_sched_barrier(0);
        gloadB(fbA);                                       // (c, ky 2)
        afrags(base, 2, faX);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(faY, fbB);
        __builtin_amdgcn_sched_barrier(0);
        gloadB(fbB);                                       // (c + 1, ky 0)
        __syncthreads();
        if (more_chunks) afrags(smem + ((c & 1) ^ 1) * BUFB, 0, faY);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(faX, fbA);
    };
    const int nc = p.Cin / BK;
    BFrag fb0, fb1;
    auto begin_tile = [&](int d) {                         // tile d: addresses, then its first window and weight loads
        set_tile(d);
        fb_voff = (unsigned)((n0 + wh * 64 + li) * 32 + lh * 16) + (unsigned)(wv * 3) * plane_bytes;
        k_ci = 0;
        b_step = 0;
        gloadA(st0);
        gloadB(fb0);
    };
    float* exch = reinterpret_cast<float*>(smem);
    const bool vec = wino_vec_ok(y, resid, out_scale, bias, p.Cout);
    const int er = (t >> 2) & 63, cg = t & 3, eh = t >> 8;
    int d = blockIdx.x;
    begin_tile(d);
    for (;;) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    lstoreA(0, transform_split(st0));
    gloadA(st1);
    __syncthreads();
    int c = 0;
    if constexpr (TP == 32) {
        bf16x8 faP[2][3], faQ[2][3];
        afrags(smem, 0, faP);
        for (; c + 1 < nc; c += 2) {
            step(c, st0, st1, fb0, fb1, faP, faQ, true);
            step(c + 1, st1, st0, fb1, fb0, faQ, faP, c + 2 < nc);
        }
        if (c < nc) step(c, st0, st1, fb0, fb1, faP, faQ, false);
    } else {
        for (; c + 1 < nc; c += 2) {
            step_plain(c, st0, st1, fb0, fb1);
            step_plain(c + 1, st1, st0, fb1, fb0);
        }
        if (c < nc) step_plain(c, st0, st1, fb0, fb1);
    }

    // ---- epilogue: as above; pair-row er of the patch = output row y0 + er / TP, pair px0 + er % TP.  The next tile's first
    // loads go out first (the coordinates of THIS tile are copied before set_tile overwrites them) --------------------------------
    const int e_pb = pb, e_n0 = n0;
    const int64_t opix = (((int64_t)pb * H + y0 + er / TP) * W + 2 * (px0 + er % TP)) * p.Cout;
    d += gridDim.x;
    const bool more = d < ntiles;
    if (more) begin_tile(d);
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                exch[((wh * 4 + wv) * WP + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * XROW + li] = acc[a][hb][e];
        __syncthreads();
        if (epi != EPI_GENERIC) {
            // (the launcher checked: 16-byte addressable rows, Cout % 4 == 0, y below 4 GB, 0 <= alpha <= 1, no accumulate)
            const int n = e_n0 + eh * 64 + hb * 32 + cg * 8;
            if (n < p.Cout) {
                const float* ex = exch + (eh * 4 * WP + er) * XROW + cg * 8;
                const unsigned yoff = (unsigned)(opix + n) * 4u, pxs = (unsigned)p.Cout * 4u;
                const float* osp = out_scale + (int64_t)e_pb * p.Cout + n;
                const float* bp = bias + n;
                const bool two = n + 4 < p.Cout;
                switch (epi) {                           // uniform
                    case EPI_PLAIN: wino_finish_fast<false, false, false>(ry, yoff, pxs, resid, osp, bp, ex, two, p.gain, p.alpha, p.act_gain, p.resid_gain); break;
                    case EPI_OS: wino_finish_fast<true, false, false>(ry, yoff, pxs, resid, osp, bp, ex, two, p.gain, p.alpha, p.act_gain, p.resid_gain); break;
                    case EPI_BA: wino_finish_fast<false, true, false>(ry, yoff, pxs, resid, osp, bp, ex, two, p.gain, p.alpha, p.act_gain, p.resid_gain); break;
                    case EPI_OS_BA: wino_finish_fast<true, true, false>(ry, yoff, pxs, resid, osp, bp, ex, two, p.gain, p.alpha, p.act_gain, p.resid_gain); break;
                    default: wino_finish_fast<true, true, true>(ry, yoff, pxs, resid, osp, bp, ex, two, p.gain, p.alpha, p.act_gain, p.resid_gain); break;
                }
            }
        } else {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int cl = cg * 8 + g * 4;
            const int n = e_n0 + eh * 64 + hb * 32 + cl;
            if (n < p.Cout) {
                const float* ex = exch + (eh * 4 * WP + er) * XROW + cl;
                const float4 m0v = *reinterpret_cast<const float4*>(ex + 0 * WP * XROW);
                const float4 m1v = *reinterpret_cast<const float4*>(ex + 1 * WP * XROW);
                const float4 m2v = *reinterpret_cast<const float4*>(ex + 2 * WP * XROW);
                const float4 m3v = *reinterpret_cast<const float4*>(ex + 3 * WP * XROW);
                const float mm[4][4] = {{m0v.x, m0v.y, m0v.z, m0v.w}, {m1v.x, m1v.y, m1v.z, m1v.w},
                                        {m2v.x, m2v.y, m2v.z, m2v.w}, {m3v.x, m3v.y, m3v.z, m3v.w}};
                wino_finish4(y, resid, out_scale, bias, p, mm, opix, e_pb, n, vec);
            }
        }
        }
        __syncthreads();
    }
    if (!more) break;
    }
}

}  // namespace

extern "C" int ideas_b3_wino_supported(const ideas_conv_params* p) {
    if (!p) return 0;
    return p->TY == 3 && p->TX == 3 && p->sy == 1 && p->sx == 1 && p->dy == 1 && p->dx == 1 && p->offy == -1 && p->offx == -1 &&
           p->OH == p->IH && p->OW == p->IW && p->YH == p->IH && p->YW == p->IW && p->osy == 1 && p->osx == 1 && p->ooy == 0 &&
           p->oox == 0 && (p->IW & 1) == 0 && p->Cin % 16 == 0 && (!p->reflect || (p->IH >= 2 && p->IW >= 2)) &&
           (int64_t)p->B * p->IH * p->IW * p->Cin * 4 < 0xffffffffLL && (int64_t)p->Cin * p->Cout * 72 < 0xffffffffLL;
}

void ideas_b3_wino_split_batched(const ideas_prep_desc* tbl, int n, int blocks, hipStream_t stream) {
    hipLaunchKernelGGL(wino_split_weights_batched_kernel, dim3(blocks), dim3(256), 0, stream, tbl, n);
}

extern "C" int ideas_b3_wino_split_weights(void* planes, const void* w, int N, int C, int64_t sn, int64_t sky, int64_t skx,
                                           int64_t sc, int64_t base, void* stream_) {
    if (!planes || !w) return IDEAS_E_NULL;
    if (N <= 0 || C <= 0) return IDEAS_E_SHAPE;
    if (C % 16 || !ideas_aligned16(planes)) return IDEAS_E_ALIGN;
    const int64_t total = (int64_t)N * 3 * (C / 4);
    const int blocks = (int)(total / 256 + 1 < 2048 ? total / 256 + 1 : 2048);
    hipLaunchKernelGGL(wino_split_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, (uint2*)planes,
                       (const float*)w, N, C, sn, sky, skx, sc, base);
    return ideas_launch_status();
}

// called by ideas_conv3x3_wino for dtype IDEAS_F32_B3 (umat = planes of ideas_b3_wino_split_weights)
int ideas_b3_wino_fwd(void* y, const void* x, const void* uplanes, const float* in_scale, const float* out_scale,
                      const float* bias, const void* resid, const ideas_conv_params* p, hipStream_t stream) {
    const int64_t M = (int64_t)p->B * p->IH * (p->IW / 2);
    const int64_t tm = ideas_cdiv(M, WP);
    const bool wide = p->Cin % 32 == 0 && p->Cout > 64;     // 8-wave 64 x 128 tile, K-step 32
    const int tn = (int)ideas_cdiv(p->Cout, wide ? 2 * WN : WN);
    if (tm * tn > 0x7fffffffLL) return IDEAS_E_SHAPE;
    const unsigned x_bytes = (unsigned)((int64_t)p->B * p->IH * p->IW * p->Cin * 4);
    const unsigned plane_bytes = (unsigned)((int64_t)3 * p->Cin * p->Cout * 2);     // one (v, plane): 3*Cin/16 steps x Cout x 32 B
    // row-sharing patch kernel: 8-wave tile shapes whose image divides into TR x TP patches (IDEAS_B3_WINO2D=0: the kernel above)
    const char* e2d = getenv("IDEAS_B3_WINO2D");           // read per call: tests toggle it in-process
    const bool use2d = !(e2d && e2d[0] == '0');
    const int W2 = p->IW / 2;
    const int TPsel = (W2 % 32 == 0) ? 32 : (W2 % 16 == 0) ? 16 : (W2 % 8 == 0) ? 8 : 0;
    // epilogue configuration (wino_finish_fast): IDEAS_B3_WINO_EPI=0 keeps the flag-testing tail everywhere (A/B measurements)
    int epi = EPI_GENERIC;
    {
        const char* eepi = getenv("IDEAS_B3_WINO_EPI");
        const bool vec = p->Cout % 4 == 0 && ((((uintptr_t)y | (uintptr_t)resid | (uintptr_t)out_scale | (uintptr_t)bias) & 15) == 0);
        if (!(eepi && eepi[0] == '0') && vec && !p->accumulate && (int64_t)p->B * p->IH * p->IW * p->Cout * 4 < 0xffffffffLL) {
            const bool ba = bias && p->act && p->alpha >= 0.f && p->alpha <= 1.f;
            if (!bias && !p->act && !resid) epi = out_scale ? EPI_OS : EPI_PLAIN;
            else if (ba && !resid) epi = out_scale ? EPI_OS_BA : EPI_BA;
            else if (ba && resid && out_scale) epi = EPI_OS_BA_RS;
        }
    }
    if (wide && use2d && TPsel && p->IH % (WP / TPsel) == 0 && p->Cin % 16 == 0) {
        auto go2 = [&](auto sc, auto rf, auto tp) {
            const int64_t nt = tm * tn;                       // persistent: one block per CU (98 KB of LDS each), a multiple of 8
            const unsigned grid = (unsigned)(nt < 256 ? nt : 256);
            hipLaunchKernelGGL((conv_b3_wino2d_kernel<decltype(sc)::value, decltype(rf)::value, decltype(tp)::value>),
                               dim3(grid), dim3(512), 0, stream, (float*)y, (const float*)x, uplanes, in_scale, out_scale,
                               bias, (const float*)resid, *p, tn, x_bytes, plane_bytes, (int)nt, epi);
        };
        using T = std::true_type;
        using F = std::false_type;
        auto go1 = [&](auto tp) {
            if (in_scale) { if (p->reflect) go2(T{}, T{}, tp); else go2(T{}, F{}, tp); }
            else { if (p->reflect) go2(F{}, T{}, tp); else go2(F{}, F{}, tp); }
        };
        if (TPsel == 32) go1(std::integral_constant<int, 32>{});
        else if (TPsel == 16) go1(std::integral_constant<int, 16>{});
        else go1(std::integral_constant<int, 8>{});
        return ideas_launch_status();
    }
    auto go = [&](auto sc, auto rf) {
        if (wide)
            hipLaunchKernelGGL((conv_b3_wino_kernel<decltype(sc)::value, decltype(rf)::value, 2>), dim3((unsigned)(tm * tn)),
                               dim3(512), 0, stream, (float*)y, (const float*)x, uplanes, in_scale, out_scale, bias,
                               (const float*)resid, *p, tn, x_bytes, plane_bytes, epi);
        else
            hipLaunchKernelGGL((conv_b3_wino_kernel<decltype(sc)::value, decltype(rf)::value, 1>), dim3((unsigned)(tm * tn)),
                               dim3(256), 0, stream, (float*)y, (const float*)x, uplanes, in_scale, out_scale, bias,
                               (const float*)resid, *p, tn, x_bytes, plane_bytes, epi);
    };
    using T = std::true_type;
    using F = std::false_type;
    if (in_scale) { if (p->reflect) go(T{}, T{}); else go(T{}, F{}); }
    else { if (p->reflect) go(F{}, T{}); else go(F{}, F{}); }
    return ideas_launch_status();
}
