// 3x3 stride-1 pad-1 convolution with a 1-D Winograd F(2,3) transform along x, on the f32 MFMA pipe, NHWC.
//
// The 3x3/s1 layers (Generator conv2s and same-resolution conv1s, every ResBlock conv1 of E / Dreal / Dco and
// their input gradients) carry ~2/3 of the step's FLOPs and the step is MFMA-bound, so the lever left once the
// implicit GEMM sits at ~0.75 of peak is doing fewer multiplies.  F(2,3) along the row axis:
//
//     out[y, 2t  ] = M0 + M1 + M2          M_v[y,t,o] = sum_{ky,ci} U_v[o][ky][ci] * V_v[y+ky-1, t, ci]
//     out[y, 2t+1] = M1 - M2 - M3
//     V0 = d0 - d2,  V1 = d1 + d2,  V2 = d2 - d1,  V3 = d1 - d3        d_j = x[., 2t-1+j, ci]   (zero / mirrored outside)
//     U0 = w0,  U1 = (w0 + w1 + w2)/2,  U2 = (w0 - w1 + w2)/2,  U3 = w2  w_kx = w[o][ky][kx][ci]
//
// i.e. four GEMMs (v = 0..3) with M = B*H*W/2 "column pairs", N = Cout, K = 3*Cin: 6 multiplies per output instead
// of 9 (1.5x fewer MFMAs), exact in exact arithmetic, f32 error of the same class as the direct kernel (the
// transforms only add; the 1/2 in U is exact).  The input transform is two vector subtractions on the operand's way
// into LDS, the output transform two adds in the epilogue on values a lane already holds (all four v of an output
// element live in the same lane), so nothing extra touches HBM.
//
// Block = 256 threads = 4 waves, tile 128 column-pairs (256 output pixels) x 64 channels x K-step 8.  Wave w owns
// rows [32w, 32w+32) x 64 channels x 4 v = 8 accumulators of 32x32.  LDS rows are 8 floats with an XOR slot swizzle
// ((row>>3)&1) instead of padding -> conflict-free ds_read_b128, 48 KB double-buffered.  Same pipeline as
// conv_igemm: one basic block per K-step, loads of step t+2 issued while step t computes.
#include "common.hpp"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int WBM = 128;  // column pairs per block
constexpr int WBN = 64;   // output channels per block
constexpr int WBK = 8;    // K depth per step

__device__ __forceinline__ float4 sub4(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 mulv4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 keepv4(bool ok, float4 v) {
    return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}
__device__ __forceinline__ int wino_swizzle(int row, int slot) { return (slot ^ ((row >> 3) & 1)) * 4; }

__device__ __forceinline__ int xcd_swz(int bid, int nblk) {
    const int q = nblk >> 3, rem = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
}

template <bool SCALE, bool REFLECT>
__global__ __launch_bounds__(256, 2) void conv3x3_wino_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                              const float* __restrict__ umat,
                                                              const float* __restrict__ in_scale,
                                                              const float* __restrict__ out_scale,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ resid, ideas_conv_params p,
                                                              int tiles_n) {
    constexpr int A_FLOATS = 4 * WBM * WBK, B_FLOATS = 4 * WBN * WBK;
    __shared__ __attribute__((aligned(16))) float smem[2 * (A_FLOATS + B_FLOATS)];
    float* As = smem;                  // [2][4][WBM][8]
    float* Bs = smem + 2 * A_FLOATS;   // [2][4][WBN][8]

    const int t = threadIdx.x;
    const int H = p.IH, W = p.IW, W2 = W >> 1;
    const int64_t M = (int64_t)p.B * H * W2;
    const int K = 3 * p.Cin;
    const int swz = xcd_swz(blockIdx.x, gridDim.x);
    const int tile_n = swz % tiles_n, tile_m = swz / tiles_n;
    const int64_t m0 = (int64_t)tile_m * WBM;
    const int n0 = tile_n * WBN;

    // ---- A staging: thread = (row r, float4 slot kq) -----------------------------------------------
    const int ar = t >> 1, akq = t & 1;
    const int64_t am = m0 + ar;
    const bool a_rowok = am < M;
    int a_b, a_y, a_tx;
    {
        const int64_t mm = a_rowok ? am : 0;
        a_tx = (int)(mm % W2);
        const int64_t q = mm / W2;
        a_y = (int)(q % H);
        a_b = (int)(q / H);
    }
    const int64_t a_base = (int64_t)a_b * H * W * p.Cin;
    int ax[4];
    bool axok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int ix = 2 * a_tx - 1 + j;
        if (REFLECT) { ix = reflect_coord(ix, W); axok[j] = true; }
        else { axok[j] = ix >= 0 && ix < W; }
        ax[j] = axok[j] ? ix : 0;
    }
    int k_ky = 0, k_ci = akq * 4;   // walker of this thread's K column (Cin % 8 == 0: a step never straddles ky)
    // ---- B staging: two (v, n, kq) items per thread ---------------------------------------------------
    const float* b_ptr[2];
    bool b_ok[2];
    int b_dst[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int item = t + 256 * j;
        const int v = item >> 7, n = (item >> 1) & 63, kq = item & 1;
        b_ok[j] = n0 + n < p.Cout;
        b_ptr[j] = umat + ((int64_t)v * p.Cout + (b_ok[j] ? n0 + n : 0)) * K + kq * 4;
        b_dst[j] = (v * WBN + n) * WBK + wino_swizzle(n, kq);
    }

    float4 rd[4], rs, rb[2];
    bool okd[4], kval;
    auto gload = [&](int kt) {
        kval = k_ky < 3;
        int iy = a_y + k_ky - 1;
        bool rowok = a_rowok && kval;
        if (REFLECT) iy = reflect_coord(iy, H);
        else rowok = rowok && iy >= 0 && iy < H;
        const int64_t rowoff = a_base + (int64_t)(rowok ? iy : 0) * W * p.Cin + k_ci;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            okd[j] = rowok && axok[j];
            rd[j] = *reinterpret_cast<const float4*>(x + (okd[j] ? rowoff + (int64_t)ax[j] * p.Cin : 0));
        }
        if (SCALE) rs = *reinterpret_cast<const float4*>(in_scale + (int64_t)a_b * p.Cin + k_ci);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            rb[j] = *reinterpret_cast<const float4*>((b_ok[j] && kval) ? b_ptr[j] + (int64_t)kt * WBK : umat);
        k_ci += WBK;
        const bool wrap = k_ci >= p.Cin;
        k_ci -= wrap ? p.Cin : 0;
        k_ky += wrap ? 1 : 0;
    };
    // the predicate of the B rows for the tile held in rb (kval belongs to the same gload)
    auto lstore = [&](int buf) {
        float4 d0 = keepv4(okd[0], rd[0]), d1 = keepv4(okd[1], rd[1]), d2 = keepv4(okd[2], rd[2]), d3 = keepv4(okd[3], rd[3]);
        float4 v0 = sub4(d0, d2), v1 = add4(d1, d2), v2 = sub4(d2, d1), v3 = sub4(d1, d3);
        if (SCALE) { v0 = mulv4(v0, rs); v1 = mulv4(v1, rs); v2 = mulv4(v2, rs); v3 = mulv4(v3, rs); }
        float* a = As + buf * A_FLOATS + ar * WBK + wino_swizzle(ar, akq);
        *reinterpret_cast<float4*>(a + 0 * WBM * WBK) = v0;
        *reinterpret_cast<float4*>(a + 1 * WBM * WBK) = v1;
        *reinterpret_cast<float4*>(a + 2 * WBM * WBK) = v2;
        *reinterpret_cast<float4*>(a + 3 * WBM * WBK) = v3;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            *reinterpret_cast<float4*>(Bs + buf * B_FLOATS + b_dst[j]) = keepv4(b_ok[j] && kval, rb[j]);
    };

    const int lane = t & 63, wave = t >> 6;
    const int li = lane & 31, lh = lane >> 5;
    f32x16 acc[4][2];
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[v][b][r] = 0.f;

    const int a_row = wave * 32 + li;
    const int a_off = a_row * WBK + wino_swizzle(a_row, lh);
    int b_off[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) b_off[b] = (b * 32 + li) * WBK + wino_swizzle(b * 32 + li, lh);

    const int nk = K / WBK;
    gload(0);
    lstore(0);
    gload(1);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        lstore(buf ^ 1);
        gload(kt + 2);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float4 fa = *reinterpret_cast<const float4*>(As + buf * A_FLOATS + v * WBM * WBK + a_off);
            float4 fb[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) fb[b] = *reinterpret_cast<const float4*>(Bs + buf * B_FLOATS + v * WBN * WBK + b_off[b]);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const float av = kk == 0 ? fa.x : kk == 1 ? fa.y : kk == 2 ? fa.z : fa.w;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const float bv = kk == 0 ? fb[b].x : kk == 1 ? fb[b].y : kk == 2 ? fb[b].z : fb[b].w;
                    acc[v][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[v][b], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue: inverse transform in registers, then the usual gain / demod / bias / act / residual --------
    int64_t* row_off = reinterpret_cast<int64_t*>(smem);
    int* row_b = reinterpret_cast<int*>(smem + 2 * WBM);
    if (t < WBM) {
        const int64_t m = m0 + t;
        int64_t off = -1;
        int b = 0;
        if (m < M) {
            const int tx = (int)(m % W2);
            const int64_t q = m / W2;
            const int yy = (int)(q % H);
            b = (int)(q / H);
            off = (((int64_t)b * H + yy) * W + 2 * tx) * p.Cout;
        }
        row_off[t] = off;
        row_b[t] = b;
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int n = n0 + b * 32 + li;
        if (n >= p.Cout) continue;
        const float bvv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int64_t off = row_off[row];
            if (off < 0) continue;
            const float m0v = acc[0][b][r], m1v = acc[1][b][r], m2v = acc[2][b][r], m3v = acc[3][b][r];
            float o[2];
            o[0] = (m0v + m1v) + m2v;
            o[1] = (m1v - m2v) - m3v;
            const float os = out_scale ? out_scale[(int64_t)row_b[row] * p.Cout + n] : 1.f;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float v = mul_rn(o[e], p.gain);
                if (out_scale) v = mul_rn(v, os);
                v = mul_then_add(v, 1.0f, bvv);
                if (p.act) v = (v > 0.f ? v : v * p.alpha) * p.act_gain;
                const int64_t yi = off + (int64_t)e * p.Cout + n;
                if (resid) v = (v + resid[yi]) * p.resid_gain;
                if (p.accumulate) y[yi] += v; else y[yi] = v;
            }
        }
    }
}

// =====================================================================================================
// weight gradient in the Winograd domain:
//     dU_v[o][ky][ci] += gain * sum_{b,y,t} dM_v[b,y,t,o] * V_v[b, y+ky-1, t, ci]
//     dM0 = g0, dM1 = g0 + g1, dM2 = g0 - g1, dM3 = -g1      g_e = gy[b, y, 2t+e, o]  (x out_scale)
// (the host folds dU back to the 3x3 taps: dw0 = dU0 + (dU1+dU2)/2, dw1 = (dU1-dU2)/2, dw2 = (dU1+dU2)/2 + dU3).
// Same structure as conv_wgrad_kernel (pixel-major LDS tiles, ds_read_b32 operands, split-K + atomics), with the
// reduction running over column PAIRS and four accumulator sets; tile 64 (o) x 128 (ky,ci) x 8 pairs.
// =====================================================================================================
constexpr int GBM = 64, GBN = 128, GBK = 8;

template <bool SCALE, bool REFLECT, bool WIDEW>
__global__ __launch_bounds__(256, 2) void conv3x3_wino_wgrad_kernel(float* __restrict__ gu, const float* __restrict__ gy,
                                                                    const float* __restrict__ x,
                                                                    const float* __restrict__ in_scale,
                                                                    const float* __restrict__ out_scale,
                                                                    ideas_conv_params p, int tiles_n,
                                                                    int64_t pairs_per_split) {
    constexpr int LDM = GBM + 4, LDN = GBN + 4;
    constexpr int G_FLOATS = 4 * GBK * LDM, X_FLOATS = 4 * GBK * LDN;
    __shared__ __attribute__((aligned(16))) float smem[2 * (G_FLOATS + X_FLOATS)];
    float* Gs = smem;                  // [2][4][GBK][LDM]
    float* Xs = smem + 2 * G_FLOATS;   // [2][4][GBK][LDN]

    const int t = threadIdx.x;
    const int H = p.IH, W = p.IW, W2 = W >> 1;
    const int Ktot = 3 * p.Cin;
    const int64_t P2 = (int64_t)p.B * H * W2;
    const int tile_n = blockIdx.x % tiles_n, tile_m = blockIdx.x / tiles_n;
    const int o0 = tile_m * GBM, n0 = tile_n * GBN;
    const int64_t pbeg = (int64_t)blockIdx.y * pairs_per_split;
    const int64_t pend = (pbeg + pairs_per_split < P2) ? pbeg + pairs_per_split : P2;
    if (pbeg >= pend) return;

    struct Walk { int b, y, tx, left; };
    auto walk_init = [&](int pr) {
        Walk wk;
        const int64_t pp = pbeg + pr;
        wk.left = (int)(pend - pp);
        const int64_t q = pp / W2;
        wk.tx = (int)(pp - q * W2);
        wk.b = (int)(q / H);
        wk.y = (int)(q - (int64_t)wk.b * H);
        return wk;
    };
    auto walk_step = [&](Walk& wk) {
        wk.left -= GBK;
        wk.tx += GBK;
        if (WIDEW) {   // W/2 >= GBK: at most one row boundary per step -> selects only, the K loop stays one basic block
            const bool wrap = wk.tx >= W2;
            wk.tx -= wrap ? W2 : 0;
            wk.y += wrap ? 1 : 0;
            const bool wrap2 = wk.y == H;
            wk.y = wrap2 ? 0 : wk.y;
            wk.b += wrap2 ? 1 : 0;
        } else {
            while (wk.tx >= W2) {
                wk.tx -= W2;
                if (++wk.y == H) { wk.y = 0; ++wk.b; }
            }
        }
    };
    // G items: threads 0..127 -> (pair row pr = t/16, channel quad oq = t%16)
    const bool g_active = t < GBK * (GBM / 4);
    const int g_pr = (t >> 4) & (GBK - 1), g_oq = t & 15;
    const int g_o = o0 + g_oq * 4;
    const bool g_ok = g_active && g_o < p.Cout;
    Walk gwk = walk_init(g_pr);
    // X items: every thread -> (pair row pr = t/32, k quad kq = t%32)
    const int x_pr = t >> 5, x_kq = t & 31;
    const int x_k = n0 + x_kq * 4;
    const bool x_ok = x_k < Ktot;
    const int x_ky = (x_ok ? x_k : 0) / p.Cin;
    const int x_ci = (x_ok ? x_k : 0) - x_ky * p.Cin;
    Walk xwk = walk_init(x_pr);

    float4 rg[2], rgs, rx[4], rxs;
    bool okg, okx[4];
    auto gload = [&]() {
        {
            const Walk wk = gwk;
            okg = g_ok && wk.left > 0;
            const int64_t off = okg ? (((int64_t)wk.b * H + wk.y) * W + 2 * wk.tx) * p.Cout + g_o : 0;
            rg[0] = *reinterpret_cast<const float4*>(gy + off);
            rg[1] = *reinterpret_cast<const float4*>(gy + off + (okg ? p.Cout : 0));
            if (SCALE) rgs = *reinterpret_cast<const float4*>(out_scale + (okg ? (int64_t)wk.b * p.Cout + g_o : 0));
            walk_step(gwk);
        }
        {
            const Walk wk = xwk;
            int iy = wk.y + x_ky - 1;
            bool rowok = x_ok && wk.left > 0;
            if (REFLECT) iy = reflect_coord(iy, H);
            else rowok = rowok && iy >= 0 && iy < H;
            const int64_t rowoff = ((int64_t)wk.b * H + (rowok ? iy : 0)) * W * p.Cin + x_ci;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int ix = 2 * wk.tx - 1 + j;
                bool ok = rowok;
                if (REFLECT) ix = reflect_coord(ix, W);
                else ok = ok && ix >= 0 && ix < W;
                okx[j] = ok;
                rx[j] = *reinterpret_cast<const float4*>(x + (ok ? rowoff + (int64_t)ix * p.Cin : 0));
            }
            if (SCALE) rxs = *reinterpret_cast<const float4*>(in_scale + (rowok ? (int64_t)wk.b * p.Cin + x_ci : 0));
            walk_step(xwk);
        }
    };
    auto lstore = [&](int buf) {
        if (g_active) {
            float4 g0 = keepv4(okg, rg[0]), g1 = keepv4(okg, rg[1]);
            if (SCALE) { g0 = mulv4(g0, rgs); g1 = mulv4(g1, rgs); }
            float* dst = Gs + buf * G_FLOATS + g_pr * LDM + g_oq * 4;
            *reinterpret_cast<float4*>(dst + 0 * GBK * LDM) = g0;
            *reinterpret_cast<float4*>(dst + 1 * GBK * LDM) = add4(g0, g1);
            *reinterpret_cast<float4*>(dst + 2 * GBK * LDM) = sub4(g0, g1);
            *reinterpret_cast<float4*>(dst + 3 * GBK * LDM) = make_float4(-g1.x, -g1.y, -g1.z, -g1.w);
        }
        float4 d0 = keepv4(okx[0], rx[0]), d1 = keepv4(okx[1], rx[1]), d2 = keepv4(okx[2], rx[2]), d3 = keepv4(okx[3], rx[3]);
        float4 v0 = sub4(d0, d2), v1 = add4(d1, d2), v2 = sub4(d2, d1), v3 = sub4(d1, d3);
        if (SCALE) { v0 = mulv4(v0, rxs); v1 = mulv4(v1, rxs); v2 = mulv4(v2, rxs); v3 = mulv4(v3, rxs); }
        float* dst = Xs + buf * X_FLOATS + x_pr * LDN + x_kq * 4;
        *reinterpret_cast<float4*>(dst + 0 * GBK * LDN) = v0;
        *reinterpret_cast<float4*>(dst + 1 * GBK * LDN) = v1;
        *reinterpret_cast<float4*>(dst + 2 * GBK * LDN) = v2;
        *reinterpret_cast<float4*>(dst + 3 * GBK * LDN) = v3;
    };

    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;     // 2 x 2 waves: 32 (o) x 64 (k) each
    const int li = lane & 31, lh = lane >> 5;
    f32x16 acc[4][2];
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[v][b][r] = 0.f;

    const int64_t nsteps = (pend - pbeg + GBK - 1) / GBK;
    gload();
    lstore(0);
    gload();
    __syncthreads();
    for (int64_t s = 0; s < nsteps; ++s) {
        const int buf = (int)(s & 1);
        lstore(buf ^ 1);
        gload();
#pragma unroll
        for (int v = 0; v < 4; ++v) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int prow = lh * 4 + kk;
                const float av = Gs[buf * G_FLOATS + (v * GBK + prow) * LDM + wm * 32 + li];
                float bv[2];
#pragma unroll
                for (int b = 0; b < 2; ++b) bv[b] = Xs[buf * X_FLOATS + (v * GBK + prow) * LDN + (wn * 2 + b) * 32 + li];
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[v][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[b], acc[v][b], 0, 0, 0);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int k = n0 + (wn * 2 + b) * 32 + li;
            if (k >= Ktot) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = o0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (o < p.Cout) atomicAdd(&gu[((int64_t)v * p.Cout + o) * Ktot + k], acc[v][b][r] * p.gain);
            }
        }
}

}  // namespace

// umat: [4][Cout][3][Cin] (v, o, ky, ci) transformed weights.  Geometry: x [B,H,W,Cin] -> y [B,H,W,Cout], 3x3,
// stride 1, padding 1 (zero or mirrored); requires W even, Cin % 8 == 0.
extern "C" int ideas_conv3x3_wino(void* y, const void* x, const void* umat, const float* in_scale, const float* out_scale,
                                  const float* bias, const void* resid, const ideas_conv_params* p, int dtype,
                                  void* stream_) {
    if (dtype == IDEAS_F32_B3) {   // umat = bf16 planes of ideas_b3_wino_split_weights
        if (!y || !x || !umat || !p) return IDEAS_E_NULL;
        if (!ideas_b3_wino_supported(p)) return IDEAS_E_UNSUPPORTED;
        if (!ideas_aligned16(x) || !ideas_aligned16(umat) || (in_scale && !ideas_aligned16(in_scale))) return IDEAS_E_ALIGN;
        return ideas_b3_wino_fwd(y, x, umat, in_scale, out_scale, bias, resid, p, (hipStream_t)stream_);
    }
    if (dtype != IDEAS_F32) return IDEAS_E_UNSUPPORTED;
    if (!y || !x || !umat || !p) return IDEAS_E_NULL;
    if (p->B <= 0 || p->IH <= 0 || p->IW <= 0 || p->Cin <= 0 || p->Cout <= 0) return IDEAS_E_SHAPE;
    if (p->TY != 3 || p->TX != 3 || p->sy != 1 || p->sx != 1 || p->OH != p->IH || p->OW != p->IW || p->YH != p->IH ||
        p->YW != p->IW || p->osy != 1 || p->osx != 1 || p->ooy != 0 || p->oox != 0)
        return IDEAS_E_UNSUPPORTED;
    if ((p->IW & 1) || (p->Cin % 8)) return IDEAS_E_ALIGN;
    if (p->reflect && (p->IH < 2 || p->IW < 2)) return IDEAS_E_SHAPE;
    if (!ideas_aligned16(x) || !ideas_aligned16(umat) || (in_scale && !ideas_aligned16(in_scale))) return IDEAS_E_ALIGN;
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t M = (int64_t)p->B * p->IH * (p->IW / 2);
    const int64_t tm = ideas_cdiv(M, WBM);
    const int tn = (int)ideas_cdiv(p->Cout, WBN);
    if (tm * tn > 0x7fffffffLL) return IDEAS_E_SHAPE;
    auto go = [&](auto sc, auto rf) {
        hipLaunchKernelGGL((conv3x3_wino_kernel<decltype(sc)::value, decltype(rf)::value>), dim3((unsigned)(tm * tn)),
                           dim3(256), 0, stream, (float*)y, (const float*)x, (const float*)umat, in_scale, out_scale, bias,
                           (const float*)resid, *p, tn);
    };
    using T = std::true_type;
    using F = std::false_type;
    if (in_scale) { if (p->reflect) go(T{}, T{}); else go(T{}, F{}); }
    else { if (p->reflect) go(F{}, T{}); else go(F{}, F{}); }
    return ideas_launch_status();
}

// gu: ZEROED [4][Cout][3][Cin]; gy [B,H,W,Cout], x [B,H,W,Cin].  Same geometry restrictions as ideas_conv3x3_wino,
// plus Cout % 4 == 0.  in_scale / out_scale: both or neither.
extern "C" int ideas_conv3x3_wino_wgrad(float* gu, const void* gy, const void* x, const float* in_scale,
                                        const float* out_scale, const ideas_conv_params* p, int dtype, void* stream_) {
    if (dtype != IDEAS_F32) return IDEAS_E_UNSUPPORTED;
    if (!gu || !gy || !x || !p) return IDEAS_E_NULL;
    if (p->B <= 0 || p->IH <= 0 || p->IW <= 0 || p->Cin <= 0 || p->Cout <= 0) return IDEAS_E_SHAPE;
    if (p->TY != 3 || p->TX != 3 || p->sy != 1 || p->sx != 1 || p->OH != p->IH || p->OW != p->IW || p->YH != p->IH ||
        p->YW != p->IW)
        return IDEAS_E_UNSUPPORTED;
    if ((p->IW & 1) || (p->Cin % 8) || (p->Cout % 4)) return IDEAS_E_ALIGN;
    if ((in_scale == nullptr) != (out_scale == nullptr)) return IDEAS_E_UNSUPPORTED;
    if (!ideas_aligned16(x) || !ideas_aligned16(gy) || (in_scale && (!ideas_aligned16(in_scale) || !ideas_aligned16(out_scale))))
        return IDEAS_E_ALIGN;
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t P2 = (int64_t)p->B * p->IH * (p->IW / 2);
    if (P2 >= 0x7fffffffLL) return IDEAS_E_SHAPE;
    const int Ktot = 3 * p->Cin;
    const int tm = (int)ideas_cdiv(p->Cout, GBM), tn = (int)ideas_cdiv(Ktot, GBN);
    const int64_t tiles = (int64_t)tm * tn;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    const int64_t slots = 2LL * n_cu;                       // 2 blocks per CU (208 VGPRs)
    const int64_t max_splits = ideas_cdiv(P2, 16 * GBK);
    int64_t splits = (2 * slots) / tiles;
    if (splits < 1) splits = 1;
    if (splits > max_splits) splits = max_splits;
    if (splits > 65535) splits = 65535;
    int64_t per = ideas_cdiv(ideas_cdiv(P2, splits), GBK) * GBK;
    splits = ideas_cdiv(P2, per);
    {
        const int64_t blocks = tiles * splits, waves = blocks / slots;
        if (waves >= 1 && blocks % slots) {
            const int64_t want = (waves * slots) / tiles;
            if (want >= 1) { per = ideas_cdiv(ideas_cdiv(P2, want), GBK) * GBK; splits = ideas_cdiv(P2, per); }
        }
    }
    auto go = [&](auto sc, auto rf) {
        if (p->IW / 2 >= GBK)
            hipLaunchKernelGGL((conv3x3_wino_wgrad_kernel<decltype(sc)::value, decltype(rf)::value, true>),
                               dim3((unsigned)tiles, (unsigned)splits), dim3(256), 0, stream, gu, (const float*)gy,
                               (const float*)x, in_scale, out_scale, *p, tn, per);
        else
            hipLaunchKernelGGL((conv3x3_wino_wgrad_kernel<decltype(sc)::value, decltype(rf)::value, false>),
                               dim3((unsigned)tiles, (unsigned)splits), dim3(256), 0, stream, gu, (const float*)gy,
                               (const float*)x, in_scale, out_scale, *p, tn, per);
    };
    using T = std::true_type;
    using F = std::false_type;
    if (in_scale) { if (p->reflect) go(T{}, T{}); else go(T{}, F{}); }
    else { if (p->reflect) go(F{}, T{}); else go(F{}, F{}); }
    return ideas_launch_status();
}

namespace {
// dU [4][Cout][3][Cin] -> the 3x3 taps, ADDED into gw (element (o, ky, kx, ci) at gw[o*so + ky*sky + kx*skx + ci*sc]); `clear`
// re-zeroes dU behind the read so that the scratch is ready for the next weight gradient without a fill launch
__global__ __launch_bounds__(256) void wino_wgrad_fold_kernel(float* __restrict__ gw, float* __restrict__ gu, int Cout, int Cin,
                                                              int64_t so, int64_t sky, int64_t skx, int64_t sc, int clear) {
    const int64_t n = (int64_t)Cout * 3 * Cin;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int ci = (int)(i % Cin);
        const int ky = (int)((i / Cin) % 3);
        const int o = (int)(i / (3 * (int64_t)Cin));
        const float u0 = gu[i], u1 = gu[n + i], u2 = gu[2 * n + i], u3 = gu[3 * n + i];
        if (clear) { gu[i] = 0.f; gu[n + i] = 0.f; gu[2 * n + i] = 0.f; gu[3 * n + i] = 0.f; }
        const float half = (u1 + u2) * 0.5f;
        float* d = gw + o * so + ky * sky + ci * sc;
        d[0] += u0 + half;
        d[skx] += (u1 - u2) * 0.5f;
        d[2 * skx] += half + u3;
    }
}
}  // namespace

extern "C" int ideas_wino_wgrad_fold(float* gw, float* gu, int Cout, int Cin, int64_t so, int64_t sky, int64_t skx, int64_t sc,
                                     int clear, void* stream_) {
    if (!gw || !gu) return IDEAS_E_NULL;
    if (Cout <= 0 || Cin <= 0) return IDEAS_E_SHAPE;
    const int64_t n = (int64_t)Cout * 3 * Cin;
    const int blocks = (int)(ideas_cdiv(n, 256) < 4096 ? ideas_cdiv(n, 256) : 4096);
    hipLaunchKernelGGL(wino_wgrad_fold_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, gw, gu, Cout, Cin, so, sky, skx, sc,
                       clear);
    return ideas_launch_status();
}
