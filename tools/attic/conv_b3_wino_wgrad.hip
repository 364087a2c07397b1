// Weight gradient of the 3x3 / stride-1 / pad-1 layers in the 1-D Winograd F(2,3) domain, contracted with the exact 3-way
// bf16 split of b3.hpp: conv_wino.hip's algebra on conv_b3_wgrad.hip's pipeline.
//
//     dU_v[o][ky][ci] += gain * sum over (b, y, t) of  dM_v(b, y, t, o) * V_v(b, y + ky - 1, t, ci)
//     dM = (g0, g0 + g1, g0 - g1, -g1)                  g_e = gy[b, y, 2t + e, o]  (x out_scale)
//     V  = (d0 - d2, d1 + d2, d2 - d1, d1 - d3)         d_j = x[b, ., 2t - 1 + j, ci]  (x in_scale)
//
// The reduction runs over column PAIRS, so the four component GEMMs together issue 4 * 3 * Cin * Cout * P/2 products against
// the direct kernel's 9 * Cin * Cout * P: 2/3 of the MFMAs and -- because every operand value is one combination of two
// loaded pixels -- 2/3 of the operand splits.  ideas_wino_wgrad_fold maps dU back to the taps:
//     dw[kx=0] = dU0 + (dU1 + dU2)/2,   dw[kx=1] = (dU1 - dU2)/2,   dw[kx=2] = (dU1 + dU2)/2 + dU3.
//
// A block owns ONE component v (blockIdx.x -> (o tile, v, (ky, ci) tile)): its two operands are then plain "fa * T[col a] +
// fb * T[col b]" combinations with wave-uniform coefficients, and the K loop is conv_b3_wgrad's: a staging thread owns
// 4 channels x 4 consecutive pairs (eight 16-byte buffer loads), combines, splits and writes per channel and plane one 8-byte
// group of 4 consecutive pairs; LDS rows are permuted exactly as there (channel c = 4 * (pos % (R/4)) + pos / (R/4)).
// Requires W % 8 == 0 (a 4-pair group never straddles a row), 16 | W/2 or W/2 | 16, and B*H*W/2 % 16 == 0.
#include "b3.hpp"
#include <type_traits>

namespace {

template <int WM, int WN, int MT, int NT, bool SCALE, bool REFLECT>
__global__ __launch_bounds__(256, 2) void conv_b3_wino_wgrad_kernel(float* __restrict__ gu, const float* __restrict__ gy,
                                                                    const float* __restrict__ x,
                                                                    const float* __restrict__ in_scale,
                                                                    const float* __restrict__ out_scale, ideas_conv_params p,
                                                                    int tiles_nv, int pairs_per_split, unsigned gy_bytes,
                                                                    unsigned x_bytes, int tiles, int splits) {
    static_assert(WM * WN == 4, "4 waves per block");
    constexpr int BM = WM * MT * 32;   // output channels of the tile
    constexpr int BN = WN * NT * 32;   // (ky, ci) columns of the tile
    static_assert(BM % 64 == 0 && BN % 64 == 0 && BM + BN <= 256, "one staging thread per 4 rows, roles per wave");
    constexpr int PLANE_A = BM * ROWB, PLANE_B = BN * ROWB;
    constexpr int BUF = 3 * (PLANE_A + PLANE_B);
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF];

    const int t = threadIdx.x;
    const int H = p.IH, W = p.IW, W2 = W >> 1;
    const int Kv = 3 * p.Cin;                          // columns of one component
    int tile, split;
    splitk_xcd_map(blockIdx.x, tiles, splits, tile, split);
    const int tn_all = tile % (4 * tiles_nv);
    const int tile_m = tile / (4 * tiles_nv);
    const int v = tn_all / tiles_nv;                   // Winograd component of this block
    const int tile_n = tn_all - v * tiles_nv;
    const int o0 = tile_m * BM;
    const int n0 = tile_n * BN;
    const int P2 = p.B * H * W2;
    const int pbeg = split * pairs_per_split;
    const int pend = pbeg + pairs_per_split < P2 ? pbeg + pairs_per_split : P2;
    if (pbeg >= pend) return;

    // ---- staging role of this wave: G (rows = output channels) or X (rows = (ky, ci) columns) ----------------------------------
    const bool is_g = __builtin_amdgcn_readfirstlane((int)(t < BM)) != 0;
    const int tt = is_g ? t : t - BM;
    const int ROWS = is_g ? BM : BN;
    const int pq = tt & 3;                           // which 4-pair group of the 16-pair step
    const int cq = (tt >> 2) % (ROWS / 4);           // which channel / column quad of the tile
    const int plane = is_g ? PLANE_A : PLANE_B;
    const int lds_base = (is_g ? 0 : 3 * PLANE_A) + cq * ROWB + ((pq * 8) ^ (((cq >> 3) & 1) << 4));
    const int lds_quarter = (ROWS / 4) * ROWB;

    // value of a pair = fa * T[row, 2t + oa] + fb * T[row, 2t + ob]      (block-uniform per role)
    //   G: dM_v of (g0, g1)                X: V_v of (d0..d3), d_j at column 2t - 1 + j
    const int oa = is_g ? 0 : (v == 0 ? -1 : 0);
    const int ob = is_g ? 1 : (v == 3 ? 2 : 1);
    const float fa = is_g ? (v == 3 ? 0.f : 1.f) : (v == 2 ? -1.f : 1.f);
    const float fb = is_g ? (v == 0 ? 0.f : (v == 1 ? 1.f : -1.f)) : ((v == 1 || v == 2) ? 1.f : -1.f);
    const bool use_a = fa != 0.f, use_b = fb != 0.f;  // a zero coefficient skips the load (out-of-range offset -> hardware zeros)

    const int kcol = n0 + cq * 4;                     // X column (ky, ci) of this thread; columns past Kv are never stored
    const int x_ky = kcol / p.Cin;
    const int sC = is_g ? p.Cout : p.Cin;
    const int s_yoff = is_g ? 0 : x_ky - 1;
    const int s_c0 = is_g ? o0 + cq * 4 : kcol - x_ky * p.Cin;
    const bool row_ok = is_g ? (s_c0 < p.Cout) : (kcol < Kv);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(is_g ? gy : x), 0, (int)(is_g ? gy_bytes : x_bytes), (int)RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rscale = __builtin_amdgcn_make_buffer_rsrc((void*)(is_g ? out_scale : in_scale), 0, SCALE ? p.B * sC * 4 : 0, (int)RSRC_FLAGS);

    // pair walk of this thread's 4-pair group: (b, y, tx), advanced by 16 pairs per step
    int w_b, w_y, w_tx;
    {
        const int pp = pbeg + pq * 4;
        const int q = pp / W2;
        w_tx = pp - q * W2;
        w_b = q / H;
        w_y = q - w_b * H;
    }
    const int d_tx = 16 % W2, d_y = 16 / W2;         // (16 | W2: d_y = 0;  W2 | 16: d_tx = 0)

    struct Stage { float4 a[4], b[4]; float4 s; };
    Stage st0, st1;
    auto gload = [&](Stage& st) {
        int iy = w_y + s_yoff;
        bool yok = row_ok;
        if (REFLECT) iy = reflect_coord(iy, H);
        else yok = yok && (unsigned)iy < (unsigned)H;
        const unsigned rowb = (unsigned)((w_b * H + iy) * W * sC + s_c0) * 4u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int ia = 2 * (w_tx + j) + oa, ib = 2 * (w_tx + j) + ob;
            bool oka = yok && use_a, okb = yok && use_b;
            if (REFLECT) { ia = reflect_coord(ia, W); ib = reflect_coord(ib, W); }
            else { oka = oka && (unsigned)ia < (unsigned)W; okb = okb && (unsigned)ib < (unsigned)W; }
            st.a[j] = buffer_load4(rsrc, oka ? rowb + (unsigned)(ia * sC) * 4u : 0xffffffffu, 0);
            st.b[j] = buffer_load4(rsrc, okb ? rowb + (unsigned)(ib * sC) * 4u : 0xffffffffu, 0);
        }
        if (SCALE) st.s = buffer_load4(rscale, (unsigned)(w_b * sC + s_c0) * 4u, 0);
        // advance 16 pairs
        w_tx += d_tx;
        const bool cx = w_tx >= W2;
        w_tx -= cx ? W2 : 0;
        w_y += d_y + (cx ? 1 : 0);
        const bool cy = w_y >= H;
        w_y -= cy ? H : 0;
        w_b += cy ? 1 : 0;
    };
    auto lstore = [&](int buf, const Stage& st) {
        unsigned char* base = smem + buf * BUF + lds_base;
        const float va[4][4] = {{st.a[0].x, st.a[0].y, st.a[0].z, st.a[0].w}, {st.a[1].x, st.a[1].y, st.a[1].z, st.a[1].w},
                                {st.a[2].x, st.a[2].y, st.a[2].z, st.a[2].w}, {st.a[3].x, st.a[3].y, st.a[3].z, st.a[3].w}};
        const float vb[4][4] = {{st.b[0].x, st.b[0].y, st.b[0].z, st.b[0].w}, {st.b[1].x, st.b[1].y, st.b[1].z, st.b[1].w},
                                {st.b[2].x, st.b[2].y, st.b[2].z, st.b[2].w}, {st.b[3].x, st.b[3].y, st.b[3].z, st.b[3].w}};
        const float sc[4] = {st.s.x, st.s.y, st.s.z, st.s.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {                // channel c of the quad: its 4 pairs -> one 8-byte group per plane
            float e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // the per-sample scale goes on the two loaded values, THEN they are combined: the other order,
                // mul_rn(fmaf(fb, b, fa * a), scale), came back from hipcc (ROCm 7.2) with wrong values in the rows whose
                // LDS chunk is swizzled (tools/probes/wino_wgrad_diag.py found it; both forms are the same polynomial).
                // Coefficients are 0 / +-1, so the products are exact and the FMA rounds once: conv_wino.hip's add / subtract
                const float aa = SCALE ? mul_rn(va[j][c], sc[c]) : va[j][c];
                const float bb = SCALE ? mul_rn(vb[j][c], sc[c]) : vb[j][c];
                e[j] = fmaf(fb, bb, fa * aa);
            }
            uint2 pl[3];
            split2(e[0], e[1], pl[0].x, pl[1].x, pl[2].x);
            split2(e[2], e[3], pl[0].y, pl[1].y, pl[2].y);
#pragma unroll
            for (int q = 0; q < 3; ++q) *reinterpret_cast<uint2*>(base + c * lds_quarter + q * plane) = pl[q];
        }
    };

    const int lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int f_swz = (lh ^ ((li >> 3) & 1)) << 4;
    const int a_off = ((wm * MT) * 32 + li) * ROWB + f_swz;
    const int b_off = 3 * PLANE_A + ((wn * NT) * 32 + li) * ROWB + f_swz;

    auto step = [&](int buf, Stage& ld, const Stage& stg) {
        gload(ld);
        const unsigned char* base = smem + buf * BUF;
        bf16x8 fa_[MT][3], fb_[NT][3];
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                fa_[a][pl] = *reinterpret_cast<const bf16x8*>(base + a_off + pl * PLANE_A + a * 32 * ROWB);
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                fb_[b][pl] = *reinterpret_cast<const bf16x8*>(base + b_off + pl * PLANE_B + b * 32 * ROWB);
        lstore(buf ^ 1, stg);
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_[a][PA[q]], fb_[b][PB[q]], acc[a][b], 0, 0, 0);
        __syncthreads();
    };
    // loads walk past pend in the last two steps: in range they fetch the next split's pairs, out of range zeros -- either way
    // that data is stored to LDS but never multiplied
    const int nsteps = (pend - pbeg) / 16;
    gload(st0);
    gload(st1);
    lstore(0, st0);
    __syncthreads();
    int s = 0;
    for (; s + 1 < nsteps; s += 2) {
        step(0, st0, st1);
        step(1, st1, st0);
    }
    if (s < nsteps) step(0, st0, st1);

    // ---- epilogue: undo the row permutation, f32 atomics into dU_v ----------------------------------------------------------------
    float* guv = gu + (int64_t)v * p.Cout * Kv;
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        const int posn = (wn * NT + b) * 32 + li;
        const int k = n0 + 4 * (posn % (BN / 4)) + posn / (BN / 4);
        if (k >= Kv) continue;
#pragma unroll
        for (int a = 0; a < MT; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int posm = (wm * MT + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int o = o0 + 4 * (posm % (BM / 4)) + posm / (BM / 4);
                if (o < p.Cout) atomicAdd(&guv[(int64_t)o * Kv + k], acc[a][b][r] * p.gain);
            }
        }
    }
}

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

template <int WM, int WN, int MT, int NT>
int launch_b3_wino_wgrad_cfg(float* gu, const void* gy, const void* x, const float* in_scale, const float* out_scale,
                             const ideas_conv_params* p, hipStream_t stream) {
    constexpr int BM_ = WM * MT * 32, BN_ = WN * NT * 32;
    const int64_t P2 = (int64_t)p->B * p->IH * (p->IW / 2);
    const int Kv = 3 * p->Cin;
    const int tm = (int)ideas_cdiv(p->Cout, BM_);
    const int tnv = (int)ideas_cdiv(Kv, BN_);
    const int64_t tiles = (int64_t)tm * 4 * tnv;
    static int occ = 0, n_cu = 0;
    if (!occ) {
        int o = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, conv_b3_wino_wgrad_kernel<WM, WN, MT, NT, true, false>, 256, 0);
        occ = (e == hipSuccess && o > 0) ? o : 2;
        int dev = 0;
        hipDeviceProp_t prop;
        n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    // split-K sizing as in conv_b3_wgrad.hip: whole waves of resident blocks, >= 16 steps per block
    const int64_t slots = (int64_t)occ * n_cu;
    const int64_t max_splits = ideas_cdiv(P2, 16 * 16);
    int64_t splits = (2 * slots) / tiles;
    if (splits < 1) splits = 1;
    if (splits > max_splits) splits = max_splits;
    if (splits > 65535) splits = 65535;
    int64_t per = ideas_cdiv(ideas_cdiv(P2, splits), 16) * 16;
    splits = ideas_cdiv(P2, per);
    {
        const int64_t blocks = tiles * splits;
        const int64_t waves = blocks / slots;
        if (waves >= 1 && blocks % slots) {
            const int64_t want = (waves * slots) / tiles;
            if (want >= 1) {
                per = ideas_cdiv(ideas_cdiv(P2, want), 16) * 16;
                splits = ideas_cdiv(P2, per);
            }
        }
    }
    const unsigned gy_bytes = (unsigned)((int64_t)p->B * p->IH * p->IW * p->Cout * 4);
    const unsigned x_bytes = (unsigned)((int64_t)p->B * p->IH * p->IW * p->Cin * 4);
    auto go = [&](auto sc, auto rf) {
        hipLaunchKernelGGL((conv_b3_wino_wgrad_kernel<WM, WN, MT, NT, decltype(sc)::value, decltype(rf)::value>),
                           dim3(splitk_grid(tiles, splits)), dim3(256), 0, stream, gu, (const float*)gy,
                           (const float*)x, in_scale, out_scale, *p, tnv, (int)per, gy_bytes, x_bytes, (int)tiles, (int)splits);
    };
    using T = std::true_type;
    using F = std::false_type;
    const bool sc = in_scale && out_scale;
    if (sc) { if (p->reflect) go(T{}, T{}); else go(T{}, F{}); }
    else { if (p->reflect) go(F{}, T{}); else go(F{}, F{}); }
    return ideas_launch_status();
}

}  // namespace

extern "C" int ideas_b3_wino_wgrad_supported(const ideas_conv_params* p) {
    if (!p) return 0;
    const int W2 = p->IW / 2;
    const int64_t P2 = (int64_t)p->B * p->IH * W2;
    if (!(p->TY == 3 && p->TX == 3 && p->sy == 1 && p->sx == 1 && p->dy == 1 && p->dx == 1 && p->offy == -1 && p->offx == -1 &&
          p->OH == p->IH && p->OW == p->IW && p->YH == p->IH && p->YW == p->IW && p->osy == 1 && p->osx == 1 && p->ooy == 0 &&
          p->oox == 0))
        return 0;
    // (small layers stay on the direct split kernel / the f32 kernels: the component grid quadruples the tiles and the fold is a
    //  second launch, which only pays once the reduction is long)
    if (p->Cout <= 32 || P2 < 16384) return 0;
    return p->IW % 8 == 0 && (W2 % 16 == 0 || 16 % W2 == 0) && 16 / W2 <= p->IH && P2 % 16 == 0 && P2 < 0x7fffffffLL &&
           p->Cin % 4 == 0 && p->Cout % 4 == 0 && (!p->reflect || (p->IH >= 2 && p->IW >= 4)) &&
           (int64_t)p->B * p->IH * p->IW * p->Cin * 4 < 0xffffffffLL && (int64_t)p->B * p->IH * p->IW * p->Cout * 4 < 0xffffffffLL;
}

// called by ideas_conv3x3_wino_wgrad for dtype IDEAS_F32_B3 once the arguments are validated and ideas_b3_wino_wgrad_supported
int ideas_b3_wino_wgrad(float* gu, const void* gy, const void* x, const float* in_scale, const float* out_scale,
                        const ideas_conv_params* p, hipStream_t stream) {
    // tile by padding waste of the (ky, ci) axis: 3*Cin columns per component in tiles of 128 or 192
    const int Kv = 3 * p->Cin;
    const int64_t w128 = ideas_cdiv(Kv, 128) * 128 * ideas_cdiv(p->Cout, 128) * 128;
    const int64_t w192 = ideas_cdiv(Kv, 192) * 192 * ideas_cdiv(p->Cout, 64) * 64;
    if (p->Cout <= 64 || w192 < w128) return launch_b3_wino_wgrad_cfg<2, 2, 1, 3>(gu, gy, x, in_scale, out_scale, p, stream);   // 64 x 192
    return launch_b3_wino_wgrad_cfg<2, 2, 2, 2>(gu, gy, x, in_scale, out_scale, p, stream);                                      // 128 x 128
}

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
