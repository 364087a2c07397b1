// Weight gradient on the bf16 matrix pipe with the exact 3-way split of b3.hpp (see conv_b3.hip for the arithmetic):
//
//     gw[o][k] += gain * sum over pixels p of  G(p, o) * X(p, k)          k = (ty, tx, ci)
//
// GEMM view: M = Cout, N = TY*TX*Cin, reduction over the B*OH*OW output pixels, 16 pixels per pipeline step (= the K
// of one v_mfma_f32_32x32x16_bf16), split-K (common.hpp: splitk_xcd_map) with f32 atomics into the caller-zeroed gw.
//
// Both operands are pixel-major in HBM (NHWC: channels contiguous), but the MFMA wants, per lane, 8 consecutive PIXELS
// of one channel.  The transpose happens in registers on the way into LDS: a staging thread owns a 4-channel x
// 4-pixel block (four coalesced 16-byte buffer loads), splits its 16 values, and writes per channel and plane one
// 8-byte group of 4 consecutive pixels.  For those stores to be conflict-free the four channels of a thread must
// not sit 4 LDS rows apart (= 128 bytes = one full turn of the banks), so LDS row `pos` holds channel
//        c = 4 * (pos % (R/4)) + pos / (R/4)          (R = rows of the tile: the thread's channel j lives in quarter j)
// and the MFMA tiles simply take 32 consecutive LDS rows: a GEMM does not care which channel an operand row is, the
// epilogue applies the same permutation to the output coordinates.  Reads are then exactly conv_b3's (conflict-free).
//
// Threads [0, BM) stage G (x out_scale for the modulated conv), threads [BM, BM+BN) stage X (x in_scale): roles are
// wave-uniform.  Padding taps use the out-of-range buffer offset (hardware zero fill); rows past Cout / columns past
// K compute on whatever the loads return and are not stored.  Requires OW % 4 == 0 (a 4-pixel group never straddles
// an image row), 16 | OW or OW | 16, and B*OH*OW % 16 == 0 -- true for every layer of the networks.
#include "b3.hpp"
#include <type_traits>

namespace {

template <int WM, int WN, int MT, int NT, bool SCALE, bool REFLECT>
__global__ __launch_bounds__(256, 2) void conv_b3_wgrad_kernel(float* __restrict__ gw, const float* __restrict__ gy,
                                                               const float* __restrict__ x,
                                                               const float* __restrict__ in_scale,
                                                               const float* __restrict__ out_scale, ideas_conv_params p,
                                                               int tiles_n, int pix_per_split, unsigned gy_bytes,
                                                               unsigned x_bytes, int tiles, int splits) {
    static_assert(WM * WN == 4, "4 waves per block");
    constexpr int BM = WM * MT * 32;   // output channels of the tile
    constexpr int BN = WN * NT * 32;   // k columns of the tile
    static_assert(BM % 64 == 0 && BN % 64 == 0 && BM + BN <= 256, "one staging thread per 4 rows, roles per wave");
    constexpr int PLANE_A = BM * ROWB, PLANE_B = BN * ROWB;
    constexpr int BUF = 3 * (PLANE_A + PLANE_B);
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF];

    const int t = threadIdx.x;
    const int Ktot = p.TY * p.TX * p.Cin;
    int tile, split;
    splitk_xcd_map(blockIdx.x, tiles, splits, tile, split);
    const int tile_n = tile % tiles_n;
    const int tile_m = tile / tiles_n;
    const int o0 = tile_m * BM;
    const int n0 = tile_n * BN;
    const int P = p.B * p.OH * p.OW;
    const int pbeg = split * pix_per_split;
    const int pend = pbeg + pix_per_split < P ? pbeg + pix_per_split : P;
    if (pbeg >= pend) return;

    // ---- staging role of this wave: G (rows = output channels) or X (rows = k columns) -----------------------------
    // One code path serves both: "read 4 consecutive pixels of channels [c0, c0+4) from an NHWC tensor [B][H][W][C] at
    // (oy*sy + yoff, (ox0+j)*sx + xoff)"; only the (wave-uniform, scalar) parameters differ, so the K loop stays one
    // basic block.  Threads past BM+BN (64x128 tile) repeat X rows: same values to the same addresses.
    const bool is_g = __builtin_amdgcn_readfirstlane((int)(t < BM)) != 0;
    const int tt = is_g ? t : t - BM;
    const int ROWS = is_g ? BM : BN;
    const int pq = tt & 3;                           // which 4-pixel group of the 16-pixel step
    const int cq = (tt >> 2) % (ROWS / 4);           // which channel / column quad of the tile
    const int plane = is_g ? PLANE_A : PLANE_B;
    // LDS byte offset of (quarter 0, row cq, pixel group pq); quarter j adds j * (ROWS/4) rows, and (ROWS/4) % 16 == 0,
    // so the swizzle bit (row bit 3) is the same in all quarters
    const int lds_base = (is_g ? 0 : 3 * PLANE_A) + cq * ROWB + ((pq * 8) ^ (((cq >> 3) & 1) << 4));
    const int lds_quarter = (ROWS / 4) * ROWB;

    // X column (tap, ci) of this thread; columns past K decode to a tap >= TY*TX: harmless (never stored)
    const int kcol = n0 + cq * 4;
    const int tap = kcol / p.Cin;
    const int x_ty = tap / p.TX, x_tx = tap - x_ty * p.TX;
    const int sH = is_g ? p.YH : p.IH, sW = is_g ? p.YW : p.IW, sC = is_g ? p.Cout : p.Cin;
    const int s_sy = is_g ? p.osy : p.sy, s_sx = is_g ? p.osx : p.sx;
    const int s_yoff = is_g ? p.ooy : x_ty * p.dy + p.offy;          // (per-thread for X: depends on the tap)
    const int s_xoff = is_g ? p.oox : x_tx * p.dx + p.offx;
    const int s_c0 = is_g ? o0 + cq * 4 : kcol - tap * p.Cin;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(is_g ? gy : x), 0, (int)(is_g ? gy_bytes : x_bytes), (int)RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rscale = __builtin_amdgcn_make_buffer_rsrc((void*)(is_g ? out_scale : in_scale), 0, SCALE ? p.B * sC * 4 : 0, (int)RSRC_FLAGS);

    // pixel walk of this thread's 4-pixel group: (b, oy, ox0), advanced by 16 pixels per step
    int w_b, w_oy, w_ox;
    {
        const int pp = pbeg + pq * 4;
        const int q = pp / p.OW;
        w_ox = pp - q * p.OW;
        w_b = q / p.OH;
        w_oy = q - w_b * p.OH;
    }
    const int d_ox = 16 % p.OW, d_oy = 16 / p.OW;    // (16 | OW: d_oy = 0;  OW | 16: d_ox = 0)

    struct Stage { float4 v[4]; float4 s; };
    Stage st0, st1;
    auto gload = [&](Stage& st) {
        int iy = w_oy * s_sy + s_yoff;
        const int ix0 = w_ox * s_sx + s_xoff;
        bool yok = true;
        if (REFLECT) iy = reflect_coord(iy, sH);
        else yok = (unsigned)iy < (unsigned)sH;
        const unsigned rowb = (unsigned)((w_b * sH + iy) * sW * sC + s_c0) * 4u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int ix = ix0 + j * s_sx;
            bool ok = yok;
            if (REFLECT) ix = reflect_coord(ix, sW);
            else ok = ok && (unsigned)ix < (unsigned)sW;
            st.v[j] = buffer_load4(rsrc, ok ? rowb + (unsigned)(ix * sC) * 4u : 0xffffffffu, 0);   // padding -> hardware zero fill
        }
        if (SCALE) st.s = buffer_load4(rscale, (unsigned)(w_b * sC + s_c0) * 4u, 0);
        // advance 16 pixels
        w_ox += d_ox;
        const bool cx = w_ox >= p.OW;
        w_ox -= cx ? p.OW : 0;
        w_oy += d_oy + (cx ? 1 : 0);
        const bool cy = w_oy >= p.OH;
        w_oy -= cy ? p.OH : 0;
        w_b += cy ? 1 : 0;
    };
    auto lstore = [&](int buf, const Stage& st) {
        unsigned char* base = smem + buf * BUF + lds_base;
        const float vv[4][4] = {{st.v[0].x, st.v[0].y, st.v[0].z, st.v[0].w}, {st.v[1].x, st.v[1].y, st.v[1].z, st.v[1].w},
                                {st.v[2].x, st.v[2].y, st.v[2].z, st.v[2].w}, {st.v[3].x, st.v[3].y, st.v[3].z, st.v[3].w}};
        const float sc[4] = {st.s.x, st.s.y, st.s.z, st.s.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {                // channel c of the quad: its 4 pixels -> one 8-byte group per plane
            float e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = SCALE ? mul_rn(vv[j][c], sc[c]) : vv[j][c];
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
        bf16x8 fa[MT][3], fb[NT][3];
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                fa[a][pl] = *reinterpret_cast<const bf16x8*>(base + a_off + pl * PLANE_A + a * 32 * ROWB);
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                fb[b][pl] = *reinterpret_cast<const bf16x8*>(base + b_off + pl * PLANE_B + b * 32 * ROWB);
        lstore(buf ^ 1, stg);
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][PA[q]], fb[b][PB[q]], acc[a][b], 0, 0, 0);
        __syncthreads();
    };
    // loads walk past pend in the last two steps: in range they fetch the next split's pixels, out of range zeros --
    // either way that data is stored to LDS but never multiplied
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

    // ---- epilogue: undo the row permutation, f32 atomics ----------------------------------------------------------
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        const int posn = (wn * NT + b) * 32 + li;
        const int k = n0 + 4 * (posn % (BN / 4)) + posn / (BN / 4);
        if (k >= Ktot) continue;
#pragma unroll
        for (int a = 0; a < MT; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int posm = (wm * MT + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int o = o0 + 4 * (posm % (BM / 4)) + posm / (BM / 4);
                if (o < p.Cout) atomicAdd(&gw[(int64_t)o * Ktot + k], acc[a][b][r] * p.gain);
            }
        }
    }
}

template <int WM, int WN, int MT, int NT>
int launch_b3_wgrad_cfg(float* gw, const void* gy, const void* x, const float* in_scale, const float* out_scale,
                        const ideas_conv_params* p, hipStream_t stream) {
    constexpr int BM_ = WM * MT * 32, BN_ = WN * NT * 32;
    const int64_t P = (int64_t)p->B * p->OH * p->OW;
    const int Ktot = p->TY * p->TX * p->Cin;
    const int tm = (int)ideas_cdiv(p->Cout, BM_);
    const int tn = (int)ideas_cdiv(Ktot, BN_);
    const int64_t tiles = (int64_t)tm * tn;
    static int occ = 0, n_cu = 0;
    if (!occ) {
        int o = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, conv_b3_wgrad_kernel<WM, WN, MT, NT, true, false>, 256, 0);
        occ = (e == hipSuccess && o > 0) ? o : 2;
        int dev = 0;
        hipDeviceProp_t prop;
        n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    // split-K sizing as in conv_igemm.hip: whole waves of resident blocks, >= 16 steps per block
    const int64_t slots = (int64_t)occ * n_cu;
    const int64_t max_splits = ideas_cdiv(P, 16 * 16);
    // two waves of resident blocks -- four where every block still reduces >= 4096 pixels: the long reductions of the 128-512
    // channel layers at >= 64x64 gain 4-6 % from the finer interleaving (164 -> 170, 169 -> 179 TFLOP/s), short ones lose to the
    // extra atomics (E.2.conv1: 152 -> 123), profiles/r02_wgrad_ab.txt
    int64_t splits = (4 * slots) / tiles;
    if (splits < 1 || P / splits < 4096) splits = (2 * slots) / tiles;
    if (splits < 1) splits = 1;
    if (splits > max_splits) splits = max_splits;
    if (splits > 65535) splits = 65535;
    int64_t per = ideas_cdiv(ideas_cdiv(P, splits), 16) * 16;
    splits = ideas_cdiv(P, per);
    {
        const int64_t blocks = tiles * splits;
        const int64_t waves = blocks / slots;
        if (waves >= 1 && blocks % slots) {
            const int64_t want = (waves * slots) / tiles;
            if (want >= 1) {
                per = ideas_cdiv(ideas_cdiv(P, want), 16) * 16;
                splits = ideas_cdiv(P, per);
            }
        }
    }
    const unsigned gy_bytes = (unsigned)((int64_t)p->B * p->YH * p->YW * p->Cout * 4);
    const unsigned x_bytes = (unsigned)((int64_t)p->B * p->IH * p->IW * p->Cin * 4);
    auto go = [&](auto sc, auto rf) {
        hipLaunchKernelGGL((conv_b3_wgrad_kernel<WM, WN, MT, NT, decltype(sc)::value, decltype(rf)::value>),
                           dim3(splitk_grid(tiles, splits)), dim3(256), 0, stream, gw, (const float*)gy,
                           (const float*)x, in_scale, out_scale, *p, tn, (int)per, gy_bytes, x_bytes, (int)tiles, (int)splits);
    };
    using T = std::true_type;
    using F = std::false_type;
    const bool sc = in_scale && out_scale;
    if (sc) { if (p->reflect) go(T{}, T{}); else go(T{}, F{}); }
    else { if (p->reflect) go(F{}, T{}); else go(F{}, F{}); }
    return ideas_launch_status();
}

}  // namespace

extern "C" int ideas_b3_wgrad_supported(const ideas_conv_params* p) {
    if (!p) return 0;
    if (ideas_b3_wgrad3_enabled() && ideas_b3_wgrad3_supported(p)) return 1;     // the tap-fused 3x3 kernel takes it
    const int64_t P = (int64_t)p->B * p->OH * p->OW;
    // (Cout <= 32 and tiny reductions stay on the f32 kernels: half-empty tiles / atomics-dominated there, measured slower;
    //  32 < Cout <= 64 runs a 64 x 192 tile -- all four waves stage, 18 MFMAs per wave and step -- which beats the f32
    //  Winograd weight gradient by 8-20 % there; a 64 x 128 tile with an idle staging wave did not)
    if (p->Cout <= 32 || (P < 16384 && (int64_t)p->TY * p->TX * p->Cin < 2048)) return 0;
    return p->Cin % 4 == 0 && p->Cout % 4 == 0 && p->OW % 4 == 0 && (p->OW % 16 == 0 || 16 % p->OW == 0) &&
           16 / p->OW <= p->OH && P % 16 == 0 && P < 0x7fffffffLL &&
           (int64_t)p->B * p->IH * p->IW * p->Cin * 4 < 0xffffffffLL && (int64_t)p->B * p->YH * p->YW * p->Cout * 4 < 0xffffffffLL;
}

// called by ideas_conv_wgrad for dtype IDEAS_F32_B3 once the arguments are validated and ideas_b3_wgrad_supported
int ideas_b3_wgrad(float* gw, const void* gy, const void* x, const float* in_scale, const float* out_scale,
                   const ideas_conv_params* p, hipStream_t stream) {
    if (p->Cout <= 64) return launch_b3_wgrad_cfg<2, 2, 1, 3>(gw, gy, x, in_scale, out_scale, p, stream);   // 64 (o) x 192 (k)
    return launch_b3_wgrad_cfg<2, 2, 2, 2>(gw, gy, x, in_scale, out_scale, p, stream);   // 128 (o) x 128 (k)
}
