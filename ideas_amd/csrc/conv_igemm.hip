// Implicit-GEMM convolution family on the gfx950 f32 matrix pipe (v_mfma_f32_32x32x2_f32), NHWC.
//
// Replaces the cuDNN convolutions of the reference (stylegan2/model.py:115-121 EqualConv2d, :258/:273 the
// groups=batch modulated convs, models.py:32-38 EqualConvTranspose2d) and their autograd backward.
//
//   GEMM view (forward family):  M = B*OH*OW output points, N = Cout, K = TY*TX*Cin, K ordered (ty,tx,ci)
//   so that with NHWC activations and OHWI weights BOTH operands are contiguous along K: every global access
//   is a 16-byte vector, no transposes anywhere.
//
//   Block = 256 threads = 4 wavefronts, tile BM x BN x 16.  Operands are staged global -> VGPR -> LDS
//   (register staging, not LDS-DMA: the gather needs per-lane predication for padding / parity phases),
//   double-buffered so the loads of K-step t+1 fly under the MFMAs of step t; one barrier per K-step.
//   LDS rows are 16 floats + 4 pad (80 B): ds_read_b128 of 16 consecutive rows then touches 16 distinct
//   16-byte slots -> conflict-free (MI355X LDS: 64 banks x 4 B, b128 serviced in 16-lane groups).
//   A wave owns MT x NT tiles of 32x32; lane (i = lane&31, h = lane>>5) reads its 8 consecutive K values
//   [8h, 8h+8) of row i with two ds_read_b128 and feeds MFMA #kk with element kk — A and B use the same K
//   permutation, so the sum over (h, kk) covers the 16-wide K-step exactly once.
//   f32 MFMA is an exact fmaf chain (no TF32-style truncation): results are f32-roundoff class.
//
//   The modulated conv never materialises per-sample weights: style s[b,ci] scales the A tile on its way
//   into LDS and demod d[b,o] scales the accumulator in the epilogue (same contraction, re-associated).
//
// Workgroup -> tile mapping is XCD-aware: hardware round-robins consecutive workgroup ids over the 8 XCDs,
// so ids are remapped to give each XCD (= each private L2) one contiguous band of M tiles with all its N
// tiles; 3x3 halos and the N-tile re-reads of the same activations then hit that XCD's L2.
#include "common.hpp"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 16;    // K (or pixel) depth of one pipeline step
constexpr int LDK = 20;   // padded LDS row length (floats) of the K-contiguous tiles

__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
// component-wise select: a ternary on the float4 aggregates becomes a select between two ADDRESSES, which forces
// the staging registers into scratch memory (seen in the ISA as scratch_store + vmcnt(0) after every load)
__device__ __forceinline__ float4 keep4(bool ok, float4 v) {
    return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

// =====================================================================================================
// forward family
// =====================================================================================================
template <int WM, int WN, int MT, int NT, bool SCALE, bool REFLECT, bool WIDE>
__global__ __launch_bounds__(256, SCALE ? 3 : 4) void conv_igemm_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                            const float* __restrict__ wmat,
                                                            const float* __restrict__ in_scale,
                                                            const float* __restrict__ out_scale,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ resid, ideas_conv_params p,
                                                            int tiles_n) {
    static_assert(WM * WN == 4, "4 waves per block");
    constexpr int BM = WM * MT * 32;
    constexpr int BN = WN * NT * 32;
    constexpr int A_PER = (BM * 4 + 255) / 256;   // float4 loads per thread for the A tile
    constexpr int B_PER = (BN * 4 + 255) / 256;
    constexpr int LDS_FLOATS = 2 * (BM + BN) * LDK;
    static_assert(LDS_FLOATS * 4 >= BM * 12, "epilogue row table must fit");
    __shared__ __attribute__((aligned(16))) float smem[LDS_FLOATS];
    float* As = smem;                    // [2][BM][LDK]
    float* Bs = smem + 2 * BM * LDK;     // [2][BN][LDK]

    const int t = threadIdx.x;
    const int64_t M = (int64_t)p.B * p.OH * p.OW;
    const int K = p.TY * p.TX * p.Cin;
    const int nblk = gridDim.x;
    const int swz = xcd_swizzle(blockIdx.x, nblk);
    const int tile_n = swz % tiles_n;
    const int tile_m = swz / tiles_n;
    const int64_t m0 = (int64_t)tile_m * BM;
    const int n0 = tile_n * BN;

    // ---- per-thread gather state ---------------------------------------------------------------
    const int kq = t & 3;  // which float4 of the 16-wide K-step
    int a_iyb[A_PER], a_ixb[A_PER], a_b[A_PER];
    bool a_ok[A_PER];
    int64_t a_base[A_PER];
#pragma unroll
    for (int j = 0; j < A_PER; ++j) {
        const int r = (t >> 2) + 64 * j;
        const int64_t m = m0 + r;
        a_ok[j] = (r < BM) && (m < M);
        const int64_t mm = a_ok[j] ? m : 0;
        const int ox = (int)(mm % p.OW);
        const int64_t q = mm / p.OW;
        const int oy = (int)(q % p.OH);
        const int b = (int)(q / p.OH);
        a_b[j] = b;
        a_iyb[j] = oy * p.sy + p.offy;
        a_ixb[j] = ox * p.sx + p.offx;
        a_base[j] = (int64_t)b * p.IH * p.IW * p.Cin;
    }
    const float* b_ptr[B_PER];
    bool b_ok[B_PER];
#pragma unroll
    for (int j = 0; j < B_PER; ++j) {
        const int r = (t >> 2) + 64 * j;
        b_ok[j] = (r < BN) && (n0 + r < p.Cout);
        b_ptr[j] = wmat + (int64_t)(b_ok[j] ? n0 + r : 0) * K + kq * 4;
    }
    // tap walker for this thread's K column
    int k_ci, k_tx, k_ty;
    {
        const int k = kq * 4;
        const int tap = k / p.Cin;
        k_ci = k - tap * p.Cin;
        k_ty = tap / p.TX;
        k_tx = tap - k_ty * p.TX;
    }

    float4 ra[A_PER], rs[A_PER], rb[B_PER];
    bool oka[A_PER], okb[B_PER];
    // gload only ISSUES loads (from a clamped, always-valid address: no exec-mask branches) and remembers the
    // predicate; scaling and zero-masking happen in lstore, i.e. after the MFMAs of the current step, so the loads
    // of step t+1 stay in flight under the matrix work of step t.
    auto gload = [&](int kt) {
        const bool kvalid = k_ty < p.TY;
#pragma unroll
        for (int j = 0; j < A_PER; ++j) {
            int iy = a_iyb[j] + k_ty * p.dy;
            int ix = a_ixb[j] + k_tx * p.dx;
            bool ok = a_ok[j] && kvalid;
            if (REFLECT) {
                iy = reflect_coord(iy, p.IH);
                ix = reflect_coord(ix, p.IW);
            } else {
                ok = ok && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
            }
            const int64_t off = ok ? a_base[j] + ((int64_t)iy * p.IW + ix) * p.Cin + k_ci : 0;
            ra[j] = *reinterpret_cast<const float4*>(x + off);
            if (SCALE) rs[j] = *reinterpret_cast<const float4*>(in_scale + (int64_t)a_b[j] * p.Cin + k_ci);
            oka[j] = ok;
        }
#pragma unroll
        for (int j = 0; j < B_PER; ++j) {
            const bool ok = b_ok[j] && kvalid;
            rb[j] = *reinterpret_cast<const float4*>(ok ? b_ptr[j] + (int64_t)kt * BK : wmat);
            okb[j] = ok;
        }
        // advance the walker by one K-step (no branches: the whole K loop stays one basic block)
        if (WIDE) {   // Cin >= 16: at most one tap boundary per step
            k_ci += BK;
            const bool wrap = k_ci >= p.Cin;
            k_ci -= wrap ? p.Cin : 0;
            k_tx += wrap ? 1 : 0;
            const bool wrap2 = k_tx == p.TX;
            k_tx = wrap2 ? 0 : k_tx;
            k_ty += wrap2 ? 1 : 0;
        } else {
            const int k = (kt + 1) * BK + kq * 4;
            const int tap = k / p.Cin;
            k_ci = k - tap * p.Cin;
            k_ty = tap / p.TX;
            k_tx = tap - k_ty * p.TX;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < A_PER; ++j) {
            const int r = (t >> 2) + 64 * j;
            float4 v = ra[j];
            if (SCALE) v = mul4(v, rs[j]);
            v = keep4(oka[j], v);
            if (r < BM) *reinterpret_cast<float4*>(As + ((buf * BM + r) * LDK + kq * 4)) = v;
        }
#pragma unroll
        for (int j = 0; j < B_PER; ++j) {
            const int r = (t >> 2) + 64 * j;
            const float4 v = keep4(okb[j], rb[j]);
            if (r < BN) *reinterpret_cast<float4*>(Bs + ((buf * BN + r) * LDK + kq * 4)) = v;
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

    // Pipeline (one barrier per K-step, loop body = ONE basic block):
    //   top of step t : registers hold tile t+1 (issued a whole step ago) -> scale/mask -> LDS[buf^1];
    //                   issue the loads of tile t+2 into the same registers;
    //   then          : fragments of tile t from LDS[buf], 8*MT*NT MFMAs;  barrier.
    // Loads beyond K are predicated off (clamped address, zeroed), so no bounds branch is needed.
    const int nk = (K + BK - 1) / BK;
    gload(0);
    lstore(0);
    gload(1);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        lstore(buf ^ 1);
        gload(kt + 2);
        float4 fa[MT][2], fb[NT][2];
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            const float* src = As + ((buf * BM + (wm * MT + a) * 32 + li) * LDK + lh * 8);
            fa[a][0] = *reinterpret_cast<const float4*>(src);
            fa[a][1] = *reinterpret_cast<const float4*>(src + 4);
        }
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const float* src = Bs + ((buf * BN + (wn * NT + b) * 32 + li) * LDK + lh * 8);
            fb[b][0] = *reinterpret_cast<const float4*>(src);
            fb[b][1] = *reinterpret_cast<const float4*>(src + 4);
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                const float4 av4 = fa[a][kk >> 2];
                const float av = (kk & 3) == 0 ? av4.x : (kk & 3) == 1 ? av4.y : (kk & 3) == 2 ? av4.z : av4.w;
#pragma unroll
                for (int b = 0; b < NT; ++b) {
                    const float4 bv4 = fb[b][kk >> 2];
                    const float bv = (kk & 3) == 0 ? bv4.x : (kk & 3) == 1 ? bv4.y : (kk & 3) == 2 ? bv4.z : bv4.w;
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a][b], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue: row table (output offset, batch index) in LDS, then masked coalesced stores --------
    int64_t* row_off = reinterpret_cast<int64_t*>(smem);
    int* row_b = reinterpret_cast<int*>(smem + 2 * BM);
    if (t < BM) {
        const int64_t m = m0 + t;
        int64_t off = -1;
        int b = 0;
        if (m < M) {
            const int ox = (int)(m % p.OW);
            const int64_t q = m / p.OW;
            const int oy = (int)(q % p.OH);
            b = (int)(q / p.OH);
            off = (((int64_t)b * p.YH + (oy * p.osy + p.ooy)) * p.YW + (ox * p.osx + p.oox)) * p.Cout;
        }
        row_off[t] = off;
        row_b[t] = b;
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        const int n = n0 + (wn * NT + b) * 32 + li;
        if (n >= p.Cout) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int a = 0; a < MT; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (wm * MT + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int64_t off = row_off[row];
                if (off < 0) continue;
                float v = mul_rn(acc[a][b][r], p.gain);      // separate roundings (no FMA contraction): the fused
                if (out_scale) v = mul_rn(v, out_scale[(int64_t)row_b[row] * p.Cout + n]);
                v = mul_then_add(v, 1.0f, bv);               // epilogue is bitwise conv -> fused_bias_act
                if (p.act) v = (v > 0.f ? v : v * p.alpha) * p.act_gain;
                if (resid) v = (v + resid[off + n]) * p.resid_gain;
                if (p.accumulate) y[off + n] += v; else y[off + n] = v;
            }
        }
    }
}

// =====================================================================================================
// weight gradient:  gw[o][k] += gain * sum_pixels G(p,o) * X(p,k)
//   GEMM view: M = Cout, N = TY*TX*Cin, reduction over the B*OH*OW pixels.  Both operands are contiguous
//   along their NON-reduced axis (channels), so tiles are stored pixel-major in LDS ([16][BM], [16][BN]) and
//   MFMA operands are read with conflict-free ds_read_b32.  The pixel axis is split over blockIdx.y
//   (split-K); partial tiles are combined with f32 atomics into the caller-zeroed gw.
// =====================================================================================================
template <int WM, int WN, int MT, int NT, bool SCALE, bool REFLECT, bool WIDEW>
__global__ __launch_bounds__(256, SCALE ? 3 : 4) void conv_wgrad_kernel(float* __restrict__ gw, const float* __restrict__ gy,
                                                            const float* __restrict__ x,
                                                            const float* __restrict__ in_scale,
                                                            const float* __restrict__ out_scale, ideas_conv_params p,
                                                            int tiles_n, int64_t pix_per_split) {
    constexpr int BM = WM * MT * 32;
    constexpr int BN = WN * NT * 32;
    constexpr int LDM = BM + 4, LDN = BN + 4;
    constexpr int G_PER = (BK * BM / 4 + 255) / 256;
    constexpr int X_PER = (BK * BN / 4 + 255) / 256;
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (LDM + LDN)];
    float* Gs = smem;                     // [2][BK][LDM]
    float* Xs = smem + 2 * BK * LDM;      // [2][BK][LDN]

    const int t = threadIdx.x;
    const int Ktot = p.TY * p.TX * p.Cin;
    const int64_t P = (int64_t)p.B * p.OH * p.OW;
    const int tile_n = blockIdx.x % tiles_n;
    const int tile_m = blockIdx.x / tiles_n;
    const int o0 = tile_m * BM;
    const int n0 = tile_n * BN;
    const int64_t pbeg = (int64_t)blockIdx.y * pix_per_split;
    const int64_t pend = (pbeg + pix_per_split < P) ? pbeg + pix_per_split : P;
    if (pbeg >= pend) return;

    // fixed per-thread columns
    int g_pr[G_PER], g_o[G_PER];
    bool g_ok[G_PER];
#pragma unroll
    for (int j = 0; j < G_PER; ++j) {
        const int idx = t + 256 * j;
        g_pr[j] = idx / (BM / 4);
        const int oq = idx % (BM / 4);
        g_o[j] = o0 + oq * 4;
        g_ok[j] = (idx < BK * BM / 4) && (g_o[j] < p.Cout);
    }
    int x_pr[X_PER], x_col[X_PER], x_ci[X_PER], x_ty[X_PER], x_tx[X_PER];
    bool x_ok[X_PER];
#pragma unroll
    for (int j = 0; j < X_PER; ++j) {
        const int idx = t + 256 * j;
        x_pr[j] = idx / (BN / 4);
        const int kq = idx % (BN / 4);
        const int k = n0 + kq * 4;
        x_col[j] = kq * 4;
        x_ok[j] = (idx < BK * BN / 4) && (k < Ktot);
        const int kk = x_ok[j] ? k : 0;
        const int tap = kk / p.Cin;
        x_ci[j] = kk - tap * p.Cin;
        x_ty[j] = tap / p.TX;
        x_tx[j] = tap - x_ty[j] * p.TX;
    }

    // pixel walkers: (b, oy, ox) of the pixel row each load slot handles, advanced by BK per step with 32-bit
    // adds/compares (no integer division in the loop); `left` = pixels remaining before pend for that slot.
    struct Walk { int b, oy, ox, left; };
    auto walk_init = [&](int pr) {
        Walk wk;
        const int64_t pp = pbeg + pr;
        wk.left = (int)(pend - pp);
        const int64_t q = pp / p.OW;
        wk.ox = (int)(pp - q * p.OW);
        wk.b = (int)(q / p.OH);
        wk.oy = (int)(q - (int64_t)wk.b * p.OH);
        return wk;
    };
    auto walk_step = [&](Walk& wk) {
        wk.left -= BK;
        wk.ox += BK;
        if (WIDEW) {   // OW >= BK: at most one row boundary per step -> selects only (single-basic-block K loop)
            const bool wrap = wk.ox >= p.OW;
            wk.ox -= wrap ? p.OW : 0;
            wk.oy += wrap ? 1 : 0;
            const bool wrap2 = wk.oy == p.OH;
            wk.oy = wrap2 ? 0 : wk.oy;
            wk.b += wrap2 ? 1 : 0;
        } else {
            while (wk.ox >= p.OW) {
                wk.ox -= p.OW;
                if (++wk.oy == p.OH) { wk.oy = 0; ++wk.b; }
            }
        }
    };
    Walk gwk[G_PER], xwk[X_PER];
#pragma unroll
    for (int j = 0; j < G_PER; ++j) gwk[j] = walk_init(g_pr[j]);
#pragma unroll
    for (int j = 0; j < X_PER; ++j) xwk[j] = walk_init(x_pr[j]);

    float4 rg[G_PER], rgs[G_PER], rx[X_PER], rxs[X_PER];
    bool okg[G_PER], okx[X_PER];
    auto gload = [&]() {   // issue only; scale + mask in lstore (after the MFMAs)
#pragma unroll
        for (int j = 0; j < G_PER; ++j) {
            const Walk wk = gwk[j];
            const bool ok = g_ok[j] && wk.left > 0;
            const int64_t off = ok ? (((int64_t)wk.b * p.YH + (wk.oy * p.osy + p.ooy)) * p.YW + (wk.ox * p.osx + p.oox)) * p.Cout + g_o[j] : 0;
            rg[j] = *reinterpret_cast<const float4*>(gy + off);
            if (SCALE) rgs[j] = *reinterpret_cast<const float4*>(out_scale + (ok ? (int64_t)wk.b * p.Cout + g_o[j] : 0));
            okg[j] = ok;
            walk_step(gwk[j]);
        }
#pragma unroll
        for (int j = 0; j < X_PER; ++j) {
            const Walk wk = xwk[j];
            int iy = wk.oy * p.sy + x_ty[j] * p.dy + p.offy;
            int ix = wk.ox * p.sx + x_tx[j] * p.dx + p.offx;
            bool ok = x_ok[j] && wk.left > 0;
            if (REFLECT) {
                iy = reflect_coord(iy, p.IH);
                ix = reflect_coord(ix, p.IW);
            } else {
                ok = ok && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
            }
            const int64_t off = ok ? (((int64_t)wk.b * p.IH + iy) * p.IW + ix) * p.Cin + x_ci[j] : 0;
            rx[j] = *reinterpret_cast<const float4*>(x + off);
            if (SCALE) rxs[j] = *reinterpret_cast<const float4*>(in_scale + (ok ? (int64_t)wk.b * p.Cin + x_ci[j] : 0));
            okx[j] = ok;
            walk_step(xwk[j]);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < G_PER; ++j) {
            const int idx = t + 256 * j;
            float4 v = rg[j];
            if (SCALE) v = mul4(v, rgs[j]);
            v = keep4(okg[j], v);
            if (idx < BK * BM / 4)
                *reinterpret_cast<float4*>(Gs + ((buf * BK + g_pr[j]) * LDM + (idx % (BM / 4)) * 4)) = v;
        }
#pragma unroll
        for (int j = 0; j < X_PER; ++j) {
            const int idx = t + 256 * j;
            float4 v = rx[j];
            if (SCALE) v = mul4(v, rxs[j]);
            v = keep4(okx[j], v);
            if (idx < BK * BN / 4) *reinterpret_cast<float4*>(Xs + ((buf * BK + x_pr[j]) * LDN + x_col[j])) = v;
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

    const int64_t nsteps = (pend - pbeg + BK - 1) / BK;
    // same pipeline as the forward kernel: registers hold step s+1 at the top (-> LDS[buf^1]), the loads of step s+2
    // are issued before the MFMAs of step s; rows beyond `pend` are predicated off, so no bounds branch
    gload();
    lstore(0);
    gload();
    __syncthreads();
    for (int64_t s = 0; s < nsteps; ++s) {
        const int buf = (int)(s & 1);
        lstore(buf ^ 1);
        gload();
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            float av[MT], bv[NT];
#pragma unroll
            for (int a = 0; a < MT; ++a) av[a] = Gs[(buf * BK + lh * 8 + kk) * LDM + (wm * MT + a) * 32 + li];
#pragma unroll
            for (int b = 0; b < NT; ++b) bv[b] = Xs[(buf * BK + lh * 8 + kk) * LDN + (wn * NT + b) * 32 + li];
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        const int k = n0 + (wn * NT + b) * 32 + li;
        if (k >= Ktot) continue;
#pragma unroll
        for (int a = 0; a < MT; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = o0 + (wm * MT + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (o < p.Cout) atomicAdd(&gw[(int64_t)o * Ktot + k], acc[a][b][r] * p.gain);
            }
        }
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

template <int WM, int WN, int MT, int NT>
int launch_fwd_cfg(void* y, const void* x, const void* wmat, const float* in_scale, const float* out_scale,
                   const float* bias, const void* resid, const ideas_conv_params* p, hipStream_t stream) {
    constexpr int BM_ = WM * MT * 32, BN_ = WN * NT * 32;
    const int64_t M = (int64_t)p->B * p->OH * p->OW;
    const int64_t tm = ideas_cdiv(M, BM_);
    const int tn = (int)ideas_cdiv(p->Cout, BN_);
    if (tm * tn > 0x7fffffffLL) return IDEAS_E_SHAPE;
    auto go = [&](auto sc, auto rf) {
        if (p->Cin >= BK)
            hipLaunchKernelGGL((conv_igemm_kernel<WM, WN, MT, NT, decltype(sc)::value, decltype(rf)::value, true>),
                               dim3((unsigned)(tm * tn)), dim3(256), 0, stream, (float*)y, (const float*)x,
                               (const float*)wmat, in_scale, out_scale, bias, (const float*)resid, *p, tn);
        else
            hipLaunchKernelGGL((conv_igemm_kernel<WM, WN, MT, NT, decltype(sc)::value, decltype(rf)::value, false>),
                               dim3((unsigned)(tm * tn)), dim3(256), 0, stream, (float*)y, (const float*)x,
                               (const float*)wmat, in_scale, out_scale, bias, (const float*)resid, *p, tn);
    };
    using T = std::true_type;
    using F = std::false_type;
    if (in_scale) { if (p->reflect) go(T{}, T{}); else go(T{}, F{}); }
    else { if (p->reflect) go(F{}, T{}); else go(F{}, F{}); }
    return ideas_launch_status();
}

template <int WM, int WN, int MT, int NT>
int launch_wgrad_cfg(float* gw, const void* gy, const void* x, const float* in_scale, const float* out_scale,
                     const ideas_conv_params* p, hipStream_t stream) {
    constexpr int BM_ = WM * MT * 32, BN_ = WN * NT * 32;
    const int64_t P = (int64_t)p->B * p->OH * p->OW;
    const int Ktot = p->TY * p->TX * p->Cin;
    const int tm = (int)ideas_cdiv(p->Cout, BM_);
    const int tn = (int)ideas_cdiv(Ktot, BN_);
    const int64_t tiles = (int64_t)tm * tn;
    // split-K sizing: the grid should be a whole number of "waves" of resident blocks, otherwise the last,
    // partially filled wave costs as much as a full one (1026 blocks on 1024 slots ran 2x the time).
    const bool sc_ = in_scale && out_scale;
    static int occ_cache[2] = {0, 0};
    if (!occ_cache[sc_]) {
        int occ = 0;
        hipError_t e = sc_ ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, conv_wgrad_kernel<WM, WN, MT, NT, true, false, true>, 256, 0)
                           : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, conv_wgrad_kernel<WM, WN, MT, NT, false, false, true>, 256, 0);
        occ_cache[sc_] = (e == hipSuccess && occ > 0) ? occ : 2;
    }
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    const int64_t slots = (int64_t)occ_cache[sc_] * n_cu;
    const int64_t max_splits = ideas_cdiv(P, 16 * BK);          // keep >= 16 pipeline steps per block
    int64_t splits = (2 * slots) / tiles;                        // aim at two full waves of blocks
    if (splits < 1) splits = 1;
    if (splits > max_splits) splits = max_splits;
    if (splits > 65535) splits = 65535;
    int64_t per = ideas_cdiv(ideas_cdiv(P, splits), BK) * BK;
    splits = ideas_cdiv(P, per);
    // rounding `per` up to BK can shave a split off; re-balance so the block count stays a multiple of the slots
    {
        const int64_t blocks = tiles * splits;
        const int64_t waves = blocks / slots;
        if (waves >= 1 && blocks % slots) {
            const int64_t want = (waves * slots) / tiles;        // largest split count giving whole waves
            if (want >= 1) {
                per = ideas_cdiv(ideas_cdiv(P, want), BK) * BK;
                splits = ideas_cdiv(P, per);
            }
        }
    }
    auto go = [&](auto sc, auto rf) {
        if (p->OW >= BK)
            hipLaunchKernelGGL((conv_wgrad_kernel<WM, WN, MT, NT, decltype(sc)::value, decltype(rf)::value, true>),
                               dim3((unsigned)tiles, (unsigned)splits), dim3(256), 0, stream, gw, (const float*)gy,
                               (const float*)x, in_scale, out_scale, *p, tn, per);
        else
            hipLaunchKernelGGL((conv_wgrad_kernel<WM, WN, MT, NT, decltype(sc)::value, decltype(rf)::value, false>),
                               dim3((unsigned)tiles, (unsigned)splits), dim3(256), 0, stream, gw, (const float*)gy,
                               (const float*)x, in_scale, out_scale, *p, tn, per);
    };
    using T = std::true_type;
    using F = std::false_type;
    const bool sc = sc_;
    if (sc) { if (p->reflect) go(T{}, T{}); else go(T{}, F{}); }
    else { if (p->reflect) go(F{}, T{}); else go(F{}, F{}); }
    return ideas_launch_status();
}

}  // namespace

extern "C" int ideas_conv_igemm(void* y, const void* x, const void* wmat, const float* in_scale, const float* out_scale,
                                const float* bias, const void* resid, const ideas_conv_params* p, int dtype,
                                void* stream_) {
    if (dtype != IDEAS_F32 && dtype != IDEAS_F32_B3 && dtype != IDEAS_BF16) return IDEAS_E_UNSUPPORTED;
    if (!y || !x || !wmat) return IDEAS_E_NULL;
    int rc = check_conv(p);
    if (rc) return rc;
    if (dtype == IDEAS_BF16) {     // bf16 x / y / resid; wmat = pack(s) of ideas_bf16_pack_weights: with an in_scale pointer one pack
                                   // PER SAMPLE made with that very scale (the pointer only selects the per-sample mode here)
        if (!ideas_bf16_conv_supported(p, in_scale != nullptr)) return IDEAS_E_UNSUPPORTED;
        if (!ideas_aligned16(x) || !ideas_aligned16(wmat) || !ideas_aligned16(y) || (in_scale && !ideas_aligned16(in_scale)) ||
            (out_scale && !ideas_aligned16(out_scale)) || (bias && !ideas_aligned16(bias)) || (resid && !ideas_aligned16(resid)))
            return IDEAS_E_ALIGN;
        return ideas_bf16_fwd(y, x, wmat, in_scale != nullptr, out_scale, bias, resid, p, (hipStream_t)stream_);
    }
    if (p->Cin % 4) return IDEAS_E_ALIGN;
    if (!ideas_aligned16(x) || !ideas_aligned16(wmat) || (in_scale && !ideas_aligned16(in_scale))) return IDEAS_E_ALIGN;
    hipStream_t stream = (hipStream_t)stream_;
    if (dtype == IDEAS_F32_B3) {   // wmat = bf16 planes of ideas_b3_split_weights
        if (!ideas_b3_conv_supported(p)) return IDEAS_E_UNSUPPORTED;
        return ideas_b3_fwd(y, x, wmat, in_scale, out_scale, bias, resid, p, stream);
    }
    if (p->Cout > 64) return launch_fwd_cfg<2, 2, 2, 2>(y, x, wmat, in_scale, out_scale, bias, resid, p, stream);  // 128x128
    if (p->Cout > 32) return launch_fwd_cfg<2, 2, 2, 1>(y, x, wmat, in_scale, out_scale, bias, resid, p, stream);  // 128x64
    return launch_fwd_cfg<4, 1, 1, 1>(y, x, wmat, in_scale, out_scale, bias, resid, p, stream);                    // 128x32
}

extern "C" int ideas_conv_igemm_multi(int n, void* y, const void* x, const void* const* wmat, const float* in_scale, const float* out_scale,
                                      const ideas_conv_params* params, int dtype, void* stream_) {
    if (dtype != IDEAS_F32_B3 && dtype != IDEAS_BF16) return IDEAS_E_UNSUPPORTED;
    if (!y || !x || !wmat || !params) return IDEAS_E_NULL;
    if (n < 1 || n > 4) return IDEAS_E_SHAPE;
    if (!ideas_aligned16(x) || (in_scale && !ideas_aligned16(in_scale))) return IDEAS_E_ALIGN;
    for (int i = 0; i < n; ++i) {
        const ideas_conv_params* p = params + i;
        int rc = check_conv(p);
        if (rc) return rc;
        if (!wmat[i]) return IDEAS_E_NULL;
        if (!ideas_aligned16(wmat[i])) return IDEAS_E_ALIGN;
        if (p->reflect || p->act || p->accumulate) return IDEAS_E_UNSUPPORTED;
        if (dtype == IDEAS_BF16 ? !ideas_bf16_conv_supported(p, in_scale != nullptr) : !ideas_b3_conv_supported(p)) return IDEAS_E_UNSUPPORTED;
        // one tensor pair: same x, same y, same channel counts
        if (p->B != params->B || p->IH != params->IH || p->IW != params->IW || p->Cin != params->Cin || p->YH != params->YH ||
            p->YW != params->YW || p->Cout != params->Cout)
            return IDEAS_E_SHAPE;
    }
    if (dtype == IDEAS_BF16)      // bf16 activations; wmat[i] = the launches' bf16 packs (per sample when in_scale is given: only its presence matters here)
        return ideas_bf16_fwd_multi(n, y, x, wmat, in_scale != nullptr, out_scale, params, (hipStream_t)stream_);
    return ideas_b3_fwd_multi(n, y, x, wmat, in_scale, out_scale, params, (hipStream_t)stream_);
}

extern "C" int ideas_conv_wgrad(float* gw, const void* gy, const void* x, const float* in_scale, const float* out_scale,
                                const ideas_conv_params* p, int dtype, void* stream_) {
    if (dtype != IDEAS_F32 && dtype != IDEAS_F32_B3 && dtype != IDEAS_BF16) return IDEAS_E_UNSUPPORTED;
    if (!gw || !gy || !x) return IDEAS_E_NULL;
    int rc = check_conv(p);
    if (rc) return rc;
    if (p->Cin % 4 || p->Cout % 4) return IDEAS_E_ALIGN;
    if ((in_scale == nullptr) != (out_scale == nullptr)) return IDEAS_E_UNSUPPORTED;  // both or neither
    if (dtype == IDEAS_BF16) {     // bf16 gy / x, f32 gw and scales
        if (!ideas_bf16_wgrad_supported(p, in_scale != nullptr)) return IDEAS_E_UNSUPPORTED;
        if (!ideas_aligned16(x) || !ideas_aligned16(gy)) return IDEAS_E_ALIGN;
        return ideas_bf16_wgrad(gw, gy, x, in_scale, out_scale, p, (hipStream_t)stream_);
    }
    if (!ideas_aligned16(x) || !ideas_aligned16(gy) || (in_scale && !ideas_aligned16(in_scale)) ||
        (out_scale && !ideas_aligned16(out_scale)))
        return IDEAS_E_ALIGN;
    if ((int64_t)p->B * p->OH * p->OW >= 0x7fffffffLL) return IDEAS_E_SHAPE;
    hipStream_t stream = (hipStream_t)stream_;
    if (dtype == IDEAS_F32_B3 && ideas_b3_pw_wgrad_ok(p, in_scale, out_scale))
        return ideas_b3_pw_wgrad(gw, gy, x, p, stream);                          // 1x1 / stride 1: flat reduction over the pixels (conv_b3_pw.hip)
    if (dtype == IDEAS_F32_B3 && ideas_b3_wgrad3_enabled() && ideas_b3_wgrad3_supported(p))
        return ideas_b3_wgrad3(gw, gy, x, in_scale, out_scale, p, stream);      // 3x3: tap-fused, rolling window (conv_b3_wgrad3.hip)
    if (dtype == IDEAS_F32_B3 && ideas_b3_wgrad_supported(p)) return ideas_b3_wgrad(gw, gy, x, in_scale, out_scale, p, stream);
    if (p->Cout > 64) return launch_wgrad_cfg<2, 2, 2, 2>(gw, gy, x, in_scale, out_scale, p, stream);  // 128 (o) x 128 (k)
    if (p->Cout > 32) return launch_wgrad_cfg<2, 2, 1, 2>(gw, gy, x, in_scale, out_scale, p, stream);  // 64 x 128
    return launch_wgrad_cfg<1, 4, 1, 1>(gw, gy, x, in_scale, out_scale, p, stream);                    // 32 x 128
}
