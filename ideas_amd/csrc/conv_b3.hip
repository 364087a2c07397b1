// Implicit-GEMM convolution on the gfx950 bf16 matrix pipe with f32-class accuracy ("bf16x3"), NHWC, f32 in/out.
//
// Same GEMM view and epilogue as conv_igemm.hip (which runs the contraction on v_mfma_f32_32x32x2_f32 = 1/16 of
// the bf16 MFMA rate).  Here every f32 operand is split into three bf16 planes
//
//        x = hi + mid + lo        hi = rne_bf16(x),  mid = rne_bf16(x - hi),  lo = x - hi - mid
//
// The split is EXACT (both residuals are exact f32 subtractions, and lo has <= 8 significant bits left), so
//        x*w = sum over the 9 plane pairs, each an exact bf16 x bf16 product accumulated in f32 by the MFMA.
// Six pairs are contracted (hi.hi, hi.mid, mid.hi, mid.mid, hi.lo, lo.hi) with v_mfma_f32_32x32x16_bf16; the three
// dropped pairs (mid.lo, lo.mid, lo.lo) are each <= 2^-24 |x*w|, the size of the f32 rounding of the product itself.
// Measured against f64 (tools/gemm_bf16x3.hip, tools/check_b3.py): max / rms error equal to the exact f32 MFMA
// kernel's (2.2-2.8e-7 / 2.5e-8 of sum|x*w| against 2.0-3.0e-7 / 2.5-3.2e-8).  Six bf16 MFMAs replace sixteen f32 ones.
//
// The matrix pipe is then no longer the only limiter: a SIMD issues the split arithmetic through the same VALU port
// as the MFMAs, so the kernel is built to keep everything else OFF the vector ALU:
//   * weights are split once per launch by ideas_b3_split_weights into step-major planes [3][K/16][Cout][16] bf16
//     (K-steps ordered (ci/16, ty, tx): taps innermost, so a pixel neighbourhood is revisited in consecutive steps):
//     the B tile of a K-step is one contiguous 4 KB block, copied global -> LDS without any arithmetic;
//   * activations are fetched with raw buffer loads: 32-bit offsets, and padding taps get an out-of-range offset,
//     for which the hardware returns zeros (no predication, no zero-fill selects);
//   * the (tap, ci) walk is block-uniform -> scalar ALU; per row a K-step costs a bit-field extract of the row's
//     precomputed invalid-tap mask, an add and an or;
//   * rows beyond M and columns beyond Cout are not masked at all: their operands are duplicates / whatever the
//     buffer returns, and their results are simply not stored (rows and columns of a GEMM are independent).
// What is left on the VALU per 16 x 16-element A chunk: 6 cvt_pk + 8 bit ops + 8 subtractions (+4 mul for the
// modulation scale).
//
// LDS: per pipeline buffer three A planes [BM][16 bf16] and three B planes [BN][16 bf16]; 32-byte rows with the two
// 16-byte halves swapped on rows with bit 3 set: ds_write_b64 (staging), ds_write_b128 (weights) and the
// ds_read_b128 operand fetch are all conflict-free (checked by enumeration against the bank rules of
// MI355X_MICROARCH.md).  Lane (i = lane&31, h = lane>>5) of a 32x32x16 MFMA holds K values [8h, 8h+8) of row i.
#include "b3.hpp"
#include <type_traits>

namespace {

// ---------------------------------------------------------------------------------------------------------------
// weights: f32 [Cout][K] (K = (ty,tx,ci) contiguous)  ->  bf16 planes [3][K/16][Cout][16], K-step = (ci/16, ty, tx)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void split_weights_kernel(uint2* __restrict__ dst, const float4* __restrict__ w, int Cout, int K,
                                                            int Cin) {
    const int64_t n4 = (int64_t)Cout * (K / 4);
    const int64_t plane = (int64_t)Cout * K / 4;   // uint2 (4 bf16) units per plane
    const int ntaps = K / Cin;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int k = (int)(i % (K / 4)) * 4;      // k = tap * Cin + ci in the f32 matrix
        const int n = (int)(i / (K / 4));
        const int tap = k / Cin, ci = k - tap * Cin;
        const int step = (ci >> 4) * ntaps + tap;  // K-steps run over the taps of one 16-channel chunk, then the next chunk
        const Split4 s = split4(w[i]);
        const int64_t o = ((int64_t)step * Cout + n) * 4 + ((ci & 15) >> 2);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) dst[pl * plane + o] = s.p[pl];
    }
}

// The same planes read straight from the parameter: element (n, ty, tx, ci) of the launch's matrix at
// w[n*sn + ty*sty + tx*stx + ci*sc] (see ideas_bf16_pack_weights_strided, conv_bf16.hip).
template <bool UNIT>
__device__ __forceinline__ void split_weights_strided_body(uint2* __restrict__ dst, const float* __restrict__ w, int Cout, int TY, int TX,
                                                           int Cin, int64_t sn, int64_t sty, int64_t stx, int64_t sc, int64_t bid,
                                                           int64_t nblk) {
    const int c4 = Cin / 4, ntaps = TY * TX;
    const int64_t n4 = (int64_t)Cout * ntaps * c4;
    const int64_t plane = n4;
    for (int64_t i = bid * 256 + threadIdx.x; i < n4; i += nblk * 256) {
        int n, tap, ci;
        if (UNIT) {
            ci = (int)(i % c4) * 4;
            tap = (int)((i / c4) % ntaps);
            n = (int)(i / ((int64_t)c4 * ntaps));
        } else {
            n = (int)(i % Cout);
            ci = (int)((i / Cout) % c4) * 4;
            tap = (int)(i / ((int64_t)Cout * c4));
        }
        const int ty = tap / TX, tx = tap - ty * TX;
        const float* src = w + n * sn + ty * sty + tx * stx + ci * sc;
        const float4 v = UNIT ? *reinterpret_cast<const float4*>(src) : make_float4(src[0], src[sc], src[2 * sc], src[3 * sc]);
        const int step = (ci >> 4) * ntaps + tap;
        const Split4 s = split4(v);
        const int64_t o = ((int64_t)step * Cout + n) * 4 + ((ci & 15) >> 2);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) dst[pl * plane + o] = s.p[pl];
    }
}

template <bool UNIT>
__global__ __launch_bounds__(256) void split_weights_strided_kernel(uint2* __restrict__ dst, const float* __restrict__ w, int Cout,
                                                                    int TY, int TX, int Cin, int64_t sn, int64_t sty, int64_t stx,
                                                                    int64_t sc) {
    split_weights_strided_body<UNIT>(dst, w, Cout, TY, TX, Cin, sn, sty, stx, sc, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(256) void split_weights_batched_kernel(const ideas_prep_desc* __restrict__ tbl, int n) {
    int local, nblk;
    const ideas_prep_desc* d = prep_lookup(tbl, n, local, nblk);
    if (d->unit) split_weights_strided_body<true>((uint2*)d->dst, d->w, d->a[0], d->a[1], d->a[2], d->a[3], d->s[0], d->s[1], d->s[2], d->s[3], local, nblk);
    else split_weights_strided_body<false>((uint2*)d->dst, d->w, d->a[0], d->a[1], d->a[2], d->a[3], d->s[0], d->s[1], d->s[2], d->s[3], local, nblk);
}

// ---------------------------------------------------------------------------------------------------------------
// forward family
// ---------------------------------------------------------------------------------------------------------------
template <int WM, int WN, int MT, int NT, bool SCALE, bool REFLECT>
__device__ __forceinline__ void conv_b3_body(float* __restrict__ y, const float* __restrict__ x,
                                             const void* __restrict__ wplanes,
                                             const float* __restrict__ in_scale,
                                             const float* __restrict__ out_scale,
                                             const float* __restrict__ bias,
                                             const float* __restrict__ resid, const ideas_conv_params& p,
                                             int tile_m, int tile_n, unsigned x_bytes, unsigned plane_bytes) {
    static_assert(WM * WN == 4, "4 waves per block");
    constexpr int BM = WM * MT * 32;
    constexpr int BN = WN * NT * 32;
    static_assert(BM % 64 == 0 && BM * 4 % 256 == 0, "A rows are staged 64 at a time by all 256 threads");
    constexpr int A_PER = BM * 4 / 256;
    constexpr int PLANE_A = BM * ROWB, PLANE_B = BN * ROWB;
    constexpr int BUF = 3 * (PLANE_A + PLANE_B);
    static_assert(2 * BUF >= BM * 12, "epilogue row table must fit");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF];

    const int t = threadIdx.x;
    const int64_t M = (int64_t)p.B * p.OH * p.OW;
    const int K = p.TY * p.TX * p.Cin;
    const int64_t m0 = (int64_t)tile_m * BM;
    const int n0 = tile_n * BN;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)x_bytes, (int)RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)wplanes, 0, (int)(3u * plane_bytes), (int)RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc((void*)in_scale, 0, SCALE ? p.B * p.Cin * 4 : 0, (int)RSRC_FLAGS);

    // ---- per-thread A rows ------------------------------------------------------------------------------------
    const int kq = t & 3;
    unsigned a_rowbase[A_PER];   // byte offset of (b, iy0, ix0, ci = 4 kq); arithmetic is mod 2^32, valid taps land < x_bytes
    unsigned a_inv[A_PER];       // bit tap = 1 -> tap (ty, tx) of this row lies in the zero padding
    unsigned a_sbase[A_PER];     // byte offset of in_scale[b][4 kq]
    int a_iyb[A_PER], a_ixb[A_PER], a_img[A_PER];   // REFLECT only
#pragma unroll
    for (int j = 0; j < A_PER; ++j) {
        const int r = (t >> 2) + 64 * j;
        int64_t m = m0 + r;
        m = m < M ? m : M - 1;                     // rows past M repeat the last row; their results are not stored
        const int ox = (int)(m % p.OW);
        const int64_t q = m / p.OW;
        const int oy = (int)(q % p.OH);
        const int b = (int)(q / p.OH);
        const int iyb = oy * p.sy + p.offy, ixb = ox * p.sx + p.offx;
        a_rowbase[j] = (unsigned)(((b * p.IH + iyb) * p.IW + ixb) * p.Cin + kq * 4) * 4u;
        a_sbase[j] = (unsigned)(b * p.Cin + kq * 4) * 4u;
        a_iyb[j] = iyb; a_ixb[j] = ixb; a_img[j] = b * p.IH;
        unsigned inv = 0;
        if (!REFLECT) {
            for (int ty = 0; ty < p.TY; ++ty)
                for (int tx = 0; tx < p.TX; ++tx) {
                    const int iy = iyb + ty * p.dy, ix = ixb + tx * p.dx;
                    const bool ok = iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
                    inv |= (ok ? 0u : 1u) << (ty * p.TX + tx);
                }
        }
        a_inv[j] = inv;
    }
    // B: thread t copies 16 bytes (row t>>1, half t&1) of each plane's contiguous [BN][16] tile
    const int brow = (t >> 1) % BN;                // BN < 128: the upper threads repeat rows (same value, same address)
    const unsigned b_voff = (unsigned)(n0 * 32 + brow * 32 + (t & 1) * 16);
    const int b_lds = 3 * PLANE_A + brow * ROWB + (((t & 1) ^ ((brow >> 3) & 1)) << 4);
    int a_lds[A_PER];
#pragma unroll
    for (int j = 0; j < A_PER; ++j) {
        const int r = (t >> 2) + 64 * j;
        a_lds[j] = r * ROWB + ((kq * 8) ^ (((r >> 3) & 1) << 4));
    }

    // block-uniform walk over K = (ty, tx, ci): scalar registers
    int k_ci = 0, k_tx = 0, k_ty = 0, k_tap = 0;

    // A: two register stages (the loads of tile t+2 are issued before tile t+1 is split, which takes the whole step);
    // B: one (it is stored to LDS untouched at the top of a step, then immediately re-loaded)
    struct Stage { float4 a[A_PER], s[A_PER]; };
    Stage st0, st1;
    uint4 rb[3];
    auto gloadA = [&](Stage& st) {
        const unsigned tapoff = (unsigned)(((k_ty * p.dy) * p.IW + k_tx * p.dx) * p.Cin + k_ci) * 4u;   // uniform
#pragma unroll
        for (int j = 0; j < A_PER; ++j) {
            unsigned off;
            if (REFLECT) {
                const int iy = reflect_coord(a_iyb[j] + k_ty * p.dy, p.IH), ix = reflect_coord(a_ixb[j] + k_tx * p.dx, p.IW);
                off = (unsigned)(((a_img[j] + iy) * p.IW + ix) * p.Cin + k_ci + kq * 4) * 4u;
            } else {
                const unsigned inv = (unsigned)__builtin_amdgcn_sbfe(a_inv[j], k_tap, 1);   // 0 or 0xffffffff
                off = (a_rowbase[j] + tapoff) | inv;   // (tiles past K: any offset is safe, in range or zero-filled)
            }
            st.a[j] = buffer_load4(rx, off, 0);
            if (SCALE) st.s[j] = buffer_load4(rs_, a_sbase[j], (unsigned)k_ci * 4u);
        }
        // K order: all taps of one 16-channel chunk, then the next chunk.  The 9 visits to a pixel neighbourhood are then
        // 9 consecutive steps (L1/L2 hits) instead of three passes over the image a third of the kernel apart (tap-major
        // order re-fetched the input from HBM once per ky: rocprofv3 FETCH_SIZE 3.6x the tensor)
        // (pure arithmetic: hipcc turns uniform selects back into scalar BRANCHES, which cut the K loop's scheduling region)
        const int nx = k_tx + 1, gx = 1 - (int)((unsigned)(nx - p.TX) >> 31);   // gx = nx >= TX
        k_tx = nx - gx * p.TX;
        const int ny = k_ty + gx, gy = 1 - (int)((unsigned)(ny - p.TY) >> 31);
        k_ty = ny - gy * p.TY;
        k_tap = (k_tap + 1) * (1 - gy);
        k_ci += gy * BK;
    };
    auto gloadB = [&](int kt) {
        const unsigned wsoff = (unsigned)kt * (unsigned)p.Cout * 32u;   // uniform
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            const float4 v = buffer_load4(rw, b_voff, wsoff + (unsigned)pl * plane_bytes);
            rb[pl] = make_uint4(__builtin_bit_cast(unsigned, v.x), __builtin_bit_cast(unsigned, v.y),
                                __builtin_bit_cast(unsigned, v.z), __builtin_bit_cast(unsigned, v.w));
        }
    };
    auto lstoreB = [&](int buf) {
        unsigned char* base = smem + buf * BUF;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint4*>(base + b_lds + pl * PLANE_B) = rb[pl];
    };
    auto lstoreA = [&](int buf, const Stage& st) {
        unsigned char* base = smem + buf * BUF;
#pragma unroll
        for (int j = 0; j < A_PER; ++j) {
            float4 v = st.a[j];
            // rounded to f32 BEFORE the split (no FMA contraction into the residual), as in the f32 kernel
            if (SCALE) v = make_float4(mul_rn(v.x, st.s[j].x), mul_rn(v.y, st.s[j].y), mul_rn(v.z, st.s[j].z), mul_rn(v.w, st.s[j].w));
            const Split4 s = split4(v);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint2*>(base + a_lds[j] + pl * PLANE_A) = s.p[pl];
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

    // Pipeline, one barrier per K-step:  step t = { issue loads of tile t+2 -> stage X;  fragments of tile t from
    // LDS[t&1];  split + store tile t+1 (stage Y, loaded a whole step ago) -> LDS[(t+1)&1];  6*MT*NT MFMAs }.
    // Tiles beyond K are fetched out of range (zeros) and never consumed: no bounds branch anywhere.
    auto step = [&](int kt, Stage& ld, const Stage& stg) {
        const int buf = kt & 1;
        lstoreB(buf ^ 1);
        gloadB(kt + 2);
        gloadA(ld);
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
        lstoreA(buf ^ 1, stg);
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][PA[q]], fb[b][PB[q]], acc[a][b], 0, 0, 0);
        __syncthreads();
    };
    const int nk = K / BK;
    gloadA(st0);
    gloadB(0);
    lstoreA(0, st0);
    lstoreB(0);
    gloadA(st1);
    gloadB(1);
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        step(kt, st0, st1);       // tile kt+2 -> st0 while tile kt+1 (st1) is split into LDS
        step(kt + 1, st1, st0);
    }
    if (kt < nk) step(kt, st0, st1);

    // ---- epilogue (identical to conv_igemm_kernel) -------------------------------------------------------------
    int64_t* row_off = reinterpret_cast<int64_t*>(smem);
    int* row_b = reinterpret_cast<int*>(smem + 8 * BM);
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
                float v = mul_rn(acc[a][b][r], p.gain);
                if (out_scale) v = mul_rn(v, out_scale[(int64_t)row_b[row] * p.Cout + n]);
                v = mul_then_add(v, 1.0f, bv);
                if (p.act) v = (v > 0.f ? v : v * p.alpha) * p.act_gain;
                if (resid) v = (v + resid[off + n]) * p.resid_gain;
                if (p.accumulate) y[off + n] += v; else y[off + n] = v;
            }
        }
    }
}

template <int WM, int WN, int MT, int NT, bool SCALE, bool REFLECT>
__global__ __launch_bounds__(256, 3) void conv_b3_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                         const void* __restrict__ wplanes,
                                                         const float* __restrict__ in_scale,
                                                         const float* __restrict__ out_scale,
                                                         const float* __restrict__ bias,
                                                         const float* __restrict__ resid, ideas_conv_params p,
                                                         int tiles_n, unsigned x_bytes, unsigned plane_bytes) {
    const int swz = xcd_swizzle(blockIdx.x, gridDim.x);
    conv_b3_body<WM, WN, MT, NT, SCALE, REFLECT>(y, x, wplanes, in_scale, out_scale, bias, resid, p, swz / tiles_n, swz % tiles_n, x_bytes,
                                                 plane_bytes);
}

// Several launches of the family in ONE grid: the output-parity phases of a stride-2 input gradient / transposed conv
// (op/conv_plan.py::plan_dgrad: 4 / 2 / 2 / 1 taps), which share x, y and the per-sample scales and differ in their geometry and
// weight planes.  Block order is launch-major, heaviest launch first (the caller's order: 4, 2, 2, 1 taps), each launch's tiles
// in their own XCD-banded order: the same schedule as back-to-back launches minus the grid ramp and tail of each, which is what
// the ~16 K-step launches of the small layers consist of (E.texture.1's input gradient: 26 -> 57 TFLOP/s).  Measured alternative
// that LOST: interleaving the launches' tiles of one image region (shared activation fetch) -- 160 -> 132 TFLOP/s on
// G.layers.7.conv1, 158 -> 135 on Dreal.1.conv2's input gradient.
struct B3Multi {
    ideas_conv_params p[4];
    const void* w[4];
    int off[5];          // first block of launch i (off[n] = grid size)
    int n, tiles_n;
};

template <int WM, int WN, int MT, int NT, bool SCALE>
__global__ __launch_bounds__(256, 3) void conv_b3_multi_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                               const float* __restrict__ in_scale,
                                                               const float* __restrict__ out_scale, B3Multi a, unsigned x_bytes) {
    const int bid = blockIdx.x;
    const int which = (bid >= a.off[1] && a.n > 1) + (bid >= a.off[2] && a.n > 2) + (bid >= a.off[3] && a.n > 3);
    const int swz = xcd_swizzle(bid - a.off[which], a.off[which + 1] - a.off[which]);
    const int tile_m = swz / a.tiles_n, tile_n = swz - tile_m * a.tiles_n;
    const ideas_conv_params& p = a.p[which];                          // block-uniform index: scalar loads from the kernel arguments
    conv_b3_body<WM, WN, MT, NT, SCALE, false>(y, x, a.w[which], in_scale, out_scale, nullptr, nullptr, p, tile_m, tile_n, x_bytes,
                                               (unsigned)((int64_t)p.TY * p.TX * p.Cin * p.Cout * 2));
}

template <int WM, int WN, int MT, int NT>
int launch_b3_multi_cfg(int n, void* y, const void* x, const void* const* wplanes, const float* in_scale, const float* out_scale,
                        const ideas_conv_params* ps, hipStream_t stream) {
    constexpr int BM_ = WM * MT * 32, BN_ = WN * NT * 32;
    B3Multi a;
    a.n = n;
    a.tiles_n = (int)ideas_cdiv(ps[0].Cout, BN_);
    int64_t blocks = 0;
    for (int i = 0; i < n; ++i) {
        a.p[i] = ps[i];
        a.w[i] = wplanes[i];
        a.off[i] = (int)blocks;
        blocks += ideas_cdiv((int64_t)ps[i].B * ps[i].OH * ps[i].OW, BM_) * a.tiles_n;
        if (blocks > 0x7fffffffLL) return IDEAS_E_SHAPE;
    }
    for (int i = n; i < 4; ++i) { a.p[i] = ps[0]; a.w[i] = wplanes[0]; a.off[i] = (int)blocks; }
    a.off[n] = (int)blocks;
    a.off[4] = (int)blocks;
    const unsigned x_bytes = (unsigned)((int64_t)ps[0].B * ps[0].IH * ps[0].IW * ps[0].Cin * 4);
    if (in_scale)
        hipLaunchKernelGGL((conv_b3_multi_kernel<WM, WN, MT, NT, true>), dim3((unsigned)blocks), dim3(256), 0, stream, (float*)y,
                           (const float*)x, in_scale, out_scale, a, x_bytes);
    else
        hipLaunchKernelGGL((conv_b3_multi_kernel<WM, WN, MT, NT, false>), dim3((unsigned)blocks), dim3(256), 0, stream, (float*)y,
                           (const float*)x, in_scale, out_scale, a, x_bytes);
    return ideas_launch_status();
}

template <int WM, int WN, int MT, int NT>
int launch_b3_cfg(void* y, const void* x, const void* wplanes, const float* in_scale, const float* out_scale,
                  const float* bias, const void* resid, const ideas_conv_params* p, hipStream_t stream) {
    constexpr int BM_ = WM * MT * 32, BN_ = WN * NT * 32;
    const int64_t M = (int64_t)p->B * p->OH * p->OW;
    const int64_t tm = ideas_cdiv(M, BM_);
    const int tn = (int)ideas_cdiv(p->Cout, BN_);
    if (tm * tn > 0x7fffffffLL) return IDEAS_E_SHAPE;
    const unsigned x_bytes = (unsigned)((int64_t)p->B * p->IH * p->IW * p->Cin * 4);
    const unsigned plane_bytes = (unsigned)((int64_t)p->TY * p->TX * p->Cin * p->Cout * 2);
    auto go = [&](auto sc, auto rf) {
        hipLaunchKernelGGL((conv_b3_kernel<WM, WN, MT, NT, decltype(sc)::value, decltype(rf)::value>),
                           dim3((unsigned)(tm * tn)), dim3(256), 0, stream, (float*)y, (const float*)x, wplanes,
                           in_scale, out_scale, bias, (const float*)resid, *p, tn, x_bytes, plane_bytes);
    };
    using T = std::true_type;
    using F = std::false_type;
    if (in_scale) { if (p->reflect) go(T{}, T{}); else go(T{}, F{}); }
    else { if (p->reflect) go(F{}, T{}); else go(F{}, F{}); }
    return ideas_launch_status();
}

}  // namespace

extern "C" int ideas_b3_conv_supported(const ideas_conv_params* p) {
    if (!p) return 0;
    return p->Cin % 16 == 0 && p->TY * p->TX <= 32 && (int64_t)p->B * p->IH * p->IW * p->Cin * 4 < 0xffffffffLL &&
           (int64_t)p->TY * p->TX * p->Cin * p->Cout * 6 < 0xffffffffLL;
}

// called by ideas_conv_igemm for dtype IDEAS_F32_B3 once the arguments are validated; `wplanes` comes from
// ideas_b3_split_weights
#ifndef SMALL_GRID_TILES
#define SMALL_GRID_TILES 160
#endif
int ideas_b3_fwd(void* y, const void* x, const void* wplanes, const float* in_scale, const float* out_scale,
                 const float* bias, const void* resid, const ideas_conv_params* p, hipStream_t stream) {
    // 1x1 layers with few input channels are bound by HBM, not by the matrix pipe: flat persistent GEMM (conv_b3_pw.hip)
    if (ideas_b3_pw_ok(p, in_scale, out_scale, resid)) return ideas_b3_pw_fwd(y, x, wplanes, bias, resid, p, stream);
    // 3x3 / stride 2 / no padding on a large grid: the LDS-image kernel stages every input pixel once per chunk instead of once per tap
    if (ideas_b3_s2img_ok(p, in_scale)) return ideas_b3_s2img_fwd(y, x, wplanes, in_scale, out_scale, bias, resid, p, stream);
    // few pixels, many channels (E's texture head on 4x4 .. 7x7 maps, Dco's last blocks on 8B patches): 128 x 128 tiles are fewer
    // than the CUs -- the 64-channel N tile doubles the blocks (E.texture.1 forward 64 -> 128 blocks)
    const int64_t t128 = ideas_cdiv((int64_t)p->B * p->OH * p->OW, 128) * ideas_cdiv(p->Cout, 128);
    if (p->Cout > 64 && t128 < SMALL_GRID_TILES) return launch_b3_cfg<2, 2, 2, 1>(y, x, wplanes, in_scale, out_scale, bias, resid, p, stream);
    if (p->Cout > 64) return launch_b3_cfg<2, 2, 2, 2>(y, x, wplanes, in_scale, out_scale, bias, resid, p, stream);  // 128x128
    if (p->Cout > 32) return launch_b3_cfg<2, 2, 2, 1>(y, x, wplanes, in_scale, out_scale, bias, resid, p, stream);  // 128x64
    return launch_b3_cfg<4, 1, 1, 1>(y, x, wplanes, in_scale, out_scale, bias, resid, p, stream);                    // 128x32
}

// ideas_conv_igemm_multi (conv_igemm.hip validates): n <= 4 launches sharing x / y / scales, dtype IDEAS_F32_B3
int ideas_b3_fwd_multi(int n, void* y, const void* x, const void* const* wplanes, const float* in_scale, const float* out_scale,
                       const ideas_conv_params* ps, hipStream_t stream) {
    const int cout = ps[0].Cout;
    ideas_conv_params strips[4];
    const void* strip_w[4];
    int nstrips = 0;
    const int rc = ideas_b3_fwd_tphase(n, y, x, wplanes, in_scale, out_scale, ps, stream, strips, strip_w, &nstrips);   // conv_b3_tphase.hip
    if (rc >= 0) {                                        // (-1 = not its geometry)
        if (rc != IDEAS_OK || nstrips == 0) return rc;
        n = nstrips; ps = strips; wplanes = strip_w;      // the last row / column of positions: generic kernel, one grid
    }
    if (cout > 64) return launch_b3_multi_cfg<2, 2, 2, 2>(n, y, x, wplanes, in_scale, out_scale, ps, stream);
    if (cout > 32) return launch_b3_multi_cfg<2, 2, 2, 1>(n, y, x, wplanes, in_scale, out_scale, ps, stream);
    return launch_b3_multi_cfg<4, 1, 1, 1>(n, y, x, wplanes, in_scale, out_scale, ps, stream);
}

void ideas_b3_split_batched(const ideas_prep_desc* tbl, int n, int blocks, hipStream_t stream) {
    hipLaunchKernelGGL(split_weights_batched_kernel, dim3(blocks), dim3(256), 0, stream, tbl, n);
}

extern "C" int ideas_b3_split_weights(void* planes, const void* wmat, int Cout, int K, int Cin, void* stream_) {
    if (!planes || !wmat) return IDEAS_E_NULL;
    if (Cout <= 0 || K <= 0 || Cin <= 0 || K % Cin) return IDEAS_E_SHAPE;
    if (Cin % 16 || !ideas_aligned16(planes) || !ideas_aligned16(wmat)) return IDEAS_E_ALIGN;
    const int64_t n4 = (int64_t)Cout * (K / 4);
    const int blocks = (int)(n4 / 256 + 1 < 2048 ? n4 / 256 + 1 : 2048);
    hipLaunchKernelGGL(split_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, (uint2*)planes,
                       (const float4*)wmat, Cout, K, Cin);
    return ideas_launch_status();
}

extern "C" int ideas_b3_split_weights_strided(void* planes, const float* w, int Cout, int TY, int TX, int Cin, int64_t sn, int64_t sty,
                                              int64_t stx, int64_t sc, void* stream_) {
    if (!planes || !w) return IDEAS_E_NULL;
    if (Cout <= 0 || TY <= 0 || TX <= 0 || Cin <= 0) return IDEAS_E_SHAPE;
    if (Cin % 16 || !ideas_aligned16(planes)) return IDEAS_E_ALIGN;
    const int64_t n4 = (int64_t)Cout * TY * TX * (Cin / 4);
    const int blocks = (int)(n4 / 256 + 1 < 2048 ? n4 / 256 + 1 : 2048);
    const bool unit = sc == 1 && ideas_aligned16(w) && sn % 4 == 0 && sty % 4 == 0 && stx % 4 == 0;
    if (unit)
        hipLaunchKernelGGL(split_weights_strided_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, (uint2*)planes, w, Cout,
                           TY, TX, Cin, sn, sty, stx, sc);
    else
        hipLaunchKernelGGL(split_weights_strided_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, (uint2*)planes, w, Cout,
                           TY, TX, Cin, sn, sty, stx, sc);
    return ideas_launch_status();
}
