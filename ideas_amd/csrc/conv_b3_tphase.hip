// The four output-parity phases of a 3x3 / stride-2 / pad-0 transposed convolution (= the input gradient of the 3x3 / stride-2
// convolution behind a Blur) in ONE pass over the input, on the bf16 matrix pipe with the exact 3-way split of b3.hpp.
//
//     y[2q+0, 2r+0] = sum_{jy,jx in {0,1}} W00[jy,jx] x[q-jy, r-jx]        (4 taps)       q, r index the INPUT grid (one more row /
//     y[2q+1, 2r+0] = sum_{jx}            W10[jx]    x[q,    r-jx]        (2 taps)       column than the input: the last phase-00
//     y[2q+0, 2r+1] = sum_{jy}            W01[jy]    x[q-jy, r   ]        (2 taps)       outputs read only x[q-1], x[r-1])
//     y[2q+1, 2r+1] =                     W11        x[q,    r   ]        (1 tap)
//
// conv_b3_multi_kernel runs these as four implicit GEMMs in one grid: every block stages (loads, scales, splits into three bf16
// planes, writes to LDS) its own 128-pixel operand tile for every tap, so an input element is fetched and split NINE times
// (4 + 2 + 2 + 1) -- 3 vector instructions per MFMA, 155-165 TFLOP/s.  Here a block owns a patch of 4 x 16 input-grid positions
// and all four phases of it: per 16-channel chunk the (4 + 1) x (16 + 1) input pixels are staged ONCE as a pixel-major LDS image
// (exactly the MFMA A-operand layout: a lane's 8 consecutive channels of one pixel), and the nine taps read it at the four
// pixel shifts (0,0) (0,-1) (-1,0) (-1,-1): a shifted window is just another row address.  108 MFMAs per wave and chunk on one
// staging pass (340 pixel-quads for 256 threads), one barrier per chunk.
//   waves: 4 = the 32-channel quarters of a 128-channel N tile; each holds 4 phases x 2 operands (2 x 16 positions each) = 8
//          accumulators (128 registers); two blocks per CU (a first version with 8-wave blocks, one per CU, gained 5 % instead
//          of 20 %).  The epilogue -- a block stores 4x the pixels it read -- is still not overlapped: see the measurements in the kernel;
//   B:     the per-phase weight planes of ideas_b3_split_weights_strided ([3][chunk * taps + tap][Cout][16] = the MFMA operand
//          layout) are fetched straight from global memory one tap group ahead, as conv_b3_wino.hip does;
//   LDS:   two buffers of 3 planes x 5 rows x 32 pixels x 32 B (30 KB; 17 pixels of a row in use); 16-byte halves of a pixel
//          swapped on pixels with bit 3 set.
// Requires the canonical pad-0 geometry of op/conv_plan.py::plan_dgrad (4 launches, taps 2x2 / 2x1 / 1x2 / 1x1, offsets 0,
// tap step -1), Cin % 16 == 0, Cout > 64 (smaller layers keep conv_b3_multi_kernel).
#include "b3.hpp"
#include <cstdlib>

namespace {

constexpr int PH = 4, PW = 16;                 // patch of input-grid positions
constexpr int SW = PW + 1, SP = (PH + 1) * SW; // staged pixels per chunk: 5 x 17 = 85
constexpr int LP = 32;                         // LDS pixels per image row: with this pitch (and the bit-3 half swap) the ds_read_b128 lane
                                               // groups {0-3,12-15,20-27} / {4-11,16-19,28-31} of two 16-position rows are conflict-free at every
                                               // shift; pitch 17 was 2-way on every operand read (enumerated against MI355X_MICROARCH.md, LDS)
constexpr int PLB = (PH + 1) * LP * ROWB;      // bytes per plane
constexpr int BUFB = 3 * PLB;

struct TPhase {
    const void* w[4];            // weight planes of the phases in canonical order (2x2, 2x1, 1x2, 1x1 taps)
    unsigned plane_bytes[4];
    int oh[4], ow[4], ooy[4], oox[4];
};

__device__ __forceinline__ int pix_off(int pix, int chunk16) { return pix * ROWB + ((chunk16 ^ ((pix >> 3) & 1)) << 4); }

template <bool SCALE>
__global__ __launch_bounds__(256, 2) void conv_b3_tphase_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                                const float* __restrict__ in_scale,
                                                                const float* __restrict__ out_scale, ideas_conv_params p, TPhase a,
                                                                int QH, int QW, int tiles_n, unsigned x_bytes) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUFB];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int ppr = (QW + PW - 1) / PW, ppi = ((QH + PH - 1) / PH) * ppr;
    const int swz = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tile_n = swz % tiles_n, tile_m = swz / tiles_n;
    const int pb = tile_m / ppi, prem = tile_m - pb * ppi;
    const int qy0 = (prem / ppr) * PH, qx0 = (prem % ppr) * PW;
    const int n0 = tile_n * 128;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)x_bytes, (int)RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc((void*)in_scale, 0, SCALE ? p.B * p.Cin * 4 : 0, (int)RSRC_FLAGS);
    __amdgpu_buffer_rsrc_t rw[4];
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) rw[ph] = __builtin_amdgcn_make_buffer_rsrc((void*)a.w[ph], 0, (int)(3u * a.plane_bytes[ph]), (int)RSRC_FLAGS);

    // ---- staging: task k = (pixel k >> 2, channel quad k & 3); every thread owns task t, the first 84 threads also task t + 256 ----
    const int quad = t & 3;
    unsigned goff[2], gmask[2];
    int lds_st[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pix = (t >> 2) + 64 * j;
        const bool live = pix < SP;
        const int r = live ? pix / SW : 0, c = live ? pix % SW : 0;
        const int iy = qy0 - 1 + r, ix = qx0 - 1 + c;
        const bool ok = live && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
        goff[j] = (unsigned)(((pb * p.IH + (ok ? iy : 0)) * p.IW + (ok ? ix : 0)) * p.Cin + quad * 4) * 4u;
        gmask[j] = ok ? 0u : 0xffffffffu;
        const int lp = r * LP + c;
        lds_st[j] = live ? lp * ROWB + ((quad * 8) ^ (((lp >> 3) & 1) << 4)) : -1;
    }
    const unsigned sbase = (unsigned)(pb * p.Cin + quad * 4) * 4u;
    struct Stage { float4 v[2], s; };
    Stage st;
    int k_ci = 0;
    auto gloadA = [&]() {
        const unsigned so = (unsigned)k_ci * 4u;
        st.v[0] = buffer_load4(rx, (goff[0] + so) | gmask[0], 0);
        if (wave < 2) st.v[1] = buffer_load4(rx, (goff[1] + so) | gmask[1], 0);   // tasks 256..339
        if (SCALE) st.s = buffer_load4(rs_, sbase, so);
        k_ci += BK;
    };
    auto lstoreA = [&](int buf) {
        unsigned char* base = smem + buf * BUFB;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (j == 1 && wave >= 2) break;
            float4 v = st.v[j];
            if (SCALE) v = make_float4(mul_rn(v.x, st.s.x), mul_rn(v.y, st.s.y), mul_rn(v.z, st.s.z), mul_rn(v.w, st.s.w));
            const Split4 s = split4(v);
            if (lds_st[j] >= 0) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint2*>(base + pl * PLB + lds_st[j]) = s.p[pl];
            }
        }
    };

    // ---- MFMA side ------------------------------------------------------------------------------------------------------------
    const int nq = wave;
    const int li = lane & 31, lh = lane >> 5;
    // operand a of shift (jy, jx): 32 positions = patch rows a*2 + (li >> 4), column li & 15; staged pixel (row + 1 - jy, col + 1 - jx)
    int a_off[4][2];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int aa = 0; aa < 2; ++aa) {
            const int pix = (aa * 2 + (li >> 4) + 1 - (s >> 1)) * LP + (li & 15) + 1 - (s & 1);
            a_off[s][aa] = pix_off(pix, lh);
        }
    const unsigned b_voff = (unsigned)((n0 + nq * 32 + li) * 32 + lh * 16);
    f32x16 acc[4][2];
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
        for (int aa = 0; aa < 2; ++aa)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ph][aa][e] = 0.f;

    struct BSet { bf16x8 f[2][3]; };           // the weights of the (up to) two taps of a group
    int chunk_b = 0;                           // chunk the NEXT gloadB call belongs to is passed explicitly
    // taps per phase: 4, 2, 2, 1; weight step of (chunk c, tap) = c * taps + tap
    auto loadB1 = [&](bf16x8 (&f)[3], int ph, int ntaps, int c, int tap) {
        const unsigned so = (unsigned)((c * ntaps + tap) * p.Cout) * 32u;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            f[pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw[ph], (int)b_voff, (int)(so + (unsigned)pl * a.plane_bytes[ph]), 0));
    };
    // groups of a chunk (shift = jy*2 + jx):   0: shift 0, (ph0 tap0), (ph1 tap0)     1: shift 0, (ph2 tap0), (ph3 tap0)
    //                                          2: shift 1, (ph0 tap1), (ph2 tap1)     3: shift 2, (ph0 tap2), (ph1 tap1)
    //                                          4: shift 3, (ph0 tap3)
    auto loadB = [&](BSet& b, int g, int c) {
        if (g == 0) { loadB1(b.f[0], 0, 4, c, 0); loadB1(b.f[1], 1, 2, c, 0); }
        else if (g == 1) { loadB1(b.f[0], 2, 2, c, 0); loadB1(b.f[1], 3, 1, c, 0); }
        else if (g == 2) { loadB1(b.f[0], 0, 4, c, 1); loadB1(b.f[1], 2, 2, c, 1); }
        else if (g == 3) { loadB1(b.f[0], 0, 4, c, 2); loadB1(b.f[1], 1, 2, c, 1); }
        else { loadB1(b.f[0], 0, 4, c, 3); }
    };
    // A operand halves (a = 0 / 1: two position rows each) are separate register sets: while one half multiplies, the other is
    // re-read for the next shift (left to itself hipcc serialised read -> wait -> MFMA: the LDS latency sat in front of every
    // third MFMA)
    auto afrag = [&](const unsigned char* base, int s, int aa, bf16x8 (&fa)[3]) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) fa[pl] = *reinterpret_cast<const bf16x8*>(base + pl * PLB + a_off[s][aa]);
    };
    // one operand half x the two taps of a weight set: two independent accumulators, 12 MFMAs (ph1 < 0: one tap)
    auto unit = [&](const bf16x8 (&fa)[3], const BSet& b, int aa, int ph0, int ph1) {
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            acc[ph0][aa] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[q]], b.f[0][PB[q]], acc[ph0][aa], 0, 0, 0);
            if (ph1 >= 0) acc[ph1][aa] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[q]], b.f[1][PB[q]], acc[ph1][aa], 0, 0, 0);
        }
    };
    (void)chunk_b;
#define SB __builtin_amdgcn_sched_barrier(0)

    // chunk c: LDS[c & 1] holds its image, X the weights of group 0; the window of chunk c + 1 is in flight in `st` (split into
    // LDS[(c + 1) & 1] under the MFMAs of the last groups); the weights of every group are fetched one group ahead.  X and Y swap
    // roles every chunk (five groups), so the loop body is two chunks.
    auto step = [&](int c, BSet& X, BSet& Y) {
        const unsigned char* base = smem + (c & 1) * BUFB;
        bf16x8 f0[3], f1[3];
        loadB(Y, 1, c);
        gloadA();
        afrag(base, 0, 0, f0);
        afrag(base, 0, 1, f1);
        SB;
        unit(f0, X, 0, 0, 1);           // group 0: shift 0, phases 0 / 1
        unit(f1, X, 1, 0, 1);
        SB;
        loadB(X, 2, c);
        SB;
        unit(f0, Y, 0, 2, 3);           // group 1: shift 0, phases 2 / 3
        SB;
        afrag(base, 1, 0, f0);
        SB;
        unit(f1, Y, 1, 2, 3);
        SB;
        loadB(Y, 3, c);
        afrag(base, 1, 1, f1);
        SB;
        unit(f0, X, 0, 0, 2);           // group 2: shift 1 (jx = 1), phases 0 / 2
        SB;
        afrag(base, 2, 0, f0);
        SB;
        unit(f1, X, 1, 0, 2);
        SB;
        loadB(X, 4, c);
        afrag(base, 2, 1, f1);
        SB;
        unit(f0, Y, 0, 0, 1);           // group 3: shift 2 (jy = 1), phases 0 / 1
        SB;
        afrag(base, 3, 0, f0);
        SB;
        lstoreA((c & 1) ^ 1);           // the split of chunk c + 1 rides under these MFMAs
        unit(f1, Y, 1, 0, 1);
        SB;
        loadB(Y, 0, c + 1);
        afrag(base, 3, 1, f1);
        SB;
        unit(f0, X, 0, 0, -1);          // group 4: shift 3, phase 0
        unit(f1, X, 1, 0, -1);
        __syncthreads();
    };
#undef SB
    const int nc = p.Cin / BK;
    // Where the time goes (tools/probes/tphase_scan.py, 512 resident blocks, B = 32, 128x128 -> 257x257, Cout = 128): 5.2 us per
    // 16-channel chunk (3.3 us of MFMA issue for the CU's two blocks) + 16 us per block for the epilogue: a block stores 4x the pixels
    // it read (128 KB), 64 MB per round of blocks = 4 TB/s, and nothing overlaps it -- without the stores the fixed cost is 2.7 us.
    // Tried and measured flat: a random start delay for the first round's blocks (de-phasing), nontemporal stores, and stores to one
    // contiguous 128 KB region per block (so it is not the 128-byte-per-KB pattern).  16-chunk layers run at 0.85 of the store-free rate.
    BSet b0, b1;
    gloadA();
    loadB(b0, 0, 0);
    lstoreA(0);
    __syncthreads();
    int c = 0;
    for (; c + 1 < nc; c += 2) {
        step(c, b0, b1);
        step(c + 1, b1, b0);
    }
    if (c < nc) step(c, b0, b1);

    // ---- epilogue: gain / demodulation, strided store of each phase.  Block-uniform 64-bit base per phase, 32-bit lane offsets
    // (the first version formed a 64-bit address per element: ~2000 vector instructions per wave behind the last MFMA) -------------
    const int n = n0 + nq * 32 + li;
    if (n >= p.Cout) return;
    const float os = out_scale ? out_scale[(int64_t)pb * p.Cout + n] : 1.f;
    const unsigned rowstep = (unsigned)(2 * p.YW) * (unsigned)p.Cout;           // one position row = two output rows
    const unsigned colstep = 2u * (unsigned)p.Cout;
    const unsigned lane_off = (unsigned)(4 * lh) * colstep + (unsigned)(nq * 32 + li);
    const bool full = qy0 + PH <= a.oh[3] && qx0 + PW <= a.ow[3];                 // block-uniform: nothing of the patch is masked
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
        float* yb = y + (((int64_t)pb * p.YH + 2 * qy0 + a.ooy[ph]) * p.YW + 2 * qx0 + a.oox[ph]) * p.Cout + n0;
#pragma unroll
        for (int aa = 0; aa < 2; ++aa)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ry = aa * 2 + (e >> 3), cx = (e & 3) + 8 * ((e >> 2) & 1);      // + 4 lh columns (lane_off)
                if (!full && (qy0 + ry >= a.oh[ph] || qx0 + cx + 4 * lh >= a.ow[ph])) continue;
                float v = mul_rn(acc[ph][aa][e], p.gain);
                if (out_scale) v = mul_rn(v, os);
                yb[lane_off + (unsigned)ry * rowstep + (unsigned)cx * colstep] = v;
            }
    }
}

}  // namespace

// The canonical geometry (see the header); `order[i]` = index of the launch with the i-th canonical tap shape.
static bool tphase_match(int n, const ideas_conv_params* ps, int (&order)[4]) {
    if (n != 4) return false;
    static const int ty[4] = {2, 2, 1, 1}, tx[4] = {2, 1, 2, 1};
    for (int i = 0; i < 4; ++i) {
        order[i] = -1;
        for (int k = 0; k < 4; ++k)
            if (ps[k].TY == ty[i] && ps[k].TX == tx[i]) order[i] = k;
        if (order[i] < 0) return false;
    }
    for (int k = 0; k < 4; ++k) {
        const ideas_conv_params& p = ps[k];
        if (p.sy != 1 || p.sx != 1 || p.dy != -1 || p.dx != -1 || p.offy != 0 || p.offx != 0 || p.osy != 2 || p.osx != 2) return false;
        if (p.ooy != (p.TY == 2 ? 0 : 1) || p.oox != (p.TX == 2 ? 0 : 1)) return false;
        if (p.reflect || p.act || p.accumulate || p.gain != ps[0].gain) return false;
        if (p.OH != (p.YH - p.ooy + 1) / 2 || p.OW != (p.YW - p.oox + 1) / 2) return false;
    }
    const ideas_conv_params& p = ps[0];
    if (p.Cin % 16 || p.Cout <= 64) return false;
    if (p.YH != 2 * p.IH + 1 || p.YW != 2 * p.IW + 1) return false;
    return (int64_t)p.B * p.IH * p.IW * p.Cin * 4 < 0xffffffffLL && (int64_t)4 * p.Cin * p.Cout * 6 < 0xffffffffLL;
}

// called by ideas_b3_fwd_multi; returns -1 when the launches are not the canonical transposed-conv phases (IDEAS_B3_TPHASE=0: never).
// The input grid of the phases has IH + 1 rows and IW + 1 columns, which no power-of-two patch divides (129 x 129 positions in 4 x 16
// patches: 12 % of the blocks' positions idle).  When IH % 4 == 0 and IW % 16 == 0 the kernel therefore covers the IH x IW positions
// exactly, and the last row and column of positions (output row 2 IH, output column 2 IW) come back as `nstrips` <= 4 small
// launches of the generic kernel for the caller to run (conv_b3_multi_kernel): 1/129 of the work each.
int ideas_b3_fwd_tphase(int n, void* y, const void* x, const void* const* wplanes, const float* in_scale, const float* out_scale,
                        const ideas_conv_params* ps, hipStream_t stream, ideas_conv_params* strips, const void** strip_w, int* nstrips) {
    *nstrips = 0;
    // Default: the modulated launches only (the generator's transposed convs).  The discriminators' stride-2 input gradients are 4 %
    // faster with this kernel in isolation, but inside the training step they run while the weight gradients occupy the side stream,
    // and there two 256-register blocks per CU co-schedule worse than the generic kernel's three 150-register ones: same-box step
    // 451.7 ms without this kernel, 454.8 with it everywhere, 449.0 with it on the modulated launches only.
    // IDEAS_B3_TPHASE = 0: never, 1: every launch of the geometry.
    const char* e = getenv("IDEAS_B3_TPHASE");
    if (e ? (e[0] == '0' || (e[0] != '1' && !in_scale)) : !in_scale) return -1;
    int order[4];
    if (!tphase_match(n, ps, order)) return -1;
    TPhase a;
    for (int i = 0; i < 4; ++i) {
        const ideas_conv_params& p = ps[order[i]];
        a.w[i] = wplanes[order[i]];
        a.plane_bytes[i] = (unsigned)((int64_t)p.TY * p.TX * p.Cin * p.Cout * 2);
        a.oh[i] = p.OH; a.ow[i] = p.OW; a.ooy[i] = p.ooy; a.oox[i] = p.oox;
    }
    const ideas_conv_params& p = ps[order[0]];
    const bool exact = p.IH % PH == 0 && p.IW % PW == 0;
    const int QH = exact ? p.IH : p.OH, QW = exact ? p.IW : p.OW;       // (the 2x2-tap phase covers every input-grid position)
    const int64_t tm = (int64_t)p.B * ((QH + PH - 1) / PH) * ((QW + PW - 1) / PW);
    const int tn = (p.Cout + 127) / 128;
    if (tm * tn > 0x7fffffffLL) return IDEAS_E_SHAPE;
    const unsigned x_bytes = (unsigned)((int64_t)p.B * p.IH * p.IW * p.Cin * 4);
    if (in_scale)
        hipLaunchKernelGGL(conv_b3_tphase_kernel<true>, dim3((unsigned)(tm * tn)), dim3(256), 0, stream, (float*)y, (const float*)x,
                           in_scale, out_scale, p, a, QH, QW, tn, x_bytes);
    else
        hipLaunchKernelGGL(conv_b3_tphase_kernel<false>, dim3((unsigned)(tm * tn)), dim3(256), 0, stream, (float*)y, (const float*)x,
                           in_scale, out_scale, p, a, QH, QW, tn, x_bytes);
    if (exact) {
        // position row IH: phases (2x2) and (2x1) [output row 2 IH, even / odd columns]; position column IW, rows < IH: phases (2x2), (1x2)
        const int rowph[2] = {0, 1}, colph[2] = {0, 2};
        for (int k = 0; k < 2; ++k) {
            ideas_conv_params s = ps[order[rowph[k]]];
            s.OH = 1; s.offy = p.IH; s.ooy = 2 * p.IH;
            strips[*nstrips] = s; strip_w[*nstrips] = wplanes[order[rowph[k]]]; ++*nstrips;
        }
        for (int k = 0; k < 2; ++k) {
            ideas_conv_params s = ps[order[colph[k]]];
            s.OW = 1; s.OH = p.IH; s.offx = p.IW; s.oox = 2 * p.IW;
            strips[*nstrips] = s; strip_w[*nstrips] = wplanes[order[colph[k]]]; ++*nstrips;
        }
    }
    return ideas_launch_status();
}
