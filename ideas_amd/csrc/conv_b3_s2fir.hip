// Blur -> 3x3 / stride-2 convolution of a downsampling ConvLayer (models.py:68-76: Blur(pad (2,2)) then EqualConv2d stride 2) as ONE
// kernel on the bf16 matrix pipe with the exact 3-way split of b3.hpp: the blurred tensor is never written to HBM and read back.
//
//     xb[i, j]      = sum_{a,b in 0..3} kv[a] kh[b] x[i + a - pad0, j + b - pad0]        (the 4x4 FIR, separable: make_kernel's outer product)
//     y[oy, ox, o]  = epilogue( sum_{ty,tx,ci} w[o,ty,tx,ci] xb[2 oy + ty, 2 ox + tx, ci] )
//
// The generic kernel (conv_b3.hip) behind a blur pass costs: one HBM-bound pass over the largest tensors of a discriminator (8.6 % of the
// f32 step's kernel time, half of it forward), and per conv block a load + split of every blurred pixel for each of its nine taps.
// Here a block owns a patch of 8 x 16 output pixels and, per 16-channel chunk, builds the 17 x 33 blurred pixels under them ONCE:
//   * four PRODUCER waves (one per SIMD) run the FIR: lane (channel quad, raw column) loads one 16-byte piece per raw row and walks
//     down the 20 raw rows of the patch; the three other horizontal taps come from the lanes to the right by DPP row shifts, the
//     vertical taps from the last four filtered rows in registers -- the arithmetic of blur4_f32_c2 (upfirdn2d.hip), same taps,
//     same FMA order, so the blurred values (and with them the kernel's result) are bitwise those of the two-kernel chain;
//   * the blurred values are split into the three bf16 planes and stored as a pixel-major LDS image (= the MFMA A layout), columns
//     de-interleaved by parity (even columns at slots 0..16, odd at 17..32 of a 40-slot image row) so that the stride-2 tap (ty, tx)
//     of 16 consecutive output pixels is 16 consecutive slots: a tap is a row address, as in conv_b3_tphase.hip;
//   * four or eight CONSUMER waves (all 128 pixels x 32 output channels each for the 128 / 256-channel tiles) multiply the nine taps
//     from the image of chunk c while the producers build chunk c + 1 in the other LDS buffer (2 x 64 KB, one block per CU, one
//     barrier per chunk); the weights ([3][chunk * 9 + tap][Cout][16], the planes of ideas_b3_split_weights) go straight from global
//     memory into operand registers two taps ahead.
// LDS slot s of the image holds 16 bf16 (32 B) per plane, 16-byte halves swapped when bit 3 of s is set; with the 40-slot pitch the
// pixel rows of an operand (80 slots apart) alias mod 16 slots, so every ds_read_b128 lane group covers 16 distinct (slot mod 8, half)
// pairs: conflict-free at all nine taps (enumerated with tools/lds_conflicts_s2fir.py against the lane groups of MI355X_MICROARCH.md).
// `xb_out` (optional): the producers of channel tile 0 also store the blurred f32 values -- the operand of the layer's weight gradient
// (conv_b3_wgrad.hip reads it as before).  Measurements, the ablations behind this shape and what the fusion can and cannot buy:
// DESIGN.md section 3.9.
#include "b3.hpp"
#ifndef S2FIR_DPP_BUILTIN
#define S2FIR_DPP_BUILTIN 0
#endif
#include <cstdlib>

#ifndef S2FIR_XB_AUX
// Cache policy of the side-output stores (A/B builds: -DS2FIR_XB_AUX=n; 0 = default, 1 = sc0, 2 = nt, 3 = sc0 nt).  A store
// instruction covers 64 contiguous bytes of a pixel (16 channels), the other chunks of the pixel's line follow thousands of cycles
// later: with the default policy the L2 fetches every partially written line (profiles/r04_pmc_b3s2: 5.29 GB fetched with the side
// output against 3.71 without).  Non-temporal: Dreal.1.conv2 with the side output 3.45 -> 3.35-3.38 ms, the others 1-3 % (same box).
#define S2FIR_XB_AUX 2
#endif

namespace {

constexpr int TR = 8, TP = 16;                 // output patch
constexpr int IR = 2 * TR + 1;                 // image rows per chunk: 17
constexpr int PITCH = 40, ODD0 = 24;           // slots per image row; first slot of the odd columns (17 even + 16 odd slots; ODD0 % 8 == 0
                                               // with ODD0 / 8 odd: the odd run's 16-byte halves sit opposite the even run's, which keeps a
                                               // producer's ds_write_b64 -- 16 lanes = 16 columns of one channel quad -- at 2-way instead of 4-way)
constexpr int PLB = IR * PITCH * ROWB;         // bytes per plane: 21760
constexpr int BUFB = 3 * PLB;                  // 65280
constexpr int BW = 9;                          // image columns per producer wave (4 waves x 9 >= 33); a wave loads 16 raw columns
constexpr int RAWR = IR + 3;                   // raw rows under the 17 image rows: 20

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct S2Fir {
    float kh[4], kv[4];        // the flipped, gain-scaled FIR as blur4_f32_c2 factors it: kh[t] = f[0][t], kv[j] = f[j][0] / f[0][0]
    int XH, XW, pad0;          // raw input [B, XH, XW, Cin]; p.IH x p.IW is the blurred size
};

__device__ __forceinline__ int slot_byte(int slot, int half) { return slot * ROWB + ((half ^ ((slot >> 3) & 1)) << 4); }

constexpr int NPW = 4;                         // producer (FIR) waves: one per SIMD
#ifndef S2FIR_ABL
#define S2FIR_ABL 0                            // ablations for timing only (wrong results): 1 no producer work, 2 no consumer work,
#endif                                         // 3 producer without global loads, 4 consumer without weight loads

// NCW: consumer (MFMA) waves, 4 or 8 (one or two per SIMD); WN of them along the channels (NCW / WN along the pixels), NB: 32-channel
// blocks per consumer wave; N tile = WN * NB * 32.
// MODE 0: the producers blur (the kernel's purpose).  MODE 1 (round 5): NO FIR -- the producers only load, scale and split the 17 x 33
// input pixels under the patch, i.e. a plain 3x3 / stride-2 convolution on the LDS-image layout (x IS the tensor p describes; f.XH = p.IH,
// f.XW = p.IW, f.pad0 = 0), with the per-sample scales of a modulated conv (in_scale before the split, out_scale in the epilogue, the
// order of conv_b3_kernel).  The generic kernel stages every input pixel once per TAP (nine times: 195-203 TFLOP/s on the input
// gradients of G's upsampling layers); here it is staged once per chunk for all nine, and the consumer side is the one that runs
// 297-304 TFLOP/s when nothing else is in its way.
template <int NCW, int WN, int NB, bool XBOUT, int MODE = 0>
__global__ __launch_bounds__((NCW + NPW) * 64, 1) void conv_b3_s2fir_kernel(float* __restrict__ y, float* __restrict__ xb_out,
                                                               const float* __restrict__ x, const void* __restrict__ wplanes,
                                                               const float* __restrict__ bias, const float* __restrict__ resid,
                                                               ideas_conv_params p, S2Fir f, int tiles_n, unsigned x_bytes,
                                                               unsigned plane_bytes, unsigned xb_bytes,
                                                               const float* __restrict__ in_scale, const float* __restrict__ out_scale) {
    constexpr int WM = NCW / WN, RB = 4 / WM;   // row blocks (32 pixels = 2 patch rows) per consumer wave
    constexpr int BN = WN * NB * 32;
    static_assert(RB == 1 || RB == 2 || RB == 4, "wave grid");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUFB];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int ppr = (p.OW + TP - 1) / TP, ppi = ((p.OH + TR - 1) / TR) * ppr;
    const int swz = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tile_n = swz % tiles_n, tile_m = swz / tiles_n;
    const int pb = tile_m / ppi, prem = tile_m - pb * ppi;
    const int oy0 = (prem / ppr) * TR, ox0 = (prem % ppr) * TP;
    const int n0 = tile_n * BN;
    const int nc = p.Cin / BK;
#define SB __builtin_amdgcn_sched_barrier(0)

    if (wave >= NCW) {
        // ============================ producer waves: FIR + split -> LDS image of chunk c (+ side output) ============================
        // A wave owns BW = 9 image columns; lane (quad = lane >> 4, px = lane & 15) loads channels 4 quad .. + 3 of raw column
        // band + px -- ONE 16-byte load per raw row and thread -- and walks down the 20 raw rows of the patch.  The horizontal taps
        // come from the lanes to the right through DPP row shifts (px + 1 .. + 3 live in the same 16-lane row: no LDS, no extra
        // loads; lanes 13-15 only feed their neighbours), the vertical ones from the last four filtered rows in registers.  Per
        // chunk: 20 loads, 17 image pixels per thread; loads run RING - 1 rows (across chunk boundaries) ahead of their use.
        // (A first version gave a thread a column pair and a row segment: five loads per row and two outputs, 45 loads per chunk
        //  with 3.2x redundancy; its loads delayed the consumers' weight loads in the CU's one memory pipe by more than everything
        //  else it did -- ablations in DESIGN.md.)
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)x_bytes, (int)RSRC_FLAGS);
        const __amdgpu_buffer_rsrc_t rxb = __builtin_amdgcn_make_buffer_rsrc((void*)xb_out, 0, XBOUT ? (int)xb_bytes : 0, (int)RSRC_FLAGS);
        const int pw = wave - NCW;
        const int quad = lane >> 4, px = lane & 15;
        const int j = pw * BW + px;                                   // image column of this lane's output
        const bool on = px < BW && j < 2 * TP + 1;
        const int iy0 = 2 * oy0 - f.pad0, ix = 2 * ox0 - f.pad0 + j;  // raw pixel under (image row 0, column j), tap (0, 0)
        const unsigned cmask = (ix >= 0 && ix < f.XW) ? 0u : 0xffffffffu;      // raw column in the zero padding
        unsigned rbad = 0;                            // bit r: raw row iy0 + r lies in the zero padding; bit 24: side-output column not owned
        // rows staged per chunk: the 20 raw rows under the 17 blurred ones, or (MODE 1) the 17 image rows + one dummy (18 = 2 x 9: the
        // ring position of a row must not depend on the chunk)
        constexpr int NROW = MODE == 0 ? RAWR : 18;
#pragma unroll
        for (int r = 0; r < NROW; ++r) rbad |= ((iy0 + r >= 0 && iy0 + r < f.XH && (MODE == 0 || r < IR)) ? 0u : 1u) << r;
        const unsigned colstep = (unsigned)p.Cin * 4u, rowstep = (unsigned)f.XW * colstep;
        // (mod 2^32: rows / columns left of the image give a wrapped offset, which its mask replaces)
        unsigned gbase = (unsigned)(((pb * f.XH + iy0) * f.XW + ix) * p.Cin + quad * 4) * 4u;
        // LDS byte address (plane 0) of column j at image row 0; one image row further = + PITCH * ROWB with the 16-byte halves
        // swapped back (PITCH / 8 is odd: bit 3 of the slot flips)
        const int slot0 = (j & 1) ? ODD0 + (j >> 1) : (j >> 1);
        int w0 = slot0 * ROWB + ((quad * 8) ^ (((slot0 >> 3) & 1) << 4));
        // side output: blurred pixel (2 oy0 + i, 2 ox0 + j) belongs to this patch unless it is the shared last row / column of an inner one
        unsigned xb_off = 0;
        if (XBOUT) {
            const int jg = 2 * ox0 + j;
            const bool last_c = ox0 + TP >= p.OW;
            const bool ok = on && tile_n == 0 && jg < p.IW && (j < 2 * TP || last_c);
            xb_off = (unsigned)(((pb * p.IH + 2 * oy0) * p.IW + jg) * p.Cin + quad * 4) * 4u;
            rbad |= (ok ? 0u : 1u) << 24;
        }
        const bool last_r = oy0 + TR >= p.OH;

        constexpr int RING = MODE == 0 ? 10 : 9, AHEAD = RING - 1;       // 20 (18) rows per chunk: the ring position of a row is the same in every chunk
        static_assert(NROW % RING == 0 && (MODE != 0 || RAWR % 4 == 0), "ring positions must not depend on the chunk");
        float4 ldv[RING];
        float4 h[4];                                     // horizontally filtered rows r - 3 .. r
        auto gload = [&](int r, int ci) {                // raw row r of the patch, chunk starting at channel ci
            const unsigned so = (unsigned)r * rowstep + (unsigned)ci * 4u;       // uniform
            ldv[r % RING] = buffer_load4(rx, (gbase + so) | cmask | (unsigned)__builtin_amdgcn_sbfe(rbad, r, 1), 0);
        };
        // The three shifted taps are v_fmac_f32 with a DPP source (row_shl:n = the lane n columns to the right, 0 past the 16-lane
        // row).  Written in assembly: hipcc keeps __builtin_amdgcn_update_dpp as a separate v_mov_b32_dpp per shift (240 per chunk;
        // its FMAs are the VOP3 / packed forms, which take no DPP operand).  v_fmac needs the tap in a VGPR.  The hazard recogniser
        // does not look inside asm, so the wait states are ours: a VALU write of the DPP source wants 2, a VALU write of EXEC 5 --
        // s_nop 4 (five states) covers whatever the compiler schedules in front (s_nop 1 was enough for what it
        // schedules today).  All 64 lanes are active here (no divergent code around the producer loop).
        // -DS2FIR_DPP_BUILTIN=1 builds the same arithmetic from __builtin_amdgcn_update_dpp + fmaf (the compiler then owns the
        // hazards): libideas_hip_dppb.so of the Makefile, which tests/test_ops_gpu.py runs through the same bitwise checks.
        float kh1 = f.kh[1], kh2 = f.kh[2], kh3 = f.kh[3];
        asm volatile("" : "+v"(kh1), "+v"(kh2), "+v"(kh3));
        auto hrow1 = [&](float v) {                      // blur4_f32_c2::hrow: fma chain over the four columns, from zero
            float a = fmaf(v, f.kh[0], 0.f);
#if S2FIR_DPP_BUILTIN
            const int vi = __builtin_bit_cast(int, v);
            a = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, vi, 0x101, 0xf, 0xf, true)), kh1, a);
            a = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, vi, 0x102, 0xf, 0xf, true)), kh2, a);
            a = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, vi, 0x103, 0xf, 0xf, true)), kh3, a);
#else
            asm volatile("s_nop 4\n\t"
                         "v_fmac_f32_dpp %0, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_fmac_f32_dpp %0, %1, %3 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_fmac_f32_dpp %0, %1, %4 row_shl:3 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                         : "+v"(a) : "v"(v), "v"(kh1), "v"(kh2), "v"(kh3));
#endif
            return a;
        };
        auto hrow = [&](int r) {
            const float4 v = ldv[r % RING];
            h[r % 4] = make_float4(hrow1(v.x), hrow1(v.y), hrow1(v.z), hrow1(v.w));
        };
        auto vsum1 = [&](float r0, float r1, float r2, float r3) {
            // blur4_f32_c2 writes r0 k0 + r1 k1 + r2 k2 + r3 k3, which hipcc contracts as fma(r0, k0, r1 k1) then the other two: same order here
            float s_ = r1 * f.kv[1];
            s_ = fmaf(r0, f.kv[0], s_);
            s_ = fmaf(r2, f.kv[2], s_);
            return fmaf(r3, f.kv[3], s_);
        };
        // image row i (raw rows i .. i + 3 filtered): vertical sum, split, LDS store (+ side output)
        auto emit = [&](unsigned char* buf, int i, int ci_chunk) {
            const float4 &r0 = h[i % 4], &r1 = h[(i + 1) % 4], &r2 = h[(i + 2) % 4], &r3 = h[(i + 3) % 4];
            const float4 o = make_float4(vsum1(r0.x, r1.x, r2.x, r3.x), vsum1(r0.y, r1.y, r2.y, r3.y), vsum1(r0.z, r1.z, r2.z, r3.z),
                                         vsum1(r0.w, r1.w, r2.w, r3.w));
            const Split4 s_ = split4(o);
            if (on) {                                    // (only the stores are conditional: no value is defined on one path only)
                unsigned char* a = buf + ((i & 1) ? (w0 ^ 16) : w0) + i * PITCH * ROWB;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint2*>(a + pl * PLB) = s_.p[pl];
            }
            if (XBOUT) {
                const unsigned rowbad = (2 * oy0 + i < p.IH && (i < 2 * TR || last_r)) ? 0u : 0xffffffffu;     // uniform
                const unsigned so = (unsigned)i * (unsigned)p.IW * colstep + (unsigned)ci_chunk * 4u;             // uniform
                const f32x4 v = {o.x, o.y, o.z, o.w};
                // (the uniform part is ADDED to the lane offset, not passed as the instruction's scalar offset: with an SGPR offset hipcc
                //  follows the ISA manual's "no wait state needed" and overwrote the data registers in the very next instruction -- on
                //  this part the second component of lanes 12-15 (mod 16) then stored the NEW value: tools/probes/dbg_xb.py)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rxb,
                                                       (int)((xb_off + so) | (unsigned)__builtin_amdgcn_sbfe(rbad, 24, 1) | rowbad), 0, S2FIR_XB_AUX);   // masked: dropped
            }
        };

        // MODE 1: image row i = input row i: scale (rounded to f32 before the split, as conv_b3_kernel does), split, store
        const __amdgpu_buffer_rsrc_t rsi = __builtin_amdgcn_make_buffer_rsrc((void*)in_scale, 0, (MODE == 1 && in_scale) ? p.B * p.Cin * 4 : 0, (int)RSRC_FLAGS);
        auto emit_plain = [&](unsigned char* buf, int i, float4 sc) {
            float4 v = ldv[i % RING];
            if (in_scale) v = make_float4(mul_rn(v.x, sc.x), mul_rn(v.y, sc.y), mul_rn(v.z, sc.z), mul_rn(v.w, sc.w));
            const Split4 s_ = split4(v);
            if (on) {
                unsigned char* a = buf + ((i & 1) ? (w0 ^ 16) : w0) + i * PITCH * ROWB;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint2*>(a + pl * PLB) = s_.p[pl];
            }
        };
#pragma unroll
        for (int r = 0; r < AHEAD; ++r) gload(r, 0);
        for (int c = 0; c < nc; ++c) {
            unsigned char* buf = smem + (c & 1) * BUFB;
            const int ci = c * BK;
            const int ci_next = c + 1 < nc ? ci + BK : ci;          // (behind the last chunk: harmless reloads of its first rows)
            // (opaque to the optimizer: nothing derived from these is hoisted out of the chunk loop)
            asm volatile("" : "+v"(gbase), "+v"(rbad), "+v"(w0));
            if (XBOUT) asm volatile("" : "+v"(xb_off));
            if constexpr (MODE == 0) {
#pragma unroll
                for (int r = 0; r < RAWR; ++r) {
                    if (S2FIR_ABL == 1) break;
                    if (S2FIR_ABL != 3) { if (r + AHEAD < RAWR) gload(r + AHEAD, ci); else gload(r + AHEAD - RAWR, ci_next); }
                    SB;
                    hrow(r);
                    if (r >= 3) emit(buf, r - 3, ci);
                    SB;
                }
            } else {
                float4 sc = make_float4(1.f, 1.f, 1.f, 1.f);
                if (in_scale) sc = buffer_load4(rsi, (unsigned)(pb * p.Cin + ci + quad * 4) * 4u, 0);
#pragma unroll
                for (int r = 0; r < NROW; ++r) {
                    if (r + AHEAD < NROW) gload(r + AHEAD, ci); else gload(r + AHEAD - NROW, ci_next);
                    SB;
                    if (r < IR) emit_plain(buf, r, sc);
                    SB;
                }
            }
            __syncthreads();                             // image of chunk c complete; the consumers are done with chunk c - 1
        }
        return;
    }

    // ================================ consumer waves: nine stride-2 taps per chunk from the LDS image ================================
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)wplanes, 0, (int)(3u * plane_bytes), (int)RSRC_FLAGS);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    // operand of tap (ty, tx), row block rb: image row 4 (wm RB + rb) + 2 (li >> 4) + ty, slot (li & 15) + {0, ODD0, 1}[tx].  Two lane
    // addresses (tap column 0 and 2; the odd columns sit ODD0 = 3 x 8 slots behind the former: bit 3 flipped, halves swapped) plus
    // immediates: + PITCH * ROWB per ty (halves swapped back for ty = 1), + ODD0 * ROWB for tx = 1, + 4 * PITCH * ROWB per row block
    // (bit 3 unchanged).
    const int a_row0 = (4 * wm * RB + 2 * (li >> 4)) * PITCH + (li & 15);
    int a_c0 = slot_byte(a_row0, lh), a_c2 = slot_byte(a_row0 + 1, lh);
    auto a_off = [&](int tap, int rb) {
        const int ty = tap / 3, tx = tap - 3 * ty;
        static_assert(ODD0 % 8 == 0 && (ODD0 / 8) % 2 == 1, "tap column 1 = tap column 0's address with the halves swapped");
        const int base = tx == 2 ? a_c2 : a_c0;
        return ((((ty & 1) != 0) != (tx == 1)) ? (base ^ 16) : base) + ty * PITCH * ROWB + (tx == 1 ? ODD0 * ROWB : 0) + rb * 4 * PITCH * ROWB;
    };
    const unsigned b_voff = (unsigned)((n0 + wn * NB * 32 + li) * 32 + lh * 16);
    f32x16 acc[RB][NB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[rb][nb][e] = 0.f;

    struct BSet { bf16x8 f[NB][3]; };
    auto loadB = [&](BSet& b, int c, int tap) {
        const unsigned so = (unsigned)((c * 9 + tap) * p.Cout) * 32u;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                b.f[nb][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, (int)(b_voff + (unsigned)nb * 1024u),
                                                                                                 (int)(so + (unsigned)pl * plane_bytes), 0));
    };
    // The operand of a tap is read in two halves of HB row blocks (RB == 1: one "half" per tap, two sets alternating by tap): while
    // one half multiplies, the other is (re-)read -- left alone, hipcc serialises read -> wait -> MFMA.
    constexpr int HB = RB >= 2 ? RB / 2 : 1;
    bf16x8 fa[2][HB][3];
    auto readA = [&](const unsigned char* base, int tap, int half, int set) {
#pragma unroll
        for (int k = 0; k < HB; ++k)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                fa[set][k][pl] = *reinterpret_cast<const bf16x8*>(base + pl * PLB + a_off(tap, (RB >= 2 ? half * HB : 0) + k));
    };
    auto mma = [&](int set, int half, const BSet& b) {
#pragma unroll
        for (int qq = 0; qq < 6; ++qq)
#pragma unroll
            for (int k = 0; k < HB; ++k)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    f32x16& d = acc[(RB >= 2 ? half * HB : 0) + k][nb];
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[set][k][PA[qq]], b.f[nb][PB[qq]], d, 0, 0, 0);
                }
    };
    // chunk c: wait for its image (barrier c), then nine taps; the weights are fetched TWO taps ahead (three register sets, 9 % 3 == 0:
    // tap t always uses set t % 3) -- the producers' loads share the CU's memory pipe and one tap of MFMAs did not cover the queue.
    BSet bs[3];
    auto step = [&](int c) {
        const unsigned char* base = smem + (c & 1) * BUFB;
        asm volatile("" : "+v"(a_c0), "+v"(a_c2));
        __syncthreads();
        readA(base, 0, 0, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (S2FIR_ABL == 2) break;
            if (S2FIR_ABL != 4) { if (tap < 7) loadB(bs[(tap + 2) % 3], c, tap + 2); else loadB(bs[(tap + 2) % 3], c + 1, tap - 7); }
            const BSet& cur = bs[tap % 3];
            if (RB >= 2) {
                readA(base, tap, 1, 1);
                SB;
                mma(0, 0, cur);
                SB;
                if (tap < 8) readA(base, tap + 1, 0, 0);
                SB;
                mma(1, 1, cur);
                SB;
            } else {
                if (tap < 8) readA(base, tap + 1, 0, (tap + 1) & 1);
                SB;
                mma(tap & 1, 0, cur);
                SB;
            }
        }
    };
    loadB(bs[0], 0, 0);
    loadB(bs[1], 0, 1);
    for (int c = 0; c < nc; ++c) step(c);
#undef SB

    // ---- epilogue (the arithmetic of conv_b3_kernel) ----------------------------------------------------------------------------------
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int n = n0 + (wn * NB + nb) * 32 + li;
        if (n >= p.Cout) continue;
        const float bv = bias ? bias[n] : 0.f;
        const float osv = (MODE == 1 && out_scale) ? out_scale[(int64_t)pb * p.Cout + n] : 1.f;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = (e & 3) + 8 * (e >> 2) + 4 * lh;
                const int oy = oy0 + 2 * (wm * RB + rb) + (m >> 4), ox = ox0 + (m & 15);
                if (oy >= p.OH || ox >= p.OW) continue;
                const int64_t off = (((int64_t)pb * p.YH + oy) * p.YW + ox) * p.Cout + n;
                float v = mul_rn(acc[rb][nb][e], p.gain);
                if (MODE == 1 && out_scale) v = mul_rn(v, osv);
                v = mul_then_add(v, 1.0f, bv);
                if (p.act) v = (v > 0.f ? v : v * p.alpha) * p.act_gain;
                if (resid) v = (v + resid[off]) * p.resid_gain;
                y[off] = v;
            }
    }
}

template <int NCW, int WN, int NB>
int launch_s2fir(void* y, void* xb, const void* x, const void* wplanes, const float* bias, const void* resid, const ideas_conv_params* p,
                 const S2Fir& f, hipStream_t stream) {
    constexpr int BN = WN * NB * 32;
    const int64_t tm = (int64_t)p->B * ideas_cdiv(p->OH, TR) * ideas_cdiv(p->OW, TP);
    const int tn = (int)ideas_cdiv(p->Cout, BN);
    if (tm * tn > 0x7fffffffLL) return IDEAS_E_SHAPE;
    const unsigned x_bytes = (unsigned)((int64_t)p->B * f.XH * f.XW * p->Cin * 4);
    const unsigned xb_bytes = (unsigned)((int64_t)p->B * p->IH * p->IW * p->Cin * 4);
    const unsigned plane_bytes = (unsigned)((int64_t)9 * p->Cin * p->Cout * 2);
    if (xb)
        hipLaunchKernelGGL((conv_b3_s2fir_kernel<NCW, WN, NB, true>), dim3((unsigned)(tm * tn)), dim3((NCW + NPW) * 64), 0, stream, (float*)y, (float*)xb,
                           (const float*)x, wplanes, bias, (const float*)resid, *p, f, tn, x_bytes, plane_bytes, xb_bytes, (const float*)nullptr,
                           (const float*)nullptr);
    else
        hipLaunchKernelGGL((conv_b3_s2fir_kernel<NCW, WN, NB, false>), dim3((unsigned)(tm * tn)), dim3((NCW + NPW) * 64), 0, stream, (float*)y, (float*)xb,
                           (const float*)x, wplanes, bias, (const float*)resid, *p, f, tn, x_bytes, plane_bytes, xb_bytes, (const float*)nullptr,
                           (const float*)nullptr);
    return ideas_launch_status();
}

template <int NCW, int WN, int NB>
int launch_s2img(void* y, const void* x, const void* wplanes, const float* in_scale, const float* out_scale, const float* bias, const void* resid,
                 const ideas_conv_params* p, hipStream_t stream) {
    constexpr int BN = WN * NB * 32;
    const int64_t tm = (int64_t)p->B * ideas_cdiv(p->OH, TR) * ideas_cdiv(p->OW, TP);
    const int tn = (int)ideas_cdiv(p->Cout, BN);
    if (tm * tn > 0x7fffffffLL) return IDEAS_E_SHAPE;
    S2Fir f;
    for (int i = 0; i < 4; ++i) f.kh[i] = f.kv[i] = 0.f;
    f.XH = p->IH; f.XW = p->IW; f.pad0 = 0;
    const unsigned x_bytes = (unsigned)((int64_t)p->B * p->IH * p->IW * p->Cin * 4);
    const unsigned plane_bytes = (unsigned)((int64_t)9 * p->Cin * p->Cout * 2);
    hipLaunchKernelGGL((conv_b3_s2fir_kernel<NCW, WN, NB, false, 1>), dim3((unsigned)(tm * tn)), dim3((NCW + NPW) * 64), 0, stream, (float*)y, (float*)nullptr,
                       (const float*)x, wplanes, bias, (const float*)resid, *p, f, tn, x_bytes, plane_bytes, 0u, in_scale, out_scale);
    return ideas_launch_status();
}

}  // namespace

// p describes the stride-2 conv on the BLURRED tensor [B, IH, IW, Cin] (TY = TX = 3, sy = sx = 2, no padding, dense output);
// the raw input is [B, xh, xw, Cin] with IH = xh + pad0 + pad1 - 3.
extern "C" int ideas_b3_blur_conv_s2_supported(const ideas_conv_params* p, int xh, int xw, int pad0) {
    if (!p) return 0;
    if (p->TY != 3 || p->TX != 3 || p->sy != 2 || p->sx != 2 || p->dy != 1 || p->dx != 1 || p->offy != 0 || p->offx != 0) return 0;
    if (p->osy != 1 || p->osx != 1 || p->ooy != 0 || p->oox != 0 || p->YH != p->OH || p->YW != p->OW || p->reflect || p->accumulate) return 0;
    if (p->OH != (p->IH - 3) / 2 + 1 || p->OW != (p->IW - 3) / 2 + 1 || p->IH < 3 || p->IW < 3) return 0;
    if (pad0 < 0 || pad0 > 3 || xh <= 0 || xw <= 0 || p->IH > xh + pad0 || p->IW > xw + pad0) return 0;   // pad1 <= 3
    if (p->IH < xh + pad0 - 3 || p->IW < xw + pad0 - 3) return 0;                                          // pad1 >= 0
    if (p->Cin % 16 || p->Cout % 4) return 0;
    return (int64_t)p->B * xh * xw * p->Cin * 4 < 0xffffffffLL && (int64_t)p->B * p->IH * p->IW * p->Cin * 4 < 0xffffffffLL &&
           (int64_t)9 * p->Cin * p->Cout * 6 < 0xffffffffLL;
}

extern "C" int ideas_b3_blur_conv_s2(void* y, void* xb_out, const void* x, const void* wplanes, const float* fir_h, const float* fir_v,
                                     const float* bias, const void* resid, const ideas_conv_params* p, int xh, int xw, int pad0,
                                     void* stream_) {
    if (!y || !x || !wplanes || !fir_h || !fir_v || !p) return IDEAS_E_NULL;
    if (!ideas_b3_blur_conv_s2_supported(p, xh, xw, pad0)) return IDEAS_E_UNSUPPORTED;
    if (!ideas_aligned16(x) || !ideas_aligned16(wplanes) || (xb_out && !ideas_aligned16(xb_out))) return IDEAS_E_ALIGN;
    // the side output is complete only when every blurred pixel lies under an output pixel's taps (the 2 OH + 1 rows of an even input)
    if (xb_out && (p->IH != 2 * p->OH + 1 || p->IW != 2 * p->OW + 1)) return IDEAS_E_UNSUPPORTED;
    S2Fir f;
    for (int i = 0; i < 4; ++i) { f.kh[i] = fir_h[i]; f.kv[i] = fir_v[i]; }
    f.XH = xh; f.XW = xw; f.pad0 = pad0;
    hipStream_t stream = (hipStream_t)stream_;
    // IDEAS_S2FIR_CFG (A/B measurements): 0 = default, 1 = eight consumer waves everywhere, 2 = four everywhere
    const char* ecfg = getenv("IDEAS_S2FIR_CFG");       // (read per call: no state in the library, include/ideas_hip.h)
    const int cfg = ecfg ? atoi(ecfg) : 0;
    const bool eight = cfg == 1 || (cfg == 0 && p->Cout > 128);
    // a consumer wave = all 128 pixels x 32 channels (every weight fragment is loaded by exactly one wave)
    if (p->Cout > 128 && eight) return launch_s2fir<8, 8, 1>(y, xb_out, x, wplanes, bias, resid, p, f, stream);   // N tile 256
    if (p->Cout > 64) return eight ? launch_s2fir<8, 4, 1>(y, xb_out, x, wplanes, bias, resid, p, f, stream)     // 128: 2 x 4 waves of 64 x 32
                                   : launch_s2fir<4, 4, 1>(y, xb_out, x, wplanes, bias, resid, p, f, stream);    //      1 x 4 waves of 128 x 32
    return eight ? launch_s2fir<8, 2, 1>(y, xb_out, x, wplanes, bias, resid, p, f, stream)                       // 64:  4 x 2 waves of 32 x 32
                 : launch_s2fir<4, 2, 1>(y, xb_out, x, wplanes, bias, resid, p, f, stream);                      //      2 x 2 waves of 64 x 32
}

// The same LDS-image kernel WITHOUT the FIR (MODE 1): a plain 3x3 / stride-2 / unpadded convolution, optionally modulated.  Taken by
// ideas_b3_fwd for grids of at least IDEAS_S2IMG_MIN_BLOCKS patches x N tiles (default 512 = two per CU; 0 in the environment: never).
int ideas_b3_s2img_ok(const ideas_conv_params* p, const float* in_scale) {
    const char* e = getenv("IDEAS_S2IMG_MIN_BLOCKS");
    const int64_t minb = e ? atoll(e) : 512;
    if (minb <= 0) return 0;
    if (p->TY != 3 || p->TX != 3 || p->sy != 2 || p->sx != 2 || p->dy != 1 || p->dx != 1 || p->offy != 0 || p->offx != 0) return 0;
    if (p->osy != 1 || p->osx != 1 || p->ooy != 0 || p->oox != 0 || p->YH != p->OH || p->YW != p->OW || p->reflect || p->accumulate) return 0;
    if (p->OH != (p->IH - 3) / 2 + 1 || p->OW != (p->IW - 3) / 2 + 1 || p->IH < 3 || p->IW < 3) return 0;
    if (p->Cin % 16 || p->Cout % 4 || p->OW < 16 || p->OH < 8) return 0;
    // Where it wins (tools/ab_s2img.py, same box, against conv_b3_kernel): the 256-channel N tile with eight consumer waves -- the
    // modulated input gradients of G's upsampling layers 190.9 -> 204.4, 205.3 -> 219.4, 195.8 -> 218.7 TFLOP/s; unmodulated 512 -> 512
    // 217 -> 227.  One 128- or 64-channel N tile amortises the staging of a patch over too few MFMAs (128 -> 128: 191 -> 166, 64 -> 64:
    // 121 -> 99) and 256 -> 256 is a tie: those keep the generic kernel.  (IDEAS_S2IMG_ALL=1: every shape the kernel covers -- tests.)
    const char* ea = getenv("IDEAS_S2IMG_ALL");
    if (!(ea && ea[0] == '1') && !(p->Cout > 128 && (in_scale || p->Cin >= 512))) return 0;
    const int nt = p->Cout > 128 ? 256 : p->Cout > 64 ? 128 : 64;
    if ((int64_t)p->B * ideas_cdiv(p->OH, TR) * ideas_cdiv(p->OW, TP) * ideas_cdiv(p->Cout, nt) < minb) return 0;
    return (int64_t)p->B * p->IH * p->IW * p->Cin * 4 < 0xffffffffLL && (int64_t)9 * p->Cin * p->Cout * 6 < 0xffffffffLL;
}

int ideas_b3_s2img_fwd(void* y, const void* x, const void* wplanes, const float* in_scale, const float* out_scale, const float* bias,
                       const void* resid, const ideas_conv_params* p, hipStream_t stream) {
    if (p->Cout > 128) return launch_s2img<8, 8, 1>(y, x, wplanes, in_scale, out_scale, bias, resid, p, stream);
    if (p->Cout > 64) return launch_s2img<4, 4, 1>(y, x, wplanes, in_scale, out_scale, bias, resid, p, stream);
    return launch_s2img<4, 2, 1>(y, x, wplanes, in_scale, out_scale, bias, resid, p, stream);
}
