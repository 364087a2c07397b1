// Weight gradient of the 3x3 convolutions on the bf16 matrix pipe (exact 3-way split of b3.hpp), tap-fused with a rolling
// activation window:
//
//     gw[o][ty][tx][ci] += gain * sum over (b, oy, ox) of  G(b, oy, ox, o) * X(b, oy*S + ty + offy, ox*S + tx + offx, ci)
//
// conv_b3_wgrad.hip treats the nine taps as nine independent column tiles of a GEMM: every block stages (loads, scales, splits
// into three bf16 planes, transposes in registers) a 128-channel x 16-pixel piece of G and of X per step for 96 MFMAs, so each
// G element is split by 9 Cin / 128 blocks and each X element by 9 Cout / 128 -- 5.8 vector instructions per MFMA, 2.4x the
// operand bytes fetched (profiles/r02_pmc_b3wg_*), 0.40 of the ceiling.  Here ONE block owns a 64 (o) x 64 (ci) tile for ALL nine
// taps and walks DOWN a 16-pixel-wide column strip of the images:
//   * per step (one output row of the strip) it stages one 16-pixel row of G and ONE new row of the X window (16 S + 2 pixels;
//     S rows for stride S) -- the other rows of the 3-row window are still in LDS from the previous steps (ring of rows);
//   * the nine taps are 9 x 6 = 54 MFMAs per wave on those operands: the tap (ty, tx) of X is the same LDS image read at row
//     slot ty and pixel offset tx.  That works because the image is kept PIXEL-major ([pixel][64 channels], exactly as the
//     float4 loads arrive: a staging thread only scales + splits 4 channels of one pixel and writes 8 bytes per plane) and the
//     MFMA operand (8 consecutive pixels of one channel per lane) is gathered by the LDS transpose read ds_read_b64_tr_b16,
//     whose per-lane row address is free -- a shifted window is just another row index.
// Per step a thread splits 8-9 values for 54 MFMAs of its wave (old kernel: 16 values for 24), G is fetched Cin/64 times and X
// Cout/64 times (+ the 2-pixel halo of a strip) instead of 9 Cin/128 and 9 Cout/128.
//
// LDS image of an operand: rows of 64 channels x bf16 = 128 B, one row per pixel, three planes; the 16-byte chunk c of row r
// sits at chunk (c ^ (((r >> 1) & 1) << 2)): the four rows r0..r0+3 a transpose block covers then fall into four different
// 8-bank groups for ANY r0 (4 r mod 8 alternates, the XOR separates rows two apart), so all three tap offsets read
// conflict-free.  Stride 2: window columns are de-interleaved by parity when they are written (row = parity * 17 + column / 2),
// so a tap still reads consecutive rows.
//
// Work decomposition: tiles = (Cout/64) x (Cin/64); a strip is 16 output columns; the B*OH output rows of a strip are cut into
// equal ranges (splits); grid = tiles x strips x splits in XCD-banded split-major order (common.hpp: the tiles of one range
// read the same pixels and meet in one L2).  A range may cross image boundaries: the window is re-primed (two bubble steps
// without MFMAs) at every image start.  Accumulation into gw with f32 atomics, as before.
//
// Requires TY = TX = 3, dilation 1, S in {1, 2}, OW % 16 == 0, Cin % 64 == 0, Cout % 64 == 0, dense G (osy = osx = 1).
#include "b3.hpp"
#include <type_traits>
#include <cstdlib>

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int S> struct W3 {
    static constexpr int XW = 15 * S + 3;          // window columns per row: 18 / 33
    static constexpr int NR = 4;                   // ring of window rows: 3 in use + 1 being written (a task is ONE row)
    static constexpr int XROWS = NR * XW;          // LDS rows (pixels) of the X image per plane
    static constexpr int GROWS = 2 * 16;           // two G buffers of 16 pixels
    static constexpr int XPLANE = XROWS * 128, GPLANE = GROWS * 128;
    static constexpr int LDS = 3 * (XPLANE + GPLANE);
    static constexpr int XLOADS = (XW * 16 + 255) / 256;   // float4 loads per thread and window row: 2 / 3
};

__device__ __forceinline__ int chunk_off(int r, int c) { return (r * 8 + (c ^ (((r >> 1) & 1) << 2))) * 16; }

template <int S, bool SCALE, bool REFLECT>
__global__ __launch_bounds__(256, 2) void conv_b3_wgrad3_kernel(float* __restrict__ gw, const float* __restrict__ gy,
                                                                const float* __restrict__ x, const float* __restrict__ in_scale,
                                                                const float* __restrict__ out_scale, ideas_conv_params p,
                                                                int tiles_ci, int tiles, int splits, int strips, int rows_per_split,
                                                                unsigned gy_bytes, unsigned x_bytes) {
    using L = W3<S>;
    constexpr int XW = L::XW, NR = L::NR;
    __shared__ __attribute__((aligned(16))) unsigned char smem[L::LDS];
    unsigned char* const sG = smem;                       // [3 planes][2 x 16 rows][128 B]
    unsigned char* const sX = smem + 3 * L::GPLANE;       // [3 planes][NR x XW rows][128 B]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int tile, split;
    splitk_xcd_map(blockIdx.x, tiles, splits, tile, split);
    const int o0 = (tile / tiles_ci) * 64, c0 = (tile % tiles_ci) * 64;
    const int strip = split % strips, range = split / strips;
    const int ox0 = strip * 16;
    const int rows_total = p.B * p.OH;
    const int R0 = range * rows_per_split;
    const int R1 = R0 + rows_per_split < rows_total ? R0 + rows_per_split : rows_total;
    if (R0 >= R1) return;

    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)gy, 0, (int)gy_bytes, (int)RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)x_bytes, (int)RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc((void*)out_scale, 0, SCALE ? p.B * p.Cout * 4 : 0, (int)RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rsi = __builtin_amdgcn_make_buffer_rsrc((void*)in_scale, 0, SCALE ? p.B * p.Cin * 4 : 0, (int)RSRC_FLAGS);

    // ---- staging role of a thread: pixel (t >> 4) [+16, +32 for the extra window columns], channel quad (t & 15) -------------
    const int quad = t & 15, px = t >> 4;
    const unsigned g_cb = (unsigned)(o0 + quad * 4) * 4u, x_cb = (unsigned)(c0 + quad * 4) * 4u;
    // LDS row of window column j: S = 1: j;  S = 2: parity * 17 + j / 2
    auto xrow_of = [](int j) { return S == 1 ? j : (j & 1) * 17 + (j >> 1); };
    int x_lrow[L::XLOADS];
    bool x_on[L::XLOADS];
#pragma unroll
    for (int k = 0; k < L::XLOADS; ++k) {
        const int j = px + 16 * k;
        x_on[k] = j < XW;
        x_lrow[k] = xrow_of(j < XW ? j : 0);
    }

    // ---- task stream (block-uniform): one task = one window row of X (+ the G row it completes) ---------------------------------
    // image b, output rows [oa, ob) of the block's range inside it: tasks iy' = oa*S + offy .. , S window rows per output row after
    // a warm-up of (3 - S) rows.  To keep one code path a task is ONE window row; an output row is computed after its last row.
    // stride 1: rows oy-1, oy, oy+1 -> tasks wy = oa-1, oa, oa+1, ..., ob;      compute(oy = wy - 1) when wy >= oa + 1
    // stride 2: rows 2oy, 2oy+1, 2oy+2 -> tasks wy = 2oa, ..., 2(ob-1)+2;        compute(oy = (wy - 2) / 2) when wy even, >= 2oa + 2
    struct Task { int b, wy, oy; bool mma, live; };      // wy: window row index relative to offy (iy = wy + offy)
    int tk_b, tk_oa, tk_ob, tk_wy, tk_g;
    bool tk_done = false;
    {
        tk_g = R0;
        tk_b = R0 / p.OH;
        tk_oa = R0 - tk_b * p.OH;
        const int left = R1 - R0;
        tk_ob = tk_oa + left < p.OH ? tk_oa + left : p.OH;
        tk_wy = tk_oa * S;
    }
    auto next_task = [&]() -> Task {
        Task k;
        k.live = !tk_done;
        k.b = tk_b;
        k.wy = tk_wy;
        const int rel = tk_wy - tk_oa * S;                 // 0, 1, 2, ...
        k.mma = k.live && rel >= 2 && ((rel - 2) % S == 0);
        k.oy = tk_oa + (rel - 2) / S;
        if (!tk_done) {
            ++tk_wy;
            if (tk_wy > (tk_ob - 1) * S + 2) {             // image (or range) finished
                tk_g += tk_ob - tk_oa;
                if (tk_g >= R1) tk_done = true;
                else {
                    ++tk_b;
                    tk_oa = 0;
                    tk_ob = R1 - tk_g < p.OH ? R1 - tk_g : p.OH;
                    tk_wy = 0;
                }
            }
        }
        return k;
    };

    struct Stage { float4 g, xv[L::XLOADS], sg, sx; };
    auto gload = [&](Stage& st, const Task& k) {
        // X window row wy of image b: iy = wy + offy; columns ix = ox0*S + offx + j
        int iy = k.wy + p.offy;
        bool yok = k.live;
        if (REFLECT) iy = reflect_coord(iy, p.IH);
        else yok = yok && (unsigned)iy < (unsigned)p.IH;
        const unsigned rowb = (unsigned)((k.b * p.IH + iy) * p.IW) * (unsigned)p.Cin * 4u + x_cb;
#pragma unroll
        for (int q = 0; q < L::XLOADS; ++q) {
            int ix = ox0 * S + p.offx + px + 16 * q;
            bool ok = yok && x_on[q];
            if (REFLECT) ix = reflect_coord(ix, p.IW);
            else ok = ok && (unsigned)ix < (unsigned)p.IW;
            st.xv[q] = buffer_load4(rx, ok ? rowb + (unsigned)ix * (unsigned)p.Cin * 4u : 0xffffffffu, 0);
        }
        // G row oy (only tasks that complete an output row carry one)
        const unsigned goff = (unsigned)(((k.b * p.OH + k.oy) * p.OW + ox0 + px) * p.Cout) * 4u + g_cb;
        st.g = buffer_load4(rg, k.mma ? goff : 0xffffffffu, 0);
        if (SCALE) {
            st.sg = buffer_load4(rso, (unsigned)(k.b * p.Cout) * 4u + g_cb, 0);
            st.sx = buffer_load4(rsi, (unsigned)(k.b * p.Cin) * 4u + x_cb, 0);
        }
    };
    auto put = [&](unsigned char* plane0, int plane_bytes, int row, float4 v, float4 sc) {
        if (SCALE) v = make_float4(mul_rn(v.x, sc.x), mul_rn(v.y, sc.y), mul_rn(v.z, sc.z), mul_rn(v.w, sc.w));
        const Split4 s = split4(v);
        unsigned char* a = plane0 + chunk_off(row, quad >> 1) + (quad & 1) * 8;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint2*>(a + pl * plane_bytes) = s.p[pl];
    };
    auto lstore = [&](const Stage& st, int n) {            // n = index of the task in the stream
        const int slot = n % NR;
#pragma unroll
        for (int q = 0; q < L::XLOADS; ++q)
            if (x_on[q]) put(sX, L::XPLANE, slot * XW + x_lrow[q], st.xv[q], st.sx);
        put(sG, L::GPLANE, (n & 1) * 16 + px, st.g, st.sg);
    };

    // ---- MFMA side: wave = (o half, ci half); 9 accumulators of 32 (o) x 32 (ci) ---------------------------------------------------
    const int wo = wave >> 1, wc = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int g_q = lane & 15, g_row = g_q >> 2, g_piece = g_q & 3, g_cblk = (lane >> 4) & 1;
    // Transpose reads with NO per-read address arithmetic.  A fragment = 8 consecutive pixel rows r0 + 8 lh .. + 7 of one channel
    // (two transpose blocks of 4 rows); its byte address is chunk_off(r, c) with the 16-byte chunk XOR-swizzled by bit 1 of the row.
    // That bit depends only on (r0 mod 4) and the lane (the lane's row offsets 8 lh and + 4 are multiples of 4), so with the ring
    // position of a step known at compile time (the step loop is unrolled over the four ring positions) an address is one of FOUR
    // lane registers per operand plus an immediate: ds_read_b64_tr_b16 v, v_base[r0 & 3] offset:imm.  (Computed per read, the 60
    // reads of a step cost 145 vector instructions next to its 54 MFMAs -- 2.7 of the kernel's 3.9 VALU per MFMA; on this part
    // every VALU instruction of a SIMD is time taken from its matrix pipe: cycles per MFMA ~ 32 + 4 x VALU per MFMA.)
    auto lane_base = [&](unsigned region, int b, int cb) {
        const int r = b + 8 * lh + g_row;
        return region + (unsigned)(r * 128 + (((cb * 2 + (g_piece >> 1)) ^ ((((b + g_row) >> 1) & 1) << 2)) << 4) + (g_piece & 1) * 8);
    };
    const unsigned ldsG = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)sG;
    const unsigned ldsX = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)sX;
    unsigned baseG[4], baseX[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) { baseG[b] = lane_base(ldsG, b, wo * 2 + g_cblk); baseX[b] = lane_base(ldsX, b, wc * 2 + g_cblk); }
    auto tr_ld = [&](unsigned addr) -> s16x4 {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)addr);
    };
    // r0: compile-time row (after unrolling); plane_off: pl * plane bytes
    auto frag = [&](const unsigned (&base)[4], int plane_off, int r0) -> bf16x8 {
        const unsigned a0 = base[r0 & 3] + (unsigned)(plane_off + (r0 & ~3) * 128);
        const s16x4 a = tr_ld(a0);
        const s16x4 b = tr_ld(a0 + 512u);
        const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        return __builtin_bit_cast(bf16x8, v);
    };

    f32x16 acc[9];
#pragma unroll
    for (int a = 0; a < 9; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    // Nine taps = 54 MFMAs per wave and step.  Two taps are multiplied at a time, alternating their accumulators (no MFMA reads the
    // accumulator the previous one writes), while the transpose reads of the NEXT two taps are already in flight: left to itself
    // the compiler issued two reads, waited for them, issued one MFMA (read latency exposed 30 times per step).
    auto compute = [&](auto ph_) {                         // ring phase PH = n % 4 of the task that completed an output row
        constexpr int PH = decltype(ph_)::value;
        bf16x8 fa[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) fa[pl] = frag(baseG, pl * L::GPLANE, (PH & 1) * 16);
        auto loadB = [&](int tap, bf16x8 (&fb)[3]) {
            const int ty = tap / 3, tx = tap - 3 * ty;
            const int r0 = ((PH + 2 + ty) % NR) * XW + (S == 1 ? tx : (tx & 1) * 17 + (tx >> 1));       // window rows = tasks n-2, n-1, n
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) fb[pl] = frag(baseX, pl * L::XPLANE, r0);
        };
        bf16x8 fb[5][2][3];                                // [pair][tap of the pair][plane] (fully unrolled: plain registers)
        loadB(0, fb[0][0]);
        loadB(1, fb[0][1]);
#pragma unroll
        for (int pr = 0; pr < 5; ++pr) {
            if (pr < 4) {
                loadB(2 * pr + 2, fb[pr + 1][0]);
                if (2 * pr + 3 < 9) loadB(2 * pr + 3, fb[pr + 1][1]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                acc[2 * pr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[q]], fb[pr][0][PB[q]], acc[2 * pr], 0, 0, 0);
                if (2 * pr + 1 < 9)
                    acc[2 * pr + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[q]], fb[pr][1][PB[q]], acc[2 * pr + 1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- pipeline: loads of task n+2 in flight, task n+1 scaled/split into LDS, task n multiplied ---------------------------------
    Stage st0, st1;
    Task k0 = next_task();
    gload(st0, k0);
    Task k1 = next_task();
    gload(st1, k1);
    lstore(st0, 0);
    __syncthreads();
    bool mma0 = k0.mma, mma1 = k1.mma, live0 = k0.live, live1 = k1.live;
    static_assert(NR == 4, "the step loop is unrolled over the ring positions");
    // one step: task n (ring phase PH = n % 4, compile-time) is multiplied, task n + 1 scaled / split into LDS, task n + 2 loaded
    auto step = [&](auto ph_, Stage& ld, const Stage& stg) {
        constexpr int PH = decltype(ph_)::value;
        const Task k2 = next_task();
        gload(ld, k2);
        if (live1) lstore(stg, PH + 1);                    // (lstore only uses its index mod 4 and mod 2)
        if (mma0) compute(ph_);
        __syncthreads();
        mma0 = mma1; live0 = live1; mma1 = k2.mma; live1 = k2.live;
    };
    // four steps per trip, so that the ring position (and with it every LDS read address) is a constant; the stage registers alternate
    while (live0) {
        step(std::integral_constant<int, 0>{}, st0, st1);
        if (!live0) break;
        step(std::integral_constant<int, 1>{}, st1, st0);
        if (!live0) break;
        step(std::integral_constant<int, 2>{}, st0, st1);
        if (!live0) break;
        step(std::integral_constant<int, 3>{}, st1, st0);
    }

    // ---- epilogue: D rows = o (r & 3) + 8 (r >> 2) + 4 lh, column = ci li; gw is OHWI ---------------------------------------------
    const int ci = c0 + wc * 32 + li;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + wo * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            atomicAdd(&gw[((int64_t)o * 9 + tap) * p.Cin + ci], acc[tap][r] * p.gain);
        }
}

// (a stride-2 variant with TWO window rows per barrier was built in round 5, measured at parity with the kernel above -- 0.93-1.06x,
// profiles/r05_wgrad3_s2_ab.txt -- and left the library in round 6: tools/attic/conv_b3_wgrad3_s2pair.hip)

template <int S>
int launch_wgrad3(float* gw, const void* gy, const void* x, const float* in_scale, const float* out_scale, const ideas_conv_params* p,
                  hipStream_t stream) {
    const int tiles_ci = p->Cin / 64, tiles = (p->Cout / 64) * tiles_ci;
    const int strips = p->OW / 16;
    const int64_t rows_total = (int64_t)p->B * p->OH;
    // blocks: about four waves of the 512 resident slots (2 per CU) where a range still has >= 128 rows, two otherwise; >= 16 rows
    const int64_t slots = 512;
    int64_t spl = (4 * slots) / ((int64_t)tiles * strips);
    if (spl < 1 || rows_total / spl < 128) spl = (2 * slots) / ((int64_t)tiles * strips);
    if (spl < 1) spl = 1;
    const int64_t max_spl = rows_total / 16 > 0 ? rows_total / 16 : 1;
    if (spl > max_spl) spl = max_spl;
    int64_t per = ideas_cdiv(rows_total, spl);
    spl = ideas_cdiv(rows_total, per);
    const int64_t splits = spl * strips;
    if ((int64_t)tiles * splits > 0x7fffffffLL) return IDEAS_E_SHAPE;
    const unsigned gy_bytes = (unsigned)((int64_t)p->B * p->YH * p->YW * p->Cout * 4);
    const unsigned x_bytes = (unsigned)((int64_t)p->B * p->IH * p->IW * p->Cin * 4);
    auto go = [&](auto sc, auto rf) {
        hipLaunchKernelGGL((conv_b3_wgrad3_kernel<S, decltype(sc)::value, decltype(rf)::value>), dim3(splitk_grid(tiles, splits)),
                           dim3(256), 0, stream, gw, (const float*)gy, (const float*)x, in_scale, out_scale, *p, tiles_ci, tiles,
                           (int)splits, strips, (int)per, gy_bytes, x_bytes);
    };
    using T = std::true_type;
    using F = std::false_type;
    const bool sc = in_scale && out_scale;
    if (sc) { if (p->reflect) go(T{}, T{}); else go(T{}, F{}); }
    else { if (p->reflect) go(F{}, T{}); else go(F{}, F{}); }
    return ideas_launch_status();
}

}  // namespace

bool ideas_b3_wgrad3_enabled() {
    const char* e = getenv("IDEAS_B3_WGRAD3");          // (read per call: no state in the library, include/ideas_hip.h)
    return !(e && e[0] == '0');
}

extern "C" int ideas_b3_wgrad3_supported(const ideas_conv_params* p) {
    if (!p) return 0;
    if (p->TY != 3 || p->TX != 3 || p->dy != 1 || p->dx != 1 || p->sy != p->sx || (p->sy != 1 && p->sy != 2)) return 0;
    if (p->osy != 1 || p->osx != 1 || p->ooy != 0 || p->oox != 0 || p->YH != p->OH || p->YW != p->OW) return 0;
    if (p->OW % 16 || p->Cin % 64 || p->Cout % 64) return 0;
    // stride 2 (the Blur -> 3x3/s2 convs, the upsampling modulated convs): two window rows, i.e. two barriers, per output row with
    // MFMAs in every other one.  Round 3 measured it 4-7 % SLOWER than conv_b3_wgrad.hip and kept it opt-in; with the address-free
    // transpose reads of round 4 it is ahead: same box, two interleaved runs of 32 iterations, 412.34 / 412.42 -> 411.49 / 411.78 ms
    // (tools/ab_step.sh).  IDEAS_B3_WGRAD3_S2=0 switches the stride-2 path off (A/B measurements).
    if (p->sy == 2) {
        const char* e = getenv("IDEAS_B3_WGRAD3_S2");
        if (e && e[0] == '0') return 0;
    }
    if (p->reflect && (p->IH < 2 || p->IW < 2)) return 0;
    // every window pixel the taps can address must lie inside what one strip stages: ix = ox*S + tx + offx, tx in 0..2
    if (p->offx > 0 || p->offx < -2 || p->offy > 0 || p->offy < -2) return 0;
    return (int64_t)p->B * p->IH * p->IW * p->Cin * 4 < 0xffffffffLL && (int64_t)p->B * p->YH * p->YW * p->Cout * 4 < 0xffffffffLL &&
           (int64_t)p->B * p->OH < 0x7fffffffLL;
}

// called by ideas_conv_wgrad for dtype IDEAS_F32_B3 once the arguments are validated and ideas_b3_wgrad3_supported
int ideas_b3_wgrad3(float* gw, const void* gy, const void* x, const float* in_scale, const float* out_scale, const ideas_conv_params* p,
                    hipStream_t stream) {
    if (p->sy == 1) return launch_wgrad3<1>(gw, gy, x, in_scale, out_scale, p, stream);
    return launch_wgrad3<2>(gw, gy, x, in_scale, out_scale, p, stream);
}
