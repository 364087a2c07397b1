// Removed from ideas_amd/csrc/conv_b3_wgrad3.hip in round 6 (not built; kept for the record -- it lived inside that file's anonymous
// namespace and uses its helpers).

// ---- stride 2, TWO window rows per step ------------------------------------------------------------------------------------------
// conv_b3_wgrad3_kernel<2> stages one window row per barrier: an output row needs two of them, so every other step is staging only
// (loads, split, barrier) with no MFMAs of its own block to hide behind -- 165-190 TFLOP/s against the stride-1 path's 200.  Here a
// task is the PAIR of window rows (2 r - 1, 2 r) and, from the second pair of an image on, output row r - 1 (window rows 2 r - 2 [the
// previous pair's second row], 2 r - 1, 2 r): one barrier and 54 MFMAs per step, one bubble step per image instead of two.
// MEASURED (tools/ab_wgrad3_s2.py, same box, profiles/r05_wgrad3_s2_ab.txt): 0.93-1.06x of the one-row kernel, i.e. parity -- the
// staging-only step was already covered by the CU's other block, and what separates stride 2 from stride 1 is the split itself:
// an output row consumes 66 window pixels instead of 18 (2.2-2.6 vector instructions per MFMA here, ~3 in the one-row kernel, 1.3 at
// stride 1), which no barrier arrangement removes.  Opt-in (IDEAS_B3_WGRAD3_S2=2), results to f32 rounding of the one-row kernel's.
//   ring:   five single-row slots, task n writes slots 2n and 2n + 1 (mod 5), its taps read slots 2n - 1, 2n, 2n + 1; five distinct
//           residues, so task n + 1 is written while task n is multiplied; the step loop is unrolled over n mod 5 (every read address
//           stays an immediate on four lane registers, as in the kernel above);
//   stage:  ONE register stage (66 pixels x 16 channel quads = 5 float4 per thread, the last one 2 pixels): the loads of task n + 1
//           are issued before the MFMAs of task n and split into LDS after them -- the two-stage pipeline above would need 64 + 144
//           (accumulators) + 60 (operands) registers, more than the 256 of two blocks per CU;
//   LDS:    3 x (165 + 32) rows x 128 B = 75.6 KB, two blocks per CU (151 of 160 KB).
struct W3P {
    static constexpr int XW = 33, NR = 5, XROWS = NR * XW;
    static constexpr int XPLANE = XROWS * 128, GPLANE = 32 * 128;
    static constexpr int LDS = 3 * (XPLANE + GPLANE);
    static constexpr int XLOADS = 5;               // (2 x 33 pixels x 16 quads + 255) / 256
};

template <bool SCALE, bool REFLECT>
__global__ __launch_bounds__(256, 2) void conv_b3_wgrad3_s2pair_kernel(float* __restrict__ gw, const float* __restrict__ gy,
                                                                       const float* __restrict__ x, const float* __restrict__ in_scale,
                                                                       const float* __restrict__ out_scale, ideas_conv_params p,
                                                                       int tiles_ci, int tiles, int splits, int strips,
                                                                       int rows_per_split, unsigned gy_bytes, unsigned x_bytes) {
    using L = W3P;
    constexpr int XW = L::XW, NR = L::NR;
    __shared__ __attribute__((aligned(16))) unsigned char smem[L::LDS];
    unsigned char* const sG = smem;                       // [3 planes][2 x 16 rows][128 B]
    unsigned char* const sX = smem + 3 * L::GPLANE;       // [3 planes][5 x 33 rows][128 B]

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

    // ---- staging role: element e = (t >> 4) + 16 k of the pair's 66 pixels (row e / 33, window column e % 33), channel quad t & 15 ----
    const int quad = t & 15, px = t >> 4;
    const unsigned g_cb = (unsigned)(o0 + quad * 4) * 4u, x_cb = (unsigned)(c0 + quad * 4) * 4u;
    int x_lrow[L::XLOADS], x_col[L::XLOADS];
    bool x_on[L::XLOADS], x_second[L::XLOADS];
#pragma unroll
    for (int k = 0; k < L::XLOADS; ++k) {
        const int e = px + 16 * k;
        x_on[k] = e < 2 * XW;
        x_second[k] = e >= XW;
        const int j = x_on[k] ? (e >= XW ? e - XW : e) : 0;
        x_col[k] = j;
        x_lrow[k] = (j & 1) * 17 + (j >> 1);               // window columns de-interleaved by parity
    }

    // ---- task stream (block-uniform): image b, output rows [oa, ob) of the block's range inside it: tasks r = oa .. ob ---------------
    struct Task { int b, r; bool first, live; };           // window rows 2 r - 1 (dead when first) and 2 r; output row r - 1 unless first
    int tk_b, tk_oa, tk_ob, tk_r, tk_g;
    bool tk_done = false;
    {
        tk_g = R0;
        tk_b = R0 / p.OH;
        tk_oa = R0 - tk_b * p.OH;
        const int left = R1 - R0;
        tk_ob = tk_oa + left < p.OH ? tk_oa + left : p.OH;
        tk_r = tk_oa;
    }
    auto next_task = [&]() -> Task {
        Task k;
        k.live = !tk_done;
        k.b = tk_b;
        k.r = tk_r;
        k.first = tk_r == tk_oa;
        if (!tk_done) {
            ++tk_r;
            if (tk_r > tk_ob) {                            // image (or range) finished
                tk_g += tk_ob - tk_oa;
                if (tk_g >= R1) tk_done = true;
                else {
                    ++tk_b;
                    tk_oa = 0;
                    tk_ob = R1 - tk_g < p.OH ? R1 - tk_g : p.OH;
                    tk_r = 0;
                }
            }
        }
        return k;
    };

    struct Stage { float4 g, xv[L::XLOADS], sg, sx; };
    auto gload = [&](Stage& st, const Task& k) {
        // window rows iy = 2 r - 1 + offy, 2 r + offy; columns ix = 2 ox0 + offx + j
        int iy0 = 2 * k.r - 1 + p.offy, iy1 = iy0 + 1;
        bool ok0 = k.live && !k.first, ok1 = k.live;
        if (REFLECT) { iy0 = reflect_coord(iy0, p.IH); iy1 = reflect_coord(iy1, p.IH); }
        else { ok0 = ok0 && (unsigned)iy0 < (unsigned)p.IH; ok1 = ok1 && (unsigned)iy1 < (unsigned)p.IH; }
        const unsigned rowb0 = (unsigned)((k.b * p.IH + (ok0 ? iy0 : 0)) * p.IW) * (unsigned)p.Cin * 4u + x_cb;
        const unsigned rowb1 = (unsigned)((k.b * p.IH + (ok1 ? iy1 : 0)) * p.IW) * (unsigned)p.Cin * 4u + x_cb;
#pragma unroll
        for (int q = 0; q < L::XLOADS; ++q) {
            int ix = ox0 * 2 + p.offx + x_col[q];
            bool ok = x_on[q] && (x_second[q] ? ok1 : ok0);
            if (REFLECT) ix = reflect_coord(ix, p.IW);
            else ok = ok && (unsigned)ix < (unsigned)p.IW;
            st.xv[q] = buffer_load4(rx, ok ? (x_second[q] ? rowb1 : rowb0) + (unsigned)ix * (unsigned)p.Cin * 4u : 0xffffffffu, 0);
        }
        const bool mma = k.live && !k.first;
        const unsigned goff = (unsigned)(((k.b * p.OH + k.r - 1) * p.OW + ox0 + px) * p.Cout) * 4u + g_cb;
        st.g = buffer_load4(rg, mma ? goff : 0xffffffffu, 0);
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
    // n5 = task index mod 5 (compile time), gbuf = task index & 1
    auto lstore = [&](const Stage& st, int n5, int gbuf) {
        const int slot0 = (2 * n5) % NR, slot1 = (2 * n5 + 1) % NR;
#pragma unroll
        for (int q = 0; q < L::XLOADS; ++q)
            if (x_on[q]) put(sX, L::XPLANE, (x_second[q] ? slot1 : slot0) * XW + x_lrow[q], st.xv[q], st.sx);
        put(sG, L::GPLANE, gbuf * 16 + px, st.g, st.sg);
    };

    // ---- MFMA side: as the kernel above ------------------------------------------------------------------------------------------------
    const int wo = wave >> 1, wc = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int g_q = lane & 15, g_row = g_q >> 2, g_piece = g_q & 3, g_cblk = (lane >> 4) & 1;
    auto lane_base = [&](unsigned region, int b, int cb) {
        const int r = b + 8 * lh + g_row;
        return region + (unsigned)(r * 128 + (((cb * 2 + (g_piece >> 1)) ^ ((((b + g_row) >> 1) & 1) << 2)) << 4) + (g_piece & 1) * 8);
    };
    const unsigned ldsG = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)sG;
    const unsigned ldsX = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)sX;
    unsigned baseX[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) baseX[b] = lane_base(ldsX, b, wc * 2 + g_cblk);
    const unsigned baseG = lane_base(ldsG, 0, wo * 2 + g_cblk);
    auto tr_ld = [&](unsigned addr) -> s16x4 {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)addr);
    };
    auto frag_at = [&](unsigned a0) -> bf16x8 {
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

    auto compute = [&](auto n5_, unsigned g_addr) {        // g_addr = baseG + (task & 1) * 16 rows
        constexpr int N5 = decltype(n5_)::value;
        bf16x8 fa[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) fa[pl] = frag_at(g_addr + (unsigned)(pl * L::GPLANE));
        auto loadB = [&](int tap, bf16x8 (&fb)[3]) {
            const int ty = tap / 3, tx = tap - 3 * ty;
            const int r0 = ((2 * N5 + 4 + ty) % NR) * XW + (tx & 1) * 17 + (tx >> 1);      // window rows 2r-2, 2r-1, 2r = slots 2n-1, 2n, 2n+1
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) fb[pl] = frag_at(baseX[r0 & 3] + (unsigned)(pl * L::XPLANE + (r0 & ~3) * 128));
        };
        bf16x8 fb[5][2][3];
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

    // ---- pipeline: task n in LDS is multiplied while the loads of task n + 1 are in flight; they are split into LDS behind the MFMAs ----
    Stage st;
    Task cur = next_task();
    gload(st, cur);
    lstore(st, 0, 0);
    __syncthreads();
    Task nxt = next_task();
    int gbuf = 0;
    auto step = [&](auto n5_) {
        constexpr int N5 = decltype(n5_)::value;
        if (nxt.live) gload(st, nxt);
        if (!cur.first) compute(n5_, baseG + (unsigned)(gbuf * 16 * 128));
        if (nxt.live) lstore(st, (N5 + 1) % NR, gbuf ^ 1);
        __syncthreads();
        gbuf ^= 1;
        cur = nxt;
        nxt = next_task();
    };
    while (cur.live) {
        step(std::integral_constant<int, 0>{});
        if (!cur.live) break;
        step(std::integral_constant<int, 1>{});
        if (!cur.live) break;
        step(std::integral_constant<int, 2>{});
        if (!cur.live) break;
        step(std::integral_constant<int, 3>{});
        if (!cur.live) break;
        step(std::integral_constant<int, 4>{});
    }

    const int ci = c0 + wc * 32 + li;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + wo * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            atomicAdd(&gw[((int64_t)o * 9 + tap) * p.Cin + ci], acc[tap][r] * p.gain);
        }
}

