// Pointwise (1x1, stride 1) convolutions with 32 / 64 / 128 input channels of the bf16 family as a flat, HBM-bound GEMM
//     y[M][Cout] = epilogue(gain * x[M][Cin] . W^T)          (bf16 in HBM, one v_mfma_f32_32x32x16_bf16 per product, f32 epilogue)
// -- the bf16 twin of conv_b3_pw.hip: the skip branches of the residual blocks (models.py:195-206, 170-178), their input gradients
// and the other 1x1 layers with few input channels.  On conv_bf16_kernel these run at 1.7-3.1 TB/s (tools/step_census2.py,
// PRECISION=bf16: [96 x 128 x 128] pixels, 64 -> 128 channels + the merge add in 0.325 ms): with K = 64 a 128 x 128 tile is two
// K-steps between a prologue that derives every row's address with divisions, three barriers and an epilogue that looks every row
// up in LDS.  A flat GEMM needs none of it:
//   * the x tile of 128 pixels is ONE contiguous run of 128 * Cin bf16: it goes global -> LDS by LDS-DMA as whole pixel rows (a
//     64-lane DMA covers 1 KB = 16 / 8 / 4 complete rows: full cache lines), with the bank swizzle of the fragment reads applied in
//     the per-lane SOURCE offset; two LDS tiles, the DMA of tile t + 1 flies under the contraction and the epilogue of tile t;
//   * a wave keeps the weight fragments of ITS output channels for all of K in registers for the whole launch (read once from the
//     packed weights of ideas_bf16_pack_weights, whose layout they already have): the K loop issues no weight load;
//   * blocks are persistent (two per CU) and walk tiles b, b + G, ...; one barrier per tile, in front of it a COUNTED vmcnt wait
//     (the tile's own stores may still be in flight; everything issued before them -- the DMA -- has landed: gfx9 returns VMEM in
//     order).  Rows past M and channels past Cout cost nothing: DMA sources past the tensor deliver zeros, stores to an
//     out-of-range offset are dropped -- and every store IS issued (no branch around it), which is what makes the count a constant.
// MFMA roles, K order (ascending 16-channel slices) and the epilogue expression are conv_bf16_body's: the results are BITWISE the
// generic kernel's (tests/test_ops_gpu.py).
#include "common.hpp"
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef unsigned short bf16_t;

namespace {

constexpr unsigned RSRC = 0x00020000u;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    const f32x2v v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v));
}
__device__ __forceinline__ float bf_lo(unsigned pk) { return __builtin_bit_cast(float, pk << 16); }
__device__ __forceinline__ float bf_hi(unsigned pk) { return __builtin_bit_cast(float, pk & 0xffff0000u); }
template <int N> __device__ __forceinline__ void wait_step() {       // (conv_bf16.hip: the LDS reads must have returned as well)
    static_assert(N >= 0 && N <= 63, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}
// lanes li / li + 32 hold channel quads 8g + 4lh of one pixel: after the swap lane (li, 0) has channels 0-15, (li, 1) 16-31 (conv_bf16.hip)
__device__ __forceinline__ void quad_exchange(const uint2 (&q)[4], uint4 (&out)[2]) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const auto rx = __builtin_amdgcn_permlane32_swap(q[g].x, q[g + 2].x, false, false);
        const auto ry = __builtin_amdgcn_permlane32_swap(q[g].y, q[g + 2].y, false, false);
        out[g] = make_uint4(rx[0], ry[0], rx[1], ry[1]);
    }
}

// KS = Cin / 32 (1, 2, 4); a block = 2 x 2 waves over 128 pixels x (64 NT) channels per pass, NP passes with their weights resident
template <int KS, int NT, int NP>
__global__ __launch_bounds__(256, 2) void conv_bf16_pw_kernel(bf16_t* __restrict__ y, const bf16_t* __restrict__ x,
                                                              const void* __restrict__ wpack, const float* __restrict__ bias,
                                                              const bf16_t* __restrict__ resid, ideas_conv_params p, unsigned M,
                                                              int ntiles) {
    constexpr int MT = 2, BM = 128;
    constexpr int CIN = KS * 32, RB = CIN * 2;           // bytes of one pixel row
    constexpr int CR = RB / 16;                          // 16-byte chunks per row: 4 / 8 / 16
    constexpr int RPP = 64 / CR;                         // rows per 1 KB DMA piece: 16 / 8 / 4
    constexpr int PIECES = BM / RPP, PER = PIECES / 4;   // DMA pieces per tile and per wave: 2 / 4 / 8
    constexpr int RSH = CR == 4 ? 2 : CR == 8 ? 1 : 0;   // chunk c of row r sits at position c ^ ((r >> RSH) & (CR - 1)): the 16 lanes of a
                                                         // ds_read_b128 group then fall into 16 different 16-byte bank groups (rows x chunks)
    constexpr int TILE = BM * RB;
    constexpr int NSL = 2 * KS;                          // 16-channel K-slices
    constexpr int NSTORE = MT * NT * NP * 2;             // 16-byte stores per wave and tile
    static_assert(NSTORE <= 63, "vmcnt immediate");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * TILE];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const unsigned x_bytes = M * (unsigned)RB, y_bytes = M * (unsigned)p.Cout * 2u;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)x_bytes, (int)RSRC);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)wpack, 0, CIN * p.Cout * 2, (int)RSRC);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, (int)y_bytes, (int)RSRC);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)resid, 0, resid ? (int)y_bytes : 0, (int)RSRC);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)bias, 0, bias ? p.Cout * 4 : 0, (int)RSRC);

    // DMA: piece q = wave + 4 j holds rows q RPP .. + RPP - 1; lane l fills LDS slot l = (row l / CR, position l % CR), which must hold
    // chunk (l % CR) ^ swizzle(row) of that row
    unsigned d_src[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int r = (wave + 4 * j) * RPP + lane / CR;
        const int c = (lane % CR) ^ ((r >> RSH) & (CR - 1));
        d_src[j] = (unsigned)(r * RB + c * 16);
    }
    auto dma = [&](int tile, int buf) {
        const unsigned base = (unsigned)tile * (unsigned)TILE;              // (rows past M: offsets past x_bytes -> zeros)
#pragma unroll
        for (int j = 0; j < PER; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(smem + buf * TILE + (wave + 4 * j) * 1024), 16, (int)(base + d_src[j]), 0, 0, 0);
    };

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    dma(tile, 0);

    // weights (A operand: lane li = output channel, half lh = channels 8 lh .. + 7 of the slice), packed [Cin/32][Cout][32] bf16 with
    // chunk c of row n at position c ^ ((n >> 2) & 3); channels past Cout: zeros
    bf16x8 fw[NP][NT][NSL];
#pragma unroll
    for (int np = 0; np < NP; ++np)
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const int n = np * (64 * NT) + (wn * NT + b) * 32 + li;
#pragma unroll
            for (int s = 0; s < NSL; ++s) {
                const unsigned off = (unsigned)(((s >> 1) * p.Cout + n) * 64 + (((2 * (s & 1) + lh) ^ ((n >> 2) & 3)) << 4));
                fw[np][b][s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, (int)(n < p.Cout ? off : 0xffffffffu), 0, 0));
            }
        }

    // B operand (pixels): row (wm MT + a) 32 + li, slice s, half lh -> chunk 2 s + lh
    int fx_off[MT];
#pragma unroll
    for (int a = 0; a < MT; ++a) fx_off[a] = ((wm * MT + a) * 32 + li) * RB;
    const int rsw = (li >> RSH) & (CR - 1);              // (the row's block offset is a multiple of 32: it does not reach the swizzle bits)

    wait_step<0>();
    __builtin_amdgcn_s_barrier();
    int buf = 0;
    for (;;) {
        const int next = tile + (int)gridDim.x;
        if (next < ntiles) dma(next, buf ^ 1);
        const unsigned char* base = smem + buf * TILE;
        unsigned cout2 = (unsigned)p.Cout * 2u;
        int lh4 = 4 * lh;
        asm volatile("" : "+s"(cout2), "+v"(lh4));       // (conv_b3_pw.hip: keeps the per-row / per-quad offsets out of the persistent loop's registers)
#pragma unroll
        for (int np = 0; np < NP; ++np) {
            if (np) __builtin_amdgcn_sched_barrier(0);   // (a pass's accumulators die before the next pass's are born)
            f32x16 acc[MT][NT];
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
#pragma unroll
            for (int s = 0; s < NSL; ++s) {
                bf16x8 fx[MT];
#pragma unroll
                for (int a = 0; a < MT; ++a) fx[a] = *reinterpret_cast<const bf16x8*>(base + fx_off[a] + (((2 * s + lh) ^ rsw) << 4));
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int b = 0; b < NT; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[np][b][s], fx[a], acc[a][b], 0, 0, 0);
                if (NP * NT * NSL > 16 && (s & 1)) __builtin_amdgcn_sched_barrier(0);   // (128 weight registers: no room for every slice's pixel fragments at once)
            }
            // ---- epilogue: lane = pixel li of block a; registers 4g .. 4g+3 = channels 8g + 4lh .. + 3 of block b (conv_bf16_body's) ----
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                const unsigned row = (unsigned)tile * (unsigned)BM + (unsigned)((wm * MT + a) * 32 + li);
                const unsigned yrow = row * cout2;                                       // rows past M: >= y_bytes
#pragma unroll
                for (int b = 0; b < NT; ++b) {
                    const int nb = np * (64 * NT) + (wn * NT + b) * 32;
                    uint2 q[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n = nb + 8 * g + lh4;
                        const bool ok = n < p.Cout;                                     // Cout % 8 == 0
                        float bvv[4] = {0.f, 0.f, 0.f, 0.f};
                        if (bias) {
                            const auto bq = __builtin_amdgcn_raw_buffer_load_b128(rb, ok ? n * 4 : (int)0xffffffffu, 0, 0);
                            const float4 bv = __builtin_bit_cast(float4, bq);
                            bvv[0] = bv.x; bvv[1] = bv.y; bvv[2] = bv.z; bvv[3] = bv.w;
                        }
                        uint2 rq = make_uint2(0u, 0u);
                        if (resid) rq = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rr, (int)(ok ? yrow + (unsigned)n * 2u : 0xffffffffu), 0, 0));
                        const float rv[4] = {bf_lo(rq.x), bf_hi(rq.x), bf_lo(rq.y), bf_hi(rq.y)};
                        float v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float u = mul_then_add(acc[a][b][4 * g + j], p.gain, bvv[j]);   // (= the generic kernel's fma(acc * gain, out_scale = 1, bias))
                            if (p.act) u = (u > 0.f ? u : u * p.alpha) * p.act_gain;
                            if (resid) u = (u + rv[j]) * p.resid_gain;
                            v[j] = u;
                        }
                        q[g] = make_uint2(pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]));
                    }
                    uint4 ch[2];
                    quad_exchange(q, ch);
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const int n = nb + 8 * c + 4 * lh4;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, ch[c]), ry,
                                                               (int)(n < p.Cout ? yrow + (unsigned)n * 2u : 0xffffffffu), 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);   // (one 32 x 32 block at a time: hoisting every block's bias / residual loads costs 50 registers)
                }
            }
        }
        if (next >= ntiles) break;
        wait_step<NSTORE>();                             // the DMA of `next` (issued before this tile's stores) has landed; my LDS reads returned
        __builtin_amdgcn_s_barrier();                    // ... everybody's; every wave is done with `buf`
        tile = next;
        buf ^= 1;
    }
}

template <int KS>
int launch_bf16_pw(void* y, const void* x, const void* wpack, const float* bias, const void* resid, const ideas_conv_params* p,
                   hipStream_t stream) {
    const int64_t M = (int64_t)p->B * p->OH * p->OW;
    const int64_t ntiles = ideas_cdiv(M, 128);
    // two persistent blocks per CU.  (Three or four where registers and LDS allow it were measured -- tools/ab_pw_bf16.py of round 5,
    // occupancy-derived grid: 64 -> 128 with the merge add 4.30 -> 3.83 TB/s, the others unchanged -- and not kept.)
    const unsigned grid = (unsigned)(ntiles < 512 ? ntiles : 512);
    auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, stream, (bf16_t*)y, (const bf16_t*)x, wpack, bias, (const bf16_t*)resid, *p,
                           (unsigned)M, (int)ntiles);
    };
    if (p->Cout <= 64) go(conv_bf16_pw_kernel<KS, 1, 1>);
    else if (p->Cout <= 128) go(conv_bf16_pw_kernel<KS, 2, 1>);
    else go(conv_bf16_pw_kernel<KS, 2, 2>);
    return ideas_launch_status();
}

}  // namespace

// 1 when ideas_bf16_pw_fwd takes the launch: a single-tap, unit-stride, unpadded geometry whose output grid IS the input grid, shared
// weights, no per-sample output scale, no accumulation; Cin 32 / 64 / 128, Cout % 8 == 0 up to 256, 16-byte aligned y.
// IDEAS_BF16_PW=0: never (A/B; read per call).
int ideas_bf16_pw_ok(const ideas_conv_params* p, int per_image, const float* out_scale, const void* y, const void* resid) {
    const char* e = getenv("IDEAS_BF16_PW");
    if (e && e[0] == '0') return 0;
    if (per_image || out_scale || p->accumulate || p->reflect) return 0;
    if (p->TY != 1 || p->TX != 1 || p->sy != 1 || p->sx != 1 || p->offy != 0 || p->offx != 0) return 0;
    if (p->osy != 1 || p->osx != 1 || p->ooy != 0 || p->oox != 0) return 0;
    if (p->OH != p->IH || p->OW != p->IW || p->YH != p->IH || p->YW != p->IW) return 0;
    if (p->Cin != 32 && p->Cin != 64 && p->Cin != 128) return 0;
    if (p->Cout % 8 || p->Cout < 8 || p->Cout > 256) return 0;
    if (!ideas_aligned16(y) || (resid && !ideas_aligned16(resid))) return 0;
    const int64_t M = (int64_t)p->B * p->OH * p->OW;
    // (+ one tile of slack: the DMA of the last, partial tile forms offsets past M rows)
    return (M + 128) * p->Cin * 2 < 0xffffffffLL && (M + 128) * p->Cout * 2 < 0xffffffffLL;
}

int ideas_bf16_pw_fwd(void* y, const void* x, const void* wpack, const float* bias, const void* resid, const ideas_conv_params* p,
                      hipStream_t stream) {
    switch (p->Cin) {
        case 32: return launch_bf16_pw<1>(y, x, wpack, bias, resid, p, stream);
        case 64: return launch_bf16_pw<2>(y, x, wpack, bias, resid, p, stream);
        default: return launch_bf16_pw<4>(y, x, wpack, bias, resid, p, stream);
    }
}
