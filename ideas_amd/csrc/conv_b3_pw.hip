// Pointwise (1x1, stride 1) convolutions with few input channels as a flat, HBM-bound GEMM  y[M][Cout] = gain * x[M][Cin] . W^T
// on the split-bf16 contraction of b3.hpp -- the skip branches of the residual blocks (models.py:195-206, 170-178: a 1x1 conv behind
// the decimating FIR / in front of the zero-stuffing one), their input gradients, and the other 1x1 layers with Cin <= 128.
//
// Why a kernel of its own (round 5): the per-launch census of the step (tools/step_census2.py) shows the 1x1 layers on the generic
// conv_b3_kernel at 16-51 TFLOP/s where they are bound by HBM, not by the matrix pipe -- [96 x 128 x 128] pixels, 64 -> 128 channels
// moves 1.2 GB and took 0.90 ms (1.3 TB/s).  With K = Cin = 64 a 128 x 128 tile is four K-steps between a prologue that derives every
// row's address with divisions, two block barriers and an epilogue that looks every row up in LDS: the tile's fixed costs are the
// kernel.  A flat GEMM needs none of it:
//   * the x tile of BM pixels is ONE contiguous run of BM * Cin floats (NHWC, no padding, no stride): thread t fetches 16-byte
//     pieces t, t + 256, ...: perfectly coalesced, and the same addresses plus a tile stride for every tile;
//   * all of K fits in LDS at once (three bf16 planes of BM x Cin = 48 KB), so a tile is: split + store, ONE barrier, contract,
//     store -- and the next tile's loads are already in flight (register prefetch) while this one is contracted;
//   * a wave owns 32 output channels and keeps its weight fragments for ALL of K in registers (Cin / 16 x 3 planes x 4 VGPRs) for
//     the whole launch when Cout <= 128: the K loop issues no weight load at all (more output channels: passes of 128, fragments
//     re-read from L2 per pass);
//   * blocks are persistent (two per CU) and walk tiles b, b + G, ...; rows past M and channels past Cout cost nothing: the loads
//     return zeros (buffer bounds), the stores are dropped by the same bounds / an out-of-range offset.
// Arithmetic, op order and epilogue (gain, bias, leaky-ReLU, residual) are those of conv_b3_kernel: each output is the same six
// plane-pair MFMA chains over K in the same order, so results are BITWISE the generic kernel's (tests/test_ops_gpu.py).
#include "b3.hpp"
#include <cstdlib>

namespace {

// NWC = waves side by side along the channels (4: passes of 128 channels, every wave all MT row blocks; 2: passes of 64 channels, the
// upper two waves take the lower half of the rows -- Cout <= 64 would leave two of four waves without work otherwise)
template <int KS, int MT, int NWC>
__global__ __launch_bounds__(256, 2) void conv_b3_pw_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                            const void* __restrict__ wplanes, const float* __restrict__ bias,
                                                            const float* __restrict__ resid, ideas_conv_params p, unsigned M,
                                                            unsigned plane_bytes, int ntiles, int npass) {
    constexpr int CIN = KS * 16, BM = MT * 32;
    constexpr int Q = CIN / 4;                       // 16-byte pieces per row
    constexpr int NV = BM * Q / 256;                 // pieces per thread and tile
    static_assert(BM * Q % 256 == 0, "tile = whole pieces per thread");
    constexpr int MTW = MT * NWC / 4;                // row blocks per wave
    static_assert(MTW >= 1, "a wave owns at least one 32-row block");
    constexpr int STEPB = BM * ROWB;                 // bytes of one K-step of one plane
    constexpr int PLANE = KS * STEPB;
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * PLANE];

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6, li = lane & 31, lh = lane >> 5;
    const int wc = wave % NWC, wr = wave / NWC;      // channel block / row group of this wave
    const unsigned x_bytes = M * (unsigned)(CIN * 4), y_bytes = M * (unsigned)p.Cout * 4u;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)x_bytes, (int)RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)wplanes, 0, (int)(3u * plane_bytes), (int)RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, (int)y_bytes, (int)RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)resid, 0, resid ? (int)y_bytes : 0, (int)RSRC_FLAGS);

    // staging: piece f = t + 256 j of the tile -> (row, quad) -> its 8 bytes in every plane
    int a_lds[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int f = t + 256 * j, row = f / Q, q = f - row * Q;
        a_lds[j] = (q >> 2) * STEPB + row * ROWB + (((q & 3) * 8) ^ (((row >> 3) & 1) << 4));
    }
    float4 pre[NV];
    auto prefetch = [&](int tile) {
        const unsigned base = (unsigned)tile * (unsigned)(BM * CIN * 4) + (unsigned)t * 16u;       // < x_bytes + one tile: fits 32 bits (launcher)
#pragma unroll
        for (int j = 0; j < NV; ++j) pre[j] = buffer_load4(rx, base + (unsigned)j * 4096u, 0);     // rows past M: zeros
    };
    auto stage = [&]() {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const Split4 s = split4(pre[j]);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint2*>(smem + a_lds[j] + pl * PLANE) = s.p[pl];
        }
    };
    // weights of pass np: this wave's 32 channels, every K-step, three planes
    bf16x8 fb[KS][3];
    auto loadB = [&](int np) {
        const unsigned voff = (unsigned)((np * (32 * NWC) + wc * 32 + li) * 32 + lh * 16);
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                fb[s][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                    rw, (int)(voff + (unsigned)pl * plane_bytes), (int)((unsigned)s * (unsigned)p.Cout * 32u), 0));
    };
    const int f_off = (wr * MTW * 32 + li) * ROWB + ((lh ^ ((li >> 3) & 1)) << 4);

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    prefetch(tile);
    if (npass == 1) loadB(0);
    for (;;) {
        stage();
        __syncthreads();
        const int next = tile + gridDim.x;
        if (next < ntiles) prefetch(next);
        __builtin_amdgcn_sched_barrier(0);            // the loads stay up here, in flight under the contraction
        for (int np = 0; np < npass; ++np) {
            if (npass > 1) loadB(np);
            f32x16 acc[MTW];
#pragma unroll
            for (int a = 0; a < MTW; ++a)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                bf16x8 fa[MTW][3];
#pragma unroll
                for (int a = 0; a < MTW; ++a)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        fa[a][pl] = *reinterpret_cast<const bf16x8*>(smem + pl * PLANE + s * STEPB + a * 32 * ROWB + f_off);
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int a = 0; a < MTW; ++a)
                        acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][PA[q]], fb[s][PB[q]], acc[a], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);    // (the fragment reads of all K-steps at once would not fit the register file)
            }
            // ---- epilogue: lane = channel n, acc[a][e] = row a*32 + (e&3) + 8 (e>>2) + 4 lh: 128-byte runs per row.  The lane part of
            // an address is ONE register (ybase), the row part is uniform and rides in the instruction's scalar offset; `cout4` is made
            // opaque per tile so that the 16 * MT row offsets are re-derived by the scalar ALU here instead of being hoisted out of the
            // persistent loop into 64 live registers (the first version spilled 250 of them).
            const int n = np * (32 * NWC) + wc * 32 + li;
            unsigned cout4 = (unsigned)p.Cout * 4u;
            asm volatile("" : "+s"(cout4));
            if (n < p.Cout) {
                const float bv = bias ? bias[n] : 0.f;
                const unsigned ybase = ((unsigned)tile * (unsigned)BM + (unsigned)(wr * MTW * 32 + 4 * lh)) * cout4 + (unsigned)n * 4u;
                // A ragged last tile (rows past M) must not depend on how the hardware range-checks the SCALAR offset (LLVM documents
                // soffset as excluded from the bounds check): there the whole offset goes into the lane register, where row >= M <=>
                // offset >= y_bytes (n < Cout; no 32-bit wrap: ideas_b3_pw_ok keeps (M + 128) * Cout * 4 below 2^32) and the access is
                // dropped / returns zero by the descriptor's bounds.  One row group at a time (sched_barrier) so that the 16 lane offsets
                // of a group are the only ones alive.  Full tiles keep the one-register form (uniform branch).
                const bool ragged = ((unsigned)tile + 1u) * (unsigned)BM > M;
#pragma unroll
                for (int a = 0; a < MTW; ++a) {
                    float rv[16];
                    if (ragged) {
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4) {
                            unsigned off[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) off[k] = ybase + (unsigned)(a * 32 + k + 8 * e4) * cout4;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                float v = mul_rn(acc[a][4 * e4 + k], p.gain);
                                v = mul_then_add(v, 1.0f, bv);
                                if (p.act) v = (v > 0.f ? v : v * p.alpha) * p.act_gain;
                                if (resid) v = (v + __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, (int)off[k], 0, 0))) * p.resid_gain;
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, (int)off[k], 0, 0);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        continue;
                    }
                    if (resid) {
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            rv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                rr, (int)ybase, (int)((unsigned)(a * 32 + (e & 3) + 8 * (e >> 2)) * cout4), 0));
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float v = mul_rn(acc[a][e], p.gain);
                        v = mul_then_add(v, 1.0f, bv);
                        if (p.act) v = (v > 0.f ? v : v * p.alpha) * p.act_gain;
                        if (resid) v = (v + rv[e]) * p.resid_gain;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, (int)ybase,
                                                              (int)((unsigned)(a * 32 + (e & 3) + 8 * (e >> 2)) * cout4), 0);
                    }
                }
            }
        }
        if (next >= ntiles) break;
        tile = next;
        __syncthreads();                              // every wave is done reading the planes
    }
}

template <int KS, int MT>
int launch_pw(void* y, const void* x, const void* wplanes, const float* bias, const void* resid, const ideas_conv_params* p,
              hipStream_t stream) {
    constexpr int BM = MT * 32;
    const int64_t M = (int64_t)p->B * p->OH * p->OW;
    const int64_t ntiles = ideas_cdiv(M, BM);
    const unsigned plane_bytes = (unsigned)((int64_t)p->Cin * p->Cout * 2);
    const unsigned grid = (unsigned)(ntiles < 512 ? ntiles : 512);        // two persistent blocks per CU
    if (p->Cout <= 64)
        hipLaunchKernelGGL((conv_b3_pw_kernel<KS, MT, 2>), dim3(grid), dim3(256), 0, stream, (float*)y, (const float*)x, wplanes, bias,
                           (const float*)resid, *p, (unsigned)M, plane_bytes, (int)ntiles, (int)ideas_cdiv(p->Cout, 64));
    else
        hipLaunchKernelGGL((conv_b3_pw_kernel<KS, MT, 4>), dim3(grid), dim3(256), 0, stream, (float*)y, (const float*)x, wplanes, bias,
                           (const float*)resid, *p, (unsigned)M, plane_bytes, (int)ntiles, (int)ideas_cdiv(p->Cout, 128));
    return ideas_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same flat GEMM for MORE input channels (Cin a multiple of 128: 256 .. 1024): K no longer fits LDS at once, so a tile of 64
// pixels walks K in chunks of 128 channels (one staging + one barrier pair per chunk, the next chunk's -- or the next tile's -- loads
// in flight under the contraction), and the weight fragments come straight from L2 one K-step ahead of the MFMAs that use them.  A
// block owns up to 256 output channels (two passes of 128 with their own accumulators; blockIdx.y = the 256-channel group), so the
// x tile is staged once for all of them.  Same arithmetic and K order as the generic kernel: bitwise the same results.
// ---------------------------------------------------------------------------------------------------------------------------
template <int NWC>
__global__ __launch_bounds__(256, 2) void conv_b3_pwk_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                             const void* __restrict__ wplanes, const float* __restrict__ bias,
                                                             const float* __restrict__ resid, ideas_conv_params p, unsigned M,
                                                             unsigned plane_bytes, int ntiles, int nchunks) {
    constexpr int KS = 8, CH = 128, MT = 2, BM = MT * 32;
    constexpr int Q = CH / 4, NV = BM * Q / 256;     // 8 pieces per thread and chunk
    constexpr int MTW = MT * NWC / 4;
    constexpr int STEPB = BM * ROWB, PLANE = KS * STEPB;
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * PLANE];

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6, li = lane & 31, lh = lane >> 5;
    const int wc = wave % NWC, wr = wave / NWC;
    const int ngrp0 = blockIdx.y * 256;              // first output channel of this block's group
    const int npass = (p.Cout - ngrp0 + 32 * NWC - 1) / (32 * NWC) < 256 / (32 * NWC) ? (p.Cout - ngrp0 + 32 * NWC - 1) / (32 * NWC) : 256 / (32 * NWC);
    const unsigned cin4 = (unsigned)p.Cin * 4u;
    const unsigned x_bytes = M * cin4, y_bytes = M * (unsigned)p.Cout * 4u;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)x_bytes, (int)RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)wplanes, 0, (int)(3u * plane_bytes), (int)RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, (int)y_bytes, (int)RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)resid, 0, resid ? (int)y_bytes : 0, (int)RSRC_FLAGS);

    // staging: piece f = t + 256 j of a chunk -> row f / 32, quad f % 32
    int a_lds[NV];
    unsigned a_off[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int f = t + 256 * j, row = f / Q, q = f - row * Q;
        a_lds[j] = (q >> 2) * STEPB + row * ROWB + (((q & 3) * 8) ^ (((row >> 3) & 1) << 4));
        a_off[j] = (unsigned)row * cin4 + (unsigned)q * 16u;
    }
    float4 pre[NV];
    auto prefetch = [&](int tile, int c) {
        const unsigned base = (unsigned)tile * (unsigned)BM * cin4 + (unsigned)c * (unsigned)(CH * 4);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const unsigned off = base + a_off[j];
            pre[j] = buffer_load4(rx, off < x_bytes ? off : 0xffffffffu, 0);      // rows past M: zeros (a row's pieces are in or out together)
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const Split4 s = split4(pre[j]);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint2*>(smem + a_lds[j] + pl * PLANE) = s.p[pl];
        }
    };
    struct BF { bf16x8 f[3]; };
    auto loadB = [&](BF& fb, int np, int kstep) {      // K-step kstep (of all of K) of this wave's 32 channels in pass np
        const unsigned voff = (unsigned)((ngrp0 + np * (32 * NWC) + wc * 32 + li) * 32 + lh * 16);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            fb.f[pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                rw, (int)(voff + (unsigned)pl * plane_bytes), (int)((unsigned)kstep * (unsigned)p.Cout * 32u), 0));
    };
    const int f_off = (wr * MTW * 32 + li) * ROWB + ((lh ^ ((li >> 3) & 1)) << 4);

    f32x16 acc[2][MTW];
    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    prefetch(tile, 0);
    for (;;) {
#pragma unroll
        for (int np = 0; np < 2; ++np)
#pragma unroll
            for (int a = 0; a < MTW; ++a)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[np][a][e] = 0.f;
        const int next = tile + gridDim.x;
        for (int c = 0; c < nchunks; ++c) {
            stage();
            __syncthreads();
            if (c + 1 < nchunks) prefetch(tile, c + 1);
            else if (next < ntiles) prefetch(next, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int np = 0; np < 2; ++np) {
                if (np < npass) {                        // uniform
                    BF fb0, fb1;
                    loadB(fb0, np, c * KS);
#pragma unroll
                    for (int s = 0; s < KS; s += 2) {
                        loadB(fb1, np, c * KS + s + 1);
                        {
                            bf16x8 fa[MTW][3];
#pragma unroll
                            for (int a = 0; a < MTW; ++a)
#pragma unroll
                                for (int pl = 0; pl < 3; ++pl)
                                    fa[a][pl] = *reinterpret_cast<const bf16x8*>(smem + pl * PLANE + s * STEPB + a * 32 * ROWB + f_off);
#pragma unroll
                            for (int q = 0; q < 6; ++q)
#pragma unroll
                                for (int a = 0; a < MTW; ++a)
                                    acc[np][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][PA[q]], fb0.f[PB[q]], acc[np][a], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (s + 2 < KS) loadB(fb0, np, c * KS + s + 2);
                        {
                            bf16x8 fa[MTW][3];
#pragma unroll
                            for (int a = 0; a < MTW; ++a)
#pragma unroll
                                for (int pl = 0; pl < 3; ++pl)
                                    fa[a][pl] = *reinterpret_cast<const bf16x8*>(smem + pl * PLANE + (s + 1) * STEPB + a * 32 * ROWB + f_off);
#pragma unroll
                            for (int q = 0; q < 6; ++q)
#pragma unroll
                                for (int a = 0; a < MTW; ++a)
                                    acc[np][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][PA[q]], fb1.f[PB[q]], acc[np][a], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            __syncthreads();                           // every wave is done reading the planes of this chunk
        }
        // ---- epilogue (conv_b3_pw_kernel's) ----
        unsigned cout4 = (unsigned)p.Cout * 4u;
        asm volatile("" : "+s"(cout4));
#pragma unroll
        for (int np = 0; np < 2; ++np) {
            const int n = ngrp0 + np * (32 * NWC) + wc * 32 + li;
            if (np < npass && n < p.Cout) {
                const float bv = bias ? bias[n] : 0.f;
                const unsigned ybase = ((unsigned)tile * (unsigned)BM + (unsigned)(wr * MTW * 32 + 4 * lh)) * cout4 + (unsigned)n * 4u;
                // (ragged last tile: row offsets in the lane register, bounds-checked -- see conv_b3_pw_kernel)
                const bool ragged = ((unsigned)tile + 1u) * (unsigned)BM > M;
#pragma unroll
                for (int a = 0; a < MTW; ++a) {
                    float rv[16];
                    if (ragged) {
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4) {
                            unsigned off[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) off[k] = ybase + (unsigned)(a * 32 + k + 8 * e4) * cout4;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                float v = mul_rn(acc[np][a][4 * e4 + k], p.gain);
                                v = mul_then_add(v, 1.0f, bv);
                                if (p.act) v = (v > 0.f ? v : v * p.alpha) * p.act_gain;
                                if (resid) v = (v + __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, (int)off[k], 0, 0))) * p.resid_gain;
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, (int)off[k], 0, 0);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        continue;
                    }
                    if (resid) {
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            rv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                rr, (int)ybase, (int)((unsigned)(a * 32 + (e & 3) + 8 * (e >> 2)) * cout4), 0));
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float v = mul_rn(acc[np][a][e], p.gain);
                        v = mul_then_add(v, 1.0f, bv);
                        if (p.act) v = (v > 0.f ? v : v * p.alpha) * p.act_gain;
                        if (resid) v = (v + rv[e]) * p.resid_gain;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, (int)ybase,
                                                              (int)((unsigned)(a * 32 + (e & 3) + 8 * (e >> 2)) * cout4), 0);
                    }
                }
            }
        }
        if (next >= ntiles) break;
        tile = next;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Weight gradient of the same layers:  gw[o][ci] += gain * sum_m gy[m][o] * x[m][ci]  -- a reduction over ALL pixels of two flat
// matrices, bound by reading them once.  conv_b3_wgrad.hip ran these at 10-110 TFLOP/s (its tiles stage 128 channels x 16 pixels per
// 96 MFMAs with a register transpose; with one tap there is nothing to amortise that over).  Here, as in conv_b3_wgrad3.hip, the
// operands stay PIXEL-major in LDS exactly as the 16-byte loads arrive (rows of 64 channels x bf16 = 128 B, three planes, 16-byte
// chunks XOR-swizzled by bit 1 of the row) and the MFMA operand -- 8 consecutive pixels of one channel per lane -- is gathered by the
// LDS transpose read ds_read_b64_tr_b16.  A block owns one 64 (o) x 64 (ci) tile and one contiguous pixel range; a step is 32 pixels:
// every thread loads, splits and stores two 16-byte pieces of each operand, every wave (a 32 x 32 quarter of the tile) issues 12
// MFMAs; loads run two steps ahead in registers, one barrier per step (LDS double-buffered).  The tile goes to gw with f32 atomics
// (order not fixed, as in every split-K weight gradient of the library).  Grid = tiles x ranges in XCD-banded range-major order: the
// tiles of one range read the same pixels and meet in one L2.
// ---------------------------------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ int pw_chunk_off(int r, int c) { return (r * 8 + (c ^ (((r >> 1) & 1) << 2))) * 16; }

__global__ __launch_bounds__(256, 2) void conv_b3_pw_wgrad_kernel(float* __restrict__ gw, const float* __restrict__ gy,
                                                                  const float* __restrict__ x, ideas_conv_params p, unsigned M,
                                                                  int tiles_ci, int tiles, int ranges, unsigned rows_per_range) {
    constexpr int STEP = 32;                             // pixels per step
    constexpr int OPB = STEP * 128;                      // bytes of one operand buffer of one plane
    constexpr int PLANE = 2 * OPB;                       // two buffers
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 3 * PLANE];
    unsigned char* const sG = smem;                      // [3 planes][2 buffers][32 rows][128 B]
    unsigned char* const sX = smem + 3 * PLANE;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int tile, range;
    splitk_xcd_map(blockIdx.x, tiles, ranges, tile, range);
    const int o0 = (tile / tiles_ci) * 64, c0 = (tile % tiles_ci) * 64;
    const unsigned m0 = (unsigned)range * rows_per_range;
    const unsigned m1 = m0 + rows_per_range < M ? m0 + rows_per_range : M;
    if (m0 >= m1) return;
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)gy, 0, (int)(M * (unsigned)p.Cout * 4u), (int)RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)(M * (unsigned)p.Cin * 4u), (int)RSRC_FLAGS);

    // staging: thread = (pixel t >> 4 [+ 16], channel quad t & 15)
    const int quad = t & 15, px = t >> 4;
    const unsigned g_cb = (unsigned)(o0 + quad * 4) * 4u, x_cb = (unsigned)(c0 + quad * 4) * 4u;
    const unsigned gstride = (unsigned)p.Cout * 4u, xstride = (unsigned)p.Cin * 4u;
    struct Stage { float4 g[2], xv[2]; };
    auto gload = [&](Stage& st, unsigned m) {            // pixels m + px, m + px + 16 (past the range: zeros)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const unsigned r = m + (unsigned)(px + 16 * k);
            const unsigned bad = r < m1 ? 0u : 0xffffffffu;
            st.g[k] = buffer_load4(rg, (r * gstride + g_cb) | bad, 0);
            st.xv[k] = buffer_load4(rx, (r * xstride + x_cb) | bad, 0);
        }
    };
    int lds_row[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) lds_row[k] = pw_chunk_off(px + 16 * k, quad >> 1) + (quad & 1) * 8;
    auto lstore = [&](const Stage& st, int buf) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const Split4 sg = split4(st.g[k]), sx = split4(st.xv[k]);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                *reinterpret_cast<uint2*>(sG + pl * PLANE + buf * OPB + lds_row[k]) = sg.p[pl];
                *reinterpret_cast<uint2*>(sX + pl * PLANE + buf * OPB + lds_row[k]) = sx.p[pl];
            }
        }
    };

    // MFMA side: wave = (o half, ci half) of the tile; fragments by transpose reads (conv_b3_wgrad3.hip: lane_base / frag)
    const int wo = wave >> 1, wc = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int g_q = lane & 15, g_row = g_q >> 2, g_piece = g_q & 3, g_cblk = (lane >> 4) & 1;
    auto lane_base = [&](unsigned region, int cb) {      // rows 8 lh + g_row (+ 0, 4 by the two reads; + 16 for the second half step)
        const int r = 8 * lh + g_row;
        return region + (unsigned)(r * 128 + (((cb * 2 + (g_piece >> 1)) ^ (((g_row >> 1) & 1) << 2)) << 4) + (g_piece & 1) * 8);
    };
    const unsigned ldsG = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)sG;
    const unsigned ldsX = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)sX;
    const unsigned baseG = lane_base(ldsG, wo * 2 + g_cblk), baseX = lane_base(ldsX, wc * 2 + g_cblk);
    auto tr_ld = [&](unsigned addr) -> s16x4 {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)addr);
    };
    auto frag = [&](unsigned base, int off) -> bf16x8 {  // 8 pixels (rows r0 + 8 lh .. + 7; r0 a multiple of 16: the swizzle bit is g_row's)
        const s16x4 a = tr_ld(base + (unsigned)off);
        const s16x4 b = tr_ld(base + (unsigned)off + 512u);
        const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        return __builtin_bit_cast(bf16x8, v);
    };
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    auto compute = [&](int buf) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {           // two 16-pixel K-steps
            bf16x8 fa[3], fb[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                fa[pl] = frag(baseG, pl * PLANE + buf * OPB + half * 16 * 128);
                fb[pl] = frag(baseX, pl * PLANE + buf * OPB + half * 16 * 128);
            }
#pragma unroll
            for (int q = 0; q < 6; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[q]], fb[PB[q]], acc, 0, 0, 0);
        }
    };

    // pipeline: loads of step n + 2 in flight, step n + 1 split into LDS[(n+1)&1], step n multiplied from LDS[n&1]
    const unsigned nsteps = (m1 - m0 + STEP - 1) / STEP;
    Stage st0, st1;
    gload(st0, m0);
    gload(st1, m0 + STEP);
    lstore(st0, 0);
    __syncthreads();
    unsigned n = 0;
    for (; n + 1 < nsteps; n += 2) {
        gload(st0, m0 + (n + 2) * STEP);
        lstore(st1, 1);
        compute(0);
        __syncthreads();
        gload(st1, m0 + (n + 3) * STEP);
        lstore(st0, 0);
        compute(1);
        __syncthreads();
    }
    if (n < nsteps) compute(0);

    // D rows = o: (e & 3) + 8 (e >> 2) + 4 lh, column = ci: li; gw is [Cout][Cin] (OHWI with one tap)
    const int ci = c0 + wc * 32 + li;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int o = o0 + wo * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        atomicAdd(&gw[(int64_t)o * p.Cin + ci], acc[e] * p.gain);
    }
}

}  // namespace

// 1 when ideas_b3_pw_fwd takes the launch: a single-tap, unit-stride, unpadded geometry whose output grid IS the input grid, without
// per-sample scales and without accumulation, Cin % 16 == 0 up to 128 or Cin % 128 == 0 up to 1024, tensors below 4 GB.
// IDEAS_B3_PW=0: never (A/B).
int ideas_b3_pw_ok(const ideas_conv_params* p, const float* in_scale, const float* out_scale, const void* resid) {
    const char* e = getenv("IDEAS_B3_PW");
    if (e && e[0] == '0') return 0;
    if (in_scale || out_scale || p->accumulate || p->reflect) return 0;
    if (p->TY != 1 || p->TX != 1 || p->sy != 1 || p->sx != 1 || p->offy != 0 || p->offx != 0) return 0;
    if (p->osy != 1 || p->osx != 1 || p->ooy != 0 || p->oox != 0) return 0;
    if (p->OH != p->IH || p->OW != p->IW || p->YH != p->IH || p->YW != p->IW) return 0;
    if (p->Cin % 16 || p->Cin < 16 || p->Cout < 16) return 0;
    // more channels: chunks of 128 (conv_b3_pwk_kernel) -- where it wins.  Same box, tools/ab_pw.py: with the residual epilogue (the
    // merge add of a skip branch, the gradient sum of a forked block input) 256 -> 512 @64x64 0.418 -> 0.271 ms, B96 @32x32 0.316 ->
    // 0.203; without it the generic kernel is 7-9 % ahead on 256 -> 128, 512 -> 256 and 512 -> 512 (compute-heavier: its 128 x 128
    // tile shares the staged pixels between more MFMAs), so those stay where they were.
    if (p->Cin > 128 && (p->Cin % 128 || p->Cin > 1024 || !resid)) return 0;
    const int64_t M = (int64_t)p->B * p->OH * p->OW;
    // (+ one tile of slack on the x AND y offsets: the prefetch and the epilogue of the last, partial tile compute offsets past M rows,
    // which must not wrap around 2^32 into the start of the tensor)
    return (M + 128) * p->Cin * 4 < 0xffffffffLL && (M + 128) * p->Cout * 4 < 0xffffffffLL && (int64_t)p->Cin * p->Cout * 6 < 0xffffffffLL;
}

int ideas_b3_pw_fwd(void* y, const void* x, const void* wplanes, const float* bias, const void* resid, const ideas_conv_params* p,
                    hipStream_t stream) {
    if (p->Cin > 128) {
        const int64_t M = (int64_t)p->B * p->OH * p->OW;
        const int64_t ntiles = ideas_cdiv(M, 64);
        const unsigned plane_bytes = (unsigned)((int64_t)p->Cin * p->Cout * 2);
        const unsigned ngrp = (unsigned)ideas_cdiv(p->Cout, 256);
        const unsigned gx = (unsigned)(ntiles < 512 / ngrp ? ntiles : (512 / ngrp > 0 ? 512 / ngrp : 1));
        if (p->Cout <= 64)
            hipLaunchKernelGGL(conv_b3_pwk_kernel<2>, dim3(gx, ngrp), dim3(256), 0, stream, (float*)y, (const float*)x, wplanes, bias,
                               (const float*)resid, *p, (unsigned)M, plane_bytes, (int)ntiles, p->Cin / 128);
        else
            hipLaunchKernelGGL(conv_b3_pwk_kernel<4>, dim3(gx, ngrp), dim3(256), 0, stream, (float*)y, (const float*)x, wplanes, bias,
                               (const float*)resid, *p, (unsigned)M, plane_bytes, (int)ntiles, p->Cin / 128);
        return ideas_launch_status();
    }
    switch (p->Cin / 16) {
        case 1: return launch_pw<1, 4>(y, x, wplanes, bias, resid, p, stream);
        case 2: return launch_pw<2, 4>(y, x, wplanes, bias, resid, p, stream);
        case 3: return launch_pw<3, 4>(y, x, wplanes, bias, resid, p, stream);
        case 4: return launch_pw<4, 4>(y, x, wplanes, bias, resid, p, stream);
        case 5: return launch_pw<5, 2>(y, x, wplanes, bias, resid, p, stream);
        case 6: return launch_pw<6, 2>(y, x, wplanes, bias, resid, p, stream);
        case 7: return launch_pw<7, 2>(y, x, wplanes, bias, resid, p, stream);
        default: return launch_pw<8, 2>(y, x, wplanes, bias, resid, p, stream);
    }
}

// Weight gradient of the same flat geometry (no per-sample scales), Cin % 64 == 0 and Cout % 64 == 0.  IDEAS_B3_PW_WGRAD=0: never.
int ideas_b3_pw_wgrad_ok(const ideas_conv_params* p, const float* in_scale, const float* out_scale) {
    const char* e = getenv("IDEAS_B3_PW_WGRAD");
    if (e && e[0] == '0') return 0;
    if (in_scale || out_scale || p->reflect) return 0;
    if (p->TY != 1 || p->TX != 1 || p->sy != 1 || p->sx != 1 || p->offy != 0 || p->offx != 0) return 0;
    if (p->osy != 1 || p->osx != 1 || p->ooy != 0 || p->oox != 0) return 0;
    if (p->OH != p->IH || p->OW != p->IW || p->YH != p->IH || p->YW != p->IW) return 0;
    if (p->Cin % 64 || p->Cout % 64) return 0;
    const int64_t M = (int64_t)p->B * p->OH * p->OW;
    return M >= 512 && (M + 64) * p->Cin * 4 < 0xffffffffLL && (M + 64) * p->Cout * 4 < 0xffffffffLL;
}

int ideas_b3_pw_wgrad(float* gw, const void* gy, const void* x, const ideas_conv_params* p, hipStream_t stream) {
    const int64_t M = (int64_t)p->B * p->OH * p->OW;
    const int tiles_ci = p->Cin / 64, tiles = (p->Cout / 64) * tiles_ci;
    // about four rounds of the 512 resident blocks, >= 256 pixels per range (multiples of the 32-pixel step)
    int64_t ranges = 2048 / tiles;
    if (ranges < 1) ranges = 1;
    if (M / ranges < 256) ranges = M / 256 > 0 ? M / 256 : 1;
    int64_t per = ideas_cdiv(ideas_cdiv(M, ranges), 32) * 32;
    ranges = ideas_cdiv(M, per);
    if ((int64_t)tiles * ranges > 0x7fffffffLL) return IDEAS_E_SHAPE;
    hipLaunchKernelGGL(conv_b3_pw_wgrad_kernel, dim3(splitk_grid(tiles, ranges)), dim3(256), 0, stream, gw, (const float*)gy,
                       (const float*)x, *p, (unsigned)M, tiles_ci, tiles, (int)ranges, (unsigned)per);
    return ideas_launch_status();
}
