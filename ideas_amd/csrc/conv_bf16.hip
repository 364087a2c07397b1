// bf16 mixed-precision convolution family for gfx950 (BASELINE.json configs[4]): bf16 activations in HBM (NHWC), bf16
// operands on v_mfma_f32_32x32x16_bf16 (ONE product per MFMA — no split), f32 accumulation, f32 epilogue, f32 master
// weights and f32 weight gradients.  Same GEMM view, parameterisation (ideas_conv_params) and epilogue as
// conv_igemm.hip / conv_b3.hip; what changes is how the operands travel:
//
//   forward family (forward convs, input gradients, the parity phases of transposed convs)
//     * K-step = 32 channels of one tap (two MFMA K-slices); LDS rows are 64 bytes, 16-byte chunk c of row r lives at
//       position c ^ ((r >> 2) & 3): the ds_read_b128 operand fetches (four 16-lane groups) are conflict-free;
//     * activations go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds): no staging registers, no ds_write.  The DMA
//       writes lane l's 16 bytes at base + 16 l, so the swizzle and the im2col gather both live in the per-lane SOURCE
//       offset; a padding tap gets offset 0xffffffff, for which the DMA writes zeros (probed: tools/probes/dma.hip);
//     * weights are packed once per optimiser step (ideas_bf16_pack_weights) into [K/32][Cout][32] bf16 with that swizzle
//       already applied, so a B tile is one contiguous block that the DMA copies linearly;
//     * modulated convs (per-sample input scale s[b, ci]): the pack kernel writes one bf16 pack per sample, w * s[b, :] -- the
//       reference's per-sample weights (stylegan2/model.py:240-248), but bf16, swizzled and alive for one launch only (at most
//       151 MB for a 512x512x3x3 layer at B = 32, 0.06 ms of HBM time next to a 0.9 ms convolution); M tiles are cut per image
//       (the last tile of an image is partial) and a block DMAs the B tiles of ITS image, so both operand paths stay pure DMA;
//     * three LDS stages and one raw s_barrier per K-step with a COUNTED s_waitcnt vmcnt: a DMA has two K-steps to land;
//     * the MFMA runs with the weights as its A operand: a lane then owns ONE pixel and four consecutive output channels per
//       accumulator quad, i.e. the epilogue packs 4 bf16 and issues 8-byte stores (4x fewer store instructions than a
//       channel-per-lane layout with 2-byte stores).
//
//   weight gradient  gw[o][k] += gain * sum_p G(p, o) X(p, k)
//     * both operands are pixel-major in HBM; the MFMA wants 8 consecutive PIXELS of one channel per lane.  The tiles are
//       DMA'd as they lie ([32 pixels][channels]) and transposed by the LDS itself: ds_read_b64_tr_b16 hands lane l the four
//       rows of column l & 15 of a 4 x 16 block whose 8-byte pieces the 16 lanes of its group point at (probed:
//       tools/probes/tr.hip).  16-byte chunks are XOR-swizzled across the four pixel rows of a block so that the 32 lanes of a
//       half-wave hit 32 different bank pairs;
//     * split-K (XCD-banded, common.hpp: splitk_xcd_map) with f32 atomics into the (flat-bucket) f32 gradient; for modulated convs every split lies
//       inside one image and the per-sample scales d[b,o] * s[b,ci] are applied to the accumulator in the epilogue.
#include "common.hpp"
#include <type_traits>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;      // storage type of a bf16 element in HBM

namespace {

constexpr int KB = 32;          // bf16 K depth of one pipeline step (two MFMA K-slices)
constexpr int ROW = 64;         // bytes per LDS row of the forward kernel
constexpr unsigned RSRC = 0x00020000u;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    const f32x2v v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v));
}
__device__ __forceinline__ float bf_lo(unsigned pk) { return __builtin_bit_cast(float, pk << 16); }
__device__ __forceinline__ float bf_hi(unsigned pk) { return __builtin_bit_cast(float, pk & 0xffff0000u); }
__device__ __forceinline__ uint4 bload4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

// ---------------------------------------------------------------------------------------------------------------
// weights: f32 [Cout][K] (K = (ty, tx, ci) contiguous) -> bf16 [K/32][Cout][32], K-step = (ci/32, ty, tx), 16-byte chunk c of
// row n stored at position c ^ ((n >> 2) & 3).  With `scale` (float [B][Cin]): B packs, pack b holds w[n][k] * scale[b][ci(k)]
// -- the reference's per-sample modulated weights (stylegan2/model.py:240-248), as bf16 and only for the life of one launch.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_weights_kernel(uint4* __restrict__ dst, const float4* __restrict__ w,
                                                           const float* __restrict__ scale, int Cout, int K, int Cin) {
    const int64_t n8 = (int64_t)Cout * (K / 8);
    const int ntaps = K / Cin;
    const int b = blockIdx.y;
    const float* sb = scale ? scale + (int64_t)b * Cin : nullptr;
    uint4* out = dst + (int64_t)b * n8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const int k = (int)(i % (K / 8)) * 8;
        const int n = (int)(i / (K / 8));
        const int tap = k / Cin, ci = k - tap * Cin;
        const int step = (ci >> 5) * ntaps + tap;
        const int c = (ci & 31) >> 3;
        float4 u = w[2 * i], v = w[2 * i + 1];
        if (sb) {
            const float4 s0 = *reinterpret_cast<const float4*>(sb + ci), s1 = *reinterpret_cast<const float4*>(sb + ci + 4);
            u = make_float4(u.x * s0.x, u.y * s0.y, u.z * s0.z, u.w * s0.w);
            v = make_float4(v.x * s1.x, v.y * s1.y, v.z * s1.z, v.w * s1.w);
        }
        out[((int64_t)step * Cout + n) * 4 + (c ^ ((n >> 2) & 3))] =
            make_uint4(pk_bf16(u.x, u.y), pk_bf16(u.z, u.w), pk_bf16(v.x, v.y), pk_bf16(v.z, v.w));
    }
}

// The same pack read straight from the PARAMETER: element (n, ty, tx, ci) of the launch's weight matrix lies at
// w[n*sn + ty*sty + tx*stx + ci*sc] (f32 elements) -- the forward matrix of an [O,I,KH,KW] tensor of any strides, the
// phase-sliced transposed matrix of an input gradient, or the matrix of a transposed conv -- so no permuted / sliced f32 copy is
// materialised first (one launch instead of copy + pack, and 2-4 launches fewer per conv and step than torch's slicing).
template <bool UNIT>
__device__ __forceinline__ void pack_weights_strided_body(uint4* __restrict__ dst, const float* __restrict__ w,
                                                          const float* __restrict__ scale, int Cout, int TY, int TX, int Cin, int64_t sn,
                                                          int64_t sty, int64_t stx, int64_t sc, int b, int64_t bid, int64_t nblk) {
    const int c8 = Cin / 8, ntaps = TY * TX;
    const int64_t n8 = (int64_t)Cout * ntaps * c8;
    const float* sb = scale ? scale + (int64_t)b * Cin : nullptr;
    uint4* out = dst + (int64_t)b * n8;
    for (int64_t i = bid * 256 + threadIdx.x; i < n8; i += nblk * 256) {
        // UNIT (sc == 1): consecutive threads read consecutive 32-byte pieces of one (n, tap) row; otherwise consecutive
        // threads take consecutive n (the contiguous index of a transposed read) of one (tap, 8-channel chunk)
        int n, tap, ci;
        if (UNIT) {
            ci = (int)(i % c8) * 8;
            tap = (int)((i / c8) % ntaps);
            n = (int)(i / ((int64_t)c8 * ntaps));
        } else {
            n = (int)(i % Cout);
            ci = (int)((i / Cout) % c8) * 8;
            tap = (int)(i / ((int64_t)Cout * c8));
        }
        const int ty = tap / TX, tx = tap - ty * TX;
        const float* src = w + n * sn + ty * sty + tx * stx + ci * sc;
        float v[8];
        if (UNIT) {
            const float4 u0 = *reinterpret_cast<const float4*>(src), u1 = *reinterpret_cast<const float4*>(src + 4);
            v[0] = u0.x; v[1] = u0.y; v[2] = u0.z; v[3] = u0.w; v[4] = u1.x; v[5] = u1.y; v[6] = u1.z; v[7] = u1.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = src[j * sc];
        }
        if (sb) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= sb[ci + j];
        }
        const int step = (ci >> 5) * ntaps + tap;
        const int c = (ci & 31) >> 3;
        out[((int64_t)step * Cout + n) * 4 + (c ^ ((n >> 2) & 3))] =
            make_uint4(pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7]));
    }
}

template <bool UNIT>
__global__ __launch_bounds__(256) void pack_weights_strided_kernel(uint4* __restrict__ dst, const float* __restrict__ w,
                                                                   const float* __restrict__ scale, int Cout, int TY, int TX, int Cin,
                                                                   int64_t sn, int64_t sty, int64_t stx, int64_t sc) {
    pack_weights_strided_body<UNIT>(dst, w, scale, Cout, TY, TX, Cin, sn, sty, stx, sc, blockIdx.y, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(256) void pack_weights_batched_kernel(const ideas_prep_desc* __restrict__ tbl, int n) {
    int local, nblk;
    const ideas_prep_desc* d = prep_lookup(tbl, n, local, nblk);
    if (d->unit) pack_weights_strided_body<true>((uint4*)d->dst, d->w, nullptr, d->a[0], d->a[1], d->a[2], d->a[3], d->s[0], d->s[1], d->s[2], d->s[3], 0, local, nblk);
    else pack_weights_strided_body<false>((uint4*)d->dst, d->w, nullptr, d->a[0], d->a[1], d->a[2], d->a[3], d->s[0], d->s[1], d->s[2], d->s[3], 0, local, nblk);
}

// ---------------------------------------------------------------------------------------------------------------
// forward family
// ---------------------------------------------------------------------------------------------------------------
// Pipeline: NST LDS stages, ONE raw barrier per K-step.  Step t:  wait until this wave's DMA pieces of tile t have landed
// (counted vmcnt: the pieces of tiles t+1 .. t+NST-2 stay in flight), barrier (=> every wave's pieces of tile t are in LDS
// and every wave is done reading tile t-1), issue the DMA of tile t+NST-1 into the stage tile t-1 occupied, then fragments +
// MFMAs of tile t.  A DMA has NST-1 steps to land instead of one.  Every wave issues exactly DMA_PER pieces per tile, so the
// count is a compile-time constant.
template <int N> __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N <= 63, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// The wait in front of a K-step's raw barrier: this wave's DMA pieces of the step have landed (counted vmcnt) AND its LDS reads of
// the previous step have returned.  The second half matters: hipcc floats `s_barrier` (and a bare vmcnt wait) up over the previous
// step's last MFMAs and over the `s_waitcnt lgkmcnt` in front of them, so a wave passed the barrier with ds_reads of stage t-1 still
// queued -- and behind the barrier the other waves' DMA overwrites exactly that stage.  Alone the reads always won that race; next
// to a kernel that keeps the LDS busy (the f32 split weight gradient on the side stream) the image kernel returned a few output
// channels off by one K-step in 20 % of its launches (tools/probes/img_vs_b3wgrad.py; cdna_hip_programming.md: "raw s_barrier +
// lgkmcnt(0)").
template <int N> __device__ __forceinline__ void wait_step() {
    static_assert(N >= 0 && N <= 63, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}

// Register-only widening of the epilogue stores.  In the accumulator layout lane (li, lh) holds, per 32-channel block, the four
// channel quads 8g + 4lh .. +3 (g = 0..3) of pixel li: 8-byte stores, 64 pieces per instruction.  v_permlane32_swap exchanges
// lanes li and li + 32 (the two halves of the SAME pixel): swapping quad g of the upper half with quad g + 2 of the lower half
// leaves lane (li, 0) with channels 0-15 and lane (li, 1) with channels 16-31 of the block -- two 16-byte stores per lane, 32
// contiguous bytes, half the store instructions, no LDS.  out[c] = channels 8 (c + 2 lh) .. + 7 of the block.  With the stores compiled
// out the 128-channel image kernel runs 1030 instead of 750 TFLOP/s at 256x256; this form recovers 820 (Dreal.1.conv1 710 -> 780,
// E.2.conv1 575 -> 840).  (The tile through LDS as whole pixel rows, 128-byte lines per pixel, was the same speed.)
__device__ __forceinline__ void quad_exchange(const uint2 (&q)[4], uint4 (&out)[2]) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const auto rx = __builtin_amdgcn_permlane32_swap(q[g].x, q[g + 2].x, false, false);
        const auto ry = __builtin_amdgcn_permlane32_swap(q[g].y, q[g + 2].y, false, false);
        out[g] = make_uint4(rx[0], ry[0], rx[1], ry[1]);
    }
}

template <int WM, int WN, int MT, int NT, int NST, bool PERIMG, bool REFLECT>
__device__ __forceinline__ void conv_bf16_body(bf16_t* __restrict__ y, const bf16_t* __restrict__ x, const void* __restrict__ wpack,
                                               const float* __restrict__ out_scale, const float* __restrict__ bias,
                                               const bf16_t* __restrict__ resid, const ideas_conv_params& p, int tile_m, int tile_n,
                                               int tiles_per_img, unsigned x_bytes, unsigned w_bytes) {
    constexpr int NW = WM * WN;           // waves per block (4 or 8)
    constexpr int BM = WM * MT * 32;      // pixels of the tile
    constexpr int BN = WN * NT * 32;      // output channels of the tile
    constexpr int A_PIECES = BM / 16, B_PIECES = BN / 16;     // 16-row (1 KiB) DMA pieces per tile
    constexpr int A_PER = (A_PIECES + NW - 1) / NW;           // per wave; a wave without a piece of its own repeats one
    constexpr int B_PER = (B_PIECES + NW - 1) / NW;
    constexpr int DMA_PER = A_PER + B_PER;
    constexpr int BUF = (BM + BN) * ROW;
    constexpr int SMEM = NST * BUF > BM * 12 ? NST * BUF : BM * 12;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t M = (int64_t)p.B * p.OH * p.OW;
    const int n0 = tile_n * BN;
    // row r of the tile -> output point.  Plain: consecutive points of the flattened (b, oy, ox) grid.  PERIMG (per-sample
    // weights): tiles are cut per image (tile_m = img * tiles_per_img + j), rows past the image's last point are clamped and
    // not stored, and the weight pack of image `img` is used.
    const int OHW = p.OH * p.OW;
    const int img = PERIMG ? tile_m / tiles_per_img : 0;
    const int64_t m0 = PERIMG ? (int64_t)img * OHW + (int64_t)(tile_m - img * tiles_per_img) * BM : (int64_t)tile_m * BM;
    const int64_t mend = PERIMG ? (int64_t)(img + 1) * OHW : M;     // first point this tile must not touch
    const int K = p.TY * p.TX * p.Cin;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)x_bytes, (int)RSRC);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)wpack, 0, (int)w_bytes, (int)RSRC);
    const unsigned w_img = PERIMG ? (unsigned)img * (unsigned)K * (unsigned)p.Cout * 2u : 0u;   // this image's pack
    const unsigned w_end = w_img + (unsigned)K * (unsigned)p.Cout * 2u;

    // ---- A: this lane's DMA slots.  Piece q (16 rows) is issued by wave q & 3; lane l fills LDS slot l of the piece:
    // row = 16 q + (l >> 2), position l & 3, which must hold chunk c = (l & 3) ^ ((row >> 2) & 3) of that row.
    unsigned a_base[A_PER], a_inv[A_PER];
    int a_iyb[A_PER], a_ixb[A_PER], a_img[A_PER], a_c8[A_PER];
#pragma unroll
    for (int j = 0; j < A_PER; ++j) {
        const int r = ((wave + NW * j) % A_PIECES) * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((r >> 2) & 3);
        int64_t m = m0 + r;
        m = m < mend ? m : mend - 1;               // rows past the end repeat the last row; their results are not stored
        const int ox = (int)(m % p.OW);
        const int64_t q = m / p.OW;
        const int oy = (int)(q % p.OH);
        const int b = (int)(q / p.OH);
        const int iyb = oy * p.sy + p.offy, ixb = ox * p.sx + p.offx;
        a_base[j] = (unsigned)(((b * p.IH + iyb) * p.IW + ixb) * p.Cin + c * 8) * 2u;   // mod 2^32; valid taps land < x_bytes
        a_iyb[j] = iyb; a_ixb[j] = ixb; a_img[j] = b * p.IH; a_c8[j] = c * 8;
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

    int k_ci = 0, k_tx = 0, k_ty = 0, k_tap = 0;   // block-uniform walk over K = (ci/32, ty, tx)
    int k_next = 0;                                // index of the next tile to fetch
    // Fetch the next tile (k_next) into stage `st`.  Tiles past K: the activation pieces are fetched from wherever the walk
    // points (in range or zero-filled), the weight pieces get an out-of-range offset (zeros) -- never multiplied.
    auto fetch = [&](int st) {
        unsigned char* base = smem + st * BUF;
        const unsigned tapoff = (unsigned)(((k_ty * p.dy) * p.IW + k_tx * p.dx) * p.Cin + k_ci) * 2u;
#pragma unroll
        for (int j = 0; j < A_PER; ++j) {
            unsigned off;
            if (REFLECT) {
                const int iy = reflect_coord(a_iyb[j] + k_ty * p.dy, p.IH), ix = reflect_coord(a_ixb[j] + k_tx * p.dx, p.IW);
                off = (unsigned)(((a_img[j] + iy) * p.IW + ix) * p.Cin + k_ci + a_c8[j]) * 2u;
            } else {
                off = (a_base[j] + tapoff) | (unsigned)__builtin_amdgcn_sbfe(a_inv[j], k_tap, 1);
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(base + ((wave + NW * j) % A_PIECES) * 1024), 16, (int)off, 0, 0, 0);
        }
        // (the K-step offset is folded into the per-lane offset: only that one is range-checked by a raw buffer)
        const unsigned soff = w_img + (unsigned)k_next * (unsigned)p.Cout * 64u;
#pragma unroll
        for (int j = 0; j < B_PER; ++j) {
            const int q = (wave + NW * j) % B_PIECES;       // narrow tiles: some waves repeat a piece (same bytes, same place)
            unsigned off = soff + (unsigned)((n0 + q * 16) * 64 + lane * 16);
            off = off < w_end ? off : 0xffffffffu;          // past this image's pack (last tiles, rows >= Cout of the last step)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, LDS_PTR(base + BM * ROW + q * 1024), 16, (int)off, 0, 0, 0);
        }
        ++k_next;
        const int nx = k_tx + 1, gx = 1 - (int)((unsigned)(nx - p.TX) >> 31);
        k_tx = nx - gx * p.TX;
        const int ny = k_ty + gx, gy = 1 - (int)((unsigned)(ny - p.TY) >> 31);
        k_ty = ny - gy * p.TY;
        k_tap = (k_tap + 1) * (1 - gy);
        k_ci += gy * KB;
    };

    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // fragment addresses: row (..)*32 + li, K-slice s, half lh -> chunk 2 s + lh at position chunk ^ ((row >> 2) & 3)
    const int rsw = (li >> 2) & 3;
    const int a_off = ((wm * MT) * 32 + li) * ROW;
    const int b_off = BM * ROW + ((wn * NT) * 32 + li) * ROW;
    const int pos0 = ((0 + lh) ^ rsw) * 16, pos1 = ((2 + lh) ^ rsw) * 16;

    const int nk = K / KB;
#pragma unroll
    for (int st = 0; st < NST - 1; ++st) fetch(st);               // prologue: tiles 0 .. NST-2
    int st_cur = 0, st_free = NST - 1;
    for (int kt = 0; kt < nk; ++kt) {
        wait_step<(NST - 2) * DMA_PER>();           // this wave's pieces of tile kt have landed, its reads of tile kt-1 returned
        __builtin_amdgcn_s_barrier();               // ... and everybody else's; all waves are done with tile kt-1
        fetch(st_free);                             // tile kt+NST-1 into the stage tile kt-1 occupied
        const unsigned char* base = smem + st_cur * BUF;
        bf16x8 fx[MT][2], fw[NT][2];
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            fx[a][0] = *reinterpret_cast<const bf16x8*>(base + a_off + a * 32 * ROW + pos0);
            fx[a][1] = *reinterpret_cast<const bf16x8*>(base + a_off + a * 32 * ROW + pos1);
        }
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            fw[b][0] = *reinterpret_cast<const bf16x8*>(base + b_off + b * 32 * ROW + pos0);
            fw[b][1] = *reinterpret_cast<const bf16x8*>(base + b_off + b * 32 * ROW + pos1);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b)   // weights as the A operand: rows of D = output channels, columns = pixels
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[b][s], fx[a][s], acc[a][b], 0, 0, 0);
        st_free = st_cur;
        st_cur = st_cur + 1 == NST ? 0 : st_cur + 1;
    }
    wait_vmcnt<0>();                                // drain the tiles fetched past K before the stages are reused
    __syncthreads();

    // ---- epilogue: lane = pixel li of block a; registers 4g..4g+3 = channels 8g + 4lh .. +3 of block b --------------------
    int64_t* row_off = reinterpret_cast<int64_t*>(smem);
    int* row_b = reinterpret_cast<int*>(smem + 8 * BM);
    if (t < BM) {
        const int64_t m = m0 + t;
        int64_t off = -1;
        int b = 0;
        if (m < mend) {
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
    const bool wide = (p.Cout & 7) == 0 && (((uintptr_t)y) & 15) == 0;      // 16-byte stores after a lane exchange (quad_exchange)
#pragma unroll
    for (int a = 0; a < MT; ++a) {
        const int row = (wm * MT + a) * 32 + li;
        const int64_t off = row_off[row];
        const int bimg = row_b[row];
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            uint2 q[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                q[g] = make_uint2(0u, 0u);
                const int n = n0 + (wn * NT + b) * 32 + 8 * g + 4 * lh;
                if (off < 0 || n >= p.Cout) continue;           // Cout % 4 == 0
                float v[4];
                const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 os = out_scale ? *reinterpret_cast<const float4*>(out_scale + (int64_t)bimg * p.Cout + n)
                                            : make_float4(1.f, 1.f, 1.f, 1.f);
                const float bvv[4] = {bv.x, bv.y, bv.z, bv.w}, osv[4] = {os.x, os.y, os.z, os.w};
                uint2 rr = make_uint2(0u, 0u);
                if (resid) rr = *reinterpret_cast<const uint2*>(resid + off + n);
                const float rv[4] = {bf_lo(rr.x), bf_hi(rr.x), bf_lo(rr.y), bf_hi(rr.y)};
                uint2 prev = make_uint2(0u, 0u);
                if (p.accumulate) prev = *reinterpret_cast<const uint2*>(y + off + n);
                const float pv[4] = {bf_lo(prev.x), bf_hi(prev.x), bf_lo(prev.y), bf_hi(prev.y)};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float u = acc[a][b][4 * g + j] * p.gain * osv[j] + bvv[j];
                    if (p.act) u = (u > 0.f ? u : u * p.alpha) * p.act_gain;
                    if (resid) u = (u + rv[j]) * p.resid_gain;
                    if (p.accumulate) u += pv[j];
                    v[j] = u;
                }
                q[g] = make_uint2(pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]));
                if (!wide) *reinterpret_cast<uint2*>(y + off + n) = q[g];
            }
            if (wide) {                                      // (block-uniform; every lane takes part in the exchange)
                uint4 ch[2];
                quad_exchange(q, ch);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int n = n0 + (wn * NT + b) * 32 + 8 * (c + 2 * lh);
                    if (off >= 0 && n < p.Cout) *reinterpret_cast<uint4*>(y + off + n) = ch[c];
                }
            }
        }
    }
}

template <int WM, int WN, int MT, int NT, int NST, bool PERIMG, bool REFLECT>
__global__ __launch_bounds__(64 * WM * WN) void conv_bf16_kernel(bf16_t* __restrict__ y, const bf16_t* __restrict__ x,
                                                           const void* __restrict__ wpack, const float* __restrict__ out_scale,
                                                           const float* __restrict__ bias, const bf16_t* __restrict__ resid,
                                                           ideas_conv_params p, int tiles_n, int tiles_per_img, unsigned x_bytes,
                                                           unsigned w_bytes) {
    const int swz = xcd_swizzle(blockIdx.x, gridDim.x);
    conv_bf16_body<WM, WN, MT, NT, NST, PERIMG, REFLECT>(y, x, wpack, out_scale, bias, resid, p, swz / tiles_n, swz % tiles_n, tiles_per_img,
                                                         x_bytes, w_bytes);
}

// Several launches of the family in ONE grid: the output-parity phases of a stride-2 input gradient / transposed conv (4 / 2 / 2 / 1
// taps; same x, y and scales, own geometry and weight pack each), launch-major and heaviest first, each launch's tiles in their own
// XCD-banded order -- as conv_b3_multi_kernel.  The bf16 phases are 50-150 us launches: four grid ramps and tails per layer were a
// third of their time.
struct BF16Multi {
    ideas_conv_params p[4];
    const void* w[4];
    unsigned w_bytes[4];
    int tpi[4];
    int off[5];
    int n, tiles_n;
};

template <int WM, int WN, int MT, int NT, bool PERIMG>
__global__ __launch_bounds__(64 * WM * WN) void conv_bf16_multi_kernel(bf16_t* __restrict__ y, const bf16_t* __restrict__ x,
                                                                 const float* __restrict__ out_scale, BF16Multi a, unsigned x_bytes) {
    const int bid = blockIdx.x;
    const int which = (bid >= a.off[1] && a.n > 1) + (bid >= a.off[2] && a.n > 2) + (bid >= a.off[3] && a.n > 3);
    const int swz = xcd_swizzle(bid - a.off[which], a.off[which + 1] - a.off[which]);
    conv_bf16_body<WM, WN, MT, NT, 3, PERIMG, false>(y, x, a.w[which], out_scale, nullptr, nullptr, a.p[which], swz / a.tiles_n,
                                                     swz % a.tiles_n, a.tpi[which], x_bytes, a.w_bytes[which]);
}

template <int WM, int WN, int MT, int NT>
int launch_bf16_multi_cfg(int n, void* y, const void* x, const void* const* wpack, int per_image, const float* out_scale,
                          const ideas_conv_params* ps, hipStream_t stream) {
    constexpr int BM_ = WM * MT * 32, BN_ = WN * NT * 32;
    BF16Multi a;
    a.n = n;
    a.tiles_n = (int)ideas_cdiv(ps[0].Cout, BN_);
    int64_t blocks = 0;
    for (int i = 0; i < 4; ++i) {
        const int k = i < n ? i : 0;
        a.p[i] = ps[k];
        a.w[i] = wpack[k];
        a.w_bytes[i] = (unsigned)((int64_t)(per_image ? ps[k].B : 1) * ps[k].TY * ps[k].TX * ps[k].Cin * ps[k].Cout * 2);
        a.tpi[i] = (int)ideas_cdiv((int64_t)ps[k].OH * ps[k].OW, BM_);
        a.off[i] = (int)blocks;
        if (i < n) {
            const int64_t tm = per_image ? (int64_t)ps[k].B * a.tpi[i] : ideas_cdiv((int64_t)ps[k].B * ps[k].OH * ps[k].OW, BM_);
            blocks += tm * a.tiles_n;
            if (blocks > 0x7fffffffLL) return IDEAS_E_SHAPE;
        }
    }
    a.off[n] = (int)blocks;
    a.off[4] = (int)blocks;
    const unsigned x_bytes = (unsigned)((int64_t)ps[0].B * ps[0].IH * ps[0].IW * ps[0].Cin * 2);
    if (per_image)
        hipLaunchKernelGGL((conv_bf16_multi_kernel<WM, WN, MT, NT, true>), dim3((unsigned)blocks), dim3(64 * WM * WN), 0, stream, (bf16_t*)y,
                           (const bf16_t*)x, out_scale, a, x_bytes);
    else
        hipLaunchKernelGGL((conv_bf16_multi_kernel<WM, WN, MT, NT, false>), dim3((unsigned)blocks), dim3(64 * WM * WN), 0, stream, (bf16_t*)y,
                           (const bf16_t*)x, out_scale, a, x_bytes);
    return ideas_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 layers (forward convs and their input gradients: ~60 % of the bf16 step's convolution time): the
// activation operand as an LDS IMAGE.  conv_bf16_kernel DMAs a 256-pixel A tile per (32-channel chunk, tap) step -- 16 of its
// 24 DMA pieces per step, every input pixel nine times -- and the issue cost of those pieces (60-185 cycles each next to 8 MFMAs
// of 32) is what bounds it: 0.27 of the MFMA peak.  Here a block owns a TR x TP patch of output pixels (4 x 64 or 8 x 32 = 256)
// and DMAs the (TR + 2) x (TP + 2) input pixels of a chunk ONCE; the nine taps read that image at nine pixel offsets (the im2col
// gather of a tap is just another row / column base of the ds_read).  Per chunk and wave: 9 weight pieces + 4 image pieces
// instead of 27.  Zero padding = out-of-range DMA source (the DMA writes zeros).
//   image:   pixel (r, c) at LDS pixel r * LP + c, 64 B (32 channels), 16-byte chunk k at position k ^ ((c >> 2) & 3): the 32
//            consecutive pixels an MFMA operand covers are conflict-free at every column offset (enumerated against the
//            ds_read_b128 lane groups of MI355X_MICROARCH.md); LP = TP + 4 keeps c mod 4 = pixel mod 4;
//   weights: as conv_bf16_kernel (packed [chunk * 9 + tap][Cout][32], three LDS stages, one piece per wave and step);
//   steps:   the nine taps of a chunk are unrolled: the weight stage (tap % 3), the image piece slots (even taps <= 6) and the
//            counted vmcnt of every position are compile-time constants; one raw barrier per tap as before;
//   waves:   8 = 4 pixel quarters x 2 channel halves of a 256 x 128 tile, weights as the MFMA A operand, epilogue of
//            conv_bf16_kernel (a lane owns one pixel and four consecutive channels per accumulator quad: 8-byte stores).
// Requires TY = TX = 3, unit stride, |tap step| = 1 with offset = -step (forward: +1 / -1, input gradient: -1 / +1), output =
// input size, W % 16 == 0, H % (256 / TP) == 0 (patches 4 x 64, 8 x 32 or 16 x 16), Cin % 32 == 0, Cout > 32 (N tile 128, or 64 for
// Cout <= 64); mirror padding (forward only) is a different source pixel per image pixel, nothing else.
// ---------------------------------------------------------------------------------------------------------------
template <int TP, int NT, bool PERIMG>
__global__ __launch_bounds__(512, 4) void conv_bf16_img_kernel(bf16_t* __restrict__ y, const bf16_t* __restrict__ x,
                                                               const void* __restrict__ wpack, const float* __restrict__ out_scale,
                                                               const float* __restrict__ bias, const bf16_t* __restrict__ resid,
                                                               ideas_conv_params p, int tiles_n, unsigned x_bytes, unsigned w_bytes) {
    constexpr int TR = 256 / TP;
    constexpr int LP = TP + 4;                       // LDS pixels per image row (TP + 2 in use)
    constexpr int NPIX = (TR + 2) * LP;
    constexpr int NPIECE = (NPIX + 15) / 16;         // 16-pixel (1 KiB) DMA pieces of an image
    static_assert(NPIECE <= 32, "four image slots per wave");
    constexpr int IMG = NPIECE * 1024;               // bytes of one image buffer
    constexpr int BN = 64 * NT;                      // output channels of the tile (128, or 64 for the narrow layers)
    constexpr int BST = BN * ROW;                    // one weight stage (BN output channels x 64 B)
    constexpr int SMEM = 2 * IMG + 3 * BST;
    static_assert(SMEM >= 256 * 12, "epilogue row table");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];
    unsigned char* const sB = smem + 2 * IMG;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int swz = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tile_n = swz % tiles_n, tile_m = swz / tiles_n;
    const int ppr = p.OW / TP, ppi = (p.OH / TR) * ppr;
    const int img = tile_m / ppi, prem = tile_m - img * ppi;
    const int y0 = (prem / ppr) * TR, x0 = (prem % ppr) * TP;
    const int n0 = tile_n * BN;
    const int K = 9 * p.Cin;
    const bool flip = p.dy < 0;                      // input gradient: tap ty reads row oy + 1 - ty

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)x_bytes, (int)RSRC);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)wpack, 0, (int)w_bytes, (int)RSRC);
    const unsigned w_img = PERIMG ? (unsigned)img * (unsigned)K * (unsigned)p.Cout * 2u : 0u;
    const unsigned w_end = w_img + (unsigned)K * (unsigned)p.Cout * 2u;

    // ---- image DMA slots of this wave: piece q = wave + 8 j (duplicates past NPIECE rewrite a piece with the same bytes);
    // lane l fills position l & 3 of LDS pixel 16 q + (l >> 2) = image pixel (r, c), which must hold chunk (l & 3) ^ ((c >> 2) & 3)
    const int swave = __builtin_amdgcn_readfirstlane(wave);      // (scalar: the DMA destinations stay in SGPRs)
    unsigned a_src[4], a_msk[4];
    int a_dst[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = (swave + 8 * j) % NPIECE;
        const int lp = 16 * q + (lane >> 2);
        const int r = lp / LP, c = lp - r * LP;
        const int k = (lane & 3) ^ ((c >> 2) & 3);
        int iy = y0 - 1 + r, ix = x0 - 1 + c;
        if (p.reflect) { iy = reflect_coord(iy, p.IH); ix = reflect_coord(ix, p.IW); }     // (mirror padding: every patch pixel exists)
        const bool ok = r < TR + 2 && c < TP + 2 && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
        a_src[j] = (unsigned)(((img * p.IH + (ok ? iy : 0)) * p.IW + (ok ? ix : 0)) * p.Cin + k * 8) * 2u;
        a_msk[j] = ok ? 0u : 0xffffffffu;
        a_dst[j] = q * 1024;
    }
    auto fetchA = [&](int buf, int j, int chunk) {     // image piece slot j of chunk `chunk` (past the last chunk: zeros, never read)
        const unsigned off = (a_src[j] + (unsigned)chunk * 64u) | a_msk[j] | (chunk * KB < p.Cin ? 0u : 0xffffffffu);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(smem + buf * IMG + a_dst[j]), 16, (int)off, 0, 0, 0);
    };
    int kb_next = 0;                                   // next weight step to fetch, (chunk * 9 + tap) order
    auto fetchB = [&](int st) {
        unsigned off = w_img + (unsigned)kb_next * (unsigned)p.Cout * 64u + (unsigned)((n0 + (swave % (4 * NT)) * 16) * 64 + lane * 16);
        off = off < w_end ? off : 0xffffffffu;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, LDS_PTR(sB + st * BST + (swave % (4 * NT)) * 1024), 16, (int)off, 0, 0, 0);
        ++kb_next;
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    f32x16 acc[2][NT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // operand a of this wave: 32 pixels of patch row prow(a), columns pcol(a) + li.  Fragment address for column shift tx and
    // K-slice s (byte offset inside the image, operand 0, patch row 0): (col pixel) * 64 + ((2 s + lh) ^ ((col >> 2) & 3)) * 16;
    // operand 1 is 32 columns (same swizzle class) or one image row further: a constant byte offset
    constexpr int A1_OFF = TP == 64 ? 32 * ROW : TP == 32 ? LP * ROW : 2 * LP * ROW;
    const int row_base = (TP == 64 ? wm : TP == 32 ? 2 * wm : 4 * wm + (li >> 4)) * LP * ROW;     // (16-wide: an operand = 2 rows x 16)
    int fx_off[3][2];
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) {
        const int c = (flip ? 2 - tx : tx) + (TP == 16 ? (li & 15) : li);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) fx_off[tx][s2] = row_base + c * ROW + (((2 * s2 + lh) ^ ((c >> 2) & 3)) << 4);
    }
    const int rsw = (li >> 2) & 3;
    const int b_off = (wn * 32 * NT + li) * ROW;
    const int pos0 = ((0 + lh) ^ rsw) * 16, pos1 = ((2 + lh) ^ rsw) * 16;

    // one tap: wait for this wave's pieces, barrier, issue the fetches of this position, fragments, 8 MFMAs
    auto tap_step = [&](auto tapc, int chunk) {
        constexpr int TAP = decltype(tapc)::value;
        constexpr int TY = TAP / 3, TX = TAP % 3;
        // in flight behind the weight piece of this step: the weight piece of the next step and the image pieces issued at the
        // two previous positions (even positions <= 6 carry one)
        constexpr int P1 = (TAP + 7) % 9, P2 = (TAP + 8) % 9;
        constexpr int NFLY = 1 + ((P1 % 2 == 0 && P1 <= 6) ? 1 : 0) + ((P2 % 2 == 0 && P2 <= 6) ? 1 : 0);
        wait_step<NFLY>();
        __builtin_amdgcn_s_barrier();
        fetchB((TAP + 2) % 3);                                             // weight step + 2 into the stage step - 1 used
        if (TAP % 2 == 0 && TAP <= 6) fetchA((chunk & 1) ^ 1, TAP / 2, chunk + 1);
        const unsigned char* img_base = smem + (chunk & 1) * IMG + (flip ? 2 - TY : TY) * LP * ROW;
        const unsigned char* wb = sB + (TAP % 3) * BST + b_off;
        bf16x8 fx[2][2], fw[NT][2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            fx[a][0] = *reinterpret_cast<const bf16x8*>(img_base + a * A1_OFF + fx_off[TX][0]);
            fx[a][1] = *reinterpret_cast<const bf16x8*>(img_base + a * A1_OFF + fx_off[TX][1]);
        }
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            fw[b][0] = *reinterpret_cast<const bf16x8*>(wb + b * 32 * ROW + pos0);
            fw[b][1] = *reinterpret_cast<const bf16x8*>(wb + b * 32 * ROW + pos1);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[b][s2], fx[a][s2], acc[a][b], 0, 0, 0);
    };

    // prologue: the image of chunk 0 (four slots), weight steps 0 and 1
#pragma unroll
    for (int j = 0; j < 4; ++j) fetchA(0, j, 0);
    fetchB(0);
    fetchB(1);
    const int nc = p.Cin / KB;
    for (int chunk = 0; chunk < nc; ++chunk) {
        tap_step(std::integral_constant<int, 0>{}, chunk);
        tap_step(std::integral_constant<int, 1>{}, chunk);
        tap_step(std::integral_constant<int, 2>{}, chunk);
        tap_step(std::integral_constant<int, 3>{}, chunk);
        tap_step(std::integral_constant<int, 4>{}, chunk);
        tap_step(std::integral_constant<int, 5>{}, chunk);
        tap_step(std::integral_constant<int, 6>{}, chunk);
        tap_step(std::integral_constant<int, 7>{}, chunk);
        tap_step(std::integral_constant<int, 8>{}, chunk);
    }
    wait_vmcnt<0>();
    __syncthreads();

    // ---- epilogue (conv_bf16_kernel's): tile row = (wm, a, li) -> output pixel of the patch ------------------------------------
    int64_t* row_off = reinterpret_cast<int64_t*>(smem);
    if (t < 256) {
        const int w4 = t >> 6, a = (t >> 5) & 1, l = t & 31;
        const int prow = TP == 64 ? w4 : TP == 32 ? 2 * w4 + a : 4 * w4 + 2 * a + (l >> 4);
        const int pcol = TP == 64 ? 32 * a + l : TP == 32 ? l : (l & 15);
        row_off[t] = (((int64_t)img * p.YH + y0 + prow) * p.YW + x0 + pcol) * p.Cout;
    }
    __syncthreads();
    const bool wide = (p.Cout & 7) == 0 && (((uintptr_t)y) & 15) == 0;      // 16-byte stores after a lane exchange (quad_exchange)
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int64_t off = row_off[wm * 64 + a * 32 + li];
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            uint2 q[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                q[g] = make_uint2(0u, 0u);
                const int n = n0 + (wn * NT + b) * 32 + 8 * g + 4 * lh;
                if (n >= p.Cout) continue;           // Cout % 4 == 0
                float v[4];
                const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 os = out_scale ? *reinterpret_cast<const float4*>(out_scale + (int64_t)img * p.Cout + n)
                                            : make_float4(1.f, 1.f, 1.f, 1.f);
                const float bvv[4] = {bv.x, bv.y, bv.z, bv.w}, osv[4] = {os.x, os.y, os.z, os.w};
                uint2 rr = make_uint2(0u, 0u);
                if (resid) rr = *reinterpret_cast<const uint2*>(resid + off + n);
                const float rv[4] = {bf_lo(rr.x), bf_hi(rr.x), bf_lo(rr.y), bf_hi(rr.y)};
                uint2 prev = make_uint2(0u, 0u);
                if (p.accumulate) prev = *reinterpret_cast<const uint2*>(y + off + n);
                const float pv[4] = {bf_lo(prev.x), bf_hi(prev.x), bf_lo(prev.y), bf_hi(prev.y)};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float u = acc[a][b][4 * g + j] * p.gain * osv[j] + bvv[j];
                    if (p.act) u = (u > 0.f ? u : u * p.alpha) * p.act_gain;
                    if (resid) u = (u + rv[j]) * p.resid_gain;
                    if (p.accumulate) u += pv[j];
                    v[j] = u;
                }
                q[g] = make_uint2(pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]));
                if (!wide) *reinterpret_cast<uint2*>(y + off + n) = q[g];
            }
            if (wide) {                                      // (block-uniform; every lane takes part in the exchange)
                uint4 ch[2];
                quad_exchange(q, ch);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int n = n0 + (wn * NT + b) * 32 + 8 * (c + 2 * lh);
                    if (n < p.Cout) *reinterpret_cast<uint4*>(y + off + n) = ch[c];
                }
            }
        }
    }
}

static int bf16_img_tp(const ideas_conv_params* p) {      // patch width of the image kernel for this geometry, 0 = not its geometry
    const char* e = getenv("IDEAS_BF16_IMG");             // (read per call: the tests toggle it in-process)
    if (e && e[0] == '0') return 0;
    if (p->TY != 3 || p->TX != 3 || p->sy != 1 || p->sx != 1 || p->osy != 1 || p->osx != 1 || p->ooy || p->oox) return 0;
    if (!((p->dy == 1 && p->offy == -1) || (p->dy == -1 && p->offy == 1)) || p->dx != p->dy || p->offx != p->offy) return 0;
    if (p->OH != p->IH || p->OW != p->IW || p->YH != p->OH || p->YW != p->OW) return 0;
    if (p->reflect && (p->dy != 1 || p->IH < 2 || p->IW < 2)) return 0;
    if (p->Cin % 32 || p->Cout <= 32 || p->OW % 16) return 0;
    const int tp = p->OW % 64 == 0 ? 64 : p->OW % 32 == 0 ? 32 : 16;
    return p->OH % (256 / tp) == 0 ? tp : 0;
}

template <int TP, int NT>
int launch_bf16_img(void* y, const void* x, const void* wpack, int per_image, const float* out_scale, const float* bias, const void* resid,
                    const ideas_conv_params* p, hipStream_t stream) {
    const int64_t tm = (int64_t)p->B * (p->OH / (256 / TP)) * (p->OW / TP);
    const int tn = (p->Cout + 64 * NT - 1) / (64 * NT);
    if (tm * tn > 0x7fffffffLL) return IDEAS_E_SHAPE;
    const unsigned x_bytes = (unsigned)((int64_t)p->B * p->IH * p->IW * p->Cin * 2);
    const unsigned w_bytes = (unsigned)((int64_t)(per_image ? p->B : 1) * 9 * p->Cin * p->Cout * 2);
    if (per_image)
        hipLaunchKernelGGL((conv_bf16_img_kernel<TP, NT, true>), dim3((unsigned)(tm * tn)), dim3(512), 0, stream, (bf16_t*)y, (const bf16_t*)x,
                           wpack, out_scale, bias, (const bf16_t*)resid, *p, tn, x_bytes, w_bytes);
    else
        hipLaunchKernelGGL((conv_bf16_img_kernel<TP, NT, false>), dim3((unsigned)(tm * tn)), dim3(512), 0, stream, (bf16_t*)y, (const bf16_t*)x,
                           wpack, out_scale, bias, (const bf16_t*)resid, *p, tn, x_bytes, w_bytes);
    return ideas_launch_status();
}

template <int NT>
int launch_bf16_img_tp(int tp, void* y, const void* x, const void* wpack, int per_image, const float* out_scale, const float* bias,
                       const void* resid, const ideas_conv_params* p, hipStream_t stream) {
    if (tp == 64) return launch_bf16_img<64, NT>(y, x, wpack, per_image, out_scale, bias, resid, p, stream);
    if (tp == 32) return launch_bf16_img<32, NT>(y, x, wpack, per_image, out_scale, bias, resid, p, stream);
    return launch_bf16_img<16, NT>(y, x, wpack, per_image, out_scale, bias, resid, p, stream);
}

template <int WM, int WN, int MT, int NT, int NST = 3>
int launch_bf16_cfg(void* y, const void* x, const void* wpack, int per_image, const float* out_scale, const float* bias,
                    const void* resid, const ideas_conv_params* p, hipStream_t stream) {
    constexpr int BM_ = WM * MT * 32, BN_ = WN * NT * 32;
    const int64_t M = (int64_t)p->B * p->OH * p->OW;
    const int tpi = (int)ideas_cdiv((int64_t)p->OH * p->OW, BM_);
    const int64_t tm = per_image ? (int64_t)p->B * tpi : ideas_cdiv(M, BM_);
    const int tn = (int)ideas_cdiv(p->Cout, BN_);
    if (tm * tn > 0x7fffffffLL) return IDEAS_E_SHAPE;
    const unsigned x_bytes = (unsigned)((int64_t)p->B * p->IH * p->IW * p->Cin * 2);
    const unsigned w_bytes = (unsigned)((int64_t)(per_image ? p->B : 1) * p->TY * p->TX * p->Cin * p->Cout * 2);
    auto go = [&](auto pi, auto rf) {
        hipLaunchKernelGGL((conv_bf16_kernel<WM, WN, MT, NT, NST, decltype(pi)::value, decltype(rf)::value>), dim3((unsigned)(tm * tn)),
                           dim3(64 * WM * WN), 0, stream, (bf16_t*)y, (const bf16_t*)x, wpack, out_scale, bias, (const bf16_t*)resid, *p, tn,
                           tpi, x_bytes, w_bytes);
    };
    using T = std::true_type;
    using F = std::false_type;
    if (per_image) { if (p->reflect) go(T{}, T{}); else go(T{}, F{}); }
    else { if (p->reflect) go(F{}, T{}); else go(F{}, F{}); }
    return ideas_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------------------------------------------
// LDS image of an operand tile: [32 pixels][R channels] bf16, 16-byte chunk (8 channels) c of pixel row r at chunk slot
// r * (R/8) + (c ^ sw(r)), where sw spreads the four rows of a transpose block over different bank groups.
template <int R>
__device__ __forceinline__ int chunk_slot(int r, int c) {
    constexpr int CPR = R / 8;
    if (CPR >= 16) return r * CPR + (c ^ ((r & 3) << 2));
    if (CPR == 8) return r * CPR + (c ^ (((r >> 1) & 1) << 2));
    return r * CPR + c;
}

template <int WM, int WN, int MT, int NT, bool SCALE, bool REFLECT>
__global__ __launch_bounds__(64 * WM * WN) void conv_bf16_wgrad_kernel(float* __restrict__ gw, const bf16_t* __restrict__ gy,
                                                                 const bf16_t* __restrict__ x, const float* __restrict__ in_scale,
                                                                 const float* __restrict__ out_scale, ideas_conv_params p,
                                                                 int tiles_n, int pix_per_split, int splits_per_img,
                                                                 unsigned gy_bytes, unsigned x_bytes, int tiles, int splits) {
    constexpr int NW = WM * WN;        // waves per block (4 or 8)
    constexpr int BM = WM * MT * 32;   // output channels of the tile
    constexpr int BN = WN * NT * 32;   // k columns of the tile
    constexpr int PIECES_G = 32 * BM * 2 / 1024, PIECES_X = 32 * BN * 2 / 1024;    // 1 KiB DMA pieces per K-step
    // every wave issues PER_G pieces of G then PER_X pieces of X per step (narrow tiles: some pieces twice) -- which operand a
    // slot belongs to is a compile-time property of the slot, so the buffer descriptor of each DMA is known to be uniform
    constexpr int PER_G = (PIECES_G + NW - 1) / NW, PER_X = (PIECES_X + NW - 1) / NW;
    constexpr int PER = PER_G + PER_X;
    constexpr int NST = 3;                  // LDS stages (see conv_bf16_kernel: counted vmcnt, one raw barrier per step)
    constexpr int BUF = 32 * (BM + BN) * 2;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * BUF];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int Ktot = p.TY * p.TX * p.Cin;
    int tile, split;
    splitk_xcd_map(blockIdx.x, tiles, splits, tile, split);
    const int tile_n = tile % tiles_n, tile_m = tile / tiles_n;
    const int o0 = tile_m * BM, n0 = tile_n * BN;
    const int P = p.B * p.OH * p.OW;
    // plain: splits cut the flattened pixel axis; SCALE: `splits_per_img` splits per image, none straddles two samples
    const int OHW = p.OH * p.OW;
    const int bimg = SCALE ? split / splits_per_img : 0;
    const int pbeg = SCALE ? bimg * OHW + (split - bimg * splits_per_img) * pix_per_split : split * pix_per_split;
    const int plim = SCALE ? (bimg + 1) * OHW : P;
    const int pend = pbeg + pix_per_split < plim ? pbeg + pix_per_split : plim;
    if (pbeg >= pend) return;

    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)gy, 0, (int)gy_bytes, (int)RSRC);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)x_bytes, (int)RSRC);

    // ---- this lane's DMA slots: piece q = wave + 4 j; LDS slot = 64 (q - first piece of its operand) + lane --------------
    int s_pix[PER];                    // pixel row (0..31) within the step
    unsigned s_cb[PER];                // byte offset of the channel chunk inside a source pixel (G: o0 + 8c; X: ci of column n0 + 8c)
    int s_yoff[PER], s_xoff[PER];      // tap shift (X) / output placement (G)
    bool s_valid[PER];
    int w_b[PER], w_oy[PER], w_ox[PER], w_p[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const bool is_g = j < PER_G;
        const int q = is_g ? (wave + NW * j) % PIECES_G : (wave + NW * (j - PER_G)) % PIECES_X;    // piece within its operand
        const int slot = q * 64 + lane;
        s_valid[j] = true;
        int r, c;
        if (is_g) {
            constexpr int CPR = BM / 8;
            r = slot / CPR;
            const int cs = slot % CPR;
            c = CPR >= 16 ? cs ^ ((r & 3) << 2) : (CPR == 8 ? cs ^ (((r >> 1) & 1) << 2) : cs);
            const int o = o0 + 8 * c;
            s_cb[j] = (unsigned)o * 2u;
            s_yoff[j] = p.ooy; s_xoff[j] = p.oox;
            if (o >= p.Cout) s_valid[j] = false;
        } else {
            constexpr int CPR = BN / 8;
            r = slot / CPR;
            const int cs = slot % CPR;
            c = CPR >= 16 ? cs ^ ((r & 3) << 2) : (CPR == 8 ? cs ^ (((r >> 1) & 1) << 2) : cs);
            const int k = n0 + 8 * c;
            const int tap = k / p.Cin, ci = k - tap * p.Cin;
            const int ty = tap / p.TX, tx = tap - ty * p.TX;
            s_cb[j] = (unsigned)ci * 2u;
            s_yoff[j] = ty * p.dy + p.offy; s_xoff[j] = tx * p.dx + p.offx;
            if (k >= Ktot) s_valid[j] = false;
        }
        s_pix[j] = r;
        const int pp = pbeg + r;
        const int qq = pp / p.OW;
        w_ox[j] = pp - qq * p.OW;
        w_b[j] = qq / p.OH;
        w_oy[j] = qq - w_b[j] * p.OH;
        w_p[j] = pp;
    }
    // a step advances 32 pixels of the flattened (b, oy, ox) axis.  32 | OW: d_oy = 0;  OW | 32: d_ox = 0;  images of <= 32 pixels
    // whose size divides 32 (the <= 4 x 4 maps of the co-occurrence discriminator's tail): 32 / OHW whole samples, same pixel
    const int d_b = (OHW <= 32 && 32 % OHW == 0) ? 32 / OHW : 0;
    const int d_ox = d_b ? 0 : 32 % p.OW, d_oy = d_b ? 0 : 32 / p.OW;

    auto dma = [&](int buf) {
        unsigned char* base = smem + buf * BUF;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const bool is_g = j < PER_G;                                   // compile-time after unrolling
            const int q = is_g ? (wave + NW * j) % PIECES_G : PIECES_G + (wave + NW * (j - PER_G)) % PIECES_X;
            const int sH = is_g ? p.YH : p.IH, sW = is_g ? p.YW : p.IW, sC = is_g ? p.Cout : p.Cin;
            int iy = w_oy[j] * (is_g ? p.osy : p.sy) + s_yoff[j];
            int ix = w_ox[j] * (is_g ? p.osx : p.sx) + s_xoff[j];
            bool ok = s_valid[j] && w_p[j] < pend;
            if (REFLECT && !is_g) { iy = reflect_coord(iy, sH); ix = reflect_coord(ix, sW); }
            else ok = ok && (unsigned)iy < (unsigned)sH && (unsigned)ix < (unsigned)sW;
            const unsigned off = (unsigned)(((w_b[j] * sH + iy) * sW + ix) * sC) * 2u + s_cb[j];
            if (is_g) __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, LDS_PTR(base + q * 1024), 16, (int)(ok ? off : 0xffffffffu), 0, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(base + q * 1024), 16, (int)(ok ? off : 0xffffffffu), 0, 0, 0);
            // advance 32 pixels
            w_p[j] += 32;
            w_ox[j] += d_ox;
            const bool cx = w_ox[j] >= p.OW;
            w_ox[j] -= cx ? p.OW : 0;
            w_oy[j] += d_oy + (cx ? 1 : 0);
            const bool cy = w_oy[j] >= p.OH;
            w_oy[j] -= cy ? p.OH : 0;
            w_b[j] += d_b + (cy ? 1 : 0);
        }
    };

    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // Transpose-read addresses.  MFMA K-slice s covers pixel rows 16 s .. 16 s + 15; lane (li, lh) needs rows 16 s + 8 lh + 0..7 of
    // channel (block) * 32 + li: two ds_read_b64_tr_b16, each over a [4 rows][16 channels] block.  Within its 16-lane group a
    // lane POINTS at row (lane & 15) >> 2, 8-byte piece lane & 3 of the block; it RECEIVES column lane & 15.
    const int g_q = lane & 15;
    const int g_row = g_q >> 2, g_piece = g_q & 3;           // which row / 8-byte piece this lane points at
    const int g_cblk = (lane >> 4) & 1;                      // 16-channel half of the 32-channel MFMA block
    auto tr_addr = [&](int tileR_is_M, int blk, int s, int half) -> int {   // byte offset inside the operand's image
        const int r = 16 * s + 8 * lh + 4 * half + g_row;
        const int c = blk * 4 + g_cblk * 2 + (g_piece >> 1);                // 16-byte chunk (8 channels) index in the row
        const int slot = tileR_is_M ? chunk_slot<BM>(r, c) : chunk_slot<BN>(r, c);
        return slot * 16 + (g_piece & 1) * 8;
    };
    int a_addr[MT][2][2], b_addr[NT][2][2];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int h = 0; h < 2; ++h) a_addr[a][s][h] = tr_addr(1, wm * MT + a, s, h);
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int h = 0; h < 2; ++h) b_addr[b][s][h] = 32 * BM * 2 + tr_addr(0, wn * NT + b, s, h);

    // The transpose reads are issued through inline asm: for the __builtin_amdgcn_ds_read_tr16_b64 form hipcc (ROCm 7.2) puts an
    // s_waitcnt vmcnt(0) in front of the first read of every step (it orders the read behind ALL pending LDS-DMA), which drains
    // the pipeline the counted wait above keeps full.  The asm form is opaque to that pass; the data hazard is handled by hand:
    // one s_waitcnt lgkmcnt(0) + sched_barrier before the MFMAs (cdna_hip_programming.md §5.4 rule 18).
    const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(smem);
    auto tr_ld = [&](unsigned addr) -> s16x4 {
        s16x4 v;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
        return v;
    };
    typedef short s16x8 __attribute__((ext_vector_type(8)));

    const int nsteps = (pend - pbeg + 31) / 32;
#pragma unroll
    for (int st = 0; st < NST - 1; ++st) dma(st);    // prologue: steps 0 .. NST-2 (past pend: zero fill)
    int st_cur = 0, st_free = NST - 1;
    for (int s_ = 0; s_ < nsteps; ++s_) {
        wait_vmcnt<(NST - 2) * PER>();               // this wave's pieces of step s_ have landed
        __builtin_amdgcn_s_barrier();                // ... and everybody else's; all waves are done with step s_-1
        dma(st_free);
        const unsigned sbase = lds0 + (unsigned)(st_cur * BUF);
        s16x4 ra[MT][2][2], rb_[NT][2][2];
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int h = 0; h < 2; ++h) ra[a][s][h] = tr_ld(sbase + (unsigned)a_addr[a][s][h]);
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int h = 0; h < 2; ++h) rb_[b][s][h] = tr_ld(sbase + (unsigned)b_addr[b][s][h]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 fa[MT][2], fb[NT][2];
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const s16x8 v = {ra[a][s][0][0], ra[a][s][0][1], ra[a][s][0][2], ra[a][s][0][3],
                                 ra[a][s][1][0], ra[a][s][1][1], ra[a][s][1][2], ra[a][s][1][3]};
                fa[a][s] = __builtin_bit_cast(bf16x8, v);
            }
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const s16x8 v = {rb_[b][s][0][0], rb_[b][s][0][1], rb_[b][s][0][2], rb_[b][s][0][3],
                                 rb_[b][s][1][0], rb_[b][s][1][1], rb_[b][s][1][2], rb_[b][s][1][3]};
                fb[b][s] = __builtin_bit_cast(bf16x8, v);
            }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][s], fb[b][s], acc[a][b], 0, 0, 0);
        st_free = st_cur;
        st_cur = st_cur + 1 == NST ? 0 : st_cur + 1;
    }
    wait_vmcnt<0>();

    // ---- epilogue: D rows = output channels (r&3) + 8 (r>>2) + 4 lh of block a, column = k column li of block b ------------
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        const int k = n0 + (wn * NT + b) * 32 + li;
        if (k >= Ktot) continue;
        const float sk = SCALE ? in_scale[(int64_t)bimg * p.Cin + (k % p.Cin)] * p.gain : p.gain;
#pragma unroll
        for (int a = 0; a < MT; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = o0 + (wm * MT + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (o >= p.Cout) continue;
                float v = acc[a][b][r] * sk;
                if (SCALE) v *= out_scale[(int64_t)bimg * p.Cout + o];
                atomicAdd(&gw[(int64_t)o * Ktot + k], v);
            }
        }
    }
}

template <int WM, int WN, int MT, int NT>
int launch_bf16_wgrad_cfg(float* gw, const void* gy, const void* x, const float* in_scale, const float* out_scale,
                          const ideas_conv_params* p, hipStream_t stream) {
    constexpr int BM_ = WM * MT * 32, BN_ = WN * NT * 32;
    const int64_t P = (int64_t)p->B * p->OH * p->OW;
    const int Ktot = p->TY * p->TX * p->Cin;
    const int tm = (int)ideas_cdiv(p->Cout, BM_);
    const int tn = (int)ideas_cdiv(Ktot, BN_);
    const int64_t tiles = (int64_t)tm * tn;
    const bool sc = in_scale && out_scale;
    const int64_t img = (int64_t)p->OH * p->OW;
    // split-K: about two waves of resident blocks (2 per CU), >= 8 steps each; modulated convs: whole splits inside one image
    const int64_t slots = (WM * WN == 8 ? 1 : 2) * 256;
    // (four waves of blocks where every block still reduces >= 4096 pixels, two otherwise: conv_b3_wgrad.hip; bf16: 572 -> 645
    //  TFLOP/s on 128->128 @256x256, 730 -> 789 on 256->512 @64x64)
    int64_t splits = (4 * slots) / tiles;
    if (splits < 1 || P / splits < 4096) splits = (2 * slots) / tiles;
    if (splits < 1) splits = 1;
    int64_t per, spi = 1;
    if (sc) {
        spi = splits / p->B;                         // splits per image
        const int64_t max_spi = ideas_cdiv(img, 32 * 4);
        if (spi > max_spi) spi = max_spi;
        if (spi < 1) spi = 1;
        per = ideas_cdiv(ideas_cdiv(img, spi), 32) * 32;
        spi = ideas_cdiv(img, per);
        splits = spi * p->B;
    } else {
        const int64_t max_splits = ideas_cdiv(P, 32 * 8);
        if (splits > max_splits) splits = max_splits;
        per = ideas_cdiv(ideas_cdiv(P, splits), 32) * 32;
        splits = ideas_cdiv(P, per);
    }
    if (tiles * splits > 0x7fffffffLL) return IDEAS_E_SHAPE;
    const unsigned gy_bytes = (unsigned)((int64_t)p->B * p->YH * p->YW * p->Cout * 2);
    const unsigned x_bytes = (unsigned)((int64_t)p->B * p->IH * p->IW * p->Cin * 2);
    auto go = [&](auto s_, auto rf) {
        hipLaunchKernelGGL((conv_bf16_wgrad_kernel<WM, WN, MT, NT, decltype(s_)::value, decltype(rf)::value>),
                           dim3(splitk_grid(tiles, splits)), dim3(64 * WM * WN), 0, stream, gw, (const bf16_t*)gy, (const bf16_t*)x,
                           in_scale, out_scale, *p, tn, (int)per, (int)spi, gy_bytes, x_bytes, (int)tiles, (int)splits);
    };
    using T = std::true_type;
    using F = std::false_type;
    if (sc) { if (p->reflect) go(T{}, T{}); else go(T{}, F{}); }
    else { if (p->reflect) go(F{}, T{}); else go(F{}, F{}); }
    return ideas_launch_status();
}

}  // namespace

extern "C" int ideas_bf16_conv_supported(const ideas_conv_params* p, int scaled) {
    if (!p) return 0;
    if (p->Cin % 32 || p->Cout % 4 || p->TY * p->TX > 32) return 0;
    if ((int64_t)p->B * p->IH * p->IW * p->Cin * 2 >= 0xffffffffLL) return 0;
    if ((int64_t)(scaled ? p->B : 1) * p->TY * p->TX * p->Cin * p->Cout * 2 >= 0xffffffffLL) return 0;   // (all per-sample packs)
    return 1;
}

extern "C" int ideas_bf16_wgrad_supported(const ideas_conv_params* p, int scaled) {
    if (!p) return 0;
    const int64_t P = (int64_t)p->B * p->OH * p->OW;
    if (p->Cin % 8 || p->Cout % 8) return 0;
    const int64_t ohw = (int64_t)p->OH * p->OW;
    if (!(ohw <= 32 && 32 % ohw == 0) && (!(p->OW % 32 == 0 || 32 % p->OW == 0) || 32 / p->OW > p->OH)) return 0;
    if (P >= 0x7fffffffLL || (int64_t)p->B * p->IH * p->IW * p->Cin * 2 >= 0xffffffffLL ||
        (int64_t)p->B * p->YH * p->YW * p->Cout * 2 >= 0xffffffffLL)
        return 0;
    (void)scaled;
    return 1;
}

extern "C" int ideas_bf16_pack_weights(void* pack, const void* wmat, const float* in_scale, int B, int Cout, int K, int Cin,
                                       void* stream_) {
    if (!pack || !wmat) return IDEAS_E_NULL;
    if (Cout <= 0 || K <= 0 || Cin <= 0 || K % Cin || B <= 0 || B > 65535) return IDEAS_E_SHAPE;
    if (Cin % 32 || !ideas_aligned16(pack) || !ideas_aligned16(wmat) || (in_scale && !ideas_aligned16(in_scale))) return IDEAS_E_ALIGN;
    if (!in_scale && B != 1) return IDEAS_E_SHAPE;
    const int64_t n8 = (int64_t)Cout * (K / 8);
    const int blocks = (int)(n8 / 256 + 1 < 2048 ? n8 / 256 + 1 : 2048);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks, in_scale ? B : 1), dim3(256), 0, (hipStream_t)stream_, (uint4*)pack,
                       (const float4*)wmat, in_scale, Cout, K, Cin);
    return ideas_launch_status();
}

void ideas_bf16_pack_batched(const ideas_prep_desc* tbl, int n, int blocks, hipStream_t stream) {
    hipLaunchKernelGGL(pack_weights_batched_kernel, dim3(blocks), dim3(256), 0, stream, tbl, n);
}

extern "C" int ideas_bf16_pack_weights_strided(void* pack, const float* w, const float* in_scale, int B, int Cout, int TY, int TX,
                                               int Cin, int64_t sn, int64_t sty, int64_t stx, int64_t sc, void* stream_) {
    if (!pack || !w) return IDEAS_E_NULL;
    if (Cout <= 0 || TY <= 0 || TX <= 0 || Cin <= 0 || B <= 0 || B > 65535) return IDEAS_E_SHAPE;
    if (Cin % 32 || !ideas_aligned16(pack) || (in_scale && !ideas_aligned16(in_scale))) return IDEAS_E_ALIGN;
    if (!in_scale && B != 1) return IDEAS_E_SHAPE;
    const int64_t n8 = (int64_t)Cout * TY * TX * (Cin / 8);
    const int blocks = (int)(n8 / 256 + 1 < 2048 ? n8 / 256 + 1 : 2048);
    const bool unit = sc == 1 && ideas_aligned16(w) && sn % 4 == 0 && sty % 4 == 0 && stx % 4 == 0;
    const dim3 grid(blocks, in_scale ? B : 1);
    if (unit)
        hipLaunchKernelGGL(pack_weights_strided_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream_, (uint4*)pack, w, in_scale, Cout,
                           TY, TX, Cin, sn, sty, stx, sc);
    else
        hipLaunchKernelGGL(pack_weights_strided_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream_, (uint4*)pack, w, in_scale, Cout,
                           TY, TX, Cin, sn, sty, stx, sc);
    return ideas_launch_status();
}

// called by ideas_conv_igemm / ideas_conv_wgrad for dtype IDEAS_BF16 once the arguments are validated
int ideas_bf16_fwd(void* y, const void* x, const void* wpack, int per_image, const float* out_scale, const float* bias,
                   const void* resid, const ideas_conv_params* p, hipStream_t stream) {
    // Measured (tools/bench_igemm.py --dtype bf16 --cfg N, profiles/r02_bf16_tile_sweep.txt): with >= 256 output channels the
    // 8-wave 256 x 128 tile is 8-10 % ahead of the 4-wave 128 x 128 tile (0.375 instead of 0.5 DMA pieces per MFMA), 256 x 256 adds
    // 1-3 % on 512-channel layers only and loses badly below; a fourth LDS stage and a register-prefetch pipeline (fragments of
    // tile t+1 read under the MFMAs of tile t) both measured 3-5 % SLOWER than three stages + counted vmcnt.
    if (ideas_bf16_pw_ok(p, per_image, out_scale, y, resid)) return ideas_bf16_pw_fwd(y, x, wpack, bias, resid, p, stream);   // flat 1x1 (conv_bf16_pw.hip)
    const int64_t rows = (int64_t)(per_image ? 1 : p->B) * p->OH * p->OW;
    if (const int tp = bf16_img_tp(p)) {                  // 3x3 / stride 1: the activation operand as an LDS image (IDEAS_BF16_IMG=0: off)
        if (p->Cout > 64) return launch_bf16_img_tp<2>(tp, y, x, wpack, per_image, out_scale, bias, resid, p, stream);
        return launch_bf16_img_tp<1>(tp, y, x, wpack, per_image, out_scale, bias, resid, p, stream);
    }
    if (p->Cout >= 256 && rows >= 256) return launch_bf16_cfg<4, 2, 2, 2>(y, x, wpack, per_image, out_scale, bias, resid, p, stream);   // 256 x 128, 8 waves
    if (p->Cout > 64) return launch_bf16_cfg<2, 2, 2, 2>(y, x, wpack, per_image, out_scale, bias, resid, p, stream);   // 128 x 128
    if (p->Cout > 32) return launch_bf16_cfg<2, 2, 2, 1>(y, x, wpack, per_image, out_scale, bias, resid, p, stream);   // 128 x 64
    return launch_bf16_cfg<4, 1, 1, 1>(y, x, wpack, per_image, out_scale, bias, resid, p, stream);                     // 128 x 32
}

// ideas_conv_igemm_multi for dtype IDEAS_BF16 (conv_igemm.hip validates): n <= 4 launches sharing x / y / scales
int ideas_bf16_fwd_multi(int n, void* y, const void* x, const void* const* wpack, int per_image, const float* out_scale,
                         const ideas_conv_params* ps, hipStream_t stream) {
    const ideas_conv_params* p = ps;
    const int64_t rows = (int64_t)(per_image ? 1 : p->B) * p->OH * p->OW;
    if (p->Cout >= 256 && rows >= 256) return launch_bf16_multi_cfg<4, 2, 2, 2>(n, y, x, wpack, per_image, out_scale, ps, stream);
    if (p->Cout > 64) return launch_bf16_multi_cfg<2, 2, 2, 2>(n, y, x, wpack, per_image, out_scale, ps, stream);
    if (p->Cout > 32) return launch_bf16_multi_cfg<2, 2, 2, 1>(n, y, x, wpack, per_image, out_scale, ps, stream);
    return launch_bf16_multi_cfg<4, 1, 1, 1>(n, y, x, wpack, per_image, out_scale, ps, stream);
}

// Tile-shape A/B for tools/bench_igemm.py --cfg (not part of the declared ABI; the production dispatch is ideas_bf16_fwd)
extern "C" int ideas_tune_bf16_fwd(int cfg, void* y, const void* x, const void* wpack, int per_image, const float* out_scale,
                                   const float* bias, const void* resid, const ideas_conv_params* p, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    switch (cfg) {
        case 0: return launch_bf16_cfg<2, 2, 2, 2, 3>(y, x, wpack, per_image, out_scale, bias, resid, p, stream);   // 128x128, 4 waves
        case 1: return launch_bf16_cfg<2, 2, 2, 2, 4>(y, x, wpack, per_image, out_scale, bias, resid, p, stream);   // + 4 stages
        case 2: return launch_bf16_cfg<2, 2, 4, 2, 3>(y, x, wpack, per_image, out_scale, bias, resid, p, stream);   // 256x128, 4 waves
        case 3: return launch_bf16_cfg<4, 2, 2, 2, 3>(y, x, wpack, per_image, out_scale, bias, resid, p, stream);   // 256x128, 8 waves
        case 4: return launch_bf16_cfg<2, 4, 4, 2, 3>(y, x, wpack, per_image, out_scale, bias, resid, p, stream);   // 256x256, 8 waves
        case 5: return launch_bf16_cfg<2, 2, 2, 2, 2>(y, x, wpack, per_image, out_scale, bias, resid, p, stream);   // 2 stages (vmcnt(0) every step)
        case 6: return launch_bf16_cfg<2, 2, 2, 4, 3>(y, x, wpack, per_image, out_scale, bias, resid, p, stream);   // 128x256, 4 waves
        default: return IDEAS_E_UNSUPPORTED;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradient of the 3x3 / stride-1 convolutions, tap-fused with a rolling activation window (the bf16 sibling of
// conv_b3_wgrad3.hip):     gw[o][ty][tx][ci] += gain * sum_{b, oy, ox} G(b, oy, ox, o) * X(b, oy + ty + offy, ox + tx + offx, ci)
// conv_bf16_wgrad_kernel treats the nine taps as nine column tiles of a GEMM: per 32-pixel step it DMAs a 128-channel piece of G
// and of X for 32 MFMAs (0.5 DMA pieces per MFMA; every G element fetched 9 Cin / 128 times, every X element 9 Cout / 128 times),
// and is bound by issuing them (13 vector instructions per MFMA with the transpose-read addressing).  Here one block owns a
// 64 (o) x 64 (ci) tile for ALL nine taps and walks down a 32-pixel-wide column strip of one image:
//   * per step (one output row of the strip) the DMA brings one 32-pixel row of G and ONE new row of the X window (34 pixels); the
//     other two window rows are still in LDS (ring of 5 rows: 3 in use, 2 in flight);
//   * the nine taps are 18 MFMAs per wave on those operands: tap (ty, tx) reads the pixel-major image ([pixel][64 channels], 128 B
//     rows, chunk c of row r at c ^ (((r >> 1) & 1) << 2) as in conv_b3_wgrad3.hip: conflict-free transpose reads at any pixel
//     offset) at window row ty and pixel offset tx through ds_read_b64_tr_b16;
//   * 3 DMA pieces per wave and step for 18 MFMAs (was 4 for 8), counted vmcnt, one raw barrier per step, two steps of DMA in flight.
// A block's range of rows lies inside ONE image (so the per-sample scales of a modulated conv multiply the accumulators in the
// epilogue, and nothing is re-primed mid-range); grid = tiles x strips x (image, part) in XCD-banded split-major order; f32
// atomics into gw.  Requires TY = TX = 3, unit stride / dilation, OW % 32 == 0, Cin % 64 == 0, Cout % 64 == 0, dense G.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int w3_chunk_off(int r, int c) { return (r * 8 + (c ^ (((r >> 1) & 1) << 2))) * 16; }

template <bool SCALE, bool REFLECT>
__global__ __launch_bounds__(256, 2) void conv_bf16_wgrad3_kernel(float* __restrict__ gw, const bf16_t* __restrict__ gy,
                                                                  const bf16_t* __restrict__ x, const float* __restrict__ in_scale,
                                                                  const float* __restrict__ out_scale, ideas_conv_params p,
                                                                  int tiles_ci, int tiles, int splits, int strips, int parts,
                                                                  int rows_per_part, unsigned gy_bytes, unsigned x_bytes) {
    constexpr int XW = 40;                      // LDS pixels per window row (34 in use; 5 DMA pieces)
    constexpr int NR = 6;                       // ring of window rows: 3 in use + 2 in flight (+ 1: the step loop is unrolled over the six ring
                                                // positions, so that every LDS read address is a lane register + an immediate)
    constexpr int NG = 3;                       // G row buffers: 1 in use + 2 in flight
    constexpr int XBYTES = NR * XW * 128, GBYTES = NG * 32 * 128;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[XBYTES + GBYTES];
    unsigned char* const sX = smem;
    unsigned char* const sG = smem + XBYTES;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    int tile, split;
    splitk_xcd_map(blockIdx.x, tiles, splits, tile, split);
    const int o0 = (tile / tiles_ci) * 64, c0 = (tile % tiles_ci) * 64;
    const int strip = split % strips, rng = split / strips;
    const int img = rng / parts, part = rng - img * parts;
    const int ox0 = strip * 32;
    const int oa = part * rows_per_part;
    const int ob = oa + rows_per_part < p.OH ? oa + rows_per_part : p.OH;
    if (oa >= ob) return;

    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)gy, 0, (int)gy_bytes, (int)RSRC);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)x_bytes, (int)RSRC);

    // ---- DMA slots of a wave per step: G piece `swave` (8 pixels), X piece `swave`, X piece 4 (every wave: same bytes) --------------
    // lane l fills position l & 7 of LDS pixel 8 q + (l >> 3), which must hold channel chunk (l & 7) ^ (((pixel >> 1) & 1) << 2)
    const int lpx = lane >> 3, lpos = lane & 7;
    unsigned g_col, x_col[2];
    bool x_ok[2];
    {
        const int px = 8 * swave + lpx;
        g_col = (unsigned)((ox0 + px) * p.Cout + o0 + 8 * (lpos ^ (((px >> 1) & 1) << 2))) * 2u;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int j = 8 * (k == 0 ? swave : 4) + lpx;              // window column 0..39
            int ix = ox0 + p.offx + j;
            bool ok = j < 34;
            if (REFLECT) ix = reflect_coord(ix, p.IW); else ok = ok && ix >= 0 && ix < p.IW;
            x_ok[k] = ok;
            x_col[k] = (unsigned)((ok ? ix : 0) * p.Cin + c0 + 8 * (lpos ^ (((j >> 1) & 1) << 2))) * 2u;
        }
    }
    // task n (n = 0, 1, ...): window row wy = oa - 1 + n relative to offy + 1 (input row oa + offy + n); it completes output row
    // oy = oa + n - 2 when n >= 2.  The last task is n = (ob - oa) + 1.
    const int ntask = ob - oa + 2;
    auto dma = [&](int n) {
        const bool live = n < ntask;
        int iy = oa + p.offy + n;
        bool yok = live;
        if (REFLECT) iy = reflect_coord(iy, p.IH); else yok = yok && iy >= 0 && iy < p.IH;
        const unsigned xrow = (unsigned)((img * p.IH + (yok ? iy : 0)) * p.IW) * (unsigned)p.Cin * 2u;
        const bool gl = live && n >= 2;
        const unsigned grow = (unsigned)((img * p.OH + (gl ? oa + n - 2 : 0)) * p.OW) * (unsigned)p.Cout * 2u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, LDS_PTR(sG + ((n % NG) * 32 + 8 * swave) * 128), 16, (int)(gl ? grow + g_col : 0xffffffffu), 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(sX + ((n % NR) * XW + 8 * swave) * 128), 16,
                                                 (int)((yok && x_ok[0]) ? xrow + x_col[0] : 0xffffffffu), 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(sX + ((n % NR) * XW + 32) * 128), 16,
                                                 (int)((yok && x_ok[1]) ? xrow + x_col[1] : 0xffffffffu), 0, 0, 0);
    };

    // ---- MFMA side: wave = (o half, ci half); 9 accumulators of 32 (o) x 32 (ci) ----------------------------------------------------
    const int wo = wave >> 1, wc = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int g_q = lane & 15, g_row = g_q >> 2, g_piece = g_q & 3, g_cblk = (lane >> 4) & 1;
    auto tr_off = [&](int r, int cb) { return w3_chunk_off(r, cb * 2 + (g_piece >> 1)) + (g_piece & 1) * 8; };
    const unsigned ldsG = (unsigned)(uintptr_t)LDS_PTR(sG), ldsX = (unsigned)(uintptr_t)LDS_PTR(sX);
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    // Transpose reads with NO per-read address arithmetic: the (r >> 1) & 1 swizzle of a pixel row depends only on (row mod 4) and the
    // lane (row-slot bases XW * slot and 32 * buffer and the lane's row offsets 8 lh, 4 h are multiples of 4), so with the ring
    // position a compile-time constant (loop unrolled over the six positions) an address is one of FOUR lane registers per operand
    // (pixel offset tx + 16 s + ... mod 4) plus an immediate.  Computed per read they were ~40 of the ~60 vector instructions of a
    // step, next to 18 MFMAs -- and a VALU instruction on a SIMD is time taken from its matrix pipe.
    auto lane_base = [&](unsigned region, int b, int cb) {
        const int r = b + 8 * lh + g_row;
        return region + (unsigned)(r * 128 + (((cb * 2 + (g_piece >> 1)) ^ ((((b + g_row) >> 1) & 1) << 2)) << 4) + (g_piece & 1) * 8);
    };
    const unsigned baseG = lane_base(ldsG, 0, wo * 2 + g_cblk);
    const unsigned bx0 = lane_base(ldsX, 0, wc * 2 + g_cblk), bx1 = lane_base(ldsX, 1, wc * 2 + g_cblk), bx2 = lane_base(ldsX, 2, wc * 2 + g_cblk);
    (void)tr_off;
#define TR_LD(dst, base, imm) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "n"(imm))
    f32x16 acc[9];
#pragma unroll
    for (int a = 0; a < 9; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    // one step with ring phase PH = n % 6 (compile-time): G buffer PH % 3, window rows at slots (PH + 4 + ty) % 6
    auto compute = [&](auto ph_) {
        constexpr int PH = decltype(ph_)::value;
        constexpr int GB = (PH % NG) * 32 * 128;
        s16x4 rg_[2][2], rx_[2][2][2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                // pixel row 16 s2 + 4 h (+ 8 lh + g_row in the lane register): a multiple of 4 -> lane base 0
                switch (s2 * 2 + h) {
                    case 0: TR_LD(rg_[0][0], baseG, GB + 0 * 128); break;
                    case 1: TR_LD(rg_[0][1], baseG, GB + 4 * 128); break;
                    case 2: TR_LD(rg_[1][0], baseG, GB + 16 * 128); break;
                    default: TR_LD(rg_[1][1], baseG, GB + 20 * 128); break;
                }
            }
        // X fragment of tap (ty, tx), K-slice s2, half h: pixel row slot * XW + 16 s2 + 4 h + tx; XW * slot is a multiple of 4
        auto readX = [&](auto tap_, s16x4 (&dst)[2][2]) {
            constexpr int tap = decltype(tap_)::value;
            constexpr int ty = tap / 3, tx = tap - 3 * ty;
            constexpr int row0 = ((PH + 4 + ty) % NR) * XW;
            // (row0 + 16 s2 + 4 h) is a multiple of 4: the lane base follows tx (0 .. 2)
            const unsigned bb = tx == 0 ? bx0 : (tx == 1 ? bx1 : bx2);
            TR_LD(dst[0][0], bb, (row0 + 0) * 128);
            TR_LD(dst[0][1], bb, (row0 + 4) * 128);
            TR_LD(dst[1][0], bb, (row0 + 16) * 128);
            TR_LD(dst[1][1], bb, (row0 + 20) * 128);
        };
        readX(std::integral_constant<int, 0>{}, rx_[0]);
        bf16x8 fa[2];
        auto tapstep = [&](auto tap_) {
            constexpr int tap = decltype(tap_)::value;
            if constexpr (tap < 8) {
                readX(std::integral_constant<int, tap + 1>{}, rx_[(tap + 1) & 1]);
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (tap == 0) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const s16x8 v = {rg_[s2][0][0], rg_[s2][0][1], rg_[s2][0][2], rg_[s2][0][3], rg_[s2][1][0], rg_[s2][1][1], rg_[s2][1][2], rg_[s2][1][3]};
                    fa[s2] = __builtin_bit_cast(bf16x8, v);
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const s16x4 a = rx_[tap & 1][s2][0], b = rx_[tap & 1][s2][1];
                const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s2], __builtin_bit_cast(bf16x8, v), acc[tap], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        tapstep(std::integral_constant<int, 0>{}); tapstep(std::integral_constant<int, 1>{}); tapstep(std::integral_constant<int, 2>{});
        tapstep(std::integral_constant<int, 3>{}); tapstep(std::integral_constant<int, 4>{}); tapstep(std::integral_constant<int, 5>{});
        tapstep(std::integral_constant<int, 6>{}); tapstep(std::integral_constant<int, 7>{}); tapstep(std::integral_constant<int, 8>{});
    };
    static_assert(NR == 6 && NG == 3, "the step loop is unrolled over the ring positions");
    // The transpose reads are inline asm (see conv_bf16_wgrad_kernel), so the waits are placed by hand: LDS operations return in
    // order, the reads of tap + 1 are issued before the MFMAs of tap, and lgkmcnt(4) = "everything but the last four reads".
    int n = 0;
    auto step = [&](auto ph_) -> bool {
        if (n >= ntask) return false;
        wait_vmcnt<3>();                            // this wave's pieces of task n have landed (task n + 1's three stay in flight)
        __builtin_amdgcn_s_barrier();               // ... and everybody else's; all waves are done with task n - 1's buffers
        dma(n + 2);
        if (n >= 2) compute(ph_);                   // (n < 2: priming, the window is not complete yet -- block-uniform)
        ++n;
        return true;
    };
    dma(0);
    dma(1);
    while (step(std::integral_constant<int, 0>{}) && step(std::integral_constant<int, 1>{}) && step(std::integral_constant<int, 2>{}) &&
           step(std::integral_constant<int, 3>{}) && step(std::integral_constant<int, 4>{}) && step(std::integral_constant<int, 5>{})) {}
#undef TR_LD
    wait_vmcnt<0>();

    // ---- epilogue: D rows = o (r & 3) + 8 (r >> 2) + 4 lh, column = ci li; gw is OHWI ---------------------------------------------
    const int ci = c0 + wc * 32 + li;
    const float sk = SCALE ? in_scale[(int64_t)img * p.Cin + ci] * p.gain : p.gain;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int o = o0 + wo * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float so = SCALE ? out_scale[(int64_t)img * p.Cout + o] * sk : sk;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) atomicAdd(&gw[((int64_t)o * 9 + tap) * p.Cin + ci], acc[tap][r] * so);
    }
}

static bool bf16_wgrad3_ok(const ideas_conv_params* p) {
    const char* e = getenv("IDEAS_BF16_WGRAD3");          // (read per call: the tests toggle it in-process)
    if (e && e[0] == '0') return false;
    if (p->TY != 3 || p->TX != 3 || p->sy != 1 || p->sx != 1 || p->dy != 1 || p->dx != 1) return false;
    if (p->osy != 1 || p->osx != 1 || p->ooy || p->oox || p->YH != p->OH || p->YW != p->OW) return false;
    if (p->OW % 32 || p->Cin % 64 || p->Cout % 64 || p->OH < 8) return false;
    if (p->offx > 0 || p->offx < -2 || p->offy > 0 || p->offy < -2) return false;
    if (p->reflect && (p->IH < 2 || p->IW < 2)) return false;
    // (small layers: the generic kernel's fewer, fatter splits win -- E.2.conv1, 64 -> 128 at 128x128, B = 32: 549 vs 459 TFLOP/s)
    return (int64_t)p->B * p->OH * p->OW * p->Cin * p->Cout >= 6000000000LL;
}

int launch_bf16_wgrad3(float* gw, const void* gy, const void* x, const float* in_scale, const float* out_scale, const ideas_conv_params* p,
                       hipStream_t stream) {
    const int tiles_ci = p->Cin / 64, tiles = (p->Cout / 64) * tiles_ci;
    const int strips = p->OW / 32;
    // parts per image: about four waves of the 512 resident blocks (two per CU) while a part keeps >= 32 rows, at least 8 rows
    const int64_t base = (int64_t)tiles * strips * p->B;
    int64_t parts = (4 * 512) / base;
    if (parts < 1) parts = 1;
    while (parts > 1 && p->OH / parts < 32) --parts;
    if (parts > p->OH / 8) parts = p->OH / 8 > 0 ? p->OH / 8 : 1;
    const int rows = (int)ideas_cdiv(p->OH, parts);
    parts = ideas_cdiv(p->OH, rows);
    const int64_t splits = (int64_t)strips * p->B * parts;
    if ((int64_t)tiles * splits > 0x7fffffffLL) return IDEAS_E_SHAPE;
    const unsigned gy_bytes = (unsigned)((int64_t)p->B * p->YH * p->YW * p->Cout * 2);
    const unsigned x_bytes = (unsigned)((int64_t)p->B * p->IH * p->IW * p->Cin * 2);
    auto go = [&](auto s_, auto rf) {
        hipLaunchKernelGGL((conv_bf16_wgrad3_kernel<decltype(s_)::value, decltype(rf)::value>), dim3(splitk_grid(tiles, splits)), dim3(256), 0,
                           stream, gw, (const bf16_t*)gy, (const bf16_t*)x, in_scale, out_scale, *p, tiles_ci, tiles, (int)splits, strips,
                           (int)parts, rows, gy_bytes, x_bytes);
    };
    using T = std::true_type;
    using F = std::false_type;
    const bool sc = in_scale && out_scale;
    if (sc) { if (p->reflect) go(T{}, T{}); else go(T{}, F{}); }
    else { if (p->reflect) go(F{}, T{}); else go(F{}, F{}); }
    return ideas_launch_status();
}

int ideas_bf16_wgrad(float* gw, const void* gy, const void* x, const float* in_scale, const float* out_scale,
                     const ideas_conv_params* p, hipStream_t stream) {
    if (bf16_wgrad3_ok(p)) return launch_bf16_wgrad3(gw, gy, x, in_scale, out_scale, p, stream);    // 3x3 / s1: tap-fused, rolling window
    // (an 8-wave 256 x 128 tile, the winner of the forward family, measured 22 % SLOWER here: half the blocks for the split-K)
    if (p->Cout > 64) return launch_bf16_wgrad_cfg<2, 2, 2, 2>(gw, gy, x, in_scale, out_scale, p, stream);   // 128 (o) x 128 (k)
    if (p->Cout > 32) return launch_bf16_wgrad_cfg<2, 2, 1, 2>(gw, gy, x, in_scale, out_scale, p, stream);   // 64 x 128
    return launch_bf16_wgrad_cfg<1, 4, 1, 1>(gw, gy, x, in_scale, out_scale, p, stream);                     // 32 x 128
}
