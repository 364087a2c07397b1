// Equalised-lr linear layers  y = scale * (x @ W^T) + bias_mul * b  (EqualLinear, stylegan2/model.py:152-160) for MANY layers
// that share their input, in one launch per direction.
//
// Every linear layer of the path is skinny: M = the batch (32 texture codes, 96 images, 256 patches), K = 2048 .. 8192, N = 8 .. 512.
// The vendor GEMM runs them as 8 workgroups of a 64 x 64 macro tile (139 us for [32 x 2048] . [2048 x 512] on 256 CUs), and the
// generator asks for SIXTEEN of them per pass -- the modulation layers of its StyledConvs (stylegan2/model.py:226,239), all applied to
// the same texture code -- plus two per layer in the backward (d texture, d weight) and the autograd adds that sum the sixteen
// d texture.  Here a "segment" is one layer (its own weight, bias, output, gradient buffers), a launch takes a table of segments:
//     ideas_linear_fwd     y_s[m, j]   = scale_s * sum_k x[m, k] W_s[j, k] + bias_mul_s * b_s[j]                for every segment s
//     ideas_linear_bwd_x   gx[m, k]    = sum_s scale_s * sum_j g_s[m, j] W_s[j, k]                               (one tensor: the sum)
//     ideas_linear_bwd_w   gW_s[j, k] (+)= scale_s * sum_m g_s[m, j] x[m, k];   gb_s[j] (+)= bias_mul_s * sum_m g_s[m, j]
// All three are HBM-bound on reading (or read-modify-writing) the weights once: 43 MB for the generator's sixteen layers.
// Arithmetic: v_mfma_f32_32x32x2_f32, an exact f32 fmaf chain (the matrix pipe only because M = 32 is its native tile; the rate is
// the vector rate and irrelevant here).  Reductions are split over waves / blocks in a FIXED order and folded through LDS or a
// caller-provided partial buffer: no atomics, results are reproducible run to run.
#include "common.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int MAXSEG = IDEAS_LINEAR_MAX_SEGMENTS;

struct SegTable { ideas_linear_seg s[MAXSEG]; int n; };

// block-uniform: the segment whose tile range holds `tile` (tile0 = exclusive prefix over the segments' 32-row tiles)
__device__ __forceinline__ int find_seg(const SegTable& t, int tile) {
    int s = 0;
#pragma unroll 1
    for (int i = 1; i < t.n; ++i) s = (t.s[i].tile0 <= tile) ? i : s;
    return s;
}

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// ---------------------------------------------------------------------------------------------------------------------------
// forward.  Block = 4 waves on ONE 32 (m) x 32 (j) output tile, each wave a quarter of K (8-column steps); lane (i, h) loads 16 bytes
// of x row m0 + i and of W row j0 + i at column k + 4 h and feeds four MFMAs with (x[q], w[q]) -- A and B share the K permutation.
// The four partial tiles meet in LDS in wave order.
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void linear_fwd_kernel(SegTable t, const float* __restrict__ x, int M, int K, int ldx, int mtiles) {
    __shared__ float part[4][32][33];
    const int tile = blockIdx.x / mtiles, mt = blockIdx.x - tile * mtiles;
    const int si = find_seg(t, tile);
    const ideas_linear_seg sg = t.s[si];
    const int j0 = (tile - sg.tile0) * 32, m0 = mt * 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int steps = K >> 3;
    const int s0 = wave * steps / 4, s1 = (wave + 1) * steps / 4;
    const int mr = min(m0 + li, M - 1), jr = min(j0 + li, sg.n - 1);
    const float* xp = x + (int64_t)mr * ldx + 4 * lh;
    const float* wp = sg.w + (int64_t)jr * sg.ldw + 4 * lh;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    int s = s0;
    for (; s + 1 < s1; s += 2) {                     // two steps in flight per iteration
        const float4 xa = ldg4(xp + 8 * s), wa = ldg4(wp + 8 * s);
        const float4 xb = ldg4(xp + 8 * s + 8), wb = ldg4(wp + 8 * s + 8);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.x, wa.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.y, wa.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.z, wa.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.w, wa.w, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xb.x, wb.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xb.y, wb.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xb.z, wb.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xb.w, wb.w, acc, 0, 0, 0);
    }
    if (s < s1) {
        const float4 xa = ldg4(xp + 8 * s), wa = ldg4(wp + 8 * s);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.x, wa.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.y, wa.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.z, wa.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.w, wa.w, acc, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) part[wave][(e & 3) + 8 * (e >> 2) + 4 * lh][li] = acc[e];        // [m][j]
    __syncthreads();
    // thread -> (m = t / 8 .. , 4 consecutive j): 1024 outputs, 4 per thread
    const int r = threadIdx.x >> 3, c = (threadIdx.x & 7) * 4;
    const int m = m0 + r;
    if (m < M) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = j0 + c + q;
            if (j < sg.n) {
                float v = ((part[0][r][c + q] + part[1][r][c + q]) + part[2][r][c + q]) + part[3][r][c + q];
                v = mul_rn(v, sg.scale);
                if (sg.bias) v += mul_rn(sg.bias[j], sg.bias_mul);
                sg.y[(int64_t)m * sg.ldy + j] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// input gradient.  The reduction runs over the concatenated output axis of all segments in groups of 8 rows of W (n_s % 8 == 0).
// Block = (32-row m tile, 512-column k slab, split sp of the groups); wave w owns columns [128 w, 128 w + 128) of the slab: per
// group lane (i, h) loads 16 bytes of g row m0 + i at j + 4 h (A) and, for q = 0..3, 16 bytes of W row j + 4 h + q at column
// k0 + 4 i (B: four column blocks per load) -> 16 MFMAs.  Partial tiles go to part[sp][m][k]; linear_fold_kernel adds the splits.
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void linear_bwd_x_kernel(SegTable t, float* __restrict__ part, int M, int K, int mtiles, int kslabs,
                                                           int splits, int groups) {
    int b = blockIdx.x;
    const int sp = b % splits; b /= splits;
    const int ks = b % kslabs, mt = b / kslabs;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int m0 = mt * 32, k0 = ks * 512 + wave * 128;
    const int g0 = (int)((int64_t)sp * groups / splits), g1 = (int)((int64_t)(sp + 1) * groups / splits);
    const int mr = min(m0 + li, M - 1);
    const bool kok = k0 + 4 * li < K;                 // (K % 4 == 0: a lane's four columns are in or out together)
    const int kc = kok ? k0 + 4 * li : 0;
    f32x16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
    // segment of group g0 (tile0 here = exclusive prefix over the segments' 8-row GROUPS, set by the launcher)
    int si = find_seg(t, g0);
    for (int g = g0; g < g1; ++g) {
        while (si + 1 < t.n && t.s[si + 1].tile0 <= g) ++si;
        const ideas_linear_seg& sg = t.s[si];
        const int j = (g - sg.tile0) * 8;
        float4 a = ldg4(sg.y + (int64_t)mr * sg.ldy + j + 4 * lh);
        a.x *= sg.scale; a.y *= sg.scale; a.z *= sg.scale; a.w *= sg.scale;
        const float* wr = sg.w + (int64_t)(j + 4 * lh) * sg.ldw + kc;
        const float4 w0 = ldg4(wr), w1 = ldg4(wr + sg.ldw), w2 = ldg4(wr + 2 * (int64_t)sg.ldw), w3 = ldg4(wr + 3 * (int64_t)sg.ldw);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, w0.x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, w0.y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, w0.z, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, w0.w, acc[3], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, w1.x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, w1.y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, w1.z, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, w1.w, acc[3], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, w2.x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, w2.y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, w2.z, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, w2.w, acc[3], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, w3.x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, w3.y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, w3.z, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, w3.w, acc[3], 0, 0, 0);
    }
    // acc[c][e]: row m0 + (e&3) + 8 (e>>2) + 4 lh, column k0 + 4 li + c  -> one 16-byte store per row
    if (kok) {
        float* pp = part + ((int64_t)sp * (mtiles * 32) + m0) * K + k0 + 4 * li;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = (e & 3) + 8 * (e >> 2) + 4 * lh;
            *reinterpret_cast<float4*>(pp + (int64_t)r * K) = make_float4(acc[0][e], acc[1][e], acc[2][e], acc[3][e]);
        }
    }
}

// gx[m, k] = sum over the splits, in split order
__global__ __launch_bounds__(256) void linear_fold_kernel(float* __restrict__ gx, const float* __restrict__ part, int M, int K, int ldgx,
                                                          int mrows, int splits) {
    const int64_t n4 = (int64_t)M * (K >> 2);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int m = (int)(i / (K >> 2)), k = (int)(i % (K >> 2)) * 4;
        float4 s = ldg4(part + (int64_t)m * K + k);
        for (int sp = 1; sp < splits; ++sp) {
            const float4 v = ldg4(part + ((int64_t)sp * mrows + m) * K + k);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        *reinterpret_cast<float4*>(gx + (int64_t)m * ldgx + k) = s;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// weight (and bias) gradient.  Block = one 32-row j tile of one segment x 512 columns; wave w owns 128 of them.  The reduction
// runs over ALL m (two rows per MFMA): lane (i, h) loads g[m + h][j0 + i] (A: coalesced along j) and 16 bytes of x[m + h] at column
// k0 + 4 i (B: four column blocks).  The tile is read-modify-written (accumulate) or stored; wave 0 of the first slab also folds
// its A values into the bias gradient.
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void linear_bwd_w_kernel(SegTable t, const float* __restrict__ x, int M, int K, int ldx, int kslabs,
                                                           int accumulate) {
    const int tile = blockIdx.x / kslabs, ks = blockIdx.x - tile * kslabs;
    const int si = find_seg(t, tile);
    const ideas_linear_seg sg = t.s[si];
    const int j0 = (tile - sg.tile0) * 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int k0 = ks * 512 + wave * 128;
    const bool kok = k0 + 4 * li < K;
    const int kc = kok ? k0 + 4 * li : 0;
    const int jr = min(j0 + li, sg.n - 1);
    f32x16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
    float bsum = 0.f;
    const float* gp = sg.y + jr;
    for (int m = 0; m < M; m += 2) {
        const int mm = m + lh;
        const bool ok = mm < M;                       // (odd M: the last pair's second row contributes zeros)
        const int mc = ok ? mm : 0;
        float a = gp[(int64_t)mc * sg.ldy];
        a = ok ? a : 0.f;
        const float4 xv = ldg4(x + (int64_t)mc * ldx + kc);
        bsum += a;
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xv.x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xv.y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xv.z, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xv.w, acc[3], 0, 0, 0);
    }
    // acc[c][e]: row j0 + (e&3) + 8 (e>>2) + 4 lh, column k0 + 4 li + c
    if (kok) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int j = j0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            if (j < sg.n) {
                float* q = sg.gw + (int64_t)j * sg.ldgw + k0 + 4 * li;
                float4 v = make_float4(mul_rn(acc[0][e], sg.scale), mul_rn(acc[1][e], sg.scale), mul_rn(acc[2][e], sg.scale),
                                       mul_rn(acc[3][e], sg.scale));
                if (accumulate) {
                    const float4 o = ldg4(q);
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                }
                *reinterpret_cast<float4*>(q) = v;
            }
        }
    }
    if (sg.gb && ks == 0 && wave == 0) {
        bsum += __shfl_xor(bsum, 32, 64);
        if (lh == 0 && j0 + li < sg.n) {
            const float v = mul_rn(bsum, sg.bias_mul);
            sg.gb[j0 + li] = accumulate ? sg.gb[j0 + li] + v : v;
        }
    }
}

int check_segs(const ideas_linear_seg* segs, int nseg, SegTable& t, int unit, bool need_w) {
    if (!segs) return IDEAS_E_NULL;
    if (nseg <= 0 || nseg > MAXSEG) return IDEAS_E_SHAPE;
    int tiles = 0;
    for (int i = 0; i < nseg; ++i) {
        t.s[i] = segs[i];
        if (segs[i].n <= 0) return IDEAS_E_SHAPE;
        if (need_w && (!segs[i].w || !ideas_aligned16(segs[i].w) || segs[i].ldw % 4)) return segs[i].w ? IDEAS_E_ALIGN : IDEAS_E_NULL;
        if (!segs[i].y) return IDEAS_E_NULL;
        t.s[i].tile0 = tiles;
        tiles += (segs[i].n + unit - 1) / unit;
    }
    t.n = nseg;
    return tiles;
}

}  // namespace

extern "C" int ideas_sizeof_linear_seg(void) { return (int)sizeof(ideas_linear_seg); }

extern "C" int ideas_linear_fwd(const ideas_linear_seg* segs, int nseg, const void* x, int M, int K, int ldx, void* stream) {
    SegTable t;
    const int tiles = check_segs(segs, nseg, t, 32, true);
    if (tiles < 0) return tiles;
    if (!x) return IDEAS_E_NULL;
    if (M <= 0 || K <= 0) return IDEAS_E_SHAPE;
    if (K % 8 || ldx % 4 || !ideas_aligned16(x)) return IDEAS_E_ALIGN;
    const int mtiles = (M + 31) / 32;
    hipLaunchKernelGGL(linear_fwd_kernel, dim3((unsigned)(tiles * mtiles)), dim3(256), 0, (hipStream_t)stream, t, (const float*)x, M, K,
                       ldx, mtiles);
    return ideas_launch_status();
}

extern "C" int64_t ideas_linear_bwd_x_workspace(int total_n, int M, int K) {
    if (total_n <= 0 || M <= 0 || K <= 0) return 0;
    const int groups = total_n / 8;
    const int mtiles = (M + 31) / 32, kslabs = (K + 511) / 512;
    int splits = 512 / (mtiles * kslabs);             // ~2 blocks per CU
    splits = splits < 1 ? 1 : splits > groups ? groups : splits > 64 ? 64 : splits;
    return (int64_t)splits * mtiles * 32 * K * 4;
}

extern "C" int ideas_linear_bwd_x(const ideas_linear_seg* segs, int nseg, void* gx, int M, int K, int ldgx, void* workspace,
                                  int64_t workspace_bytes, void* stream) {
    SegTable t;
    const int groups = check_segs(segs, nseg, t, 8, true);
    if (groups < 0) return groups;
    if (!gx || !workspace) return IDEAS_E_NULL;
    if (M <= 0 || K <= 0) return IDEAS_E_SHAPE;
    int total = 0;
    for (int i = 0; i < nseg; ++i) {
        if (segs[i].n % 8 || segs[i].ldy % 4 || !ideas_aligned16(segs[i].y)) return IDEAS_E_ALIGN;
        total += segs[i].n;
    }
    if (K % 4 || ldgx % 4 || !ideas_aligned16(gx) || !ideas_aligned16(workspace)) return IDEAS_E_ALIGN;
    const int64_t need = ideas_linear_bwd_x_workspace(total, M, K);
    if (workspace_bytes < need) return IDEAS_E_SHAPE;
    const int mtiles = (M + 31) / 32, kslabs = (K + 511) / 512;
    const int splits = (int)(need / ((int64_t)mtiles * 32 * K * 4));
    hipLaunchKernelGGL(linear_bwd_x_kernel, dim3((unsigned)(mtiles * kslabs * splits)), dim3(256), 0, (hipStream_t)stream, t,
                       (float*)workspace, M, K, mtiles, kslabs, splits, groups);
    const int64_t n4 = (int64_t)M * (K / 4);
    const unsigned fb = (unsigned)(n4 / 256 + 1 < 1024 ? n4 / 256 + 1 : 1024);
    hipLaunchKernelGGL(linear_fold_kernel, dim3(fb), dim3(256), 0, (hipStream_t)stream, (float*)gx, (const float*)workspace, M, K, ldgx,
                       mtiles * 32, splits);
    return ideas_launch_status();
}

extern "C" int ideas_linear_bwd_w(const ideas_linear_seg* segs, int nseg, const void* x, int M, int K, int ldx, int accumulate,
                                  void* stream) {
    SegTable t;
    const int tiles = check_segs(segs, nseg, t, 32, false);
    if (tiles < 0) return tiles;
    if (!x) return IDEAS_E_NULL;
    if (M <= 0 || K <= 0) return IDEAS_E_SHAPE;
    for (int i = 0; i < nseg; ++i) {
        if (!segs[i].gw) return IDEAS_E_NULL;
        if (segs[i].ldgw % 4 || !ideas_aligned16(segs[i].gw)) return IDEAS_E_ALIGN;
    }
    if (K % 4 || ldx % 4 || !ideas_aligned16(x)) return IDEAS_E_ALIGN;
    const int kslabs = (K + 511) / 512;
    hipLaunchKernelGGL(linear_bwd_w_kernel, dim3((unsigned)(tiles * kslabs)), dim3(256), 0, (hipStream_t)stream, t, (const float*)x, M, K,
                       ldx, kslabs, accumulate);
    return ideas_launch_status();
}
