// Experiment (not part of the library): is a 3-way bf16 split of f32 operands, contracted with six
// v_mfma_f32_32x32x16_bf16 products per tile, (a) f32-class accurate and (b) faster than the exact
// v_mfma_f32_32x32x2_f32 path on gfx950?   C[M,N] = A[M,K] · B[N,K]^T, f32 in / f32 out.
//
//   x = hi + mid + lo exactly (hi = rne_bf16(x), mid = rne_bf16(x-hi), lo = x-hi-mid, 8 significand bits each)
//   x·y ≈ hi·hi + hi·mid + mid·hi + mid·mid + hi·lo + lo·hi      (dropped terms ≤ 2^-25 |x·y| each)
//
// build: hipcc -O3 --offload-arch=gfx950 tools/gemm_bf16x3.hip -o tools/gemm_bf16x3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int BM = 128, BN = 128, BK = 32, LDK = BK + 8;  // 80-byte rows: conflict-free ds_read_b128

struct Split4 { bf16x4 p[3]; };

__device__ __forceinline__ Split4 split4(float4 v) {
    Split4 s;
    float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        __bf16 h = (__bf16)x[i];
        float r1 = x[i] - (float)h;
        __bf16 m = (__bf16)r1;
        float r2 = r1 - (float)m;
        s.p[0][i] = h; s.p[1][i] = m; s.p[2][i] = (__bf16)r2;
    }
    return s;
}

template <int NPROD>
__global__ __launch_bounds__(256, 2) void gemm_split(const float* __restrict__ A, const float* __restrict__ B,
                                                     float* __restrict__ C, int M, int N, int K, int lda) {
    __shared__ __attribute__((aligned(16))) __bf16 sA[3][BM][LDK];
    __shared__ __attribute__((aligned(16))) __bf16 sB[3][BN][LDK];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;
    const int nbn = N / BN;
    const int nblk = gridDim.x, q8 = nblk >> 3, xcd = blockIdx.x & 7;   // XCD-contiguous bands of M tiles (nblk % 8 == 0)
    const int swz = xcd * q8 + (blockIdx.x >> 3);
    const int m0 = (swz / nbn) * BM, n0 = (swz % nbn) * BN;
    const int lr = t >> 3, lq = t & 7;  // staging: row lr + 32 i, float4 column lq

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[4], rb[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = *reinterpret_cast<const float4*>(A + (size_t)(m0 + lr + 32 * i) * lda + k0 + lq * 4);
            rb[i] = *reinterpret_cast<const float4*>(B + (size_t)(n0 + lr + 32 * i) * K + k0 + lq * 4);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            Split4 a = split4(ra[i]), b = split4(rb[i]);
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                *reinterpret_cast<bf16x4*>(&sA[p][lr + 32 * i][lq * 4]) = a.p[p];
                *reinterpret_cast<bf16x4*>(&sB[p][lr + 32 * i][lq * 4]) = b.p[p];
            }
        }
    };

    gload(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        lstore();
        __syncthreads();
        if (k0 + BK < K) gload(k0 + BK);
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            bf16x8 a[2][3], b[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    a[i][p] = *reinterpret_cast<const bf16x8*>(&sA[p][wm * 64 + i * 32 + (lane & 31)][kk * 16 + (lane >> 5) * 8]);
                    b[i][p] = *reinterpret_cast<const bf16x8*>(&sB[p][wn * 64 + i * 32 + (lane & 31)][kk * 16 + (lane >> 5) * 8]);
                }
            // small terms first, the hi·hi product last
            constexpr int PA[9] = {2, 0, 1, 1, 0, 0, 1, 2, 2};
            constexpr int PB[9] = {0, 2, 1, 0, 1, 0, 2, 1, 2};
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                const bool use = (NPROD == 9) || (q < 6 && (NPROD == 6 || q >= 6 - NPROD));
                if (!use) continue;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[q]], b[j][PB[q]], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = n0 + wn * 64 + j * 32 + (lane & 31);
                C[(size_t)row * N + col] = acc[i][j][r];
            }
}

// exact-f32 MFMA comparison kernel with the same tiling (v_mfma_f32_32x32x2_f32)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 3))) void gemm_f32(const float* __restrict__ A, const float* __restrict__ B,
                                                   float* __restrict__ C, int M, int N, int K, int lda) {
    constexpr int LDF = BK + 4;
    __shared__ __attribute__((aligned(16))) float sA[BM][LDF];
    __shared__ __attribute__((aligned(16))) float sB[BN][LDF];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;
    const int nbn = N / BN;
    const int nblk = gridDim.x, q8 = nblk >> 3, xcd = blockIdx.x & 7;   // XCD-contiguous bands of M tiles (nblk % 8 == 0)
    const int swz = xcd * q8 + (blockIdx.x >> 3);
    const int m0 = (swz / nbn) * BM, n0 = (swz % nbn) * BN;
    const int lr = t >> 3, lq = t & 7;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 ra[4], rb[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = *reinterpret_cast<const float4*>(A + (size_t)(m0 + lr + 32 * i) * lda + k0 + lq * 4);
            rb[i] = *reinterpret_cast<const float4*>(B + (size_t)(n0 + lr + 32 * i) * K + k0 + lq * 4);
        }
    };
    gload(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float4*>(&sA[lr + 32 * i][lq * 4]) = ra[i];
            *reinterpret_cast<float4*>(&sB[lr + 32 * i][lq * 4]) = rb[i];
        }
        __syncthreads();
        if (k0 + BK < K) gload(k0 + BK);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = sA[wm * 64 + i * 32 + (lane & 31)][kk + (lane >> 5)];
                b[i] = sB[wn * 64 + i * 32 + (lane & 31)][kk + (lane >> 5)];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = n0 + wn * 64 + j * 32 + (lane & 31);
                C[(size_t)row * N + col] = acc[i][j][r];
            }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 32768, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 4608;
    const int lda = argc > 4 ? atoi(argv[4]) : K;
    const int reps = 10, RS = 48;  // rows sampled for the f64 check
    std::vector<float> hA((size_t)M * lda + K), hB((size_t)N * K), hC((size_t)M * N);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f; };
    auto gauss = [&]() { float u = 0; for (int i = 0; i < 6; ++i) u += rnd(); return (u - 3.0f) * 1.414f; };
    for (auto& v : hA) v = gauss() * (1.0f + 3.0f * rnd());  // mixed sign, a few octaves of magnitude
    for (auto& v : hB) v = gauss() * 0.05f;
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, hC.size() * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    // f64 truth and the abs-dot normaliser on sampled rows
    std::vector<double> ref((size_t)RS * N), nrm((size_t)RS * N);
    std::vector<int> rows(RS);
    for (int i = 0; i < RS; ++i) rows[i] = (int)(((long long)i * 2654435761LL) % M);
    for (int i = 0; i < RS; ++i)
        for (int n = 0; n < N; ++n) {
            double acc = 0, na = 0;
            const float* a = &hA[(size_t)rows[i] * lda]; const float* b = &hB[(size_t)n * K];
            for (int k = 0; k < K; ++k) { double p = (double)a[k] * (double)b[k]; acc += p; na += fabs(p); }
            ref[(size_t)i * N + n] = acc; nrm[(size_t)i * N + n] = na;
        }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 grid((M / BM) * (N / BN)), block(256);
    const double flop = 2.0 * M * N * K;
    auto run = [&](const char* name, auto launch) {
        CK(hipMemset(dC, 0, hC.size() * 4));
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
        double emax = 0, erms = 0;
        for (int i = 0; i < RS; ++i)
            for (int n = 0; n < N; ++n) {
                double e = fabs((double)hC[(size_t)rows[i] * N + n] - ref[(size_t)i * N + n]) / nrm[(size_t)i * N + n];
                emax = fmax(emax, e); erms += e * e;
            }
        printf("%-14s %8.3f ms  %7.1f TF/s (algorithmic)   err/absdot: max %.3e rms %.3e\n", name, ms, flop / ms * 1e-9, emax,
               sqrt(erms / (RS * (double)N)));
    };
    printf("M=%d N=%d K=%d lda=%d\n", M, N, K, lda);
    run("f32 mfma", [&] { hipLaunchKernelGGL(gemm_f32, grid, block, 0, 0, dA, dB, dC, M, N, K, lda); });
    run("bf16x3 / 9", [&] { hipLaunchKernelGGL(gemm_split<9>, grid, block, 0, 0, dA, dB, dC, M, N, K, lda); });
    run("bf16x3 / 6", [&] { hipLaunchKernelGGL(gemm_split<6>, grid, block, 0, 0, dA, dB, dC, M, N, K, lda); });
    run("bf16x3 / 3", [&] { hipLaunchKernelGGL(gemm_split<3>, grid, block, 0, 0, dA, dB, dC, M, N, K, lda); });
    run("bf16 / 1", [&] { hipLaunchKernelGGL(gemm_split<1>, grid, block, 0, 0, dA, dB, dC, M, N, K, lda); });
    return 0;
}
