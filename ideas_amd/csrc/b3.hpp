// Shared pieces of the split-bf16 ("b3") kernels: exact 3-way f32 -> bf16 split, buffer loads, plane-pair order.
#pragma once
#include "common.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int BK = 16;     // f32 K depth of one pipeline step = K of one bf16 MFMA
constexpr int ROWB = 32;   // bytes per LDS row (16 bf16)
constexpr unsigned RSRC_FLAGS = 0x00020000u;   // raw buffer, 32-bit data format

// two f32 -> one dword of two RNE bf16 (v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo_f32(unsigned pk) { return __builtin_bit_cast(float, pk << 16); }
__device__ __forceinline__ float hi_f32(unsigned pk) { return __builtin_bit_cast(float, pk & 0xffff0000u); }

// exact three-way split of four f32 into packed bf16 planes
struct Split4 { uint2 p[3]; };
__device__ __forceinline__ void split2(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = pack_bf16(a, b);
    const float ra = a - lo_f32(h), rb = b - hi_f32(h);
    m = pack_bf16(ra, rb);
    l = pack_bf16(ra - lo_f32(m), rb - hi_f32(m));   // exact: <= 8 significant bits are left
}
// the same from doubles (operands that are sums of f32 values, e.g. Winograd-transformed weights): the residuals are carried in
// double, so the three planes hold the leading ~26 bits of the EXACT value instead of those of its f32 rounding
__device__ __forceinline__ void split2d(double a, double b, unsigned& h, unsigned& m, unsigned& l) {
    h = pack_bf16((float)a, (float)b);
    const double ra = a - (double)lo_f32(h), rb = b - (double)hi_f32(h);
    m = pack_bf16((float)ra, (float)rb);
    l = pack_bf16((float)(ra - (double)lo_f32(m)), (float)(rb - (double)hi_f32(m)));
}
__device__ __forceinline__ Split4 split4(float4 v) {
    Split4 s;
    split2(v.x, v.y, s.p[0].x, s.p[1].x, s.p[2].x);
    split2(v.z, v.w, s.p[0].y, s.p[1].y, s.p[2].y);
    return s;
}

// (bit_cast the WHOLE result: indexing the builtin's return value element-wise makes the optimizer shrink the load to
// one dword and splat it)
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 buffer_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const f32x4 f = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
    return make_float4(f.x, f.y, f.z, f.w);
}

// plane pairs, smallest terms first
constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
constexpr int PB[6] = {0, 2, 1, 0, 1, 0};

}  // namespace
