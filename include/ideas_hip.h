/*
 * ideas_hip.h — C ABI of libideas_hip.so, the MI355X (gfx950) kernels behind the IDEAS hot path.
 *
 * Drop-in boundary.  The reference binds two pybind11 modules whose single entry
 * points take/return torch::Tensor:
 *     fused.fused_bias_act(input, bias, refer, act, grad, alpha, scale)      stylegan2/op/fused_bias_act.cpp:11-21
 *     upfirdn2d_op.upfirdn2d(input, kernel, up_x, up_y, down_x, down_y,
 *                            pad_x0, pad_x1, pad_y0, pad_y1)                   stylegan2/op/upfirdn2d.cpp:12-23
 * and leaves every convolution to cuDNN through ATen (stylegan2/model.py:115-121,258,273; models.py:32-38).
 * This library replaces all of them with plain-C entry points:
 *
 *   - raw device pointers, int dims, a dtype/layout enum and a hipStream_t (passed as void*);
 *   - the CALLER allocates every output (so memory stays with its own allocator);
 *   - no global state, no allocation, no host synchronisation: every call only enqueues on `stream`.  The only thing a call reads
 *     besides its arguments is a handful of A/B measurement switches in the process environment, looked up PER CALL (nothing is
 *     cached in the library): IDEAS_B3_WINO2D, IDEAS_B3_TPHASE, IDEAS_B3_WGRAD3, IDEAS_B3_WGRAD3_S2, IDEAS_S2FIR_CFG,
 *     IDEAS_BF16_IMG, IDEAS_BF16_WGRAD3, IDEAS_B3_PW, IDEAS_B3_PW_WGRAD, IDEAS_BF16_PW, IDEAS_B3_WINO_EPI, IDEAS_S2IMG_MIN_BLOCKS
 *     -- each "0" selects the older kernel of its family, unset = the default dispatch.
 *     What IS memoised per process: immutable device properties (the CU count and the occupancy hipOccupancy... reports for the
 *     library's own split-K kernels), used to size grids;
 *   - return value: 0 = enqueued; negative = argument error (IDEAS_E_*); positive = hipError_t of the launch.
 *
 * Layouts.  IDEAS_NCHW is the reference's layout (bias index (i / inner) % C,
 * fused_bias_act_kernel.cu:29).  IDEAS_NHWC is the MI355X fast path: channels innermost, so bias index is
 * i % C, every load is a coalesced 16-byte vector and the 3x3 contraction's K axis (ky,kx,ci) is contiguous.
 * The convolution entry points are NHWC-only.
 */
#ifndef IDEAS_HIP_H
#define IDEAS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IDEAS_ABI_VERSION 4   /* 2: the per-sample reductions of the modulated-conv backward accumulate in double (round 3);
                                 3: ideas_demod_bwd overwrites dot_d, ideas_weight_prep_batched added;
                                 4: the entry points added since 3 are REQUIRED by the Python binding -- ideas_patch_resize,
                                    ideas_image_u8_to_f32, ideas_stream_create, ideas_linear_fwd / _bwd_x / _bwd_w +
                                    ideas_sizeof_linear_seg (minimum version for the EqualLinear API), and what round 6 adds
                                    (see the end of this header).  A binding checks `ideas_abi_version() >= 4` before it looks
                                    any of them up. */

enum { IDEAS_NCHW = 0, IDEAS_NHWC = 1 };
/* `dtype` of the convolution entry points.  Tensors are f32 in HBM for both values.
 *   IDEAS_F32     contraction on the f32 matrix instruction (v_mfma_f32_32x32x2_f32): an exact fmaf chain.
 *   IDEAS_F32_B3  operands split exactly into three bf16 planes, six plane-pair products on the bf16 matrix
 *                 instruction with f32 accumulation (csrc/conv_b3.hip): the same f32 error class at 2.67x the matrix
 *                 rate.  Shapes the split kernels do not cover (Cin % 16 != 0) run the IDEAS_F32 kernel.
 *   IDEAS_BF16    bf16 mixed precision (BASELINE.json configs[4]; the reference's ops dispatch half as well,
 *                 fused_bias_act_kernel.cu:78, upfirdn2d_kernel.cu:311): activations (x, y, gy, resid) are bf16 in HBM,
 *                 contraction on v_mfma_f32_32x32x16_bf16 with f32 accumulation and an f32 epilogue; weights stay f32 masters
 *                 (the MFMA kernels read a bf16 pack of them, ideas_bf16_pack_weights), scales / biases / weight gradients f32. */
enum { IDEAS_F32 = 0, IDEAS_F32_B3 = 1, IDEAS_BF16 = 2 };

enum {
    IDEAS_OK = 0,
    IDEAS_E_NULL = -1,      /* a required pointer is NULL                 */
    IDEAS_E_SHAPE = -2,     /* a dimension is <= 0 or inconsistent        */
    IDEAS_E_UNSUPPORTED = -3, /* dtype / layout / mode not implemented   */
    IDEAS_E_ALIGN = -4      /* pointer or channel count not 16-B aligned where the kernel needs it */
};

int ideas_abi_version(void);
int ideas_sizeof_conv_params(void);   /* lets a binding check its mirror of ideas_conv_params */
const char* ideas_strerror(int code);

/* Stream helper (no reference counterpart: the reference runs everything on torch's current stream).  Creates a non-blocking HIP
 * stream of the lowest (prio < 0), default (prio == 0) or highest (prio > 0) priority of the device; the caller owns it
 * (ideas_stream_destroy).  The host side runs the weight-gradient kernels on a lowest-priority stream so that they fill what the
 * input-gradient chain on the caller's stream leaves free instead of competing with it for compute units. */
int ideas_stream_create(void** out_stream, int prio);
int ideas_stream_destroy(void* stream);

/* ------------------------------------------------------------------------------------------------
 * fused bias + leaky-ReLU.  Replaces fused_bias_act_op (fused_bias_act_kernel.cu:52-98).
 *
 *   v   = x[i] + (b ? b[channel(i)] : 0)
 *   y   = (grad == 0) ?  (v   > 0 ? v : v * alpha) * scale            -- act=3, grad=0  (forward)
 *       : (grad == 1) ?  (ref > 0 ? v : v * alpha) * scale            -- act=3, grad=1  (backward / grad-grad)
 *       :                0                                            -- grad == 2
 *   act == 1 is the linear variant (y = v * scale; 0 for grad == 2).
 *
 * `n` = total elements, `C` = channels, `inner` = product of dims after the channel dim (NCHW only; a
 * [B,C] matrix is inner = 1).  `ref` is required iff grad == 1.  `bias_grad` (optional, grad == 1 only):
 * if non-NULL it must be a ZEROED float[C]; the kernel adds sum over (n,h,w) of y into it, fusing the
 * reference's separate grad_input.sum(dim) pass (fused_act.py:33-38).
 * ---------------------------------------------------------------------------------------------- */
int ideas_fused_bias_act(void* y, const void* x, const void* b, const void* ref, float* bias_grad,
                         int64_t n, int C, int64_t inner, int layout,
                         int act, int grad, float alpha, float scale, int dtype, void* stream);

/* Per-channel sum of a channels-innermost tensor (x: n elements, NHWC or [B,C]; any C <= 8192): out[c] (+)= sum of x[.., c].
 * The gradient of a plain conv bias (EqualConv2d with bias and no activation = G.to_rgb, stylegan2/model.py:94-123,
 * models.py:120), which autograd takes as grad.sum((0, 2, 3)).  `clear` != 0 zeroes out[0..C) first.  (additive since ABI 3.) */
int ideas_channel_sum(float* out, const void* x, int64_t n, int C, int clear, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * upfirdn2d.  Replaces upfirdn2d_op (upfirdn2d_kernel.cu:209-368): zero-stuff by `up`, pad (negative pad
 * crops), correlate with the FLIPPED kh x kw FIR, decimate by `down`.  x is [B,C,in_h,in_w] (NCHW) or
 * [B,in_h,in_w,C] (NHWC); y has out = (in*up + pad0 + pad1 - k) / down + 1 per axis.  `fir` is a device
 * float[kh*kw], row-major, kh,kw <= 8.  `gain` multiplies the FIR (1.0f for the plain op).
 * The gradient is the same call with up<->down swapped, the FIR flipped and
 * pads (k - p0 - 1, in*up - out*down + p0 - up + 1)  (upfirdn2d.py:111-114).
 * ---------------------------------------------------------------------------------------------- */
int ideas_upfirdn2d(void* y, const void* x, const float* fir,
                    int B, int C, int in_h, int in_w, int out_h, int out_w,
                    int kh, int kw, int up_x, int up_y, int down_x, int down_y,
                    int pad_x0, int pad_y0, float gain, int flip,
                    int layout, int dtype, void* stream);

/* The 4x4 NHWC blur (up = down = 1; C % 4 == 0) with the elementwise pass that always follows it on the IDEAS path folded into
 * its store, so the blurred tensor is never written and read back:
 *   mode 1 (IDEAS_BLUR_ACT_BWD):  y = (ref > 0 ? v : v*alpha) * scale,  bias_grad[c] += sum over (b,h,w) of y
 *        v = the blur of x.  The gradient of a downsampling ConvLayer's Blur (models.py:69, upfirdn2d.py:19-45) followed by the
 *        leaky-ReLU backward of the conv layer below it (fused_act.py:20-47); `ref` = that layer's saved output, same shape and
 *        dtype as y; `bias_grad` float[C], accumulated into (not zeroed here).
 *   mode 2 (IDEAS_BLUR_BIAS_ACT): y = lrelu(v + bias[c], alpha) * scale
 *        the Blur of an upsampling ModulatedConv2d followed by its FusedLeakyReLU (stylegan2/model.py:258-261, 371-377).
 * Operation order is that of ideas_upfirdn2d -> ideas_fused_bias_act (for bf16 the blur is rounded to bf16 in between, as the
 * two-kernel path stores it), so f32 results are bitwise the unfused ones.  `flip` / `gain` / pads as in ideas_upfirdn2d. */
/* The zero-stuffing 4x4 NHWC FIR (up = 2, down = 1; C % 4 == 0) of ideas_upfirdn2d plus a tensor of the output's shape:
 * y = fir(x) + resid.  Forward: the residual merge behind an upsampling skip branch (models.py:160-178: (conv2(...) + skip) with
 * the skip ending in a Blur); backward: the sum of the two input gradients of a downsampling ResBlock, whose skip branch starts
 * with the decimating FIR this kernel is the adjoint of.  Saves the separate add pass (3 tensor passes). */
int ideas_fir_up2_add(void* y, const void* x, const float* fir, const void* resid, int B, int C, int in_h, int in_w, int out_h,
                      int out_w, int pad_x0, int pad_y0, float gain, int flip, int dtype, void* stream);

#define IDEAS_BLUR_ACT_BWD 1
#define IDEAS_BLUR_BIAS_ACT 2
int ideas_blur_fused(void* y, const void* x, const float* fir, int B, int C, int in_h, int in_w, int out_h, int out_w,
                     int pad_x0, int pad_y0, float gain, int flip, int mode, const void* ref, const float* bias,
                     float* bias_grad, float alpha, float scale, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution on the f32 MFMA pipe (v_mfma_f32_32x32x2_f32), NHWC.
 * Replaces the cuDNN calls behind F.conv2d / F.conv_transpose2d (stylegan2/model.py:115-121,258,273;
 * models.py:32-38) and, with in_scale/out_scale, the per-sample weight materialisation of
 * ModulatedConv2d (stylegan2/model.py:240-248):  y[b,:,:,o] = out_scale[b,o] * sum w[o,t,i] * in_scale[b,i] * x[b,..,i].
 *
 * One launch computes, for every point (b, oy, ox) of a logical OH x OW grid and every output channel o:
 *     acc = sum over taps (ty,tx) in [0,TY)x[0,TX) and ci in [0,Cin) of
 *              wmat[o][(ty*TX+tx)*Cin + ci] * X(b, oy*sy + ty*dy + offy, ox*sx + tx*dx + offx, ci)
 *     where X(...) is 0 outside [0,IH)x[0,IW) (or mirrored, reflect=1), times in_scale[b*Cin+ci] if given;
 *     y[b, oy*osy + ooy, ox*osx + oox, o] = epilogue(acc)
 * The output tensor is [B, YH, YW, Cout].  This one parameterisation covers forward convs (sy = stride,
 * dy = 1, off = -pad), input gradients of stride-1 convs (flipped/transposed wmat), and the four parity
 * phases of stride-2 transposed convs / stride-2 input gradients (osy = 2, ooy = phase).
 *
 * epilogue(acc): v = acc * gain * (out_scale ? out_scale[b*Cout+o] : 1) + (bias ? bias[o] : 0);
 *                if act: v = (v > 0 ? v : v*alpha) * act_gain;
 *                if resid: v = (v + resid[same index as y]) * resid_gain;
 *                if accumulate: y += v else y = v.
 * Requires Cin % 4 == 0 and 16-byte aligned x / wmat.
 * ---------------------------------------------------------------------------------------------- */
typedef struct ideas_conv_params {
    int B, IH, IW, Cin;          /* input  [B,IH,IW,Cin]                    */
    int YH, YW, Cout;            /* output tensor [B,YH,YW,Cout]            */
    int OH, OW;                  /* logical grid computed by this launch    */
    int TY, TX;                  /* taps                                    */
    int sy, sx, dy, dx, offy, offx;
    int osy, osx, ooy, oox;      /* output placement                        */
    int reflect;                 /* 1 = mirror out-of-range input coords    */
    int act;                     /* 1 = leaky-ReLU epilogue                 */
    float alpha, act_gain, resid_gain;
    int accumulate;
    float gain;                  /* uniform multiplier on the accumulator (equalised-lr weight scale) */
} ideas_conv_params;

int ideas_conv_igemm(void* y, const void* x, const void* wmat, const float* in_scale, const float* out_scale,
                     const float* bias, const void* resid, const ideas_conv_params* p, int dtype, void* stream);

/* Split-bf16 ("b3") operand preparation for dtype IDEAS_F32_B3 (csrc/conv_b3.hip).
 *   ideas_b3_conv_supported  1 if ideas_conv_igemm(..., IDEAS_F32_B3, ...) covers the geometry (Cin % 16 == 0, <= 32 taps,
 *                            x and the weight planes < 4 GiB: the kernel addresses them with 32-bit buffer offsets).
 *   ideas_b3_split_weights   wmat f32 [Cout][K] (the same matrix ideas_conv_igemm takes for IDEAS_F32) -> `planes`,
 *                            3*Cout*K bf16 laid out [3][K/16][Cout][16] with the K-steps
 *                            ordered (ci/16, ty, tx); K = taps*Cin, Cin % 16 == 0.  With IDEAS_F32_B3 the `wmat`
 *                            argument of ideas_conv_igemm is this buffer.  Activations are split inside the kernel. */
int ideas_b3_conv_supported(const ideas_conv_params* p);
int ideas_b3_wgrad_supported(const ideas_conv_params* p);   /* 1 if ideas_conv_wgrad(IDEAS_F32_B3) runs the split kernel; otherwise it
                                                                runs the IDEAS_F32 kernel (same arguments, same result class) */
int ideas_b3_wgrad3_supported(const ideas_conv_params* p);  /* 1 if that split weight gradient is the tap-fused 3x3 kernel with a rolling
                                                                activation window (csrc/conv_b3_wgrad3.hip: 3x3, stride 1 or 2,
                                                                OW % 16 == 0, Cin % 64 == 0, Cout % 64 == 0); informational */
int ideas_b3_split_weights(void* planes, const void* wmat, int Cout, int K, int Cin, void* stream);
/* The same planes read straight from a parameter of any strides: element (n, ty, tx, ci) of the launch's weight matrix is
 * w[n*sn + ty*sty + tx*stx + ci*sc] (floats; w already points at the first element).  Covers the forward matrix of an
 * [O,I,KH,KW] tensor in either memory format, the phase-sliced transposed matrix of an input gradient (tap step = the conv
 * stride) and the matrix of a transposed conv, without materialising a permuted / sliced f32 copy first. */
int ideas_b3_split_weights_strided(void* planes, const float* w, int Cout, int TY, int TX, int Cin, int64_t sn, int64_t sty,
                                   int64_t stx, int64_t sc, void* stream);
/* The same for ideas_conv3x3_wino(IDEAS_F32_B3): Winograd-transformed AND split weights, 12*N*3*C bf16 laid out
 * [4 v][3 planes][3*C/16 steps][N][16] (step = (c/16)*3 + ky).  Element (n, ky, kx, c) of the 3x3 kernel is read at
 * w[base + n*sn + ky*sky + kx*skx + c*sc] (floats), so the forward matrix (n = o, c = i) and the flipped / transposed one
 * of the input gradient (n = i, c = o) both come straight from the OHWI parameter. */
int ideas_b3_wino_supported(const ideas_conv_params* p);
int ideas_b3_wino_split_weights(void* planes, const void* w, int N, int C, int64_t sn, int64_t sky, int64_t skx, int64_t sc,
                                int64_t base, void* stream);

/* bf16 mixed precision (dtype IDEAS_BF16 of ideas_conv_igemm / ideas_conv_wgrad; csrc/conv_bf16.hip).
 *   ideas_bf16_conv_supported   1 if ideas_conv_igemm(..., IDEAS_BF16, ...) covers the geometry: Cin % 32 == 0, Cout % 4 == 0,
 *                               <= 32 taps, tensors < 4 GiB (`scaled`: an in_scale will be passed; M tiles are then cut per sample
 *                               and the block scales its weight tile by in_scale[b, :]).
 *   ideas_bf16_wgrad_supported  the same for ideas_conv_wgrad: Cin % 8 == 0, Cout % 8 == 0, OW a divisor or a multiple of 32.
 *   ideas_bf16_pack_weights     wmat f32 [Cout][K] (K = taps*Cin, the matrix ideas_conv_igemm takes for IDEAS_F32) -> `pack`,
 *                               Cout*K bf16 laid out [K/32][Cout][32], K-steps ordered (ci/32, ty, tx), 16-byte chunk c of row
 *                               n stored at position c ^ ((n >> 2) & 3) (the LDS swizzle, so the kernel's DMA is linear).
 *                               in_scale == NULL: one pack (B must be 1).  in_scale = float[B][Cin] (modulated conv): B packs,
 *                               pack b = bf16(w[n][k] * in_scale[b][ci(k)]) -- the reference's per-sample weights
 *                               (stylegan2/model.py:240-248) for the life of one launch.
 *                               With IDEAS_BF16 the `wmat` argument of ideas_conv_igemm is this buffer; pass the same in_scale
 *                               to ideas_conv_igemm (it selects the per-sample packs; the scale itself is already applied). */
int ideas_bf16_conv_supported(const ideas_conv_params* p, int scaled);
int ideas_bf16_wgrad_supported(const ideas_conv_params* p, int scaled);
int ideas_bf16_pack_weights(void* pack, const void* wmat, const float* in_scale, int B, int Cout, int K, int Cin, void* stream);
/* ideas_bf16_pack_weights reading the matrix through strides (see ideas_b3_split_weights_strided). */
int ideas_bf16_pack_weights_strided(void* pack, const float* w, const float* in_scale, int B, int Cout, int TY, int TX, int Cin,
                                    int64_t sn, int64_t sty, int64_t stx, int64_t sc, void* stream);
/* The three derived-weight forms above for MANY parameters in one launch per form (the forms are remade after every optimiser
 * step: one batched launch instead of ~100 small ones per form and network group).  `table` = n descriptors IN DEVICE MEMORY
 * (the caller builds them once: parameter and destination addresses do not change between steps), `op` selects the form, and
 * block0 / nblocks give each descriptor's share of the `total_blocks` 256-thread blocks (descriptors sorted by block0, first 0,
 * contiguous).  Per element the kernels run the single-tensor bodies: results are bitwise those of the calls they replace.
 *   IDEAS_PREP_B3_SPLIT   dst = planes of ideas_b3_split_weights_strided(w, Cout=a[0], TY=a[1], TX=a[2], Cin=a[3], s = sn,sty,stx,sc)
 *   IDEAS_PREP_B3_WINO    dst = planes of ideas_b3_wino_split_weights(w, N=a[0], C=a[1], s = sn,sky,skx,sc,base)
 *   IDEAS_PREP_BF16_PACK  dst = pack of ideas_bf16_pack_weights_strided(w, NULL, 1, Cout=a[0], TY=a[1], TX=a[2], Cin=a[3], s = ...) */
enum { IDEAS_PREP_B3_SPLIT = 0, IDEAS_PREP_B3_WINO = 1, IDEAS_PREP_BF16_PACK = 2 };
typedef struct ideas_prep_desc {
    void* dst;
    const float* w;
    int64_t s[5];
    int a[4];
    int unit;        /* 1: unit channel stride and 16-byte aligned rows (vector reads), as the single-tensor entry points decide */
    int block0;
    int nblocks;
    int pad_;
} ideas_prep_desc;
int ideas_sizeof_prep_desc(void);
int ideas_weight_prep_batched(const ideas_prep_desc* table, int n, int op, int total_blocks, void* stream);
/* 1 if ideas_conv_direct / ideas_conv_wgrad_direct with IDEAS_BF16 are the intended path for the geometry: the HBM-bound
 * pointwise layers with <= 8 input or output channels (from-RGB, to-RGB and their gradients).  Everything else without a bf16
 * MFMA kernel (a handful of tiny layers) is computed by the caller in f32 on casts. */
int ideas_bf16_direct_supported(const ideas_conv_params* p);

/* Weight gradient of the same family:  for every o, tap, ci
 *     gw[o][(ty*TX+tx)*Cin + ci] (+)= sum over (b,oy,ox) of  G(b,oy,ox,o) * X(b, iy, ix, ci)
 * with G = gy[b, oy*osy+ooy, ox*osx+oox, o] * (out_scale ? out_scale[b*Cout+o] : 1),
 *      X as above (zero / reflect outside, times in_scale).  gw must be ZEROED by the caller unless it is
 * meant to accumulate; split-K partial sums (times `gain`) are added with float atomics (order not
 * deterministic).
 * Requires Cin % 4 == 0 and Cout % 4 == 0.
 */
int ideas_conv_wgrad(float* gw, const void* gy, const void* x, const float* in_scale, const float* out_scale,
                     const ideas_conv_params* p, int dtype, void* stream);

/* 3x3, stride 1, padding 1 (zero or mirrored) convolution through a 1-D Winograd F(2,3) transform along x: four
 * GEMMs with K = 3*Cin instead of one with K = 9*Cin (1.5x fewer MFMAs), same epilogue as ideas_conv_igemm.
 * `umat` = transformed weights [4][Cout][3][Cin]:  U0 = w[..,kx=0], U1 = (w0+w1+w2)/2, U2 = (w0-w1+w2)/2, U3 = w[..,kx=2]
 * (for an input gradient pass the weights flipped in ky,kx with in/out channels swapped).  p must describe the plain
 * geometry (TY = TX = 3, s = 1, OH = IH, OW = IW); requires IW even and Cin % 8 == 0. */
int ideas_conv3x3_wino(void* y, const void* x, const void* umat, const float* in_scale, const float* out_scale,
                       const float* bias, const void* resid, const ideas_conv_params* p, int dtype, void* stream);

/* Weight gradient of the same layers in the Winograd domain: gu (ZEROED float [4][Cout][3][Cin]) receives
 * dU_v = gain * sum dM_v (x) V_v with dM = (g0, g0+g1, g0-g1, -g1); the caller folds it back to the 3x3 taps:
 * dw[kx=0] = dU0 + (dU1+dU2)/2, dw[kx=1] = (dU1-dU2)/2, dw[kx=2] = (dU1+dU2)/2 + dU3.  Cout % 4 == 0 as well. */
int ideas_conv3x3_wino_wgrad(float* gu, const void* gy, const void* x, const float* in_scale, const float* out_scale,
                             const ideas_conv_params* p, int dtype, void* stream);
/* (dtype IDEAS_F32 only: the split-bf16 form of this call, measured no faster than the direct split weight gradient in rounds 2-5,
 * left the library in round 6 -- tools/attic/conv_b3_wino_wgrad.hip.)
 * ideas_wino_wgrad_fold ADDS the folded taps to gw -- element (o, ky, kx, ci) at gw[o*so + ky*sky + kx*skx + ci*sc] (floats), so
 * the OHWI gradient of a parameter or a strided view of a flat gradient bucket -- and, with `clear`, re-zeroes dU behind the
 * read (a persistent scratch then never needs a fill launch).  Replaces the autograd weight gradient of the F.conv2d calls at
 * stylegan2/model.py:115-121 (EqualConv2d) and :262-277 (ModulatedConv2d, same-resolution branch) for the 3x3 layers. */
int ideas_wino_wgrad_fold(float* gw, float* gu, int Cout, int Cin, int64_t so, int64_t sky, int64_t skx, int64_t sc, int clear,
                          void* stream);

/* Up to four launches of ideas_conv_igemm(IDEAS_F32_B3) on ONE tensor pair in one grid: the output-parity phases of a stride-2
 * input gradient / transposed convolution (stylegan2/model.py:250-261, models.py:32-38: the zero-stuffed taps are never multiplied;
 * 4 / 2 / 2 / 1 taps for a 3x3 kernel).  params[i] and wmat[i] (planes of ideas_b3_split_weights) describe launch i; x, y, the
 * per-sample scales, B / IH / IW / Cin / YH / YW / Cout are shared; no bias / resid / act / accumulate / reflect.  The launches run
 * back to back inside one grid, in the order given (put the one with the most taps first): no grid ramp / tail per launch.
 * dtype IDEAS_BF16: the same for the bf16 family, wmat[i] = the packs of ideas_bf16_pack_weights_strided (one per sample when
 * in_scale is given -- only its presence is used, the scale is in the packs).  When the launches are the canonical 3x3 / stride-2
 * / pad-0 phases and Cout > 64, IDEAS_F32_B3 runs them as ONE pass over a shared LDS image of the input (csrc/conv_b3_tphase.hip). */
int ideas_conv_igemm_multi(int n, void* y, const void* x, const void* const* wmat, const float* in_scale, const float* out_scale,
                           const ideas_conv_params* params, int dtype, void* stream);

/* Blur -> 3x3 / stride-2 convolution of a downsampling ConvLayer in ONE kernel (csrc/conv_b3_s2fir.hip, dtype IDEAS_F32_B3 only):
 * replaces upfirdn2d_op(x, kernel, pad) followed by F.conv2d(stride=2) (models.py:68-76 -> stylegan2/model.py:88-91, 115-121) without
 * writing the blurred tensor to HBM: a block builds the blurred pixels under its 8 x 16 output patch in LDS (the separable 4-tap FIR
 * in f32, then the exact 3-way bf16 split) and contracts the nine stride-2 taps from that image.
 *   p        the stride-2 convolution on the BLURRED tensor [B, IH, IW, Cin]: TY = TX = 3, sy = sx = 2, no padding, dense output
 *            (YH = OH = (IH - 3) / 2 + 1); epilogue fields (gain, act, alpha, act_gain, resid_gain) as ideas_conv_igemm;
 *   x        the RAW input [B, xh, xw, Cin] f32 NHWC; IH = xh + pad0 + pad1 - 3 with 0 <= pad0, pad1 <= 3 (the Blur's pads);
 *   fir_h/v  HOST pointers to the 4 + 4 factors of the flipped, gain-scaled FIR: xb[i,j] = sum_a fir_v[a] * (sum_b fir_h[b] *
 *            x[i + a - pad0, j + b - pad0]) -- evaluated in exactly that order with fused multiply-adds, as the stand-alone blur does;
 *   wplanes  the planes of ideas_b3_split_weights for the [Cout][3*3*Cin] matrix; bias / resid optional (float[Cout] / y's shape);
 *   xb_out   optional [B, IH, IW, Cin] f32: the blurred tensor as a side output (the operand of the layer's weight gradient); needs
 *            IH == 2*OH + 1 and IW == 2*OW + 1 (every blurred pixel lies under some output pixel's taps), else IDEAS_E_UNSUPPORTED.
 * ideas_b3_blur_conv_s2_supported: 1 if the geometry is covered (Cin % 16 == 0, Cout % 4 == 0, tensors < 4 GiB). */
int ideas_b3_blur_conv_s2_supported(const ideas_conv_params* p, int xh, int xw, int pad0);
int ideas_b3_blur_conv_s2(void* y, void* xb_out, const void* x, const void* wplanes, const float* fir_h, const float* fir_v,
                          const float* bias, const void* resid, const ideas_conv_params* p, int xh, int xw, int pad0, void* stream);

/* Generic direct convolution (VALU) with the same parameterisation and epilogue; any Cin/Cout. Used for the
 * handful of tiny-K layers (RGB / N-channel inputs) where the MFMA tile would be empty. */
/* (dtype IDEAS_F32, or IDEAS_BF16: bf16 x / y / resid / gy with f32 weights, scales, bias and weight gradient) */
int ideas_conv_direct(void* y, const void* x, const void* wmat, const float* in_scale, const float* out_scale,
                      const float* bias, const void* resid, const ideas_conv_params* p, int dtype, void* stream);
int ideas_conv_wgrad_direct(float* gw, const void* gy, const void* x, const float* in_scale, const float* out_scale,
                            const ideas_conv_params* p, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Style demodulation (stylegan2/model.py:243-244):  d[b,o] = rsqrt( sum_i s[b,i]^2 * wsq[o,i] + eps )
 * where wsq[o,i] = scale^2 * sum_k W[o,i,k]^2.  One wavefront per (b,o), shuffle reduction over Cin.
 * ---------------------------------------------------------------------------------------------- */
int ideas_demod(float* d, const float* s, const float* wsq, int B, int Cin, int Cout, float eps, void* stream);
/* wsq[o][i] = scale2 * sum_{ky,kx} W[o][i][ky][kx]^2 of a 4-D f32 weight of strides (so, si, sky, skx) (elements). */
int ideas_weight_sqsum(float* wsq, const float* w, int Cout, int Cin, int KH, int KW, int64_t so, int64_t si, int64_t sky, int64_t skx,
                       float scale2, void* stream);
/* The same table accumulated and stored in double (the operand of ideas_demod_bwd). */
int ideas_weight_sqsum_f64(double* wsq, const float* w, int Cout, int Cin, int KH, int KW, int64_t so, int64_t si, int64_t sky,
                           int64_t skx, double scale2, void* stream);
/* Style gradient of a modulated conv from the per-sample reductions of its backward (ideas_pixel_dot / ideas_act_bwd_dot), evaluated
 * in DOUBLE (its two terms cancel to a small remainder; stylegan2/model.py:239-248 gets the same gradient from autograd through the
 * materialised per-sample weights):
 *   q[b,o]  = sum_i s[b,i]^2 wsq[o,i],  dt = (q + eps)^(-1/2)                      (re-evaluated here from s and the double wsq)
 *   gq[b,o] = -0.5 * (dot_d[b,o] / d[b,o]) * dt^3      (dL/dq; dot_d = <gy, y>, y = d * conv with d = the f32 factor of the forward)
 *   gs[b,i] = (s != 0 ? dot_s[b,i] / s[b,i] : 0) + 2 s[b,i] * sum_o gq[b,o] * wsq[o,i]      (dot_s = <x, gx>, gx = s * dL/d(s x))
 * d == NULL (no demodulation): only the first term of gs; gq / dot_d / wsq are then unused.
 * dot_d is IN/OUT: on return it holds gq in double (the second launch reads it back; gq is its float copy for ideas_demod_wgrad). */
int ideas_demod_bwd(float* gs, float* gq, const double* dot_s, double* dot_d, const float* d, const float* s, const double* wsq,
                    int B, int Cin, int Cout, float eps, void* stream);
/* Weight gradient through the demodulation, ADDED in place:  gw[o][i][k] += coef * W[o][i][k] * sum_b gq[b,o] * s[b,i]^2
 * (coef = 2 * scale^2).  W has strides (so, si, sky, skx), gw strides (go, gi, gky, gkx). */
int ideas_demod_wgrad(float* gw, const float* w, const float* gq, const float* s, int B, int Cout, int Cin, int KH, int KW, int64_t so,
                      int64_t si, int64_t sky, int64_t skx, int64_t go, int64_t gi, int64_t gky, int64_t gkx, float coef, void* stream);

/* Per-(b,c) sum over pixels of a[b,p,c]*g[b,p,c] (NHWC).  Gives d(style) and d(demod) of the modulated conv
 * without materialising per-sample weights.  out must be ZEROED double[B*C]: products and sums are double throughout (the kernel is
 * HBM-bound; see ideas_demod_bwd for why the width matters).  C <= 6144. */
int ideas_pixel_dot(double* out, const void* a, const void* g, int B, int64_t P, int C, int dtype, void* stream);

/* Backward prologue of a fused (modulated conv + bias + leaky-ReLU), NHWC, C % 4 == 0.  One pass over the incoming
 * gradient gy and the saved POST-activation output `out` [B,P,C]:
 *     gpre      = (out > 0 ? gy : gy*alpha) * act_gain
 *     bias_grad[c] += sum_{b,p} gpre                       (ZEROED float[C])
 *     dot[b,c]     += sum_p gpre * (inverse_act(out) - bias[c])   (ZEROED double[B*C]; = <gpre, demodulated conv output>; C <= 3072)
 * so neither the pre-activation tensor nor a separate bias-gradient / pixel-dot pass is needed.
 * gpre_scale (optional float[B*C]): the STORED gpre is multiplied by it (bias_grad and dot are not) -- the bf16 path passes
 * the demodulation factor here, so its input- and weight-gradient kernels take the already-scaled gradient. */
int ideas_act_bwd_dot(void* gpre, float* bias_grad, double* dot, const void* gy, const void* out, const float* bias,
                      const float* gpre_scale, int B, int64_t P, int C, float alpha, float act_gain, int dtype, void* stream);

/* Adjoint of ReflectionPad2d(pad) in NHWC: gx [B,H,W,C] = fold of gpadded [B,H+2pad,W+2pad,C] (mirrored border rows /
 * columns added back onto their sources).  Any C (16-byte vectors when C % 4 == 0).  Used by the input gradient of the reflect-padded 3x3 convs of
 * E / Gstru / Ex (models.py:102-106). */
int ideas_reflect_fold(void* gx, const void* gpadded, int B, int H, int W, int C, int pad, int dtype, void* stream);

/* Fused Adam (beta1 = 0, as every optimiser of train.py:417-432) + optional EMA (utils.py:55-60) over ONE flat f32
 * buffer aliasing all parameters of an optimiser group (n % 4 == 0, 16-byte aligned):
 *     v = beta2*v + (1-beta2)*g*g;  p -= lr * g / (sqrt(v)/sqrt(bias_correction2) + eps);  ema = d*ema + (1-d)*p
 * bias_correction2 = 1 - beta2^step (computed by the caller); ema may be NULL. */
int ideas_adam_ema(float* p, const float* g, float* v, float* ema, int64_t n, float lr, float beta2, float eps,
                   float bias_correction2, float ema_decay, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Input pipeline (dataset.py:10-85 + the transform of train.py:443-449), device side.
 * x_u8: decoded images, uint8 [B][H][W][C] (PIL's HWC order).  y: f32 [B][H][W][C] = the NHWC activation layout.
 *   y = ((x / 255) - mean) / stdv           -- ToTensor then Normalize(mean, stdv), the same two roundings
 * flip_u8 (optional, uint8 [B]): non-zero -> that sample is mirrored horizontally (RandomHorizontalFlip; the caller draws). */
int ideas_image_u8_to_f32(float* y, const void* x_u8, const void* flip_u8, int B, int H, int W, int C, float mean, float stdv,
                          void* stream);

/* ------------------------------------------------------------------------------------------------
 * patchify_image (utils.py:127-149), device side: crop `n_crop` boxes (host int[n_crop][4] = y, x, h, w in source pixels, the
 * same boxes for every image, n_crop <= 64) out of x [B][H][W][C] (NHWC, C = 3 or 1; f32 or bf16) and resize each bilinearly
 * (F.interpolate(mode="bilinear", align_corners=False)) to out_h x out_w.  y: [B * n_crop][out_h][out_w][C], image-major (the
 * torch.stack(patches, 1).view(B * n_crop, ...) order of utils.py:147-149), x's dtype.  One launch for all boxes.
 * ideas_patch_resize_bwd: gx f32 [B][H][W][C] += the adjoint applied to gy (y's shape and dtype); `clear` != 0 zeroes gx first.
 * Overlapping crops accumulate with f32 atomics (the summation order is not fixed, as in torch's own backward).
 * (additive since ABI version 3.) */
int ideas_patch_resize(void* y, const void* x, const int* boxes, int n_crop, int B, int C, int H, int W, int out_h, int out_w,
                       int dtype, void* stream);
int ideas_patch_resize_bwd(float* gx, const void* gy, const int* boxes, int n_crop, int B, int C, int H, int W, int out_h,
                           int out_w, int clear, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * EqualLinear (stylegan2/model.py:131-160:  F.linear(input, weight * scale, bias * lr_mul)) for MANY layers sharing one input, one
 * launch per direction.  Replaces the ATen / vendor-GEMM calls behind F.linear on this path: every linear layer of IDEAS is skinny
 * (M = batch <= a few hundred rows, K = 32 .. 8192, N = 1 .. 512), and the generator applies sixteen of them (the modulation layers
 * of its StyledConvs, stylegan2/model.py:226,239) to the same texture code.  (additive since ABI version 3.)
 *
 * A segment = one layer.  All matrices are f32 row-major with explicit row strides (in floats, multiples of 4; 16-byte aligned bases):
 *     w    [n][K]   the layer's weight (ldw)               bias [n] or NULL
 *     y    [M][n]   forward: the output;  backward: the incoming gradient g = dL/dy (ldy)
 *     gw   [n][K]   ideas_linear_bwd_w: the weight gradient (ldgw);   gb [n] or NULL: the bias gradient
 *     scale         the equalised-lr factor (1/sqrt(K) * lr_mul);   bias_mul = lr_mul
 * tile0 / pad_ are scratch of the library (any value).
 *   ideas_linear_fwd     y_s[m][j]   = scale_s * sum_k x[m][k] w_s[j][k] + bias_mul_s * bias_s[j]             (K % 8 == 0)
 *   ideas_linear_bwd_x   gx[m][k]    = sum_s scale_s * sum_j y_s[m][j] w_s[j][k]       (ONE tensor: the sum over the segments, i.e. what
 *                                      autograd would add up for an input used by every layer; n_s % 8 == 0, K % 4 == 0).
 *                                      `workspace`: >= ideas_linear_bwd_x_workspace(sum n_s, M, K) bytes of device scratch
 *                                      (split partial sums, folded in a fixed order: no atomics, reproducible).
 *   ideas_linear_bwd_w   gw_s[j][k] (+)= scale_s * sum_m y_s[m][j] x[m][k];   gb_s[j] (+)= bias_mul_s * sum_m y_s[m][j]
 *                                      (`accumulate` != 0: add to what gw / gb hold -- e.g. the optimiser's gradient buffer; K % 4 == 0)
 * Arithmetic: v_mfma_f32_32x32x2_f32 (exact f32 fused multiply-adds), fixed summation order. */
#define IDEAS_LINEAR_MAX_SEGMENTS 32
typedef struct ideas_linear_seg {
    const float* w;
    const float* bias;
    float* y;
    float* gw;
    float* gb;
    int n, ldw, ldy, ldgw;
    float scale, bias_mul;
    int tile0, pad_;
} ideas_linear_seg;
int ideas_sizeof_linear_seg(void);
int ideas_linear_fwd(const ideas_linear_seg* segs, int nseg, const void* x, int M, int K, int ldx, void* stream);
int64_t ideas_linear_bwd_x_workspace(int total_n, int M, int K);
int ideas_linear_bwd_x(const ideas_linear_seg* segs, int nseg, void* gx, int M, int K, int ldgx, void* workspace,
                       int64_t workspace_bytes, void* stream);
int ideas_linear_bwd_w(const ideas_linear_seg* segs, int nseg, const void* x, int M, int K, int ldx, int accumulate, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* IDEAS_HIP_H */
