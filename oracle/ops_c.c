/* ORACLE — test infrastructure only (tests/, smoke(), bench.py's cpu_baseline leg).  NOT the product path.
 *
 * Plain-C, loop-per-definition restatement of the four custom ops of the IDEAS hot path, NCHW float32 with
 * double accumulation, written straight from the reference's definitions:
 *   oracle_fused_bias_act   stylegan2/op/fused_bias_act_kernel.cu:18-49  (act*10+grad switch, bias index (i/step)%C)
 *   oracle_upfirdn2d        stylegan2/op/upfirdn2d.py:159-200 / upfirdn2d_kernel.cu:49-105
 *   oracle_conv2d           F.conv2d as EqualConv2d calls it, stylegan2/model.py:115-121 (weight * scale, stride, pad)
 *   oracle_conv_transpose2d models.py:32-38 (weight [Cin,Cout,k,k], stride, padding 0)
 *   oracle_modulated_conv2d stylegan2/model.py:236-277 same-resolution branch + the conv_transpose of the upsample
 *                           branch (the blur that follows is oracle_upfirdn2d), per-sample weights as the reference
 * Pinned by tests/test_oracle_golden.py::test_c_oracle_* against the vectors captured from the reference.
 * Pure C99, no dependencies; only ever sized for the small parity cases.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

void oracle_fused_bias_act(float* out, const float* x, const float* b, const float* ref, long n, int C, long step_b,
                           int act, int grad, float alpha, float scale) {
    for (long i = 0; i < n; ++i) {
        float v = x[i];
        if (b) v += b[(i / step_b) % C];
        float r = ref ? ref[i] : 0.0f, y;
        switch (act * 10 + grad) {
            default:
            case 10: case 11: y = v; break;
            case 12: case 32: y = 0.0f; break;
            case 30: y = (v > 0.0f) ? v : v * alpha; break;
            case 31: y = (r > 0.0f) ? v : v * alpha; break;
        }
        out[i] = y * scale;
    }
}

/* x [planes, in_h, in_w] -> y [planes, out_h, out_w]; k [kh, kw]; correlates with the FLIPPED kernel */
void oracle_upfirdn2d(float* y, const float* x, const float* k, long planes, int in_h, int in_w, int kh, int kw,
                      int up, int down, int pad0, int pad1) {
    const int out_h = (in_h * up + pad0 + pad1 - kh) / down + 1;
    const int out_w = (in_w * up + pad0 + pad1 - kw) / down + 1;
    for (long p = 0; p < planes; ++p)
        for (int oy = 0; oy < out_h; ++oy)
            for (int ox = 0; ox < out_w; ++ox) {
                double acc = 0.0;
                for (int ky = 0; ky < kh; ++ky) {
                    int uy = oy * down + ky - pad0;
                    if (uy < 0 || uy % up) continue;
                    int iy = uy / up;
                    if (iy >= in_h) continue;
                    for (int kx = 0; kx < kw; ++kx) {
                        int ux = ox * down + kx - pad0;
                        if (ux < 0 || ux % up) continue;
                        int ix = ux / up;
                        if (ix >= in_w) continue;
                        acc += (double)x[(p * in_h + iy) * in_w + ix] * (double)k[(kh - 1 - ky) * kw + (kw - 1 - kx)];
                    }
                }
                y[(p * out_h + oy) * out_w + ox] = (float)acc;
            }
}

/* groups-aware direct conv: x [B, G*Cin, H, W], w [G*Cout, Cin, k, k] -> y [B, G*Cout, OH, OW] */
static void conv2d_grouped(float* y, const float* x, const float* w, const float* bias, int B, int G, int Cin, int Cout,
                           int H, int W, int k, int stride, int pad, float wscale) {
    const int OH = (H + 2 * pad - k) / stride + 1, OW = (W + 2 * pad - k) / stride + 1;
    for (int b = 0; b < B; ++b)
        for (int g = 0; g < G; ++g)
            for (int o = 0; o < Cout; ++o)
                for (int oy = 0; oy < OH; ++oy)
                    for (int ox = 0; ox < OW; ++ox) {
                        double acc = 0.0;
                        for (int i = 0; i < Cin; ++i)
                            for (int ky = 0; ky < k; ++ky) {
                                int iy = oy * stride + ky - pad;
                                if (iy < 0 || iy >= H) continue;
                                for (int kx = 0; kx < k; ++kx) {
                                    int ix = ox * stride + kx - pad;
                                    if (ix < 0 || ix >= W) continue;
                                    float wv = w[(((long)(g * Cout + o) * Cin + i) * k + ky) * k + kx] * wscale;
                                    acc += (double)x[(((long)b * G * Cin + g * Cin + i) * H + iy) * W + ix] * (double)wv;
                                }
                            }
                        if (bias) acc += bias[g * Cout + o];
                        y[(((long)b * G * Cout + g * Cout + o) * OH + oy) * OW + ox] = (float)acc;
                    }
}

void oracle_conv2d(float* y, const float* x, const float* w, const float* bias, int B, int Cin, int Cout, int H, int W,
                   int k, int stride, int pad) {
    conv2d_grouped(y, x, w, bias, B, 1, Cin, Cout, H, W, k, stride, pad, 1.0f / sqrtf((float)(Cin * k * k)));
}

/* x [B,Cin,H,W], w [Cin,Cout,k,k] (already multiplied by any scale), padding 0 -> y [B,Cout,(H-1)s+k,(W-1)s+k] */
static void conv_transpose2d_raw(float* y, const float* x, const float* w, int B, int Cin, int Cout, int H, int W, int k,
                                 int stride) {
    const int OH = (H - 1) * stride + k, OW = (W - 1) * stride + k;
    double* acc = (double*)calloc((size_t)B * Cout * OH * OW, sizeof(double));
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < Cin; ++i)
            for (int iy = 0; iy < H; ++iy)
                for (int ix = 0; ix < W; ++ix) {
                    double xv = x[(((long)b * Cin + i) * H + iy) * W + ix];
                    for (int o = 0; o < Cout; ++o)
                        for (int ky = 0; ky < k; ++ky)
                            for (int kx = 0; kx < k; ++kx)
                                acc[(((long)b * Cout + o) * OH + iy * stride + ky) * OW + ix * stride + kx] +=
                                    xv * (double)w[(((long)i * Cout + o) * k + ky) * k + kx];
                }
    for (long t = 0; t < (long)B * Cout * OH * OW; ++t) y[t] = (float)acc[t];
    free(acc);
}

void oracle_conv_transpose2d(float* y, const float* x, const float* w, int B, int Cin, int Cout, int H, int W, int k,
                             int stride) {
    const float s = 1.0f / sqrtf((float)(Cin * k * k));
    float* ws = (float*)malloc(sizeof(float) * (size_t)Cin * Cout * k * k);
    for (long t = 0; t < (long)Cin * Cout * k * k; ++t) ws[t] = w[t] * s;
    conv_transpose2d_raw(y, x, ws, B, Cin, Cout, H, W, k, stride);
    free(ws);
}

/* style [B,Cin] is the already-affine modulation; weight [Cout,Cin,k,k].  upsample=0: y [B,Cout,H,W] (pad k/2);
 * upsample=1: y [B,Cout,2H+1,2W+1] (the conv_transpose only; blur with oracle_upfirdn2d afterwards). */
void oracle_modulated_conv2d(float* y, const float* x, const float* style, const float* weight, int B, int Cin, int Cout,
                             int H, int W, int k, int demodulate, int upsample) {
    const float scale = 1.0f / sqrtf((float)(Cin * k * k));
    const long wn = (long)Cout * Cin * k * k;
    float* wb = (float*)malloc(sizeof(float) * wn);
    for (int b = 0; b < B; ++b) {
        for (int o = 0; o < Cout; ++o) {
            double ss = 0.0;
            for (int i = 0; i < Cin; ++i)
                for (int t = 0; t < k * k; ++t) {
                    float v = scale * weight[((long)o * Cin + i) * k * k + t] * style[(long)b * Cin + i];
                    wb[((long)o * Cin + i) * k * k + t] = v;
                    ss += (double)v * v;
                }
            if (demodulate) {
                float d = 1.0f / sqrtf((float)ss + 1e-8f);
                for (long t = 0; t < (long)Cin * k * k; ++t) wb[(long)o * Cin * k * k + t] *= d;
            }
        }
        if (!upsample) {
            conv2d_grouped(y + (long)b * Cout * H * W, x + (long)b * Cin * H * W, wb, NULL, 1, 1, Cin, Cout, H, W, k, 1,
                           k / 2, 1.0f);
        } else {
            float* wt = (float*)malloc(sizeof(float) * wn); /* [Cin,Cout,k,k] as the reference's transpose(1,2) */
            for (int o = 0; o < Cout; ++o)
                for (int i = 0; i < Cin; ++i)
                    memcpy(wt + ((long)i * Cout + o) * k * k, wb + ((long)o * Cin + i) * k * k, sizeof(float) * k * k);
            const int OH = 2 * H + 1, OW = 2 * W + 1;
            conv_transpose2d_raw(y + (long)b * Cout * OH * OW, x + (long)b * Cin * H * W, wt, 1, Cin, Cout, H, W, k, 2);
            free(wt);
        }
    }
    free(wb);
}
