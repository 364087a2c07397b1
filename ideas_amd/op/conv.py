"""Convolutions with explicit first- and second-order backward on the gfx950 implicit-GEMM kernels.

Plays the role the north_star calls ``conv2d_gradfix``: ``conv2d`` / ``conv_transpose2d`` whose backward is
built from *differentiable* Functions (input-gradient, weight-gradient), so ``d_r1_loss``'s
``autograd.grad(..., create_graph=True)`` (utils.py:112-118) works through the discriminators.  Replaces
the ATen/cuDNN calls at stylegan2/model.py:115-121 (EqualConv2d) and models.py:32-38 (EqualConvTranspose2d).

All activations are NHWC in memory (``torch.channels_last`` on a logical [B,C,H,W] tensor); weights
[O,I,KH,KW] are used in OHWI order (free when the parameter itself is channels_last).  The equalised-lr
``scale`` is folded into the kernel's accumulator gain instead of a ``weight * scale`` pass.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Tuple

import torch
from torch.autograd import Function

from .. import _lib
from ..precision import to_act
from . import conv_plan
from .conv_plan import ConvGeom, Launch, convT_out_size, plan_dgrad, plan_fwd, plan_wgrad

CL = torch.channels_last

# 1-D Winograd F(2,3) for the 3x3 / stride-1 / pad-1 layers (csrc/conv_wino.hip): 1.5x fewer MFMAs.
# IDEAS_WINOGRAD=0 falls back to the direct implicit GEMM (A/B measurements, debugging).
import os as _os
WINOGRAD = _os.environ.get("IDEAS_WINOGRAD", "1") != "0"
# Contraction arithmetic of the MFMA convolutions (include/ideas_hip.h): "f32" = f32 matrix instruction, "b3" = exact
# 3-way bf16 split + six bf16 MFMA products (f32-class error, 2.67x the matrix rate).
MATH = {"f32": _lib.F32, "b3": _lib.F32_B3}[_os.environ.get("IDEAS_MATH", "b3")]


def wino_weights(w_ohwi: torch.Tensor) -> torch.Tensor:
    """[O,3,3,I] (o, ky, kx, ci) -> U [4,O,3,I]: U0 = w0, U1 = (w0+w1+w2)/2, U2 = (w0-w1+w2)/2, U3 = w2 over kx."""
    # (sums in double, ONE rounding to f32: the transformed weights are shared by every pixel, so their rounding error is
    #  coherent across the backward's pixel reductions -- see csrc/conv_b3_wino.hip::wino_split_weights_kernel)
    w_ohwi = w_ohwi.double()
    w0, w1, w2 = w_ohwi[:, :, 0], w_ohwi[:, :, 1], w_ohwi[:, :, 2]
    s = w0 + w2
    return torch.stack((w0, (s + w1) * 0.5, (s - w1) * 0.5, w2)).float().contiguous()


def _wino_ok(g: "ConvGeom", cin: int, width: int, fwd: bool = True) -> bool:
    if fwd and MATH == _lib.F32_B3 and cin % 16 == 0:
        return False          # the split-bf16 direct kernel outruns the f32 Winograd kernel
    return WINOGRAD and g.kh == 3 and g.kw == 3 and g.stride == 1 and g.pad == 1 and width % 2 == 0 and cin % 8 == 0


# Winograd F(2,3) variant of the split-bf16 kernel for the 3x3/s1/p1 layers (csrc/conv_b3_wino.hip); IDEAS_B3_WINO=0 keeps them
# on the direct split kernel.
B3_WINO = _os.environ.get("IDEAS_B3_WINO", "1") != "0"
# (diagnostics) the Winograd kernel for the forward convs only / the input gradients only
B3_WINO_FWD = _os.environ.get("IDEAS_B3_WINO_FWD", "1") != "0"
B3_WINO_DGRAD = _os.environ.get("IDEAS_B3_WINO_DGRAD", "1") != "0"


def _b3_wino_ok(g: "ConvGeom", cin: int, cout: int, width: int) -> bool:
    return (MATH == _lib.F32_B3 and B3_WINO and g.kh == 3 and g.kw == 3 and g.stride == 1 and g.pad == 1 and width % 2 == 0
            and cin % 16 == 0 and cout % 4 == 0)


def b3_wino_planes(w: torch.Tensor, transposed: bool) -> torch.Tensor:
    """Winograd-transformed, split weights of a [O,I,3,3] parameter for ideas_conv3x3_wino(IDEAS_F32_B3): the forward matrix,
    or (transposed) the flipped one of the input gradient, read straight from the OHWI memory of the parameter."""
    def make():
        o, i = w.shape[0], w.shape[1]
        wm = w.permute(0, 2, 3, 1)
        wm = wm if wm.is_contiguous() else wm.contiguous()
        n, c, sn, sky, skx, sc, base = (i, o, 1, -3 * i, -i, 9 * i, 8 * i) if transposed else (o, i, 9 * i, 3 * i, i, 1, 0)
        pl = torch.empty(12 * n * 3 * c, device=w.device, dtype=torch.bfloat16)
        _lib.check(_lib.load().ideas_b3_wino_split_weights(_lib.ptr(pl), _lib.ptr(wm), n, c, sn, sky, skx, sc, base,
                                                            _lib.stream_ptr()), "ideas_b3_wino_split_weights")
        return pl
    prep = None
    wm0 = w.permute(0, 2, 3, 1)
    if wm0.is_contiguous() and w.shape[1] % 16 == 0:        # read in place: the same call as an entry of the batched refill
        o, i = w.shape[0], w.shape[1]
        n, c, sn, sky, skx, sc, base = (i, o, 1, -3 * i, -i, 9 * i, 8 * i) if transposed else (o, i, 9 * i, 3 * i, i, 1, 0)
        if c % 16 == 0:
            prep = (_lib.PREP_B3_WINO, 12 * n * 3 * c, (n, c), (sn, sky, skx, sc, base), 0, wm0.data_ptr(), n * 3 * (c // 4))
    return conv_plan.cached(w, ("b3wino", transposed), make, prep)


def launch_wino(y, x, umat, b, cin, h, w, cout, gain, reflect, in_scale=None, out_scale=None, bias=None, resid=None,
                act=False, alpha=0.2, act_gain=1.0, resid_gain=1.0, dtype=_lib.F32) -> None:
    p = _lib.ConvParams(b, h, w, cin, h, w, cout, h, w, 3, 3, 1, 1, 1, 1, -1, -1, 1, 1, 0, 0, int(reflect), int(act),
                        alpha, act_gain, resid_gain, 0, gain)
    rc = _lib.load().ideas_conv3x3_wino(_lib.ptr(y), _lib.ptr(x), _lib.ptr(umat), _lib.ptr(in_scale), _lib.ptr(out_scale),
                                        _lib.ptr(bias), _lib.ptr(resid), C.byref(p), dtype, _lib.stream_ptr())
    _lib.check(rc, "ideas_conv3x3_wino")


def _nhwc(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32 and t.dtype != torch.bfloat16:
        raise RuntimeError(f"ideas_amd conv: only float32 and bfloat16 activations are implemented, got {t.dtype}")
    return t if t.is_contiguous(memory_format=CL) else t.contiguous(memory_format=CL)


BF = torch.bfloat16


def _f32(t):
    return None if t is None else t.float()


def bf16_pack(L: Launch, in_scale=None) -> torch.Tensor:
    """bf16 pack of a forward-family weight matrix for ideas_conv_igemm(IDEAS_BF16), read from the parameter through the strides
    of ``L.wview``.  Plain convs: one pack, memoised like the b3 planes.  Modulated convs (in_scale [B, Cin]): one pack per
    sample, w * in_scale[b]; memoised on (weights, styles) for the iteration -- G is applied to the same texture code two or three
    times -- inside a byte budget (conv_plan.STYLE_BUDGET_MB, oldest entries dropped first: the packs of a style that cannot
    recur, e.g. the D phase's T2, do not stay resident until the G step)."""
    v = L.wview
    nb = 1 if in_scale is None else L.B

    def make():
        pk = torch.empty(nb * L.Cout * L.TY * L.TX * L.Cin, device=v.device, dtype=BF)
        sn, sty, stx, sc = v.stride()
        _lib.check(_lib.load().ideas_bf16_pack_weights_strided(_lib.ptr(pk), _lib.ptr(v), _lib.ptr(in_scale), nb, L.Cout, L.TY, L.TX,
                                                                L.Cin, sn, sty, stx, sc, _lib.stream_ptr()),
                   "ideas_bf16_pack_weights_strided")
        return pk
    if L.wsrc is None:
        return make()
    if in_scale is not None:           # per-sample packs: the same (weights, styles) pair comes back within an iteration (conv_plan.cached_on)
        return conv_plan.cached_on(L.wsrc, ("bf16s",) + L.wkey, in_scale, make, budget=True)
    sn, sty, stx, sc = v.stride()
    unit = sc == 1 and v.data_ptr() % 16 == 0 and sn % 4 == 0 and sty % 4 == 0 and stx % 4 == 0
    prep = (_lib.PREP_BF16_PACK, L.Cout * L.TY * L.TX * L.Cin, (L.Cout, L.TY, L.TX, L.Cin), (sn, sty, stx, sc), unit, v.data_ptr(),
            L.Cout * L.TY * L.TX * (L.Cin // 8)) if L.Cin % 32 == 0 else None
    return conv_plan.cached(L.wsrc, ("bf16",) + L.wkey, make, prep)


def b3_planes(L: Launch) -> torch.Tensor:
    """Split-bf16 planes of a forward-family weight matrix for ideas_conv_igemm(IDEAS_F32_B3), memoised per optimiser step."""
    v = L.wview

    def split():
        pl = torch.empty(3 * L.Cout * L.TY * L.TX * L.Cin, device=v.device, dtype=BF)
        sn, sty, stx, sc = v.stride()
        _lib.check(_lib.load().ideas_b3_split_weights_strided(_lib.ptr(pl), _lib.ptr(v), L.Cout, L.TY, L.TX, L.Cin, sn, sty, stx, sc,
                                                               _lib.stream_ptr()), "ideas_b3_split_weights_strided")
        return pl
    if L.wsrc is None:
        return split()
    sn, sty, stx, sc = v.stride()
    unit = sc == 1 and v.data_ptr() % 16 == 0 and sn % 4 == 0 and sty % 4 == 0 and stx % 4 == 0
    prep = (_lib.PREP_B3_SPLIT, 3 * L.Cout * L.TY * L.TX * L.Cin, (L.Cout, L.TY, L.TX, L.Cin), (sn, sty, stx, sc), unit, v.data_ptr(),
            L.Cout * L.TY * L.TX * (L.Cin // 4)) if L.Cin % 16 == 0 else None
    return conv_plan.cached(L.wsrc, ("b3",) + L.wkey, split, prep)


def _params(L: Launch, gain: float, accumulate: bool = False, act: bool = False, alpha: float = 0.2,
            act_gain: float = 1.0, resid_gain: float = 1.0) -> _lib.ConvParams:
    return _lib.ConvParams(L.B, L.IH, L.IW, L.Cin, L.YH, L.YW, L.Cout, L.OH, L.OW, L.TY, L.TX, L.sy, L.sx, L.dy, L.dx,
                           L.offy, L.offx, L.osy, L.osx, L.ooy, L.oox, L.reflect, int(act), alpha, act_gain,
                           resid_gain, int(accumulate), gain)


def launch_fwd(y: torch.Tensor, x: torch.Tensor, L: Launch, gain: float, in_scale=None, out_scale=None, bias=None,
               resid=None, act: bool = False, alpha: float = 0.2, act_gain: float = 1.0, resid_gain: float = 1.0,
               accumulate: bool = False) -> None:
    """Enqueue one forward-family launch (MFMA implicit GEMM when Cin % 4 == 0, VALU direct otherwise)."""
    lib = _lib.load()
    p = _params(L, gain, accumulate, act, alpha, act_gain, resid_gain)
    if x.dtype == BF:
        if lib.ideas_bf16_conv_supported(C.byref(p), int(in_scale is not None)):
            rc = lib.ideas_conv_igemm(_lib.ptr(y), _lib.ptr(x), _lib.ptr(bf16_pack(L, in_scale)), _lib.ptr(in_scale), _lib.ptr(out_scale),
                                      _lib.ptr(bias), _lib.ptr(resid), C.byref(p), _lib.BF16, _lib.stream_ptr())
            _lib.check(rc, "ideas_conv_igemm[bf16]")
            return
        if lib.ideas_bf16_direct_supported(C.byref(p)) and in_scale is None and out_scale is None:
            rc = lib.ideas_conv_direct(_lib.ptr(y), _lib.ptr(x), _lib.ptr(L.wmat), None, None, _lib.ptr(bias), _lib.ptr(resid),
                                       C.byref(p), _lib.BF16, _lib.stream_ptr())
            _lib.check(rc, "ideas_conv_direct[bf16]")
            return
        # geometries without a bf16 kernel (tiny layers: N-channel / 8-channel ends, odd sizes): f32 kernels on casts
        partial = L.osy != 1 or L.osx != 1 or L.OH != L.YH or L.OW != L.YW     # a parity phase writes only its own pixels
        y32 = y.float() if (accumulate or partial) else torch.empty(y.shape, device=y.device, dtype=torch.float32, memory_format=CL)
        launch_fwd(y32, x.float(), L, gain, in_scale, out_scale, bias, _f32(resid), act, alpha, act_gain, resid_gain, accumulate)
        y.copy_(y32)
        return
    if MATH == _lib.F32_B3 and lib.ideas_b3_conv_supported(C.byref(p)):
        rc = lib.ideas_conv_igemm(_lib.ptr(y), _lib.ptr(x), _lib.ptr(b3_planes(L)), _lib.ptr(in_scale), _lib.ptr(out_scale),
                                  _lib.ptr(bias), _lib.ptr(resid), C.byref(p), _lib.F32_B3, _lib.stream_ptr())
        _lib.check(rc, "ideas_conv_igemm[b3]")
        return
    fn = lib.ideas_conv_igemm if (L.Cin % 4 == 0) else lib.ideas_conv_direct
    rc = fn(_lib.ptr(y), _lib.ptr(x), _lib.ptr(L.wmat), _lib.ptr(in_scale), _lib.ptr(out_scale), _lib.ptr(bias),
            _lib.ptr(resid), C.byref(p), _lib.F32, _lib.stream_ptr())
    _lib.check(rc, "ideas_conv_igemm" if L.Cin % 4 == 0 else "ideas_conv_direct")


B3_MULTI = _os.environ.get("IDEAS_B3_MULTI", "1") != "0"


def launch_multi(y: torch.Tensor, x: torch.Tensor, launches, gain: float, in_scale=None, out_scale=None) -> bool:
    """The output-parity phases of a stride-2 input gradient / transposed conv as ONE grid (ideas_conv_igemm_multi) where the
    split-bf16 kernel covers all of them; False -> the caller launches them one by one."""
    n = len(launches)
    if not B3_MULTI or n < 2 or n > 4:
        return False
    # heaviest launch first (include/ideas_hip.h: the launches run back to back inside one grid, so the lightest one's tail is the
    # one left exposed): plan_dgrad hands the parity phases over in (ry, rx) order, e.g. 1 / 2 / 2 / 4 taps for a 3x3 stride-2 pad-1 conv
    launches = sorted(launches, key=lambda L: -(L.TY * L.TX * L.OH * L.OW))
    lib = _lib.load()
    ps = (_lib.ConvParams * n)(*[_params(L, gain) for L in launches])
    if x.dtype == BF:           # bf16 family (csrc/conv_bf16.hip::conv_bf16_multi_kernel): the launches' packs, per sample when modulated
        if not all(lib.ideas_bf16_conv_supported(C.byref(ps[i]), int(in_scale is not None)) for i in range(n)):
            return False
        packs = [bf16_pack(L, in_scale) for L in launches]
        ws = (C.c_void_p * n)(*[_lib.ptr(pk) for pk in packs])
        rc = lib.ideas_conv_igemm_multi(n, _lib.ptr(y), _lib.ptr(x), ws, _lib.ptr(in_scale), _lib.ptr(out_scale), ps, _lib.BF16,
                                        _lib.stream_ptr())
        _lib.check(rc, "ideas_conv_igemm_multi[bf16]")
        return True
    if x.dtype != torch.float32 or MATH != _lib.F32_B3:
        return False
    if not all(lib.ideas_b3_conv_supported(C.byref(ps[i])) for i in range(n)):
        return False
    planes = [b3_planes(L) for L in launches]
    ws = (C.c_void_p * n)(*[_lib.ptr(pl) for pl in planes])
    rc = lib.ideas_conv_igemm_multi(n, _lib.ptr(y), _lib.ptr(x), ws, _lib.ptr(in_scale), _lib.ptr(out_scale), ps, _lib.F32_B3,
                                    _lib.stream_ptr())
    _lib.check(rc, "ideas_conv_igemm_multi")
    return True


def launch_wgrad(gw: torch.Tensor, gy: torch.Tensor, x: torch.Tensor, L: Launch, gain: float, in_scale=None,
                 out_scale=None) -> None:
    lib = _lib.load()
    p = _params(L, gain)
    if x.dtype == BF:
        if lib.ideas_bf16_wgrad_supported(C.byref(p), int(in_scale is not None)) and L.Cout > 8 and L.Cin > 8:
            rc = lib.ideas_conv_wgrad(_lib.ptr(gw), _lib.ptr(gy), _lib.ptr(x), _lib.ptr(in_scale), _lib.ptr(out_scale), C.byref(p),
                                      _lib.BF16, _lib.stream_ptr())
            _lib.check(rc, "ideas_conv_wgrad[bf16]")
            return
        if lib.ideas_bf16_direct_supported(C.byref(p)) and in_scale is None and out_scale is None:
            rc = lib.ideas_conv_wgrad_direct(_lib.ptr(gw), _lib.ptr(gy), _lib.ptr(x), None, None, C.byref(p), _lib.BF16,
                                             _lib.stream_ptr())
            _lib.check(rc, "ideas_conv_wgrad_direct[bf16]")
            return
        return launch_wgrad(gw, gy.float(), x.float(), L, gain, in_scale, out_scale)
    mfma = (L.Cin % 4 == 0) and (L.Cout % 4 == 0)
    fn = lib.ideas_conv_wgrad if mfma else lib.ideas_conv_wgrad_direct
    rc = fn(_lib.ptr(gw), _lib.ptr(gy), _lib.ptr(x), _lib.ptr(in_scale), _lib.ptr(out_scale), C.byref(p),
            MATH if mfma else _lib.F32, _lib.stream_ptr())
    _lib.check(rc, "ideas_conv_wgrad" if mfma else "ideas_conv_wgrad_direct")


# ----------------------------------------------------------------------------------------------------
# raw primitives of the "conv" geometry (no autograd).  lin = per-(b, input-channel) scale of the launch,
# lout = per-(b, output-channel) scale of the launch.
# ----------------------------------------------------------------------------------------------------

def conv_fwd_raw(x, w, g: ConvGeom, gain: float, lin=None, lout=None, bias=None, act=False, act_gain=1.0,
                 resid=None, resid_gain=1.0, alpha=0.2):
    x = _nhwc(x)
    if resid is not None:
        resid = _nhwc(resid)
    if x.dtype == torch.float32 and B3_WINO_FWD and _b3_wino_ok(g, x.shape[1], w.shape[0], x.shape[3]):
        b, ci, h, wd = x.shape
        co = w.shape[0]
        y = torch.empty((b, co, h, wd), device=x.device, dtype=x.dtype, memory_format=CL)
        launch_wino(y, x, b3_wino_planes(w, False), b, ci, h, wd, co, gain, g.reflect, lin, lout, bias, resid, act=act,
                    alpha=alpha, act_gain=act_gain, resid_gain=resid_gain, dtype=_lib.F32_B3)
        return y
    if x.dtype == torch.float32 and _wino_ok(g, x.shape[1], x.shape[3]):
        b, ci, h, wd = x.shape
        co = w.shape[0]
        y = torch.empty((b, co, h, wd), device=x.device, dtype=x.dtype, memory_format=CL)
        launch_wino(y, x, wino_weights(w.permute(0, 2, 3, 1)), b, ci, h, wd, co, gain, g.reflect, lin, lout, bias, resid,
                    act=act, alpha=alpha, act_gain=act_gain, resid_gain=resid_gain)
        return y
    L = plan_fwd(x.shape, w, g)
    y = torch.empty((L.B, L.Cout, L.YH, L.YW), device=x.device, dtype=x.dtype, memory_format=CL)
    launch_fwd(y, x, L, gain, lin, lout, bias, resid, act=act, alpha=alpha, act_gain=act_gain, resid_gain=resid_gain)
    return y


def conv_dgrad_raw(gy, w, g: ConvGeom, in_hw: Tuple[int, int], gain: float, lin=None, lout=None, resid=None):
    """Input gradient of the conv geometry (also: forward of the matching transposed conv).  ``resid`` (the result's shape): added
    in the kernel epilogue -- single-launch geometries only (stride 1, no mirror padding), e.g. the 1x1 skip convs."""
    gy = _nhwc(gy)
    if resid is not None:
        if g.reflect or g.stride != 1:
            raise RuntimeError("conv_dgrad_raw(resid=...) needs a single-launch geometry")
        launches, need_zero = plan_dgrad(gy.shape, w, g, in_hw)
        if len(launches) != 1 or need_zero:
            raise RuntimeError("conv_dgrad_raw(resid=...) needs a single-launch geometry")
        gx = torch.empty((gy.shape[0], w.shape[1], in_hw[0], in_hw[1]), device=gy.device, dtype=gy.dtype, memory_format=CL)
        resid = _nhwc(resid)
        launch_fwd(gx, gy, launches[0], gain, lin, lout, resid=resid if resid.dtype == gy.dtype else resid.to(gy.dtype), resid_gain=1.0)
        return gx
    if g.reflect:
        # gradient w.r.t. the reflect-padded input, then fold the mirrored border back
        gp = ConvGeom(g.kh, g.kw, g.stride, 0, False)
        ph, pw = in_hw[0] + 2 * g.pad, in_hw[1] + 2 * g.pad
        gxp = conv_dgrad_raw(gy, w, gp, (ph, pw), gain, lin, lout)
        b, c = gxp.shape[0], gxp.shape[1]
        gx = torch.empty((b, c, in_hw[0], in_hw[1]), device=gxp.device, dtype=gxp.dtype, memory_format=CL)
        if gxp.dtype == BF and c % 4:
            return conv_dgrad_fold32(gxp, in_hw, g.pad)
        rc = _lib.load().ideas_reflect_fold(_lib.ptr(gx), _lib.ptr(gxp), b, in_hw[0], in_hw[1], c, g.pad, _lib.act_dtype(gxp),
                                            _lib.stream_ptr())
        _lib.check(rc, "ideas_reflect_fold")
        return gx
    f32 = gy.dtype == torch.float32
    if f32 and B3_WINO_DGRAD and in_hw == (gy.shape[2], gy.shape[3]) and _b3_wino_ok(g, gy.shape[1], w.shape[1], gy.shape[3]):
        b, co, h, wd = gy.shape
        ci = w.shape[1]
        gx = torch.empty((b, ci, h, wd), device=gy.device, dtype=gy.dtype, memory_format=CL)
        launch_wino(gx, gy, b3_wino_planes(w, True), b, co, h, wd, ci, gain, False, lin, lout, dtype=_lib.F32_B3)
        return gx
    if f32 and in_hw == (gy.shape[2], gy.shape[3]) and _wino_ok(g, gy.shape[1], gy.shape[3]):
        # dgrad of a 3x3/s1/p1 conv = the same conv with the taps flipped and the channel roles swapped
        b, co, h, wd = gy.shape
        ci = w.shape[1]
        gx = torch.empty((b, ci, h, wd), device=gy.device, dtype=gy.dtype, memory_format=CL)
        u = wino_weights(w.flip(2, 3).permute(1, 2, 3, 0))       # [I, ky', kx', O]
        launch_wino(gx, gy, u, b, co, h, wd, ci, gain, False, lin, lout)
        return gx
    launches, need_zero = plan_dgrad(gy.shape, w, g, in_hw)
    b, ci = gy.shape[0], w.shape[1]
    gx = torch.empty((b, ci, in_hw[0], in_hw[1]), device=gy.device, dtype=gy.dtype, memory_format=CL)
    if need_zero:
        gx.zero_()
    if not launch_multi(gx, gy, launches, gain, lin, lout):
        for L in launches:
            launch_fwd(gx, gy, L, gain, lin, lout)
    return gx


def conv_dgrad_fold32(gxp: torch.Tensor, in_hw, pad: int) -> torch.Tensor:
    """bf16 reflect fold for channel counts without a bf16 kernel (C % 4 != 0: the N-channel / RGB ends): f32 kernel on a cast."""
    g32 = gxp.float()
    b, c = g32.shape[0], g32.shape[1]
    gx = torch.empty((b, c, in_hw[0], in_hw[1]), device=g32.device, dtype=torch.float32, memory_format=CL)
    rc = _lib.load().ideas_reflect_fold(_lib.ptr(gx), _lib.ptr(g32), b, in_hw[0], in_hw[1], c, pad, _lib.F32, _lib.stream_ptr())
    _lib.check(rc, "ideas_reflect_fold")
    return gx.to(BF)


_GU = {}


def _wino_gu_scratch(n: int, device) -> torch.Tensor:
    """ZEROED f32 scratch for a Winograd-domain gradient dU, one per (stream, size) (a fold with ``clear`` re-zeroes it behind its read)."""
    key = (_lib.stream_ptr(), n, str(device))
    buf = _GU.get(key)
    if buf is None:
        buf = _GU[key] = torch.zeros(n, device=device, dtype=torch.float32)
    return buf


def _wino_fold(gu, out, co: int, ci: int, w_shape, device, clear: bool = True):
    """dU [4, O, 3, I] -> the 3x3 taps added into ``out`` (any strides) or into a fresh zeroed OHWI gradient."""
    tgt = out if out is not None else torch.zeros(tuple(w_shape), device=device, dtype=torch.float32).contiguous(memory_format=CL)
    so, si, sky, skx = tgt.stride()
    rc = _lib.load().ideas_wino_wgrad_fold(_lib.ptr(tgt), _lib.ptr(gu), co, ci, so, sky, skx, si, int(clear), _lib.stream_ptr())
    _lib.check(rc, "ideas_wino_wgrad_fold")
    return tgt


PRESCALE_MOD_PIX = int(_os.environ.get("IDEAS_PRESCALE_MOD_PIX", "256"))


def conv_wgrad_raw(gy, x, g: ConvGeom, w_shape, gain: float, lin=None, lout=None, out=None):
    """Weight gradient [O,I,KH,KW] (channels_last, i.e. OHWI in memory).  lin scales x, lout scales gy.
    ``out`` (an OHWI-contiguous tensor of that shape): ADD the gradient to it instead of returning a new tensor — the
    split-K kernels accumulate with atomics anyway, so this costs neither a zero-fill nor an add pass."""
    gy, x = _nhwc(gy), _nhwc(x)
    if gy.dtype != x.dtype:               # (double-backward corner: a f32 cotangent meeting a bf16 activation)
        gy, x = gy.float(), x.float()
    if (lin is None) != (lout is None):   # the MFMA wgrad kernels take both per-sample scales or neither
        if lin is None:
            lin = torch.ones((x.shape[0], x.shape[1]), device=x.device, dtype=torch.float32)
        else:
            lout = torch.ones((gy.shape[0], gy.shape[1]), device=x.device, dtype=torch.float32)
    L = plan_wgrad(x.shape, gy.shape, g)
    if x.dtype == BF and lin is not None and gy.shape[2] * gy.shape[3] <= PRESCALE_MOD_PIX:
        # small images of a modulated layer (bf16): the kernels apply the per-sample scales to the accumulators, so a split-K block
        # cannot cross a sample and a 16x16 image gives each block 8 K-steps under a 128x128 atomic epilogue (145 TFLOP/s).
        # Scale the (few-MB) operands instead and run the unscaled kernel, whose splits are free to span samples.
        x = (x * lin[:, :, None, None]).to(BF)
        gy = (gy * lout[:, :, None, None]).to(BF)
        lin = lout = None
    b3 = x.dtype == BF or (MATH == _lib.F32_B3 and bool(_lib.load().ideas_b3_wgrad_supported(C.byref(_params(L, gain)))))
    if not b3 and _wino_ok(g, x.shape[1], x.shape[3], fwd=False) and gy.shape[1] % 4 == 0 and tuple(w_shape[2:]) == (3, 3):
        b, ci, h, wd = x.shape
        co = gy.shape[1]
        gu = _wino_gu_scratch(4 * co * 3 * ci, x.device)          # zeroed once; the fold re-zeroes it behind its read
        p = _lib.ConvParams(b, h, wd, ci, h, wd, co, h, wd, 3, 3, 1, 1, 1, 1, -1, -1, 1, 1, 0, 0, int(g.reflect), 0, 0.2,
                            1.0, 1.0, 0, gain)
        rc = _lib.load().ideas_conv3x3_wino_wgrad(_lib.ptr(gu), _lib.ptr(gy), _lib.ptr(x), _lib.ptr(lin), _lib.ptr(lout),
                                                  C.byref(p), _lib.F32, _lib.stream_ptr())
        _lib.check(rc, "ideas_conv3x3_wino_wgrad")
        return _wino_fold(gu, out, co, ci, w_shape, x.device, clear=True)
    if (lin is None) != (lout is None):   # the MFMA wgrad kernel takes both per-sample scales or neither
        if lin is None:
            lin = torch.ones((x.shape[0], x.shape[1]), device=x.device, dtype=torch.float32)
        else:
            lout = torch.ones((gy.shape[0], gy.shape[1]), device=x.device, dtype=torch.float32)
    mfma = (L.Cin % 4 == 0) and (L.Cout % 4 == 0)       # the atomics-accumulating kernels
    if out is not None and mfma and tuple(out.shape) == tuple(w_shape) and out.is_contiguous(memory_format=CL):
        launch_wgrad(out, gy, x, L, gain, lin, lout)
        return out
    gw = torch.empty(tuple(w_shape), device=x.device, dtype=torch.float32, memory_format=CL).zero_()
    launch_wgrad(gw, gy, x, L, gain, lin, lout)
    return gw if out is None else out.add_(gw)


# ----------------------------------------------------------------------------------------------------
# Gradient sink.  Inside ``with grad_sink(params):`` the weight gradients of those parameters are produced on a side
# stream and accumulated straight into their (pre-existing, e.g. flat-bucket) ``.grad`` — the Functions return None
# for the weight, so autograd neither allocates, zero-fills nor adds.  The weight-gradient kernels (MFMA-bound) then
# overlap the HBM-bound elementwise backward passes of the following layers on the main stream.  Leaving the context
# joins the side stream.  Only plain backward passes qualify (no create_graph), and only parameters named by the caller:
# a Function cannot see the ``inputs=`` filter of ``torch.autograd.backward``.
# ----------------------------------------------------------------------------------------------------
_SINK = {"ids": None, "stream": None}
# IDEAS_SINK_PRIORITY=low puts the side stream on the device's LOWEST priority (ideas_stream_create).  Measured round 5, same box,
# interleaved: f32 412.2 -> 413.6 ms, bf16 154.7 -> 154.2 ms (noise) -- and the same with the whole iteration on a highest-priority
# stream (414.2 ms).  A kernel trace shows 20-us torch adds of the main stream taking up to 1.9 ms next to a weight-gradient grid, but
# the chip is busy throughout: the queue priority changes who waits, not how much work the compute units retire.  Default: off.
SINK_LOW_PRIORITY = _os.environ.get("IDEAS_SINK_PRIORITY", "default") == "low"


class grad_sink:
    """``defer=True``: leaving the context does NOT join the side stream; the caller does (``join()``) before it consumes the
    gradients.  The D phase's weight gradients are wanted only by the discriminators' optimiser step, which the step defers to the
    first discriminator call of the G phase (train_step._Deferred) -- until then they may keep running under the generator
    forwards of the G phase instead of holding the main stream at the end of the backward pass."""

    def __init__(self, params, defer: bool = False):
        self.ids = {id(p) for p in params if p.grad is not None and p.is_cuda}
        self.defer = defer
        self.pending = False

    def __enter__(self):
        if not self.ids:            # nothing to sink (no pre-existing device gradients): plain autograd
            return self
        if _SINK["stream"] is None:
            _SINK["stream"] = _lib.make_stream(-1) if SINK_LOW_PRIORITY else torch.cuda.Stream()
        _SINK["ids"] = self.ids
        return self

    def __exit__(self, *exc):
        if _SINK["ids"] is not None:
            _SINK["ids"] = None
            if self.defer and exc[0] is None:
                self.pending = True
            else:
                torch.cuda.current_stream().wait_stream(_SINK["stream"])

    def join(self):
        if self.pending:
            self.pending = False
            torch.cuda.current_stream().wait_stream(_SINK["stream"])


def _sink_target(w: torch.Tensor):
    ids = _SINK["ids"]
    if ids is None or torch.is_grad_enabled():
        return None
    base = w._base if w._base is not None else w
    if id(base) not in ids:
        return None
    gr = base.grad
    if gr is None or gr.shape != base.shape or gr.stride() != base.stride():
        return None
    if base is w:
        return gr
    if w.numel() != base.numel() or w.data_ptr() != base.data_ptr():
        return None
    return gr.as_strided(w.shape, w.stride())


_SIDE_STREAM = _os.environ.get("IDEAS_SIDE_STREAM", "1") != "0"


def weight_grad(w: torch.Tensor, compute, *uses):
    """``compute(out)`` -> the gradient of ``w`` (added to ``out`` when that is not None).  Returns it, or None after
    sinking it into ``w.grad`` on the side stream (``uses``: the tensors the kernels read, for the allocator)."""
    tgt = _sink_target(w)
    if tgt is None:
        return compute(None)
    if not _SIDE_STREAM:                 # (A/B only: same in-place accumulation, on the current stream)
        compute(tgt)
        return None
    side, cur = _SINK["stream"], torch.cuda.current_stream()
    side.wait_stream(cur)
    for t in uses:
        if t is not None:
            t.record_stream(side)
    with torch.cuda.stream(side):
        compute(tgt)
    return None


# ----------------------------------------------------------------------------------------------------
# dense convolution with double backward
# ----------------------------------------------------------------------------------------------------

class _Conv(Function):
    """y = gain * conv(x, w).  backward -> _ConvDgrad / _ConvWgrad (both differentiable)."""

    @staticmethod
    def forward(ctx, x, w, g: ConvGeom, gain: float):
        ctx.g, ctx.gain = g, gain
        x = _nhwc(x)
        ctx.save_for_backward(x, w)
        return conv_fwd_raw(x, w, g, gain)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = _ConvDgrad.apply(gy, w, ctx.g, ctx.gain, (x.shape[2], x.shape[3]))
        if ctx.needs_input_grad[1]:
            gw = _wgrad(w, gy, x, ctx.g, ctx.gain)
        return gx, gw, None, None


class _ConvAdd(Function):
    """y = gain * conv(x, w) + r with the add in the conv epilogue (the residual merge of a ResBlock: one elementwise pass
    over the block output saved).  Same single rounding of the sum as conv -> torch.add, so bitwise the unfused result."""

    @staticmethod
    def forward(ctx, x, w, r, g: ConvGeom, gain: float):
        ctx.g, ctx.gain = g, gain
        x = _nhwc(x)
        ctx.save_for_backward(x, w)
        return conv_fwd_raw(x, w, g, gain, resid=r, resid_gain=1.0)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = _ConvDgrad.apply(gy, w, ctx.g, ctx.gain, (x.shape[2], x.shape[3]))
        if ctx.needs_input_grad[1]:
            gw = _wgrad(w, gy, x, ctx.g, ctx.gain)
        return gx, gw, (gy if ctx.needs_input_grad[2] else None), None, None


def _wgrad(w, gy, x, g: ConvGeom, gain: float):
    """Weight gradient of a dense conv: differentiable Function normally, sunk into w.grad inside grad_sink."""
    if _sink_target(w) is None:
        return _ConvWgrad.apply(gy, x, g, gain, tuple(w.shape))
    gy, x = _nhwc(gy), _nhwc(x)
    return weight_grad(w, lambda out: conv_wgrad_raw(gy, x, g, tuple(w.shape), gain, out=out), gy, x)


class _ConvDgrad(Function):
    """gx = gain * conv^T(gy, w).  Linear in (gy, w): its backward is the forward conv and a wgrad."""

    @staticmethod
    def forward(ctx, gy, w, g: ConvGeom, gain: float, in_hw):
        ctx.g, ctx.gain, ctx.in_hw = g, gain, in_hw
        gy = _nhwc(gy)
        ctx.save_for_backward(gy, w)
        return conv_dgrad_raw(gy, w, g, in_hw, gain)

    @staticmethod
    def backward(ctx, ggx):
        gy, w = ctx.saved_tensors
        g_gy = g_w = None
        if ctx.needs_input_grad[0]:
            g_gy = _Conv.apply(ggx, w, ctx.g, ctx.gain)
        if ctx.needs_input_grad[1]:
            g_w = _wgrad(w, gy, ggx, ctx.g, ctx.gain)
        return g_gy, g_w, None, None, None


class _ConvWgrad(Function):
    """gw = gain * sum gy (x) x.  Bilinear in (gy, x)."""

    @staticmethod
    def forward(ctx, gy, x, g: ConvGeom, gain: float, w_shape):
        ctx.g, ctx.gain = g, gain
        gy, x = _nhwc(gy), _nhwc(x)
        ctx.save_for_backward(gy, x)
        return conv_wgrad_raw(gy, x, g, w_shape, gain)

    @staticmethod
    def backward(ctx, ggw):
        gy, x = ctx.saved_tensors
        g_gy = g_x = None
        if ctx.needs_input_grad[0]:
            g_gy = _Conv.apply(x, ggw, ctx.g, ctx.gain)
        if ctx.needs_input_grad[1]:
            g_x = _ConvDgrad.apply(gy, ggw, ctx.g, ctx.gain, (x.shape[2], x.shape[3]))
        return g_gy, g_x, None, None, None


class _ForkConv(Function):
    """x -> (x, gain * conv(x, w)) for a stride-1 conv without mirror padding (the 1x1 conv an upsampling skip branch starts with,
    models.py:170-172 after the reordering of ConvLayer.forward): owning the fork lets the backward add the gradient of the other
    use of x in the input-gradient kernel's epilogue instead of autograd's separate accumulation pass."""

    @staticmethod
    def forward(ctx, x, w, g: ConvGeom, gain: float):
        ctx.g, ctx.gain = g, gain
        x = _nhwc(x)
        ctx.save_for_backward(x, w)
        ctx.set_materialize_grads(False)
        return x.view_as(x), conv_fwd_raw(x, w, g, gain)

    @staticmethod
    def backward(ctx, ga, gh):
        x, w = ctx.saved_tensors
        if gh is None:
            return ga, None, None, None
        gx = gw = None
        if ctx.needs_input_grad[0]:
            if ga is not None and not torch.is_grad_enabled() and ga.is_contiguous(memory_format=CL) and ga.dtype == x.dtype:
                gx = conv_dgrad_raw(_nhwc(gh), w, ctx.g, (x.shape[2], x.shape[3]), ctx.gain, resid=ga)
            else:
                gx = _ConvDgrad.apply(gh, w, ctx.g, ctx.gain, (x.shape[2], x.shape[3]))
                gx = gx if ga is None else gx + ga
        if ctx.needs_input_grad[1]:
            gw = _wgrad(w, gh, x, ctx.g, ctx.gain)
        return gx, gw, None, None


def fork_conv2d(input: torch.Tensor, weight: torch.Tensor, padding: int = 0, gain: float = 1.0):
    """``(input, gain * F.conv2d(input, weight, padding=padding))`` with the gradient sum of the two uses of ``input`` fused into
    the conv's input-gradient kernel."""
    _lib.require_cuda(input, weight)
    g = ConvGeom(weight.shape[2], weight.shape[3], 1, padding, False)
    return _ForkConv.apply(to_act(input), weight, g, float(gain))


_BIAS_SUM_HIP = _os.environ.get("IDEAS_BIAS_SUM_HIP", "1") != "0"      # 0: autograd's composite sum (A/B only)


class _AddBias(Function):
    """y + bias[c] with the bias gradient on ``ideas_channel_sum`` (autograd's sum over (0, 2, 3) of a channels_last tensor with
    C = 3 ran as ONE block: 1.0 ms per G.to_rgb backward).  Under create_graph the backward is the differentiable composite.
    The sum follows torch's type promotion: a bf16 activation plus the f32 parameter is an f32 tensor (G's image and Ex's message
    keep the f32 bias and one rounding, as before this Function existed); the incoming gradient has that dtype."""

    @staticmethod
    def forward(ctx, y, bias):
        ctx.bias_ref = bias
        ctx.y_dtype = y.dtype
        return y + bias.view(1, -1, 1, 1)

    @staticmethod
    def backward(ctx, gy):
        gb = None
        if ctx.needs_input_grad[1]:
            if torch.is_grad_enabled() or not _BIAS_SUM_HIP:
                gb = gy.sum((0, 2, 3)).to(ctx.bias_ref.dtype)
            else:
                from .fused_act import bias_sink, channel_sum
                tgt = bias_sink(ctx.bias_ref)
                gb = channel_sum(gy, into=tgt)
                gb = None if gb is None else gb.to(ctx.bias_ref.dtype)
        return (gy if gy.dtype == ctx.y_dtype else gy.to(ctx.y_dtype)), gb


def conv2d(input: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, stride: int = 1,
           padding: int = 0, reflect: bool = False, gain: float = 1.0, resid: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``gain * F.conv2d(input, weight, stride, padding) + bias`` (zero padding, or mirror padding if ``reflect``);
    ``resid``: a tensor of the output's shape added in the conv epilogue (differentiable)."""
    _lib.require_cuda(input, weight, bias)
    g = ConvGeom(weight.shape[2], weight.shape[3], stride, padding, reflect)
    if resid is not None:
        if bias is not None:
            raise RuntimeError("conv2d(resid=...) with a conv bias is not used on this path")
        return _ConvAdd.apply(to_act(input), weight, to_act(resid), g, float(gain))
    y = _Conv.apply(to_act(input), weight, g, float(gain))
    if bias is not None:
        y = _AddBias.apply(y, bias)
    return y


class _ConvBiasAct(Function):
    """y = lrelu(gain * conv(x, w) + b) * act_gain with bias + activation in the conv epilogue (one pass saved).
    Same op order as the separate fused_bias_act kernel, so results are bitwise those of conv2d -> fused_leaky_relu.
    The backward is composed of differentiable Functions, so double backward (R1) still works."""

    @staticmethod
    def forward(ctx, x, w, b, g: ConvGeom, gain: float, slope: float, act_gain: float):
        x = _nhwc(x)
        y = conv_fwd_raw(x, w, g, gain, bias=b.contiguous(), act=True, act_gain=act_gain, alpha=slope)
        ctx.g, ctx.gain, ctx.slope, ctx.act_gain = g, gain, slope, act_gain
        ctx.bias_ref = b
        ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, gy):
        from .fused_act import FusedLeakyReLUFunctionBackward
        x, w, y = ctx.saved_tensors
        from .fused_act import bias_act_raw, bias_sink
        tgt = bias_sink(ctx.bias_ref) if ctx.needs_input_grad[2] else None
        if tgt is not None:          # gradient sink: bias gradient accumulated by the kernel into bias.grad
            g_pre, gb = bias_act_raw(gy, None, y, 1, ctx.slope, ctx.act_gain, bias_grad_into=tgt)
        else:
            # (a frozen layer -- the discriminators in the G phase -- needs no bias gradient: no zero-fill, no reduction in the kernel)
            g_pre, gb = FusedLeakyReLUFunctionBackward.apply(gy, y, ctx.slope, ctx.act_gain, bool(ctx.needs_input_grad[2]))
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = _ConvDgrad.apply(g_pre, w, ctx.g, ctx.gain, (x.shape[2], x.shape[3]))
        if ctx.needs_input_grad[1]:
            gw = _wgrad(w, g_pre, x, ctx.g, ctx.gain)
        return gx, gw, (gb if (ctx.needs_input_grad[2] and tgt is None) else None), None, None, None, None


class _ConvBiasActBlur(Function):
    """upfirdn2d(lrelu(gain * conv(x, w) + b) * act_gain, fir, pad): conv1 of a downsampling ResBlock together with the Blur that
    opens its conv2 (models.py:185-187, 69).  Forward = the two kernels of the unfused chain (bitwise the same values).  Backward of
    a plain pass: ONE kernel takes the blur's adjoint, applies the leaky-ReLU mask of the saved activation and reduces the bias
    gradient (ideas_blur_fused) -- the gradient of the blurred tensor is never written back and re-read; when a graph is being
    built (R1's double backward) the backward is composed of the differentiable Functions instead."""

    @staticmethod
    def forward(ctx, x, w, b, g: ConvGeom, gain: float, slope: float, act_gain: float, fir, pad2):
        from .upfirdn2d import blur_geometry, upfirdn2d_raw
        x = _nhwc(x)
        y1 = conv_fwd_raw(x, w, g, gain, bias=b.contiguous(), act=True, act_gain=act_gain, alpha=slope)
        pad4, out_hw, g_pad = blur_geometry((y1.shape[2], y1.shape[3]), fir, pad2)
        yb = upfirdn2d_raw(y1, fir, (1, 1), (1, 1), pad4, out_hw, flip=True)
        ctx.g, ctx.gain, ctx.slope, ctx.act_gain = g, gain, slope, act_gain
        ctx.pad4, ctx.g_pad, ctx.out_hw = pad4, g_pad, out_hw
        ctx.bias_ref = b
        ctx.save_for_backward(x, w, y1, fir)
        return yb

    @staticmethod
    def backward(ctx, gyb):
        from .fused_act import FusedLeakyReLUFunctionBackward, bias_sink
        from .upfirdn2d import BLUR_ACT_BWD, UpFirDn2dBackward, blur_fused_ok, blur_fused_raw
        x, w, y1, fir = ctx.saved_tensors
        need_b = ctx.needs_input_grad[2]
        gb = None
        if torch.is_grad_enabled() or not blur_fused_ok(y1, fir):
            g1 = UpFirDn2dBackward.apply(gyb, fir, (1, 1), (1, 1), ctx.pad4, ctx.g_pad, tuple(y1.shape), ctx.out_hw)
            g_pre, gb = FusedLeakyReLUFunctionBackward.apply(g1, y1, ctx.slope, ctx.act_gain, bool(need_b))
        else:
            tgt = bias_sink(ctx.bias_ref) if need_b else None
            if tgt is None:
                gb = torch.zeros(y1.shape[1], device=y1.device, dtype=torch.float32)
            g_pre = blur_fused_raw(_nhwc(gyb), fir, ctx.g_pad, (y1.shape[2], y1.shape[3]), False, BLUR_ACT_BWD, ref=y1,
                                   bias_grad=tgt if tgt is not None else gb, alpha=ctx.slope, scale=ctx.act_gain)
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = _ConvDgrad.apply(g_pre, w, ctx.g, ctx.gain, (x.shape[2], x.shape[3]))
        if ctx.needs_input_grad[1]:
            gw = _wgrad(w, g_pre, x, ctx.g, ctx.gain)
        return gx, gw, (gb if need_b else None), None, None, None, None, None, None


# ----------------------------------------------------------------------------------------------------
# Blur -> 3x3 / stride-2 conv (+ bias + leaky-ReLU) of a downsampling ConvLayer in ONE kernel (csrc/conv_b3_s2fir.hip): the
# blurred tensor is built in LDS under each output patch and never makes the round trip through HBM.  IDEAS_BLUR_CONV=0 keeps the
# blur a separate pass (A/B measurements).
# ----------------------------------------------------------------------------------------------------
BLUR_CONV = _os.environ.get("IDEAS_BLUR_CONV", "1") != "0"
BLUR_CONV_MIN_OW = int(_os.environ.get("IDEAS_BLUR_CONV_MIN_OW", "16"))      # below: the 8 x 16 output patch of the kernel would idle
BLUR_CONV_MIN_BLOCKS = int(_os.environ.get("IDEAS_BLUR_CONV_MIN_BLOCKS", "512"))   # down_pair_ok: below, the two-kernel chain wins


def fir_factors(fir: torch.Tensor, flip: bool = True):
    """Host-side factors (kh[4], kv[4]) of a separable 4x4 FIR, derived the way the stand-alone blur kernel derives them on the
    device (upfirdn2d.hip::blur4_f32_c2: kh = first row of the flipped table, kv = first column / its first element, in f32), or
    None for a rank > 1 table.  One synchronising copy per FIR tensor: the tables are constant module buffers (stylegan2/model.py:84)."""
    hit = getattr(fir, "_ideas_factors", None)
    if hit is not None and hit[0] == (fir._version, flip, fir.data_ptr()):
        return hit[1]
    import numpy as np
    res = None
    if tuple(fir.shape) == (4, 4):
        f = fir.detach().to(torch.float32).cpu().numpy().astype(np.float32)
        if flip:
            f = f[::-1, ::-1]
        kh = f[0].astype(np.float32)
        kv = (f[:, 0] / f[0, 0]).astype(np.float32) if f[0, 0] != 0 else np.zeros(4, np.float32)
        if float(np.abs(f - np.outer(kv, kh)).max()) <= 1e-6 * float(np.abs(f).max()):
            res = ((C.c_float * 4)(*[float(v) for v in kh]), (C.c_float * 4)(*[float(v) for v in kv]))
    try:
        fir._ideas_factors = ((fir._version, flip, fir.data_ptr()), res)
    except Exception:      # (a tensor subclass without a __dict__)
        pass
    return res


def _blur_conv_plan(x_shape, w: torch.Tensor, fir: torch.Tensor, pad2):
    """(Launch of the stride-2 conv on the blurred tensor, blurred height, width), or None when the fused kernel does not apply."""
    if not BLUR_CONV or MATH != _lib.F32_B3 or tuple(w.shape[2:]) != (3, 3) or tuple(fir.shape) != (4, 4):
        return None
    b, ci, h, wd = x_shape
    p0, p1 = int(pad2[0]), int(pad2[1])
    hb, wb = h + p0 + p1 - 3, wd + p0 + p1 - 3
    if hb < 3 or wb < 3 or ci != w.shape[1]:
        return None
    L = plan_fwd((b, ci, hb, wb), w, ConvGeom(3, 3, 2, 0, False))
    if L.OW < BLUR_CONV_MIN_OW or L.OH < 8:
        return None
    if not _lib.load().ideas_b3_blur_conv_s2_supported(C.byref(_params(L, 1.0)), h, wd, p0):
        return None
    return L, hb, wb


def blur_conv_s2_ok(x: torch.Tensor, w: torch.Tensor, fir: torch.Tensor, pad2, want_xb: bool = False) -> bool:
    if x.dtype != torch.float32 or not x.is_cuda:
        return False
    pl = _blur_conv_plan(tuple(x.shape), w, fir, pad2)
    if pl is None or fir_factors(fir) is None:
        return False
    L, hb, wb = pl
    return (not want_xb) or (hb == 2 * L.OH + 1 and wb == 2 * L.OW + 1)


def blur_conv_s2_raw(x, w, fir, pad2, gain: float, bias=None, act: bool = False, act_gain: float = 1.0, alpha: float = 0.2,
                     resid=None, want_xb: bool = False):
    """``epilogue(gain * conv2d(upfirdn2d(x, fir, pad=pad2), w, stride=2))`` in one launch -> (y, blurred tensor or None).
    The caller has checked ``blur_conv_s2_ok``."""
    x = _nhwc(x)
    L, hb, wb = _blur_conv_plan(tuple(x.shape), w, fir, pad2)
    kh, kv = fir_factors(fir)
    p = _params(L, gain, False, act, alpha, act_gain, 1.0)
    y = torch.empty((L.B, L.Cout, L.YH, L.YW), device=x.device, dtype=x.dtype, memory_format=CL)
    xb = torch.empty((L.B, L.Cin, hb, wb), device=x.device, dtype=x.dtype, memory_format=CL) if want_xb else None
    if resid is not None:
        resid = _nhwc(resid)
    rc = _lib.load().ideas_b3_blur_conv_s2(_lib.ptr(y), _lib.ptr(xb), _lib.ptr(x), _lib.ptr(b3_planes(L)), kh, kv,
                                           _lib.ptr(None if bias is None else bias.contiguous()), _lib.ptr(resid), C.byref(p),
                                           x.shape[2], x.shape[3], int(pad2[0]), _lib.stream_ptr())
    _lib.check(rc, "ideas_b3_blur_conv_s2")
    return y, xb


class _DownPair(Function):
    """The body of a downsampling ResBlock (models.py:185-187 with ConvLayer :68-76, :120-128) as one autograd node:

        y1 = lrelu(gain1 * conv(x, w1) + b1) * ag1                      conv1 (3x3, zero or mirror padding)
        y2 = lrelu(gain2 * conv_s2(blur(y1), w2) + b2) * ag2            conv2's Blur + stride-2 conv + activation: ONE kernel

    The blurred tensor exists only as the side output the weight gradient of w2 reads (written by the conv kernel, not by a blur
    pass), and not at all when w2 is frozen (the discriminators in the G phase).  Backward of a plain pass: leaky-ReLU backward of
    y2 -> input gradient of the stride-2 conv -> blur adjoint + leaky-ReLU mask of y1 + bias gradient in one kernel
    (ideas_blur_fused) -> conv1's gradients; under create_graph (R1) the same chain composed of the differentiable Functions."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, g1: ConvGeom, gain1: float, slope1: float, ag1: float, fir, pad2, gain2: float,
                slope2: float, ag2: float):
        from .upfirdn2d import blur_geometry
        x = _nhwc(x)
        y1 = conv_fwd_raw(x, w1, g1, gain1, bias=b1.contiguous(), act=True, act_gain=ag1, alpha=slope1)
        need_xb = ctx.needs_input_grad[3]
        y2, yb = blur_conv_s2_raw(y1, w2, fir, pad2, gain2, bias=b2, act=True, act_gain=ag2, alpha=slope2, want_xb=need_xb)
        pad4, out_hw, g_pad = blur_geometry((y1.shape[2], y1.shape[3]), fir, pad2)
        ctx.g1, ctx.gain1, ctx.slope1, ctx.ag1 = g1, gain1, slope1, ag1
        ctx.gain2, ctx.slope2, ctx.ag2 = gain2, slope2, ag2
        ctx.pad4, ctx.g_pad, ctx.out_hw = pad4, g_pad, out_hw
        ctx.b1_ref, ctx.b2_ref = b1, b2
        ctx.save_for_backward(x, w1, y1, w2, y2, fir, yb)
        return y2

    @staticmethod
    def backward(ctx, gy2):
        from .fused_act import FusedLeakyReLUFunctionBackward, bias_act_raw, bias_sink
        from .upfirdn2d import BLUR_ACT_BWD, UpFirDn2dBackward, blur_fused_ok, blur_fused_raw, upfirdn2d_raw
        x, w1, y1, w2, y2, fir, yb = ctx.saved_tensors
        need = ctx.needs_input_grad
        g2 = ConvGeom(3, 3, 2, 0, False)
        gb1 = gb2 = None
        tgt2 = bias_sink(ctx.b2_ref) if need[4] else None
        if tgt2 is not None:
            g_pre2, _ = bias_act_raw(gy2, None, y2, 1, ctx.slope2, ctx.ag2, bias_grad_into=tgt2)
        else:
            g_pre2, gb2 = FusedLeakyReLUFunctionBackward.apply(gy2, y2, ctx.slope2, ctx.ag2, bool(need[4]))
        gw2 = None
        if need[3]:
            if yb is None:      # (cannot happen: w2 required a gradient in the forward, so the side output was written)
                yb = upfirdn2d_raw(y1, fir, (1, 1), (1, 1), ctx.pad4, ctx.out_hw, flip=True)
            gw2 = _wgrad(w2, g_pre2, yb, g2, ctx.gain2)
        gyb = _ConvDgrad.apply(g_pre2, w2, g2, ctx.gain2, ctx.out_hw)
        if torch.is_grad_enabled() or not blur_fused_ok(y1, fir):
            g1 = UpFirDn2dBackward.apply(gyb, fir, (1, 1), (1, 1), ctx.pad4, ctx.g_pad, tuple(y1.shape), ctx.out_hw)
            g_pre1, gb1 = FusedLeakyReLUFunctionBackward.apply(g1, y1, ctx.slope1, ctx.ag1, bool(need[2]))
        else:
            tgt1 = bias_sink(ctx.b1_ref) if need[2] else None
            if tgt1 is None:
                gb1 = torch.zeros(y1.shape[1], device=y1.device, dtype=torch.float32)
            g_pre1 = blur_fused_raw(_nhwc(gyb), fir, ctx.g_pad, (y1.shape[2], y1.shape[3]), False, BLUR_ACT_BWD, ref=y1,
                                    bias_grad=tgt1 if tgt1 is not None else gb1, alpha=ctx.slope1, scale=ctx.ag1)
            if tgt1 is not None:
                gb1 = None
        gx = gw1 = None
        if need[0]:
            gx = _ConvDgrad.apply(g_pre1, w1, ctx.g1, ctx.gain1, (x.shape[2], x.shape[3]))
        if need[1]:
            gw1 = _wgrad(w1, g_pre1, x, ctx.g1, ctx.gain1)
        return (gx, gw1, (gb1 if need[2] else None), gw2, (gb2 if (need[4] and tgt2 is None) else None),
                None, None, None, None, None, None, None, None, None)


def down_pair_ok(input: torch.Tensor, w1, w2, fir, pad2, padding1: int = 1) -> bool:
    """Shapes / precision ``down_pair`` covers: f32 activations (the bf16 path keeps its own kernels), the split-bf16 contraction,
    a separable 4x4 FIR, >= 8 x 16 output pixels per image, Cin % 16 == 0; with a weight gradient pending also the blurred size
    2 OH + 1 (every blurred pixel is then written by the conv kernel's side output)."""
    from ..precision import activation_dtype
    if not input.is_cuda or activation_dtype() != torch.float32 or input.dim() != 4:
        return False
    g1 = ConvGeom(w1.shape[2], w1.shape[3], 1, padding1, False)
    oh, ow = g1.out_size(input.shape[2], input.shape[3])
    pl = _blur_conv_plan((input.shape[0], w1.shape[0], oh, ow), w2, fir, pad2)
    if pl is None or fir_factors(fir) is None:
        return False
    if torch.is_grad_enabled() and w2.requires_grad and not (pl[1] == 2 * pl[0].OH + 1 and pl[2] == 2 * pl[0].OW + 1):
        return False
    # Fewer than two workgroups per CU: every block's producers repeat the blur for its N tile and nothing hides the tail, the
    # blur kernel + generic stride-2 conv is as fast or faster (profiles/r04_blur_conv_microbench.txt: E.4.conv2 0.29 against 0.25 ms,
    # Dreal.4.conv2 a tie)
    L = pl[0]
    nt = 256 if L.Cout > 128 else 128 if L.Cout > 64 else 64
    return L.B * ((L.OH + 7) // 8) * ((L.OW + 15) // 16) * ((L.Cout + nt - 1) // nt) >= BLUR_CONV_MIN_BLOCKS


def down_pair(input: torch.Tensor, w1, b1, w2, b2, fir, pad2, padding1: int = 1, reflect1: bool = False, gain1: float = 1.0,
              gain2: float = 1.0, negative_slope: float = 0.2, scale1: float = 2 ** 0.5, scale2: float = 2 ** 0.5,
              resid: Optional[torch.Tensor] = None):
    """``act2(conv_s2(blur(act1(conv(input, w1)), fir, pad2), w2))``, the body of a downsampling ResBlock, with the Blur inside the
    stride-2 conv's kernel; the caller has checked ``down_pair_ok``.  ``resid`` (no-grad passes only): added in the last epilogue."""
    _lib.require_cuda(input, w1, b1, w2, b2, fir)
    input = to_act(input)
    g1 = ConvGeom(w1.shape[2], w1.shape[3], 1, padding1, reflect1)
    grad = torch.is_grad_enabled() and (input.requires_grad or w1.requires_grad or b1.requires_grad or w2.requires_grad or b2.requires_grad)
    if not grad:
        y1 = conv_fwd_raw(_nhwc(input), w1, g1, float(gain1), bias=b1.contiguous(), act=True, act_gain=float(scale1),
                          alpha=float(negative_slope))
        return blur_conv_s2_raw(y1, w2, fir, pad2, float(gain2), bias=b2, act=True, act_gain=float(scale2),
                                alpha=float(negative_slope), resid=None if resid is None else to_act(resid))[0]
    if resid is not None:
        raise RuntimeError("down_pair(resid=...) is the no-grad fast path")
    return _DownPair.apply(input, w1, b1, w2, b2, g1, float(gain1), float(negative_slope), float(scale1), fir,
                           (int(pad2[0]), int(pad2[1])), float(gain2), float(negative_slope), float(scale2))


def conv2d_bias_act(input: torch.Tensor, weight: torch.Tensor, act_bias: torch.Tensor, stride: int = 1, padding: int = 0,
                    reflect: bool = False, gain: float = 1.0, negative_slope: float = 0.2,
                    scale: float = 2 ** 0.5, resid: Optional[torch.Tensor] = None, post_blur=None) -> torch.Tensor:
    """``fused_leaky_relu(gain * conv2d(input, weight), act_bias, negative_slope, scale)`` in one kernel.
    ``resid`` (inference only): the residual branch, added in the same epilogue.
    ``post_blur = (fir, (pad0, pad1))``: also apply ``upfirdn2d(., fir, pad=pad)`` to the result (the Blur of the next, downsampling,
    layer) so that the backward can fuse the blur's adjoint with the activation's (``_ConvBiasActBlur``)."""
    _lib.require_cuda(input, weight, act_bias)
    g = ConvGeom(weight.shape[2], weight.shape[3], stride, padding, reflect)
    input = to_act(input)
    if post_blur is not None:
        if resid is not None:
            raise RuntimeError("conv2d_bias_act: post_blur and resid are exclusive")
        fir, pad2 = post_blur
        return _ConvBiasActBlur.apply(input, weight, act_bias, g, float(gain), float(negative_slope), float(scale), fir,
                                      (int(pad2[0]), int(pad2[1])))
    if resid is not None:
        resid = to_act(resid)
        if torch.is_grad_enabled() and (input.requires_grad or weight.requires_grad or act_bias.requires_grad or resid.requires_grad):
            raise RuntimeError("conv2d_bias_act(resid=...) is the no-grad fast path")
        return conv_fwd_raw(input, weight, g, float(gain), bias=act_bias.contiguous(), act=True, act_gain=float(scale),
                            alpha=float(negative_slope), resid=resid, resid_gain=1.0)
    return _ConvBiasAct.apply(input, weight, act_bias, g, float(gain), float(negative_slope), float(scale))


class _ConvT(Function):
    """Transposed conv (weight [I,O,KH,KW], stride s, padding 0) = dgrad of the conv reading w as [O'=I, I'=O]."""

    @staticmethod
    def forward(ctx, x, w, g: ConvGeom, gain: float):
        ctx.g, ctx.gain = g, gain
        x = _nhwc(x)
        ctx.save_for_backward(x, w)
        return conv_dgrad_raw(x, w, g, convT_out_size(x.shape[2], x.shape[3], g), gain)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = _Conv.apply(gy, w, ctx.g, ctx.gain)
        if ctx.needs_input_grad[1]:
            gw = _wgrad(w, x, gy, ctx.g, ctx.gain)
        return gx, gw, None, None


def conv_transpose2d(input: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, stride: int = 1,
                     gain: float = 1.0) -> torch.Tensor:
    """``gain * F.conv_transpose2d(input, weight, stride=stride, padding=0) + bias``."""
    _lib.require_cuda(input, weight, bias)
    g = ConvGeom(weight.shape[2], weight.shape[3], stride, 0, False)
    y = _ConvT.apply(to_act(input), weight, g, float(gain))
    if bias is not None:
        y = _AddBias.apply(y, bias)
    return y
