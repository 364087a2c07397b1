"""ctypes loader of the plain-C oracle (oracle/ops_c.c).  Test infrastructure only."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_F = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")


def load():
    lib = C.CDLL(os.path.join(_HERE, "liboracle_ops.so"))
    lib.oracle_fused_bias_act.argtypes = [_F, _F, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_long, C.c_int, C.c_int, C.c_float, C.c_float]
    lib.oracle_upfirdn2d.argtypes = [_F, _F, _F, C.c_long] + [C.c_int] * 8
    lib.oracle_conv2d.argtypes = [_F, _F, _F, C.c_void_p] + [C.c_int] * 8
    lib.oracle_conv_transpose2d.argtypes = [_F, _F, _F] + [C.c_int] * 7
    lib.oracle_modulated_conv2d.argtypes = [_F, _F, _F, _F] + [C.c_int] * 8
    for f in ("oracle_fused_bias_act", "oracle_upfirdn2d", "oracle_conv2d", "oracle_conv_transpose2d", "oracle_modulated_conv2d"):
        getattr(lib, f).restype = None
    return lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def fused_bias_act(x, b, ref, act, grad, alpha, scale):
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    step = int(np.prod(x.shape[2:])) if x.ndim > 2 else 1
    b = None if b is None else np.ascontiguousarray(b, np.float32)
    ref = None if ref is None else np.ascontiguousarray(ref, np.float32)
    load().oracle_fused_bias_act(out, x, _p(b), _p(ref), x.size, x.shape[1], step, act, grad, alpha, scale)
    return out


def upfirdn2d(x, k, up=1, down=1, pad=(0, 0)):
    x = np.ascontiguousarray(x, np.float32)
    k = np.ascontiguousarray(k, np.float32)
    n, c, h, w = x.shape
    oh = (h * up + pad[0] + pad[1] - k.shape[0]) // down + 1
    ow = (w * up + pad[0] + pad[1] - k.shape[1]) // down + 1
    y = np.empty((n, c, oh, ow), np.float32)
    load().oracle_upfirdn2d(y, x, k, n * c, h, w, k.shape[0], k.shape[1], up, down, pad[0], pad[1])
    return y


def conv2d(x, w, bias, stride, pad):
    x, w = np.ascontiguousarray(x, np.float32), np.ascontiguousarray(w, np.float32)
    b, ci, h, ww = x.shape
    co, _, k, _ = w.shape
    y = np.empty((b, co, (h + 2 * pad - k) // stride + 1, (ww + 2 * pad - k) // stride + 1), np.float32)
    bias = None if bias is None else np.ascontiguousarray(bias, np.float32)
    load().oracle_conv2d(y, x, w, _p(bias), b, ci, co, h, ww, k, stride, pad)
    return y


def conv_transpose2d(x, w, stride):
    x, w = np.ascontiguousarray(x, np.float32), np.ascontiguousarray(w, np.float32)
    b, ci, h, ww = x.shape
    _, co, k, _ = w.shape
    y = np.empty((b, co, (h - 1) * stride + k, (ww - 1) * stride + k), np.float32)
    load().oracle_conv_transpose2d(y, x, w, b, ci, co, h, ww, k, stride)
    return y


def modulated_conv2d(x, style, weight, demodulate=True, upsample=False):
    x, style = np.ascontiguousarray(x, np.float32), np.ascontiguousarray(style, np.float32)
    weight = np.ascontiguousarray(weight.reshape(weight.shape[-4:]), np.float32)
    b, ci, h, w = x.shape
    co, _, k, _ = weight.shape
    y = np.empty((b, co, 2 * h + 1, 2 * w + 1) if upsample else (b, co, h, w), np.float32)
    load().oracle_modulated_conv2d(y, x, style, weight, b, ci, co, h, w, k, int(demodulate), int(upsample))
    return y
