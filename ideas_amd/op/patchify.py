"""The crop + bilinear-resize half of ``patchify_image`` (utils.py:127-149) on the HIP kernel ``ideas_patch_resize``:
all boxes of a call in one launch instead of one ``F.interpolate`` per box plus a stack.  No CPU branch."""
from __future__ import annotations

import ctypes as C
from typing import Sequence, Tuple

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib

CL = torch.channels_last


def _boxes_arg(boxes: Sequence[Tuple[int, int, int, int]]):
    flat = [int(v) for bx in boxes for v in bx]
    return (C.c_int * len(flat))(*flat)


class _PatchResize(Function):
    @staticmethod
    def forward(ctx, img, boxes, out_hw):
        _lib.require_cuda(img)
        if img.dim() != 4:
            raise RuntimeError("patch_resize expects a 4-D [B, C, H, W] tensor")
        dt = _lib.act_dtype(img)
        b, c, h, w = img.shape
        x = img if img.is_contiguous(memory_format=CL) else img.contiguous(memory_format=CL)
        n = len(boxes)
        y = torch.empty((b * n, c, out_hw[0], out_hw[1]), device=img.device, dtype=img.dtype, memory_format=CL)
        rc = _lib.load().ideas_patch_resize(_lib.ptr(y), _lib.ptr(x), _boxes_arg(boxes), n, b, c, h, w, out_hw[0], out_hw[1], dt,
                                            _lib.stream_ptr())
        _lib.check(rc, "ideas_patch_resize")
        ctx.boxes, ctx.shape, ctx.out_hw, ctx.dtype = tuple(boxes), (b, c, h, w), tuple(out_hw), img.dtype
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        b, c, h, w = ctx.shape
        gy = gy.to(ctx.dtype)
        gy = gy if gy.is_contiguous(memory_format=CL) else gy.contiguous(memory_format=CL)
        gx = torch.empty((b, c, h, w), device=gy.device, dtype=torch.float32, memory_format=CL)
        rc = _lib.load().ideas_patch_resize_bwd(_lib.ptr(gx), _lib.ptr(gy), _boxes_arg(ctx.boxes), len(ctx.boxes), b, c, h, w,
                                                ctx.out_hw[0], ctx.out_hw[1], 1, _lib.act_dtype(gy), _lib.stream_ptr())
        _lib.check(rc, "ideas_patch_resize_bwd")
        return gx.to(ctx.dtype), None, None


def patch_resize(img: torch.Tensor, boxes: Sequence[Tuple[int, int, int, int]], out_hw: Tuple[int, int]) -> torch.Tensor:
    """Crops ``boxes`` ((y, x, h, w) each, shared by the batch) of ``img`` [B, C, H, W], each resized bilinearly
    (``align_corners=False``) to ``out_hw``; [B * len(boxes), C, oh, ow], image-major, channels_last."""
    return _PatchResize.apply(img, tuple(tuple(int(v) for v in bx) for bx in boxes), tuple(out_hw))
