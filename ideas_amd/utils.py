"""Training helpers of the IDEAS step — host-side mirror of the reference's utils.py.

``requires_grad`` (:50-52), ``accumulate`` (:55-60), ``message_to_tensor`` (:74-83), ``tensor_to_message``
(:86-97), ``d_logistic_loss`` (:105-109), ``d_r1_loss`` (:112-118), ``g_nonsaturating_loss`` (:121-124),
``patchify_image`` (:127-149).  Same names, arguments and results; differences are mechanical:
the message codec works on whatever device its input lives on (the reference allocates on CPU), the EMA is
two multi-tensor launches instead of two per parameter, and ``patchify_image`` can take the crop boxes
explicitly so a step can be replayed without emulating three interleaved RNG streams.
"""
from __future__ import annotations

import os
import random
from typing import List, Optional, Sequence, Tuple

import torch
from torch import autograd
from torch.nn import functional as F

Box = Tuple[int, int, int, int]  # (c_y, c_x, c_h, c_w)


def requires_grad(model, flag: bool = True) -> None:
    for p in model.parameters():
        p.requires_grad_(flag)


@torch.no_grad()
def accumulate(model1, model2, decay: float = 0.999) -> None:
    """EMA of the *parameters* (buffers untouched): p1 = decay * p1 + (1 - decay) * p2."""
    par2 = dict(model2.named_parameters())
    dst, src = [], []
    for k, p in model1.named_parameters():
        dst.append(p.data)
        src.append(par2[k].data)
    if not dst:
        return
    torch._foreach_mul_(dst, decay)
    torch._foreach_add_(dst, src, alpha=1 - decay)


def message_to_tensor(message: torch.Tensor, sigma: int, delta: float,
                      jitter: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Bits [B, L*sigma] -> floats [B, L] in [-1, 1]: sigma bits select one of 2^sigma bins, plus a uniform
    jitter of +-delta*step inside the bin.  ``jitter`` (U[0,1), shape [B, L]) makes the draw explicit."""
    step = 2 / 2 ** sigma
    nums = torch.zeros(message.shape[0], message.shape[1] // sigma, device=message.device, dtype=message.dtype)
    for i in range(sigma):
        nums += message[:, i::sigma] * 2 ** (sigma - i - 1)
    secret = step * (nums + 0.5) - 1
    if jitter is None:
        jitter = torch.rand_like(secret)
    r = step * delta
    return secret + (jitter * r * 2 - r)


def tensor_to_message(secret_tensor: torch.Tensor, sigma: int) -> torch.Tensor:
    """Inverse of ``message_to_tensor``: clamp, rescale, peel off sigma bits MSB first (an integer decision)."""
    msg = torch.zeros(secret_tensor.shape[0], secret_tensor.shape[1] * sigma, device=secret_tensor.device,
                      dtype=secret_tensor.dtype)
    step = 2 / 2 ** sigma
    nums = (torch.clamp(secret_tensor, min=-1, max=1) + 1) / step
    for i in range(sigma):
        bit = (nums >= 2 ** (sigma - i - 1)).to(nums.dtype)
        msg[:, i::sigma] = bit
        nums = nums - bit * 2 ** (sigma - i - 1)
    return msg


def image_grid(images: torch.Tensor, nrow: int, value_range=(-1.0, 1.0), padding: int = 2) -> torch.Tensor:
    """uint8 [H', W', 3] grid of a [B, C, H, W] batch: what ``torchvision.utils.save_image(sample, path, nrow=..., normalize=True,
    range=(-1, 1))`` writes for the sample sheet of train.py:295-303, restated (torchvision is not a dependency of this build):
    clamp to the range, map it to [0, 1], lay the images out ``nrow`` per row with ``padding`` black pixels around each, then
    ``* 255 + 0.5``, clamp and truncate to bytes."""
    x = images.detach().float().cpu()
    lo, hi = float(value_range[0]), float(value_range[1])
    x = (x.clamp(min=lo, max=hi) - lo) / max(hi - lo, 1e-5)
    b, c, h, w = x.shape
    if c == 1:
        x = x.expand(b, 3, h, w)
    xmaps = min(int(nrow), b)
    ymaps = -(-b // xmaps)
    grid = torch.zeros(3, (h + padding) * ymaps + padding, (w + padding) * xmaps + padding)
    for k in range(b):
        y0, x0 = (k // xmaps) * (h + padding) + padding, (k % xmaps) * (w + padding) + padding
        grid[:, y0:y0 + h, x0:x0 + w] = x[k]
    return grid.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8)


def save_image_grid(images: torch.Tensor, path: str, nrow: int, value_range=(-1.0, 1.0)) -> None:
    """Write the sample sheet (train.py:295-301): PNG by PIL, as torchvision's save_image does."""
    from PIL import Image
    Image.fromarray(image_grid(images, nrow, value_range).numpy()).save(path)


def d_logistic_loss(real_pred, fake_pred):
    return F.softplus(-real_pred).mean() + F.softplus(fake_pred).mean()


def d_r1_loss(real_pred, real_img):
    (grad_real,) = autograd.grad(outputs=real_pred.sum(), inputs=real_img, create_graph=True)
    return grad_real.pow(2).reshape(grad_real.shape[0], -1).sum(1).mean()


def g_nonsaturating_loss(fake_pred):
    return F.softplus(-fake_pred).mean()




def draw_boxes(height: int, width: int, n_crop: int, min_size: float = 1 / 8, max_size: float = 1 / 4) -> List[Box]:
    """The random half of ``patchify_image``: sizes from the torch CPU generator, offsets from ``random``."""
    size = torch.rand(n_crop) * (max_size - min_size) + min_size
    hs = (size * height).type(torch.int64).tolist()
    ws = (size * width).type(torch.int64).tolist()
    return [(random.randrange(0, height - ch), random.randrange(0, width - cw), ch, cw) for ch, cw in zip(hs, ws)]


def patchify_image(img: torch.Tensor, n_crop: int, min_size: float = 1 / 8, max_size: float = 1 / 4,
                   boxes: Optional[Sequence[Box]] = None) -> torch.Tensor:
    """``n_crop`` random crops (one box per crop for the whole batch), bilinearly resized to H/4 x W/4 and stacked
    image-major: [B*n_crop, C, H/4, W/4].  Differentiable w.r.t. ``img``."""
    b, c, h, w = img.shape
    if boxes is None:
        boxes = draw_boxes(h, w, n_crop, min_size, max_size)
    th, tw = int(h * max_size), int(w * max_size)
    if img.is_cuda:
        if c not in (1, 3) or len(boxes) > 64:                # (train.py never asks for it: RGB images, 8 or 32 boxes)
            raise RuntimeError(f"patchify_image on a device tensor: csrc/patchify.hip covers 1 or 3 channels and <= 64 boxes, got {c} / "
                               f"{len(boxes)} (no eager fallback on the device)")
        from .op.patchify import patch_resize                 # one launch for all boxes
        return patch_resize(img, boxes, (th, tw))
    # host tensors only: the restatement of utils.py:141-147 that the CPU host-logic tests drive the step with
    patches = [F.interpolate(img[:, :, y:y + ch, x:x + cw], size=(th, tw), mode="bilinear", align_corners=False)
               for (y, x, ch, cw) in boxes]
    return torch.stack(patches, 1).reshape(-1, c, th, tw)
