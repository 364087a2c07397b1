"""Input pipeline — SURVEY.md §8(f) row 3.  Mirrors ``dataset.py:10-85`` (LMDB / folder datasets, PIL resize) and the
transform + loader of ``train.py:443-464`` (RandomHorizontalFlip, ToTensor, Normalize(0.5, 0.5), RandomSampler).

What moves to the device: the datasets hand out DECODED, RESIZED uint8 HWC images (PIL does the decode and the resize
on the host workers, exactly as the reference); flip + ToTensor + Normalize run as one HIP kernel
(``ideas_image_u8_to_f32``, csrc/image_io.hip) that writes the f32 NHWC layout of the networks — 4x less PCIe traffic than
shipping normalised f32, and bit-identical values.  ``DeviceLoader`` overlaps the host->device copy of batch t+1 with the
training step on batch t (pinned buffers, a copy stream).  Multi-GPU: ``ShardSampler`` gives every rank a disjoint,
equally long slice of one seeded permutation per epoch (the role of DistributedSampler in stylegan2/train.py:33-41).
"""
from __future__ import annotations

import os
from io import BytesIO
from typing import Iterator, List, Optional

import numpy as np
import torch
from torch.utils import data

from . import _lib

IMG_EXTENSIONS = ['webp', '.png', '.jpg', '.jpeg', '.ppm', '.bmp', '.pgm', '.tif', '.tiff']   # dataset.py:50 (sic: 'webp')


def _decode(img, resolution: int) -> torch.Tensor:
    """PIL image -> uint8 [R, R, 3] (dataset.py:45,70: ``Image.open(..).resize((R, R))``, PIL's default filter)."""
    img = img.convert("RGB").resize((resolution, resolution))   # convert: a no-op for the RGB files the reference assumes
    return torch.from_numpy(np.asarray(img, dtype=np.uint8).copy())


def list_files(path: str) -> List[str]:
    """Every file below ``path``, sorted (imutils.paths.list_files + sorted, dataset.py:56)."""
    out = []
    for root, _, files in os.walk(path):
        out.extend(os.path.join(root, f) for f in files)
    return sorted(out)


class NormalDataset(data.Dataset):
    """Folder of image files (dataset.py:53-73)."""

    def __init__(self, path: str, resolution: int = 256, max_num: int = 70000):
        listed = list_files(path)
        self.files = [f for f in listed[:min(max_num, len(listed))] if any(f.lower().endswith(e) for e in IMG_EXTENSIONS)]
        self.resolution = resolution

    def __len__(self) -> int:
        return len(self.files)

    def __getitem__(self, index: int) -> torch.Tensor:
        from PIL import Image
        with Image.open(self.files[index]) as img:
            return _decode(img, self.resolution)


class LMDBDataset(data.Dataset):
    """LMDB of encoded images (dataset.py:10-47).  Needs the ``lmdb`` package (not in the build image: raises clearly).  Executed in
    the tests against an in-process stand-in of that API (tests/lmdb_standin.py) -- beside the reference's own class on the same
    store (tests/test_data.py); the LMDB file format itself is the package's business, not this class's."""

    def __init__(self, path: str, resolution: int = 256, max_num: int = 70000):
        try:
            import lmdb
        except ImportError as e:   # pragma: no cover - depends on the image
            raise ImportError("LMDBDataset needs the 'lmdb' package (dataset.py:3); use dataset_type='normal'") from e
        self.env = lmdb.open(path, max_readers=32, readonly=True, lock=False, readahead=False, meminit=False)
        if not self.env:
            raise IOError('Cannot open lmdb dataset', path)
        self.keys = []
        with self.env.begin(write=False) as txn:
            for idx, (key, _) in enumerate(txn.cursor()):
                self.keys.append(key)
                if idx > max_num:
                    break
        self.resolution = resolution

    def __len__(self) -> int:
        return len(self.keys)

    def __getitem__(self, index: int) -> torch.Tensor:
        from PIL import Image
        with self.env.begin(write=False) as txn:
            img_bytes = txn.get(self.keys[index])
        return _decode(Image.open(BytesIO(img_bytes)), self.resolution)


def set_dataset(type: str, path: str, resolution: int, max_num: int = 70000) -> data.Dataset:
    """dataset.py:76-85 (the transform argument is gone: it runs on the device)."""
    if type == 'lmdb':
        return LMDBDataset(path, resolution, max_num)
    if type == 'normal':
        return NormalDataset(path, resolution, max_num)
    raise NotImplementedError(type)


class ShardSampler(data.Sampler):
    """Rank ``rank`` of ``world``'s slice of one permutation per epoch (seeded identically on every rank), padded by
    wrap-around so that all ranks see the same number of samples; ``world == 1`` with shuffle is RandomSampler's role
    (utils.py:42-47)."""

    def __init__(self, length: int, shuffle: bool = True, rank: int = 0, world: int = 1, seed: int = 0):
        if not 0 <= rank < world:
            raise ValueError("rank out of range")
        self.length, self.shuffle, self.rank, self.world, self.seed = length, shuffle, rank, world, seed
        self.epoch = 0
        self.per_rank = -(-length // world)

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def __len__(self) -> int:
        return self.per_rank

    def __iter__(self) -> Iterator[int]:
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed * 1000003 + self.epoch)
            order = torch.randperm(self.length, generator=g).tolist()
        else:
            order = list(range(self.length))
        total = self.per_rank * self.world
        order = (order * (total // max(len(order), 1) + 1))[:total] if order else []
        return iter(order[self.rank:total:self.world])


def data_sampler(dataset, shuffle: bool, rank: int = 0, world: int = 1, seed: int = 0) -> ShardSampler:
    return ShardSampler(len(dataset), shuffle, rank, world, seed)


def u8_to_f32(x_u8: torch.Tensor, flip: Optional[torch.Tensor] = None, mean: float = 0.5, std: float = 0.5) -> torch.Tensor:
    """uint8 [B,H,W,C] on the device -> f32 [B,C,H,W] (channels_last memory) = Normalize(ToTensor(x)), samples with
    ``flip[b] != 0`` mirrored horizontally.  Bit-identical to the torchvision transforms of train.py:443-449."""
    _lib.require_cuda(x_u8, flip)
    if x_u8.dtype != torch.uint8 or x_u8.dim() != 4 or not x_u8.is_contiguous():
        raise RuntimeError("u8_to_f32 expects a contiguous uint8 [B,H,W,C] tensor")
    b, h, w, c = x_u8.shape
    if flip is not None:
        flip = flip.to(torch.uint8).contiguous()
        if flip.numel() != b:
            raise RuntimeError("flip must have one entry per sample")
    y = torch.empty((b, c, h, w), device=x_u8.device, dtype=torch.float32, memory_format=torch.channels_last)
    rc = _lib.load().ideas_image_u8_to_f32(_lib.ptr(y), _lib.ptr(x_u8), _lib.ptr(flip), b, h, w, c, float(mean), float(std),
                                           _lib.stream_ptr())
    _lib.check(rc, "ideas_image_u8_to_f32")
    return y


class DeviceLoader:
    """Batches of normalised, randomly flipped images on the device.  The host side is a ``DataLoader`` over a uint8
    dataset; batch t+1 is copied (pinned memory, copy stream) while the caller trains on batch t.  ``last_flips`` holds the
    flip draws of the batch just returned (tests, reproducibility)."""

    def __init__(self, dataset: data.Dataset, batch_size: int, sampler: Optional[data.Sampler] = None, device="cuda",
                 num_workers: int = 0, flip: bool = True, seed: int = 0, drop_last: bool = False):
        self.device = torch.device(device)
        self.loader = data.DataLoader(dataset, batch_size=batch_size, sampler=sampler, num_workers=num_workers,
                                      pin_memory=True, drop_last=drop_last)
        self.flip = flip
        self.gen = torch.Generator().manual_seed(seed)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.last_flips: Optional[torch.Tensor] = None

    def __len__(self) -> int:
        return len(self.loader)

    def _stage(self, it):
        try:
            host = next(it)
        except StopIteration:
            return None
        flips = (torch.rand(host.shape[0], generator=self.gen) < 0.5).to(torch.uint8) if self.flip else None
        with torch.cuda.stream(self.copy_stream):
            dev = host.to(self.device, non_blocking=True)
            fdev = flips.to(self.device, non_blocking=True) if flips is not None else None
        return dev, fdev, flips

    def __iter__(self):
        it = iter(self.loader)
        nxt = self._stage(it)
        while nxt is not None:
            dev, fdev, flips = nxt
            cur = torch.cuda.current_stream(self.device)
            cur.wait_stream(self.copy_stream)
            dev.record_stream(cur)
            if fdev is not None:
                fdev.record_stream(cur)
            nxt = self._stage(it)                      # batch t+1 starts its copy before batch t is consumed
            self.last_flips = flips
            yield u8_to_f32(dev, fdev)


def sample_data(loader):
    """utils.py:63-66."""
    while True:
        for batch in loader:
            yield batch
