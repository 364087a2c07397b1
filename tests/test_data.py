"""Input pipeline (ideas_amd/data.py): datasets + sampler on CPU, the conversion kernel and the device
loader on the GPU.  The reference's dataset.py / torchvision transforms cannot be imported here (lmdb, imutils,
torchvision are absent), so parity is against their published semantics restated inline: PIL ``resize((R, R))``,
``ToTensor`` = uint8/255 in CHW, ``Normalize(0.5, 0.5)`` = (t - 0.5) / 0.5, ``RandomHorizontalFlip`` = mirror along W."""
import os

import numpy as np
import pytest
import torch

from ideas_amd import data as D


def _make_images(root, n=7):
    from PIL import Image
    rng = np.random.RandomState(0)
    os.makedirs(os.path.join(root, "sub"), exist_ok=True)
    names = []
    for i in range(n):
        h, w = rng.randint(20, 70), rng.randint(20, 70)
        arr = rng.randint(0, 256, size=(h, w, 3), dtype=np.uint8)
        name = os.path.join(root, "sub" if i % 3 == 0 else "", f"img_{i:02d}.png")
        Image.fromarray(arr).save(name)
        names.append(name)
    open(os.path.join(root, "notes.txt"), "w").write("not an image")
    return sorted(names)


def test_normal_dataset_lists_decodes_and_resizes(tmp_path):
    from PIL import Image
    names = _make_images(str(tmp_path))
    ds = D.NormalDataset(str(tmp_path), resolution=32)
    assert ds.files == names                                   # sorted, recursive, extension-filtered (dataset.py:53-62)
    for i in (0, 3, len(ds) - 1):
        want = np.asarray(Image.open(names[i]).resize((32, 32)))
        got = ds[i]
        assert got.dtype == torch.uint8 and tuple(got.shape) == (32, 32, 3)
        assert np.array_equal(got.numpy(), want)
    assert len(D.NormalDataset(str(tmp_path), resolution=32, max_num=3)) <= 3
    with pytest.raises(NotImplementedError):
        D.set_dataset("zip", str(tmp_path), 32)


def test_shard_sampler_partitions_one_permutation():
    n, world = 103, 4
    shards = [list(D.ShardSampler(n, True, r, world, seed=5)) for r in range(world)]
    assert all(len(s) == 26 for s in shards)                   # ceil(103 / 4), equal on every rank
    flat = [i for s in shards for i in s]
    assert set(flat) == set(range(n)) and len(flat) == 104     # everything once, one wrap-around pad
    again = [list(D.ShardSampler(n, True, r, world, seed=5)) for r in range(world)]
    assert again == shards                                     # seeded: identical on every process
    s = D.ShardSampler(n, True, 0, world, seed=5)
    s.set_epoch(1)
    assert list(s) != shards[0]                                # reshuffled per epoch
    assert list(D.ShardSampler(10, False)) == list(range(10))  # sequential, single rank (utils.py:46-47)
    assert sorted(D.data_sampler(list(range(9)), True)) == list(range(9))


def _reference_transform(u8_hwc: torch.Tensor, flip: bool) -> torch.Tensor:
    t = u8_hwc.permute(2, 0, 1).float().div(255)               # ToTensor
    if flip:
        t = t.flip(-1)                                         # RandomHorizontalFlip (applied to the PIL image: same pixels)
    return t.sub(0.5).div(0.5)                                 # Normalize((0.5,)*3, (0.5,)*3)


@pytest.mark.gpu
def test_u8_to_f32_is_bitwise_totensor_normalize_flip():
    torch.manual_seed(0)
    x = torch.randint(0, 256, (5, 17, 23, 3), dtype=torch.uint8)
    x[0, 0, 0] = torch.tensor([0, 255, 128], dtype=torch.uint8)
    flips = torch.tensor([0, 1, 1, 0, 1], dtype=torch.uint8)
    y = D.u8_to_f32(x.cuda(), flips.cuda())
    assert y.shape == (5, 3, 17, 23) and y.is_contiguous(memory_format=torch.channels_last)
    want = torch.stack([_reference_transform(x[b], bool(flips[b])) for b in range(5)])
    assert torch.equal(y.cpu(), want)                          # same two roundings as the reference's transforms
    assert torch.equal(D.u8_to_f32(x.cuda()).cpu(), torch.stack([_reference_transform(x[b], False) for b in range(5)]))
    with pytest.raises(RuntimeError):
        D.u8_to_f32(x)                                         # no CPU fallback


@pytest.mark.gpu
def test_device_loader_end_to_end(tmp_path):
    _make_images(str(tmp_path), n=9)
    ds = D.set_dataset("normal", str(tmp_path), 16)
    loader = D.DeviceLoader(ds, batch_size=4, sampler=D.data_sampler(ds, shuffle=False), seed=3)
    seen = 0
    for batch in loader:
        b = batch.shape[0]
        assert batch.is_cuda and batch.shape[1:] == (3, 16, 16)
        want = torch.stack([_reference_transform(ds[seen + i], bool(loader.last_flips[i])) for i in range(b)])
        assert torch.equal(batch.cpu(), want)
        seen += b
    assert seen == 9
    it = D.sample_data(loader)                                 # endless iterator of train.py:56 / utils.py:63-66
    assert next(it).shape[0] == 4
