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


def _png_bytes(arr):
    from io import BytesIO
    from PIL import Image
    b = BytesIO()
    Image.fromarray(arr).save(b, format="PNG")
    return b.getvalue()


def _lmdb_store(root, n=9):
    """n encoded images under the stand-in's layout; keys as the StyleGAN2 tooling writes them (f'{resolution}-{index:05d}')."""
    import lmdb_standin
    rng = np.random.RandomState(3)
    arrs = [rng.randint(0, 256, size=(rng.randint(20, 70), rng.randint(20, 70), 3), dtype=np.uint8) for _ in range(n)]
    lmdb_standin.write_store(root, [(f"256-{i:05d}".encode(), _png_bytes(a)) for i, a in enumerate(arrs)])
    return arrs


def test_lmdb_dataset_executes_against_the_api_standin(tmp_path, monkeypatch):
    """``LMDBDataset`` (dataset.py:10-47) had never run -- the image has no ``lmdb``.  tests/lmdb_standin.py provides
    the object protocol the class uses (open / begin / cursor / get) over a directory of files; with it in ``sys.modules`` the class
    enumerates the keys in LMDB's (sorted) order, applies the reference's ``idx > max_num`` cut-off (which keeps max_num + 2 keys),
    decodes and resizes each value exactly as the folder dataset does."""
    import sys
    from PIL import Image
    import lmdb_standin
    monkeypatch.setitem(sys.modules, "lmdb", lmdb_standin)
    arrs = _lmdb_store(str(tmp_path / "db"))
    ds = D.set_dataset("lmdb", str(tmp_path / "db"), 32)
    assert isinstance(ds, D.LMDBDataset) and len(ds) == 9
    assert ds.keys == sorted(ds.keys) and ds.keys[0] == b"256-00000"
    assert ds.env.kw == dict(max_readers=32, readonly=True, lock=False, readahead=False, meminit=False)      # dataset.py:12-19
    for i in (0, 4, 8):
        want = np.asarray(Image.fromarray(arrs[i]).resize((32, 32)))
        got = ds[i]
        assert got.dtype == torch.uint8 and tuple(got.shape) == (32, 32, 3) and np.array_equal(got.numpy(), want)
    assert len(D.LMDBDataset(str(tmp_path / "db"), 32, max_num=3)) == 5          # idx 0..4: the break comes after the append (dataset.py:27-30)
    with pytest.raises(Exception):
        D.LMDBDataset(str(tmp_path / "missing"), 32)
    # through the loader's collate path on the host (uint8 batches; the device half is test_device_loader_end_to_end)
    batch = torch.stack([ds[i] for i in range(4)])
    assert tuple(batch.shape) == (4, 32, 32, 3)


@pytest.mark.skipif(not os.path.isdir(os.environ.get("IDEAS_REFERENCE", "/root/reference")), reason="needs the reference checkout (build container only)")
def test_lmdb_and_folder_datasets_against_the_reference_classes(tmp_path, monkeypatch):
    """The reference's OWN dataset.py (LMDBDataset and NormalDataset, dataset.py:10-73) imported in the build container with the lmdb
    stand-in and a two-line imutils stub, given a transform that returns the PIL image's pixels: same length, same order, same pixels as
    ideas_amd.data's classes on the same store / folder."""
    import sys
    import types
    import lmdb_standin
    ref = os.environ.get("IDEAS_REFERENCE", "/root/reference")
    paths = types.ModuleType("imutils.paths")
    paths.list_files = lambda p: (os.path.join(r, f) for r, _, fs in os.walk(p) for f in fs)
    imutils = types.ModuleType("imutils")
    imutils.paths = paths
    for name, mod in (("lmdb", lmdb_standin), ("imutils", imutils), ("imutils.paths", paths)):
        monkeypatch.setitem(sys.modules, name, mod)
    monkeypatch.syspath_prepend(ref)
    monkeypatch.delitem(sys.modules, "dataset", raising=False)
    import dataset as RD
    to_u8 = lambda img: torch.from_numpy(np.asarray(img.convert("RGB"), dtype=np.uint8).copy())
    _lmdb_store(str(tmp_path / "db"), n=7)
    for max_num in (70000, 2):
        a, b = RD.LMDBDataset(str(tmp_path / "db"), to_u8, 48, max_num), D.LMDBDataset(str(tmp_path / "db"), 48, max_num)
        assert len(a) == len(b) and a.keys == b.keys
        assert all(torch.equal(a[i], b[i]) for i in range(len(a)))
    _make_images(str(tmp_path / "folder"))
    a, b = RD.NormalDataset(str(tmp_path / "folder"), to_u8, 48), D.NormalDataset(str(tmp_path / "folder"), 48)
    assert len(a) == len(b) == 7 and [os.path.basename(f) for f in a.files] == [os.path.basename(f) for f in b.files]
    assert all(torch.equal(a[i], b[i]) for i in range(len(a)))
    monkeypatch.delitem(sys.modules, "dataset", raising=False)


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
