"""An in-process stand-in for the part of the ``lmdb`` API that dataset.py:10-47 uses -- NOT an LMDB implementation.

The build image has no ``lmdb`` package (and no network to get one), so ``ideas_amd.data.LMDBDataset`` could never execute.  This module
gives it -- and the reference's own ``LMDBDataset``, imported from /root/reference in the build container -- the same object protocol
over a directory of ``<hex key>.bin`` files: ``open(path, max_readers=, readonly=, lock=, readahead=, meminit=)`` -> environment;
``env.begin(write=False)`` -> a context-managed transaction; ``txn.cursor()`` iterates ``(key, value)`` in ascending key order (LMDB's
order); ``txn.get(key)`` -> bytes or None.  ``write_store`` is the test-side writer.  What this exercises is the dataset code on either
side (key enumeration incl. the ``idx > max_num`` cut-off, byte -> PIL decode, resize); the LMDB file format itself is out of reach here,
which DESIGN.md states."""
import io
import os


class _Txn:
    def __init__(self, root):
        self.root = root

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def _keys(self):
        return sorted(bytes.fromhex(f[:-4]) for f in os.listdir(self.root) if f.endswith(".bin"))

    def cursor(self):
        for k in self._keys():
            yield k, self.get(k)

    def get(self, key):
        p = os.path.join(self.root, key.hex() + ".bin")
        if not os.path.exists(p):
            return None
        with io.open(p, "rb") as f:          # (this module's own `open` is the lmdb entry point)
            return f.read()


class Environment:
    def __init__(self, path, **kw):
        if not os.path.isdir(path):
            raise Error(f"{path}: No such file or directory")
        self.path, self.kw = path, kw

    def begin(self, write=False):
        assert not write
        return _Txn(self.path)

    def __bool__(self):
        return True


class Error(Exception):
    pass


def open(path, **kw):             # noqa: A001 - the name the real module exports
    return Environment(path, **kw)


def write_store(path, items):
    os.makedirs(path, exist_ok=True)
    for k, v in items:
        with io.open(os.path.join(path, k.hex() + ".bin"), "wb") as f:
            f.write(v)
