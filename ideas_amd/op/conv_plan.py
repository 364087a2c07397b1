"""Geometry of the convolution family: turns conv / transposed-conv / their gradients into launches of the
one parameterised kernel family (``ideas_conv_params``, include/ideas_hip.h).  Pure Python integers and
tensor *views* only — no device work — so the derivations are unit-tested on CPU against F.conv2d.

Conventions.  A "conv" geometry is (KH, KW, stride s, zero/reflect padding p) with weight w[O, I, KH, KW]:

    forward   y[oy]  = sum_ky w[ky] * x[oy*s + ky - p]
    dgrad     gx[iy] = sum_{ky : (iy + p - ky) % s == 0} w[ky] * gy[(iy + p - ky) / s]
    wgrad     gw[ky] = sum_oy gy[oy] * x[oy*s + ky - p]

A transposed conv (weight [I, O, KH, KW], stride s, padding 0) is the dgrad of the conv whose weight is the
same tensor read as [O' = I, I' = O, KH, KW]; its own input gradient is that conv's forward and its weight
gradient is that conv's wgrad with the roles of x and gy swapped.

dgrad is decomposed by output parity r = iy mod s (one launch per (ry, rx)): with c = (r + p) mod s the
contributing taps are ky = c + s*j, j = 0..J-1, J = ceil((KH - c) / s), and gy is read at q + e - j where
iy = s*q + r and e = (r + p - c) / s — i.e. a stride-1 gather with tap step -1.  No multiply-by-zero work is
issued for the stuffed zeros of a strided transposed conv (4 / 2 / 2 / 1 taps instead of 9 for 3x3, s = 2).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import torch


@dataclass(frozen=True)
class ConvGeom:
    kh: int
    kw: int
    stride: int = 1
    pad: int = 0
    reflect: bool = False

    def out_size(self, ih: int, iw: int) -> Tuple[int, int]:
        return ((ih + 2 * self.pad - self.kh) // self.stride + 1, (iw + 2 * self.pad - self.kw) // self.stride + 1)


@dataclass
class Launch:
    """One kernel launch: integer fields of ideas_conv_params + the [Cout, TY, TX, Cin] weight matrix of the launch as a strided
    VIEW of the parameter (``wview``: no device work).  The b3 / bf16 operand preparation reads it through its strides
    (ideas_*_strided); ``wmat`` materialises the dense f32 copy only for the kernels that take one."""
    B: int
    IH: int
    IW: int
    Cin: int
    YH: int
    YW: int
    Cout: int
    OH: int
    OW: int
    TY: int
    TX: int
    sy: int
    sx: int
    dy: int
    dx: int
    offy: int
    offx: int
    osy: int = 1
    osx: int = 1
    ooy: int = 0
    oox: int = 0
    reflect: int = 0
    wview: Optional[torch.Tensor] = None     # forward family: [Cout, TY, TX, Cin] view of the parameter
    wsrc: Optional[torch.Tensor] = None      # the parameter (view) it was derived from, and how: cache key for
    wkey: Optional[tuple] = None             # the derived forms (dense f32 copy, split-bf16 planes, bf16 pack)
    w_slice: Optional[Tuple[int, int]] = None  # wgrad of a phase: (cy, cx) tap offsets inside the full kernel

    @property
    def wmat(self) -> Optional[torch.Tensor]:
        """Dense f32 [Cout, TY, TX, Cin] (free when the view is already contiguous, e.g. a channels_last parameter)."""
        v = self.wview
        if v is None or v.is_contiguous():
            return v
        return v.contiguous() if self.wsrc is None else cached(self.wsrc, ("dense",) + self.wkey, v.contiguous)


# ---------------------------------------------------------------------------------------------------------------
# Derived-weight cache.  The networks are applied 2-3 times per iteration with unchanged weights (E on X and on the
# generated images, G on two code pairs, the discriminators in the D and the G phase), and every application re-derives
# the same matrices: the phase-sliced / transposed copies for input gradients, the split-bf16 planes.  train_step
# switches the cache on for the duration of an iteration and clears it after every optimiser step; outside of that
# (plain op calls, tests) nothing is cached, so in-place weight updates by anyone can never be missed.
# ---------------------------------------------------------------------------------------------------------------
_CACHE = None

# Batched refill.  Most derived forms are "parameter -> bf16 planes / pack" by one of three kernels; remade one by one they are ~390
# launches of 5-25 us per iteration (f32; ~290 in bf16), most too small to fill the chip.  A miss that comes with a ``prep``
# description (op, destination elements, dims, strides, source address) is remembered across iterations; from then on
# ``cache_begin`` (all of them) and ``cache_clear(params)`` (those derived from the stepped parameters) remake the remembered forms
# with ONE launch per op (ideas_weight_prep_batched) into persistent buffers, and the first use finds them in the cache.
# Entries whose parameter died or moved are dropped.  IDEAS_PREFILL=0 keeps the one-by-one path.
import os as _os
import weakref as _weakref

PREFILL = _os.environ.get("IDEAS_PREFILL", "1") != "0"
STYLE_CACHE = _os.environ.get("IDEAS_STYLE_CACHE", "1") != "0"
_RECORDED = {}      # cache key -> (prep, weakref of the base parameter, its data_ptr when recorded)
_PREP_STATE = {}    # tuple of cache keys -> per-op launch state (device table, blocks, persistent outputs)
_KEYS_OF = {}       # (span set (None = everything), precision) -> the sorted tuple of recorded keys inside it
_OUT_BUF = {}       # cache key -> its persistent destination buffer, shared by every span selection that contains the key


def _prep_state(keys):
    import ctypes as C
    from .. import _lib
    st = _PREP_STATE.get(keys)
    if st is not None:
        return st
    by_op = {}
    for k in keys:
        by_op.setdefault(_RECORDED[k][0][0], []).append(k)
    st = []
    for op, ks in sorted(by_op.items()):
        descs = (_lib.PrepDesc * len(ks))()
        outs, block0 = [], 0
        dev = None
        for i, k in enumerate(ks):
            (_, numel, a, sv, unit, src, work), ref, _p0 = _RECORDED[k]
            base = ref()
            dev = base.device
            out = _OUT_BUF.get(k)
            if out is None or out.numel() != numel or out.device != dev:
                out = _OUT_BUF[k] = torch.empty(numel, device=dev, dtype=torch.bfloat16)
            nb = int(min(work // 256 + 1, 2048))
            d = descs[i]
            d.dst, d.w = out.data_ptr(), src
            for j in range(5):
                d.s[j] = sv[j] if j < len(sv) else 0
            for j in range(4):
                d.a[j] = a[j] if j < len(a) else 0
            d.unit, d.block0, d.nblocks = int(unit), block0, nb
            block0 += nb
            outs.append((k, out))
        table = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(dev)
        st.append((op, table, len(ks), block0, outs))
    _PREP_STATE[keys] = st
    return st


def _prefill(spans=None) -> None:
    """Remake the remembered derived forms (of the parameters inside ``spans``, or all) into the cache, one launch per op."""
    if not PREFILL or _CACHE is None or not _RECORDED:
        return
    import bisect
    from .. import _lib
    dead = [k for k, (_, ref, p0) in _RECORDED.items() if ref() is None or ref().data_ptr() != p0]
    if dead:
        for k in dead:
            del _RECORDED[k]
            _OUT_BUF.pop(k, None)
        _PREP_STATE.clear()
        _KEYS_OF.clear()
    # Forms of the other arithmetic mode are not remade: a process that ran f32 iterations and then switches to bf16 activations
    # (bench.py --also-bf16) would otherwise split every weight into b3 planes after each optimiser step for nothing (the few
    # tiny layers a mode still computes with the other mode's kernels miss the cache and are remade one by one, as before).
    from ..precision import activation_dtype
    bf = activation_dtype() == torch.bfloat16
    live = lambda k: str(k[3][0]).startswith("bf16") == bf
    # (the selection + sort is ~1.5 ms of host time for ~400 entries, three times per iteration, at points where the GPU queue is
    #  empty -- right behind an optimiser step; memoised per span set until _RECORDED changes)
    sel = (None if spans is None else tuple((a, b) for a, b in spans), bf)
    keys = _KEYS_OF.get(sel)
    if keys is None:
        if spans is None:
            keys = tuple(sorted((k for k in _RECORDED if live(k)), key=lambda k: (k[0], repr(k[1:]))))
        else:
            starts = [m[0] for m in spans]
            keys = []
            for k in _RECORDED:
                i = bisect.bisect_right(starts, k[0]) - 1
                if i >= 0 and k[0] < spans[i][1] and live(k):
                    keys.append(k)
            keys = tuple(sorted(keys, key=lambda k: (k[0], repr(k[1:]))))
        _KEYS_OF[sel] = keys
    if not keys:
        return
    lib = _lib.load()
    for op, table, n, blocks, outs in _prep_state(keys):
        _lib.check(lib.ideas_weight_prep_batched(table.data_ptr(), n, op, blocks, _lib.stream_ptr()), "ideas_weight_prep_batched")
        for k, out in outs:
            _CACHE[k] = out


def cache_begin() -> None:
    global _CACHE
    _CACHE = {}
    _prefill(None)


def cache_clear(params=None, refill: bool = True) -> None:
    """Drop the derived weights -- all of them, or (``params``: the tensors an optimiser just stepped) only those derived from these
    parameters: the D step leaves E's / G's packs alone, which the G phase of the same iteration then reuses.  ``refill``: remake
    the remembered forms of those parameters right away in one batched launch per op (pointless after the last step of an
    iteration, whose cache ends anyway)."""
    if _CACHE is None:
        return
    if params is None:
        _CACHE.clear()
        _DEPS.clear()
        if refill:
            _prefill(None)
        return
    spans = sorted((p.data_ptr(), p.data_ptr() + p.numel() * p.element_size()) for p in params)
    merged = []
    for a, b in spans:                       # fused optimisers: the parameters are views of one flat buffer -> a few long spans
        if merged and a <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], b)
        else:
            merged.append([a, b])
    import bisect
    starts = [m[0] for m in merged]
    def stepped(addr):
        i = bisect.bisect_right(starts, addr) - 1
        return i >= 0 and addr < merged[i][1]
    for k in [k for k in _CACHE]:
        # an entry derived from SEVERAL parameters (model.styles_for: all sixteen modulation layers under the first one's key) lists
        # the others in _DEPS: stepping any of them drops it
        if stepped(k[0]) or any(stepped(a) for a in _DEPS.get(k, ())):
            del _CACHE[k]
            _DEPS.pop(k, None)
    if refill:
        _prefill(merged)


def cache_end() -> None:
    global _CACHE
    _CACHE = None


def cached(w: torch.Tensor, key, make, prep=None):
    """`make()` memoised on (storage address, shape, key) while the cache is on and `w` is (a view of) a Parameter.
    ``prep`` = (op, destination bf16 elements, dims a[<=4], strides s[<=5], unit, source address, work items): the same result as
    one entry of a batched launch (see _prefill); remembered on a miss."""
    if _CACHE is None:
        return make()
    base = w._base if w._base is not None else w
    if not isinstance(base, torch.nn.Parameter):
        return make()
    k = (w.data_ptr(), tuple(w.shape), tuple(w.stride()), key)
    v = _CACHE.get(k)
    if v is None:
        v = make()
        _CACHE[k] = v
        if prep is not None and PREFILL and k not in _RECORDED:
            _RECORDED[k] = (prep, _weakref.ref(base), base.data_ptr())
            _PREP_STATE.clear()
            _KEYS_OF.clear()
    return v


# Byte budget of the per-sample bf16 weight packs memoised by cached_on(..., budget=True): tens to hundreds of MB each at B = 32.
# G's layers are visited cyclically (forward packs of layers 1..16, again for the next pass over the same styles, then the input
# gradients' packs), so an evicting policy -- FIFO or LRU alike -- drops exactly the entries the next pass asks for once the working
# set exceeds the budget.  Admission control instead: entries are admitted until the budget is full and then stay for
# the iteration (hit rate = budget / working set, never zero); a miss beyond the budget is made, used and dropped by its caller.
# B = 32 at 256x256 the packs of one iteration peak at 5.8 GB (of 288 GB): the default of 8 GB admits all of them -- measured per
# iteration (tools/probes/pack_cache_stats.py): 70 hits / 113 admitted / 0 rejected, against 54 / 91 / 38 at 4 GB; bf16 step 153.4 ->
# 152.0 ms, same box.  BUDGET_STATS counts what happened.
STYLE_BUDGET_MB = int(_os.environ.get("IDEAS_STYLE_BUDGET_MB", "8192"))
_BUDGETED = {}      # cache key -> bytes of the admitted entries still in _CACHE
_BUDGET_TOTAL = [0]
BUDGET_STATS = {"hit": 0, "admitted": 0, "rejected": 0, "peak_bytes": 0}


def _budget_admit(k, nbytes: int) -> bool:
    def sweep():                             # cache_clear / a new iteration drop entries behind our back
        for dead in [q for q in _BUDGETED if q not in _CACHE]:
            _BUDGET_TOTAL[0] -= _BUDGETED.pop(dead)
    if len(_BUDGETED) and next(iter(_BUDGETED)) not in _CACHE:
        sweep()
    if _BUDGET_TOTAL[0] + nbytes > STYLE_BUDGET_MB << 20:
        sweep()
    if _BUDGET_TOTAL[0] + nbytes > STYLE_BUDGET_MB << 20:
        BUDGET_STATS["rejected"] += 1
        return False
    _BUDGETED[k] = nbytes
    _BUDGET_TOTAL[0] += nbytes
    BUDGET_STATS["admitted"] += 1
    BUDGET_STATS["peak_bytes"] = max(BUDGET_STATS["peak_bytes"], _BUDGET_TOTAL[0])
    return True


_DEPS = {}          # cache key -> addresses of further parameters the entry was derived from (cached_on(..., deps=...))


def cached_on(w: torch.Tensor, key, t: torch.Tensor, make, budget: bool = False, deps=()):
    """`make()` memoised on a parameter (as ``cached``) AND on the identity of a second tensor `t` (address + version counter) --
    the per-sample styles of a modulated conv: G is applied to the same texture code two or three times per iteration (train.py:58,
    :68, :145-160), and everything derived from (weights, styles) -- the styles themselves, the demodulation factors, the bf16
    per-sample weight packs -- is then the same.  The entry keeps `t` alive, so its address cannot be reused by another tensor
    while the entry exists.  ``deps``: further parameters the value depends on (cache_clear drops the entry when any of them is
    stepped, not only `w`).  IDEAS_STYLE_CACHE=0 switches it off (A/B)."""
    if _CACHE is None or not STYLE_CACHE:
        return make()
    base = w._base if w._base is not None else w
    if not isinstance(base, torch.nn.Parameter):
        return make()
    k = (w.data_ptr(), tuple(w.shape), tuple(w.stride()), key, t.data_ptr(), t._version, tuple(t.shape))
    v = _CACHE.get(k)
    if v is None:
        v = (t, make())
        if budget and torch.is_tensor(v[1]) and not _budget_admit(k, v[1].numel() * v[1].element_size()):
            return v[1]                      # over budget: used once by the caller, not kept
        _CACHE[k] = v
        if deps:
            _DEPS[k] = tuple(d.data_ptr() for d in deps)
    elif budget:
        BUDGET_STATS["hit"] += 1
    return v[1]


def peek_on(w: torch.Tensor, key, t: torch.Tensor):
    """The memoised value of ``cached_on(w, key, t, ...)`` if there is one, else None."""
    if _CACHE is None or not STYLE_CACHE:
        return None
    v = _CACHE.get((w.data_ptr(), tuple(w.shape), tuple(w.stride()), key, t.data_ptr(), t._version, tuple(t.shape)))
    return None if v is None else v[1]


def plan_fwd(x_shape, w: torch.Tensor, g: ConvGeom) -> Launch:
    b, ci, ih, iw = x_shape
    co = w.shape[0]
    oh, ow = g.out_size(ih, iw)
    return Launch(B=b, IH=ih, IW=iw, Cin=ci, YH=oh, YW=ow, Cout=co, OH=oh, OW=ow, TY=g.kh, TX=g.kw,
                  sy=g.stride, sx=g.stride, dy=1, dx=1, offy=-g.pad, offx=-g.pad, reflect=int(g.reflect),
                  wview=w.permute(0, 2, 3, 1), wsrc=w, wkey=("fwd",))


def _phase(r: int, p: int, s: int, k: int):
    c = (r + p) % s
    j = 0 if c >= k else -(-(k - c) // s)
    e = (r + p - c) // s
    return c, j, e


def plan_dgrad(gy_shape, w: torch.Tensor, g: ConvGeom, in_hw: Tuple[int, int]) -> Tuple[List[Launch], bool]:
    """Launches computing gx[B, I, IH, IW] from gy[B, O, OH, OW]; second value: must gx be pre-zeroed."""
    if g.reflect:
        raise ValueError("dgrad of a reflect-padded conv: compute the padded gradient (pad=0 geometry on the padded "
                         "size) and fold it with reflection_pad2d_backward")
    b, co, oh, ow = gy_shape
    ci = w.shape[1]
    ih, iw = in_hw
    s, p = g.stride, g.pad
    launches: List[Launch] = []
    need_zero = False
    for ry in range(s):
        cy, jy, ey = _phase(ry, p, s, g.kh)
        nqy = -(-(ih - ry) // s) if ih > ry else 0
        for rx in range(s):
            cx, jx, ex = _phase(rx, p, s, g.kw)
            nqx = -(-(iw - rx) // s) if iw > rx else 0
            if nqy <= 0 or nqx <= 0:
                continue
            if jy == 0 or jx == 0:
                need_zero = True
                continue
            # wmat[i][(jy, jx, o)] = w[o][i][cy + s*jy][cx + s*jx]
            launches.append(Launch(B=b, IH=oh, IW=ow, Cin=co, YH=ih, YW=iw, Cout=ci, OH=nqy, OW=nqx, TY=jy, TX=jx,
                                   sy=1, sx=1, dy=-1, dx=-1, offy=ey, offx=ex, osy=s, osx=s, ooy=ry, oox=rx,
                                   wview=w[:, :, cy::s, cx::s].permute(1, 2, 3, 0), wsrc=w, wkey=("dgrad", s, cy, cx)))
    return launches, need_zero


def plan_wgrad(x_shape, gy_shape, g: ConvGeom) -> Launch:
    """gw[O, KH, KW, I] (OHWI) from x[B, I, IH, IW] and gy[B, O, OH, OW]."""
    b, ci, ih, iw = x_shape
    _, co, oh, ow = gy_shape
    return Launch(B=b, IH=ih, IW=iw, Cin=ci, YH=oh, YW=ow, Cout=co, OH=oh, OW=ow, TY=g.kh, TX=g.kw,
                  sy=g.stride, sx=g.stride, dy=1, dx=1, offy=-g.pad, offx=-g.pad, reflect=int(g.reflect))


def convT_out_size(ih: int, iw: int, g: ConvGeom) -> Tuple[int, int]:
    """Output size of conv_transpose2d(stride s, padding p): (ih-1)*s - 2p + kh."""
    return (ih - 1) * g.stride - 2 * g.pad + g.kh, (iw - 1) * g.stride - 2 * g.pad + g.kw
