"""Test-only torch emulation of the ``ideas_conv_params`` launch semantics (include/ideas_hip.h).

Lets the CPU suite check the geometry derivations of ``ideas_amd.op.conv_plan`` (forward / parity-phase
dgrad / wgrad / transposed conv) against F.conv2d & friends without a GPU.  Never imported by the product.
"""
import torch


def _coords(n_out, taps, s, d, off, n_in, reflect):
    o = torch.arange(n_out).view(-1, 1)
    t = torch.arange(taps).view(1, -1)
    i = o * s + t * d + off
    if reflect:
        i = torch.where(i < 0, -i, i)
        i = torch.where(i >= n_in, 2 * n_in - 2 - i, i)
        ok = torch.ones_like(i, dtype=torch.bool)
    else:
        ok = (i >= 0) & (i < n_in)
    return i.clamp(0, n_in - 1), ok


def gather(x, L, in_scale=None):
    """x [B,IH,IW,Cin] -> patches [B,OH,OW,TY,TX,Cin] as the kernel would read them."""
    iy, oky = _coords(L.OH, L.TY, L.sy, L.dy, L.offy, L.IH, L.reflect)
    ix, okx = _coords(L.OW, L.TX, L.sx, L.dx, L.offx, L.IW, L.reflect)
    if in_scale is not None:
        x = x * in_scale.view(L.B, 1, 1, L.Cin)
    g = x[:, iy][:, :, :, ix]                      # [B,OH,TY,OW,TX,C]
    m = (oky.view(L.OH, L.TY, 1, 1) & okx.view(1, 1, L.OW, L.TX)).to(x.dtype)
    g = g * m.view(1, L.OH, L.TY, L.OW, L.TX, 1)
    return g.permute(0, 1, 3, 2, 4, 5)             # [B,OH,OW,TY,TX,C]


def run_fwd(x, L, y, in_scale=None, out_scale=None, gain=1.0):
    """Execute one forward-family launch into y [B,YH,YW,Cout] (in place)."""
    pat = gather(x, L, in_scale)
    w = L.wmat.reshape(L.Cout, L.TY, L.TX, L.Cin)
    v = torch.einsum("bhwyxc,oyxc->bhwo", pat, w) * gain
    if out_scale is not None:
        v = v * out_scale.view(L.B, 1, 1, L.Cout)
    y[:, L.ooy::L.osy, L.oox::L.osx][:, :L.OH, :L.OW] = v
    return y


def run_wgrad(gy, x, L, in_scale=None, out_scale=None, gain=1.0):
    """gw [Cout,TY,TX,Cin] of one wgrad launch (gy [B,YH,YW,Cout], x [B,IH,IW,Cin])."""
    pat = gather(x, L, in_scale)
    g = gy[:, L.ooy::L.osy, L.oox::L.osx][:, :L.OH, :L.OW]
    if out_scale is not None:
        g = g * out_scale.view(L.B, 1, 1, L.Cout)
    return torch.einsum("bhwo,bhwyxc->oyxc", g, pat) * gain
