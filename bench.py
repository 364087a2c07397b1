#!/usr/bin/env python3
"""IDEAS headline benchmark: train images/sec of the full G+D+Ex iteration at 256x256 (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W          (N > 1 and no WORLD_SIZE in the environment: spawns the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one iteration of train.py:48-221 (D phase, lazy R1 when iter % 16 == 0, G phase, Ex step, EMA)
on a synthetic batch already resident in HBM.  Workload: BASELINE.json configs[2] — N=1, sigma=1, 256x256,
batch 32 per GPU, full-width networks (channel 32, texture 2048), f32, through the HIP kernels of
libideas_hip.so.  Weak scaling: every rank owns its own 32-image shard; gradients are averaged with one RCCL
all-reduce per optimiser group (ideas_amd/ddp.py).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import random
import sys
import time

# RCCL / device-tensor sharing between the ranks of one node needs dmabuf IPC on this driver; the launcher normally exports it, a
# bare `python -m torch.distributed.run ... bench.py` may not (read when the HIP runtime initialises, i.e. after this line)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# forward GFLOP per sample at R=256, full width (measured by hooks on the reference)
F_E, F_G, F_GS, F_EX, F_DR, F_DC = 16.52, 95.99, 0.21, 0.19, 53.26, 1.00
PEAK_BF16_MFMA_TFLOPS = 2500.0   # dense, MI355X_MICROARCH.md (measured 2495)
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz
PEAK_HBM_GBS = 8000.0


def flop_per_image(elided: bool, d_reg_every: int = 16, shared: bool = True) -> float:
    """Algorithmic GFLOP per image per iteration at R=256 (rule: bwd with weight+input grads = 2F, input-only = 1F).
    `shared`: E(X) and G(S1,T1) evaluated once per iteration instead of once per phase (train_step.share_forward)."""
    d_phase = ((0.0 if shared else F_E + F_G) + F_GS + 2 * F_G + 4 * F_DR + 48 * F_DC) + 2 * (4 * F_DR + 48 * F_DC)
    g_fwd = 2 * F_E + F_GS + 3 * F_G + 3 * F_DR + 40 * F_DC + F_EX
    g_bwd = 2 * (2 * F_E + F_GS + 3 * F_G + F_EX) + 3 * F_DR + 8 * F_DC
    second = 0.0 if elided else 2 * (F_EX + F_E + F_G + F_GS) + 2 * F_E
    r1 = 6 * (F_DR + 40 * F_DC) / d_reg_every
    return d_phase + g_fwd + g_bwd + second + r1


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=48, help="timed iterations (48 = three lazy-R1 iterations at d_reg_every 16)")
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--precision", choices=["f32", "bf16"], default="f32",
                   help="activation dtype: f32 = the reference's arithmetic class (headline); bf16 = BASELINE.json configs[4] "
                        "mixed precision (bf16 activations + bf16 MFMA, f32 accumulation, f32 master weights)")
    p.add_argument("--batch", type=int, default=32, help="images per GPU")
    p.add_argument("--image-size", type=int, default=256)
    p.add_argument("--N", type=int, default=1)
    p.add_argument("--channel", type=int, default=32, help="base width (train.py:355); anything but 32 is a test configuration, named in the line")
    p.add_argument("--texture-channel", type=int, default=2048, help="texture code width (train.py:358)")
    p.add_argument("--literal-second-backward", action="store_true",
                   help="re-traverse Ex->E->G->Gstru for the Ex gradient exactly like train.py:214-215")
    p.add_argument("--no-share-forward", action="store_true",
                   help="evaluate E(X) and G(S1,T1) in both phases like train.py:58,68,145,155 (same results, +4.6 %% FLOPs)")
    p.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam + per-tensor EMA instead of FusedAdamEMA")
    p.add_argument("--cpu-baseline", choices=["auto", "skip"], default="auto")
    p.add_argument("--roofline", choices=["on", "off", "only"], default="on")
    p.add_argument("--roofline-launches", type=int, default=20)
    p.add_argument("--also-bf16", choices=["on", "off"], default="on",
                   help="after the f32 window, time a short bf16 window in the same process and report it under \"bf16\"")
    p.add_argument("--bf16-steps", type=int, default=16)
    p.add_argument("--bf16-warmup", type=int, default=4)
    return p.parse_args()


def roofline_probe_bf16(device, batch: int, launches: int):
    """bf16 mode: the dominant kernel is conv_bf16_img_kernel<64,2,true> (csrc/conv_bf16.hip; IDEAS_BF16_IMG=0: conv_bf16_kernel) on the same heaviest instance,
    G.layers.7.conv2.  One bf16 MFMA product per algorithmic product: `peak` is the dense bf16 MFMA peak itself."""
    from ideas_amd.op import conv as CV, conv_plan
    from ideas_amd.op.conv_plan import ConvGeom
    g = torch.Generator(device="cpu").manual_seed(7)
    x = torch.randn(batch, 128, 256, 256, generator=g).to(device).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(128, 128, 3, 3, generator=g).to(device).contiguous(memory_format=torch.channels_last))
    s = (torch.randn(batch, 128, generator=g) * 0.5 + 1).to(device)
    d = (torch.rand(batch, 128, generator=g) + 0.5).to(device)
    geom = ConvGeom(3, 3, 1, 1, False)
    conv_plan.cache_begin()              # as inside train_iteration: the bf16 weight pack is made once per optimiser step
    try:
        for _ in range(3):
            CV.conv_fwd_raw(x, w, geom, 0.03, lin=s, lout=d)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(launches):
            CV.conv_fwd_raw(x, w, geom, 0.03, lin=s, lout=d)
        e1.record()
        torch.cuda.synchronize()
    finally:
        conv_plan.cache_end()
    ms = e0.elapsed_time(e1) / launches
    flops = 2.0 * batch * 256 * 256 * 128 * 128 * 9
    achieved = flops / (ms * 1e-3) / 1e12
    alg_bytes = 2.0 * batch * 256 * 256 * 128 * 2          # read x + write y, bf16
    traffic, note = _pmc_traffic("conv_bf16_img_kernel", "pmc_bf16") if batch == 32 else (None, None)
    return {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_BF16_MFMA_TFLOPS, 4), "traffic": traffic, "traffic_source": note,
            "kernel": "conv_bf16_img_kernel<64,2,true> (the activation operand as an LDS image: one DMA of the 6 x 66 input pixels per 32-channel "
                      "chunk, nine taps read it at pixel offsets; v_mfma_f32_32x32x16_bf16, one product per MFMA, "
                      "f32 accumulate, per-sample weight packs) on G.layers.7.conv2: 3x3 modconv 128->128 @256x256, B=%d" % batch,
            "flop_per_launch": flops, "ms_per_launch": round(ms, 4),
            "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_hbm_gbs": round(alg_bytes / (ms * 1e-3) / 1e9, 1),
            "hbm_frac_of_8tbs": round(alg_bytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}


def roofline_probe(device, batch: int, launches: int):
    """Dominant kernel on its heaviest instance, G.layers.7.conv2: modulated 3x3, 128 -> 128 channels at 256x256
    (19.33 GFLOP per sample), timed with HIP events on the launch stream over `launches` back-to-back
    launches.  `achieved` = ALGORITHMIC f32 FLOPs / time.  Which kernel runs depends on the dispatch:
      IDEAS_MATH=b3 (default)   conv_b3_wino2d_kernel<true,false,32> (IDEAS_B3_WINO2D=0: conv_b3_wino_kernel<true,false,2>;
                                IDEAS_B3_WINO=0: conv_b3_kernel<2,2,2,2,true,false>): every f32
                                product = six bf16 MFMA products, so the pipe's ceiling for this arithmetic is the dense
                                bf16 peak / 6 (the Winograd variant issues 2/3 of them; `peak` does not credit that);
      IDEAS_MATH=f32            conv3x3_wino_kernel<true,false> (1-D Winograd on the f32 MFMA: executes 2/3 of the
                                multiplies), or with IDEAS_WINOGRAD=0 the direct conv_igemm_kernel<2,2,2,2,true,false,true>."""
    from ideas_amd import _lib
    from ideas_amd.op import conv as CV
    from ideas_amd.op.conv_plan import ConvGeom
    g = torch.Generator(device="cpu").manual_seed(7)
    x = torch.randn(batch, 128, 256, 256, generator=g).to(device).contiguous(memory_format=torch.channels_last)
    w = torch.randn(128, 128, 3, 3, generator=g).to(device).contiguous(memory_format=torch.channels_last)
    s = (torch.randn(batch, 128, generator=g) * 0.5 + 1).to(device)
    d = (torch.rand(batch, 128, generator=g) + 0.5).to(device)
    geom = ConvGeom(3, 3, 1, 1, False)
    for _ in range(3):
        CV.conv_fwd_raw(x, w, geom, 0.03, lin=s, lout=d)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(launches):
        CV.conv_fwd_raw(x, w, geom, 0.03, lin=s, lout=d)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / launches
    flops = 2.0 * batch * 256 * 256 * 128 * 128 * 9
    achieved = flops / (ms * 1e-3) / 1e12
    where = " on G.layers.7.conv2: 3x3 modconv 128->128 @256x256, B=%d" % batch
    if CV.MATH == _lib.F32_B3:
        peak = PEAK_BF16_MFMA_TFLOPS / 6.0
        b3w = CV.B3_WINO
        kname = "conv_b3_wino2d_kernel" if b3w else "conv_b3_kernel"
        traffic, traffic_note = _pmc_traffic(kname, "pmc_b3w" if b3w else "pmc_b3") if batch == 32 else (None, None)
        executed = achieved * (4.0 if b3w else 6.0)       # bf16 MFMA FLOPs issued per algorithmic f32 FLOP
        return {"bound": "mfma", "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_note,
                "kernel": ("conv_b3_wino2d_kernel<true,false,32> (1-D Winograd F(2,3): 2/3 of the products; 2 x 32 pair patches whose "
                           "input rows are staged once for the three ky taps; " if b3w else
                           "conv_b3_kernel<2,2,2,2,true,false> (") +
                          "f32 operands split exactly into 3 bf16 planes, 6 v_mfma_f32_32x32x16_bf16 products per f32 "
                          "product, f32 accumulate)" + where,
                "peak_note": "dense bf16 MFMA peak %.0f TFLOP/s / 6 plane products per f32 product (Winograd not credited); "
                             "the f32 MFMA peak is %.1f" % (PEAK_BF16_MFMA_TFLOPS, PEAK_F32_MFMA_TFLOPS),
                "flop_per_launch": flops, "ms_per_launch": round(ms, 4),
                "executed_bf16_tflops": round(executed, 1), "mfma_executed_frac": round(executed / PEAK_BF16_MFMA_TFLOPS, 4),
                "vs_f32_mfma_peak": round(achieved / PEAK_F32_MFMA_TFLOPS, 4)}
    wino = CV.WINOGRAD
    traffic, traffic_note = _pmc_traffic_of("wino", "r02_pmc_f32") if (wino and batch == 32) else (None, None)
    kernel = ("conv3x3_wino_kernel<true,false> (1-D Winograd F(2,3); executes 2/3 of the algorithmic multiplies)" if wino
              else "conv_igemm_kernel<2,2,2,2,true,false,true> (direct implicit GEMM)")
    return {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_note,
            "kernel": kernel + where, "flop_per_launch": flops, "ms_per_launch": round(ms, 4),
            "mfma_executed_frac": round(achieved * (2.0 / 3.0 if wino else 1.0) / PEAK_F32_MFMA_TFLOPS, 4)}


def roofline_probe_wgrad(device, batch: int, launches: int, bf16: bool):
    """Second roofline entry: the weight-gradient kernel of the same G.layers.7.conv2 launch -- the kernel family
    with the largest summed share of the step after the forward family -- timed with HIP events, accumulating into an existing
    gradient buffer exactly as inside grad_sink.  Same algorithmic FLOPs and the same peak as the forward entry."""
    from ideas_amd import _lib
    from ideas_amd.op import conv as CV
    from ideas_amd.op.conv_plan import ConvGeom
    g = torch.Generator(device="cpu").manual_seed(8)
    adt = torch.bfloat16 if bf16 else torch.float32
    x = torch.randn(batch, 128, 256, 256, generator=g).to(device).to(adt).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(batch, 128, 256, 256, generator=g).to(device).to(adt).contiguous(memory_format=torch.channels_last)
    s = (torch.randn(batch, 128, generator=g) * 0.5 + 1).to(device)
    d = (torch.rand(batch, 128, generator=g) + 0.5).to(device)
    acc = torch.zeros(128, 128, 3, 3, device=device).contiguous(memory_format=torch.channels_last)
    geom = ConvGeom(3, 3, 1, 1, False)
    run = lambda: CV.conv_wgrad_raw(gy, x, geom, (128, 128, 3, 3), 0.03, lin=s, lout=d, out=acc)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(launches):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / launches
    flops = 2.0 * batch * 256 * 256 * 128 * 128 * 9
    achieved = flops / (ms * 1e-3) / 1e12
    if bf16:
        peak, kern = PEAK_BF16_MFMA_TFLOPS, ("conv_bf16_wgrad3_kernel<true,false> (tap-fused 3x3: one DMA'd G row + one new X window row per step, nine taps on "
                                             "ds_read_b64_tr_b16 transpose reads at pixel offsets, one bf16 product per MFMA)")
    elif CV.MATH == _lib.F32_B3:
        peak = PEAK_BF16_MFMA_TFLOPS / 6.0
        kern = ("conv_b3_wgrad3_kernel<1,true,false> (tap-fused 3x3, rolling activation window in LDS; exact 3-way bf16 split of both operands, "
                "6 bf16 MFMA products per f32 product)")
    else:
        peak, kern = PEAK_F32_MFMA_TFLOPS, "conv3x3_wino_wgrad_kernel / conv_wgrad_kernel (f32 MFMA)"
    elem = 2 if bf16 else 4
    traffic = note = None
    if batch == 32 and (bf16 or CV.MATH == _lib.F32_B3):
        traffic, note = _pmc_traffic("conv_bf16_wgrad3_kernel" if bf16 else "conv_b3_wgrad3_kernel", "pmc_bf16wg" if bf16 else "pmc_b3wg")
    return {"bound": "mfma", "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
            "traffic": traffic, "traffic_source": note, "kernel": kern + " on the weight gradient of G.layers.7.conv2: 128->128 @256x256, B=%d, column strips x row ranges in "
            "XCD-banded order (csrc/common.hpp)" % batch, "flop_per_launch": flops, "ms_per_launch": round(ms, 4),
            "algorithmic_bytes_per_launch": 2.0 * batch * 256 * 256 * 128 * elem}


def roofline_probe_hbm(device, batch: int, launches: int, bf16: bool):
    """Third entry: the largest HBM-bound kernel of the step, the 4x4 blur (blur4_f32_c2 / blur4_bf16x8_c2, upfirdn2d.hip), on the
    activation it is most expensive on (Dreal.1 / G.7: 128 channels at 256x256).  `achieved` = ALGORITHMIC bytes (read the
    input once, write the output once) / time, against the 8 TB/s HBM3E peak."""
    from ideas_amd.model import make_kernel
    from ideas_amd.op.upfirdn2d import upfirdn2d_raw
    g = torch.Generator(device="cpu").manual_seed(9)
    adt = torch.bfloat16 if bf16 else torch.float32
    x = torch.randn(batch, 128, 256, 256, generator=g).to(device).to(adt).contiguous(memory_format=torch.channels_last)
    fir = make_kernel((1, 3, 3, 1)).to(device)
    run = lambda: upfirdn2d_raw(x, fir, (1, 1), (1, 1), (2, 2, 2, 2), (257, 257), True)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(launches):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / launches
    nbytes = (batch * 128 * 256 * 256 + batch * 128 * 257 * 257) * (2 if bf16 else 4)
    gbs = nbytes / (ms * 1e-3) / 1e9
    traffic, note = _pmc_traffic("blur4_bf16x8_c2" if bf16 else "blur4_f32_c2", "pmc_blurbf16" if bf16 else "pmc_blurf32", smallest_grid=True) if batch == 32 else (None, None)
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": traffic,
            "traffic_source": note,
            "kernel": ("blur4_bf16x8_c2<0>" if bf16 else "blur4_f32_c2<0>") + " (4x4 FIR of a downsampling ConvLayer, two output columns per thread) on "
            "[%d,128,256,256] -> 257x257, pad (2,2)" % batch, "algorithmic_bytes_per_launch": nbytes, "ms_per_launch": round(ms, 4)}


def _time_launches(run, launches):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(launches):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / launches


def roofline_probe_direct(device, batch: int, launches: int, bf16: bool):
    """Fourth entry: the DIRECT forward-family kernel (the stride-2 / transposed layers the Winograd kernel does
    not take; a quarter of the f32 step's kernel time) on its heaviest instance, G.layers.7.conv1: the stride-2 transposed 3x3
    modconv 256 -> 128 from 128x128 to 257x257 (9.66 GFLOP per sample; the four output-parity phases with
    4/2/2/1 taps, so no product is spent on stuffed zeros).  One call = everything conv_dgrad_raw launches for it."""
    from ideas_amd import _lib
    from ideas_amd.op import conv as CV, conv_plan
    from ideas_amd.op.conv_plan import ConvGeom, convT_out_size
    g = torch.Generator(device="cpu").manual_seed(10)
    adt = torch.bfloat16 if bf16 else torch.float32
    x = torch.randn(batch, 256, 128, 128, generator=g).to(device).to(adt).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(128, 256, 3, 3, generator=g).to(device).contiguous(memory_format=torch.channels_last))
    s = (torch.randn(batch, 256, generator=g) * 0.5 + 1).to(device)
    d = (torch.rand(batch, 128, generator=g) + 0.5).to(device)
    geom = ConvGeom(3, 3, 2, 0, False)
    out_hw = convT_out_size(128, 128, geom)
    wt = w.transpose(0, 1)
    conv_plan.cache_begin()
    try:
        ms = _time_launches(lambda: CV.conv_dgrad_raw(x, wt, geom, out_hw, 0.02, lin=s, lout=d), launches)
    finally:
        conv_plan.cache_end()
    flops = 2.0 * batch * 128 * 128 * 256 * 128 * 9
    achieved = flops / (ms * 1e-3) / 1e12
    if bf16:
        peak, kern = PEAK_BF16_MFMA_TFLOPS, "conv_bf16_multi_kernel (LDS-DMA implicit GEMM, the four parity phases in one grid)"
    elif CV.MATH == _lib.F32_B3:
        peak, kern = PEAK_BF16_MFMA_TFLOPS / 6.0, ("conv_b3_tphase_kernel<true> (the four parity phases from one LDS image of the input; exact 3-way bf16 "
                                                   "split, 6 bf16 MFMA products per f32 product) + the edge strips on conv_b3_multi_kernel")
    else:
        peak, kern = PEAK_F32_MFMA_TFLOPS, "conv_igemm_kernel (f32 MFMA)"
    elem = 2 if bf16 else 4
    traffic = note = None
    if batch == 32 and not bf16 and CV.MATH == _lib.F32_B3:
        traffic, note = _pmc_traffic("conv_b3_tphase_kernel", "pmc_b3tp")
    return {"bound": "mfma", "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
            "traffic": traffic, "traffic_source": note, "kernel": kern + " on G.layers.7.conv1: stride-2 transposed 3x3 modconv 256->128, 128x128 -> 257x257, B=%d "
            "(all output-parity phases of the layer)" % batch, "flop_per_launch": flops, "ms_per_launch": round(ms, 4),
            "algorithmic_bytes_per_launch": float(batch * (128 * 128 * 256 + 257 * 257 * 128) * elem)}


def roofline_probe_s2(device, batch: int, launches: int):
    """Stride-2 entry: Blur(pad (2,2)) -> 3x3 / stride-2 conv of Dreal.1.conv2 (128 -> 128 channels, 256x256 -> 128x128,
    on the 3B images of the discriminator's fake pass) as ONE kernel, conv_b3_s2fir_kernel (csrc/conv_b3_s2fir.hip); beside it the
    two-kernel chain it replaces (blur4_f32_c2 + conv_b3_kernel) and the layer's weight gradient (tap-fused stride-2 kernel).
    FLOPs: the convolution's only (the FIR's are not counted)."""
    from ideas_amd import _lib
    from ideas_amd.model import make_kernel
    from ideas_amd.op import conv as CV, conv_plan
    from ideas_amd.op.conv_plan import ConvGeom
    from ideas_amd.op.upfirdn2d import upfirdn2d_raw
    if CV.MATH != _lib.F32_B3:
        return None
    g = torch.Generator(device="cpu").manual_seed(12)
    B3 = 3 * batch
    x = torch.randn(B3, 128, 256, 256, generator=g).to(device).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(128, 128, 3, 3, generator=g).to(device).contiguous(memory_format=torch.channels_last))
    bias = torch.zeros(128, device=device)
    fir = make_kernel((1, 3, 3, 1)).to(device)
    geom = ConvGeom(3, 3, 2, 0, False)
    flops = 2.0 * B3 * 128 * 128 * 128 * 128 * 9
    peak = PEAK_BF16_MFMA_TFLOPS / 6.0
    conv_plan.cache_begin()
    try:
        if not CV.blur_conv_s2_ok(x, w, fir, (2, 2)):
            return None
        ms_f = _time_launches(lambda: CV.blur_conv_s2_raw(x, w, fir, (2, 2), 0.03, bias=bias, act=True, act_gain=1.0), launches)
        ms_fx = _time_launches(lambda: CV.blur_conv_s2_raw(x, w, fir, (2, 2), 0.03, bias=bias, act=True, act_gain=1.0, want_xb=True), launches)
        xb = upfirdn2d_raw(x, fir, (1, 1), (1, 1), (2, 2, 2, 2), (257, 257), True)
        ms_blur = _time_launches(lambda: upfirdn2d_raw(x, fir, (1, 1), (1, 1), (2, 2, 2, 2), (257, 257), True), launches)
        ms_conv = _time_launches(lambda: CV.conv_fwd_raw(xb, w, geom, 0.03, bias=bias, act=True), launches)
        gy = torch.randn(B3, 128, 128, 128, generator=g).to(device).contiguous(memory_format=torch.channels_last)
        acc = torch.zeros(128, 128, 3, 3, device=device).contiguous(memory_format=torch.channels_last)
        ms_wg = _time_launches(lambda: CV.conv_wgrad_raw(gy, xb, geom, (128, 128, 3, 3), 0.03, out=acc), launches)
    finally:
        conv_plan.cache_end()
    tf = lambda ms: round(flops / (ms * 1e-3) / 1e12, 2)
    traffic, note = _pmc_traffic("conv_b3_s2fir_kernel<4, 4, 1, false,", "pmc_b3s2") if batch == 32 else (None, None)    # (not the side-output variant)
    alg = float(B3 * (256 * 256 + 128 * 128) * 128 * 4)
    return {"bound": "mfma", "achieved": tf(ms_f), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(tf(ms_f) / peak, 4), "traffic": traffic,
            "traffic_source": note,
            "kernel": "conv_b3_s2fir_kernel<4,4,1,false> (Blur + 3x3 / stride-2 conv + bias + leaky-ReLU in one kernel: four producer waves build "
                      "the blurred, 3-way split LDS image of an 8 x 16 output patch per 16-channel chunk -- one 16-byte load per raw row and "
                      "thread, horizontal taps by DPP row shifts --, four consumer waves contract the nine stride-2 taps from it; 6 bf16 "
                      "MFMA products per f32 product) on Dreal.1.conv2: 128->128, 256x256 -> 128x128, B=%d (3 x %d)" % (B3, batch),
            "flop_per_launch": flops, "ms_per_launch": round(ms_f, 4), "algorithmic_bytes_per_launch": alg,
            "with_blurred_side_output": {"ms_per_launch": round(ms_fx, 4), "tflops": tf(ms_fx)},
            "two_kernel_chain": {"blur_ms": round(ms_blur, 4), "conv_ms": round(ms_conv, 4), "conv_tflops": tf(ms_conv),
                                 "chain_ms": round(ms_blur + ms_conv, 4), "chain_tflops": tf(ms_blur + ms_conv)},
            "weight_gradient": {"kernel": ("conv_b3_wgrad_kernel (generic split weight gradient, stride 2)" if os.environ.get("IDEAS_B3_WGRAD3_S2", "1") == "0"
                                           else "conv_b3_wgrad3_kernel<2,false,false> (tap-fused, one window row per barrier)") + " on the blurred tensor",
                                "ms_per_launch": round(ms_wg, 4), "tflops": tf(ms_wg), "frac": round(tf(ms_wg) / peak, 4)}}


def roofline_probe_bias_act_bwd(device, batch: int, launches: int, bf16: bool):
    """Second HBM entry: the backward of fused_leaky_relu (bias_act_nhwc_v4, grad = 1: gradient in, saved activation in, gradient out,
    the bias gradient reduced in the same pass -- fused_act.py:20-49 + the separate .sum of :38) on [B,128,256,256].
    Algorithmic bytes: 3 tensors (12 B/elem in f32)."""
    from ideas_amd.op.fused_act import bias_act_raw
    g = torch.Generator(device="cpu").manual_seed(11)
    adt = torch.bfloat16 if bf16 else torch.float32
    gy = torch.randn(batch, 128, 256, 256, generator=g).to(device).to(adt).contiguous(memory_format=torch.channels_last)
    out = torch.randn(batch, 128, 256, 256, generator=g).to(device).to(adt).contiguous(memory_format=torch.channels_last)
    acc = torch.zeros(128, device=device)
    ms = _time_launches(lambda: bias_act_raw(gy, None, out, 1, 0.2, 2 ** 0.5, bias_grad_into=acc), launches)
    nbytes = 3.0 * gy.numel() * (2 if bf16 else 4)
    gbs = nbytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": None,
            "kernel": "bias_act_nhwc_v4 (grad=1: leaky-ReLU backward + bias-gradient reduction in one pass) on [%d,128,256,256] %s"
                      % (batch, "bf16" if bf16 else "f32"), "algorithmic_bytes_per_launch": nbytes, "ms_per_launch": round(ms, 4)}


def _src_sha(files):
    import hashlib
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(ROOT, "ideas_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


PMC_ROUNDS = ("r06", "r05")      # newest committed PMC set first; a set is used only while its sidecar hash matches the kernel sources


def _pmc_traffic(kernel_substr: str, name: str, smallest_grid: bool = False):
    """`name` = "pmc_<set>": the first round of PMC_ROUNDS whose profiles/<round>_pmc_<set>_* passes are current (see _pmc_traffic_of)."""
    note = None
    for rnd in PMC_ROUNDS:
        t, n_ = _pmc_traffic_of(kernel_substr, rnd + "_" + name, smallest_grid)
        if t is not None:
            return t, n_
        note = note or n_
    return None, note


def _pmc_traffic_of(kernel_substr: str, prefix: str, smallest_grid: bool = False):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/<prefix>_*.csv;
    the counters cannot be read from inside this process): 2 x FETCH_SIZE (gfx950 reports half the bytes of a wide
    coalesced read, MI355X_MICROARCH.md §HBM) + WRITE_SIZE, both in KB.  The passes carry a sidecar
    profiles/<prefix>_source.json = {"files": [...], "sha": ...} written by tools/collect_profiles.sh with the hash of the
    kernel sources they measured; if the sources changed since (or there is no sidecar) the figure is stale -> null.
    ``smallest_grid``: the pass holds launches of that kernel on several shapes (the stride-2 probe calls the blur on 3B images);
    the entry's launch is the one with the smallest grid."""
    import csv
    try:
        side = json.load(open(os.path.join(ROOT, "profiles", prefix + "_source.json")))
        if _src_sha(side["files"]) != side["sha"]:
            return None, "profiles/%s_* are older than csrc/%s: stale, not reported" % (prefix, ",".join(side["files"]))
    except Exception:
        return None, "no PMC pass of the current kernel sources under profiles/ (%s_source.json)" % prefix
    try:
        vals = {}
        for name, fn in (("FETCH_SIZE", prefix + "_fetch_size.csv"), ("WRITE_SIZE", prefix + "_write_size.csv")):
            rows = [(int(r["Grid_Size"]), float(r["Counter_Value"])) for r in csv.DictReader(open(os.path.join(ROOT, "profiles", fn)))
                    if r["Counter_Name"] == name and kernel_substr in r["Kernel_Name"]]
            if smallest_grid:
                g0 = min(g for g, _ in rows)
                rows = [r for r in rows if r[0] == g0]
            vals[name] = sum(v for _, v in rows) / len(rows)
        return int((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024), \
            "profiles/%s_fetch_size.csv + %s_write_size.csv (separate --pmc passes; FETCH_SIZE x2)" % (prefix, prefix)
    except Exception:
        return None, None


def eager_complete(a):
    """Complete eager iterations at the headline shape (tools/eager_cached.sh): profiles/r06_eager_full.json (round 6's re-measurement
    from the committed find-db; r04_eager_full.json is round 4's) -- MIOpen's default find
    mode, every convolution searched once and then served from the find-db -- when it exists, else profiles/r04_eager_fast.json --
    MIOPEN_FIND_MODE=FAST, the immediate-mode solver choice without a search.  Used only while the sidecar matches: same torch
    build, same oracle / comparator sources, same batch -- otherwise stale -> None."""
    import hashlib
    src = b"".join(open(os.path.join(ROOT, f), "rb").read() for f in ("oracle/torch_ref.py", "tests/eager_baseline.py"))
    sha = hashlib.sha256(src).hexdigest()[:16]
    for name in ("r06_eager_full.json", "r04_eager_full.json", "r04_eager_fast.json"):      # newest measurement first
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
            if d["source_sha16"] != sha or d["torch_version"] != torch.__version__ or d["batch"] != a.batch:
                continue
            if d.get("convs") == "MIOpen" and d["steps"] >= 1:
                return dict(d, file="profiles/" + name)
        except Exception:
            continue
    return None


def vs_rocm_eager(ips: float, a, world: int):
    """The north_star's target (>= 1.5x the stock PyTorch-ROCm eager path at 256x256 on one MI355X), two measurements:

    * `ratio_lower_bound` (round 2, profiles/r02_eager_comparator.json, tools/eager_partial.py; B = 32, f32, cudnn.benchmark=True,
      every call in isolation with the searched-best solver): the time the eager step spends in 334 of its 561 convolution calls
      (94 % of its convolution FLOPs), plus -- measured separately, tests/eager_baseline.py --stub-convs -- the time of the same
      eager step with every convolution replaced by an allocation (bias/act, per-sample weight materialisation, residual merges,
      pads, resampling, losses, Adam).  Eager PyTorch issues all of it on one stream, so the sum is a LOWER bound of its iteration
      time (the 227 unmeasured convolution calls count as zero) and the ratio a lower bound on ours / it.
    * `complete_iteration` (round 4, tools/eager_cached.sh): whole iterations of tests/eager_baseline.py (the oracle's step on cuda:0,
      composite torch ops, MIOpen convolutions) -- see eager_complete().  This image ships no gfx950 find-db, so MIOpen's default
      mode searches ~560 problem-directions on first use (single FIR convolutions on [B*C, 1, H, W] views take 4-12 minutes each);
      the find-db is carried between GPU calls as a tarball."""
    headline = world == 1 and a.image_size == 256 and a.N == 1 and a.precision == "f32" and (a.channel, a.texture_channel) == (32, 2048)
    if not headline:
        return None          # the comparator was measured for the f32 headline configuration only (f32 NCHW)
    out = {}
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r02_eager_comparator.json")))
        if a.batch == d["batch"]:
            nonconv = d.get("nonconv_ms_per_iteration", 0.0)
            lb_ms = d["measured_conv_ms_per_iteration"] + nonconv
            ub = d["batch"] / (lb_ms * 1e-3)
            out = {"measured": "lower bound of the eager iteration time = measured convolution calls + measured conv-free step",
                   "eager_conv_ms_measured": d["measured_conv_ms_per_iteration"], "conv_flop_coverage": d["conv_flop_coverage"],
                   "eager_conv_tflops": d["measured_conv_tflops"], "eager_nonconv_ms_measured": nonconv,
                   "eager_iteration_ms_lower_bound": round(lb_ms, 1), "eager_images_per_sec_upper_bound": round(ub, 2),
                   "ratio_lower_bound": round(ips / ub, 3),
                   "ratio_with_unmeasured_convs_at_same_rate":
                       round(ips / (d["batch"] / ((d["extrapolated_conv_ms_at_same_rate"] + nonconv) * 1e-3)), 3),
                   "source": "profiles/r02_eager_comparator.json (tools/eager_partial.py, tests/eager_baseline.py --stub-convs)"}
    except Exception:
        pass
    full = eager_complete(a)
    if full is not None:
        out["complete_iteration"] = {
            "find_mode": full.get("find_mode", "default"), "eager_images_per_sec": full["eager_gpu_images_per_sec"],
            "eager_ms_per_step": full["ms_per_step"], "eager_steps_timed": full["steps"],
            "ratio": round(ips / full["eager_gpu_images_per_sec"], 3),
            "source": "%s (torch %s, sources %s)" % (full["file"], full["torch_version"], full["source_sha16"])}
    return out or None


def _cpu_baseline_worker(R: int, threads: int, B: int = 1, warm: int = 0, iters: int = 1):
    """Runs in a child process: `warm` untimed + `iters` timed oracle iterations (D phase with backward and Adam step, G/Ex phase
    with both backwards and Adam steps) at batch B; prints the per-iteration seconds."""
    import oracle.torch_ref as O
    from ideas_amd.models import init_model
    from ideas_amd import train_step as TS
    torch.set_num_threads(threads)
    big = R >= 256
    args = TS.default_args(image_size=R, use_dco=big)
    torch.manual_seed(0)
    nets = {}
    for n in ("E", "G", "Gstru", "Ex", "Dreal", "Dco", "Ddist"):
        if n == "Dco" and not big:
            continue
        m = init_model(TS.NET_CLASSES[n], args)
        nets[n] = {k: v.detach().contiguous().requires_grad_(v.is_floating_point() and not k.endswith("kernel"))
                   for k, v in m.state_dict().items()}
    cfg = O.Cfg(image_size=R)
    sargs = O.StepArgs(use_dco=big)
    X = torch.rand(B, 3, R, R) * 2 - 1
    random.seed(0)
    s = R // 16
    d_params = [p for n in ("Dreal", "Dco", "Ddist") if n in nets for p in nets[n].values() if p.requires_grad]
    g_params = [p for n in ("E", "G", "Gstru") for p in nets[n].values() if p.requires_grad]
    ex_params = [p for p in nets["Ex"].values() if p.requires_grad]
    r = 16 / 17
    opts = (torch.optim.Adam(d_params, lr=0.002 * r, betas=(0.0, 0.99 ** r)), torch.optim.Adam(g_params, lr=0.002, betas=(0.0, 0.99)),
            torch.optim.Adam(ex_params, lr=0.002, betas=(0.0, 0.99)))

    def one():
        dr = O.StepDraws(Z_d=torch.rand(B, 1, s, s) * 2 - 1, T2_d=torch.rand(B, 2048) * 2 - 1,
                         Z_g=torch.rand(B, 1, s, s) * 2 - 1, T2_g=torch.rand(B, 2048) * 2 - 1)
        if big:
            dr.boxes_d_fake, dr.boxes_d_real = O.draw_boxes(R, R, 8), O.draw_boxes(R, R, 8)
            dr.boxes_d_ref, dr.boxes_g_fake, dr.boxes_g_ref = O.draw_boxes(R, R, 32), O.draw_boxes(R, R, 8), O.draw_boxes(R, R, 32)
        total, _, _ = O.d_phase(nets, cfg, sargs, X, dr)
        for p, g in zip(d_params, torch.autograd.grad(total, d_params, allow_unused=True)):
            p.grad = g
        opts[0].step()
        total, ex_loss, _, _ = O.g_phase(nets, cfg, sargs, X, dr, 1)
        gex = torch.autograd.grad(ex_loss, ex_params, retain_graph=True)
        for p, g in zip(g_params, torch.autograd.grad(total, g_params, allow_unused=True)):
            p.grad = g
        opts[1].step()
        for p, g in zip(ex_params, gex):
            p.grad = g
        opts[2].step()

    for _ in range(warm):
        one()
    times = []
    for _ in range(iters):
        t0 = time.perf_counter()
        one()
        times.append(time.perf_counter() - t0)
    print(json.dumps({"seconds": sum(times) / len(times), "each": [round(t, 2) for t in times], "threads": torch.get_num_threads()}))


def _cpu_run(R, B, warm, iters, limit):
    import subprocess
    threads = min(os.cpu_count() or 1, 64)
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(R), str(threads), str(B), str(warm), str(iters)],
                           capture_output=True, text=True, timeout=limit, cwd=ROOT)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode == 0 and line:
            return json.loads(line[-1]), threads
    except subprocess.TimeoutExpired:
        pass
    return None, threads


def cpu_baseline():
    """The oracle's step (oracle/torch_ref.py, the CPU restatement pinned to the reference) on the host cores, at the workload of
    the headline: 256x256 with Dco, full width.  Bounded sample: 1 warm-up + 2 timed iterations (D phase + G/Ex phase, forward,
    backward and the three Adam steps; no lazy-R1 pass) at batch 1 -- a batch-32 iteration would take minutes.  Child process (thread count =
    min(host threads, 64)) so a slow host cannot hang the bench; falls back to the 64x64 sub-step if it does not finish."""
    for R, limit in ((256, 240), (64, 120)):
        d, threads = _cpu_run(R, 1, 1, 2, limit)
        if d is not None:
            return {"value": round(1.0 / d["seconds"], 5), "unit": "images/sec", "cores": d["threads"], "kind": "port",
                    "sample": "mean of 2 timed iterations after 1 warm-up (D phase + G/Ex phase fwd+bwd + Adam steps, no R1), batch 1, "
                              "%dx%d, full width, %s; %s s" % (R, R, "with Dco" if R >= 256 else "Dco-less sub-step", d["each"])}
    return {"value": None, "unit": "images/sec", "cores": threads, "kind": "port", "sample": "timed out"}


def cpu_baseline_config1():
    """BASELINE.json configs[0] beside it: 64x64, batch 4, the Dco-less sub-step (the reference's Dco cannot run below 256x256,
    models.py:400): 1 warm-up + 3 timed iterations on the host cores."""
    d, threads = _cpu_run(64, 4, 1, 3, 240)
    if d is not None:
        return {"value": round(4.0 / d["seconds"], 4), "unit": "images/sec", "cores": d["threads"], "kind": "port",
                "sample": "mean of 3 timed iterations after 1 warm-up (D phase + G/Ex phase fwd+bwd + Adam steps, Dco-less sub-step, no "
                          "R1), batch 4, 64x64, full width; %s s" % d["each"]}
    return {"value": None, "unit": "images/sec", "cores": threads, "kind": "port", "sample": "timed out"}


def extraction_probe(trainer, args, X, device):
    """BASELINE.json's metric names the secret-bit extraction accuracy next to images/sec: the sender / receiver block of
    train.py:249-293 (bits -> Z -> Gstru -> G -> E -> Ex -> bits, sigma = 1, delta = 50 %) on the EMA networks as they stand after the
    timed window, all B images on the GPU (train_step.extraction_test).  The weights are a few dozen iterations from random
    initialisation, so ACC sits near 0.5 -- what the entry pins is that the DECISIONS are the reference arithmetic's: image 0 goes,
    with the same EMA weights, message, jitter and texture code, through the CPU oracle in the cpu_baseline leg (finish_extraction)."""
    import tempfile
    from ideas_amd import train_step as TS
    g = torch.Generator().manual_seed(4321)
    B, s = X.shape[0], X.shape[-1] // 16
    M = torch.randint(0, 2, (B, args.N * s * s), generator=g).float()
    jitter = torch.rand(B, args.N * s * s, generator=g)
    T2 = torch.rand(B, args.texture_channel, generator=g) * 2 - 1
    hat_Z, hat_M, acc, l1 = TS.extraction_test(trainer, args, X, M, T2.to(device), False, jitter=jitter)
    fd, path = tempfile.mkstemp(suffix=".pt", prefix="ideas_extraction_")
    os.close(fd)
    nets = {n: {k: v.detach().float().cpu().contiguous() for k, v in trainer[n + "_ema"].state_dict().items()} for n in ("E", "G", "Gstru", "Ex")}
    torch.save({"nets": nets, "X": X[:1].float().cpu().contiguous(), "M": M[:1], "jitter": jitter[:1], "T2": T2[:1],
                "cfg": dict(channel=args.channel, structure_channel=args.structure_channel, texture_channel=args.texture_channel, N=args.N,
                            image_size=int(X.shape[-1]), channel_multiplier=args.channel_multiplier)}, path)
    return {"bits": int(M.numel()), "acc": round(float(acc), 5), "l1": round(float(l1), 5), "sigma": 1, "delta": 0.5, "images": B,
            "nets": "EMA copies after the timed window (random initialisation + warm-up + timed iterations: ACC ~ 0.5 by construction)",
            "_hat_Z0": hat_Z[:1].float().cpu(), "_hat_M0": hat_M[:1].float().cpu(), "_M0": M[:1].clone(), "_file": path}


def _oracle_extraction_worker(path: str, threads: int):
    """Child process of the cpu_baseline leg: the CPU oracle's sender / receiver block on the dumped EMA weights (one image)."""
    import oracle.torch_ref as O
    torch.set_num_threads(threads)
    d = torch.load(path, map_location="cpu", weights_only=False)
    cfg = O.Cfg(**d["cfg"])
    hat_Z, hat_M, acc, l1 = O.extraction_test(d["nets"], cfg, d["X"], d["M"], d["jitter"], d["T2"], False)
    print(json.dumps({"hat_Z": hat_Z.flatten().tolist(), "hat_M": hat_M.flatten().tolist(), "acc": float(acc), "l1": float(l1)}))


def finish_extraction(e, f32: bool):
    """cpu_baseline leg: run the oracle child on the dump of extraction_probe and compare image 0's decisions bit for bit."""
    import subprocess
    if not e or "_file" not in e:
        return e
    path, z_gpu, m_gpu, m0 = e.pop("_file"), e.pop("_hat_Z0"), e.pop("_hat_M0"), e.pop("_M0")
    try:
        threads = min(os.cpu_count() or 1, 64)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--oracle-extraction-worker", path, str(threads)],
                           capture_output=True, text=True, timeout=240, cwd=ROOT)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            e["oracle"] = "failed: " + (r.stderr or "")[-200:]
            return e
        d = json.loads(line[-1])
        z_ref = torch.tensor(d["hat_Z"]).view_as(z_gpu)
        m_ref = torch.tensor(d["hat_M"]).view_as(m_gpu)
        flipped = int((m_ref != m_gpu).sum())
        e["oracle_image0"] = {"bits": int(m_ref.numel()), "decisions_equal_oracle": flipped == 0, "bits_flipped": flipped,
                              "hat_Z_rel_err": float("%.3g" % float((z_gpu - z_ref).abs().max() / z_ref.abs().max())),
                              "acc_oracle": round(d["acc"], 5), "acc_gpu": round(float(1 - (m_gpu - m0).abs().mean()), 5),
                              "min_abs_hat_Z": float("%.3g" % float(z_ref.abs().min())), "max_abs_hat_Z": float("%.3g" % float(z_ref.abs().max())),
                              # (bf16 policy, DESIGN 3.6: a flipped bit must sit at the decision boundary -- |hat_Z| of every flipped bit, oracle's value)
                              "abs_hat_Z_of_flipped_bits": [float("%.3g" % v) for v in z_ref.flatten()[(m_ref != m_gpu).flatten()].abs().tolist()][:16],
                              "oracle": "oracle/torch_ref.py::extraction_test (f32, CPU, %d threads) on the same EMA weights, message, jitter, "
                                        "texture code" % threads + ("" if f32 else "; this run is bf16 mixed precision: flipped bits are reported, not promised zero")}
    finally:
        try:
            os.remove(path)
        except OSError:
            pass
    return e


def _self_launch(a) -> int:
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): start the N ranks here, one process per
    GPU, exactly as the driver's `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` does
    (the shape of stylegan2/train.py:372-373 under torch.distributed.launch).  Rank 0's JSON line is the only stdout line."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < a.gpus and os.environ.get("IDEAS_BENCH_SHARE_GPU") != "1":
        raise SystemExit(f"--gpus {a.gpus} but only {have} GPU(s) are visible (IDEAS_BENCH_SHARE_GPU=1 + IDEAS_DIST_BACKEND=gloo "
                         "runs several ranks on one device, for tests of the multi-rank path only)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(8, (os.cpu_count() or 8) // a.gpus))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env, cwd=ROOT).returncode


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "--cpu-baseline-worker":
        _cpu_baseline_worker(*[int(v) for v in sys.argv[2:7]])
        return
    if len(sys.argv) >= 4 and sys.argv[1] == "--oracle-extraction-worker":
        _oracle_extraction_worker(sys.argv[2], int(sys.argv[3]))
        return
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_self_launch(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    # IDEAS_BENCH_SHARE_GPU=1 (+ IDEAS_DIST_BACKEND=gloo) lets the multi-rank code path be exercised on a 1-GPU box
    if os.environ.get("IDEAS_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    if world > 1:
        # one process per GPU on a shared host: give every rank its own slice of the host cores (the step issues ~5000 small
        # launches per iteration from Python; ranks competing for the same cores would serialise them)
        try:
            cores = sorted(os.sched_getaffinity(0))
            per = max(1, len(cores) // world)
            mine = cores[int(os.environ.get("LOCAL_RANK", "0")) * per:(int(os.environ.get("LOCAL_RANK", "0")) + 1) * per]
            if mine:
                os.sched_setaffinity(0, mine)
                torch.set_num_threads(max(1, min(8, len(mine))))
        except (AttributeError, OSError):
            pass
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # IDEAS_DDP_FORCE_COLLECTIVE=1 (test switch, ideas_amd/ddp.py): one rank under a launcher still joins a process group and runs
    # every collective of the multi-rank path through RCCL (mean over one rank = identity)
    from ideas_amd.ddp import force_collective
    forced = world == 1 and force_collective() and "MASTER_ADDR" in os.environ
    if world > 1 or forced:
        backend = os.environ.get("IDEAS_DIST_BACKEND", "nccl")      # "nccl" is RCCL on ROCm
        if backend == "nccl":
            # NO device_id: binding the group to the device makes ProcessGroupNCCL create its communicator eagerly, and on this stack
            # (torch 2.10 + RCCL 2.26) every later iteration is then 13.5 ms slower WITHOUT a single collective being issued (f32 393.3 ->
            # 406.9 ms, same box; tools/probes/pg_init_overhead.py, profiles/r06_pg_init_overhead.txt); the lazily created communicator
            # (first collective on the current device, which set_device fixed above) costs nothing (392.9 ms)
            dist.init_process_group(backend="nccl", init_method="env://")
        else:
            dist.init_process_group(backend=backend, init_method="env://")

    from ideas_amd import _lib
    _lib.load()

    bf16 = a.precision == "bf16"
    if a.roofline == "only":
        print(json.dumps(rooflines(device, a.batch, a.roofline_launches, bf16)))
        return

    res = run_steps(a, a.precision, a.steps, a.warmup, device, world, rank)
    if rank != 0:
        if dist.is_initialized():
            dist.destroy_process_group()
        return
    out = step_line(a, a.precision, res, world)
    errors = {}

    def leg(name, fn):
        """An auxiliary measurement must never cost the line its headline: failures are reported under `probe_errors`."""
        try:
            return fn()
        except Exception as e:
            errors[name] = repr(e)[:300]
            return None
    if res.get("roofline_weighted"):
        out["roofline_weighted"] = res["roofline_weighted"]
    if res.get("r1_reweighted"):
        out["r1_reweighted"] = res["r1_reweighted"]
    out["vs_rocm_eager"] = leg("vs_rocm_eager", lambda: vs_rocm_eager(out["value"], a, world))
    if a.roofline == "on":
        out.update(leg("rooflines", lambda: rooflines(device, a.batch, a.roofline_launches, bf16, errors)) or {})
    r2 = None
    if a.also_bf16 == "on" and not bf16 and world == 1 and not dist.is_initialized():
        # BASELINE.json configs[4]'s single-GPU part, timed by the same process right after the f32 window: same step, same
        # synthetic batch, bf16 activations (ideas_amd/precision.py).  A short window (R1 falls on its last step or not at all is
        # stated in r1_steps_in_window); the full-length figure is `python bench.py --precision bf16`.
        def bf16_leg():
            r2_ = run_steps(a, "bf16", a.bf16_steps, a.bf16_warmup, device, world, rank)
            l2 = step_line(a, "bf16", r2_, world)
            out["bf16"] = {k: l2[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "dtype", "step_tflops", "step_frac_of_ceiling", "losses")}
            out["bf16"]["r1_steps_in_window"] = l2["config"]["r1_steps_in_window"]
            out["bf16"]["workload"] = l2["config"]["workload"]
            if r2_.get("roofline_weighted"):
                out["bf16"]["roofline_weighted"] = r2_["roofline_weighted"]
            if r2_.get("r1_reweighted"):
                out["bf16"]["r1_reweighted"] = r2_["r1_reweighted"]
            if a.roofline == "on":
                out["bf16"]["roofline"] = roofline_probe_bf16(device, a.batch, a.roofline_launches)
            return r2_
        r2 = leg("bf16", bf16_leg)
    if a.cpu_baseline == "auto" and world == 1:
        out["cpu_baseline"] = leg("cpu_baseline", cpu_baseline) or {"value": None, "unit": "images/sec", "cores": None, "kind": "port", "sample": "failed"}
        out["cpu_baseline_config1"] = leg("cpu_baseline_config1", cpu_baseline_config1)
        # the metric's second half ("secret-bit extraction acc"): GPU decisions against the CPU oracle on identical weights
        out["extraction"] = leg("extraction", lambda: finish_extraction(res.get("extraction"), not bf16))
        if r2 is not None and "bf16" in out:
            out["bf16"]["extraction"] = leg("extraction_bf16", lambda: finish_extraction(r2.get("extraction"), False))
    if errors:
        out["probe_errors"] = errors
    print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


def rooflines(device, batch, launches, bf16, errors=None):
    probe = roofline_probe_bf16 if bf16 else roofline_probe
    out = {"roofline": probe(device, batch, launches)}            # the required entry: its failure is the caller's to report

    def opt(name, fn):
        try:
            v = fn()
            if v is not None:
                out[name] = v
        except Exception as e:
            if errors is None:
                raise
            errors[name] = repr(e)[:300]
    if not bf16:
        opt("roofline_s2", lambda: roofline_probe_s2(device, batch, max(4, launches // 2)))
    opt("roofline_wgrad", lambda: roofline_probe_wgrad(device, batch, launches, bf16))
    opt("roofline_direct", lambda: roofline_probe_direct(device, batch, launches, bf16))
    opt("roofline_hbm", lambda: roofline_probe_hbm(device, batch, launches, bf16))
    opt("roofline_hbm_bias_act_bwd", lambda: roofline_probe_bias_act_bwd(device, batch, launches, bf16))
    return out


def roofline_weighted(run_iteration, bf16: bool):
    """the headline `roofline` object names ONE launch of ONE kernel; this one covers the step.  After the timed
    window one more iteration is run with every libideas_hip.so launch timed in isolation (device synchronised around each call,
    tools/step_census2.py) and grouped by family = entry point + kernel size / stride.  `frac` = the MFMA families' total ALGORITHMIC
    FLOPs / their total isolated time / `peak` (the time-weighted mean of the per-family fractions); `families` lists each family's
    share of the isolated step and its own fraction; `isolated_ms` is what the iteration costs with nothing overlapped."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import step_census2 as SC
    peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_BF16_MFMA_TFLOPS / 6.0
    stats = SC.census(run_iteration)
    rows, tot, ct, cf = SC.families(stats, peak)
    return {"bound": "mfma", "achieved": round(cf / ct / 1e9, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(cf / ct / 1e9 / peak, 4),
            "what": "all MFMA convolution launches of one iteration (no lazy-R1 branch), each timed in isolation: total algorithmic FLOPs / "
                    "total time; `families`: isolated ms, share of the isolated iteration, TFLOP/s and fraction per family (HBM-bound "
                    "families have no fraction here: see roofline_hbm*)",
            "isolated_ms": round(tot, 1), "mfma_ms": round(ct, 1), "launches": sum(v[0] for v in stats.values()),
            "families": [r for r in rows if r["ms"] >= 0.5]}


def _barrier(device):
    """dist.barrier() on THIS rank's device (the group is not bound to one: see the init_process_group call)."""
    if dist.get_backend() == "nccl":
        dist.barrier(device_ids=[device.index])
    else:
        dist.barrier()


def run_steps(a, precision_name, steps, warmup, device, world, rank):
    """Build the trainer in `precision_name`, run `warmup` untimed and exactly `steps` timed iterations between
    barrier + synchronize pairs; returns the max-over-ranks wall time and what the line needs.  Frees everything on return."""
    from ideas_amd import precision, train_step as TS
    from ideas_amd.models import init_model
    from ideas_amd.ddp import GradReducer
    precision.set_activation_dtype(precision_name)
    # below 256x256 the reference's co-occurrence discriminator cannot run (models.py:400): the
    # step is then the Dco-less sub-step, as in the parity fixtures (tests/golden/step_r128.npz)
    args = TS.default_args(image_size=a.image_size, batch_size=a.batch, N=a.N, use_dco=a.image_size >= 256,
                           channel=a.channel, texture_channel=a.texture_channel,
                           elide_second_backward=not a.literal_second_backward, num_iters=10 ** 9,
                           share_forward=not a.no_share_forward)
    torch.manual_seed(0)              # identical replicas on every rank (no broadcast needed)
    trainer = TS.build_trainer(args, "cpu", init_model)
    for v in trainer.values():
        if isinstance(v, torch.nn.Module):
            v.to(device)
    if not a.torch_adam:
        from ideas_amd.optim import fuse_optimizers
        fuse_optimizers(trainer, args)      # one fused Adam(beta1=0)+EMA launch per group on flat buffers
    random.seed(1000 + rank)
    torch.manual_seed(1000 + rank)
    gx = torch.Generator().manual_seed(1234 + rank)
    X = (torch.rand(a.batch, 3, a.image_size, a.image_size, generator=gx) * 2 - 1).to(device)
    X = X.contiguous(memory_format=torch.channels_last)
    dist_on = dist.is_initialized()           # world > 1, or the one-rank RCCL exercise of IDEAS_DDP_FORCE_COLLECTIVE=1
    reducer = GradReducer() if dist_on else None

    def step(idx):
        return TS.train_iteration(trainer, args, X, idx, reducer=reducer)

    for j in range(warmup):
        step(args.d_reg_every * 1000 + j)          # first warm-up iteration exercises the R1 branch
    if dist_on:
        _barrier(device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(1, steps + 1):
        losses = step(i)
    torch.cuda.synchronize()
    t_own = time.perf_counter() - t0               # this rank's own K steps (before it waits for the slowest rank)
    if dist_on:
        _barrier(device)
    dt = time.perf_counter() - t0
    ranks_seen = None
    if dist_on:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # who took part: every rank's own time and the physical device it ran on (PCI bus id where torch exposes it -- a launcher that
        # hands each rank ONE visible device makes every index 0 --, else the device index).  Two ranks on one device would make the
        # "N GPUs" of the line untrue: fail loudly (IDEAS_BENCH_SHARE_GPU=1, the 1-GPU test of this code path, is the only exception).
        props = torch.cuda.get_device_properties(device)
        ident = [float(getattr(props, k, -1)) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")]
        info = torch.tensor([float(rank), t_own, float(device.index)] + ident, device=device, dtype=torch.float64)
        got = [torch.empty_like(info) for _ in range(dist.get_world_size())]
        dist.all_gather(got, info)
        rows = sorted([g_.tolist() for g_ in got])
        ranks_seen = [{"rank": int(r[0]), "images_per_sec": round(a.batch * steps / r[1], 2), "ms_per_step": round(r[1] / steps * 1e3, 2),
                       "device_index": int(r[2]), "pci": "%04x:%02x:%02x" % tuple(int(v) for v in r[3:6]) if min(r[3:6]) >= 0 else None}
                      for r in rows]
        keys = [(r["pci"] if r["pci"] is not None else r["device_index"]) for r in ranks_seen]
        if len(set(keys)) != len(keys) and os.environ.get("IDEAS_BENCH_SHARE_GPU") != "1":
            raise SystemExit(f"bench.py: {len(keys)} ranks on {len(set(keys))} distinct device(s) {keys}: every rank needs its own GPU "
                             "(LOCAL_RANK -> device index; check the launcher's device visibility)")
    # lazy-R1 weighting of the window: K timed steps hold floor(K / 16) R1 iterations, not K / 16 (20 steps: 1 in 20).  The extra cost of
    # an R1 iteration is measured right here (two R1 and two plain iterations, each between synchronisations) so that the line can
    # state the rate at exactly one R1 iteration in sixteen next to the measured one.
    r1w = None
    if not dist_on and getattr(a, "roofline", "off") != "off":          # (--roofline off: A/B and profiling runs time their window and nothing else)
        def timed(idx):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            step(idx)
            torch.cuda.synchronize()
            return time.perf_counter() - t1
        base = (steps // args.d_reg_every + 2) * args.d_reg_every
        t_r1 = min(timed(base), timed(base + args.d_reg_every))
        t_pl = min(timed(base + 1), timed(base + 2))
        n_r1 = sum(1 for i in range(1, steps + 1) if i % args.d_reg_every == 0)
        extra = max(0.0, t_r1 - t_pl)
        ms_w = ((dt - n_r1 * extra) / steps + extra / args.d_reg_every) * 1e3
        r1w = {"r1_iteration_extra_ms": round(extra * 1e3, 2), "r1_steps_in_window": n_r1, "steps": steps,
               "ms_per_step_at_one_r1_in_%d" % args.d_reg_every: round(ms_w, 2),
               "images_per_sec_at_one_r1_in_%d" % args.d_reg_every: round(a.batch / (ms_w * 1e-3), 3),
               "note": "`value` is the measured window as it is; this re-weights its R1 share to exactly 1 / d_reg_every (the FLOP model's)"}
    extraction = None
    if world == 1 and not dist_on and getattr(a, "cpu_baseline", "skip") == "auto":
        try:
            extraction = extraction_probe(trainer, args, X, device)
        except Exception as e:                                   # never lose the line to the accuracy probe
            extraction = {"error": repr(e)[:300]}
    weighted = None
    if getattr(a, "roofline", "off") == "on" and world == 1 and a.image_size == 256 and (a.channel, a.texture_channel) == (32, 2048):
        idx = steps + 3
        idx += 1 if idx % args.d_reg_every == 0 else 0                                   # (an iteration without the lazy-R1 branch)
        try:
            weighted = roofline_weighted(lambda: step(idx), precision_name == "bf16")
        except Exception as e:
            weighted = {"error": repr(e)[:300]}
    res = {"dt": dt, "steps": steps, "warmup": warmup, "roofline_weighted": weighted, "r1_reweighted": r1w, "extraction": extraction,
           "n_r1": sum(1 for i in range(1, steps + 1) if i % args.d_reg_every == 0),
           "losses": {k: round(float(v.detach()), 4) for k, v in losses.items() if v.numel() == 1},
           "ranks_seen": ranks_seen,
           "bucket_bytes": ({k: 4 * int(trainer[k].flat_g.numel()) for k in ("d_optim", "g_optim", "ex_optim") if hasattr(trainer[k], "flat_g")}
                            if dist_on else None)}
    del trainer, reducer, losses, X
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    precision.set_activation_dtype("f32")
    return res


def step_line(a, precision_name, res, world):
    from ideas_amd import _lib
    from ideas_amd.op import conv as _CV
    bf16 = precision_name == "bf16"
    dt, steps = res["dt"], res["steps"]
    if bf16:
        conv_math = ("bf16 activations in HBM, bf16 MFMA (one product per MFMA) with f32 accumulation and f32 epilogues; f32 master "
                     "weights, gradients and optimiser state (ideas_amd/precision.py, csrc/conv_bf16.hip)")
        ceiling, ceiling_note = PEAK_BF16_MFMA_TFLOPS, "dense bf16 MFMA peak"
    elif _CV.MATH == _lib.F32_B3:
        conv_math = ("f32 tensors; MFMA convs contract an exact 3-way bf16 split of both operands (6 bf16 MFMA products per "
                     "f32 product, f32 accumulate; f32 error class, tests/test_ops_gpu.py)")
        ceiling, ceiling_note = PEAK_BF16_MFMA_TFLOPS / 6.0, "dense bf16 MFMA peak / 6 plane products per f32 product"
    else:
        conv_math = "f32 MFMA (v_mfma_f32_32x32x2_f32)" + (" + 1-D Winograd F(2,3)" if _CV.WINOGRAD else "")
        ceiling, ceiling_note = PEAK_F32_MFMA_TFLOPS, "f32 MFMA peak"
    ips = world * a.batch * steps / dt
    gflop_img = flop_per_image(not a.literal_second_backward, shared=not a.no_share_forward)
    full = a.image_size == 256 and (a.channel, a.texture_channel) == (32, 2048)      # the configuration the FLOP model describes
    tfl = ips / world * gflop_img / 1e3
    return {
        "metric": "train images/sec at %dx%d (G+D+Ex step)" % (a.image_size, a.image_size), "value": round(ips, 3), "unit": "images/sec",
        "n_gpus": (dist.get_world_size() if dist.is_initialized() else 1), "steps": steps, "warmup": res["warmup"], "ms_per_step": round(dt / steps * 1e3, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": precision_name, "data": "synthetic",
        "config": {"workload": "IDEAS N=%d sigma=1 %dx%d batch=%d/GPU full G+D+Ex iteration (lazy R1 every 16, EMA), "
                               "%s nets, HIP kernels (BASELINE.json configs[%d]%s)"
                               % (a.N, a.image_size, a.image_size, a.batch,
                                  "full-width" if (a.channel, a.texture_channel) == (32, 2048) else
                                  "NARROW (channel %d, texture %d: a test configuration, not the benchmark)" % (a.channel, a.texture_channel),
                                  1 if a.image_size == 128 else (4 if bf16 else (3 if a.N == 2 else 2)),
                                  "" if a.image_size >= 256 else "; Dco-less sub-step: the reference's Dco cannot run below 256x256"),
                   "global_batch": world * a.batch, "parallelism": "dp%d" % world, "r1_steps_in_window": res["n_r1"],
                   "ranks": (dist.get_world_size() if dist.is_initialized() else 1),
                   "dist_backend": (dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else "")) if dist.is_initialized() else None,
                   "collectives_forced_at_one_rank": (True if (dist.is_initialized() and world == 1) else None),
                   "ranks_seen": res.get("ranks_seen"),
                   "grad_exchange": ("one mean all-reduce per optimiser group on its flat gradient buffer, 1/world folded into the collective; "
                                     "D group overlapped with the G-phase generator forwards, Ex group with the G-side backward") if dist.is_initialized() else None,
                   "allreduce_bytes_per_iteration": res["bucket_bytes"],
                   "second_backward": "literal" if a.literal_second_backward else "elided (Ex grad over Ex sub-graph)",
                   "shared_forward": "E(X), G(S1,T1) evaluated once per iteration" if not a.no_share_forward else "off",
                   "conv_arithmetic": conv_math},
        # (the FLOP model is the 256x256 full-width one of SURVEY.md §8(d); other configurations report throughput only)
        "step_gflop_per_image": round(gflop_img, 1) if full else None,
        "step_tflops": round(tfl, 2) if full else None,   # algorithmic FLOPs of the step / time, per GPU
        # the whole step (HBM-bound passes included) over the matrix ceiling of the arithmetic its convolutions actually run
        "step_frac_of_ceiling": round(tfl / ceiling, 4) if full else None,
        "step_ceiling_tflops": round(ceiling, 1), "step_ceiling_note": ceiling_note,
        # for orientation only: the f32 MFMA instruction (157.3 TFLOP/s) is what the reference's arithmetic would use on this chip
        "step_vs_f32_mfma_peak": round(tfl / PEAK_F32_MFMA_TFLOPS, 4) if full else None,
        "losses": res["losses"],
    }


if __name__ == "__main__":
    main()
