"""Kernel-level parity on a real MI355X: every C-ABI entry point against plain torch fp32 on CPU.

fp32 mode (exact-fp32 MFMA) must agree to ~1e-4 of the tensor scale; bf16 mode to 2e-2 (operand
rounding; SURVEY.md §8d tolerances)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from shot_vae_amd import _lib as L          # noqa: E402
from shot_vae_amd import geometry as G      # noqa: E402

DT = {"f32": (L.SV_F32, torch.float32, 2e-4), "bf16": (L.SV_BF16, torch.bfloat16, 2.5e-2)}
ACC = torch.float64      # sv_acc_t: the BatchNorm statistics / backward-sum accumulators are doubles (ABI 6)


def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_KEEP = []


def p(t):
    """device pointer of t; keeps t alive (an inline temporary would otherwise be freed -- and its
    block reused by the next temporary -- before the asynchronous kernel runs)."""
    if t is None:
        return None
    _KEEP.append(t)
    if len(_KEEP) > 4096:
        torch.cuda.synchronize()
        del _KEEP[:2048]
    return C.c_void_p(t.data_ptr())


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-20))


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def repack(master, g, transpose, dt):
    code, tdt, _ = DT[dt]
    N, T, Cc = master.shape
    dst = torch.zeros(max(G.packed_size(g), 1), dtype=tdt, device=dev())
    m = master.to(dev()).contiguous()
    L.call("sv_repack", code, p(m), N, T, Cc, int(transpose), C.byref(g), p(dst), st())
    return dst


def run_igemm(g, dt, x, w, pro=None, bias=None, residual=None, stats=False, ex=None):
    code, tdt, _ = DT[dt]
    d = dev()
    xd = x.to(d, tdt).contiguous()
    out = torch.full((g.B, g.Hout, g.Wout, g.ldo), 7.0, dtype=tdt, device=d)
    a = L.SvIgemmArgs()
    keep = [xd, out, w]
    a.x, a.w, a.out = xd.data_ptr(), w.data_ptr(), out.data_ptr()
    if pro is not None:
        sc, sh = pro[0].to(d).float().contiguous(), pro[1].to(d).float().contiguous()
        keep += [sc, sh]
        a.pro_scale, a.pro_shift, a.pro_slope = sc.data_ptr(), sh.data_ptr(), pro[2]
    if bias is not None:
        bd = bias.to(d).float().contiguous()
        keep.append(bd)
        a.bias = bd.data_ptr()
    if residual is not None:
        rd = residual.to(d, tdt).contiguous()
        keep.append(rd)
        a.residual = rd.data_ptr()
    sums = None
    R = 4      # replicated accumulators: block b adds to copy b % R
    a.replicas = R
    if stats:
        sums = torch.zeros(R, 2 * g.N, device=d, dtype=ACC)
        a.stats = sums.data_ptr()
    if ex is not None:
        exd = ex["x"].to(d, tdt).contiguous()
        vec = [ex[k].to(d).float().contiguous() for k in ("scale", "shift", "mean", "rstd")]
        keep += [exd] + vec
        sums = torch.zeros(R, 2 * g.N, device=d, dtype=ACC)
        a.ex, a.ex_scale, a.ex_shift, a.ex_mean, a.ex_rstd = [t.data_ptr() for t in [exd] + vec]
        a.ex_slope, a.bsums = ex["slope"], sums.data_ptr()
    L.call("sv_igemm", C.byref(g), code, C.byref(a), st())
    torch.cuda.synchronize()
    return out.float().cpu(), (None if sums is None else sums.sum(0).float().cpu())


def bq(t, dt):
    """round to the storage dtype (so the CPU reference sees the same operands)"""
    return t.to(DT[dt][1]).float()


CONV_CASES = [
    # B, Cin, N, H, k, stride, pad
    (4, 16, 32, 8, 3, 1, 1),
    (8, 32, 32, 16, 3, 1, 1),
    (3, 32, 64, 16, 3, 2, 1),
    (5, 64, 128, 8, 3, 2, 1),
    (4, 16, 32, 8, 1, 1, 0),
    (4, 32, 64, 16, 1, 2, 0),
    (2, 160, 160, 8, 3, 1, 1),
    (6, 128, 128, 8, 3, 1, 1),
    (2, 16, 16, 32, 3, 1, 1),
    # shapes that take the LDS-halo 3x3 kernel (conv3x3.hip) in its different tilings
    (4, 32, 32, 32, 3, 1, 1),
    (4, 64, 64, 16, 3, 1, 1),
    (4, 64, 128, 8, 3, 1, 1),
    (2, 32, 96, 16, 3, 1, 1),
    (1, 320, 320, 16, 3, 1, 1),
]


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_forward_fused(dt, case):
    B, Cin, N, H, k, stride, pad = case
    tol = DT[dt][2]
    torch.manual_seed(1)
    x = bq(torch.randn(B, Cin, H, H), dt)
    w = bq(torch.randn(N, Cin, k, k) / (Cin * k * k) ** 0.5, dt)
    scale, shift = torch.rand(Cin) + 0.5, torch.randn(Cin) * 0.3
    bias = torch.randn(N) * 0.2
    a = bq(F.leaky_relu(x * scale[None, :, None, None] + shift[None, :, None, None], 0.01), dt)
    Ho = (H + 2 * pad - k) // stride + 1
    res = bq(torch.randn(B, N, Ho, Ho), dt)
    y = F.conv2d(a, w, bias, stride, pad) + res
    g = G.conv_like(B, H, H, Cin, N, k, stride, pad)
    master = w.permute(0, 2, 3, 1).reshape(N, k * k, Cin).contiguous()
    wp = repack(master, g, False, dt)
    out, sums = run_igemm(g, dt, nhwc(x), wp, pro=(scale, shift, 0.01), bias=bias, residual=nhwc(res), stats=True)
    assert rel(nchw(out), y) < tol, ("out", rel(nchw(out), y))
    s1, s2 = y.sum((0, 2, 3)), (y * y).sum((0, 2, 3))
    assert rel(sums[:N], s1) < max(tol, 1e-3) * 3 and rel(sums[N:], s2) < max(tol, 1e-3) * 3


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_dgrad_with_activation_backward(dt, case):
    B, Cin, N, H, k, stride, pad = case
    tol = DT[dt][2]
    torch.manual_seed(2)
    w = bq(torch.randn(N, Cin, k, k) / (Cin * k * k) ** 0.5, dt)
    Ho = (H + 2 * pad - k) // stride + 1
    dy = bq(torch.randn(B, N, Ho, Ho), dt)
    xraw = bq(torch.randn(B, Cin, H, H), dt)            # the raw tensor whose BN+act fed this conv
    scale, shift = torch.rand(Cin) + 0.5, torch.randn(Cin) * 0.3
    mean, rstd = torch.randn(Cin) * 0.1, torch.rand(Cin) + 0.5
    da = F.conv_transpose2d(dy, w, None, stride, pad, output_padding=H - ((Ho - 1) * stride - 2 * pad + k))
    u = xraw * scale[None, :, None, None] + shift[None, :, None, None]
    gref = da * torch.where(u > 0, torch.ones_like(u), torch.full_like(u, 0.01))
    xh = (xraw - mean[None, :, None, None]) * rstd[None, :, None, None]
    g = G.convT_like(B, Ho, Ho, N, Cin, k, stride, pad)
    master = w.permute(0, 2, 3, 1).reshape(N, k * k, Cin).contiguous()
    wp = repack(master, g, True, dt)
    out, sums = run_igemm(g, dt, nhwc(dy), wp,
                          ex=dict(x=nhwc(xraw), scale=scale, shift=shift, mean=mean, rstd=rstd, slope=0.01))
    assert rel(nchw(out), gref) < tol, ("g", rel(nchw(out), gref))
    assert rel(sums[:Cin], gref.sum((0, 2, 3))) < max(tol, 1e-3) * 3
    assert rel(sums[Cin:], (gref * xh).sum((0, 2, 3))) < max(tol, 1e-3) * 3


@pytest.mark.parametrize("B,pro,stats,Gn,budget", [(1, True, True, 1, 0), (3, True, True, 1, 0), (70, True, True, 1, 0), (37, False, False, 1, 0),
                                                  (300, True, True, 1, 0), (65, True, True, 4, 0), (64, True, False, 2, 0), (96, True, True, 1, 8)])
@pytest.mark.parametrize("Cc,N,H", [(64, 128, 16), (32, 64, 32)])
def test_register_resident_stride2_forward(B, pro, stats, Gn, budget, Cc, N, H):
    """sconv.hip (the stride-2 3x3 forward convolutions of WideResNet blocks 2 / 3, wideresnet.py:29-30: register-resident weights,
    parity-split LDS image, bands of 8 output rows) against torch fp32 on the same bf16 operands -- one band per block, several with
    an odd count, the second band of an image (its top row is data, not padding), with / without the BatchNorm + LeakyReLU prologue
    and the statistics, batched groups, a small block budget -- and against the kernels it replaces."""
    torch.manual_seed(B)
    d = dev()
    Ho = H // 2
    x = bq(torch.randn(Gn * B, Cc, H, H), "bf16")
    w = bq(torch.randn(N, Cc, 3, 3) / (Cc * 9) ** 0.5, "bf16")
    scale, shift = torch.rand(Gn, Cc) + 0.5, torch.randn(Gn, Cc) * 0.3
    master = w.permute(0, 2, 3, 1).reshape(N, 9, Cc).contiguous()
    g = G.conv_like(B, H, H, Cc, N, 3, 2, 1)
    wp = repack(master, g, False, "bf16")
    xd = nhwc(x).to(d, torch.bfloat16).contiguous()
    scd, shd = scale.to(d).contiguous(), shift.to(d).contiguous()
    R = 4

    def run(disable):
        out = torch.full((Gn * B, Ho, Ho, N), 7.0, dtype=torch.bfloat16, device=d)
        sums = torch.zeros(Gn, R, 2 * N, device=d, dtype=ACC)
        a = L.SvIgemmArgs()
        a.x, a.w, a.out, a.replicas, a.groups, a.block_budget = xd.data_ptr(), wp.data_ptr(), out.data_ptr(), R, Gn, budget
        if pro:
            a.pro_scale, a.pro_shift, a.pro_slope = scd.data_ptr(), shd.data_ptr(), 0.01
        if stats:
            a.stats = sums.data_ptr()
        with L.options(disable=disable):
            L.call("sv_igemm", C.byref(g), L.SV_BF16, C.byref(a), st())
        torch.cuda.synchronize()
        return out.float().cpu(), sums.sum(1).float().cpu()

    out, sums = run(0)
    ref_out, ref_sums = run(L.K_SCONV)
    for gi in range(Gn):
        xs = x[gi * B:(gi + 1) * B]
        act = bq(F.leaky_relu(xs * scale[gi][None, :, None, None] + shift[gi][None, :, None, None], 0.01), "bf16") if pro else xs
        y = F.conv2d(act, w, None, 2, 1)
        o = nchw(out[gi * B:(gi + 1) * B])
        assert rel(o, y) < 4e-3, (gi, rel(o, y))
        assert (o - bq(y, "bf16")).abs().max() <= 2.0 ** -6 * y.abs().max()
        if stats:
            assert rel(sums[gi, :N], y.sum((0, 2, 3))) < 3e-3
            assert rel(sums[gi, N:], (y * y).sum((0, 2, 3))) < 3e-3
            assert rel(sums[gi], ref_sums[gi]) < 1e-3
    assert rel(out, ref_out) < 6e-3


@pytest.mark.parametrize("B", [1, 3, 70, 300])
def test_thin_layers_dgrad(B):
    """thconv.hip, activation-backward form: the data gradient of the 16 -> 32 convolution of block 1 (32 -> 16 at 32x32) against
    torch fp32, and against the LDS-halo kernel (same test with the kernel switched off)."""
    case = (B, 16, 32, 32, 3, 1, 1)
    test_conv_dgrad_with_activation_backward("bf16", case)
    with L.options(disable=L.K_THCONV):
        test_conv_dgrad_with_activation_backward("bf16", case)


@pytest.mark.parametrize("Cc,N,H,stride", [(16, 32, 32, 1), (32, 64, 32, 2), (16, 160, 32, 1)])
@pytest.mark.parametrize("B,pro,Gn", [(1, True, 1), (3, False, 1), (37, True, 3), (130, True, 4)])
def test_pointwise_shortcut_forward(B, pro, Gn, Cc, N, H, stride):
    """pconv.hip (the 1x1 shortcut convolutions of the WideResNet, wideresnet.py:41-43: B fragments straight from global memory,
    BatchNorm + LeakyReLU in registers, no LDS in the loop) against torch fp32 on the same bf16 operands -- groups with their own
    coefficients, with / without the prologue -- and against the gather-GEMM it replaces."""
    torch.manual_seed(B + N)
    d = dev()
    Ho = H // stride
    x = bq(torch.randn(Gn * B, Cc, H, H), "bf16")
    w = bq(torch.randn(N, Cc, 1, 1) / Cc ** 0.5, "bf16")
    scale, shift = torch.rand(Gn, Cc) + 0.5, torch.randn(Gn, Cc) * 0.3
    g = G.conv_like(B, H, H, Cc, N, 1, stride, 0)
    wp = repack(w.reshape(N, 1, Cc).contiguous(), g, False, "bf16")
    xd = nhwc(x).to(d, torch.bfloat16).contiguous()
    scd, shd = scale.to(d).contiguous(), shift.to(d).contiguous()

    def run(disable):
        out = torch.full((Gn * B, Ho, Ho, N), 7.0, dtype=torch.bfloat16, device=d)
        a = L.SvIgemmArgs()
        a.x, a.w, a.out, a.replicas, a.groups = xd.data_ptr(), wp.data_ptr(), out.data_ptr(), 1, Gn
        if pro:
            a.pro_scale, a.pro_shift, a.pro_slope = scd.data_ptr(), shd.data_ptr(), 0.01
        with L.options(disable=disable):
            L.call("sv_igemm", C.byref(g), L.SV_BF16, C.byref(a), st())
        torch.cuda.synchronize()
        return out.float().cpu()

    out, ref_out = run(0), run(L.K_PCONV)
    for gi in range(Gn):
        xs = x[gi * B:(gi + 1) * B]
        act = bq(F.leaky_relu(xs * scale[gi][None, :, None, None] + shift[gi][None, :, None, None], 0.01), "bf16") if pro else xs
        y = F.conv2d(act, w, None, stride, 0)
        o = nchw(out[gi * B:(gi + 1) * B])
        assert rel(o, y) < 4e-3, (gi, rel(o, y))
        assert (o - bq(y, "bf16")).abs().max() <= 2.0 ** -6 * y.abs().max()
    assert rel(out, ref_out) < 6e-3


@pytest.mark.parametrize("B", [1, 3, 70, 300])
@pytest.mark.parametrize("Cc,N,H", [(64, 128, 16), (32, 64, 32)])
def test_register_resident_stride2_dgrad(B, Cc, N, H):
    """tconv.hip, EX form: the data gradients of the two stride-2 3x3 convolutions of the WideResNet (64 -> 128 at 16x16, 32 -> 64 at
    32x32; wideresnet.py:29-30: phases of 1 / 2 / 2 / 4 taps, register-resident weights) with the activation-backward epilogue,
    against torch fp32 -- and against the LDS-halo kernels they replace (same test with the kernel switched off)."""
    case = (B, Cc, N, H, 3, 2, 1)
    test_conv_dgrad_with_activation_backward("bf16", case)
    with L.options(disable=L.K_TCONVR):
        test_conv_dgrad_with_activation_backward("bf16", case)
    # the two kernels against each other, groups with their own constants and accumulators, a small block budget
    torch.manual_seed(B)
    d = dev()
    Gn, Ho = 2, H // 2
    g = G.convT_like(B, Ho, Ho, N, Cc, 3, 2, 1)
    w = torch.randn(G.packed_size(g), device=d).bfloat16() * 0.05
    dy = torch.randn(Gn * B, Ho, Ho, N, device=d).bfloat16()
    xraw = torch.randn(Gn * B, H, H, Cc, device=d).bfloat16()
    vec = [(torch.rand(Gn, Cc, device=d) + 0.5), torch.randn(Gn, Cc, device=d) * 0.3, torch.randn(Gn, Cc, device=d) * 0.1, torch.rand(Gn, Cc, device=d) + 0.5]
    R = 4

    def run(disable, budget):
        out = torch.full((Gn * B, H, H, Cc), 7.0, dtype=torch.bfloat16, device=d)
        sums = torch.zeros(Gn, R, 2 * Cc, device=d, dtype=ACC)
        a = L.SvIgemmArgs()
        a.x, a.w, a.out, a.replicas, a.groups, a.block_budget = dy.data_ptr(), w.data_ptr(), out.data_ptr(), R, Gn, budget
        a.ex, a.ex_scale, a.ex_shift, a.ex_mean, a.ex_rstd = [t.data_ptr() for t in [xraw] + vec]
        a.ex_slope, a.bsums = 0.0, sums.data_ptr()
        with L.options(disable=disable):
            L.call("sv_igemm", C.byref(g), L.SV_BF16, C.byref(a), st())
        torch.cuda.synchronize()
        return out.float().cpu(), sums.sum(1).float().cpu()

    ref_out, ref_sums = run(L.K_TCONVR, 0)
    for budget in (0, 16):
        out, sums = run(0, budget)
        assert rel(out, ref_out) < 6e-3
        assert (out - ref_out).abs().max() <= 2.0 ** -6 * ref_out.abs().max()
        assert rel(sums, ref_sums) < 2e-3


@pytest.mark.parametrize("case", [(32, 160, 160, 32, 3, 1, 1), (64, 160, 160, 16, 3, 1, 1)])
def test_conv3x3_wide_multitile(case):
    """Shapes large enough to take the multi-tile MFMA-bound kernel (conv3x3m): forward with every fusion and
    the data gradient with the activation-backward epilogue."""
    test_conv_forward_fused("bf16", case)
    test_conv_dgrad_with_activation_backward("bf16", case)


@pytest.mark.parametrize("case", [(4, 160, 160, 8, 3, 1, 1), (8, 128, 128, 8, 3, 1, 1), (1, 96, 128, 32, 3, 1, 1),
                                  (2, 128, 256, 16, 3, 1, 1), (3, 160, 320, 32, 3, 1, 1), (4, 640, 640, 8, 3, 1, 1),
                                  (2, 96, 192, 16, 3, 1, 1)])
def test_conv3x3_wide_mfma(case):
    """Shapes that take the 256-pixel x 160/128/64-channel LDS-DMA pipelined kernel (conv3x3w.hip): all channel-tile
    widths, the three image sizes (one / several images per tile), odd and even chunk counts, several channel tiles
    per pixel tile, a grid with idle tail blocks; forward with every fusion and the data gradient with the
    activation-backward epilogue (reversed tap order).  (The dispatcher only picks this kernel for grids of at least
    one block per CU; SV_OPT_WIDE_MIN_BLOCKS lowers that bound for these small parity shapes.)"""
    # (the 160-channel-tile shapes would otherwise take conv3x3x.hip)
    with L.options(wide_min_blocks=1, disable=L.K_CONV3X3X):
        test_conv_forward_fused("bf16", case)
        test_conv_dgrad_with_activation_backward("bf16", case)


@pytest.mark.parametrize("case", [(4, 160, 160, 8, 3, 1, 1), (3, 160, 320, 32, 3, 1, 1), (4, 640, 640, 8, 3, 1, 1),
                                  (2, 320, 160, 16, 3, 1, 1), (72, 160, 160, 32, 3, 1, 1), (70, 96, 320, 16, 3, 1, 1)])
def test_conv3x3_one_wave_per_simd(case):
    """The gap-scheduled, persistent one-block-per-CU variant of the wide kernel (conv3x3x.hip, the default for
    160-channel tiles): the three image sizes, odd / even chunk counts, one and several channel tiles, blocks with one,
    two and three items (the last two cases: 288 / 140 items on 256 blocks); forward with every fusion and the data gradient
    with the activation-backward epilogue."""
    with L.options(wide_min_blocks=1):
        test_conv_forward_fused("bf16", case)
        test_conv_dgrad_with_activation_backward("bf16", case)


@pytest.mark.parametrize("case", [(5, 160, 320, 32, 3, 2, 1), (3, 320, 640, 16, 3, 2, 1), (5, 160, 320, 32, 1, 2, 0),
                                  (3, 16, 160, 32, 3, 1, 1), (7, 16, 160, 32, 1, 1, 0), (2, 320, 160, 8, 3, 2, 1),
                                  (5, 64, 128, 16, 3, 2, 1), (3, 128, 256, 8, 1, 2, 0), (3, 128, 128, 16, 1, 2, 0)])
def test_generic_gemm_256_row_tiles(case):
    """The 256-row x 160 / 128-channel tiles of the generic gather-GEMM (igemm.hip, MS = 4): the stride-2 3x3 convolutions
    and 1x1 shortcuts of WRN-28-10 and its first layers; ragged row counts (B * Hq * Wq not a multiple of 256), k tails
    (Cin * taps not a multiple of 32), one / two / four channel tiles; forward with every fusion and the data gradient
    (four phases for stride 2, three of them without taps for the 1x1 shortcut: the zero-store path) with the
    activation-backward epilogue."""
    with L.options(wide_min_blocks=1, disable=L.K_HALO | L.K_HALOP):
        test_conv_forward_fused("bf16", case)
        test_conv_dgrad_with_activation_backward("bf16", case)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("H,Cin,N,B", [(1, 1024, 512, 6), (2, 512, 256, 4), (8, 128, 64, 3), (16, 64, 16, 2)])
def test_convT_forward_and_dgrad(dt, H, Cin, N, B):
    tol = DT[dt][2]
    torch.manual_seed(3)
    x = bq(torch.randn(B, Cin, H, H), dt)
    w = bq(torch.randn(Cin, N, 4, 4) / (Cin * 4) ** 0.5, dt)
    scale, shift = torch.rand(Cin) + 0.5, torch.randn(Cin) * 0.3
    a = bq(F.relu(x * scale[None, :, None, None] + shift[None, :, None, None]), dt)
    y = F.conv_transpose2d(a, w, None, 2, 1)
    master = w.permute(1, 2, 3, 0).reshape(N, 16, Cin).contiguous()
    g = G.convT_like(B, H, H, Cin, N, 4, 2, 1)
    out, sums = run_igemm(g, dt, nhwc(x), repack(master, g, False, dt), pro=(scale, shift, 0.0), stats=True)
    assert rel(nchw(out), y) < tol
    assert rel(sums[:N], y.sum((0, 2, 3))) < max(tol, 1e-3) * 3
    dy = bq(torch.randn_like(y), dt)
    dx = F.conv2d(dy, w, None, 2, 1)
    gd = G.conv_like(B, 2 * H, 2 * H, N, Cin, 4, 2, 1)
    out, _ = run_igemm(gd, dt, nhwc(dy), repack(master, gd, True, dt))
    assert rel(nchw(out), dx) < tol


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_gemm_ragged_batch(dt):
    """ConvTranspose2d(latent,1024,k=1) on a 1x1 map == GEMM; batch not a multiple of the tile."""
    tol = DT[dt][2]
    torch.manual_seed(4)
    B, Cin, N = 37, 144, 1024
    x = bq(torch.randn(B, Cin), dt)
    w = bq(torch.randn(N, Cin) / Cin ** 0.5, dt)
    g = G.conv_like(B, 1, 1, Cin, N, 1, 1, 0)
    out, sums = run_igemm(g, dt, x.view(B, 1, 1, Cin), repack(w.view(N, 1, Cin), g, False, dt), stats=True)
    y = x @ w.t()
    assert rel(out.view(B, N), y) < tol
    assert rel(sums[N:], (y * y).sum(0)) < max(tol, 1e-3) * 3


WG_CASES = CONV_CASES + [(40, 16, 16, 32, 3, 1, 1), (16, 32, 32, 32, 3, 1, 1), (32, 64, 64, 16, 3, 1, 1),
                         # wide layers (160 x 32 slabs, wgrad3x3w): several splits, two channel tiles, all image sizes
                         (16, 160, 160, 32, 3, 1, 1), (8, 96, 320, 16, 3, 1, 1), (32, 160, 160, 8, 3, 1, 1),
                         # the 32x32x16 narrow kernel (wgrad3x3m): 64-channel n tiles at every image size, 32-channel n tiles (N = 96)
                         (32, 128, 128, 8, 3, 1, 1), (8, 64, 96, 16, 3, 1, 1), (16, 32, 64, 8, 3, 1, 1), (12, 64, 64, 32, 3, 1, 1)]


@pytest.mark.parametrize("dt,use_tr", [("f32", 0), ("bf16", 0), ("bf16", 1)])
@pytest.mark.parametrize("case", WG_CASES)
def test_conv_wgrad(dt, use_tr, case):
    B, Cin, N, H, k, stride, pad = case
    code, tdt, tol = DT[dt]
    torch.manual_seed(5)
    x = bq(torch.randn(B, Cin, H, H), dt)
    scale, shift = torch.rand(Cin) + 0.5, torch.randn(Cin) * 0.3
    a = bq(F.leaky_relu(x * scale[None, :, None, None] + shift[None, :, None, None], 0.01), dt)
    Ho = (H + 2 * pad - k) // stride + 1
    dy = bq(torch.randn(B, N, Ho, Ho), dt)
    wref = torch.nn.grad.conv2d_weight(a, (N, Cin, k, k), dy, stride, pad)
    g = G.conv_like(B, H, H, Cin, N, k, stride, pad)
    d = dev()
    xd, dyd = nhwc(x).to(d, tdt), nhwc(dy).to(d, tdt)
    sc, sh = scale.to(d), shift.to(d)
    dw = torch.zeros(N, k * k, Cin, device=d)
    ws = torch.full((8 * 1024 * 1024,), float("nan"), device=d)      # workspace contents are irrelevant on entry
    for it in range(2):   # accumulates: two calls == 2x; once with and once without the slab workspace
        L.call("sv_wgrad", C.byref(g), code, p(xd), p(sc), p(sh), 0.01, p(dyd), p(dw), 0, use_tr,
               p(ws) if it else None, ws.numel() if it else 0, 1, st())
    torch.cuda.synchronize()
    got = dw.cpu().view(N, k, k, Cin).permute(0, 3, 1, 2) / 2
    assert rel(got, wref) < tol, rel(got, wref)


@pytest.mark.parametrize("N,pro", [(32, True), (16, False), (160, True)])
@pytest.mark.parametrize("B,Gn,budget", [(1, 1, 0), (3, 1, 0), (70, 1, 0), (33, 4, 0), (96, 2, 16)])
def test_thin_layers_wgrad(B, Gn, budget, N, pro):
    """thwgrad.hip (weight gradients of the thin 3x3 layers at 32x32 -- 16 -> 32 with the BatchNorm + LeakyReLU prologue, the stem
    16 -> 16 without -- : the whole gradient in every block, every band staged once) against torch fp32 on the same bf16 operands: all
    four bands of an image, batched groups with their own coefficients (the gradient of the shared weights sums over them), a small
    block budget, accumulation into a non-zero gradient -- and against the tap-fused LDS-halo kernel it replaces.  N = 160 (the same
    layer at width 10): the output channels split over five blocks of a band slot."""
    torch.manual_seed(B + N)
    d = dev()
    Cin, H = 16, 32
    x = bq(torch.randn(Gn * B, Cin, H, H), "bf16")
    dy = bq(torch.randn(Gn * B, N, H, H), "bf16")
    scale, shift = torch.rand(Gn, Cin) + 0.5, torch.randn(Gn, Cin) * 0.3
    wref = torch.zeros(N, Cin, 3, 3)
    for gi in range(Gn):
        xs = x[gi * B:(gi + 1) * B]
        a = bq(F.leaky_relu(xs * scale[gi][None, :, None, None] + shift[gi][None, :, None, None], 0.01), "bf16") if pro else xs
        wref += torch.nn.grad.conv2d_weight(a, (N, Cin, 3, 3), dy[gi * B:(gi + 1) * B], 1, 1)
    g = G.conv_like(B, H, H, Cin, N, 3, 1, 1)
    xd, dyd = nhwc(x).to(d, torch.bfloat16).contiguous(), nhwc(dy).to(d, torch.bfloat16).contiguous()
    sc, sh = scale.to(d).contiguous(), shift.to(d).contiguous()
    ws = torch.full((8 * 1024 * 1024,), float("nan"), device=d)

    def run(disable):
        dw = torch.full((N, 9, Cin), 0.5, device=d)
        a = L.SvWgradArgs()
        a.x, a.dy, a.dw = xd.data_ptr(), dyd.data_ptr(), dw.data_ptr()
        if pro:
            a.pro_scale, a.pro_shift, a.pro_slope = sc.data_ptr(), sh.data_ptr(), 0.01
        a.splits, a.use_tr, a.ws, a.ws_elems, a.groups, a.block_budget = 0, 1, ws.data_ptr(), ws.numel(), Gn, budget
        with L.options(disable=disable):
            L.call("sv_wgrad_ex", C.byref(g), L.SV_BF16, C.byref(a), st())
        torch.cuda.synchronize()
        return (dw.cpu() - 0.5).view(N, 3, 3, Cin).permute(0, 3, 1, 2)

    got, ref = run(0), run(L.K_THWGRAD)
    assert rel(got, wref) < 2e-3, rel(got, wref)
    assert rel(got, ref) < 2e-3, rel(got, ref)


@pytest.mark.parametrize("kind,pro", [("conv", False), ("conv", True), ("convT", True), ("convT", False)])
@pytest.mark.parametrize("B,Gn,budget", [(1, 1, 0), (3, 1, 0), (70, 1, 0), (33, 4, 0), (96, 2, 16), (600, 1, 0)])
def test_thin_4x4_stride2_wgrad(B, Gn, budget, kind, pro):
    """k4wgrad.hip (weight gradients of svhn_VAE's thin 4x4 stride-2 layers: Conv2d(3 (16 padded), 32, 4, 2, 1) at 32x32, svhn_vae.py:62,
    and ConvTranspose2d(32, 3 (16 padded), 4, 2, 1) at 16x16, svhn_vae.py:131; the whole gradient in every block, every image staged
    once) against torch fp32 on the same bf16 operands -- the load prologue on the layer's input, batched groups with their own
    coefficients, a small block budget, more images than blocks, accumulation into a non-zero gradient -- and against the generic
    gather kernel it replaces."""
    torch.manual_seed(B + len(kind))
    d = dev()
    if kind == "conv":
        Cin, N, H = 16, 32, 32
        g = G.conv_like(B, H, H, Cin, N, 4, 2, 1)
        Ho = H // 2
    else:
        Cin, N, H = 32, 16, 16
        g = G.convT_like(B, H, H, Cin, N, 4, 2, 1)
        Ho = 2 * H
    x = bq(torch.randn(Gn * B, Cin, H, H), "bf16")
    dy = bq(torch.randn(Gn * B, N, Ho, Ho), "bf16")
    scale, shift = torch.rand(Gn, Cin) + 0.5, torch.randn(Gn, Cin) * 0.3
    wref = torch.zeros(N, 16, Cin)                       # master layout [n][ky * 4 + kx][c]
    for gi in range(Gn):
        xs = x[gi * B:(gi + 1) * B]
        a = bq(F.leaky_relu(xs * scale[gi][None, :, None, None] + shift[gi][None, :, None, None], 0.01), "bf16") if pro else xs
        if kind == "conv":
            wref += torch.nn.grad.conv2d_weight(a, (N, Cin, 4, 4), dy[gi * B:(gi + 1) * B], 2, 1).permute(0, 2, 3, 1).reshape(N, 16, Cin)
        else:
            w0 = torch.zeros(Cin, N, 4, 4, requires_grad=True)
            (F.conv_transpose2d(a, w0, None, 2, 1) * dy[gi * B:(gi + 1) * B]).sum().backward()
            wref += w0.grad.permute(1, 2, 3, 0).reshape(N, 16, Cin)
    xd, dyd = nhwc(x).to(d, torch.bfloat16).contiguous(), nhwc(dy).to(d, torch.bfloat16).contiguous()
    sc, sh = scale.to(d).contiguous(), shift.to(d).contiguous()
    ws = torch.full((8 * 1024 * 1024,), float("nan"), device=d)

    def run(disable, with_ws=True):
        dw = torch.full((N, 16, Cin), 0.5, device=d)
        a = L.SvWgradArgs()
        a.x, a.dy, a.dw = xd.data_ptr(), dyd.data_ptr(), dw.data_ptr()
        if pro:
            a.pro_scale, a.pro_shift, a.pro_slope = sc.data_ptr(), sh.data_ptr(), 0.01
        a.splits, a.use_tr, a.groups, a.block_budget = 0, 1, Gn, budget
        if with_ws:                       # (per-block slabs + sv_slab_reduce; without: the blocks add to dw themselves)
            a.ws, a.ws_elems = ws.data_ptr(), ws.numel()
        with L.options(disable=disable):
            L.call("sv_wgrad_ex", C.byref(g), L.SV_BF16, C.byref(a), st())
        torch.cuda.synchronize()
        return dw.cpu() - 0.5

    got, got_atomic, ref = run(0), run(0, False), run(L.K_THWGRAD)
    assert rel(got, wref) < 2e-3, rel(got, wref)
    assert rel(got_atomic, wref) < 2e-3, rel(got_atomic, wref)
    assert rel(got, ref) < 2e-3, rel(got, ref)


@pytest.mark.parametrize("Cin,N,H", [(32, 64, 32), (64, 128, 16)])
@pytest.mark.parametrize("B,Gn,budget,pro", [(1, 1, 0, True), (3, 1, 0, False), (70, 1, 0, True), (33, 4, 0, True), (96, 2, 16, True), (512, 1, 256, True)])
def test_banded_stride2_wgrad(B, Gn, budget, pro, Cin, N, H):
    """s2wgrad.hip (weight gradients of the stride-2 3x3 convolutions, wideresnet.py:29-30: 32 -> 64 at 32x32 with the whole gradient in
    every block, 64 -> 128 at 16x16 with the output channels split over four blocks of an XCD; bands of 8 output rows staged once into
    a parity-split image) against torch fp32 on the same bf16 operands -- both bands of an image, odd counts, batched groups with
    their own prologue coefficients, a small block budget, accumulation into a non-zero gradient -- and against the generic kernel
    it replaces."""
    torch.manual_seed(B)
    d = dev()
    x = bq(torch.randn(Gn * B, Cin, H, H), "bf16")
    dy = bq(torch.randn(Gn * B, N, H // 2, H // 2), "bf16")
    scale, shift = torch.rand(Gn, Cin) + 0.5, torch.randn(Gn, Cin) * 0.3
    wref = torch.zeros(N, Cin, 3, 3)
    for gi in range(Gn):
        xs = x[gi * B:(gi + 1) * B]
        a = bq(F.leaky_relu(xs * scale[gi][None, :, None, None] + shift[gi][None, :, None, None], 0.01), "bf16") if pro else xs
        wref += torch.nn.grad.conv2d_weight(a, (N, Cin, 3, 3), dy[gi * B:(gi + 1) * B], 2, 1)
    g = G.conv_like(B, H, H, Cin, N, 3, 2, 1)
    xd, dyd = nhwc(x).to(d, torch.bfloat16).contiguous(), nhwc(dy).to(d, torch.bfloat16).contiguous()
    sc, sh = scale.to(d).contiguous(), shift.to(d).contiguous()
    ws = torch.full((8 * 1024 * 1024,), float("nan"), device=d)

    def run(disable):
        dw = torch.full((N, 9, Cin), 0.5, device=d)
        a = L.SvWgradArgs()
        a.x, a.dy, a.dw = xd.data_ptr(), dyd.data_ptr(), dw.data_ptr()
        if pro:
            a.pro_scale, a.pro_shift, a.pro_slope = sc.data_ptr(), sh.data_ptr(), 0.01
        a.splits, a.use_tr, a.ws, a.ws_elems, a.groups, a.block_budget = 0, 1, ws.data_ptr(), ws.numel(), Gn, budget
        with L.options(disable=disable):
            L.call("sv_wgrad_ex", C.byref(g), L.SV_BF16, C.byref(a), st())
        torch.cuda.synchronize()
        return (dw.cpu() - 0.5).view(N, 3, 3, Cin).permute(0, 3, 1, 2)

    got, ref = run(0), run(L.K_S2WGRAD)
    assert rel(got, wref) < 2e-3, rel(got, wref)
    assert rel(got, ref) < 2e-3, rel(got, ref)


@pytest.mark.parametrize("dt,use_tr", [("f32", 0), ("bf16", 1)])
@pytest.mark.parametrize("H,Cin,N,B", [(1, 1024, 512, 6), (4, 256, 128, 4), (16, 64, 16, 2)])
def test_convT_wgrad(dt, use_tr, H, Cin, N, B):
    code, tdt, tol = DT[dt]
    torch.manual_seed(6)
    x = bq(torch.randn(B, Cin, H, H), dt)
    dy = bq(torch.randn(B, N, 2 * H, 2 * H), dt)
    w = torch.zeros(Cin, N, 4, 4, requires_grad=True)
    F.conv_transpose2d(x, w, None, 2, 1).backward(dy)
    g = G.convT_like(B, H, H, Cin, N, 4, 2, 1)
    d = dev()
    dw = torch.zeros(N, 16, Cin, device=d)
    L.call("sv_wgrad", C.byref(g), code, p(nhwc(x).to(d, tdt)), None, None, 0.0, p(nhwc(dy).to(d, tdt)), p(dw), 0,
           use_tr, None, 0, 1, st())
    torch.cuda.synchronize()
    got = dw.cpu().view(N, 4, 4, Cin).permute(3, 0, 1, 2)
    assert rel(got, w.grad) < tol


# ------------------------------------------------------------------------------------------ small kernels
def test_bn_finalize_and_bwd_apply():
    torch.manual_seed(7)
    d = dev()
    B, Cc, H = 6, 32, 4
    x = torch.randn(B, Cc, H, H) * 1.5 + 0.3
    gamma, beta = torch.rand(Cc) + 0.5, torch.randn(Cc)
    rm, rv = torch.randn(Cc) * 0.1, torch.rand(Cc) + 0.5
    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm2, rv2 = rm.clone(), rv.clone()
    y = F.batch_norm(xr, rm2, rv2, gr, br, True, 0.1, 1e-5)
    gup = torch.randn_like(y)
    res = torch.randn_like(y)
    (y * gup).sum().backward()
    n = B * H * H
    xn = nhwc(x).to(d)
    stats = torch.cat([xn.reshape(-1, Cc).sum(0), (xn.reshape(-1, Cc) ** 2).sum(0)]).contiguous()
    stats = torch.stack([stats * 0.25, stats * 0.75]).to(ACC).contiguous()        # two replicas (sv_acc_t: doubles)
    outs = [torch.zeros(Cc, device=d) for _ in range(4)]
    rmd, rvd = rm.to(d), rv.to(d)
    L.call("sv_bn_finalize", p(stats), 2, Cc, float(n), p(gamma.to(d)), p(beta.to(d)), 1e-5, 0.1, p(rmd), p(rvd),
           p(outs[0]), p(outs[1]), p(outs[2]), p(outs[3]), 1, st())
    assert rel(rmd, rm2) < 1e-5 and rel(rvd, rv2) < 1e-5
    mean, var = x.mean((0, 2, 3)), x.var((0, 2, 3), unbiased=False)
    assert rel(outs[2], mean) < 1e-4 and rel(outs[3], torch.rsqrt(var + 1e-5)) < 1e-4
    yk = xn * outs[0] + outs[1]
    assert rel(yk, nhwc(y.detach())) < 1e-4
    # backward apply (g = upstream gradient w.r.t. y)
    gn = nhwc(gup).to(d)
    xh = (xn - outs[2]) * outs[3]
    bsums = torch.cat([gn.reshape(-1, Cc).sum(0), (gn * xh).reshape(-1, Cc).sum(0)]).contiguous()
    bsums = torch.stack([bsums * 0.5, bsums * 0.5]).to(ACC).contiguous()
    dgam, dbet = torch.zeros(Cc, device=d), torch.zeros(Cc, device=d)
    br_ = (L.SvBnBranch * 1)()
    gmd = gamma.to(d)
    br_[0].g, br_[0].bsums, br_[0].gamma = gn.data_ptr(), bsums.data_ptr(), gmd.data_ptr()
    br_[0].dgamma, br_[0].dbeta, br_[0].replicas = dgam.data_ptr(), dbet.data_ptr(), 2
    dx = torch.empty_like(xn)
    rn = nhwc(res).to(d)
    L.call("sv_bn_bwd_apply", L.SV_F32, n, Cc, Cc, p(xn), p(outs[2]), p(outs[3]), float(n), br_, 1, p(rn), p(dx), 1, st())
    torch.cuda.synchronize()
    assert rel(nchw(dx.cpu()), xr.grad + res) < 1e-4
    assert rel(dgam, gr.grad) < 1e-4 and rel(dbet, br.grad) < 1e-4


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_pool_head_sample(dt):
    code, tdt, tol = DT[dt]
    torch.manual_seed(8)
    d = dev()
    B, Cc, HW, ldc, K = 11, 64, 16, 128, 10
    x = bq(torch.randn(B, HW, Cc), dt)
    scale, shift = torch.rand(Cc) + 0.5, torch.randn(Cc) * 0.2
    mean, rstd = torch.randn(Cc) * 0.1, torch.rand(Cc) + 0.5
    W = torch.randn(2 * ldc + K, Cc) / 8
    bias = torch.randn(2 * ldc + K) * 0.1
    eps, u = torch.randn(B, ldc), torch.rand(B, K)
    xr = x.clone().requires_grad_(True)
    Wr, br = W.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    feat = F.leaky_relu(xr * scale + shift, 0.01).mean(1)
    o = F.linear(feat, Wr, br)
    mu, ls, la = o[:, :ldc], o[:, ldc:2 * ldc], F.log_softmax(o[:, 2 * ldc:], 1)
    z = mu + torch.exp(ls) * eps
    gum = -torch.log(-torch.log(u + 1e-12) + 1e-12)
    c = torch.softmax((la + gum) / 0.67, 1)
    lat = torch.cat([z, c], 1)
    dlat = bq(torch.randn(B, ldc + K), dt)
    dmu0, dls0, dla0 = torch.randn(B, ldc), torch.randn(B, ldc), torch.randn(B, K)
    ((lat * dlat).sum() + (mu * dmu0).sum() + (ls * dls0).sum() + (la * dla0).sum()).backward()
    # --- forward on device
    xd = x.to(d, tdt)
    featd = torch.empty(B, Cc, device=d)
    sc, sh, mn, rs = scale.to(d), shift.to(d), mean.to(d), rstd.to(d)
    L.call("sv_pool_fwd", code, p(xd), p(sc), p(sh), 0.01, B, HW, Cc, Cc, p(featd), 1, st())
    assert rel(featd, feat.detach()) < 1e-4
    mud, lsd, lad = torch.empty(B, ldc, device=d), torch.empty(B, ldc, device=d), torch.empty(B, K, device=d)
    Wd, bd = W.to(d), bias.to(d)
    L.call("sv_head_fwd", p(featd), B, Cc, p(Wd), p(bd), ldc, K, p(mud), p(lsd), p(lad), st())
    assert rel(mud, mu.detach()) < 1e-4 and rel(lsd, ls.detach()) < 1e-4 and rel(lad, la.detach()) < 1e-4
    Lpad = 144
    latd = torch.full((B, Lpad), 5.0, device=d, dtype=tdt)
    csoft = torch.empty(B, K, device=d)
    epsd, ud = eps.to(d), u.to(d)
    L.call("sv_sample_fwd", code, p(mud), p(lsd), p(lad), p(epsd), p(ud), None, None, 0.0, None, 0, 0.67, B, ldc, K, Lpad,
           p(latd), p(csoft), st())
    assert rel(latd[:, :ldc + K].float(), lat.detach()) < max(tol, 1e-4)
    assert float(latd[:, ldc + K:].float().abs().max()) == 0.0
    # --- backward on device
    dmud, dlsd, dlad = dmu0.to(d).clone(), dls0.to(d).clone(), dla0.to(d).clone()
    dl = torch.zeros(B, Lpad, device=d, dtype=tdt)
    dl[:, :ldc + K] = dlat.to(d, tdt)
    L.call("sv_sample_bwd", code, p(dl), p(lsd), p(epsd), p(csoft), 0, 0.67, B, ldc, K, Lpad, p(dmud), p(dlsd), p(dlad), st())
    dfeat = torch.empty(B, Cc, device=d)
    dW, db = torch.zeros_like(Wd), torch.zeros_like(bd)
    ws = torch.empty(B, 2 * ldc + K, device=d)
    L.call("sv_head_bwd", p(featd), B, Cc, p(Wd), ldc, K, p(lad), p(dmud), p(dlsd), p(dlad), p(dfeat), p(dW), p(db),
           p(ws), st())
    assert rel(dW, Wr.grad) < 2e-3 and rel(db, br.grad) < 2e-3
    g = torch.empty(B, HW, Cc, device=d, dtype=tdt)
    bs = torch.zeros(2 * Cc, device=d, dtype=ACC)
    L.call("sv_pool_bwd", code, p(xd), p(sc), p(sh), 0.01, p(mn), p(rs), p(dfeat), B, HW, Cc, Cc, p(g), p(bs), 1, st())
    torch.cuda.synchronize()
    gref = xr.grad / scale     # the kernel emits dL/d(BN output); gamma*rstd is applied by sv_bn_bwd_apply
    assert rel(g.float(), gref) < max(tol, 2e-3)
    xh = (x - mean) * rstd
    assert rel(bs[:Cc], gref.sum((0, 1))) < max(tol, 2e-3)
    assert rel(bs[Cc:], (gref * xh).sum((0, 1))) < max(tol, 2e-3)


def test_sampler_label_modes():
    d = dev()
    B, ldc, K, Lpad = 5, 128, 10, 144
    torch.manual_seed(9)
    mu, ls, la = torch.randn(B, ldc), torch.randn(B, ldc) * 0.1, F.log_softmax(torch.randn(B, K), 1)
    eps = torch.randn(B, ldc)
    la_, lb_ = torch.randint(0, K, (B,)), torch.randint(0, K, (B,))
    for mode, lam in ((1, 0.0), (2, 0.8)):
        lat = torch.empty(B, Lpad, device=d)
        cs = torch.empty(B, K, device=d)
        lam_dev = torch.tensor([lam], device=d) if mode == 2 else None      # mode 2: lambda through the device pointer
        L.call("sv_sample_fwd", L.SV_F32, p(mu.to(d)), p(ls.to(d)), p(la.to(d)), p(eps.to(d)), None, p(la_.to(d)),
               p(lb_.to(d)), -1.0 if mode == 2 else lam, p(lam_dev), mode, 0.67, B, ldc, K, Lpad, p(lat), p(cs), st())
        c = F.one_hot(la_, K).float()
        if mode == 2:
            c = lam * c + (1 - lam) * F.one_hot(lb_, K).float()
        assert rel(lat[:, ldc:ldc + K], c) < 1e-6
        assert rel(lat[:, :ldc], mu + torch.exp(ls) * eps) < 1e-6


@pytest.mark.parametrize("bce", [1, 0])
def test_elbo_cls_post(bce):
    d = dev()
    torch.manual_seed(10)
    B, ldc, K = 7, 128, 10
    x = torch.rand(B, 3, 32, 32)
    xr = (torch.randn(B, 3, 32, 32) * 2).requires_grad_(True)
    mu = (torch.randn(B, ldc) * 0.7).requires_grad_(True)
    ls = (torch.randn(B, ldc) * 0.3 - 0.5).requires_grad_(True)
    la = F.log_softmax(torch.randn(B, K) * 2, 1).requires_grad_(True)
    sig = 0.5
    if bce:
        r = F.binary_cross_entropy_with_logits(xr, x, reduction="sum") / B
    else:
        r = F.mse_loss(torch.sigmoid(xr), x, reduction="sum") / (2 * B * sig ** 2)
    kc = 0.5 * torch.sum(mu * mu + torch.exp(2 * ls) - 2 * ls - 1) / B
    kd = torch.sum(torch.exp(la) * (la - torch.log(torch.tensor(1.0 / K)))) / B
    gw = torch.tensor([0.7, -1.3, 2.1])
    (gw[0] * r + gw[1] * kc + gw[2] * kd).backward()
    out3 = torch.zeros(3, device=d)
    args = [p(x.to(d)), p(xr.detach().to(d)), 3 * 32 * 32, p(mu.detach().to(d)), p(ls.detach().to(d)),
            p(la.detach().to(d)), B, ldc, K, bce, sig]
    keep = [x.to(d), xr.detach().to(d), mu.detach().to(d), ls.detach().to(d), la.detach().to(d)]
    args = [p(keep[0]), p(keep[1]), 3 * 32 * 32, p(keep[2]), p(keep[3]), p(keep[4]), B, ldc, K, bce, sig]
    L.call("sv_elbo_fwd", *args, p(out3), st())
    ref = torch.stack([r, kc, kd]).detach()
    assert torch.allclose(out3.cpu(), ref, rtol=2e-5, atol=1e-6), (out3.cpu(), ref)
    dxr, dmu, dls, dla = (torch.empty_like(k) for k in (keep[1], keep[2], keep[3], keep[4]))
    gd = gw.to(d)
    L.call("sv_elbo_bwd", *args, p(gd), p(dxr), p(dmu), p(dls), p(dla), st())
    assert rel(dxr, xr.grad) < 1e-4 and rel(dmu, mu.grad) < 1e-5 and rel(dls, ls.grad) < 1e-5
    assert rel(dla, la.grad) < 1e-5
    # ClsCriterion + posterior terms
    lab = torch.softmax(torch.randn(B, K), 1)
    wgt = torch.rand(B)
    la2 = la.detach().clone().requires_grad_(True)
    c = -torch.mean(torch.sum(la2 * lab, 1) * wgt)
    c.backward()
    o = torch.zeros(1, device=d)
    L.call("sv_cls_fwd", p(keep[4]), p(lab.to(d)), p(wgt.to(d)), B, K, p(o), st())
    assert abs(float(o) - float(c.detach())) < 1e-5
    dp = torch.empty(B, K, device=d)
    one = torch.ones(1, device=d)
    L.call("sv_cls_bwd", p(lab.to(d)), p(wgt.to(d)), B, K, p(one), p(dp), st())
    assert rel(dp, la2.grad) < 1e-5
    mt, stt = torch.randn(B, ldc), torch.rand(B, ldc)
    mu2, ls2 = mu.detach().clone().requires_grad_(True), ls.detach().clone().requires_grad_(True)
    q = (F.mse_loss(mu2, mt, reduction="sum") + F.mse_loss(torch.exp(ls2), stt, reduction="sum")) / B
    q.backward()
    o.zero_()
    L.call("sv_post_fwd", p(keep[2]), p(keep[3]), p(mt.to(d)), p(stt.to(d)), B, ldc, p(o), st())
    assert abs(float(o) - float(q.detach())) < 1e-4 * abs(float(q.detach()))
    d1, d2 = torch.empty(B, ldc, device=d), torch.empty(B, ldc, device=d)
    L.call("sv_post_bwd", p(keep[2]), p(keep[3]), p(mt.to(d)), p(stt.to(d)), B, ldc, p(one), p(d1), p(d2), st())
    assert rel(d1, mu2.grad) < 1e-5 and rel(d2, ls2.grad) < 1e-5


def test_mix_lerp_optimal_match_sgd_layout():
    d = dev()
    torch.manual_seed(11)
    B, D = 9, 128
    a = torch.randn(B, 3, 8, 8)
    idx = torch.randperm(B)
    out = torch.empty_like(a, device=d)
    L.call("sv_mix_lerp", p(a.to(d)), p(idx.to(d)), 0.3, None, B, 3 * 64, 0, p(out), st())
    assert rel(out, 0.3 * a + 0.7 * a[idx]) < 1e-6
    L.call("sv_mix_lerp", p(a.to(d)), p(idx.to(d)), -5.0, p(torch.tensor([0.3], device=d)), B, 3 * 64, 1, p(out), st())
    assert rel(out, 0.3 * a.exp() + 0.7 * a[idx].exp()) < 1e-5
    mu, ls = torch.randn(B, D), torch.randn(B, D) * 0.3
    kl = torch.zeros(B, B)
    for i in range(B):
        for j in range(B):
            s1, s2 = torch.exp(ls[i]), torch.exp(ls[j])
            kl[i, j] = torch.sum(ls[j] - ls[i]) + 0.5 * torch.sum(s1 ** 2 / s2 ** 2) + \
                0.5 * torch.sum((mu[i] - mu[j]) ** 2 / s2 ** 2) - 0.5 * D
    ref = torch.topk(kl, 2, largest=False)[1][:, 1]
    got = torch.empty(B, dtype=torch.int64, device=d)
    L.call("sv_optimal_match", p(mu.to(d)), p(ls.to(d)), B, D, p(got), st())
    assert torch.equal(got.cpu(), ref)
    # SGD, two steps, vs torch.optim.SGD
    n = 1003
    w = torch.randn(n)
    pr = torch.nn.Parameter(w.clone())
    opt = torch.optim.SGD([pr], lr=0.1, momentum=0.9, weight_decay=5e-4)
    pd, vd = w.to(d).clone(), torch.zeros(n, device=d)
    for step in range(2):
        g = torch.randn(n)
        pr.grad = g.clone()
        opt.step()
        L.call("sv_sgd", p(pd), p((g * 4).to(d)), p(vd), n, 0.1, 0.9, 5e-4, 0.25, int(step == 0), st())
    assert rel(pd, pr.detach()) < 1e-6
    # layout round trip
    img = torch.rand(3, 3, 4, 4)
    for dt in ("f32", "bf16"):
        code, tdt, _ = DT[dt]
        o = torch.empty(3, 4, 4, 16, device=d, dtype=tdt)
        L.call("sv_nchw_to_nhwc", code, p(img.to(d)), 3, 3, 4, 4, 16, p(o), st())
        assert rel(o[..., :3].float(), nhwc(bq(img, dt))) < 1e-6 and float(o[..., 3:].float().abs().max()) == 0
        back = torch.empty(3, 3, 4, 4, device=d)
        L.call("sv_nhwc_to_nchw", code, p(o), 3, 3, 4, 4, 16, p(back), st())
        assert rel(back, bq(img, dt)) < 1e-6
    torch.cuda.synchronize()


@pytest.mark.parametrize("mode", ["nchw", "nhwc_bf16", "nhwc_f32", "eval"])
def test_augment_pipeline(mode):
    """sv_augment (gather + reflect pad + flip + crop + ToTensor, lib/dataloader.py:58-70) against the numpy oracle:
    bit-exact in fp32, exact after bf16 rounding in the stem's NHWC16 layout; every crop corner / flip combination."""
    from oracle import augment_oracle as A
    from shot_vae_amd.data import DeviceDataset
    rs = np.random.RandomState(3)
    data = rs.randint(0, 256, size=(50, 32, 32, 3)).astype(np.uint8)
    labels = rs.randint(0, 10, size=50)
    ds = DeviceDataset(torch.from_numpy(data), labels, device=dev())
    B = 37
    index = torch.from_numpy(rs.randint(0, 50, size=B)).to(dev())
    params = np.stack([rs.randint(0, 9, size=B), rs.randint(0, 9, size=B), rs.randint(0, 2, size=B)], 1).astype(np.int32)
    params[:4] = [[0, 0, 0], [8, 8, 1], [0, 8, 1], [8, 0, 0]]
    pd = torch.from_numpy(params).to(dev())
    if mode == "eval":
        out, lab = ds.batch(index, train=False)
        ref = A.batch(data, index.cpu().numpy(), None)
        assert np.array_equal(out.cpu().numpy(), ref)
    elif mode == "nchw":
        out, lab = ds.batch(index, train=True, params=pd)
        ref = A.batch(data, index.cpu().numpy(), params)
        assert np.array_equal(out.cpu().numpy(), ref)
        assert out.shape == (B, 3, 32, 32) and float(out.min()) >= 0.0 and float(out.max()) <= 1.0
    else:
        dt = "bf16" if mode == "nhwc_bf16" else "fp32"
        out, lab = ds.batch(index, train=True, params=pd, nhwc_dtype=dt, cpad=16)
        ref = torch.from_numpy(A.batch(data, index.cpu().numpy(), params)).permute(0, 2, 3, 1)
        ref = ref.to(torch.bfloat16).float() if dt == "bf16" else ref
        assert np.array_equal(out[..., :3].float().cpu().numpy(), ref.numpy())
        assert float(out[..., 3:].float().abs().max()) == 0.0
    assert np.array_equal(lab.cpu().numpy(), labels[index.cpu().numpy()])
    # the drawn parameters stay in range
    dr = ds.draw(1000).cpu().numpy()
    assert dr[:, :2].min() >= 0 and dr[:, :2].max() <= 8 and set(np.unique(dr[:, 2])) <= {0, 1}


# ------------------------------------------------------------------------------------------ batched launches (groups)
@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("case", [(4, 16, 32, 8, 3, 1, 1), (4, 32, 32, 32, 3, 1, 1), (4, 64, 64, 16, 3, 1, 1),
                                  (3, 32, 64, 16, 3, 2, 1), (4, 32, 64, 16, 1, 2, 0), (2, 160, 160, 8, 3, 1, 1)])
def test_batched_groups_equal_separate_launches(dt, case):
    """sv_igemm / sv_wgrad with groups = 3 (blockIdx.y = group: tensors back to back, coefficient vectors [G][C],
    accumulators [G][R][2N]) against three separate launches on the slices: forward with prologue + residual + statistics,
    data gradient with the activation-backward epilogue, weight gradient (sum over the groups)."""
    B, Cin, N, H, k, stride, pad = case
    code, tdt, tol = DT[dt]
    Gn, d, R = 3, dev(), 4
    torch.manual_seed(11)
    Ho = (H + 2 * pad - k) // stride + 1
    x = torch.randn(Gn * B, H, H, Cin, device=d).to(tdt)
    sc, sh = (torch.rand(Gn, Cin, device=d) + 0.5).contiguous(), (torch.randn(Gn, Cin, device=d) * 0.3).contiguous()
    w = bq(torch.randn(N, k * k, Cin) / (k * k * Cin) ** 0.5, dt)
    gf = G.conv_like(B, H, H, Cin, N, k, stride, pad)
    wf = repack(w, gf, False, dt)
    resid = torch.randn(Gn * B, Ho, Ho, N, device=d).to(tdt)

    def fwd(xs, scs, shs, rs, groups):
        out = torch.zeros(xs.shape[0], Ho, Ho, N, dtype=tdt, device=d)
        stats = torch.zeros(groups, R, 2 * N, device=d, dtype=ACC)
        a = L.SvIgemmArgs()
        a.x, a.w, a.out, a.residual = xs.data_ptr(), wf.data_ptr(), out.data_ptr(), rs.data_ptr()
        a.pro_scale, a.pro_shift, a.pro_slope = scs.data_ptr(), shs.data_ptr(), 0.01
        a.stats, a.replicas, a.groups = stats.data_ptr(), R, groups
        L.call("sv_igemm", C.byref(gf), code, C.byref(a), st())
        torch.cuda.synchronize()
        return out, stats.sum(1)

    ob, sb = fwd(x, sc, sh, resid, Gn)
    for gi in range(Gn):
        o1, s1 = fwd(x[gi * B:(gi + 1) * B].contiguous(), sc[gi].contiguous(), sh[gi].contiguous(),
                     resid[gi * B:(gi + 1) * B].contiguous(), 1)
        assert torch.equal(ob[gi * B:(gi + 1) * B], o1), "group %d of the batched forward differs" % gi
        assert rel(sb[gi], s1[0]) < 1e-5
    # data gradient with the activation-backward epilogue
    gd = G.convT_like(B, Ho, Ho, N, Cin, k, stride, pad)
    wd = repack(w, gd, True, dt)
    dy = torch.randn(Gn * B, Ho, Ho, N, device=d).to(tdt)
    emu, ers = (torch.randn(Gn, Cin, device=d) * 0.1).contiguous(), (torch.rand(Gn, Cin, device=d) + 0.5).contiguous()

    def dgrad(dys, xs, scs, shs, mus, rss, groups):
        dx = torch.zeros(xs.shape[0], H, H, Cin, dtype=tdt, device=d)
        bs = torch.zeros(groups, R, 2 * Cin, device=d, dtype=ACC)
        a = L.SvIgemmArgs()
        a.x, a.w, a.out = dys.data_ptr(), wd.data_ptr(), dx.data_ptr()
        a.ex, a.ex_scale, a.ex_shift, a.ex_mean, a.ex_rstd = (t.data_ptr() for t in (xs, scs, shs, mus, rss))
        a.ex_slope, a.bsums, a.replicas, a.groups = 0.01, bs.data_ptr(), R, groups
        L.call("sv_igemm", C.byref(gd), code, C.byref(a), st())
        torch.cuda.synchronize()
        return dx, bs.sum(1)

    db, bb = dgrad(dy, x, sc, sh, emu, ers, Gn)
    for gi in range(Gn):
        sl = slice(gi * B, (gi + 1) * B)
        d1, b1 = dgrad(dy[sl].contiguous(), x[sl].contiguous(), sc[gi].contiguous(), sh[gi].contiguous(),
                       emu[gi].contiguous(), ers[gi].contiguous(), 1)
        assert torch.equal(db[sl], d1), "group %d of the batched data gradient differs" % gi
        assert rel(bb[gi], b1[0]) < 1e-5
    # weight gradient: the batched launch sums over the groups
    ws = torch.empty(8 * 1024 * 1024, device=d)
    dwb = torch.zeros(N, k * k, Cin, device=d)
    L.call("sv_wgrad", C.byref(gf), code, p(x), p(sc), p(sh), 0.01, p(dy), p(dwb), 0, int(dt == "bf16"), p(ws), ws.numel(),
           Gn, st())
    dws = torch.zeros(N, k * k, Cin, device=d)
    for gi in range(Gn):
        sl = slice(gi * B, (gi + 1) * B)
        L.call("sv_wgrad", C.byref(gf), code, p(x[sl].contiguous()), p(sc[gi].contiguous()), p(sh[gi].contiguous()), 0.01,
               p(dy[sl].contiguous()), p(dws), 0, int(dt == "bf16"), p(ws), ws.numel(), 1, st())
    torch.cuda.synchronize()
    assert rel(dwb, dws) < 1e-4, rel(dwb, dws)


def test_batched_bn_kernels_equal_separate_launches():
    """sv_bn_finalize / sv_bn_bwd_apply / sv_pool_fwd / sv_pool_bwd / sv_bn_running_update with groups = 3 against the
    per-group calls; the running statistics receive the groups' momentum updates in group order."""
    Gn, B, HW, Cc, R, d = 3, 4, 64, 32, 2, dev()
    torch.manual_seed(3)
    n = B * HW
    x = torch.randn(Gn, n, Cc, device=d) * (1 + torch.arange(Gn, device=d).view(Gn, 1, 1)) + 0.3
    gamma, beta = torch.rand(Cc, device=d) + 0.5, torch.randn(Cc, device=d)
    stats = torch.stack([torch.stack([torch.cat([x[g].sum(0), (x[g] ** 2).sum(0)]) * f for f in (0.25, 0.75)]) for g in range(Gn)]).to(ACC).contiguous()
    A = (Gn * Cc + 63) // 64 * 64
    bnbuf = torch.zeros(4 * A, device=d)
    o = bnbuf.data_ptr()
    L.call("sv_bn_finalize", p(stats), R, Cc, float(n), p(gamma), p(beta), 1e-5, 0.1, None, None, C.c_void_p(o),
           C.c_void_p(o + 4 * A), C.c_void_p(o + 8 * A), C.c_void_p(o + 12 * A), Gn, st())
    torch.cuda.synchronize()
    sc, sh, mn, rs = (bnbuf[i * A: i * A + Gn * Cc].view(Gn, Cc) for i in range(4))
    for g in range(Gn):
        mu, var = x[g].mean(0), x[g].var(0, unbiased=False)
        assert rel(mn[g], mu) < 1e-4 and rel(rs[g], torch.rsqrt(var + 1e-5)) < 1e-4
        assert rel(sc[g], gamma * torch.rsqrt(var + 1e-5)) < 1e-4
    # running statistics: three ordered momentum updates from (mean, rstd)
    bufs = torch.cat([torch.zeros(Cc), torch.ones(Cc)]).to(d)
    tab = torch.tensor([0, 0, Cc, Cc], dtype=torch.int32, device=d)
    cnt = torch.tensor([float(n)], device=d)
    L.call("sv_bn_running_update", p(tab), p(cnt), 1, p(bnbuf), p(bufs), 1e-5, 0.1, 64, Gn, st())
    rm, rv = torch.zeros(Cc, device=d), torch.ones(Cc, device=d)
    for g in range(Gn):
        rm = 0.9 * rm + 0.1 * x[g].mean(0)
        rv = 0.9 * rv + 0.1 * x[g].var(0, unbiased=True)
    torch.cuda.synchronize()
    assert rel(bufs[:Cc], rm) < 1e-4 and rel(bufs[Cc:], rv) < 1e-3
    # backward apply, batched against per group
    gup = torch.randn(Gn, n, Cc, device=d)
    res = torch.randn(Gn, n, Cc, device=d)
    xh = (x - mn.view(Gn, 1, Cc)) * rs.view(Gn, 1, Cc)
    bsums = torch.stack([torch.stack([torch.cat([gup[g].sum(0), (gup[g] * xh[g]).sum(0)]) * 0.5] * 2) for g in range(Gn)]).to(ACC).contiguous()

    def apply(xs, gs, rsd, mns, rss, bss, groups):
        dgam, dbet = torch.zeros(Cc, device=d), torch.zeros(Cc, device=d)
        br_ = (L.SvBnBranch * 1)()
        br_[0].g, br_[0].bsums, br_[0].gamma = gs.data_ptr(), bss.data_ptr(), gamma.data_ptr()
        br_[0].dgamma, br_[0].dbeta, br_[0].replicas = dgam.data_ptr(), dbet.data_ptr(), 2
        dx = torch.empty_like(xs)
        L.call("sv_bn_bwd_apply", L.SV_F32, n, Cc, Cc, p(xs), p(mns), p(rss), float(n), br_, 1, p(rsd), p(dx), groups, st())
        torch.cuda.synchronize()
        return dx, dgam, dbet

    dxb, dgb, dbb = apply(x.contiguous(), gup.contiguous(), res.contiguous(), mn.contiguous(), rs.contiguous(), bsums, Gn)
    dg_sum, db_sum = torch.zeros(Cc, device=d), torch.zeros(Cc, device=d)
    for g in range(Gn):
        d1, dg1, db1 = apply(x[g].contiguous(), gup[g].contiguous(), res[g].contiguous(), mn[g].contiguous(),
                             rs[g].contiguous(), bsums[g].contiguous(), 1)
        assert torch.equal(dxb[g], d1)
        dg_sum += dg1
        db_sum += db1
    assert rel(dgb, dg_sum) < 1e-5 and rel(dbb, db_sum) < 1e-5
    # pooling with per-group coefficients
    xp = x.view(Gn * B, HW, Cc).contiguous()
    feat = torch.zeros(Gn * B, Cc, device=d)
    L.call("sv_pool_fwd", L.SV_F32, p(xp), p(sc.contiguous()), p(sh.contiguous()), 0.01, Gn * B, HW, Cc, Cc, p(feat), Gn, st())
    torch.cuda.synchronize()
    for g in range(Gn):
        want = F.leaky_relu(x[g].view(B, HW, Cc) * sc[g] + sh[g], 0.01).mean(1)
        assert rel(feat[g * B:(g + 1) * B], want) < 1e-5


def test_repack_batch_equals_per_layer_repack():
    """sv_repack_batch (one launch for every layer / direction / phase) against sv_repack, on the WRN-10-1 plan."""
    import shot_vae_amd as S
    m = S.VariationalAutoEncoder("wideresnet-10-1", num_input_channels=3, img_size=(32, 32), data_parallel=False,
                                 continuous_latent_dim=128, disc_latent_dim=10, small_input=True).cuda()
    eng, plan = m._engine, m._plan
    eng.ensure_packs()
    torch.cuda.synchronize()
    ref = torch.zeros_like(eng.packs)
    base, pb, es = eng.param.data_ptr(), ref.data_ptr(), ref.element_size()
    for cv in plan.convs:
        mm = C.c_void_p(base + 4 * cv.master_off)
        L.call("sv_repack", eng.code, mm, cv.N, cv.T, cv.Cin, 0, C.byref(cv.geom_fwd(1)), C.c_void_p(pb + es * cv.fwd_off), st())
        L.call("sv_repack", eng.code, mm, cv.N, cv.T, cv.Cin, 1, C.byref(cv.geom_dgrad(1)), C.c_void_p(pb + es * cv.dgrad_off), st())
    torch.cuda.synchronize()
    assert torch.equal(ref.view(torch.int16), eng.packs.view(torch.int16))
    assert float(eng.packs.float().abs().sum()) > 0


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("case", [c for c in CONV_CASES if not (c[4] == 3 and c[5] == 1 and c[1] % 32 == 0 and c[2] % 32 == 0)])
def test_halo_kernel_conv_cases(dt, case):
    """halo.hip over its whole range (SV_OPT_HALO_ALL): stride-2 3x3, 1x1 (both strides), the 16-channel layers -- forward
    with every fusion and the data gradient with the activation-backward epilogue, against torch."""
    with L.options(halo_all=1):
        test_conv_forward_fused(dt, case)
        test_conv_dgrad_with_activation_backward(dt, case)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("B,Cc,HW,Gn", [(16, 128, 64, 2), (6, 640, 64, 1), (8, 64, 16, 1), (12, 128, 64, 4), (11, 64, 16, 1),
                                        (8, 16, 4, 2)])
def test_pool_kernels_vectorised(dt, B, Cc, HW, Gn):
    """sv_pool_fwd / sv_pool_bwd (BatchNorm + LeakyReLU + global average pool and its backward with the BatchNorm-backward
    sums) in the 16-byte form (images x pixel parts x 8-channel groups per block; several images per block, 640 channels =
    one image per block with idle threads) and the element-wise fallback (B = 11), batched over groups with their own
    coefficients."""
    code, tdt, tol = DT[dt]
    torch.manual_seed(12)
    d = dev()
    Bg = B // Gn
    x = bq(torch.randn(B, HW, Cc), dt)
    scale, shift = torch.rand(Gn, Cc) + 0.5, torch.randn(Gn, Cc) * 0.2
    mean, rstd = torch.randn(Gn, Cc) * 0.1, torch.rand(Gn, Cc) + 0.5
    dfeat = torch.randn(B, Cc)
    gi = torch.arange(B) // Bg
    u = x * scale[gi][:, None, :] + shift[gi][:, None, :]
    feat = F.leaky_relu(u, 0.01).mean(1)
    gref = (dfeat / HW)[:, None, :] * torch.where(u > 0, torch.ones_like(u), torch.full_like(u, 0.01))
    xh = (x - mean[gi][:, None, :]) * rstd[gi][:, None, :]
    xd = x.to(d, tdt).contiguous()
    sc, sh, mn, rs = [t.to(d).contiguous() for t in (scale, shift, mean, rstd)]
    featd = torch.empty(B, Cc, device=d)
    L.call("sv_pool_fwd", code, p(xd), p(sc), p(sh), 0.01, B, HW, Cc, Cc, p(featd), Gn, st())
    assert rel(featd, feat) < (1e-4 if dt == "f32" else 1e-3)
    gd = torch.empty(B, HW, Cc, device=d, dtype=tdt)
    bs = torch.zeros(Gn, 2 * Cc, device=d, dtype=ACC)
    dfd = dfeat.to(d)
    L.call("sv_pool_bwd", code, p(xd), p(sc), p(sh), 0.01, p(mn), p(rs), p(dfd), B, HW, Cc, Cc, p(gd), p(bs), Gn, st())
    torch.cuda.synchronize()
    assert rel(gd.float(), gref) < tol
    for k in range(Gn):
        sl = slice(k * Bg, (k + 1) * Bg)
        assert rel(bs[k, :Cc], gref[sl].sum((0, 1))) < max(tol, 1e-3) * 3
        assert rel(bs[k, Cc:], (gref[sl] * xh[sl]).sum((0, 1))) < max(tol, 1e-3) * 3


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("M,N,ld", [(5000, 16, 16), (70001, 16, 16), (4099, 24, 32), (9000, 128, 128), (300, 16, 16), (4500, 4, 4),
                                    (2048, 256, 256), (2048, 1024, 1024)])   # (the linear layers of config 5: few rows, 128 per block)
def test_colsum(dt, M, N, ld):
    """sv_colsum (bias gradients: column sums over all rows, accumulating into `out`): the 16-byte-load kernel (N and ld
    multiples of 8, >= 4096 rows; ragged row counts, a channel-group count that does not divide 256, a padded row
    stride) and the element-wise kernel."""
    code, tdt, tol = DT[dt]
    torch.manual_seed(11)
    d = dev()
    y = bq(torch.randn(M, ld), dt)
    yd = y.to(d, tdt).contiguous()
    out = torch.full((N,), 0.5, device=d)
    L.call("sv_colsum", code, p(yd), M, N, ld, p(out), st())
    torch.cuda.synchronize()
    ref = y[:, :N].double().sum(0).float() + 0.5
    assert (out.cpu() - ref).abs().max() < 2e-3 * (M ** 0.5), (out.cpu() - ref).abs().max()


@pytest.mark.parametrize("case", [(3, 16, 160, 32, 3, 1, 1), (5, 16, 128, 16, 3, 1, 1), (2, 32, 320, 8, 3, 1, 1)])
def test_halo_kernel_thin_output_default_dispatch(case):
    """Default dispatch (no option): the data gradient of a 3x3 layer with 16 / 32 input and >= 128 output channels -- few
    output channels of the gather-GEMM, many input channels (first block of WRN-28-10) -- takes the channel-chunked LDS-halo
    kernel; the forward of the same layer is covered alongside."""
    test_conv_dgrad_with_activation_backward("bf16", case)
    test_conv_forward_fused("bf16", case)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("H,Cin,N,B", [(2, 512, 256, 4), (4, 256, 128, 4), (8, 128, 64, 3), (16, 64, 16, 2)])
def test_halo_kernel_convT_cases(dt, H, Cin, N, B):
    """... and the ConvTranspose2d(4, 2, 1) decoder layers (four sub-pixel phases in one block) with their data gradients
    (4x4 stride-2 convolutions)."""
    with L.options(halo_all=1, disable=L.K_TCONVR):
        test_convT_forward_and_dgrad(dt, H, Cin, N, B)


@pytest.mark.parametrize("B,pro,Gn,fold", [(1, True, 1, False), (3, False, 1, False), (70, True, 2, False), (33, True, 2, True), (300, True, 1, True)])
def test_register_resident_last_decoder_layer(B, pro, Gn, fold):
    """tconv.hip's 16x16 kernel in its forward form: ConvTranspose2d(64, 16 (3 padded), 4, 2, 1) at 16x16 -> 32x32 (decoder.py:58) with
    the BatchNorm + ReLU prologue (finished coefficients or folded finalisation), no epilogue fusion -- against torch fp32 on the
    same bf16 operands and against the LDS-halo kernel it replaces."""
    torch.manual_seed(B)
    d = dev()
    H, Cin, N = 16, 64, 16
    x = bq(torch.randn(Gn * B, Cin, H, H) * 1.5 + 0.3, "bf16")
    w = bq(torch.randn(Cin, N, 4, 4) / (Cin * 4) ** 0.5, "bf16")
    g = G.convT_like(B, H, H, Cin, N, 4, 2, 1)
    wp = repack(w.permute(1, 2, 3, 0).reshape(N, 16, Cin).contiguous(), g, False, "bf16")
    xd = nhwc(x).to(d, torch.bfloat16).contiguous()
    gamma, beta = (torch.rand(Cin, device=d) + 0.5), torch.randn(Cin, device=d) * 0.2
    count, R = float(B * H * H), 16
    xf = xd.float().view(Gn, -1, Cin)
    stats = torch.cat([xf.sum(1)[:, None, :] / R, (xf * xf).sum(1)[:, None, :] / R], dim=2).repeat(1, R, 1).to(ACC).contiguous()     # [G][R][2C]
    coef = torch.zeros(4, Gn, Cin, device=d)
    L.call("sv_bn_finalize", p(stats), R, Cin, count, p(gamma), p(beta), 1e-5, 0.1, None, None, p(coef[0]), p(coef[1]), p(coef[2]), p(coef[3]), Gn, st())

    def run(disable, folded):
        out = torch.full((Gn * B, 2 * H, 2 * H, N), 7.0, dtype=torch.bfloat16, device=d)
        c2 = torch.zeros(4, Gn, Cin, device=d)
        a = L.SvIgemmArgs()
        a.x, a.w, a.out, a.replicas, a.groups = xd.data_ptr(), wp.data_ptr(), out.data_ptr(), 1, Gn
        if pro and folded:
            a.pro_scale, a.pro_shift, a.pro_slope = c2[0].data_ptr(), c2[1].data_ptr(), 0.0
            a.fold_stats, a.fold_replicas, a.fold_count, a.fold_eps = stats.data_ptr(), R, count, 1e-5
            a.fold_gamma, a.fold_beta, a.fold_mean, a.fold_rstd = gamma.data_ptr(), beta.data_ptr(), c2[2].data_ptr(), c2[3].data_ptr()
        elif pro:
            a.pro_scale, a.pro_shift, a.pro_slope = coef[0].data_ptr(), coef[1].data_ptr(), 0.0
        with L.options(disable=disable):
            L.call("sv_igemm", C.byref(g), L.SV_BF16, C.byref(a), st())
        torch.cuda.synchronize()
        if pro and folded:
            assert rel(c2, coef) < 2e-6
        return out.float().cpu()

    out, ref_out = run(0, fold), run(L.K_TCONVR, False)
    sc, sh = coef[0].cpu(), coef[1].cpu()
    for gi in range(Gn):
        xs = x[gi * B:(gi + 1) * B]
        act = bq(F.relu(xs * sc[gi][None, :, None, None] + sh[gi][None, :, None, None]), "bf16") if pro else xs
        y = F.conv_transpose2d(act, w, None, 2, 1)
        o = nchw(out[gi * B:(gi + 1) * B])
        assert rel(o, y) < 4e-3, (gi, rel(o, y))
    assert rel(out, ref_out) < 6e-3


@pytest.mark.parametrize("Cin", [64, 32])
@pytest.mark.parametrize("B,Gn,budget", [(1, 1, 0), (3, 1, 0), (70, 2, 0), (300, 1, 0), (64, 2, 16)])
def test_register_resident_last_decoder_layer_dgrad(B, Gn, budget, Cin):
    """dconv.hip: the data gradient of ConvTranspose2d(64, 16 (3 padded), 4, 2, 1) -- a 4x4 stride-2 convolution 16 -> 64 at 32x32
    (decoder.py:58) -- with the activation-backward epilogue of the BatchNorm + ReLU in front of that layer, against torch fp32 on the
    same bf16 operands (both bands of an image: top / bottom padding rows, the row shared by the bands) and against the LDS-halo
    kernel it replaces.  Cin = 32: the same for svhn_VAE's last layer, ConvTranspose2d(32, 3, 4, 2, 1) (svhn_vae.py:131; one band)."""
    torch.manual_seed(B)
    d = dev()
    N, H = 16, 16                                # the ConvTranspose's channels: its data gradient maps N -> Cin
    w = bq(torch.randn(Cin, N, 4, 4) / (Cin * 4) ** 0.5, "bf16")
    dy = bq(torch.randn(Gn * B, N, 2 * H, 2 * H), "bf16")
    xraw = bq(torch.randn(Gn * B, Cin, H, H), "bf16")
    scale, shift = torch.rand(Gn, Cin) + 0.5, torch.randn(Gn, Cin) * 0.3
    mean, rstd = torch.randn(Gn, Cin) * 0.1, torch.rand(Gn, Cin) + 0.5
    gd = G.conv_like(B, 2 * H, 2 * H, N, Cin, 4, 2, 1)
    wp = repack(w.permute(1, 2, 3, 0).reshape(N, 16, Cin).contiguous(), gd, True, "bf16")
    dyd, xd = nhwc(dy).to(d, torch.bfloat16).contiguous(), nhwc(xraw).to(d, torch.bfloat16).contiguous()
    vec = [t.to(d).contiguous() for t in (scale, shift, mean, rstd)]
    R = 4

    def run(disable):
        out = torch.full((Gn * B, H, H, Cin), 7.0, dtype=torch.bfloat16, device=d)
        sums = torch.zeros(Gn, R, 2 * Cin, device=d, dtype=ACC)
        a = L.SvIgemmArgs()
        a.x, a.w, a.out, a.replicas, a.groups, a.block_budget = dyd.data_ptr(), wp.data_ptr(), out.data_ptr(), R, Gn, budget
        a.ex, a.ex_scale, a.ex_shift, a.ex_mean, a.ex_rstd = [t.data_ptr() for t in [xd] + vec]
        a.ex_slope, a.bsums = 0.0, sums.data_ptr()
        with L.options(disable=disable):
            L.call("sv_igemm", C.byref(gd), L.SV_BF16, C.byref(a), st())
        torch.cuda.synchronize()
        return out.float().cpu(), sums.sum(1).float().cpu()

    out, sums = run(0)
    ref_out, ref_sums = run(L.K_TCONVR_EX)
    for gi in range(Gn):
        sl = slice(gi * B, (gi + 1) * B)
        da = F.conv2d(dy[sl], w, None, 2, 1)
        u = xraw[sl] * scale[gi][None, :, None, None] + shift[gi][None, :, None, None]
        gref = da * (u > 0).float()
        xh = (xraw[sl] - mean[gi][None, :, None, None]) * rstd[gi][None, :, None, None]
        assert rel(nchw(out[sl]), gref) < 4e-3
        assert rel(sums[gi, :Cin], gref.sum((0, 2, 3))) < 3e-3
        assert rel(sums[gi, Cin:], (gref * xh).sum((0, 2, 3))) < 3e-3
    assert rel(out, ref_out) < 6e-3 and rel(sums, ref_sums) < 2e-3


@pytest.mark.parametrize("B,Gn,budget", [(1, 1, 0), (5, 1, 0), (70, 2, 0), (300, 1, 0), (64, 2, 16)])
def test_register_resident_first_svhn_convolution(B, Gn, budget):
    """dconv.hip's bias form: svhn_VAE's first layer, Conv2d(3 (16 padded), 32, 4, 2, 1) with its bias (svhn_vae.py:62), against
    torch fp32 on the same bf16 operands and against the LDS-halo kernel it replaces."""
    torch.manual_seed(B)
    d = dev()
    Cin, N, H = 16, 32, 32
    w = bq(torch.randn(N, Cin, 4, 4) / (3 * 16) ** 0.5, "bf16")
    w[:, 3:] = 0
    x = bq(torch.randn(Gn * B, Cin, H, H), "bf16")
    x[:, 3:] = 0
    bias = torch.randn(N) * 0.2                  # (shared by the groups of a batched launch)
    g = G.conv_like(B, H, H, Cin, N, 4, 2, 1)
    wp = repack(w.permute(0, 2, 3, 1).reshape(N, 16, Cin).contiguous(), g, False, "bf16")
    xd, bd = nhwc(x).to(d, torch.bfloat16).contiguous(), bias.to(d).contiguous()

    def run(disable):
        out = torch.full((Gn * B, H // 2, H // 2, N), 7.0, dtype=torch.bfloat16, device=d)
        a = L.SvIgemmArgs()
        a.x, a.w, a.out, a.replicas, a.groups, a.block_budget = xd.data_ptr(), wp.data_ptr(), out.data_ptr(), 1, Gn, budget
        a.bias = bd.data_ptr()
        with L.options(disable=disable):
            L.call("sv_igemm", C.byref(g), L.SV_BF16, C.byref(a), st())
        torch.cuda.synchronize()
        return out.float().cpu()

    out, ref_out = run(0), run(L.K_TCONVR_EX)
    for gi in range(Gn):
        sl = slice(gi * B, (gi + 1) * B)
        y = F.conv2d(x[sl], w, bias, 2, 1)
        assert rel(nchw(out[sl]), y) < 4e-3
    assert rel(out, ref_out) < 6e-3


@pytest.mark.parametrize("B,groups,R", [(5, 1, 8), (40, 4, 32), (3, 2, 256)])
def test_register_resident_convT_folds_its_batchnorm(B, groups, R):
    """tconv.hip's forward form with sv_igemm_args::fold_*: every block derives scale / shift from the raw statistics, block 0 of a
    group stores the four vectors -- against sv_bn_finalize + the same launch with finished coefficients."""
    d = dev()
    torch.manual_seed(B)
    H, Cin, N = 8, 128, 64
    g = G.convT_like(B, H, H, Cin, N, 4, 2, 1)
    x = (torch.randn(groups * B, H, H, Cin, device=d) * 1.5 + 0.3).bfloat16()
    w = (torch.randn(G.packed_size(g), device=d) * 0.05).bfloat16()
    count = float(B * H * H)
    xf = x.float().view(groups, -1, Cin)
    parts = torch.rand(groups, R, 1, device=d) + 0.1
    parts = parts / parts.sum(1, keepdim=True)
    stats = torch.cat([xf.sum(1)[:, None, :] * parts, (xf * xf).sum(1)[:, None, :] * parts], dim=2).to(ACC).contiguous()     # [G][R][2C]
    gamma, beta = (torch.rand(Cin, device=d) + 0.5), torch.randn(Cin, device=d) * 0.2

    def launch(fold):
        coef = torch.zeros(4, groups, Cin, device=d)
        out = torch.zeros(groups * B, 2 * H, 2 * H, N, dtype=torch.bfloat16, device=d)
        a = L.SvIgemmArgs()
        a.x, a.w, a.out, a.groups, a.replicas = x.data_ptr(), w.data_ptr(), out.data_ptr(), groups, 1
        a.pro_scale, a.pro_shift, a.pro_slope = coef[0].data_ptr(), coef[1].data_ptr(), 0.0
        if fold:
            a.fold_stats, a.fold_replicas, a.fold_count, a.fold_eps = stats.data_ptr(), R, count, 1e-5
            a.fold_gamma, a.fold_beta = gamma.data_ptr(), beta.data_ptr()
            a.fold_mean, a.fold_rstd = coef[2].data_ptr(), coef[3].data_ptr()
        else:
            L.call("sv_bn_finalize", p(stats), R, Cin, count, p(gamma), p(beta), 1e-5, 0.1, None, None, p(coef[0]), p(coef[1]),
                   p(coef[2]), p(coef[3]), groups, st())
        L.call("sv_igemm", C.byref(g), L.SV_BF16, C.byref(a), st())
        torch.cuda.synchronize()
        return coef, out.float()

    c1, o1 = launch(True)
    c0, o0 = launch(False)
    assert rel(c1, c0) < 2e-6
    assert rel(o1, o0) < 1e-3 and bool(torch.isfinite(o1).all())


@pytest.mark.parametrize("B,pro,stats,Gn,budget", [(1, True, True, 1, 0), (3, True, True, 1, 0), (70, True, True, 1, 0), (37, False, False, 1, 0),
                                                  (600, True, True, 1, 0), (130, True, True, 4, 0), (64, True, False, 2, 0), (96, True, True, 1, 8)])
def test_register_resident_convT_128_64(B, pro, stats, Gn, budget):
    """tconv.hip (ConvTranspose2d(4, 2, 1) 128 -> 64 at 8x8, decoder.py:40-47: the weights of a sub-pixel phase live in the
    registers of the persistent block) against torch fp32 on the same bf16 operands -- one image per block, several images per
    block with an odd count (the two LDS images alternate), with / without the BatchNorm + ReLU prologue and the statistics,
    batched groups with their own coefficients and accumulators, a small block budget -- and against the LDS-halo kernel it
    replaces."""
    torch.manual_seed(B)
    H, Cin, N = 8, 128, 64
    d = dev()
    x = bq(torch.randn(Gn * B, Cin, H, H), "bf16")
    w = bq(torch.randn(Cin, N, 4, 4) / (Cin * 4) ** 0.5, "bf16")
    scale, shift = torch.rand(Gn, Cin) + 0.5, torch.randn(Gn, Cin) * 0.3
    master = w.permute(1, 2, 3, 0).reshape(N, 16, Cin).contiguous()
    g = G.convT_like(B, H, H, Cin, N, 4, 2, 1)
    wp = repack(master, g, False, "bf16")
    xd = nhwc(x).to(d, torch.bfloat16).contiguous()
    scd, shd = scale.to(d).contiguous(), shift.to(d).contiguous()
    R = 4

    def run(disable):
        out = torch.full((Gn * B, 2 * H, 2 * H, N), 7.0, dtype=torch.bfloat16, device=d)
        sums = torch.zeros(Gn, R, 2 * N, device=d, dtype=ACC)
        a = L.SvIgemmArgs()
        a.x, a.w, a.out, a.replicas, a.groups, a.block_budget = xd.data_ptr(), wp.data_ptr(), out.data_ptr(), R, Gn, budget
        if pro:
            a.pro_scale, a.pro_shift, a.pro_slope = scd.data_ptr(), shd.data_ptr(), 0.0
        if stats:
            a.stats = sums.data_ptr()
        with L.options(disable=disable):
            L.call("sv_igemm", C.byref(g), L.SV_BF16, C.byref(a), st())
        torch.cuda.synchronize()
        return out.float().cpu(), sums.sum(1).float().cpu()

    out, sums = run(0)
    ref_out, ref_sums = run(L.K_TCONVR)
    for gi in range(Gn):
        xs = x[gi * B:(gi + 1) * B]
        act = bq(F.relu(xs * scale[gi][None, :, None, None] + shift[gi][None, :, None, None]), "bf16") if pro else xs
        y = F.conv_transpose2d(act, w, None, 2, 1)
        o = nchw(out[gi * B:(gi + 1) * B])
        assert rel(o, y) < 4e-3, (gi, rel(o, y))
        assert (o - bq(y, "bf16")).abs().max() <= 2.0 ** -6 * y.abs().max()
        if stats:
            assert rel(sums[gi, :N], y.sum((0, 2, 3))) < 3e-3
            assert rel(sums[gi, N:], (y * y).sum((0, 2, 3))) < 3e-3
            assert rel(sums[gi], ref_sums[gi]) < 1e-3
    assert rel(out, ref_out) < 6e-3        # (two bf16 roundings of sums formed in different orders)


@pytest.mark.parametrize("case", [(40, 16, 16, 32, 3, 1, 1), (16, 16, 32, 32, 3, 1, 1), (8, 32, 64, 32, 3, 2, 1),
                                  (8, 64, 128, 16, 3, 2, 1), (4, 32, 64, 16, 3, 2, 1)])
def test_tap_fused_wgrad_conv(case):
    """hwgrad.hip (all taps of a layer in one block, LDS-halo) on the thin / stride-2 convolutions against torch; once as a
    plain launch (accumulating twice) and once batched over three groups."""
    B, Cin, N, H, k, stride, pad = case
    code, tdt, tol = DT["bf16"]
    torch.manual_seed(5)
    d = dev()
    Gn = 3
    x = bq(torch.randn(Gn * B, Cin, H, H), "bf16")
    scale, shift = torch.rand(Gn, Cin) + 0.5, torch.randn(Gn, Cin) * 0.3
    Ho = (H + 2 * pad - k) // stride + 1
    dy = bq(torch.randn(Gn * B, N, Ho, Ho), "bf16")
    wref = torch.zeros(N, Cin, k, k)
    for gi in range(Gn):
        sl = slice(gi * B, (gi + 1) * B)
        a = bq(F.leaky_relu(x[sl] * scale[gi][None, :, None, None] + shift[gi][None, :, None, None], 0.01), "bf16")
        wref += torch.nn.grad.conv2d_weight(a, (N, Cin, k, k), dy[sl], stride, pad)
    g = G.conv_like(B, H, H, Cin, N, k, stride, pad)
    xd, dyd = nhwc(x).to(d, tdt), nhwc(dy).to(d, tdt)
    sc, sh = scale.to(d).contiguous(), shift.to(d).contiguous()
    ws = torch.full((16 * 1024 * 1024,), float("nan"), device=d)
    with L.options(halo_all=1):
        dw = torch.zeros(N, k * k, Cin, device=d)
        L.call("sv_wgrad", C.byref(g), code, p(xd), p(sc), p(sh), 0.01, p(dyd), p(dw), 0, 1, p(ws), ws.numel(), Gn, st())
        dw2 = torch.zeros(N, k * k, Cin, device=d)
        with L.options(disable=L.K_HWGRAD):
            L.call("sv_wgrad", C.byref(g), code, p(xd), p(sc), p(sh), 0.01, p(dyd), p(dw2), 0, 1, p(ws), ws.numel(), Gn, st())
    torch.cuda.synchronize()
    got = dw.cpu().view(N, k, k, Cin).permute(0, 3, 1, 2)
    assert rel(got, wref) < tol, rel(got, wref)
    assert rel(dw, dw2) < 2e-3, rel(dw, dw2)             # against the generic kernel on the same bf16 operands


@pytest.mark.parametrize("case", [(5, 160, 320, 32, 3, 2, 1), (3, 320, 640, 16, 3, 2, 1), (7, 160, 320, 32, 1, 2, 0),
                                  (3, 320, 160, 8, 3, 2, 1), (9, 160, 160, 8, 1, 1, 0)])
def test_wide_cooperative_wgrad(case):
    """The 160 x 160 cooperative tiles of the generic weight gradient (wgrad.hip, wgradc_kernel: the stride-2 3x3 layers and
    1x1 shortcuts of WRN-28-10) against torch and against the 64 x 64 wave-private tiles; plain (accumulating twice) and
    batched over three groups with their own BatchNorm coefficients; m-ranges that are not multiples of 32 rows."""
    B, Cin, N, H, k, stride, pad = case
    code, tdt, tol = DT["bf16"]
    torch.manual_seed(6)
    d = dev()
    Gn = 3
    x = bq(torch.randn(Gn * B, Cin, H, H), "bf16")
    scale, shift = torch.rand(Gn, Cin) + 0.5, torch.randn(Gn, Cin) * 0.3
    Ho = (H + 2 * pad - k) // stride + 1
    dy = bq(torch.randn(Gn * B, N, Ho, Ho), "bf16")
    wref = torch.zeros(N, Cin, k, k)
    for gi in range(Gn):
        sl = slice(gi * B, (gi + 1) * B)
        a = bq(F.leaky_relu(x[sl] * scale[gi][None, :, None, None] + shift[gi][None, :, None, None], 0.01), "bf16")
        wref += torch.nn.grad.conv2d_weight(a, (N, Cin, k, k), dy[sl], stride, pad)
    g = G.conv_like(B, H, H, Cin, N, k, stride, pad)
    xd, dyd = nhwc(x).to(d, tdt), nhwc(dy).to(d, tdt)
    sc, sh = scale.to(d).contiguous(), shift.to(d).contiguous()
    ws = torch.full((16 * 1024 * 1024,), float("nan"), device=d)
    with L.options(wide_min_blocks=1):
        dw = torch.zeros(N, k * k, Cin, device=d)
        L.call("sv_wgrad", C.byref(g), code, p(xd), p(sc), p(sh), 0.01, p(dyd), p(dw), 0, 1, p(ws), ws.numel(), Gn, st())
        dw2 = torch.zeros(N, k * k, Cin, device=d)
        with L.options(disable=L.K_WGRAD_WIDE):
            L.call("sv_wgrad", C.byref(g), code, p(xd), p(sc), p(sh), 0.01, p(dyd), p(dw2), 0, 1, p(ws), ws.numel(), Gn, st())
        g1 = G.conv_like(Gn * B, H, H, Cin, N, k, stride, pad)          # one group, no prologue, two accumulating calls
        dw3 = torch.zeros(N, k * k, Cin, device=d)
        for _ in range(2):
            L.call("sv_wgrad", C.byref(g1), code, p(xd), None, None, 0.01, p(dyd), p(dw3), 0, 1, None, 0, 1, st())
    torch.cuda.synchronize()
    got = dw.cpu().view(N, k, k, Cin).permute(0, 3, 1, 2)
    assert rel(got, wref) < tol, rel(got, wref)
    assert rel(dw, dw2) < 2e-3, rel(dw, dw2)
    w3 = torch.nn.grad.conv2d_weight(x, (N, Cin, k, k), dy, stride, pad)
    assert rel(dw3.cpu().view(N, k, k, Cin).permute(0, 3, 1, 2) / 2, w3) < tol


@pytest.mark.parametrize("H,Cin,N,B", [(4, 256, 128, 32), (8, 128, 64, 16), (16, 64, 16, 8)])
def test_tap_fused_wgrad_convT(H, Cin, N, B):
    """... and on the ConvTranspose2d(4, 2, 1) decoder layers (four phases of dy, one input region)."""
    code, tdt, tol = DT["bf16"]
    torch.manual_seed(6)
    x = bq(torch.randn(B, Cin, H, H), "bf16")
    dy = bq(torch.randn(B, N, 2 * H, 2 * H), "bf16")
    w = torch.zeros(Cin, N, 4, 4, requires_grad=True)
    F.conv_transpose2d(F.relu(x), w, None, 2, 1).backward(dy)
    g = G.convT_like(B, H, H, Cin, N, 4, 2, 1)
    d = dev()
    one, zero = torch.ones(Cin, device=d), torch.zeros(Cin, device=d)
    ws = torch.full((16 * 1024 * 1024,), float("nan"), device=d)
    dw = torch.zeros(N, 16, Cin, device=d)
    with L.options(halo_all=1):
        L.call("sv_wgrad", C.byref(g), code, p(nhwc(x).to(d, tdt)), p(one), p(zero), 0.0, p(nhwc(dy).to(d, tdt)), p(dw), 0, 1,
               p(ws), ws.numel(), 1, st())
    torch.cuda.synchronize()
    got = dw.cpu().view(N, 4, 4, Cin).permute(3, 0, 1, 2)
    assert rel(got, w.grad) < tol, rel(got, w.grad)


# ------------------------------------------------------------------------------------------ deterministic accumulation
@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("case", [(8, 32, 32, 16, 3, 1, 1), (6, 64, 64, 16, 3, 1, 1), (5, 64, 128, 8, 3, 2, 1), (8, 16, 32, 32, 3, 1, 1),
                                  (4, 32, 64, 16, 1, 2, 0), (4, 160, 160, 8, 3, 1, 1), (16, 128, 128, 8, 3, 1, 1),
                                  # several M ranges of the generic / the cooperative wide weight gradient: a zeroed slab each
                                  (48, 64, 128, 16, 3, 2, 1), (24, 160, 320, 16, 3, 2, 1)])
def test_deterministic_mode_reproduces_bit_for_bit(dt, case):
    """SV_OPT_DETERMINISTIC: the BatchNorm statistics of a forward, the BatchNorm-backward sums of a data gradient (groups =
    2) and the weight gradient are IDENTICAL over repeated launches (one adder per accumulator address: replicas sized by
    sv_igemm_query_blocks; ordered slab reduction / a single M range for the weights), agree with the atomic path to
    rounding, and a replica count below 4 x blocks is refused."""
    B, Cin, N, H, k, stride, pad = case
    code, tdt, tol = DT[dt]
    Gn, d = 2, dev()
    torch.manual_seed(5)
    Ho = (H + 2 * pad - k) // stride + 1
    x = torch.randn(Gn * B, H, H, Cin, device=d).to(tdt)
    sc, sh = (torch.rand(Gn, Cin, device=d) + 0.5).contiguous(), (torch.randn(Gn, Cin, device=d) * 0.3).contiguous()
    emu, ers = (torch.randn(Gn, Cin, device=d) * 0.1).contiguous(), (torch.rand(Gn, Cin, device=d) + 0.5).contiguous()
    w = bq(torch.randn(N, k * k, Cin) / (k * k * Cin) ** 0.5, dt)
    gf = G.conv_like(B, H, H, Cin, N, k, stride, pad)
    gd = G.convT_like(B, Ho, Ho, N, Cin, k, stride, pad)
    wf, wd = repack(w, gf, False, dt), repack(w, gd, True, dt)
    dy = torch.randn(Gn * B, Ho, Ho, N, device=d).to(tdt)
    ws = torch.empty(8 * 1024 * 1024, device=d)

    def run(det, replicas=None):
        with L.options(deterministic=int(det), wide_min_blocks=1):
            a = L.SvIgemmArgs()
            out = torch.zeros(Gn * B, Ho, Ho, N, dtype=tdt, device=d)
            a.x, a.w, a.out = x.data_ptr(), wf.data_ptr(), out.data_ptr()
            a.pro_scale, a.pro_shift, a.pro_slope, a.groups = sc.data_ptr(), sh.data_ptr(), 0.01, Gn
            a.stats = x.data_ptr()
            R = replicas or (L.det_replicas(gf, code, a) if det else 4)
            stats = torch.zeros(Gn, R, 2 * N, device=d, dtype=ACC)
            a.stats, a.replicas = stats.data_ptr(), R
            L.call("sv_igemm", C.byref(gf), code, C.byref(a), st())
            a2 = L.SvIgemmArgs()
            dx = torch.zeros(Gn * B, H, H, Cin, dtype=tdt, device=d)
            a2.x, a2.w, a2.out, a2.groups = dy.data_ptr(), wd.data_ptr(), dx.data_ptr(), Gn
            a2.ex, a2.ex_scale, a2.ex_shift, a2.ex_mean, a2.ex_rstd = (t.data_ptr() for t in (x, sc, sh, emu, ers))
            a2.ex_slope, a2.bsums = 0.01, x.data_ptr()
            R2 = replicas or (L.det_replicas(gd, code, a2) if det else 4)
            bs = torch.zeros(Gn, R2, 2 * Cin, device=d, dtype=ACC)
            a2.bsums, a2.replicas = bs.data_ptr(), R2
            L.call("sv_igemm", C.byref(gd), code, C.byref(a2), st())
            dw = torch.zeros(N, k * k, Cin, device=d)
            L.call("sv_wgrad", C.byref(gf), code, p(x), p(sc), p(sh), 0.01, p(dy), p(dw), 0, int(dt == "bf16"), p(ws), ws.numel(),
                   Gn, st())
            torch.cuda.synchronize()
            # the consumers' order: replicas in index order
            return out, stats.double().sum(1).float(), stats, dx, bs, dw

    # (the reference launch takes the kernels the deterministic mode takes: the register-resident stride-2 data gradients of
    #  tconv.hip decline that mode, and their outputs differ from the LDS-halo kernels' in the last bf16 bit)
    with L.options(disable=L.K_TCONVR_EX):
        ref = run(False)
    first = run(True)
    for _ in range(3):
        again = run(True)
        assert torch.equal(again[2], first[2]), "statistics differ between two deterministic launches"
        assert torch.equal(again[4], first[4]), "BatchNorm-backward sums differ between two deterministic launches"
        assert torch.equal(again[5], first[5]), "weight gradient differs between two deterministic launches"
        assert torch.equal(again[0], first[0]) and torch.equal(again[3], first[3])
    # every replica address received at most one add: a replica row is either untouched or one wave's sums
    assert rel(first[1], ref[1]) < 1e-4 and rel(first[4].sum(1), ref[4].sum(1)) < 1e-4
    assert rel(first[5], ref[5]) < 2e-4 and torch.equal(first[0], ref[0]) and torch.equal(first[3], ref[3])
    with pytest.raises(L.ShotVaeHipError, match="replicas"):
        run(True, replicas=2)


def test_deterministic_two_pass_reductions():
    """SV_OPT_DETERMINISTIC at sizes where the small reductions take several blocks: column sums, the ELBO / classification /
    posterior loss sums, the pooling backward (BatchNorm-backward sums) and sv_bn_bwd_apply over thousands of accumulator
    replicas (folded 256 : 1 first; dgamma / dbeta of a batched launch by one block) -- per-block slots in the library's
    scratch ring + an ordered second pass: bit-identical over repeats, equal to the default (atomic) mode to rounding."""
    d = dev()
    torch.manual_seed(21)
    M, N = 300000, 16
    y = torch.randn(M, N, device=d).to(torch.bfloat16)
    B, ldc, K = 96, 128, 10
    x = torch.rand(B, 3, 32, 32, device=d)
    xr = torch.randn(B, 3, 32, 32, device=d) * 2
    mu, ls = torch.randn(B, ldc, device=d) * 0.7, torch.randn(B, ldc, device=d) * 0.3 - 0.5
    la = F.log_softmax(torch.randn(B, K, device=d) * 2, 1)
    lab = torch.softmax(torch.randn(B, K, device=d), 1)
    mt, stt = torch.randn(B, ldc, device=d), torch.rand(B, ldc, device=d)
    Bp, Cc, HW, Gn = 512, 128, 64, 2
    xp = torch.randn(Bp, HW, Cc, device=d).to(torch.bfloat16)
    sc, sh = torch.rand(Gn, Cc, device=d) + 0.5, torch.randn(Gn, Cc, device=d) * 0.2
    mn, rs = torch.randn(Gn, Cc, device=d) * 0.1, torch.rand(Gn, Cc, device=d) + 0.5
    dfeat = torch.randn(Bp, Cc, device=d)
    # sv_bn_bwd_apply: 2 groups, 2 branches, 1 536 replicas per group
    Mb, Cb, R = 4096, 64, 1536
    xb = torch.randn(Gn, Mb, Cb, device=d).to(torch.bfloat16)
    g1, g2 = (torch.randn(Gn, Mb, Cb, device=d).to(torch.bfloat16) for _ in range(2))
    bmean, brstd = torch.randn(Gn, Cb, device=d) * 0.1, torch.rand(Gn, Cb, device=d) + 0.5
    bs1, bs2 = (torch.randn(Gn, R, 2 * Cb, device=d).to(ACC) for _ in range(2))
    gam1, gam2 = torch.rand(Cb, device=d) + 0.5, torch.rand(Cb, device=d) + 0.5

    def run(det):
        with L.options(deterministic=int(det)):
            cs = torch.full((N,), 0.5, device=d)
            L.call("sv_colsum", L.SV_BF16, p(y), M, N, N, p(cs), st())
            out3, oc, op = torch.zeros(3, device=d), torch.zeros(1, device=d), torch.zeros(1, device=d)
            L.call("sv_elbo_fwd", p(x), p(xr), 3 * 32 * 32, p(mu), p(ls), p(la), B, ldc, K, 1, 1.0, p(out3), st())
            L.call("sv_cls_fwd", p(la), p(lab), None, B, K, p(oc), st())
            L.call("sv_post_fwd", p(mu), p(ls), p(mt), p(stt), B, ldc, p(op), st())
            gd = torch.empty(Bp, HW, Cc, device=d, dtype=torch.bfloat16)
            bs = torch.zeros(Gn, 2 * Cc, device=d, dtype=ACC)
            L.call("sv_pool_bwd", L.SV_BF16, p(xp), p(sc), p(sh), 0.01, p(mn), p(rs), p(dfeat), Bp, HW, Cc, Cc, p(gd), p(bs), Gn, st())
            br_ = (L.SvBnBranch * 2)()
            dg = [torch.zeros(Cb, device=d) for _ in range(4)]
            for k, (g_, b_, gm) in enumerate(((g1, bs1, gam1), (g2, bs2, gam2))):
                br_[k].g, br_[k].bsums, br_[k].gamma, br_[k].replicas = g_.data_ptr(), b_.data_ptr(), gm.data_ptr(), R
                br_[k].dgamma, br_[k].dbeta = dg[2 * k].data_ptr(), dg[2 * k + 1].data_ptr()
            dx = torch.empty_like(xb)
            L.call("sv_bn_bwd_apply", L.SV_BF16, Mb, Cb, Cb, p(xb), p(bmean), p(brstd), float(Mb), br_, 2, None, p(dx), Gn, st())
            torch.cuda.synchronize()
            return [cs, out3, oc, op, gd, bs, dx] + dg

    ref = run(False)
    first = run(True)
    for _ in range(3):
        again = run(True)
        for a_, b_ in zip(again, first):
            assert torch.equal(a_, b_)
    names = ["colsum", "elbo", "cls", "post", "pool g", "pool sums", "bn dx", "dgamma1", "dbeta1", "dgamma2", "dbeta2"]
    for nm, a_, b_ in zip(names, first, ref):
        assert rel(a_.float(), b_.float()) < (1e-2 if nm == "bn dx" else 2e-4), nm
    assert rel(first[0].cpu(), y.float().sum(0).cpu() + 0.5) < 1e-4
    assert rel(first[7], bs1[:, :, Cb:].double().sum((0, 1)).float()) < 1e-4 and rel(first[8], bs1[:, :, :Cb].double().sum((0, 1)).float()) < 1e-4


@pytest.mark.parametrize("case", [(64, 32, 32, 32, 3, 1, 1), (64, 128, 128, 8, 3, 1, 1), (32, 32, 64, 32, 3, 2, 1), (32, 32, 64, 32, 1, 2, 0),
                                  (64, 256, 128, 4, 3, 1, 1), (16, 160, 160, 32, 3, 1, 1)])
def test_start_signal_forks_a_second_stream(case):
    """sv_igemm_args::start_flag / sv_stream_wait_flag (ABI 5): the first block of every kernel of the sv_igemm family stores
    the value when it starts, i.e. after everything enqueued before it on its stream -- a second stream that waits for the flag
    sees that work's results without an event between the streams.  Stream A: a long producer chain, then the signalling
    convolution; stream B: the wait, then a copy of the producer's result.  Every dispatch target of the table (persistent
    narrow / wide / 160-channel 3x3, stride-2 and 1x1 LDS-halo kernels, the generic gather GEMM) is one case."""
    B, Cin, N, H, k, stride, pad = case
    d = dev()
    torch.manual_seed(3)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    g = G.conv_like(B, H, H, Cin, N, k, stride, pad)
    Ho = (H + 2 * pad - k) // stride + 1
    w = bq(torch.randn(N, k * k, Cin) / (k * k * Cin) ** 0.5, "bf16")
    wp = repack(w, g, False, "bf16")
    x = torch.randn(B, H, H, Cin, device=d).to(torch.bfloat16)
    out = torch.empty(B, Ho, Ho, N, device=d, dtype=torch.bfloat16)
    big = torch.zeros(64 * 1024 * 1024, device=d)
    got = torch.empty_like(big)
    torch.cuda.synchronize()
    for rep in range(3):
        flag, value = C.c_void_p(), C.c_uint32()
        L.call("sv_stream_flag_next", C.c_void_p(sa.cuda_stream), C.byref(flag), C.byref(value))
        with torch.cuda.stream(sa):
            for _ in range(6):
                big.add_(1.0)                    # ~2 ms of producer work in front of the signalling launch
            a = L.SvIgemmArgs()
            a.x, a.w, a.out = x.data_ptr(), wp.data_ptr(), out.data_ptr()
            a.start_flag, a.start_value = flag.value, value.value
            L.call("sv_igemm", C.byref(g), L.SV_BF16, C.byref(a), C.c_void_p(sa.cuda_stream))
        L.call("sv_stream_wait_flag", C.c_void_p(sb.cuda_stream), flag, value)
        with torch.cuda.stream(sb):
            got.copy_(big)
        torch.cuda.synchronize()
        assert float(got.min()) == 6.0 * (rep + 1) and float(got.max()) == 6.0 * (rep + 1)
    assert L.lib().sv_flag_timeouts() == 0
    ref = F.conv2d(nchw(x.float().cpu()), w.reshape(N, k, k, Cin).permute(0, 3, 1, 2), None, stride, pad)
    assert rel(nchw(out.float().cpu()), ref) < DT["bf16"][2]


def test_flag_fork_fails_closed_on_a_timeout():
    """A side-stream wait whose signal never comes gives up after ~3 s -- and must not pass silently: the sticky host-mapped
    counter makes the next check raise (Engine._join_side, FlatSGD.step, dp's all-reduce all call L.check_flag_timeouts),
    without a copy or a synchronisation of its own."""
    import shot_vae_amd as S
    d = dev()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    assert L.lib().sv_flag_timeouts() == 0
    L.check_flag_timeouts()
    flag, value = C.c_void_p(), C.c_uint32()
    L.call("sv_stream_flag_next", C.c_void_p(sa.cuda_stream), C.byref(flag), C.byref(value))     # ... and nobody signals it
    L.call("sv_stream_wait_flag", C.c_void_p(sb.cuda_stream), flag, value)
    assert L.lib().sv_flag_timeouts() == 0            # (reading the counter does not wait for the kernel)
    sb.synchronize()
    try:
        assert L.lib().sv_flag_timeouts() == 1
        with pytest.raises(L.ShotVaeHipError, match="timed out"):
            L.check_flag_timeouts("test")
        model = S.VariationalAutoEncoder("wideresnet-10-1", num_input_channels=3, img_size=(32, 32), data_parallel=False,
                                         continuous_latent_dim=128, disc_latent_dim=10, small_input=True).to(d)
        opt = S.FlatSGD(model)
        opt.zero_grad()
        before = model._engine.param.clone()
        with pytest.raises(L.ShotVaeHipError, match="timed out"):
            opt.step()                                # the update is refused, the parameters are untouched
        assert torch.equal(before, model._engine.param)
    finally:
        L.call("sv_flag_timeouts_reset")
    # (the signal arrives late: the stream's sequence stays consistent for the tests that follow)
    a = torch.zeros(1, device=d)
    assert L.lib().sv_flag_timeouts() == 0 and float(a) == 0.0


def test_flag_fork_timeout_in_the_first_backward_is_recoverable():
    """the FIRST flag-forked backward of an engine is verified with one synchronisation; a wait that gave up there has
    corrupted nothing yet (optimizer step and all-reduce come after the backward): the engine switches to event forks, CLEARS the
    counter, releases the side stream's operands and raises for this step only -- a caller that drops the step continues."""
    import shot_vae_amd as S
    from shot_vae_amd.engine import Engine
    d = dev()
    K, B = 10, 8
    model = S.VariationalAutoEncoder("wideresnet-10-1", num_input_channels=3, img_size=(32, 32), data_parallel=False,
                                     continuous_latent_dim=128, disc_latent_dim=K, small_input=True, rng="device").to(d).train()
    elbo, cls = S.VAECriterion(discrete_dim=K).cuda(), S.ClsCriterion()
    opt = S.FlatSGD(model, lr=0.01)
    opt.zero_grad()
    il, iu = torch.rand(B, 3, 32, 32, device=d), torch.rand(B, 3, 32, 32, device=d)
    ll = torch.randint(0, K, (B,), device=d)
    saved = (Engine.flag_fork, Engine._flag_forks_verified)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    try:
        Engine.flag_fork, Engine._flag_forks_verified = True, False
        model._engine.flag_fork = True
        flag, value = C.c_void_p(), C.c_uint32()
        L.call("sv_stream_flag_next", C.c_void_p(sa.cuda_stream), C.byref(flag), C.byref(value))     # a wait nobody signals
        L.call("sv_stream_wait_flag", C.c_void_p(sb.cuda_stream), flag, value)
        with pytest.raises(L.ShotVaeHipError, match="drop the step"):
            S.train_step_grouped(model, elbo, cls, opt, il, ll, iu, S.schedule(10))
        assert L.lib().sv_flag_timeouts() == 0 and model._engine.flag_fork is False and Engine._flag_forks_verified
        assert not model._engine._side_keep
        opt.zero_grad()
        before = model._engine.param.clone()
        ls, lu = S.train_step_grouped(model, elbo, cls, opt, il, ll, iu, S.schedule(10))      # event forks: a valid step
        torch.cuda.synchronize()
        assert bool(torch.isfinite(ls)) and not torch.equal(before, model._engine.param)
    finally:
        Engine.flag_fork, Engine._flag_forks_verified = saved
        L.call("sv_flag_timeouts_reset")
        torch.cuda.synchronize()


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("B,Cc,H,Gn", [(6, 32, 32, 2), (4, 64, 16, 1), (3, 16, 8, 3)])
def test_compact_shortcut_branch_equals_strided(dt, B, Cc, H, Gn):
    """sv_gather_even + sv_bn_branch::sparse < 0 (ABI 5): a BatchNorm-backward branch that exists at the even positions only,
    stored compactly [M / 4][C] (the data gradient of a stride-2 1x1 layer run as a dense product over the stride-2 grid),
    gives bit for bit the dx of the same branch in the strided form (sparse > 0); the gather returns exactly the even
    positions."""
    code, tdt, tol = DT[dt]
    d = dev()
    torch.manual_seed(31)
    x = torch.randn(Gn * B, H, H, Cc, device=d).to(tdt)
    g1 = torch.randn(Gn * B, H, H, Cc, device=d).to(tdt)
    gs = torch.full((Gn * B, H, H, Cc), float("nan"), device=d).to(tdt)           # strided: odd positions are never read
    gs[:, ::2, ::2] = torch.randn(Gn * B, H // 2, H // 2, Cc, device=d).to(tdt)
    gc = torch.empty(Gn * B, H // 2, H // 2, Cc, device=d, dtype=tdt)
    L.call("sv_gather_even", code, p(gs), Gn * B, H, H, Cc, p(gc), st())
    torch.cuda.synchronize()
    assert torch.equal(gc, gs[:, ::2, ::2].contiguous())
    mean, rstd = (torch.randn(Gn, Cc, device=d) * 0.1).contiguous(), (torch.rand(Gn, Cc, device=d) + 0.5).contiguous()
    gam1, gam2 = torch.rand(Cc, device=d) + 0.5, torch.rand(Cc, device=d) + 0.5
    R = 4
    bs1, bs2 = torch.randn(Gn, R, 2 * Cc, device=d).to(ACC), torch.randn(Gn, R, 2 * Cc, device=d).to(ACC)
    wl1 = int(H).bit_length()

    def run(g2, sparse):
        br_ = (L.SvBnBranch * 2)()
        dg = [torch.zeros(Cc, device=d) for _ in range(4)]
        for k, (g_, b_, gm, sp) in enumerate(((g1, bs1, gam1, 0), (g2, bs2, gam2, sparse))):
            br_[k].g, br_[k].bsums, br_[k].gamma, br_[k].replicas, br_[k].sparse = g_.data_ptr(), b_.data_ptr(), gm.data_ptr(), R, sp
            br_[k].dgamma, br_[k].dbeta = dg[2 * k].data_ptr(), dg[2 * k + 1].data_ptr()
        dx = torch.empty_like(x)
        L.call("sv_bn_bwd_apply", code, B * H * H, Cc, Cc, p(x), p(mean), p(rstd), float(B * H * H), br_, 2, None, p(dx), Gn, st())
        torch.cuda.synchronize()
        return dx

    a, b = run(gs, wl1), run(gc, -wl1)
    assert bool(torch.isfinite(a.float()).all()) and torch.equal(a, b)


def test_rank_permutation():
    """sv_rank_permutation: perm = argsort of the keys (ties: the lower index first), several batches per launch, sizes up
    to a full minibatch."""
    d = dev()
    torch.manual_seed(2)
    for n, nb in ((1, 1), (7, 3), (512, 2), (1000, 1), (4096, 2)):
        keys = torch.rand(nb, n, device=d)
        if n > 4:
            keys[:, 3] = keys[:, 1]                     # a tie
        perm = torch.full((nb, n), -1, dtype=torch.int64, device=d)
        L.call("sv_rank_permutation", p(keys), n, nb, p(perm), st())
        torch.cuda.synchronize()
        ref = torch.argsort(keys, dim=1, stable=True)
        assert torch.equal(perm, ref), (n, nb)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("case", [(6, 32, 64, 32), (4, 64, 128, 16), (3, 160, 320, 16)])
def test_sparse_stride2_shortcut_gradient(dt, case):
    """Data gradient of a stride-2 1x1 shortcut with sv_igemm_args::sparse_out (the three tapless output parities are left
    unwritten) + sv_bn_bwd_apply with sv_bn_branch::sparse (they are not read): the same dx, BatchNorm sums and dgamma /
    dbeta as the dense pair, with the skipped positions of the gradient tensor poisoned."""
    B, Cin, N, H = case
    code, tdt, tol = DT[dt]
    d = dev()
    torch.manual_seed(13)
    Ho = H // 2
    x = torch.randn(B, H, H, Cin, device=d).to(tdt)                 # the shortcut's input (raw, pre-BatchNorm)
    dy = torch.randn(B, Ho, Ho, N, device=d).to(tdt)
    g1 = torch.randn(B, H, H, Cin, device=d).to(tdt)                # the other branch's gradient (dense)
    w = bq(torch.randn(N, 1, Cin) / Cin ** 0.5, dt)
    gd = G.convT_like(B, Ho, Ho, N, Cin, 1, 2, 0)
    wd = repack(w, gd, True, dt)
    sc, sh = torch.rand(Cin, device=d) + 0.5, torch.randn(Cin, device=d) * 0.3
    mean, rstd = torch.randn(Cin, device=d) * 0.1, torch.rand(Cin, device=d) + 0.5
    gamma = torch.rand(Cin, device=d) + 0.5
    R = 4
    res = []
    with L.options(wide_min_blocks=1):
        for sparse in (0, 1):
            gi = torch.full((B, H, H, Cin), float("nan"), dtype=tdt, device=d)
            bs = torch.zeros(R, 2 * Cin, device=d, dtype=ACC)
            a = L.SvIgemmArgs()
            a.x, a.w, a.out, a.replicas, a.sparse_out = dy.data_ptr(), wd.data_ptr(), gi.data_ptr(), R, sparse
            a.ex, a.ex_scale, a.ex_shift, a.ex_mean, a.ex_rstd = (t.data_ptr() for t in (x, sc, sh, mean, rstd))
            a.ex_slope, a.bsums = 0.01, bs.data_ptr()
            L.call("sv_igemm", C.byref(gd), code, C.byref(a), st())
            bs1 = torch.zeros(R, 2 * Cin, device=d, dtype=ACC)
            bs1[0, :Cin] = g1.float().sum((0, 1, 2))
            bs1[0, Cin:] = (g1.float() * ((x.float() - mean) * rstd)).sum((0, 1, 2))
            dgam, dbet = torch.zeros(2, Cin, device=d), torch.zeros(2, Cin, device=d)
            br = (L.SvBnBranch * 2)()
            for k, (gt, bt) in enumerate(((g1, bs1), (gi, bs))):
                br[k].g, br[k].bsums, br[k].gamma = gt.data_ptr(), bt.data_ptr(), gamma.data_ptr()
                br[k].dgamma, br[k].dbeta, br[k].replicas = dgam[k].data_ptr(), dbet[k].data_ptr(), R
            br[1].sparse = (H.bit_length()) if sparse else 0
            dx = torch.empty(B, H, H, Cin, dtype=tdt, device=d)
            L.call("sv_bn_bwd_apply", code, B * H * H, Cin, Cin, p(x), p(mean), p(rstd), float(B * H * H), br, 2, None, p(dx), 1, st())
            torch.cuda.synchronize()
            res.append((gi, bs.sum(0), dx, dgam.clone(), dbet.clone()))
    (gi0, b0, dx0, dg0, db0), (gi1, b1, dx1, dg1, db1) = res
    assert bool(torch.isfinite(gi0.float()).all())                                  # dense: every position written
    even = gi1[:, ::2, ::2]
    assert torch.equal(even, gi0[:, ::2, ::2]) and bool(torch.isnan(gi1[:, 1::2].float()).all())   # sparse: odd rows untouched
    assert float(gi0[:, 1::2].float().abs().max()) == 0.0 and float(gi0[:, :, 1::2].float().abs().max()) == 0.0
    # (the two runs accumulate the shortcut branch's sums with float atomics in their own order: equal to rounding)
    assert bool(torch.isfinite(dx1.float()).all()) and rel(dx1, dx0) < (1e-5 if dt == "f32" else 1e-2)
    assert rel(b1, b0) < 1e-5 and rel(dg1, dg0) < 1e-5 and rel(db1, db0) < 1e-5


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("halo_all", [0, 1])
@pytest.mark.parametrize("B,Cin,N,H,k,stride,groups,R", [(8, 32, 32, 32, 3, 1, 1, 8), (16, 64, 64, 16, 3, 1, 4, 32),
                                                       (4, 32, 64, 16, 3, 2, 2, 4), (8, 128, 128, 8, 3, 1, 1, 2),
                                                       (6, 16, 32, 32, 1, 1, 3, 16), (4, 16, 32, 32, 3, 1, 2, 64),
                                                       # enough tiles for the wide kernel conv3x3w, which folds too (every block
                                                       # stores the coefficients its chunk DMAs then read)
                                                       (1024, 128, 128, 8, 3, 1, 1, 32), (512, 128, 128, 8, 3, 1, 2, 32),
                                                       # the stride-2 forwards with register-resident weights (sconv.hip) fold too
                                                       (9, 64, 128, 16, 3, 2, 2, 16), (5, 32, 64, 32, 3, 2, 3, 64),
                                                       # ... and the pointwise shortcuts (pconv.hip)
                                                       (7, 32, 64, 32, 1, 2, 2, 8), (5, 64, 128, 16, 1, 2, 4, 64)])
def test_folded_batchnorm_finalisation(dt, B, Cin, N, H, k, stride, groups, R, halo_all):
    """sv_igemm_args::fold_* (ABI 4): the BatchNorm in front of a conv-like layer finalised BY the launch -- inside the
    persistent 3x3 kernel (every block derives scale / shift from the raw statistics, block 0 stores the four vectors), inside the
    wide kernel conv3x3w (128 / 256 input channels: every block stores them), or by the sv_bn_finalize launch sv_igemm issues
    itself for the other kernels -- against sv_bn_finalize + the same launch with
    finished coefficients: identical coefficient vectors (2e-6: the replicas are summed in another order) and outputs.
    halo_all = 1 sends the thin / strided shapes to the persistent LDS-halo kernel (halop), which folds too."""
    code, tdt, tol = DT[dt]
    if halo_all and k == 3 and stride == 1 and Cin >= 32:
        pytest.skip("stride-1 3x3 layers do not reach halo.hip")
    if dt == "f32" and Cin > 32 and k == 3 and stride == 1:
        pytest.skip("fp32 operands: the persistent kernel covers 32 input channels")
    d = dev()
    torch.manual_seed(17 + Cin + H)
    pad = k // 2
    g = G.conv_like(B, H, H, Cin, N, k, stride, pad)
    x = (torch.randn(groups * B, H, H, Cin, device=d) * 1.5 + 0.3).to(tdt)
    master = (torch.randn(N, k * k, Cin) / (k * k * Cin) ** 0.5)
    w = repack(master, g, False, dt)
    count = float(B * H * H)
    # raw statistics of x per group, spread over R replicas as a producer would leave them
    xf = x.float().view(groups, -1, Cin)
    parts = torch.rand(groups, R, 1, device=d) + 0.1
    parts = parts / parts.sum(1, keepdim=True)
    stats = torch.cat([xf.sum(1)[:, None, :] * parts, (xf * xf).sum(1)[:, None, :] * parts], dim=2).to(ACC).contiguous()     # [G][R][2C]
    gamma, beta = (torch.rand(Cin, device=d) + 0.5), torch.randn(Cin, device=d) * 0.2
    Ho = g.Hout

    def launch(fold):
        coef = torch.zeros(4, groups, Cin, device=d)
        out = torch.zeros(groups * B, Ho, Ho, N, dtype=tdt, device=d)
        a = L.SvIgemmArgs()
        a.x, a.w, a.out, a.groups, a.replicas = x.data_ptr(), w.data_ptr(), out.data_ptr(), groups, 1
        a.pro_scale, a.pro_shift, a.pro_slope = coef[0].data_ptr(), coef[1].data_ptr(), 0.01
        if fold:
            a.fold_stats, a.fold_replicas, a.fold_count, a.fold_eps = stats.data_ptr(), R, count, 1e-5
            a.fold_gamma, a.fold_beta = gamma.data_ptr(), beta.data_ptr()
            a.fold_mean, a.fold_rstd = coef[2].data_ptr(), coef[3].data_ptr()
        else:
            L.call("sv_bn_finalize", p(stats), R, Cin, count, p(gamma), p(beta), 1e-5, 0.1, None, None, p(coef[0]), p(coef[1]),
                   p(coef[2]), p(coef[3]), groups, st())
        L.call("sv_igemm", C.byref(g), code, C.byref(a), st())
        torch.cuda.synchronize()
        return coef, out.float()

    with L.options(halo_all=halo_all):
        c0, o0 = launch(False)
        c1, o1 = launch(True)
    assert float((c0 - c1).abs().max() / c0.abs().max()) < 2e-6
    assert float(c1[3].min()) > 0                                   # rstd written for every group
    assert float((o0 - o1).abs().max() / o0.abs().max()) < (1e-5 if dt == "f32" else 1e-2)
    # and against torch: batch-norm + LeakyReLU + convolution of group 0
    xg = x[:B].float().permute(0, 3, 1, 2).cpu()
    mu, var = xg.mean((0, 2, 3)), xg.var((0, 2, 3), unbiased=False)
    yn = (xg - mu[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + 1e-5) * gamma.cpu()[None, :, None, None] + beta.cpu()[None, :, None, None]
    act = bq(F.leaky_relu(yn, 0.01), dt)
    wt = bq(master, dt).view(N, k, k, Cin).permute(0, 3, 1, 2)
    ref = F.conv2d(act, wt, None, stride, pad)
    assert rel(nchw(o1[:B]), ref) < max(tol, 3e-3)
