"""Host logic: the gather-GEMM geometry + weight packing reproduce torch's convolutions and their
gradients (CPU; kernels are emulated by tests/_emu.py)."""
import pytest
import torch
import torch.nn.functional as F

from shot_vae_amd import geometry as G
from tests import _emu as E


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("k,stride,pad,H", [(3, 1, 1, 8), (3, 2, 1, 8), (1, 1, 0, 4), (1, 2, 0, 8), (3, 1, 1, 2)])
def test_conv_forward_dgrad_wgrad(k, stride, pad, H):
    torch.manual_seed(0)
    B, Cin, N = 2, 16, 32
    x = torch.randn(B, Cin, H, H, requires_grad=True)
    w = torch.randn(N, Cin, k, k, requires_grad=True)
    y = F.conv2d(x, w, None, stride, pad)
    dy = torch.randn_like(y)
    y.backward(dy)
    master = w.detach().permute(0, 2, 3, 1).reshape(N, k * k, Cin)          # [N][T][Cin]
    gf = G.conv_like(B, H, H, Cin, N, k, stride, pad)
    out = E.emu_igemm(gf, nhwc(x.detach()), E.emu_repack(master, gf, False))
    assert torch.allclose(nchw(out), y.detach(), atol=1e-4)
    gd = G.convT_like(B, H // stride, H // stride, N, Cin, k, stride, pad)   # dgrad: roles swapped
    dx = E.emu_igemm(gd, nhwc(dy), E.emu_repack(master, gd, True))
    assert dx.shape[1] == H
    assert torch.allclose(nchw(dx), x.grad, atol=1e-4)
    dw = E.emu_wgrad(gf, nhwc(x.detach()), nhwc(dy))
    assert torch.allclose(dw.view(N, k, k, Cin).permute(0, 3, 1, 2), w.grad, atol=1e-3)


@pytest.mark.parametrize("H", [1, 2, 4])
def test_convT_4x4_s2_forward_dgrad_wgrad(H):
    torch.manual_seed(1)
    B, Cin, N = 2, 32, 16
    x = torch.randn(B, Cin, H, H, requires_grad=True)
    w = torch.randn(Cin, N, 4, 4, requires_grad=True)                        # torch ConvT layout
    y = F.conv_transpose2d(x, w, None, 2, 1)
    dy = torch.randn_like(y)
    y.backward(dy)
    master = w.detach().permute(1, 2, 3, 0).reshape(N, 16, Cin)             # [cout][ky*4+kx][cin]
    gf = G.convT_like(B, H, H, Cin, N, 4, 2, 1)
    out = E.emu_igemm(gf, nhwc(x.detach()), E.emu_repack(master, gf, False))
    assert torch.allclose(nchw(out), y.detach(), atol=1e-4)
    gd = G.conv_like(B, 2 * H, 2 * H, N, Cin, 4, 2, 1)
    dx = E.emu_igemm(gd, nhwc(dy), E.emu_repack(master, gd, True))
    assert torch.allclose(nchw(dx), x.grad, atol=1e-4)
    dw = E.emu_wgrad(gf, nhwc(x.detach()), nhwc(dy))
    assert torch.allclose(dw.view(N, 4, 4, Cin).permute(3, 0, 1, 2), w.grad, atol=1e-3)
    if H == 1:   # 1x1 input: only one of the four taps per phase can ever be inside the image
        assert all(gf.phase[p].ntap == 1 for p in range(4))


def test_prologue_residual_and_padding_semantics():
    """zero padding applies AFTER BN+LeakyReLU (wideresnet.py:27-30 order)."""
    torch.manual_seed(2)
    B, C, H = 2, 16, 4
    x = torch.randn(B, C, H, H)
    w = torch.randn(C, C, 3, 3)
    scale, shift = torch.rand(C) + 0.5, torch.randn(C)
    a = F.leaky_relu(x * scale[None, :, None, None] + shift[None, :, None, None], 0.01)
    r = torch.randn(B, C, H, H)
    y = F.conv2d(a, w, None, 1, 1) + r
    g = G.conv_like(B, H, H, C, C, 3, 1, 1)
    master = w.permute(0, 2, 3, 1).reshape(C, 9, C)
    out = E.emu_igemm(g, nhwc(x), E.emu_repack(master, g, False), pro=(scale, shift, 0.01), residual=nhwc(r))
    assert torch.allclose(nchw(out), y, atol=1e-4)
