"""sv_bwd3x3 (ABI 7): the fused backward of a narrow stride-1 3x3 convolution -- data gradient + activation backward + the two
BatchNorm-backward sums + weight gradient in ONE launch -- against the PAIR of launches it replaces (sv_igemm with the `ex`
epilogue, sv_wgrad_ex) and against torch's fp32 autograd of the same lines (wideresnet.py:27-35 backward).

Gates: the data gradient is BIT-EQUAL to sv_igemm's (same MFMA shape, same accumulation order, same epilogue arithmetic);
the two sums agree to 1e-6 (fp32 partial sums over another tile partition, added in fp64); the weight gradient agrees with
sv_wgrad_ex to 1e-4 (same bf16 operands, another summation order) and with torch fp32 on the rounded operands to 2e-3."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from shot_vae_amd import _lib as L          # noqa: E402
from shot_vae_amd import geometry as G      # noqa: E402
from tests.test_kernels_gpu import ACC, bq, dev, nchw, nhwc, p, rel, repack, st      # noqa: E402

CH = 32
# B (per group), H, groups, block budget, two-tensor dy
CASES = [(16, 32, 2, 0, False), (16, 32, 2, 0, True),
         (3, 32, 1, 0, False),                    # fewer tiles than blocks
         (24, 32, 1, 0, True),                    # the steady state of the register pipeline (> 3 tiles per block at budget 64)
         (40, 32, 4, 64, True), (40, 32, 4, 64, False),
         (8, 16, 1, 0, True), (8, 16, 3, 0, False),          # 16 x 16 maps: eight rows per tile
         (6, 8, 2, 0, True), (6, 8, 1, 16, False),           # 8 x 8 maps: two images per tile (spacer rows in the LDS halo)
         (130, 32, 4, 0, True)]                   # 4 x 130 images: many tiles per block, all four groups


def _inputs(B, H, Gn, lin2, seed):
    d = dev()
    torch.manual_seed(seed)
    bf = torch.bfloat16
    t = {}
    t["dy"] = torch.randn(Gn * B, H, H, CH, device=d).to(bf)
    t["c1"] = (torch.randn(Gn * B, H, H, CH, device=d) * 1.3 - 0.2).to(bf) if lin2 else None
    t["x"] = (torch.randn(Gn * B, H, H, CH, device=d) * 0.9 + 0.1).to(bf)
    t["w"] = bq(torch.randn(CH, 9, CH) / (9 * CH) ** 0.5, "bf16")          # master weights [N][tap][Cin]
    t["sc"] = (torch.rand(Gn, CH, device=d) + 0.5).contiguous()
    t["sh"] = (torch.randn(Gn, CH, device=d) * 0.3).contiguous()
    t["mean"] = (torch.randn(Gn, CH, device=d) * 0.1).contiguous()
    t["rstd"] = (torch.rand(Gn, CH, device=d) + 0.5).contiguous()
    t["coef"] = None
    if lin2:
        coef = torch.empty(3, Gn, CH, device=d)
        coef[0] = torch.rand(Gn, CH, device=d) + 0.5
        coef[1] = torch.randn(Gn, CH, device=d) * 0.2
        coef[2] = torch.randn(Gn, CH, device=d) * 0.05
        t["coef"] = coef.contiguous()
    return t


def _pair(t, B, H, Gn, budget, slope, R):
    """the launches sv_bwd3x3 replaces: sv_igemm (data gradient, `ex` epilogue[, two-tensor prologue]) + sv_wgrad_ex"""
    d = dev()
    gd = G.convT_like(B, H, H, CH, CH, 3, 1, 1)
    gf = G.conv_like(B, H, H, CH, CH, 3, 1, 1)
    wd = repack(t["w"], gd, True, "bf16")
    g = torch.full_like(t["x"], 7.0)
    bs = torch.zeros(Gn, R, 2 * CH, device=d, dtype=ACC)
    a = L.SvIgemmArgs()
    a.x, a.w, a.out, a.groups, a.block_budget = t["dy"].data_ptr(), wd.data_ptr(), g.data_ptr(), Gn, budget
    po = None
    if t["coef"] is not None:
        po = torch.empty_like(t["dy"])
        a.pro_scale, a.pro_scale2, a.pro_shift, a.pro_slope = (t["coef"][0].data_ptr(), t["coef"][1].data_ptr(),
                                                               t["coef"][2].data_ptr(), 1.0)
        a.x2, a.pro_out = t["c1"].data_ptr(), po.data_ptr()
    a.ex, a.ex_scale, a.ex_shift, a.ex_mean, a.ex_rstd = (q.data_ptr() for q in (t["x"], t["sc"], t["sh"], t["mean"], t["rstd"]))
    a.ex_slope, a.bsums, a.replicas = slope, bs.data_ptr(), R
    L.call("sv_igemm", C.byref(gd), L.SV_BF16, C.byref(a), st())
    ws = torch.empty(16 * 1024 * 1024, device=d)
    dw = torch.zeros(CH, 9, CH, device=d)
    b = L.SvWgradArgs()
    b.x, b.pro_scale, b.pro_shift, b.pro_slope = t["x"].data_ptr(), t["sc"].data_ptr(), t["sh"].data_ptr(), slope
    b.dy, b.dw, b.use_tr, b.ws, b.ws_elems, b.groups, b.block_budget = (t["dy"].data_ptr(), dw.data_ptr(), 1, ws.data_ptr(),
                                                                        ws.numel(), Gn, budget)
    if t["coef"] is not None:
        b.dy2, b.dy_scale, b.dy_scale2, b.dy_shift = (t["c1"].data_ptr(), t["coef"][0].data_ptr(), t["coef"][1].data_ptr(),
                                                      t["coef"][2].data_ptr())
    L.call("sv_wgrad_ex", C.byref(gf), L.SV_BF16, C.byref(b), st())
    torch.cuda.synchronize()
    return g, bs, dw, po, wd, gd


def _fused(t, wd, gd, Gn, budget, slope, R, ws=None):
    d = dev()
    g = torch.full_like(t["x"], 7.0)
    bs = torch.zeros(Gn, R, 2 * CH, device=d, dtype=ACC)
    dw = torch.zeros(CH, 9, CH, device=d)
    ws = ws if ws is not None else torch.full((4 * 1024 * 1024,), float("nan"), device=d)
    a = L.SvBwd3x3Args()
    a.dy, a.x, a.w, a.out = t["dy"].data_ptr(), t["x"].data_ptr(), wd.data_ptr(), g.data_ptr()
    if t["coef"] is not None:
        a.dy2, a.dy_scale, a.dy_scale2, a.dy_shift = (t["c1"].data_ptr(), t["coef"][0].data_ptr(), t["coef"][1].data_ptr(),
                                                      t["coef"][2].data_ptr())
    a.x_scale, a.x_shift, a.x_mean, a.x_rstd, a.x_slope = (t["sc"].data_ptr(), t["sh"].data_ptr(), t["mean"].data_ptr(),
                                                           t["rstd"].data_ptr(), slope)
    a.bsums, a.replicas, a.groups, a.dw, a.ws, a.ws_elems, a.block_budget = (bs.data_ptr(), R, Gn, dw.data_ptr(), ws.data_ptr(),
                                                                             ws.numel(), budget)
    L.call("sv_bwd3x3", C.byref(gd), L.SV_BF16, C.byref(a), st())
    torch.cuda.synchronize()
    return g, bs, dw


@pytest.mark.parametrize("B,H,Gn,budget,lin2", CASES)
def test_fused_backward_equals_the_pair_it_replaces(B, H, Gn, budget, lin2):
    slope, R = 0.01, 4
    t = _inputs(B, H, Gn, lin2, 4000 + B + H + Gn)
    g_ref, bs_ref, dw_ref, po, wd, gd = _pair(t, B, H, Gn, budget, slope, R)
    g, bs, dw = _fused(t, wd, gd, Gn, budget, slope, R)
    assert bool(torch.isfinite(g.float()).all()) and bool(torch.isfinite(dw).all())
    assert torch.equal(g, g_ref), float((g.float() - g_ref.float()).abs().max())
    s, s_ref = bs.sum(1), bs_ref.sum(1)
    scale = float(s_ref.abs().max())
    assert float((s - s_ref).abs().max()) < 2e-6 * scale + 1e-3, float((s - s_ref).abs().max())
    assert rel(dw, dw_ref) < 1e-4, rel(dw, dw_ref)
    # a second run reproduces the first: the data gradient bit for bit, the weight gradient to the order of its float atomics
    g2, bs2, dw2 = _fused(t, wd, gd, Gn, budget, slope, R)
    assert torch.equal(g2, g) and rel(dw2, dw) < 1e-6


@pytest.mark.parametrize("B,H,Gn,lin2", [(6, 32, 2, False), (6, 32, 2, True), (4, 16, 1, True), (4, 8, 2, False)])
def test_fused_backward_against_torch_autograd(B, H, Gn, lin2):
    """against torch fp32 on the operands the kernel multiplies (dy and act(x) rounded to bf16), group by group"""
    d = dev()
    slope, R = 0.01, 2
    t = _inputs(B, H, Gn, lin2, 5000 + B + H)
    gd = G.convT_like(B, H, H, CH, CH, 3, 1, 1)
    wd = repack(t["w"], gd, True, "bf16")
    g, bs, dw = _fused(t, wd, gd, Gn, 0, slope, R)
    wt = t["w"].reshape(CH, 3, 3, CH).permute(0, 3, 1, 2).contiguous().to(d)
    dw_ref = torch.zeros(CH, CH, 3, 3, device=d)
    g_ref = []
    for gi in range(Gn):
        sl = slice(gi * B, (gi + 1) * B)
        dy = t["dy"][sl].float()
        if lin2:
            dy = dy * t["coef"][0][gi] + (t["c1"][sl].float() * t["coef"][1][gi] + t["coef"][2][gi])
        dy = bq(dy, "bf16")
        u = t["x"][sl].float() * t["sc"][gi] + t["sh"][gi]
        act = bq(torch.where(u > 0, u, u * slope), "bf16")
        dw_ref += torch.nn.grad.conv2d_weight(nchw(act), (CH, CH, 3, 3), nchw(dy), 1, 1)
        z = torch.zeros(B, CH, H, H, device=d, requires_grad=True)
        F.conv2d(z, wt, None, 1, 1).backward(nchw(dy))
        g_ref.append(nhwc(z.grad) * torch.where(u > 0, torch.ones_like(u), torch.full_like(u, slope)))
    g_ref = torch.cat(g_ref)
    assert rel(g.float(), g_ref) < 1e-2, rel(g.float(), g_ref)
    got = dw.view(CH, 3, 3, CH).permute(0, 3, 1, 2)
    assert rel(got, dw_ref) < 2e-3, rel(got, dw_ref)
    gi_ = torch.arange(Gn * B, device=d) // B
    xh = (t["x"].float() - t["mean"][gi_][:, None, None, :]) * t["rstd"][gi_][:, None, None, :]
    s1 = g_ref.view(Gn, -1, CH).sum(1)
    s2 = (g_ref * xh).view(Gn, -1, CH).sum(1)
    got_s = bs.sum(1)
    tol = 2e-2 * float(g_ref.abs().mean()) * (B * H * H) ** 0.5 * 4
    assert float((got_s[:, :CH] - s1).abs().max()) < tol and float((got_s[:, CH:] - s2).abs().max()) < 3 * tol


def test_fused_backward_argument_checks():
    d = dev()
    t = _inputs(4, 32, 1, False, 1)
    gd = G.convT_like(4, 32, 32, CH, CH, 3, 1, 1)
    wd = repack(t["w"], gd, True, "bf16")
    with pytest.raises(L.ShotVaeHipError, match="workspace"):
        _fused(t, wd, gd, 1, 0, 0.01, 2, ws=torch.empty(1024, device=d))
    g64 = G.convT_like(4, 16, 16, 64, 64, 3, 1, 1)
    with pytest.raises(L.ShotVaeHipError, match="32 input and 32 output"):
        _fused(t, wd, g64, 1, 0, 0.01, 2)
    gf = G.conv_like(4, 32, 32, CH, CH, 3, 1, 1)          # the FORWARD geometry: taps in the other order
    with pytest.raises(L.ShotVaeHipError, match="data-gradient tap"):
        _fused(t, wd, gf, 1, 0, 0.01, 2)
    with L.options(deterministic=1):
        with pytest.raises(L.ShotVaeHipError, match="deterministic"):
            _fused(t, wd, gd, 1, 0, 0.01, 2)
