"""sv_bwd3x3 (ABI 7): the fused backward of a narrow stride-1 3x3 convolution -- data gradient + activation backward + the two
BatchNorm-backward sums + weight gradient in ONE launch -- against the PAIR of launches it replaces (sv_igemm with the `ex`
epilogue, sv_wgrad_ex) and against torch's fp32 autograd of the same lines (wideresnet.py:27-35 backward).

Gates: the data gradient is BIT-EQUAL to sv_igemm's (same MFMA shape, same accumulation order, same epilogue arithmetic);
the two sums agree to 1e-6 (fp32 partial sums over another tile partition, added in fp64); the weight gradient agrees with
sv_wgrad_ex to 1e-4 (same bf16 operands, another summation order) and with torch fp32 on the rounded operands to 2e-3.
The two-tensor form (dy = dy_scale * dy + dy_scale2 * dy2 + dy_shift formed in the kernel's load path) is held to the same gates against
the pair run on that tensor MATERIALISED by the test (the kernel's two fused multiply-adds emulated in fp64), and -- end to end with
sv_bn_bwd_affine -- against torch's own autograd of BatchNorm2d(train) -> conv (test_two_tensor_form_against_torch_autograd)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from shot_vae_amd import _lib as L          # noqa: E402
from shot_vae_amd import geometry as G      # noqa: E402
from tests.test_kernels_gpu import ACC, bq, dev, nchw, nhwc, p, rel, repack, st      # noqa: E402

CH = 32
# B (per group), H, groups, block budget, form of dy: 0 = a tensor, 1 = two-tensor (BatchNorm backward), 2 = + residual and side output
CASES = [(16, 32, 2, 0, 0), (16, 32, 2, 0, 1), (16, 32, 2, 0, 2),
         (3, 32, 1, 0, 0), (3, 32, 1, 0, 2),      # fewer tiles than blocks
         (24, 32, 1, 0, 1),                       # the steady state of the register pipeline (> 3 tiles per block at budget 64)
         (40, 32, 4, 64, 1), (40, 32, 4, 64, 0), (40, 32, 4, 64, 2),
         (8, 16, 1, 0, 1), (8, 16, 3, 0, 0), (8, 16, 2, 0, 2),           # 16 x 16 maps: eight rows per tile
         (6, 8, 2, 0, 1), (6, 8, 1, 16, 0), (6, 8, 2, 0, 2),             # 8 x 8 maps: two images per tile (spacer rows in the LDS halo)
         (130, 32, 4, 0, 1), (130, 32, 4, 0, 2)]  # 4 x 130 images: many tiles per block, all four groups


def _inputs(B, H, Gn, lin2, seed):
    d = dev()
    torch.manual_seed(seed)
    bf = torch.bfloat16
    t = {}
    t["dy"] = torch.randn(Gn * B, H, H, CH, device=d).to(bf)
    t["c1"] = (torch.randn(Gn * B, H, H, CH, device=d) * 1.3 - 0.2).to(bf) if lin2 else None
    t["res"] = (torch.randn(Gn * B, H, H, CH, device=d) * 0.8).to(bf) if lin2 == 2 else None
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


def materialise(t, B):
    """dy_scale * dy + (dy_scale2 * dy2 + dy_shift), rounded as the kernel rounds it: two fp32 fused multiply-adds (each emulated by
    an fp64 multiply-add -- exact for these operands -- and ONE rounding to fp32), then one rounding to bf16"""
    gi = torch.arange(t["dy"].shape[0], device=t["dy"].device) // B
    ca, cb, cc = (t["coef"][k][gi][:, None, None, :].double() for k in range(3))
    inner = (t["c1"].double() * cb + cc).float()
    out = (t["dy"].double() * ca + inner.double()).float()
    if t.get("res") is not None:
        out = out + t["res"].float()             # the skip connection's gradient: a plain fp32 add behind the two multiply-adds
    return out.to(torch.bfloat16)


def _pair(t, B, H, Gn, budget, slope, R):
    """the launches sv_bwd3x3 replaces: sv_igemm (data gradient, `ex` epilogue) + sv_wgrad_ex[, on the materialised two-tensor dy]"""
    d = dev()
    dy = materialise(t, B) if t["coef"] is not None else t["dy"]
    gd = G.convT_like(B, H, H, CH, CH, 3, 1, 1)
    gf = G.conv_like(B, H, H, CH, CH, 3, 1, 1)
    wd = repack(t["w"], gd, True, "bf16")
    g = torch.full_like(t["x"], 7.0)
    bs = torch.zeros(Gn, R, 2 * CH, device=d, dtype=ACC)
    a = L.SvIgemmArgs()
    a.x, a.w, a.out, a.groups, a.block_budget = dy.data_ptr(), wd.data_ptr(), g.data_ptr(), Gn, budget
    a.ex, a.ex_scale, a.ex_shift, a.ex_mean, a.ex_rstd = (q.data_ptr() for q in (t["x"], t["sc"], t["sh"], t["mean"], t["rstd"]))
    a.ex_slope, a.bsums, a.replicas = slope, bs.data_ptr(), R
    L.call("sv_igemm", C.byref(gd), L.SV_BF16, C.byref(a), st())
    ws = torch.empty(16 * 1024 * 1024, device=d)
    dw = torch.zeros(CH, 9, CH, device=d)
    b = L.SvWgradArgs()
    b.x, b.pro_scale, b.pro_shift, b.pro_slope = t["x"].data_ptr(), t["sc"].data_ptr(), t["sh"].data_ptr(), slope
    b.dy, b.dw, b.use_tr, b.ws, b.ws_elems, b.groups, b.block_budget = (dy.data_ptr(), dw.data_ptr(), 1, ws.data_ptr(),
                                                                        ws.numel(), Gn, budget)
    L.call("sv_wgrad_ex", C.byref(gf), L.SV_BF16, C.byref(b), st())
    torch.cuda.synchronize()
    return g, bs, dw, dy, wd, gd


def _fused(t, wd, gd, Gn, budget, slope, R, ws=None):
    d = dev()
    g = torch.full_like(t["x"], 7.0)
    bs = torch.zeros(Gn, R, 2 * CH, device=d, dtype=ACC)
    dw = torch.zeros(CH, 9, CH, device=d)
    ws = ws if ws is not None else torch.full((10 * 1024 * 1024,), float("nan"), device=d)
    a = L.SvBwd3x3Args()
    a.dy, a.x, a.w, a.out = t["dy"].data_ptr(), t["x"].data_ptr(), wd.data_ptr(), g.data_ptr()
    if t.get("fold") is not None:                 # (ABI 8) the coefficients derived in the launch from the raw sums
        a.dy2 = t["c1"].data_ptr()
        bs2, R2, count, gamma2, mean2, rstd2, dgam, dbet = t["fold"]
        a.fold_bsums, a.fold_replicas, a.fold_count = bs2.data_ptr(), R2, count
        a.fold_gamma, a.fold_mean, a.fold_rstd, a.fold_dgamma, a.fold_dbeta = (q.data_ptr() for q in (gamma2, mean2, rstd2, dgam, dbet))
    elif t["coef"] is not None:
        a.dy2, a.dy_scale, a.dy_scale2, a.dy_shift = (t["c1"].data_ptr(), t["coef"][0].data_ptr(), t["coef"][1].data_ptr(),
                                                      t["coef"][2].data_ptr())
    if t.get("res") is not None:
        t["dy_out"] = torch.full_like(t["dy"], 5.0)
        a.dy3, a.dy_out = t["res"].data_ptr(), t["dy_out"].data_ptr()
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
    g_ref, bs_ref, dw_ref, dy_eff, wd, gd = _pair(t, B, H, Gn, budget, slope, R)
    g, bs, dw = _fused(t, wd, gd, Gn, budget, slope, R)
    assert bool(torch.isfinite(g.float()).all()) and bool(torch.isfinite(dw).all())
    if lin2 == 2:                                 # the side output: the formed gradient itself, every element written exactly once
        assert torch.equal(t["dy_out"], dy_eff), float((t["dy_out"].float() - dy_eff.float()).abs().max())
    assert torch.equal(g, g_ref), float((g.float() - g_ref.float()).abs().max())
    s, s_ref = bs.sum(1), bs_ref.sum(1)
    scale = float(s_ref.abs().max())
    assert float((s - s_ref).abs().max()) < 2e-6 * scale + 1e-3, float((s - s_ref).abs().max())
    assert rel(dw, dw_ref) < 1e-4, rel(dw, dw_ref)
    # a second run reproduces the first: the data gradient bit for bit, the weight gradient to the order of its float atomics
    g2, bs2, dw2 = _fused(t, wd, gd, Gn, budget, slope, R)
    assert torch.equal(g2, g) and rel(dw2, dw) < 1e-6


@pytest.mark.parametrize("B,H,Gn,lin2", [(6, 32, 2, 0), (6, 32, 2, 1), (6, 32, 2, 2), (4, 16, 1, 1), (4, 8, 2, 0), (4, 8, 1, 2)])
def test_fused_backward_against_torch_autograd(B, H, Gn, lin2):
    """against torch fp32 on the operands the kernel multiplies (dy and act(x) rounded to bf16), group by group"""
    d = dev()
    slope, R = 0.01, 2
    t = _inputs(B, H, Gn, lin2, 5000 + B + H)
    gd = G.convT_like(B, H, H, CH, CH, 3, 1, 1)
    wd = repack(t["w"], gd, True, "bf16")
    g, bs, dw = _fused(t, wd, gd, Gn, 0, slope, R)
    dy_all = materialise(t, B) if lin2 else None
    wt = t["w"].reshape(CH, 3, 3, CH).permute(0, 3, 1, 2).contiguous().to(d)
    dw_ref = torch.zeros(CH, CH, 3, 3, device=d)
    g_ref = []
    for gi in range(Gn):
        sl = slice(gi * B, (gi + 1) * B)
        dy = (dy_all[sl] if lin2 else t["dy"][sl]).float()
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
    g128 = G.convT_like(4, 8, 8, 128, 128, 3, 1, 1)
    with pytest.raises(L.ShotVaeHipError, match="32 -> 32 channels"):
        _fused(t, wd, g128, 1, 0, 0.01, 2)
    g64 = G.convT_like(4, 32, 32, 64, 64, 3, 1, 1)          # 64 channels: 16-pixel maps only
    with pytest.raises(L.ShotVaeHipError, match="64 -> 64 channels on 16-pixel maps"):
        _fused(t, wd, g64, 1, 0, 0.01, 2)
    gf = G.conv_like(4, 32, 32, CH, CH, 3, 1, 1)          # the FORWARD geometry: taps in the other order
    with pytest.raises(L.ShotVaeHipError, match="data-gradient tap"):
        _fused(t, wd, gf, 1, 0, 0.01, 2)
    with L.options(deterministic=1):
        with pytest.raises(L.ShotVaeHipError, match="deterministic"):
            _fused(t, wd, gd, 1, 0, 0.01, 2)


@pytest.mark.parametrize("B,H,Gn", [(16, 32, 2), (24, 32, 1), (8, 16, 4), (6, 8, 1)])
def test_two_tensor_form_against_torch_autograd(B, H, Gn):
    """conv1's whole backward behind norm2 in ONE launch, against TORCH's own fp32 autograd of  BatchNorm2d(train) -> [conv1's input and
    weight gradient]  (wideresnet.py:27-35 backward), not against sv_bn_bwd_apply.  Given g2 = dL/d(norm2's output) and c1 = norm2's
    raw input: sv_bn_bwd_affine turns the two sums of g2 (as the data gradient behind norm2 leaves them, R replicas) into per-channel
    coefficients (and adds dgamma / dbeta); sv_bwd3x3 reads g2, c1 and conv1's raw input once, forms dc1 = dL/dc1 on the way in, and
    produces g1 (norm1's activation backward applied), its two sums, and conv1's weight gradient."""
    d = dev()
    torch.manual_seed(1000 + B + H)
    bf = torch.bfloat16
    c = CH
    g2 = torch.randn(Gn * B, H, H, c, device=d).to(bf)
    c1 = (torch.randn(Gn * B, H, H, c, device=d) * 1.7 + 0.4).to(bf)
    tin = torch.randn(Gn * B, H, H, c, device=d).to(bf)
    gamma2 = torch.rand(c, device=d) + 0.5
    w = bq(torch.randn(c, 9, c) / (9 * c) ** 0.5, "bf16")
    sc1, sh1 = (torch.rand(Gn, c, device=d) + 0.5).contiguous(), (torch.randn(Gn, c, device=d) * 0.3).contiguous()
    mean1, rstd1 = (torch.randn(Gn, c, device=d) * 0.1).contiguous(), (torch.rand(Gn, c, device=d) + 0.5).contiguous()
    slope, eps, count = 0.01, 1e-5, float(B * H * H)
    wt = w.reshape(c, 3, 3, c).permute(0, 3, 1, 2).contiguous().to(d)
    g1_ref, dgam_ref, dbet_ref = [], torch.zeros(c, device=d), torch.zeros(c, device=d)
    dw_ref = torch.zeros(c, c, 3, 3, device=d)
    mean2, rstd2 = torch.empty(Gn, c, device=d), torch.empty(Gn, c, device=d)
    for gi in range(Gn):
        sl = slice(gi * B, (gi + 1) * B)
        xg = nchw(c1[sl].float()).requires_grad_(True)
        gam, bet = gamma2.clone().requires_grad_(True), torch.zeros(c, device=d, requires_grad=True)
        F.batch_norm(xg, None, None, gam, bet, True, 0.1, eps).backward(nchw(g2[sl].float()))
        dgam_ref += gam.grad
        dbet_ref += bet.grad
        mean2[gi] = xg.detach().mean((0, 2, 3))
        rstd2[gi] = (xg.detach().var((0, 2, 3), unbiased=False) + eps).rsqrt()
        dc1 = nchw(bq(nhwc(xg.grad), "bf16"))                 # rounded to the storage type, as the kernel's MFMA operand is
        z = torch.zeros(B, c, H, H, device=d, requires_grad=True)
        F.conv2d(z, wt, None, 1, 1).backward(dc1)
        u = tin[sl].float() * sc1[gi] + sh1[gi]
        g1_ref.append(nhwc(z.grad) * torch.where(u > 0, torch.ones_like(u), torch.full_like(u, slope)))
        act = bq(torch.where(u > 0, u, u * slope), "bf16")
        dw_ref += torch.nn.grad.conv2d_weight(nchw(act), (c, c, 3, 3), dc1, 1, 1)
    g1_ref = torch.cat(g1_ref)
    gi_ = torch.arange(Gn * B, device=d) // B
    xh2 = (c1.float() - mean2[gi_][:, None, None, :]) * rstd2[gi_][:, None, None, :]
    R = 4
    gf = g2.float().view(Gn, B * H * H, c)
    bs2 = torch.zeros(Gn, R, 2 * c, device=d, dtype=ACC)
    for r in range(R):
        rows = slice(r * (B * H * H) // R, (r + 1) * (B * H * H) // R)
        bs2[:, r, :c] = gf[:, rows].sum(1)
        bs2[:, r, c:] = (gf * xh2.view(Gn, -1, c))[:, rows].sum(1)
    coef = torch.empty(3, Gn, c, device=d)
    dgam, dbet = torch.zeros(c, device=d), torch.zeros(c, device=d)
    L.call("sv_bn_bwd_affine", p(bs2), R, c, count, p(gamma2), p(mean2), p(rstd2), p(dgam), p(dbet), p(coef[0]), p(coef[1]), p(coef[2]),
           Gn, st())
    t = {"dy": g2, "c1": c1, "x": tin, "w": w, "sc": sc1, "sh": sh1, "mean": mean1, "rstd": rstd1, "coef": coef}
    gd = G.convT_like(B, H, H, c, c, 3, 1, 1)
    wd = repack(w, gd, True, "bf16")
    g1, bs1, dw = _fused(t, wd, gd, Gn, 0, slope, R)
    # the coefficients' side effect: dgamma / dbeta of norm2 exactly as autograd has them (fp32 sums)
    assert rel(dgam, dgam_ref) < 1e-4 and rel(dbet, dbet_ref) < 1e-4
    # (ABI 8) the same launch deriving the coefficients itself from the raw sums: the same outputs (the sums' partial order differs: the
    # coefficients agree to rounding, an element of the formed gradient may round the other way), dgamma / dbeta added once
    dgam8, dbet8 = torch.zeros(c, device=d), torch.zeros(c, device=d)
    t8 = dict(t, coef=None, fold=(bs2, R, count, gamma2, mean2, rstd2, dgam8, dbet8))
    g18, bs18, dw8 = _fused(t8, wd, gd, Gn, 0, slope, R)
    assert rel(dgam8, dgam) < 1e-6 and rel(dbet8, dbet) < 1e-6
    assert rel(g18.float(), g1.float()) < 2e-3 and rel(dw8, dw) < 1e-3 and rel(bs18.sum(1).float(), bs1.sum(1).float()) < 1e-3
    # g1: the transposed convolution of ONE bf16 rounding of torch's BatchNorm backward + norm1's activation backward
    assert rel(g1.float(), g1_ref) < 2.5e-2, rel(g1.float(), g1_ref)
    got = dw.view(c, 3, 3, c).permute(0, 3, 1, 2)
    assert rel(got, dw_ref) < 4e-3, rel(got, dw_ref)          # (dc1's bf16 rounding differs from torch's in a few elements)
    xh1 = (tin.float() - mean1[gi_][:, None, None, :]) * rstd1[gi_][:, None, None, :]
    s1 = g1_ref.view(Gn, -1, c).sum(1)
    s2 = (g1_ref * xh1).view(Gn, -1, c).sum(1)
    got_s = bs1.sum(1)
    tol = 2e-2 * float(g1_ref.abs().mean()) * (B * H * H) ** 0.5 * 4          # a sum of rounding errors, not of the values
    assert float((got_s[:, :c] - s1).abs().max()) < tol and float((got_s[:, c:] - s2).abs().max()) < 3 * tol


@pytest.mark.parametrize("ch,B,H,Gn,res", [(32, 12, 32, 2, True), (32, 12, 32, 1, False), (64, 12, 16, 2, True), (64, 12, 16, 4, False)])
def test_fold_derives_the_coefficients_of_sv_bn_bwd_affine(monkeypatch, ch, B, H, Gn, res):
    """ABI 8 (sv_bwd3x3_args::fold_*): the launch that derives dy_scale / dy_scale2 / dy_shift from the raw backward sums itself against
    the same launch with the coefficients sv_bn_bwd_affine makes of those sums -- two-tensor and residual form, both kernels; dgamma /
    dbeta are added exactly once (by block 0 of every group)."""
    import sys
    monkeypatch.setattr(sys.modules[__name__], "CH", ch)
    d = dev()
    t = _inputs(B, H, Gn, 2 if res else 1, 77 + ch + Gn)
    R, count = 8, float(B * H * H)
    bs2 = (torch.randn(Gn, R, 2 * ch, device=d) * 40).to(ACC).contiguous()
    gamma2, mean2, rstd2 = torch.rand(ch, device=d) + 0.5, (torch.randn(Gn, ch, device=d) * 0.2).contiguous(), (torch.rand(Gn, ch, device=d) + 0.5).contiguous()
    coef = torch.empty(3, Gn, ch, device=d)
    dgam, dbet = torch.zeros(ch, device=d), torch.zeros(ch, device=d)
    L.call("sv_bn_bwd_affine", p(bs2), R, ch, count, p(gamma2), p(mean2), p(rstd2), p(dgam), p(dbet), p(coef[0]), p(coef[1]), p(coef[2]), Gn, st())
    gd = G.convT_like(B, H, H, ch, ch, 3, 1, 1)
    wd = repack(t["w"], gd, True, "bf16")
    ta = dict(t, coef=coef)
    g_a, bs_a, dw_a = _fused(ta, wd, gd, Gn, 0, 0.01, 4)
    out_a = ta.get("dy_out")
    dgam8, dbet8 = torch.zeros(ch, device=d), torch.zeros(ch, device=d)
    tb = dict(t, coef=None, fold=(bs2, R, count, gamma2, mean2, rstd2, dgam8, dbet8))
    g_b, bs_b, dw_b = _fused(tb, wd, gd, Gn, 0, 0.01, 4)
    assert rel(dgam8, dgam) < 1e-6 and rel(dbet8, dbet) < 1e-6
    assert rel(g_b.float(), g_a.float()) < 2e-3 and rel(dw_b, dw_a) < 1e-3 and rel(bs_b.sum(1).float(), bs_a.sum(1).float()) < 1e-3
    if res:
        assert rel(tb["dy_out"].float(), out_a.float()) < 2e-3


# ---- 64 channels on 16 x 16 maps (bwd3x3g.hip): the same checks with the module's channel count switched ------------------------------
CASES64 = [(16, 16, 2, 0, 0), (16, 16, 2, 0, 1), (16, 16, 2, 0, 2),
           (1, 16, 1, 0, 1), (3, 16, 1, 0, 2),        # fewer tiles than blocks (4 / 12 tiles)
           (40, 16, 4, 64, 1), (40, 16, 4, 64, 0), (40, 16, 4, 64, 2),
           (70, 16, 1, 0, 1), (33, 16, 2, 24, 2),     # odd tile counts per block: the tail behind the pairs
           (130, 16, 4, 0, 1), (130, 16, 4, 0, 2), (130, 16, 4, 0, 0)]


@pytest.fixture
def ch64(monkeypatch):
    import sys
    monkeypatch.setattr(sys.modules[__name__], "CH", 64)


@pytest.mark.parametrize("B,H,Gn,budget,lin2", CASES64)
def test_fused_backward_64_channels_equals_the_pair_it_replaces(ch64, B, H, Gn, budget, lin2):
    test_fused_backward_equals_the_pair_it_replaces(B, H, Gn, budget, lin2)


@pytest.mark.parametrize("B,H,Gn,lin2", [(6, 16, 2, 0), (6, 16, 2, 1), (6, 16, 2, 2), (20, 16, 1, 1)])
def test_fused_backward_64_channels_against_torch_autograd(ch64, B, H, Gn, lin2):
    test_fused_backward_against_torch_autograd(B, H, Gn, lin2)


@pytest.mark.parametrize("B,H,Gn", [(16, 16, 2), (24, 16, 1), (8, 16, 4)])
def test_two_tensor_form_64_channels_against_torch_autograd(ch64, B, H, Gn):
    test_two_tensor_form_against_torch_autograd(B, H, Gn)
