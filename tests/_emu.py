"""Plain-torch CPU emulation of the *semantics* of sv_repack / sv_igemm / sv_wgrad (test infra).

Used by the CPU tests to prove that the geometry tables and weight packings built by
shot_vae_amd.geometry reproduce torch's conv2d / conv_transpose2d and their gradients before any
kernel runs; the GPU tests then only have to show kernel == these semantics."""
import torch


def emu_repack(master, g, transpose):
    """master [N][T_orig][C] -> flat packed tensor (per phase [n'][ntap][c'])."""
    N, T, C = master.shape
    out = torch.zeros(sum(g.phase[p].ntap for p in range(g.nphase)) * N * C)
    for p in range(g.nphase):
        ph = g.phase[p]
        if ph.ntap == 0:
            continue
        sel = master[:, [ph.torig[t] for t in range(ph.ntap)], :]      # [N][ntap][C]
        if transpose:
            sel = sel.permute(2, 1, 0)                                   # [C][ntap][N]
        out[ph.w_off: ph.w_off + sel.numel()] = sel.reshape(-1)
    return out


def _act(x, pro):
    if pro is None:
        return x
    scale, shift, slope = pro
    u = x * scale + shift
    return torch.where(u > 0, u, u * slope)


def _gather(g, ph, t, xa):
    """activated input at tap t for all (b,qy,qx): [B,Hq,Wq,Cin], zero outside the image."""
    B = g.B
    out = torch.zeros(B, g.Hq, g.Wq, g.Cin)
    dy, dx = ph.dy[t], ph.dx[t]
    for qy in range(g.Hq):
        iy = qy * g.sy + dy
        if not 0 <= iy < g.Hin:
            continue
        for qx in range(g.Wq):
            ix = qx * g.sx + dx
            if 0 <= ix < g.Win:
                out[:, qy, qx] = xa[:, iy, ix, :g.Cin]
    return out


def emu_igemm(g, x, wp, pro=None, bias=None, residual=None):
    """x [B,Hin,Win,ldx] -> out [B,Hout,Wout,ldo] (channels >= N left zero)."""
    xa = _act(x[..., :g.Cin], pro)
    out = torch.zeros(g.B, g.Hout, g.Wout, g.ldo)
    for p in range(g.nphase):
        ph = g.phase[p]
        acc = torch.zeros(g.B, g.Hq, g.Wq, g.N)
        if ph.ntap:
            w = wp[ph.w_off: ph.w_off + g.N * ph.ntap * g.Cin].view(g.N, ph.ntap, g.Cin)
            for t in range(ph.ntap):
                acc += _gather(g, ph, t, xa) @ w[:, t, :].t()
        if bias is not None:
            acc = acc + bias
        out[:, ph.ooy::g.osy, ph.oox::g.osx, :g.N] = acc
    if residual is not None:
        out[..., :g.N] += residual[..., :g.N]
    return out


def emu_wgrad(g, x, dy, pro=None):
    """dW in the MASTER layout [N][T_orig][Cin]."""
    xa = _act(x[..., :g.Cin], pro)
    dw = torch.zeros(g.N, g.T_orig, g.Cin)
    for p in range(g.nphase):
        ph = g.phase[p]
        d = dy[:, ph.ooy::g.osy, ph.oox::g.osx, :g.N].reshape(-1, g.N)
        for t in range(ph.ntap):
            a = _gather(g, ph, t, xa).reshape(-1, g.Cin)
            dw[:, ph.torig[t], :] += d.t() @ a
    return dw
