"""Gather-GEMM geometry (sv_geom) of every conv-like layer of the SHOT-VAE step.

Two families cover everything (derivations in DESIGN.md):
  conv_like   out[oy] = sum_ky x[s*oy - pad + ky] w[ky]        (Conv2d forward, ConvTranspose2d dgrad)
  convT_like  out[oy] = sum_{iy,ky: oy = s*iy - pad + ky} ...  (ConvTranspose2d forward, Conv2d dgrad),
              split into s*s sub-pixel phases so that every output pixel only visits its valid taps.
Reference layers: shot_vae_model/wideresnet.py:13-14,29-30,34-35,41-43; shot_vae_model/decoder.py:13-58.
"""
from ._lib import MAX_PHASES, MAX_TAPS, SvGeom


def _fill_phase(ph, ooy, oox, taps, w_off):
    assert len(taps) <= MAX_TAPS, "too many taps"
    ph.ooy, ph.oox, ph.ntap, ph.w_off = ooy, oox, len(taps), w_off
    for t in range(MAX_TAPS):
        dy, dx, to = taps[t] if t < len(taps) else (0, 0, 0)
        ph.dy[t], ph.dx[t], ph.torig[t] = dy, dx, to


def conv_like(B, Hin, Win, Cin, N, k, stride, pad, ldx=None, ldo=None):
    """Regular strided convolution as one phase with (up to) k*k taps."""
    g = SvGeom()
    Ho, Wo = (Hin + 2 * pad - k) // stride + 1, (Win + 2 * pad - k) // stride + 1
    g.B, g.Hin, g.Win, g.Cin, g.ldx = B, Hin, Win, Cin, ldx or Cin
    g.Hq, g.Wq, g.sy, g.sx = Ho, Wo, stride, stride
    g.Hout, g.Wout, g.N, g.ldo, g.osy, g.osx = Ho, Wo, N, ldo or N, 1, 1
    g.T_orig, g.nphase = k * k, 1
    taps = []
    for ky in range(k):
        dy = ky - pad
        if not any(0 <= q * stride + dy < Hin for q in range(Ho)):
            continue    # this tap never lands inside the image
        for kx in range(k):
            dx = kx - pad
            if not any(0 <= q * stride + dx < Win for q in range(Wo)):
                continue
            taps.append((dy, dx, ky * k + kx))
    _fill_phase(g.phase[0], 0, 0, taps, 0)
    return g


def convT_like(B, Hin, Win, Cin, N, k, stride, pad, ldx=None, ldo=None):
    """Transposed-convolution-shaped op producing exactly stride*Hin x stride*Win outputs
    (out[oy] = sum over {iy,ky : oy = stride*iy - pad + ky}), as stride^2 sub-pixel phases: phase
    (py,px) writes output pixels (stride*q + p) and visits only the taps with
    (p + pad - ky) % stride == 0, reading input pixel q + (p + pad - ky) // stride.  Covers
    ConvTranspose2d(4,2,1) forward and the data gradient of any Conv2d whose input size is a
    multiple of its stride (3x3 s1/s2 p1, 1x1 s1/s2)."""
    g = SvGeom()
    g.B, g.Hin, g.Win, g.Cin, g.ldx = B, Hin, Win, Cin, ldx or Cin
    g.Hq, g.Wq, g.sy, g.sx = Hin, Win, 1, 1
    g.Hout, g.Wout, g.N, g.ldo, g.osy, g.osx = Hin * stride, Win * stride, N, ldo or N, stride, stride
    g.T_orig, g.nphase = k * k, stride * stride
    assert g.nphase <= MAX_PHASES
    w_off = 0
    for py in range(stride):
        for px in range(stride):
            taps = []
            for ky in range(k):
                if (py + pad - ky) % stride:
                    continue
                dy = (py + pad - ky) // stride
                if not any(0 <= q + dy < Hin for q in range(Hin)):
                    continue
                for kx in range(k):
                    if (px + pad - kx) % stride:
                        continue
                    dx = (px + pad - kx) // stride
                    if not any(0 <= q + dx < Win for q in range(Win)):
                        continue
                    taps.append((dy, dx, ky * k + kx))
            _fill_phase(g.phase[py * stride + px], py, px, taps, w_off)
            w_off += len(taps) * Cin * N
    return g


def packed_size(g):
    """Elements of the packed weight buffer of geometry g."""
    return sum(g.phase[p].ntap for p in range(g.nphase)) * g.Cin * g.N


def with_batch(g, B):
    """Copy of g for another batch size (everything else is batch independent)."""
    h = SvGeom.from_buffer_copy(g)
    h.B = B
    return h
