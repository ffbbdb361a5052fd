"""sv_bwd3x3 alone at the headline size (4 x 512 images, 32 channels, 32 x 32): time, GB/s of its algorithmic bytes (3 passes, 4 in the
two-tensor form), beside the launches it replaces run back to back on one stream (sv_bn_bwd_apply + sv_igemm + sv_wgrad_ex).

    python tools/bwdf_bench.py [B] [H] [groups] [budget] [channels]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shot_vae_amd import _lib as L          # noqa: E402
from shot_vae_amd import geometry as G      # noqa: E402
import tests.test_fused_bwd_gpu as T        # noqa: E402
from tests.test_fused_bwd_gpu import _fused, _inputs, _pair      # noqa: E402

CH = int(sys.argv[5]) if len(sys.argv) > 5 else 32          # 64: bwd3x3g.hip (16 x 16 maps)
T.CH = CH


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    Gn = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    budget = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    d = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    tensor_bytes = Gn * B * H * H * CH * 2
    for lin2 in (0, 1, 2):
        t = _inputs(B, H, Gn, lin2, 7)
        g_ref, bs_ref, dw_ref, _, wd, gd = _pair(t, B, H, Gn, 0, 0.01, 4)
        ws = torch.empty(10 * 1024 * 1024, device=d)
        g, bs, dw = _fused(t, wd, gd, Gn, budget, 0.01, 4, ws=ws)
        ok = torch.equal(g, g_ref)
        err = float((dw - dw_ref).abs().max() / dw_ref.abs().max())
        us = timed(lambda: _fused_nosync(t, wd, gd, Gn, budget, ws))
        passes = (3, 4, 6)[lin2]
        us_pair = timed(lambda: _pair_nosync(t, B, H, Gn, wd, gd))
        print("%s  fused %7.1f us  %6.0f GB/s of %d passes (%.0f MB)   pair back to back %7.1f us   g bit-equal %s  dw rel %.1e"
              % (("plain     ", "two-tensor", "residual  ")[lin2], us, passes * tensor_bytes / us / 1e3, passes, passes * tensor_bytes / 1e6,
                 us_pair, ok, err))


_bufs = {}


def _fused_nosync(t, wd, gd, Gn, budget, ws):
    key = ("f", t["coef"] is not None, t.get("res") is not None)
    if key not in _bufs:
        d = t["x"].device
        _bufs[key] = (torch.empty_like(t["x"]), torch.zeros(Gn, 4, 2 * CH, device=d, dtype=torch.float64), torch.zeros(CH, 9, CH, device=d))
    g, bs, dw = _bufs[key]
    a = L.SvBwd3x3Args()
    a.dy, a.x, a.w, a.out = t["dy"].data_ptr(), t["x"].data_ptr(), wd.data_ptr(), g.data_ptr()
    if t["coef"] is not None:
        a.dy2, a.dy_scale, a.dy_scale2, a.dy_shift = (t["c1"].data_ptr(), t["coef"][0].data_ptr(), t["coef"][1].data_ptr(),
                                                      t["coef"][2].data_ptr())
    if t.get("res") is not None:
        a.dy3, a.dy_out = t["res"].data_ptr(), t["dy_out"].data_ptr()
    a.x_scale, a.x_shift, a.x_mean, a.x_rstd, a.x_slope = (t["sc"].data_ptr(), t["sh"].data_ptr(), t["mean"].data_ptr(),
                                                           t["rstd"].data_ptr(), 0.01)
    a.bsums, a.replicas, a.groups, a.dw, a.ws, a.ws_elems, a.block_budget = (bs.data_ptr(), 4, Gn, dw.data_ptr(), ws.data_ptr(),
                                                                             ws.numel(), budget)
    L.call("sv_bwd3x3", C.byref(gd), L.SV_BF16, C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream))


def _pair_nosync(t, B, H, Gn, wd, gd):
    """what the step runs for such a layer today, on ONE stream: [sv_bn_bwd_apply], data gradient, weight gradient"""
    d = t["x"].device
    key = ("p", t["coef"] is not None, t.get("res") is not None)
    if key not in _bufs:
        _bufs[key] = (torch.empty_like(t["x"]), torch.zeros(Gn, 4, 2 * CH, device=d, dtype=torch.float64), torch.zeros(CH, 9, CH, device=d),
                      torch.empty(16 * 1024 * 1024, device=d), torch.empty_like(t["x"]), G.conv_like(B, H, H, CH, CH, 3, 1, 1))
    g, bs, dw, ws, dc1, gf = _bufs[key]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    dy = t["dy"]
    if t["coef"] is not None:
        # the streaming pass the two-tensor form removes (same bytes as sv_bn_bwd_apply: two reads, one write)
        arr = (L.SvBnBranch * 1)()
        arr[0].g, arr[0].bsums, arr[0].gamma, arr[0].replicas = t["dy"].data_ptr(), bs.data_ptr(), t["sc"].data_ptr(), 4
        arr[0].dgamma = arr[0].dbeta = None
        L.call("sv_bn_bwd_apply", L.SV_BF16, Gn * 0 + t["x"].numel() // CH // Gn, CH, CH, C.c_void_p(t["c1"].data_ptr()),
               C.c_void_p(t["mean"].data_ptr()), C.c_void_p(t["rstd"].data_ptr()), float(B * H * H), arr, 1,
               C.c_void_p(t["res"].data_ptr()) if t.get("res") is not None else None, C.c_void_p(dc1.data_ptr()), Gn, st)
        dy = dc1
    a = L.SvIgemmArgs()
    a.x, a.w, a.out, a.groups = dy.data_ptr(), wd.data_ptr(), g.data_ptr(), Gn
    a.ex, a.ex_scale, a.ex_shift, a.ex_mean, a.ex_rstd = (q.data_ptr() for q in (t["x"], t["sc"], t["sh"], t["mean"], t["rstd"]))
    a.ex_slope, a.bsums, a.replicas = 0.01, bs.data_ptr(), 4
    L.call("sv_igemm", C.byref(gd), L.SV_BF16, C.byref(a), st)
    b = L.SvWgradArgs()
    b.x, b.pro_scale, b.pro_shift, b.pro_slope = t["x"].data_ptr(), t["sc"].data_ptr(), t["sh"].data_ptr(), 0.01
    b.dy, b.dw, b.use_tr, b.ws, b.ws_elems, b.groups = dy.data_ptr(), dw.data_ptr(), 1, ws.data_ptr(), ws.numel(), Gn
    L.call("sv_wgrad_ex", C.byref(gf), L.SV_BF16, C.byref(b), st)


if __name__ == "__main__":
    main()
