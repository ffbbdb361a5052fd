"""Micro-benchmark of single conv-like layers through the C ABI (MI355X): forward with the fused BN prologue /
residual / statistics epilogue, data gradient with the activation-backward epilogue, weight gradient.

    python tools/layer_bench.py                 # the WRN-28-10 body shapes of SURVEY.md §8d at B=512
    python tools/layer_bench.py 512 32 32 32    # B C H N: any stride-1 3x3 layer

Prints microseconds per launch, TFLOP/s and the fraction of the 2.5 PFLOP/s dense bf16 MFMA peak."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from shot_vae_amd import _lib as L          # noqa: E402
from shot_vae_amd import geometry as G      # noqa: E402

PEAK = 2.5e15


def timed(fn, iters=int(os.environ.get("SV_BENCH_ITERS", "20")), warm=int(os.environ.get("SV_BENCH_WARM", "3"))):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def bench_layer(B, Cin, H, N, what=("fwd", "dgrad", "wgrad")):
    KS, STR = int(os.environ.get("SV_BENCH_K", "3")), int(os.environ.get("SV_BENCH_S", "1"))     # kernel size, stride
    if os.environ.get("SV_BENCH_T"):
        return bench_convT_layer(B, Cin, H, N, what)
    if KS != 3 or STR != 1 or os.environ.get("SV_BENCH_NOPRO"):
        return bench_odd_layer(B, Cin, H, N, KS, STR, what)
    d = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    bf = torch.bfloat16
    x = torch.randn(B, H, H, Cin, device=d).to(bf)
    dy = torch.randn(B, H, H, N, device=d).to(bf)
    master = (torch.randn(N, 9, Cin, device=d) / (9 * Cin) ** 0.5).contiguous()
    flops = 2.0 * B * H * H * 9 * Cin * N
    R = int(os.environ.get("SV_BENCH_R", "8"))
    res = {}
    if "fwd" in what:
        g = G.conv_like(B, H, H, Cin, N, 3, 1, 1)
        wp = torch.zeros(G.packed_size(g), dtype=bf, device=d)
        L.call("sv_repack", L.SV_BF16, C.c_void_p(master.data_ptr()), N, 9, Cin, 0, C.byref(g), C.c_void_p(wp.data_ptr()), st)
        out = torch.empty(B, H, H, N, dtype=bf, device=d)
        resid = torch.randn(B, H, H, N, device=d).to(bf)
        sc, sh = torch.rand(Cin, device=d) + 0.5, torch.randn(Cin, device=d) * 0.3
        stats = torch.zeros(R, 2 * N, device=d, dtype=torch.float64)
        a = L.SvIgemmArgs()
        a.x, a.w, a.out, a.residual = x.data_ptr(), wp.data_ptr(), out.data_ptr(), resid.data_ptr()
        a.pro_scale, a.pro_shift, a.pro_slope = sc.data_ptr(), sh.data_ptr(), 0.01
        a.stats, a.replicas = stats.data_ptr(), R
        res["fwd"] = timed(lambda: L.call("sv_igemm", C.byref(g), L.SV_BF16, C.byref(a), st))
    if "dgrad" in what:
        g = G.convT_like(B, H, H, N, Cin, 3, 1, 1)
        wp = torch.zeros(G.packed_size(g), dtype=bf, device=d)
        L.call("sv_repack", L.SV_BF16, C.c_void_p(master.data_ptr()), N, 9, Cin, 1, C.byref(g), C.c_void_p(wp.data_ptr()), st)
        out = torch.empty(B, H, H, Cin, dtype=bf, device=d)
        vec = [torch.rand(Cin, device=d) + 0.5 for _ in range(4)]
        bs = torch.zeros(R, 2 * Cin, device=d, dtype=torch.float64)
        a = L.SvIgemmArgs()
        a.x, a.w, a.out = dy.data_ptr(), wp.data_ptr(), out.data_ptr()
        a.ex, a.ex_scale, a.ex_shift, a.ex_mean, a.ex_rstd = [t.data_ptr() for t in [x] + vec]
        a.ex_slope, a.bsums, a.replicas = 0.01, bs.data_ptr(), R
        res["dgrad"] = timed(lambda: L.call("sv_igemm", C.byref(g), L.SV_BF16, C.byref(a), st))
    if "wgrad" in what:
        g = G.conv_like(B, H, H, Cin, N, 3, 1, 1)
        dw = torch.zeros(N, 9, Cin, device=d)
        sc, sh = torch.rand(Cin, device=d) + 0.5, torch.randn(Cin, device=d) * 0.3
        ws = torch.empty(16 << 20, device=d)
        res["wgrad"] = timed(lambda: L.call("sv_wgrad", C.byref(g), L.SV_BF16, C.c_void_p(x.data_ptr()),
                                            C.c_void_p(sc.data_ptr()), C.c_void_p(sh.data_ptr()), C.c_float(0.01),
                                            C.c_void_p(dy.data_ptr()), C.c_void_p(dw.data_ptr()), 0, 1,
                                            C.c_void_p(ws.data_ptr()), ws.numel(), 1, st))
    for form, kind in enumerate(("bwd", "bwd2", "bwd3")):
        # sv_bwd3x3, the fused backward (32 / 64 channels): dy a tensor / two-tensor form / residual form; one group of B images
        if kind not in what:
            continue
        assert (Cin == 32 and N == 32) or (Cin == 64 and N == 64 and H == 16), "sv_bwd3x3: 32 -> 32 channels, or 64 -> 64 on 16 x 16 maps"
        g = G.convT_like(B, H, H, N, Cin, 3, 1, 1)
        wp = torch.zeros(G.packed_size(g), dtype=bf, device=d)
        L.call("sv_repack", L.SV_BF16, C.c_void_p(master.data_ptr()), N, 9, Cin, 1, C.byref(g), C.c_void_p(wp.data_ptr()), st)
        out, dyo = torch.empty(B, H, H, Cin, dtype=bf, device=d), torch.empty(B, H, H, N, dtype=bf, device=d)
        y2, r3 = torch.randn(B, H, H, N, device=d).to(bf), torch.randn(B, H, H, N, device=d).to(bf)
        vec = [torch.rand(Cin, device=d) + 0.5 for _ in range(4)]
        cof = [torch.rand(N, device=d) + 0.5, torch.randn(N, device=d) * 0.2, torch.randn(N, device=d) * 0.05]
        bs = torch.zeros(R, 2 * Cin, device=d, dtype=torch.float64)
        dw, ws = torch.zeros(N, 9, Cin, device=d), torch.empty(10 << 20, device=d)
        a = L.SvBwd3x3Args()
        a.dy, a.x, a.w, a.out = dy.data_ptr(), x.data_ptr(), wp.data_ptr(), out.data_ptr()
        if form >= 1:
            a.dy2, a.dy_scale, a.dy_scale2, a.dy_shift = y2.data_ptr(), cof[0].data_ptr(), cof[1].data_ptr(), cof[2].data_ptr()
        if form == 2:
            a.dy3, a.dy_out = r3.data_ptr(), dyo.data_ptr()
        a.x_scale, a.x_shift, a.x_mean, a.x_rstd, a.x_slope = [t.data_ptr() for t in vec] + [0.01]
        a.bsums, a.replicas, a.groups, a.dw, a.ws, a.ws_elems = bs.data_ptr(), R, 1, dw.data_ptr(), ws.data_ptr(), ws.numel()
        a.block_budget = int(os.environ.get("SV_BENCH_FUSED_BLOCKS", "248"))
        res[kind] = timed(lambda: L.call("sv_bwd3x3", C.byref(g), L.SV_BF16, C.byref(a), st))
    for k, us in res.items():
        fl = flops * (2 if k.startswith("bwd") else 1)
        print(f"B={B} Cin={Cin} N={N} H={H} {k:6s} {us:9.1f} us  {fl / us / 1e6:8.1f} TFLOP/s  "
              f"{fl / us / 1e-6 / PEAK:6.3f} of bf16 MFMA peak", flush=True)
    return res


def bench_odd_layer(B, Cin, H, N, KS, STR, what):
    """The other conv layers (stride 2, 1x1): SV_BENCH_K / SV_BENCH_S; H = input size."""
    d = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    bf = torch.bfloat16
    pad = KS // 2
    Ho = (H + 2 * pad - KS) // STR + 1
    T = KS * KS
    x = torch.randn(B, H, H, Cin, device=d).to(bf)
    dy = torch.randn(B, Ho, Ho, N, device=d).to(bf)
    master = (torch.randn(N, T, Cin, device=d) / (T * Cin) ** 0.5).contiguous()
    flops = 2.0 * B * Ho * Ho * T * Cin * N
    R = int(os.environ.get("SV_BENCH_R", "8"))
    res = {}
    if "fwd" in what:
        g = G.conv_like(B, H, H, Cin, N, KS, STR, pad)
        wp = torch.zeros(G.packed_size(g), dtype=bf, device=d)
        L.call("sv_repack", L.SV_BF16, C.c_void_p(master.data_ptr()), N, T, Cin, 0, C.byref(g), C.c_void_p(wp.data_ptr()), st)
        out = torch.empty(B, Ho, Ho, N, dtype=bf, device=d)
        sc, sh = torch.rand(Cin, device=d) + 0.5, torch.randn(Cin, device=d) * 0.3
        stats = torch.zeros(R, 2 * N, device=d, dtype=torch.float64)
        a = L.SvIgemmArgs()
        a.x, a.w, a.out = x.data_ptr(), wp.data_ptr(), out.data_ptr()
        if os.environ.get("SV_BENCH_NOPRO"):            # the stem: bias, no BatchNorm prologue
            bias = torch.randn(N, device=d)
            a.bias = bias.data_ptr()
        else:
            a.pro_scale, a.pro_shift, a.pro_slope = sc.data_ptr(), sh.data_ptr(), 0.01
        a.replicas = R
        if KS != 1:                                      # (the 1x1 shortcuts of the step have no BatchNorm behind them: no statistics)
            a.stats = stats.data_ptr()
        res["fwd"] = timed(lambda: L.call("sv_igemm", C.byref(g), L.SV_BF16, C.byref(a), st))
    if "dgrad" in what:
        g = G.convT_like(B, Ho, Ho, N, Cin, KS, STR, pad)
        wp = torch.zeros(G.packed_size(g), dtype=bf, device=d)
        L.call("sv_repack", L.SV_BF16, C.c_void_p(master.data_ptr()), N, T, Cin, 1, C.byref(g), C.c_void_p(wp.data_ptr()), st)
        out = torch.empty(B, H, H, Cin, dtype=bf, device=d)
        vec = [torch.rand(Cin, device=d) + 0.5 for _ in range(4)]
        bs = torch.zeros(R, 2 * Cin, device=d, dtype=torch.float64)
        a = L.SvIgemmArgs()
        a.x, a.w, a.out = dy.data_ptr(), wp.data_ptr(), out.data_ptr()
        a.ex, a.ex_scale, a.ex_shift, a.ex_mean, a.ex_rstd = [t.data_ptr() for t in [x] + vec]
        a.ex_slope, a.bsums, a.replicas = 0.01, bs.data_ptr(), R
        res["dgrad"] = timed(lambda: L.call("sv_igemm", C.byref(g), L.SV_BF16, C.byref(a), st))
    if "wgrad" in what:
        g = G.conv_like(B, H, H, Cin, N, KS, STR, pad)
        dw = torch.zeros(N, T, Cin, device=d)
        sc, sh = torch.rand(Cin, device=d) + 0.5, torch.randn(Cin, device=d) * 0.3
        ws = torch.empty(16 << 20, device=d)
        res["wgrad"] = timed(lambda: L.call("sv_wgrad", C.byref(g), L.SV_BF16, C.c_void_p(x.data_ptr()),
                                            C.c_void_p(sc.data_ptr()), C.c_void_p(sh.data_ptr()), C.c_float(0.01),
                                            C.c_void_p(dy.data_ptr()), C.c_void_p(dw.data_ptr()), 0, 1,
                                            C.c_void_p(ws.data_ptr()), ws.numel(), 1, st))
    for k, us in res.items():
        print(f"B={B} Cin={Cin} N={N} H={H} k={KS} s={STR} {k:6s} {us:9.1f} us  {flops / us / 1e6:8.1f} TFLOP/s  "
              f"{flops / us / 1e-6 / PEAK:6.3f} of bf16 MFMA peak", flush=True)
    return res


def bench_convT_layer(B, Cin, H, N, what):
    """The decoder's ConvTranspose2d(4, 2, 1) layers (SV_BENCH_T=1): H = INPUT size, output 2H; forward with the BatchNorm +
    ReLU prologue and the next BatchNorm's statistics, data gradient (a 4x4 stride-2 convolution of dy) with the
    activation-backward epilogue, weight gradient."""
    d = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    bf = torch.bfloat16
    Ho = 2 * H
    x = torch.randn(B, H, H, Cin, device=d).to(bf)
    dy = torch.randn(B, Ho, Ho, N, device=d).to(bf)
    master = (torch.randn(N, 16, Cin, device=d) / (4 * Cin) ** 0.5).contiguous()
    flops = 2.0 * B * H * H * 16 * Cin * N          # the PyTorch count
    R = int(os.environ.get("SV_BENCH_R", "8"))
    res = {}
    gf = G.convT_like(B, H, H, Cin, N, 4, 2, 1)
    if "fwd" in what:
        wp = torch.zeros(G.packed_size(gf), dtype=bf, device=d)
        L.call("sv_repack", L.SV_BF16, C.c_void_p(master.data_ptr()), N, 16, Cin, 0, C.byref(gf), C.c_void_p(wp.data_ptr()), st)
        out = torch.empty(B, Ho, Ho, N, dtype=bf, device=d)
        sc, sh = torch.rand(Cin, device=d) + 0.5, torch.randn(Cin, device=d) * 0.3
        stats = torch.zeros(R, 2 * N, device=d, dtype=torch.float64)
        a = L.SvIgemmArgs()
        a.x, a.w, a.out = x.data_ptr(), wp.data_ptr(), out.data_ptr()
        if not os.environ.get("SV_BENCH_NOPRO"):      # (set: the materialised activations of the small decoder layers)
            a.pro_scale, a.pro_shift, a.pro_slope = sc.data_ptr(), sh.data_ptr(), 0.0
        a.stats, a.replicas = stats.data_ptr(), R
        res["fwd"] = timed(lambda: L.call("sv_igemm", C.byref(gf), L.SV_BF16, C.byref(a), st))
    if "dgrad" in what:
        g = G.conv_like(B, Ho, Ho, N, Cin, 4, 2, 1)
        wp2 = torch.zeros(G.packed_size(g), dtype=bf, device=d)
        L.call("sv_repack", L.SV_BF16, C.c_void_p(master.data_ptr()), N, 16, Cin, 1, C.byref(g), C.c_void_p(wp2.data_ptr()), st)
        out2 = torch.empty(B, H, H, Cin, dtype=bf, device=d)
        vec = [torch.rand(Cin, device=d) + 0.5 for _ in range(4)]
        bs = torch.zeros(R, 2 * Cin, device=d, dtype=torch.float64)
        a2 = L.SvIgemmArgs()
        a2.x, a2.w, a2.out = dy.data_ptr(), wp2.data_ptr(), out2.data_ptr()
        a2.ex, a2.ex_scale, a2.ex_shift, a2.ex_mean, a2.ex_rstd = [t.data_ptr() for t in [x] + vec]
        a2.ex_slope, a2.bsums, a2.replicas = 0.0, bs.data_ptr(), R
        res["dgrad"] = timed(lambda: L.call("sv_igemm", C.byref(g), L.SV_BF16, C.byref(a2), st))
    if "wgrad" in what:
        dw = torch.zeros(N, 16, Cin, device=d)
        sc2, sh2 = torch.rand(Cin, device=d) + 0.5, torch.randn(Cin, device=d) * 0.3
        ws = torch.empty(16 << 20, device=d)
        res["wgrad"] = timed(lambda: L.call("sv_wgrad", C.byref(gf), L.SV_BF16, C.c_void_p(x.data_ptr()),
                                            C.c_void_p(sc2.data_ptr()), C.c_void_p(sh2.data_ptr()), C.c_float(0.0),
                                            C.c_void_p(dy.data_ptr()), C.c_void_p(dw.data_ptr()), 0, 1,
                                            C.c_void_p(ws.data_ptr()), ws.numel(), 1, st))
    es = 2
    nb = es * (B * H * H * Cin + B * Ho * Ho * N)
    for k, us in res.items():
        by = nb + (4 if k == "wgrad" else es) * 16 * Cin * N + (es * B * H * H * Cin if k == "dgrad" else 0)
        print(f"B={B} Cin={Cin} N={N} H={H} convT4x4s2 {k:6s} {us:9.1f} us  {flops / us / 1e6:8.1f} TFLOP/s  "
              f"{flops / us / 1e-6 / PEAK:6.3f} of bf16 MFMA peak  {by / us / 1e3:7.1f} GB/s", flush=True)
    return res


if __name__ == "__main__":
    if os.environ.get("SV_BENCH_HALO_ALL"):             # A/B: let the LDS-halo kernels take every layer they can run
        L.call("sv_set_option", L.OPT_HALO_ALL, 1)
    if os.environ.get("SV_BENCH_PERSISTENT_BLOCKS"):    # e.g. 256: the budget a body weight gradient gets in the paired backward
        L.call("sv_set_option", L.OPT_PERSISTENT_BLOCKS, int(os.environ["SV_BENCH_PERSISTENT_BLOCKS"]))
    if os.environ.get("SV_BENCH_WIDE_MIN_BLOCKS"):      # threshold of the 256-row tiles of igemm.hip (default 256)
        L.call("sv_set_option", L.OPT_WIDE_MIN_BLOCKS, int(os.environ["SV_BENCH_WIDE_MIN_BLOCKS"]))
    if os.environ.get("SV_BENCH_ENABLE"):             # kernels that are off by default (SV_OPT_ENABLE_MASK)
        L.call("sv_set_option", L.OPT_ENABLE_MASK, int(os.environ["SV_BENCH_ENABLE"]))
    if os.environ.get("SV_BENCH_DISABLE"):
        L.call("sv_set_option", L.OPT_DISABLE_MASK, int(os.environ["SV_BENCH_DISABLE"]))
    if len(sys.argv) >= 5:
        B, Cc, H, N = map(int, sys.argv[1:5])
        bench_layer(B, Cc, H, N, tuple(sys.argv[5:]) or ("fwd", "dgrad", "wgrad"))
    else:
        for Cc, H in ((160, 32), (320, 16), (640, 8)):
            bench_layer(512, Cc, H, Cc)
