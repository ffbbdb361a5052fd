"""Per-call time table of one config-5 iteration (svhn_VAE smooth-ELBO, B per loader): every library call of the eager
iteration bracketed by events on the current stream (torch's own glue kernels fall between the brackets and are not listed).
    python tools/probes/svhn_layers.py [B=1024] [iters=10] [sweep]
sweep: the same table under several dispatcher option sets side by side (which general kernel serves a layer best)."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import shot_vae_amd as S                      # noqa: E402
from shot_vae_amd import _lib as L            # noqa: E402
from shot_vae_amd import smooth as SM         # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
torch.manual_seed(1)
model = SM.SmoothVAE((3, 32, 32), {"cont": 32, "disc": [10]}, temperature=0.67, compute_dtype="bf16").cuda().train()
loss_fn = SM.SmoothELBOLoss()
opt = S.FlatAdam(model.parameters(), lr=1e-3, capturable=True)
g = torch.Generator(device="cuda").manual_seed(1234)
u = torch.rand(B, 3, 32, 32, device="cuda", generator=g) * 2 - 1
l = torch.rand(B, 3, 32, 32, device="cuda", generator=g) * 2 - 1
y = torch.randint(0, 10, (B,), device="cuda", generator=g)

real_call = L.call
log, recording = [], [False]


def describe(name, args):
    if name in ("sv_igemm", "sv_wgrad", "sv_wgrad_ex"):
        gm = args[0]._obj
        return "%-12s B=%d %dx%d Cin=%d -> %dx%d N=%d phases=%d taps=%d" % (name, gm.B, gm.Hin, gm.Win, gm.Cin, gm.Hout, gm.Wout, gm.N,
                                                                            gm.nphase, gm.phase[0].ntap)
    return name


def call(name, *args):
    if not recording[0]:
        return real_call(name, *args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = real_call(name, *args)
    e1.record()
    log.append((describe(name, args), e0, e1))
    return r


L.call = call
SM.L.call = call


def table(opts):
    del log[:]
    recording[0] = False
    with L.options(**opts):
        for it in range(3):
            SM.smooth_train_step(model, loss_fn, opt, u, l, y)
        torch.cuda.synchronize()
        recording[0] = True
        w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        w0.record()
        for it in range(iters):
            SM.smooth_train_step(model, loss_fn, opt, u, l, y)
        w1.record()
        torch.cuda.synchronize()
    recording[0] = False
    n = len(log) // iters
    tab = collections.OrderedDict()
    for i, (d, e0, e1) in enumerate(log):
        tab.setdefault((i % n, d), []).append(e0.elapsed_time(e1) * 1e3)
    return [(k, sorted(v)[len(v) // 2]) for k, v in tab.items()], w0.elapsed_time(w1) / iters


SETS = [("default", {})]
if len(sys.argv) > 3:
    SETS += [("no BIG", dict(disable=L.K_IGEMM_BIG)), ("no DMA", dict(disable=L.K_IGEMM_DMA)), ("no HALO/P", dict(disable=L.K_HALO | L.K_HALOP)),
             ("HALO_ALL", dict(halo_all=1)), ("no ALIGNED", dict(disable=L.K_IGEMM_ALIGNED)), ("no KV2", dict(disable=L.K_IGEMM_KV2)),
             ("no WG_WIDE", dict(disable=L.K_WGRAD_WIDE)), ("no WG_INCR", dict(disable=L.K_WGRAD_INCR)), ("no HWGRAD", dict(disable=L.K_HWGRAD))]
cols = [table(o) for _, o in SETS]
print(" " * 4 + "".join("%11s" % nm for nm, _ in SETS))
for r, ((i, d), _) in enumerate(cols[0][0]):
    print("%3d " % i + "".join("%11.1f" % c[0][r][1] for c in cols) + "  " + d)
print("sum " + "".join("%11.1f" % sum(v for _, v in c[0]) for c in cols) + "  us of library calls (event-bracketed)")
print("wall" + "".join("%11.3f" % c[1] for c in cols) + "  ms per iteration (eager, bracketed: slower than the timed step)")
