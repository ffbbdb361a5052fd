"""What a family of launches costs in WALL time of the default step: the step timed with that family skipped (results wrong by
construction).  usage: skip_probe.py [wgrad_body|wgrad_all|bn_apply|none]..."""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 3)[0])
import shot_vae_amd as S                      # noqa: E402
from shot_vae_amd import _lib as L            # noqa: E402
from shot_vae_amd.engine import Engine        # noqa: E402

K, B = 10, 512
torch.manual_seed(1)
model = S.VariationalAutoEncoder("wideresnet-28-2", num_input_channels=3, img_size=(32, 32), data_parallel=True,
                                 continuous_latent_dim=128, disc_latent_dim=K, small_input=True, compute_dtype="bf16",
                                 rng="device").cuda().train()
elbo, cls = S.VAECriterion(discrete_dim=K, bce_reconstruction=True).cuda(), S.ClsCriterion()
opt = S.FlatSGD(model, lr=0.02, momentum=0.9, weight_decay=5e-4)
opt.zero_grad()
sch = S.schedule(10)
il, iu = torch.rand(B, 3, 32, 32, device="cuda"), torch.rand(B, 3, 32, 32, device="cuda")
ll = torch.randint(0, K, (B,), device="cuda")
orig_wgrad, orig_call = Engine._wgrad, L.call
mode = {"skip": "none"}


def wgrad(self, g, x, pro, dy, dw_ptr, tag=None, groups=1, budget=0):
    body = g.nphase == 1 and g.phase[0].ntap == 9 and g.sy == 1 and g.Cin == g.N and g.Cin >= 32
    if mode["skip"] == "wgrad_all" or (mode["skip"] == "wgrad_body" and body):
        return
    return orig_wgrad(self, g, x, pro, dy, dw_ptr, tag, groups, budget)


def call(name, *a):
    if mode["skip"] == "bn_apply" and name == "sv_bn_bwd_apply":
        return
    if mode["skip"] == "bn_finalize" and name == "sv_bn_finalize":
        return
    return orig_call(name, *a)


Engine._wgrad = wgrad
import shot_vae_amd.engine as E               # noqa: E402
E.L.call = call
import os
if os.environ.get("SV_FOLD"):
    model._engine.fold_bn = bool(int(os.environ["SV_FOLD"]))
if os.environ.get("SV_MAT_HIN"):
    model._engine.materialize_max_hin = int(os.environ["SV_MAT_HIN"])
for m in sys.argv[1:] or ["none", "wgrad_body", "wgrad_all", "bn_apply", "bn_finalize", "none"]:
    mode["skip"] = m
    for _ in range(5):
        S.train_step_grouped(model, elbo, cls, opt, il, ll, iu, sch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        S.train_step_grouped(model, elbo, cls, opt, il, ll, iu, sch)
    torch.cuda.synchronize()
    print("%-12s %.3f ms/step" % (m, (time.perf_counter() - t0) / 30 * 1e3), flush=True)
