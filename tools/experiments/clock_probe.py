"""Shader clock inside the step (tools/probes/clock_probe.hip): a one-wave probe launched on the main stream right behind the forward
and right behind the backward of the default grouped step, with and without an ENABLE mask (SV_ENABLE=1048576: cconv forward).
MHz = 100 * d(s_memtime) / d(s_memrealtime)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import shot_vae_amd as S  # noqa: E402
import shot_vae_amd.train as TR  # noqa: E402
from shot_vae_amd import _lib as L  # noqa: E402

P = C.CDLL(os.path.join(ROOT, "gpurun_out", "libclockprobe.so"))
P.sv_clock_probe.argtypes = [C.c_void_p, C.c_void_p]
K, B = 10, 512
torch.manual_seed(1)
model = S.VariationalAutoEncoder("wideresnet-28-2", num_input_channels=3, img_size=(32, 32), data_parallel=True, continuous_latent_dim=128,
                                 disc_latent_dim=K, small_input=True, compute_dtype="bf16", rng="device").cuda().train()
elbo, cls = S.VAECriterion(discrete_dim=K, bce_reconstruction=True).cuda(), S.ClsCriterion()
opt = S.FlatSGD(model, lr=0.02, momentum=0.9, weight_decay=5e-4)
opt.zero_grad()
sch = S.schedule(10)
il, iu = torch.rand(B, 3, 32, 32, device="cuda"), torch.rand(B, 3, 32, 32, device="cuda")
ll = torch.randint(0, K, (B,), device="cuda")
eng = model._engine
out = torch.zeros(2, 2, dtype=torch.int64, device="cuda")
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731


def run(mask, steps=30):
    L.call("sv_set_option", L.OPT_ENABLE_MASK, mask)
    clocks = []
    for s in range(steps):
        S.train_step_grouped(model, elbo, cls, opt, il, ll, iu, sch)
        P.sv_clock_probe(out[1].data_ptr(), st())          # behind the whole step (its backward + update)
        if s >= 10:
            torch.cuda.synchronize()
            o = out.cpu()
            clocks.append(100.0 * float(o[1, 0]) / float(o[1, 1]))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for s in range(40):
        S.train_step_grouped(model, elbo, cls, opt, il, ll, iu, sch)
    e1.record()
    torch.cuda.synchronize()
    return sum(clocks) / len(clocks), e0.elapsed_time(e1) / 40


for rep in range(2):
    for mask in (0, L.K_CCONV, L.K_CCONV_EX):
        mhz, ms = run(mask)
        print("enable=%8d   clock behind the step %.0f MHz   %.3f ms/step" % (mask, mhz, ms), flush=True)
