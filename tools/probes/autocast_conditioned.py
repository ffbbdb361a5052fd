"""What bf16 operands cost the GRADIENT of one SHOT-VAE step, measured on the reference arithmetic itself (CPU, no GPU needed):
the fp32 oracle against the same oracle under torch's CPU bf16 autocast (convolutions / linear layers on bf16 operands, fp32
accumulation) -- at the ill-conditioned initial weights (step 0) and at the CONDITIONED point of tests/test_loss_curve_gpu.py
(default initialisation + N fp32 oracle SGD steps).  The HIP bf16 path is gated to do at least as well as this.

    python tools/probes/autocast_conditioned.py [steps=60] [B=64]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import loss_curve as LC                       # noqa: E402
from oracle import closed_form as C          # noqa: E402
from oracle import shotvae_oracle as O       # noqa: E402


def grads(st, name, batch, nz, sch, autocast):
    s = {k: v.detach().clone() for k, v in st.items()}
    for k in s:
        if O.is_param(k):
            s[k].requires_grad_(True)
    il, ll, iu, lu = batch
    with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
        O.train_step(s, name, il, ll, iu, nz, sch)
    return torch.cat([s[k].grad.double().flatten() for k in s if O.is_param(k) and not k.endswith("conv0.bias")])


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    name, K, lr = "wideresnet-28-2", 10, 0.02
    sch = O.schedule(10)
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    for label, n in (("initial weights", 0), ("after %d fp32 SGD steps" % steps, steps)):
        _, st = LC.run_oracle(name, K, B, n, lr, sch, return_state=True)
        batch = LC.batches(4, B, K)[n % 4]
        nz = C.make_noise(B, B, K, stream0=9000 + 100 * n)
        g32, g16 = grads(st, name, batch, nz, sch, False), grads(st, name, batch, nz, sch, True)
        print("%-28s torch bf16 autocast of the oracle vs its fp32 gradient: flat cosine %.4f, relative L2 %.3f"
              % (label, float(g32 @ g16 / g32.norm() / g16.norm()), float((g32 - g16).norm() / g32.norm())), flush=True)


if __name__ == "__main__":
    main()
