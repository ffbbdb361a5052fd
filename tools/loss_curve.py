"""Loss-curve overlay (BASELINE.json north_star: "loss curves overlaying the CPU reference within tolerance").

Trains WRN-28-2 / K = 10 for N steps of the SHOT-VAE loop body (main_shot_vae.py:280-366: 4 forwards, 2 backwards, SGD
with momentum 0.9 / weight decay 5e-4 at the warm-up learning rate 0.02, :223-225) three times from the same
initialisation, on the same synthetic batches and with IDENTICAL scripted host noise (eps, Gumbel u, pairings, lambdas):

    oracle_fp32   the CPU oracle (torch fp32; golden-pinned to the reference)
    hip_fp32      the HIP path in fp32-operand mode (exact-fp32 MFMA)
    hip_bf16      the HIP path in bf16 (the throughput mode bench.py measures)

and writes one JSON line per step with the twelve loss terms of each run:  python tools/loss_curve.py [steps] [B] [out]
(tests/test_loss_curve_gpu.py runs the same function and gates the curves)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import closed_form as C          # noqa: E402
from oracle import shotvae_oracle as O       # noqa: E402
from tests import _cases as T                # noqa: E402

TERMS = T.SCALARS


def batches(n, B, K):
    """n different synthetic batches (closed-form generators: the GPU box regenerates them)"""
    return [C.make_batch(B, B, K, stream0=7000 + 10 * i) for i in range(n)]


def run_oracle(name, K, B, steps, lr, sch, nbatch=4, return_state=False):
    st = O.default_init(name, K=K, seed=1)
    for k in st:
        if O.is_param(k):
            st[k].requires_grad_(True)
    mom, data, curve = {}, batches(nbatch, B, K), []
    for s in range(steps):
        il, ll, iu, lu = data[s % nbatch]
        nz = C.make_noise(B, B, K, stream0=9000 + 100 * s)
        out = O.train_step(st, name, il, ll, iu, nz, sch)
        O.sgd_step(st, mom, lr=lr, momentum=0.9, weight_decay=5e-4)
        curve.append({k: float(out[k]) for k in TERMS})
    if return_state:                      # the trained state (parameters + BatchNorm buffers), detached
        return curve, {k: v.detach().clone() for k, v in st.items()}
    return curve


def run_hip(name, K, B, steps, lr, sch, dtype, nbatch=4, grouped=False):
    import shot_vae_amd as S
    model = S.VariationalAutoEncoder(name, num_input_channels=3, img_size=(32, 32), data_parallel=True,
                                     continuous_latent_dim=128, disc_latent_dim=K, small_input=True, compute_dtype=dtype)
    model.load_state_dict({k: v.detach() for k, v in O.default_init(name, K=K, seed=1).items()})
    model = model.cuda().train()
    elbo, cls = S.VAECriterion(discrete_dim=K, bce_reconstruction=True).cuda(), S.ClsCriterion()
    opt = S.FlatSGD(model, lr=lr, momentum=0.9, weight_decay=5e-4)
    opt.zero_grad()
    data = [tuple(t.cuda() for t in b) for b in batches(nbatch, B, K)]
    step = S.train_step_grouped if grouped else S.train_step
    curve = []
    for s in range(steps):
        il, ll, iu, lu = data[s % nbatch]
        nz = C.make_noise(B, B, K, stream0=9000 + 100 * s)
        with T.rng_for_step(nz):
            out = step(model, elbo, cls, opt, il, ll, iu, sch, return_outputs=True)
        curve.append({k: float(out[k]) for k in TERMS})
    return curve


def deviations(curve, ref):
    """per term: max over steps of |a - b| / max(|b|, floor) (floor = 1 % of the term's largest reference value)"""
    dev = {}
    for k in TERMS:
        b = np.array([r[k] for r in ref])
        a = np.array([c[k] for c in curve])
        dev[k] = float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-2 * np.abs(b).max() + 1e-12)))
    return dev


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", "loss_curves.jsonl")
    name, K, lr = "wideresnet-28-2", 10, 0.02
    sch = O.schedule(10)
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    runs = {"oracle_fp32": run_oracle(name, K, B, steps, lr, sch),
            "hip_fp32": run_hip(name, K, B, steps, lr, sch, "fp32"),
            "hip_bf16": run_hip(name, K, B, steps, lr, sch, "bf16"),
            "hip_bf16_grouped": run_hip(name, K, B, steps, lr, sch, "bf16", grouped=True)}
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        f.write(json.dumps({"meta": dict(net=name, K=K, B_l=B, B_u=B, steps=steps, lr=lr, momentum=0.9, weight_decay=5e-4,
                                         epoch_scalars=sch, terms=TERMS,
                                         max_rel_deviation_vs_oracle={r: deviations(c, runs["oracle_fp32"])
                                                                      for r, c in runs.items() if r != "oracle_fp32"})}) + "\n")
        for s in range(steps):
            f.write(json.dumps({"step": s, **{r: c[s] for r, c in runs.items()}}) + "\n")
    for r, c in runs.items():
        if r != "oracle_fp32":
            d = deviations(c, runs["oracle_fp32"])
            print(r, "max rel deviation vs oracle:", {k: round(v, 5) for k, v in d.items()})
    print("loss_sup first/last:", {r: (round(c[0]["loss_sup"], 5), round(c[-1]["loss_sup"], 5)) for r, c in runs.items()})


if __name__ == "__main__":
    main()
