"""Shared golden-case table + helpers (inputs are regenerated from oracle/closed_form.py)."""
import os

import numpy as np
import torch

from oracle import closed_form as C
from oracle import shotvae_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# tag -> (net, K, Bl, Bu, bce, x_sigma, om, dmi, steps)
STEP_CASES = {
    "ref_step_wrn10_1_br": ("wideresnet-10-1", 10, 4, 6, True, 1.0, False, 2.3, 2),
    "ref_step_wrn10_1_om": ("wideresnet-10-1", 10, 4, 6, True, 1.0, True, 2.3, 1),
    "ref_step_wrn28_2_br": ("wideresnet-28-2", 10, 4, 4, True, 1.0, False, 2.3, 1),
    "ref_step_wrn28_2_mse": ("wideresnet-28-2", 10, 4, 4, False, 0.5, False, 2.3, 1),
    "ref_step_wrn28_10_k100": ("wideresnet-28-10", 100, 2, 2, True, 1.0, False, 4.6, 1),
    # round 4: a larger ragged batch (one launch sequence per loader in the grouped step) and --om with B_l == B_u (three groups,
    # the pairing kernel, the fourth) on the headline network
    "ref_step_wrn28_2_b16_24": ("wideresnet-28-2", 10, 16, 24, True, 1.0, False, 2.3, 1),
    "ref_step_wrn28_2_om_b16": ("wideresnet-28-2", 10, 16, 16, True, 1.0, True, 2.3, 1),
}
SCALARS = ["recon_l", "klc_l", "kld_l", "recon_u", "klc_u", "kld_u", "disc_post_l", "cont_post_l",
           "disc_post_u", "cont_post_u", "loss_sup", "loss_unsup"]
TENSORS = ["%s%d" % (n, i) for i in (1, 2, 3, 4) for n in ("rec", "mu", "ls", "la")] + ["sm_img", "mx_img"]


def load(tag):
    return np.load(os.path.join(GOLDEN, tag + ".npz"))


def sample_idx(n, k=16):
    return np.unique(np.linspace(0, n - 1, num=min(k, n)).astype(np.int64))


def rel_err(a, b):
    """max|a-b| / max(|b|) -- the 'relative to tensor scale' error the parity gates use."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-30))


def oracle_run(tag):
    """Run the oracle on a golden case; returns (per-step outputs, grad norms/samples, final state)."""
    name, K, Bl, Bu, bce, x_sigma, om, dmi, steps = STEP_CASES[tag]
    st = C.make_state(name, K=K)
    for k in st:
        if O.is_param(k):
            st[k].requires_grad_(True)
    sch = O.schedule(10, dmi=dmi)
    mom = {}
    outs = []
    for s in range(steps):
        il, ll, iu, lu = C.make_batch(Bl, Bu, K, stream0=7000 + 10 * s)
        nz = C.make_noise(Bl, Bu, K, stream0=9000 + 100 * s)
        out = O.train_step(st, name, il, ll, iu, nz, sch, bce=bce, x_sigma=x_sigma, optimal_match=om)
        pk = [k for k in st if O.is_param(k)]
        out["grad_norm"] = np.array([float(st[k].grad.double().norm()) for k in pk])
        out["grad_sample"] = np.concatenate(
            [st[k].grad.reshape(-1)[torch.from_numpy(sample_idx(st[k].numel()))].numpy() for k in pk])
        outs.append(out)
        O.sgd_step(st, mom, lr=0.1, momentum=0.9, weight_decay=5e-4)
    return outs, st


# ---- scripted host RNG (same technique as tests/golden/make_goldens.py) -------------------------
import contextlib


@contextlib.contextmanager
def scripted_rng(randn=(), rand=(), randperm=(), beta=()):
    """Replace torch.randn / torch.rand / torch.randperm / numpy.random.beta with scripted queues so
    that the reference-compatible host-RNG code paths consume exactly the fixture's noise."""
    q = dict(randn=list(randn), rand=list(rand), randperm=list(randperm), beta=list(beta))
    saved = (torch.randn, torch.rand, torch.randperm, np.random.beta)

    def pop(kind, size=None):
        v = q[kind].pop(0)
        if size is not None and torch.is_tensor(v):
            assert tuple(v.shape) == tuple(size), (kind, tuple(v.shape), tuple(size))
        return v.clone() if torch.is_tensor(v) else v

    def _sz(s):
        return s[0] if len(s) == 1 and not isinstance(s[0], int) else s

    torch.randn = lambda *s, **k: pop("randn", _sz(s))
    torch.rand = lambda *s, **k: pop("rand", _sz(s))
    torch.randperm = lambda n, **k: pop("randperm", (n,))
    np.random.beta = lambda a, b: pop("beta")
    try:
        yield
    finally:
        torch.randn, torch.rand, torch.randperm, np.random.beta = saved
        for k, v in q.items():
            assert not v, "unused scripted %s draws: %d" % (k, len(v))


def rng_for_step(nz, om=False):
    return scripted_rng(randn=[nz["eps1"], nz["eps2"], nz["eps3"], nz["eps4"]], rand=[nz["u3"], nz["u4"]],
                        randperm=[nz["perm_l"]] + ([] if om else [nz["perm_u"]]),
                        beta=[nz["lam_l"], nz["lam_u"]])
