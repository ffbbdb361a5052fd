"""The oracle (oracle/shotvae_oracle.py) against the reference's own outputs (tests/golden/*.npz).

These goldens are the only pin for parity: the reference ships no tests."""
import numpy as np
import pytest
import torch

from oracle import closed_form as C
from oracle import shotvae_oracle as O
from tests import _cases as T

TOL = 2e-5   # fp32 CPU vs fp32 CPU, different op order in a few places


@pytest.mark.parametrize("tag", list(T.STEP_CASES))
def test_step_matches_reference(tag):
    if "28_10" in tag:
        torch.set_num_threads(8)
    g = T.load(tag)
    outs, st = T.oracle_run(tag)
    for s, out in enumerate(outs):
        for k in T.SCALARS:
            ref = float(g["s%d.%s" % (s, k)])
            assert abs(float(out[k]) - ref) <= TOL * max(1.0, abs(ref)), (tag, s, k, float(out[k]), ref)
        for k in T.TENSORS:
            assert T.rel_err(out[k].numpy(), g["s%d.%s" % (s, k)]) < TOL, (tag, s, k)
        gn, gr = out["grad_norm"], g["s%d.grad_norm" % s]
        # conv0.bias has an analytically-zero gradient (a BN follows on every path): compare with an
        # absolute floor tied to the largest gradient norm
        assert np.all(np.abs(gn - gr) <= 1e-3 * gr + 1e-6 * gr.max()), (tag, s, "grad_norm")
        # (two fp32 implementations of a 28-layer network with batch statistics: the strided gradient sample agrees to 1e-4 at
        #  the 2-6 image fixtures; at 16 / 24 images the summation orders of torch's conv backward differ more -- measured 5e-4)
        gs_tol = 1e-4 if max(T.STEP_CASES[tag][2], T.STEP_CASES[tag][3]) <= 6 else 2e-3
        assert T.rel_err(out["grad_sample"], g["s%d.grad_sample" % s]) < gs_tol, (tag, s, "grad_sample")
    names = [str(n) for n in g["meta.param_names"]]
    pn = np.array([float(st[k].detach().double().norm()) for k in names])
    assert np.max(np.abs(pn - g["final.param_norm"]) / g["final.param_norm"]) < 1e-5
    ps = np.concatenate([st[k].detach().reshape(-1)[torch.from_numpy(T.sample_idx(st[k].numel()))].numpy()
                         for k in names])
    assert T.rel_err(ps, g["final.param_sample"]) < 1e-5
    for k in g.files:
        if k.startswith("final.buf."):
            key = k[len("final.buf."):]
            assert T.rel_err(st[key].numpy(), g[k]) < 1e-5, key


def test_eval_forward_matches_reference():
    g = T.load("ref_eval_wrn10_1")
    st = C.make_state("wideresnet-10-1", K=10)
    il, ll, iu, lu = C.make_batch(4, 4, 10)
    nz = C.make_noise(4, 4, 10)
    with torch.no_grad():
        rec, mu, ls, la = O.vae_forward(st, "wideresnet-10-1", iu, nz["eps3"], u=nz["u3"], training=False)
    for k, v in (("rec", rec), ("mu", mu), ("ls", ls), ("la", la)):
        assert T.rel_err(v.numpy(), g[k]) < TOL, k


def _fn_inputs():
    B, K, D = 6, 10, 128
    x = C.uniform((B, 3, 32, 32), 11)
    xr = C.normal((B, 3, 32, 32), 12) * 2.0
    mu = C.normal((B, D), 13) * 0.7
    ls = C.normal((B, D), 14) * 0.3 - 0.5
    la = torch.log_softmax(C.normal((B, K), 15) * 2.0, dim=1)
    lab = (torch.arange(B) * 3 + 1) % K
    return B, K, x, xr, mu, ls, la, lab


def test_criteria_and_mixup_functions_match_reference():
    g = T.load("ref_functions")
    B, K, x, xr, mu, ls, la, lab = _fn_inputs()
    for bce, sig in ((True, 1.0), (False, 1.0), (False, 0.5)):
        r = O.vae_criterion(x, xr, mu, ls, la, x_sigma=sig, bce=bce)
        ref = g["crit_bce%d_sig%g" % (int(bce), sig)]
        assert np.allclose([float(v) for v in r], ref, rtol=1e-5), (bce, sig)
    soft = torch.softmax(C.normal((B, K), 16), dim=1)
    w = C.uniform((B,), 17)
    assert abs(float(O.cls_criterion(la, soft)) - float(g["cls_soft"])) < 1e-5
    assert abs(float(O.cls_criterion(la, soft, w)) - float(g["cls_weighted"])) < 1e-5
    perm = C.permutation(B, 18)
    outs = O.mix_with_index(x, mu, ls, la, 0.81, perm)
    for v, n in zip(outs, ["img", "mu", "sigma", "alpha"]):
        assert T.rel_err(v.numpy(), g["ls_" + n]) < 1e-6, n
    assert np.array_equal(lab[perm].numpy(), g["ls_label"])
    outs = O.mix_with_index(x, mu, ls, la, 0.42, perm)
    for v, n in zip(outs, ["img", "mu", "sigma", "alpha"]):
        assert T.rel_err(v.numpy(), g["mx_" + n]) < 1e-6, n
    idx = O.optimal_match_index(mu, ls)
    outs = O.mix_with_index(x, mu, ls, la, 0.42, idx)
    for v, n in zip(outs, ["img", "mu", "sigma", "alpha"]):
        assert T.rel_err(v.numpy(), g["om_" + n]) < 1e-6, n


def test_known_answers():
    """Analytic closed forms of lib/criterion.py:44-56 and main_shot_vae.py:518-520."""
    B, K, D = 3, 10, 128
    z = torch.zeros(B, D)
    x = torch.full((B, 3, 32, 32), 0.25)
    uni = torch.full((B, K), float(np.log(1.0 / K)))
    r, kc, kd = O.vae_criterion(x, torch.zeros_like(x), z, z, uni, bce=True)
    assert abs(float(r) - 3 * 32 * 32 * np.log(2.0)) < 1e-2
    assert abs(float(kc)) < 1e-7 and abs(float(kd)) < 1e-5
    peaked = torch.log_softmax(torch.tensor([[60.0] + [0.0] * (K - 1)]).repeat(B, 1), dim=1)
    assert abs(float(O.vae_criterion(x, x, z, z, peaked)[2]) - np.log(K)) < 1e-4
    assert abs(O.alpha_schedule(0, 200, 2.0) - 2.0 * np.exp(-5.0)) < 1e-12
    assert O.alpha_schedule(200, 200, 2.0) == 2.0 and O.alpha_schedule(300, 200, 2.0) == 2.0


def test_m2_baseline_step_matches_reference():
    """oracle.m2_step (main_M2_vae.py:259-305) against the reference run by tests/golden/make_goldens.py (m2 case)."""
    name, K, B = "wideresnet-10-1", 10, 6
    g = T.load("ref_m2_step_wrn10_1")
    st = C.make_state(name, K=K)
    for k in st:
        if O.is_param(k):
            st[k].requires_grad_(True)
    sch = O.schedule(10)
    il, ll, iu, lu = C.make_batch(B, B, K, stream0=7300)
    nz = C.make_noise(B, B, K, stream0=9300)
    out = O.m2_step(st, name, il, ll, iu, lu, nz, sch)
    for k in ("recon_l", "klc_l", "kld_l", "recon_u", "klc_u", "kld_u", "disc_post_l", "kl_inference", "loss_sup",
              "loss_unsup"):
        assert abs(float(out[k]) - float(g[k])) <= 2e-5 * max(1.0, abs(float(g[k]))), k
    for k in ("rec1", "mu1", "ls1", "la1", "rec3", "mu3", "ls3", "la3"):
        assert T.rel_err(out[k].numpy(), g[k]) < 2e-5, k
    pk = [k for k in st if O.is_param(k)]
    gn = np.array([float(st[k].grad.double().norm()) for k in pk])
    assert np.max(np.abs(gn - g["grad_norm"])) < 2e-4 * np.max(g["grad_norm"])


SMOOTH_SCALARS = ["loss", "loss_u", "loss_l", "recon_u", "cont_u", "disc_u", "recon_l", "cont_l", "disc_l", "cls_l"]
SMOOTH_TENSORS = ["rec_u", "mean_u", "logvar_u", "alpha_u", "rec_l", "mean_l", "logvar_l", "alpha_l"]


@pytest.mark.parametrize("kind,Bu,Bl", [("svhn", 6, 4), ("mnist", 4, 6)])
def test_smooth_elbo_iteration_matches_reference(kind, Bu, Bl):
    """oracle.smooth_oracle (svhn_VAE / mnist_VAE forward, the trainer's loss, Adam) against one iteration of the
    reference model driven by tests/golden/make_goldens.py (smooth cases)."""
    from oracle import smooth_oracle as SO
    g = T.load("ref_smooth_" + kind)
    st = SO.make_state(kind)
    for k in st:
        st[k].requires_grad_(True)
    unl, lab, label, nz = SO.make_inputs(kind, Bu, Bl)
    out = SO.train_iteration(st, kind, unl, lab, label, nz, int(g["meta.num_steps"]))
    for k in SMOOTH_SCALARS:
        assert abs(float(out[k]) - float(g[k])) <= 2e-5 * max(1.0, abs(float(g[k]))), (k, float(out[k]), float(g[k]))
    for k in SMOOTH_TENSORS:
        assert T.rel_err(out[k].numpy(), g[k]) < 2e-5, k
    gn = np.array([float(st[k].grad.double().norm()) for k in st])
    assert np.max(np.abs(gn - g["grad_norm"])) < 2e-4 * np.max(g["grad_norm"])
    gs = np.concatenate([st[k].grad.reshape(-1)[torch.from_numpy(T.sample_idx(st[k].numel()))].numpy() for k in st])
    assert T.rel_err(gs, g["grad_sample"]) < 2e-4
    SO.adam_step(st, {})
    pn = np.array([float(st[k].detach().double().norm()) for k in st])
    assert np.max(np.abs(pn - g["final.param_norm"]) / g["final.param_norm"]) < 1e-5
    ps = np.concatenate([st[k].detach().reshape(-1)[torch.from_numpy(T.sample_idx(st[k].numel()))].numpy() for k in st])
    assert T.rel_err(ps, g["final.param_sample"]) < 1e-5
