"""Generate golden vectors by running the REFERENCE implementation on CPU.

Runs only in the build container (needs /root/reference, read-only).  It imports the reference's
own modules (shot_vae_model.vae, lib.criterion, lib.utils.mixup) behind a ``.cuda()`` no-op shim
(SURVEY.md Appendix B), loads closed-form weights (oracle/closed_form.py), feeds explicit noise by
temporarily replacing torch.randn / torch.rand / torch.randperm / numpy.random.beta, and drives the
step of main_shot_vae.py:280-366 (that script cannot be imported: argparse + torchvision at import).

Outputs: tests/golden/*.npz  (reference OUTPUTS only -- inputs are regenerated from closed forms).

    python tests/golden/make_goldens.py
"""
import contextlib
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import closed_form as C          # noqa: E402
from oracle import shotvae_oracle as O       # noqa: E402  (only alpha schedule defaults + key helpers)

REF = "/root/reference"


def import_reference():
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)
    from shot_vae_model.vae import VariationalAutoEncoder
    from lib.criterion import VAECriterion, ClsCriterion
    from lib.utils.mixup import mixup_vae_data, label_smoothing
    return VariationalAutoEncoder, VAECriterion, ClsCriterion, mixup_vae_data, label_smoothing


@contextlib.contextmanager
def scripted_rng(randn=(), rand=(), randperm=(), beta=()):
    """Replace the host RNG entry points the reference uses with scripted queues."""
    q = dict(randn=list(randn), rand=list(rand), randperm=list(randperm), beta=list(beta))
    saved = (torch.randn, torch.rand, torch.randperm, np.random.beta)

    def pop(kind, size=None):
        v = q[kind].pop(0)
        if size is not None:
            assert tuple(v.shape) == tuple(size), (kind, v.shape, size)
        return v.clone() if torch.is_tensor(v) else v

    torch.randn = lambda *s, **k: pop("randn", s[0] if len(s) == 1 and not isinstance(s[0], int) else s)
    torch.rand = lambda *s, **k: pop("rand", s[0] if len(s) == 1 and not isinstance(s[0], int) else s)
    torch.randperm = lambda n, **k: pop("randperm", (n,))
    np.random.beta = lambda a, b: pop("beta")
    try:
        yield
    finally:
        torch.randn, torch.rand, torch.randperm, np.random.beta = saved
        for k, v in q.items():
            assert not v, "unused scripted %s draws: %d" % (k, len(v))


def alpha_schedule(epoch, max_epoch, alpha_max):
    import math
    return alpha_max * math.exp(-5 * (1 - min(1, epoch / max_epoch)) ** 2)


def reference_step(model, elbo_criterion, cls_criterion, label_smoothing, mixup_vae_data,
                   image_l, label_l, image_u, label_u, K, sch, epsilon, om):
    """Body of the loop at main_shot_vae.py:281-364 (one iteration), driving reference objects."""
    import torch.nn.functional as F
    batch_size_l, batch_size_u = image_l.size(0), image_u.size(0)
    label_onehot_l = torch.zeros(batch_size_l, K).scatter_(1, label_l.view(-1, 1), 1)
    rec1, mu1, ls1, la1 = model(image_l, disc_label=label_l)
    recon_l, klc_l, kld_l = elbo_criterion(image_l, rec1, mu1, ls1, la1)
    prior_l = sch["kl_beta_c"] * torch.abs(klc_l - sch["cmi"]) + sch["kl_beta_d"] * torch.abs(kld_l - sch["dmi"])
    elbo_l = recon_l + prior_l
    with torch.no_grad():
        sm_img, sm_mu, sm_sigma, sm_alpha, sm_label, lam_l = label_smoothing(
            image_l, mu1, ls1, la1, epsilon=epsilon, disc_label=label_l)
        sm_onehot = torch.zeros(batch_size_l, K).scatter_(1, sm_label.view(-1, 1), 1)
    rec2, mu2, ls2, la2, *_ = model(sm_img, True, label_l, sm_label, lam_l)
    disc_post_l = lam_l * cls_criterion(la2, label_onehot_l) + (1 - lam_l) * cls_criterion(la2, sm_onehot)
    cont_post_l = (F.mse_loss(mu2, sm_mu, reduction="sum")
                   + F.mse_loss(torch.exp(ls2), sm_sigma, reduction="sum")) / batch_size_l
    elbo_l = elbo_l + sch["kl_beta_c"] * sch["pwm"] * cont_post_l
    loss_sup = sch["ew"] * elbo_l + disc_post_l
    loss_sup.backward()
    rec3, mu3, ls3, la3 = model(image_u)
    recon_u, klc_u, kld_u = elbo_criterion(image_u, rec3, mu3, ls3, la3)
    prior_u = sch["kl_beta_c"] * torch.abs(klc_u - sch["cmi"]) + sch["kl_beta_d"] * torch.abs(kld_u - sch["dmi"])
    elbo_u = recon_u + prior_u
    with torch.no_grad():
        mx_img, mx_mu, mx_sigma, mx_alpha, lam_u = mixup_vae_data(image_u, mu3, ls3, la3, optimal_match=om)
    rec4, mu4, ls4, la4, *_ = model(mx_img)
    disc_post_u = cls_criterion(la4, mx_alpha)
    cont_post_u = (F.mse_loss(mu4, mx_mu, reduction="sum")
                   + F.mse_loss(torch.exp(ls4), mx_sigma, reduction="sum")) / batch_size_u
    elbo_u = elbo_u + sch["kl_beta_c"] * sch["pwm"] * cont_post_u
    loss_unsup = sch["ew"] * elbo_u + sch["ucw"] * disc_post_u
    loss_unsup.backward()
    loc = dict(locals())
    keys = ["recon_l", "klc_l", "kld_l", "recon_u", "klc_u", "kld_u", "disc_post_l", "cont_post_l",
            "disc_post_u", "cont_post_u", "loss_sup", "loss_unsup"] + \
           ["%s%d" % (n, i) for i in (1, 2, 3, 4) for n in ("rec", "mu", "ls", "la")] + ["sm_img", "mx_img"]
    return {k: loc[k].detach().clone() for k in keys}


def grad_sample_idx(n, k=16):
    return np.unique(np.linspace(0, n - 1, num=min(k, n)).astype(np.int64))


def run_step_case(tag, name, K, Bl, Bu, bce, x_sigma=1.0, om=False, epoch=10, dmi=2.3, steps=1):
    VAE, VAECriterion, ClsCriterion, mixup_vae_data, label_smoothing = import_reference()
    model = VAE(encoder_name=name, num_input_channels=3, drop_rate=0, img_size=(32, 32),
                data_parallel=False, continuous_latent_dim=128, disc_latent_dim=K,
                sample_temperature=0.67, small_input=True)
    st = C.make_state(name, K=K)
    sd = model.state_dict()
    assert list(sd.keys()) == list(st.keys()), "state_dict key order mismatch"
    model.load_state_dict(st)
    model.train()
    elbo = VAECriterion(discrete_dim=K, x_sigma=x_sigma, bce_reconstruction=bce)
    cls = ClsCriterion()
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=5e-4)
    opt.zero_grad()
    sch = dict(cmi=alpha_schedule(epoch, 200, 0.0), dmi=alpha_schedule(epoch, 200, dmi),
               ew=alpha_schedule(epoch, 400, 1e-3), kl_beta_c=alpha_schedule(epoch, 200, 1e-3),
               kl_beta_d=alpha_schedule(epoch, 200, 1e-3), pwm=alpha_schedule(epoch, 200, 1.0),
               ucw=alpha_schedule(epoch, round(0.4 * 600), 1.0))
    rec = {}
    for s in range(steps):
        il, ll, iu, lu = C.make_batch(Bl, Bu, K, stream0=7000 + 10 * s)
        nz = C.make_noise(Bl, Bu, K, stream0=9000 + 100 * s)
        randperm = [nz["perm_l"]] + ([] if om else [nz["perm_u"]])
        with scripted_rng(randn=[nz["eps1"], nz["eps2"], nz["eps3"], nz["eps4"]],
                          rand=[nz["u3"], nz["u4"]], randperm=randperm,
                          beta=[nz["lam_l"], nz["lam_u"]]):
            out = reference_step(model, elbo, cls, label_smoothing, mixup_vae_data, il, ll, iu, lu,
                                 K, sch, 0.1, om)
        pre = "s%d." % s
        for k, v in out.items():
            rec[pre + k] = v.numpy()
        names = [k for k, _ in model.named_parameters()]
        gn = np.array([float(p.grad.double().norm()) for _, p in model.named_parameters()])
        rec[pre + "grad_norm"] = gn
        rec[pre + "grad_sample"] = np.concatenate(
            [p.grad.reshape(-1)[torch.from_numpy(grad_sample_idx(p.numel()))].numpy()
             for _, p in model.named_parameters()])
        opt.step()
        opt.zero_grad()
    sd = model.state_dict()
    rec["final.param_norm"] = np.array([float(sd[k].double().norm()) for k in names])
    rec["final.param_sample"] = np.concatenate(
        [sd[k].reshape(-1)[torch.from_numpy(grad_sample_idx(sd[k].numel()))].numpy() for k in names])
    for k, v in sd.items():
        if k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"):
            rec["final.buf." + k] = v.numpy()
    rec["meta.param_names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), **rec)
    print(tag, "loss_sup", float(out["loss_sup"]), "loss_unsup", float(out["loss_unsup"]),
          "bytes", os.path.getsize(os.path.join(HERE, tag + ".npz")))


def reference_m2_step(model, elbo_criterion, cls_criterion, image_l, label_l, image_u, label_u, K, sch):
    """Body of the loop at main_M2_vae.py:259-305 (the M2 baseline: no mixup, two forwards, two backwards)."""
    batch_size = image_l.size(0)
    label_onehot_l = torch.zeros(batch_size, K).scatter_(1, label_l.view(-1, 1), 1)
    rec1, mu1, ls1, la1 = model(image_l, disc_label=label_l)
    recon_l, klc_l, kld_l = elbo_criterion(image_l, rec1, mu1, ls1, la1)
    prior_l = sch["kl_beta_c"] * torch.abs(klc_l - sch["cmi"]) + sch["kl_beta_d"] * torch.abs(kld_l - sch["dmi"])
    elbo_l = recon_l + prior_l
    disc_post_l = cls_criterion(la1, label_onehot_l)
    loss_sup = sch["ew"] * elbo_l + disc_post_l
    loss_sup.backward()
    rec3, mu3, ls3, la3 = model(image_u)
    with torch.no_grad():
        label_smooth_u = torch.zeros(batch_size, K).scatter_(1, label_u.view(-1, 1), 1 - 0.001 - 0.001 / (K - 1))
        label_smooth_u = label_smooth_u + torch.ones(label_smooth_u.size()) * 0.001 / (K - 1)
        disc_alpha_u = torch.exp(la3)
        inference_kl = disc_alpha_u * la3 - disc_alpha_u * torch.log(label_smooth_u)
        kl_inference = torch.sum(inference_kl) / batch_size
    recon_u, klc_u, kld_u = elbo_criterion(image_u, rec3, mu3, ls3, la3)
    prior_u = sch["kl_beta_c"] * torch.abs(klc_u - sch["cmi"]) + sch["kl_beta_d"] * torch.abs(kld_u - sch["dmi"])
    elbo_u = recon_u + prior_u
    loss_unsup = sch["ew"] * elbo_u
    loss_unsup.backward()
    loc = dict(locals())
    keys = ["recon_l", "klc_l", "kld_l", "recon_u", "klc_u", "kld_u", "disc_post_l", "kl_inference", "loss_sup",
            "loss_unsup", "rec1", "mu1", "ls1", "la1", "rec3", "mu3", "ls3", "la3"]
    return {k: loc[k].detach().clone() for k in keys}


def run_m2_case(tag, name, K, B, epoch=10, dmi=2.3):
    VAE, VAECriterion, ClsCriterion, _, _ = import_reference()
    model = VAE(encoder_name=name, num_input_channels=3, drop_rate=0, img_size=(32, 32), data_parallel=False,
                continuous_latent_dim=128, disc_latent_dim=K, sample_temperature=0.67, small_input=True)
    model.load_state_dict(C.make_state(name, K=K))
    model.train()
    elbo, cls = VAECriterion(discrete_dim=K, x_sigma=1.0, bce_reconstruction=True), ClsCriterion()
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=5e-4)
    opt.zero_grad()
    sch = dict(cmi=alpha_schedule(epoch, 200, 0.0), dmi=alpha_schedule(epoch, 200, dmi),
               ew=alpha_schedule(epoch, 400, 1e-3), kl_beta_c=alpha_schedule(epoch, 200, 1e-3),
               kl_beta_d=alpha_schedule(epoch, 200, 1e-3))
    il, ll, iu, lu = C.make_batch(B, B, K, stream0=7300)
    nz = C.make_noise(B, B, K, stream0=9300)
    with scripted_rng(randn=[nz["eps1"], nz["eps3"]], rand=[nz["u3"]]):
        out = reference_m2_step(model, elbo, cls, il, ll, iu, lu, K, sch)
    rec = {k: v.numpy() for k, v in out.items()}
    names = [k for k, _ in model.named_parameters()]
    rec["grad_norm"] = np.array([float(p.grad.double().norm()) for _, p in model.named_parameters()])
    rec["grad_sample"] = np.concatenate(
        [p.grad.reshape(-1)[torch.from_numpy(grad_sample_idx(p.numel()))].numpy() for _, p in model.named_parameters()])
    opt.step()
    sd = model.state_dict()
    rec["final.param_norm"] = np.array([float(sd[k].double().norm()) for k in names])
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), **rec)
    print(tag, "loss_sup", float(out["loss_sup"]), "loss_unsup", float(out["loss_unsup"]), "kl_inf",
          float(out["kl_inference"]), "bytes", os.path.getsize(os.path.join(HERE, tag + ".npz")))


def run_smooth_case(tag, kind, Bu, Bl, num_steps=7):
    """One iteration of the one-stage smooth-ELBO trainer on the REFERENCE model (smooth_vae_model/{svhn,mnist}_vae.py).
    The trainer script (main_smooth_ELBO_svhn.py) has argparse at import, so the loop body (:152-176) and the loss
    (:228-388) are restated here, driving the reference model and torch.optim.Adam."""
    import math
    import torch.nn.functional as F
    from oracle import smooth_oracle as SO
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)
    if kind == "svhn":
        from smooth_vae_model.svhn_vae import svhn_VAE as Model
        img = (3, 32, 32)
    else:
        from smooth_vae_model.mnist_vae import mnist_VAE as Model
        img = (1, 32, 32)
    spec = {"cont": 32, "disc": [10]}
    model = Model(img_size=img, latent_spec=spec, temperature=0.67, use_cuda=False)
    st = SO.make_state(kind)
    sd = model.state_dict()
    assert list(sd.keys()) == list(st.keys()), (list(sd.keys()), list(st.keys()))
    assert all(tuple(sd[k].shape) == tuple(st[k].shape) for k in sd)
    model.load_state_dict(st)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    unl, lab, label, nz = SO.make_inputs(kind, Bu, Bl)
    cap_c, cap_d, cls_alpha, EPS = (0.0, 50, 50000, 1), (0.0, 50, 50000, 1), 1500.0, 1e-12

    def loss_function(data, recon, dist, lbl=None):
        npix = img[0] * img[1] * img[2]
        recon_loss = F.mse_loss(recon.view(-1, npix), data.view(-1, npix)) * npix
        mean, logvar = dist["cont"]
        kl_values = -0.5 * (1 + logvar - mean.pow(2) - logvar.exp())
        kl_c = torch.sum(torch.mean(kl_values, dim=0))
        cc = min((cap_c[1] - cap_c[0]) * num_steps / float(cap_c[2]) + cap_c[0], cap_c[1])
        cont_loss = cap_c[3] * torch.abs(cc - kl_c)
        alpha = dist["disc"][0]
        neg_entropy = torch.sum(alpha * torch.log(alpha + EPS), dim=1)
        kl_d = torch.Tensor([np.log(10)]) + torch.mean(neg_entropy, dim=0)
        kl_d = torch.sum(torch.cat([kl_d]))
        dc = min(min((cap_d[1] - cap_d[0]) * num_steps / float(cap_d[2]) + cap_d[0], cap_d[1]), float(np.log(10)))
        disc_loss = cap_d[3] * torch.abs(dc - kl_d)
        cls = 0
        if lbl is not None:
            one_hot = torch.Tensor(np.eye(10)[lbl.cpu()])
            cls = cls_alpha * nn.BCELoss()(alpha, one_hot)
        return recon_loss + cont_loss + disc_loss + cls, (recon_loss, cont_loss, disc_loss, cls)

    # host RNG of the model: torch.zeros(size).normal_() and torch.rand(size) -- scripted in call order
    q_n, q_u = [nz["eps_u"], nz["eps_l"]], [nz["u_u"], nz["u_l"]]
    saved = (torch.Tensor.normal_, torch.rand)
    torch.Tensor.normal_ = lambda self, *a, **k: self.copy_(q_n.pop(0))
    torch.rand = lambda *s, **k: q_u.pop(0).clone()
    try:
        opt.zero_grad()
        rec_u, dist_u, _, _ = model(unl)
        loss_u, split_u = loss_function(unl, rec_u, dist_u)
        rec_l, dist_l, _, dsamp = model(lab, label)
        loss_l, split_l = loss_function(lab, rec_l, dist_l, label)
        loss = loss_u + loss_l
        loss.backward()
    finally:
        torch.Tensor.normal_, torch.rand = saved
    assert not q_n and not q_u
    rec = dict(loss=loss, loss_u=loss_u, loss_l=loss_l, recon_u=split_u[0], cont_u=split_u[1], disc_u=split_u[2],
               recon_l=split_l[0], cont_l=split_l[1], disc_l=split_l[2], cls_l=split_l[3], rec_u=rec_u,
               mean_u=dist_u["cont"][0], logvar_u=dist_u["cont"][1], alpha_u=dist_u["disc"][0], rec_l=rec_l,
               mean_l=dist_l["cont"][0], logvar_l=dist_l["cont"][1], alpha_l=dist_l["disc"][0])
    rec = {k: v.detach().numpy() for k, v in rec.items()}
    names = [k for k, _ in model.named_parameters()]
    rec["grad_norm"] = np.array([float(p.grad.double().norm()) for _, p in model.named_parameters()])
    rec["grad_sample"] = np.concatenate(
        [p.grad.reshape(-1)[torch.from_numpy(grad_sample_idx(p.numel()))].numpy() for _, p in model.named_parameters()])
    opt.step()
    sd = model.state_dict()
    rec["final.param_norm"] = np.array([float(sd[k].double().norm()) for k in names])
    rec["final.param_sample"] = np.concatenate(
        [sd[k].reshape(-1)[torch.from_numpy(grad_sample_idx(sd[k].numel()))].numpy() for k in names])
    rec["meta.num_steps"] = np.array(num_steps)
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), **rec)
    print(tag, "loss", float(loss), "recon_u", float(split_u[0]), "cls", float(split_l[3]), "bytes",
          os.path.getsize(os.path.join(HERE, tag + ".npz")))


def run_eval_case(tag, name, K, B):
    VAE, *_ = import_reference()
    model = VAE(encoder_name=name, num_input_channels=3, drop_rate=0, img_size=(32, 32),
                data_parallel=False, continuous_latent_dim=128, disc_latent_dim=K,
                sample_temperature=0.67, small_input=True)
    model.load_state_dict(C.make_state(name, K=K))
    model.eval()
    il, ll, iu, lu = C.make_batch(B, B, K)
    nz = C.make_noise(B, B, K)
    with torch.no_grad(), scripted_rng(randn=[nz["eps3"]], rand=[nz["u3"]]):
        rec, mu, ls, la = model(iu)
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), rec=rec.numpy(), mu=mu.numpy(), ls=ls.numpy(),
                        la=la.numpy())
    print(tag, float(rec.abs().mean()))


def run_fn_cases(tag):
    """Direct calls of the reference criteria / mixup functions."""
    _, VAECriterion, ClsCriterion, mixup_vae_data, label_smoothing = import_reference()
    B, K, D = 6, 10, 128
    x = C.uniform((B, 3, 32, 32), 11)
    xr = C.normal((B, 3, 32, 32), 12) * 2.0
    mu = C.normal((B, D), 13) * 0.7
    ls = C.normal((B, D), 14) * 0.3 - 0.5
    la = torch.log_softmax(C.normal((B, K), 15) * 2.0, dim=1)
    lab = (torch.arange(B) * 3 + 1) % K
    rec = {}
    for bce, sig in ((True, 1.0), (False, 1.0), (False, 0.5)):
        r, kc, kd = VAECriterion(discrete_dim=K, x_sigma=sig, bce_reconstruction=bce)(x, xr, mu, ls, la)
        rec["crit_bce%d_sig%g" % (int(bce), sig)] = np.array([float(r), float(kc), float(kd)])
    soft = torch.softmax(C.normal((B, K), 16), dim=1)
    w = C.uniform((B,), 17)
    rec["cls_soft"] = np.array(float(ClsCriterion()(la, soft)))
    rec["cls_weighted"] = np.array(float(ClsCriterion()(la, soft, w)))
    perm = C.permutation(B, 18)
    with scripted_rng(randperm=[perm], beta=[0.81]):
        outs = label_smoothing(x, mu, ls, la, epsilon=0.1, disc_label=lab)
    for i, n in enumerate(["img", "mu", "sigma", "alpha", "label"]):
        rec["ls_" + n] = outs[i].numpy()
    rec["ls_lam"] = np.array(outs[5])
    with scripted_rng(randperm=[perm], beta=[0.42]):
        outs = mixup_vae_data(x, mu, ls, la, optimal_match=False)
    for i, n in enumerate(["img", "mu", "sigma", "alpha"]):
        rec["mx_" + n] = outs[i].numpy()
    with scripted_rng(beta=[0.42]):
        outs = mixup_vae_data(x, mu, ls, la, optimal_match=True)
    for i, n in enumerate(["img", "mu", "sigma", "alpha"]):
        rec["om_" + n] = outs[i].numpy()
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), **rec)
    print(tag, "ok")


def run_keys_case(tag):
    """state_dict keys + shapes of the reference model, in its order, for both data_parallel layouts (the reference wraps
    eleven sub-modules in nn.DataParallel inside the model: wideresnet.py:78-93, vae.py:108-132, decoder.py:63-64)."""
    import json
    VAE, *_ = import_reference()
    rec = {}
    for name, K in (("wideresnet-28-2", 10), ("wideresnet-10-1", 10), ("wideresnet-28-10", 100)):
        for dp in (False, True):
            model = VAE(encoder_name=name, num_input_channels=3, drop_rate=0, img_size=(32, 32), data_parallel=dp,
                        continuous_latent_dim=128, disc_latent_dim=K, sample_temperature=0.67, small_input=True)
            rec["%s|K=%d|dp=%d" % (name, K, int(dp))] = [[k, list(v.shape)] for k, v in model.state_dict().items()]
    with open(os.path.join(HERE, tag + ".json"), "w") as f:
        json.dump(rec, f, indent=0)
    print(tag, {k: len(v) for k, v in rec.items()})


def run_monitor_valid_case(tag, name, K):
    """(a) the Train/KL_Inference monitor of main_shot_vae.py:329-339 (KL(q(y|x) || smoothed one-hot label_u), computed on
    the unlabelled forward in train mode); (b) the body of valid() / test() (main_shot_vae.py:409-458 = :461-510) over two
    batches of unequal size: eval-mode forward, the three AverageMeter means, the `ELBO` scalar, top-1 / top-5."""
    import torch.nn.functional as F
    VAE, VAECriterion, *_ = import_reference()
    sys.path.insert(0, os.path.join(REF, "lib", "utils"))
    from lib.utils.avgmeter import AverageMeter
    model = VAE(encoder_name=name, num_input_channels=3, drop_rate=0, img_size=(32, 32), data_parallel=False,
                continuous_latent_dim=128, disc_latent_dim=K, sample_temperature=0.67, small_input=True)
    model.load_state_dict(C.make_state(name, K=K))
    rec = {}
    # (a)
    model.train()
    il, ll, iu, lu = C.make_batch(4, 6, K)
    nz = C.make_noise(4, 6, K)
    kl_inferences = AverageMeter()
    with scripted_rng(randn=[nz["eps3"]], rand=[nz["u3"]]):
        reconstruction_u, norm_mean_u, norm_log_sigma_u, disc_log_alpha_u = model(iu)
    batch_size_u = iu.size(0)
    with torch.no_grad():
        label_smooth_u = torch.zeros(batch_size_u, K).scatter_(1, lu.view(-1, 1), 1 - 0.001 - 0.001 / (K - 1))
        label_smooth_u = label_smooth_u + torch.ones(label_smooth_u.size()) * 0.001 / (K - 1)
        disc_alpha_u = torch.exp(disc_log_alpha_u)
        inference_kl = disc_alpha_u * disc_log_alpha_u - disc_alpha_u * torch.log(label_smooth_u)
    kl_inferences.update(float(torch.sum(inference_kl) / batch_size_u), batch_size_u)
    rec["kl_inference"] = np.array(kl_inferences.avg)
    # (b): a fresh model (the train-mode forward above moved the running statistics)
    model.load_state_dict(C.make_state(name, K=K))
    for bce, x_sigma in ((True, 1.0), (False, 0.5)):
        elbo_criterion = VAECriterion(discrete_dim=K, x_sigma=x_sigma, bce_reconstruction=bce)
        continuous_kl_losses, discrete_kl_losses = AverageMeter(), AverageMeter()
        mse_losses, elbo_losses = AverageMeter(), AverageMeter()
        model.eval()
        all_score, all_label = [], []
        for i, B in enumerate((6, 4)):
            image, label, _, _ = C.make_batch(B, B, K, stream0=7500 + 10 * i)
            noise = C.make_noise(B, B, K, stream0=9500 + 100 * i)
            label_onehot = torch.zeros(label.size(0), K).scatter_(1, label.view(-1, 1), 1)
            batch_size = image.size(0)
            with torch.no_grad(), scripted_rng(randn=[noise["eps3"][:B]], rand=[noise["u3"][:B]]):
                reconstruction, norm_mean, norm_log_sigma, disc_log_alpha, *_ = model(image)
            reconstruct_loss, continuous_kl_loss, discrete_kl_loss = elbo_criterion(image, reconstruction, norm_mean,
                                                                                    norm_log_sigma, disc_log_alpha)
            mse_loss = F.mse_loss(torch.sigmoid(reconstruction.detach()), image.detach(), reduction="sum") / (
                2 * image.size(0) * (x_sigma ** 2))
            mse_losses.update(float(mse_loss), image.size(0))
            all_score.append(torch.exp(disc_log_alpha))
            all_label.append(label_onehot)
            continuous_kl_losses.update(float(continuous_kl_loss.item()), batch_size)
            discrete_kl_losses.update(float(discrete_kl_loss.item()), batch_size)
            elbo_losses.update(float(mse_loss + 0.01 * (continuous_kl_loss + discrete_kl_loss)), image.size(0))
        all_score = torch.cat(all_score, dim=0).detach()
        all_label = torch.cat(all_label, dim=0).detach()
        _, y_true = torch.topk(all_label, k=1, dim=1)
        _, y_pred = torch.topk(all_score, k=5, dim=1)
        top1 = float(torch.sum(y_true == y_pred[:, :1]).item()) / y_true.size(0)
        top5 = float(torch.sum(y_true == y_pred).item()) / y_true.size(0)
        pre = "valid_bce%d." % int(bce)
        rec[pre + "klc"] = np.array(continuous_kl_losses.avg)
        rec[pre + "kld"] = np.array(discrete_kl_losses.avg)
        rec[pre + "mse"] = np.array(mse_losses.avg)
        rec[pre + "elbo"] = np.array(elbo_losses.avg)
        rec[pre + "top1"] = np.array(top1)
        rec[pre + "top5"] = np.array(top5)
        rec[pre + "score"] = all_score.numpy()
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), **rec)
    print(tag, {k: (float(v) if v.ndim == 0 else v.shape) for k, v in rec.items()})


def run_ssl_sampler_case(tag):
    """lib/dataloader.py:142-190 -- get_cifar10_ssl_sampler / get_cifar100_ssl_sampler, the semi-supervised index split of
    main_shot_vae.py:156-167 -- imported from the reference and driven with SCRIPTED torch.randperm draws.  The module's top-level
    `from torchvision import transforms, datasets` (torchvision is not installed here; the two functions use torch alone) is
    satisfied by an empty stub module IN THIS GENERATOR ONLY.  Records labels, the per-class permutations and the three index lists
    (SubsetRandomSampler.indices) of each sampler."""
    import types
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tv.transforms, tv.datasets = types.ModuleType("torchvision.transforms"), types.ModuleType("torchvision.datasets")
        sys.modules.update({"torchvision": tv, "torchvision.transforms": tv.transforms, "torchvision.datasets": tv.datasets})
    sys.path.insert(0, REF)
    from lib.dataloader import get_cifar10_ssl_sampler, get_cifar100_ssl_sampler
    rec = {}
    g = torch.Generator().manual_seed(20260)
    for name, fn, K, N, nv, na in (("c10", get_cifar10_ssl_sampler, 10, 230, 5, 4), ("c100", get_cifar100_ssl_sampler, 100, 900, 2, 3),
                                   ("c10_ragged", get_cifar10_ssl_sampler, 10, 97, 3, 20)):      # annotated > what a class has left
        labels = torch.randint(0, K, (N,), generator=g)
        counts = [int((labels == c).sum()) for c in range(K)]
        perms = [torch.randperm(n, generator=g) for n in counts]
        with scripted_rng(randperm=perms):
            sv, sl, su = fn(labels, nv, na, K)
        rec[name + ".labels"] = labels.numpy()
        rec[name + ".perm_cat"] = torch.cat(perms).numpy()
        rec[name + ".args"] = np.array([nv, na, K])
        rec[name + ".valid"] = np.array(list(sv.indices), dtype=np.int64)
        rec[name + ".train_l"] = np.array(list(sl.indices), dtype=np.int64)
        rec[name + ".train_u"] = np.array(list(su.indices), dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), **rec)
    print(tag, {k: v.shape for k, v in rec.items()})


if __name__ == "__main__":
    assert os.path.isdir(REF), "reference not mounted; goldens are generated in the build container"
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == "m2":          # only the M2 fixture
        run_m2_case("ref_m2_step_wrn10_1", "wideresnet-10-1", 10, 6)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "extra":       # only the round-2 fixtures (keys, monitor KL, valid() metrics)
        run_keys_case("ref_state_keys")
        run_monitor_valid_case("ref_monitor_valid_wrn10_1", "wideresnet-10-1", 10)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "round4":      # only the round-4 fixtures: larger, ragged batches and --om with B_l = B_u
        run_step_case("ref_step_wrn28_2_b16_24", "wideresnet-28-2", 10, 16, 24, True)
        run_step_case("ref_step_wrn28_2_om_b16", "wideresnet-28-2", 10, 16, 16, True, om=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ssl":         # only the round-6 fixture: the semi-supervised samplers
        run_ssl_sampler_case("ref_ssl_samplers")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "smooth":      # only the smooth-ELBO fixtures
        run_smooth_case("ref_smooth_svhn", "svhn", 6, 4)
        run_smooth_case("ref_smooth_mnist", "mnist", 4, 6)
        sys.exit(0)
    run_fn_cases("ref_functions")
    run_eval_case("ref_eval_wrn10_1", "wideresnet-10-1", 10, 4)
    run_step_case("ref_step_wrn10_1_br", "wideresnet-10-1", 10, 4, 6, True, steps=2)
    run_step_case("ref_step_wrn10_1_om", "wideresnet-10-1", 10, 4, 6, True, om=True)
    run_step_case("ref_step_wrn28_2_br", "wideresnet-28-2", 10, 4, 4, True)
    run_step_case("ref_step_wrn28_2_mse", "wideresnet-28-2", 10, 4, 4, False, x_sigma=0.5)
    run_step_case("ref_step_wrn28_10_k100", "wideresnet-28-10", 100, 2, 2, True, dmi=4.6)
    run_step_case("ref_step_wrn28_2_b16_24", "wideresnet-28-2", 10, 16, 24, True)
    run_step_case("ref_step_wrn28_2_om_b16", "wideresnet-28-2", 10, 16, 16, True, om=True)
    run_m2_case("ref_m2_step_wrn10_1", "wideresnet-10-1", 10, 6)
    run_smooth_case("ref_smooth_svhn", "svhn", 6, 4)
    run_smooth_case("ref_smooth_mnist", "mnist", 4, 6)
    run_keys_case("ref_state_keys")
    run_monitor_valid_case("ref_monitor_valid_wrn10_1", "wideresnet-10-1", 10)
    run_ssl_sampler_case("ref_ssl_samplers")
