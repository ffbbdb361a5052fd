"""The SHOT-VAE training step (body of the loop at main_shot_vae.py:280-366) on the HIP path, plus the
per-epoch schedule (:270-279, :518-520)."""
import math

import torch

from . import dp
from .criterion import continuous_posterior_loss
from .mixup import label_smoothing, mixup_vae_data


def alpha_schedule(epoch, max_epoch, alpha_max):
    """main_shot_vae.py:518-520"""
    return alpha_max * math.exp(-5 * (1 - min(1, epoch / max_epoch)) ** 2)


def schedule(epoch, epochs=600, cmi=0.0, dmi=2.3, kbmc=1e-3, kbmd=1e-3, akb=200, ewm=1e-3, aew=400, pwm=1.0,
             apw=200, wrd=1.0, wmf=0.4):
    """Scalars of main_shot_vae.py:270-279 (defaults: Cifar10 branch, dmi=2.3 from :139)."""
    return dict(cmi=alpha_schedule(epoch, akb, cmi), dmi=alpha_schedule(epoch, akb, dmi),
                ew=alpha_schedule(epoch, aew, ewm), kl_beta_c=alpha_schedule(epoch, akb, kbmc),
                kl_beta_d=alpha_schedule(epoch, akb, kbmd), pwm=alpha_schedule(epoch, apw, pwm),
                ucw=alpha_schedule(epoch, round(wmf * epochs), wrd))


def one_hot(label, K):
    return torch.zeros(label.shape[0], K, device=label.device).scatter_(1, label.view(-1, 1), 1)


def train_step_overlapped(model, elbo_criterion, cls_criterion, optimizer, image_l, label_l, image_u, sch,
                          epsilon=0.1, distributed=False):
    """Same step, the labelled branch ((1),(2), backward) and the unlabelled branch ((3),(4), backward) issued on two
    HIP streams: they are independent until the optimizer step (both only read the weights and add to the flat
    gradient buffer with atomics), so the latency-bound small kernels of one branch (decoder, heads, BN
    finalisation) run beside the other branch's convolutions."""
    K = model._plan.K
    cur = torch.cuda.current_stream()
    st = getattr(model, "_branch_streams", None)
    if st is None:
        st = model._branch_streams = (torch.cuda.Stream(), torch.cuda.Stream())
    eng = model._engine
    eng.ensure_packs()
    model._attach_grads()
    for s in st:
        s.wait_stream(cur)
    # BN running statistics: every forward defers its momentum update into its slot; they are applied after the
    # join in the reference's order (1)(2)(3)(4) (apply_pending), so the result does not depend on stream timing
    with torch.cuda.stream(st[0]):
        onehot_l = one_hot(label_l, K)
        eng.defer_slot = 0
        rec1, mu1, ls1, la1 = model(image_l, disc_label=label_l)
    with torch.cuda.stream(st[1]):
        eng.defer_slot = 2
        rec3, mu3, ls3, la3 = model(image_u)
    with torch.cuda.stream(st[0]):
        recon_l, klc_l, kld_l = elbo_criterion(image_l, rec1, mu1, ls1, la1)
        elbo_l = recon_l + sch["kl_beta_c"] * torch.abs(klc_l - sch["cmi"]) + sch["kl_beta_d"] * torch.abs(kld_l - sch["dmi"])
        with torch.no_grad():
            sm_img, sm_mu, sm_sigma, sm_alpha, sm_label, lam_l = label_smoothing(
                image_l, mu1, ls1, la1, epsilon=epsilon, disc_label=label_l)
            sm_onehot = one_hot(sm_label, K)
        eng.defer_slot = 1
        rec2, mu2, ls2, la2, *_ = model(sm_img, True, label_l, sm_label, lam_l)
    with torch.cuda.stream(st[1]):
        recon_u, klc_u, kld_u = elbo_criterion(image_u, rec3, mu3, ls3, la3)
        elbo_u = recon_u + sch["kl_beta_c"] * torch.abs(klc_u - sch["cmi"]) + sch["kl_beta_d"] * torch.abs(kld_u - sch["dmi"])
        with torch.no_grad():
            mx_img, mx_mu, mx_sigma, mx_alpha, lam_u = mixup_vae_data(image_u, mu3, ls3, la3)
        eng.defer_slot = 3
        rec4, mu4, ls4, la4, *_ = model(mx_img)
    eng.defer_slot = None
    with torch.cuda.stream(st[0]):
        disc_post_l = lam_l * cls_criterion(la2, onehot_l) + (1 - lam_l) * cls_criterion(la2, sm_onehot)
        elbo_l = elbo_l + sch["kl_beta_c"] * sch["pwm"] * continuous_posterior_loss(mu2, ls2, sm_mu, sm_sigma)
        loss_sup = sch["ew"] * elbo_l + disc_post_l
        loss_sup.backward()
    with torch.cuda.stream(st[1]):
        elbo_u = elbo_u + sch["kl_beta_c"] * sch["pwm"] * continuous_posterior_loss(mu4, ls4, mx_mu, mx_sigma)
        loss_unsup = sch["ew"] * elbo_u + sch["ucw"] * cls_criterion(la4, mx_alpha)
        loss_unsup.backward()
    for s in st:
        cur.wait_stream(s)
    eng.apply_pending()
    if optimizer is not None:
        scale = dp.all_reduce_gradients(model.flat_parameters()[1]) if distributed else 1.0
        optimizer.step(scale) if hasattr(optimizer, "_steps") else optimizer.step()
        optimizer.zero_grad()
    return loss_sup.detach(), loss_unsup.detach()


def train_step(model, elbo_criterion, cls_criterion, optimizer, image_l, label_l, image_u, sch, epsilon=0.1,
               optimal_match=False, distributed=False, return_outputs=False):
    """One step: 4 forwards, 2 backwards, (all-reduce,) SGD.  Inputs are device tensors.
    Returns the two scalar losses (device tensors) and optionally every intermediate the parity
    tests compare against the oracle."""
    K = model._plan.K
    Bl, Bu = image_l.size(0), image_u.size(0)
    onehot_l = one_hot(label_l, K)
    # (1) labelled forward                                                   :288-295
    rec1, mu1, ls1, la1 = model(image_l, disc_label=label_l)
    recon_l, klc_l, kld_l = elbo_criterion(image_l, rec1, mu1, ls1, la1)
    prior_l = sch["kl_beta_c"] * torch.abs(klc_l - sch["cmi"]) + sch["kl_beta_d"] * torch.abs(kld_l - sch["dmi"])
    elbo_l = recon_l + prior_l
    with torch.no_grad():                                                    # :297-310
        sm_img, sm_mu, sm_sigma, sm_alpha, sm_label, lam_l = label_smoothing(
            image_l, mu1, ls1, la1, epsilon=epsilon, disc_label=label_l)
        sm_onehot = one_hot(sm_label, K)
    # (2) mixed labelled forward                                              :311-324
    rec2, mu2, ls2, la2, *_ = model(sm_img, True, label_l, sm_label, lam_l)
    disc_post_l = lam_l * cls_criterion(la2, onehot_l) + (1 - lam_l) * cls_criterion(la2, sm_onehot)
    cont_post_l = continuous_posterior_loss(mu2, ls2, sm_mu, sm_sigma)
    elbo_l = elbo_l + sch["kl_beta_c"] * sch["pwm"] * cont_post_l
    loss_sup = sch["ew"] * elbo_l + disc_post_l
    loss_sup.backward()
    # (3) unlabelled forward                                                  :327-346
    rec3, mu3, ls3, la3 = model(image_u)
    recon_u, klc_u, kld_u = elbo_criterion(image_u, rec3, mu3, ls3, la3)
    prior_u = sch["kl_beta_c"] * torch.abs(klc_u - sch["cmi"]) + sch["kl_beta_d"] * torch.abs(kld_u - sch["dmi"])
    elbo_u = recon_u + prior_u
    with torch.no_grad():                                                    # :348-355
        mx_img, mx_mu, mx_sigma, mx_alpha, lam_u = mixup_vae_data(image_u, mu3, ls3, la3,
                                                                  optimal_match=optimal_match)
    # (4) mixed unlabelled forward                                            :356-364
    rec4, mu4, ls4, la4, *_ = model(mx_img)
    disc_post_u = cls_criterion(la4, mx_alpha)
    cont_post_u = continuous_posterior_loss(mu4, ls4, mx_mu, mx_sigma)
    elbo_u = elbo_u + sch["kl_beta_c"] * sch["pwm"] * cont_post_u
    loss_unsup = sch["ew"] * elbo_u + sch["ucw"] * disc_post_u
    loss_unsup.backward()
    # gradient exchange + update                                              :365-366
    if optimizer is not None:
        scale = dp.all_reduce_gradients(model.flat_parameters()[1]) if distributed else 1.0
        optimizer.step(scale) if hasattr(optimizer, "_steps") else optimizer.step()
        optimizer.zero_grad()
    if not return_outputs:
        return loss_sup.detach(), loss_unsup.detach()
    loc = dict(locals())
    keys = ["recon_l", "klc_l", "kld_l", "recon_u", "klc_u", "kld_u", "disc_post_l", "cont_post_l", "disc_post_u",
            "cont_post_u", "loss_sup", "loss_unsup", "sm_img", "mx_img"] + \
           ["%s%d" % (n, i) for i in (1, 2, 3, 4) for n in ("rec", "mu", "ls", "la")]
    return {k: loc[k].detach() for k in keys}
