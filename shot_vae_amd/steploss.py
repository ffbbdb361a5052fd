"""The loss stage of the SHOT-VAE step (main_shot_vae.py:289-323,340-363) as ONE autograd node over the C ABI.

The reference composes its two objectives from ~40 small tensor operations (criteria, abs, scalar products, the mixed
targets, two one-hot scatters) and autograd walks them again backwards: on the MI355X path that was ~65 micro-kernels per
step with the GPU idle between them (tools/step_timeline.py: 1.4 ms of gaps).  Here the stage is 9 launches forward (the
two ELBO reductions, one launch for every target of the mixed forwards, two ClsCriterion and two posterior reductions, the
scalar composition) and 7 backward (upstream scaling + the seven gradient kernels, each writing its slice of the network's
output gradients directly) -- the same kernels VAECriterion / ClsCriterion / continuous_posterior_loss use, the same
formulas; only the two label terms of :316-318 are merged into one soft label (ClsCriterion is linear in it)."""
import ctypes as C

import torch

from . import _lib as L

TERMS = ["recon_l", "klc_l", "kld_l", "recon_u", "klc_u", "kld_u", "disc_post_l", "cont_post_l", "disc_post_u", "cont_post_u",
         "loss_sup", "loss_unsup"]


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _lam(v):
    """(host float, device pointer or None) of a mixing coefficient given as a float or as a device scalar"""
    if torch.is_tensor(v):
        return 0.0, _p(v)
    return float(v), None


class _ShotLossFn(torch.autograd.Function):
    """rec [2B] = reconstructions of forwards (1), (3); mu / ls / la [4B] in the group order (1) (3) (2) (4)."""

    @staticmethod
    def forward(ctx, rec, mu, ls, la, image_l, image_u, label_l, perm_l, perm_u, lam_l, lam_u, sch, bce, x_sigma):
        for t in (rec, mu, ls, la, image_l, image_u):
            if not t.is_cuda:
                raise L.ShotVaeHipError("shot_vae_amd losses run on an MI355X only (no CPU fallback)")
        B = image_l.shape[0]
        D, K = mu.shape[1], la.shape[1]
        dev = mu.device
        rec, mu, ls, la = rec.contiguous(), mu.contiguous(), ls.contiguous(), la.contiguous()
        image_l, image_u = image_l.contiguous().float(), image_u.contiguous().float()
        npi = image_l[0].numel()
        f32 = dict(dtype=torch.float32, device=dev)
        terms = torch.zeros(12, **f32)
        coef = torch.empty(10, **f32)
        tgt = torch.empty(4 * B * D + 2 * B * K, **f32)           # sm_mu | sm_sigma | mx_mu | mx_sigma | lab_mix | mx_alpha
        sm_mu, sm_sigma, mx_mu, mx_sigma = (tgt[i * B * D:(i + 1) * B * D].view(B, D) for i in range(4))
        lab_mix = tgt[4 * B * D: 4 * B * D + B * K].view(B, K)
        mx_alpha = tgt[4 * B * D + B * K:].view(B, K)
        g = lambda t, i: t[i * B:(i + 1) * B]                      # group i of a [4B] / [2B] tensor (contiguous rows)
        st = _st()
        tp = terms.data_ptr()
        fp = lambda i: C.c_void_p(tp + 4 * i)
        # (1), (3): the ELBO terms                                                                   :289-295, :340-346
        L.call("sv_elbo_fwd", _p(image_l), _p(g(rec, 0)), npi, _p(g(mu, 0)), _p(g(ls, 0)), _p(g(la, 0)), B, D, K, int(bce),
               float(x_sigma), fp(0), st)
        L.call("sv_elbo_fwd", _p(image_u), _p(g(rec, 1)), npi, _p(g(mu, 1)), _p(g(ls, 1)), _p(g(la, 1)), B, D, K, int(bce),
               float(x_sigma), fp(3), st)
        # the targets of (2) and (4), detached                                                      :297-310, :348-355
        ll, ld = _lam(lam_l)
        lu, lud = _lam(lam_u)
        L.call("sv_shot_targets", _p(g(mu, 0)), _p(g(ls, 0)), _p(g(mu, 1)), _p(g(ls, 1)), _p(g(la, 1)), _p(label_l), _p(perm_l),
               _p(perm_u), ll, ld, lu, lud, B, D, K, _p(sm_mu), _p(sm_sigma), _p(lab_mix), _p(mx_mu), _p(mx_sigma), _p(mx_alpha),
               st)
        # (2): posterior terms of the smoothed labelled forward                                      :316-321
        L.call("sv_cls_fwd", _p(g(la, 2)), _p(lab_mix), None, B, K, fp(6), st)
        L.call("sv_post_fwd", _p(g(mu, 2)), _p(g(ls, 2)), _p(sm_mu), _p(sm_sigma), B, D, fp(7), st)
        # (4): posterior terms of the mixed unlabelled forward                                       :358-361
        L.call("sv_cls_fwd", _p(g(la, 3)), _p(mx_alpha), None, B, K, fp(8), st)
        L.call("sv_post_fwd", _p(g(mu, 3)), _p(g(ls, 3)), _p(mx_mu), _p(mx_sigma), B, D, fp(9), st)
        s = L.SvShotSchedule(*[float(sch[k]) for k in ("ew", "kl_beta_c", "kl_beta_d", "cmi", "dmi", "pwm", "ucw")])
        L.call("sv_shot_compose", _p(terms), C.byref(s), _p(coef), st)
        ctx.save_for_backward(rec, mu, ls, la, image_l, image_u, tgt, coef)
        ctx.cfg = (B, D, K, npi, bce, x_sigma)
        ctx.mark_non_differentiable(terms)
        return terms[10], terms[11], terms

    @staticmethod
    def backward(ctx, g_sup, g_unsup, _g_terms, unit_upstream=False):
        rec, mu, ls, la, image_l, image_u, tgt, coef = ctx.saved_tensors
        B, D, K, npi, bce, x_sigma = ctx.cfg
        dev = mu.device
        f32 = dict(dtype=torch.float32, device=dev)
        sm_mu, sm_sigma, mx_mu, mx_sigma = (tgt[i * B * D:(i + 1) * B * D] for i in range(4))
        lab_mix = tgt[4 * B * D: 4 * B * D + B * K]
        mx_alpha = tgt[4 * B * D + B * K:]
        g = lambda t, i: t[i * B:(i + 1) * B]
        st = _st()
        if unit_upstream:              # both upstream gradients are 1: the coefficients ARE the scaled gradients
            gvec = coef
        else:
            gvec = torch.empty(10, **f32)
            gs = g_sup.contiguous().float().view(1) if g_sup is not None else None
            gu = g_unsup.contiguous().float().view(1) if g_unsup is not None else None
            L.call("sv_shot_scale", _p(coef), _p(gs), _p(gu), _p(gvec), st)
        gp = gvec.data_ptr()
        fp = lambda i: C.c_void_p(gp + 4 * i)
        # every slice of the four gradients is written by exactly one kernel (=, not +=)
        d_rec, d_mu, d_ls, d_la = torch.empty_like(rec), torch.empty_like(mu), torch.empty_like(ls), torch.empty_like(la)
        for i, img in ((0, image_l), (1, image_u)):
            L.call("sv_elbo_bwd", _p(img), _p(g(rec, i)), npi, _p(g(mu, i)), _p(g(ls, i)), _p(g(la, i)), B, D, K, int(bce),
                   float(x_sigma), fp(3 * i), _p(g(d_rec, i)), _p(g(d_mu, i)), _p(g(d_ls, i)), _p(g(d_la, i)), st)
        L.call("sv_cls_bwd", _p(lab_mix), None, B, K, fp(6), _p(g(d_la, 2)), st)
        L.call("sv_post_bwd", _p(g(mu, 2)), _p(g(ls, 2)), _p(sm_mu), _p(sm_sigma), B, D, fp(7), _p(g(d_mu, 2)), _p(g(d_ls, 2)), st)
        L.call("sv_cls_bwd", _p(mx_alpha), None, B, K, fp(8), _p(g(d_la, 3)), st)
        L.call("sv_post_bwd", _p(g(mu, 3)), _p(g(ls, 3)), _p(mx_mu), _p(mx_sigma), B, D, fp(9), _p(g(d_mu, 3)), _p(g(d_ls, 3)), st)
        return (d_rec, d_mu, d_ls, d_la) + (None,) * 10


def shot_loss_step_groups(outs, grads, image_l, image_u, label_l, perm_l, perm_u, lam_l, lam_u, sch, bce=True, x_sigma=1.0):
    """The loss stage WITHOUT autograd, for a step that drives the network's backward itself (train_step_grouped): the
    forward reductions and, with upstream gradients 1 for both objectives (`(loss_sup + loss_unsup).backward()`), the
    gradients w.r.t. the network outputs -- 9 + 6 launches issued back to back by ONE call into the library
    (sv_shot_loss_step2: at ~5 us per kernel the stage is bound by the per-launch host time of a Python caller).
    outs[i] = (rec or None, mu, ls, la) of forward i in (1, 2, 3, 4), grads[i] = (d_rec or None, d_mu, d_ls, d_la): the
    tensors the gradients are WRITTEN to (row slices of whatever the caller's launches need; forwards (1), (2) have B_l
    rows, (3), (4) B_u rows).  Returns terms[12]."""
    Bl, Bu, D, K = image_l.shape[0], image_u.shape[0], outs[1][1].shape[1], outs[1][3].shape[1]
    for i in (1, 2, 3, 4):
        for t in outs[i] + grads[i]:
            if t is None:
                continue
            if not t.is_cuda:
                raise L.ShotVaeHipError("shot_vae_amd losses run on an MI355X only (no CPU fallback)")
            assert t.is_contiguous() and t.dtype == torch.float32 and t.shape[0] == (Bl if i < 3 else Bu), (i, t.shape, t.dtype)
    f32 = dict(dtype=torch.float32, device=image_l.device)
    image_l, image_u = image_l.contiguous().float(), image_u.contiguous().float()
    label_l = label_l.long().contiguous()
    terms = torch.zeros(12, **f32)
    scratch = torch.empty(10 + 2 * (Bl + Bu) * D + (Bl + Bu) * K, **f32)          # coef | the targets
    a = L.SvShotLossArgs2()
    for slot, i in enumerate((1, 3, 2, 4)):                 # the library's group order
        a.mu[slot], a.ls[slot], a.la[slot] = (t.data_ptr() for t in outs[i][1:])
        a.d_mu[slot], a.d_ls[slot], a.d_la[slot] = (t.data_ptr() for t in grads[i][1:])
    for slot, i in enumerate((1, 3)):
        a.rec[slot], a.d_rec[slot] = outs[i][0].data_ptr(), grads[i][0].data_ptr()
    a.image_l, a.image_u, a.label_l = image_l.data_ptr(), image_u.data_ptr(), label_l.data_ptr()
    a.perm_l, a.perm_u = perm_l.data_ptr(), perm_u.data_ptr()
    keep = []
    for name, v in (("lam_l", lam_l), ("lam_u", lam_u)):
        if torch.is_tensor(v):
            keep.append(v)
            setattr(a, name + "_dev", v.data_ptr())
        else:
            setattr(a, name, float(v))
    a.Bl, a.Bu, a.D, a.K, a.bce, a.n_per_img, a.x_sigma = Bl, Bu, D, K, int(bce), image_l[0].numel(), float(x_sigma)
    a.sch = L.SvShotSchedule(*[float(sch[k]) for k in ("ew", "kl_beta_c", "kl_beta_d", "cmi", "dmi", "pwm", "ucw")])
    a.terms, a.coef, a.tgt = terms.data_ptr(), scratch.data_ptr(), scratch.data_ptr() + 40
    L.call("sv_shot_loss_step2", C.byref(a), _st())
    return terms


def shot_loss_step(rec, mu, ls, la, image_l, image_u, label_l, perm_l, perm_u, lam_l, lam_u, sch, bce=True, x_sigma=1.0):
    """shot_loss_step_groups for the outputs of ONE batched launch of four equal groups in the order (1) (3) (2) (4)
    (rec: groups (1), (3) only).  Returns (terms[12], d_rec [2B], d_mu, d_ls, d_la [4B])."""
    B = image_l.shape[0]
    rec, mu, ls, la = rec.contiguous(), mu.contiguous(), ls.contiguous(), la.contiguous()
    d_rec, d_mu, d_ls, d_la = torch.empty_like(rec), torch.empty_like(mu), torch.empty_like(ls), torch.empty_like(la)
    g = lambda t, k: t[k * B:(k + 1) * B]
    outs, grads = {}, {}
    for slot, i in enumerate((1, 3, 2, 4)):
        outs[i] = (g(rec, slot) if slot < 2 else None, g(mu, slot), g(ls, slot), g(la, slot))
        grads[i] = (g(d_rec, slot) if slot < 2 else None, g(d_mu, slot), g(d_ls, slot), g(d_la, slot))
    terms = shot_loss_step_groups(outs, grads, image_l, image_u, label_l, perm_l, perm_u, lam_l, lam_u, sch, bce, x_sigma)
    return terms, d_rec, d_mu, d_ls, d_la


class _Ctx:
    """stand-in for the autograd context when the node's forward / backward are called directly"""

    def save_for_backward(self, *ts):
        self.saved_tensors = ts

    def mark_non_differentiable(self, *ts):
        pass


def shot_losses(rec, mu, ls, la, image_l, image_u, label_l, perm_l, perm_u, lam_l, lam_u, sch, bce=True, x_sigma=1.0):
    """(loss_supervised, loss_unsupervised, terms[12]) of one step from the batched outputs of the four forwards in the
    group order (1) (3) (2) (4) (rec: groups (1), (3) only); terms in the order of steploss.TERMS, detached."""
    return _ShotLossFn.apply(rec, mu, ls, la, image_l, image_u, label_l.long().contiguous(), perm_l, perm_u, lam_l, lam_u,
                             sch, bce, x_sigma)
