"""The SHOT-VAE training step (body of the loop at main_shot_vae.py:280-366) on the HIP path, plus the
per-epoch schedule (:270-279, :518-520)."""
import math

import numpy as np
import torch

from . import dp
from .criterion import continuous_posterior_loss
from .mixup import _lerp, device_permutation, label_smoothing, mixup_vae_data, optimal_match_index
from .steploss import TERMS, shot_loss_step_groups
from .trace import step_range


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


def inference_kl(disc_log_alpha_u, label_u):
    """The Train/KL_Inference monitor of main_shot_vae.py:330-339: KL(q(y|x) || smoothed one-hot of the (held-back)
    label of the unlabelled batch), mean over the batch.  A device scalar -- the reference's float() (a device-to-host
    sync every step) is left to the caller, so that the step stays asynchronous / capturable."""
    with torch.no_grad():
        B, K = disc_log_alpha_u.shape
        smooth = torch.zeros(B, K, device=disc_log_alpha_u.device).scatter_(
            1, label_u.view(-1, 1), 1 - 0.001 - 0.001 / (K - 1))
        smooth = smooth + 0.001 / (K - 1)
        alpha = torch.exp(disc_log_alpha_u)
        return (alpha * disc_log_alpha_u - alpha * torch.log(smooth)).sum() / B


def _bucketed(model, distributed):
    """distributed = "bucketed": the decoder-first two-bucket exchange (dp.DecoderFirstAllReduce); True: one all-reduce"""
    if distributed != "bucketed":
        return None
    ar = getattr(model, "_bucket_allreduce", None)
    if ar is None:
        ar = model._bucket_allreduce = dp.DecoderFirstAllReduce(model)
    return ar


def apply_update(model, optimizer, distributed=False):
    """Gradient exchange + optimizer step + zero_grad (main_shot_vae.py:365-366).  With N ranks the flat gradient buffer
    holds the SUM over ranks after the all-reduce; FlatSGD folds the 1/N into its kernel, any other optimizer (a plain
    torch.optim.SGD over model.parameters()) gets the buffer scaled first, so the effective learning rate never depends
    on the optimizer class.  distributed: False | True (one all-reduce of the flat buffer) | "bucketed" (decoder bucket
    started during the backward by the step, encoder bucket here)."""
    grad = model.flat_parameters()[1]
    ar = _bucketed(model, distributed)
    if ar is not None:
        scale = ar.finish()
    else:
        scale = dp.all_reduce_gradients(grad) if distributed else 1.0
    if hasattr(optimizer, "_steps"):
        optimizer.step(grad_scale=scale)
    else:
        if scale != 1.0:
            grad.mul_(scale)
        optimizer.step()
    optimizer.zero_grad()


class DeviceRng:
    """Device-side source of the step's host-RNG draws, so that the step can be captured into a hipGraph: the two
    mixup coefficients come from tables of numpy Beta draws (the reference's distributions: Beta(eps,eps) for label
    smoothing, mixup.py:31; Beta(2,2) for mixup, mixup.py:7) indexed by a device counter, the pairings from a
    device-side random permutation."""

    def __init__(self, device, epsilon=0.1, n=4096, seed=0):
        self.rs = np.random.RandomState(seed)          # the SAME seed on every rank: all ranks agree on the lambdas
        self.epsilon, self.n = epsilon, n
        self.lam_l = torch.empty(n, dtype=torch.float32, device=device)
        self.lam_u = torch.empty(n, dtype=torch.float32, device=device)
        self.counter = torch.zeros(1, dtype=torch.int64, device=device)
        self.refill()

    def refill(self):
        """Fresh draws into the same device tensors (outside a graph: GraphedTrainStep calls it every n replays, so the
        tables are never recycled)."""
        n = self.n
        tl = self.rs.beta(self.epsilon, self.epsilon, size=n) if self.epsilon > 0 else np.ones(n)
        self.lam_l.copy_(torch.tensor(tl, dtype=torch.float32))
        self.lam_u.copy_(torch.tensor(self.rs.beta(2.0, 2.0, size=n), dtype=torch.float32))

    def next_lams(self):
        i = torch.remainder(self.counter, self.n)
        lam_l, lam_u = self.lam_l.index_select(0, i), self.lam_u.index_select(0, i)     # shape [1] each
        self.counter += 1
        return lam_l, lam_u


def train_step_overlapped(model, elbo_criterion, cls_criterion, optimizer, image_l, label_l, image_u, sch,
                          epsilon=0.1, distributed=False, device_rng=None, optimal_match=False, label_u=None):
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
    dl_l = dl_u = None
    if device_rng is not None:           # capturable: lambdas / pairings drawn on the device
        dl_l, dl_u = device_rng.next_lams()
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
            if device_rng is None:
                sm_img, sm_mu, sm_sigma, sm_alpha, sm_label, lam_l = label_smoothing(
                    image_l, mu1, ls1, la1, epsilon=epsilon, disc_label=label_l)
                lam_l0 = lam_l
            else:
                sm_img, sm_mu, sm_sigma, sm_alpha, sm_label, lam_l = label_smoothing(
                    image_l, mu1, ls1, la1, epsilon=epsilon, disc_label=label_l, lam=dl_l,
                    index=device_permutation(image_l.size(0), image_l.device))
                lam_l0 = lam_l.reshape(())
            sm_onehot = one_hot(sm_label, K)
        eng.defer_slot = 1
        rec2, mu2, ls2, la2, *_ = model(sm_img, True, label_l, sm_label, lam_l)
    with torch.cuda.stream(st[1]):
        recon_u, klc_u, kld_u = elbo_criterion(image_u, rec3, mu3, ls3, la3)
        elbo_u = recon_u + sch["kl_beta_c"] * torch.abs(klc_u - sch["cmi"]) + sch["kl_beta_d"] * torch.abs(kld_u - sch["dmi"])
        with torch.no_grad():
            kl_inf = inference_kl(la3, label_u) if label_u is not None else None
            if device_rng is None:
                mx_img, mx_mu, mx_sigma, mx_alpha, lam_u = mixup_vae_data(image_u, mu3, ls3, la3,
                                                                          optimal_match=optimal_match)
            else:
                mx_img, mx_mu, mx_sigma, mx_alpha, lam_u = mixup_vae_data(
                    image_u, mu3, ls3, la3, optimal_match=optimal_match, lam=dl_u,
                    index=None if optimal_match else device_permutation(image_u.size(0), image_u.device))
        eng.defer_slot = 3
        rec4, mu4, ls4, la4, *_ = model(mx_img)
    eng.defer_slot = None
    model._last_lams = (lam_l0, lam_u)
    with torch.cuda.stream(st[0]):
        disc_post_l = lam_l0 * cls_criterion(la2, onehot_l) + (1 - lam_l0) * cls_criterion(la2, sm_onehot)
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
        apply_update(model, optimizer, distributed)
    if label_u is not None:
        return loss_sup.detach(), loss_unsup.detach(), kl_inf
    return loss_sup.detach(), loss_unsup.detach()


def _launch_plan(ragged, optimal_match):
    """Which forwards of the step ((1) labelled, (2) smoothed labelled, (3) unlabelled, (4) mixed unlabelled) share a batched
    launch sequence.  Inside a launch the forwards whose reconstruction enters the loss ((1), (3)) come first."""
    if not ragged and not optimal_match:
        return [[1, 3, 2, 4]]                       # the usual step: ONE launch sequence, four BatchNorm groups
    if not ragged:
        return [[1, 3, 2], [4]]                     # --om: the pairing of (4) needs mu / log_sigma of (3) (mixup.py:9-18)
    if not optimal_match:
        return [[1, 2], [3, 4]]                     # B_l != B_u (the last labelled batch of an epoch): one launch per loader
    return [[1, 2], [3], [4]]


def _input_stream(model, dev):
    """the stream the input side of a grouped step is issued on (one per model and device)"""
    pool = getattr(model, "_input_streams", None)
    if pool is None:
        pool = model._input_streams = {}
    s = pool.get(dev)
    if s is None:
        s = pool[dev] = torch.cuda.Stream(device=dev)
    return s


def train_step_grouped(model, elbo_criterion, cls_criterion, optimizer, image_l, label_l, image_u, sch, epsilon=0.1,
                       distributed=False, device_rng=None, label_u=None, return_outputs=False, optimal_match=False,
                       input_stream=False):
    """The same step with the four forwards as batched launch sequences and no autograd graph.
    Legal because, without --om, the INPUTS of the mixed forwards (2) and (4) depend only on the raw images, a pairing and
    lambda (mixup.py:22,36) -- only their loss TARGETS depend on the outputs of (1) and (3) -- and all four use the same
    weights (optimizer.step comes last, main_shot_vae.py:365).  Each forward keeps its own BatchNorm batch statistics
    (groups of the batched launches) and the running statistics receive the four momentum updates in the reference's
    order; the backward accumulates the same gradient sum as the reference's two backward() calls.

    B_l == B_u, no --om: ONE launch sequence of four groups -- a quarter of the launches at four times the rows.
    --om (`optimal_match`): groups (1)(3)(2) batched, then the pairing kernel on the outputs of (3), then (4) alone.
    B_l != B_u (main_shot_vae.py:280 zips a 4 000-label loader, 7 x 512 + 416, with the unlabelled one): one launch
    sequence per loader, (1)(2) and (3)(4).  Both together: (1)(2), (3), (4).
    Host RNG (model.rng == "host") is consumed in the reference's order, so identical seeds give identical noise.

    input_stream=True (the usual step only): the INPUT side -- noise, pairings, both mixed image batches, the concatenation
    and the NHWC conversion, ~90 us of small kernels that depend on nothing but the batch -- is issued on a stream of its
    own, so that it runs beside whatever the main stream still has queued (the previous step's backward: the host issues a
    step in a third of the time the GPU takes) instead of in front of the first convolution.  The caller guarantees that
    image_l / label_l / image_u are COMPLETE when the step is called (resident tensors, or produced on another stream
    that has been synchronised with): the input stream does not wait for the main stream."""
    plan = model._plan
    K, ldc = plan.K, plan.ldc
    Bl, Bu = image_l.size(0), image_u.size(0)
    rows = {1: Bl, 2: Bl, 3: Bu, 4: Bu}
    dev = image_l.device
    eng = model._engine
    launches = _launch_plan(Bl != Bu, optimal_match)
    main = torch.cuda.current_stream() if image_l.is_cuda else None
    side_in = None
    if (input_stream and main is not None and len(launches) == 1 and not torch.cuda.is_current_stream_capturing()
            and eng.prof_tags is None):
        side_in = _input_stream(model, dev)
        eng.ensure_packs()                    # (the weight re-pack belongs to the main stream, behind the optimizer step)
        torch.cuda.set_stream(side_in)
    try:
        with step_range("inputs"):
            prep = _grouped_inputs(model, image_l, label_l, image_u, epsilon, device_rng, optimal_match, launches,
                                   side_in is not None)
    finally:
        if side_in is not None:
            torch.cuda.set_stream(main)
    if side_in is not None:
        main.wait_stream(side_in)
        for t in prep["tensors"]:                # allocated on the input stream's pool, used (and freed) on the main stream
            if torch.is_tensor(t):
                t.record_stream(main)
    return _grouped_body(model, elbo_criterion, cls_criterion, optimizer, image_l, label_l, image_u, sch, distributed,
                         label_u, return_outputs, optimal_match, launches, prep)


def _grouped_inputs(model, image_l, label_l, image_u, epsilon, device_rng, optimal_match, launches, whole):
    """The input side of a grouped step: every random draw in the reference's order, the smoothed labelled batch and -- for
    the single-launch plan when `whole` -- the mixed unlabelled batch, the concatenated images and their NHWC16 form."""
    plan = model._plan
    K, ldc = plan.K, plan.ldc
    Bl, Bu = image_l.size(0), image_u.size(0)
    rows = {1: Bl, 2: Bl, 3: Bu, 4: Bu}
    dev = image_l.device
    eng = model._engine
    e_all = u_all = None
    # ---- every random draw of the step, in the reference's order ---------------------------------
    # The reconstructions of the mixed forwards are dead values in the reference (main_shot_vae.py:311,356: `*_`): their last
    # ConvTranspose and their whole decoder backward are skipped (forward_groups(rec_groups=...)); BatchNorm running
    # statistics still receive the four updates in the reference's order (update_order / defer slots).
    eps, u = {}, {}
    perm_u = None
    device_noise = not (device_rng is None and model.rng == "host")
    if not device_noise:
        eps[1] = torch.randn(Bl, ldc)
        lam_l = np.random.beta(epsilon, epsilon) if epsilon > 0 else 1
        perm_l = torch.randperm(Bl).to(dev)
        eps[2], eps[3], u[3] = torch.randn(Bl, ldc), torch.randn(Bu, ldc), torch.rand(Bu, K)
        lam_u = np.random.beta(2.0, 2.0)
        if not optimal_match:
            perm_u = torch.randperm(Bu).to(dev)
        eps[4], u[4] = torch.randn(Bu, ldc), torch.rand(Bu, K)
        for k in eps:
            eps[k] = eps[k].to(dev)
        for k in u:
            u[k] = u[k].to(dev)
    else:
        e_all = torch.randn(2 * (Bl + Bu), ldc, device=dev)
        u_all = torch.rand(2 * (Bl + Bu), K, device=dev)
        o = 0
        for k in (1, 3, 2, 4):                              # (the order of the single launch: its tensors are used as they are)
            eps[k], u[k] = e_all[o:o + rows[k]], u_all[o:o + rows[k]]
            o += rows[k]
        if Bl == Bu:
            perm_l, perm_u = device_permutation(Bl, dev, 2)      # both pairings: one key draw, one launch
        else:
            perm_l, perm_u = device_permutation(Bl, dev), device_permutation(Bu, dev)
        if device_rng is not None:
            lam_l, lam_u = device_rng.next_lams()          # device scalars: capturable
        else:
            lam_l = np.random.beta(epsilon, epsilon) if epsilon > 0 else 1
            lam_u = np.random.beta(2.0, 2.0)
    perm_l = perm_l.long().contiguous()
    model._last_lams = (lam_l, lam_u)          # (data-parallel runs check that every rank used the same pair)
    with torch.no_grad():
        sm_img = _lerp(image_l, perm_l, lam_l, False)                        # mixup.py:36
        sm_label = label_l[perm_l]
    images = {1: image_l, 2: sm_img, 3: image_u}
    specs = {1: dict(disc_label=label_l), 3: dict(), 4: dict(),
             2: dict(mixup=True, disc_label=label_l, disc_pseudo_label=sm_label, mixup_lam=lam_l)}
    mx_img = image_cat = x16 = None
    if whole:
        with torch.no_grad():
            perm_u = perm_u.long().contiguous()
            mx_img = images[4] = _lerp(image_u, perm_u, lam_u, False)            # mixup.py:22
            image_cat = torch.cat([images[k].float() for k in launches[0]])
            x16 = eng.to_nhwc16(image_cat)
    return dict(eps=eps, u=u, e_all=e_all, u_all=u_all, device_noise=device_noise, perm_l=perm_l, perm_u=perm_u, lam_l=lam_l,
                lam_u=lam_u, sm_img=sm_img, sm_label=sm_label, images=images, specs=specs, mx_img=mx_img, image_cat=image_cat,
                x16=x16, tensors=[e_all, u_all, perm_l, perm_u, sm_img, sm_label, mx_img, image_cat, x16, lam_l, lam_u] +
                list(eps.values()) + list(u.values()))


def _grouped_body(model, elbo_criterion, cls_criterion, optimizer, image_l, label_l, image_u, sch, distributed, label_u,
                  return_outputs, optimal_match, launches, prep):
    """forward(s), loss stage, backward(s) and the update of a grouped step on the prepared inputs"""
    plan = model._plan
    K, ldc = plan.K, plan.ldc
    Bl, Bu = image_l.size(0), image_u.size(0)
    rows = {1: Bl, 2: Bl, 3: Bu, 4: Bu}
    dev = image_l.device
    eng = model._engine
    eps, u, e_all, u_all, device_noise = prep["eps"], prep["u"], prep["e_all"], prep["u_all"], prep["device_noise"]
    perm_l, perm_u, lam_l, lam_u = prep["perm_l"], prep["perm_u"], prep["lam_l"], prep["lam_u"]
    sm_img, images, specs, mx_img = prep["sm_img"], prep["images"], prep["specs"], prep["mx_img"]
    # forward, loss stage and backward are driven from THIS thread, without an autograd graph: the step's structure is
    # fixed ((loss_sup + loss_unsup).backward() = upstream gradients 1), and the autograd engine's worker-thread hand-over
    # left the GPU idle between the loss kernels and the network's backward
    outs, ctxs = {}, []
    for li, ids in enumerate(launches):
        if 4 in ids and mx_img is None:
            with torch.no_grad():
                if optimal_match:                                            # mixup.py:9-18 on the outputs of forward (3)
                    perm_u = optimal_match_index(outs[3][1], outs[3][2])
                perm_u = perm_u.long().contiguous()
                mx_img = images[4] = _lerp(image_u, perm_u, lam_u, False)    # mixup.py:22
        B = rows[ids[0]]
        zu = None
        us = []
        for k in ids:
            if k in u and specs[k].get("disc_label") is None:
                us.append(u[k])
            else:
                zu = torch.zeros(B, K, device=dev) if zu is None else zu
                us.append(zu)
        if device_noise and len(launches) == 1:     # drawn in this launch's group order (uniform rows of label groups are unused)
            e_cat, u_cat = e_all, u_all
        else:
            e_cat = torch.cat([eps[k] for k in ids]) if len(ids) > 1 else eps[ids[0]]
            u_cat = torch.cat(us) if len(ids) > 1 else us[0]
        nrec = sum(1 for k in ids if k in (1, 3))
        order = sorted(range(len(ids)), key=lambda j: ids[j])                # running statistics: the reference's forward order
        if len(launches) > 1:
            eng.defer_slot = li          # (launch li holds forwards that all precede those of launch li + 1)
        try:
            with step_range("forward"):
                rec, mu, ls, la, fctx = model.forward_groups_direct([images[k] for k in ids], [specs[k] for k in ids], eps=e_cat,
                                                                    u=u_cat, rec_groups=nrec, update_order=order,
                                                                    image_cat=prep["image_cat"] if len(launches) == 1 else None,
                                                                    x16=prep["x16"] if len(launches) == 1 else None)
        finally:
            eng.defer_slot = None
        ctxs.append((ids, fctx, rec, mu, ls, la))
        for j, k in enumerate(ids):
            outs[k] = (rec[j * B:(j + 1) * B] if j < nrec else None, mu[j * B:(j + 1) * B], ls[j * B:(j + 1) * B],
                       la[j * B:(j + 1) * B])
    if len(launches) > 1:
        eng.apply_pending()              # BatchNorm running statistics: slots in launch order, groups in forward order
    kl_inference = inference_kl(outs[3][3], label_u) if label_u is not None else None            # :330-339 (monitor)
    # the loss stage (steploss.py): 9 launches forward, 6 backward, no tensor algebra in between        :289-323, :340-363
    grads, per_launch = {}, []
    for ids, fctx, rec, mu, ls, la in ctxs:
        B = rows[ids[0]]
        d = (torch.empty_like(rec) if rec is not None else None, torch.empty_like(mu), torch.empty_like(ls), torch.empty_like(la))
        per_launch.append(d)
        for j, k in enumerate(ids):
            grads[k] = (d[0][j * B:(j + 1) * B] if outs[k][0] is not None else None,) + tuple(t[j * B:(j + 1) * B] for t in d[1:])
    with step_range("loss"):
        terms = shot_loss_step_groups(outs, grads, image_l, image_u, label_l, perm_l, perm_u, lam_l, lam_u, sch,
                                      bce=elbo_criterion.bce_reconstruction, x_sigma=elbo_criterion.x_sigma)
    loss_sup, loss_unsup = terms[10], terms[11]
    # backward(s): launches without a reconstruction first; the decoder-first gradient bucket of a data-parallel step is armed
    # for the LAST backward that runs the decoder (its gradients are complete once that one has issued them)   :324 + :364
    seq = sorted(range(len(ctxs)), key=lambda i: per_launch[i][0] is not None)
    for n, i in enumerate(seq):
        if n == len(seq) - 1 and optimizer is not None and _bucketed(model, distributed) is not None:
            _bucketed(model, distributed).arm()
        with step_range("backward"):
            model.backward_direct(ctxs[i][1], *per_launch[i], own_grads=True)
    if optimizer is not None:
        with step_range("update"):
            apply_update(model, optimizer, distributed)
    if not return_outputs:
        if label_u is not None:
            return loss_sup.detach(), loss_unsup.detach(), kl_inference
        return loss_sup.detach(), loss_unsup.detach()
    res = {k: terms[i].detach() for i, k in enumerate(TERMS)}
    res.update(sm_img=sm_img, mx_img=mx_img, rec1=outs[1][0], rec3=outs[3][0], perm_u=perm_u)
    for i in (1, 2, 3, 4):
        for n, t in zip(("mu", "ls", "la"), outs[i][1:]):
            res["%s%d" % (n, i)] = t
    if label_u is not None:
        res["kl_inference"] = kl_inference
    return {k: v.detach() for k, v in res.items()}        # (rec2 / rec4 are not computed: dead values of the reference step)


def train_step(model, elbo_criterion, cls_criterion, optimizer, image_l, label_l, image_u, sch, epsilon=0.1,
               optimal_match=False, distributed=False, return_outputs=False, label_u=None):
    """One step: 4 forwards, 2 backwards, (all-reduce,) SGD.  Inputs are device tensors.
    Returns the two scalar losses (device tensors) -- with `label_u` (the held-back labels of the unlabelled batch) also
    the Train/KL_Inference monitor of main_shot_vae.py:330-339 -- and optionally every intermediate the parity tests
    compare against the oracle."""
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
    kl_inference = inference_kl(la3, label_u) if label_u is not None else None      # :330-339
    recon_u, klc_u, kld_u = elbo_criterion(image_u, rec3, mu3, ls3, la3)
    prior_u = sch["kl_beta_c"] * torch.abs(klc_u - sch["cmi"]) + sch["kl_beta_d"] * torch.abs(kld_u - sch["dmi"])
    elbo_u = recon_u + prior_u
    with torch.no_grad():                                                    # :348-355
        # (--om: the pairing is computed here only so that return_outputs can report it; mixup_vae_data draws nothing else
        #  in front of lambda either way, mixup.py:30-39)
        perm_u = optimal_match_index(mu3, ls3) if optimal_match else None
        mx_img, mx_mu, mx_sigma, mx_alpha, lam_u = mixup_vae_data(image_u, mu3, ls3, la3, index=perm_u)
    model._last_lams = (lam_l, lam_u)
    # (4) mixed unlabelled forward                                            :356-364
    rec4, mu4, ls4, la4, *_ = model(mx_img)
    disc_post_u = cls_criterion(la4, mx_alpha)
    cont_post_u = continuous_posterior_loss(mu4, ls4, mx_mu, mx_sigma)
    elbo_u = elbo_u + sch["kl_beta_c"] * sch["pwm"] * cont_post_u
    loss_unsup = sch["ew"] * elbo_u + sch["ucw"] * disc_post_u
    if optimizer is not None and _bucketed(model, distributed) is not None:
        _bucketed(model, distributed).arm()               # the LAST backward of the step (the first one's gradients are in)
    loss_unsup.backward()
    # gradient exchange + update                                              :365-366
    if optimizer is not None:
        apply_update(model, optimizer, distributed)
    if not return_outputs:
        if label_u is not None:
            return loss_sup.detach(), loss_unsup.detach(), kl_inference
        return loss_sup.detach(), loss_unsup.detach()
    loc = dict(locals())
    keys = ["recon_l", "klc_l", "kld_l", "recon_u", "klc_u", "kld_u", "disc_post_l", "cont_post_l", "disc_post_u",
            "cont_post_u", "loss_sup", "loss_unsup", "sm_img", "mx_img"] + \
           ["%s%d" % (n, i) for i in (1, 2, 3, 4) for n in ("rec", "mu", "ls", "la")] + \
           (["kl_inference"] if label_u is not None else []) + (["perm_u"] if perm_u is not None else [])
    return {k: loc[k].detach() for k in keys}


def m2_train_step(model, elbo_criterion, cls_criterion, optimizer, image_l, label_l, image_u, label_u, sch,
                  distributed=False, return_outputs=False):
    """One step of the M2 baseline loop (main_M2_vae.py:259-305) on the HIP path: the same model and criteria without
    the mixup forwards -- labelled forward with the one-hot label + cross-entropy on q(y|x), unlabelled forward with
    the Gumbel-softmax sample, two backward passes accumulating into the flat gradient buffer, (all-reduce,) SGD.
    Also returns the monitored KL(q(y|x) || smoothed label) of :285-291."""
    K = model._plan.K
    B = image_l.size(0)
    onehot_l = one_hot(label_l, K)
    rec1, mu1, ls1, la1 = model(image_l, disc_label=label_l)
    recon_l, klc_l, kld_l = elbo_criterion(image_l, rec1, mu1, ls1, la1)
    elbo_l = recon_l + sch["kl_beta_c"] * torch.abs(klc_l - sch["cmi"]) + sch["kl_beta_d"] * torch.abs(kld_l - sch["dmi"])
    disc_post_l = cls_criterion(la1, onehot_l)
    loss_sup = sch["ew"] * elbo_l + disc_post_l
    loss_sup.backward()
    rec3, mu3, ls3, la3 = model(image_u)
    kl_inference = inference_kl(la3, label_u)
    recon_u, klc_u, kld_u = elbo_criterion(image_u, rec3, mu3, ls3, la3)
    elbo_u = recon_u + sch["kl_beta_c"] * torch.abs(klc_u - sch["cmi"]) + sch["kl_beta_d"] * torch.abs(kld_u - sch["dmi"])
    loss_unsup = sch["ew"] * elbo_u
    loss_unsup.backward()
    if optimizer is not None:
        apply_update(model, optimizer, distributed)
    if not return_outputs:
        return loss_sup.detach(), loss_unsup.detach(), kl_inference
    loc = dict(locals())
    keys = ["recon_l", "klc_l", "kld_l", "recon_u", "klc_u", "kld_u", "disc_post_l", "kl_inference", "loss_sup",
            "loss_unsup", "rec1", "mu1", "ls1", "la1", "rec3", "mu3", "ls3", "la3"]
    return {k: loc[k].detach() for k in keys}


class GraphedTrainStep:
    """(Do not capture while an RCCL (`nccl`) process group has work in flight: its watchdog thread polls events, which is an
    error during another thread's stream capture on this stack -- bench.py issues eagerly in that case.)
    The two-stream step captured once into a hipGraph and replayed: ~1100 kernel launches per step stop costing
    ~12 ms of host time (measured: the eager step is host-bound below that).  The graph holds the weight re-packing,
    the four forwards, both backwards and the deferred BN running-stat updates; the gradient all-reduce and the SGD
    kernel stay outside (eager), so the collective is an ordinary RCCL call and lr can change without re-capturing.
    Needs rng="device".  Re-capture (build a new object) when the per-epoch schedule scalars change.
    The `warmup` eager steps of the constructor are REAL training steps on the construction batch (parameter, momentum and
    BatchNorm running-statistic updates): pass the first batch of the epoch, or warmup=1 with a throw-away learning rate.
    After them the lambda tables are redrawn and the device counter is reset, so replay k reads entry (k - 1) mod n and a
    refill happens exactly when the counter wraps (no entry is used twice)."""

    def __init__(self, model, elbo_criterion, cls_criterion, optimizer, image_l, label_l, image_u, sch, epsilon=0.1,
                 distributed=False, seed=0, warmup=2, optimal_match=False, label_u=None, schedule="grouped"):
        assert model.rng == "device", "graph capture needs device-side noise: VariationalAutoEncoder(..., rng='device')"
        # "grouped": the four forwards as batched launch sequences (train_step_grouped; also --om and B_l != B_u);
        # "two-stream": the labelled and the unlabelled branch on two HIP streams (train_step_overlapped)
        self.schedule = schedule
        self.model, self.opt, self.distributed = model, optimizer, distributed
        self.il, self.ll, self.iu = image_l.clone(), label_l.clone(), image_u.clone()
        self.lu = label_u.clone() if label_u is not None else None
        self.om = optimal_match
        self.rng = DeviceRng(image_l.device, epsilon, seed=seed)
        self.replays = 0
        self.args = (elbo_criterion, cls_criterion, sch, epsilon)
        self.stream = torch.cuda.Stream()
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            for _ in range(warmup):                 # also creates every per-stream cache the capture relies on
                self._body()
                self._update()
        cur.wait_stream(self.stream)
        torch.cuda.synchronize()
        self.rng.refill()                           # the warm-up steps consumed entries 0 .. warmup-1: fresh tables,
        self.rng.counter.zero_()                    # and replay k reads entry (k - 1) mod n
        model._engine.mark_dirty()                  # the captured sequence must start with the weight re-packing
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: other threads (RCCL's watchdog polling its events at N > 1, the autograd worker's allocator
        # calls) must not abort the capture
        with torch.cuda.graph(self.graph, stream=self.stream, capture_error_mode="thread_local"):
            self.losses = self._body()

    def _body(self):
        e, c, sch, eps = self.args
        eng = self.model._engine
        if self.schedule == "grouped":
            return train_step_grouped(self.model, e, c, None, self.il, self.ll, self.iu, sch, epsilon=eps,
                                      device_rng=self.rng, label_u=self.lu, optimal_match=self.om)
        keep = eng.wgrad_side_stream
        # nested side streams inside the two branch streams crash hipGraph instantiation on ROCm 7.2 (and add
        # nothing to the two-stream schedule, measured), so the captured body keeps wgrads on the branch streams
        eng.wgrad_side_stream = False
        try:
            return train_step_overlapped(self.model, e, c, None, self.il, self.ll, self.iu, sch, epsilon=eps,
                                         device_rng=self.rng, optimal_match=self.om, label_u=self.lu)
        finally:
            eng.wgrad_side_stream = keep

    def _update(self):
        apply_update(self.model, self.opt, self.distributed)

    def __call__(self, image_l=None, label_l=None, image_u=None, label_u=None):
        if image_l is not None:
            self.il.copy_(image_l)
            self.ll.copy_(label_l)
            self.iu.copy_(image_u)
            if label_u is not None and self.lu is not None:
                self.lu.copy_(label_u)
        if self.replays and self.replays % self.rng.n == 0:      # the device counter wrapped: this replay reads entry 0 again
            self.rng.refill()                                   # -> fresh Beta draws first (stream-ordered before the replay)
        self.replays += 1
        self.graph.replay()
        self._update()
        return self.losses
