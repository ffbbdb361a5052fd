"""CPU oracle for the SHOT-VAE training hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain torch-CPU fp32 *restatement* of the reference algorithm
(FengHZ/SHOT-VAE).  It is written functionally over a flat ``state`` dict whose
keys are the reference's ``state_dict`` names (``data_parallel=False`` layout),
so that it shares nothing structurally with the reference's nn.Module code.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.  The product package (``shot_vae_amd/``) never does; the
product path fails loudly when the HIP extension is missing.

Parity pin: the reference has no tests or golden vectors of its own
(SURVEY.md §4), so this oracle is pinned against outputs of the reference
itself, generated in the build container by ``tests/golden/make_goldens.py``
(imports /root/reference with a ``.cuda()`` no-op shim) and committed under
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every case.

Reference citations (relative to /root/reference):
  encoder      shot_vae_model/wideresnet.py:8-114
  decoder      shot_vae_model/decoder.py:4-69
  assembly     shot_vae_model/vae.py:10-151
  criteria     lib/criterion.py:8-57, 93-108
  mixup        lib/utils/mixup.py:5-41, 93-99
  train step   main_shot_vae.py:261-366, 518-520
"""
import math
import re

import torch
import torch.nn.functional as F

LEAKY_SLOPE = 0.01      # nn.LeakyReLU default (wideresnet.py:28,33,40,91)
BN_EPS = 1e-5           # nn.BatchNorm2d default
BN_MOMENTUM = 0.1
GUMBEL_EPS = 1e-12      # vae.py:68


# --------------------------------------------------------------------------- #
# architecture description
# --------------------------------------------------------------------------- #
def parse_wideresnet(name):
    """'wideresnet-D-W' -> (depth, width, units_per_stage).  wideresnet.py:72-74,110."""
    nums = re.findall(r"\d+", name)
    depth, width = int(nums[0]), int(nums[1])
    if "wideresnet" not in name:
        raise NotImplementedError("{} not implemented".format(name))
    assert (depth - 4) % 6 == 0, "depth should be 6n+4"
    return depth, width, (depth - 4) // 6


def encoder_units(name):
    """List of (prefix, cin, cout, stride, has_shortcut_conv) for every residual unit."""
    _, width, n = parse_wideresnet(name)
    widths = [int(16 * width), int(32 * width), int(64 * width)]
    units = []
    cin = 16
    for s, w in enumerate(widths):
        for u in range(n):
            stride = 2 if (s > 0 and u == 0) else 1
            c_in = cin if u == 0 else w
            units.append((
                "feature_extractor.encoder.wideblock%d.wide_block.wideunit%d" % (s + 1, u + 1),
                c_in, w, stride, (c_in != w) or stride != 1))
        cin = w
    return units, widths[-1]


DECODER_CONVT = [0, 3, 6, 9, 12, 15]     # decoder.py:12-58 Sequential indices
DECODER_BN = [1, 4, 7, 10, 13]
DECODER_WIDTHS = [1024, 512, 256, 128, 64]


def state_shapes(name, in_ch=3, ldc=128, K=10, img=32):
    """Ordered {key: shape} of the reference state_dict (data_parallel=False)."""
    units, cfeat = encoder_units(name)
    sh = {}

    def bn(prefix, c):
        sh[prefix + ".weight"] = (c,)
        sh[prefix + ".bias"] = (c,)
        sh[prefix + ".running_mean"] = (c,)
        sh[prefix + ".running_var"] = (c,)
        sh[prefix + ".num_batches_tracked"] = ()

    sh["feature_extractor.encoder.pre_process.conv0.weight"] = (16, in_ch, 3, 3)
    sh["feature_extractor.encoder.pre_process.conv0.bias"] = (16,)
    for p, ci, co, stride, sc in units:
        bn(p + ".f_block.norm1", ci)
        sh[p + ".f_block.conv1.weight"] = (co, ci, 3, 3)
        bn(p + ".f_block.norm2", co)
        sh[p + ".f_block.conv2.weight"] = (co, co, 3, 3)
        if sc:
            bn(p + ".i_block.norm", ci)
            sh[p + ".i_block.conv.weight"] = (co, ci, 1, 1)
    bn("feature_extractor.encoder.transition.norm", cfeat)
    sh["continuous_inference.mean.fc.weight"] = (ldc, cfeat)
    sh["continuous_inference.mean.fc.bias"] = (ldc,)
    sh["continuous_inference.log_sigma.fc.weight"] = (ldc, cfeat)
    sh["continuous_inference.log_sigma.fc.bias"] = (ldc,)
    sh["disc_latent_inference.fc.weight"] = (K, cfeat)
    sh["disc_latent_inference.fc.bias"] = (K,)
    k0 = img // 32
    chans = [ldc + K] + DECODER_WIDTHS + [in_ch]
    for i, idx in enumerate(DECODER_CONVT):
        ks = k0 if i == 0 else 4
        sh["feature_reconstructor.decoder.%d.weight" % idx] = (chans[i], chans[i + 1], ks, ks)
        if i < 5:
            bn("feature_reconstructor.decoder.%d" % DECODER_BN[i], chans[i + 1])
    return sh


def is_param(key):
    return not (key.endswith("running_mean") or key.endswith("running_var")
                or key.endswith("num_batches_tracked"))


def default_init(name, in_ch=3, ldc=128, K=10, img=32, seed=1):
    """PyTorch-default init distribution (SURVEY §8b 'Init'): conv/convT/linear weights and
    biases U(+-1/sqrt(fan_in)), BN gamma=1 beta=0, running (0,1).  Same *distribution* as the
    reference, own RNG stream."""
    g = torch.Generator().manual_seed(seed)
    st = {}
    for k, shp in state_shapes(name, in_ch, ldc, K, img).items():
        if k.endswith("num_batches_tracked"):
            st[k] = torch.zeros((), dtype=torch.int64)
        elif k.endswith("running_mean"):
            st[k] = torch.zeros(shp)
        elif k.endswith("running_var"):
            st[k] = torch.ones(shp)
        elif len(shp) == 1 and ("norm" in k or re.search(r"decoder\.\d+\.(weight|bias)$", k)):
            st[k] = torch.ones(shp) if k.endswith("weight") else torch.zeros(shp)
        else:
            if len(shp) == 4:
                fan_in = shp[1] * shp[2] * shp[3]     # also what torch uses for ConvTranspose2d
            elif len(shp) == 2:
                fan_in = shp[1]
            else:                                     # bias: fan_in of its weight
                w = st[k[:-4] + "weight"]
                fan_in = w[0].numel()
            bound = 1.0 / math.sqrt(fan_in)
            st[k] = (torch.rand(shp, generator=g) * 2 - 1) * bound
    return st


# --------------------------------------------------------------------------- #
# model forward
# --------------------------------------------------------------------------- #
def _bn(st, prefix, x, training, update):
    """Train-mode BatchNorm2d: biased var to normalise, unbiased into running_var."""
    w, b = st[prefix + ".weight"], st[prefix + ".bias"]
    rm, rv = st[prefix + ".running_mean"], st[prefix + ".running_var"]
    if not training:
        return F.batch_norm(x, rm, rv, w, b, False, BN_MOMENTUM, BN_EPS)
    dims = (0, 2, 3)
    mean = x.mean(dims)
    var = x.var(dims, unbiased=False)
    if update:
        n = x.numel() // x.shape[1]
        with torch.no_grad():
            rm.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.detach())
            rv.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * var.detach() * n / max(n - 1, 1))
            st[prefix + ".num_batches_tracked"] += 1
    xh = (x - mean[None, :, None, None]) * torch.rsqrt(var + BN_EPS)[None, :, None, None]
    return xh * w[None, :, None, None] + b[None, :, None, None]


def encoder_forward(st, name, x, training=True, update=True):
    """wideresnet.py:27-49 (unit), :76-99 (net)."""
    units, _ = encoder_units(name)
    t = F.conv2d(x, st["feature_extractor.encoder.pre_process.conv0.weight"],
                 st["feature_extractor.encoder.pre_process.conv0.bias"], stride=1, padding=1)
    for p, ci, co, stride, sc in units:
        a = F.leaky_relu(_bn(st, p + ".f_block.norm1", t, training, update), LEAKY_SLOPE)
        c1 = F.conv2d(a, st[p + ".f_block.conv1.weight"], None, stride=stride, padding=1)
        a2 = F.leaky_relu(_bn(st, p + ".f_block.norm2", c1, training, update), LEAKY_SLOPE)
        c2 = F.conv2d(a2, st[p + ".f_block.conv2.weight"], None, stride=1, padding=1)
        if sc:
            ai = F.leaky_relu(_bn(st, p + ".i_block.norm", t, training, update), LEAKY_SLOPE)
            t = c2 + F.conv2d(ai, st[p + ".i_block.conv.weight"], None, stride=stride, padding=0)
        else:
            t = c2 + t
    return F.leaky_relu(_bn(st, "feature_extractor.encoder.transition.norm", t, training, update),
                        LEAKY_SLOPE)


def decoder_forward(st, latent, training=True, update=True):
    """decoder.py:12-62: ConvT(k0,s1,p0) BN ReLU, 4x[ConvT(4,2,1) BN ReLU], ConvT(4,2,1)."""
    h = latent
    for i, idx in enumerate(DECODER_CONVT):
        w = st["feature_reconstructor.decoder.%d.weight" % idx]
        if i == 0:
            h = F.conv_transpose2d(h, w, None, stride=1, padding=0)
        else:
            h = F.conv_transpose2d(h, w, None, stride=2, padding=1)
        if i < 5:
            h = F.relu(_bn(st, "feature_reconstructor.decoder.%d" % DECODER_BN[i], h, training, update))
    return h


def sample_latent(mu, log_sigma, log_alpha, eps, u=None, label=None, mixup=False,
                  label_mix=None, lam=None, temperature=0.67):
    """vae.py:23-86 with the noise made explicit (eps ~ N(0,1), u ~ U[0,1))."""
    z = mu + torch.exp(log_sigma) * eps
    K = log_alpha.shape[1]
    if label is not None:
        c = F.one_hot(label, K).to(mu.dtype)
        if mixup:
            c = lam * c + (1 - lam) * F.one_hot(label_mix, K).to(mu.dtype)
    else:
        gumbel = -torch.log(-torch.log(u + GUMBEL_EPS) + GUMBEL_EPS)
        c = torch.softmax((log_alpha + gumbel) / temperature, dim=1)
    return torch.cat([z, c], dim=1)[:, :, None, None]


def vae_forward(st, name, x, eps, u=None, label=None, mixup=False, label_mix=None, lam=None,
                temperature=0.67, training=True, update=True):
    """vae.py:140-151 -> (reconstruction logits, mean, log_sigma, log_alpha)."""
    feat = encoder_forward(st, name, x, training, update).mean(dim=(2, 3))
    mu = F.linear(feat, st["continuous_inference.mean.fc.weight"], st["continuous_inference.mean.fc.bias"])
    ls = F.linear(feat, st["continuous_inference.log_sigma.fc.weight"],
                  st["continuous_inference.log_sigma.fc.bias"])
    la = F.log_softmax(F.linear(feat, st["disc_latent_inference.fc.weight"],
                                st["disc_latent_inference.fc.bias"]), dim=1)
    latent = sample_latent(mu, ls, la, eps, u, label, mixup, label_mix, lam, temperature)
    return decoder_forward(st, latent, training, update), mu, ls, la


# --------------------------------------------------------------------------- #
# criteria, mixup
# --------------------------------------------------------------------------- #
def vae_criterion(x, x_rec, mu, ls, la, x_sigma=1.0, bce=True):
    """criterion.py:32-57 -> (recon, KL_c, KL_d), each a 0-dim tensor."""
    B = x.shape[0]
    K = la.shape[1]
    if bce:
        recon = (torch.clamp(x_rec, min=0) - x_rec * x + torch.log1p(torch.exp(-x_rec.abs()))).sum() / B
    else:
        recon = ((torch.sigmoid(x_rec) - x) ** 2).sum() / (2 * B * x_sigma ** 2)
    kl_c = 0.5 * (mu * mu + torch.exp(2 * ls) - 2 * ls - 1).sum() / B
    # the reference builds the prior as float32(log(float32(1/K)))
    log_prior = float(torch.log(torch.tensor(1.0 / K, dtype=torch.float32)))
    kl_d = (torch.exp(la) * (la - log_prior)).sum() / B
    return recon, kl_c, kl_d


def cls_criterion(predict, label, batch_weight=None):
    """criterion.py:97-108; label may be a soft distribution."""
    s = (predict * label).sum(dim=1)
    if batch_weight is not None:
        s = s * batch_weight
    return -s.mean()


def pairwise_gaussian_kl(mu, ls):
    """KL(N_i || N_j) for all pairs; closed form of mixup.py:93-99 (vectorised)."""
    var = torch.exp(2 * ls)
    inv = 1.0 / var
    d = mu.shape[1]
    t1 = ls.sum(1)[None, :] - ls.sum(1)[:, None]
    t2 = 0.5 * var @ inv.t()
    t3 = 0.5 * ((mu * mu) @ inv.t() - 2 * mu @ (mu * inv).t() + (mu * mu * inv).sum(1)[None, :])
    return t1 + t2 + t3 - 0.5 * d


def optimal_match_index(mu, ls):
    """mixup.py:9-18: second-smallest entry per row of the pairwise-KL matrix."""
    kl = pairwise_gaussian_kl(mu, ls)
    return torch.topk(kl, 2, largest=False)[1][:, 1]


def mix_with_index(image, mu, ls, la, lam, index):
    """mixup.py:22-25 / :36-39: sigma and alpha are mixed in linear space."""
    return (lam * image + (1 - lam) * image[index],
            lam * mu + (1 - lam) * mu[index],
            lam * torch.exp(ls) + (1 - lam) * torch.exp(ls[index]),
            lam * torch.exp(la) + (1 - lam) * torch.exp(la[index]))


def alpha_schedule(epoch, max_epoch, alpha_max):
    """main_shot_vae.py:518-520."""
    return alpha_max * math.exp(-5 * (1 - min(1, epoch / max_epoch)) ** 2)


def schedule(epoch, epochs=600, cmi=0.0, dmi=2.3, kbmc=1e-3, kbmd=1e-3, akb=200, ewm=1e-3,
             aew=400, pwm=1.0, apw=200, wrd=1.0, wmf=0.4):
    """Per-epoch scalars of main_shot_vae.py:270-279 (Cifar10 defaults, dmi=2.3 from :139)."""
    return dict(cmi=alpha_schedule(epoch, akb, cmi), dmi=alpha_schedule(epoch, akb, dmi),
                ew=alpha_schedule(epoch, aew, ewm), kl_beta_c=alpha_schedule(epoch, akb, kbmc),
                kl_beta_d=alpha_schedule(epoch, akb, kbmd), pwm=alpha_schedule(epoch, apw, pwm),
                ucw=alpha_schedule(epoch, round(wmf * epochs), wrd))


# --------------------------------------------------------------------------- #
# the training step (main_shot_vae.py:280-366)
# --------------------------------------------------------------------------- #
def train_step(st, name, image_l, label_l, image_u, noise, sch, bce=True, x_sigma=1.0,
               temperature=0.67, optimal_match=False, backward=True):
    """One SHOT-VAE step up to (not including) optimizer.step().

    ``st`` tensors that are parameters must have requires_grad=True; their .grad accumulates
    over both backward passes exactly as in the reference.  ``noise`` holds the host-RNG draws in
    reference order: eps1, lam_l, perm_l, eps2, eps3, u3, lam_u, perm_u, eps4, u4.
    Returns a dict of the scalars and tensors the parity tests compare."""
    out = {}
    K = st["disc_latent_inference.fc.bias"].shape[0]
    Bl, Bu = image_l.shape[0], image_u.shape[0]
    onehot_l = F.one_hot(label_l, K).float()
    # (1) labelled forward, c = one-hot                                   :288-295
    rec1, mu1, ls1, la1 = vae_forward(st, name, image_l, noise["eps1"], label=label_l,
                                      temperature=temperature)
    recon_l, klc_l, kld_l = vae_criterion(image_l, rec1, mu1, ls1, la1, x_sigma, bce)
    prior_l = sch["kl_beta_c"] * torch.abs(klc_l - sch["cmi"]) + sch["kl_beta_d"] * torch.abs(kld_l - sch["dmi"])
    elbo_l = recon_l + prior_l
    # label smoothing (no grad)                                           :297-310
    with torch.no_grad():
        lam_l, perm_l = noise["lam_l"], noise["perm_l"]
        sm_img, sm_mu, sm_sigma, sm_alpha = mix_with_index(image_l, mu1, ls1, la1, lam_l, perm_l)
        sm_label = label_l[perm_l]
        sm_onehot = F.one_hot(sm_label, K).float()
    # (2) mixed labelled forward                                          :311-324
    rec2, mu2, ls2, la2 = vae_forward(st, name, sm_img, noise["eps2"], label=label_l, mixup=True,
                                      label_mix=sm_label, lam=lam_l, temperature=temperature)
    disc_post_l = lam_l * cls_criterion(la2, onehot_l) + (1 - lam_l) * cls_criterion(la2, sm_onehot)
    cont_post_l = (((mu2 - sm_mu) ** 2).sum() + ((torch.exp(ls2) - sm_sigma) ** 2).sum()) / Bl
    elbo_l = elbo_l + sch["kl_beta_c"] * sch["pwm"] * cont_post_l
    loss_sup = sch["ew"] * elbo_l + disc_post_l
    if backward:
        loss_sup.backward()
    # (3) unlabelled forward, c = gumbel-softmax                          :327-346
    rec3, mu3, ls3, la3 = vae_forward(st, name, image_u, noise["eps3"], u=noise["u3"],
                                      temperature=temperature)
    recon_u, klc_u, kld_u = vae_criterion(image_u, rec3, mu3, ls3, la3, x_sigma, bce)
    prior_u = sch["kl_beta_c"] * torch.abs(klc_u - sch["cmi"]) + sch["kl_beta_d"] * torch.abs(kld_u - sch["dmi"])
    elbo_u = recon_u + prior_u
    # mixup (no grad)                                                     :348-355
    with torch.no_grad():
        lam_u = noise["lam_u"]
        perm_u = optimal_match_index(mu3, ls3) if optimal_match else noise["perm_u"]
        mx_img, mx_mu, mx_sigma, mx_alpha = mix_with_index(image_u, mu3, ls3, la3, lam_u, perm_u)
    # (4) mixed unlabelled forward                                        :356-364
    rec4, mu4, ls4, la4 = vae_forward(st, name, mx_img, noise["eps4"], u=noise["u4"],
                                      temperature=temperature)
    disc_post_u = cls_criterion(la4, mx_alpha)
    cont_post_u = (((mu4 - mx_mu) ** 2).sum() + ((torch.exp(ls4) - mx_sigma) ** 2).sum()) / Bu
    elbo_u = elbo_u + sch["kl_beta_c"] * sch["pwm"] * cont_post_u
    loss_unsup = sch["ew"] * elbo_u + sch["ucw"] * disc_post_u
    if backward:
        loss_unsup.backward()
    out.update(recon_l=recon_l, klc_l=klc_l, kld_l=kld_l, recon_u=recon_u, klc_u=klc_u, kld_u=kld_u,
               disc_post_l=disc_post_l, cont_post_l=cont_post_l, disc_post_u=disc_post_u,
               cont_post_u=cont_post_u, loss_sup=loss_sup, loss_unsup=loss_unsup,
               rec1=rec1, mu1=mu1, ls1=ls1, la1=la1, rec2=rec2, mu2=mu2, ls2=ls2, la2=la2,
               rec3=rec3, mu3=mu3, ls3=ls3, la3=la3, rec4=rec4, mu4=mu4, ls4=ls4, la4=la4,
               sm_img=sm_img, mx_img=mx_img, perm_u=perm_u)
    return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}


def m2_step(st, name, image_l, label_l, image_u, label_u, noise, sch, bce=True, x_sigma=1.0, temperature=0.67,
            backward=True):
    """One step of the M2 baseline loop (main_M2_vae.py:259-305): labelled forward with the one-hot label +
    cross-entropy, unlabelled forward with the Gumbel-softmax sample, no mixup; gradients of both backward passes
    accumulate.  ``noise``: eps1 (labelled), eps3 / u3 (unlabelled).  Also the monitored KL(q(y|x) || smoothed label)
    of :285-291."""
    K = st["disc_latent_inference.fc.bias"].shape[0]
    B = image_l.shape[0]
    onehot_l = F.one_hot(label_l, K).float()
    rec1, mu1, ls1, la1 = vae_forward(st, name, image_l, noise["eps1"], label=label_l, temperature=temperature)
    recon_l, klc_l, kld_l = vae_criterion(image_l, rec1, mu1, ls1, la1, x_sigma, bce)
    elbo_l = recon_l + sch["kl_beta_c"] * torch.abs(klc_l - sch["cmi"]) + sch["kl_beta_d"] * torch.abs(kld_l - sch["dmi"])
    disc_post_l = cls_criterion(la1, onehot_l)
    loss_sup = sch["ew"] * elbo_l + disc_post_l
    if backward:
        loss_sup.backward()
    rec3, mu3, ls3, la3 = vae_forward(st, name, image_u, noise["eps3"], u=noise["u3"], temperature=temperature)
    with torch.no_grad():
        smooth = torch.zeros(B, K).scatter_(1, label_u.view(-1, 1), 1 - 0.001 - 0.001 / (K - 1)) + 0.001 / (K - 1)
        alpha = torch.exp(la3)
        kl_inference = (alpha * la3 - alpha * torch.log(smooth)).sum() / B
    recon_u, klc_u, kld_u = vae_criterion(image_u, rec3, mu3, ls3, la3, x_sigma, bce)
    elbo_u = recon_u + sch["kl_beta_c"] * torch.abs(klc_u - sch["cmi"]) + sch["kl_beta_d"] * torch.abs(kld_u - sch["dmi"])
    loss_unsup = sch["ew"] * elbo_u
    if backward:
        loss_unsup.backward()
    out = dict(recon_l=recon_l, klc_l=klc_l, kld_l=kld_l, recon_u=recon_u, klc_u=klc_u, kld_u=kld_u,
               disc_post_l=disc_post_l, kl_inference=kl_inference, loss_sup=loss_sup, loss_unsup=loss_unsup,
               rec1=rec1, mu1=mu1, ls1=ls1, la1=la1, rec3=rec3, mu3=mu3, ls3=ls3, la3=la3)
    return {k: v.detach() for k, v in out.items()}


def sgd_step(st, momentum_buf, lr=0.1, momentum=0.9, weight_decay=5e-4):
    """torch.optim.SGD semantics (main_shot_vae.py:198,365): g += wd*p; v = mom*v + g (v = g on
    the first step); p -= lr*v.  Clears .grad afterwards (:366)."""
    with torch.no_grad():
        for k, p in st.items():
            if not is_param(k) or p.grad is None:
                continue
            g = p.grad + weight_decay * p
            if k not in momentum_buf:
                momentum_buf[k] = g.clone()
            else:
                momentum_buf[k].mul_(momentum).add_(g)
            p.sub_(lr * momentum_buf[k])
            p.grad = None


def make_noise(Bl, Bu, K, ldc=128, seed=0, epsilon=0.1):
    """Host-RNG draws for one step in the reference's consumption order."""
    import numpy as np
    g = torch.Generator().manual_seed(seed)
    rs = np.random.RandomState(seed)
    n = {}
    n["eps1"] = torch.randn(Bl, ldc, generator=g)
    n["lam_l"] = float(rs.beta(epsilon, epsilon)) if epsilon > 0 else 1.0
    n["perm_l"] = torch.randperm(Bl, generator=g)
    n["eps2"] = torch.randn(Bl, ldc, generator=g)
    n["eps3"] = torch.randn(Bu, ldc, generator=g)
    n["u3"] = torch.rand(Bu, K, generator=g)
    n["lam_u"] = float(rs.beta(2.0, 2.0))
    n["perm_u"] = torch.randperm(Bu, generator=g)
    n["eps4"] = torch.randn(Bu, ldc, generator=g)
    n["u4"] = torch.rand(Bu, K, generator=g)
    return n
