"""Drop-ins for lib/criterion.py: VAECriterion (:8-57) and ClsCriterion (:93-108), each one fused HIP
reduction kernel forward and one backward."""
import ctypes as C

import torch
from torch import nn

from . import _lib as L


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.ShotVaeHipError("shot_vae_amd criteria run on an MI355X only (no CPU fallback)")


class _ElboFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, x_rec, mu, ls, la, bce, x_sigma):
        _need_gpu(x, x_rec, mu, ls, la)
        x, x_rec = x.contiguous().float(), x_rec.contiguous().float()
        mu, ls, la = mu.contiguous().float(), ls.contiguous().float(), la.contiguous().float()
        B = x.shape[0]
        out = torch.zeros(3, dtype=torch.float32, device=x.device)
        L.call("sv_elbo_fwd", _p(x), _p(x_rec), x[0].numel(), _p(mu), _p(ls), _p(la), B, mu.shape[1], la.shape[1],
               int(bce), float(x_sigma), _p(out), _st())
        ctx.save_for_backward(x, x_rec, mu, ls, la)
        ctx.cfg = (bce, x_sigma)
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, g0, g1, g2):
        x, x_rec, mu, ls, la = ctx.saved_tensors
        bce, x_sigma = ctx.cfg
        z = torch.zeros((), device=x.device)
        g = torch.stack([g0 if g0 is not None else z, g1 if g1 is not None else z, g2 if g2 is not None else z]).float()
        dxr, dmu, dls, dla = torch.empty_like(x_rec), torch.empty_like(mu), torch.empty_like(ls), torch.empty_like(la)
        L.call("sv_elbo_bwd", _p(x), _p(x_rec), x[0].numel(), _p(mu), _p(ls), _p(la), x.shape[0], mu.shape[1],
               la.shape[1], int(bce), float(x_sigma), _p(g), _p(dxr), _p(dmu), _p(dls), _p(dla), _st())
        return None, dxr, dmu, dls, dla, None, None


class VAECriterion(nn.Module):
    """(x, x_reconstructed, z_mean, z_log_sigma, disc_log_alpha) -> (reconstruct_loss, continuous_kl_loss,
    disc_kl_loss); BCE-with-logits sum/B or MSE(sigmoid)/(2 B sigma^2), KL to N(0,I), KL to the uniform prior."""

    def __init__(self, discrete_dim=10, x_sigma=1, bce_reconstruction=True):
        super(VAECriterion, self).__init__()
        self.x_sigma = x_sigma
        self.bce_reconstruction = bce_reconstruction
        self.discrete_dim = discrete_dim
        # the reference's attribute (lib/criterion.py:29-30: the log of the uniform prior, [1, K] on the GPU); the kernel has the
        # prior as the constant -log K, the tensor exists for callers that read it.  Like the reference's, it is a plain attribute
        # (not a buffer: .cuda() / .to() do not move it, state_dict() does not hold it) -- created on the GPU when there is one.
        prior = torch.full((1, discrete_dim), 1.0 / discrete_dim).log()
        self.disc_log_prior_param = prior.cuda() if torch.cuda.is_available() else prior

    def forward(self, x, x_reconstructed, z_mean, z_log_sigma, disc_log_alpha):
        assert disc_log_alpha.shape[1] == self.discrete_dim
        return _ElboFn.apply(x, x_reconstructed, z_mean, z_log_sigma, disc_log_alpha, self.bce_reconstruction,
                             self.x_sigma)


class _ClsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, predict, label, weight):
        _need_gpu(predict, label, weight)
        predict, label = predict.contiguous().float(), label.contiguous().float()
        weight = weight.contiguous().float().view(-1) if weight is not None else None
        B, K = predict.shape
        out = torch.zeros(1, dtype=torch.float32, device=predict.device)
        L.call("sv_cls_fwd", _p(predict), _p(label), _p(weight), B, K, _p(out), _st())
        ctx.save_for_backward(label, weight)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        label, weight = ctx.saved_tensors
        B, K = label.shape
        dp = torch.empty_like(label)
        g = g.contiguous().float().view(1)
        L.call("sv_cls_bwd", _p(label), _p(weight), B, K, _p(g), _p(dp), _st())
        return dp, None, None


class ClsCriterion(nn.Module):
    """-mean_b sum_c predict*label (label may be soft), optional per-sample weight."""

    def __init__(self):
        super(ClsCriterion, self).__init__()

    def forward(self, predict, label, batch_weight=None):
        return _ClsFn.apply(predict, label, batch_weight)


class _PostFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, ls, mu_t, sigma_t):
        _need_gpu(mu, ls, mu_t, sigma_t)
        mu, ls = mu.contiguous().float(), ls.contiguous().float()
        mu_t, sigma_t = mu_t.contiguous().float(), sigma_t.contiguous().float()
        B, D = mu.shape
        out = torch.zeros(1, dtype=torch.float32, device=mu.device)
        L.call("sv_post_fwd", _p(mu), _p(ls), _p(mu_t), _p(sigma_t), B, D, _p(out), _st())
        ctx.save_for_backward(mu, ls, mu_t, sigma_t)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        mu, ls, mu_t, sigma_t = ctx.saved_tensors
        dmu, dls = torch.empty_like(mu), torch.empty_like(ls)
        g = g.contiguous().float().view(1)
        L.call("sv_post_bwd", _p(mu), _p(ls), _p(mu_t), _p(sigma_t), mu.shape[0], mu.shape[1], _p(g), _p(dmu),
               _p(dls), _st())
        return dmu, dls, None, None


def continuous_posterior_loss(norm_mean, norm_log_sigma, target_mean, target_sigma):
    """(mse_sum(mean, target_mean) + mse_sum(exp(log_sigma), target_sigma)) / B  --  the expression at
    main_shot_vae.py:319-321 and :359-361 as one fused kernel."""
    return _PostFn.apply(norm_mean, norm_log_sigma, target_mean, target_sigma)
