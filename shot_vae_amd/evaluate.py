"""The evaluation pass of the reference loop -- valid() / test() of main_shot_vae.py:409-458 / :461-510 (the two bodies
are identical up to the TensorBoard tag) -- on the HIP path: eval-mode forward (BatchNorm with running statistics, the
sampler still stochastic: vae.py:37), the ELBO terms, the reported reconstruction metric MSE(sigmoid(rec), x) / (2 B
sigma^2) (always MSE, whatever --br says: :427-429), `ELBO` = mse + 0.01 (KL_c + KL_d) (:435), top-1 / top-5 from
disc_log_alpha (:441-447).

Every per-batch quantity stays on the device (the reference calls float() five times per batch = five host syncs);
`Evaluator.result()` does the one device-to-host copy of an epoch."""
import ctypes as C

import torch

from . import _lib as L


def _p(t):
    return C.c_void_p(t.data_ptr())


class Evaluator:
    """Running means of valid() / test() (lib/utils/avgmeter.py semantics: batch means weighted by batch size).

        ev = Evaluator(model, elbo_criterion)
        for image, label in loader: ev.update(image, label)
        top1, top5 = ev.result()["top1"], ev.result()["top5"]"""

    KEYS = ("klc", "kld", "mse", "elbo")

    def __init__(self, model, elbo_criterion, topk=5):
        self.model, self.crit, self.topk = model, elbo_criterion, topk
        self.reset()

    def reset(self):
        self.acc = None          # device: [sum klc*B, sum kld*B, sum mse*B, sum elbo*B, top1 hits, topk hits]
        self.count = 0

    def update(self, image, label):
        model, crit = self.model, self.crit
        if not image.is_cuda:
            raise L.ShotVaeHipError("Evaluator: inputs must be on an MI355X (no CPU fallback)")
        image = image.float().contiguous()
        label = label.long().contiguous()
        B = image.size(0)
        was_training = model.training
        model.eval()                                                     # :414
        try:
            with torch.no_grad():
                rec, mu, ls, la, *_ = model(image)                       # :423-424
                mu, ls, la = mu.contiguous(), ls.contiguous(), la.contiguous()
                out = torch.zeros(3, dtype=torch.float32, device=image.device)
                # MSE(sigmoid(rec), x) / (2 B sigma^2) + the two KL terms in one fused reduction (bce = 0): :425-429
                L.call("sv_elbo_fwd", _p(image), _p(rec.contiguous()), image[0].numel(), _p(mu), _p(ls), _p(la), B,
                       mu.shape[1], la.shape[1], 0, float(crit.x_sigma), _p(out),
                       C.c_void_p(torch.cuda.current_stream().cuda_stream))
                if self.acc is None:
                    self.acc = torch.zeros(6, dtype=torch.float32, device=image.device)
                mse, klc, kld = out[0], out[1], out[2]
                self.acc[:4] += B * torch.stack([klc, kld, mse, mse + 0.01 * (klc + kld)])        # :430-435
                L.call("sv_topk_hits", _p(la), _p(label), B, la.shape[1], self.topk, _p(self.acc[4:]),
                       C.c_void_p(torch.cuda.current_stream().cuda_stream))                       # :441-447
        finally:
            model.train(was_training)
        self.count += B
        return out

    def result(self):
        """dict(klc, kld, mse, elbo, top1, top5) as Python floats (one host sync)."""
        if self.acc is None or self.count == 0:
            return dict(klc=0.0, kld=0.0, mse=0.0, elbo=0.0, top1=0.0, top5=0.0)
        a = (self.acc / self.count).tolist()
        return dict(klc=a[0], kld=a[1], mse=a[2], elbo=a[3], top1=a[4], top5=a[5])


def evaluate(model, elbo_criterion, batches, topk=5):
    """valid() / test(): `batches` yields (image, label) device tensors; returns the result dict.  The reference returns
    (top1, top5) (:458)."""
    ev = Evaluator(model, elbo_criterion, topk)
    for image, label in batches:
        ev.update(image, label)
    return ev.result()
