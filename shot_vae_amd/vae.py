"""Drop-in for the reference's ``shot_vae_model.vae.VariationalAutoEncoder`` (vae.py:89-151) whose
forward/backward run as hand-written HIP kernels on MI355X.

Same constructor, same positional ``forward`` signature, same 4-tuple result, same ``state_dict``
keys (with or without the ``.module`` segments ``data_parallel=True`` produces in the reference), so
the loop of main_shot_vae.py:261-383 runs unchanged.  Differences (all documented in DESIGN.md):
  * ``data_parallel`` only selects the key naming; multi-GPU is one process per GPU + one RCCL
    all-reduce of the flat gradient buffer (see dp.py), not nn.DataParallel;
  * gradients of the parameters are accumulated by the kernels straight into one flat fp32 buffer
    (``p.grad`` are views of it);
  * only the wideresnet encoders on 32x32 inputs are implemented (the BASELINE.json configs).
"""
import torch
from torch import nn

from . import _lib as L
from .engine import Engine, Plan

# modules that the reference wraps in nn.DataParallel (wideresnet.py:78-93, vae.py:108-132, decoder.py:63-64)
_DP_WRAPPED = ("feature_extractor.encoder.pre_process", "feature_extractor.encoder.wideblock1",
               "feature_extractor.encoder.wideblock2", "feature_extractor.encoder.wideblock3",
               "feature_extractor.encoder.transition", "continuous_inference.mean",
               "continuous_inference.log_sigma", "disc_latent_inference", "feature_reconstructor.decoder")


class _Node(nn.Module):
    """Name-space module: only carries parameters / buffers so state_dict keys match the reference."""

    def forward(self, *a, **k):
        raise RuntimeError("this sub-module is a parameter container; call the VariationalAutoEncoder")


def _dp_key(key):
    for w in _DP_WRAPPED:
        if key.startswith(w + "."):
            return w + ".module." + key[len(w) + 1:]
    return key


class _VAEFunction(torch.autograd.Function):
    """One autograd node for the whole network (one or several batched instances of it): fused HIP forward,
    hand-written HIP backward."""

    @staticmethod
    def forward(ctx, anchor, model, image, groups, eps, u, rec_groups=None, update_order=None):
        eng = model._engine
        rec, mu, ls, la, f = eng.forward(image, groups, eps, u, model._temperature, model.training, keep=True,
                                         rec_groups=rec_groups, update_order=update_order)
        ctx.model, ctx.f = model, f
        # an output that enters no loss term arrives as None in backward (not as a zero tensor): the reconstruction of
        # the mixed forwards (main_shot_vae.py:311,356) -- their decoder backward is then skipped, as autograd does in the
        # reference
        ctx.set_materialize_grads(False)
        return rec, mu, ls, la

    @staticmethod
    def backward(ctx, d_rec, d_mu, d_ls, d_la):
        model, f = ctx.model, ctx.f
        if f is None:
            raise RuntimeError("backward through the same SHOT-VAE forward twice is not supported")
        if not f.training:
            # an eval-mode forward normalises with the running statistics and saves no batch statistics: the
            # batch-statistics BatchNorm backward below would read uninitialised mean / rstd
            raise NotImplementedError("backward through an eval-mode forward (BatchNorm with running statistics) is not "
                                      "implemented; the reference never does it (main_shot_vae.py:423-424: no_grad)")
        ctx.f = None
        model._attach_grads()
        eng = model._engine
        eng.backward(f, d_rec.contiguous().float() if d_rec is not None else None, d_mu, d_ls, d_la)
        return (None,) * 8


class VariationalAutoEncoder(nn.Module):
    def __init__(self, encoder_name, num_input_channels=1, drop_rate=0, img_size=(160, 160), data_parallel=True,
                 continuous_latent_dim=100, disc_latent_dim=10, sample_temperature=0.67, small_input=False,
                 compute_dtype="bf16", rng="host"):
        super(VariationalAutoEncoder, self).__init__()
        if "wideresnet" not in encoder_name:
            # densenet / preactresnet encoders exist in the reference (vae.py:93-104) but are outside
            # every BASELINE.json config; same error type as the reference's fall-through (vae.py:106)
            raise NotImplementedError("{} not implemented".format(encoder_name))
        if drop_rate != 0:
            raise NotImplementedError("drop_rate != 0 is not implemented (main_shot_vae.py:60 default is 0)")
        if not small_input:
            raise NotImplementedError("small_input=False (7x7 stem + max-pool) is not implemented; the "
                                      "CIFAR/SVHN configs of main_shot_vae.py use small_input=True")
        if tuple(img_size) != (32, 32):
            raise NotImplementedError("only 32x32 inputs are implemented")
        plan = Plan(encoder_name, in_ch=num_input_channels, img=img_size[0], ldc=continuous_latent_dim,
                    K=int(disc_latent_dim))
        self._plan = plan
        self._engine = Engine(plan, compute_dtype)
        self._temperature = sample_temperature
        self._data_parallel = data_parallel
        self._disc_latent_dim = disc_latent_dim
        self.rng = rng
        self._views = []          # (parameter, flat offset/spec) for re-pointing after device moves
        self._engine.init_default()
        self._build_tree()
        self.feature_extractor.num_feature_channel = plan.cfeat
        self._anchor = None
        # in-place updates by a torch optimizer bump the parameters' version counters: that is how the
        # engine learns that its packed weight shadows are stale
        self._engine.version_probe = lambda: sum(v[0]._version for v in self._views)

    # ------------------------------------------------------------------ module tree / state_dict
    def _node(self, path):
        m = self
        for part in path:
            if not hasattr(m, part):
                m.add_module(part, _Node())
            m = getattr(m, part)
        return m

    def _flat_view(self, flat, kind, payload):
        if kind == "conv":
            return payload.torch_view(flat)
        if kind == "mat":
            off, (r, c) = payload
            return flat[off: off + r * c].view(r, c)
        off, n = payload
        return flat[off: off + n]

    def _build_tree(self):
        eng, plan = self._engine, self._plan
        for name in ("feature_extractor", "global_avg", "continuous_inference", "disc_latent_inference", "sample",
                     "feature_reconstructor"):
            self.add_module(name, _Node())
        for key, kind, payload in plan.state_items():
            k = _dp_key(key) if self._data_parallel else key
            parts = k.split(".")
            node = self._node(parts[:-1])
            if kind in ("conv", "mat", "vec"):
                prm = nn.Parameter(self._flat_view(eng.param, kind, payload))
                node.register_parameter(parts[-1], prm)
                self._views.append((prm, kind, payload))
            elif kind == "rm":
                node.register_buffer(parts[-1], eng.bufs[payload.rm_off: payload.rm_off + payload.C])
            elif kind == "rv":
                node.register_buffer(parts[-1], eng.bufs[payload.rv_off: payload.rv_off + payload.C])
            else:
                node.register_buffer(parts[-1], eng.nbt[payload.index])

    def _repoint(self):
        """Make every Parameter / buffer a view of the (possibly moved) flat storage again."""
        eng, plan = self._engine, self._plan
        for prm, kind, payload in self._views:
            prm.data = self._flat_view(eng.param, kind, payload)
            prm.grad = None
        for key, kind, payload in plan.state_items():
            if kind in ("rm", "rv", "nbt"):
                k = _dp_key(key) if self._data_parallel else key
                parts = k.split(".")
                node = self._node(parts[:-1])
                if kind == "rm":
                    t = eng.bufs[payload.rm_off: payload.rm_off + payload.C]
                elif kind == "rv":
                    t = eng.bufs[payload.rv_off: payload.rv_off + payload.C]
                else:
                    t = eng.nbt[payload.index]
                node._buffers[parts[-1]] = t

    def _apply(self, fn, recurse=True):
        # keep ONE flat storage: move the flat buffers, then re-point the views (nn.Module._apply would
        # give every parameter its own allocation)
        self._engine.to(fn)
        self._repoint()
        self._anchor = None
        return self

    def _attach_grads(self):
        """p.grad = view of the flat gradient buffer.  If the optimizer dropped them
        (zero_grad(set_to_none=True)), the flat buffer is zeroed first."""
        eng = self._engine
        first = self._views[0][0]
        if first.grad is not None and first.grad.data_ptr() == eng.grad.data_ptr() + 4 * self._views[0][2].master_off:
            return
        eng.grad.zero_()
        for prm, kind, payload in self._views:
            prm.grad = self._flat_view(eng.grad, kind, payload)

    def load_state_dict(self, state_dict, strict=True):
        """Accepts both the data_parallel=True ('.module.') and =False key layouts of the reference."""
        own = set(self.state_dict().keys())
        fixed = {}
        for k, v in state_dict.items():
            if k in own:
                fixed[k] = v
                continue
            plain = k.replace(".module.", ".")
            alt = _dp_key(plain) if self._data_parallel else plain
            fixed[alt if alt in own else k] = v
        out = super(VariationalAutoEncoder, self).load_state_dict(fixed, strict)
        self._engine.mark_dirty()
        return out

    # ------------------------------------------------------------------ reference API
    def flat_parameters(self):
        """(param, grad) flat fp32 buffers: what FlatSGD updates and dp.all_reduce reduces."""
        return self._engine.param, self._engine.grad

    def _draw_noise(self, B, dev, gumbel):
        """noise in the reference's order: randn for z (vae.py:37,82), then rand for gumbel (vae.py:52,69)"""
        plan = self._plan
        if self.rng == "host":
            eps = torch.randn(B, plan.ldc).to(dev)
            u = torch.rand(B, plan.K).to(dev) if gumbel else None
        else:
            eps = torch.randn(B, plan.ldc, device=dev)
            u = torch.rand(B, plan.K, device=dev) if gumbel else None
        return eps, u

    @staticmethod
    def _group_spec(mixup, disc_label, disc_pseudo_label, mixup_lam):
        """(mode, label, label_mix, lam) of one forward call's arguments (vae.py:38-52)"""
        if disc_label is None:
            return (0, None, None, 0.0)
        label = disc_label.view(-1).long().contiguous()
        if mixup:
            lam = mixup_lam.reshape(1).float() if torch.is_tensor(mixup_lam) else float(mixup_lam)
            return (2, label, disc_pseudo_label.view(-1).long().contiguous(), lam)
        return (1, label, None, 0.0)

    def forward(self, input_img, mixup=False, disc_label=None, disc_pseudo_label=None, mixup_lam=None):
        if not input_img.is_cuda:
            raise L.ShotVaeHipError("VariationalAutoEncoder: input is not on an MI355X (no CPU fallback)")
        spec = self._group_spec(mixup, disc_label, disc_pseudo_label, mixup_lam)
        eps, u = self._draw_noise(input_img.size(0), input_img.device, spec[0] == 0)
        return self._run(input_img, [spec], eps, u)

    def forward_groups(self, images, specs, eps=None, u=None, rec_groups=None, update_order=None):
        """Several forward calls as ONE batched launch sequence (extension; see Engine.forward): images = list of equally
        sized batches, specs = list of dicts with the keyword arguments of forward() (mixup, disc_label,
        disc_pseudo_label, mixup_lam).  Equivalent to calling forward() on each batch -- every group keeps its own
        BatchNorm batch statistics and the running statistics receive the groups' momentum updates in list order -- at a
        fraction of the launches.  eps / u: the noise to use ([G * B, ldc] / [G * B, K]); drawn here if None, group by
        group in the reference's order.  rec_groups = Gd: only the first Gd batches' reconstructions are produced and
        differentiated (the others' last ConvTranspose and decoder backward are skipped; reconstruction is [Gd * B, ...]);
        update_order[k] = the list position of the reference's k-th forward (order of the BatchNorm running-statistic
        updates; default list order).  Returns the 4-tuple of forward() with the groups concatenated along dim 0."""
        G = len(images)
        B = images[0].size(0)
        if any(im.size(0) != B for im in images):
            raise ValueError("forward_groups: the batches must have equal sizes")
        dev = images[0].device
        gs = [self._group_spec(sp.get("mixup", False), sp.get("disc_label"), sp.get("disc_pseudo_label"),
                               sp.get("mixup_lam")) for sp in specs]
        if eps is None:
            pairs = [self._draw_noise(B, dev, g[0] == 0) for g in gs]
            eps = torch.cat([e for e, _ in pairs])
            if any(uu is not None for _, uu in pairs):
                z = torch.zeros(B, self._plan.K, device=dev)
                u = torch.cat([uu if uu is not None else z for _, uu in pairs])
        return self._run(torch.cat([im.float() for im in images]), gs, eps, u, rec_groups, update_order)

    def forward_groups_direct(self, images, specs, eps, u, rec_groups=None, update_order=None, image_cat=None, x16=None):
        """forward_groups for a caller that runs the backward itself (no autograd node): returns (rec, mu, ls, la, ctx); pass
        ctx and the gradients w.r.t. the four outputs to backward_direct().  image_cat / x16: the concatenated images and
        their NHWC16 form when the caller has already made them (train_step_grouped's input-side stream)."""
        gs = [self._group_spec(sp.get("mixup", False), sp.get("disc_label"), sp.get("disc_pseudo_label"),
                               sp.get("mixup_lam")) for sp in specs]
        with torch.no_grad():
            if image_cat is None:
                image_cat = torch.cat([im.float() for im in images])
            return self._engine.forward(image_cat, gs, eps, u, self._temperature, self.training, keep=True,
                                        rec_groups=rec_groups, update_order=update_order, x16=x16)

    def backward_direct(self, ctx, d_rec, d_mu, d_ls, d_la, own_grads=False):
        """accumulates the parameter gradients of a forward_groups_direct() call into the flat gradient buffer (p.grad).
        own_grads: d_mu / d_ls / d_la are the caller's own, freshly written tensors and may be modified in place"""
        if not ctx.training:
            raise NotImplementedError("backward through an eval-mode forward (BatchNorm with running statistics) is not implemented")
        self._attach_grads()
        with torch.no_grad():
            self._engine.backward(ctx, d_rec, d_mu, d_ls, d_la, own_grads)

    def _run(self, image, groups, eps, u, rec_groups=None, update_order=None):
        eng = self._engine
        dev = image.device
        if torch.is_grad_enabled():
            if self._anchor is None or self._anchor.device != dev:
                self._anchor = torch.zeros(1, device=dev, requires_grad=True)
            return _VAEFunction.apply(self._anchor, self, image, groups, eps, u, rec_groups, update_order)
        rec, mu, ls, la, _ = eng.forward(image, groups, eps, u, self._temperature, self.training, keep=False,
                                         rec_groups=rec_groups, update_order=update_order)
        return rec, mu, ls, la
