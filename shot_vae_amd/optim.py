"""SGD(momentum, weight decay) on the flat parameter buffer: one HIP kernel per step instead of 107
per-tensor updates.  Semantics of torch.optim.SGD as used at main_shot_vae.py:198,365-366
(no Nesterov, dampening 0, first step v = g)."""
import ctypes as C

import torch

from . import _lib as L


class FlatSGD(torch.optim.Optimizer):
    """A torch.optim.Optimizer (one param group over model.parameters()): torch.optim.lr_scheduler.MultiStepLR(FlatSGD(...)) -- the
    reference's schedule, main_shot_vae.py:199,252-254 -- and anything else that walks param_groups works on it; step() is one
    sv_sgd launch on the engine's flat buffers, not the per-tensor loop."""

    def __init__(self, model, lr=0.1, momentum=0.9, weight_decay=5e-4):
        self.model = model
        self._steps = 0
        super().__init__([p for p, _, _ in model._views], dict(lr=lr, momentum=momentum, weight_decay=weight_decay))

    def zero_grad(self, set_to_none=False):          # (the gradients are views of ONE buffer: zeroed, never dropped)
        eng = self.model._engine
        eng.grad.zero_()
        self.model._attach_grads()

    def step(self, closure=None, grad_scale=1.0):
        if closure is not None:
            raise NotImplementedError("FlatSGD.step: no closure (the reference's loop has none, main_shot_vae.py:365)")
        eng = self.model._engine
        g = self.param_groups[0]
        if eng.mom is None or eng.mom.device != eng.param.device:
            eng.mom = torch.zeros_like(eng.param)
            self._steps = 0
        L.check_flag_timeouts("FlatSGD.step")      # never apply gradients a failed side-stream wait may have corrupted
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.call("sv_sgd", C.c_void_p(eng.param.data_ptr()), C.c_void_p(eng.grad.data_ptr()),
               C.c_void_p(eng.mom.data_ptr()), eng.param.numel(), float(g["lr"]), float(g["momentum"]),
               float(g["weight_decay"]), float(grad_scale), int(self._steps == 0), st)
        self._steps += 1
        eng.mark_dirty()

    # ---- checkpoint format: torch.optim.SGD's (main_shot_vae.py:237-242 saves optimizer.state_dict(), :207 loads it), so a
    #      checkpoint written by the reference loop resumes here and vice versa.  Parameter ids follow model.parameters().
    def _views(self):
        return self.model._views          # [(nn.Parameter, kind, payload)] in registration (= reference) order

    def state_dict(self):
        eng = self.model._engine
        g = self.param_groups[0]
        n = len(self._views())
        group = dict(lr=g["lr"], momentum=g["momentum"], dampening=0, weight_decay=g["weight_decay"], nesterov=False,
                     maximize=False, foreach=None, differentiable=False, fused=None, params=list(range(n)))
        state = {}
        if eng.mom is not None and self._steps > 0:
            for i, (_, kind, payload) in enumerate(self._views()):
                state[i] = dict(momentum_buffer=self.model._flat_view(eng.mom, kind, payload).clone())
        return dict(state=state, param_groups=[group])

    def load_state_dict(self, sd):
        eng = self.model._engine
        g = sd["param_groups"][0]
        if g.get("nesterov") or g.get("dampening", 0) != 0:
            raise NotImplementedError("FlatSGD implements the reference's SGD (no Nesterov, dampening 0)")
        self.param_groups[0].update(lr=g["lr"], momentum=g["momentum"], weight_decay=g["weight_decay"])
        if "initial_lr" in g:                      # (written by an lr scheduler: it resumes from there)
            self.param_groups[0]["initial_lr"] = g["initial_lr"]
        state = sd.get("state", {})
        bufs = {int(k): v.get("momentum_buffer") for k, v in state.items()}
        if not any(b is not None for b in bufs.values()):
            eng.mom, self._steps = None, 0
            return
        eng.mom = torch.zeros_like(eng.param)
        for i, (_, kind, payload) in enumerate(self._views()):
            b = bufs.get(i)
            if b is not None:
                self.model._flat_view(eng.mom, kind, payload).copy_(b.to(eng.param.device))
        self._steps = 1           # momentum buffers exist: the next step is not a "first step" (v = g)


class FlatAdam:
    """torch.optim.Adam(params, lr, betas, eps) -- no weight decay, no amsgrad: the optimizer of the smooth-ELBO trainers
    (main_smooth_ELBO_svhn.py:428) -- on ONE flat fp32 buffer: the parameters are re-pointed to views of it (as are their
    .grad), so a step is one sv_adam launch instead of ~60 small kernels per tensor list, and a data-parallel step reduces
    one buffer.  Construct it AFTER moving the module to the GPU.  capturable=True keeps the step count on the device, so
    that step() can be captured into a hipGraph.  state_dict() / load_state_dict() use torch.optim.Adam's format."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, capturable=False):
        self.params = [p for p in params]
        assert self.params and all(p.is_cuda and p.dtype == torch.float32 for p in self.params), \
            "FlatAdam: fp32 parameters on an MI355X (move the module to the GPU first)"
        dev = self.params[0].device
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + 63) // 64 * 64
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros_like(self.flat)
        self.m, self.v = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        for p, o in zip(self.params, self.offsets):
            self.flat[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + p.numel()].view(p.shape)
        self._attach()
        self.param_groups = [dict(lr=lr, betas=tuple(betas), eps=eps)]
        self.steps = 0
        self.step_dev = torch.zeros((), dtype=torch.float32, device=dev) if capturable else None

    def _attach(self):
        for p, o in zip(self.params, self.offsets):
            g = p.grad
            if g is None or g.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                p.grad = self.flat_grad[o:o + p.numel()].view(p.shape)

    def zero_grad(self, set_to_none=False):
        self.flat_grad.zero_()
        self._attach()

    def step(self, grad_scale=1.0):
        g = self.param_groups[0]
        self.steps += 1
        if self.step_dev is not None:
            self.step_dev += 1
        self._attach()
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.call("sv_adam", C.c_void_p(self.flat.data_ptr()), C.c_void_p(self.flat_grad.data_ptr()), C.c_void_p(self.m.data_ptr()),
               C.c_void_p(self.v.data_ptr()), self.flat.numel(), float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
               float(g["eps"]), float(self.steps), C.c_void_p(self.step_dev.data_ptr()) if self.step_dev is not None else None,
               float(grad_scale), st)

    def state_dict(self):
        g = self.param_groups[0]
        group = dict(lr=g["lr"], betas=g["betas"], eps=g["eps"], weight_decay=0, amsgrad=False, maximize=False, foreach=None,
                     capturable=self.step_dev is not None, differentiable=False, fused=None, decoupled_weight_decay=False,
                     params=list(range(len(self.params))))
        state = {}
        if self.step_dev is not None:
            # hipGraph replays advance only the captured device counter: it is the truth (a D2H read, outside any capture)
            self.steps = int(round(float(self.step_dev.item())))
        if self.steps > 0:
            for i, (p, o) in enumerate(zip(self.params, self.offsets)):
                state[i] = dict(step=torch.tensor(float(self.steps)), exp_avg=self.m[o:o + p.numel()].view(p.shape).clone(),
                                exp_avg_sq=self.v[o:o + p.numel()].view(p.shape).clone())
        return dict(state=state, param_groups=[group])

    def load_state_dict(self, sd):
        g = sd["param_groups"][0]
        if g.get("weight_decay", 0) != 0 or g.get("amsgrad", False):
            raise NotImplementedError("FlatAdam implements the reference's Adam (no weight decay, no amsgrad)")
        self.param_groups = [dict(lr=g["lr"], betas=tuple(g["betas"]), eps=g["eps"])]
        self.m.zero_()
        self.v.zero_()
        self.steps = 0
        for k, st in sd.get("state", {}).items():
            i = int(k)
            p, o = self.params[i], self.offsets[i]
            self.m[o:o + p.numel()].copy_(st["exp_avg"].reshape(-1).to(self.m.device))
            self.v[o:o + p.numel()].copy_(st["exp_avg_sq"].reshape(-1).to(self.v.device))
            self.steps = max(self.steps, int(float(st["step"])))
        if self.step_dev is not None:
            self.step_dev.fill_(float(self.steps))
