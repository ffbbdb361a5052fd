"""SGD(momentum, weight decay) on the flat parameter buffer: one HIP kernel per step instead of 107
per-tensor updates.  Semantics of torch.optim.SGD as used at main_shot_vae.py:198,365-366
(no Nesterov, dampening 0, first step v = g)."""
import ctypes as C

import torch

from . import _lib as L


class FlatSGD:
    def __init__(self, model, lr=0.1, momentum=0.9, weight_decay=5e-4):
        self.model = model
        self.param_groups = [dict(lr=lr, momentum=momentum, weight_decay=weight_decay)]
        self._steps = 0

    def zero_grad(self):
        eng = self.model._engine
        eng.grad.zero_()
        self.model._attach_grads()

    def step(self, grad_scale=1.0):
        eng = self.model._engine
        g = self.param_groups[0]
        if eng.mom is None or eng.mom.device != eng.param.device:
            eng.mom = torch.zeros_like(eng.param)
            self._steps = 0
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
        self.param_groups = [dict(lr=g["lr"], momentum=g["momentum"], weight_decay=g["weight_decay"])]
        if g.get("nesterov") or g.get("dampening", 0) != 0:
            raise NotImplementedError("FlatSGD implements the reference's SGD (no Nesterov, dampening 0)")
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
