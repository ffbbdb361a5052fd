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

    def state_dict(self):
        eng = self.model._engine
        return dict(param_groups=self.param_groups, steps=self._steps,
                    momentum=None if eng.mom is None else eng.mom.clone())

    def load_state_dict(self, sd):
        self.param_groups = sd["param_groups"]
        self._steps = sd["steps"]
        if sd["momentum"] is not None:
            self.model._engine.mom = sd["momentum"].to(self.model._engine.param.device).clone()
