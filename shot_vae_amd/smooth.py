"""One-stage smooth-ELBO VAEs on the HIP path: drop-ins for smooth_vae_model/svhn_vae.py
(svhn_VAE), smooth_vae_model/mnist_vae.py (mnist_VAE), the loss of the Trainer in main_smooth_ELBO_svhn.py:228-388 and
one iteration of its loop (:152-176).

Every convolution, transposed convolution and Linear layer of these models is the same gather-GEMM the SHOT-VAE step
uses (sv_igemm / sv_wgrad / sv_colsum through the C ABI): 4x4 stride-2 convs as `conv_like`, the 4x4 stride-2 transposed
convs as `convT_like` (four sub-pixel phases), the Linear layers as 1x1 GEMMs (the two that touch the 4x4 feature map
with their weights permuted between the reference's (c, y, x) flattening and the NHWC one); the ReLU between two layers is
the consumer's load prologue (scale 1, shift 0, slope 0) and, in the backward, the producer's activation-backward
epilogue, so no activation tensor is ever materialised.  Each layer is one torch.autograd.Function over those calls (its weights pre-packed, and its weight / bias gradients scattered
back, by the table-driven launches of `_WeightPlan`: two gathers per forward, one scatter per backward pass, for all layers);
the heads' softmax, both samplers and the decoder input are one launch each way (sv_smooth_latent_fwd / _bwd), the Tanh +
layout change of the reconstruction another (sv_tanh_to_nchw), the trainer's loss two launches forward and one backward
(sv_smooth_elbo_fwd / _bwd), and Adam one launch on a flat parameter buffer (optim.FlatAdam over sv_adam;
torch.optim.Adam also works).  Parameters keep the reference's names and shapes (state_dict compatible).

There is no CPU fallback: the layers raise when the HIP library or a GPU is missing."""
import ctypes as C
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from . import geometry as G

EPS = 1e-12
KINDS = {"svhn": dict(in_ch=3, widths=(32, 64, 128), dec=(64, 32), hidden=512),
         "mnist": dict(in_ch=1, widths=(32, 64, 64), dec=(32, 32), hidden=256)}


def _vp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _pad16(n):
    return (n + 15) // 16 * 16


_CONST = {}


def _one_zero(n, dev):
    """constant [n] vectors of ones / zeros (the identity BatchNorm coefficients of the ReLU prologue / epilogue), made once
    per size and device instead of two fill launches per layer and direction"""
    key = (n, dev)
    v = _CONST.get(key)
    if v is None:
        v = _CONST[key] = (torch.ones(n, device=dev), torch.zeros(n, device=dev))
    return v


def _al64(n):
    return (n + 63) // 64 * 64


# layer -> the layer that consumes its output through a ReLU and nothing else (no re-flattening in between): the consumer's data
# gradient leaves sum(dy) per channel of THIS layer's output in its epilogue sums
BIAS_FROM_NEXT = {"c1": "c2", "c2": "c3", "f1": "heads", "g1": "g2", "t1": "t2", "t2": "t3"}


class _WeightPlan:
    """Everything between the nn.Parameters of a SmoothVAE (torch's own layouts, reference names and shapes) and the kernels,
    as TABLES: one sv_param_gather launch packs every layer's forward and data-gradient weights (bf16 / fp32 MFMA operand order,
    per sub-pixel phase) straight from the parameter tensors, a second one the padded fp32 bias vectors; the weight / bias
    gradients of a backward pass accumulate in ONE zeroed master-layout scratch (sv_wgrad / sv_colsum add) and one
    sv_param_scatter_add launch adds them to the parameters' .grad tensors when the backward pass ends.  (Round 4 issued, per
    iteration, 19 re-pack launches, 19 gradient-add kernels of autograd, and the permute / cat / pad copies of the Linear layers
    whose rows are a permuted (c, y, x) flattening: ~60 of the iteration's ~140 launches.)

    A source is (weight, bias, first row n0, rows) -- the three heads share one GEMM; a description maps (n, tap, c) of the layer
    onto the parameter: n = n_hi * n_lo_count + n_lo with strides (sn_hi, sn_lo), taps st, channels sc."""

    def __init__(self, model, dev):
        self.dev, self.tdt, self.code = dev, model._tdt, model._code()
        self.layers = model._L
        es_off = 0
        self.fwd_off, self.dg_off, self.bias_off, self.dw_off, self.db_off, self.bs_off = {}, {}, {}, {}, {}, {}
        boff = goff = 0
        for name, l in self.layers.items():
            self.fwd_off[name] = es_off
            es_off = _al64(es_off + max(G.packed_size(l.geom_fwd(1)), 1))
            if name != "c1":                       # (the image needs no gradient: no data-gradient pack for the first layer)
                self.dg_off[name] = es_off
                es_off = _al64(es_off + max(G.packed_size(l.geom_dgrad(1)), 1))
            self.bias_off[name] = boff
            boff = _al64(boff + l.N)
            self.dw_off[name] = goff
            goff = _al64(goff + l.N * l.T * l.Cin)
            self.db_off[name] = goff
            goff = _al64(goff + l.N)
            self.bs_off[name] = goff               # the (unused) sums of the identity BatchNorm of the ReLU-backward epilogue: doubles
            goff = _al64(goff + 4 * l.Cin)
        self.pack = torch.zeros(es_off, dtype=self.tdt, device=dev)          # padding rows / columns stay zero for ever
        self.bias = torch.zeros(boff, dtype=torch.float32, device=dev)
        self.gscr = torch.zeros(goff, dtype=torch.float32, device=dev)
        self.n_pass = goff
        self.key = self.gkey = None
        self.model = model
        self.flushed, self.cb_task = True, None
        self.wg_ws = torch.empty(512 * 8192, dtype=torch.float32, device=dev)     # sv_wgrad's workspace (16 MB: the launches of a pass run one after the other)
        self.wver = None                 # version stamp of the parameter VALUES the packs hold (sum of the tensors' version counters)
        self.ran = set()                 # layers whose backward ran in the current pass
        self._gbuf = {}                  # persistent .grad buffers (id(param) -> tensor): re-attached, zeroed, never re-allocated
        self._fresh = []                 # parameters whose .grad this pass had to create (None before it)

    # ---- job tables ---------------------------------------------------------------------------------------------------------
    @staticmethod
    def _job(ptr, dst_off, size, N, C, ntap, transpose, n_real, c_real, sn_hi, sn_lo, n_lo_count, st, sc, torig=None, dst_ld=0):
        j = L.SvParamJob()
        j.ptr, j.dst_off, j.size, j.dst_ld = ptr, dst_off, size, dst_ld
        j.sn_hi, j.sn_lo, j.st, j.sc, j.n_lo_count = sn_hi, sn_lo, st, sc, n_lo_count
        j.N, j.C, j.ntap, j.transpose, j.n_real, j.c_real = N, C, ntap, transpose, n_real, c_real
        for t in range(L.MAX_TAPS):
            j.torig[t] = (torig[t] if torig is not None else t) if t < ntap else 0
        return j

    def _upload(self, jobs):
        b0 = 0
        for j in jobs:
            j.block0 = b0
            b0 += max((j.size + 1023) // 1024, 1)
        arr = (L.SvParamJob * len(jobs))(*jobs)
        raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.dev)
        return raw, len(jobs), b0

    def _build(self, model, grads):
        """grads=False: the gather tables (weights, biases); True: the scatter table of the gradients"""
        srcs = model._sources()
        wj, bj, gj = [], [], []
        for name, l in self.layers.items():
            for (w, b, n0, rows, d) in srcs[name]:
                # d: description of (n, tap, c) on the weight tensor -- T (taps of the DESCRIPTION), C (padded), c_real, n_lo_count,
                #    sn_hi, sn_lo, st, sc, bias strides bn_hi / bn_lo; d["pseudo"]: T > 1 describes the channel index of a 1x1 layer (f1)
                T, Cd, cr = d["T"], d["C"], d["c_real"]
                common = dict(n_real=rows, c_real=cr, sn_hi=d["sn_hi"], sn_lo=d["sn_lo"], n_lo_count=d["n_lo_count"], st=d["st"], sc=d["sc"])
                bcommon = dict(n_real=rows, c_real=1, sn_hi=d["bn_hi"], sn_lo=d["bn_lo"], n_lo_count=d["n_lo_count"], st=0, sc=0)
                if grads:
                    # master layout [N][T][Cin] (+ db [N]) in the scratch -> += the parameter's .grad
                    gT, gC = (T, Cd) if d.get("pseudo") else (l.T, l.Cin)
                    gj.append(self._job(w.grad.data_ptr(), self.dw_off[name] + n0 * gT * gC, rows * gT * cr, rows, gC, gT, 0, **common))
                    nxt = BIAS_FROM_NEXT.get(name) if not L.det_stats() else None      # (fixed-order mode: per-wave replicas elsewhere)
                    if nxt is not None:
                        # the bias gradient = the column sums of this layer's dy = the first of the two per-channel sums the NEXT
                        # layer's data gradient accumulated in its activation-backward epilogue (doubles): no sv_colsum pass
                        gj.append(self._job(b.grad.data_ptr(), self.bs_off[nxt] + 2 * n0, rows, rows, 1, 1, 1, **bcommon))
                    else:
                        gj.append(self._job(b.grad.data_ptr(), self.db_off[name] + n0, rows, rows, 1, 1, 0, **bcommon))
                    continue
                gf, gd = l.geom_fwd(1), l.geom_dgrad(1)
                multi = len(srcs[name]) > 1
                Nrows = rows if multi else l.N                        # a single source also owns the padding rows (zeros)
                if d.get("pseudo"):
                    # forward pack [n][cin'] with cin' = t * C + c: exactly the dense [n][t][c] order of the description
                    wj.append(self._job(w.data_ptr(), self.fwd_off[name], Nrows * T * Cd, Nrows, Cd, T, 0, **common))
                    # data-gradient pack [cin'][n]: the transposed matrix described row by row (cin' = (t, c): a two-level row index)
                    wj.append(self._job(w.data_ptr(), self.dg_off[name], T * Cd * l.N, T * Cd, l.N, 1, 0, n_real=T * cr, c_real=rows,
                                        sn_hi=d["st"], sn_lo=d["sc"], n_lo_count=Cd, st=0, sc=d["sn_hi"]))
                else:
                    for ph in range(gf.nphase):
                        P = gf.phase[ph]
                        if P.ntap:
                            wj.append(self._job(w.data_ptr(), self.fwd_off[name] + P.w_off + n0 * P.ntap * l.Cin, Nrows * P.ntap * l.Cin,
                                                Nrows, l.Cin, P.ntap, 0, torig=list(P.torig), **common))
                    for ph in range(gd.nphase if name in self.dg_off else 0):
                        P = gd.phase[ph]
                        if not P.ntap:
                            continue
                        if multi:
                            # several sources fill column ranges [n0, n0 + rows) of ONE [c][n] pack (1x1 layers): each is the
                            # transposed matrix of its parameter described row by row, with the pack's row stride
                            assert P.ntap == 1 and d["n_lo_count"] == 1
                            wj.append(self._job(w.data_ptr(), self.dg_off[name] + P.w_off + n0, l.Cin * rows, l.Cin, rows, 1, 0,
                                                n_real=cr, c_real=rows, sn_hi=d["sc"], sn_lo=0, n_lo_count=1, st=0, sc=d["sn_hi"],
                                                dst_ld=l.N))
                        else:
                            wj.append(self._job(w.data_ptr(), self.dg_off[name] + P.w_off, l.N * P.ntap * l.Cin, l.N, l.Cin, P.ntap, 1,
                                                torig=list(P.torig), **common))
                # bias: [rows] (a permuted flattening for g2) into the layer's padded fp32 vector
                bj.append(self._job(b.data_ptr(), self.bias_off[name] + n0, Nrows, Nrows, 1, 1, 0, **bcommon))
        if grads:
            self.gjobs = self._upload(gj)
        else:
            self.wjobs, self.bjobs = self._upload(wj), self._upload(bj)

    def refresh(self, model, for_backward):
        """packs + biases from the CURRENT parameter values (two launches); with a backward pass to come, the gradient scratch
        is cleared (one launch).  The tables hold absolute pointers: rebuilt when a parameter or its .grad moved."""
        k = tuple(p.data_ptr() for p in model.parameters())
        if k != self.key:
            self._build(model, False)
            self.key = k
        self.wver = sum(p._version for p in model.parameters())
        raw, n, nb = self.wjobs
        L.call("sv_param_gather", self.code, _vp(raw), n, nb, _vp(self.pack), _st())
        raw, n, nb = self.bjobs
        L.call("sv_param_gather", L.SV_F32, _vp(raw), n, nb, _vp(self.bias), _st())
        if for_backward:
            self.gscr.zero_()
            self.flushed = False
            self.cb_task = None            # (a backward pass that died before its callback ran must not mute the next one)

    def _grad_table(self, model):
        """Every parameter gets a .grad the scatter can add into.  The buffers are PERSISTENT: torch's default
        `zero_grad(set_to_none=True)` drops .grad every iteration -- a fresh tensor each time moved every pointer of the scatter table,
        i.e. a table rebuild and a synchronous host-to-device upload per iteration; the same tensors are zeroed (one
        foreach launch) and re-attached instead.  Parameters that had no .grad before this pass are remembered: flush() hands
        .grad = None back to those whose layer took no part in it (torch leaves them None, and Adam skips them)."""
        self._fresh, zero = [], []
        for p in model.parameters():
            if p.grad is None or not p.grad.is_contiguous() or p.grad.dtype != torch.float32:
                buf = self._gbuf.get(id(p))
                if buf is None or buf.shape != p.shape or buf.device != p.device:
                    buf = self._gbuf[id(p)] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                else:
                    zero.append(buf)
                p.grad = buf
                self._fresh.append(p)
        if zero:
            torch._foreach_zero_(zero)
        k = tuple(p.grad.data_ptr() for p in model.parameters()) + (L.det_stats(),)
        if k != self.gkey:
            self._build(model, True)
            self.gkey = k

    def before_layer_backward(self, model, name=None, wver=None):
        """called by every layer's backward: a SECOND backward pass without a forward in between starts from a cleared scratch;
        the first layer of a pass queues the flush for the end of that pass"""
        if wver is not None and wver != self.wver:
            # (torch raises its in-place version error here: the graph was built on other parameter values than the packs hold now)
            raise RuntimeError("shot_vae_amd.smooth: the parameters were modified (optimizer step / in-place update) after the forward "
                               "pass this backward belongs to -- its data gradients would use the NEW weights.  Run the forward again.")
        if self.flushed:
            self.gscr.zero_()
            self.flushed = False
            self.ran = set()
        if name is not None:
            self.ran.add(name)
        # one flush per BACKWARD PASS, keyed by the autograd engine's graph task -- not by a sticky flag: the engine drops its queued
        # callbacks when a node of the pass raises (OOM retry, KeyboardInterrupt, ShotVaeHipError), a flag set there would never be
        # cleared and every later pass would leave .grad untouched without an error
        task = torch._C._current_graph_task_id()
        if task != self.cb_task or task < 0:
            if self.cb_task is not None and not self.flushed:
                self.gscr.zero_()          # the pass that queued last never flushed: its partial gradients are not this pass's
                self.ran = {name} if name is not None else set()
            self.cb_task = task
            torch.autograd.Variable._execution_engine.queue_callback(self.flush)

    def flush(self):
        """end of a backward pass: the scratch's weight / bias gradients += the parameters' .grad (one launch)"""
        self.cb_task = None
        if self.flushed:
            return
        self._grad_table(self.model)
        raw, n, nb = self.gjobs
        L.call("sv_param_scatter_add", _vp(raw), n, nb, _vp(self.gscr), _st())
        self.flushed = True
        if self._fresh and len(self.ran) < len(self.layers):
            # layers this pass never reached (e.g. the decoder under a loss on encode() alone): their scratch is zero, the scatter
            # added nothing -- parameters that had no .grad before the pass get None back, as torch leaves them
            idle = set()
            srcs = self.model._sources()
            for lname in self.layers:
                if lname not in self.ran:
                    for (w, b, *_rest) in srcs[lname]:
                        idle.update((id(w), id(b)))
            for p in self._fresh:
                if id(p) in idle:
                    p.grad = None
        self._fresh = []


class _Layer:
    """Static description of one conv-like layer: kind 'conv' (Conv2d / Linear-as-conv) or 'convT'."""

    def __init__(self, kind, k, stride, pad, cin, n, hin, cin_real=None, n_real=None):
        self.kind, self.k, self.stride, self.pad = kind, k, stride, pad
        self.Cin, self.N, self.Hin = cin, n, hin
        self.cin_real, self.n_real = cin_real or cin, n_real or n
        self.T = k * k
        self._g = {}

    @property
    def Hout(self):
        if self.kind == "conv":
            return (self.Hin + 2 * self.pad - self.k) // self.stride + 1
        return self.Hin * self.stride

    def geom_fwd(self, B):
        g = self._g.get(("f", B))
        if g is None:
            f = G.conv_like if self.kind == "conv" else G.convT_like
            g = self._g[("f", B)] = f(B, self.Hin, self.Hin, self.Cin, self.N, self.k, self.stride, self.pad)
        return g

    def geom_dgrad(self, B):
        g = self._g.get(("d", B))
        if g is None:
            f = G.convT_like if self.kind == "conv" else G.conv_like
            g = self._g[("d", B)] = f(B, self.Hout, self.Hout, self.N, self.Cin, self.k, self.stride, self.pad)
        return g


class _ConvLikeFn(torch.autograd.Function):
    """out = conv_like(ReLU?(x)) + bias through sv_igemm on the pre-packed weights of the model's _WeightPlan; backward =
    sv_igemm (data gradient with the ReLU backward as its epilogue), sv_wgrad and sv_colsum into the plan's gradient scratch.
    The parameters are not inputs of the node (`token` keeps it in the graph): their gradients reach .grad through the
    plan's one scatter launch at the end of the backward pass."""

    @staticmethod
    def forward(ctx, x, token, model, name, relu_in):
        if not x.is_cuda:
            raise L.ShotVaeHipError("shot_vae_amd runs on an MI355X only (no CPU fallback)")
        plan, layer = model._plan_for(x.device), model._L[name]
        code, tdt = plan.code, plan.tdt
        x = x.contiguous()
        B, dev = x.shape[0], x.device
        gf = layer.geom_fwd(B)
        es = plan.pack.element_size()
        out = torch.empty(B, layer.Hout, layer.Hout, layer.N, dtype=tdt, device=dev)
        a = L.SvIgemmArgs()
        a.x, a.w, a.out, a.replicas = x.data_ptr(), plan.pack.data_ptr() + es * plan.fwd_off[name], out.data_ptr(), 1
        if relu_in:
            one, zero = _one_zero(layer.Cin, dev)
            a.pro_scale, a.pro_shift, a.pro_slope = one.data_ptr(), zero.data_ptr(), 0.0
        a.bias = plan.bias.data_ptr() + 4 * plan.bias_off[name]
        L.call("sv_igemm", C.byref(gf), code, C.byref(a), _st())
        ctx.save_for_backward(x)
        ctx.cfg = (model, name, relu_in)
        ctx.wver = plan.wver
        return out

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        model, name, relu_in = ctx.cfg
        plan, layer = model._plan_for(x.device), model._L[name]
        code = plan.code
        plan.before_layer_backward(model, name, ctx.wver)
        B, dev = x.shape[0], x.device
        dy = dy.contiguous()
        gf, gd = layer.geom_fwd(B), layer.geom_dgrad(B)
        one, zero = _one_zero(layer.Cin, dev)
        es = plan.pack.element_size()
        gbase = plan.gscr.data_ptr()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            a = L.SvIgemmArgs()
            a.x, a.w, a.out, a.replicas = dy.data_ptr(), plan.pack.data_ptr() + es * plan.dg_off[name], dx.data_ptr(), 1
            if relu_in:     # ReLU backward fused as the activation-backward epilogue (BN part: identity statistics)
                a.ex, a.ex_scale, a.ex_shift = x.data_ptr(), one.data_ptr(), zero.data_ptr()
                a.ex_mean, a.ex_rstd, a.ex_slope, a.bsums = zero.data_ptr(), one.data_ptr(), 0.0, gbase + 4 * plan.bs_off[name]
                if L.det_stats():            # (the sums are not used here, but the launch wants one replica per wave)
                    a.replicas = L.det_replicas(gd, code, a)
                    ctx.bs_keep = torch.zeros(2 * a.replicas * 2 * layer.Cin, dtype=torch.float32, device=dev)
                    a.bsums = ctx.bs_keep.data_ptr()
            L.call("sv_igemm", C.byref(gd), code, C.byref(a), _st())
        # (the workspace: per-block slabs of the thin layers' gradients, k4wgrad.hip -- 256 blocks adding 8 192 floats each to the same
        #  addresses cost more than the whole product)
        L.call("sv_wgrad", C.byref(gf), code, _vp(x), _vp(one) if relu_in else None, _vp(zero) if relu_in else None, 0.0, _vp(dy),
               C.c_void_p(gbase + 4 * plan.dw_off[name]), 0, 1, _vp(plan.wg_ws), plan.wg_ws.numel(), 1, _st())
        if name not in BIAS_FROM_NEXT or L.det_stats():      # (see _WeightPlan: most bias gradients are a by-product of the next layer's data gradient)
            L.call("sv_colsum", code, _vp(dy), dy.numel() // layer.N, layer.N, layer.N, C.c_void_p(gbase + 4 * plan.db_off[name]), _st())
        return dx, None, None, None, None


class _LatentFn(torch.autograd.Function):
    """Everything between the head GEMM and the decoder (svhn_vae.py:137-208) as one launch each way: softmax of the
    discrete logits, the reparameterised normal sample, the Gumbel-softmax sample, one-hot of the label, the decoder's
    padded input (sv_smooth_latent_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, o, eps, u, label, temperature, training, Dc, Dd, Lpad, code):
        B, ldo, dev = o.shape[0], o.shape[-1], o.device
        o = o.contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        mean, logvar = torch.empty(B, Dc, **f32), torch.empty(B, Dc, **f32)
        alpha, gs = torch.empty(B, Dd, **f32), torch.empty(B, Dd, **f32)
        latent = torch.empty(B, 1, 1, Lpad, dtype=o.dtype, device=dev)
        lat32 = torch.empty(B, Dc + Dd, **f32)
        L.call("sv_smooth_latent_fwd", code, _vp(o), ldo, _vp(eps), _vp(u), _vp(label), float(temperature), int(training), B, Dc,
               Dd, Lpad, _vp(mean), _vp(logvar), _vp(alpha), _vp(gs), _vp(latent), _vp(lat32), _st())
        ctx.save_for_backward(logvar, eps, alpha, gs)
        ctx.cfg = (temperature, training, label is None, Dc, Dd, Lpad, ldo, code, o.dtype)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(gs, lat32)
        return mean, logvar, alpha, gs, latent, lat32

    @staticmethod
    def backward(ctx, d_mean, d_logvar, d_alpha, _d_gs, d_latent, _d_lat32):
        logvar, eps, alpha, gs = ctx.saved_tensors
        temperature, training, sample_path, Dc, Dd, Lpad, ldo, code, tdt = ctx.cfg
        B, dev = logvar.shape[0], logvar.device
        if d_latent is None:
            d_latent = torch.zeros(B, Lpad, dtype=tdt, device=dev)
        c = lambda t: t.contiguous().float() if t is not None else None
        d_mean, d_logvar, d_alpha = c(d_mean), c(d_logvar), c(d_alpha)
        d_latent = d_latent.contiguous()
        d_o = torch.empty(B, 1, 1, ldo, dtype=tdt, device=dev)
        L.call("sv_smooth_latent_bwd", code, _vp(d_latent), Lpad, _vp(d_mean), _vp(d_logvar), _vp(d_alpha), _vp(logvar), _vp(eps),
               _vp(alpha), _vp(gs), float(temperature), int(training), int(sample_path), B, Dc, Dd, _vp(d_o), ldo, _st())
        return (d_o,) + (None,) * 9


class _TanhNchwFn(torch.autograd.Function):
    """reconstruction = tanh(decoder output[..., :C]) as NCHW fp32 (svhn_vae.py:118-120): one launch each way"""

    @staticmethod
    def forward(ctx, f, C_, code):
        f = f.contiguous()
        B, H, W, ld = f.shape
        out = torch.empty(B, C_, H, W, dtype=torch.float32, device=f.device)
        L.call("sv_tanh_to_nchw", code, _vp(f), B, C_, H, W, ld, _vp(out), _st())
        ctx.save_for_backward(out)
        ctx.cfg = (ld, code, f.dtype)
        return out

    @staticmethod
    def backward(ctx, d_out):
        out, = ctx.saved_tensors
        ld, code, tdt = ctx.cfg
        B, C_, H, W = out.shape
        d_f = torch.empty(B, H, W, ld, dtype=tdt, device=out.device)
        L.call("sv_tanh_to_nchw_bwd", code, _vp(d_out.contiguous().float()), _vp(out), B, C_, H, W, ld, _vp(d_f), _st())
        return d_f, None, None


class _SmoothLossFn(torch.autograd.Function):
    """Trainer._loss_function (main_smooth_ELBO_svhn.py:228-310) in two launches forward (reductions, composition) and one
    backward (sv_smooth_elbo_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, data, recon, mean, logvar, alpha, label, sched, steps_dev):
        for t in (data, recon, mean, logvar, alpha):
            if not t.is_cuda:
                raise L.ShotVaeHipError("shot_vae_amd runs on an MI355X only (no CPU fallback)")
        data, recon = data.contiguous().float(), recon.contiguous().float()
        mean, logvar, alpha = mean.contiguous().float(), logvar.contiguous().float(), alpha.contiguous().float()
        label = label.contiguous().long() if label is not None else None
        B, Dc, Dd = data.shape[0], mean.shape[1], alpha.shape[1]
        terms = torch.zeros(9, dtype=torch.float32, device=data.device)
        coef = torch.empty(4, dtype=torch.float32, device=data.device)
        L.call("sv_smooth_elbo_fwd", _vp(data), _vp(recon), data[0].numel(), _vp(mean), _vp(logvar), _vp(alpha), _vp(label), B, Dc,
               Dd, C.byref(sched), _vp(steps_dev), _vp(terms), _vp(coef), _st())
        ctx.save_for_backward(data, recon, mean, logvar, alpha, coef)
        ctx.label = label
        ctx.mark_non_differentiable(terms)
        return terms[4], terms

    @staticmethod
    def backward(ctx, g, _g_terms):
        data, recon, mean, logvar, alpha, coef = ctx.saved_tensors
        B, Dc, Dd = data.shape[0], mean.shape[1], alpha.shape[1]
        d_rec, d_mean, d_logvar, d_alpha = (torch.empty_like(t) for t in (recon, mean, logvar, alpha))
        g = g.contiguous().float().view(1)
        L.call("sv_smooth_elbo_bwd", _vp(data), _vp(recon), data[0].numel(), _vp(mean), _vp(logvar), _vp(alpha), _vp(ctx.label), B,
               Dc, Dd, _vp(coef), _vp(g), _vp(d_rec), _vp(d_mean), _vp(d_logvar), _vp(d_alpha), _st())
        return None, d_rec, d_mean, d_logvar, d_alpha, None, None, None


class SmoothVAE(nn.Module):
    """svhn_VAE / mnist_VAE (smooth_vae_model/svhn_vae.py:8-300): same constructor arguments, parameter names and
    forward contract -- forward(x, label=None) -> (reconstruction after Tanh, {'cont': [mean, logvar], 'disc': [alpha]},
    latent sample, [discrete sample]).  `kind` selects the widths ('svhn': 3x32x32 input, 'mnist': 1x32x32)."""

    def __init__(self, img_size, latent_spec, temperature=.67, use_cuda=True, kind=None, compute_dtype="bf16"):
        super().__init__()
        kind = kind or ("svhn" if img_size[0] == 3 else "mnist")
        k = KINDS[kind]
        assert tuple(img_size) == (k["in_ch"], 32, 32), "the MI355X path covers the reference's 32x32 inputs"
        assert "cont" in latent_spec and len(latent_spec.get("disc", [])) == 1, \
            "one continuous block and one categorical latent (the trainers' latent_spec)"
        self.kind, self.img_size, self.latent_spec, self.temperature = kind, tuple(img_size), latent_spec, temperature
        self.use_cuda, self.compute_dtype = use_cuda, compute_dtype
        self.num_pixels = img_size[0] * img_size[1] * img_size[2]
        self.is_continuous = self.is_discrete = True
        self.latent_cont_dim, self.latent_disc_dim = latent_spec["cont"], latent_spec["disc"][0]
        self.latent_dim = self.latent_cont_dim + self.latent_disc_dim
        self.hidden_dim = k["hidden"]
        w1, w2, w3 = k["widths"]
        d1, d2 = k["dec"]
        self.reshape = (w3, 4, 4)
        ch, h, lat = k["in_ch"], self.hidden_dim, self.latent_dim
        # parameters in the reference's modules (same state_dict keys / shapes); only their tensors are used
        self.img_to_features = nn.Sequential(nn.Conv2d(ch, w1, 4, 2, 1), nn.ReLU(), nn.Conv2d(w1, w2, 4, 2, 1), nn.ReLU(),
                                             nn.Conv2d(w2, w3, 4, 2, 1), nn.ReLU())
        self.features_to_hidden = nn.Sequential(nn.Linear(w3 * 16, h), nn.ReLU())
        self.fc_mean, self.fc_log_var = nn.Linear(h, self.latent_cont_dim), nn.Linear(h, self.latent_cont_dim)
        self.fc_alphas = nn.ModuleList([nn.Linear(h, self.latent_disc_dim)])
        self.latent_to_features = nn.Sequential(nn.Linear(lat, h), nn.ReLU(), nn.Linear(h, w3 * 16), nn.ReLU())
        self.features_to_img = nn.Sequential(nn.ConvTranspose2d(w3, d1, 4, 2, 1), nn.ReLU(),
                                             nn.ConvTranspose2d(d1, d2, 4, 2, 1), nn.ReLU(),
                                             nn.ConvTranspose2d(d2, ch, 4, 2, 1), nn.Tanh())
        NH = 2 * self.latent_cont_dim + self.latent_disc_dim
        self._L = dict(
            c1=_Layer("conv", 4, 2, 1, 16, w1, 32, cin_real=ch), c2=_Layer("conv", 4, 2, 1, w1, w2, 16),
            c3=_Layer("conv", 4, 2, 1, w2, w3, 8), f1=_Layer("conv", 1, 1, 0, w3 * 16, h, 1),
            heads=_Layer("conv", 1, 1, 0, h, _pad16(NH), 1, n_real=NH),
            g1=_Layer("conv", 1, 1, 0, _pad16(lat), h, 1, cin_real=lat), g2=_Layer("conv", 1, 1, 0, h, w3 * 16, 1),
            t1=_Layer("convT", 4, 2, 1, w3, d1, 4), t2=_Layer("convT", 4, 2, 1, d1, d2, 8),
            t3=_Layer("convT", 4, 2, 1, d2, 16, 16, n_real=ch))
        self._tdt = torch.bfloat16 if compute_dtype == "bf16" else torch.float32
        self._plans = {}
        self._tokens = {}

    def _sources(self):
        """name -> [(weight, bias, first row, rows, description)]: how (n, tap, c) of every layer's GEMM lies in the module's own
        parameters (Conv2d OIHW, ConvTranspose2d IOHW, Linear [out][in]); see _WeightPlan."""
        e, d_, lf, fh = self.img_to_features, self.features_to_img, self.latent_to_features, self.features_to_hidden[0]
        Ls = self._L
        w3, h, lat = self.reshape[0], self.hidden_dim, self.latent_dim

        def conv(l):          # OIHW [n][c][16]
            return dict(T=16, C=l.Cin, c_real=l.cin_real, n_lo_count=1, sn_hi=l.cin_real * 16, sn_lo=0, st=1, sc=16, bn_hi=1, bn_lo=0)

        def convT(l):         # IOHW [c][n][16]
            return dict(T=16, C=l.Cin, c_real=l.cin_real, n_lo_count=1, sn_hi=16, sn_lo=0, st=1, sc=l.n_real * 16, bn_hi=1, bn_lo=0)

        def lin(cin_pad, cin):   # [n][c]
            return dict(T=1, C=cin_pad, c_real=cin, n_lo_count=1, sn_hi=cin, sn_lo=0, st=0, sc=1, bn_hi=1, bn_lo=0)

        out = {}
        for nm, mod in (("c1", e[0]), ("c2", e[2]), ("c3", e[4])):
            out[nm] = [(mod.weight, mod.bias, 0, Ls[nm].n_real, conv(Ls[nm]))]
        # Linear over features.view(B, -1) of the NCHW map, computed on the NHWC map flattened as (y, x, c): input index
        # cin' = (4 y + x) * w3 + c of the GEMM <-> column c * 16 + (4 y + x) of the parameter: sixteen pseudo-taps
        out["f1"] = [(fh.weight, fh.bias, 0, h, dict(T=16, C=w3, c_real=w3, n_lo_count=1, sn_hi=w3 * 16, sn_lo=0, st=1, sc=16, bn_hi=1,
                                                     bn_lo=0, pseudo=True))]
        cdim, ddim = self.latent_cont_dim, self.latent_disc_dim
        out["heads"] = [(self.fc_mean.weight, self.fc_mean.bias, 0, cdim, lin(h, h)),
                        (self.fc_log_var.weight, self.fc_log_var.bias, cdim, cdim, lin(h, h)),
                        (self.fc_alphas[0].weight, self.fc_alphas[0].bias, 2 * cdim, ddim, lin(h, h))]
        out["g1"] = [(lf[0].weight, lf[0].bias, 0, h, lin(Ls["g1"].Cin, lat))]
        # Linear whose output is viewed as (c, 4, 4): the GEMM writes NHWC directly, row n' = (4 y + x) * w3 + c <-> parameter row
        # c * 16 + (4 y + x): a two-level row index (rows and bias alike)
        out["g2"] = [(lf[2].weight, lf[2].bias, 0, w3 * 16, dict(T=1, C=h, c_real=h, n_lo_count=w3, sn_hi=h, sn_lo=16 * h, st=0, sc=1,
                                                                 bn_hi=1, bn_lo=16))]
        for nm, mod in (("t1", d_[0]), ("t2", d_[2]), ("t3", d_[4])):
            out[nm] = [(mod.weight, mod.bias, 0, Ls[nm].n_real, convT(Ls[nm]))]
        return out

    def _plan_for(self, dev):
        key = (dev, self.compute_dtype)
        pl = self._plans.get(key)
        if pl is None:
            pl = self._plans[key] = _WeightPlan(self, dev)
        return pl

    def _token(self, dev):
        """a scalar that requires grad: keeps the layer nodes in the autograd graph (their parameters are not node inputs)"""
        t = self._tokens.get(dev)
        if t is None:
            t = self._tokens[dev] = torch.zeros((), device=dev, requires_grad=True)
        return t

    def begin_iteration(self, device):
        """Packs the weights and biases of every layer from the current parameter values (two launches) and -- when a backward
        pass may follow -- clears the gradient scratch (one launch).  Called by every forward."""
        self._plan_for(device).refresh(self, for_backward=torch.is_grad_enabled())

    # ---- layers --------------------------------------------------------------------------------------------------
    def _run(self, name, x, relu_in):
        return _ConvLikeFn.apply(x, self._token(x.device), self, name, relu_in)

    def _code(self):
        return L.SV_BF16 if self.compute_dtype == "bf16" else L.SV_F32

    def _heads(self, x, x2=None):
        """image -> raw head outputs [B,1,1,pad16(2*cont + disc)] = [mean | logvar | logits | pad] (svhn_vae.py:137-160).
        x2: a second batch that follows the first (the two forwards of an iteration as one pass: both go straight into their
        halves of the NHWC tensor -- no concatenated copy of the images)"""
        B1 = x.shape[0]
        B = B1 + (x2.shape[0] if x2 is not None else 0)
        x16 = torch.empty(B, 32, 32, 16, dtype=self._tdt, device=x.device)                       # NHWC16 (layout edge)
        L.call("sv_nchw_to_nhwc", self._code(), _vp(x.contiguous().float()), B1, x.shape[1], 32, 32, 16, _vp(x16), _st())
        if x2 is not None:
            L.call("sv_nchw_to_nhwc", self._code(), _vp(x2.contiguous().float()), B - B1, x2.shape[1], 32, 32, 16, _vp(x16[B1:]), _st())
        y = self._run("c1", x16, False)
        y = self._run("c2", y, True)
        y = self._run("c3", y, True)
        w3 = self.reshape[0]
        # (Linear over the flattened map and the three heads as ONE GEMM: the row / column permutations live in the weight plan)
        hid = self._run("f1", y.view(B, 1, 1, w3 * 16), True)                                     # [B,1,1,hidden] raw
        return self._run("heads", hid, True)

    def _latent(self, o, label):
        """head outputs -> (mean, logvar, alpha, Gumbel-softmax sample, decoder input, latent_sample): one launch
        (sv_smooth_latent_fwd).  Host RNG order of the reference: randn for z (svhn_vae.py:176), then rand for the Gumbel
        noise (:192) -- drawn in training mode for labelled data too (:205-207)."""
        B, dev = o.shape[0], o.device
        eps = u = None
        if self.training:
            eps = torch.randn((B, self.latent_cont_dim), device=dev).to(dev).float().contiguous()
            u = torch.rand((B, self.latent_disc_dim), device=dev).to(dev).float().contiguous()
        return _LatentFn.apply(o, eps, u, label.long().contiguous() if label is not None else None, self.temperature,
                               self.training, self.latent_cont_dim, self.latent_disc_dim, self._L["g1"].Cin, self._code())

    def encode(self, x):
        """-> (mean, logvar, alpha) (svhn_vae.py:137-166)"""
        self.begin_iteration(x.device)
        o = self._heads(x).view(x.shape[0], -1).float()
        c = self.latent_cont_dim
        return o[:, :c], o[:, c:2 * c], F.softmax(o[:, 2 * c:2 * c + self.latent_disc_dim], dim=1)

    def sample_normal(self, mean, logvar):
        if self.training:
            return mean + torch.exp(0.5 * logvar) * torch.randn(mean.shape, device=mean.device).to(mean.device)
        return mean

    def sample_gumbel_softmax(self, alpha):
        if self.training:
            unif = torch.rand(alpha.shape, device=alpha.device).to(alpha.device)
            gumbel = -torch.log(-torch.log(unif + EPS) + EPS)
            return F.softmax((torch.log(alpha + EPS) + gumbel) / self.temperature, dim=1)
        return F.one_hot(alpha.argmax(1), alpha.shape[1]).float()

    def _decode_padded(self, z):
        """z: [B,1,1,pad16(latent)] in the compute dtype -> reconstruction NCHW fp32 after Tanh"""
        B = z.shape[0]
        f = self._run("g1", z, False)
        w3 = self.reshape[0]
        f = self._run("g2", f, True).view(B, 4, 4, w3)                                                    # NHWC
        f = self._run("t1", f, True)
        f = self._run("t2", f, True)
        f = self._run("t3", f, True)                                                                      # [B,32,32,16]
        return _TanhNchwFn.apply(f, self.img_size[0], self._code())

    def decode(self, latent_sample):
        B = latent_sample.shape[0]
        lat = self._L["g1"].Cin
        z = F.pad(latent_sample, (0, lat - latent_sample.shape[1])).to(self._tdt).view(B, 1, 1, lat)
        self.begin_iteration(z.device)
        return self._decode_padded(z)

    def forward(self, x, label=None):
        self.begin_iteration(x.device)
        mean, logvar, alpha, gs, latent, latent_sample = self._latent(self._heads(x), label)
        latent_dist = {"cont": [mean, logvar], "disc": [alpha]}
        disc_sample = [gs] if label is not None else []      # drawn (and unused) for labelled data, as svhn_vae.py:205-207
        return self._decode_padded(latent), latent_dist, latent_sample, disc_sample


svhn_VAE = mnist_VAE = SmoothVAE


class SmoothELBOLoss:
    """Trainer._loss_function (main_smooth_ELBO_svhn.py:228-310): num_pixels * MSE + gamma_c |C_c(t) - KL_c| +
    gamma_d |C_d(t) - KL_d| + alpha * BCE(q(y|x), one-hot label); capacities grow linearly with `num_steps`."""

    def __init__(self, cont_capacity=(0.0, 50, 50000, 1), disc_capacity=(0.0, 50, 50000, 1), alpha=1500.0):
        self.cont_capacity, self.disc_capacity, self.alpha, self.num_steps = cont_capacity, disc_capacity, alpha, 0
        self.steps_dev = None       # optional device scalar used instead of num_steps (hipGraph replay)

    def _schedule(self):
        cc, dc = self.cont_capacity, self.disc_capacity
        return L.SvSmoothSchedule(float(cc[0]), float(cc[1]), float(cc[2]), float(cc[3]), float(dc[0]), float(dc[1]), float(dc[2]),
                                  float(dc[3]), float(self.alpha), float(self.num_steps))

    def __call__(self, data, recon_data, latent_dist, label=None):
        mean, logvar = latent_dist["cont"]
        loss, terms = _SmoothLossFn.apply(data, recon_data, mean, logvar, latent_dist["disc"][0], label, self._schedule(),
                                          self.steps_dev)
        return loss, (terms[5], terms[6], terms[7], terms[8])


def both_forwards(model, loss_fn, unlabeled_data, labeled_data, label):
    """The two forwards of one trainer iteration (main_smooth_ELBO_svhn.py:157-168: model(unlabeled), model(labeled, label))
    as ONE pass over the concatenated batch: these models have no BatchNorm -- no operation couples two samples -- so the
    encoder, the head GEMM and the decoder see 2 B rows once instead of B rows twice (half the launches of a launch-bound
    iteration, one weight re-pack per layer); only the samplers (the labelled half takes the one-hot label) and the two loss
    evaluations stay per half.  Host RNG order of the reference: randn, rand of the unlabelled forward, then of the labelled
    one.  Returns (loss_u, split_u, loss_l, split_l, rec_u, dist_u, rec_l, dist_l)."""
    Bu = unlabeled_data.shape[0]
    model.begin_iteration(unlabeled_data.device)
    o = model._heads(unlabeled_data, labeled_data)
    mean_u, logvar_u, alpha_u, _, lat_u, _ = model._latent(o[:Bu], None)
    mean_l, logvar_l, alpha_l, _, lat_l, _ = model._latent(o[Bu:], label)
    rec = model._decode_padded(torch.cat([lat_u, lat_l]))
    rec_u, rec_l = rec[:Bu], rec[Bu:]
    dist_u = {"cont": [mean_u, logvar_u], "disc": [alpha_u]}
    dist_l = {"cont": [mean_l, logvar_l], "disc": [alpha_l]}
    loss_u, split_u = loss_fn(unlabeled_data, rec_u, dist_u)
    loss_l, split_l = loss_fn(labeled_data, rec_l, dist_l, label)
    return loss_u, split_u, loss_l, split_l, rec_u, dist_u, rec_l, dist_l


def smooth_train_step(model, loss_fn, optimizer, unlabeled_data, labeled_data, label, return_outputs=False,
                      distributed=False, batched=True):
    """One iteration of Trainer._train_epoch (main_smooth_ELBO_svhn.py:152-176).  distributed=True: one process per
    GPU, every rank on its shard of both batches, ONE all-reduce of the bucketed gradients before the optimizer step.
    batched (default): both forwards as one pass over the concatenated batch (both_forwards); False: two model(...) calls."""
    loss_fn.num_steps += 1
    if optimizer is not None:
        optimizer.zero_grad()
    if batched:
        loss_u, split_u, loss_l, split_l, rec_u, dist_u, rec_l, dist_l = both_forwards(model, loss_fn, unlabeled_data,
                                                                                       labeled_data, label)
    else:
        rec_u, dist_u, _, _ = model(unlabeled_data)
        loss_u, split_u = loss_fn(unlabeled_data, rec_u, dist_u)
        rec_l, dist_l, _, _ = model(labeled_data, label)
        loss_l, split_l = loss_fn(labeled_data, rec_l, dist_l, label)
    loss = loss_u + loss_l
    loss.backward()
    scale = None
    if distributed:
        from . import dp
        if hasattr(optimizer, "flat_grad"):        # FlatAdam: the gradients ARE one flat buffer -> one all-reduce, 1/world in the kernel
            scale = dp.all_reduce_gradients(optimizer.flat_grad)
        else:
            dp.all_reduce_module_gradients(model)
    if optimizer is not None:
        if scale is not None:
            optimizer.step(grad_scale=scale)
        else:
            optimizer.step()
    if not return_outputs:
        return loss.detach()
    out = dict(loss=loss, loss_u=loss_u, loss_l=loss_l, recon_u=split_u[0], cont_u=split_u[1], disc_u=split_u[2],
               recon_l=split_l[0], cont_l=split_l[1], disc_l=split_l[2], cls_l=split_l[3], rec_u=rec_u,
               mean_u=dist_u["cont"][0], logvar_u=dist_u["cont"][1], alpha_u=dist_u["disc"][0], rec_l=rec_l,
               mean_l=dist_l["cont"][0], logvar_l=dist_l["cont"][1], alpha_l=dist_l["disc"][0])
    return {k: v.detach() for k, v in out.items()}


class GraphedSmoothStep:
    """(Do not capture while an RCCL (`nccl`) process group has work in flight: its watchdog thread polls events, which is an
    error during another thread's stream capture on this stack -- bench.py issues eagerly in that case.)
    One trainer iteration (both forwards, the loss, backward, Adam) captured once into a hipGraph and replayed: the
    eager iteration is ~120 kernel launches plus torch glue and entirely host-bound (7 ms at batch 1024; SURVEY.md §8d
    config 5).  The optimizer must be graph-capturable (torch.optim.Adam(..., capturable=True)); the capacity schedule
    follows a device-side step counter; noise comes from torch's graph-safe Philox generator."""

    def __init__(self, model, loss_fn, optimizer, unlabeled_data, labeled_data, label, warmup=3, distributed=False):
        # distributed=True (one process per GPU): the graph holds both forwards, the loss and the backward; the ONE gradient
        # all-reduce (FlatAdam's flat buffer, or the bucketed module gradients) and the optimizer step follow eagerly, so
        # the collective is an ordinary RCCL call
        self.model, self.loss_fn, self.opt, self.distributed = model, loss_fn, optimizer, distributed
        self.u, self.l, self.y = unlabeled_data.clone(), labeled_data.clone(), label.clone()
        loss_fn.steps_dev = torch.full((), float(loss_fn.num_steps), device=self.u.device)
        self.stream = torch.cuda.Stream()
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            for _ in range(warmup):
                self._body()
                if distributed:
                    self._update()
        cur.wait_stream(self.stream)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.stream, capture_error_mode="thread_local"):
            self.loss = self._body()

    def _body(self):
        lf = self.loss_fn
        lf.steps_dev += 1
        self.opt.zero_grad(set_to_none=False)
        loss_u, _, loss_l = both_forwards(self.model, lf, self.u, self.l, self.y)[:3]
        loss = loss_u + loss_l
        loss.backward()
        if not self.distributed:
            self.opt.step()
        return loss.detach()

    def _update(self):
        from . import dp
        if hasattr(self.opt, "flat_grad"):
            self.opt.step(grad_scale=dp.all_reduce_gradients(self.opt.flat_grad))
        else:
            dp.all_reduce_module_gradients(self.model)
            self.opt.step()

    def __call__(self, unlabeled_data=None, labeled_data=None, label=None):
        if unlabeled_data is not None:
            self.u.copy_(unlabeled_data)
            self.l.copy_(labeled_data)
            self.y.copy_(label)
        self.graph.replay()
        if self.distributed:
            self._update()
        self.loss_fn.num_steps += 1
        return self.loss
