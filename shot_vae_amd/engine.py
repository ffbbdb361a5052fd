"""Host-side plan of the SHOT-VAE network on MI355X: flat fp32 parameter / gradient buffers, packed
weight shadows, and the launch sequences of one forward and one backward over the C ABI
(include/shotvae_hip.h).  PyTorch is used for device memory and streams only.

Reference being replaced: shot_vae_model/wideresnet.py:8-114 (encoder), shot_vae_model/decoder.py:4-69,
shot_vae_model/vae.py:10-151 (heads, sampler, assembly) and autograd's backward of those.

Data layout in HBM
  activations     NHWC, element type = compute dtype (bf16 throughput mode / fp32 parity mode)
  master weights  fp32, conv-like layers as [Cout][ky*k+kx][Cin] (so torch's OIHW / IOHW tensors are
                  plain strided views of it), everything in ONE flat buffer -> one SGD launch, one
                  RCCL all-reduce of the matching flat gradient buffer
  packed weights  compute dtype, per layer one forward pack and one data-gradient pack
                  ([n][tap][c], k contiguous = MFMA operand order), refreshed after each optimizer step
"""
import ctypes as C
import os
import math
import re

import torch

from . import _lib as L
from . import geometry as G
_GEOM = G          # (functions below use G for the number of groups)

LEAKY_SLOPE = 0.01     # nn.LeakyReLU default, wideresnet.py:28
BN_EPS = 1e-5
BN_MOMENTUM = 0.1
CPAD = 16              # channel padding granule of the MFMA kernels


def _align(n, a=64):
    return (n + a - 1) // a * a


def _pad16(c):
    return (c + CPAD - 1) // CPAD * CPAD


def parse_wideresnet(name):
    """'wideresnet-D-W' (wideresnet.py:102-114).  Errors mirror the reference's."""
    depth, width = re.findall(r"\d+", name)           # ValueError if not exactly two integers
    depth, width = int(depth), int(width)
    assert (depth - 4) % 6 == 0, "depth should be 6n+4"
    return depth, width, (depth - 4) // 6


class ConvSpec:
    """One conv-like layer (Conv2d / ConvTranspose2d / the k=1 ConvTranspose GEMM)."""

    def __init__(self, key, kind, k, stride, pad, cin, n, hin, cin_real=None, n_real=None):
        self.key, self.kind, self.k, self.stride, self.pad = key, kind, k, stride, pad
        self.Cin, self.N, self.Hin = cin, n, hin
        self.cin_real, self.n_real = cin_real or cin, n_real or n
        self.T = k * k
        self.master_off = None
        self.fwd_off = self.dgrad_off = None
        self._g = {}

    @property
    def Hout(self):
        if self.kind == "conv":
            return (self.Hin + 2 * self.pad - self.k) // self.stride + 1
        return self.Hin * self.stride

    def geom_fwd(self, B):
        g = self._g.get(("f", B))
        if g is None:
            if self.kind == "conv":
                g = G.conv_like(B, self.Hin, self.Hin, self.Cin, self.N, self.k, self.stride, self.pad)
            else:
                g = G.convT_like(B, self.Hin, self.Hin, self.Cin, self.N, self.k, self.stride, self.pad)
            self._g[("f", B)] = g
        return g

    def geom_dgrad(self, B):
        """input = gradient w.r.t. this layer's output, output = gradient w.r.t. its input."""
        g = self._g.get(("d", B))
        if g is None:
            ho = self.Hout
            if self.kind == "conv":
                g = G.convT_like(B, ho, ho, self.N, self.Cin, self.k, self.stride, self.pad)
            else:
                g = G.conv_like(B, ho, ho, self.N, self.Cin, self.k, self.stride, self.pad)
            self._g[("d", B)] = g
        return g

    def torch_view(self, master):
        """The reference-shaped (OIHW for Conv2d, IOHW for ConvTranspose2d) strided view."""
        m = master[self.master_off: self.master_off + self.N * self.T * self.Cin].view(self.N, self.k, self.k, self.Cin)
        if self.kind == "conv":
            return m.permute(0, 3, 1, 2)[: self.n_real, : self.cin_real]
        return m.permute(3, 0, 1, 2)[: self.cin_real, : self.n_real]


class BNSpec:
    def __init__(self, key, c, slope):
        self.key, self.C, self.slope = key, c, slope
        self.gamma_off = self.beta_off = self.rm_off = self.rv_off = None
        self.buf_off = None      # offset into the per-forward (scale, shift, mean, rstd) scratch
        self.index = None


class Plan:
    """Architecture + memory layout (device independent)."""

    def __init__(self, encoder_name, in_ch=3, img=32, ldc=128, K=10):
        if "wideresnet" not in encoder_name:
            raise NotImplementedError("{} not implemented".format(encoder_name))
        if img != 32 or in_ch > CPAD:
            raise NotImplementedError("the MI355X path covers 32x32 inputs with <= 16 channels "
                                      "(BASELINE.json configs); got img=%s ch=%s" % (img, in_ch))
        self.name, self.in_ch, self.img, self.ldc, self.K = encoder_name, in_ch, img, ldc, K
        _, width, n_units = parse_wideresnet(encoder_name)
        self.widths = [int(16 * width), int(32 * width), int(64 * width)]
        self.cfeat = self.widths[-1]
        self.Lpad = _pad16(ldc + K)
        self.NH = 2 * ldc + K
        self.convs, self.bns, self.params = [], [], []     # params: (key, offset, shape-or-spec)
        self.units = []
        off = [0]

        def alloc(n):
            o = off[0]
            off[0] = _align(o + n)
            return o

        def add_conv(spec):
            spec.master_off = alloc(spec.N * spec.T * spec.Cin)
            self.convs.append(spec)
            self.params.append((spec.key + ".weight" if not spec.key.endswith("weight") else spec.key, spec))
            return spec

        def add_vec(key, n):
            o = alloc(n)
            self.params.append((key, (o, n)))
            return o

        def add_bn(key, c, slope):
            b = BNSpec(key, c, slope)
            b.gamma_off = add_vec(key + ".weight", c)
            b.beta_off = add_vec(key + ".bias", c)
            b.index = len(self.bns)
            self.bns.append(b)
            return b

        enc = "feature_extractor.encoder."
        self.stem = add_conv(ConvSpec(enc + "pre_process.conv0", "conv", 3, 1, 1, CPAD, 16, img,
                                      cin_real=in_ch))
        self.stem_bias_off = add_vec(enc + "pre_process.conv0.bias", 16)
        cin, h = 16, img
        for s, w in enumerate(self.widths):
            for u in range(n_units):
                stride = 2 if (s > 0 and u == 0) else 1
                ci = cin if u == 0 else w
                p = enc + "wideblock%d.wide_block.wideunit%d." % (s + 1, u + 1)
                unit = dict(cin=ci, cout=w, stride=stride, hin=h)
                unit["bn1"] = add_bn(p + "f_block.norm1", ci, LEAKY_SLOPE)
                unit["conv1"] = add_conv(ConvSpec(p + "f_block.conv1", "conv", 3, stride, 1, ci, w, h))
                unit["bn2"] = add_bn(p + "f_block.norm2", w, LEAKY_SLOPE)
                unit["conv2"] = add_conv(ConvSpec(p + "f_block.conv2", "conv", 3, 1, 1, w, w, h // stride))
                if ci != w or stride != 1:
                    unit["bni"] = add_bn(p + "i_block.norm", ci, LEAKY_SLOPE)
                    unit["convi"] = add_conv(ConvSpec(p + "i_block.conv", "conv", 1, stride, 0, ci, w, h))
                h //= stride
                self.units.append(unit)
            cin = w
        self.hfeat = h
        self.bn_t = add_bn(enc + "transition.norm", self.cfeat, LEAKY_SLOPE)
        # the three heads share one [2*ldc+K][C] matrix (vae.py:113-129)
        self.head_w_off = alloc(self.NH * self.cfeat)
        self.head_b_off = alloc(self.NH)
        c = self.cfeat
        for nm, r0, r1 in (("continuous_inference.mean.fc", 0, ldc),
                           ("continuous_inference.log_sigma.fc", ldc, 2 * ldc),
                           ("disc_latent_inference.fc", 2 * ldc, self.NH)):
            self.params.append((nm + ".weight", (self.head_w_off + r0 * c, (r1 - r0, c))))
            self.params.append((nm + ".bias", (self.head_b_off + r0, r1 - r0)))
        # decoder (decoder.py:12-58): ConvT(latent,1024,k=img/32) then 5x ConvT(4,2,1)
        dec = "feature_reconstructor.decoder."
        chans = [1024, 512, 256, 128, 64]
        self.dec_convs, self.dec_bns = [], []
        self.dec_convs.append(add_conv(ConvSpec(dec + "0", "convT", 1, 1, 0, self.Lpad, chans[0], 1,
                                                cin_real=ldc + K)))
        hh = 1
        for i in range(5):
            self.dec_bns.append(add_bn(dec + "%d" % (3 * i + 1), chans[i], 0.0))
            nout = chans[i + 1] if i < 4 else _pad16(in_ch)
            self.dec_convs.append(add_conv(ConvSpec(dec + "%d" % (3 * i + 3), "convT", 4, 2, 1, chans[i], nout, hh,
                                                    n_real=(chans[i + 1] if i < 4 else in_ch))))
            hh *= 2
        self.n_param = off[0]
        self.dec_off = self.dec_convs[0].master_off     # the decoder's parameters are the tail [dec_off, n_param) of the flat buffer
        # BN running statistics and per-forward scratch
        o = 0
        for b in self.bns:
            b.rm_off, b.rv_off = o, o + _align(b.C)
            o += 2 * _align(b.C)
        self.n_buf = o
        o = 0
        for b in self.bns:
            b.buf_off = o
            o += 4 * _align(b.C)
        self.n_bnbuf = o
        # packed weights
        o = 0
        for cv in self.convs:
            cv.fwd_off = o
            o = _align(o + G.packed_size(cv.geom_fwd(1)))
            cv.dgrad_off = o
            o = _align(o + G.packed_size(cv.geom_dgrad(1)))
        self.n_pack = o

    # -- reference state_dict keys (data_parallel=False naming) in the reference's order ---------
    def state_items(self):
        """yield (key, kind, payload): kind in {'conv','vec','mat','rm','rv','nbt'}."""
        bn_by_key = {b.key: b for b in self.bns}
        out = []
        for key, payload in self.params:
            if isinstance(payload, ConvSpec):
                out.append((key, "conv", payload))
            elif isinstance(payload[1], tuple):
                out.append((key, "mat", payload))
            else:
                out.append((key, "vec", payload))
            if key.endswith(".bias") and key[:-5] in bn_by_key:
                b = bn_by_key[key[:-5]]
                out.append((b.key + ".running_mean", "rm", b))
                out.append((b.key + ".running_var", "rv", b))
                out.append((b.key + ".num_batches_tracked", "nbt", b))
        return out


_SERIALISED = None


def _dispatch_serialised():
    """True when the environment serialises kernel dispatch across streams (rocprofv3 --pmc counter collection,
    AMD_SERIALIZE_KERNEL, HIP_LAUNCH_BLOCKING): a kernel that WAITS for another stream's kernel to start (the device-side fork,
    sv_stream_wait_flag) would then sit in front of it until its time-out -- those runs fork with events."""
    global _SERIALISED
    if _SERIALISED is None:
        e = os.environ
        _SERIALISED = (e.get("ROCPROF_COUNTER_COLLECTION", "0") not in ("", "0") or bool(e.get("ROCPROF_COUNTERS")) or
                       e.get("AMD_SERIALIZE_KERNEL", "0") not in ("", "0") or e.get("HIP_LAUNCH_BLOCKING", "0") not in ("", "0"))
    return _SERIALISED


def _vp(x):
    return C.c_void_p(x)


def _replicas(rows):
    """Copies of a per-channel accumulator for a tensor with `rows` pixels: the gather-GEMM runs one
    block per 128 rows and each block issues one atomic per channel, so keep ~<=128 adders per address."""
    tiles, r = rows // 128, 1
    while r < 32 and tiles // (2 * r) >= 16:
        r *= 2
    return r


class FwdCtx:
    """Everything one forward leaves behind for its backward."""
    pass


class Engine:
    """Device state + launch sequences.  All methods enqueue on torch's current stream."""

    def __init__(self, plan, compute_dtype="bf16"):
        self.plan = plan
        self.set_compute_dtype(compute_dtype)
        p = plan
        self.param = torch.zeros(p.n_param, dtype=torch.float32)
        self.grad = torch.zeros(p.n_param, dtype=torch.float32)
        self.mom = None
        self.bufs = torch.zeros(p.n_buf, dtype=torch.float32)
        self.nbt = torch.zeros(len(p.bns), dtype=torch.int64)
        self.packs = None
        self._pack_key = None
        self._manual_epoch = 0
        self.use_tr = 1
        self.prof_tags = None
        self.prof_cost = {}
        self.defer_slot = None        # k: this forward's running-stat update is deferred into slot k (apply_pending)
        self._pending = {}
        self._run_tables = {}
        self._repack_tables = {}
        # backward of a body convolution: its weight gradient (side stream) and its data gradient (main stream) read the
        # same dY and x and start together; with HALF the persistent-block budget each, one block of either kernel sits on
        # every CU, the two walk the same pixel ranges on the same XCDs and the follower finds the operands in L2
        # (config 2: 9.26 -> 9.09 ms; 192 / 128 blocks are slower).  0 = both kernels with the full budget, one after the other.
        # A pair budget above SV_OPT_PERSISTENT_BLOCKS would not be "half": it is clamped there at launch time.
        self.pair_blocks = 256
        # ... of the stride-2 unit's first convolution (its data gradient is a whole-CU kernel of tconv.hip: with a budget it leaves
        # CUs to the weight gradient beside it); 0 = no budget: the two effectively run one after the other.  Measured: neutral on the
        # boxes where the two streams overlap well anyway (6.75 either way), -0.10 ms on the others (6.96 -> 6.86)
        self.pair_blocks_strided = 256
        self._bn_layouts = {}
        self.version_probe = None     # callable: summed version counters of the nn.Parameters (set by the module)
        # one-shot callback fired by backward() as soon as every decoder gradient has been ISSUED (main + side stream): the
        # data-parallel step starts the all-reduce of the decoder's 88 % of the gradient bytes there (dp.DecoderFirstAllReduce)
        self.bucket_hook = None
        # bench.py's per-launch timing pass issues the step on ONE stream (HIP events bracket every launch); with this set it
        # still gives the paired launches the block budgets they run with in the timed, two-stream step
        self.prof_paired = False
        self._side_keep = {}          # main stream -> operands of the weight gradients in flight on its side stream
        self._pending_wgrads = []     # weight gradients waiting for the next fork (fork_every)
        for b in p.bns:
            self.bufs[b.rv_off: b.rv_off + b.C] = 1.0

    def set_compute_dtype(self, compute_dtype):
        assert compute_dtype in ("bf16", "fp32"), compute_dtype
        self.compute_dtype = compute_dtype
        self.code = L.SV_BF16 if compute_dtype == "bf16" else L.SV_F32
        self.tdtype = torch.bfloat16 if compute_dtype == "bf16" else torch.float32
        self.packs = None
        self._pack_key = None

    # ------------------------------------------------------------------------------- storage
    def to(self, fn):
        """Apply a torch `_apply` function (device move) to the flat buffers."""
        new = fn(self.param)
        if new.dtype != torch.float32:
            raise TypeError("shot_vae_amd keeps fp32 master weights; choose the compute dtype with "
                            "compute_dtype='bf16'|'fp32' instead of .half()/.bfloat16()")
        self.param, self.grad = new, fn(self.grad)
        self.bufs, self.nbt = fn(self.bufs), fn(self.nbt)
        if self.mom is not None:
            self.mom = fn(self.mom)
        self.packs, self._pack_key = None, None

    def init_default(self, seed=None):
        """PyTorch default initialisation (the reference defines no custom init):
        conv / convT / linear weights and biases U(+-1/sqrt(fan_in)), BN gamma 1, beta 0."""
        g = torch.Generator()
        if seed is not None:
            g.manual_seed(seed)
        else:
            g.manual_seed(int(torch.randint(0, 2 ** 31 - 1, (1,))))
        p = self.plan
        self.param.zero_()
        for key, kind, payload in p.state_items():
            if kind == "conv":
                v = payload.torch_view(self.param)
                fan_in = v.shape[1] * v.shape[2] * v.shape[3]
                v.copy_((torch.rand(v.shape, generator=g) * 2 - 1) / math.sqrt(fan_in))
            elif kind == "mat":
                off, (r, c) = payload
                self.param[off: off + r * c] = ((torch.rand(r * c, generator=g) * 2 - 1) / math.sqrt(c)).to(self.param.device)
            elif kind == "vec":
                off, n = payload
                if key.endswith("conv0.bias"):
                    fan = p.in_ch * 9
                    self.param[off: off + n] = ((torch.rand(n, generator=g) * 2 - 1) / math.sqrt(fan)).to(self.param.device)
                elif key.endswith("fc.bias"):
                    self.param[off: off + n] = ((torch.rand(n, generator=g) * 2 - 1) / math.sqrt(p.cfeat)).to(self.param.device)
                elif key.endswith(".weight"):
                    self.param[off: off + n] = 1.0
        self.mark_dirty()

    def mark_dirty(self):
        self._manual_epoch += 1

    # ------------------------------------------------------------------------------- helpers
    def _stream(self):
        return _vp(torch.cuda.current_stream().cuda_stream)

    def _require_gpu(self, t):
        if not t.is_cuda or not self.param.is_cuda:
            raise L.ShotVaeHipError("shot_vae_amd runs on an MI355X only (HIP kernels, no CPU fallback): "
                                    "move the model and its inputs to cuda first")
        L.lib()

    def _tag(self, name, g=None, extra_out_reads=0, wgrad=False, groups=1, writes_out=True, extra_in=0):
        """File the next launch under `name` for sv_prof_collect and remember its algorithmic cost:
        bytes = input + output (+ fused residual / raw-tensor reads) + weights, flops = 2*M*N*K."""
        if self.prof_tags is None:
            return
        t = self.prof_tags.setdefault(name, len(self.prof_tags))
        L.lib().sv_prof_tag(t)
        if g is not None:
            es = self.packs.element_size()
            taps = sum(g.phase[p].ntap for p in range(g.nphase))
            rows = groups * g.B * g.Hq * g.Wq
            n_in = groups * g.B * g.Hin * g.Win * g.Cin
            n_out = groups * g.B * g.Hout * g.Wout * g.N
            n_w = taps * g.Cin * g.N
            flops = 2.0 * rows * taps * g.Cin * g.N
            if wgrad:
                nbytes = es * (n_in + n_out * (1 + extra_out_reads)) + 4 * n_w
            else:
                nbytes = es * (n_in * (1 + extra_in) + n_out * (int(writes_out) + extra_out_reads) + n_w)
            self._cost(name, nbytes, flops)

    def _cost(self, name, nbytes, flops=0.0):
        """algorithmic bytes / flops of one launch filed under `name` (summed over launches; bench.py divides)"""
        if self.prof_tags is None:
            return
        b, f, n = self.prof_cost.get(name, (0.0, 0.0, 0))
        self.prof_cost[name] = (b + float(nbytes), f + float(flops), n + 1)

    def ensure_packs(self):
        ver = self.param._version + (self.version_probe() if self.version_probe is not None else 0)
        key = (ver, self._manual_epoch, self.param.data_ptr(), self.compute_dtype)
        if self.packs is not None and key == self._pack_key:
            return
        p = self.plan
        if self.packs is None or self.packs.dtype != self.tdtype or self.packs.device != self.param.device:
            self.packs = torch.zeros(p.n_pack, dtype=self.tdtype, device=self.param.device)
        # every (layer, direction, phase) pack in ONE launch (sv_repack_batch); the job table is plan-static
        tab = self._repack_tables.get(self.param.device)
        if tab is None:
            jobs, b0 = [], 0
            for cv in p.convs:
                for tr, g, off in ((0, cv.geom_fwd(1), cv.fwd_off), (1, cv.geom_dgrad(1), cv.dgrad_off)):
                    for ph in range(g.nphase):
                        P = g.phase[ph]
                        if P.ntap == 0:
                            continue
                        j = L.SvRepackJob()
                        j.master_off, j.dst_off, j.size = cv.master_off, off + P.w_off, cv.N * cv.Cin * P.ntap
                        j.N, j.T_orig, j.C, j.transpose, j.ntap, j.block0 = cv.N, cv.T, cv.Cin, tr, P.ntap, b0
                        for t in range(L.MAX_TAPS):
                            j.torig[t] = P.torig[t]
                        b0 += (j.size + 1023) // 1024
                        jobs.append(j)
            arr = (L.SvRepackJob * len(jobs))(*jobs)
            raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.param.device)
            tab = self._repack_tables[self.param.device] = (raw, len(jobs), b0)
        raw, njobs, nblocks = tab
        L.call("sv_repack_batch", self.code, _vp(self.param.data_ptr()), _vp(raw.data_ptr()), njobs, nblocks,
               _vp(self.packs.data_ptr()), self._stream())
        self._pack_key = key

    def _igemm(self, g, x, w_ptr, out, pro=None, bias=None, residual=None, stats=None, ex=None, tag=None, groups=1,
               budget=0, sparse_out=False):
        a = L.SvIgemmArgs()
        a.groups = groups
        a.block_budget = budget
        a.sparse_out = int(bool(sparse_out))
        a.x, a.w, a.out = x.data_ptr(), w_ptr, out.data_ptr()
        if pro is not None:
            a.pro_scale, a.pro_shift, a.pro_slope = pro[0], pro[1], pro[2]
            if len(pro) > 3 and pro[3] is not None:
                # the BatchNorm in front of this layer is finalised BY the launch (sv_igemm_args::fold_*): the persistent
                # kernels derive the coefficients in every block, sv_igemm runs sv_bn_finalize first for the others
                (a.fold_stats, a.fold_replicas, a.fold_count, a.fold_gamma, a.fold_beta, a.fold_mean, a.fold_rstd) = pro[3]
                a.fold_eps = BN_EPS
        if bias is not None:
            a.bias = bias
        if residual is not None:
            a.residual = residual.data_ptr()
        a.replicas = 1
        # deterministic mode (SV_OPT_DETERMINISTIC): the accumulator needs one replica per wave of the launch's grid, which
        # only the dispatcher knows -- `stats` / the bsums slot of `ex` is then a callable(replicas) -> pointer that
        # allocates it once the grid has been queried
        if stats is not None:
            if callable(stats):
                a.stats = x.data_ptr()           # (placeholder for the query: the dispatch does not depend on it)
                a.replicas = L.det_replicas(g, self.code, a)
                a.stats = stats(a.replicas)
            else:
                a.stats, a.replicas = stats
        if ex is not None:
            a.ex = ex[0].data_ptr()
            a.ex_scale, a.ex_shift, a.ex_mean, a.ex_rstd, a.ex_slope = ex[1:6]
            if callable(ex[6]):
                a.bsums = x.data_ptr()
                a.replicas = L.det_replicas(g, self.code, a)
                a.bsums = ex[6](a.replicas)
            else:
                a.bsums, a.replicas = ex[6:]
        if tag:
            self._tag(tag, g, (residual is not None) + (ex is not None), groups=groups)
        if self._start_signal is not None:       # (see _wgrad_async: this launch forks the side stream when it starts)
            (a.start_flag, a.start_value), self._start_signal = self._start_signal, None
        L.call("sv_igemm", C.byref(g), self.code, C.byref(a), self._stream())

    _ws_elems = 16 * 1024 * 1024       # 64 MiB of fp32 partial-slab workspace for sv_wgrad

    def _wg_ws(self):
        """wgrad partial-slab workspace, one per stream (two backwards may run concurrently)."""
        pool = getattr(self, "_ws", None)
        if pool is None:
            pool = self._ws = {}
        key = (torch.cuda.current_stream().cuda_stream, self.param.device)
        ws = pool.get(key)
        if ws is None:
            ws = pool[key] = torch.empty(self._ws_elems, dtype=torch.float32, device=self.param.device)
        return ws

    wgrad_side_stream = True      # weight gradients on a side stream (they are off the backward's critical path)
    sparse_shortcut_grad = True      # stride-2 1x1 shortcuts: data gradient written / read at the even positions only
    compact_shortcut_grad = True     # ... as a dense 1x1 product over the stride-2 grid, stored compactly (bf16)
    materialize_decoder_act = True   # BatchNorm + ReLU of the first decoder layers' inputs as a pass of its own (see forward)
    materialize_max_hin = 4          # ... for the layers whose input map is at most this large
    light_fork = True                # fork events without the system-scope fence (sv_stream_fork)
    fork_every = 1                   # weight gradients per side-stream fork
    # sv_bwd3x3 (ABI 7 / 8): data gradient + weight gradient of a 32 -> 32 (64 -> 64) channel body convolution in ONE launch that reads every
    # operand once.  1 = conv1 of the same-shape units in the two-tensor form (norm2's BatchNorm backward formed in the kernel's
    # load path: the sv_bn_bwd_apply pass between the unit's two data gradients disappears: 8 tensor passes -> 4);
    # 2 = also conv2 of the unit IN FRONT of such a unit in the residual form: the NEXT unit's norm1 backward + skip connection
    # (the sv_bn_bwd_apply pass at the unit boundary, 4 passes) formed in its load path and written once as a side output
    # (9 passes -> 6); 3 = every other 32-channel conv2 as well (5 passes -> 3: slower than the two-stream pair, an A/B switch).
    # bf16, not in deterministic mode.
    fused_bwd = 2
    fold_bn_bwd = True               # (ABI 8) sv_bwd3x3 derives the BatchNorm backward's coefficients itself: no sv_bn_bwd_affine launch
    fused_channels = (32, 64)        # body widths that take sv_bwd3x3 (64: bwd3x3g.hip, 16 x 16 maps; same-box A/B 6.19-6.23 -> 6.09-6.12 ms)
    # blocks of a fused-backward launch (one 512-thread block per CU: a block's eight waves fill the SIMDs' register files).  NOT 256:
    # a single CU that hosts anything else -- the side stream's spinning wait_flag_kernel, the tail of a slab reduction -- cannot take
    # a block, the 256th block then runs as a second round and the launch takes twice as long (152 -> 240-247 us in the step's
    # trace, round 6).  248 = 31 per XCD leaves every XCD one CU for the neighbours.
    fused_blocks = 248
    flag_fork = True                 # paired launches: the data gradient's start signal forks the side stream (no event)
    _fused_since_fork = False        # an sv_bwd3x3 launch has been issued since the last fork of the side stream
    event_fork_after_fused = True    # (A/B switch: bench.py --event-fork-after-fused 0)
    _start_signal = None
    _pending_wgrads = ()
    fold_bn = True                   # BatchNorm finalisation folded into the consuming sv_igemm launch (sv_igemm_args::fold_*)

    def _side(self):
        """side stream paired with the current stream"""
        pool = getattr(self, "_side_streams", None)
        if pool is None:
            pool = self._side_streams = {}
        cur = torch.cuda.current_stream()
        s = pool.get(cur.cuda_stream)
        if s is None:
            s = pool[cur.cuda_stream] = torch.cuda.Stream()
        return cur, s

    def _wgrad_async(self, g, x, pro, dy, dw_ptr, tag=None, groups=1, budget=0, then=None):
        """Enqueue the weight gradient behind everything issued so far, on the side stream: the dgrad -> BN-apply
        chain continues on the main stream without waiting for it (joined at the end of backward).
        `then`: a callable that issues the main-stream launch paired with this weight gradient (the layer's data gradient).
        It is issued FIRST, so that the main stream, the critical path, never waits for the host to finish the side stream's
        bookkeeping (tools/step_timeline.py showed ~8 us of idle main stream per pair), and it carries the fork: its first
        block stores a sequence number the side stream waits for (sv_igemm_args::start_flag, sv_stream_wait_flag) -- no event
        in the main stream's queue.  Without a `then` (or where dispatch is serialised: profilers) the fork is an event of the
        library's pool.  Either way the weight gradient depends only on what preceded the pair."""
        # (not under hipGraph capture: a captured step replays a hundred cross-stream edges slower than one stream --
        #  11.6 against 11.05 ms, measured -- and tensors freed during capture would need to outlive the side stream)
        if not self.wgrad_side_stream or self.prof_tags is not None or torch.cuda.is_current_stream_capturing():
            self._wgrad(g, x, pro, dy, dw_ptr, tag, groups, budget)
            return then() if then is not None else None
        cur, side = self._side()
        # `fork_every` weight gradients share one fork: every fork is a marker in the main stream's queue that costs it ~6 us of
        # idle time in front of the next kernel (tools/probes/step_list.py: a gap before every data gradient)
        self._pending_wgrads.append((g, x, pro, dy, dw_ptr, tag, groups, budget))
        if len(self._pending_wgrads) < self.fork_every:
            return then() if then is not None else None
        # (the first pair behind fused-backward launches forks by event: the host runs ahead, the side stream's wait_flag_kernel for
        #  this pair would start as soon as the side stream is idle and spin on a CU THROUGH the fused launches in between -- 825 us in
        #  the trace -- and a fused launch whose waves fill the SIMDs' register files cannot share a CU with it: the workgroups the
        #  dispatcher had dealt to that CU's shader engine waited for a second round, 137 -> 236 us per launch of the residual form)
        after_fused, self._fused_since_fork = self._fused_since_fork and self.event_fork_after_fused, False
        if self.flag_fork and not after_fused and not _dispatch_serialised() and then is not None and len(self._pending_wgrads) == 1:
            # device-side fork: the paired main-stream launch (`then`, an sv_igemm) announces its own START through a flag word
            # (sv_igemm_args::start_flag) -- everything this weight gradient depends on has completed by then -- and the side
            # stream waits for the flag: no event, no marker in the main stream's queue
            flag, value = C.c_void_p(), C.c_uint32()
            L.call("sv_stream_flag_next", _vp(cur.cuda_stream), C.byref(flag), C.byref(value))
            self._start_signal = (flag.value, value.value)
            try:
                out = then()
            finally:
                armed, self._start_signal = self._start_signal, None
            if armed is None:          # consumed by the launch
                L.call("sv_stream_wait_flag", _vp(side.cuda_stream), flag, value)
                return self._issue_pending(cur, side, out)
            # (`then` launched nothing that could carry the signal: the ordinary fork, behind it)
            L.call("sv_stream_fork", _vp(cur.cuda_stream), _vp(side.cuda_stream), int(self.light_fork))
            return self._issue_pending(cur, side, out)
        return self._flush_wgrads(cur, side, then)

    def _flush_wgrads(self, cur, side, then=None):
        """Fork the side stream off the main stream here and issue the pending weight gradients on it.  The fork is an event
        of the library's pool WITHOUT the system-scope fence of an ordinary event (sv_stream_fork): both streams are on this
        device."""
        L.call("sv_stream_fork", _vp(cur.cuda_stream), _vp(side.cuda_stream), int(self.light_fork))
        out = then() if then is not None else None
        return self._issue_pending(cur, side, out)

    def _issue_pending(self, cur, side, out=None):
        with torch.cuda.stream(side):
            for (g, x, pro, dy, dw_ptr, tag, groups, budget) in self._pending_wgrads:
                self._wgrad(g, x, pro, dy, dw_ptr, tag, groups, budget)
        # the operands stay referenced until the streams are joined at the end of backward (no record_stream bookkeeping
        # per tensor: two allocator calls per weight gradient on the host's critical path)
        keep = self._side_keep.setdefault(cur.cuda_stream, [])
        for w in self._pending_wgrads:
            keep.append((w[1], w[3]))
        self._pending_wgrads = []
        return out

    _flag_forks_verified = False     # the first backward that forked by flag has been checked with a synchronisation

    def _join_side(self):
        if self.wgrad_side_stream and self.prof_tags is None and not torch.cuda.is_current_stream_capturing():
            cur, side = self._side()
            if self._pending_wgrads:
                self._flush_wgrads(cur, side)
            cur.wait_stream(side)
            try:
                if self.flag_fork and not _dispatch_serialised():
                    # fail closed (sv_stream_wait_flag): the sticky time-out counter is host-mapped, the check is a memory read.  The
                    # environment sniffing of _dispatch_serialised() cannot know every serialising tool, so the FIRST flag-forked
                    # backward of an engine is verified with one synchronisation.  A wait that gave up THERE is recoverable: nothing
                    # has consumed this backward's gradients yet (the optimizer step and the all-reduce come after it), so the engine
                    # switches to event forks for good, clears the counter and reports THIS step invalid -- a caller that drops it
                    # (zero_grad, next batch) continues on valid steps.
                    if not self._flag_forks_verified:
                        torch.cuda.synchronize()
                        Engine._flag_forks_verified = True
                        n = L.lib().sv_flag_timeouts()
                        if n:
                            Engine.flag_fork = False
                            self.flag_fork = False
                            L.call("sv_flag_timeouts_reset")
                            raise L.ShotVaeHipError(
                                "%d side-stream wait(s) for a data gradient's start signal timed out in the first backward of this "
                                "engine (kernel dispatch is serialised: profiler, debugger): the gradients of THIS backward are invalid "
                                "-- drop the step (zero_grad) -- and nothing has consumed them yet.  The engine now forks by events "
                                "(Engine.flag_fork = False); the following steps are valid." % n)
                    # a LATER time-out is seen by this memory read only once the waiting kernel has run -- possibly a step late, after
                    # sv_sgd / the all-reduce applied the corrupted gradients: that one is fatal (check_flag_timeouts says so)
                    L.check_flag_timeouts("Engine.backward")
            finally:
                # the side stream's operands may be released now (also when the check raises): the main stream, on which the
                # allocator will hand their memory out again, is ordered behind everything the side stream did
                self._side_keep.pop(cur.cuda_stream, None)

    def _wgrad(self, g, x, pro, dy, dw_ptr, tag=None, groups=1, budget=0):
        if tag:
            self._tag(tag, g, wgrad=True, groups=groups)
        a = L.SvWgradArgs()
        a.x, a.dy, a.dw = x.data_ptr(), dy.data_ptr(), dw_ptr
        if pro is not None:
            a.pro_scale, a.pro_shift, a.pro_slope = pro[0], pro[1], pro[2]
        a.splits, a.use_tr, a.ws, a.ws_elems = 0, self.use_tr, self._wg_ws().data_ptr(), self._ws_elems
        a.groups, a.block_budget = groups, budget          # the budget is an argument of THIS launch, not process state
        L.call("sv_wgrad_ex", C.byref(g), self.code, C.byref(a), self._stream())

    def _bwd3x3(self, cv, B, dy, lin2, x, bn_ptrs, slope, out, bsums, replicas, tag, groups, budget=0, res=None):
        """sv_bwd3x3: data gradient (activation-backward epilogue of the BatchNorm in front: bn_ptrs = scale, shift, mean, rstd) and
        weight gradient of the stride-1 3x3 convolution `cv` in one launch; lin2 = (dy2 tensor, scale, scale2, shift pointers);
        res = (dy3 tensor, dy_out tensor): the residual form"""
        g = cv.geom_dgrad(B)
        if tag:
            # algorithmic cost: dy [+ dy2] and x read once, g written; both products' flops
            n = groups * g.B * g.Hin * g.Win * g.Cin
            es = self.packs.element_size()
            if self.prof_tags is not None:
                L.lib().sv_prof_tag(self.prof_tags.setdefault(tag, len(self.prof_tags)))
            self._cost(tag, es * n * (3 + (lin2 is not None) + 2 * (res is not None)) + 4 * 9 * g.Cin * g.N, 2 * 2.0 * n * 9 * g.N)
        a = L.SvBwd3x3Args()
        a.dy, a.x, a.out = dy.data_ptr(), x.data_ptr(), out.data_ptr()
        if lin2 is not None and len(lin2) == 2:
            # (ABI 8) the coefficients derived in the launch from the BatchNorm's raw backward sums: lin2 = (dy2 tensor, (bsums, replicas,
            # count, gamma, mean, rstd, dgamma, dbeta))
            a.dy2 = lin2[0].data_ptr()
            (a.fold_bsums, a.fold_replicas, a.fold_count, a.fold_gamma, a.fold_mean, a.fold_rstd, a.fold_dgamma, a.fold_dbeta) = lin2[1]
        elif lin2 is not None:
            a.dy2, a.dy_scale, a.dy_scale2, a.dy_shift = lin2[0].data_ptr(), lin2[1], lin2[2], lin2[3]
        if res is not None:
            a.dy3, a.dy_out = res[0].data_ptr(), res[1].data_ptr()
        a.x_scale, a.x_shift, a.x_mean, a.x_rstd = bn_ptrs
        a.x_slope = slope
        a.w = self.packs.data_ptr() + self.packs.element_size() * cv.dgrad_off
        a.bsums, a.replicas, a.groups = bsums, replicas, groups
        a.dw = self.grad.data_ptr() + 4 * cv.master_off
        a.ws, a.ws_elems, a.block_budget = self._wg_ws().data_ptr(), self._ws_elems, budget or self.fused_blocks
        self._fused_since_fork = True
        L.call("sv_bwd3x3", C.byref(g), self.code, C.byref(a), self._stream())

    # ------------------------------------------------------------------------------- forward
    def _bn_layout(self, G):
        """Offsets (in floats) of every BatchNorm's [scale | shift | mean | rstd] block in the per-forward scratch for G
        groups: each of the four arrays is [G][C], padded to the alignment as a whole."""
        lay = self._bn_layouts.get(G)
        if lay is None:
            off, o = {}, 0
            for b in self.plan.bns:
                off[b.index] = o
                o += 4 * _align(G * b.C)
            lay = self._bn_layouts[G] = (off, o)
        return lay

    def forward(self, image, groups, eps, u, temperature, training, keep, rec_groups=None, update_order=None, x16=None):
        """One BATCHED forward of G = len(groups) independent instances of the network that share the weights (the
        forwards (1)-(4) of a SHOT-VAE step, main_shot_vae.py:288,311,329,356, or a single one): every launch carries
        the G groups (sv_igemm_args::groups), each with its OWN BatchNorm batch statistics -- the reference's
        semantics of separate model(...) calls -- so the step costs a quarter of the launches at four times the rows.

        image   NCHW fp32 on device, [G * B] images: the groups back to back, B images each
        groups  list of (mode, label, label_mix, lam): the sampler mode of each group (vae.py:38-52): 0 Gumbel-softmax,
                1 one-hot(label), 2 lam * onehot(label) + (1 - lam) * onehot(label_mix)
        eps, u  [G * B, ldc] Gaussian noise, [G * B, K] uniform noise (rows of groups with mode != 0 are ignored) or None
        rec_groups  Gd <= G: only the first Gd groups' reconstructions are produced (and their decoder differentiated).  The
                reconstructions of the mixed forwards (2) and (4) enter no loss term (main_shot_vae.py:311,356: `*_`), so the
                step puts (1) and (3) first and skips the last ConvTranspose of the other two and their whole decoder
                backward -- the decoder up to its last BatchNorm still runs for every group: its running statistics are
                updated by every train-mode forward.  rec is then [Gd * B, ...].
        update_order  order[k] = the group of the reference's k-th forward (running-statistic updates; default group order)
        x16     optional: the NHWC16 image tensor (sv_nchw_to_nhwc of `image`) when the caller has already made it -- the
                grouped step converts on its input-side stream, beside the previous step's backward
        Returns (rec NCHW fp32, mu, ls, la, ctx-or-None), all [G * B, ...]."""
        self._require_gpu(image)
        p = self.plan
        G = len(groups)
        Bt = image.shape[0]
        assert Bt % G == 0, "the groups of a batched forward have equal batch sizes"
        B = Bt // G
        Gd = G if rec_groups is None else int(rec_groups)
        assert 0 <= Gd <= G
        dev = image.device
        T = self.tdtype
        st = self._stream()
        self.ensure_packs()
        image = image.contiguous().float()
        pbase, bbase = self.param.data_ptr(), self.bufs.data_ptr()
        pk, es = self.packs.data_ptr(), self.packs.element_size()

        # per-forward scratch: BN statistics [G][R][2C] (zeroed), BN affine/mean/rstd [G][C]
        n_stat = 0
        stat_off, stat_rep, stat_c = {}, {}, {}

        def stat_slot(name, c, rows):
            nonlocal n_stat
            stat_c[name] = c
            stat_off[name] = n_stat
            stat_rep[name] = _replicas(rows)
            n_stat += _align(G * stat_rep[name] * 2 * c)

        stat_slot("t0", 16, B * p.img * p.img)
        hs = p.img
        for i, un in enumerate(p.units):
            hs //= un["stride"]
            stat_slot("c1_%d" % i, un["cout"], B * hs * hs)
            stat_slot("t%d" % (i + 1), un["cout"], B * hs * hs)
        for i in range(5):
            stat_slot("h%d" % i, p.dec_convs[i].N, B * p.dec_convs[i].Hout ** 2)
        stats = torch.zeros(n_stat, dtype=torch.float64, device=dev)       # sv_acc_t: the accumulators are doubles (ABI 6)
        sbase = stats.data_ptr()
        det = training and L.det_stats()
        det_stats = {}                 # deterministic mode: name -> (tensor [G][R][2C], R), sized by the launch's grid
        bn_off, n_bnbuf = self._bn_layout(G)
        bnbuf = torch.empty(n_bnbuf, dtype=torch.float32, device=dev)
        nb = bnbuf.data_ptr()
        # running statistics: always through the deferred update (sv_bn_running_update_ex from the saved mean / rstd) -- a
        # finalisation folded into its consumer kernel has no running-statistics side effect
        defer = training

        def bn_ptrs(b):
            a = _align(G * b.C)
            o = nb + 4 * bn_off[b.index]
            return o, o + 4 * a, o + 8 * a, o + 12 * a       # scale, shift, mean, rstd: [G][C] each

        def finalize(b, stat_name, count, fold=False):
            """(scale, shift, slope[, fold]) of BatchNorm b.  fold=True (the consumer is an sv_igemm launch): no launch here --
            the consumer finalises (engine._igemm); not in deterministic mode (thousands of accumulator replicas)."""
            sc, sh, mn, rs = bn_ptrs(b)
            if training and fold and self.fold_bn and not det:
                return (sc, sh, b.slope, (sbase + 8 * stat_off[stat_name], stat_rep[stat_name], float(count),
                                          pbase + 4 * b.gamma_off, pbase + 4 * b.beta_off, mn, rs))
            if training:
                if det:
                    sp, sr = det_stats[stat_name][0].data_ptr(), det_stats[stat_name][1]
                else:
                    sp, sr = sbase + 8 * stat_off[stat_name], stat_rep[stat_name]
                L.call("sv_bn_finalize", _vp(sp), sr, b.C, float(count),
                       _vp(pbase + 4 * b.gamma_off), _vp(pbase + 4 * b.beta_off), BN_EPS, BN_MOMENTUM,
                       None if defer else _vp(bbase + 4 * b.rm_off), None if defer else _vp(bbase + 4 * b.rv_off),
                       _vp(sc), _vp(sh), _vp(mn), _vp(rs), G, st)
            else:
                for gi in range(G):       # running statistics: the same affine for every group
                    L.call("sv_bn_eval_affine", b.C, _vp(pbase + 4 * b.gamma_off), _vp(pbase + 4 * b.beta_off),
                           _vp(bbase + 4 * b.rm_off), _vp(bbase + 4 * b.rv_off), BN_EPS, _vp(sc + 4 * gi * b.C),
                           _vp(sh + 4 * gi * b.C), st)
            return (sc, sh, b.slope)

        def sptr(name):
            # in eval mode BN uses running statistics; batch statistics are not accumulated
            if not training:
                return None
            if det:
                def alloc(replicas, name=name):
                    t = torch.zeros(G * replicas * 2 * stat_c[name], dtype=torch.float64, device=dev)
                    det_stats[name] = (t, replicas)
                    return t.data_ptr()
                return alloc
            return (sbase + 8 * stat_off[name], stat_rep[name])

        f = FwdCtx()
        f.B, f.G, f.groups, f.temperature, f.training = B, G, groups, temperature, training
        f.Gd = Gd
        f.bnbuf, f.bn_off = bnbuf, bn_off
        # stem (wideresnet.py:13-14): NCHW fp32 -> NHWC16, conv3x3 + bias, stats of t0
        if x16 is None:
            x16 = self.to_nhwc16(image)
        t = torch.empty(Bt, p.img, p.img, 16, dtype=T, device=dev)
        self._igemm(p.stem.geom_fwd(B), x16, pk + es * p.stem.fwd_off, t, bias=pbase + 4 * p.stem_bias_off,
                    stats=sptr("t0"), tag="fwd:stem", groups=G)
        f.x16, f.t, f.c1, f.pro = x16, [t], [], []
        h = p.img
        for i, un in enumerate(p.units):
            cnt_in = B * h * h
            tin = f.t[-1]
            pro1 = finalize(un["bn1"], "t%d" % i, cnt_in, fold=True)
            ho = h // un["stride"]
            c1 = torch.empty(Bt, ho, ho, un["cout"], dtype=T, device=dev)
            self._igemm(un["conv1"].geom_fwd(B), tin, pk + es * un["conv1"].fwd_off, c1, pro=pro1,
                        stats=sptr("c1_%d" % i), tag="fwd:conv3x3_%dx%d_s%d" % (un["cin"], un["cout"], un["stride"]),
                        groups=G)
            pro2 = finalize(un["bn2"], "c1_%d" % i, B * ho * ho, fold=True)
            tout = torch.empty(Bt, ho, ho, un["cout"], dtype=T, device=dev)
            if "convi" in un:
                proi = finalize(un["bni"], "t%d" % i, cnt_in, fold=True)
                sc = torch.empty(Bt, ho, ho, un["cout"], dtype=T, device=dev)
                self._igemm(un["convi"].geom_fwd(B), tin, pk + es * un["convi"].fwd_off, sc, pro=proi,
                            tag="fwd:conv1x1_%dx%d" % (un["cin"], un["cout"]), groups=G)
                res = sc
            else:
                proi = None
                res = tin
            self._igemm(un["conv2"].geom_fwd(B), c1, pk + es * un["conv2"].fwd_off, tout, pro=pro2, residual=res,
                        stats=sptr("t%d" % (i + 1)), tag="fwd:conv3x3_%dx%d_s1" % (un["cout"], un["cout"]), groups=G)
            f.c1.append(c1)
            f.t.append(tout)
            f.pro.append((pro1[:3], pro2[:3], proi[:3] if proi is not None else None))
            h = ho
        # transition BN + LeakyReLU + global average pool (wideresnet.py:90-91, vae.py:143)
        prot = finalize(p.bn_t, "t%d" % len(p.units), B * h * h)
        feat = torch.empty(Bt, p.cfeat, dtype=torch.float32, device=dev)
        L.call("sv_pool_fwd", self.code, _vp(f.t[-1].data_ptr()), _vp(prot[0]), _vp(prot[1]), prot[2], Bt, h * h,
               p.cfeat, p.cfeat, _vp(feat.data_ptr()), G, st)
        mu = torch.empty(Bt, p.ldc, dtype=torch.float32, device=dev)
        ls = torch.empty(Bt, p.ldc, dtype=torch.float32, device=dev)
        la = torch.empty(Bt, p.K, dtype=torch.float32, device=dev)
        L.call("sv_head_fwd", _vp(feat.data_ptr()), Bt, p.cfeat, _vp(pbase + 4 * p.head_w_off),
               _vp(pbase + 4 * p.head_b_off), p.ldc, p.K, _vp(mu.data_ptr()), _vp(ls.data_ptr()),
               _vp(la.data_ptr()), st)
        latent = torch.empty(Bt, p.Lpad, dtype=T, device=dev)
        csoft = torch.empty(Bt, p.K, dtype=torch.float32, device=dev)
        for gi, (mode, label, label_mix, lam) in enumerate(groups):      # the sampler mode differs per group (vae.py:38-52)
            r0 = gi * B
            L.call("sv_sample_fwd", self.code, _vp(mu[r0:].data_ptr()), _vp(ls[r0:].data_ptr()), _vp(la[r0:].data_ptr()),
                   _vp(eps[r0:].data_ptr()), _vp(u[r0:].data_ptr()) if (u is not None and mode == 0) else None,
                   _vp(label.data_ptr()) if label is not None else None,
                   _vp(label_mix.data_ptr()) if label_mix is not None else None,
                   0.0 if torch.is_tensor(lam) else float(lam), _vp(lam.data_ptr()) if torch.is_tensor(lam) else None,
                   mode, float(temperature), B, p.ldc, p.K, p.Lpad, _vp(latent[r0:].data_ptr()),
                   _vp(csoft[r0:].data_ptr()), st)
        # decoder (decoder.py:12-58)
        f.h = []
        x = latent.view(Bt, 1, 1, p.Lpad)
        pro = None
        f.dpro = []
        f.ha = {}
        for i, cv in enumerate(p.dec_convs):
            ho = cv.Hout
            gl = G if i < 5 else Gd        # the last ConvTranspose (no BatchNorm behind it): only where rec is needed
            if gl == 0:                    # a launch of mixed forwards only (the --om / ragged schedules): no reconstruction
                f.h.append(None)
                continue
            out = torch.empty(gl * B, ho, ho, cv.N, dtype=T, device=dev)
            if pro is not None and self.materialize_decoder_act and cv.Hin <= self.materialize_max_hin:
                # weight-heavy layers (1x1 ... 4x4 maps, 1024 ... 256 channels): BatchNorm + ReLU once, as a pass over a few
                # MB, and a prologue-free GEMM (the LDS-DMA loader) -- fused, the transform was redone per channel tile
                xa = torch.empty_like(x)
                L.call("sv_bn_act", self.code, _vp(x.data_ptr()), _vp(pro[0]), _vp(pro[1]), pro[2], B * cv.Hin * cv.Hin, cv.Cin,
                       _vp(xa.data_ptr()), G, st)
                f.ha[i] = xa
                self._igemm(cv.geom_fwd(B), xa, pk + es * cv.fwd_off, out, stats=sptr("h%d" % i), tag="fwd:dec%d" % i, groups=gl)
            else:
                self._igemm(cv.geom_fwd(B), x, pk + es * cv.fwd_off, out, pro=pro,
                            stats=sptr("h%d" % i) if i < 5 else None, tag="fwd:dec%d" % i, groups=gl)
            f.h.append(out)
            if i < 5:
                # (consumed by the next layer's sv_igemm prologue -- folded -- unless that layer takes the materialised form, or
                #  is the last ConvTranspose of a launch that runs it for the reconstructed groups only: the running statistics
                #  need the mean / rstd of EVERY group)
                pro = finalize(p.dec_bns[i], "h%d" % i, B * ho * ho,
                               fold=not (self.materialize_decoder_act and p.dec_convs[i + 1].Hin <= self.materialize_max_hin) and (i < 4 or Gd == G))
                f.dpro.append(pro[:3])
                x = out
        rec = None
        if Gd > 0:
            rec = torch.empty(Gd * B, p.in_ch, p.img, p.img, dtype=torch.float32, device=dev)
            L.call("sv_nhwc_to_nchw", self.code, _vp(f.h[5].data_ptr()), Gd * B, p.in_ch, p.img, p.img, p.dec_convs[5].N,
                   _vp(rec.data_ptr()), st)
        if training:
            if defer:      # running stats + counter applied by apply_pending(), in slot order (= the reference's forward order)
                slot = self.defer_slot if self.defer_slot is not None else len(self._pending)
                self._pending[slot] = (bnbuf, B, G, update_order)
                if self.defer_slot is None:
                    self.apply_pending()
            else:
                self.nbt += 1
        if not keep:
            return rec, mu, ls, la, None
        f.prot, f.feat, f.mu, f.ls, f.la = prot, feat, mu, ls, la
        f.eps, f.csoft, f.latent = eps, csoft, latent
        f.keep = (groups, u, stats)
        return rec, mu, ls, la, f

    def to_nhwc16(self, image):
        """NCHW fp32 images -> the stem's NHWC tensor with 16 (zero-padded) channels in the compute dtype, on the current stream"""
        p = self.plan
        image = image.contiguous().float()
        x16 = torch.empty(image.shape[0], p.img, p.img, CPAD, dtype=self.tdtype, device=image.device)
        L.call("sv_nchw_to_nhwc", self.code, _vp(image.data_ptr()), image.shape[0], p.in_ch, p.img, p.img, CPAD,
               _vp(x16.data_ptr()), self._stream())
        return x16

    def _bn_counts(self, B):
        """samples per channel of every BatchNorm (in plan order) for batch size B"""
        cnt = {b.index: float(rows) for b, rows in self._bn_rows(B)}
        cnt[self.plan.bn_t.index] = float(B * self.plan.hfeat * self.plan.hfeat)
        return [cnt[b.index] for b in self.plan.bns]

    def apply_pending(self):
        """Apply the deferred running-statistic updates in slot order, the groups of a batched forward in group order (=
        the reference's forward order) on the current stream and advance num_batches_tracked."""
        p = self.plan
        dev = self.param.device
        total = 0
        for k in sorted(self._pending):
            bnbuf, B, G, order = self._pending[k]
            tab = self._run_tables.get((dev, "tab", G))
            if tab is None:
                bn_off, _ = self._bn_layout(G)
                rows = []
                for b in p.bns:
                    rows += [bn_off[b.index], b.rm_off, b.rv_off, b.C]
                tab = self._run_tables[(dev, "tab", G)] = torch.tensor(rows, dtype=torch.int32, device=dev)
            if not torch.cuda.is_current_stream_capturing():
                bnbuf.record_stream(torch.cuda.current_stream())      # allocated on a branch stream, read here
            key = (dev, B)
            cnt = self._run_tables.get(key)
            if cnt is None:
                cnt = self._run_tables[key] = torch.tensor(self._bn_counts(B), dtype=torch.float32, device=dev)
            oarr = (C.c_int32 * G)(*order) if order is not None else None
            L.call("sv_bn_running_update_ex", _vp(tab.data_ptr()), _vp(cnt.data_ptr()), len(p.bns), _vp(bnbuf.data_ptr()),
                   _vp(self.bufs.data_ptr()), BN_EPS, BN_MOMENTUM, 64, G, oarr, self._stream())
            total += G
        self.nbt += total
        self._pending = {}

    def _bn_rows(self, B):
        """(BNSpec, pixels of the tensor it normalises, per group) for every BatchNorm."""
        p = self.plan
        out, h = [], p.img
        for un in p.units:
            out.append((un["bn1"], B * h * h))
            if "bni" in un:
                out.append((un["bni"], B * h * h))
            h //= un["stride"]
            out.append((un["bn2"], B * h * h))
        out.append((p.bn_t, 1))          # its sums come from sv_pool_bwd (plain atomics, one copy)
        for i, b in enumerate(p.dec_bns):
            out.append((b, B * p.dec_convs[i].Hout ** 2))
        return out

    # ------------------------------------------------------------------------------- backward
    def backward(self, f, d_rec, d_mu, d_ls, d_la, own_grads=False):
        """Gradients accumulate (+=) into self.grad; nothing is returned (inputs need no grad).  Batched like the
        forward it belongs to: every launch carries the G groups, the weight gradients sum over them."""
        try:
            self._backward(f, d_rec, d_mu, d_ls, d_la, own_grads)
        except BaseException:
            # the operands kept alive for the side stream must not survive a failed backward into the next step
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            self._side_keep.clear()
            self._pending_wgrads = []
            raise

    def _backward(self, f, d_rec, d_mu, d_ls, d_la, own_grads=False):
        p = self.plan
        B, G, T = f.B, f.G, self.tdtype
        Bt = B * G
        dev = f.mu.device
        st = self._stream()
        pbase, gbase = self.param.data_ptr(), self.grad.data_ptr()
        pk, es = self.packs.data_ptr(), self.packs.element_size()
        nb = f.bnbuf.data_ptr()
        # one zeroed scratch for every (sum g, sum g*xhat) pair of this backward: [G][R][2C] per BatchNorm
        bs_rep, bs_rel, tot = {}, {}, 0
        for b, rows in self._bn_rows(B):
            bs_rep[b.index] = _replicas(rows)
            bs_rel[b.index] = tot
            tot += _align(G * bs_rep[b.index] * 2 * b.C)
        bsums = torch.zeros(tot, dtype=torch.float64, device=dev)          # sv_acc_t
        bs_off = {k: bsums.data_ptr() + 8 * v for k, v in bs_rel.items()}
        det = L.det_stats()
        det_keep = []                  # deterministic mode: per-BatchNorm accumulators sized by the producing launch's grid

        def bnp(b):
            a = _align(G * b.C)
            q = nb + 4 * f.bn_off[b.index]
            return q, q + 4 * a, q + 8 * a, q + 12 * a

        def ex_of(b, raw, Gx=None):
            Gx = Gx or G
            sc, sh, mn, rs = bnp(b)
            if det:
                def alloc(replicas, b=b):
                    t = torch.zeros(Gx * replicas * 2 * b.C, dtype=torch.float64, device=dev)
                    det_keep.append(t)
                    bs_off[b.index], bs_rep[b.index] = t.data_ptr(), replicas
                    return t.data_ptr()
                return (raw, sc, sh, mn, rs, b.slope, alloc, None)
            return (raw, sc, sh, mn, rs, b.slope, bs_off[b.index], bs_rep[b.index])

        def bn_apply(raw, branches, residual, count, Gx=None, sparse=(), compact=False):
            """branches: [(g tensor, BNSpec)] sharing `raw`; returns dL/d(raw) (+ residual).  count = rows of ONE group.
            sparse: indices of branches whose g was written with sparse_out (defined at even positions only)."""
            Gx = Gx or G
            arr = (L.SvBnBranch * len(branches))()
            for k, (g, b) in enumerate(branches):
                arr[k].g = g.data_ptr()
                arr[k].sparse = ((-1 if compact else 1) * int(raw.shape[2]).bit_length()) if k in sparse else 0     # +-(log2(W) + 1)
                arr[k].bsums = bs_off[b.index]
                arr[k].gamma = pbase + 4 * b.gamma_off
                arr[k].dgamma = gbase + 4 * b.gamma_off
                arr[k].dbeta = gbase + 4 * b.beta_off
                arr[k].replicas = bs_rep[b.index]
            b0 = branches[0][1]
            _, _, mn, rs = bnp(b0)
            dx = torch.empty_like(raw)
            cc = raw.shape[-1]
            # reads x and one g per branch (+ the residual), writes dx
            # (a sparse branch is read at one position in four)
            self._cost("sv_bn_bwd_apply", raw.numel() * raw.element_size() *
                       (2 + len(branches) - 0.75 * len(sparse) + (residual is not None)))
            L.call("sv_bn_bwd_apply", self.code, raw.numel() // cc // Gx, cc, cc, _vp(raw.data_ptr()), _vp(mn), _vp(rs),
                   float(count), arr, len(branches), _vp(residual.data_ptr()) if residual is not None else None,
                   _vp(dx.data_ptr()), Gx, st)
            return dx

        def bn_fold(b, count):
            """the arguments of sv_bwd3x3_args::fold_* for BatchNorm b (the launch derives its backward's coefficients itself and adds
            dgamma / dbeta: no sv_bn_bwd_affine launch in front of it)"""
            _, _, mn, rs = bnp(b)
            return (bs_off[b.index], bs_rep[b.index], float(count), pbase + 4 * b.gamma_off, mn, rs, gbase + 4 * b.gamma_off,
                    gbase + 4 * b.beta_off)

        def bn_affine(b, count):
            """scale_g, scale_x, shift [G][C] of BatchNorm b's backward from its sums (one small launch, which also adds dgamma /
            dbeta): the coefficients of a data gradient that forms dx = scale_g * g + scale_x * x + shift in its load path"""
            _, _, mn, rs = bnp(b)
            a_ = _align(G * b.C)
            coef = torch.empty(3 * a_, dtype=torch.float32, device=dev)
            q = coef.data_ptr()
            L.call("sv_bn_bwd_affine", _vp(bs_off[b.index]), bs_rep[b.index], b.C, float(count), _vp(pbase + 4 * b.gamma_off), _vp(mn),
                   _vp(rs), _vp(gbase + 4 * b.gamma_off), _vp(gbase + 4 * b.beta_off), _vp(q), _vp(q + 4 * a_), _vp(q + 8 * a_), G, st)
            return coef, q, q + 4 * a_, q + 8 * a_

        # ---- decoder: only the first Gd groups carry a reconstruction gradient (forward(..., rec_groups)); none at all when
        #      the caller's loss does not use the reconstruction (d_rec is None: the mixed forwards of the sequential step) --
        Gd = f.Gd if d_rec is not None else 0
        Bd = B * Gd
        # (the sampler's backward ADDS to these: a caller that hands over freshly written tensors of its own -- the grouped step's
        #  loss stage -- passes own_grads and saves three copies; autograd's gradients are cloned)
        def mine(t, like):
            if t is None:
                return torch.zeros_like(like)
            if own_grads and t.is_contiguous() and t.dtype == torch.float32:
                return t
            return t.contiguous().float().clone()
        dmu, dls, dla = mine(d_mu, f.mu), mine(d_ls, f.ls), mine(d_la, f.la)
        if Gd > 0:
            last = p.dec_convs[5]
            D = torch.empty(Bd, p.img, p.img, last.N, dtype=T, device=dev)
            L.call("sv_nchw_to_nhwc", self.code, _vp(d_rec.data_ptr()), Bd, p.in_ch, p.img, p.img, last.N,
                   _vp(D.data_ptr()), st)
            for i in range(5, 0, -1):
                cv, b = p.dec_convs[i], p.dec_bns[i - 1]
                hin = f.h[i - 1][:Bd]                     # the groups are back to back: the first Gd are a prefix
                g = torch.empty_like(hin)
                # (where the forward materialised BatchNorm + ReLU of this layer's input, the weight gradient reads that)
                wx, wpro = (f.ha[i][:Bd], None) if i in f.ha else (hin, f.dpro[i - 1])
                # (the last layer's data gradient is a whole-CU kernel of dconv.hip: with a budget it leaves CUs to the weight gradient
                #  beside it -- Engine.pair_blocks_strided)
                dbud = self.pair_blocks_strided if (i == 5 and self.wgrad_side_stream and self.prof_tags is None) else 0
                self._wgrad_async(cv.geom_fwd(B), wx, wpro, D, gbase + 4 * cv.master_off, tag="wgrad:dec%d" % i,
                                  groups=Gd, then=lambda: self._igemm(cv.geom_dgrad(B), D, pk + es * cv.dgrad_off, g,
                                                                      ex=ex_of(b, hin, Gd), tag="dgrad:dec%d" % i, groups=Gd,
                                                                      budget=dbud))
                D = bn_apply(hin, [(g, b)], None, hin.numel() // hin.shape[-1] // Gd, Gd)
            cv = p.dec_convs[0]
            lat4 = f.latent.view(Bt, 1, 1, p.Lpad)[:Bd]
            dlat = torch.empty(Bd, 1, 1, p.Lpad, dtype=T, device=dev)
            self._wgrad_async(cv.geom_fwd(B), lat4, None, D, gbase + 4 * cv.master_off, tag="wgrad:dec0", groups=Gd,
                              then=lambda: self._igemm(cv.geom_dgrad(B), D, pk + es * cv.dgrad_off, dlat, tag="dgrad:dec0",
                                                       groups=Gd))
            # ---- sampler (the latent's gradient reaches mu / log_sigma / log_alpha) -------------------------------
            dl2 = dlat.view(Bd, p.Lpad)
            for gi, (mode, _, _, _) in enumerate(f.groups[:Gd]):
                r0 = gi * B
                L.call("sv_sample_bwd", self.code, _vp(dl2[r0:].data_ptr()), _vp(f.ls[r0:].data_ptr()),
                       _vp(f.eps[r0:].data_ptr()), _vp(f.csoft[r0:].data_ptr()), mode, float(f.temperature), B, p.ldc, p.K,
                       p.Lpad, _vp(dmu[r0:].data_ptr()), _vp(dls[r0:].data_ptr()), _vp(dla[r0:].data_ptr()), st)
        # grad[dec_off:] is complete once the launches issued so far have run -- but only a backward that DID run the decoder
        # may say so: autograd visits forward (4) of the sequential step first, whose reconstruction enters no loss (Gd = 0);
        # firing there would reduce the decoder bucket before forward (3)'s decoder gradients exist.  The hook stays armed
        # (finish() falls back to the single all-reduce if no decoder backward follows).
        if self.bucket_hook is not None and Gd > 0:
            hook, self.bucket_hook = self.bucket_hook, None
            if self._pending_wgrads:       # (fork_every > 1: the decoder's last weight gradients must be ISSUED before the hook)
                self._flush_wgrads(*self._side())
            hook()
        # ---- heads + pool ---------------------------------------------------------------------------------------
        dfeat = torch.empty(Bt, p.cfeat, dtype=torch.float32, device=dev)
        ws = torch.empty(Bt, p.NH, dtype=torch.float32, device=dev)
        L.call("sv_head_bwd", _vp(f.feat.data_ptr()), Bt, p.cfeat, _vp(pbase + 4 * p.head_w_off), p.ldc, p.K,
               _vp(f.la.data_ptr()), _vp(dmu.data_ptr()), _vp(dls.data_ptr()), _vp(dla.data_ptr()),
               _vp(dfeat.data_ptr()), _vp(gbase + 4 * p.head_w_off), _vp(gbase + 4 * p.head_b_off),
               _vp(ws.data_ptr()), st)
        tl = f.t[-1]
        hw = tl.shape[1] * tl.shape[2]
        g = torch.empty_like(tl)
        sc, sh, mn, rs = bnp(p.bn_t)
        L.call("sv_pool_bwd", self.code, _vp(tl.data_ptr()), _vp(sc), _vp(sh), p.bn_t.slope, _vp(mn), _vp(rs),
               _vp(dfeat.data_ptr()), Bt, hw, p.cfeat, p.cfeat, _vp(g.data_ptr()), _vp(bs_off[p.bn_t.index]), G, st)
        D = bn_apply(tl, [(g, p.bn_t)], None, B * hw)
        # ---- encoder units, last to first (wideresnet.py:45-49 backward) ------------------------
        deferred = None
        for i in range(len(p.units) - 1, -1, -1):
            un = p.units[i]
            tin, c1 = f.t[i], f.c1[i]
            pro1, pro2, proi = f.pro[i]
            c = un["cout"]
            # paired launches: the weight gradient (side stream) and the data gradient (main stream) of a body convolution
            # each get `pair` persistent blocks as a per-launch argument (sv_igemm_args / sv_wgrad_args::block_budget)
            pair = self.pair_blocks if ((self.wgrad_side_stream and self.prof_tags is None and
                                         not torch.cuda.is_current_stream_capturing()) or self.prof_paired) else 0
            pair = min(pair, L.lib().sv_get_option(L.OPT_PERSISTENT_BLOCKS))
            same = un["stride"] == 1 and un["cin"] == c
            cnt2 = c1.numel() // c // G
            g2 = torch.empty_like(c1)
            hmap = c1.shape[1]
            fusable = (c == 32 and hmap in (8, 16, 32) and (B * hmap) % (128 // hmap) == 0) or (c == 64 and hmap == 16)
            fb = self.fused_bwd if (self.code == L.SV_BF16 and not det and fusable and c in self.fused_channels) else 0
            if deferred is not None:
                # the unit BEHIND this one left its boundary pass to this launch: D = dL/d(this unit's output) = norm1's BatchNorm
                # backward of g1n + the skip connection's gradient, formed in conv2's load path from (g1n, the next unit's raw input
                # = this unit's output, the gradient at the next unit's output) and written once (the previous unit's skip needs it)
                g1n, tinn, bn1n, Dres, cntn = deferred
                deferred = None
                D = torch.empty_like(Dres)
                if self.fold_bn_bwd:
                    lin2n = (tinn, bn_fold(bn1n, cntn))
                else:
                    coefn, sgn, sxn, shn = bn_affine(bn1n, cntn)
                    lin2n = (tinn, sgn, sxn, shn)
                self._bwd3x3(un["conv2"], B, g1n, lin2n, c1, bnp(un["bn2"]), un["bn2"].slope, g2,
                             bs_off[un["bn2"].index], bs_rep[un["bn2"].index], "bwd:conv3x3_%dx%d_s1+bn+skip" % (c, c), G, res=(Dres, D))
                del lin2n, g1n, Dres
            elif fb >= 3:
                self._bwd3x3(un["conv2"], B, D, None, c1, bnp(un["bn2"]), un["bn2"].slope, g2, bs_off[un["bn2"].index],
                             bs_rep[un["bn2"].index], "bwd:conv3x3_%dx%d_s1" % (c, c), G)
            else:
                self._wgrad_async(un["conv2"].geom_fwd(B), c1, pro2, D, gbase + 4 * un["conv2"].master_off,
                                  tag="wgrad:conv3x3_%dx%d_s1" % (c, c), groups=G, budget=pair,
                                  then=lambda: self._igemm(un["conv2"].geom_dgrad(B), D, pk + es * un["conv2"].dgrad_off, g2,
                                                           ex=ex_of(un["bn2"], c1), tag="dgrad:conv3x3_%dx%d_s1" % (c, c),
                                                           groups=G, budget=pair))
            # (the in-situ table times every launch ALONE: the strided pair's budget only makes sense beside its weight gradient)
            pair1 = pair if same else (min(pair, self.pair_blocks_strided) if self.prof_tags is None else 0)
            cnt = tin.numel() // tin.shape[-1] // G
            g1 = torch.empty_like(tin)
            tag1 = "conv3x3_%dx%d_s%d" % (un["cin"], c, un["stride"])
            if fb >= 1 and same:
                # conv1's WHOLE backward in one launch: dc1 = norm2's BatchNorm backward of g2 is formed in its load path from
                # (g2, c1) and the finished sums (sv_bn_bwd_affine), both products run from that one LDS image
                if self.fold_bn_bwd:
                    lin21 = (c1, bn_fold(un["bn2"], cnt2))
                else:
                    coef, sg, sx, sh = bn_affine(un["bn2"], cnt2)
                    lin21 = (c1, sg, sx, sh)
                self._bwd3x3(un["conv1"], B, g2, lin21, tin, bnp(un["bn1"]), un["bn1"].slope, g1,
                             bs_off[un["bn1"].index], bs_rep[un["bn1"].index], "bwd:" + tag1 + "+bn", G)
                del lin21
                dc1 = None
            else:
                dc1 = bn_apply(c1, [(g2, un["bn2"])], None, cnt2)
                self._wgrad_async(un["conv1"].geom_fwd(B), tin, pro1, dc1, gbase + 4 * un["conv1"].master_off,
                                  tag="wgrad:" + tag1, groups=G, budget=pair1,
                                  then=lambda: self._igemm(un["conv1"].geom_dgrad(B), dc1, pk + es * un["conv1"].dgrad_off, g1,
                                                           ex=ex_of(un["bn1"], tin), tag="dgrad:" + tag1, groups=G, budget=pair1))
            del g2
            del dc1
            cnt = tin.numel() // tin.shape[-1] // G
            if "convi" in un and un["stride"] == 2 and self.compact_shortcut_grad and self.code == L.SV_BF16 and not det:
                # the stride-2 shortcut's data gradient as a DENSE 1x1 product over the stride-2 grid: the raw tensor's even
                # positions are gathered once (sv_gather_even), the gradient is stored compactly and sv_bn_bwd_apply reads it
                # that way (sv_bn_branch::sparse < 0) -- instead of a strided epilogue operand and strided 64-byte row stores
                Hq = tin.shape[1] // 2
                tin_c = torch.empty(tin.shape[0], Hq, Hq, un["cin"], dtype=tin.dtype, device=dev)
                L.call("sv_gather_even", self.code, _vp(tin.data_ptr()), tin.shape[0], tin.shape[1], tin.shape[2], un["cin"],
                       _vp(tin_c.data_ptr()), st)
                gi_ = torch.empty_like(tin_c)
                gdense = _GEOM.conv_like(B, Hq, Hq, c, un["cin"], 1, 1, 0)
                self._wgrad_async(un["convi"].geom_fwd(B), tin, proi, D, gbase + 4 * un["convi"].master_off,
                                  tag="wgrad:conv1x1_%dx%d" % (un["cin"], c), groups=G,
                                  then=lambda: self._igemm(gdense, D, pk + es * un["convi"].dgrad_off, gi_,
                                                           ex=ex_of(un["bni"], tin_c),
                                                           tag="dgrad:conv1x1_%dx%d" % (un["cin"], c), groups=G))
                D = bn_apply(tin, [(g1, un["bn1"]), (gi_, un["bni"])], None, cnt, sparse=(1,), compact=True)
                del tin_c
            elif "convi" in un:
                gi_ = torch.empty_like(tin)
                # a stride-2 shortcut's data gradient is zero at three of four positions: those are neither written nor read
                sp = un["stride"] == 2 and self.sparse_shortcut_grad
                self._wgrad_async(un["convi"].geom_fwd(B), tin, proi, D, gbase + 4 * un["convi"].master_off,
                                  tag="wgrad:conv1x1_%dx%d" % (un["cin"], c), groups=G,
                                  then=lambda: self._igemm(un["convi"].geom_dgrad(B), D, pk + es * un["convi"].dgrad_off, gi_,
                                                           ex=ex_of(un["bni"], tin), sparse_out=sp,
                                                           tag="dgrad:conv1x1_%dx%d" % (un["cin"], c), groups=G))
                D = bn_apply(tin, [(g1, un["bn1"]), (gi_, un["bni"])], None, cnt, sparse=(1,) if sp else ())
            elif fb >= 2 and same and i > 0 and p.units[i - 1]["cout"] == c:
                # the boundary pass (norm1's backward + the skip connection) is left to conv2 of the unit in front (see above)
                deferred = (g1, tin, un["bn1"], D, cnt)
                D = None
            else:
                D = bn_apply(tin, [(g1, un["bn1"])], D, cnt)
        assert deferred is None
        # ---- stem: weight + bias gradients (the image needs none) ---------------------------------
        self._wgrad_async(p.stem.geom_fwd(B), f.x16, None, D, gbase + 4 * p.stem.master_off, tag="wgrad:stem", groups=G)
        L.call("sv_colsum", self.code, _vp(D.data_ptr()), D.numel() // 16, 16, 16, _vp(gbase + 4 * p.stem_bias_off), st)
        self._join_side()
