"""ctypes binding of libshotvae_hip.so (the C ABI declared in include/shotvae_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call fails, the product
path raises.  ``build()`` compiles the library in-tree with hipcc for gfx950."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
# SV_LIB_PATH: a diagnostic / A-B build of the library (tools/ab.sh builds variants into scratch paths and selects them
# here; the shipped library is never overwritten by a tool)
LIB_PATH = os.environ.get("SV_LIB_PATH") or os.path.join(HERE, "libshotvae_hip.so")
CSRC = os.path.join(HERE, "csrc")

SV_F32, SV_BF16 = 0, 1
ABI_VERSION = 8                  # include/shotvae_hip.h: SV_ABI_VERSION
MAX_TAPS, MAX_PHASES = 16, 4


class SvPhase(C.Structure):
    _fields_ = [("ooy", C.c_int32), ("oox", C.c_int32), ("ntap", C.c_int32),
                ("dy", C.c_int8 * MAX_TAPS), ("dx", C.c_int8 * MAX_TAPS), ("torig", C.c_int8 * MAX_TAPS),
                ("w_off", C.c_int64)]


class SvGeom(C.Structure):
    _fields_ = [("B", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32), ("Cin", C.c_int32), ("ldx", C.c_int32),
                ("Hq", C.c_int32), ("Wq", C.c_int32), ("sy", C.c_int32), ("sx", C.c_int32),
                ("Hout", C.c_int32), ("Wout", C.c_int32), ("N", C.c_int32), ("ldo", C.c_int32),
                ("osy", C.c_int32), ("osx", C.c_int32), ("T_orig", C.c_int32), ("nphase", C.c_int32),
                ("phase", SvPhase * MAX_PHASES)]


class SvIgemmArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("pro_scale", C.c_void_p), ("pro_shift", C.c_void_p), ("pro_slope", C.c_float),
                ("w", C.c_void_p), ("bias", C.c_void_p), ("residual", C.c_void_p), ("out", C.c_void_p),
                ("stats", C.c_void_p), ("ex", C.c_void_p), ("ex_scale", C.c_void_p), ("ex_shift", C.c_void_p),
                ("ex_mean", C.c_void_p), ("ex_rstd", C.c_void_p), ("ex_slope", C.c_float), ("bsums", C.c_void_p),
                ("replicas", C.c_int32), ("groups", C.c_int32), ("block_budget", C.c_int32), ("flags", C.c_int32), ("sparse_out", C.c_int32), ("reserved0", C.c_int32),
                ("fold_stats", C.c_void_p), ("fold_gamma", C.c_void_p), ("fold_beta", C.c_void_p), ("fold_mean", C.c_void_p),
                ("fold_rstd", C.c_void_p), ("fold_count", C.c_float), ("fold_eps", C.c_float), ("fold_replicas", C.c_int32),
                ("reserved1", C.c_int32), ("start_flag", C.c_void_p), ("start_value", C.c_uint32), ("reserved2", C.c_int32)]


class SvWgradArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("pro_scale", C.c_void_p), ("pro_shift", C.c_void_p), ("pro_slope", C.c_float),
                ("dy", C.c_void_p), ("dw", C.c_void_p), ("splits", C.c_int32), ("use_tr", C.c_int32), ("ws", C.c_void_p),
                ("ws_elems", C.c_int64), ("groups", C.c_int32), ("block_budget", C.c_int32)]


class SvBwd3x3Args(C.Structure):
    _fields_ = [("dy", C.c_void_p), ("dy2", C.c_void_p), ("dy_scale", C.c_void_p), ("dy_scale2", C.c_void_p), ("dy_shift", C.c_void_p),
                ("dy3", C.c_void_p), ("dy_out", C.c_void_p), ("x", C.c_void_p), ("x_scale", C.c_void_p), ("x_shift", C.c_void_p), ("x_mean", C.c_void_p), ("x_rstd", C.c_void_p),
                ("x_slope", C.c_float), ("w", C.c_void_p), ("out", C.c_void_p), ("bsums", C.c_void_p), ("replicas", C.c_int32),
                ("groups", C.c_int32), ("dw", C.c_void_p), ("ws", C.c_void_p), ("ws_elems", C.c_int64), ("block_budget", C.c_int32),
                ("reserved0", C.c_int32),
                ("fold_bsums", C.c_void_p), ("fold_gamma", C.c_void_p), ("fold_mean", C.c_void_p), ("fold_rstd", C.c_void_p),
                ("fold_dgamma", C.c_void_p), ("fold_dbeta", C.c_void_p), ("fold_count", C.c_float), ("fold_replicas", C.c_int32)]


class SvRepackJob(C.Structure):
    _fields_ = [("master_off", C.c_int64), ("dst_off", C.c_int64), ("size", C.c_int64), ("N", C.c_int32),
                ("T_orig", C.c_int32), ("C", C.c_int32), ("transpose", C.c_int32), ("ntap", C.c_int32),
                ("block0", C.c_int32), ("torig", C.c_int8 * MAX_TAPS)]


class SvParamJob(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("dst_off", C.c_int64), ("size", C.c_int64), ("dst_ld", C.c_int64), ("sn_hi", C.c_int64), ("sn_lo", C.c_int64),
                ("st", C.c_int64), ("sc", C.c_int64), ("n_lo_count", C.c_int32), ("N", C.c_int32), ("C", C.c_int32),
                ("ntap", C.c_int32), ("transpose", C.c_int32), ("n_real", C.c_int32), ("c_real", C.c_int32), ("block0", C.c_int32),
                ("torig", C.c_int8 * MAX_TAPS)]


class SvShotSchedule(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("ew", "kl_beta_c", "kl_beta_d", "cmi", "dmi", "pwm", "ucw")]


class SvSmoothSchedule(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("cont_min", "cont_max", "cont_iters", "cont_gamma", "disc_min", "disc_max",
                                         "disc_iters", "disc_gamma", "alpha_cls", "steps")]


class SvShotLossArgs(C.Structure):
    _fields_ = [("rec", C.c_void_p), ("mu", C.c_void_p), ("ls", C.c_void_p), ("la", C.c_void_p), ("image_l", C.c_void_p),
                ("image_u", C.c_void_p), ("label_l", C.c_void_p), ("perm_l", C.c_void_p), ("perm_u", C.c_void_p),
                ("lam_l", C.c_float), ("lam_l_dev", C.c_void_p), ("lam_u", C.c_float), ("lam_u_dev", C.c_void_p),
                ("B", C.c_int32), ("D", C.c_int32), ("K", C.c_int32), ("bce", C.c_int32), ("n_per_img", C.c_int64),
                ("x_sigma", C.c_float), ("sch", SvShotSchedule), ("terms", C.c_void_p), ("coef", C.c_void_p), ("tgt", C.c_void_p),
                ("d_rec", C.c_void_p), ("d_mu", C.c_void_p), ("d_ls", C.c_void_p), ("d_la", C.c_void_p)]


class SvShotLossArgs2(C.Structure):
    _fields_ = [("rec", C.c_void_p * 2), ("mu", C.c_void_p * 4), ("ls", C.c_void_p * 4), ("la", C.c_void_p * 4),
                ("image_l", C.c_void_p), ("image_u", C.c_void_p), ("label_l", C.c_void_p), ("perm_l", C.c_void_p),
                ("perm_u", C.c_void_p), ("lam_l", C.c_float), ("lam_l_dev", C.c_void_p), ("lam_u", C.c_float),
                ("lam_u_dev", C.c_void_p), ("Bl", C.c_int32), ("Bu", C.c_int32), ("D", C.c_int32), ("K", C.c_int32),
                ("bce", C.c_int32), ("reserved0", C.c_int32), ("n_per_img", C.c_int64), ("x_sigma", C.c_float),
                ("sch", SvShotSchedule), ("terms", C.c_void_p), ("coef", C.c_void_p), ("tgt", C.c_void_p),
                ("d_rec", C.c_void_p * 2), ("d_mu", C.c_void_p * 4), ("d_ls", C.c_void_p * 4), ("d_la", C.c_void_p * 4)]


class SvBnBranch(C.Structure):
    _fields_ = [("g", C.c_void_p), ("bsums", C.c_void_p), ("gamma", C.c_void_p), ("dgamma", C.c_void_p),
                ("dbeta", C.c_void_p), ("replicas", C.c_int32), ("sparse", C.c_int32)]


P, I, I64, F = C.c_void_p, C.c_int, C.c_int64, C.c_float
_PROTOS = {
    "sv_igemm": [C.POINTER(SvGeom), I, C.POINTER(SvIgemmArgs), P],
    "sv_igemm_query_blocks": [C.POINTER(SvGeom), I, C.POINTER(SvIgemmArgs), C.POINTER(C.c_int)],
    "sv_wgrad": [C.POINTER(SvGeom), I, P, P, P, F, P, P, I, I, P, I64, I, P],
    "sv_wgrad_ex": [C.POINTER(SvGeom), I, C.POINTER(SvWgradArgs), P],
    "sv_bwd3x3": [C.POINTER(SvGeom), I, C.POINTER(SvBwd3x3Args), P],
    "sv_colsum": [I, P, I64, I, I, P, P],
    "sv_bn_finalize": [P, I, I, F, P, P, F, F, P, P, P, P, P, P, I, P],
    "sv_bn_eval_affine": [I, P, P, P, P, F, P, P, P],
    "sv_bn_act": [I, P, P, P, F, I64, I, P, I, P],
    "sv_bn_running_update": [P, P, I, P, P, F, F, I, I, P],
    "sv_bn_running_update_ex": [P, P, I, P, P, F, F, I, I, C.POINTER(C.c_int32), P],
    "sv_bn_bwd_apply": [I, I64, I, I, P, P, P, F, C.POINTER(SvBnBranch), I, P, P, I, P],
    "sv_pool_fwd": [I, P, P, P, F, I, I, I, I, P, I, P],
    "sv_pool_bwd": [I, P, P, P, F, P, P, P, I, I, I, I, P, P, I, P],
    "sv_head_fwd": [P, I, I, P, P, I, I, P, P, P, P],
    "sv_head_bwd": [P, I, I, P, I, I, P, P, P, P, P, P, P, P, P],
    "sv_sample_fwd": [I, P, P, P, P, P, P, P, F, P, I, F, I, I, I, I, P, P, P],
    "sv_sample_bwd": [I, P, P, P, P, I, F, I, I, I, I, P, P, P, P],
    "sv_elbo_fwd": [P, P, I64, P, P, P, I, I, I, I, F, P, P],
    "sv_elbo_bwd": [P, P, I64, P, P, P, I, I, I, I, F, P, P, P, P, P, P],
    "sv_cls_fwd": [P, P, P, I, I, P, P],
    "sv_cls_bwd": [P, P, I, I, P, P, P],
    "sv_topk_hits": [P, P, I, I, I, P, P],
    "sv_post_fwd": [P, P, P, P, I, I, P, P],
    "sv_post_bwd": [P, P, P, P, I, I, P, P, P, P],
    "sv_mix_lerp": [P, P, F, P, I, I64, I, P, P],
    "sv_optimal_match": [P, P, I, I, P, P],
    "sv_rank_permutation": [P, I, I, P, P],
    "sv_shot_targets": [P, P, P, P, P, P, P, P, F, P, F, P, I, I, I, P, P, P, P, P, P, P],
    "sv_shot_compose": [P, C.POINTER(SvShotSchedule), P, P],
    "sv_shot_scale": [P, P, P, P, P],
    "sv_shot_loss_step": [C.POINTER(SvShotLossArgs), P],
    "sv_shot_loss_step2": [C.POINTER(SvShotLossArgs2), P],
    "sv_shot_targets2": [P, P, P, P, P, P, P, P, F, P, F, P, I, I, I, I, P, P, P, P, P, P, P],
    "sv_sgd": [P, P, P, I64, F, F, F, F, I, P],
    "sv_adam": [P, P, P, P, I64, F, F, F, F, F, P, F, P],
    "sv_smooth_latent_fwd": [I, P, I, P, P, P, F, I, I, I, I, I, P, P, P, P, P, P, P],
    "sv_smooth_latent_bwd": [I, P, I, P, P, P, P, P, P, P, F, I, I, I, I, I, P, I, P],
    "sv_tanh_to_nchw": [I, P, I, I, I, I, I, P, P],
    "sv_tanh_to_nchw_bwd": [I, P, P, I, I, I, I, I, P, P],
    "sv_smooth_elbo_fwd": [P, P, I64, P, P, P, P, I, I, I, C.POINTER(SvSmoothSchedule), P, P, P, P],
    "sv_smooth_elbo_bwd": [P, P, I64, P, P, P, P, I, I, I, P, P, P, P, P, P, P],
    "sv_nchw_to_nhwc": [I, P, I, I, I, I, I, P, P],
    "sv_nhwc_to_nchw": [I, P, I, I, I, I, I, P, P],
    "sv_repack": [I, P, I, I, I, I, C.POINTER(SvGeom), P, P],
    "sv_repack_strided": [I, P, I, I, I64, I64, I64, I, I, I, I, C.POINTER(SvGeom), P, P],
    "sv_repack_batch": [I, P, P, I, I, P, P],
    "sv_param_gather": [I, P, I, I, P, P],
    "sv_param_scatter_add": [P, I, I, P, P],
    "sv_augment": [I, P, P, P, I, I, I, I, I, I, P, P],
    "sv_prof_enable": [I],
    "sv_prof_nested_tag": [I],
    "sv_prof_nested_tag_kind": [I, I],
    "sv_prof_tag": [I],
    "sv_prof_collect": [I, C.POINTER(C.c_double), C.POINTER(C.c_int)],
    "sv_debug_wgrad_tile_program": [I, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "sv_debug_conv_chunk_program": [C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "sv_set_option": [I, I],
    "sv_get_option": [I],
    "sv_bn_bwd_affine": [P, I, I, F, P, P, P, P, P, P, P, P, I, P],
    "sv_gather_even": [I, P, I, I, I, I, P, P],
    "sv_stream_fork": [P, P, I],
    "sv_stream_flag_next": [P, P, P],
    "sv_stream_wait_flag": [P, P, C.c_uint32],
    "sv_flag_timeouts": [],
    "sv_flag_timeouts_reset": [],
    "sv_version": [],
}
OPT_DISABLE_MASK, OPT_WIDE_MIN_BLOCKS, OPT_HALO_ALL, OPT_PERSISTENT_BLOCKS, OPT_DETERMINISTIC, OPT_ENABLE_MASK = 0, 1, 2, 3, 4, 5
(K_CONV3X3, K_CONV3X3P, K_CONV3X3M, K_CONV3X3W, K_CONV3X3X, K_WGRAD3X3, K_WGRAD3X3W, K_IGEMM_KV2, K_HALO, K_HALOP, K_HWGRAD, K_IGEMM_BIG,
 K_WGRAD_WIDE, K_IGEMM_ALIGNED, K_IGEMM_DMA, K_WGRAD_INCR, K_WGRAD3X3M, K_TCONVR, K_TCONVR_EX, K_SCONV) = (1 << i for i in range(20))
# (bits 20-22 and 25 belonged to experiments -- cconv / swgrad / the thconv forward forms / fwd3x3f: tools/experiments/)
K_PCONV, K_THCONV, K_THWGRAD, K_S2WGRAD = 1 << 23, 1 << 24, 1 << 26, 1 << 27
EXPORTS = sorted(list(_PROTOS) + ["sv_last_error"])

_lib = None


class ShotVaeHipError(RuntimeError):
    pass


def build(verbose=False):
    """Compile libshotvae_hip.so in-tree (hipcc --offload-arch=gfx950)."""
    r = subprocess.run(["make", "-C", CSRC, "-j4"], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:])
        print(r.stderr[-4000:])
    if r.returncode != 0:
        raise ShotVaeHipError("building libshotvae_hip.so failed")
    return LIB_PATH


def lib():
    """The loaded library; raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ShotVaeHipError(
                "libshotvae_hip.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C shot_vae_amd/csrc`.  There is no CPU / PyTorch fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, args in _PROTOS.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = C.c_int
        L.sv_last_error.argtypes = []
        L.sv_last_error.restype = C.c_char_p
        # the structs above mirror ONE ABI: a stale library would misread accumulator widths / struct sizes silently
        if L.sv_version() != ABI_VERSION:
            raise ShotVaeHipError("libshotvae_hip.so is ABI %d, this package is written for ABI %d: rebuild (make -C shot_vae_amd/csrc)"
                                  % (L.sv_version(), ABI_VERSION))
        _lib = L
    return _lib


class options:
    """with options(disable=K_CONV3X3X, wide_min_blocks=1): ...  -- dispatcher options for the duration of a block
    (tests / tools: compare a specialised kernel with the general one)."""

    def __init__(self, disable=None, wide_min_blocks=None, halo_all=None, persistent_blocks=None, deterministic=None, enable=None):
        self.new = {OPT_DISABLE_MASK: disable, OPT_WIDE_MIN_BLOCKS: wide_min_blocks, OPT_HALO_ALL: halo_all,
                    OPT_PERSISTENT_BLOCKS: persistent_blocks, OPT_DETERMINISTIC: deterministic, OPT_ENABLE_MASK: enable}

    def __enter__(self):
        self.old = {k: lib().sv_get_option(k) for k in self.new}
        for k, v in self.new.items():
            if v is not None:
                call("sv_set_option", k, int(v))
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            call("sv_set_option", k, v)
        return False


# in-situ timing (bench.py): tag name -> id; every entry point is filed under its own name unless the engine filed the
# launch under a per-layer tag first (Engine._tag, for sv_igemm / sv_wgrad)
prof_tags = None
_LAYER_TAGGED = ("sv_igemm", "sv_igemm_query_blocks", "sv_wgrad", "sv_wgrad_ex", "sv_bwd3x3", "sv_prof_tag", "sv_prof_enable", "sv_set_option")


def deterministic():
    """SV_OPT_DETERMINISTIC = 1: fixed summation order everywhere (include/shotvae_hip.h)"""
    return lib().sv_get_option(OPT_DETERMINISTIC) == 1


def det_stats():
    """SV_OPT_DETERMINISTIC = 1 or 2: the BatchNorm statistics and backward sums of the conv-like launches (2: only those) are
    accumulated in a fixed order -- their accumulators are sized by the launch's grid (det_replicas)"""
    return lib().sv_get_option(OPT_DETERMINISTIC) != 0


def det_replicas(g, code, a):
    """replica count sv_igemm needs for `stats` / `bsums` in deterministic mode: next power of two >= 4 * blocks"""
    blocks = C.c_int(0)
    if a.replicas < 1:
        a.replicas = 1                 # (the argument check wants a power of two; the query does not use it)
    call("sv_igemm_query_blocks", C.byref(g), code, C.byref(a), C.byref(blocks))
    r = 1
    while r < 4 * blocks.value:
        r *= 2
    return r


def check_flag_timeouts(where=""):
    """Fail closed on the device-side fork (sv_stream_wait_flag): a wait that gave up let a weight gradient read operands that
    were not written yet -- the gradients of that step (and, after an all-reduce, of every rank) are garbage.  The counter is
    sticky and lives in host-mapped memory: reading it costs neither a copy nor a synchronisation, so every step checks it
    (Engine._join_side, FlatSGD.step, dp's all-reduce); a time-out that has not been executed yet is caught one step later."""
    n = lib().sv_flag_timeouts()
    if n:
        raise ShotVaeHipError(
            "%d side-stream wait(s) for a data gradient's start signal timed out%s: weight gradients ran on unfinished operands.  "
            "FATAL for this run: the time-out is only visible once the waiting kernel has executed, so the optimizer step (and, with "
            "several ranks, the all-reduce) of that step may already have applied the corrupted gradients -- the parameters and the "
            "momentum buffer cannot be trusted: restore the last checkpoint, do not save one.  Kernel dispatch is serialised or the "
            "main queue stalled > 3 s (profiler, debugger, shared GPU): set Engine.flag_fork = False (event forks) and re-run.  (The "
            "counter stays set until sv_flag_timeouts_reset(): every later step raises too.)" % (n, (" (" + where + ")") if where else ""))


def call(name, *args):
    if prof_tags is not None and name not in _LAYER_TAGGED:
        lib().sv_prof_tag(prof_tags.setdefault(name, len(prof_tags)))
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise ShotVaeHipError("%s failed (%d): %s" % (name, rc, lib().sv_last_error().decode()))
