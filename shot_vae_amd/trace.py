"""Tracing hooks of the training path (the reference's only instrumentation is TensorBoard scalars and a timer,
main_shot_vae.py:367-383,256-258).  Two things, both off unless asked for and free when off:

* roctx ranges around the phases of a step (inputs / forward / loss / backward / update), so that a `rocprofv3 --marker-trace` run
  shows them beside the kernels.  `enable_ranges(True)` (or SV_TRACE_RANGES=1) loads libroctx64.so; without it every range is a no-op.
* `StepLogger(path)`: one JSON line per step with the three ELBO terms of both batches, the posterior terms and the two objectives --
  what the reference sends to `writer.add_scalar` per epoch (`Train/KL_Inference`, main_shot_vae.py:371) and what its progress line
  prints, at step granularity.  It takes the dict `train_step*(..., return_outputs=True)` returns; reading the scalars is ONE
  device-to-host copy of a stacked tensor (the reference syncs once per step too: main_shot_vae.py:339)."""
import contextlib
import ctypes
import json
import os
import time

import torch

_roctx = None
_enabled = None


def enable_ranges(on=True):
    """switch the roctx ranges on / off; returns whether they are active (False when libroctx64.so cannot be loaded)"""
    global _roctx, _enabled
    if on and _roctx is None:
        for name in ("libroctx64.so", "/opt/rocm/lib/libroctx64.so", "librocprofiler-sdk-roctx.so"):
            try:
                lib = ctypes.CDLL(name)
                lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                lib.roctxRangePushA.restype = ctypes.c_int
                lib.roctxRangePop.restype = ctypes.c_int
                _roctx = lib
                break
            except (OSError, AttributeError):
                continue
    _enabled = bool(on and _roctx is not None)
    return _enabled


def ranges_enabled():
    global _enabled
    if _enabled is None:
        _enabled = enable_ranges(os.environ.get("SV_TRACE_RANGES", "0") not in ("", "0"))
    return _enabled


@contextlib.contextmanager
def step_range(name):
    """with step_range("backward"): ...  -- a roctx range named "shot_vae/<name>" (no-op unless enabled)"""
    if not ranges_enabled():
        yield
        return
    _roctx.roctxRangePushA(("shot_vae/" + name).encode())
    try:
        yield
    finally:
        _roctx.roctxRangePop()


SCALARS = ("recon_l", "klc_l", "kld_l", "recon_u", "klc_u", "kld_u", "disc_post_l", "cont_post_l", "disc_post_u", "cont_post_u",
           "loss_sup", "loss_unsup")


class StepLogger:
    """JSONL writer: {"step": n, "time": t, "recon_l": ..., ..., "loss_unsup": ...[, "kl_inference": ...][, extra...]} per call."""

    def __init__(self, path, flush_every=50):
        d = os.path.dirname(os.path.abspath(path))
        os.makedirs(d, exist_ok=True)
        self._f = open(path, "a")
        self._n, self._flush_every, self._t0 = 0, flush_every, time.time()

    def log(self, outputs, step=None, **extra):
        keys = [k for k in SCALARS + ("kl_inference",) if k in outputs]
        vals = torch.stack([outputs[k].detach().float().reshape(()) for k in keys]).cpu().tolist()       # one copy, one sync
        rec = {"step": self._n if step is None else int(step), "time": round(time.time() - self._t0, 6)}
        rec.update(zip(keys, vals))
        rec.update(extra)
        self._f.write(json.dumps(rec) + "\n")
        self._n += 1
        if self._n % self._flush_every == 0:
            self._f.flush()
        return rec

    def close(self):
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
