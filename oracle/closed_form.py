"""Closed-form (RNG-free, bit-portable) generators for golden-vector cases  --  TEST INFRASTRUCTURE.

Both ``tests/golden/make_goldens.py`` (which runs the *reference* in the build container) and the
parity tests (which run the oracle / the HIP path, possibly on the GPU box) regenerate identical
weights, inputs and noise from these integer-hash formulas, so the committed fixtures only need to
hold the reference's *outputs*.
"""
import math

import numpy as np
import torch

from . import shotvae_oracle as O

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def hash_uniform(n, stream):
    """n float64 values in [0,1): splitmix64 of (index, stream); exact integer arithmetic."""
    with np.errstate(over="ignore"):
        x = (np.arange(n, dtype=np.uint64) + np.uint64(stream) * np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        x = x ^ (x >> np.uint64(31))
    return (x >> np.uint64(11)).astype(np.float64) / float(1 << 53)


def uniform(shape, stream, lo=0.0, hi=1.0):
    n = int(np.prod(shape)) if len(shape) else 1
    v = hash_uniform(n, stream) * (hi - lo) + lo
    return torch.from_numpy(v.astype(np.float32)).reshape(shape)


def normal(shape, stream):
    """Box-Muller on two hash streams (float64 math, rounded to fp32 once)."""
    n = int(np.prod(shape))
    u1 = hash_uniform(n, stream * 2 + 1_000_003)
    u2 = hash_uniform(n, stream * 2 + 1_000_004)
    v = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * math.pi * u2)
    return torch.from_numpy(v.astype(np.float32)).reshape(shape)


def permutation(n, stream):
    return torch.from_numpy(np.argsort(hash_uniform(n, stream), kind="stable").astype(np.int64))


def make_state(name, in_ch=3, ldc=128, K=10, img=32, stream0=100):
    """Every parameter / buffer of the reference state_dict from the hash generator: conv, convT
    and linear tensors U(+-1/sqrt(fan_in)); BN gamma in [0.8,1.2], beta in [-0.1,0.1],
    running_mean in [-0.05,0.05], running_var in [0.9,1.1]."""
    st = {}
    for i, (k, shp) in enumerate(O.state_shapes(name, in_ch, ldc, K, img).items()):
        s = stream0 + i
        if k.endswith("num_batches_tracked"):
            st[k] = torch.zeros((), dtype=torch.int64)
        elif k.endswith("running_mean"):
            st[k] = uniform(shp, s, -0.05, 0.05)
        elif k.endswith("running_var"):
            st[k] = uniform(shp, s, 0.9, 1.1)
        elif len(shp) == 1 and (".norm" in k or k.split(".")[-2].isdigit()):
            st[k] = uniform(shp, s, 0.8, 1.2) if k.endswith("weight") else uniform(shp, s, -0.1, 0.1)
        else:
            if len(shp) == 4:
                fan_in = shp[1] * shp[2] * shp[3]
            elif len(shp) == 2:
                fan_in = shp[1]
            else:
                fan_in = int(np.prod(st[k[:-4] + "weight"].shape[1:]))
            b = 1.0 / math.sqrt(fan_in)
            st[k] = uniform(shp, s, -b, b)
    return st


def make_batch(Bl, Bu, K, in_ch=3, img=32, stream0=7000):
    image_l = uniform((Bl, in_ch, img, img), stream0)
    image_u = uniform((Bu, in_ch, img, img), stream0 + 1)
    label_l = (torch.arange(Bl) * 7 + 3) % K
    label_u = (torch.arange(Bu) * 5 + 1) % K
    return image_l, label_l, image_u, label_u


def make_noise(Bl, Bu, K, ldc=128, stream0=9000, lam_l=0.93, lam_u=0.37):
    return dict(eps1=normal((Bl, ldc), stream0), lam_l=lam_l, perm_l=permutation(Bl, stream0 + 1),
                eps2=normal((Bl, ldc), stream0 + 2), eps3=normal((Bu, ldc), stream0 + 3),
                u3=uniform((Bu, K), stream0 + 4), lam_u=lam_u, perm_u=permutation(Bu, stream0 + 5),
                eps4=normal((Bu, ldc), stream0 + 6), u4=uniform((Bu, K), stream0 + 7))
