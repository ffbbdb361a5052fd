"""Drop-ins for lib/utils/mixup.py: mixup_vae_data (:5-26) and label_smoothing (:29-41).

Same host-RNG consumption as the reference (numpy beta, then a CPU torch.randperm moved to the device),
so identical seeds give identical pairings; the gather-lerp itself is one HIP kernel per tensor and the
--om nearest-neighbour search (an O(B^2) Python loop in the reference) is one HIP kernel."""
import ctypes as C

import numpy as np
import torch

from . import _lib as L


def _p(t):
    return C.c_void_p(t.data_ptr())


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _lerp(a, index, lam, exp_space):
    if not a.is_cuda:
        raise L.ShotVaeHipError("shot_vae_amd mixup runs on an MI355X only (no CPU fallback)")
    a = a.contiguous().float()
    out = torch.empty_like(a)
    B = a.shape[0]
    if torch.is_tensor(lam):
        L.call("sv_mix_lerp", _p(a), _p(index), 0.0, _p(lam), B, a[0].numel(), int(exp_space), _p(out), _st())
    else:
        L.call("sv_mix_lerp", _p(a), _p(index), float(lam), None, B, a[0].numel(), int(exp_space), _p(out), _st())
    return out


def optimal_match_index(z_mean, z_log_sigma):
    """index[i] = second-smallest entry of row i of the pairwise KL(N_i || N_j) matrix
    (mixup.py:9-18 with gaussian_kl_divergence_calculation, mixup.py:93-99)."""
    mu, ls = z_mean.contiguous().float(), z_log_sigma.contiguous().float()
    idx = torch.empty(mu.shape[0], dtype=torch.int64, device=mu.device)
    L.call("sv_optimal_match", _p(mu), _p(ls), mu.shape[0], mu.shape[1], _p(idx), _st())
    return idx


def device_permutation(n, device, count=1):
    """`count` random permutations of n drawn on the device (capturable into a hipGraph, unlike a CPU randperm + copy):
    uniform keys + one rank-counting launch (sv_rank_permutation).  Returns int64 [n] (count == 1) or [count, n]."""
    if torch.device(device).type != "cuda" or n > 16384:
        # host-side callers (gloo tests of the non-host-RNG path) and batches beyond the rank-counting kernel's LDS budget
        perm = torch.rand(count, n, device=device).argsort(1)
        return perm[0] if count == 1 else perm
    keys = torch.rand(count * n, device=device)
    perm = torch.empty(count, n, dtype=torch.int64, device=device)
    L.call("sv_rank_permutation", _p(keys), n, count, _p(perm), _st())
    return perm[0] if count == 1 else perm


def mixup_vae_data(image, z_mean, z_log_sigma, disc_log_alpha, optimal_match=False, lam=None, index=None):
    """Returns mixed image, mean, sigma (linear space), alpha (linear space), lambda.  `lam` / `index` (extensions):
    a device scalar / a device index tensor to use instead of the host draws (numpy beta, CPU randperm)."""
    if lam is None:
        lam = np.random.beta(2.0, 2.0)
    batch_size = image.size()[0]
    if optimal_match:
        index = optimal_match_index(z_mean, z_log_sigma)
    elif index is None:
        index = torch.randperm(batch_size).to(image.device)
    index = index.long().contiguous()
    return (_lerp(image, index, lam, False), _lerp(z_mean, index, lam, False),
            _lerp(z_log_sigma, index, lam, True), _lerp(disc_log_alpha, index, lam, True), lam)


def label_smoothing(image, z_mean, z_log_sigma, disc_log_alpha, epsilon=0.1, disc_label=None, lam=None, index=None):
    if lam is None:
        if epsilon > 0:
            lam = np.random.beta(epsilon, epsilon)
        else:
            lam = 1
    batch_size = image.size()[0]
    if index is None:
        index = torch.randperm(batch_size).to(image.device)
    index = index.long().contiguous()
    return (_lerp(image, index, lam, False), _lerp(z_mean, index, lam, False),
            _lerp(z_log_sigma, index, lam, True), _lerp(disc_log_alpha, index, lam, True),
            disc_label[index], lam)
