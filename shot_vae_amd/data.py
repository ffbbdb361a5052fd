"""Device-side input pipeline: the CIFAR training transforms of lib/dataloader.py:58-70 and the
semi-supervised index split of :142-166, for a dataset that lives in HBM as uint8 (CIFAR-10/100: 150 MB) -- at
>80 k images/s per GPU four CPU DataLoader workers cannot feed the step.  One HIP kernel gathers a batch, reflect-pads,
flips, crops and converts it; the random draws (crop offsets, flips, pairings) are device tensors."""
import ctypes as C

import torch

from . import _lib as L


class DeviceDataset:
    """images: uint8 tensor [N][H][W][C] (the layout of the CIFAR arrays), labels: int64 [N]; both moved to `device`."""

    def __init__(self, images_u8_nhwc, labels, device="cuda", pad=4):
        assert images_u8_nhwc.dtype == torch.uint8 and images_u8_nhwc.dim() == 4
        self.images = images_u8_nhwc.contiguous().to(device)
        self.labels = torch.as_tensor(labels, dtype=torch.int64).to(device)
        self.pad = pad

    def draw(self, n, generator=None):
        """(oy, ox, flip) per sample, int32 [n][3], on the device: RandomCrop offsets in [0, 2*pad], flip with p = 0.5."""
        dev = self.images.device
        off = torch.randint(0, 2 * self.pad + 1, (n, 2), device=dev, generator=generator, dtype=torch.int32)
        flip = torch.randint(0, 2, (n, 1), device=dev, generator=generator, dtype=torch.int32)
        return torch.cat([off, flip], 1).contiguous()

    def batch(self, index, train=True, params=None, generator=None, nhwc_dtype=None, cpad=16):
        """index: int64 device tensor [B].  Returns (images, labels): images fp32 NCHW in [0, 1] (the model's input), or,
        with nhwc_dtype ("bf16" / "fp32"), the NHWC [B][H][W][cpad] tensor the stem convolution reads."""
        if not self.images.is_cuda:
            raise L.ShotVaeHipError("shot_vae_amd data pipeline runs on an MI355X only (no CPU fallback)")
        index = index.to(self.images.device, torch.int64).contiguous()
        B = index.numel()
        _, H, W, Cc = self.images.shape
        if train and params is None:
            params = self.draw(B, generator)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if nhwc_dtype is None:
            out = torch.empty(B, Cc, H, W, dtype=torch.float32, device=self.images.device)
            code, cp = L.SV_F32, 0
        else:
            tdt = torch.bfloat16 if nhwc_dtype == "bf16" else torch.float32
            out = torch.empty(B, H, W, cpad, dtype=tdt, device=self.images.device)
            code, cp = (L.SV_BF16 if nhwc_dtype == "bf16" else L.SV_F32), cpad
        L.call("sv_augment", code, C.c_void_p(self.images.data_ptr()), C.c_void_p(index.data_ptr()),
               C.c_void_p(params.data_ptr()) if (train and params is not None) else None, B, H, W, Cc, self.pad, cp,
               C.c_void_p(out.data_ptr()), st)
        return out, self.labels[index]


def ssl_split(labels, valid_per_class, annotated_per_class, num_classes, generator=None):
    """get_cifar10_ssl_sampler / get_cifar100_ssl_sampler (lib/dataloader.py:142-190): per class a random permutation,
    the first `valid_per_class` indices for validation, the next `annotated_per_class` labelled, everything after the
    validation part unlabelled (the labelled part included, as in the reference).  Returns three int64 index tensors
    on the labels' device (feed them to a random sampler / torch.randperm)."""
    labels = torch.as_tensor(labels)
    valid, tl, tu = [], [], []
    for c in range(num_classes):
        loc = torch.nonzero(labels == c).view(-1)
        loc = loc[torch.randperm(loc.numel(), generator=generator, device=loc.device)]
        valid.append(loc[:valid_per_class])
        tl.append(loc[valid_per_class:valid_per_class + annotated_per_class])
        tu.append(loc[valid_per_class:])
    return torch.cat(valid), torch.cat(tl), torch.cat(tu)
