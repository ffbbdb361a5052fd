"""CPU oracle (test infrastructure only) of the reference's CIFAR training-input pipeline, lib/dataloader.py:58-70:

    transforms.Pad(4, padding_mode='reflect') -> RandomHorizontalFlip() -> RandomCrop(32) -> ToTensor()

restated in numpy from torchvision 0.4's documented semantics.  PARITY of this transform chain is UNPINNED by reference outputs:
torchvision is not installed in this image and the chain's arithmetic lives there (the reference only composes it), so no fixture
can be generated; tests/test_data_cpu.py holds the restatement to two independent formulations instead (a closed-form index map,
torch's own reflect padding).  See DESIGN.md.
  * Pad(p, 'reflect') = numpy.pad(mode='reflect') on H and W (no edge repeat);
  * the flip acts on the PADDED image, before the crop;
  * RandomCrop(32) of the 40x40 padded image takes rows i..i+31, columns j..j+31, 0 <= i, j <= 8;
  * ToTensor: uint8 HWC -> float32 CHW divided by 255.
Also ssl_split: get_cifar10_ssl_sampler / get_cifar100_ssl_sampler (lib/dataloader.py:142-190) for given per-class permutations --
PINNED: tests/golden/ref_ssl_samplers.npz holds the reference's own outputs (make_goldens.py `ssl` imports the two functions and
scripts torch.randperm), tests/test_data_cpu.py::test_ssl_split_matches_the_reference_samplers."""
import numpy as np


def augment(img_u8_hwc, oy, ox, flip, pad=4):
    """one image uint8 [H][W][C] -> float32 [C][H][W]"""
    H, W, _ = img_u8_hwc.shape
    p = np.pad(img_u8_hwc, ((pad, pad), (pad, pad), (0, 0)), mode="reflect")
    if flip:
        p = p[:, ::-1, :]
    c = p[oy:oy + H, ox:ox + W, :]
    return (c.astype(np.float32) / np.float32(255.0)).transpose(2, 0, 1)


def batch(data_u8, index, params, pad=4):
    """data uint8 [N][H][W][C]; index [B]; params int [B][3] = (oy, ox, flip) or None (evaluation: ToTensor only)"""
    out = []
    for b, i in enumerate(index):
        if params is None:
            out.append((data_u8[i].astype(np.float32) / np.float32(255.0)).transpose(2, 0, 1))
        else:
            out.append(augment(data_u8[i], int(params[b][0]), int(params[b][1]), bool(params[b][2]), pad))
    return np.stack(out)


def ssl_split(labels, valid_per_class, annotated_per_class, num_classes, perms):
    """lib/dataloader.py:142-166 with the per-class permutations given: (valid, train_l, train_u) index lists.
    train_u contains the labelled part as well (the reference's comment at :160-161)."""
    valid, tl, tu = [], [], []
    for c in range(num_classes):
        loc = np.nonzero(labels == c)[0]
        loc = loc[perms[c]]
        valid += loc[:valid_per_class].tolist()
        tl += loc[valid_per_class:valid_per_class + annotated_per_class].tolist()
        tu += loc[valid_per_class:].tolist()
    return valid, tl, tu
