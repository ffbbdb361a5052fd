"""Input-pipeline oracle (oracle/augment_oracle.py) against independent formulations, and the host-side index split
(CPU: no kernels)."""
import numpy as np
import torch

from oracle import augment_oracle as A


def test_augment_oracle_matches_index_formula():
    """out[y][x] = img[r(oy + y - pad)][r((flip ? Wp-1-(ox+x) : ox+x) - pad)], r = reflect without edge repeat --
    the closed form the HIP kernel implements -- against the numpy.pad / slice restatement."""
    rs = np.random.RandomState(0)
    img = rs.randint(0, 256, size=(32, 32, 3)).astype(np.uint8)

    def r(t, n):
        return -t if t < 0 else (2 * n - 2 - t if t >= n else t)

    for oy, ox, flip in [(0, 0, 0), (8, 8, 1), (4, 4, 0), (4, 4, 1), (3, 7, 1), (8, 0, 0)]:
        ref = A.augment(img, oy, ox, flip)
        got = np.empty_like(ref)
        for y in range(32):
            for x in range(32):
                px = ox + x
                sx = r((39 - px if flip else px) - 4, 32)
                got[:, y, x] = img[r(oy + y - 4, 32), sx, :].astype(np.float32) / np.float32(255)
        assert np.array_equal(ref, got), (oy, ox, flip)
    # identity crop without flip returns the image itself
    assert np.array_equal(A.augment(img, 4, 4, 0), img.transpose(2, 0, 1).astype(np.float32) / np.float32(255))
    # flip at the centred crop mirrors the image
    assert np.array_equal(A.augment(img, 4, 4, 1), img[:, ::-1].transpose(2, 0, 1).astype(np.float32) / np.float32(255))


def test_ssl_split_counts_and_membership():
    """shot_vae_amd.data.ssl_split against the oracle restatement of lib/dataloader.py:142-166 for the same
    per-class permutations, plus the structural properties (class balance, labelled part inside the unlabelled)."""
    from shot_vae_amd.data import ssl_split
    K, n = 10, 500
    rs = np.random.RandomState(1)
    labels = rs.randint(0, K, size=n)
    g = torch.Generator().manual_seed(5)
    valid, tl, tu = ssl_split(torch.tensor(labels), 5, 7, K, generator=g)
    g2 = torch.Generator().manual_seed(5)
    perms = [torch.randperm(int((labels == c).sum()), generator=g2).numpy() for c in range(K)]
    v2, l2, u2 = A.ssl_split(labels, 5, 7, K, perms)
    assert valid.tolist() == v2 and tl.tolist() == l2 and tu.tolist() == u2
    assert len(valid) == 5 * K and len(tl) == 7 * K and len(tu) == n - 5 * K
    assert set(tl.tolist()) <= set(tu.tolist()) and not (set(valid.tolist()) & set(tu.tolist()))
    for c in range(K):
        assert int((labels[tl.numpy()] == c).sum()) == 7


def test_ssl_split_matches_the_reference_samplers():
    """PINNED: get_cifar10_ssl_sampler / get_cifar100_ssl_sampler of the reference itself (lib/dataloader.py:142-190, imported by
    tests/golden/make_goldens.py `ssl` with scripted torch.randperm draws; fixture ref_ssl_samplers.npz) against BOTH the oracle's
    restatement (the recorded permutations) and shot_vae_amd.data.ssl_split (the same draws replayed through torch.randperm):
    identical validation / labelled / unlabelled index lists, order included -- also where a class has fewer samples left than
    `annotated_num_per_class` asks for."""
    import os
    from shot_vae_amd import data as D
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_ssl_samplers.npz"))
    for name in ("c10", "c100", "c10_ragged"):
        labels = g[name + ".labels"]
        nv, na, K = (int(v) for v in g[name + ".args"])
        counts = [int((labels == c).sum()) for c in range(K)]
        cat, perms, o = g[name + ".perm_cat"], [], 0
        for n in counts:
            perms.append(cat[o:o + n])
            o += n
        want = tuple(g[name + "." + k].tolist() for k in ("valid", "train_l", "train_u"))
        assert tuple(A.ssl_split(labels, nv, na, K, perms)) == want, name
        queue = [torch.from_numpy(p) for p in perms]
        real = torch.randperm
        torch.randperm = lambda n, **kw: queue.pop(0)
        try:
            got = D.ssl_split(torch.from_numpy(labels), nv, na, K)
        finally:
            torch.randperm = real
        assert not queue and tuple(t.tolist() for t in got) == want, name


def test_augment_oracle_padding_matches_torch_reflect():
    """a second witness for the unpinned transform chain: torch.nn.functional.pad(mode='reflect') (no edge repeat, the semantics of
    torchvision's Pad(4, padding_mode='reflect')) + flip of the padded image + crop + /255, against the numpy restatement"""
    import torch.nn.functional as F
    rs = np.random.RandomState(3)
    img = rs.randint(0, 256, size=(32, 32, 3)).astype(np.uint8)
    t = torch.from_numpy(img).permute(2, 0, 1).float()[None]
    padded = F.pad(t, (4, 4, 4, 4), mode="reflect")[0]
    for oy, ox, flip in [(0, 0, 0), (8, 8, 1), (2, 5, 1), (7, 1, 0)]:
        q = padded.flip(2) if flip else padded
        want = (q[:, oy:oy + 32, ox:ox + 32] / 255.0).numpy()
        assert np.array_equal(A.augment(img, oy, ox, flip), want.astype(np.float32)), (oy, ox, flip)
