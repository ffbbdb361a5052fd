"""Diagnostic (build conv3x3w.hip with -DSV_W3_STAMP): prologue / main-loop / epilogue cycles of the first blocks."""
import ctypes as C, sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from shot_vae_amd import _lib as L, geometry as G
B, Cin, H, N = map(int, sys.argv[1:5])
d = torch.device("cuda:0"); st = C.c_void_p(torch.cuda.current_stream().cuda_stream); bf = torch.bfloat16
x = torch.randn(B, H, H, Cin, device=d).to(bf)
master = (torch.randn(N, 9, Cin, device=d) / (9 * Cin) ** 0.5).contiguous()
g = G.conv_like(B, H, H, Cin, N, 3, 1, 1)
wp = torch.zeros(G.packed_size(g), dtype=bf, device=d)
L.call("sv_repack", L.SV_BF16, C.c_void_p(master.data_ptr()), N, 9, Cin, 0, C.byref(g), C.c_void_p(wp.data_ptr()), st)
out = torch.empty(B, H, H, N, dtype=bf, device=d); resid = torch.randn(B, H, H, N, device=d).to(bf)
sc, sh = torch.rand(Cin, device=d) + 0.5, torch.randn(Cin, device=d) * 0.3
stats = torch.zeros(8 * 4096 + 8 * 2 * N, device=d)
a = L.SvIgemmArgs()
a.x, a.w, a.out, a.residual = x.data_ptr(), wp.data_ptr(), out.data_ptr(), resid.data_ptr()
a.pro_scale, a.pro_shift, a.pro_slope = sc.data_ptr(), sh.data_ptr(), 0.01
a.stats, a.replicas = stats.data_ptr(), 8
for _ in range(5):
    L.call("sv_igemm", C.byref(g), L.SV_BF16, C.byref(a), st)
torch.cuda.synchronize()
TR = 256 // H
nb = 8 * ((B * H // TR + 7) // 8) * (N // (160 if N % 160 == 0 else 128))
t = stats[:8 * nb].view(nb, 8).cpu().double()
t0 = t[:, 3].min()
start, end = (t[:, 3] - t0) / 100.0, (t[:, 4] - t0) / 100.0          # microseconds
dur = end - start
print("blocks", nb, "kernel span %.1f us" % end.max().item(), "block duration us: mean %.1f min %.1f max %.1f" % (dur.mean(), dur.min(), dur.max()))
print("cycles prologue / loop / epilogue (mean):", t[:, :3].mean(0).tolist())
for lo in range(0, int(end.max().item()) + 1, 25):
    act = ((start <= lo) & (end > lo)).sum().item()
    print("t=%4d us: %4d blocks running, %4d started so far" % (lo, act, (start <= lo).sum().item()))
hw = t[:, 5].long()
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7; xcc = t[:, 6].long()
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
first = key[start < 5.0]
import collections
c = collections.Counter(first.tolist())
print("first-wave blocks:", len(first), "distinct CUs:", len(c), "blocks per CU histogram:", collections.Counter(c.values()))
# lockstep check: for every block, the start-time distance to the nearest OTHER block of the same CU that overlaps it
import numpy as np
keys = key.numpy(); s_ = start.numpy(); e_ = end.numpy()
dist = []
for k in np.unique(keys):
    idx = np.where(keys == k)[0]
    for i in idx:
        others = [j for j in idx if j != i and s_[j] < e_[i] and e_[j] > s_[i]]
        if others:
            dist.append(min(abs(s_[j] - s_[i]) for j in others))
dist = np.array(dist)
dm = dur.mean().item()
print("co-resident blocks: start-time distance to the partner, as a fraction of the block duration (0 = lockstep, 0.5 = ideal stagger):")
print("  quantiles 10/25/50/75/90 %%: %s   (block duration %.1f us)" % (np.round(np.quantile(dist / dm, [0.1, 0.25, 0.5, 0.75, 0.9]), 2).tolist(), dm))
