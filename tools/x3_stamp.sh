#!/bin/bash
# Diagnostic build of conv3x3x.hip with per-block stamps: prologue / K loop / epilogue / (wait + barrier) cycles.
cd "$(dirname "$0")/../shot_vae_amd/csrc" || exit 1
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -I../../include -Wno-unused-function $SV_EXTRA_FLAGS"
/opt/rocm/bin/hipcc $FLAGS -DSV_X3_STAMP -c conv3x3x.hip -o conv3x3x.o 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC igemm.o halo.o hwgrad.o conv3x3.o conv3x3w.o conv3x3x.o wgrad.o wgrad3x3.o small.o runtime.o -o ../libshotvae_hip.so
cd ../.. && for s in "512 160 32 160" "512 640 8 640"; do python - $s <<'PY'
import ctypes as C, sys, torch
sys.path.insert(0, ".")
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
import os
a.x, a.w, a.out = x.data_ptr(), wp.data_ptr(), out.data_ptr()
if not os.environ.get('X3_NO_RESID'): a.residual = resid.data_ptr()
a.pro_scale, a.pro_shift, a.pro_slope = sc.data_ptr(), sh.data_ptr(), 0.01
a.stats, a.replicas = stats.data_ptr(), 8
stats_keep = a.stats
for _ in range(3):
    L.call("sv_igemm", C.byref(g), L.SV_BF16, C.byref(a), st)
torch.cuda.synchronize()
t = stats[16 * N:16 * N + 8 * 256].view(256, 8).cpu().double()
t = t[t[:, 4] > 0]
m = t.mean(0)
items = m[4]
steps = 9 * Cin // 32
print("Cin %d: %d persistent blocks, %.1f items each; cycles: prologue (once) %.0f;  per item: K loop %.0f (%.0f per tap of 20 MFMAs; MFMA-only stream: 660), of which wait+barrier %.0f;  epilogue %.0f  (its second 32-channel group: accumulator dump to LDS %.0f, read back + residual + convert + store %.0f, statistics %.0f)" % (
    Cin, len(t), items, m[0], m[1] / items, m[1] / items / steps, m[3] / items, m[2] / items, m[5], m[6], m[7]))
PY
done
