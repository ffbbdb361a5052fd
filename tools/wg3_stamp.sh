#!/bin/bash
# Diagnostic build of wgrad3x3_kernel with in-kernel stamps: where a 128-pixel tile iteration spends its cycles.
cd "$(dirname "$0")/../shot_vae_amd/csrc" || exit 1
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -I../../include -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS -DSV_WG3_STAMP -c wgrad3x3.hip -o wgrad3x3.o 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC igemm.o halo.o hwgrad.o conv3x3.o conv3x3w.o conv3x3x.o wgrad.o wgrad3x3.o small.o runtime.o -o ../libshotvae_hip.so
cd ../.. && python - "$@" <<'PY'
import ctypes as C, sys, torch
sys.path.insert(0, ".")
from shot_vae_amd import _lib as L, geometry as G
B, Cin, H, N = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (512, 32, 32, 32)))
d = torch.device("cuda:0"); st = C.c_void_p(torch.cuda.current_stream().cuda_stream); bf = torch.bfloat16
x = torch.randn(B, H, H, Cin, device=d).to(bf); dy = torch.randn(B, H, H, N, device=d).to(bf)
g = G.conv_like(B, H, H, Cin, N, 3, 1, 1)
dw = torch.zeros(N, 9, Cin, device=d); sc, sh = torch.rand(Cin, device=d) + 0.5, torch.randn(Cin, device=d) * 0.3
ws = torch.zeros(16 << 20, device=d)
for _ in range(5):
    L.call("sv_wgrad", C.byref(g), L.SV_BF16, C.c_void_p(x.data_ptr()), C.c_void_p(sc.data_ptr()), C.c_void_p(sh.data_ptr()),
           C.c_float(0.01), C.c_void_p(dy.data_ptr()), C.c_void_p(dw.data_ptr()), 0, 1, C.c_void_p(ws.data_ptr()), ws.numel(), st)
torch.cuda.synchronize()
t = ws[8 << 20:(8 << 20) + 8 * 512].view(512, 8).cpu().double()
t = t[t[:, 5] > 0]
per = t[:, :5] / t[:, 5:6]
ld = (t[:, 6] / t[:, 5]).mean().item()
e = ws[9 << 20:(9 << 20) + 8 * 512].view(512, 8).cpu().double()
e = e[e[:, 2] > 0]
print("per block: prologue %.0f  epilogue (publish) %.0f  whole kernel %.0f cycles" % (e[:, 0].mean(), e[:, 1].mean(), e[:, 2].mean()))
print("wait for prefetch %.0f" % (t[:, 7] / t[:, 5]).mean().item())
print("blocks %d, tiles per block %.1f; cycles per tile: store %.0f  barrier1 %.0f  issue next loads %.0f  fragment reads + mfma %.0f  barrier2 %.0f  total %.0f" % (
    len(t), t[:, 5].mean(), per[:, 0].mean(), per[:, 1].mean(), ld, per[:, 2].mean(), per[:, 3].mean(), per[:, 4].mean()))
PY
