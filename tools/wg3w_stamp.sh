#!/bin/bash
# Diagnostic build of wgrad3x3w_kernel (wide layers) with in-kernel stamps.  usage: wg3w_stamp.sh B C H N
cd "$(dirname "$0")/../shot_vae_amd/csrc" || exit 1
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -I../../include -Wno-unused-function $SV_EXTRA_FLAGS"
/opt/rocm/bin/hipcc $FLAGS -DSV_WG3_STAMP -c wgrad3x3.hip -o wgrad3x3.o 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC igemm.o halo.o hwgrad.o conv3x3.o conv3x3w.o conv3x3x.o wgrad.o wgrad3x3.o small.o runtime.o -o ../libshotvae_hip.so
cd ../.. && python - "$@" <<'PY'
import ctypes as C, sys, torch
sys.path.insert(0, ".")
from shot_vae_amd import _lib as L, geometry as G
B, Cin, H, N = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (512, 160, 32, 160)))
d = torch.device("cuda:0"); st = C.c_void_p(torch.cuda.current_stream().cuda_stream); bf = torch.bfloat16
x = torch.randn(B, H, H, Cin, device=d).to(bf); dy = torch.randn(B, H, H, N, device=d).to(bf)
g = G.conv_like(B, H, H, Cin, N, 3, 1, 1)
dw = torch.zeros(N, 9, Cin, device=d); sc, sh = torch.rand(Cin, device=d) + 0.5, torch.randn(Cin, device=d) * 0.3
ws = torch.zeros(32 << 20, device=d)
for _ in range(3):
    L.call("sv_wgrad", C.byref(g), L.SV_BF16, C.c_void_p(x.data_ptr()), C.c_void_p(sc.data_ptr()), C.c_void_p(sh.data_ptr()),
           C.c_float(0.01), C.c_void_p(dy.data_ptr()), C.c_void_p(dw.data_ptr()), 0, 1, C.c_void_p(ws.data_ptr()), 24 << 20, st)  # noqa
torch.cuda.synchronize()
t = ws[24 << 20:(24 << 20) + 8 * 2048].view(2048, 8).cpu().double()
t = t[t[:, 5] > 0]
per = t[:, :5] / t[:, 5:6]
print("blocks %d, tiles per block %.1f (min %d max %d); block cycles mean %.0f max %.0f" % (len(t), t[:, 5].mean(), t[:, 5].min(), t[:, 5].max(), t[:, 4].mean(), t[:, 4].max()))
print("s_memtime ticks per tile: %.0f (MFMA-only stream, tools/probes/issue_probe.hip: 19.5 ticks per MFMA = 3512); phase 0 %.0f  phase 1 %.0f  phase 2 %.0f  wait+barrier %.0f  phase 3 %.0f (878 each)" % (
    per[:, 4].mean(), per[:, 0].mean(), per[:, 1].mean(), per[:, 2].mean(), (t[:, 6] / t[:, 5]).mean(), per[:, 3].mean()))
PY
