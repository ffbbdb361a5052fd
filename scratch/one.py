import ctypes as C, sys, torch
sys.path.insert(0, '.')
from shot_vae_amd import _lib as L, geometry as G
d = torch.device("cuda:0")
B, H, Cin, N = 512, 32, 32, 32
mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
g = G.conv_like(B, H, H, Cin, N, 3, 1, 1)
dt = torch.bfloat16
x = torch.randn(B, H, H, Cin, device=d).to(dt); w = (torch.randn(G.packed_size(g), device=d) * 0.05).to(dt)
out = torch.empty(B, H, H, N, device=d, dtype=dt); res = torch.randn(B, H, H, N, device=d).to(dt)
sc, sh = torch.rand(Cin, device=d) + 0.5, torch.randn(Cin, device=d)
stats = torch.zeros(32 * 2 * N, device=d)
a = L.SvIgemmArgs()
a.x, a.w, a.out = x.data_ptr(), w.data_ptr(), out.data_ptr()
a.replicas = 32
if mode == "fwd":
    a.pro_scale, a.pro_shift, a.pro_slope = sc.data_ptr(), sh.data_ptr(), 0.01
    a.residual = res.data_ptr(); a.stats = stats.data_ptr()
elif mode == "bare":
    pass
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(10):
    L.call("sv_igemm", C.byref(g), L.SV_BF16, C.byref(a), st)
torch.cuda.synchronize()
