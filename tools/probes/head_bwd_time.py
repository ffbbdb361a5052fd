"""sv_head_bwd at the grouped size of config 2 (4 x 512 samples, C = 128, ldc = 128, K = 10): run under
rocprofv3 --kernel-trace --stats to see its two kernels separately."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from shot_vae_amd import _lib as L
B, Cc, ldc, K = 2048, 128, 128, 10
NH = 2 * ldc + K
d = torch.device("cuda:0")
p = lambda t: C.c_void_p(t.data_ptr())
feat, W = torch.randn(B, Cc, device=d), torch.randn(NH, Cc, device=d)
la = torch.log_softmax(torch.randn(B, K, device=d), 1)
dmu, dls, dla = torch.randn(B, ldc, device=d), torch.randn(B, ldc, device=d), torch.randn(B, K, device=d)
dfeat, dW, db, ws = torch.zeros(B, Cc, device=d), torch.zeros(NH, Cc, device=d), torch.zeros(NH, device=d), torch.zeros(B, NH, device=d)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(30):
    L.call("sv_head_bwd", p(feat), B, Cc, p(W), ldc, K, p(la), p(dmu), p(dls), p(dla), p(dfeat), p(dW), p(db), p(ws), st)
torch.cuda.synchronize()
ref = (torch.cat([dmu, dls, dla - la.exp() * dla.sum(1, keepdim=True)], 1).t() @ feat) * 30
print("dW rel err", float((dW - ref).abs().max() / ref.abs().max()))
