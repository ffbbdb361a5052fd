"""Throughput of one smooth-ELBO trainer iteration (svhn_VAE, BASELINE config 5 shape: batch 1024 per loader) on one GPU."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shot_vae_amd as S
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for dt in ("bf16", "fp32"):
    torch.manual_seed(0)
    m = S.SmoothVAE((3, 32, 32), {"cont": 32, "disc": [10]}, compute_dtype=dt).cuda().train()
    lf, opt = S.SmoothELBOLoss(), torch.optim.Adam(m.parameters(), lr=1e-3)
    u, l, y = torch.rand(B, 3, 32, 32, device="cuda") * 2 - 1, torch.rand(B, 3, 32, 32, device="cuda") * 2 - 1, torch.randint(0, 10, (B,), device="cuda")
    for _ in range(3): S.smooth_train_step(m, lf, opt, u, l, y)
    torch.cuda.synchronize(); t0 = time.time(); n = 20
    for _ in range(n): S.smooth_train_step(m, lf, opt, u, l, y)
    torch.cuda.synchronize(); dt_s = (time.time() - t0) / n
    print("svhn_VAE smooth-ELBO iteration, %s, B_u=B_l=%d: %.2f ms  (%.0f images/s)" % (dt, B, dt_s * 1e3, 2 * B / dt_s))
    opt2 = torch.optim.Adam(m.parameters(), lr=1e-3, capturable=True)
    g = S.GraphedSmoothStep(m, lf, opt2, u, l, y)
    for _ in range(3): g()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): g()
    torch.cuda.synchronize(); dt_s = (time.time() - t0) / n
    print("    hipGraph replay:                        %s, B_u=B_l=%d: %.2f ms  (%.0f images/s)" % (dt, B, dt_s * 1e3, 2 * B / dt_s))
