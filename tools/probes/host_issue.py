"""How long does the HOST need to issue one eager grouped step (bench.py's default path)?  If it is close to the GPU's step
time the step is launch-bound and every host hiccup shows up as idle GPU.  python tools/probes/host_issue.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import shot_vae_amd as S
from shot_vae_amd.train import train_step_grouped
K, B = 10, 512
torch.manual_seed(1)
model = S.VariationalAutoEncoder("wideresnet-28-2", num_input_channels=3, img_size=(32, 32), data_parallel=True,
                                 continuous_latent_dim=128, disc_latent_dim=K, small_input=True,
                                 compute_dtype="bf16", rng="device").cuda().train()
elbo, cls = S.VAECriterion(discrete_dim=K, bce_reconstruction=True).cuda(), S.ClsCriterion()
opt = S.FlatSGD(model, lr=0.02, momentum=0.9, weight_decay=5e-4); opt.zero_grad()
sch = S.schedule(10)
il, iu, ll = torch.rand(B, 3, 32, 32, device="cuda"), torch.rand(B, 3, 32, 32, device="cuda"), torch.randint(0, K, (B,), device="cuda")
step = lambda: train_step_grouped(model, elbo, cls, opt, il, ll, iu, sch)
for _ in range(5): step()
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for _ in range(N): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host issue %.3f ms/step, wall %.3f ms/step (%d steps queued without a sync)" % (1e3 * (t1 - t0) / N, 1e3 * (t2 - t0) / N, N))
# the host alone: the same loop with the GPU already idle at every step
host = []
for _ in range(N):
    torch.cuda.synchronize()
    t = time.perf_counter()
    step()
    host.append(time.perf_counter() - t)
torch.cuda.synchronize()
print("host issue with an idle GPU: mean %.3f ms, min %.3f ms" % (1e3 * sum(host) / N, 1e3 * min(host)))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
