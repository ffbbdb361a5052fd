"""How long does the HOST spend in hipGraphLaunch for the captured step?  (python tools/replay_host_time.py)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shot_vae_amd as S
from shot_vae_amd.train import GraphedTrainStep
K, B = 10, 512
torch.manual_seed(1)
model = S.VariationalAutoEncoder("wideresnet-28-2", num_input_channels=3, img_size=(32, 32), data_parallel=True,
                                 continuous_latent_dim=128, disc_latent_dim=K, small_input=True,
                                 compute_dtype="bf16", rng="device").cuda().train()
elbo, cls = S.VAECriterion(discrete_dim=K, bce_reconstruction=True).cuda(), S.ClsCriterion()
opt = S.FlatSGD(model, lr=0.02, momentum=0.9, weight_decay=5e-4); opt.zero_grad()
sch = S.schedule(10)
il, iu, ll = torch.rand(B, 3, 32, 32, device="cuda"), torch.rand(B, 3, 32, 32, device="cuda"), torch.randint(0, K, (B,), device="cuda")
g = GraphedTrainStep(model, elbo, cls, opt, il, ll, iu, sch)
for _ in range(5): g()
torch.cuda.synchronize()
host, wall = [], []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.graph.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    g._update()
    host.append(t1 - t0); wall.append(t2 - t0)
print("graph.replay() host time %.3f ms (min %.3f); replay + sync wall %.3f ms" % (1e3 * sum(host) / len(host), 1e3 * min(host), 1e3 * sum(wall) / len(wall)))
