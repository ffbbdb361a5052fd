"""The N > 1 code path on ONE MI355X: two ranks on the same device, torch.distributed over gloo (the collective through
host memory) -- a functional check of the data-parallel step (bench.py's own launcher: fresh rank processes, broadcast,
per-rank noise seeds, shared lambdas, ONE all-reduce of the flat gradient buffer, 1/world folded into the SGD kernel), not a
measurement.  RCCL itself needs one device per rank: the driver's multi-GPU run covers it."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json, torch
sys.path.insert(0, %r)
import torch.distributed as dist
import shot_vae_amd as S
from shot_vae_amd import dp
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.manual_seed(1 + 7 * rank)                 # ranks start from DIFFERENT weights: the broadcast must fix that
K, B = 10, 16
model = S.VariationalAutoEncoder("wideresnet-10-1", num_input_channels=3, img_size=(32, 32), data_parallel=True,
                                 continuous_latent_dim=128, disc_latent_dim=K, small_input=True, compute_dtype="fp32",
                                 rng="device").cuda().train()
dp.broadcast_parameters(model)
elbo, cls = S.VAECriterion(discrete_dim=K).cuda(), S.ClsCriterion()
opt = S.FlatSGD(model, lr=0.05)
opt.zero_grad()
torch.manual_seed(100 + rank); torch.cuda.manual_seed(100 + rank)       # per-rank data shard + noise
il, iu = torch.rand(B, 3, 32, 32, device="cuda"), torch.rand(B, 3, 32, 32, device="cuda")
ll = torch.randint(0, K, (B,), device="cuda")
rng = S.DeviceRng("cuda", seed=0)               # same seed everywhere: all ranks agree on the mixup lambdas
import numpy as np
np.random.seed(5)                               # (the sequential step draws its lambdas from numpy: same on every rank)
SCHEDULE = %r
losses = []
for step in range(3):
    dmode = "bucketed" if step else True        # step 0: one all-reduce; then the decoder-first buckets
    if SCHEDULE == "grouped":
        ls, lu = S.train_step_grouped(model, elbo, cls, opt, il, ll, iu, S.schedule(10), distributed=dmode, device_rng=rng)
    else:                                       # the reference's order: four autograd nodes, two backward() calls
        ls, lu = S.train_step(model, elbo, cls, opt, il, ll, iu, S.schedule(10), distributed=dmode)
    losses.append((float(ls), float(lu)))
p = model._engine.param.detach().cpu()
gathered = [torch.zeros_like(p) for _ in range(world)]
dist.all_gather(gathered, p)
lam = [float(rng.lam_l[i]) for i in range(3)]
if rank == 0:
    print(json.dumps({"max_param_diff": float((gathered[0] - gathered[1]).abs().max()), "losses": losses,
                      "finite": bool(torch.isfinite(p).all()), "param_norm": float(p.norm()), "lam": lam}))
dist.barrier()
dist.destroy_process_group()
'''


def _run_two_ranks(script, port0=29600, world=2):
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port0 + os.getpid() % 300), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=560) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    line = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("schedule", ["grouped", "sequential"])
def test_two_ranks_on_one_gpu_keep_identical_parameters(tmp_path, schedule):
    """(sequential: autograd visits forward (4) -- no decoder gradient -- before forward (3); the decoder bucket must not be
    reduced before (3)'s decoder backward has been issued, or the replicas keep rank-local decoder gradients and diverge)"""
    script = tmp_path / "worker.py"
    script.write_text(WORKER % (ROOT, schedule))
    res = _run_two_ranks(script)
    assert res["finite"] and res["param_norm"] > 0
    # the ranks saw different data and noise, started from different weights, and still hold the same parameters
    assert res["max_param_diff"] == 0.0, res
    assert all(abs(a) < 1e3 for pair in res["losses"] for a in pair)


STRAGGLER_WORKER = r'''
import os, sys, json, time, torch
sys.path.insert(0, %r)
import torch.distributed as dist
import shot_vae_amd as S
from shot_vae_amd import dp
from shot_vae_amd import _lib as L
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.manual_seed(1 + 7 * rank)
K, B = 10, 16
model = S.VariationalAutoEncoder("wideresnet-10-1", num_input_channels=3, img_size=(32, 32), data_parallel=True,
                                 continuous_latent_dim=128, disc_latent_dim=K, small_input=True, compute_dtype="bf16",
                                 rng="device").cuda().train()
dp.broadcast_parameters(model)
elbo, cls = S.VAECriterion(discrete_dim=K).cuda(), S.ClsCriterion()
opt = S.FlatSGD(model, lr=0.05)
opt.zero_grad()
torch.manual_seed(100 + rank); torch.cuda.manual_seed(100 + rank)
il, iu = torch.rand(B, 3, 32, 32, device="cuda"), torch.rand(B, 3, 32, 32, device="cuda")
ll = torch.randint(0, K, (B,), device="cuda")
rng = S.DeviceRng("cuda", seed=0)
eng = model._engine
real_backward = eng.backward
stalls = []
def slow_backward(*a, **k):
    # the straggler: host AND device stall between this rank's forward and its backward -- the other rank has long issued its
    # backward, fired its decoder bucket and waits in the collective; this rank's side stream forks by start signal behind the stall
    time.sleep(0.4)
    torch.cuda._sleep(int(3e8))
    stalls.append(time.time())
    return real_backward(*a, **k)
for step in range(4):
    eng.backward = slow_backward if rank == (step %% world) else real_backward      # the ranks take turns
    t0 = time.time()
    ls, lu = S.train_step_grouped(model, elbo, cls, opt, il, ll, iu, S.schedule(10), distributed="bucketed", device_rng=rng)
    torch.cuda.synchronize()
p = eng.param.detach().cpu()
gathered = [torch.zeros_like(p) for _ in range(world)]
dist.all_gather(gathered, p)
timeouts = torch.tensor([float(L.lib().sv_flag_timeouts())])
dist.all_reduce(timeouts)
if rank == 0:
    print(json.dumps({"max_param_diff": float((gathered[0] - gathered[1]).abs().max()), "finite": bool(torch.isfinite(p).all()),
                      "param_norm": float(p.norm()), "flag_timeouts": float(timeouts), "stalls": len(stalls),
                      "flag_fork": bool(eng.flag_fork)}))
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.timeout(600)
def test_two_ranks_with_a_straggler_keep_identical_parameters(tmp_path):
    """ADVICE r04 / VERDICT r05: the decoder-first bucket and the start-signal fork under a STRAGGLER.  The ranks take turns stalling
    (host sleep + device spin) between their forward and their backward, four bucketed steps in bf16: the early rank sits in the
    decoder bucket's collective with its encoder backward still running, the late rank's side stream waits for start signals behind
    a stalled main queue.  No wait may time out (the fail-closed counter stays 0 on both ranks), the replicas must hold bit-identical
    parameters."""
    script = tmp_path / "straggler.py"
    script.write_text(STRAGGLER_WORKER % ROOT)
    res = _run_two_ranks(script, port0=29950)
    assert res["finite"] and res["param_norm"] > 0 and res["stalls"] == 2
    assert res["flag_timeouts"] == 0.0, res
    assert res["max_param_diff"] == 0.0, res


EQUIV_WORKER = r'''
import os, sys, json, torch
sys.path.insert(0, %r)
import torch.distributed as dist
import shot_vae_amd as S
from shot_vae_amd import dp
from shot_vae_amd import _lib as L
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
K, B = 10, 16
L.call("sv_set_option", L.OPT_DETERMINISTIC, 1)          # fixed summation order: the comparison is down to the exchange


def make():
    torch.manual_seed(3)
    m = S.VariationalAutoEncoder("wideresnet-10-1", num_input_channels=3, img_size=(32, 32), data_parallel=True,
                                 continuous_latent_dim=128, disc_latent_dim=K, small_input=True, compute_dtype="fp32",
                                 rng="device").cuda().train()
    o = S.FlatSGD(m, lr=0.05, momentum=0.9, weight_decay=5e-4)
    o.zero_grad()
    return m, o


def shard_inputs(r):
    """data shard + noise stream of rank r (what the DP worker seeds itself with)"""
    torch.manual_seed(100 + r); torch.cuda.manual_seed(100 + r)
    il, iu = torch.rand(B, 3, 32, 32, device="cuda"), torch.rand(B, 3, 32, 32, device="cuda")
    return il, torch.randint(0, K, (B,), device="cuda"), iu


elbo, cls = S.VAECriterion(discrete_dim=K).cuda(), S.ClsCriterion()
sch = S.schedule(10)
# ---- the data-parallel run: this rank's shard, one all-reduce, 1/world in the SGD kernel, two steps ------------------------
model, opt = make()
dp.broadcast_parameters(model)
il, ll, iu = shard_inputs(rank)
rng = S.DeviceRng("cuda", seed=0)
for step in range(2):
    S.train_step_grouped(model, elbo, cls, opt, il, ll, iu, sch, distributed=True, device_rng=rng)
p_dp = model._engine.param.detach().clone()
bufs_dp = model._engine.bufs.detach().clone()
res = None
if rank == 0:
    # ---- ONE process over the concatenated shards, each shard with its own BatchNorm statistics: the shards' steps
    #      accumulate into the flat gradient buffer (no update in between), then one SGD step on the mean ----------------
    ref, ropt = make()
    rng2 = S.DeviceRng("cuda", seed=0)
    gens = {}
    for step in range(2):
        lams = rng2.next_lams()                     # both shards of a step use the SAME pair, like the ranks do
        snap = ref._engine.bufs.detach().clone()
        for r in range(world):
            if step == 0:
                gens[r] = shard_inputs(r) + (torch.cuda.get_rng_state(),)
            il_r, ll_r, iu_r, state = gens[r]
            torch.cuda.set_rng_state(state)         # continue rank r's device noise stream where its last step left it

            class Fixed:                            # DeviceRng stand-in: this step's pair, for every shard
                def next_lams(self):
                    return lams
            if r > 0:
                ref._engine.bufs.copy_(snap)        # running statistics are rank-local: rank 0's are compared below
            S.train_step_grouped(ref, elbo, cls, None, il_r, ll_r, iu_r, sch, device_rng=Fixed())
            if r == 0:
                bufs0 = ref._engine.bufs.detach().clone()
            gens[r] = (il_r, ll_r, iu_r, torch.cuda.get_rng_state())
        ref._engine.bufs.copy_(bufs0)
        ropt.step(grad_scale=1.0 / world)
        ropt.zero_grad()
    p_ref = ref._engine.param.detach()
    d = (p_dp - p_ref).abs().max() / p_ref.abs().max()
    db = (bufs_dp - ref._engine.bufs).abs().max() / ref._engine.bufs.abs().max()
    moved = (p_dp - make()[0]._engine.param).abs().max()
    res = {"rel_param_diff": float(d), "rel_buf_diff": float(db), "moved": float(moved)}
    print(json.dumps(res))
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 8])
def test_two_rank_step_equals_single_process_over_both_shards(tmp_path, world):
    """SURVEY.md 4, tier 5: an N-rank data-parallel run == ONE process running the shards as separate BatchNorm groups of
    the same weights and stepping on the mean gradient.  N = 2 and N = 8 ranks (gloo, all on the one GPU: the world size of
    BASELINE configs 3 / 5), two steps with momentum and weight decay; rank 0 then replays every shard itself (same data, same
    device noise streams, same lambdas, gradients accumulated in the flat buffer, `FlatSGD.step(1 / world)`) and compares
    parameters and its BatchNorm running statistics."""
    script = tmp_path / "equiv.py"
    script.write_text(EQUIV_WORKER % ROOT)
    res = _run_two_ranks(script, port0=29950, world=world)
    assert res["moved"] > 1e-4, res                        # the steps did change the weights
    assert res["rel_param_diff"] < 1e-5, res               # (sum order of the exchange vs in-buffer accumulation)
    assert res["rel_buf_diff"] < 1e-5, res


@pytest.mark.timeout(900)
def test_bench_two_ranks_gloo_on_one_gpu():
    """bench.py --gpus 2 started WITHOUT torchrun (its own launcher), both ranks on the one GPU over gloo, strong scaling."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SV_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--batch", "64", "--scaling", "strong", "--net", "wideresnet-10-1", "--no-cpu-baseline",
                        "--no-roofline"], capture_output=True, text=True, env=env, timeout=850)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["value"] > 0
    assert out["config"]["global_batch"] == 128                 # 2 loaders x 64 images in total, 32 per rank and loader
    # the lambda contract through the BENCH path: both ranks used the same mixup coefficients
    assert out["config"]["lambda_equal_across_ranks"] is True
    # --scaling strong probes both launch modes and reports both
    assert set(out["config"]["launch_probe"]) == {"eager_ms", "graph_ms"}


@pytest.mark.timeout(1200)
def test_bench_eight_ranks_gloo_on_one_gpu_headline_network():
    """The world size of BASELINE configs[2] without an 8-GPU node: bench.py --gpus 8 --scaling strong on the HEADLINE network
    (WRN-28-2, global B_l = B_u = 512 -> 64 per rank and loader), all eight ranks on the one GPU over gloo: its own launcher,
    the sharding arithmetic, the lambda contract, one all-reduce per step and the 1/8 in the SGD kernel at N = 8.  A functional
    check -- eight processes time-slice one device, the images/s figure means nothing."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SV_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1",
                        "--batch", "512", "--scaling", "strong", "--no-cpu-baseline", "--no-roofline", "--no-extras", "--graph", "0"],
                       capture_output=True, text=True, env=env, timeout=1100)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 8 and out["scaling"] == "strong" and out["value"] > 0
    assert out["config"]["global_batch"] == 1024 and out["config"]["parallelism"] == "dp8"
    assert "B_l=B_u=64 per GPU" in out["config"]["workload"]
    assert out["config"]["lambda_equal_across_ranks"] is True
    assert out["config"]["collective"].startswith("1 RCCL all-reduce")       # (the gloo stand-in of it here)
    import math
    assert math.isfinite(out["loss_sup"]) and math.isfinite(out["loss_unsup"])


@pytest.mark.timeout(900)
def test_bench_svhn_workload_two_ranks_gloo_on_one_gpu():
    """bench.py --workload svhn --gpus 2 (BASELINE config 5's data-parallel iteration: per-rank shard, graph of forwards +
    backward, ONE all-reduce of FlatAdam's flat gradient buffer, sv_adam with 1/world) on one GPU over gloo."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SV_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "svhn", "--gpus", "2", "--steps", "3",
                        "--warmup", "2", "--batch", "64"], capture_output=True, text=True, env=env, timeout=850)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["value"] > 0 and "svhn_VAE" in out["metric"]
    assert out["config"]["global_batch"] == 256 and set(out["config"]["launch_probe"]) == {"eager_ms", "graph_ms"}


def _single_rank_rccl_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SV_DIST_BACKEND")}
    env.update({"SV_DP_SINGLE_RANK": "1", "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                "MASTER_PORT": str(29500 + os.getpid() % 2000), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    return env


@pytest.mark.timeout(900)
@pytest.mark.parametrize("allreduce", ["bucketed", "single"])
def test_bench_headline_step_through_rccl_single_rank(allreduce):
    """The data-parallel step of BASELINE config 2 at FULL size with its collectives sent through RCCL on the hardware: one
    rank (a one-GPU box cannot hold two RCCL ranks), backend "nccl", the decoder-first bucket on the communication stream
    under the encoder's backward + the encoder bucket after it (or the single all-reduce).  With one rank the sums are the
    identity, so the step must give finite losses of the usual size at (nearly) the usual speed: what is checked is that
    process-group start-up, the stream / event edges around the asynchronous work handle and the 1/world path of the
    optimizer run on a real device."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "10", "--warmup", "3",
                        "--no-cpu-baseline", "--no-roofline", "--no-extras", "--allreduce", allreduce],
                       capture_output=True, text=True, env=_single_rank_rccl_env(), timeout=850)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["value"] > 0
    assert "RCCL" in out["config"]["collective"]
    assert ("two buckets" in out["config"]["collective"]) == (allreduce == "bucketed")
    assert out["config"]["lambda_equal_across_ranks"] is True
    assert 0 < out["loss_sup"] < 1e4 and 0 < abs(out["loss_unsup"]) < 1e4
    assert out["ms_per_step"] < 14.0, out["ms_per_step"]       # 8.0 ms without the collectives


@pytest.mark.timeout(900)
@pytest.mark.parametrize("batch", [512, 64])
def test_bench_graph_replay_next_to_rccl_single_rank(batch):
    """Strong-scaling readiness: the step's hipGraph is captured BEFORE the RCCL process group is created (no stream capture
    next to its watchdog), the group is created, parameters / optimizer state are broadcast, and the graph is then REPLAYED
    with the all-reduce + sv_sgd launch following eagerly -- on one rank here (a one-GPU box), at the headline batch and at
    the 64 images per rank of an 8-GPU strong-scaling run, where the eager step is host-bound."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "10", "--warmup", "3",
                        "--batch", str(batch), "--graph", "1", "--no-cpu-baseline", "--no-roofline", "--no-extras"],
                       capture_output=True, text=True, env=_single_rank_rccl_env(), timeout=850)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["config"]["launch"] == "hipGraph replay", out["config"]
    assert "RCCL" in out["config"]["collective"] and out["value"] > 0
    assert 0 < out["loss_sup"] < 1e4 and 0 < abs(out["loss_unsup"]) < 1e4


@pytest.mark.timeout(900)
def test_bench_svhn_through_rccl_single_rank():
    """BASELINE config 5's data-parallel iteration (graph of forwards + backward, one all-reduce of FlatAdam's flat gradient
    buffer, sv_adam with 1/world) with the all-reduce sent through RCCL: one rank, backend "nccl", B = 1024."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "svhn", "--gpus", "1", "--steps", "10",
                        "--warmup", "3", "--batch", "1024"], capture_output=True, text=True, env=_single_rank_rccl_env(),
                       timeout=850)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["value"] > 0 and "RCCL" in out["config"]["collective"]
    assert out["loss"] == out["loss"] and abs(out["loss"]) < 1e7
