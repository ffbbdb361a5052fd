#!/usr/bin/env python
"""bench.py -- images/sec of the SHOT-VAE training step on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

One step = the whole loop body of main_shot_vae.py:280-366 on one batch of synthetic CIFAR-shaped
input already resident in HBM: forwards (1)-(4), both backwards, (all-reduce,) SGD.  Workload at every
N: BASELINE.json configs[1] per GPU (WRN-28-2, K=10, B_l=B_u=512, bf16) -> weak scaling; the only
data-path collective is one RCCL all-reduce of the flat gradient buffer per step.
Prints ONE JSON line (rank 0)."""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch                      # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0             # MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--net", default="wideresnet-28-2")
    ap.add_argument("--classes", type=int, default=10)
    ap.add_argument("--batch", type=int, default=512, help="B_l = B_u per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--prof-steps", type=int, default=3)
    ap.add_argument("--graph", type=int, default=1, help="1 (default): replay the step from a captured hipGraph")
    ap.add_argument("--overlap", type=int, default=1,
                    help="1 (default): labelled / unlabelled branches of the step on two HIP streams; 0: one stream")
    return ap.parse_args()


def cpu_baseline(net, K):
    """The oracle (a CPU port of the reference step, golden-pinned to the reference) on the host cores,
    bounded sample: B_l=B_u=64, 1 warm-up + 2 timed steps."""
    from oracle import shotvae_oracle as O
    B = 64
    torch.manual_seed(1)
    st = O.default_init(net, K=K, seed=1)
    for k in st:
        if O.is_param(k):
            st[k].requires_grad_(True)
    il, ll, iu = torch.rand(B, 3, 32, 32), torch.randint(0, K, (B,)), torch.rand(B, 3, 32, 32)
    sch, mom = O.schedule(10), {}
    times = []
    for s in range(3):
        nz = O.make_noise(B, B, K, seed=s)
        t0 = time.time()
        O.train_step(st, net, il, ll, iu, nz, sch)
        O.sgd_step(st, mom)
        times.append(time.time() - t0)
    t = sum(times[1:]) / 2
    return {"value": round(2 * B / t, 2), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%s B_l=B_u=%d fp32 torch-CPU oracle, 2 timed steps (%.2f s/step)" % (net, B, t)}


def pmc_traffic(tag, a):
    """HBM bytes per launch of kernel `tag` from the PMC counters.  A counter pass cannot run inside this process (and a
    GPU-initialised process must not start a profiler), so the per-launch traffic of the conv-like kernels at the
    headline shapes is collected by tools/pmc_traffic.py (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes,
    gfx950 correction applied) and committed under profiles/; it is deterministic for a given kernel and shape."""
    import glob
    if a.dtype != "bf16" or a.batch != 512:
        return None, "no PMC profile for this batch / dtype"
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
    if not files:
        return None, "profiles/*pmc_traffic.json missing"
    tab = json.load(open(files[-1]))
    if tag not in tab:
        return None, "kernel not in " + os.path.basename(files[-1])
    return tab[tag]["traffic_bytes"], "profiles/%s: %s" % (os.path.basename(files[-1]), tab[tag]["formula"])


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, "launch with --nproc-per-node %d (WORLD_SIZE=%d)" % (a.gpus, world)
    torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    if world > 1:
        # "nccl" is RCCL on ROCm.  SV_DIST_BACKEND=gloo lets the N > 1 code path be exercised on a one-GPU box (both
        # ranks on the same device, the collective through host memory): a functional check, not a measurement.
        dist.init_process_group(os.environ.get("SV_DIST_BACKEND", "nccl"), rank=rank, world_size=world)

    import shot_vae_amd as S
    from shot_vae_amd import _lib as L
    from shot_vae_amd import dp

    K, B = a.classes, a.batch
    torch.manual_seed(1)
    model = S.VariationalAutoEncoder(a.net, num_input_channels=3, img_size=(32, 32), data_parallel=True,
                                     continuous_latent_dim=128, disc_latent_dim=K, small_input=True,
                                     compute_dtype=a.dtype, rng="device").cuda().train()
    if world > 1:
        dp.broadcast_parameters(model._engine.param, model._engine.bufs)
    elbo, cls = S.VAECriterion(discrete_dim=K, bce_reconstruction=True).cuda(), S.ClsCriterion()
    opt = S.FlatSGD(model, lr=0.02, momentum=0.9, weight_decay=5e-4)          # epoch-0 warm-up lr (:223-225)
    opt.zero_grad()
    sch = S.schedule(10)
    g = torch.Generator(device="cuda").manual_seed(1234 + rank)
    il = torch.rand(B, 3, 32, 32, device="cuda", generator=g)
    iu = torch.rand(B, 3, 32, 32, device="cuda", generator=g)
    ll = torch.randint(0, K, (B,), device="cuda", generator=g)

    from shot_vae_amd.train import GraphedTrainStep, train_step_overlapped

    graphed, graph_note = None, "eager"
    if a.graph and a.overlap:
        try:
            graphed = GraphedTrainStep(model, elbo, cls, opt, il, ll, iu, sch, distributed=world > 1)
            graph_note = "hipGraph replay"
        except Exception as e:        # capture unsupported on this stack: run eagerly, say so in the output
            graphed, graph_note = None, "eager (graph capture failed: %s)" % type(e).__name__
            torch.cuda.synchronize()

    def step():
        if graphed is not None:
            return graphed()
        if a.overlap:
            return train_step_overlapped(model, elbo, cls, opt, il, ll, iu, sch, distributed=world > 1)
        return S.train_step(model, elbo, cls, opt, il, ll, iu, sch, distributed=world > 1)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ls, lu = step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    assert torch.isfinite(ls).all() and torch.isfinite(lu).all(), "non-finite loss"
    images = 2 * B * world * a.steps
    headline = a.net == "wideresnet-28-2" and K == 10 and B == 512
    metric = "images/sec/step WRN-28-2 SHOT-VAE CIFAR-10 bs512" if headline else \
        "images/sec/step %s SHOT-VAE K=%d bs%d (not the BASELINE.json headline config)" % (a.net, K, B)
    out = {"metric": metric, "value": round(images / dt, 1),
           "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": round(1000 * dt / a.steps, 3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
           "config": {"workload": "SHOT-VAE train step (4 fwd + 2 bwd + SGD) %s K=%d ldc=128, B_l=B_u=%d per GPU, "
                                  "synthetic 3x32x32 in HBM, random init" % (a.net, K, B),
                      "global_batch": 2 * B * world, "parallelism": "dp%d" % world,
                      "schedule": "two-stream (labelled || unlabelled branch)" if a.overlap else "single stream",
                      "launch": graph_note,
                      "collective": "1 RCCL all-reduce of the flat fp32 gradient buffer per step" if world > 1 else "none"},
           "loss_sup": round(float(ls), 5), "loss_unsup": round(float(lu), 5)}

    # ---- roofline of the dominant kernel: HIP events around every conv-like launch (separate pass) ----
    if not a.no_roofline:
        eng = model._engine
        eng.prof_tags, eng.prof_cost = {}, {}
        L.lib().sv_prof_enable(1)
        for _ in range(a.prof_steps):       # eager, single stream: HIP events bracket every conv-like launch
            S.train_step(model, elbo, cls, opt, il, ll, iu, sch, distributed=world > 1)
        ntag = len(eng.prof_tags) + 1
        ms = (ctypes.c_double * ntag)()
        cnt = (ctypes.c_int * ntag)()
        L.lib().sv_prof_collect(ntag, ms, cnt)
        L.lib().sv_prof_enable(0)
        names = {v: k for k, v in eng.prof_tags.items()}
        rows = []
        for i in range(ntag):
            if cnt[i] and i in names and names[i] in eng.prof_cost:
                nbytes, flops = eng.prof_cost[names[i]]
                rows.append(dict(name=names[i], total_ms=ms[i], launches=cnt[i], avg_us=1000 * ms[i] / cnt[i],
                                 bytes=nbytes, flops=flops))
        tot = sum(r["total_ms"] for r in rows)
        if rows:
            d = max(rows, key=lambda r: r["total_ms"])
            ai = d["flops"] / d["bytes"]
            peak_t = MFMA_PEAK_TFLOPS[a.dtype]
            if ai < peak_t * 1e12 / (HBM_PEAK_GBS * 1e9):
                ach = d["bytes"] / (d["avg_us"] * 1e-6) / 1e9
                roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 4)}
            else:
                ach = d["flops"] / (d["avg_us"] * 1e-6) / 1e12
                roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak_t, "unit": "TFLOP/s",
                        "frac": round(ach / peak_t, 4)}
            traffic, traffic_src = pmc_traffic(d["name"], a)
            roof.update(traffic=traffic, traffic_source=traffic_src, kernel=d["name"], avg_us=round(d["avg_us"], 2), launches_per_step=d["launches"] // a.prof_steps,
                        algorithmic_bytes=d["bytes"], algorithmic_flops=d["flops"],
                        share_of_conv_kernel_time=round(d["total_ms"] / tot, 3),
                        conv_kernel_ms_per_step=round(tot / a.prof_steps, 3))
            out["roofline"] = roof
            if rank == 0 and os.environ.get("SV_BENCH_TABLE"):
                for r in sorted(rows, key=lambda r: -r["total_ms"]):
                    print("# %-28s %6d launches  avg %9.2f us  total %8.3f ms/step  %7.1f GB/s  %7.2f TFLOP/s" % (
                        r["name"], r["launches"] // a.prof_steps, r["avg_us"], r["total_ms"] / a.prof_steps,
                        r["bytes"] / r["avg_us"] / 1e3, r["flops"] / r["avg_us"] / 1e6), file=sys.stderr)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a.net, K)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
