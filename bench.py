#!/usr/bin/env python
"""bench.py -- images/sec of the SHOT-VAE training step on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

One step = the whole loop body of main_shot_vae.py:280-366 on one batch of synthetic CIFAR-shaped
input already resident in HBM: forwards (1)-(4), both backwards, (all-reduce,) SGD.  Workload:
BASELINE.json configs[1] (WRN-28-2, K=10, B_l=B_u=512, bf16) per GPU (--scaling weak, default) or as the
GLOBAL batch split over the ranks (--scaling strong: 512/N per loader per rank, BASELINE configs[2]);
the only data-path collective is one RCCL all-reduce of the flat gradient buffer per step.
`python bench.py --gpus N` without torchrun starts its own N ranks (fresh processes, one per GPU).
Prints ONE JSON line (rank 0)."""
import argparse
import ctypes
import json
import os
import socket
import subprocess
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
    ap.add_argument("--graph", type=int, default=0,
                    help="1: replay the step from a captured hipGraph (one stream); 0 (default): issue it eagerly, weight "
                         "gradients on a side stream -- the grouped step is ~280 launches, which the host keeps ahead of")
    ap.add_argument("--schedule", default="grouped", choices=["grouped", "two-stream", "sequential"],
                    help="grouped (default): the four forwards of the step as one batched launch sequence; two-stream: the "
                         "labelled / unlabelled branches on two HIP streams; sequential: the reference's order, one stream")
    ap.add_argument("--wgrad-side", type=int, default=1, help="weight gradients on a side stream (off the critical path)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher check without a GPU: the ranks rendezvous over gloo, agree on a max-reduced time and "
                         "rank 0 prints a JSON stub (tests/test_bench_launch_cpu.py)")
    ap.add_argument("--disable", type=int, default=0,
                    help="dispatcher mask SV_OPT_DISABLE_MASK (A/B runs of the specialised kernels; 0 = all enabled)")
    ap.add_argument("--persistent-blocks", type=int, default=0,
                    help="SV_OPT_PERSISTENT_BLOCKS: block budget of the persistent narrow kernels (0 = library default, 512)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): --batch per loader PER GPU; strong: --batch per loader in total, split over the GPUs")
    return ap.parse_args()


def spawn_ranks(n):
    """`python bench.py --gpus N` outside torchrun: start N fresh rank processes (before this process touches the GPU),
    one per device, torchrun-style environment, rank 0's JSON line passed through; non-zero exit if any rank fails."""
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = procs[0].communicate()[0]
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.stderr.write("bench.py: ranks failed (rank, exit code): %s\n" % bad)
        sys.exit(1)
    sys.exit(0)


def cpu_baseline(net, K):
    """The oracle (a CPU port of the reference step, golden-pinned to the reference) on the host cores, bounded sample
    (BASELINE.md 4): B_l=B_u=64; thread count chosen by a one-step sweep over {8,16,32,64} (all cores of a big host
    oversubscribe torch's intra-op pool: 128 threads ran 3.6x slower than 8), then 2 warm-up + 5 timed steps at the best
    count; one timed step on ONE thread; 1 warm-up + 2 timed steps at B=256."""
    from oracle import shotvae_oracle as O
    ncpu = os.cpu_count() or 1
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass

    def run(B, threads, warm, timed):
        torch.set_num_threads(threads)
        torch.manual_seed(1)
        st = O.default_init(net, K=K, seed=1)
        for k in st:
            if O.is_param(k):
                st[k].requires_grad_(True)
        il, ll, iu = torch.rand(B, 3, 32, 32), torch.randint(0, K, (B,)), torch.rand(B, 3, 32, 32)
        sch, mom, times = O.schedule(10), {}, []
        for s in range(warm + timed):
            nz = O.make_noise(B, B, K, seed=s)
            t0 = time.perf_counter()
            O.train_step(st, net, il, ll, iu, nz, sch)
            O.sgd_step(st, mom)
            times.append(time.perf_counter() - t0)
        return sum(times[warm:]) / timed

    B = 64
    cands = sorted({t for t in (8, 16, 32, 64) if t <= ncpu} or {ncpu})
    sweep = {t: run(B, t, 1, 1) for t in cands}
    best = min(sweep, key=sweep.get)
    t_best = run(B, best, 2, 5)
    t_one = run(B, 1, 0, 1)
    t_256 = run(256, best, 1, 2)
    return {"value": round(2 * B / t_best, 2), "unit": "images/s", "cores": best, "kind": "port",
            "sample": "%s B_l=B_u=%d fp32 torch-CPU oracle, %d threads (best of one-step sweep %s), 2 warm-up + 5 timed "
                      "steps (%.2f s/step)" % (net, B, best, {t: round(2 * B / v, 1) for t, v in sweep.items()}, t_best),
            "one_thread_images_per_s": round(2 * B / t_one, 2),
            "b256_images_per_s": round(512 / t_256, 2), "b256_s_per_step": round(t_256, 2),
            "host": {"cpu_model": model, "os_cpu_count": ncpu}}


def pmc_traffic(tag, a):
    """HBM bytes per launch of kernel `tag` from the PMC counters.  A counter pass cannot run inside this process (and a
    GPU-initialised process must not start a profiler), so the per-launch traffic of the conv-like kernels at the
    headline shapes is collected by tools/pmc_traffic.py (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes,
    gfx950 correction applied) and committed under profiles/; it is deterministic for a given kernel and shape."""
    import glob
    if a.dtype != "bf16" or a.batch != 512 or a.schedule != "grouped" or a.scaling != "weak":
        return None, "no PMC profile for this batch / dtype / schedule"
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
    if not files:
        return None, "profiles/*pmc_traffic.json missing"
    tab = json.load(open(files[-1]))
    if tag not in tab:
        return None, "kernel not in " + os.path.basename(files[-1])
    return tab[tag]["traffic_bytes"], "profiles/%s: %s" % (os.path.basename(files[-1]), tab[tag]["formula"])


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        spawn_ranks(a.gpus)                       # does not return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node %d, or without torchrun)"
                 % (a.gpus, world, a.gpus))
    if a.dry_run:
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.barrier()
            dist.destroy_process_group()
        if a.scaling == "strong" and a.batch % world:
            sys.exit("bench.py: --scaling strong needs --batch (%d) divisible by --gpus (%d)" % (a.batch, world))
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "max_over_ranks": float(t), "scaling": a.scaling,
                              "per_rank_batch": a.batch // world if a.scaling == "strong" else a.batch}))
        return
    ndev = torch.cuda.device_count()
    if world > ndev and os.environ.get("SV_DIST_BACKEND", "nccl") == "nccl":
        sys.exit("bench.py: %d ranks but %d visible GPUs (RCCL needs one device per rank)" % (world, ndev))
    torch.cuda.set_device(local % max(ndev, 1))
    if world > 1:
        # "nccl" is RCCL on ROCm.  SV_DIST_BACKEND=gloo lets the N > 1 code path be exercised on a one-GPU box (both
        # ranks on the same device, the collective through host memory): a functional check, not a measurement.
        dist.init_process_group(os.environ.get("SV_DIST_BACKEND", "nccl"), rank=rank, world_size=world)

    import shot_vae_amd as S
    from shot_vae_amd import _lib as L
    from shot_vae_amd import dp
    if a.disable:
        L.call("sv_set_option", L.OPT_DISABLE_MASK, a.disable)
    if a.persistent_blocks:
        L.call("sv_set_option", L.OPT_PERSISTENT_BLOCKS, a.persistent_blocks)

    K = a.classes
    if a.scaling == "strong":
        if a.batch % world:
            sys.exit("bench.py: --scaling strong needs --batch (%d) divisible by --gpus (%d)" % (a.batch, world))
        B = a.batch // world              # the global batch of 2 x --batch images is split over the ranks
    else:
        B = a.batch
    torch.manual_seed(1)                  # identical initial weights on every rank (and a broadcast below)
    model = S.VariationalAutoEncoder(a.net, num_input_channels=3, img_size=(32, 32), data_parallel=True,
                                     continuous_latent_dim=128, disc_latent_dim=K, small_input=True,
                                     compute_dtype=a.dtype, rng="device").cuda().train()
    if world > 1:
        dp.broadcast_parameters(model)
    # per-rank noise streams (eps, Gumbel u, pairings): seed + rank; the mixup coefficients come from DeviceRng tables
    # seeded identically on every rank, so all ranks agree on lambda
    torch.manual_seed(1 + rank)
    torch.cuda.manual_seed(1 + rank)
    elbo, cls = S.VAECriterion(discrete_dim=K, bce_reconstruction=True).cuda(), S.ClsCriterion()
    opt = S.FlatSGD(model, lr=0.02, momentum=0.9, weight_decay=5e-4)          # epoch-0 warm-up lr (:223-225)
    opt.zero_grad()
    sch = S.schedule(10)
    g = torch.Generator(device="cuda").manual_seed(1234 + rank)
    il = torch.rand(B, 3, 32, 32, device="cuda", generator=g)
    iu = torch.rand(B, 3, 32, 32, device="cuda", generator=g)
    ll = torch.randint(0, K, (B,), device="cuda", generator=g)

    from shot_vae_amd.train import GraphedTrainStep, train_step_grouped, train_step_overlapped
    model._engine.wgrad_side_stream = bool(a.wgrad_side)

    graphed, graph_note = None, "eager, weight gradients on a side stream"
    if a.graph and a.schedule != "sequential":
        try:
            graphed = GraphedTrainStep(model, elbo, cls, opt, il, ll, iu, sch, distributed=world > 1, schedule=a.schedule)
            graph_note = "hipGraph replay"
        except Exception as e:        # capture unsupported on this stack: run eagerly, say so in the output
            graphed, graph_note = None, "eager (graph capture failed: %s)" % type(e).__name__
            torch.cuda.synchronize()

    def step():
        if graphed is not None:
            return graphed()
        if a.schedule == "grouped":
            return train_step_grouped(model, elbo, cls, opt, il, ll, iu, sch, distributed=world > 1)
        if a.schedule == "two-stream":
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
    headline = a.net == "wideresnet-28-2" and K == 10 and a.batch == 512
    metric = "images/sec/step WRN-28-2 SHOT-VAE CIFAR-10 bs512" if headline else \
        "images/sec/step %s SHOT-VAE K=%d bs%d (not the BASELINE.json headline config)" % (a.net, K, B)
    out = {"metric": metric, "value": round(images / dt, 1),
           "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": round(1000 * dt / a.steps, 3), "higher_is_better": True, "scaling": a.scaling,
           "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
           "config": {"workload": "SHOT-VAE train step (4 fwd + 2 bwd + SGD) %s K=%d ldc=128, B_l=B_u=%d per GPU, "
                                  "synthetic 3x32x32 in HBM, random init" % (a.net, K, B),
                      "global_batch": 2 * B * world, "parallelism": "dp%d" % world,
                      "schedule": {"grouped": "grouped (forwards (1)-(4) as one batched launch sequence, 4 BatchNorm groups)",
                                   "two-stream": "two-stream (labelled || unlabelled branch)",
                                   "sequential": "sequential (reference order)"}[a.schedule],
                      "launch": graph_note,
                      "collective": "1 RCCL all-reduce of the flat fp32 gradient buffer per step" if world > 1 else "none"},
           "loss_sup": round(float(ls), 5), "loss_unsup": round(float(lu), 5)}

    # ---- roofline of the dominant kernel: HIP events around EVERY launch of the library (separate pass) ----
    if not a.no_roofline:
        eng = model._engine
        eng.prof_tags, eng.prof_cost = {}, {}
        L.prof_tags = eng.prof_tags
        L.lib().sv_prof_enable(1)
        eng.wgrad_side_stream = False
        for _ in range(a.prof_steps):       # eager, single stream: HIP events bracket every launch of the timed schedule
            if a.schedule == "grouped":
                train_step_grouped(model, elbo, cls, opt, il, ll, iu, sch, distributed=world > 1)
            else:
                S.train_step(model, elbo, cls, opt, il, ll, iu, sch, distributed=world > 1)
        ntag = len(eng.prof_tags) + 1
        ms = (ctypes.c_double * ntag)()
        cnt = (ctypes.c_int * ntag)()
        L.lib().sv_prof_collect(ntag, ms, cnt)
        L.lib().sv_prof_enable(0)
        tags = dict(eng.prof_tags)
        L.prof_tags = eng.prof_tags = None
        rows = []
        for name, i in tags.items():
            if not cnt[i]:
                continue
            nbytes, flops, nl = eng.prof_cost.get(name, (0.0, 0.0, 0))
            rows.append(dict(name=name, total_ms=ms[i], launches=cnt[i], avg_us=1000 * ms[i] / cnt[i],
                             bytes=nbytes / nl if nl else None, flops=flops / nl if nl else None))
        tot = sum(r["total_ms"] for r in rows)
        costed = [r for r in rows if r["bytes"]]
        if costed:
            d = max(costed, key=lambda r: r["total_ms"])
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
            roof.update(traffic=traffic, traffic_source=traffic_src, kernel=d["name"], avg_us=round(d["avg_us"], 2),
                        launches_per_step=d["launches"] // a.prof_steps,
                        algorithmic_bytes=d["bytes"], algorithmic_flops=d["flops"],
                        share_of_kernel_time=round(d["total_ms"] / tot, 3),
                        kernel_ms_per_step=round(tot / a.prof_steps, 3),
                        launches_per_step_all=sum(r["launches"] for r in rows) // a.prof_steps)
            # the five largest entries of the same pass, so that no large kernel stays invisible behind the dominant one
            top = sorted(rows, key=lambda r: -r["total_ms"])[:5]
            roof["top5"] = [{"kernel": r["name"], "ms_per_step": round(r["total_ms"] / a.prof_steps, 3),
                             "launches_per_step": r["launches"] // a.prof_steps, "avg_us": round(r["avg_us"], 2),
                             "GBps": round(r["bytes"] / r["avg_us"] / 1e3, 1) if r["bytes"] else None} for r in top]
            out["roofline"] = roof
            if rank == 0 and os.environ.get("SV_BENCH_TABLE"):
                for r in sorted(rows, key=lambda r: -r["total_ms"]):
                    print("# %-28s %6d launches  avg %9.2f us  total %8.3f ms/step  %9s GB/s  %9s TFLOP/s" % (
                        r["name"], r["launches"] // a.prof_steps, r["avg_us"], r["total_ms"] / a.prof_steps,
                        "%.1f" % (r["bytes"] / r["avg_us"] / 1e3) if r["bytes"] else "-",
                        "%.2f" % (r["flops"] / r["avg_us"] / 1e6) if r["flops"] else "-"), file=sys.stderr)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a.net, K)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
