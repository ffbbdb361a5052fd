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
    ap.add_argument("--graph", type=int, default=None,
                    help="1: replay the step from a captured hipGraph (one stream); 0: issue it eagerly, weight gradients on a "
                         "side stream (the grouped step is ~280 launches, which the host keeps ahead of at B = 512); -1: time "
                         "a few steps of both and keep the faster (both times go into the JSON line).  Default: 0 for "
                         "--scaling weak, -1 for --scaling strong (small per-rank batches are launch-bound)")
    ap.add_argument("--workload", default="shotvae", choices=["shotvae", "svhn"],
                    help="shotvae (default): the SHOT-VAE step of main_shot_vae.py:280-366 (BASELINE configs 2-4); svhn: one "
                         "smooth-ELBO iteration of svhn_VAE, main_smooth_ELBO_svhn.py:152-176 (config 5; --batch 1024)")
    ap.add_argument("--allreduce", default="single", choices=["bucketed", "single"],
                    help="N > 1: single (default, SURVEY.md 5.2) = ONE all-reduce of the flat gradient buffer after the backward; "
                         "bucketed = the decoder's 88 %% of the gradient bytes are all-reduced on a communication stream while the "
                         "encoder's backward runs, the encoder bucket after it (dp.DecoderFirstAllReduce) -- the documented "
                         "deviation, to be switched on once an 8-GPU run shows the single exchange is not hidden")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra keys of the headline line (config-4 step, WRN-28-10 layer table)")
    ap.add_argument("--schedule", default="grouped", choices=["grouped", "two-stream", "sequential"],
                    help="grouped (default): the four forwards of the step as one batched launch sequence; two-stream: the "
                         "labelled / unlabelled branches on two HIP streams; sequential: the reference's order, one stream")
    ap.add_argument("--wgrad-side", type=int, default=1, help="weight gradients on a side stream (off the critical path)")
    ap.add_argument("--pair-blocks", type=int, default=-1,
                    help="Engine.pair_blocks: block budget of each kernel of a paired weight / data gradient launch (-1 = the "
                         "engine's default, 256; 0 = both with the full budget)")
    ap.add_argument("--pair-blocks-strided", type=int, default=-1, help="block budget of the stride-2 units' first convolution pair (-1 = the engine's default)")
    ap.add_argument("--input-stream", type=int, default=0,
                    help="1: the input side of the grouped step (noise, pairings, mixed batches, layout change) on a stream of its "
                         "own, beside the previous step's backward (train_step_grouped(input_stream=True)); 0 (default): in front of "
                         "the first convolution on the main stream -- measured the same (7.22 vs 7.21 ms)")
    ap.add_argument("--compact-shortcut", type=int, default=1, help="0: the stride-2 shortcuts' data gradients in the strided (sparse_out) form")
    ap.add_argument("--fused-bwd", type=int, default=-1,
                    help="sv_bwd3x3, the one-launch backward of the 32-channel body convolutions: 0 = off (data / weight gradient pair + "
                         "sv_bn_bwd_apply), 1 = conv1 of the same-shape units with norm2's BatchNorm backward in its load path, 2 = also conv2 "
                         "of the unit in front with the unit boundary's BatchNorm backward + skip connection in its load path, 3 = every "
                         "other 32-channel conv2 too, -1 = the engine default")
    ap.add_argument("--event-fork-after-fused", type=int, default=1,
                    help="Engine.event_fork_after_fused: 0 = the pair behind fused-backward launches forks by flag like every other pair (the "
                         "side stream's wait_flag_kernel then spins through the fused launches: A/B)")
    ap.add_argument("--fold-bn-bwd", type=int, default=1, help="Engine.fold_bn_bwd (0: a sv_bn_bwd_affine launch in front of every fused backward: A/B)")
    ap.add_argument("--fused-channels", default="", help="Engine.fused_channels, comma-separated (default: the engine's; '32,64' adds the 64-channel stage)")
    ap.add_argument("--fused-blocks", type=int, default=0, help="Engine.fused_blocks: blocks of a fused-backward launch (0 = the engine's default, 248)")
    ap.add_argument("--flag-fork", type=int, default=1, help="0: event forks for the paired weight gradients instead of the start signal")
    ap.add_argument("--fork-every", type=int, default=0, help="weight gradients per side-stream fork (0 = the engine's default)")
    ap.add_argument("--light-fork", type=int, default=1,
                    help="0: the side-stream forks of the backward use ordinary events (system-scope fence) instead of sv_stream_fork's")
    ap.add_argument("--deterministic", type=int, default=0,
                    help="1: SV_OPT_DETERMINISTIC (fixed summation order everywhere: bit-reproducible steps; what it costs)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher check without a GPU: the ranks rendezvous over gloo, agree on a max-reduced time and "
                         "rank 0 prints a JSON stub (tests/test_bench_launch_cpu.py)")
    ap.add_argument("--disable", type=int, default=0,
                    help="dispatcher mask SV_OPT_DISABLE_MASK (A/B runs of the specialised kernels; 0 = all enabled)")
    ap.add_argument("--enable", type=int, default=0, help="dispatcher mask SV_OPT_ENABLE_MASK (kernels that are off by default)")
    ap.add_argument("--persistent-blocks", type=int, default=0,
                    help="SV_OPT_PERSISTENT_BLOCKS: block budget of the persistent narrow kernels (0 = library default, 512)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): --batch per loader PER GPU; strong: --batch per loader in total, split over the GPUs")
    return ap.parse_args()


def spawn_ranks(n):
    """`python bench.py --gpus N` outside torchrun: start N fresh rank processes (before this process touches the GPU; never
    a re-exec of a GPU-initialised process), one per device, torchrun-style environment, rank 0's JSON line passed through.
    Watchdog: every child is polled; as soon as one exits non-zero the others (by PID -- they may be sitting in the RCCL
    rendezvous waiting for it) are terminated and the launcher exits 1."""
    import tempfile
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    out0 = tempfile.TemporaryFile()
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    bad = []
    while True:
        rcs = [p.poll() for p in procs]
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad or all(rc is not None for rc in rcs):
            break
        time.sleep(0.2)
    if bad:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    out0.seek(0)
    sys.stdout.write(out0.read().decode())
    sys.stdout.flush()
    if bad:
        sys.stderr.write("bench.py: ranks failed (rank, exit code): %s; the remaining ranks were terminated\n" % bad)
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
    src = "profiles/%s: %s" % (os.path.basename(files[-1]), tab[tag]["formula"])
    if tab[tag].get("algorithmic_bytes"):
        # (the counter pass measures ONE form of the tag -- e.g. the forward WITH its residual operand -- while `algorithmic_bytes` of the
        #  timed pass averages the step's launches of that tag: the ratio that speaks of re-reads is traffic / this figure)
        src += "; the measured launch's own algorithmic bytes: %d (traffic / that = %.3f)" % (
            tab[tag]["algorithmic_bytes"], tab[tag]["traffic_bytes"] / tab[tag]["algorithmic_bytes"])
    return tab[tag]["traffic_bytes"], src


def _shot_setup(S, net, K, Bl, Bu, dtype="bf16", seed=4321, dmi=2.3):
    torch.manual_seed(1)
    model = S.VariationalAutoEncoder(net, num_input_channels=3, img_size=(32, 32), data_parallel=True,
                                     continuous_latent_dim=128, disc_latent_dim=K, small_input=True,
                                     compute_dtype=dtype, rng="device").cuda().train()
    elbo, cls = S.VAECriterion(discrete_dim=K, bce_reconstruction=True).cuda(), S.ClsCriterion()
    opt = S.FlatSGD(model, lr=0.02, momentum=0.9, weight_decay=5e-4)
    opt.zero_grad()
    g = torch.Generator(device="cuda").manual_seed(seed)
    il = torch.rand(Bl, 3, 32, 32, device="cuda", generator=g)
    iu = torch.rand(Bu, 3, 32, 32, device="cuda", generator=g)
    ll = torch.randint(0, K, (Bl,), device="cuda", generator=g)
    return model, elbo, cls, opt, S.schedule(10, dmi=dmi), il, ll, iu


def _time_ms(fn, warm, steps):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, out


def extras(S):
    """Extra keys of the headline line (same JSON object, the headline fields are untouched), so that they are in the
    driver-written record and not only under profiles/:
      config4                 the BASELINE configs[3] step (WRN-28-10, K = 100, B = 256)
      wrn28_10_layer_table    the per-kernel table behind the north-star bar ">= 40 % of the bf16 MFMA roofline on the WRN-28-10
                              encoder conv at batch 512" (0.2416 TFLOP per launch at B = 512, <= 242 us to pass)
      headline_variants       the headline network through the OTHER paths a real epoch takes: the ragged last labelled batch
                              (B_l = 416, main_shot_vae.py:280), --om, the sequential (reference-order, autograd) step, the
                              bit-reproducible deterministic mode, the fp32-operand parity mode
      per_rank_batch_table    single-GPU ms/step at the per-rank batches of a strong-scaling run (512 / N per loader), eager issue
                              and hipGraph replay: the per-rank efficiency curve that bounds the strong-scaling result
      config5                 BASELINE configs[4]: one smooth-ELBO iteration of svhn_VAE at B = 1024 per loader"""
    from shot_vae_amd.train import GraphedTrainStep, train_step_grouped
    res = {}
    net, K, B = "wideresnet-28-10", 100, 256
    model, elbo, cls, opt, sch, il, ll, iu = _shot_setup(S, net, K, B, B, dmi=4.6)
    warm, steps = 3, 8
    ms, (ls, lu) = _time_ms(lambda: train_step_grouped(model, elbo, cls, opt, il, ll, iu, sch), warm, steps)
    tflop_step = 32.43                     # 63.34 GFLOP per dataloader image x 512
    res["config4"] = {"workload": "SHOT-VAE train step %s K=%d B_l=B_u=%d bf16 (BASELINE configs[3]), grouped, eager" % (net, K, B),
                      "ms_per_step": round(ms, 3), "images_per_s": round(2 * B / ms * 1e3, 1), "steps": steps, "warmup": warm,
                      "TFLOPs": round(tflop_step / ms * 1e3, 1),
                      "frac_of_bf16_mfma_peak": round(tflop_step / ms * 1e3 / MFMA_PEAK_TFLOPS["bf16"], 4),
                      "finite": bool(torch.isfinite(ls).all() and torch.isfinite(lu).all())}
    del model, opt
    torch.cuda.empty_cache()
    res.update(layer_table())

    # ---- the headline network through the other paths of a real epoch -----------------------------------------------------
    net, K = "wideresnet-28-2", 10
    var = {}
    model, elbo, cls, opt, sch, il, ll, iu = _shot_setup(S, net, K, 512, 512)
    t, _ = _time_ms(lambda: train_step_grouped(model, elbo, cls, opt, il[:416], ll[:416], iu, sch), 3, 10)
    var["ragged_Bl416_Bu512"] = {"ms_per_step": round(t, 3), "images_per_s": round(928 / t * 1e3, 1),
                                 "schedule": "grouped: one launch sequence per loader, (1)(2) at 416 and (3)(4) at 512 images"}
    t, _ = _time_ms(lambda: train_step_grouped(model, elbo, cls, opt, il, ll, iu, sch, optimal_match=True), 3, 10)
    var["optimal_match"] = {"ms_per_step": round(t, 3), "images_per_s": round(1024 / t * 1e3, 1),
                            "schedule": "grouped: (1)(3)(2) batched, sv_optimal_match on the outputs of (3), then (4)"}
    t, _ = _time_ms(lambda: S.train_step(model, elbo, cls, opt, il, ll, iu, sch), 3, 10)
    var["sequential"] = {"ms_per_step": round(t, 3), "images_per_s": round(1024 / t * 1e3, 1),
                         "schedule": "the reference's order through autograd: 4 forwards, 2 backward() calls, one stream + side stream"}
    from shot_vae_amd import _lib as L_
    with L_.options(deterministic=1):
        t, _ = _time_ms(lambda: train_step_grouped(model, elbo, cls, opt, il, ll, iu, sch), 3, 10)
    var["deterministic_mode"] = {"ms_per_step": round(t, 3), "images_per_s": round(1024 / t * 1e3, 1),
                                 "note": "SV_OPT_DETERMINISTIC: every accumulation in a fixed order, two runs agree bit for bit"}
    # the DEFAULT mode twice from the same state, weights, inputs and noise seeds: what two runs of the timed path differ by
    # (round 4: cosine 0.968-0.983 -- fp32 atomics of the BatchNorm statistics; the accumulators are doubles since ABI 6)
    import numpy as np
    st0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    flats = []
    for rep in range(2):
        model.load_state_dict(st0)
        opt.zero_grad()
        torch.manual_seed(11)
        torch.cuda.manual_seed(11)
        np.random.seed(11)
        train_step_grouped(model, elbo, cls, None, il, ll, iu, sch)
        torch.cuda.synchronize()
        flats.append(model.flat_parameters()[1].detach().double().clone())
    var["default_mode_repeatability"] = {
        "flat_gradient_cosine": round(float(flats[0] @ flats[1] / flats[0].norm() / flats[1].norm()), 9),
        "flat_gradient_relative_l2_difference": float((flats[0] - flats[1]).norm() / flats[0].norm()),
        "note": "two runs of the timed (default) path from identical state and seeds"}
    del flats, st0
    del model, opt
    torch.cuda.empty_cache()
    model, elbo, cls, opt, sch, il, ll, iu = _shot_setup(S, net, K, 512, 512, dtype="fp32")
    t, _ = _time_ms(lambda: train_step_grouped(model, elbo, cls, opt, il, ll, iu, sch), 2, 5)
    var["fp32_parity_mode"] = {"ms_per_step": round(t, 3), "images_per_s": round(1024 / t * 1e3, 1),
                               "note": "fp32 operands on v_mfma_f32_16x16x4_f32: the mode that meets the 1e-3 parity gate"}
    del model, opt
    torch.cuda.empty_cache()
    res["headline_variants"] = var

    # ---- per-rank batch sizes of a strong-scaling run, on this one GPU -----------------------------------------------------
    rows = []
    for Br in (256, 128, 64):
        model, elbo, cls, opt, sch, il, ll, iu = _shot_setup(S, net, K, Br, Br)
        te, _ = _time_ms(lambda: train_step_grouped(model, elbo, cls, opt, il, ll, iu, sch), 3, 12)
        row = {"per_rank_batch": Br, "ranks_at_global_512": 512 // Br, "eager_ms": round(te, 3)}
        try:
            gs = GraphedTrainStep(model, elbo, cls, opt, il, ll, iu, sch)
            tg, _ = _time_ms(gs, 3, 12)
            row["graph_ms"] = round(tg, 3)
            del gs
        except Exception as e:
            row["graph_ms"] = None
            row["graph_error"] = type(e).__name__
            torch.cuda.synchronize()
        best = min(v for v in (row["eager_ms"], row["graph_ms"]) if v)
        row["images_per_s_per_rank"] = round(2 * Br / best * 1e3, 1)
        rows.append(row)
        del model, opt
        torch.cuda.empty_cache()
    res["per_rank_batch_table"] = {
        "note": "single GPU, no collective: ms/step of one rank of a strong-scaling run (global 512 per loader); N x images_per_s_per_rank "
                "is the ceiling of that run before the all-reduce",
        "rows": rows}

    # ---- BASELINE configs[4] -------------------------------------------------------------------------------------------------
    try:
        c5 = svhn_measure(S, 1024, "bf16", steps=20, warmup=3, want=-1, multi=False, rank=0)
        res["config5"] = {"workload": "svhn_VAE smooth-ELBO iteration (2 fwd + 1 bwd + Adam), B_u=B_l=1024, bf16 (BASELINE configs[4], "
                                      "one GPU)", "ms_per_step": round(c5["ms"], 3),
                          "images_per_s": round(2048 / c5["ms"] * 1e3, 1), "launch": c5["mode"], "launch_probe": c5["probe"],
                          "finite": c5["finite"]}
    except Exception as e:          # never lose the headline line to an extra
        res["config5"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return res


def layer_table(windows=5):
    """The nine WRN-28-10 body-convolution rows behind the north-star bar: `windows` short measurements per kernel, MEDIAN
    and MIN of them (a 20-launch window is 4-6 ms: one clock / driver hiccup inside it doubles an average, so the minimum is
    the kernel's speed and the median what a run sees); `passing` counts rows whose MEDIAN meets the bar."""
    import contextlib
    import statistics
    from tools import layer_bench as LB
    rows = []
    with contextlib.redirect_stdout(sys.stderr):          # (the tool prints its own table: keep stdout to the one JSON line)
        for Cc, H in ((160, 32), (320, 16), (640, 8)):
            r = {}
            for _ in range(windows):
                for k, us in LB.bench_layer(512, Cc, H, Cc).items():
                    r.setdefault(k, []).append(us)
            row = {"layer": "conv3x3 s1 %d->%d @%dx%d, B=512, 0.2416 TFLOP" % (Cc, Cc, H, H)}
            for k, v in r.items():
                med, mn = statistics.median(v), min(v)
                row[k + "_us"] = round(med, 1)
                row[k + "_us_min"] = round(mn, 1)
                row[k + "_frac"] = round(0.2416e12 / (med * 1e-6) / (MFMA_PEAK_TFLOPS["bf16"] * 1e12), 4)
                row[k + "_frac_best"] = round(0.2416e12 / (mn * 1e-6) / (MFMA_PEAK_TFLOPS["bf16"] * 1e12), 4)
            rows.append(row)
    return {"wrn28_10_layer_table": {
        "bar": "frac >= 0.40 (<= 242 us)", "peak_TFLOPs": MFMA_PEAK_TFLOPS["bf16"],
        "method": "tools/layer_bench.py in-process: 3 warm-up + 20 launches per window, %d windows: *_us / *_frac = MEDIAN window, "
                  "*_us_min / *_frac_best = fastest window" % windows,
        "rows": rows,
        "passing": sum(1 for r in rows for k in ("fwd", "dgrad", "wgrad") if r[k + "_frac"] >= 0.40),
        "passing_best_window": sum(1 for r in rows for k in ("fwd", "dgrad", "wgrad") if r[k + "_frac_best"] >= 0.40),
        "of": 3 * len(rows)}}


def _multi(world):
    """The collectives run with more than one rank -- or with ONE rank under SV_DP_SINGLE_RANK=1 (shot_vae_amd/dp.py: the same
    RCCL calls through the hardware of a one-GPU box; a functional check)."""
    return world > 1 or os.environ.get("SV_DP_SINGLE_RANK") == "1"


def svhn_measure(S, B, dtype, steps, warmup, want, multi, rank, init_pg=None):
    """One smooth-ELBO iteration of svhn_VAE at B images per loader on this GPU: eager issue against hipGraph replay
    (want: 1 graph, 0 eager, -1 probe both and keep the faster).  `init_pg`: called AFTER the graph has been captured and
    before anything is timed -- the RCCL process group must not be alive during a stream capture (its watchdog thread polls
    events), a replay next to it is an ordinary launch; the parameters are broadcast once the group exists."""
    torch.manual_seed(1)
    model = S.SmoothVAE((3, 32, 32), {"cont": 32, "disc": [10]}, temperature=0.67, compute_dtype=dtype).cuda().train()
    loss_fn = S.SmoothELBOLoss()
    opt = S.FlatAdam(model.parameters(), lr=1e-3, capturable=True)
    torch.manual_seed(1 + rank)
    torch.cuda.manual_seed(1 + rank)
    g = torch.Generator(device="cuda").manual_seed(1234 + rank)
    u = torch.rand(B, 3, 32, 32, device="cuda", generator=g) * 2 - 1
    l = torch.rand(B, 3, 32, 32, device="cuda", generator=g) * 2 - 1
    y = torch.randint(0, 10, (B,), device="cuda", generator=g)
    mode, probe, graphed = "eager", None, None
    if want != 0:
        try:
            graphed = S.GraphedSmoothStep(model, loss_fn, opt, u, l, y, warmup=2, distributed=multi)
        except Exception as e:
            mode = "eager (graph capture failed: %s)" % type(e).__name__
            torch.cuda.synchronize()
    if init_pg is not None:
        init_pg()
    if multi:
        for p_ in model.parameters():           # (views of FlatAdam's flat buffer: the graph reads the same storage)
            dist.broadcast(p_.data, 0)
        for t_ in (opt.m, opt.v):
            dist.broadcast(t_, 0)
        if opt.step_dev is not None:
            dist.broadcast(opt.step_dev, 0)

    def eager():
        return S.smooth_train_step(model, loss_fn, opt, u, l, y, distributed=multi)

    def sync():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, n):
        sync()
        t = time.perf_counter()
        for _ in range(n):
            out = fn()
        sync()
        return (time.perf_counter() - t) / n * 1e3, out

    for _ in range(max(warmup, 2)):
        eager()
    if graphed is not None:
        if want < 0:
            pe, pg = timed(eager, 10)[0], timed(graphed, 10)[0]
            if multi:                   # the slower rank decides, every rank decides the same
                pv = torch.tensor([pe, pg], device="cuda", dtype=torch.float64)
                dist.all_reduce(pv, op=dist.ReduceOp.MAX)
                pe, pg = float(pv[0]), float(pv[1])
            probe = {"eager_ms": round(pe, 3), "graph_ms": round(pg, 3)}
            if probe["graph_ms"] < probe["eager_ms"]:
                mode = "hipGraph replay (faster than eager issue in the probe)"
            else:
                graphed, mode = None, "eager (faster than hipGraph replay in the probe)"
        else:
            mode = "hipGraph replay"
    step = graphed if graphed is not None else eager
    ms, loss = timed(step, steps)
    dt = torch.tensor([ms], device="cuda", dtype=torch.float64)
    if multi:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    return {"ms": float(dt), "loss": float(loss), "finite": bool(torch.isfinite(loss).all()), "mode": mode, "probe": probe}


def svhn_workload(a, rank, world, init_pg):
    """--workload svhn: one smooth-ELBO iteration of svhn_VAE (BASELINE configs[4]; main_smooth_ELBO_svhn.py:152-176: unlabelled
    forward + loss, labelled forward + loss, one backward, Adam) on --batch images per loader per GPU.  The iteration is ~130
    launches of tens of microseconds: launch-bound, so the run probes eager issue against a hipGraph replay and keeps the faster
    (both in the JSON); at N > 1 the graph holds forward + backward and ONE all-reduce of FlatAdam's flat gradient buffer +
    the sv_adam launch follow it eagerly (the graph is captured before the process group is created)."""
    import shot_vae_amd as S
    multi = _multi(world)
    B = a.batch if a.scaling == "weak" else a.batch // world
    r = svhn_measure(S, B, a.dtype, a.steps, a.warmup, a.graph if a.graph is not None else -1, multi, rank, init_pg)
    ms, mode, probe = r["ms"], r["mode"], r["probe"]
    assert r["finite"], "non-finite loss"
    # algorithmic work: 1.1332e7 MACs per forwarded image, x3 for forward + both gradients, x2 flop per MAC
    flops = 2 * B * 1.1332e7 * 2 * 3
    out = {"metric": "images/sec/step svhn_VAE smooth-ELBO bs%d (BASELINE configs[4])" % a.batch, "value": round(2 * B * world / ms * 1e3, 1),
           "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3),
           "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
           "config": {"workload": "svhn_VAE smooth-ELBO iteration (2 fwd + 1 bwd + Adam), B_u=B_l=%d per GPU, synthetic 3x32x32 in "
                                  "[-1,1] in HBM, random init" % B, "global_batch": 2 * B * world, "parallelism": "dp%d" % world,
                      "launch": mode, "optimizer": "FlatAdam (one sv_adam launch on a flat buffer)",
                      "collective": "1 RCCL all-reduce of the flat gradient buffer (2.49 M floats) per iteration" if multi else "none"},
           "loss": round(r["loss"], 4), "TFLOPs": round(flops / ms / 1e9, 2),
           "note": "launch-bound workload (SURVEY.md 8d): 0.07 TFLOP and 0.43 GB per iteration"}
    if probe is not None:
        out["config"]["launch_probe"] = probe
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


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
        if os.environ.get("SV_BENCH_FAIL_RANK") == str(rank):      # launcher test: this rank dies before the rendezvous
            sys.exit(3)
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
    multi = _multi(world)
    # "nccl" is RCCL on ROCm.  SV_DIST_BACKEND=gloo lets the N > 1 code path be exercised on a one-GPU box (both
    # ranks on the same device, the collective through host memory): a functional check, not a measurement.
    backend = os.environ.get("SV_DIST_BACKEND", "nccl")

    def init_pg():
        """Creates the process group -- AFTER the step's hipGraph has been captured when one is wanted: no stream capture while
        an RCCL process group is alive (its watchdog thread polls hipEventQuery on outstanding work, an error during another
        thread's capture on this stack); replaying next to it is an ordinary launch, and the all-reduce + optimizer launch
        stay outside the graph."""
        if multi and not dist.is_initialized():
            # (device_id binds the communicator -- and every barrier -- to this rank's GPU instead of a guess from the rank)
            kw = {"device_id": torch.device("cuda", local % max(ndev, 1))} if backend == "nccl" else {}
            dist.init_process_group(backend, rank=rank, world_size=world, **kw)

    if a.workload == "svhn":
        return svhn_workload(a, rank, world, init_pg)
    import shot_vae_amd as S
    from shot_vae_amd import _lib as L
    from shot_vae_amd import dp
    if a.disable:
        L.call("sv_set_option", L.OPT_DISABLE_MASK, a.disable)
    if a.enable:
        L.call("sv_set_option", L.OPT_ENABLE_MASK, a.enable)
    if a.persistent_blocks:
        L.call("sv_set_option", L.OPT_PERSISTENT_BLOCKS, a.persistent_blocks)
    if a.deterministic:
        L.call("sv_set_option", L.OPT_DETERMINISTIC, a.deterministic)

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
    # per-rank noise streams (eps, Gumbel u, pairings): torch seeds = seed + rank.  The mixup coefficients are the lambda
    # contract of SURVEY.md 5.2 -- every rank must use the SAME lambda_l / lambda_u in a step: the eager step draws them
    # from numpy's global generator, seeded IDENTICALLY on every rank here (and consumed in lockstep: two draws per step);
    # the captured step reads DeviceRng tables built from the same seed on every rank.
    import numpy as np
    np.random.seed(20240 + 1)
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
    if a.pair_blocks >= 0:
        model._engine.pair_blocks = a.pair_blocks
    if a.pair_blocks_strided >= 0:
        model._engine.pair_blocks_strided = a.pair_blocks_strided
    model._engine.light_fork = bool(a.light_fork)
    model._engine.flag_fork = bool(a.flag_fork)
    if a.fused_bwd >= 0:
        model._engine.fused_bwd = a.fused_bwd
    if a.fused_blocks:
        model._engine.fused_blocks = a.fused_blocks
    model._engine.event_fork_after_fused = bool(a.event_fork_after_fused)
    model._engine.fold_bn_bwd = bool(a.fold_bn_bwd)
    if a.fused_channels:
        model._engine.fused_channels = tuple(int(v) for v in a.fused_channels.split(","))
    model._engine.compact_shortcut_grad = bool(a.compact_shortcut)
    if a.fork_every:
        model._engine.fork_every = a.fork_every
    dmode = False if not multi else ("bucketed" if a.allreduce == "bucketed" else True)

    mode = a.graph if a.graph is not None else (-1 if a.scaling == "strong" else 0)
    graphed, graph_note, probe = None, "eager, weight gradients on a side stream", None
    if mode and a.schedule != "sequential":
        # captured BEFORE the process group exists (see init_pg): the constructor's warm-up steps are rank-local updates, the
        # broadcast below re-synchronises parameters, BatchNorm buffers and the optimizer state
        assert not dist.is_initialized()
        try:
            graphed = GraphedTrainStep(model, elbo, cls, opt, il, ll, iu, sch, distributed=dmode, schedule=a.schedule)
            graph_note = "hipGraph replay"
        except Exception as e:        # capture unsupported on this stack: run eagerly, say so in the output
            graphed, graph_note = None, "eager (graph capture failed: %s)" % type(e).__name__
            torch.cuda.synchronize()
    init_pg()
    if multi:
        dp.broadcast_parameters(model, optimizer=opt)

    def eager_step():
        if a.schedule == "grouped":
            return train_step_grouped(model, elbo, cls, opt, il, ll, iu, sch, distributed=dmode, input_stream=bool(a.input_stream))
        if a.schedule == "two-stream":
            return train_step_overlapped(model, elbo, cls, opt, il, ll, iu, sch, distributed=dmode)
        return S.train_step(model, elbo, cls, opt, il, ll, iu, sch, distributed=dmode)

    def sync():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    if mode < 0 and graphed is not None:
        # launch-mode probe: a few steps of each mode (every rank runs the same sequence: the collectives stay matched),
        # the slower rank's time decides, every rank takes the same decision
        def probe_ms(fn, n=6):
            for _ in range(2):
                fn()
            sync()
            t = time.perf_counter()
            for _ in range(n):
                fn()
            sync()
            v = torch.tensor([(time.perf_counter() - t) / n * 1e3], device="cuda", dtype=torch.float64)
            if multi:
                dist.all_reduce(v, op=dist.ReduceOp.MAX)
            return float(v)
        probe = {"eager_ms": round(probe_ms(eager_step), 3), "graph_ms": round(probe_ms(graphed), 3)}
        if probe["eager_ms"] <= probe["graph_ms"]:
            graphed, graph_note = None, "eager, weight gradients on a side stream (faster than hipGraph replay in the probe)"
        else:
            graph_note = "hipGraph replay (faster than eager issue in the probe)"

    def step():
        return graphed() if graphed is not None else eager_step()

    for _ in range(a.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ls, lu = step()
    sync()
    dt = time.perf_counter() - t0
    if multi:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    assert torch.isfinite(ls).all() and torch.isfinite(lu).all(), "non-finite loss"
    # the lambda contract: every rank used the same mixup coefficients in the last step
    lam_equal = None
    lams = getattr(model, "_last_lams", None)
    if multi and lams is not None:
        hi = torch.tensor([float(v) for v in lams], device="cuda", dtype=torch.float64)
        lo = hi.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        lam_equal = bool(torch.equal(hi, lo))
        assert lam_equal, "ranks disagree on the mixup coefficients: max %s, min %s" % (hi.tolist(), lo.tolist())
    # the device-side forks of the backward (sv_igemm_args::start_flag): a wait that gave up means a weight gradient ran early
    L.check_flag_timeouts("bench.py timed region")      # (every step checks it too: Engine._join_side, FlatSGD.step)
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
                      "collective": "none" if not multi else (
                          "RCCL all-reduce of the flat fp32 gradient buffer in two buckets per step: decoder tail (88 % of the "
                          "bytes) on a communication stream under the encoder's backward, encoder + heads after it"
                          if dmode == "bucketed" and graphed is None and a.schedule != "two-stream" else
                          "1 RCCL all-reduce of the flat fp32 gradient buffer per step")},
           "loss_sup": round(float(ls), 5), "loss_unsup": round(float(lu), 5)}
    if headline and a.dtype == "bf16":
        # the whole step against SURVEY.md 8d's MINIMUM model (BatchNorm / activation fully fused into the convolutions: every
        # conv reads X, W and writes Y; dgrad reads dY, W, writes dX; wgrad reads X, dY): 14.26 GB and 3.058 TFLOP per step.
        # The real pass structure moves about three times that, so this fraction is the step-level measure of
        # wasted traffic, next to the per-kernel `roofline` below.
        t_s = dt / a.steps
        out["step_vs_minimum"] = {"min_hbm_bytes": 14.26e9, "GBps": round(14.26e9 / t_s / 1e9, 1),
                                  "frac_of_hbm_peak": round(14.26e9 / t_s / (HBM_PEAK_GBS * 1e9), 4),
                                  "flops": 3.058e12, "TFLOPs": round(3.058e12 / t_s / 1e12, 1),
                                  "frac_of_mfma_peak": round(3.058e12 / t_s / (MFMA_PEAK_TFLOPS["bf16"] * 1e12), 4),
                                  "tensor_passes_per_residual_unit_backward": {"pair": 17, "fused_32ch": 10, "fused_64ch": 10},
                                  "note": "17 algorithmic passes per unit with the data / weight gradient pair (dgrad 3, bn-backward "
                                          "3 / 4, wgrad 2, twice; the paired weight gradients find dY / x in L2: PMC 1.07-1.28x their own "
                                          "bytes); 10 for the same-shape 32- and 64-channel units, whose backward is TWO launches of sv_bwd3x3 "
                                          "(Engine.fused_bwd = 2): conv1's data + weight gradient with norm2's BatchNorm backward in the "
                                          "load path (4 passes instead of 8), conv2's with the next unit's norm1 backward + skip "
                                          "gradient in it (6 instead of 9)"}
    if probe is not None:
        out["config"]["launch_probe"] = probe
    if lam_equal is not None:
        out["config"]["lambda_equal_across_ranks"] = lam_equal

    # ---- roofline: HIP events around EVERY launch of the library (separate single-stream passes of the same step) ----
    if not a.no_roofline:
        eng = model._engine
        peak_t = MFMA_PEAK_TFLOPS[a.dtype]

        def prof_pass(paired):
            """One stream (HIP events bracket every launch).  paired: the body convolutions' weight / data gradients with the
            HALF block budgets they are launched with in the timed two-stream step (Engine.pair_blocks) -- what the timed step
            issues, each kernel timed without its partner; not paired: every launch with the full budget."""
            eng.prof_tags, eng.prof_cost = {}, {}
            L.prof_tags = eng.prof_tags
            # (the BatchNorm finalisations sv_igemm issues itself for folded launches: a tag of their own, not the layer's)
            L.lib().sv_prof_nested_tag(eng.prof_tags.setdefault("sv_bn_finalize(folded)", len(eng.prof_tags)))
            L.lib().sv_prof_enable(1)
            side, eng.wgrad_side_stream, eng.prof_paired = eng.wgrad_side_stream, False, bool(paired and eng.wgrad_side_stream)
            try:
                for _ in range(a.prof_steps):
                    if a.schedule == "grouped":
                        train_step_grouped(model, elbo, cls, opt, il, ll, iu, sch, distributed=dmode)
                    else:
                        S.train_step(model, elbo, cls, opt, il, ll, iu, sch, distributed=dmode)
                ntag = len(eng.prof_tags) + 1
                ms = (ctypes.c_double * ntag)()
                cnt = (ctypes.c_int * ntag)()
                L.lib().sv_prof_collect(ntag, ms, cnt)
            finally:
                L.lib().sv_prof_enable(0)
                L.lib().sv_prof_nested_tag(-1)
                tags = dict(eng.prof_tags)
                L.prof_tags = eng.prof_tags = None
                eng.wgrad_side_stream, eng.prof_paired = side, False
            rows = []
            for name, i in tags.items():
                if not cnt[i]:
                    continue
                # (a negative sum was round 4's nested-scope bug: an outer scope's end event left unrecorded / stale)
                assert ms[i] >= 0.0, "in-situ timing: tag %s sums to %g ms over %d launches" % (name, ms[i], cnt[i])
                nbytes, flops, nl = eng.prof_cost.get(name, (0.0, 0.0, 0))
                rows.append(dict(name=name, total_ms=ms[i], launches=cnt[i], avg_us=1000 * ms[i] / cnt[i],
                                 bytes=nbytes / nl if nl else None, flops=flops / nl if nl else None))
            return rows

        # every costed launch against ITS roofline, by family (floor = max(bytes / HBM peak, flops / MFMA peak) per launch):
        # the body 3x3 layers, BatchNorm backward, and the odd layers (decoder, stride-2, 1x1 shortcuts, 16-channel layers)
        def family(n):
            if n.startswith("sv_"):
                return n
            body = any(n.split("+")[0].endswith("conv3x3_%dx%d_s1" % (c, c)) for c in (32, 64, 128, 160, 320, 640))
            return "body_conv3x3_" + n.split(":")[0] if body else "odd_layers_" + n.split(":")[0]

        def families(rows):
            fam = {}
            for r in rows:
                if not r["bytes"]:
                    continue
                f = fam.setdefault(family(r["name"]), {"ms": 0.0, "floor_ms": 0.0, "launches": 0})
                floor_us = max(r["bytes"] / (HBM_PEAK_GBS * 1e3), r["flops"] / (peak_t * 1e6))
                f["ms"] += r["total_ms"] / a.prof_steps
                f["floor_ms"] += floor_us * r["launches"] / a.prof_steps / 1e3
                f["launches"] += r["launches"] // a.prof_steps
            return fam

        def conv_block(fam):
            """the convolution families together: time-weighted fraction of their rooflines (= sum of floors / sum of times)"""
            conv = [v for k, v in fam.items() if not k.startswith("sv_")]
            t, fl = sum(v["ms"] for v in conv), sum(v["floor_ms"] for v in conv)
            return {"ms_per_step": round(t, 3), "floor_ms": round(fl, 3), "frac_of_roofline": round(fl / t, 4) if t else None,
                    "launches_per_step": sum(v["launches"] for v in conv),
                    "definition": "every conv-like launch (forward, data gradient, weight gradient; body and odd layers): "
                                  "sum over launches of max(bytes / 8 TB/s, flops / MFMA peak) divided by the sum of their times"}

        paired = a.schedule == "grouped" and eng.wgrad_side_stream and eng.pair_blocks > 0
        rows = prof_pass(paired)
        tot = sum(r["total_ms"] for r in rows)
        costed = [r for r in rows if r["bytes"]]
        if costed:
            d = max(costed, key=lambda r: r["total_ms"])
            ai = d["flops"] / d["bytes"]
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
                        launches_per_step_all=sum(r["launches"] for r in rows) // a.prof_steps,
                        block_budgets="paired: the body convolutions' weight / data gradients with %d blocks each, as the timed "
                                      "two-stream step launches them (each timed alone here)" % eng.pair_blocks if paired else
                                      "full budget for every launch")
            # the five largest entries of the same pass, so that no large kernel stays invisible behind the dominant one
            top = sorted(rows, key=lambda r: -r["total_ms"])[:5]
            roof["top5"] = [{"kernel": r["name"], "ms_per_step": round(r["total_ms"] / a.prof_steps, 3),
                             "launches_per_step": r["launches"] // a.prof_steps, "avg_us": round(r["avg_us"], 2),
                             "GBps": round(r["bytes"] / r["avg_us"] / 1e3, 1) if r["bytes"] else None} for r in top]
            fam = families(rows)
            roof["conv"] = conv_block(fam)
            roof["families"] = {k: {"ms_per_step": round(v["ms"], 3), "floor_ms": round(v["floor_ms"], 3),
                                    "frac_of_roofline": round(v["floor_ms"] / v["ms"], 3), "launches_per_step": v["launches"]}
                                for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])}
            if paired:          # the same step with the full block budget for every launch, for comparison
                rows_full = prof_pass(False)
                fam_full = families(rows_full)
                roof["full_budget"] = {"kernel_ms_per_step": round(sum(r["total_ms"] for r in rows_full) / a.prof_steps, 3),
                                       "conv": {k: v for k, v in conv_block(fam_full).items() if k != "definition"},
                                       "families_frac": {k: round(v["floor_ms"] / v["ms"], 3) for k, v in
                                                         sorted(fam_full.items(), key=lambda kv: -kv[1]["ms"])}}
            out["roofline"] = roof
            if rank == 0 and os.environ.get("SV_BENCH_TABLE"):
                for tag_, rr in (("", rows),) + ((("full-budget ", rows_full),) if paired else ()):
                    for r in sorted(rr, key=lambda r: -r["total_ms"]):
                        print("# %s%-28s %6d launches  avg %9.2f us  total %8.3f ms/step  %9s GB/s  %9s TFLOP/s" % (
                            tag_, r["name"], r["launches"] // a.prof_steps, r["avg_us"], r["total_ms"] / a.prof_steps,
                            "%.1f" % (r["bytes"] / r["avg_us"] / 1e3) if r["bytes"] else "-",
                            "%.2f" % (r["flops"] / r["avg_us"] / 1e6) if r["flops"] else "-"), file=sys.stderr)
    if rank == 0 and world == 1 and headline and a.dtype == "bf16" and not a.no_extras:
        del model, opt, graphed
        torch.cuda.empty_cache()
        out.update(extras(S))
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a.net, K)
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
