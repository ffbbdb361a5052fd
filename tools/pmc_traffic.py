"""HBM traffic of the dominant conv-like kernels from the PMC counters (run ON the GPU box, from the repo root):

    python tools/pmc_traffic.py [out.json]

For every layer in LAYERS it runs tools/layer_bench.py (that one kernel, a few launches) under rocprofv3 in two separate
counter passes -- FETCH_SIZE and WRITE_SIZE do not fit into the TCC slots of one pass -- with --kernel-trace only
(MI355X_MICROARCH.md, "rocprofv3 PMC slots" / "HBM"), and prices the traffic per launch as

    traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes

(FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests of a wide coalesced stream at 64 bytes,
hence the factor 2; WRITE_SIZE is exact for 16-byte-per-lane stores and float atomics).  This process never touches the
GPU itself; the profiled program is started directly after `--`.  bench.py reads the JSON (profiles/rNN_pmc_traffic.json)
to fill `roofline.traffic` for the kernel it finds dominant."""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# tag (bench.py's name of the launch) -> (B, Cin, H, N, kind, substrings of the kernel names that make up one launch).
# B = the images ONE launch of the grouped step processes: 4 x 512 at BASELINE config 2, 4 x 256 at config 4 (a batched
# launch of G groups of B images runs the same blocks as one group of G * B images)
LAYERS = {
    "wgrad:conv3x3_32x32_s1": (2048, 32, 32, 32, "wgrad", ["wgrad3x3m_kernel", "wgrad3x3_kernel", "slab_reduce_kernel"]),
    "fwd:conv3x3_32x32_s1": (2048, 32, 32, 32, "fwd", ["conv3x3p_kernel"]),
    "dgrad:conv3x3_32x32_s1": (2048, 32, 32, 32, "dgrad", ["conv3x3p_kernel"]),
    "wgrad:conv3x3_64x64_s1": (2048, 64, 16, 64, "wgrad", ["wgrad3x3m_kernel", "wgrad3x3_kernel", "slab_reduce_kernel"]),
    "fwd:conv3x3_64x64_s1": (2048, 64, 16, 64, "fwd", ["conv3x3p_kernel"]),
    "dgrad:conv3x3_64x64_s1": (2048, 64, 16, 64, "dgrad", ["conv3x3p_kernel"]),
    "wgrad:conv3x3_128x128_s1": (2048, 128, 8, 128, "wgrad", ["wgrad3x3m_kernel", "wgrad3x3_kernel", "slab_reduce_kernel"]),
    "fwd:conv3x3_128x128_s1": (2048, 128, 8, 128, "fwd", ["conv3x3_kernel", "conv3x3w_kernel"]),
    "dgrad:conv3x3_128x128_s1": (2048, 128, 8, 128, "dgrad", ["conv3x3_kernel", "conv3x3w_kernel"]),
    "bwd:conv3x3_32x32_s1+bn": (2048, 32, 32, 32, "bwd2", ["bwd3x3f_kernel", "slab_reduce_kernel"]),
    "bwd:conv3x3_32x32_s1+bn+skip": (2048, 32, 32, 32, "bwd3", ["bwd3x3f_kernel", "slab_reduce_kernel"]),
    "bwd:conv3x3_64x64_s1+bn": (2048, 64, 16, 64, "bwd2", ["bwd3x3g_kernel", "slab_reduce_kernel"]),
    "bwd:conv3x3_64x64_s1+bn+skip": (2048, 64, 16, 64, "bwd3", ["bwd3x3g_kernel", "slab_reduce_kernel"]),
    "fwd:conv3x3_160x160_s1": (1024, 160, 32, 160, "fwd", ["conv3x3x_kernel", "conv3x3w_kernel"]),
    "wgrad:conv3x3_160x160_s1": (1024, 160, 32, 160, "wgrad", ["wgrad3x3w_kernel", "slab_reduce_kernel"]),
}
ITERS, WARM = 4, 1
# every conv-like kernel of the library (substrings of the mangled names).  A dispatch is filed under the LONGEST name it contains
# ("wgrad_kernel" is a substring of "hwgrad_kernel" / "thwgrad_kernel" / "s2wgrad_kernel").  tools/pmc_layer.py / pmc_sq.py match
# against this list unless kernel names are given: the round-5 kernels were missing from their hand-written defaults and
# profiles/r05_pmc_odd.txt held `kernels [...] not found` where their counters should have been.
ALL_KERNELS = ["igemm_kernel", "igemm_dma_kernel", "halo_kernel", "halop_kernel", "hwgrad_kernel", "wgrad_kernel", "wgradc_kernel",
               "slab_reduce_kernel", "conv3x3_kernel", "conv3x3p_kernel", "conv3x3m_kernel", "conv3x3w_kernel", "conv3x3x_kernel",
               "wgrad3x3_kernel", "wgrad3x3m_kernel", "wgrad3x3w_kernel", "tconvr_kernel", "tconvx16_kernel", "sconv_kernel",
               "pconv_kernel", "dconv_kernel", "thconv_kernel", "thwgrad_kernel", "s2wgrad_kernel", "bwd3x3f_kernel", "bwd3x3g_kernel", "k4wgrad_kernel"]


def kernel_of(mangled, names):
    """the name of `names` this dispatch belongs to (longest match), or None"""
    hit = [n for n in names if n in mangled]
    return max(hit, key=len) if hit else None


def one_pass(tag, counter, outdir):
    B, Cin, H, N, kind, names = LAYERS[tag]
    d = os.path.join(outdir, tag.replace(":", "_") + "_" + counter)
    env = dict(os.environ, SV_BENCH_ITERS=str(ITERS), SV_BENCH_WARM=str(WARM))
    if kind == "wgrad" and Cin <= 128:
        # the narrow weight gradients run with HALF the persistent-block budget in the step (paired with the data gradient on
        # the other stream: Engine.pair_blocks) -- half the blocks publish half the partial slabs
        env["SV_BENCH_PERSISTENT_BLOCKS"] = "256"
    cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
           "python3", os.path.join(ROOT, "tools", "layer_bench.py"), str(B), str(Cin), str(H), str(N), kind]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=ROOT)
    if r.returncode != 0:
        raise RuntimeError("rocprofv3 failed: " + r.stderr[-2000:])
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    assert files, "no counter_collection.csv under " + d
    per = {}            # kernel-name substring -> list of counter values (one per dispatch, in order)
    for row in csv.DictReader(open(files[0])):
        if row.get("Counter_Name") != counter:
            continue
        nm = kernel_of(row["Kernel_Name"], names)
        if nm:
            per.setdefault(nm, []).append(float(row["Counter_Value"]))
    total = 0.0
    for nm, vals in per.items():
        n_launch = ITERS + WARM
        per_launch = len(vals) // n_launch           # dispatches of this kernel per launch of the layer
        assert per_launch >= 1 and len(vals) == per_launch * n_launch, (tag, nm, len(vals))
        total += sum(vals[per_launch * WARM:]) / ITERS
    assert per, "kernels %s not found in %s" % (names, files[0])
    return total, sorted(per)


# kernels that have no single-layer harness: measured inside the WHOLE step (bench.py under the counter pass), mean over
# every dispatch of the kernel.  tag -> (kernel-name substring, algorithmic bytes are taken from bench.py's own line)
STEP_KERNELS = {"sv_bn_bwd_apply": "bn_bwd_apply_kernel"}


def step_pass(counter, outdir):
    d = os.path.join(outdir, "step_" + counter)
    cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
           "python3", os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-roofline",
           "--no-extras", "--flag-fork", "0"]      # (counter collection serialises dispatch: a kernel that waits for another
                                                    #  stream's kernel to start would sit in front of it; the engine also detects it)
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT)
    if r.returncode != 0:
        raise RuntimeError("rocprofv3 failed: " + r.stderr[-2000:])
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    assert files, "no counter_collection.csv under " + d
    per, rows = {}, []
    for row in csv.DictReader(open(files[0])):
        if row.get("Counter_Name") != counter:
            continue
        rows.append((int(row.get("Dispatch_Id", len(rows))), row["Kernel_Name"], float(row["Counter_Value"])))
        for tag, nm in STEP_KERNELS.items():
            if nm in row["Kernel_Name"]:
                per.setdefault(tag, []).append(float(row["Counter_Value"]))
    res = {tag: (sum(v) / len(v), len(v)) for tag, v in per.items()}
    # the WHOLE step: every dispatch between the last two sgd_kernel launches (the second timed step), all kernels summed
    rows.sort()
    sgd = [i for i, r in enumerate(rows) if "sgd_kernel" in r[1]]
    if len(sgd) >= 2:
        seg = rows[sgd[-2] + 1: sgd[-1] + 1]
        res["__step__"] = (sum(r[2] for r in seg), len(seg))
    return res


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "pmc_traffic.json")
    outdir = os.path.join(ROOT, "gpurun_out", "pmc")
    os.makedirs(outdir, exist_ok=True)
    res = {}
    if len(sys.argv) > 2 and sys.argv[2] == "step" and os.path.exists(out):
        res = json.load(open(out))          # add the step-level kernels to an existing table
    fetch, write = step_pass("FETCH_SIZE", outdir), step_pass("WRITE_SIZE", outdir)
    for tag in STEP_KERNELS:
        f, n = fetch[tag]
        w, _ = write[tag]
        res[tag] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "traffic_bytes": (2 * f + w) * 1024, "kernels": [STEP_KERNELS[tag]],
                    "shape": "every launch of the kernel in the BASELINE config-2 step (bench.py, grouped schedule)",
                    "formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024, per launch, mean of the %d launches of 3 steps (2 timed + 1 warm-up) of "
                               "bench.py --no-extras" % n}
        print(tag, json.dumps(res[tag]), flush=True)
    if "__step__" in fetch and "__step__" in write:
        f, n = fetch["__step__"]
        w, _ = write["__step__"]
        res["whole_step"] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "traffic_bytes": (2 * f + w) * 1024, "dispatches": n,
                             "minimum_model_bytes": 14.26e9, "ratio_to_minimum": (2 * f + w) * 1024 / 14.26e9,
                             "formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024 summed over EVERY dispatch of one BASELINE config-2 step "
                                        "(sgd_kernel to sgd_kernel) of bench.py --no-extras; SURVEY.md 8d's minimum model is 14.26 GB"}
        print("whole_step", json.dumps(res["whole_step"]), flush=True)
    if len(sys.argv) > 2 and sys.argv[2] == "step":
        json.dump(res, open(out, "w"), indent=1)
        return
    for tag in LAYERS:
        B, Cin, H, N, kind, _ = LAYERS[tag]
        fetch, k1 = one_pass(tag, "FETCH_SIZE", outdir)
        write, _ = one_pass(tag, "WRITE_SIZE", outdir)
        es = 2
        if kind == "wgrad":
            alg = es * B * H * H * (Cin + N) + 4 * 9 * Cin * N
        elif kind in ("bwd2", "bwd3"):     # the fused backward: two-tensor form 4 passes, residual form 6, + the slabs (written, read back)
            alg = es * B * H * H * Cin * (4 if kind == "bwd2" else 6) + es * 9 * Cin * N + 4 * 9 * Cin * N
        else:       # forward: x, y, residual (+ weights); data gradient: dy, dx, the raw tensor of the activation backward
            alg = es * B * H * H * (Cin + 2 * N) + es * 9 * Cin * N
        res[tag] = {"FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
                    "traffic_bytes": (2 * fetch + write) * 1024, "algorithmic_bytes": alg,
                    "kernels": k1, "shape": {"B": B, "Cin": Cin, "H": H, "N": N, "kind": kind},
                    "formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024, per launch, mean of %d launches" % ITERS}
        if kind == "wgrad" and Cin <= 128:
            res[tag]["persistent_blocks"] = 256
        print(tag, json.dumps(res[tag]), flush=True)
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
