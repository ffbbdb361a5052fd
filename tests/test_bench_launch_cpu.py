"""bench.py's launcher: `python bench.py --gpus N` outside torchrun starts its own N ranks (fresh processes, torchrun-style
environment), passes rank 0's single JSON line through and fails if any rank fails.  --dry-run replaces the GPU work by a
gloo rendezvous + max-reduce, so this runs on CPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, env=e,
                          timeout=600)


def test_bench_spawns_its_own_ranks():
    r = _run("--gpus", "2", "--dry-run", "--scaling", "strong", "--batch", "512")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]      # (gloo itself prints a connection note)
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["max_over_ranks"] == 2.0 and out["per_rank_batch"] == 256


def test_bench_fails_when_a_rank_fails():
    r = _run("--gpus", "2", "--dry-run", "--scaling", "strong", "--batch", "511")
    assert r.returncode != 0


def test_bench_watchdog_ends_the_ranks_left_in_the_rendezvous():
    """Rank 1 dies before the rendezvous: rank 0 would sit in it until the store times out (minutes); the launcher's
    watchdog ends it and reports the failure within seconds."""
    import time
    t0 = time.time()
    r = _run("--gpus", "2", "--dry-run", env=dict(SV_BENCH_FAIL_RANK="1"))
    assert r.returncode == 1 and "ranks failed" in r.stderr and "(1, 3)" in r.stderr, r.stderr[-500:]
    assert time.time() - t0 < 60


def test_bench_rejects_mismatched_world_size():
    r = _run("--gpus", "4", "--dry-run", env=dict(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                                                  MASTER_PORT="29999"))
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
