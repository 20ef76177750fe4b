"""`python bench.py --gpus N` must really start N ranks (round-1 finding: the flag was parsed and ignored).
Checked here without a GPU: TC_BENCH_LAUNCH_TEST makes every rank stop after the rendezvous and the 3-float
all-reduce (gloo), so this exercises exactly the launcher, the WORLD_SIZE cross-check and the one-line
output contract."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        if k not in env_extra:
            env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=300)


def test_gpus_flag_spawns_that_many_ranks():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"TC_BENCH_LAUNCH_TEST": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["weight"] == 2 * 38400.0


def test_mismatch_between_flag_and_launcher_fails_loudly():
    r = _run(["--gpus", "4"], {"TC_BENCH_LAUNCH_TEST": "1", "WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0
    assert "WORLD_SIZE=2" in (r.stderr + r.stdout)
