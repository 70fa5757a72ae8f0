"""`python bench.py --gpus N` must start its own ranks (VERDICT r01 / contract: the driver may call it that way) and must
refuse a WORLD_SIZE that contradicts --gpus.  Here (no GPU) the ranks run the stub step (MONORTM_BENCH_STUB=1): launcher,
rendezvous on 127.0.0.1, gloo world of N, GatherPlan, barrier, max-over-ranks timing, one JSON line from rank 0."""
import json
import os
import subprocess
import sys

from common import ROOT


def _run(args, env_extra, timeout=300):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_bench_launches_its_own_ranks():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"], {"MONORTM_BENCH_STUB": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["n_ranks_seen"] == 2 and j["steps"] == 3 and j["stub"] is True


def test_bench_single_rank_stub_needs_no_launcher():
    r = _run(["--steps", "2"], {"MONORTM_BENCH_STUB": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip())["n_gpus"] == 1


def test_world_size_mismatch_is_refused():
    r = _run(["--gpus", "4", "--steps", "1"], {"MONORTM_BENCH_STUB": "1", "WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr
