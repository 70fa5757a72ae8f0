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


import pytest  # noqa: E402


@pytest.mark.gpu
def test_bench_two_ranks_on_one_card_is_the_drivers_command():
    """The driver's 8-GPU command cannot be rehearsed here, but everything except the RCCL transport can: `bench.py --gpus 2`
    starts its two ranks itself, both on the one card of the test box (MONORTM_BENCH_BACKEND=gloo: RCCL refuses two ranks on
    one device), each takes its 512-profile block of configs[3], the spectral outputs are gathered to rank 0 every step by
    GatherPlan, and rank 0 prints ONE line: strong scaling, two ranks seen, evals of BOTH ranks in `value`."""
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--min-seconds", "0.2", "--no-pmc", "--no-cpu-baseline", "--no-extra"],
             {"MONORTM_BENCH_BACKEND": "gloo"}, timeout=360)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["n_ranks_seen"] == 2 and j["steps"] == 3 and j["scaling"] == "strong"
    assert j["config"]["profiles_per_gpu"] == 512 and "1024" in j["config"]["workload"]
    assert len(j["rank_devices"]) == 2
    # configs[3] whole: 1024 profiles x 50 channels x 64 layers x 500 lines = 1.638e9 evals per step over both ranks
    evals = j["value"] * j["ms_per_step"] * 1e-3
    assert abs(evals - 1024 * 50 * 64 * 500) <= 1e-6 * evals, evals
    assert j["profiles_per_sec"] > 0 and j["dtype"] == "f64" and "stub" not in j
