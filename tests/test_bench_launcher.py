"""`python bench.py --gpus N` must start its own ranks (VERDICT r01 / contract: the driver may call it that way) and must
refuse a WORLD_SIZE that contradicts --gpus.  Here (no GPU) the ranks run the stub step (MONORTM_BENCH_STUB=1): launcher,
rendezvous on 127.0.0.1, gloo world of N, GatherPlan, barrier, max-over-ranks timing, one JSON line from rank 0."""
import json
import os
import subprocess
import sys

import pytest

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


def test_gather_every_k_steps(tmp_path):
    """--gather-every 3 on two gloo ranks: 7 steps carry two collectives (steps 3 and 6), the compact line says so."""
    r = _run(["--gpus", "2", "--steps", "7", "--warmup", "1", "--gather-every", "3", "--detail-file", str(tmp_path / "d.json")], {"MONORTM_BENCH_STUB": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["n_ranks_seen"] == 2 and j["gather_every"] == 3 and j["gathers_in_timed_region"] == 2


def test_bench_single_rank_stub_needs_no_launcher(tmp_path):
    r = _run(["--steps", "2", "--detail-file", str(tmp_path / "d.json")], {"MONORTM_BENCH_STUB": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_last_stdout_line_is_compact_strict_json(tmp_path):
    """the driver keeps the last 8 KB of stdout; round 5's single 22 KB line left `parsed: null`.  The LAST line
    must be a small strict-JSON object with the contract's keys; the detail (here the stub's 16 KB of padding) goes to
    the detail file and to stderr."""
    det = tmp_path / "detail.json"
    r = _run(["--steps", "2", "--detail-file", str(det)], {"MONORTM_BENCH_STUB": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout.rstrip("\n").splitlines()
    last = out[-1]
    assert len(last) < 4096, len(last)
    j = json.loads(last, parse_constant=lambda c: pytest.fail(f"non-finite constant {c} in the compact line"))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "config"):
        assert k in j, k
    assert set(j["workloads"]) == {f"w{i}" for i in range(8)} and j["workloads"]["w3"] == {"value": 3.0, "ms_per_step": 1.0, "dtype": "f64", "lines_ms": None, "frac": 0.5}
    assert len(out) == 1                               # stdout carries the compact line and nothing else
    err = [ln for ln in r.stderr.splitlines() if ln.startswith("BENCH_DETAIL {")]
    assert len(err) == 1 and len(err[0]) > 16000       # the detail really is big ...
    full = json.load(open(det))
    assert full["workloads"]["w0"]["roofline"]["pad"] == "x" * 2000 and full["steps"] == 2   # ... and is kept whole beside it


def test_compact_line_of_a_real_record():
    """The compact form of round 5's committed 22 KB record (profiles/r05_z_bench.json): every key the contract names survives,
    NaN / Infinity do not, and the size limit holds."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_z_bench.json")))
    full["roofline"]["traffic"] = float("nan")          # a counter that failed must not break strict JSON
    full["workloads"]["c3"]["value"] = float("inf")
    c = bench.compact_line(full)
    line = json.dumps(c, allow_nan=False)
    assert len(line) < 4096, len(line)
    assert c["value"] == pytest.approx(full["value"], rel=1e-5) and c["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    assert c["config"]["workload"].startswith("c4") and c["config"]["wavenumbers"] == 50
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "kernel"):
        assert k in c["roofline"], k
    assert c["roofline"]["traffic"] is None and c["workloads"]["c3"]["value"] is None
    assert c["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-5)
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c["cpu_baseline"], k
    assert set(c["workloads"]) >= {"c3", "c5full", "c4shard", "c2lc", "c4brd"}
    assert c["workloads"]["c5full"]["frac"] == pytest.approx(full["workloads"]["c5full"]["roofline"]["frac"], rel=1e-5)


def test_world_size_mismatch_is_refused():
    r = _run(["--gpus", "4", "--steps", "1"], {"MONORTM_BENCH_STUB": "1", "WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr




@pytest.mark.gpu
def test_bench_two_ranks_gather_every_second_step():
    """--gather-every 2 with the real step on two ranks sharing the card: the block a gather reads is not rewritten while it is in
    flight (timed_steps waits before the step after next), the line says how often it gathered and what the gather costs alone."""
    r = _run(["--gpus", "2", "--steps", "4", "--warmup", "1", "--min-seconds", "0.2", "--gather-every", "2", "--no-pmc", "--no-cpu-baseline", "--no-extra"],
             {"MONORTM_BENCH_BACKEND": "gloo"}, timeout=360)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    j = json.loads(r.stdout.rstrip("\n").splitlines()[-1])
    assert j["n_gpus"] == 2 and j["n_ranks_seen"] == 2 and j["gather_every"] == 2 and j["gather_us_alone"] > 0
    evals = j["value"] * j["ms_per_step"] * 1e-3
    assert abs(evals - 1024 * 50 * 64 * 500) <= 1e-5 * evals, evals


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
    assert len(lines) == 1 and r.stdout.rstrip("\n").splitlines()[-1] == lines[0] and len(lines[0]) < 4096, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["n_ranks_seen"] == 2 and j["steps"] == 3 and j["scaling"] == "strong"
    assert j["gather_every"] == 1 and j["gather_us_alone"] > 0
    assert j["config"]["profiles_per_gpu"] == 512 and "1024" in j["config"]["workload"]
    full = json.loads([ln for ln in r.stderr.splitlines() if ln.startswith("BENCH_DETAIL ")][0][len("BENCH_DETAIL "):])
    assert len(full["rank_devices"]) == 2
    # configs[3] whole: 1024 profiles x 50 channels x 64 layers x 500 lines = 1.638e9 evals per step over both ranks
    evals = j["value"] * j["ms_per_step"] * 1e-3
    assert abs(evals - 1024 * 50 * 64 * 500) <= 1e-5 * evals, evals   # (the compact line carries six significant digits)
    assert j["profiles_per_sec"] > 0 and j["dtype"] == "f64" and "stub" not in j
