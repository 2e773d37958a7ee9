"""CPU: `python bench.py --gpus N` starts N ranks itself (no WORLD_SIZE in the environment), the ranks rendezvous on
127.0.0.1, time the same K steps between barriers, take the max over ranks and rank 0 prints ONE JSON line.  The step
is the --stub hook (gloo, no device): this covers the launcher and the rank-side harness, not the kernels."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(kw)
    return env


def test_gpus_2_spawns_two_ranks_and_reports_their_world_size():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--stub", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                            # rank 0 only
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 3 and res["warmup"] == 1 and res["scaling"] == "weak"
    assert res["ranks"]["world_size_of_the_process_group"] == 2 and res["ranks"]["gpus_argument"] == 2
    assert "self-spawned" in res["ranks"]["started_by"] and res["ranks"]["backend"] == "gloo"
    assert res["value"] > 0 and abs(res["value"] - 2 * 4 * 3 / (res["ms_per_step"] * 3e-3)) < 1e-6 * res["value"]


def test_external_launcher_environment_and_gpus_mismatch():
    ok = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--stub", "--steps", "2", "--warmup", "0"],
                        capture_output=True, text=True, timeout=300, cwd=ROOT,
                        env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29655"))
    assert ok.returncode == 0, ok.stderr[-2000:]
    res = json.loads([l for l in ok.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 1 and "external launcher" in res["ranks"]["started_by"]
    bad = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--stub", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT,
                         env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29656"))
    assert bad.returncode != 0 and "--gpus 4 but WORLD_SIZE=1" in bad.stderr


def test_a_failing_rank_fails_the_launcher():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--stub", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=_env(M3_BENCH_STUB_FAIL_RANK="1"), cwd=ROOT)
    assert r.returncode != 0
    assert "rank 1 exited with status" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
