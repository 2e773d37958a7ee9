"""The code the driver launches: bench.py end to end on one GPU, including the N > 1 branch of its step
(graph replays + packed snapshot + asynchronous RCCL all-gather + hand-over of the receive views) under a ONE-rank RCCL
process group - the builder has one GPU at a time, so this is the only way that branch executes before the
driver's 8-GPU run.  Runs in a child process (its own process group, its own HIP context)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--pairs-per-gpu", "1", "--no-b1", "--no-cpu-baseline", *extra],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_bench_step_with_rccl_gather_one_rank():
    res = _run(["--force-dist"])
    assert res["n_gpus"] == 1 and res["value"] > 0 and "RCCL all-gather" in res["config"]["launch"]
    ex = res["exchange"]                     # the pre-allocated packed exchange returned this rank's own results (views, no copies)
    assert ex["own_row_equals_local_results"] and ex["fields"] == 7 and ex["gathered_shapes"][0][:2] == [1, 1]
    assert res["match_valid_frac"] > 0.5 and res["valid_frac"] > 0.5          # the legs ran on accepted points
    assert res["gn_pose_max_abs_err_vs_true_sim3"] < 5e-3                      # ... and the solve found the scene's Sim(3)
    s = res["stage_ms"]
    assert abs(s["sum"] - (s["infer"] + s["match"] + s["gn"])) < 1e-2
    assert 0.5 * res["ms_per_step"] < s["sum"] < 1.05 * res["ms_per_step"]     # device stage times add up to the step
    for k in ("m3_iter_proj", "m3_refine_matches", "m3_track_gn_ray_dist_batch"):
        assert 0 < res["hbm_rooflines"][k]["frac"] < 1
    r = res["roofline"]                      # at 1 pair/GPU the dense GEMMs take the small-tile kernel: check the MFMA total
    assert r["bound"] == "mfma" and 0 < r["all_mfma_kernels_tflops"] < r["peak"]
