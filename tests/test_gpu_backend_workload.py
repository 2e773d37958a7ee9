"""BASELINE configs[3] and configs[4] at the size of their per-GPU shard / at full edge size.

  * the driver's pairs workload at 8 pairs per GPU (configs[3]'s shard), reduced-depth network at 512 x 512, N > 1 branch under
    a one-rank RCCL group;
  * the backend workload (configs[4]: edge-sharded FactorGraph re-match + blocks + solve) end to end on a small graph;
  * the block kernel at configs[4]'s edge size: 64 keyframes x 262 144 points, 372 directed edges (the SURVEY 8d config-5
    graph cut to what one test can hold), sample edges against oracle/gn_rays.edge_blocks in float64."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from mast3r_slam import kernels, synthetic
from oracle import gn_rays as og

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--no-b1", "--no-cpu-baseline", "--force-dist", *extra],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_pairs_workload_at_the_config4_shard_size():
    res = _bench(["--model", "tiny", "--pairs-per-gpu", "8"], 29621)
    assert res["n_gpus"] == 1 and res["config"]["pairs_per_gpu"] == 8 and res["config"]["image"] == [512, 512]
    assert res["ranks"]["backend"] == "nccl" and "RCCL all-gather" in res["config"]["launch"]
    assert res["match_valid_frac"] > 0.5 and res["valid_frac"] > 0.5 and res["gn_pose_max_abs_err_vs_true_sim3"] < 5e-3
    assert res["value"] > 0 and abs(res["value"] - 8 / (res["ms_per_step"] * 1e-3)) < 1e-6 * res["value"]


def test_pairs_workload_with_the_fast_reciprocal_nn_matcher():
    """--matcher fast_nn: the match leg is the MFMA fast-reciprocal-NN matcher (the matcher BASELINE.json's north_star
    names) feeding the same gather + 10-iteration solve through sparse (index, validity) maps: most seeds must find a
    reciprocal match on the smooth scene and the solve must recover the scene's Sim(3) from them (bench.py exits
    non-zero otherwise)."""
    res = _bench(["--model", "tiny", "--pairs-per-gpu", "4", "--matcher", "fast_nn"], 29626)
    assert res["config"]["matcher"] == "fast_nn" and "reciprocal" in res["config"]["workload"]
    seeds = 64 * 64
    assert 0.5 * seeds / (512 * 512) < res["match_valid_frac"] <= seeds / (512 * 512)
    assert res["gn_pose_max_abs_err_vs_true_sim3"] < 2e-2
    assert res["stage_ms"]["match"] > 0 and res["value"] > 0


def test_default_line_carries_a_backend_block():
    """The driver runs the default command only, so BASELINE configs[4] is part of ITS line: after the timed region the
    pairs workload runs 16 undirected edges of the 256-keyframe graph through the full network (symmetric decode from cached
    tokens + both match directions on fp16 features), the rays-GN blocks and the 1785-unknown dense step (reference flow
    slam.py:292-319 -> global_opt.py:49-166).  The block must be there, on valid matches, and its one Gauss-Newton iteration
    must at least halve the pose error of the keyframes it constrains."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert res["config"]["pairs_per_gpu"] == 8 and "roofline" in res and "batch1" in res
    b = res["backend"]
    assert "256 keyframes" in b["workload"] and b["edges_kept"] == 16 and b["match_valid_frac"] > 0.9
    assert b["edges_per_s"] > 0 and b["stage_ms"]["rematch"] > 0 and b["stage_ms"]["solve"] > 0
    e0, e1 = b["pose_max_abs_err_before_after"]
    assert e1 < 0.5 * e0, (e0, e1)
    blk = b["hbm_rooflines"]["m3_gn_rays_blocks"]
    assert 0 < blk["frac"] < 1 and blk["unit"] == "GB/s"
    assert b["dense_step"]["unknowns"] == 7 * 255 and b["dense_step"]["avg_us"] > 0


def test_backend_workload_small_graph_end_to_end():
    """12 keyframes, 6 undirected edges on this rank, reduced-depth network, 256 x 256: decode from cached tokens, both
    matching directions with fp16 features, blocks, one-rank RCCL gather of the blocks, dense step."""
    res = _bench(["--workload", "backend", "--model", "tiny", "--keyframes", "12", "--edges-per-gpu", "6", "--edge-batch", "4",
                  "--image", "256", "256", "--gn-iters", "3"], 29622)
    assert res["unit"] == "edges/s" and res["n_gpus"] == 1 and res["config"]["edges_per_gpu"] == 6
    assert res["edges_kept"] == 6 and res["match_valid_frac"] > 0.7     # 256 x 256: the 40-px trajectory moves ~15 % of the pixels out of view
    e0, e1 = res["pose_max_abs_err_before_after"]
    assert e1 < 0.5 * e0                                      # three GN iterations pulled the perturbed poses towards the truth
    assert abs(res["value"] - 6 / (res["ms_per_step"] * 1e-3)) < 1e-6 * res["value"]
    h = res["hbm_rooflines"]
    assert 0 < h["m3_gn_rays_blocks"]["frac"] < 1 and 0 < h["m3_refine_matches_f16"]["frac"] < 1
    assert res["dense_step"]["unknowns"] == 7 * 11 and res["dense_step"]["avg_us"] > 0


def test_config5_edge_size_blocks_vs_oracle(dev):
    """64 keyframes x 262 144 points, each linked to its previous three in both directions (372 directed edges, 4 GB of
    per-edge arrays - BASELINE configs[4]'s edge size).  idx = the scene's true correspondences rounded to pixels."""
    K_, h, w = 64, 512, 512
    n = h * w
    sc = synthetic.keyframe_graph_scene(K_, h, w, dev, seed=3)
    ii_u, jj_u = synthetic.chain_edges(K_)
    ii = torch.tensor(ii_u + jj_u, dtype=torch.int32, device=dev)
    jj = torch.tensor(jj_u + ii_u, dtype=torch.int32, device=dev)
    e = ii.numel()
    assert e == 372
    vv, uu = torch.meshgrid(torch.arange(h, device=dev), torch.arange(w, device=dev), indexing="ij")
    sh = sc["shift"]
    d = sh[jj.long()] - sh[ii.long()]                            # pixel of keyframe jj -> pixel of keyframe ii
    u = (uu.reshape(1, n) + torch.round(d[:, 0:1])).clamp(0, w - 1).long()
    v = (vv.reshape(1, n) + torch.round(d[:, 1:2])).clamp(0, h - 1).long()
    idx = (u + w * v).to(torch.int32)
    g = torch.Generator(device="cpu").manual_seed(1)
    valid = (torch.rand((e, n), generator=g) > 0.3).to(dev)
    Q = (torch.rand((e, n), generator=g) * 3 + 1).to(dev)
    poses = sc["poses"].clone()
    poses[1:, :3] += (torch.randn((K_ - 1, 3), generator=g) * 0.01).to(dev)
    blocks = kernels.gn_rays_blocks(poses, sc["Xs"], sc["C"], ii, jj, idx, valid, Q).cpu().numpy()
    assert blocks.shape == (e, 36) and np.isfinite(blocks).all()
    P = poses.cpu().numpy().astype(np.float64)
    Xs, Cs = sc["Xs"].cpu().numpy(), sc["C"].cpu().numpy()
    iu = np.triu_indices(7)
    for k in (0, 200, e - 1):
        Hjj, gj, cnt = og.edge_blocks(P[:, :3], P[:, 3:7], P[:, 7], Xs, Cs, int(ii[k]), int(jj[k]), idx[k].cpu().numpy(),
                                      valid[k].cpu().numpy(), Q[k].cpu().numpy())
        assert blocks[k, 35] == cnt and cnt > 0.5 * n
        # float32 per-point terms summed over ~180k points in float64 (same bounds as the 65 536-point test)
        assert np.abs(blocks[k, :28] - Hjj[iu]).max() <= 2e-4 * np.abs(Hjj).max()
        assert np.abs(blocks[k, 28:35] - gj).max() <= 2e-3 * np.abs(gj).max() + 1e-3
    # size-independent properties at this size: reproducible bits, and an edge evaluated alone equals its row
    again = kernels.gn_rays_blocks(poses, sc["Xs"], sc["C"], ii, jj, idx, valid, Q).cpu().numpy()
    assert np.array_equal(again, blocks)
    one = kernels.gn_rays_blocks(poses, sc["Xs"], sc["C"], ii[200:201], jj[200:201], idx[200:201], valid[200:201], Q[200:201])
    assert np.abs(one.cpu().numpy()[0] - blocks[200]).max() <= 1e-9 * np.abs(blocks[200]).max()


def _bench_two_ranks(extra):
    """`python bench.py --gpus 2` (the self-launcher) with both ranks on THIS GPU: RCCL refuses two ranks on one device, so
    the collectives go through gloo (--dist-backend gloo --single-device, test hooks); everything else - the launcher, the
    sharding of pairs / edges, the device kernels on each rank, the gathered-block solve - is the N > 1 code the driver's
    multi-GPU run executes."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-b1", "--no-cpu-baseline", "--dist-backend", "gloo", "--single-device", "--model", "tiny", *extra],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_two_ranks_pairs_workload():
    res = _bench_two_ranks(["--pairs-per-gpu", "2", "--image", "256", "256"])
    assert res["n_gpus"] == 2 and res["ranks"]["world_size_of_the_process_group"] == 2 and res["ranks"]["backend"] == "gloo"
    assert res["config"]["global_pairs"] == 4 and res["value"] > 0
    assert res["match_valid_frac"] > 0.5 and res["valid_frac"] > 0.5


def test_two_ranks_edge_sharded_backend_workload():
    """12 keyframes, 2 x 6 edges: each rank matches and linearises only its 6 edges, the 36-double blocks of all 24
    directed edges are gathered, both ranks run the same dense step - and the solve moves the perturbed poses of the WHOLE
    graph (which no rank could do from its own edges alone) towards the truth."""
    res = _bench_two_ranks(["--workload", "backend", "--keyframes", "12", "--edges-per-gpu", "6", "--edge-batch", "3",
                            "--image", "256", "256", "--gn-iters", "3"])
    assert res["n_gpus"] == 2 and res["config"]["global_edges"] == 12 and res["config"]["edges_per_gpu"] == 6
    assert res["edges_kept"] == 12 and res["match_valid_frac"] > 0.7
    e0, e1 = res["pose_max_abs_err_before_after"]
    assert e1 < 0.5 * e0
    assert abs(res["value"] - 12 / (res["ms_per_step"] * 1e-3)) < 1e-6 * res["value"]
