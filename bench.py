#!/usr/bin/env python3
"""Benchmark of the MASt3R-SLAM per-frame hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--pairs-per-gpu P]
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One step = one pass of the hot path over one batch of P synthetic keyframe pairs per GPU:
  two-view network (ViT-L encoder on 2P images, two 12-block decoders, DPT + feature heads)
  -> dense matching (prep -> iter_proj -> refine_matches -> occlusion test)
  -> Gauss-Newton Sim(3) tracking solve (10 iterations) per pair
  [N > 1] -> RCCL all-gather of the per-pair results (pointmaps, confidences, indices, validity).
Pairs are independent: each rank works on its own P pairs (weak scaling), the only collective
is the result all-gather.  Inputs are resident in HBM before the timed region.  Weights are
seeded random (no checkpoint can be fetched), data is synthetic - both stated in the JSON.

Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` (dominant kernel =
the bf16 MFMA GEMM, algorithmic FLOPs / HIP-event time per launch) and `cpu_baseline` (the CPU
oracle timed on this host, rank 0, N=1 only, on one pair).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "mast3r-slam_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

H = W = 512
MFMA_BF16_PEAK_TFLOPS = 2500.0      # dense bf16, MI355X_MICROARCH.md "Chip-level parameters"
HBM_PEAK_GBS = 8000.0               # HBM3E peak, same table (6.29 TB/s measured by a float4 copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs-per-gpu", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) and run the result all-gather even with one rank")
    ap.add_argument("--no-b1", action="store_true", help="skip the extra batch=1 (BASELINE configs[1]) latency measurement")
    return ap.parse_args()


def pmc_traffic(prefixes):
    """HBM bytes per launch of the kernels whose name starts with one of `prefixes`, from the newest committed PMC
    summary of this command (FETCH_SIZE and WRITE_SIZE cannot be collected inside the timed run: separate rocprofv3
    passes, tools/prof_summary.py).  (None, None) when no summary is present."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, None
    try:
        kern = json.load(open(files[-1]))["kernels"]
    except (OSError, ValueError, KeyError):
        return None, None
    sel = [v for k, v in kern.items() if k.startswith(prefixes)]
    n = sum(v["launches"] for v in sel)
    return (sum(v["hbm_bytes_per_launch"] * v["launches"] for v in sel) / n if n else None), os.path.basename(files[-1])


def make_inputs(synthetic, pairs, rank, dev):
    base = rank * pairs
    im1 = np.stack([synthetic.textured_image(H, W, 2 * (base + p)) for p in range(pairs)])
    im2 = np.stack([synthetic.textured_image(H, W, 2 * (base + p) + 1) for p in range(pairs)])
    return torch.from_numpy(im1).to(dev), torch.from_numpy(im2).to(dev)


def make_scene(synthetic, pairs, rank, dev):
    """What the matcher and the Gauss-Newton solve run on: P smooth two-view scenes of
    the benchmark size - pointmaps of both views in the frame's coordinates, 24-d descriptors, confidences and
    descriptor confidences drawn so that the tracker's gates (tracker.py:108-113: C > 0, Q > 1.5) pass for most
    points, and the keyframe's own canonical pointmap Xk = T * X21 under a known small Sim(3) (2 degrees about y,
    t = (0.05, 0, 0.01), s = 1.02) that the solve has to recover.  Random-init weights give neither matchable
    geometry nor confidences above the gates, so feeding the network's outputs here would time a matcher on
    white noise and a solver on ~0 valid points (round-1 verdict)."""
    base = rank * pairs
    sc = synthetic.geometric_pair(H, W, seed=1000 + base, batch=pairs)
    rng = np.random.default_rng(2000 + base)
    n = H * W
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ang = np.deg2rad(2.0)
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    Xk = (1.02 * sc["X21"].reshape(pairs, n, 3).astype(np.float64) @ R.T + np.array([0.05, 0.0, 0.01])).astype(np.float32)
    u = lambda lo, hi: rng.uniform(lo, hi, size=(pairs, n)).astype(np.float32)
    return dict(X11=t(sc["X11"]), X21=t(sc["X21"]), D11=t(sc["D11"]), D21=t(sc["D21"]), Xk=t(Xk),
                Cf=t(u(1.0, 3.0)), Ck=t(u(1.0, 3.0)), Qf=t(u(1.0, 4.0)), Qk=t(u(1.0, 4.0)),
                T_true=np.array([0.05, 0.0, 0.01, 0.0, np.sin(ang / 2), 0.0, np.cos(ang / 2), 1.02]))


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (the HIP path is the product; no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from mast3r_slam import _ffi, config, matching, model as model_mod, ops, synthetic, tracker
    from mast3r_slam import dist as m3dist
    if not os.path.exists(_ffi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()

    P = args.pairs_per_gpu
    config.set_config({"matching": {"use_simple": False}})            # the iter_proj + refine matcher
    net = model_mod.Mast3rFull(seed=0, device=dev, precision="bf16")
    im1, im2 = make_inputs(synthetic, P, rank, dev)
    sc = make_scene(synthetic, P, rank, dev)
    ident = torch.tensor([0, 0, 0, 0, 0, 0, 1, 1], dtype=torch.float32, device=dev)
    n = H * W
    tcfg = config.get_config()["tracking"]
    ev = lambda: torch.cuda.Event(enable_timing=True)

    # ---- the three legs of a step.  Each writes into tensors that stay alive (graph-static buffers). ----
    def leg_infer():
        o1, o2 = net.reconstruct_batch(im1, im2)
        return o1, o2

    def leg_match():
        return matching.match(sc["X11"], sc["X21"], sc["D11"], sc["D21"])

    def leg_gn(idx, valid):
        # FrameTracker.track's data flow (tracker.py:88-123, :177-214): frame = view 1, keyframe = view 2;
        # gather the frame's points at the match index, gate on confidences, then the 10-iteration solve of
        # all P problems in one launch sequence
        Xf, Qk, vo, vk, cnt = tracker.track_gather(sc["X11"].reshape(P, n, 3), sc["Cf"], sc["Ck"], sc["Qf"], sc["Qk"],
                                                   idx, valid.reshape(P, n), tcfg["C_conf"], tcfg["Q_conf"])
        poses, T_rel, info = tracker.opt_pose_ray_dist_sim3(Xf, sc["Xk"], ident, ident, Qk, vo, tcfg, fixed_iters=True)
        return poses, T_rel, info, vo

    def wire(o1, o2, idx, valid, poses):
        """What travels: pointmaps + confidences fp32, match index int32, validity u8, poses."""
        return (o1["pts3d"], o2["pts3d"], o1["conf"], o2["conf"], idx.to(torch.int32), valid, poses)

    # eager warm-up (lazy allocations, attribute setup), then capture each leg into its own hipGraph: the step
    # is ~1600 launches, replay removes the host launch path (a B=1 step is launch-bound when issued eagerly).
    # Three graphs instead of one so that stream events BETWEEN the replays give device time per stage inside
    # the timed region itself.
    for _ in range(2):
        o1, o2 = leg_infer(); idx, valid = leg_match(); gn = leg_gn(idx, valid)
    torch.cuda.synchronize()
    graphs = None
    if not args.no_graph:
        try:
            graphs = [torch.cuda.CUDAGraph() for _ in range(3)]
            with torch.cuda.graph(graphs[0]):
                o1, o2 = leg_infer()
            with torch.cuda.graph(graphs[1]):
                idx, valid = leg_match()
            with torch.cuda.graph(graphs[2]):
                gn = leg_gn(idx, valid)
            torch.cuda.synchronize()
        except Exception as e:                                   # noqa: BLE001 - reported, never silent
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            graphs = None
    state = {"o1": o1, "o2": o2, "idx": idx, "valid": valid, "gn": gn}
    marks = []           # per timed step: 4 events (start, after infer, after match, after GN)
    pending = []

    def run_step(timed=False):
        m = [ev() for _ in range(4)] if timed else None
        if m: m[0].record()
        if graphs is not None:
            graphs[0].replay()
        else:
            state["o1"], state["o2"] = leg_infer()
        if m: m[1].record()
        if graphs is not None:
            graphs[1].replay()
        else:
            state["idx"], state["valid"] = leg_match()
        if m: m[2].record()
        if graphs is not None:
            graphs[2].replay()
        else:
            state["gn"] = leg_gn(state["idx"], state["valid"])
        if m:
            m[3].record()
            marks.append(m)
        if dist is not None:
            # snapshot the result buffers (pack = one cat kernel) and let RCCL gather the snapshot on its own
            # stream while the next step computes; at most one gather in flight
            if pending:
                pending.pop().wait()
            pending.append(m3dist.all_gather_results(wire(state["o1"], state["o2"], state["idx"], state["valid"],
                                                          state["gn"][0]), async_op=True))

    for _ in range(args.warmup):
        run_step()

    def barrier():
        if dist is not None:
            if pending:
                pending.pop().wait()                               # the last step's gather belongs to the timed region
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step(timed=True)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    stage_ms = {k: sum(m[i].elapsed_time(m[i + 1]) for m in marks) / len(marks)
                for i, k in enumerate(("infer", "match", "gn"))}
    stage_ms["sum"] = sum(stage_ms.values())

    # ---- did the match and GN legs do the work they name? ------------------------------------------------
    poses, T_rel, info, vo = state["gn"]
    match_valid_frac = float(state["valid"].float().mean())
    valid_frac = float(vo.float().mean())
    pose_err = float(np.abs(T_rel.double().cpu().numpy() - sc["T_true"][None]).max())
    if match_valid_frac < 0.5 or valid_frac < 0.5:
        raise SystemExit(f"bench invalid: match_valid_frac={match_valid_frac:.3f} valid_frac={valid_frac:.3f} (< 0.5): "
                         "the matcher / Gauss-Newton legs would be timed on rejected points")

    # ---- untimed instrumented pass (eager, launches serialised): per-launch device time by kernel family ----
    ops.PROFILE = []
    _ffi.PROFILE = {}
    _ffi.PROFILE_NAMES = ("m3_prep_iter_proj", "m3_iter_proj", "m3_refine_matches", "m3_match_epilogue",
                          "m3_track_gather_batch", "m3_track_gn_ray_dist_batch")
    leg_infer(); i2, v2 = leg_match(); leg_gn(i2, v2)           # single stream: the launches are serialised
    torch.cuda.synchronize()
    prof, ops.PROFILE = ops.PROFILE, None
    cprof, _ffi.PROFILE = _ffi.PROFILE, None
    by_kind = {}
    for kind, flops, e0, e1, nbytes in prof:
        d = by_kind.setdefault(kind, [0.0, 0.0, 0, 0.0])
        d[0] += flops; d[1] += e0.elapsed_time(e1) * 1e-3; d[2] += 1; d[3] += nbytes
    g = by_kind.get("gemm256", [0.0, 1.0, 1, 0.0])
    traffic, traffic_src = pmc_traffic(("k_gemm256<0",))
    gemm_tflops = g[0] / g[1] / 1e12
    model_flops = net.flops_per_pair(H, W) * P
    # HBM-bound kernel families: algorithmic bytes per call from SURVEY 8d (per pair at 512x512, fp32) x P pairs
    MB = 1e6
    iters = int(tcfg["max_iters"])
    hbm_rows = {
        "m3_prep_iter_proj": ("k_prep", 21.0 * MB * P, 1),
        "m3_iter_proj": ("k_iter_proj (+ k_iter_reduce / k_iter_limit, early-stop second pass)", 17.0 * MB * P, 1),
        "m3_refine_matches": ("k_refine_lds<24> / k_refine<24>", 54.5 * MB * P, 1),
        "m3_match_epilogue": ("k_epilogue", 9.7 * MB * P, 1),
        "m3_track_gather_batch": ("k_track_gather", (3.15 + 4 * 1.05 + 2.1 + 0.26 + 3.15 + 1.05 + 0.52) * MB * P, 1),
        "m3_track_gn_ray_dist_batch": (f"k_track_accum + k_track_solve, per GN iteration ({iters} per call)", 8.7 * MB * P, iters),
    }
    hbm = {}
    for name, (kern, nbytes, div) in hbm_rows.items():
        evs = cprof.get(name, [])
        if not evs:
            continue
        us = sum(a.elapsed_time(b) for a, b in evs) / len(evs) / div * 1e3
        hbm[name] = {"kernels": kern, "algorithmic_bytes": nbytes, "avg_us": us, "achieved": nbytes / us / 1e3,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / us / 1e3 / HBM_PEAK_GBS}

    result = {
        "metric": "keyframe-pairs/sec (512x512 two-view infer+match+GN)",
        "value": world * P * args.steps / elapsed,
        "unit": "pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic: 512x512 textured pairs through the network (seeded random-init weights, no checkpoint available "
                "offline); matcher + Gauss-Newton on smooth synthetic two-view scenes of the same size (SURVEY 8d configs 2-3)",
        "config": {"workload": f"{P} keyframe pairs/GPU at 512x512 (BASELINE configs[3] per-GPU shard): "
                               "two-view MASt3R ViT-L infer (bf16 trunk, fp16 heads, fp32 accumulate) + iter_proj/refine match "
                               "+ 10-iter GN tracking" + ("" if world == 1 else " + RCCL all-gather of results"),
                   "pairs_per_gpu": P, "global_pairs": world * P, "image": [H, W], "gn_iters": iters,
                   "parallelism": f"pair-sharded x{world}",
                   "launch": ("3 hipGraph replays per step (infer | match | GN)" + ("" if dist is None else " + RCCL all-gather of the previous step overlapped on the communicator stream")) if graphs is not None else "eager"},
        "stage_ms": {k: round(v, 3) for k, v in stage_ms.items()},
        "stage_ms_note": "device time between stream events recorded around the three graph replays of every TIMED step (mean); sum ~ ms_per_step",
        "match_valid_frac": round(match_valid_frac, 4),
        "valid_frac": round(valid_frac, 4),
        "gn_pose_max_abs_err_vs_true_sim3": pose_err,
        "model_tflop_per_step": model_flops / 1e12,
        "roofline": {"bound": "mfma", "kernel": "k_gemm256 (16-bit MFMA GEMM, 256x256x64 / 256x192x64 ping-pong tiles; dense launches only - "
                                                "its implicit-GEMM conv launches and the small-problem kernel are listed under other_kernels_tflops)",
                     "achieved": gemm_tflops, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": gemm_tflops / MFMA_BF16_PEAK_TFLOPS,
                     "traffic": traffic, "traffic_unit": "HBM bytes per launch (rocprofv3 --pmc FETCH_SIZE x2 / WRITE_SIZE passes "
                                                         f"of this command, profiles/{traffic_src})",
                     "algorithmic_bytes_per_launch": g[3] / max(g[2], 1),
                     "launches": g[2], "avg_launch_us": g[1] / max(g[2], 1) * 1e6,
                     "other_kernels_tflops": {k: v[0] / v[1] / 1e12 for k, v in by_kind.items() if k != "gemm256"},
                     "all_mfma_kernels_tflops": sum(v[0] for v in by_kind.values()) / sum(v[1] for v in by_kind.values()) / 1e12},
        "hbm_rooflines": hbm,
    }

    if world == 1 and not args.no_b1:
        # for the record: the same matcher on what the random-weight network emits (scattered gathers)
        o1, o2 = state["o1"], state["o2"]
        for _ in range(2):
            matching.match(o1["pts3d"], o2["pts3d"], o1["desc"], o2["desc"])
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(5):
            matching.match(o1["pts3d"], o2["pts3d"], o1["desc"], o2["desc"])
        e1.record(); torch.cuda.synchronize()
        result["match_ms_on_random_weight_network_output"] = round(e0.elapsed_time(e1) / 5, 3)

    if world == 1 and not args.no_b1:
        # "fp16 features" (BASELINE configs[4]): the same dense matcher with both descriptor maps stored as half -
        # m3_refine_matches_f16 moves 29.3 instead of 54.5 MB per map, same fp32 scoring
        D11h, D21h = sc["D11"].half(), sc["D21"].half()
        _ffi.PROFILE = {}
        _ffi.PROFILE_NAMES = ("m3_refine_matches_f16",)
        for _ in range(3):
            i16, v16 = matching.match(sc["X11"], sc["X21"], D11h, D21h)
        torch.cuda.synchronize()
        evs, _ffi.PROFILE = _ffi.PROFILE.get("m3_refine_matches_f16", []), None
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(5):
            i16, v16 = matching.match(sc["X11"], sc["X21"], D11h, D21h)
        e1.record(); torch.cuda.synchronize()
        idx, valid = state["idx"], state["valid"]
        both = (valid & v16)[..., 0]
        us16 = sum(a.elapsed_time(b) for a, b in evs[1:]) / max(len(evs) - 1, 1) * 1e3
        result["fp16_features"] = {"match_ms": round(e0.elapsed_time(e1) / 5, 3),
                                   "match_valid_frac": round(float(v16.float().mean()), 4),
                                   "idx_agreement_with_fp32_features": round(float((idx == i16)[both].float().mean()), 5),
                                   "m3_refine_matches_f16": {"algorithmic_bytes": 29.3 * MB * P, "avg_us": us16,
                                                             "achieved": 29.3 * MB * P / us16 / 1e3, "unit": "GB/s",
                                                             "frac": 29.3 * MB * P / us16 / 1e3 / HBM_PEAK_GBS}}

    if world == 1 and not args.no_b1:
        # matcher variant named by north_star: fast reciprocal NN (MASt3R sec. 3.3) on the same scene, 64 x 64 seeds
        # (subsample 8), fp16 descriptors, device-side loop (3 rounds), per pair; not part of the timed step
        d1, d2 = sc["D21"][0].half(), sc["D11"][0].half()
        for _ in range(2):
            matching.fast_reciprocal_nn_device(d1, d2, subsample=8, max_iter=3)
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(3):
            p1, p2 = matching.fast_reciprocal_nn_device(d1, d2, subsample=8, max_iter=3)
        e1.record(); torch.cuda.synchronize()
        result["fast_nn_matcher"] = {"ms_per_pair": round(e0.elapsed_time(e1) / 3, 3), "reciprocal_pairs": int(p1.numel()),
                                     "seeds": 4096, "note": "m3_nn_search_mfma, fp16 descriptors, 3 rounds x 2 searches of 4096 x 262144 x 24"}

    if world == 1 and not args.no_b1 and P != 1:
        # BASELINE configs[1]: one pair per step (latency regime), same pipeline, graph-replayed
        a1, b1 = im1[:1].contiguous(), im2[:1].contiguous()
        s1 = {k: (v[:1].contiguous() if isinstance(v, torch.Tensor) else v) for k, v in sc.items()}

        def step1():
            net.reconstruct_batch(a1, b1)
            i1, v1 = matching.match(s1["X11"], s1["X21"], s1["D11"], s1["D21"])
            Xf, Qk, vo1, vk, cnt = tracker.track_gather(s1["X11"].reshape(1, n, 3), s1["Cf"], s1["Ck"], s1["Qf"], s1["Qk"],
                                                        i1, v1.reshape(1, n), tcfg["C_conf"], tcfg["Q_conf"])
            return tracker.opt_pose_ray_dist_sim3(Xf, s1["Xk"], ident, ident, Qk, vo1, tcfg, fixed_iters=True)
        for _ in range(2):
            step1()
        torch.cuda.synchronize()
        run1 = step1
        if graphs is not None:
            g1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1):
                keep = step1()
            run1 = g1.replay
        run1(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(10):
            run1()
        torch.cuda.synchronize()
        ms1 = (time.perf_counter() - t1) / 10 * 1e3
        result["batch1"] = {"workload": "BASELINE configs[1]: 1 pair/step at 512x512, same pipeline", "pairs_per_s": 1e3 / ms1,
                            "ms_per_pair": ms1}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args, net)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _host_threads(requested: int) -> int:
    if requested:
        return requested
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))            # the GPU box grants a 16-core share per GPU


def cpu_baseline(args, net):
    """The CPU oracle (kind "port": our restatement, pinned to the reference's numpy twins for matching;
    torch fp32 for the network) timed on this host on ONE complete 512x512 pair - the same work the GPU
    does per pair: full encoder (2 views), both decoders, both heads, prep + iter_proj + refine_matches
    over all points, 10 GN iterations.  About 10 s on 16 cores."""
    from mast3r_slam import synthetic
    from oracle import matching as om
    from oracle import model as omodel
    from oracle import tracking as ot
    threads = _host_threads(args.cpu_threads)
    torch.set_num_threads(threads)
    note = lambda s: print(f"[cpu_baseline] {s}", file=sys.stderr, flush=True)
    w, cfg = net.host_weights, net.cfg
    im = torch.from_numpy(np.stack([synthetic.textured_image(H, W, 0), synthetic.textured_image(H, W, 1)]))
    with torch.no_grad():
        t0 = time.perf_counter()
        f, pos = omodel.encode(w, im, cfg)
        t_enc = time.perf_counter() - t0
        note(f"encoder done ({t_enc:.1f} s)")
        t0 = time.perf_counter()
        o1, o2 = omodel.decode(w, f[:1], f[1:], pos, cfg)
        t_dec = time.perf_counter() - t0
        note(f"decoder done ({t_dec:.1f} s)")
        t0 = time.perf_counter()
        r1 = omodel.head(w, "downstream_head1", o1, H, W, tuple(cfg["hooks"]))
        r2 = omodel.head(w, "downstream_head2", o2, H, W, tuple(cfg["hooks"]))
        t_head = time.perf_counter() - t0
        note(f"heads done ({t_head:.1f} s)")
    X11, X21 = r1["pts3d"].numpy(), r2["pts3d"].numpy()
    D11, D21 = r1["desc"].numpy(), r2["desc"].numpy()
    n = H * W
    t0 = time.perf_counter()
    rays, tgt, p0 = om.prep_for_iter_proj(X11, X21, None)
    p, vproj = om.iter_proj(rays, tgt, p0, 10, 1e-8, 1e-6, "batch")
    pi = om.refine_matches(D11, D21.reshape(1, n, -1), p.astype(np.int32), 3, 2)
    t_match = time.perf_counter() - t0
    note(f"matching done ({t_match:.1f} s)")
    ident = np.array([0, 0, 0, 0, 0, 0, 1, 1], dtype=np.float64)
    t0 = time.perf_counter()
    ot.opt_pose_ray_dist_sim3(X11.reshape(n, 3), X21.reshape(n, 3), ident, ident, np.full(n, 2.0),
                              np.ones(n, bool), fixed_iters=10)
    t_gn = time.perf_counter() - t0
    note(f"GN done ({t_gn:.1f} s)")
    total = t_enc + t_dec + t_head + t_match + t_gn
    return {"value": 1.0 / total, "unit": "pairs/s", "cores": threads, "kind": "port",
            "sample": "1 complete pair 512x512: ViT-L encoder x2, both decoders, DPT + feature heads (torch-CPU fp32), "
                      "prep + iter_proj + refine_matches on all 262144 points (numpy oracle), 10 GN iterations (float64)",
            "seconds": {"encoder": round(t_enc, 2), "decoder": round(t_dec, 2), "heads": round(t_head, 2),
                        "match": round(t_match, 2), "gn": round(t_gn, 2)}}


if __name__ == "__main__":
    main()
