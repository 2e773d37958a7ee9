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


def pmc_traffic_per_launch(prefixes):
    """HBM bytes per launch of the dense GEMM kernels, from the committed PMC summary of this command
    (FETCH_SIZE and WRITE_SIZE cannot be collected inside the timed run: separate rocprofv3 passes,
    tools/prof_summary.py).  None when the summary is absent."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        kern = json.load(open(path))["kernels"]
    except (OSError, ValueError, KeyError):
        return None
    sel = [v for k, v in kern.items() if k.startswith(prefixes)]
    n = sum(v["launches"] for v in sel)
    return sum(v["hbm_bytes_per_launch"] * v["launches"] for v in sel) / n if n else None


def make_inputs(model_mod, synthetic, pairs, rank, dev):
    base = rank * pairs
    im1 = np.stack([synthetic.textured_image(H, W, 2 * (base + p)) for p in range(pairs)])
    im2 = np.stack([synthetic.textured_image(H, W, 2 * (base + p) + 1) for p in range(pairs)])
    return torch.from_numpy(im1).to(dev), torch.from_numpy(im2).to(dev)


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
    net = model_mod.Mast3rFull(seed=0, device=dev)
    im1, im2 = make_inputs(model_mod, synthetic, P, rank, dev)
    ident = torch.tensor([0, 0, 0, 0, 0, 0, 1, 1], dtype=torch.float32, device=dev)
    n = H * W
    tcfg = config.get_config()["tracking"]
    stage_ms = {"infer": 0.0, "match": 0.0, "gn": 0.0, "gather": 0.0}
    ev = lambda: torch.cuda.Event(enable_timing=True)

    def compute(timers=None, marks=None):
        mark = (lambda: (marks.append(ev()), marks[-1].record())) if timers is not None else (lambda: None)
        mark()
        o1, o2 = net.reconstruct_batch(im1, im2)
        mark()
        idx, valid = matching.match(o1["pts3d"], o2["pts3d"], o1["desc"], o2["desc"])
        mark()
        # frame = view 1 (its own camera), keyframe = view 2; the keyframe's canonical points stand in as
        # X_ji (same shapes and data flow as FrameTracker.track, tracker.py:88-123); all P solves batched
        Xf, Qk, vo, vk, cnt = tracker.track_gather(
            o1["pts3d"].reshape(P, n, 3), o1["conf"].reshape(P, n), o2["conf"].reshape(P, n),
            o1["desc_conf"].reshape(P, n), o2["desc_conf"].reshape(P, n), idx, valid.reshape(P, n),
            tcfg["C_conf"], tcfg["Q_conf"])
        poses, T_rel, info = tracker.opt_pose_ray_dist_sim3(Xf, o2["pts3d"].reshape(P, n, 3), ident, ident, Qk, vo, tcfg,
                                                           fixed_iters=True)
        mark()
        return (o1["pts3d"], o2["pts3d"], o1["conf"], o2["conf"], idx, valid, poses)

    def wire(out):
        """What travels: pointmaps + confidences fp32, match index int32, validity u8, poses."""
        return out[:4] + (out[4].to(torch.int32),) + out[5:]

    def step(timers=None):
        marks = []
        out = compute(timers, marks)
        if dist is not None:
            out = m3dist.all_gather_results(wire(out))
        if timers is not None:
            marks.append(ev()); marks[-1].record()
            torch.cuda.synchronize()
            for k, (a, b) in zip(("infer", "match", "gn", "gather"), zip(marks[:-1], marks[1:])):
                timers[k] += a.elapsed_time(b)
        return out

    # Capture the compute part of the step (~1500 launches) into a hipGraph: replay removes the host launch
    # path, which matters for small per-GPU batches (a B=1 step is launch-bound when issued eagerly).  The
    # RCCL all-gather (N > 1) is issued eagerly on the graph's static result buffers after each replay.
    graph = None
    if not args.no_graph:
        try:
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                graph_out = compute()
            torch.cuda.synchronize()
        except Exception as e:                                   # noqa: BLE001 - reported, never silent
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            graph = None
    if graph is None:
        run_step = step
    elif dist is None:
        run_step = graph.replay
    else:
        pending = []

        def run_step():
            # replay, snapshot the static result buffers (pack = one cat kernel), and let RCCL gather the
            # snapshot on its own stream while the next replay computes; at most one gather in flight
            graph.replay()
            if pending:
                pending.pop().wait()
            pending.append(m3dist.all_gather_results(wire(graph_out), async_op=True))

    for _ in range(args.warmup):
        run_step()

    def barrier():
        if dist is not None:
            if graph is not None and pending:
                pending.pop().wait()                               # the last step's gather belongs to the timed region
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- untimed instrumented passes: per-stage times and per-launch MFMA kernel timing --------------
    step(stage_ms)
    ops.PROFILE = []
    conc, net.concurrent_heads = net.concurrent_heads, False      # per-launch timing wants the launches serialised
    step()
    torch.cuda.synchronize()
    net.concurrent_heads = conc
    prof, ops.PROFILE = ops.PROFILE, None
    by_kind = {}
    for kind, flops, e0, e1, nbytes in prof:
        d = by_kind.setdefault(kind, [0.0, 0.0, 0, 0.0])
        d[0] += flops; d[1] += e0.elapsed_time(e1) * 1e-3; d[2] += 1; d[3] += nbytes
    g = by_kind.get("gemm256", [0.0, 1.0, 1, 0.0])
    traffic = pmc_traffic_per_launch(("k_gemm256<0",))
    gemm_tflops = g[0] / g[1] / 1e12
    model_flops = net.flops_per_pair(H, W) * P

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
        "data": "synthetic 512x512 textured pairs; seeded random-init weights (no checkpoint available offline)",
        "config": {"workload": f"{P} keyframe pairs/GPU at 512x512 (BASELINE configs[3] per-GPU shard): "
                               "two-view MASt3R ViT-L infer + iter_proj/refine match + 10-iter GN tracking"
                               + ("" if world == 1 else " + RCCL all-gather of results"),
                   "pairs_per_gpu": P, "global_pairs": world * P, "image": [H, W], "gn_iters": tcfg["max_iters"],
                   "parallelism": f"pair-sharded x{world}", "launch": ("hipGraph replay" + ("" if dist is None else " + RCCL all-gather of the previous step overlapped on the communicator stream")) if graph is not None else "eager"},
        "stage_ms": {k: round(v, 3) for k, v in stage_ms.items()},
        "model_tflop_per_step": model_flops / 1e12,
        "roofline": {"bound": "mfma", "kernel": "k_gemm256 (bf16 MFMA GEMM, 256x256x64 / 256x192x64 ping-pong tiles; dense launches only - "
                                                "its implicit-GEMM conv launches and the small-problem kernel are listed under other_kernels_tflops)",
                     "achieved": gemm_tflops, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": gemm_tflops / MFMA_BF16_PEAK_TFLOPS,
                     "traffic": traffic, "traffic_unit": "HBM bytes per launch (rocprofv3 --pmc FETCH_SIZE x2 / WRITE_SIZE passes "
                                                         "of this command, profiles/r01_pmc_traffic.json)",
                     "algorithmic_bytes_per_launch": g[3] / max(g[2], 1),
                     "launches": g[2], "avg_launch_us": g[1] / max(g[2], 1) * 1e6,
                     "other_kernels_tflops": {k: v[0] / v[1] / 1e12 for k, v in by_kind.items() if k != "gemm256"},
                     "all_mfma_kernels_tflops": sum(v[0] for v in by_kind.values()) / sum(v[1] for v in by_kind.values()) / 1e12},
    }

    if world == 1 and not args.no_b1:
        # random weights give meaningless geometry (scattered gathers), so the matcher is
        # ALSO timed on a smooth synthetic two-view scene of the same size (what real pointmaps look like)
        sc = synthetic.geometric_pair(H, W, seed=0, batch=1)
        gt = lambda k: torch.from_numpy(sc[k]).to(dev).repeat(P, 1, 1, 1)
        gX11, gX21, gD11, gD21 = gt("X11"), gt("X21"), gt("D11"), gt("D21")
        for _ in range(2):
            matching.match(gX11, gX21, gD11, gD21)
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(5):
            matching.match(gX11, gX21, gD11, gD21)
        e1.record(); torch.cuda.synchronize()
        result["match_ms_geometric_scene"] = round(e0.elapsed_time(e1) / 5, 3)

    if world == 1 and not args.no_b1 and P != 1:
        # BASELINE configs[1]: one pair per step (latency regime), same pipeline, graph-replayed
        a1, b1 = im1[:1].contiguous(), im2[:1].contiguous()

        def step1():
            o1, o2 = net.reconstruct_batch(a1, b1)
            idx, valid = matching.match(o1["pts3d"], o2["pts3d"], o1["desc"], o2["desc"])
            Xf, Qk, vo, vk, cnt = tracker.track_gather(
                o1["pts3d"].reshape(1, n, 3), o1["conf"].reshape(1, n), o2["conf"].reshape(1, n),
                o1["desc_conf"].reshape(1, n), o2["desc_conf"].reshape(1, n), idx, valid.reshape(1, n),
                tcfg["C_conf"], tcfg["Q_conf"])
            return tracker.opt_pose_ray_dist_sim3(Xf, o2["pts3d"].reshape(1, n, 3), ident, ident, Qk, vo, tcfg,
                                                  fixed_iters=True)
        for _ in range(2):
            step1()
        torch.cuda.synchronize()
        run1 = step1
        if graph is not None:
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
