#!/usr/bin/env python3
"""Benchmark of the MASt3R-SLAM hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload pairs|backend] [--pairs-per-gpu P]

Launch forms (both give one process per GPU over RCCL):
  * `python bench.py --gpus N` with no WORLD_SIZE in the environment: THIS process starts the N ranks itself (child
    processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1, before anything here touches the GPU),
    relays rank 0's JSON line and exits non-zero if any rank fails;
  * `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`: the ranks come from the
    launcher's environment; --gpus must equal WORLD_SIZE.

--workload pairs (default, BASELINE.json's metric; P = 8 pairs per GPU = configs[3]'s per-GPU shard).  One step =
one pass of the per-frame hot path over P synthetic keyframe pairs per GPU:
  two-view network (ViT-L encoder on 2P images, two 12-block decoders, DPT + feature heads)
  -> dense matching (prep -> iter_proj -> refine_matches -> occlusion test)
  -> Gauss-Newton Sim(3) tracking solve (10 iterations) per pair
  [N > 1] -> RCCL all-gather of the per-pair results (pointmaps, confidences, indices, validity).
--workload backend (BASELINE configs[4]: 256-keyframe loop-closure re-match + local-BA blocks, fp16 features).
One step = every rank re-matches its shard of the graph's edges (96 of 762 per GPU) from CACHED encoder tokens
(symmetric decode, both matching directions, fp16 descriptors) through the edge-sharded FactorGraph, evaluates the
per-edge normal-equation blocks of its edges, [N > 1] all-gathers 36 doubles per directed edge, and solves the
1785-unknown system (blocked float64 Cholesky) - reference flow slam.py:292-319 -> global_opt.py:49-211.

Units (pairs / edges) are independent: each rank works on its own shard (weak scaling); inputs are resident in HBM
before the timed region.  Weights are seeded random (no checkpoint can be fetched), data is synthetic - both stated
in the JSON.  Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` (dominant kernel = the 16-bit
MFMA GEMM, algorithmic FLOPs / HIP-event time per launch) and `cpu_baseline` (the CPU oracle timed on this host,
rank 0, N = 1 only, on a bounded sample).
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

H = W = 512
MFMA_BF16_PEAK_TFLOPS = 2500.0      # dense bf16, MI355X_MICROARCH.md "Chip-level parameters"
HBM_PEAK_GBS = 8000.0               # HBM3E peak, same table (6.29 TB/s measured by a float4 copy)
MB = 1e6


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=("pairs", "backend"), default="pairs")
    ap.add_argument("--pairs-per-gpu", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--matcher", choices=["iter_proj", "fast_nn"], default="iter_proj",
                    help="match leg of the pairs workload: the reference's dense matcher (iter_proj + refine_matches, default) or "
                         "the fast reciprocal nearest-neighbour matcher BASELINE.json's north_star names (MFMA search, fp16 descriptors)")
    ap.add_argument("--precision", choices=("bf16", "fp16"), default="fp16",
                    help="16-bit operand type of the ViT trunk: fp16 (default: the reference's and load_mast3r's default precision, "
                         "mast3r_utils.py:51 - the mode that holds BASELINE.json's 1e-3 pointmap tolerance on trained-like weight "
                         "statistics, tests/test_gpu_full_model.py TRAINED_TOL) or bf16 (the type BASELINE configs[1] names; same MFMA "
                         "rate, 3 mantissa bits fewer, 1.4e-3 / 2.6e-3 on that family); the heads are fp16 either way")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) and run the collectives even with one rank")
    ap.add_argument("--dist-overhead", action="store_true",
                    help="also measure the dist_overhead block with --model tiny (always on for the full model unless --no-b1)")
    ap.add_argument("--no-b1", action="store_true", help="skip the extra measurements outside the timed step (batch 1, fp16 features, fast NN)")
    # backend workload (BASELINE configs[4])
    ap.add_argument("--keyframes", type=int, default=256)
    ap.add_argument("--edges-per-gpu", type=int, default=96, help="undirected edges per rank (762 / 8 rounded up)")
    ap.add_argument("--edge-batch", type=int, default=8, help="edges per symmetric decode (2x as many pair decodes)")
    ap.add_argument("--gn-iters", type=int, default=1, help="Gauss-Newton iterations per step (SURVEY 8d config 5: one block pass)")
    ap.add_argument("--model", choices=("full", "tiny"), default="full", help="tiny: reduced depth, for tests only")
    ap.add_argument("--image", type=int, nargs=2, default=[H, W], metavar=("H", "W"))
    ap.add_argument("--dist-backend", choices=("nccl", "gloo"), default="nccl",
                    help="test hook: gloo lets several ranks share ONE GPU (RCCL refuses two ranks on a device); the data "
                         "path is unchanged, only the collectives travel through host memory")
    ap.add_argument("--single-device", action="store_true", help="test hook: every rank uses cuda:0")
    ap.add_argument("--stub", action="store_true",
                    help="test hook: gloo on the CPU with a trivial step (launcher, rendezvous, barrier, max-over-ranks timing, JSON)")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` starts the N ranks itself
def launch(args) -> int:
    """Start one child process per rank (before this process has imported torch or touched the GPU: children are
    plain fork+exec of the interpreter), relay rank 0's stdout, return non-zero if any rank failed."""
    import subprocess
    import tempfile
    import threading
    n = args.gpus
    # rendezvous through a file in a directory of this launch's own (torch.distributed FileStore): no TCP port is
    # guessed, so launchers started at the same moment (tests, concurrent benches) cannot collide or cross-connect
    rdzv_dir = tempfile.mkdtemp(prefix="m3bench_rdzv_")
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   M3_BENCH_RDZV_FILE=os.path.join(rdzv_dir, "store"), M3_BENCH_SPAWNED="1")
        env.pop("MASTER_PORT", None)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL across processes needs it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(procs[0].stdout), daemon=True)
    reader.start()
    rc = 0
    live = set(range(n))
    while live and rc == 0:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0:
                rc = code if code > 0 else 1
                print(f"[bench] rank {r} exited with status {code}; stopping the other ranks", file=sys.stderr, flush=True)
                break
        time.sleep(0.05)
    for r in live:                                   # a failed rank leaves the others at a barrier: end exactly those
        procs[r].terminate()
    for p in procs:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()
    reader.join(timeout=10)
    import shutil
    shutil.rmtree(rdzv_dir, ignore_errors=True)
    sys.stdout.write("".join(lines))
    sys.stdout.flush()
    if rc == 0 and not any(l.startswith("{") for l in lines):
        print("[bench] rank 0 printed no JSON line", file=sys.stderr)
        rc = 1
    return rc


# ----------------------------------------------------------------------------------------------------------------
def pmc_traffic(prefixes, tag=""):
    """HBM bytes per launch of the kernels whose name starts with one of `prefixes`, from the newest committed PMC
    summary of this command (FETCH_SIZE and WRITE_SIZE cannot be collected inside the timed run: separate rocprofv3
    passes, tools/prof_summary.py).  (None, None) when no summary is present."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_traffic{tag}.json")))
    if not files:
        return None, None
    try:
        kern = json.load(open(files[-1]))["kernels"]
    except (OSError, ValueError, KeyError):
        return None, None
    sel = [v for k, v in kern.items() if k.startswith(prefixes)]
    n = sum(v["launches"] for v in sel)
    return (sum(v["hbm_bytes_per_launch"] * v["launches"] for v in sel) / n if n else None), os.path.basename(files[-1])


def graph_kernel_nodes(graph) -> int:
    """Kernel nodes of a captured torch.cuda.CUDAGraph(keep_graph=True): hipGraphGetNodes + hipGraphNodeGetType through the HIP
    runtime torch has loaded (launches per replay; -1 when the runtime does not export the calls)."""
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        raw = ctypes.c_void_p(graph.raw_cuda_graph())
        n = ctypes.c_size_t(0)
        if hip.hipGraphGetNodes(raw, None, ctypes.byref(n)) != 0:
            return -1
        nodes = (ctypes.c_void_p * n.value)()
        if hip.hipGraphGetNodes(raw, nodes, ctypes.byref(n)) != 0:
            return -1
        kernels = 0
        for i in range(n.value):
            t = ctypes.c_int(-1)
            if hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(t)) == 0 and t.value == 0:    # hipGraphNodeTypeKernel
                kernels += 1
        return kernels
    except Exception:                                       # noqa: BLE001 - a diagnostic, never fatal
        return -1


def gemm_roofline(prof, tag=""):
    """`roofline` object from ops.PROFILE records (kind, flops, e0, e1, bytes) of one instrumented eager pass."""
    by_kind = {}
    for kind, flops, e0, e1, nbytes in prof:
        d = by_kind.setdefault(kind, [0.0, 0.0, 0, 0.0])
        d[0] += flops; d[1] += e0.elapsed_time(e1) * 1e-3; d[2] += 1; d[3] += nbytes
    g = by_kind.get("gemm256", [0.0, 1.0, 1, 0.0])
    traffic, traffic_src = pmc_traffic(("k_gemm256<0",), tag)
    tf = g[0] / g[1] / 1e12
    return {"bound": "mfma",
            "kernel": "k_gemm256 (16-bit MFMA GEMM, 256x256x64 / 256x192x64 ping-pong tiles; dense launches only - its "
                      "implicit-GEMM conv launches and the small-problem kernels are listed under other_kernels_tflops)",
            "achieved": tf, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_BF16_PEAK_TFLOPS,
            "traffic": traffic,
            "traffic_unit": "HBM bytes per launch (rocprofv3 --pmc FETCH_SIZE x2 / WRITE_SIZE passes of this command, "
                            f"profiles/{traffic_src})",
            "algorithmic_bytes_per_launch": g[3] / max(g[2], 1), "launches": g[2],
            "avg_launch_us": g[1] / max(g[2], 1) * 1e6,
            "other_kernels_tflops": {k: v[0] / v[1] / 1e12 for k, v in by_kind.items() if k != "gemm256"},
            "all_mfma_kernels_tflops": sum(v[0] for v in by_kind.values()) / max(sum(v[1] for v in by_kind.values()), 1e-12) / 1e12}


VALU_PEAK_GINSTR = 1024 * 2.4 / 2.0      # wave64 VALU instructions per ns the chip can issue: 1 024 SIMD-32s, 2 cycles per
                                          # instruction at the 2.4 GHz maximum clock (MI355X_MICROARCH.md, cycle constants)


def valu_rows(cprof, rows):
    """Kernels the design classes as VALU-issue-bound (bit-exact matcher arithmetic): issue-rate roofline.  rows:
    {entry point: (kernels, issue units per call, note)} - issue units = wave64 VALU instructions from the committed PMC
    summary (profiles/r03_matcher_pmc.md, SQ_INSTS_VALU), float64 instructions counted twice (half rate)."""
    out = {}
    for name, (kern, units, note) in rows.items():
        evs = cprof.get(name, [])
        if not evs:
            continue
        us = sum(a.elapsed_time(b) for a, b in evs) / len(evs) * 1e3
        out[name] = {"bound": "valu", "kernels": kern, "issue_units_per_call": units, "avg_us": us,
                     "achieved": units / us / 1e3, "peak": VALU_PEAK_GINSTR, "unit": "G wave-instructions/s",
                     "frac": units / us / 1e3 / VALU_PEAK_GINSTR, "note": note}
    return out


def hbm_rows(cprof, rows):
    """rows: {entry point: (kernels, algorithmic bytes per call, divisor)} -> hbm_rooflines dict from _ffi.PROFILE."""
    out = {}
    for name, (kern, nbytes, div) in rows.items():
        evs = cprof.get(name, [])
        if not evs:
            continue
        us = sum(a.elapsed_time(b) for a, b in evs) / len(evs) / div * 1e3
        out[name] = {"kernels": kern, "algorithmic_bytes": nbytes, "avg_us": us, "achieved": nbytes / us / 1e3,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / us / 1e3 / HBM_PEAK_GBS}
    return out


# ----------------------------------------------------------------------------------------------------------------
class StubWorkload:
    """--stub: a step is one small tensor operation on the CPU plus (N > 1) one gloo all-gather - no device, no
    library.  Exists so that the launcher and the rank-side harness run in the CPU test suite."""
    unit, metric = "units/s", "stub units/sec (test hook)"
    dtype = "f32"

    def __init__(self, args, ctx):
        import torch
        self.torch, self.ctx = torch, ctx
        self.units_per_step = 4
        self.x = torch.arange(64, dtype=torch.float32) + ctx.rank

    def warm(self):
        pass

    def step(self, timed=False):
        if os.environ.get("M3_BENCH_STUB_FAIL_RANK") == str(self.ctx.rank):
            raise SystemExit(3)                                        # test hook: a rank that dies mid-run
        self.x = self.x * 1.0001
        if self.ctx.dist is not None:
            out = [self.torch.empty_like(self.x) for _ in range(self.ctx.world)]
            self.ctx.dist.all_gather(out, self.x)

    def drain(self):
        pass

    def report(self, result, args):
        result["data"] = "stub"
        result["config"] = {"workload": "stub step (test hook)", "parallelism": f"x{self.ctx.world}"}


class PairsWorkload:
    unit, metric = "pairs/s", "keyframe-pairs/sec (512x512 two-view infer+match+GN)"
    dtype = "fp16"

    def __init__(self, args, ctx):
        import numpy as np
        import torch
        from mast3r_slam import config, matching, model as model_mod, synthetic, tracker
        from mast3r_slam import dist as m3dist
        self.args, self.ctx, self.torch, self.np = args, ctx, torch, np
        self.matching, self.tracker, self.m3dist, self.synthetic = matching, tracker, m3dist, synthetic
        self.h, self.w = args.image
        P = self.P = args.pairs_per_gpu
        self.units_per_step = P
        dev = ctx.dev
        config.set_config({"matching": {"use_simple": False}})            # the iter_proj + refine matcher
        cfg = model_mod.TINY_CFG if args.model == "tiny" else None
        self.net = model_mod.Mast3rFull(seed=0, device=dev, precision=args.precision, cfg=cfg)
        self.dtype = args.precision
        base = ctx.rank * P
        im = lambda off: torch.from_numpy(np.stack([synthetic.textured_image(self.h, self.w, 2 * (base + p) + off) for p in range(P)])).to(dev)
        self.im1, self.im2 = im(0), im(1)
        self.sc = self._make_scene(base)
        if args.matcher == "fast_nn":
            self.sc["D11h"], self.sc["D21h"] = self.sc["D11"].half(), self.sc["D21"].half()
        self.ident = torch.tensor([0, 0, 0, 0, 0, 0, 1, 1], dtype=torch.float32, device=dev)
        self.n = self.h * self.w
        self.tcfg = config.get_config()["tracking"]
        self.graphs = None
        self.marks, self.state = [], {}
        self.gather, self.gathered = None, None
        self.graph_launches = None

    def _make_scene(self, base):
        """What the matcher and the Gauss-Newton solve run on: P smooth two-view scenes of
        the benchmark size - pointmaps of both views in the frame's coordinates, 24-d descriptors, confidences and
        descriptor confidences drawn so that the tracker's gates (tracker.py:108-113: C > 0, Q > 1.5) pass for most
        points, and the keyframe's own canonical pointmap Xk = T * X21 under a known small Sim(3) (2 degrees about y,
        t = (0.05, 0, 0.01), s = 1.02) that the solve has to recover.  Random-init weights give neither matchable
        geometry nor confidences above the gates, so feeding the network's outputs here would time a matcher on
        white noise and a solver on ~0 valid points (round-1 verdict)."""
        np, torch, P, dev = self.np, self.torch, self.P, self.ctx.dev
        sc = self.synthetic.geometric_pair(self.h, self.w, seed=1000 + base, batch=P)
        rng = np.random.default_rng(2000 + base)
        n = self.h * self.w
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        ang = np.deg2rad(2.0)
        R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
        Xk = (1.02 * sc["X21"].reshape(P, n, 3).astype(np.float64) @ R.T + np.array([0.05, 0.0, 0.01])).astype(np.float32)
        u = lambda lo, hi: rng.uniform(lo, hi, size=(P, n)).astype(np.float32)
        return dict(X11=t(sc["X11"]), X21=t(sc["X21"]), D11=t(sc["D11"]), D21=t(sc["D21"]), Xk=t(Xk),
                    Cf=t(u(1.0, 3.0)), Ck=t(u(1.0, 3.0)), Qf=t(u(1.0, 4.0)), Qk=t(u(1.0, 4.0)),
                    T_true=np.array([0.05, 0.0, 0.01, 0.0, np.sin(ang / 2), 0.0, np.cos(ang / 2), 1.02]))

    # ---- the three legs of a step.  Each writes into tensors that stay alive (graph-static buffers). ----
    def leg_infer(self):
        return self.net.reconstruct_batch(self.im1, self.im2)

    def leg_match(self):
        sc = self.sc
        if self.args.matcher == "fast_nn":
            # MASt3R sec. 3.3: reciprocal nearest neighbours in descriptor space from a subsampled seed grid (4096 seeds per
            # pair at 512x512), all P pairs per launch; the sparse matches become the tracker's (index, validity) maps:
            # idx[pair, pixel of view 2] = pixel of view 1
            m = self.matching.fast_reciprocal_nn_maps(sc["D11h"], sc["D21h"], subsample=8, max_iter=3)   # device-only: capturable
            return m["idx"], m["valid"]
        return self.matching.match(sc["X11"], sc["X21"], sc["D11"], sc["D21"])

    def leg_gn(self, idx, valid):
        # FrameTracker.track's data flow (tracker.py:88-123, :177-214): frame = view 1, keyframe = view 2;
        # gather the frame's points at the match index, gate on confidences, then the 10-iteration solve of
        # all P problems in one launch sequence
        sc, P, n, tcfg = self.sc, self.P, self.n, self.tcfg
        Xf, Qk, vo, vk, cnt = self.tracker.track_gather(sc["X11"].reshape(P, n, 3), sc["Cf"], sc["Ck"], sc["Qf"], sc["Qk"],
                                                        idx, valid.reshape(P, n), tcfg["C_conf"], tcfg["Q_conf"])
        poses, T_rel, info = self.tracker.opt_pose_ray_dist_sim3(Xf, sc["Xk"], self.ident, self.ident, Qk, vo, tcfg, fixed_iters=True)
        return poses, T_rel, info, vo

    @staticmethod
    def wire(o1, o2, idx, valid, poses):
        """What travels: pointmaps + confidences fp32, match index int32, validity u8, poses."""
        return (o1["pts3d"], o2["pts3d"], o1["conf"], o2["conf"], idx, valid, poses)

    def warm(self):
        """eager warm-up (lazy allocations, attribute setup), then capture each leg into its own hipGraph: the step
        is ~1600 launches, replay removes the host launch path (a B=1 step is launch-bound when issued eagerly).
        Three graphs instead of one so that stream events BETWEEN the replays give device time per stage inside
        the timed region itself."""
        torch = self.torch
        for _ in range(2):
            o1, o2 = self.leg_infer(); idx, valid = self.leg_match(); gn = self.leg_gn(idx, valid)
        torch.cuda.synchronize()
        if not self.args.no_graph:
            try:
                graphs = [torch.cuda.CUDAGraph(keep_graph=True) for _ in range(3)]
                with torch.cuda.graph(graphs[0]):
                    o1, o2 = self.leg_infer()
                with torch.cuda.graph(graphs[1]):
                    idx, valid = self.leg_match()
                with torch.cuda.graph(graphs[2]):
                    gn = self.leg_gn(idx, valid)
                torch.cuda.synchronize()
                self.graphs = graphs
                self.graph_launches = [graph_kernel_nodes(g) for g in graphs]
            except Exception as e:                                   # noqa: BLE001 - reported, never silent
                print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
                self.graphs = None
        self.state = {"o1": o1, "o2": o2, "idx": idx, "valid": valid, "gn": gn}

    def step(self, timed=False, exchange=None):
        torch, st, graphs = self.torch, self.state, self.graphs
        m = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if timed else None
        if m: m[0].record()
        if graphs is not None:
            graphs[0].replay()
        else:
            st["o1"], st["o2"] = self.leg_infer()
        if m: m[1].record()
        if graphs is not None and graphs[1] is not None:
            graphs[1].replay()
        else:
            idx, valid = self.leg_match()
            if graphs is not None:                                    # the GN graph reads the buffers it was captured on
                st["idx"].copy_(idx); st["valid"].copy_(valid)
            else:
                st["idx"], st["valid"] = idx, valid
        if m: m[2].record()
        if graphs is not None:
            graphs[2].replay()
        else:
            st["gn"] = self.leg_gn(st["idx"], st["valid"])
        if m:
            m[3].record()
            self.marks.append(m)
        if (self.ctx.dist is not None) if exchange is None else exchange:
            self.exchange()

    def exchange(self):
        """Snapshot the result buffers into the pre-allocated packed send buffer (one copy kernel per field; the index
        is narrowed to int32 inside its copy) and let RCCL gather the snapshot on the communicator's stream while the
        next step computes; at most one gather in flight.  Nothing is allocated here (dist.PackedGather), and the
        gathered results are strided views of the receive buffer - no unpacking copies."""
        torch, st = self.torch, self.state
        fields = self.wire(st["o1"], st["o2"], st["idx"], st["valid"], st["gn"][0])
        if self.gather is None:
            dts = [torch.int32 if t.dtype == torch.int64 else t.dtype for t in fields]
            self.gather = self.m3dist.PackedGather(fields, dtypes=dts, group=self.ctx.group)
        self.gather.post(fields)

    def drain(self):
        if self.gather is not None:
            self.gathered = self.gather.wait()                      # the last step's gather belongs to the timed region

    def report(self, result, args):
        torch, np, P, ctx, sc, n, tcfg = self.torch, self.np, self.P, self.ctx, self.sc, self.n, self.tcfg
        from mast3r_slam import _ffi, ops
        marks = self.marks
        if self.gathered is not None:
            # the exchange delivered this rank's last results: its row of every gathered view against the local tensors
            st = self.state
            local = self.wire(st["o1"], st["o2"], st["idx"], st["valid"], st["gn"][0])
            same = all(bool(torch.equal(g[ctx.rank], l.to(g.dtype))) for g, l in zip(self.gathered, local))
            result["exchange"] = {"bytes_per_rank": int(self.gather.nbytes), "fields": len(local),
                                  "own_row_equals_local_results": same,
                                  "gathered_shapes": [list(g.shape) for g in self.gathered]}
            if not same:
                raise SystemExit("bench invalid: the all-gathered results differ from this rank's local results")
        stage_ms = {k: sum(m[i].elapsed_time(m[i + 1]) for m in marks) / len(marks) for i, k in enumerate(("infer", "match", "gn"))}
        stage_ms["sum"] = sum(stage_ms.values())
        # ---- did the match and GN legs do the work they name? ----
        poses, T_rel, info, vo = self.state["gn"]
        match_valid_frac = float(self.state["valid"].float().mean())
        valid_frac = float(vo.float().mean())
        pose_err = float(np.abs(T_rel.double().cpu().numpy() - sc["T_true"][None]).max())
        sparse = args.matcher == "fast_nn"
        if sparse:
            seeds = (self.h // 8) * (self.w // 8)
            recip = float(self.state["valid"].sum()) / (P * seeds)
            if recip < 0.5 or not pose_err < 2e-2:
                raise SystemExit(f"bench invalid: {recip:.3f} of the seeds found a reciprocal match, pose error {pose_err:.3g}")
        elif match_valid_frac < 0.5 or valid_frac < 0.5:
            raise SystemExit(f"bench invalid: match_valid_frac={match_valid_frac:.3f} valid_frac={valid_frac:.3f} (< 0.5): "
                             "the matcher / Gauss-Newton legs would be timed on rejected points")
        # ---- untimed instrumented pass (eager, launches serialised): per-launch device time by kernel family ----
        ops.PROFILE = []
        _ffi.PROFILE = {}
        _ffi.PROFILE_NAMES = ("m3_prep_iter_proj", "m3_iter_proj", "m3_refine_matches", "m3_match_epilogue",
                              "m3_track_gather_batch", "m3_track_gn_ray_dist_batch")
        if self.graphs is not None:
            # queue-ahead: two replays of the inference graph (~50 ms of device work) go first, so the eager launches below pile
            # up BEHIND them in the stream and run back to back as they do inside the graph - an event pair then brackets the
            # kernel's device time (what rocprofv3 reports for the timed replays), not the host's launch latency in front of
            # an idle GPU (r04: 100.2 us per dense launch by events against 92.8 us by rocprofv3 for this reason)
            for _ in range(2):
                self.graphs[0].replay()
        self.leg_infer(); i2, v2 = self.leg_match(); self.leg_gn(i2, v2)           # single stream: the launches are serialised
        torch.cuda.synchronize()
        prof, ops.PROFILE = ops.PROFILE, None
        cprof, _ffi.PROFILE = _ffi.PROFILE, None
        iters = int(tcfg["max_iters"])
        # HBM-bound kernel families: algorithmic bytes per call from SURVEY 8d (per pair at 512x512, fp32) x P pairs
        scale = P * (self.h * self.w) / (512 * 512)
        hbm = hbm_rows(cprof, {
            "m3_prep_iter_proj": ("k_prep", 21.0 * MB * scale, 1),
            "m3_iter_proj": ("k_iter_proj (+ k_iter_reduce / k_iter_limit, early-stop second pass)", 17.0 * MB * scale, 1),
            "m3_refine_matches": ("k_refine_lds<24> / k_refine<24>", 54.5 * MB * scale, 1),
            "m3_match_epilogue": ("k_epilogue", 9.7 * MB * scale, 1),
            "m3_track_gather_batch": ("k_track_gather", (3.15 + 4 * 1.05 + 2.1 + 0.26 + 3.15 + 1.05 + 0.52) * MB * scale, 1),
            "m3_track_gn_ray_dist_batch": (f"k_track_accum + k_track_solve, per GN iteration ({iters} per call)", 8.7 * MB * scale, iters),
        })
        valu = valu_rows(cprof, {
            "m3_iter_proj": ("k_iter_proj", (57.8e6 + 2 * 28.9e6) / 8 * scale,
                             "264 VALU instructions per LM step and point, a third of them float64 (numpy-twin interpolation)"),
            "m3_refine_matches": ("k_refine_lds<24, float, 3>", 98.8e6 / 8 * scale,
                                  "49 x 24 separately rounded multiply + add per point (bit-exact summation order)"),
        })
        world = ctx.world
        result.update({
            "data": "synthetic: 512x512 textured pairs through the network (seeded random-init weights, no checkpoint available "
                    "offline); matcher + Gauss-Newton on smooth synthetic two-view scenes of the same size (SURVEY 8d configs 2-3)",
            "config": {"workload": f"{P} keyframe pairs/GPU at {self.h}x{self.w} (BASELINE configs[3] per-GPU shard): "
                                   f"two-view MASt3R ViT-L infer ({args.precision} trunk, fp16 heads, fp32 accumulate"
                                   + (", LayerNorms folded into the GEMMs, hi+lo fp16 residual stream" if getattr(self.net, "ln_fold", False) else "") + ") + "
                                   + ("fast reciprocal-NN match (4096-seed grid, fp16 descriptors, MFMA search) " if sparse else "iter_proj/refine match ")
                                   + "+ 10-iter GN tracking" + ("" if ctx.dist is None else " + RCCL all-gather of results"),
                       "pairs_per_gpu": P, "global_pairs": world * P, "image": [self.h, self.w], "gn_iters": iters, "matcher": args.matcher,
                       "parallelism": f"pair-sharded x{world}",
                       "launch": ("3 hipGraph replays per step (infer | match | GN)" + ("" if ctx.dist is None else " + RCCL all-gather of the previous step overlapped on the communicator stream")) if self.graphs is not None else "eager"},
            "launches_per_step": None if not self.graph_launches else
                dict(zip(("infer", "match", "gn"), self.graph_launches), total=sum(self.graph_launches),
                     note="kernel nodes of the three captured hipGraphs (hipGraphGetNodes)"),
            "stage_ms": {k: round(v, 3) for k, v in stage_ms.items()},
            "stage_ms_note": "device time between stream events recorded around the three graph replays of every TIMED step (mean); sum ~ ms_per_step",
            "match_valid_frac": round(match_valid_frac, 4),
            "valid_frac": round(valid_frac, 4),
            "gn_pose_max_abs_err_vs_true_sim3": pose_err,
            "model_tflop_per_step": self.net.flops_per_pair(self.h, self.w) * P / 1e12,
            "roofline": gemm_roofline(prof),
            "hbm_rooflines": hbm,
            "valu_rooflines": valu,
        })
        if world == 1 and not args.no_b1:
            self._extras(result, args)
            if args.model == "full" or args.dist_overhead:
                self._dist_overhead(result, args)
        if ctx.rank == 0 and world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline_pairs(args, self.net)

    def _dist_overhead(self, result, args, steps: int = 10, rounds: int = 3):
        """What the N > 1 step adds on ONE GPU: the same graph replays with and without the result exchange (snapshot into the
        packed send buffer + a one-rank RCCL all-gather on the communicator's stream + the views of the receive buffer), in
        interleaved rounds on one high-priority compute stream.  The xGMI transfer itself is not in it (one rank)."""
        import statistics
        import tempfile
        import torch.distributed as tdist
        torch = self.torch
        own = not tdist.is_initialized()
        if own:
            f = tempfile.NamedTemporaryFile(prefix="m3_bench_rdzv1_", delete=False)
            f.close(); os.unlink(f.name)
            tdist.init_process_group("nccl", rank=0, world_size=1, init_method="file://" + f.name, device_id=self.ctx.dev)

        def run(with_exchange):
            for _ in range(2):
                self.step(exchange=with_exchange)
            self.drain(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step(exchange=with_exchange)
            if with_exchange:
                self.drain()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / steps * 1e3
        hp = torch.cuda.Stream(priority=-1)
        hp.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(hp):
            arms = [(run(False), run(True)) for _ in range(rounds)]
        torch.cuda.synchronize()
        without, with_x = statistics.median(a for a, _ in arms), statistics.median(b for _, b in arms)
        result["dist_overhead"] = {"ms_per_step_without_exchange": round(without, 3), "ms_per_step_with_exchange": round(with_x, 3),
                                   "delta_ms": round(with_x - without, 3), "packed_bytes_per_rank": int(self.gather.nbytes),
                                   "rounds": [[round(a, 3), round(b, 3)] for a, b in arms],
                                   "note": "one rank: dist.PackedGather.post (7 field copies into the pre-allocated send buffer) + "
                                           "all_gather_into_tensor(async_op=True) on RCCL's stream + wait (stream-ordered) + strided "
                                           "views of the receive buffer; compute on a high-priority stream; medians of interleaved rounds"}
        if own:
            self.gather = None
            tdist.destroy_process_group()

    def _extras(self, result, args):
        """Measurements outside the timed step (N = 1 only)."""
        torch, sc, P, n, tcfg = self.torch, self.sc, self.P, self.n, self.tcfg
        matching, tracker = self.matching, self.tracker
        from mast3r_slam import _ffi
        ev = lambda: torch.cuda.Event(enable_timing=True)
        # for the record: the same matcher on what the random-weight network emits (scattered gathers)
        o1, o2 = self.state["o1"], self.state["o2"]
        for _ in range(2):
            matching.match(o1["pts3d"], o2["pts3d"], o1["desc"], o2["desc"])
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(5):
            matching.match(o1["pts3d"], o2["pts3d"], o1["desc"], o2["desc"])
        e1.record(); torch.cuda.synchronize()
        result["match_ms_on_random_weight_network_output"] = round(e0.elapsed_time(e1) / 5, 3)

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
        idx, valid = self.state["idx"], self.state["valid"]
        both = (valid & v16)[..., 0]
        us16 = sum(a.elapsed_time(b) for a, b in evs[1:]) / max(len(evs) - 1, 1) * 1e3
        result["fp16_features"] = {"match_ms": round(e0.elapsed_time(e1) / 5, 3),
                                   "match_valid_frac": round(float(v16.float().mean()), 4),
                                   "idx_agreement_with_fp32_features": round(float((idx == i16)[both].float().mean()), 5),
                                   "m3_refine_matches_f16": {"algorithmic_bytes": 29.3 * MB * P, "avg_us": us16,
                                                             "achieved": 29.3 * MB * P / us16 / 1e3, "unit": "GB/s",
                                                             "frac": 29.3 * MB * P / us16 / 1e3 / HBM_PEAK_GBS}}

        # matcher variant named by north_star: fast reciprocal NN (MASt3R sec. 3.3) on the same scenes, all P pairs in one
        # set of launches (m3_frnn_pack once per map, m3_frnn_round per round), 64 x 64 seeds per pair (subsample 8),
        # fp16 descriptors, 3 rounds; not part of the timed step.  MFMA roofline per search launch: 2 S N D flops
        # (D = 24; the kernel multiplies K padded to 32) over the device time of the k_nn_mfma launch.
        d1, d2 = sc["D21"].half(), sc["D11"].half()

        def time_frnn(prune):
            for _ in range(2):
                matching.fast_reciprocal_nn_maps(d1, d2, subsample=8, max_iter=3, prune=prune)
            e0, e1 = ev(), ev()
            e0.record()
            for _ in range(3):
                out = matching.fast_reciprocal_nn_maps(d1, d2, subsample=8, max_iter=3, prune=prune)
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / 3, out
        ms_all, mp = time_frnn(True)                  # the default: exact block-bound search (m3_frnn_round_pruned)
        ms_brute, mb = time_frnn(False)               # every seed against every pixel (m3_frnn_round / _active)
        same = all(bool(torch.equal(mp[k], mb[k])) for k in ("map1", "pairs", "count", "idx", "valid"))
        # the rounds alone, HIP events around the C-ABI calls: round 0 runs every seed (two full searches), rounds 1, 2 only
        # the seeds that have not converged
        names = ("m3_frnn_round", "m3_frnn_round_active", "m3_frnn_round_pruned")
        _ffi.PROFILE, _ffi.PROFILE_NAMES = {}, names
        matching.fast_reciprocal_nn_maps(d1, d2, subsample=8, max_iter=3, prune=False)
        matching.fast_reciprocal_nn_maps(d1, d2, subsample=8, max_iter=3, prune=True)
        torch.cuda.synchronize()
        prof_r, _ffi.PROFILE = _ffi.PROFILE, None
        us_of = lambda name: [a_.elapsed_time(b_) * 1e3 for a_, b_ in prof_r.get(name, [])]
        us_round = (us_of("m3_frnn_round") or [0.0])[0]
        seeds = (self.h // 8) * (self.w // 8)
        fl_search = 2.0 * P * seeds * (self.h * self.w) * 24
        result["fast_nn_matcher"] = {"ms_per_pair": round(ms_all / P, 3), "ms_all_pairs": round(ms_all, 3), "pairs": P,
                                     "reciprocal_pairs": int(mp["count"].sum()), "seeds_per_pair": seeds,
                                     "pruned_round_us": [round(u, 1) for u in us_of("m3_frnn_round_pruned")],
                                     "ms_per_step_with_this_matcher": round(result["stage_ms"]["infer"] + ms_all + result["stage_ms"]["gn"], 3),
                                     "pairs_per_s_with_this_matcher": round(P * 1e3 / (result["stage_ms"]["infer"] + ms_all + result["stage_ms"]["gn"]), 1),
                                     "derived_note": "the two fields above = this run's infer and GN legs + this matcher's time (the timed step "
                                                     "uses the reference's dense matcher; `--matcher fast_nn` times the step with this one)",
                                     "brute_force": {"ms_all_pairs": round(ms_brute, 3), "same_outputs_bit_for_bit": same,
                                                     "round_us": us_round,
                                                     "active_round_us": [round(u, 1) for u in us_of("m3_frnn_round_active")]},
                                     "roofline": {"bound": "mfma", "kernel": "k_nn_mfma<1> (brute force: two full searches + bookkeeping, m3_frnn_round, round 0)",
                                                  "achieved": 2 * fl_search / us_round / 1e6, "peak": MFMA_BF16_PEAK_TFLOPS,
                                                  "unit": "TFLOP/s", "frac": 2 * fl_search / us_round / 1e6 / MFMA_BF16_PEAK_TFLOPS,
                                                  "flops_note": "2 S N D with D = 24 (K is padded to 32 on the matrix core: x 4/3 issued)"},
                                     "note": f"fast_reciprocal_nn_maps on {P} pairs at once (hipGraph-capturable: fixed-shape device outputs, no "
                                             f"host synchronisation): m3_frnn_pack x 2 + m3_frnn_blockstats x 2 + 3 x m3_frnn_round_pruned + "
                                             f"m3_frnn_collect, fp16 descriptors, {seeds} seeds x {self.h * self.w} pixels per search.  The "
                                             f"default search is EXACT with block bounds (8 x 8 pixel tiles: centroid + radius, Cauchy-Schwarz): "
                                             f"same index and score as the brute-force search, which stays the device-side fallback and whose "
                                             f"MFMA rate the roofline block quotes"}

        # calibrated tracking solve (tracker.py:326-406; pixel + log-depth residuals through a pinhole K): same streams as
        # the ray-distance solve (Xf 12 + Xk 12 + Qk 4 + valid 1 B per point and iteration), all P problems per launch
        import numpy as np
        hh, ww = self.h, self.w
        Kc = np.array([[float(ww), 0, ww / 2], [0, float(ww), hh / 2], [0, 0, 1]], dtype=np.float32)
        gat = tracker.track_gather(sc["X11"].reshape(P, n, 3), sc["Cf"], sc["Ck"], sc["Qf"], sc["Qk"], self.state["idx"],
                                   self.state["valid"].reshape(P, n), tcfg["C_conf"], tcfg["Q_conf"])
        Xf_c, Qk_c, vo_c = gat[0], gat[1], gat[2]
        Xk_c = tracker.constrain_points_to_ray((hh, ww), sc["Xk"].reshape(-1, 3), Kc).reshape(P, n, 3)
        Xf_c = tracker.constrain_points_to_ray((hh, ww), Xf_c.reshape(-1, 3), Kc).reshape(P, n, 3)
        run_c = lambda: tracker.opt_pose_calib_sim3(Xf_c, Xk_c, self.ident, self.ident, Qk_c, vo_c, Kc, (hh, ww), tcfg,
                                                    fixed_iters=True)
        for _ in range(2):
            run_c()
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(5):
            info_c = run_c()[2]
        e1.record(); torch.cuda.synchronize()
        it_c = int(tcfg["max_iters"])
        us_c = e0.elapsed_time(e1) / 5 * 1e3 / it_c
        by_c = 29.0 * P * n
        result["calibrated_tracking"] = {"entry": "m3_track_gn_calib_batch", "kernels": "k_track_accum_calib + k_track_solve, per GN iteration",
                                         "iterations": it_c, "avg_us_per_iteration": round(us_c, 2),
                                         "algorithmic_bytes": by_c, "achieved": round(by_c / us_c / 1e3, 1), "unit": "GB/s",
                                         "peak": HBM_PEAK_GBS, "frac": round(by_c / us_c / 1e3 / HBM_PEAK_GBS, 4),
                                         "status": [int(x) for x in info_c[:, 3].tolist()][:2]}

        self._backend_block(result, args)
        if args.model == "full":
            # the OTHER trunk precision through the same graphed inference leg, for the record: the timed step runs
            # args.precision (default fp16 = load_mast3r's and the reference's default, the mode that holds 1e-3 on trained-like
            # weight statistics); BASELINE configs[1] names bf16
            from mast3r_slam import model as model_mod
            other = "bf16" if args.precision == "fp16" else "fp16"
            net2 = model_mod.Mast3rFull(weights=self.net.host_weights, device=self.ctx.dev, precision=other)
            for _ in range(2):
                net2.reconstruct_batch(self.im1, self.im2)
            torch.cuda.synchronize()
            run2 = lambda: net2.reconstruct_batch(self.im1, self.im2)
            if self.graphs is not None:
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2):
                    keep2 = net2.reconstruct_batch(self.im1, self.im2)        # noqa: F841 - the graph's static outputs
                run2 = g2.replay
            run2(); torch.cuda.synchronize()
            e0, e1 = ev(), ev()
            e0.record()
            for _ in range(5):
                run2()
            e1.record(); torch.cuda.synchronize()
            ms2 = e0.elapsed_time(e1) / 5
            # the same per-launch measurement of the dense GEMM launches on this trunk (queued behind two graph replays)
            from mast3r_slam import ops as ops2
            for _ in range(2):
                run2()
            ops2.PROFILE = []
            net2.reconstruct_batch(self.im1, self.im2)
            torch.cuda.synchronize()
            prof2, ops2.PROFILE = ops2.PROFILE, None
            roof2 = gemm_roofline(prof2)
            result[f"{other}_trunk"] = {"infer_ms": round(ms2, 3), f"{args.precision}_trunk_infer_ms": result["stage_ms"]["infer"],
                                        "dense_gemm_roofline": {"achieved": roof2["achieved"], "frac": roof2["frac"], "avg_launch_us": roof2["avg_launch_us"],
                                                                "launches": roof2["launches"],
                                                                "note": "k_gemm256 dense launches of THIS trunk (no LayerNorm fold on the bf16 trunk: its "
                                                                        "launches do only the GEMM + epilogue work, the LayerNorms run as 85 separate kernels)"},
                                        "ms_per_step_with_this_infer_leg": round(ms2 + result["stage_ms"]["match"] + result["stage_ms"]["gn"], 3),
                                        "note": "precision='fp16' (load_mast3r's default): fp16 GEMM / q / k operands, bf16 softmax probabilities "
                                                "and v (M3_DT_F16_PVBF16); precision='bf16': bf16 trunk operands (BASELINE configs[1]); fp16 heads "
                                                "either way; same graph-replayed inference leg on the same images"}
            del net2

        if P != 1:
            # BASELINE configs[1]: one pair per step (latency regime), same pipeline, graph-replayed
            net, ident = self.net, self.ident
            a1, b1 = self.im1[:1].contiguous(), self.im2[:1].contiguous()
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
            launches1 = None
            if self.graphs is not None:
                g1 = torch.cuda.CUDAGraph(keep_graph=True)
                with torch.cuda.graph(g1):
                    keep = step1()          # noqa: F841 - the graph's static outputs
                run1 = g1.replay
                launches1 = graph_kernel_nodes(g1)
            run1(); torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(10):
                run1()
            torch.cuda.synchronize()
            ms1 = (time.perf_counter() - t1) / 10 * 1e3
            # the tracker's real per-frame call (mast3r_match_asymmetric, mast3r_utils.py:451-500, with frame.feat cached as
            # this repo's operator does): the KEYFRAME's encoder tokens exist already, only the new frame is encoded
            tok_kf, grid1 = net.encode_tokens(b1)
            tok_kf = tok_kf.clone()

            def step1_cached():
                tok_f, _ = net.encode_tokens(a1)
                net.decode_heads(tok_f, tok_kf, 1, grid1)
                i1, v1 = matching.match(s1["X11"], s1["X21"], s1["D11"], s1["D21"])
                Xf, Qk, vo1, vk, cnt = tracker.track_gather(s1["X11"].reshape(1, n, 3), s1["Cf"], s1["Ck"], s1["Qf"], s1["Qk"],
                                                            i1, v1.reshape(1, n), tcfg["C_conf"], tcfg["Q_conf"])
                return tracker.opt_pose_ray_dist_sim3(Xf, s1["Xk"], ident, ident, Qk, vo1, tcfg, fixed_iters=True)
            for _ in range(2):
                step1_cached()
            torch.cuda.synchronize()
            run1c = step1_cached
            if self.graphs is not None:
                g1c = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g1c):
                    keepc = step1_cached()  # noqa: F841
                run1c = g1c.replay
            run1c(); torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(10):
                run1c()
            torch.cuda.synchronize()
            ms1c = (time.perf_counter() - t1) / 10 * 1e3
            from mast3r_slam import ops
            ops.PROFILE = []
            net.reconstruct_batch(a1, b1)
            torch.cuda.synchronize()
            prof1, ops.PROFILE = ops.PROFILE, None
            fl = sum(p[1] for p in prof1)
            sec = sum(p[2].elapsed_time(p[3]) for p in prof1) * 1e-3
            result["batch1"] = {"workload": "BASELINE configs[1]: 1 pair/step at 512x512, same pipeline", "pairs_per_s": 1e3 / ms1,
                                "ms_per_pair": ms1, "launches_per_pair": launches1,
                                "ms_per_frame_with_cached_keyframe_tokens": ms1c,
                                "cached_note": "the per-frame tracking call: the keyframe's encoder tokens are cached (Frame.feat), one image is "
                                               "encoded, then decoders + heads + match + 10 GN iterations",
                                "roofline": {"bound": "mfma", "kernel": "all MFMA kernels of one pair (dense GEMMs, convolutions, attention)",
                                             "achieved": fl / sec / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                             "frac": fl / sec / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                                             "whole_pair_frac": net.flops_per_pair(self.h, self.w) / (ms1 * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS}}


    def _backend_block(self, result, args, edges: int = 16, steps: int = 2):
        """BASELINE configs[4] in the default line: a short run of the backend workload (`--workload backend`, reference flow
        slam.py:292-319 -> global_opt.py:49-166) after the timed region - `edges` undirected edges of the 256-keyframe graph
        through the FULL network (symmetric decode from cached tokens, both match directions on fp16 features), the rays-GN
        blocks of their directed edges and the 1785-unknown dense step.  The pairs model is reused (same weights; the
        descriptor output switched to half storage)."""
        import copy
        torch = self.torch
        bargs = copy.copy(args)
        bargs.keyframes, bargs.edges_per_gpu, bargs.edge_batch, bargs.gn_iters, bargs.no_cpu_baseline = 256, edges, 8, 1, True
        net16 = copy.copy(self.net)
        net16.desc_dtype = torch.float16
        wl = BackendWorkload(bargs, self.ctx, net=net16)
        wl.warm()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            wl.step(timed=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        r = {}
        wl.report(r, bargs)
        result["backend"] = {"workload": r["config"]["workload"], "edges_per_s": wl.total_edges / (ms * 1e-3), "ms_per_step": ms,
                             "steps": steps, "stage_ms": r["stage_ms"], "match_valid_frac": r["match_valid_frac"],
                             "edges_kept": r["edges_kept"], "pose_max_abs_err_before_after": r["pose_max_abs_err_before_after"],
                             "hbm_rooflines": {"m3_gn_rays_blocks": r["hbm_rooflines"].get("m3_gn_rays_blocks")},
                             "dense_step": r["dense_step"], "roofline": r["roofline"],
                             "note": "outside the timed region of this command; `python bench.py --workload backend` times the "
                                     "96-edge per-GPU shard of BASELINE configs[4] as the headline"}
        del wl


class BackendWorkload:
    """BASELINE configs[4] per-GPU shard.  reference flow: slam.py:292-319 -> global_opt.py:49-138 (add_factors with
    mast3r_match_symmetric) -> :168-211 (solve_GN_rays) -> kernels.py:262-322."""
    unit, metric = "edges/s", "graph-edges/sec (256-keyframe loop-closure re-match + local-BA blocks + 1785-dim step, fp16 features)"
    dtype = "fp16"

    def __init__(self, args, ctx, net=None):
        import numpy as np
        import torch
        from mast3r_slam import config, frame as frame_mod, matching, model as model_mod, synthetic
        from mast3r_slam.global_opt import FactorGraph
        self.args, self.ctx, self.torch, self.np = args, ctx, torch, np
        self.matching = matching
        self.dtype = args.precision
        self.h, self.w = args.image
        dev = ctx.dev
        config.set_config({"matching": {"use_simple": False}})
        self.lcfg = config.get_config()["local_opt"]
        cfg = model_mod.TINY_CFG if args.model == "tiny" else None
        if net is None:
            net = model_mod.Mast3rFull(seed=0, device=dev, precision=args.precision, cfg=cfg, features="fp16")
        self.net = net
        K = self.K = args.keyframes
        ii, jj = synthetic.chain_edges(K)
        tot = min(len(ii), ctx.world * args.edges_per_gpu)            # weak scaling: 96 edges per rank, all 762 at N = 8
        self.ii, self.jj = ii[:tot], jj[:tot]
        self.total_edges = tot
        from mast3r_slam.dist import shard_range
        self.mine = shard_range(tot, ctx.rank, ctx.world)
        self.units_per_step = tot / ctx.world                          # run_rank multiplies by world
        self.scene = sc = synthetic.keyframe_graph_scene(K, self.h, self.w, dev, seed=7)
        n = self.n = self.h * self.w
        t = self.h // 16 * (self.w // 16)
        g = torch.Generator(device="cpu").manual_seed(11)
        # cached encoder tokens of every keyframe (what Keyframes.feat holds, frame.py:157-158): random, unit scale
        feats = torch.randn((K, t, 1024), generator=g).to(dev, self.net.tdt)
        gy, gx = torch.meshgrid(torch.arange(self.h // 16), torch.arange(self.w // 16), indexing="ij")
        pos = torch.stack([gx.reshape(-1), gy.reshape(-1)], -1).to(dev)
        img = torch.zeros((3, self.h, self.w), dtype=torch.uint8, device=dev)
        shape = torch.tensor([[self.h, self.w]], dtype=torch.int32)
        noise = torch.randn((K, 3), generator=g).to(dev) * 0.01
        noise[0] = 0                                                   # the pinned keyframe keeps its pose
        self.T_init = sc["poses"].clone()
        self.T_init[:, :3] += noise
        self.frames = frame_mod.Keyframes()
        for k in range(K):
            f = frame_mod.Frame(frame_id=k, img=img, img_shape=shape, img_true_shape=shape, T_WC=self.T_init[k:k + 1].clone())
            f.X_canon, f.C, f.N, f.N_updates = sc["Xs"][k], sc["C"][k][:, None], 1, 1
            f.feat, f.pos = feats[k], pos
            self.frames.append(f)
        self.fg = FactorGraph(self.net, self.frames, group=ctx.group, batch=args.edge_batch)
        self.marks = []
        self.cursor = 0
        self.last_fracs = None
        self.batch_cache = {}

    # the match function handed to add_factors (mast3r_utils.mast3r_match_symmetric's signature)
    def match_fn(self, model, feat_i, pos_i, feat_j, pos_j, shape_i, shape_j):
        """(1) the network's share of a re-match: the symmetric decode of these edges from the cached tokens - real work,
        outputs dropped (random-init weights give nothing matchable); (2) the matcher's share on the synthetic
        keyframe graph: both directions as one batch of 2b maps, fp16 descriptors."""
        from mast3r_slam.mast3r_utils import _decode_symmetric      # what mast3r_match_symmetric runs before its matcher
        torch, sc = self.torch, self.scene
        b = feat_i.shape[0]
        _decode_symmetric(model, feat_i, feat_j, shape_i)
        sel = range(self.mine.start + self.cursor, self.mine.start + self.cursor + b)
        key = (self.cursor, b)
        self.cursor += b
        h, w, n = self.h, self.w, self.n
        if key in self.batch_cache:
            # the matcher's inputs of this batch: in the real pipeline they ARE the decoder outputs; here they are
            # assembled from the synthetic graph once (first, untimed step) and kept resident (0.5 GB per 8 edges)
            X11, X21, D11, D21, qs = self.batch_cache[key]
            idx, valid = self.matching.match(X11, X21, D11, D21)
            return (idx[:b], idx[b:], valid[:b], valid[b:]) + qs
        i = torch.tensor([self.ii[k] for k in sel], device=self.ctx.dev)
        j = torch.tensor([self.jj[k] for k in sel], device=self.ctx.dev)
        P = sc["poses"]

        def into(frame_ids, pts_ids):
            """world points of keyframes pts_ids expressed in the camera frames of keyframes frame_ids: R^T (Pw - t) / s."""
            Pw = sc["Pw"][pts_ids]
            T = P[frame_ids]
            q = T[:, 3:7]
            v = Pw - T[:, None, :3]
            qv = -q[:, None, :3].expand_as(v)                          # inverse rotation
            u = 2.0 * torch.cross(qv, v, dim=-1)
            return ((v + q[:, None, 3:4] * u + torch.cross(qv, u, dim=-1)) / T[:, None, 7:8]).reshape(-1, h, w, 3)
        X11 = torch.cat([sc["Xs"][i].reshape(b, h, w, 3), sc["Xs"][j].reshape(b, h, w, 3)])      # ii | jj
        X21 = torch.cat([into(i, j), into(j, i)])                                                # ji | ij
        D11 = torch.cat([sc["D"][i], sc["D"][j]])
        D21 = torch.cat([sc["D"][j], sc["D"][i]])
        idx, valid = self.matching.match(X11, X21, D11, D21)
        q = lambda name, ids: sc[name][ids].reshape(b, n, 1)
        qs = (q("Qself", i), q("Qself", j), q("Qother", j), q("Qother", i))
        self.batch_cache[key] = (X11, X21, D11, D21, qs)
        return (idx[:b], idx[b:], valid[:b], valid[b:]) + qs

    def _one(self, timed=False):
        torch = self.torch
        m = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if timed else None
        for k, f in enumerate(self.frames._frames):
            f.T_WC = self.T_init[k:k + 1]                             # every step starts from the same perturbed poses
        self.fg.reset()
        self.cursor = 0
        if m: m[0].record()
        ok = self.fg.add_factors(self.ii, self.jj, self.lcfg.get("min_match_frac", 0.1), self.match_fn)
        if not ok:
            raise SystemExit("bench invalid: add_factors rejected every edge")
        if m: m[1].record()
        self.fg.solve_GN_rays(max_iters=self.args.gn_iters)
        if m:
            m[2].record()
            self.marks.append(m)

    def warm(self):
        self._one()
        self.torch.cuda.synchronize()

    def step(self, timed=False):
        self._one(timed)

    def drain(self):
        pass

    def report(self, result, args):
        torch, np, ctx, sc = self.torch, self.np, self.ctx, self.scene
        from mast3r_slam import _ffi, kernels, ops
        marks = self.marks
        stage_ms = {k: sum(m[i].elapsed_time(m[i + 1]) for m in marks) / len(marks) for i, k in enumerate(("rematch", "solve"))}
        stage_ms["sum"] = sum(stage_ms.values())
        fg = self.fg
        kept = int(fg.ii.numel())
        # did the legs do the work they name?  matches valid, poses of the keyframes this graph constrains moved towards the truth
        vm = float(fg.valid_match_j.float().mean()) if fg.valid_match_j is not None else 0.0
        touched = torch.unique(torch.cat([fg.ii, fg.jj])).long()
        cur = torch.cat([f.T_WC for f in self.frames._frames])[touched]
        err0 = float((self.T_init[touched] - sc["poses"][touched]).abs().max())
        err1 = float((cur - sc["poses"][touched]).abs().max())
        if vm < 0.5 or kept < 0.9 * self.total_edges:
            raise SystemExit(f"bench invalid: match_valid_frac={vm:.3f}, {kept} of {self.total_edges} edges kept")
        # ---- instrumented eager pass: one decode chunk by kernel family, the block kernel and the dense step alone ----
        b = min(args.edge_batch, len(self.mine))
        sel = list(self.mine)[:b]
        cat = lambda xs: torch.cat([x[None] for x in xs])
        fi, fj = cat([self.frames[self.ii[k]].feat for k in sel]), cat([self.frames[self.jj[k]].feat for k in sel])
        shp = [self.frames[0].img_true_shape] * b
        from mast3r_slam.mast3r_utils import mast3r_decode_symmetric_batch
        ops.PROFILE = []
        mast3r_decode_symmetric_batch(self.net, fi, None, fj, None, shp, shp)
        torch.cuda.synchronize()
        prof, ops.PROFILE = ops.PROFILE, None
        uniq = fg.get_unique_kf_idx()
        Xs, T_WCs, Cs = fg._get_poses_points(uniq)
        li, lj, lidx, lvalid, lQ, graph = fg._local_edges(uniq)
        e_dir = int(li.numel())
        _ffi.PROFILE = {}
        _ffi.PROFILE_NAMES = ("m3_gn_rays_blocks", "m3_gn_rays_step", "m3_iter_proj", "m3_refine_matches_f16", "m3_prep_iter_proj",
                              "m3_match_epilogue")
        for _ in range(3):
            blocks = kernels.gn_rays_blocks(T_WCs, Xs, Cs, li, lj, lidx, lvalid, lQ, sigma_ray=self.lcfg["sigma_ray"],
                                            C_thresh=self.lcfg["C_conf"], Q_thresh=self.lcfg["Q_conf"])
        # the dense step on this rank's blocks (at N = 1 that is the whole system; the keyframes without an edge keep the
        # regularised identity rows): assembly + blocked Cholesky of 7 (K - 1) unknowns + retraction
        _, local_h, num_free = kernels._local_map(np.arange(self.K), np.arange(self.K), self.K, self.lcfg["pin"])
        L = _ffi.lib()
        dim = 7 * num_free
        hbuf = torch.empty(int(L.m3_gn_rays_hbuf_doubles(dim)), dtype=torch.float64, device=ctx.dev)
        info = torch.zeros(4, dtype=torch.float64, device=ctx.dev)
        local = torch.from_numpy(local_h).to(ctx.dev)
        # uniq may be a subset of the K keyframes (N = 1: 96 edges touch 34 of them); the step below is timed on the
        # full-size system so that the reported Cholesky is the 1785-dim one whatever N is
        posk = lambda e: uniq[e.long()].to(torch.int32).contiguous()
        twc = self.T_init.clone()
        for _ in range(3):
            _ffi.call("m3_gn_rays_info_init", _ffi.ptr(info), _ffi.stream_ptr())
            _ffi.call("m3_gn_rays_step", _ffi.ptr(twc), _ffi.ptr(blocks), _ffi.ptr(posk(li)), _ffi.ptr(posk(lj)), _ffi.ptr(local),
                      _ffi.ptr(hbuf), _ffi.ptr(info), self.K, e_dir, num_free, float(self.lcfg["delta_norm"]), _ffi.stream_ptr())
        self.cursor = 0
        self.match_fn(self.net, fi, None, fj, None, shp, shp)          # the matcher's kernels on one chunk (2b maps)
        torch.cuda.synchronize()
        cprof, _ffi.PROFILE = _ffi.PROFILE, None
        scale = 2 * b * (self.h * self.w) / (512 * 512)
        hbm = hbm_rows(cprof, {
            "m3_gn_rays_blocks": ("k_gn_blocks<0> + k_gn_reduce", 10.8 * MB * e_dir * (self.h * self.w) / (512 * 512), 1),
            "m3_prep_iter_proj": ("k_prep", 21.0 * MB * scale, 1),
            "m3_iter_proj": ("k_iter_proj (+ reduce kernels)", 17.0 * MB * scale, 1),
            "m3_refine_matches_f16": ("k_refine_lds<24, half>", 29.3 * MB * scale, 1),
            "m3_match_epilogue": ("k_epilogue", 9.7 * MB * scale, 1),
        })
        step_evs = cprof.get("m3_gn_rays_step", [])
        chol_us = sum(a.elapsed_time(c) for a, c in step_evs[1:]) / max(len(step_evs) - 1, 1) * 1e3
        world = ctx.world
        per_edge_tflop = 2 * (self.net.flops_per_pair(self.h, self.w) - 2 * self._enc_flops())
        result.update({
            "data": "synthetic: cached encoder tokens (random, unit scale) through the decoders + heads (seeded random-init weights, "
                    "no checkpoint available offline); matcher + Gauss-Newton blocks on a synthetic 256-keyframe graph of the same "
                    "size (circular trajectory over one smooth surface, SURVEY 8d config 5), fp16 descriptors; the matcher's input maps "
                    "of every edge batch (what the decoder would hand it) are assembled once and stay resident in HBM",
            "config": {"workload": f"{len(self.mine)} of {self.total_edges} graph edges per GPU, {self.K} keyframes at {self.h}x{self.w} "
                                   f"(BASELINE configs[4] per-GPU shard): symmetric decode from cached tokens ({self.dtype} trunk, fp16 heads) "
                                   "+ iter_proj/refine match in both directions on fp16 features + rays-GN blocks of the rank's "
                                   f"{e_dir} directed edges" + ("" if ctx.dist is None else " + RCCL all-gather of 36 doubles per directed edge")
                                   + f" + {dim}-unknown step ({args.gn_iters} GN iteration per step)",
                       "keyframes": self.K, "edges_per_gpu": len(self.mine), "global_edges": self.total_edges,
                       "edge_batch": args.edge_batch, "image": [self.h, self.w], "gn_iters": args.gn_iters,
                       "parallelism": f"edge-sharded x{world}", "launch": "eager (add_factors has one host sync per call, as the reference)"},
            "stage_ms": {k: round(v, 3) for k, v in stage_ms.items()},
            "stage_ms_note": "device time between stream events around add_factors (decode + match) and solve_GN_rays of every TIMED step",
            "match_valid_frac": round(vm, 4), "edges_kept": kept,
            "pose_max_abs_err_before_after": [err0, err1],
            "model_tflop_per_step": per_edge_tflop * len(self.mine) / 1e12,
            "roofline": gemm_roofline(prof, "_backend"),
            "hbm_rooflines": hbm,
            "dense_step": {"unknowns": dim, "avg_us": chol_us,
                           "kernels": "k_gn_assemble + blocked float64 Cholesky (k_chol_*) + forward / backward substitution + k_gn_retract"},
        })
        if ctx.rank == 0 and world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline_backend(args, self)

    def _enc_flops(self):
        c = self.net.cfg
        t = (self.h // 16) * (self.w // 16)
        E, r = c["enc_dim"], c["mlp_ratio"]
        return c["enc_depth"] * (2 * t * E * 3 * E + 2 * t * E * E + 4 * t * E * r * E + 4 * t * t * E) + 2 * t * 768 * E


# ----------------------------------------------------------------------------------------------------------------
def run_rank(args) -> int:
    from types import SimpleNamespace
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus N` (it spawns "
                         "the ranks itself) or under a launcher with --nproc-per-node equal to --gpus")
    dist = None
    if args.stub:
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a ROCm device (the HIP path is the product; no CPU fallback)")
        if args.single_device:
            local = 0
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        rdzv = os.environ.get("M3_BENCH_RDZV_FILE")                   # set by launch(): per-launch FileStore, no port
        if rdzv:
            kw = dict(init_method="file://" + rdzv)
        else:                                                        # external launcher: its MASTER_ADDR / MASTER_PORT
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            kw = {}
        if args.stub or args.dist_backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world, **kw)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, **kw)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")
    ctx = SimpleNamespace(rank=rank, world=world, local=local, dev=dev, dist=dist, group=None if dist is None else dist.group.WORLD)
    if not args.stub:
        from mast3r_slam import _ffi
        if not os.path.exists(_ffi.LIB_PATH):
            import __graft_entry__
            __graft_entry__.build()
    wl = (StubWorkload if args.stub else BackendWorkload if args.workload == "backend" else PairsWorkload)(args, ctx)
    if dist is not None and not args.stub and dev.type == "cuda":
        # RCCL's kernels run on the communicator's (normal-priority) stream: put the compute on a high-priority one, so that
        # where both want a CU the GEMM's workgroups (one per CU, 128 KiB of LDS) are dispatched first
        hp = torch.cuda.Stream(device=dev, priority=-1)
        hp.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(hp)
    wl.warm()
    for _ in range(args.warmup):
        wl.step()

    def barrier():
        wl.drain()
        if dist is not None:
            dist.barrier()
        if not args.stub:
            torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step(timed=True)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    n_ranks = dist.get_world_size() if dist is not None else 1
    result = {
        "metric": wl.metric,
        "value": n_ranks * wl.units_per_step * args.steps / elapsed,
        "unit": wl.unit,
        "n_gpus": n_ranks,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": wl.dtype,
        "ranks": {"world_size_of_the_process_group": n_ranks, "gpus_argument": args.gpus,
                  "started_by": "bench.py --gpus N (self-spawned ranks)" if os.environ.get("M3_BENCH_SPAWNED") else
                                ("external launcher (WORLD_SIZE in the environment)" if "WORLD_SIZE" in os.environ else "single process"),
                  "backend": None if dist is None else dist.get_backend()},
    }
    wl.report(result, args)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch(args))
    sys.exit(run_rank(args))


# ----------------------------------------------------------------------------------------------------------------
def _host_threads(requested: int) -> int:
    if requested:
        return requested
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))            # the GPU box grants a 16-core share per GPU


def cpu_baseline_pairs(args, net):
    """The CPU oracle (kind "port": our restatement, pinned to the reference's numpy twins for matching;
    torch fp32 for the network) timed on this host on ONE complete 512x512 pair - the same work the GPU
    does per pair: full encoder (2 views), both decoders, both heads, prep + iter_proj + refine_matches
    over all points, 10 GN iterations.  About 10 s on 16 cores."""
    import numpy as np
    import torch
    from mast3r_slam import synthetic
    from oracle import matching as om
    from oracle import model as omodel
    from oracle import tracking as ot
    h, w = args.image
    threads = _host_threads(args.cpu_threads)
    torch.set_num_threads(threads)
    note = lambda s: print(f"[cpu_baseline] {s}", file=sys.stderr, flush=True)
    wt, cfg = net.host_weights, net.cfg
    im = torch.from_numpy(np.stack([synthetic.textured_image(h, w, 0), synthetic.textured_image(h, w, 1)]))
    with torch.no_grad():
        t0 = time.perf_counter()
        f, pos = omodel.encode(wt, im, cfg)
        t_enc = time.perf_counter() - t0
        note(f"encoder done ({t_enc:.1f} s)")
        t0 = time.perf_counter()
        o1, o2 = omodel.decode(wt, f[:1], f[1:], pos, cfg)
        t_dec = time.perf_counter() - t0
        note(f"decoder done ({t_dec:.1f} s)")
        t0 = time.perf_counter()
        r1 = omodel.head(wt, "downstream_head1", o1, h, w, tuple(cfg["hooks"]))
        r2 = omodel.head(wt, "downstream_head2", o2, h, w, tuple(cfg["hooks"]))
        t_head = time.perf_counter() - t0
        note(f"heads done ({t_head:.1f} s)")
    X11, X21 = r1["pts3d"].numpy(), r2["pts3d"].numpy()
    D11, D21 = r1["desc"].numpy(), r2["desc"].numpy()
    n = h * w
    t0 = time.perf_counter()
    rays, tgt, p0 = om.prep_for_iter_proj(X11, X21, None)
    p, vproj = om.iter_proj(rays, tgt, p0, 10, 1e-8, 1e-6, "batch")
    om.refine_matches(D11, D21.reshape(1, n, -1), p.astype(np.int32), 3, 2)
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
            "sample": f"1 complete pair {h}x{w}: ViT-L encoder x2, both decoders, DPT + feature heads (torch-CPU fp32), "
                      f"prep + iter_proj + refine_matches on all {n} points (numpy oracle), 10 GN iterations (float64)",
            "seconds": {"encoder": round(t_enc, 2), "decoder": round(t_dec, 2), "heads": round(t_head, 2),
                        "match": round(t_match, 2), "gn": round(t_gn, 2)}}


def cpu_baseline_backend(args, wl):
    """The CPU oracle on ONE graph edge of the backend workload (kind "port"): symmetric decode of the edge from cached
    tokens (both decoders + both heads, twice: torch-CPU fp32), both matching directions on the synthetic keyframe
    graph (numpy oracle; descriptors widened from the stored half), the two directed edges' normal-equation blocks
    (float64 oracle).  The 1785-dim solve is not included (numpy's LAPACK would be timed, not the oracle)."""
    import numpy as np
    import torch
    from oracle import gn_rays as og
    from oracle import matching as om
    from oracle import model as omodel
    h, w, n = wl.h, wl.w, wl.n
    threads = _host_threads(args.cpu_threads)
    torch.set_num_threads(threads)
    note = lambda s: print(f"[cpu_baseline] {s}", file=sys.stderr, flush=True)
    wt, cfg = wl.net.host_weights, wl.net.cfg
    i, j = wl.ii[wl.mine.start], wl.jj[wl.mine.start]
    fi, fj = wl.frames[i].feat.float().cpu()[None], wl.frames[j].feat.float().cpu()[None]
    gy, gx = torch.meshgrid(torch.arange(h // 16), torch.arange(w // 16), indexing="ij")
    pos = torch.stack([gy.reshape(-1), gx.reshape(-1)], -1)
    t0 = time.perf_counter()
    with torch.no_grad():
        for a, b in ((fi, fj), (fj, fi)):
            o1, o2 = omodel.decode(wt, a, b, pos, cfg)
            omodel.head(wt, "downstream_head1", o1, h, w, tuple(cfg["hooks"]))
            omodel.head(wt, "downstream_head2", o2, h, w, tuple(cfg["hooks"]))
    t_dec = time.perf_counter() - t0
    note(f"symmetric decode done ({t_dec:.1f} s)")
    sc = wl.scene
    cpu = lambda x: x.cpu().numpy()
    P = cpu(sc["poses"]).astype(np.float64)

    def into(fr, pts):
        t, q, s = P[fr, :3], P[fr, 3:7], P[fr, 7]
        v = cpu(sc["Pw"][pts]).astype(np.float64) - t
        qv = -np.broadcast_to(q[:3], v.shape)
        u = 2.0 * np.cross(qv, v)
        return ((v + q[3] * u + np.cross(qv, u)) / s).astype(np.float32).reshape(1, h, w, 3)
    t0 = time.perf_counter()
    idxs = []
    for a, b in ((i, j), (j, i)):
        X11, X21 = cpu(sc["Xs"][a]).reshape(1, h, w, 3), into(a, b)
        D11, D21 = cpu(sc["D"][a].float())[None], cpu(sc["D"][b].float())[None]
        rays, tgt, p0 = om.prep_for_iter_proj(X11, X21, None)
        p, _ = om.iter_proj(rays, tgt, p0, 10, 1e-8, 1e-6, "batch")
        pr = om.refine_matches(D11, D21.reshape(1, n, -1), p.astype(np.int32), 3, 2)
        idxs.append((pr[0, :, 0] + w * pr[0, :, 1]).astype(np.int64))
    t_match = time.perf_counter() - t0
    note(f"matching done ({t_match:.1f} s)")
    Xs, Cs = cpu(sc["Xs"][[i, j]]), cpu(sc["C"][[i, j]])
    t0 = time.perf_counter()
    for (a, b), idx in zip(((0, 1), (1, 0)), idxs):
        og.edge_blocks(P[[i, j], :3], P[[i, j], 3:7], P[[i, j], 7], Xs, Cs, a, b, idx, np.ones(n, bool), np.full(n, 2.0, np.float32))
    t_blk = time.perf_counter() - t0
    note(f"blocks done ({t_blk:.1f} s)")
    total = t_dec + t_match + t_blk
    return {"value": 1.0 / total, "unit": "edges/s", "cores": threads, "kind": "port",
            "sample": f"1 graph edge at {h}x{w}: symmetric decode from cached tokens (2 x (both decoders + both heads), torch-CPU fp32), "
                      f"prep + iter_proj + refine_matches in both directions on all {n} points (numpy oracle), 2 directed-edge blocks (float64)",
            "seconds": {"decode": round(t_dec, 2), "match": round(t_match, 2), "blocks": round(t_blk, 2)}}


if __name__ == "__main__":
    main()
