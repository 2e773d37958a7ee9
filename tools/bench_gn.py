#!/usr/bin/env python3
"""Gauss-Newton legs alone on the benchmark scene (P = 8 pairs x 262144 points): tracking solve (10 iterations) and
one backend block pass at config-5 scale.  Run under rocprofv3 --kernel-trace --stats for per-kernel times."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam_amd")]
import numpy as np, torch
from mast3r_slam import config, matching, synthetic, tracker
dev = torch.device("cuda:0")
P, n = 8, 512 * 512
config.set_config({"matching": {"use_simple": False}})


def make_scene():
    """The benchmark's match / GN scene (bench.py PairsWorkload._make_scene): P smooth two-view scenes, confidences that pass
    the tracker's gates, keyframe pointmap = view-2 points under a known small Sim(3)."""
    g = synthetic.geometric_pair(512, 512, seed=1000, batch=P)
    rng = np.random.default_rng(2000)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ang = np.deg2rad(2.0)
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    Xk = (1.02 * g["X21"].reshape(P, n, 3).astype(np.float64) @ R.T + np.array([0.05, 0.0, 0.01])).astype(np.float32)
    u = lambda lo, hi: rng.uniform(lo, hi, size=(P, n)).astype(np.float32)
    return dict(X11=t(g["X11"]), X21=t(g["X21"]), D11=t(g["D11"]), D21=t(g["D21"]), Xk=t(Xk),
                Cf=t(u(1.0, 3.0)), Ck=t(u(1.0, 3.0)), Qf=t(u(1.0, 4.0)), Qk=t(u(1.0, 4.0)))


sc = make_scene()
tcfg = config.get_config()["tracking"]
ident = torch.tensor([0, 0, 0, 0, 0, 0, 1, 1], dtype=torch.float32, device=dev)
idx, valid = matching.match(sc["X11"], sc["X21"], sc["D11"], sc["D21"])
def gn():
    Xf, Qk, vo, vk, cnt = tracker.track_gather(sc["X11"].reshape(P, n, 3), sc["Cf"], sc["Ck"], sc["Qf"], sc["Qk"], idx, valid.reshape(P, n), tcfg["C_conf"], tcfg["Q_conf"])
    return tracker.opt_pose_ray_dist_sim3(Xf, sc["Xk"], ident, ident, Qk, vo, tcfg, fixed_iters=True)
for _ in range(3):
    gn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    out = gn()
e1.record(); torch.cuda.synchronize()
print(f"track_gather + 10-iteration GN solve, P={P}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us;  T_rel[0] = {out[1][0].tolist()}")
for _ in range(5):
    matching.match(sc["X11"], sc["X21"], sc["D11"], sc["D21"])
torch.cuda.synchronize()

# ---- backend blocks at config-5 scale: 64 keyframes x 262144 points, each linked to its previous 3 (both directions)
from mast3r_slam import kernels
K_, P_ = 64, 512 * 512
g = synthetic.gn_graph(K_, P_, 0, seed=17, chain=True, pose_noise=0.0)
args = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in g]
E = len(g[3])
for _ in range(2):
    kernels.gn_rays_blocks(*args)
torch.cuda.synchronize()
e0.record()
for _ in range(5):
    kernels.gn_rays_blocks(*args)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 5 * 1e3
print(f"gn_rays_blocks {E} directed edges x {P_} points: {us:.0f} us = {E * 10.8e6 / us / 1e6:.2f} TB/s of the 10.8 MB/edge algorithmic traffic")

# ---- fast-NN search: 4096 seeds x 262144 pixels x 24 dims
rng = np.random.default_rng(0)
def unit(n):
    v = rng.normal(size=(n, 24)).astype(np.float32); return v / np.linalg.norm(v, axis=1, keepdims=True)
Qn, DBn = torch.from_numpy(unit(4096))[None].to(dev), torch.from_numpy(unit(262144))[None].to(dev)
for name, q, d, m in (("fma fp32", Qn, DBn, "fma"), ("mfma fp32 (hi+lo)", Qn, DBn, "mfma"), ("mfma fp16", Qn.half(), DBn.half(), "mfma")):
    for _ in range(2):
        matching.nn_search(q, d, method=m)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10):
        matching.nn_search(q, d, method=m)
    e1.record(); torch.cuda.synchronize()
    print(f"nn_search 4096 x 262144 x 24 [{name}]: {e0.elapsed_time(e1) / 10 * 1e3:.0f} us")
