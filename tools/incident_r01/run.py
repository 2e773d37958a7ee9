#!/usr/bin/env python3
"""Round-1 incident 'k_track_accum miscompile' (DESIGN.md section 9): rebuild the ORIGINAL kernel (tracking_v0.hip =
tracking.hip at 169907d^, float h[36] per point indexed by a running k++ in fully unrolled loops) under several
compiler settings and compare its 36 normal-equation sums with the float64 oracle on one problem.  One run, no loops:
the wrong entries were deterministic."""
import ctypes as C, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam_amd")]
import numpy as np, torch
from mast3r_slam import synthetic
from oracle import tracking as ot

pr = synthetic.tracking_problem(96, 128, seed=3)
n = pr["Xk"].shape[0]
Xf = pr["Xf_canon"][pr["idx"]]
T0 = np.array([0.01, -0.02, 0.005, 0.0, 0.01, 0.0, 1.0, 1.01], np.float32); T0[3:7] /= np.linalg.norm(T0[3:7])
ref = ot.normal_equations(Xf, pr["Xk"], T0.astype(np.float64), pr["Qk"], pr["valid"]) if hasattr(ot, "normal_equations") else None
dev = torch.device("cuda:0")
t = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).to(dev if dt is None else dev)
dXf, dXk, dQ, dV, dT = t(Xf), t(pr["Xk"]), t(pr["Qk"]), t(pr["valid"].astype(np.uint8)), t(T0)

def run(lib):
    L = C.CDLL(lib)
    L.m3_track_ws_doubles.restype = C.c_int64
    ws = torch.zeros(int(L.m3_track_ws_doubles()), dtype=torch.float64, device=dev)
    out = torch.zeros(36, dtype=torch.float64, device=dev)
    args = [C.c_void_p(x.data_ptr()) for x in (dXf, dXk, dQ, dV, dT, out, ws)] + [C.c_int(n), C.c_float(1.345), C.c_float(0.003), C.c_float(10.0), C.c_void_p(0)]
    rc = L.m3_track_normal_eq(*args)
    torch.cuda.synchronize()
    assert rc == 0, rc
    return out.cpu().numpy()

good = run(os.path.join(ROOT, "mast3r-slam_amd", "lib", "libm3slam_hip.so"))
print("current kernel (reference for the variants), H[0..6]:", good[:7])
variants = {"O3 (round-1 flags)": ["-O3"], "O3 -fno-slp-vectorize": ["-O3", "-fno-slp-vectorize"], "O1": ["-O1"],
            "O3 -ffp-contract=off": ["-O3", "-ffp-contract=off"]}
idx = {(i, j): k for k, (i, j) in enumerate((i, j) for i in range(7) for j in range(i, 7))}
for name, fl in variants.items():
    so = f"/tmp/track_v0_{abs(hash(name))}.so"
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", *fl, "-fPIC", "-std=c++17", "-fhip-fp32-correctly-rounded-divide-sqrt",
           "-shared", "-o", so, os.path.join(HERE, "tracking_v0.hip"), os.path.join(HERE, "core.hip")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        print(name, "BUILD FAILED", r.stderr[-400:]); continue
    got = run(so)
    rel = np.abs(got - good) / (np.abs(good) + 1e-30)
    bad = [(k, float(rel[k])) for k in range(36) if rel[k] > 1e-4]
    names = {v: k for k, v in idx.items()}
    print(f"{name}: max rel diff {rel.max():.3e}; entries off by > 1e-4:", [(names.get(k, ('g/cost', k)), f"{e:.2e}") for k, e in bad])
