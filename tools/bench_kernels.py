#!/usr/bin/env python3
"""The reference's own kernel benchmark (benchmark_all_kernels.py: iter_proj, refine_matches, gauss_newton_rays /
_points / _calib on random inputs; published numbers for Apple M4 Pro in BASELINE.md section 1) re-run on the
MI355X path at the SAME shapes.  Two timings per configuration:
  level-1  numpy in / numpy out through mast3r_slam.kernels (what the reference's dispatch module does; includes
           the PCIe copies and one synchronisation per call)
  device   tensors resident on the GPU, HIP-event time of the call
Prints a markdown table (kept under profiles/)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "mast3r-slam_amd")):
    sys.path.insert(0, p)
from mast3r_slam import kernels, synthetic  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(42)


def t_host(fn, n=10):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def t_dev(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def to_dev(*a):
    return [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in a]


rows = []
# iter_proj (benchmark_all_kernels.py:45-104): random rays / points / p_init
for (b, h, w, n), ref in (((1, 64, 64, 1000), "3.60 / 11.22"), ((1, 128, 128, 5000), "11.42 / 0.89"),
                          ((1, 256, 256, 20000), "41.89 / 1.74"), ((2, 384, 512, 50000), "254.4 / 8.72")):
    rays = rng.standard_normal((b, h, w, 9)).astype(np.float32)
    pts = rng.standard_normal((b, n, 3)).astype(np.float32)
    p0 = np.stack([rng.random((b, n)) * (w - 1), rng.random((b, n)) * (h - 1)], -1).astype(np.float32)
    d = to_dev(rays, pts, p0)
    rows.append((f"iter_proj b={b} {h}x{w}, {n} pts, 10 LM iters", ref,
                 t_host(lambda: kernels.iter_proj(rays, pts, p0)), t_dev(lambda: kernels.iter_proj(*d))))
# refine_matches (:107-160): D = 32 / 64, radius 3, dilation 2
for (b, h, w, dd, n), ref in (((1, 384, 512, 32, 1000), "41.5 / 9.04"), ((1, 384, 512, 64, 5000), "206.1 / 14.41"),
                              ((2, 384, 512, 64, 10000), "826.9 / 56.82")):
    D11 = rng.standard_normal((b, h, w, dd)).astype(np.float32)
    D21 = rng.standard_normal((b, n, dd)).astype(np.float32)
    p1 = np.stack([rng.integers(0, w, (b, n)), rng.integers(0, h, (b, n))], -1).astype(np.int32)
    d = to_dev(D11, D21, p1)
    rows.append((f"refine_matches b={b} {h}x{w}, D={dd}, {n} pts (r=3, dil=2)", ref,
                 t_host(lambda: kernels.refine_matches(D11, D21, p1, 3, 2)), t_dev(lambda: kernels.refine_matches(*d, 3, 2))))
# Gauss-Newton (:163-260): 3 iterations, pin = 1
K = np.array([[500.0, 0, 320], [0, 500.0, 240], [0, 0, 1]], np.float32)
for (kf, npts, ne), refs in (((5, 200, 8), ("132.7 / 5.70", "121.4 / 5.37", "4.58 / 3.58")),
                             ((10, 500, 15), ("595.9 / 10.87", "577.5 / 13.81", "28.49 / 9.43")),
                             ((20, 1000, 30), ("2384.8 / 43.54", "2289.9 / 43.74", "91.95 / 34.94"))):
    g = synthetic.gn_graph(kf, npts, num_edges=ne, seed=42)[:8]
    d = to_dev(*g)
    rows.append((f"gauss_newton_rays {kf} KF / {npts} pts / {ne} edges, 3 iters", refs[0],
                 t_host(lambda: kernels.gauss_newton_rays(*g, max_iter=3)), t_dev(lambda: kernels.gauss_newton_rays(*d, max_iter=3))))
    rows.append((f"gauss_newton_points {kf} KF / {npts} pts / {ne} edges", refs[1],
                 t_host(lambda: kernels.gauss_newton_points(*g, max_iter=3)), t_dev(lambda: kernels.gauss_newton_points(*d, max_iter=3))))
    gc = list(g); gc[1] = np.abs(g[1]) + 0.1                                      # positive depths, as the reference benchmark
    dc = to_dev(*gc)
    rows.append((f"gauss_newton_calib {kf} KF / {npts} pts / {ne} edges", refs[2],
                 t_host(lambda: kernels.gauss_newton_calib(gc[0], gc[1], gc[2], K, *gc[3:], (640, 480), max_iter=3)),
                 t_dev(lambda: kernels.gauss_newton_calib(dc[0], dc[1], dc[2], K, *dc[3:], (640, 480), max_iter=3))))

print("| configuration (reference benchmark_all_kernels.py) | reference numpy / Metal on M4 Pro, ms | MI355X level-1 (numpy in/out), ms | MI355X device-resident, ms |")
print("|---|---|---|---|")
for name, ref, th, td in rows:
    print(f"| {name} | {ref} | {th:.3f} | {td:.3f} |")
