#!/usr/bin/env python3
"""Timing probe for the big-tile GEMM kernels on three shapes (encoder qkv plain, proj + fp32 residual, the K = 7168 feature
MLP): M3_GEMM_TILE picks the kernel (130 = k_gemm_duo), M3_DUO_DBG its timing-experiment switches.  Interleaved rounds, median."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam_amd")]
import torch
from mast3r_slam import ops, _ffi
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
n_rep = int(sys.argv[1]) if len(sys.argv) > 1 else 5
CASES = [("qkv plain", 16384, 3072, 1024, "bf16"), ("fc1 gelu", 16384, 4096, 1024, "gelu"), ("proj acc", 16384, 1024, 1024, "acc"),
         ("fc2 acc", 16384, 1024, 4096, "acc"), ("feat fc2", 8192, 6400, 7168, "bf16")]
EPI = {"bf16": ops.EPI_BF16, "gelu": ops.EPI_BF16_GELU, "acc": ops.EPI_F32_ACCUM}
runs = []
for name, m, n, k, epi in CASES:
    a = torch.randn(m, k, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(n, k, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    b = torch.randn(n, generator=g).to(dev)
    x = torch.zeros(m, n, device=dev) if epi == "acc" else None
    runs.append((name, 2.0 * m * n * k, (lambda a=a, w=w, b=b, e=EPI[epi], x=x: ops.gemm(a, w, b, e, out=x, resid=x))))
def t(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for _, _, fn in runs:
    fn(); fn()
torch.cuda.synchronize()
res = {name: [] for name, _, _ in runs}
for _ in range(n_rep):
    for name, fl, fn in runs:
        res[name].append(t(fn))
occ = int(_ffi.lib().m3_gemm_duo_occupancy()) if hasattr(_ffi.lib(), "m3_gemm_duo_occupancy") else -1
print(f"tile={os.environ.get('M3_GEMM_TILE', '-')} dbg={os.environ.get('M3_DUO_DBG', '0')} occupancy(duo)={occ} | " +
      " | ".join(f"{name} {statistics.median(v):.1f} us {fl / statistics.median(v) / 1e6:.0f} TF" for (name, fl, _), v in zip(runs, res.values())), flush=True)
