#!/usr/bin/env python3
"""What the fused RoPE epilogue costs per launch: the q|k|v projection with a plain 16-bit store, with the rotation computed
per element (v_sin / v_cos), and with the per-workgroup LDS cos / sin table (ops.rope_bound).  Interleaved rounds, medians."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam_amd")]
import torch
from mast3r_slam import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
gy, gx = torch.meshgrid(torch.arange(32), torch.arange(32), indexing="ij")
mk = lambda: torch.stack([gy.reshape(-1), gx.reshape(-1)], -1).to(torch.int32).to(dev).contiguous()
plain_pos, tab_pos = mk(), ops.rope_bound(mk(), 32)
def t(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for (M, N, K, rc) in ((16384, 3072, 1024, 2048), (16384, 1536, 768, 768), (2048, 3072, 1024, 2048), (2048, 768, 768, 768)):
    for dt in (torch.float16, torch.bfloat16):
        a = torch.randn(M, K, generator=g).to(dt).to(dev)
        w = (torch.randn(N, K, generator=g) * 0.05).to(dt).to(dev); b = torch.randn(N, generator=g).to(dev)
        runs = {
            "plain store": lambda: ops.gemm_ex(a, w, b, ops.EPI_BF16),
            "rope per element": lambda: ops.gemm_ex(a, w, b, ops.EPI_BF16_ROPE, rope=(plain_pos, rc, rc // 2, 0.18)),
            "rope LDS table": lambda: ops.gemm_ex(a, w, b, ops.EPI_BF16_ROPE, rope=(tab_pos, rc, rc // 2, 0.18)),
        }
        for fn in runs.values():
            fn(); fn()
        torch.cuda.synchronize()
        res = {k: [] for k in runs}
        for _ in range(7):
            for k, fn in runs.items():
                res[k].append(t(fn))
        print(f"{M}x{N}x{K} rope_cols {rc} {str(dt)[6:]} | " + " | ".join(f"{k} {statistics.median(v):.1f}" for k, v in res.items()), flush=True)
