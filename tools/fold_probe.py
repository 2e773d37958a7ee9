#!/usr/bin/env python3
"""Cost of the LayerNorm fold per launch: the consumer projections (fc1 + GELU, qkv + RoPE) with and without the folded
normalisation, the residual launches on an fp32 stream and on the hi / lo stream, and the LayerNorm kernel they replace.
M = 16384 (8 pairs) and 2048 (one pair).  Interleaved rounds, medians."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam_amd")]
import torch
from mast3r_slam import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
C = 1024
pos = torch.stack(torch.meshgrid(torch.arange(32), torch.arange(32), indexing="ij"), -1).reshape(-1, 2).to(torch.int32).to(dev)
def t(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for M in (16384, 2048):
    x = (torch.randn(M, C, generator=g) * 1.5).to(dev)
    gam = torch.ones(C, device=dev); bet = torch.zeros(C, device=dev)
    w1 = (torch.randn(4096, C, generator=g) * 0.05).half().to(dev); b1 = torch.randn(4096, generator=g).to(dev)
    wq = (torch.randn(3072, C, generator=g) * 0.05).half().to(dev); bq = torch.randn(3072, generator=g).to(dev)
    wp = (torch.randn(C, C, generator=g) * 0.05).half().to(dev); bp = torch.randn(C, generator=g).to(dev)
    a = torch.randn(M, C, generator=g).half().to(dev)
    hl = ops.ln_hl_buffers(M, C, dev)
    ops.gemm_ex(a, wp, bp, ops.EPI_F32, hl=hl)
    cs1 = w1.float().sum(1).contiguous(); csq = wq.float().sum(1).contiguous()
    xn = ops.layernorm(x, gam, bet, dtype=torch.float16)
    xs = x.clone()
    runs = {
        "layernorm kernel": lambda: ops.layernorm(x, gam, bet, dtype=torch.float16),
        "fc1+gelu plain": lambda: ops.gemm(xn, w1, b1, ops.EPI_BF16_GELU),
        "fc1+gelu plain, A = hi plane": lambda: ops.gemm(hl[0], w1, b1, ops.EPI_BF16_GELU),
        "fc1+gelu folded": lambda: ops.gemm_ex(hl[0], w1, b1, ops.EPI_BF16_GELU, fold_in=(hl[2], cs1)),
        "qkv+rope plain": lambda: ops.gemm_rope(xn, wq, bq, pos, 2048, q_cols=1024, q_scale=0.18),
        "qkv+rope folded": lambda: ops.gemm_ex(hl[0], wq, bq, ops.EPI_BF16_ROPE, rope=(pos, 2048, 1024, 0.18), fold_in=(hl[2], csq)),
        "proj fp32 stream": lambda: ops.gemm(a, wp, bp, ops.EPI_F32_ACCUM, out=xs, resid=xs),
        "proj hi/lo stream": lambda: ops.gemm_ex(a, wp, bp, ops.EPI_F32_ACCUM, hl=hl),
    }
    for fn in runs.values():
        fn(); fn()
    torch.cuda.synchronize()
    res = {k: [] for k in runs}
    for _ in range(7):
        for k, fn in runs.items():
            res[k].append(t(fn))
    print(f"M={M} dbg={os.environ.get('M3_FOLD_DBG', '0')} | " + " | ".join(f"{k} {statistics.median(v):.1f}" for k, v in res.items()), flush=True)
