#!/usr/bin/env python3
"""Run-to-run bit reproducibility of single operators at given shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam_amd")]
import torch
from mast3r_slam import ops, _ffi
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)

def rep(name, fn, n=6):
    outs = [fn().clone() for _ in range(n)]
    torch.cuda.synchronize()
    bad = [i for i in range(1, n) if not torch.equal(outs[0], outs[i])]
    print(f"{name}: {'deterministic' if not bad else 'DIFFERS in runs ' + str(bad)}", flush=True)

for (b, h, t) in ((2, 16, 672), (1, 16, 672), (2, 16, 640), (2, 12, 576), (2, 16, 1024)):
    c = h * 64
    qkv = torch.randn(b * t, 3 * c, generator=g).bfloat16().to(dev)
    def attn():
        out = torch.zeros(b * t, c, dtype=torch.bfloat16, device=dev)
        ops.attention(qkv, qkv[:, c:], qkv[:, 2 * c:], out, nbatch=b, heads=h, tq=t, tk=t, q_row_stride=3 * c,
                      kv_row_stride=3 * c, o_row_stride=c, q_batch_stride=t * 3 * c, kv_batch_stride=t * 3 * c, o_batch_stride=t * c)
        return out
    rep(f"attention b{b} h{h} t{t}", attn)
for (m, n, k) in ((1344, 3072, 1024), (1344, 4096, 1024), (1344, 1024, 4096), (1344, 1024, 1024), (672, 3072, 1024), (1280, 4096, 1024)):
    a = torch.randn(m, k, generator=g).bfloat16().to(dev)
    w = (torch.randn(n, k, generator=g) * 0.05).bfloat16().to(dev)
    bias = torch.randn(n, generator=g).to(dev)
    tile = _ffi.lib().m3_gemm_pick_tile(m, n, 1)
    rep(f"gemm {m}x{n}x{k} tile {tile} bf16", lambda: ops.gemm(a, w, bias, ops.EPI_BF16))
    rep(f"gemm {m}x{n}x{k} tile {tile} gelu", lambda: ops.gemm(a, w, bias, ops.EPI_BF16_GELU))
    r = torch.randn(m, n, generator=g).to(dev)
    rep(f"gemm {m}x{n}x{k} tile {tile} f32acc", lambda: ops.gemm(a, w, bias, ops.EPI_F32_ACCUM, resid=r))
x = torch.randn(1344, 1024, generator=g).to(dev); gm = torch.randn(1024, generator=g).to(dev)
rep("layernorm 1344", lambda: ops.layernorm(x, gm, gm))
