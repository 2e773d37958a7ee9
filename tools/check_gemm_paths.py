#!/usr/bin/env python3
"""Prints a digest of GEMM outputs for a fixed list of large dense problems and epilogues.  Run it under different
dispatcher overrides (M3_GEMM_4W=0/1, M3_GEMM_TILE=...) and diff the text: all big-tile kernels accumulate K in the
same order, so the digests must be identical.  Also checks each output against a float64 reference on a sample."""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam_amd")]
import torch
from mast3r_slam import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
pos = torch.stack(torch.meshgrid(torch.arange(32), torch.arange(32), indexing="ij"), -1).reshape(-1, 2).to(dev)
inv = 1.0 / (100.0 ** (torch.arange(0, 32, 2, dtype=torch.float32) / 32.0))
ang = torch.arange(33, dtype=torch.float32)[:, None] * inv[None]
rtok = ops.rope_token_table(pos, torch.stack([ang.cos(), ang.sin()], -1).to(dev))
CASES = [(16384, 3072, 1024, "rope", torch.bfloat16), (16384, 1024, 1024, "acc", torch.bfloat16),
         (16384, 4096, 1024, "gelu", torch.bfloat16), (16384, 1024, 4096, "acc", torch.bfloat16),
         (8192, 7168, 1792, "gelu", torch.float16), (8192, 6400, 7168, "bf16", torch.float16),
         (4000, 2048, 512, "bf16", torch.bfloat16), (4100, 1000, 384, "add", torch.bfloat16),
         (8192, 2048, 128, "relu", torch.float16)]
EPI = {"bf16": ops.EPI_BF16, "gelu": ops.EPI_BF16_GELU, "acc": ops.EPI_F32_ACCUM, "rope": ops.EPI_BF16_ROPE,
       "add": ops.EPI_BF16_ADD, "relu": ops.EPI_BF16_RELU}
for m, n, k, epi, dt in CASES:
    a = torch.randn(m, k, generator=g).to(dt).to(dev)
    w = (torch.randn(n, k, generator=g) * 0.05).to(dt).to(dev)
    b = torch.randn(n, generator=g).to(dev)
    if epi == "rope":
        out = ops.gemm_rope(a, w, b, rtok, n // 3 * 2, q_cols=n // 3, q_scale=0.25)
    elif epi == "acc":
        x = torch.randn(m, n, generator=g).to(dev); r = x.clone()
        out = ops.gemm(a, w, b, EPI[epi], out=x, resid=x)
    elif epi == "add":
        r = torch.randn(m, n, generator=g).to(dt).to(dev)
        out = ops.gemm(a, w, b, EPI[epi], resid=r)
    else:
        out = ops.gemm(a, w, b, EPI[epi])
    torch.cuda.synchronize()
    err = float("nan")
    if epi in ("bf16", "acc", "add", "relu"):
        rows = torch.tensor([0, 1, 255, 256, m // 2 + 3, m - 1], device=dev)
        ref = a[rows].double() @ w.double().T + b.double()
        if epi in ("acc", "add"): ref = ref + r[rows].double()
        if epi == "relu": ref = ref.clamp_min(0)
        err = float((out[rows].double() - ref).abs().max() / ref.abs().max())
    print(f"{m}x{n}x{k} {epi} {str(dt)[6:]} tile={ops.gemm_pick_tile(m, n) if hasattr(ops, 'gemm_pick_tile') else '?'} "
          f"sha={hashlib.sha1(out.cpu().contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()[:16]} rel_err={err:.2e}")
