#!/usr/bin/env python3
"""Where do the 64x64-tile GEMM's 16-bit outputs differ from run to run?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam_amd")]
import torch
from mast3r_slam import ops, _ffi
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
m, n, k = 1344, 3072, 1024
a = torch.randn(m, k, generator=g).bfloat16().to(dev)
w = (torch.randn(n, k, generator=g) * 0.05).bfloat16().to(dev)
bias = torch.randn(n, generator=g).to(dev)
print("tile", _ffi.lib().m3_gemm_pick_tile(m, n, 1))
ref = ops.gemm(a, w, bias, ops.EPI_F32).clone()
ref2 = ops.gemm(a, w, bias, ops.EPI_F32)
print("f32 deterministic:", torch.equal(ref, ref2))
want = ref.bfloat16()
for run in range(4):
    out = ops.gemm(a, w, bias, ops.EPI_BF16)
    bad = (out != want).nonzero()
    print(f"run {run}: {bad.shape[0]} elements differ from bf16(f32 result)")
    if bad.shape[0]:
        mm, nn = bad[:, 0], bad[:, 1]
        print("   rows%64:", sorted(set((mm % 64).tolist()))[:40])
        print("   cols%64:", sorted(set((nn % 64).tolist()))[:70])
        print("   tiles (m/64, n/64):", sorted(set(zip((mm // 64).tolist(), (nn // 64).tolist())))[:20])
        d = (out.float() - want.float())[mm, nn]
        print("   max abs diff", float(d.abs().max()), "example", bad[:5].tolist(), out[mm[:5], nn[:5]].tolist(), want[mm[:5], nn[:5]].tolist())
