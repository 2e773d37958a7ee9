#!/usr/bin/env python3
"""torch.mm (hipBLASLt) on the four encoder GEMM shapes of an 8-pair step, random bf16 operands, to be run under
`rocprofv3 --kernel-trace` (kernel names carry the library's macro tile / wave layout; the trace has LDS bytes, VGPR
counts, grid and workgroup sizes) and under `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` (matrix-pipe duty per
launch).  Yardstick only: the product never calls the library."""
import sys
import torch

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
P = 8
ME = 2048 * P
SHAPES = [("qkv", ME, 3072, 1024), ("proj", ME, 1024, 1024), ("fc1", ME, 4096, 1024), ("fc2", ME, 1024, 4096)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ops = []
for name, m, nn, k in SHAPES:
    a = torch.randn(m, k, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(nn, k, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    ops.append((name, m, nn, k, a, w))
for name, m, nn, k, a, w in ops:
    for _ in range(3):
        torch.mm(a, w.T)
torch.cuda.synchronize()
for name, m, nn, k, a, w in ops:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        torch.mm(a, w.T)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"{name}: {m}x{nn}x{k}  {us:.1f} us  {2.0 * m * nn * k / us / 1e6:.0f} TFLOP/s", flush=True)
