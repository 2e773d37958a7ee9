#!/usr/bin/env python3
"""Attention launches of one benchmark step in isolation (encoder: 16 images x 16 heads x 1024 tokens; decoder: 16 x 12),
prescaled and plain kernels; run under rocprofv3 for counters."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam_amd")]
import torch
from mast3r_slam import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (b, h, t) in ((16, 16, 1024), (16, 12, 1024)):
    c = h * 64
    qkv = (torch.randn(b * t, 3 * c, generator=g)).bfloat16().to(dev)
    out = torch.empty(b * t, c, dtype=torch.bfloat16, device=dev)
    for pre in (False, True):
        fn = lambda: ops.attention(qkv, qkv[:, c:], qkv[:, 2 * c:], out, nbatch=b, heads=h, tq=t, tk=t, q_row_stride=3 * c,
                                   kv_row_stride=3 * c, o_row_stride=c, q_batch_stride=t * 3 * c, kv_batch_stride=t * 3 * c,
                                   o_batch_stride=t * c, prescaled=pre)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"attention b{b} h{h} t{t} prescaled={pre}: {us:.1f} us = {4.0 * b * h * t * t * 64 / us / 1e6:.0f} TFLOP/s")
