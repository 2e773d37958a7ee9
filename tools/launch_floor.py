#!/usr/bin/env python3
"""Per-launch floor of a dependent kernel chain replayed from a hipGraph on this device: N launches of a trivial kernel
(m3_cast_f32_dt on 256 elements), each reading what the previous one wrote, captured once and replayed.  The batch-1
(one pair per frame) pipeline is ~640 dependent launches per pair; this is what they cost before doing any work."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam_amd")]
import torch
from mast3r_slam import ops

dev = torch.device("cuda:0")
x = torch.zeros(256, dtype=torch.float32, device=dev)
for n in (200, 1000):
    def chain():
        y = x
        for _ in range(n // 2):
            h = ops.cast_f32(y, torch.bfloat16)          # f32 -> bf16
            y = h.float()                                # torch elementwise kernel back to f32: the chain stays dependent
        return y
    chain(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        keep = chain()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"{n} dependent trivial launches in a hipGraph: {e0.elapsed_time(e1) / 10 * 1e3 / n:.2f} us per launch")
