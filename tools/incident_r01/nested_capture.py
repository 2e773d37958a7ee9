#!/usr/bin/env python3
"""Round-1 incident: 'hipStreamEndCapture crashes when the CALLER has forked streams around decode_heads'
(model.py decode_heads forks head 2 onto a side stream).  Each case runs in its own child process and reports
ok / exception / signal.  Cases:
  A  torch only: two-level event fork/join inside a capture, all streams created BEFORE the capture
  B  as A, inner side stream created lazily INSIDE the capture (what decode_heads did on first use)
  C  as B, and the inner branch's output gets record_stream(outer stream) inside the capture
  D  the model: caller forks two chains (each reconstruct_batch with its forked heads) inside one capture
"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

CASES = {}
CASES["A"] = '''
import torch
d = torch.device("cuda:0"); x = torch.ones(1 << 20, device=d)
s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
def inner(cur, side, x):
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        b = x * 3
    a = x * 2
    cur.wait_stream(side)
    return a + b
for _ in range(2):
    inner(torch.cuda.current_stream(), s3, x)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    with torch.cuda.stream(s1):
        y1 = inner(s1, s3, x)
    y0 = inner(cur, s2, x)
    cur.wait_stream(s1)
    out = y0 + y1
g.replay(); torch.cuda.synchronize(); print("value", float(out[0]))
'''
# A0: one level only (the outer fork; no inner fork) - the shape GraphedReconstruct captures every day
CASES["A0"] = CASES["A"].replace("        y1 = inner(s1, s3, x)", "        y1 = x * 5").replace("    y0 = inner(cur, s2, x)", "    y0 = x * 7")
# A1: inner fork only on the ORIGIN stream's branch, the forked branch s1 stays linear
CASES["A1"] = CASES["A"].replace("        y1 = inner(s1, s3, x)", "        y1 = x * 5")
# A2: inner fork only on the FORKED branch s1 (a fork from a stream that itself joined the capture by an event)
CASES["A2"] = CASES["A"].replace("    y0 = inner(cur, s2, x)", "    y0 = x * 7")
# A3: as A2 but no tensor allocation inside the capture (pre-allocated outputs, in-place ops): allocator out of the picture
CASES["A3"] = '''
import torch
d = torch.device("cuda:0"); x = torch.ones(1 << 20, device=d)
a, b, c, o = (torch.zeros_like(x) for _ in range(4))
s1, s3 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    with torch.cuda.stream(s1):
        s3.wait_stream(s1)
        with torch.cuda.stream(s3):
            torch.mul(x, 3, out=b)
        torch.mul(x, 2, out=a)
        s1.wait_stream(s3)
        a.add_(b)
    torch.mul(x, 7, out=c)
    cur.wait_stream(s1)
    torch.add(a, c, out=o)
g.replay(); torch.cuda.synchronize(); print("value", float(o[0]))
'''
CASES["B"] = CASES["A"].replace("def inner(cur, side, x):\n    side.wait_stream(cur)", "POOL = {}\ndef inner(cur, side, x):\n    side = POOL.setdefault(cur.cuda_stream, None) or POOL.__setitem__(cur.cuda_stream, torch.cuda.Stream()) or POOL[cur.cuda_stream]\n    side.wait_stream(cur)").replace("for _ in range(2):\n    inner(torch.cuda.current_stream(), s3, x)\n", "")
CASES["C"] = CASES["B"].replace("    cur.wait_stream(side)\n    return a + b", "    cur.wait_stream(side)\n    b.record_stream(cur)\n    return a + b")
CASES["D"] = '''
import sys
sys.path[:0] = [%r, %r]
import numpy as np, torch
from mast3r_slam import model as M, synthetic
dev = torch.device("cuda:0")
net = M.Mast3rFull(weights=M.init_random_weights(M.TINY_CFG, seed=1), cfg=M.TINY_CFG, device=dev)
net.concurrent_heads = True
h, w = 128, 256
a = [torch.from_numpy(synthetic.textured_image(h, w, s)[None]).to(dev) for s in range(4)]
for _ in range(2):
    net.reconstruct_batch(a[0], a[1]); net.reconstruct_batch(a[2], a[3])
torch.cuda.synchronize()
s1 = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    with torch.cuda.stream(s1):
        o_b = net.reconstruct_batch(a[2], a[3])
    o_a = net.reconstruct_batch(a[0], a[1])
    cur.wait_stream(s1)
g.replay(); torch.cuda.synchronize()
e_a = net.reconstruct_batch(a[0], a[1]); e_b = net.reconstruct_batch(a[2], a[3])
print("equal to eager:", all(torch.equal(o_a[v][k], e_a[v][k]) and torch.equal(o_b[v][k], e_b[v][k]) for v in range(2) for k in e_a[v]))
''' % (ROOT, os.path.join(ROOT, "mast3r-slam_amd"))

for name in (sys.argv[1:] or sorted(CASES)):
    r = subprocess.run([sys.executable, "-c", CASES[name]], capture_output=True, text=True, timeout=300)
    tail = (r.stdout.strip().splitlines() or [""])[-1]
    err = [l for l in r.stderr.strip().splitlines() if "amdgpu.ids" not in l][-3:]
    print(f"case {name}: rc={r.returncode} stdout[-1]={tail!r} stderr[-3:]={err}", flush=True)
