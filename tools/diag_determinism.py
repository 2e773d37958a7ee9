#!/usr/bin/env python3
"""Is reconstruct_batch bit-reproducible (eager vs eager, eager vs hipGraph replay, heads forked or not) at a shape?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam_amd")]
import numpy as np, torch
from mast3r_slam import model as M, synthetic

h, w = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
cfg = M.TINY_CFG
net = M.Mast3rFull(weights=M.init_random_weights(cfg, seed=1), cfg=cfg, device=dev)
im1 = synthetic.textured_image(h, w, 0)[None]; im2 = synthetic.textured_image(h, w, 1)[None]

def diff(a, b, tag):
    for v in range(2):
        for k in a[v]:
            d = (a[v][k].float() - b[v][k].float()).abs().max().item()
            if d != 0:
                print(f"  {tag}: view {v} {k} max abs diff {d:.3e}")
    print(f"  {tag}: done")

for conc in (False, True):
    net.concurrent_heads = conc
    print("concurrent_heads", conc)
    e = [net.reconstruct_batch(im1, im2) for _ in range(3)]
    torch.cuda.synchronize()
    diff(e[0], e[1], "eager0 vs eager1"); diff(e[1], e[2], "eager1 vs eager2")
    g = net.graphed(1, h, w)
    q = g(im1, im2); q = tuple({k: t.clone() for k, t in o.items()} for o in q)
    q2 = g(im1, im2)
    torch.cuda.synchronize()
    diff(e[0], q, "eager vs graph"); diff(q, q2, "graph vs graph")
# encoder tokens
t0 = net.encode_tokens(net._as_images(im1))[0].clone(); t1 = net.encode_tokens(net._as_images(im1))[0]
print("encoder tokens equal:", torch.equal(t0, t1))

# ---- stage bisection -------------------------------------------------------------------------------
net.concurrent_heads = False
imgs = net._as_images(np.concatenate([im1, im2], 0))
ta, grid = net.encode_tokens(imgs); ta = ta.clone()
tb, _ = net.encode_tokens(imgs)
print("encoder (2 images) equal:", torch.equal(ta, tb))
m = grid[0] * grid[1]
taps_a = net.decode_tokens(ta[:m], ta[m:], 1, grid)
taps_a = [[t.clone() for t in v] for v in taps_a]
taps_b = net.decode_tokens(ta[:m], ta[m:], 1, grid)
for v in range(2):
    for i, (x, y) in enumerate(zip(taps_a[v], taps_b[v])):
        print(f"decoder view {v} tap {i} equal:", torch.equal(x, y), float((x.float() - y.float()).abs().max()))
ha = net.head("downstream_head1", taps_a[0], 1, grid); ha = {k: t.clone() for k, t in ha.items()}
hb = net.head("downstream_head1", taps_a[0], 1, grid)
for k in ha:
    print("head1", k, "equal:", torch.equal(ha[k], hb[k]))
