#!/usr/bin/env python3
"""Device time of the GEMM launches of one benchmark step (P = 8 pairs at 512x512), by shape and epilogue, through
the C ABI - and torch.mm (hipBLASLt) on the same operands as a yardstick measured in the same process (devices of
the pool differ by up to 12 %: compare ratios, not absolutes across runs).  Interleaved rounds, median."""
import os, sys, statistics
P = int(sys.argv[sys.argv.index("--pairs") + 1]) if "--pairs" in sys.argv else 8     # pairs per step: M = 2048 P (encoder), 1024 P (decoder / heads)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam_amd")]
import torch
from mast3r_slam import ops, _ffi
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
T = 1024
pos = torch.stack(torch.meshgrid(torch.arange(32), torch.arange(32), indexing="ij"), -1).reshape(-1, 2).to(dev)
inv = 1.0 / (100.0 ** (torch.arange(0, 32, 2, dtype=torch.float32) / 32.0))
ang = torch.arange(33, dtype=torch.float32)[:, None] * inv[None]
rtok = ops.rope_bound(pos.to(torch.int32).contiguous(), 32)      # position mode with the range promised (what the model uses); the f32 table of ops.rope_token_table(pos,
                                             # torch.stack([ang.cos(), ang.sin()], -1).to(dev)) selects the table mode

def mk(m, n, k, dt, groups):
    shp = (groups, m, k) if groups == 2 else (m, k)
    a = torch.randn(*shp, generator=g).to(dt).to(dev)
    w = [(torch.randn(n, k, generator=g) * 0.05).to(dt).to(dev) for _ in range(groups)]
    b = [torch.randn(n, generator=g).to(dev) for _ in range(groups)]
    return a, w, b

ME, MD = 2048 * P, 1024 * P
CASES = [  # name, m, n, k, epi, dtype, groups, launches per step
    ("enc qkv+rope", ME, 3072, 1024, "rope", torch.bfloat16, 1, 24),
    ("enc qkv plain", ME, 3072, 1024, "bf16", torch.bfloat16, 1, 0),
    ("enc proj f32acc", ME, 1024, 1024, "acc", torch.bfloat16, 1, 24),
    ("enc fc1 gelu", ME, 4096, 1024, "gelu", torch.bfloat16, 1, 24),
    ("enc fc2 f32acc", ME, 1024, 4096, "acc", torch.bfloat16, 1, 24),
    ("dec kv+rope x2", MD, 1536, 768, "rope", torch.bfloat16, 2, 12),
    ("dec qkv+rope x2", MD, 2304, 768, "rope", torch.bfloat16, 2, 12),
    ("dec q+rope x2", MD, 768, 768, "rope", torch.bfloat16, 2, 12),
    ("dec proj f32acc x2", MD, 768, 768, "acc", torch.bfloat16, 2, 24),
    ("dec fc1 gelu x2", MD, 3072, 768, "gelu", torch.bfloat16, 2, 12),
    ("dec fc2 f32acc x2", MD, 768, 3072, "acc", torch.bfloat16, 2, 12),
    ("feat fc1 gelu f16 x2", MD, 7168, 1792, "gelu", torch.float16, 2, 1),
    ("feat fc2 f16 x2", MD, 6400, 7168, "bf16", torch.float16, 2, 1),
]
EPI = {"bf16": ops.EPI_BF16, "gelu": ops.EPI_BF16_GELU, "acc": ops.EPI_F32_ACCUM, "rope": ops.EPI_BF16_ROPE}
runs = []
for name, m, n, k, epi, dt, groups, per_step in CASES:
    a, w, b = mk(m, n, k, dt, groups)
    x = torch.zeros((groups, m, n) if groups == 2 else (m, n), device=dev) if epi == "acc" else None
    rcols = n // 64 // 3 * 2 * 64 if n % 192 == 0 and n // 64 % 3 == 0 and epi == "rope" and "qkv" in name else n
    if groups == 2:
        if epi == "rope":
            fn = lambda a=a, w=w, b=b, rc=rcols: ops.gemm_grouped2(a, w[0], w[1], b[0], b[1], ops.EPI_BF16_ROPE, rope=(rtok, rc))
        else:
            fn = lambda a=a, w=w, b=b, e=EPI[epi], x=x: ops.gemm_grouped2(a, w[0], w[1], b[0], b[1], e, out=x, resid=x)
        ref = lambda a=a, w=w: (torch.mm(a[0], w[0].T), torch.mm(a[1], w[1].T))
        mm_ = lambda a=a, w=w, b=b: [torch.addmm(b[i].to(a.dtype), a[i], w[i].T) for i in range(2)]
    else:
        if epi == "rope":
            fn = lambda a=a, w=w, b=b, rc=rcols: ops.gemm_rope(a, w[0], b[0], rtok, rc)
        else:
            fn = lambda a=a, w=w, b=b, e=EPI[epi], x=x: ops.gemm(a, w[0], b[0], e, out=x, resid=x)
        ref = lambda a=a, w=w: torch.mm(a, w[0].T)
        mm_ = lambda a=a, w=w, b=b: [torch.addmm(b[0].to(a.dtype), a, w[0].T)]
    # the SAME work through the library: GEMM (+ bias) and then the epilogue as one separate elementwise pass over the output -
    # fp32 residual: x += y; GELU: erf GELU in place; RoPE: ONE in-place multiply as a stand-in (a real rotation reads two
    # tables on top: lower bound); plain: nothing
    xs = None if x is None else (list(x) if groups == 2 else [x])
    if epi == "acc":
        eq = lambda mm_=mm_, xs=xs: [xi.add_(yi) for xi, yi in zip(xs, mm_())]
    elif epi == "gelu":
        eq = lambda mm_=mm_: [torch.nn.functional.gelu(yi) for yi in mm_()]
    elif epi == "rope":
        eq = lambda mm_=mm_: [yi.mul_(1.0009765625) for yi in mm_()]
    else:
        eq = mm_
    runs.append((name, 2.0 * groups * m * n * k, fn, ref, per_step, int(_ffi.lib().m3_gemm_pick_tile(m, n, groups)), eq))

GRAPH = "--graph" in sys.argv        # time hipGraph replays of 16 launches (small launches are host-bound when issued eagerly)
_graphs = {}

def t(fn, n=8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if GRAPH:
        g_ = _graphs.get(id(fn))
        if g_ is None:
            fn(); torch.cuda.synchronize()
            g_ = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_):
                for _ in range(16):
                    fn()
            _graphs[id(fn)] = g_
        g_.replay()
        e0.record(); g_.replay(); e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 16 * 1e3
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for name, fl, fn, ref, _, _, eq in runs:
    fn(); ref(); eq()
torch.cuda.synchronize()
res = {name: ([], [], []) for name, *_ in runs}
for rnd in range(5):
    for name, fl, fn, ref, _, _, eq in runs:
        res[name][0].append(t(fn)); res[name][1].append(t(ref)); res[name][2].append(t(eq))
print(f"P = {P} pairs per step; M3_GEMM_TILE={os.environ.get('M3_GEMM_TILE', '-')}\n")
print("| launch | tile | us (ours, fused epilogue) | TFLOP/s | torch.mm us (no epilogue) | TFLOP/s | ours/mm time | torch.addmm + the epilogue as a separate pass, us | ours / that | per step ms |")
print("|---|---|---|---|---|---|---|---|---|---|")
tot = tot_eq = 0.0
for name, fl, fn, ref, per_step, tile, eq in runs:
    a, b, c = (statistics.median(res[name][i]) for i in range(3))
    tot += a * per_step / 1e3
    tot_eq += c * per_step / 1e3
    print(f"| {name} | {tile} | {a:.1f} | {fl / a / 1e6:.0f} | {b:.1f} | {fl / b / 1e6:.0f} | {a / b:.2f} | {c:.1f} | {a / c:.2f} | {a * per_step / 1e3:.2f} |")
print(f"\nsum over one step's launches: {tot:.2f} ms; the same work through torch.addmm + separate epilogue passes: {tot_eq:.2f} ms")
