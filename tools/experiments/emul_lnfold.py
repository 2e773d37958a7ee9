"""EXPERIMENT (round 4, negative result - DESIGN.md section 10): python tools/experiments/emul_lnfold.py [plain|trained_like]
CPU emulation: LayerNorm folded into the consumer GEMM (raw x rounded to 16 bits, gamma folded into the weights, mean / rstd
applied in the epilogue) against the shipped form (normalised values rounded to 16 bits).  Encoder + decoder trunk only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # tools/experiments/ -> repo root
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam_amd"), os.path.join(ROOT, "tools")]
import torch, torch.nn.functional as F
import emul_precision as E
from oracle import model as OM
from mast3r_slam import model as M, synthetic

fam = sys.argv[1] if len(sys.argv) > 1 else "trained_like"
torch.set_num_threads(8)
cfg = dict(M.FULL_CFG)
w = M.init_random_weights(cfg, seed=0, family=fam)
h = wd = 512
im1 = torch.from_numpy(synthetic.textured_image(h, wd, 0)[None]); im2 = torch.from_numpy(synthetic.textured_image(h, wd, 1)[None])
run = lambda: OM.reconstruct(w, im1, im2, cfg)

class Fold:
    def __init__(self, R): self.R = R
    def __enter__(self):
        self.saved = (OM._ln, OM._lin)
        o_ln, o_lin = self.saved
        R = self.R
        def ln(x, w_, p):
            y = o_ln(x, w_, p)
            y._fold = (x, w_[p + ".weight"], w_[p + ".bias"])
            return y
        def lin(x, w_, p):
            f = getattr(x, "_fold", None)
            if f is None:
                return o_lin(x, w_, p)
            raw, gm, bt = f
            W, b = w_[p + ".weight"], w_[p + ".bias"]
            mu = raw.mean(-1, keepdim=True); rstd = (raw.var(-1, unbiased=False, keepdim=True) + OM.EPS).rsqrt()
            Wf = R(W * gm[None, :]); s = Wf.sum(1)
            bp = F.linear(bt, W) + b
            return rstd * (F.linear(R(raw), Wf) - mu * s) + bp
        OM._ln, OM._lin = ln, lin
        return self
    def __exit__(self, *a): OM._ln, OM._lin = self.saved

with torch.no_grad():
    ref = run()
    for name, R in (("bf16", E.RB), ("fp16", E.RH)):
        mode = dict(E.MODES["bf16+f16h" if name == "bf16" else "fp16+bf16pv"])
        with E.Emul(mode):
            a = run()
        with E.Emul(mode):
            with Fold(R):
                b = run()
        for v in range(2):
            print(f"{fam} {name} view {v+1}: shipped pts3d {E.rel(a[v]['pts3d'], ref[v]['pts3d']):.2e} desc {E.rel(a[v]['desc'], ref[v]['desc']):.2e} | "
                  f"LN folded pts3d {E.rel(b[v]['pts3d'], ref[v]['pts3d']):.2e} desc {E.rel(b[v]['desc'], ref[v]['desc']):.2e}", flush=True)
