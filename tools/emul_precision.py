#!/usr/bin/env python3
"""CPU emulation of 16-bit operand rounding on the fp32 oracle network (oracle/model.py): where does a precision mode
lose the pointmap?  Test / analysis infrastructure only (imports oracle/).

    python tools/emul_precision.py [--family plain|trained_like] [--res H W] [--depth ENC DEC] [--stats] [--stages]

Every GEMM / convolution input, q, k (after RoPE), v and the softmax probabilities are rounded to the 16-bit type of
the stage (what the HIP kernels do: fp32 accumulation, fp32 residual stream, fp32 LayerNorm statistics); the rel-L2
distance of the outputs to the un-rounded run is printed per mode:
    bf16        bf16 everywhere
    bf16+f16h   bf16 trunk (encoder + decoders), fp16 heads (precision="bf16", BASELINE configs[1])
    fp16        fp16 everywhere (precision="fp16", load_mast3r's default)
    fp16+bf16pv fp16 everywhere except v and the softmax probabilities (bf16: the fast attention loop's operand type)
--stages additionally rounds one stage at a time (bf16) to split the error.
--stats prints what the "trained_like" family is built to show: residual-stream outlier ratio, LayerNorm gain spread,
largest attention logit per row, largest GELU input.
DESIGN.md section 4 quotes the numbers; the GPU reproduces the emulation to ~3 digits on the plain family.
"""
from __future__ import annotations

import argparse
import math
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "mast3r-slam_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from mast3r_slam import model as M, synthetic  # noqa: E402
from oracle import model as OM  # noqa: E402

RB = lambda x: x.bfloat16().float()
RH = lambda x: x.half().float()


def stage_of(p: str) -> str:
    if p.startswith("enc_blocks") or p.startswith("patch"):
        return "enc"
    if p.startswith("dec_blocks") or p == "decoder_embed":
        return "dec"
    if "head_local_features" in p:
        return "feat"
    return "dpt"


class Emul:
    """Context manager patching oracle.model so that the operands of the selected stages are rounded.
    rnd: {stage: rounding function}; stages: enc, dec, dpt, feat, attn_enc, attn_dec."""

    def __init__(self, rnd: dict, stats: dict | None = None):
        self.rnd, self.stats = rnd, stats

    def __enter__(self):
        self.saved = (OM._lin, OM._conv, OM._mha, OM.rope2d, torch.softmax, F.conv_transpose2d, OM.self_attn, OM.cross_attn,
                      F.gelu)
        o_lin, o_conv, o_mha, o_rope, o_sm, o_ct, o_sa, o_ca, o_gelu = self.saved
        rnd, st = self.rnd, self.stats
        ident = lambda x: x
        cur = {"attn": ident}
        OM._lin = lambda x, w_, p: o_lin(rnd.get(stage_of(p), ident)(x), w_, p)
        OM._conv = lambda x, w_, p, stride=1, padding=0: o_conv(rnd.get("dpt", ident)(x), w_, p, stride, padding)
        F.conv_transpose2d = lambda x, wt, b=None, **k: o_ct(rnd.get("dpt", ident)(x), wt, b, **k)

        def mha(q, k, v, heads):
            r = cur.get("pv", cur["attn"])
            return o_mha(q, k, r(v), heads)                      # q, k are rounded after RoPE (as the fused epilogue does)
        OM._mha = mha
        OM.rope2d = lambda x, p, c, s: cur["attn"](o_rope(x, p, c, s))

        def softmax(x, dim=-1):
            if st is not None:
                st.setdefault("max_logit", []).append(float(x.amax(-1).float().mean()))
                st.setdefault("max_logit_p99", []).append(float(x.amax(-1).flatten().kthvalue(max(1, int(0.99 * x.amax(-1).numel()))).values))
            return cur.get("pv", cur["attn"])(o_sm(x, dim=dim))
        torch.softmax = softmax

        def sa(x, w, p, heads, pos, cos, sin):
            cur["attn"] = rnd.get("attn_enc" if p.startswith("enc") else "attn_dec", ident)
            if "pv" in rnd:
                cur["pv"] = rnd["pv"]                            # v and the probabilities in their own type (q, k keep "attn")
            return o_sa(x, w, p, heads, pos, cos, sin)

        def ca(x, y, w, p, heads, px, py, cos, sin):
            cur["attn"] = rnd.get("attn_dec", ident)
            if "pv" in rnd:
                cur["pv"] = rnd["pv"]
            return o_ca(x, y, w, p, heads, px, py, cos, sin)
        OM.self_attn, OM.cross_attn = sa, ca
        if st is not None:
            def gelu(x):
                st.setdefault("gelu_in_absmax", []).append(float(x.abs().max()))
                st.setdefault("gelu_in_frac_gt6", []).append(float((x.abs() > 6).float().mean()))
                return o_gelu(x)
            F.gelu = gelu
        return self

    def __exit__(self, *a):
        (OM._lin, OM._conv, OM._mha, OM.rope2d, torch.softmax, F.conv_transpose2d, OM.self_attn, OM.cross_attn, F.gelu) = self.saved


MODES = {
    "bf16": dict(enc=RB, dec=RB, dpt=RB, feat=RB, attn_enc=RB, attn_dec=RB),
    "bf16+f16h": dict(enc=RB, dec=RB, dpt=RH, feat=RH, attn_enc=RB, attn_dec=RB),
    "fp16": dict(enc=RH, dec=RH, dpt=RH, feat=RH, attn_enc=RH, attn_dec=RH),
    "fp16+bf16pv": dict(enc=RH, dec=RH, dpt=RH, feat=RH, attn_enc=RH, attn_dec=RH, pv=RB),   # q, k fp16; v and P bf16
}


def rel(a, b):
    return float((a - b).norm() / b.norm())


def stream_stats(w, im, cfg):
    """Outlier ratio of the fp32 residual stream after every encoder block: max over channels of the per-channel median
    |x| divided by the median over channels."""
    x = OM.normalize_image(im)
    pos = OM.patch_positions(x.shape[2], x.shape[3])
    cos, sin = OM.rope_tables(max(x.shape[2], x.shape[3]) // 16 + 1)
    x = F.conv2d(x, w["patch_embed.proj.weight"], w["patch_embed.proj.bias"], stride=16).flatten(2).transpose(1, 2)
    out = []
    for i in range(cfg["enc_depth"]):
        p = f"enc_blocks.{i}"
        x = x + OM.self_attn(OM._ln(x, w, p + ".norm1"), w, p + ".attn", cfg["enc_heads"], pos, cos, sin)
        x = x + OM.mlp(OM._ln(x, w, p + ".norm2"), w, p + ".mlp")
        ch = x.abs().flatten(0, 1).median(0).values
        out.append((float(ch.median()), float(ch.max() / ch.median())))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--family", default="plain", choices=("plain", "trained_like"))
    ap.add_argument("--res", type=int, nargs=2, default=[512, 512])
    ap.add_argument("--depth", type=int, nargs=2, default=None, metavar=("ENC", "DEC"))
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--stats", action="store_true")
    ap.add_argument("--stages", action="store_true")
    ap.add_argument("--threads", type=int, default=0)
    a = ap.parse_args()
    torch.set_num_threads(a.threads or min(16, os.cpu_count() or 1))
    cfg = dict(M.FULL_CFG)
    if a.depth:
        e, d = a.depth
        cfg.update(enc_depth=e, dec_depth=d, hooks=(0, max(1, d // 2), max(1, d * 3 // 4), d))
    h, wd = a.res
    w = M.init_random_weights(cfg, seed=a.seed, family=a.family)
    im1 = torch.from_numpy(synthetic.textured_image(h, wd, 0)[None])
    im2 = torch.from_numpy(synthetic.textured_image(h, wd, 1)[None])
    run = lambda: OM.reconstruct(w, im1, im2, cfg)
    with torch.no_grad():
        st = {} if a.stats else None
        t0 = time.time()
        with Emul({}, st):
            ref = run()
        print(f"fp32 reference: {time.time() - t0:.1f} s   family={a.family} res={h}x{wd} depth={cfg['enc_depth']}+{cfg['dec_depth']}", flush=True)
        if a.stats:
            g = torch.cat([v for k, v in w.items() if ".norm" in k and k.endswith(".weight")])
            print(f"LayerNorm gain: p1 {float(g.kthvalue(max(1, g.numel() // 100)).values):.3f}  p99 "
                  f"{float(g.kthvalue(g.numel() * 99 // 100).values):.3f}")
            ml, p99 = st["max_logit"], st["max_logit_p99"]
            print(f"attention: mean over rows of the largest logit, per layer: min {min(ml):.1f} median {sorted(ml)[len(ml) // 2]:.1f} max {max(ml):.1f}; "
                  f"p99 row maximum up to {max(p99):.1f}")
            print(f"GELU input: |x| max {max(st['gelu_in_absmax']):.1f}, fraction |x| > 6: {sum(st['gelu_in_frac_gt6']) / len(st['gelu_in_frac_gt6']):.4f}")
            ss = stream_stats(w, torch.cat([im1, im2]), cfg)
            print("encoder residual stream (median |x|, outlier ratio max/median over channels) after blocks 0, 1, mid, last:",
                  [f"{m:.2f} / {r:.0f}x" for m, r in (ss[0], ss[min(1, len(ss) - 1)], ss[len(ss) // 2], ss[-1])])
            for v in range(2):
                d = ref[v]["pts3d"].norm(dim=-1)
                print(f"view {v}: |pts3d| median {float(d.median()):.3g} max {float(d.max()):.3g}; conf median {float(ref[v]['conf'].median()):.3g}; "
                      f"desc_conf median {float(ref[v]['desc_conf'].median()):.3g}")
        for name, rnd in MODES.items():
            with Emul(rnd):
                out = run()
            for v in range(2):
                print(f"{name:10s} view {v + 1}: " + "  ".join(f"{k} {rel(out[v][k], ref[v][k]):.2e}" for k in ("pts3d", "conf", "desc", "desc_conf")), flush=True)
        if a.stages:
            for stg in ("enc", "dec", "attn_enc", "attn_dec", "dpt", "feat"):
                with Emul({stg: RB}):
                    out = run()
                print(f"bf16 in {stg:9s} only: pts3d " + " / ".join(f"{rel(out[v]['pts3d'], ref[v]['pts3d']):.2e}" for v in range(2))
                      + "  desc " + " / ".join(f"{rel(out[v]['desc'], ref[v]['desc']):.2e}" for v in range(2)), flush=True)


if __name__ == "__main__":
    main()
