"""CPU oracle (torch fp32) for the two-view network.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED against the reference: /root/reference contains no source, weights or
tests for this arithmetic - it lives in the un-vendored submodule thirdparty/mlx-mast3r
(.gitmodules:10-12, no pinned commit) and is reached only through
model.encode / model.reconstruct (mast3r_utils.py:278,281,347-355).  The architecture below
is the PUBLIC MASt3R definition (checkpoint MASt3R_ViTLarge_BaseDecoder_512_catmlp_dpt_metric;
DUSt3R / CroCo-v2 code structure), restated from its published description:

  encoder   ViT-L/16: PatchEmbed conv16/16 -> 24 x [LN, MHA(16 x 64, RoPE-2D base 100), LN, MLP 4096 GELU]
            -> LN (eps 1e-6 everywhere)
  decoder   Linear 1024->768, then 12 x two parallel blocks (one per view):
            x += SA(LN1 x); x += CA(LN2 x, LNy y_other) ; x += MLP(LN3 x)   (12 heads x 64, MLP 3072)
            -> LN on the last output
  DPT head  taps {encoder out, dec 6, dec 9, dec 12}: 1x1 conv (+ convT x4 / convT x2 / id / conv3 s2)
            -> 3x3 to 256 -> 4 RefineNet fusion blocks (2 residual conv units, x2 bilinear
            align_corners, 1x1) -> conv3 256->128, x2, conv3 128->128, ReLU, conv1 128->4
            pts3d = xyz/|xyz| * expm1(|xyz|), conf = 1 + exp(c)
  features  MLP(cat(enc 1024, dec 768) -> 7168 GELU -> 25*16*16), pixel-shuffle 16,
            desc = first 24 channels L2-normalised, desc_conf = exp(last)

What the reference DOES pin and this file honours: output keys and shapes of `reconstruct`
(mast3r_utils.py:284-294: pts3d [H,W,3], conf [H,W,1], desc [H,W,24], desc_conf), patch 16,
backbone width 1024 (mast3r_utils.py:104-109, frame.py:157-158).

Weights: a dict name -> fp32 CPU tensor using the public checkpoint's key names.  Matrix
weights are expected to be bf16-representable already (the product stores them in bf16), so
oracle-vs-HIP differences come from activation rounding and accumulation order only.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

EPS = 1e-6


def _ln(x, w, p):
    return F.layer_norm(x, (x.shape[-1],), w[p + ".weight"], w[p + ".bias"], EPS)


def _lin(x, w, p):
    return F.linear(x, w[p + ".weight"], w[p + ".bias"])


def rope_tables(max_pos: int, base: float = 100.0):
    """cos/sin [max_pos,16] for the 16 frequencies base^(-i/16)."""
    inv = 1.0 / (base ** (torch.arange(0, 32, 2, dtype=torch.float32) / 32.0))
    ang = torch.arange(max_pos, dtype=torch.float32)[:, None] * inv[None, :]
    return ang.cos(), ang.sin()


def rope2d(x, pos_yx, cos, sin):
    """x [B,h,T,64]; pos_yx [T,2] (y,x).  First 32 dims rotate with y, last 32 with x."""
    out = []
    for blk in range(2):
        v = x[..., blk * 32:(blk + 1) * 32]
        c = cos[pos_yx[:, blk]][None, None]          # [1,1,T,16]
        s = sin[pos_yx[:, blk]][None, None]
        x1, x2 = v[..., :16], v[..., 16:]
        out += [x1 * c - x2 * s, x2 * c + x1 * s]
    return torch.cat(out, dim=-1)


def _mha(q, k, v, heads):
    b, tq, c = q.shape
    tk = k.shape[1]
    d = c // heads
    q = q.view(b, tq, heads, d).transpose(1, 2)
    k = k.view(b, tk, heads, d).transpose(1, 2)
    v = v.view(b, tk, heads, d).transpose(1, 2)
    return q, k, v


def self_attn(x, w, p, heads, pos, cos, sin):
    b, t, c = x.shape
    qkv = _lin(x, w, p + ".qkv")
    q, k, v = _mha(qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:], heads)
    q, k = rope2d(q, pos, cos, sin), rope2d(k, pos, cos, sin)
    a = torch.softmax(q @ k.transpose(-1, -2) * (1.0 / math.sqrt(c // heads)), dim=-1)
    return _lin((a @ v).transpose(1, 2).reshape(b, t, c), w, p + ".proj")


def cross_attn(x, y, w, p, heads, pos_x, pos_y, cos, sin):
    b, t, c = x.shape
    q, k, v = _mha(_lin(x, w, p + ".projq"), _lin(y, w, p + ".projk"), _lin(y, w, p + ".projv"), heads)
    q, k = rope2d(q, pos_x, cos, sin), rope2d(k, pos_y, cos, sin)
    a = torch.softmax(q @ k.transpose(-1, -2) * (1.0 / math.sqrt(c // heads)), dim=-1)
    return _lin((a @ v).transpose(1, 2).reshape(b, t, c), w, p + ".proj")


def mlp(x, w, p):
    return _lin(F.gelu(_lin(x, w, p + ".fc1")), w, p + ".fc2")


def patch_positions(h, w):
    gy, gx = torch.meshgrid(torch.arange(h // 16), torch.arange(w // 16), indexing="ij")
    return torch.stack([gy.reshape(-1), gx.reshape(-1)], dim=-1)          # [T,2] (y,x)


def normalize_image(img_u8):
    """uint8 [B,H,W,3] -> float [B,3,H,W] in [-1,1] (resize_img, mast3r_utils.py:186-188)."""
    return ((img_u8.float() / 255.0 - 0.5) / 0.5).permute(0, 3, 1, 2)


def encode(w, img_u8, cfg):
    """-> tokens [B,T,1024] (after enc_norm), pos [T,2]."""
    x = normalize_image(img_u8)
    b, _, h, wd = x.shape
    pos = patch_positions(h, wd)
    cos, sin = rope_tables(max(h, wd) // 16 + 1)
    x = F.conv2d(x, w["patch_embed.proj.weight"], w["patch_embed.proj.bias"], stride=16)
    x = x.flatten(2).transpose(1, 2)
    for i in range(cfg["enc_depth"]):
        p = f"enc_blocks.{i}"
        x = x + self_attn(_ln(x, w, p + ".norm1"), w, p + ".attn", cfg["enc_heads"], pos, cos, sin)
        x = x + mlp(_ln(x, w, p + ".norm2"), w, p + ".mlp")
    return _ln(x, w, "enc_norm"), pos


def decode(w, f1, f2, pos, cfg):
    """-> two lists of 1 + dec_depth token tensors (encoder output first, last one normed)."""
    cos, sin = rope_tables(int(pos.max()) + 2)
    o1, o2 = [f1], [f2]
    f1, f2 = _lin(f1, w, "decoder_embed"), _lin(f2, w, "decoder_embed")
    h = cfg["dec_heads"]
    for i in range(cfg["dec_depth"]):
        new = []
        for p, x, y in ((f"dec_blocks.{i}", f1, f2), (f"dec_blocks2.{i}", f2, f1)):
            x = x + self_attn(_ln(x, w, p + ".norm1"), w, p + ".attn", h, pos, cos, sin)
            x = x + cross_attn(_ln(x, w, p + ".norm2"), _ln(y, w, p + ".norm_y"), w, p + ".cross_attn", h,
                               pos, pos, cos, sin)
            x = x + mlp(_ln(x, w, p + ".norm3"), w, p + ".mlp")
            new.append(x)
        f1, f2 = new
        o1.append(f1)
        o2.append(f2)
    o1[-1], o2[-1] = _ln(o1[-1], w, "dec_norm"), _ln(o2[-1], w, "dec_norm")
    return o1, o2


def _conv(x, w, p, stride=1, padding=0):
    return F.conv2d(x, w[p + ".weight"], w.get(p + ".bias"), stride=stride, padding=padding)


def _rcu(x, w, p):
    out = _conv(F.relu(x), w, p + ".conv1", padding=1)
    out = _conv(F.relu(out), w, p + ".conv2", padding=1)
    return out + x


def _fusion(w, p, x0, x1=None):
    out = x0
    if x1 is not None:
        out = out + _rcu(x1, w, p + ".resConfUnit1")
    out = _rcu(out, w, p + ".resConfUnit2")
    out = F.interpolate(out, scale_factor=2, mode="bilinear", align_corners=True)
    return _conv(out, w, p + ".out_conv")


def dpt_head(w, p, toks, gh, gw, hooks=(0, 6, 9, 12)):
    """toks: list of 1 + dec_depth token tensors; -> raw [B,4,H,W]."""
    layers = [toks[k].transpose(1, 2).reshape(toks[k].shape[0], -1, gh, gw) for k in hooks]
    l0 = F.conv_transpose2d(_conv(layers[0], w, p + ".act_postprocess.0.0"), w[p + ".act_postprocess.0.1.weight"],
                            w[p + ".act_postprocess.0.1.bias"], stride=4)
    l1 = F.conv_transpose2d(_conv(layers[1], w, p + ".act_postprocess.1.0"), w[p + ".act_postprocess.1.1.weight"],
                            w[p + ".act_postprocess.1.1.bias"], stride=2)
    l2 = _conv(layers[2], w, p + ".act_postprocess.2.0")
    l3 = _conv(_conv(layers[3], w, p + ".act_postprocess.3.0"), w, p + ".act_postprocess.3.1", stride=2, padding=1)
    ls = [_conv(l, w, p + f".scratch.layer_rn.{i}", padding=1) for i, l in enumerate((l0, l1, l2, l3))]
    path4 = _fusion(w, p + ".scratch.refinenet4", ls[3])[:, :, :ls[2].shape[2], :ls[2].shape[3]]
    path3 = _fusion(w, p + ".scratch.refinenet3", path4, ls[2])
    path2 = _fusion(w, p + ".scratch.refinenet2", path3, ls[1])
    path1 = _fusion(w, p + ".scratch.refinenet1", path2, ls[0])
    out = _conv(path1, w, p + ".head.0", padding=1)
    out = F.interpolate(out, scale_factor=2, mode="bilinear", align_corners=True)
    out = F.relu(_conv(out, w, p + ".head.2", padding=1))
    return _conv(out, w, p + ".head.4")


def head(w, p, toks, h, wd, hooks=(0, 6, 9, 12)):
    """-> dict pts3d [B,H,W,3], conf [B,H,W], desc [B,H,W,24], desc_conf [B,H,W]."""
    gh, gw = h // 16, wd // 16
    raw = dpt_head(w, p + ".dpt", toks, gh, gw, hooks).permute(0, 2, 3, 1)
    xyz = raw[..., :3]
    d = xyz.norm(dim=-1, keepdim=True)
    pts = xyz / d.clip(min=1e-8) * torch.expm1(d)
    conf = 1.0 + raw[..., 3].exp()
    cat = torch.cat([toks[0], toks[-1]], dim=-1)
    f = _lin(F.gelu(_lin(cat, w, p + ".head_local_features.fc1")), w, p + ".head_local_features.fc2")
    b = f.shape[0]
    f = F.pixel_shuffle(f.transpose(1, 2).reshape(b, -1, gh, gw), 16).permute(0, 2, 3, 1)     # [B,H,W,25]
    desc = f[..., :24]
    desc = desc / desc.norm(dim=-1, keepdim=True).clip(min=1e-12)
    return dict(pts3d=pts, conf=conf, desc=desc, desc_conf=f[..., 24].exp())


def reconstruct(w, img1_u8, img2_u8, cfg):
    """model.reconstruct(img1, img2) (mast3r_utils.py:355): both results in view-1's frame.
    img*_u8 uint8 [B,H,W,3].  Returns (out1, out2) dicts of fp32 tensors."""
    b, h, wd, _ = img1_u8.shape
    f, pos = encode(w, torch.cat([img1_u8, img2_u8], 0), cfg)
    o1, o2 = decode(w, f[:b], f[b:], pos, cfg)
    hooks = tuple(cfg.get("hooks", (0, 6, 9, 12)))
    return head(w, "downstream_head1", o1, h, wd, hooks), head(w, "downstream_head2", o2, h, wd, hooks)
