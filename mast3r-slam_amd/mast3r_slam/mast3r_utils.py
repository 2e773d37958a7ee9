"""Level-3 operator API on the MI355X: the functions the SLAM loop injects as callables.

Mirror of /root/reference/src/mlx_mast3r_slam/mast3r_utils.py (same names, argument meaning and
return tuples; `__all__` :800-823): load_mast3r :47, frame_to_numpy :210, downsample :234,
mast3r_inference_mono :255, mast3r_asymmetric_inference :324, mast3r_symmetric_inference :382,
mast3r_match_asymmetric :451, mast3r_match_symmetric :503, mast3r_decode_symmetric_batch :572.
Tensors are torch tensors on the ROCm device instead of mx.array.

Differences, on purpose:
  * mast3r_match_symmetric / mast3r_decode_symmetric_batch are REAL (decode from cached encoder
    tokens, both directions, batched over B pairs); the reference bodies are stubs that return
    identity matches / zeros (:556-569, :611-616).
  * cached `frame.feat` is reused; the reference re-encodes both images on every call (:345-355).
  * *_batch variants take P pairs at once (the data-parallel unit that shards across GPUs).
resize_img (:132-207) is the same host-side PIL preprocessing; the retrieval database (:640-795) is
outside the hot path and not provided.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .config import get_config
from .model import Mast3rFull
from . import matching

__all__ = [
    "load_mast3r", "resize_img", "frame_to_numpy", "downsample", "mast3r_inference_mono", "mast3r_asymmetric_inference",
    "mast3r_symmetric_inference", "mast3r_match_asymmetric", "mast3r_match_symmetric",
    "mast3r_decode_symmetric_batch", "mast3r_match_asymmetric_batch",
]


def load_mast3r(model_type: str = "dunemast3r", variant: str = "base", resolution: int = 336,
                precision: str = "fp16", weights_path: Optional[str] = None, **kw) -> Mast3rFull:
    """mast3r_utils.py:47-80, with the reference's own defaults ("dunemast3r", "base", 336, "fp16") - a caller that
    relies on them gets the reference's behaviour or a loud error, never a different model.  Only the full ViT-L
    model the hot path names ("mast3r_full") is provided: the DUNE default raises NotImplementedError (out of scope,
    DESIGN.md section 7), so the call a SLAM loop makes here is `load_mast3r("mast3r_full", resolution=512, ...)`.
    `precision`: "fp16" | "fp32" | "bf16" as the reference (see model.py for what each selects on the MI355X);
    `weights_path` (extension): a torch / safetensors state dict with the public MASt3R key names."""
    if model_type == "mast3r_full":
        return Mast3rFull.from_pretrained(resolution=resolution, precision=precision, weights_path=weights_path, **kw)
    if model_type == "dunemast3r":
        raise NotImplementedError("model_type 'dunemast3r' (DUNE encoder + MASt3R decoder, the reference's default) is outside "
                                  "this build's scope; pass model_type='mast3r_full' (ViT-L encoder, resolution=512)")
    raise ValueError(f"Unknown model type: {model_type}. Use 'dunemast3r' or 'mast3r_full'")


def resize_img(img, size: int, square_ok: bool = False, return_transformation: bool = False):
    """mast3r_utils.py:132-207 (host side, PIL - as the reference): uint8 or float[0,1] [H,W,3] ->
    dict(img float32 [1,H',W',3] in [-1,1], true_shape int32 [[H',W']], unnormalized_img uint8 [H',W',3]).
    size 512: long edge -> 512 (LANCZOS when shrinking, BICUBIC otherwise), centre crop to multiples of
    16, a square result becomes 4:3 unless square_ok; size 224: short edge -> 224, centre square crop."""
    from PIL import Image
    a = img.cpu().numpy() if isinstance(img, torch.Tensor) else np.asarray(img)
    if a.dtype in (np.float32, np.float64):
        a = (a * 255).astype(np.uint8) if a.max() <= 1.0 else a.astype(np.uint8)
    pil = Image.fromarray(a)
    w1, h1 = pil.size

    def to_long_edge(im, long_edge):
        s = max(im.size)
        kind = Image.LANCZOS if s > long_edge else Image.BICUBIC
        return im.resize(tuple(int(round(x * long_edge / s)) for x in im.size), kind)

    pil = to_long_edge(pil, round(size * max(w1 / h1, h1 / w1)) if size == 224 else size)
    w, h = pil.size
    cx, cy = w // 2, h // 2
    if size == 224:
        half = min(cx, cy)
        box = (cx - half, cy - half, cx + half, cy + half)
    else:
        halfw, halfh = ((2 * cx) // 16) * 8, ((2 * cy) // 16) * 8
        if not square_ok and w == h:
            halfh = int(3 * halfw / 4)
        box = (cx - halfw, cy - halfh, cx + halfw, cy + halfh)
    pil = pil.crop(box)
    raw = np.asarray(pil)
    res = {"img": torch.from_numpy(((raw.astype(np.float32) / 255.0 - 0.5) / 0.5)[None]),
           "true_shape": torch.tensor([[pil.size[1], pil.size[0]]], dtype=torch.int32),
           "unnormalized_img": raw}
    if return_transformation:
        return res, (w1 / w, h1 / h, (w - pil.size[0]) / 2, (h - pil.size[1]) / 2)
    return res


def frame_to_numpy(frame) -> torch.Tensor:
    """mast3r_utils.py:210-226: frame image -> uint8 [H,W,3] (a device tensor here)."""
    img = frame.img
    if isinstance(img, np.ndarray):
        img = torch.from_numpy(img)
    if img.dim() == 3 and img.shape[0] == 3:
        img = img.permute(1, 2, 0)
    if img.dtype != torch.uint8:
        img = (img * 255).to(torch.uint8) if float(img.max()) <= 1.0 else img.to(torch.uint8)
    return img.contiguous()


def downsample(X, C, D, Q):
    """mast3r_utils.py:234-252."""
    k = get_config().get("dataset", {}).get("img_downsample", 1)
    if k > 1:
        X = X[..., ::k, ::k, :].contiguous()
        C = C[..., ::k, ::k].contiguous()
        D = D[..., ::k, ::k, :].contiguous()
        Q = Q[..., ::k, ::k].contiguous()
    return X, C, D, Q


def _feat(model: Mast3rFull, frame) -> torch.Tensor:
    if frame.feat is None:
        frame.feat = model.encode(frame_to_numpy(frame).to(model.device))
    return frame.feat


def _grid(frame):
    img = frame.img
    h, w = (img.shape[1], img.shape[2]) if img.shape[0] == 3 else (img.shape[0], img.shape[1])
    return h // 16, w // 16


def _stack(outs):
    X = torch.cat([o["pts3d"] for o in outs], 0)
    C = torch.cat([o["conf"] for o in outs], 0)
    D = torch.cat([o["desc"] for o in outs], 0)
    Q = torch.cat([o["desc_conf"] for o in outs], 0)
    return X, C, D, Q


def mast3r_inference_mono(model: Mast3rFull, frame):
    """mast3r_utils.py:255-321 -> (Xii [H*W,3], Cii [H*W,1], feat [T,1024], pos [T,2] (x,y))."""
    f = _feat(model, frame)
    gh, gw = _grid(frame)
    o1, o2 = model.decode_heads(f, f, 1, (gh, gw))
    X, C, D, Q = downsample(*_stack([o1, o2]))
    h, w = X.shape[1:3]
    gy, gx = torch.meshgrid(torch.arange(gh), torch.arange(gw), indexing="ij")
    pos = torch.stack([gx.reshape(-1), gy.reshape(-1)], dim=-1).to(X.device)
    return X[0].reshape(h * w, 3), C[0].reshape(h * w, 1), frame.feat, pos


def mast3r_asymmetric_inference(model: Mast3rFull, frame_i, frame_j):
    """mast3r_utils.py:324-379 -> X [2,H,W,3], C [2,H,W], D [2,H,W,24], Q [2,H,W]."""
    fi, fj = _feat(model, frame_i), _feat(model, frame_j)
    o_i, o_j = model.decode_heads(fi, fj, 1, _grid(frame_i))
    return downsample(*_stack([o_i, o_j]))


def mast3r_symmetric_inference(model: Mast3rFull, frame_i, frame_j):
    """mast3r_utils.py:382-443 -> [4,...] in the order (ii, ji, jj, ij); both directions are decoded
    as one batch of two pairs."""
    fi, fj = _feat(model, frame_i), _feat(model, frame_j)
    o1, o2 = model.decode_heads(torch.stack([fi, fj]), torch.stack([fj, fi]), 2, _grid(frame_i))
    # pair 0 = (i,j): o1[0]=ii, o2[0]=ji ; pair 1 = (j,i): o1[1]=jj, o2[1]=ij
    pick = lambda o, k: {n: v[k:k + 1] for n, v in o.items()}
    return downsample(*_stack([pick(o1, 0), pick(o2, 0), pick(o1, 1), pick(o2, 1)]))


def mast3r_match_asymmetric(model: Mast3rFull, frame_i, frame_j, idx_i2j_init=None):
    """mast3r_utils.py:451-500 -> (idx_i2j [1,N], valid_match_j [1,N,1], Xii, Cii, Qii, Xji, Cji, Qji)."""
    X, C, D, Q = mast3r_asymmetric_inference(model, frame_i, frame_j)
    h, w = X.shape[1:3]
    idx, valid = matching.match(X[0:1], X[1:2], D[0:1], D[1:2], idx_1_to_2_init=idx_i2j_init)
    n = h * w
    return (idx, valid, X[0:1].reshape(1, n, 3), C[0:1].reshape(1, n, 1), Q[0:1].reshape(1, n, 1),
            X[1:2].reshape(1, n, 3), C[1:2].reshape(1, n, 1), Q[1:2].reshape(1, n, 1))


def mast3r_match_asymmetric_batch(model: Mast3rFull, imgs_i, imgs_j, idx_i2j_init=None):
    """P pairs at once from raw uint8 images [P,H,W,3]: encode + decode + heads + match.
    Returns (idx [P,N], valid [P,N,1], X [2,P,H,W,3], C [2,P,H,W], D [2,P,H,W,24], Q [2,P,H,W])."""
    o1, o2 = model.reconstruct_batch(imgs_i, imgs_j)
    idx, valid = matching.match(o1["pts3d"], o2["pts3d"], o1["desc"], o2["desc"], idx_1_to_2_init=idx_i2j_init)
    st = lambda k: torch.stack([o1[k], o2[k]])
    return idx, valid, st("pts3d"), st("conf"), st("desc"), st("desc_conf")


def _hw(shape):
    s = shape[0]
    s = s.cpu() if isinstance(s, torch.Tensor) else np.asarray(s)
    return int(s.reshape(-1)[0]), int(s.reshape(-1)[1])


def _decode_symmetric(model: Mast3rFull, feat_i, feat_j, shape_i):
    """Both decode directions of B pairs as ONE batch of 2B: items [0, B) are the pairs (i, j), items [B, 2B) the pairs
    (j, i).  Returns the two branch outputs: o1[:B] = ii, o1[B:] = jj (view-1 branch), o2[:B] = ji, o2[B:] = ij."""
    b = feat_i.shape[0]
    h, w = _hw(shape_i)
    grid = (h // 16, w // 16)
    fi = feat_i.to(model.device, model.tdt)          # cached tokens are stored in the trunk's 16-bit type
    fj = feat_j.to(model.device, model.tdt)
    o1, o2 = model.decode_heads(torch.cat([fi, fj], 0), torch.cat([fj, fi], 0), 2 * b, grid)
    return o1, o2, b


def mast3r_decode_symmetric_batch(model: Mast3rFull, feat_i, pos_i, feat_j, pos_j, shape_i, shape_j):
    """mast3r_utils.py:572-632 -> X [4,B,H,W,3], C [4,B,H,W], D [4,B,H,W,24], Q [4,B,H,W] in the order
    (ii, ji, jj, ij).  feat_* [B,T,1024] cached encoder tokens."""
    o1, o2, b = _decode_symmetric(model, feat_i, feat_j, shape_i)
    sl = lambda o, lo: {n: v[lo:lo + b] for n, v in o.items()}
    parts = [sl(o1, 0), sl(o2, 0), sl(o1, b), sl(o2, b)]           # ii, ji, jj, ij
    X = torch.stack([p["pts3d"] for p in parts])
    C = torch.stack([p["conf"] for p in parts])
    D = torch.stack([p["desc"] for p in parts])
    Q = torch.stack([p["desc_conf"] for p in parts])
    return downsample(X, C, D, Q)


def mast3r_match_symmetric(model: Mast3rFull, feat_i, pos_i, feat_j, pos_j, shape_i, shape_j):
    """mast3r_utils.py:503-569 -> (idx_i2j, idx_j2i [B,N], valid_match_j, valid_match_i [B,N,1],
    Qii, Qjj, Qji, Qij [B,N,1]).  Both matching directions run as one batch of 2B maps."""
    if get_config().get("dataset", {}).get("img_downsample", 1) <= 1:
        # match(X11, X21): i->j uses (ii, ji), j->i uses (jj, ij) - in the 2B batch that IS the view-1 branch against the
        # view-2 branch, so the decoder outputs go to the matcher as they are (the [4,B,...] stack of
        # mast3r_decode_symmetric_batch and its re-concatenation copied 1.6 GB per 8 edges at 512x512)
        o1, o2, b = _decode_symmetric(model, feat_i, feat_j, shape_i)
        n = o1["pts3d"].shape[1] * o1["pts3d"].shape[2]
        idx, valid = matching.match(o1["pts3d"], o2["pts3d"], o1["desc"], o2["desc"])
        q1, q2 = o1["desc_conf"].reshape(2 * b, n, 1), o2["desc_conf"].reshape(2 * b, n, 1)
        return idx[:b], idx[b:], valid[:b], valid[b:], q1[:b], q1[b:], q2[:b], q2[b:]
    X, C, D, Q = mast3r_decode_symmetric_batch(model, feat_i, pos_i, feat_j, pos_j, shape_i, shape_j)
    b, h, w = X.shape[1:4]
    n = h * w
    X11 = torch.cat([X[0], X[2]], 0)
    X21 = torch.cat([X[1], X[3]], 0)
    D11 = torch.cat([D[0], D[2]], 0)
    D21 = torch.cat([D[1], D[3]], 0)
    idx, valid = matching.match(X11, X21, D11, D21)
    q = lambda k: Q[k].reshape(b, n, 1)
    return idx[:b], idx[b:], valid[:b], valid[b:], q(0), q(2), q(1), q(3)
