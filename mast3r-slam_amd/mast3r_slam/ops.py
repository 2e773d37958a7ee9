"""Thin torch-tensor wrappers over the network operators of the C ABI (include/m3slam_model.h).

16-bit activations / weights (torch.bfloat16 or torch.float16: the `dtype` argument of the *_dt entry
points follows the tensors), fp32 accumulation.  All ops are stream-ordered; outputs are
allocated by the caller-facing wrapper (torch caching allocator) and passed as raw pointers.
"""
from __future__ import annotations

import os

import ctypes as C

import torch
from torch.utils.weak import WeakTensorKeyDictionary

from . import _ffi

EPI_BF16, EPI_BF16_GELU, EPI_F32, EPI_F32_ACCUM, EPI_BF16_RELU, EPI_BF16_ADD, EPI_BF16_ROPE = range(7)
EPI_INPUT_RELU = 0x100                          # flag (conv3x3*): the convolution reads relu(x) without a relu(x) tensor
_F32_EPIS = (EPI_F32, EPI_F32_ACCUM)

_zero16 = {}

H16 = (torch.bfloat16, torch.float16)          # the two 16-bit storage types of the network operators
DT_CODE = {torch.bfloat16: 0, torch.float16: 1}   # M3_DT_BF16 / M3_DT_F16 (include/m3slam_model.h)
DT_F16_PVBF16 = 2                              # M3_DT_F16_PVBF16: fp16 q / k / o, bf16 v and probabilities (RoPE GEMMs, attention)


def _pv_code(dt: int, pv_bf16: bool) -> int:
    if not pv_bf16:
        return dt
    if dt != DT_CODE[torch.float16]:
        raise TypeError("pv_bf16 (M3_DT_F16_PVBF16) is the attention form of the fp16 trunk: tensors must be torch.float16")
    return DT_F16_PVBF16


def _same16(a, *others):
    """dtype code of `a`; every other (non-None) 16-bit tensor must have the same type."""
    for t in others:
        if t is not None and t.dtype != a.dtype:
            raise TypeError(f"mixed 16-bit types in one launch: {a.dtype} vs {t.dtype}")
    return DT_CODE[a.dtype]

# When set to a list, every MFMA GEMM / implicit-GEMM conv launch appends
# (kind, algorithmic_flops, start_event, end_event, algorithmic_bytes): events are recorded on the
# current stream, i.e. the stream the kernel is launched on (bench.py's roofline measurement);
# algorithmic bytes = operands read once + output written once (+ residual read once).
PROFILE = None


def _prof_begin():
    if PROFILE is None:
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _gemm_kind(m, n, groups=1):
    """Profile tag naming the kernel family the dispatcher picks (k_gemm256 = 256-row ping-pong kernel)."""
    if PROFILE is None:
        return "gemm"
    return "gemm256" if int(_ffi.lib().m3_gemm_pick_tile(m, n, groups)) >= 192 else "gemm_small"


PROFILE_SHAPES = None      # optional list: (kind, description) per profiled launch, same order as PROFILE


def _prof_end(e0, kind, flops, nbytes=0.0, desc=""):
    if e0 is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        PROFILE.append((kind, flops, e0, e1, nbytes))
        if PROFILE_SHAPES is not None:
            PROFILE_SHAPES.append((kind, desc))


def zero_page(dev):
    """16 zero bytes on `dev` (padding source of the implicit-GEMM convolution)."""
    z = _zero16.get(dev)
    if z is None:
        z = torch.zeros(64, dtype=torch.uint8, device=dev)
        _zero16[dev] = z
    return z


def gemm(a: torch.Tensor, w: torch.Tensor, bias=None, epi: int = EPI_BF16, out=None, resid=None):
    """out[M,N] = epi(a[M,K] @ w[N,K].T + bias).  a, w bf16; out bf16 or f32 by epilogue."""
    a = _ffi.check(a, H16, "a")
    w = _ffi.check(w, H16, "w")
    dt = _same16(a, w)
    m, k = a.shape
    n = w.shape[0]
    if w.shape[1] != k:
        raise ValueError(f"K mismatch: a {tuple(a.shape)} vs w {tuple(w.shape)}")
    odt = torch.float32 if epi in _F32_EPIS else a.dtype
    if out is None:
        out = torch.empty((m, n), dtype=odt, device=a.device)
    else:
        if out.dtype != odt or out.shape[0] != m or out.shape[1] < n or out.stride(1) != 1:
            raise ValueError("bad `out`")
    ldc = out.stride(0)
    if bias is not None:
        bias = _ffi.check(bias, torch.float32, "bias", (n,))
    if resid is not None and (resid.dtype != odt or resid.stride(0) != ldc):
        raise ValueError("residual must match out dtype/stride")
    e0 = _prof_begin()
    _ffi.call("m3_gemm_dt", _ffi.ptr(a), _ffi.ptr(w), _ffi.ptr(bias), _ffi.ptr(out), _ffi.ptr(resid),
              m, n, k, ldc, epi, dt, _ffi.stream_ptr())
    esz = out.element_size()
    _prof_end(e0, _gemm_kind(m, n), 2.0 * m * n * k, 2.0 * (m * k + n * k) + esz * m * n * (1 if resid is None else 2),
              f"gemm {m}x{n}x{k} epi{epi} {a.dtype}")
    return out


def rope_token_table(pos_yx, cos_sin):
    """pos_yx int [T,2] (y,x), cos_sin f32 [max_pos,16,2] -> per-token table f32 [T,2,2,16] = (token, axis y|x,
    cos|sin, frequency): what the fused RoPE epilogue reads, no position lookup on the device."""
    return cos_sin[pos_yx.long()].transpose(-1, -2).contiguous()


QK_PRESCALE = 0.125 * 1.4426950408889634     # softmax scale (head dim 64) * log2(e), folded into q by the RoPE epilogue


ROPE_BASE = 100.0                            # CroCo "RoPE100"


_ROPE_BOUND = WeakTensorKeyDictionary()       # keyed by tensor identity, dropped with the tensor


def rope_bound(pos_yx, bound: int):
    """Promise that every entry of the int32 position tensor `pos_yx` is in [0, bound): the RoPE epilogues given this tensor then
    build their cos / sin table once per workgroup in LDS (m3_gemm_desc.rope_max_pos; same values, ~12 us less per 16384-row
    projection).  Returns the tensor.  Without the promise (or bound > 64) the coefficients are computed per element."""
    if pos_yx.dtype != torch.int32 or bound < 1:
        raise ValueError("rope_bound: int32 positions and a positive bound")
    _ROPE_BOUND[pos_yx] = int(bound)
    return pos_yx


def _rope_pmax(t) -> int:
    if os.environ.get("M3_ROPE_TABLE", "1") == "0":            # A/B switch: per-element v_sin / v_cos (same results)
        return 0
    b = _ROPE_BOUND.get(t, 0) if isinstance(t, torch.Tensor) else 0
    return b if 0 < b <= 64 else 0


def _rope_table(t):
    """The `rope` operand of the fused epilogue: int32 [T,2] grid positions (y, x) - cos/sin computed in the kernel -
    or float32 [T,2,2,16] per-token cos/sin table (rope_token_table).  Returns (tensor, tokens_per_image, by_position)."""
    if isinstance(t, torch.Tensor) and t.dtype == torch.int32:
        t = _ffi.check(t, torch.int32, "rope positions", (None, 2))
        return t, t.shape[0], True
    t = _ffi.check(t, torch.float32, "rope_tok", (None, 2, 2, 16))
    return t, t.shape[0], False


def gemm_rope(a, w, bias, rope_tok, rope_cols: int, q_cols: int = 0, q_scale: float = 1.0, base: float = ROPE_BASE,
              pv_bf16: bool = False):
    """16-bit out[M,N] = a @ w.T + bias with RoPE-2D applied to the 64-wide heads in columns < rope_cols;
    rope_tok: int32 [tokens_per_image,2] token grid positions (frequencies base^(-i/16)) or the f32
    [tokens_per_image,2,2,16] table of rope_token_table.  pv_bf16 (fp16 tensors only): the columns >= rope_cols
    (v) of the fp16 output buffer hold bf16 values - what attention(..., pv_bf16=True) reads."""
    rope_tok, tokens_per_image, by_pos = _rope_table(rope_tok)
    a = _ffi.check(a, H16, "a")
    w = _ffi.check(w, H16, "w")
    dt = _pv_code(_same16(a, w), pv_bf16)
    m, k = a.shape
    n = w.shape[0]
    if by_pos and float(base) == ROPE_BASE:                   # the descriptor entry point carries the position bound (rope_bound)
        return gemm_ex(a, w, bias, EPI_BF16_ROPE, rope=(rope_tok, rope_cols, q_cols, q_scale), pv_bf16=pv_bf16)
    out = torch.empty((m, n), dtype=a.dtype, device=a.device)
    e0 = _prof_begin()
    if by_pos:
        _ffi.call("m3_gemm_rope_pos_dt", _ffi.ptr(a), _ffi.ptr(w), _ffi.ptr(bias), _ffi.ptr(out), m, n, k, n,
                  _ffi.ptr(rope_tok), tokens_per_image, float(base), rope_cols, int(q_cols), float(q_scale), dt, _ffi.stream_ptr())
    else:
        _ffi.call("m3_gemm_rope_dt", _ffi.ptr(a), _ffi.ptr(w), _ffi.ptr(bias), _ffi.ptr(out), m, n, k, n,
                  _ffi.ptr(rope_tok), tokens_per_image, rope_cols, int(q_cols), float(q_scale), dt, _ffi.stream_ptr())
    _prof_end(e0, _gemm_kind(m, n), 2.0 * m * n * k, 2.0 * (m * k + n * k + m * n))
    return out


def conv3x3(x: torch.Tensor, w: torch.Tensor, bias=None, epi: int = EPI_BF16, stride: int = 1, resid=None, out=None,
            relu_input: bool = False, direct=None):
    """x NHWC bf16 [B,H,W,Cin], w bf16 [Cout,3,3,Cin] -> NHWC [B,OH,OW,Cout], padding 1.  relu_input: conv(relu(x)).
    direct: None = the direct-convolution kernel when its grid fills the chip (conv3x3_direct_ok), True / False force
    one form - both return the same bits."""
    x = _ffi.check(x, H16, "x")
    w = _ffi.check(w, H16, "w")
    dt = _same16(x, w)
    b, h, wd, cin = x.shape
    cout = w.shape[0]
    if tuple(w.shape[1:]) != (3, 3, cin):
        raise ValueError(f"weight must be [Cout,3,3,{cin}], got {tuple(w.shape)}")
    oh, ow = (h + 2 - 3) // stride + 1, (wd + 2 - 3) // stride + 1
    odt = torch.float32 if epi in _F32_EPIS else x.dtype
    if bias is not None:
        bias = _ffi.check(bias, torch.float32, "bias", (cout,))
    if resid is not None:
        resid = _ffi.check(resid, odt, "resid", (b, oh, ow, cout))
    if out is None and direct is not False and epi in (EPI_BF16, EPI_BF16_RELU, EPI_BF16_ADD) and (direct or conv3x3_direct_ok(x, cout, stride)):
        return _conv3x3_direct(x, w, None, bias, None, epi, resid, relu_input)      # same bits as the implicit-GEMM form
    if out is None:
        out = torch.empty((b, oh, ow, cout), dtype=odt, device=x.device)
    ws_bytes = int(_ffi.lib().m3_conv3x3_splitk_bytes(b, h, wd, cin, cout, stride))
    # fp32 partial planes of the split-K path, from torch's caching allocator: stream-ordered, capture-safe
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device) if ws_bytes > 0 else None
    e0 = _prof_begin()
    _ffi.call("m3_conv3x3_dt", _ffi.ptr(x), _ffi.ptr(w), _ffi.ptr(bias), _ffi.ptr(out), _ffi.ptr(resid),
              _ffi.ptr(zero_page(x.device)), b, h, wd, cin, cout, stride, epi | (EPI_INPUT_RELU if relu_input else 0),
              _ffi.ptr(ws), ws_bytes, dt, _ffi.stream_ptr())
    _prof_end(e0, "conv3x3", 2.0 * b * oh * ow * cout * 9 * cin,
              2.0 * (b * h * wd * cin + cout * 9 * cin) + out.element_size() * b * oh * ow * cout * (1 if resid is None else 2),
              f"conv3x3 {b}x{h}x{wd} {cin}->{cout} s{stride} epi{epi} splitk_ws={ws_bytes}")
    return out


def conv3x3_relu_head4(x: torch.Tensor, w: torch.Tensor, bias, w4: torch.Tensor, b4: torch.Tensor):
    """Tail of the DPT head in one launch: relu(conv3x3(x, w) + bias) [128 ch, not materialised] -> 1x1
    projection w4 [4,128] + b4 -> (pts3d [B,H,W,3], conf [B,H,W]) f32 with the pointmap post-processing."""
    x = _ffi.check(x, H16, "x")
    b, h, wd, cin = x.shape
    w = _ffi.check(w, H16, "w", (128, 3, 3, cin))
    w4 = _ffi.check(w4, H16, "w4", (4, 128))
    dt = _same16(x, w, w4)
    b4 = _ffi.check(b4, torch.float32, "b4", (4,))
    if bias is not None:
        bias = _ffi.check(bias, torch.float32, "bias", (128,))
    pts = torch.empty((b, h, wd, 3), dtype=torch.float32, device=x.device)
    conf = torch.empty((b, h, wd), dtype=torch.float32, device=x.device)
    e0 = _prof_begin()
    _ffi.call("m3_conv3x3_relu_head4_dt", _ffi.ptr(x), _ffi.ptr(w), _ffi.ptr(bias), _ffi.ptr(w4), _ffi.ptr(b4),
              _ffi.ptr(pts), _ffi.ptr(conf), _ffi.ptr(zero_page(x.device)), b, h, wd, cin, dt, _ffi.stream_ptr())
    _prof_end(e0, "conv3x3", 2.0 * b * h * wd * 128 * (9 * cin + 4), 2.0 * (b * h * wd * cin + 128 * 9 * cin) + 16.0 * b * h * wd,
              f"conv3x3+head4 {b}x{h}x{wd} {cin}->128->4")
    return pts, conf


def conv3x3_grouped2(x, w0, w1, b0, b1, epi: int = EPI_BF16, stride: int = 1, resid=None, relu_input: bool = False, direct=None):
    """Two same-shape 3x3 convolutions in one launch (the two DPT heads): x NHWC [2,B,H,W,Cin], group g uses
    (w_g [Cout,3,3,Cin], b_g) -> [2,B,OH,OW,Cout]."""
    x = _ffi.check(x, H16, "x")
    if x.dim() != 5 or x.shape[0] != 2:
        raise ValueError(f"x must be [2,B,H,W,Cin], got {tuple(x.shape)}")
    _, b, h, wd, cin = x.shape
    w0 = _ffi.check(w0, H16, "w0", (None, 3, 3, cin))
    w1 = _ffi.check(w1, H16, "w1", tuple(w0.shape))
    dt = _same16(x, w0, w1)
    cout = w0.shape[0]
    oh, ow = (h + 2 - 3) // stride + 1, (wd + 2 - 3) // stride + 1
    odt = torch.float32 if epi in _F32_EPIS else x.dtype
    if resid is not None:
        resid = _ffi.check(resid, odt, "resid", (2, b, oh, ow, cout))
    if direct is not False and epi in (EPI_BF16, EPI_BF16_RELU, EPI_BF16_ADD) and (direct or conv3x3_direct_ok(x, cout, stride)):
        return _conv3x3_direct(x, w0, w1, b0, b1, epi, resid, relu_input)
    out = torch.empty((2, b, oh, ow, cout), dtype=odt, device=x.device)
    ws_bytes = 2 * int(_ffi.lib().m3_conv3x3_splitk_bytes(b, h, wd, cin, cout, stride))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device) if ws_bytes > 0 else None
    e0 = _prof_begin()
    _ffi.call("m3_conv3x3_grouped2_dt", _ffi.ptr(x), _ffi.ptr(w0), _ffi.ptr(w1), _ffi.ptr(b0), _ffi.ptr(b1), _ffi.ptr(out),
              _ffi.ptr(resid), _ffi.ptr(zero_page(x.device)), b, h, wd, cin, cout, stride,
              epi | (EPI_INPUT_RELU if relu_input else 0), _ffi.ptr(ws), ws_bytes, dt, _ffi.stream_ptr())
    _prof_end(e0, "conv3x3", 4.0 * b * oh * ow * cout * 9 * cin,
              2.0 * (2.0 * (b * h * wd * cin + cout * 9 * cin) + out.element_size() * b * oh * ow * cout * (1 if resid is None else 2)),
              f"conv3x3 x2 {b}x{h}x{wd} {cin}->{cout} s{stride} epi{epi} splitk_ws={ws_bytes}")
    return out


DIRECT_CONV_MIN_WGS = None     # direct convolution once its grid fills the chip (one workgroup per CU); same bits either way.
                               # None: the device's CU count (m3_device_cu_count, 256 on an MI355X); an int overrides it


def _direct_conv_min_wgs() -> int:
    return int(DIRECT_CONV_MIN_WGS) if DIRECT_CONV_MIN_WGS is not None else int(_ffi.lib().m3_device_cu_count())


def conv3x3_direct_ok(x, cout: int, stride: int = 1) -> bool:
    """Whether conv3x3 / conv3x3_grouped2 on x (NHWC [B,H,W,Cin] or [2,B,H,W,Cin]) takes the direct-convolution kernel."""
    g = 2 if x.dim() == 5 else 1
    b, h, w, cin = x.shape[-4:]
    if stride != 1 or cin not in (128, 256) or cout not in (128, 256) or h % 16 or w % 16 or os.environ.get("M3_DIRECT_CONV", "1") == "0":
        return False
    if int(_ffi.lib().m3_conv3x3_splitk_bytes(1, h, w, cin, cout, 1)) > 0:
        return False        # a geometry the implicit-GEMM form runs as split-K (partial planes: another summation order)
    return (h // 16) * ((w + 31) // 32) * b * (cout // 128) * g >= _direct_conv_min_wgs()


def _conv3x3_direct(x, w0, w1, b0, b1, epi, resid, relu_input):
    grouped = x.dim() == 5
    b, h, wd, cin = x.shape[-4:]
    cout = w0.shape[0]
    dt = _same16(x, w0, w1)
    out = torch.empty(tuple(x.shape[:-1]) + (cout,), dtype=x.dtype, device=x.device)
    e0 = _prof_begin()
    _ffi.call("m3_conv3x3_direct_grouped2_dt", _ffi.ptr(x), _ffi.ptr(w0), _ffi.ptr(w1), _ffi.ptr(b0), _ffi.ptr(b1), _ffi.ptr(out),
              _ffi.ptr(resid), _ffi.ptr(zero_page(x.device)), b, h, wd, cin, cout, epi | (EPI_INPUT_RELU if relu_input else 0), dt,
              _ffi.stream_ptr())
    g = 2 if grouped else 1
    _prof_end(e0, "conv_direct", 2.0 * g * b * h * wd * cout * 9 * cin,
              g * (2.0 * (b * h * wd * cin + cout * 9 * cin) + 2.0 * b * h * wd * cout * (1 if resid is None else 2)),
              f"conv3x3 direct{' x2' if grouped else ''} {b}x{h}x{wd} {cin}->{cout} epi{epi}")
    return out


def dpt_tail_grouped2(x, w0, w1, b0, b1, w40, w41, b40, b41, upsample: bool = True):
    """dpt_tail for both heads in one launch: x [2,B,h,w,128] -> (pts [2,B,H,W,3], conf [2,B,H,W])."""
    x = _ffi.check(x, H16, "x")
    if x.dim() != 5 or x.shape[0] != 2 or x.shape[-1] != 128:
        raise ValueError(f"x must be [2,B,h,w,128], got {tuple(x.shape)}")
    _, b, ih, iw, _ = x.shape
    ws = [_ffi.check(w, H16, "w", (128, 3, 3, 128)) for w in (w0, w1)]
    w4s = [_ffi.check(w, H16, "w4", (4, 128)) for w in (w40, w41)]
    dt = _same16(x, *ws, *w4s)
    h, wd = (2 * ih, 2 * iw) if upsample else (ih, iw)
    pts = torch.empty((2, b, h, wd, 3), dtype=torch.float32, device=x.device)
    conf = torch.empty((2, b, h, wd), dtype=torch.float32, device=x.device)
    e0 = _prof_begin()
    _ffi.call("m3_dpt_tail_grouped2_dt", _ffi.ptr(x), _ffi.ptr(ws[0]), _ffi.ptr(ws[1]), _ffi.ptr(b0), _ffi.ptr(b1),
              _ffi.ptr(w4s[0]), _ffi.ptr(w4s[1]), _ffi.ptr(b40), _ffi.ptr(b41), _ffi.ptr(pts), _ffi.ptr(conf),
              _ffi.ptr(zero_page(x.device)), b, h, wd, 1 if upsample else 0, dt, _ffi.stream_ptr())
    _prof_end(e0, "conv_tail", 4.0 * b * h * wd * 128 * (9 * 128 + 4), 2.0 * (2.0 * (b * ih * iw * 128 + 128 * 9 * 128) + 16.0 * b * h * wd),
              f"dpt_tail x2 {b}x{h}x{wd} 128->128->4 upsample={upsample}")
    return pts, conf


def conv3x3_up_direct(x, w, bias, upsample: bool = True):
    """Direct 3x3 convolution to 128 channels with the x2 bilinear (align_corners) upsample of its input fused in
    (DPT head.0): x NHWC [B,h,w,Cin] (Cin 256 or 128) -> [B,H,W,128], H = 2h with the upsample.  The upsampled map is
    never written."""
    x = _ffi.check(x, H16, "x")
    b, ih, iw, cin = x.shape
    w = _ffi.check(w, H16, "w", (128, 3, 3, cin))
    dt = _same16(x, w)
    if bias is not None:
        bias = _ffi.check(bias, torch.float32, "bias", (128,))
    h, wd = (2 * ih, 2 * iw) if upsample else (ih, iw)
    out = torch.empty((b, h, wd, 128), dtype=x.dtype, device=x.device)
    e0 = _prof_begin()
    _ffi.call("m3_conv3x3_up_direct_dt", _ffi.ptr(x), _ffi.ptr(w), _ffi.ptr(bias), _ffi.ptr(out), _ffi.ptr(zero_page(x.device)),
              b, h, wd, cin, 1 if upsample else 0, dt, _ffi.stream_ptr())
    _prof_end(e0, "conv_direct", 2.0 * b * h * wd * 128 * 9 * cin, 2.0 * (b * ih * iw * cin + 128 * 9 * cin + b * h * wd * 128),
              f"conv3x3_up_direct {b}x{h}x{wd} {cin}->128 upsample={upsample}")
    return out


def conv3x3_up_direct_grouped2(x, w0, w1, b0, b1, upsample: bool = True):
    """conv3x3_up_direct for both heads in one launch: x [2,B,h,w,Cin] -> [2,B,H,W,128]; head g uses (w_g, b_g)."""
    x = _ffi.check(x, H16, "x")
    if x.dim() != 5 or x.shape[0] != 2:
        raise ValueError(f"x must be [2,B,h,w,Cin], got {tuple(x.shape)}")
    _, b, ih, iw, cin = x.shape
    w0 = _ffi.check(w0, H16, "w0", (128, 3, 3, cin))
    w1 = _ffi.check(w1, H16, "w1", (128, 3, 3, cin))
    dt = _same16(x, w0, w1)
    h, wd = (2 * ih, 2 * iw) if upsample else (ih, iw)
    out = torch.empty((2, b, h, wd, 128), dtype=x.dtype, device=x.device)
    e0 = _prof_begin()
    _ffi.call("m3_conv3x3_up_direct_grouped2_dt", _ffi.ptr(x), _ffi.ptr(w0), _ffi.ptr(w1), _ffi.ptr(b0), _ffi.ptr(b1), _ffi.ptr(out),
              _ffi.ptr(zero_page(x.device)), b, h, wd, cin, 1 if upsample else 0, dt, _ffi.stream_ptr())
    _prof_end(e0, "conv_direct", 4.0 * b * h * wd * 128 * 9 * cin, 4.0 * (b * ih * iw * cin + 128 * 9 * cin + b * h * wd * 128),
              f"conv3x3_up_direct x2 {b}x{h}x{wd} {cin}->128 upsample={upsample}")
    return out


def dpt_tail(x: torch.Tensor, w: torch.Tensor, bias, w4: torch.Tensor, b4: torch.Tensor, upsample: bool = True):
    """DPT tail as one direct-convolution launch: [x2 bilinear upsample of x] -> conv3x3 128->128 + ReLU -> 1x1 -> 4
    -> (pts3d [B,H,W,3], conf [B,H,W]) f32.  x NHWC [B,H/2,W/2,128] (upsample) or [B,H,W,128]; 16-bit dtype."""
    x = _ffi.check(x, H16, "x")
    b, ih, iw, cin = x.shape
    if cin != 128:
        raise ValueError(f"dpt_tail needs 128 input channels, got {cin}")
    w = _ffi.check(w, H16, "w", (128, 3, 3, 128))
    w4 = _ffi.check(w4, H16, "w4", (4, 128))
    b4 = _ffi.check(b4, torch.float32, "b4", (4,))
    dt = _same16(x, w, w4)
    if bias is not None:
        bias = _ffi.check(bias, torch.float32, "bias", (128,))
    h, wd = (2 * ih, 2 * iw) if upsample else (ih, iw)
    pts = torch.empty((b, h, wd, 3), dtype=torch.float32, device=x.device)
    conf = torch.empty((b, h, wd), dtype=torch.float32, device=x.device)
    e0 = _prof_begin()
    _ffi.call("m3_dpt_tail_dt", _ffi.ptr(x), _ffi.ptr(w), _ffi.ptr(bias), _ffi.ptr(w4), _ffi.ptr(b4), _ffi.ptr(pts),
              _ffi.ptr(conf), _ffi.ptr(zero_page(x.device)), b, h, wd, 1 if upsample else 0, dt, _ffi.stream_ptr())
    _prof_end(e0, "conv_tail", 2.0 * b * h * wd * 128 * (9 * 128 + 4), 2.0 * (b * ih * iw * 128 + 128 * 9 * 128) + 16.0 * b * h * wd,
              f"dpt_tail {b}x{h}x{wd} 128->128->4 upsample={upsample}")
    return pts, conf


def attention(q, k, v, out, *, nbatch, heads, tq, tk, q_row_stride, kv_row_stride, o_row_stride,
              q_batch_stride, kv_batch_stride, o_batch_stride, kv_batch_shift=0, scale=0.125, prescaled=False,
              pv_bf16=False):
    """Fused MHA (head dim 64).  q/k/v/out are (views into) 16-bit device tensors; the strides are in
    elements, so q, k, v may be column slices of one projection buffer.  prescaled=True: q already carries
    scale * log2(e) (gemm_rope(..., q_cols, q_scale=QK_PRESCALE)); `scale` is then ignored.  pv_bf16=True (prescaled,
    fp16 tensors): v holds bf16 values (gemm_rope(..., pv_bf16=True)) and the probabilities are bf16 - the fast
    deferred-maximum loop with fp16 q / k."""
    for name, t in (("q", q), ("k", k), ("v", v), ("out", out)):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype in H16):
            raise TypeError(f"{name}: expected a bf16 / fp16 tensor on the ROCm device")
    if pv_bf16 and not prescaled:
        raise ValueError("pv_bf16 needs prescaled=True")
    dt = _pv_code(_same16(q, k, v, out), pv_bf16)
    e0 = _prof_begin()
    if prescaled:
        _ffi.call("m3_attention_prescaled_dt", q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), q_row_stride,
                  kv_row_stride, o_row_stride, q_batch_stride, kv_batch_stride, o_batch_stride, nbatch, heads, tq, tk,
                  kv_batch_shift, dt, _ffi.stream_ptr())
    else:
        _ffi.call("m3_attention_dt", q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), q_row_stride,
                  kv_row_stride, o_row_stride, q_batch_stride, kv_batch_stride, o_batch_stride, nbatch, heads, tq, tk,
                  kv_batch_shift, float(scale), dt, _ffi.stream_ptr())
    _prof_end(e0, "attention", 4.0 * nbatch * heads * tq * tk * 64, 2.0 * nbatch * heads * 64 * (2 * tq + 2 * tk))
    return out


def rope2d_(x, pos_yx, cos_sin, *, row_stride, tokens, heads, tokens_per_image):
    """In-place RoPE-2D on `heads` 64-wide heads starting at x.data_ptr()."""
    _ffi.call("m3_rope2d_dt", x.data_ptr(), _ffi.ptr(pos_yx), _ffi.ptr(cos_sin), row_stride, tokens, heads,
              tokens_per_image, DT_CODE[x.dtype], _ffi.stream_ptr())
    return x


def layernorm(x, gamma, beta, eps=1e-6, out=None, dtype=torch.bfloat16):
    x = _ffi.check(x, torch.float32, "x")
    m, c = x.shape
    if out is None:
        out = torch.empty((m, c), dtype=dtype, device=x.device)
    _ffi.call("m3_layernorm_dt", _ffi.ptr(x), _ffi.ptr(gamma), _ffi.ptr(beta), _ffi.ptr(out), m, c, float(eps),
              DT_CODE[out.dtype], _ffi.stream_ptr())
    return out


def patchify16(img_u8, dtype=torch.bfloat16, out=None):
    """uint8 [B,H,W,3] -> patch rows [B*T,768] (16-bit); `out`: write into this [B*T,768] slice of a larger token matrix
    (two image batches patchified into the halves of ONE matrix need no concatenated image tensor)."""
    img_u8 = _ffi.check(img_u8, torch.uint8, "img")
    if img_u8.data_ptr() % 8:                      # a uint8 view with an odd storage offset: the kernel reads 8-byte pieces
        img_u8 = img_u8.clone()                    # (the C entry point rejects a misaligned pointer with a status)
    b, h, w, _ = img_u8.shape
    rows = b * (h // 16) * (w // 16)
    if out is None:
        out = torch.empty((rows, 768), dtype=dtype, device=img_u8.device)
    elif out.dtype != dtype or tuple(out.shape) != (rows, 768) or not out.is_contiguous():
        raise ValueError("bad `out`")
    _ffi.call("m3_patchify16_dt", _ffi.ptr(img_u8), _ffi.ptr(out), b, h, w, DT_CODE[dtype], _ffi.stream_ptr())
    return out


def cast_f32(x, dtype=torch.bfloat16):
    """fp32 -> bf16 / fp16, round to nearest even."""
    x = _ffi.check(x, torch.float32, "x")
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    _ffi.call("m3_cast_f32_dt", _ffi.ptr(x), _ffi.ptr(out), x.numel(), DT_CODE[dtype], _ffi.stream_ptr())
    return out


def f32_to_bf16(x):
    return cast_f32(x, torch.bfloat16)


def cast16(x, dtype):
    """bf16 <-> fp16 (through fp32, round to nearest even; bf16 -> fp16 is exact for |x| in [6.1e-5, 65504])."""
    x = _ffi.check(x, H16, "x")
    if x.dtype == dtype:
        return x
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    _ffi.call("m3_cast16", _ffi.ptr(x), _ffi.ptr(out), x.numel(), DT_CODE[x.dtype], DT_CODE[dtype], _ffi.stream_ptr())
    return out


def relu(x):
    x = _ffi.check(x, H16, "x")
    out = torch.empty_like(x)
    _ffi.call("m3_relu_bf16", _ffi.ptr(x), _ffi.ptr(out), x.numel(), _ffi.stream_ptr())
    return out


def concat2(a, b):
    a = _ffi.check(a, H16, "a")
    b = _ffi.check(b, H16, "b")
    _same16(a, b)
    m = a.shape[0]
    out = torch.empty((m, a.shape[1] + b.shape[1]), dtype=a.dtype, device=a.device)
    _ffi.call("m3_concat2_bf16", _ffi.ptr(a), _ffi.ptr(b), _ffi.ptr(out), m, a.shape[1], b.shape[1],
              _ffi.stream_ptr())
    return out


def unshuffle(x, b, h, w, s, c, cpad=None):
    """[B*h*w, s*s*c] -> NHWC [B,h*s,w*s,cpad] (extra channels zero)."""
    cpad = cpad or c
    x = _ffi.check(x, H16, "x", (b * h * w, s * s * c))
    alloc = torch.zeros if cpad != c else torch.empty
    out = alloc((b, h * s, w * s, cpad), dtype=x.dtype, device=x.device)
    _ffi.call("m3_unshuffle_bf16", _ffi.ptr(x), _ffi.ptr(out), b, h, w, s, c, cpad, _ffi.stream_ptr())
    return out


def upsample2x(x):
    x = _ffi.check(x, H16, "x")
    b, h, w, c = x.shape
    out = torch.empty((b, 2 * h, 2 * w, c), dtype=x.dtype, device=x.device)
    _ffi.call("m3_upsample2x_dt", _ffi.ptr(x), _ffi.ptr(out), b, h, w, c, DT_CODE[x.dtype], _ffi.stream_ptr())
    return out


def add_upsample2x(low, y):
    """bilinear x2 (align_corners) of low [B,h,w,C], cropped to y's [B,OH,OW,C] (OH <= 2h, OW <= 2w), plus y - one launch,
    one rounding; the DPT fusion block's 'upsampled coarser path + refined skip connection'."""
    low = _ffi.check(low, H16, "low")
    y = _ffi.check(y, low.dtype, "y")
    b, h, w, c = low.shape
    if y.dim() != 4 or y.shape[0] != b or y.shape[3] != c or y.shape[1] > 2 * h or y.shape[2] > 2 * w:
        raise ValueError(f"y must be [{b},<={2 * h},<={2 * w},{c}], got {tuple(y.shape)}")
    out = torch.empty_like(y)
    _ffi.call("m3_add_upsample2x_dt", _ffi.ptr(low), _ffi.ptr(y), _ffi.ptr(out), b, h, w, y.shape[1], y.shape[2], c,
              DT_CODE[low.dtype], _ffi.stream_ptr())
    return out


def pts_post(raw):
    """[...,4] f32 -> pts3d [...,3], conf [...]."""
    raw = _ffi.check(raw, torch.float32, "raw")
    p = raw.numel() // 4
    pts = torch.empty(raw.shape[:-1] + (3,), dtype=torch.float32, device=raw.device)
    conf = torch.empty(raw.shape[:-1], dtype=torch.float32, device=raw.device)
    _ffi.call("m3_pts_post", _ffi.ptr(raw), _ffi.ptr(pts), _ffi.ptr(conf), p, _ffi.stream_ptr())
    return pts, conf


def desc_post(f, b, h, w, desc_dtype=torch.float32):
    """Pixel shuffle + L2 normalisation of the feature head: desc [b,h,w,24] (float32, or float16 = "fp16 features"),
    desc_conf [b,h,w] float32."""
    f = _ffi.check(f, H16, "f", (b * (h // 16) * (w // 16), 6400))
    if desc_dtype not in (torch.float32, torch.float16):
        raise ValueError(f"desc_dtype must be torch.float32 or torch.float16, got {desc_dtype}")
    desc = torch.empty((b, h, w, 24), dtype=desc_dtype, device=f.device)
    dconf = torch.empty((b, h, w), dtype=torch.float32, device=f.device)
    _ffi.call("m3_desc_post_f16" if desc_dtype == torch.float16 else "m3_desc_post_dt", _ffi.ptr(f), _ffi.ptr(desc),
              _ffi.ptr(dconf), b, h, w, DT_CODE[f.dtype], _ffi.stream_ptr())
    return desc, dconf


def add(a, b):
    a = _ffi.check(a, H16, "a")
    b = _ffi.check(b, H16, "b", tuple(a.shape))
    out = torch.empty_like(a)
    _ffi.call("m3_add_dt", _ffi.ptr(a), _ffi.ptr(b), _ffi.ptr(out), a.numel(), _same16(a, b), _ffi.stream_ptr())
    return out


def gemm_grouped2(a, w0, w1, b0, b1, epi=EPI_BF16, out=None, resid=None, rope=None, pv_bf16=False):
    """Two same-shape GEMMs in one launch.  a [2,M,K] bf16, weights [N,K] x2 -> out [2,M,N].
    rope = (positions int32 [T,2] or table f32 [T,2,2,16], rope_cols[, q_cols, q_scale]) with epi=EPI_BF16_ROPE;
    pv_bf16 as gemm_rope."""
    a = _ffi.check(a, H16, "a")
    if a.dim() != 3 or a.shape[0] != 2:
        raise ValueError(f"a must be [2,M,K], got {tuple(a.shape)}")
    _, m, k = a.shape
    w0 = _ffi.check(w0, H16, "w0")
    w1 = _ffi.check(w1, H16, "w1", tuple(w0.shape))
    if pv_bf16 and epi != EPI_BF16_ROPE:
        raise ValueError("pv_bf16 goes with epi=EPI_BF16_ROPE")
    dt = _pv_code(_same16(a, w0, w1), pv_bf16)
    n = w0.shape[0]
    odt = torch.float32 if epi in _F32_EPIS else a.dtype
    if out is None:
        out = torch.empty((2, m, n), dtype=odt, device=a.device)
    elif out.dtype != odt or out.dim() != 3 or out.shape[0] != 2 or out.shape[1] != m or out.shape[2] < n or not out.is_contiguous():
        raise ValueError("bad `out`")
    ldc = out.shape[2]                                           # > n: the extra columns (zero padding) are left alone
    if resid is not None and (resid.dtype != odt or tuple(resid.shape) != tuple(out.shape) or not resid.is_contiguous()):
        raise ValueError("bad `resid`")
    rtok, rc, qc, qs = (tuple(rope) + (0, 1.0))[:4] if rope is not None else (None, 0, 0, 1.0)
    tpi, by_pos = 0, False
    if rtok is not None:
        rtok, tpi, by_pos = _rope_table(rtok)
    if by_pos:
        if epi != EPI_BF16_ROPE or resid is not None:
            raise ValueError("rope positions go with epi=EPI_BF16_ROPE and no residual")
        return gemm_ex(a, w0, b0, EPI_BF16_ROPE, out=out, w1=w1, bias1=b1, rope=(rtok, rc, qc, qs), pv_bf16=pv_bf16)
    e0 = _prof_begin()
    _ffi.call("m3_gemm_grouped2_dt", _ffi.ptr(a), _ffi.ptr(w0), _ffi.ptr(w1), _ffi.ptr(b0), _ffi.ptr(b1), _ffi.ptr(out),
              _ffi.ptr(resid), m, n, k, ldc, m * k, m * ldc, epi, _ffi.ptr(rtok), tpi, rc, int(qc), float(qs), dt, _ffi.stream_ptr())
    _prof_end(e0, _gemm_kind(m, n, 2), 4.0 * m * n * k,
              2.0 * (2.0 * (m * k + n * k) + out.element_size() * m * n * (1 if resid is None else 2)))
    return out


LN_EPS = 1e-6


def ln_fold_buffers(rows: int, cols: int, dtype, device, groups: int = 1):
    """Buffers of the LayerNorm fold for a [rows, cols] (or [2, rows, cols]) fp32 stream: (x16, stats) - the 16-bit copy of the
    stream and the partial statistics [slots, rows, 2] (slot-major; ln_slot_count) that a producer GEMM (gemm_ex(..., fold_out=...)) fills."""
    slots = ln_slot_count(rows, cols, groups)
    lead = (2,) if groups == 2 else ()
    return (torch.empty(lead + (rows, cols), dtype=dtype, device=device),
            torch.empty(lead + (slots, rows, 2), dtype=torch.float32, device=device))


def ln_slot_count(rows: int, cols: int, groups: int = 1) -> int:
    """Statistics slots per row a producer launch of a [rows, cols] stream writes (m3_ln_slot_count): nodes of the rows'
    canonical sum tree, cols / 64 pairs, cols / 128 halves or one slot per 256- / 192-column top node - the widest the tile
    the launch will be dispatched to allows."""
    n = int(_ffi.lib().m3_ln_slot_count(int(rows), int(cols), int(groups)))
    if n <= 0:
        raise ValueError(f"no LayerNorm-fold statistics for a [{rows}, {cols}] stream")
    return n


def ln_hl_buffers(rows: int, cols: int, device, groups: int = 1):
    """The hi / lo form of a [rows, cols] (or [2, rows, cols]) residual stream: (hi, lo, stats) - two fp16 planes with
    x = hi + lo (22 significant bits) and the slot-major row statistics.  gemm_ex(..., hl=...) keeps all three up to date;
    `hi` is what the projections behind a LayerNorm multiply (gemm_ex(hi, ..., fold_in=(stats, ...)))."""
    hi, st = ln_fold_buffers(rows, cols, torch.float16, device, groups)
    return hi, torch.empty_like(hi), st


def hl_to_f32(hl):
    """fp32 view of a hi / lo stream (the two LayerNorms that still run as kernels - enc_norm, dec_norm - read it)."""
    return torch.add(hl[0].float(), hl[1])


def layernorm_hl(hl, g0, b0, g1=None, b1=None, eps=1e-6, dtype=torch.float16):
    """LayerNorm of a hi / lo stream (ln_hl_buffers) in one pass over the two planes: [M,C] -> [M,C], or [2,M,C] -> [2,M,C]
    with (g1, b1) for the second group.  Bit-identical to layernorm(hl_to_f32(hl), ...)."""
    hi, lo = hl[0], hl[1]
    if hi.dtype != torch.float16 or lo.dtype != torch.float16 or hi.shape != lo.shape or not (hi.is_contiguous() and lo.is_contiguous()):
        raise ValueError("bad hi / lo planes")
    c = hi.shape[-1]
    rows = hi.numel() // c
    split = hi.shape[-2] if hi.dim() == 3 else rows
    if hi.dim() == 3 and (hi.shape[0] != 2 or g1 is None or b1 is None):
        raise ValueError("two groups need [2,M,C] planes and both parameter sets")
    out = torch.empty(hi.shape, dtype=dtype, device=hi.device)
    g1 = g0 if g1 is None else g1
    b1 = b0 if b1 is None else b1
    _ffi.call("m3_layernorm_hl_dt", _ffi.ptr(hi), _ffi.ptr(lo), _ffi.ptr(g0), _ffi.ptr(b0), _ffi.ptr(g1), _ffi.ptr(b1), _ffi.ptr(out),
              rows, c, split, float(eps), DT_CODE[dtype], _ffi.stream_ptr())
    return out


def gemm_ex(a, w, bias=None, epi: int = EPI_BF16, out=None, resid=None, w1=None, bias1=None, rope=None, pv_bf16: bool = False,
            fold_in=None, fold_out=None, a_swap: bool = False, hl=None):
    """m3_gemm_ex: one or two groups, any epilogue, with the LayerNorm fold (include/m3slam_model.h).
      a [M,K] (w1 None) or [2,M,K]; w (and w1) [N,K]; out / resid like gemm / gemm_grouped2; rope = (positions int32 [T,2],
      rope_cols[, q_cols, q_scale]) with EPI_BF16_ROPE.
      fold_out = (x16, stats): PRODUCER - with EPI_F32 / EPI_F32_ACCUM also writes the 16-bit copy of the fp32 output and the
        rows' partial statistics (ln_fold_buffers).
      fold_in = (stats, colsum[, colsum1]): CONSUMER - `a` is the 16-bit copy of the raw stream, w carries gamma, bias carries
        beta . W^T; the epilogue applies rstd * (acc - mean * colsum) before the bias.
      a_swap (two groups): group g multiplies a[1 - g] (and reads the statistics of stream 1 - g) - the decoder's
        cross-attention memory is the other view's stream.
      hl = (hi, lo, stats) (ln_hl_buffers; fp16): PRODUCER on a hi / lo stream, in place - EPI_F32: the stream becomes
        a @ w^T + bias; EPI_F32_ACCUM: it is updated by that product.  No fp32 tensor is written; returns None."""
    grouped = w1 is not None
    a = _ffi.check(a, H16, "a")
    w = _ffi.check(w, H16, "w")
    if grouped:
        if a.dim() != 3 or a.shape[0] != 2:
            raise ValueError(f"a must be [2,M,K], got {tuple(a.shape)}")
        w1 = _ffi.check(w1, H16, "w1", tuple(w.shape))
    elif a.dim() != 2:
        raise ValueError(f"a must be [M,K], got {tuple(a.shape)}")
    m, k = a.shape[-2:]
    n = w.shape[0]
    if w.shape[1] != k:
        raise ValueError(f"K mismatch: a {tuple(a.shape)} vs w {tuple(w.shape)}")
    if pv_bf16 and epi != EPI_BF16_ROPE:
        raise ValueError("pv_bf16 goes with epi=EPI_BF16_ROPE")
    dt = _pv_code(_same16(a, w, w1), pv_bf16)
    odt = torch.float32 if epi in _F32_EPIS else a.dtype
    lead = (2,) if grouped else ()
    if hl is not None:
        if epi not in _F32_EPIS or a.dtype != torch.float16 or out is not None or resid is not None or fold_out is not None:
            raise ValueError("hl goes with EPI_F32 / EPI_F32_ACCUM on fp16 operands, without out / resid / fold_out")
        hi, lo, stats = hl
        for t_ in (hi, lo):
            if t_.dtype != torch.float16 or tuple(t_.shape) != lead + (m, n) or not t_.is_contiguous():
                raise ValueError("bad hi / lo planes (ln_hl_buffers)")
        slots = ln_slot_count(m, n, 2 if grouped else 1)
        if stats.dtype != torch.float32 or tuple(stats.shape) != lead + (slots, m, 2) or not stats.is_contiguous():
            raise ValueError("bad statistics buffer (ln_hl_buffers)")
        d = _ffi.GemmDesc()
        d.A, d.W, d.W1 = a.data_ptr(), w.data_ptr(), _ffi.ptr(w1)
        d.bias, d.bias1 = _ffi.ptr(bias), _ffi.ptr(bias1)
        d.c16, d.c_lo, d.stats_out, d.stats_slots = hi.data_ptr(), lo.data_ptr(), stats.data_ptr(), slots
        if epi == EPI_F32_ACCUM:
            d.R, d.r_lo = hi.data_ptr(), lo.data_ptr()
        d.M, d.N, d.K, d.ldc, d.epilogue, d.dtype, d.groups = m, n, k, n, epi, dt, 2 if grouped else 1
        d.a_gstride, d.c_gstride = (m * k, m * n) if grouped else (0, 0)
        d.stats_gstride = m * slots * 2 if grouped else 0
        e0 = _prof_begin()
        _ffi.call("m3_gemm_ex", C.addressof(d), _ffi.stream_ptr())
        g_ = 2 if grouped else 1
        _prof_end(e0, _gemm_kind(m, n, g_), 2.0 * g_ * m * n * k,
                  g_ * (2.0 * (m * k + n * k) + 4.0 * m * n * (1 if epi == EPI_F32 else 2) + 8.0 * m * (n // 32)),
                  f"gemm_ex hl {g_}x{m}x{n}x{k} epi{epi}")
        return None
    if out is None:
        out = torch.empty(lead + (m, n), dtype=odt, device=a.device)
    elif out.dtype != odt or tuple(out.shape[:-1]) != lead + (m,) or out.shape[-1] < n or not out.is_contiguous():
        raise ValueError("bad `out`")
    ldc = out.shape[-1]
    if resid is not None and (resid.dtype != odt or tuple(resid.shape) != tuple(out.shape) or not resid.is_contiguous()):
        raise ValueError("bad `resid`")
    d = _ffi.GemmDesc()
    d.A = (a[1] if (grouped and a_swap) else a).data_ptr()
    d.W, d.W1 = w.data_ptr(), _ffi.ptr(w1)
    d.bias, d.bias1 = _ffi.ptr(bias), _ffi.ptr(bias1)
    d.C, d.R = out.data_ptr(), _ffi.ptr(resid)
    d.M, d.N, d.K, d.ldc, d.epilogue, d.dtype, d.groups = m, n, k, ldc, epi, dt, 2 if grouped else 1
    d.a_gstride = (-m * k if a_swap else m * k) if grouped else 0
    d.c_gstride = m * ldc if grouped else 0
    if rope is not None:
        rtok, rc, qc, qs = (tuple(rope) + (0, 1.0))[:4]
        d.rope_max_pos = _rope_pmax(rtok)
        rtok = _ffi.check(rtok, torch.int32, "rope positions", (None, 2))
        d.rope_pos, d.tokens_per_image, d.rope_base = rtok.data_ptr(), rtok.shape[0], float(ROPE_BASE)
        d.rope_cols, d.q_cols, d.q_scale = int(rc), int(qc), float(qs)
    if fold_out is not None:
        x16, stats = fold_out
        slots = ln_slot_count(m, n, 2 if grouped else 1)
        if (x16.dtype != a.dtype or tuple(x16.shape) != tuple(out.shape) or not x16.is_contiguous() or stats.dtype != torch.float32
                or tuple(stats.shape) != lead + (slots, m, 2) or not stats.is_contiguous() or ldc != n):
            raise ValueError("bad fold_out buffers (ln_fold_buffers)")
        d.c16, d.stats_out, d.stats_slots = x16.data_ptr(), stats.data_ptr(), slots
        d.stats_gstride = m * slots * 2 if grouped else 0
    if fold_in is not None:
        stats, cs0 = fold_in[0], fold_in[1]
        cs1 = fold_in[2] if len(fold_in) > 2 else None
        if (stats.dtype != torch.float32 or stats.dim() != len(lead) + 3 or tuple(stats.shape[:len(lead)]) != lead
                or tuple(stats.shape[-2:]) != (m, 2) or not stats.is_contiguous()):
            raise ValueError("bad fold_in statistics")
        slots = stats.shape[-3]                                  # the producer's tree level (checked in C)
        cs0 = _ffi.check(cs0, torch.float32, "colsum", (n,))
        if grouped:
            cs1 = _ffi.check(cs1, torch.float32, "colsum1", (n,))
        per = m * slots * 2
        d.ln_stats = (stats[1] if (grouped and a_swap) else stats).data_ptr()
        d.ln_colsum, d.ln_colsum1 = cs0.data_ptr(), _ffi.ptr(cs1)
        d.ln_slots, d.ln_eps = slots, float(LN_EPS)
        d.ln_gstride = (-per if a_swap else per) if grouped else 0
    e0 = _prof_begin()
    _ffi.call("m3_gemm_ex", C.addressof(d), _ffi.stream_ptr())
    g = 2 if grouped else 1
    nb = 2.0 * (m * k + n * k) + out.element_size() * m * n * (1 if resid is None else 2)
    if fold_out is not None:
        nb += 2.0 * m * n + 8.0 * m * (n // 32)
    _prof_end(e0, _gemm_kind(m, n, g), 2.0 * g * m * n * k, g * nb, f"gemm_ex {g}x{m}x{n}x{k} epi{epi} {a.dtype}")
    return out


def layernorm_dual2(x, own, cross, eps=1e-6, dtype=torch.bfloat16):
    """x f32 [2,M,C] -> (y_own, y_cross), both 16-bit [2,M,C], in one pass over x: y_own[g] = LN(x[g]) with own[g] =
    (gamma, beta); y_cross[g] = LN(x[1-g]) with cross[g] - i.e. layernorm_grouped2(x, *own) and
    layernorm_grouped2(x, *cross, swap=True) together."""
    x = _ffi.check(x, torch.float32, "x")
    _, m, c = x.shape
    y_own = torch.empty((2, m, c), dtype=dtype, device=x.device)
    y_cross = torch.empty((2, m, c), dtype=dtype, device=x.device)
    (ga0, ba0), (ga1, ba1) = own
    (gb0, bb0), (gb1, bb1) = cross
    _ffi.call("m3_layernorm_dual2_dt", _ffi.ptr(x), _ffi.ptr(ga0), _ffi.ptr(ba0), _ffi.ptr(ga1), _ffi.ptr(ba1), _ffi.ptr(gb0),
              _ffi.ptr(bb0), _ffi.ptr(gb1), _ffi.ptr(bb1), _ffi.ptr(y_own), _ffi.ptr(y_cross), m, c, float(eps), DT_CODE[dtype],
              _ffi.stream_ptr())
    return y_own, y_cross


def layernorm_grouped2(x, g0, b0, g1, b1, swap=False, eps=1e-6, dtype=torch.bfloat16):
    """x f32 [2,M,C] -> 16-bit [2,M,C]; group v uses (g_v, b_v); swap=True normalises the OTHER group's rows."""
    x = _ffi.check(x, torch.float32, "x")
    _, m, c = x.shape
    out = torch.empty((2, m, c), dtype=dtype, device=x.device)
    _ffi.call("m3_layernorm_grouped2_dt", _ffi.ptr(x), _ffi.ptr(g0), _ffi.ptr(b0), _ffi.ptr(g1), _ffi.ptr(b1),
              _ffi.ptr(out), m, c, m if swap else 0, float(eps), DT_CODE[dtype], _ffi.stream_ptr())
    return out
