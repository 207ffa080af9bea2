"""Operator-level Python entry points: torch tensors in, HIP kernels through the C ABI, torch tensors out.

torch is used for device memory and streams only; every wrapper checks device / dtype / contiguity and raises if the
input is not a ROCm tensor -- nothing here computes on the CPU or with torch ops.
"""
from __future__ import annotations

import collections
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import F16, F32, check, lib

_DT = {torch.float16: F16, torch.float32: F32}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev(t: torch.Tensor, name: str, dtypes=None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"clipmi: `{name}` must be a tensor on the GPU (got {getattr(t, 'device', type(t))}); "
                           "the HIP path has no CPU fallback")
    if dtypes is not None and t.dtype not in dtypes:
        raise TypeError(f"clipmi: `{name}` has dtype {t.dtype}, expected one of {dtypes}")
    # launches go to the CURRENT device's stream with raw pointers: a tensor that lives on another GPU would be handed to
    # the wrong device.  One process drives one GPU here; anything else must say so with torch.cuda.device(...).
    if t.device.index != torch.cuda.current_device():
        raise RuntimeError(f"clipmi: `{name}` is on {t.device} but the current device is cuda:{torch.cuda.current_device()}; "
                           f"run the call under `with torch.cuda.device({t.device.index}):`")
    return t if t.is_contiguous() else t.contiguous()


def _opt(t: Optional[torch.Tensor], name: str, dtypes) -> Tuple[Optional[torch.Tensor], Optional[int]]:
    if t is None:
        return None, None
    t = _dev(t, name, dtypes)
    return t, t.data_ptr()


def gemm_f16(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None,
             residual: Optional[torch.Tensor] = None, epilogue: int = _lib.EPI_NONE,
             out_dtype: torch.dtype = torch.float16, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``epi(a[M,K] @ w[N,K]^T)`` -- nn.Linear and friends (reference clip/model.py:174-176,183,422,611)."""
    a = _dev(a, "a", (torch.float16,))
    w = _dev(w, "w", (torch.float16,))
    M, K = a.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise ValueError(f"gemm: a is [{M},{K}] but w is {tuple(w.shape)}")
    bias, pb = _opt(bias, "bias", (torch.float32,))
    residual, pr = _opt(residual, "residual", (torch.float16,) if epilogue == _lib.EPI_BIAS_RESIDUAL16_RELU else (torch.float32,))
    if residual is not None and tuple(residual.shape) != (M, N):
        raise ValueError(f"gemm: residual is {tuple(residual.shape)}, expected [{M},{N}]")
    if out is None:
        out = torch.empty(M, N, dtype=out_dtype, device=a.device)
    check(lib.clipmi_gemm_f16(a.data_ptr(), K, w.data_ptr(), K, pb, pr, out.data_ptr(), N, _DT[out.dtype],
                              M, N, K, epilogue, _stream()), "clipmi_gemm_f16")
    return out


def gemm_residual_f16(a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, x16: torch.Tensor):
    """``x16 <- fp16(x16 + a @ w^T + bias)`` in place (one rounding of the fp32 sum): the residual GEMM of a block on the fp16
    stream (reference clip/model.py:186-187).  Returns (stats fp32 [8, M, 2], parts): the LayerNorm-fold row partials
    ``stats[t, m] = (sum, sum of squares)`` over the t-th column tile of the rounded row m, ``parts`` tiles of them."""
    import ctypes as C
    a = _dev(a, "a", (torch.float16,))
    w = _dev(w, "w", (torch.float16,))
    bias = _dev(bias, "bias", (torch.float32,))
    if not (isinstance(x16, torch.Tensor) and x16.is_cuda and x16.dtype == torch.float16 and x16.is_contiguous()):
        raise TypeError("gemm_residual_f16: x16 must be a contiguous fp16 GPU tensor (it is updated in place)")
    M, K = a.shape
    N = w.shape[0]
    if w.shape[1] != K or tuple(x16.shape) != (M, N) or bias.numel() != N:
        raise ValueError(f"gemm_residual_f16: a {tuple(a.shape)}, w {tuple(w.shape)}, bias {tuple(bias.shape)}, x16 {tuple(x16.shape)}")
    stats = torch.empty(8, M, 2, dtype=torch.float32, device=a.device)
    parts = C.c_int(0)
    check(lib.clipmi_gemm_residual_f16(a.data_ptr(), K, w.data_ptr(), K, bias.data_ptr(), x16.data_ptr(), N, stats.data_ptr(),
                                       C.byref(parts), M, N, K, _stream()), "clipmi_gemm_residual_f16")
    return stats, parts.value


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5,
              out_dtype: Optional[torch.dtype] = None, gather_rows: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32-statistics LayerNorm over the last dim (reference clip/model.py:153-159)."""
    x = _dev(x, "x", (torch.float16, torch.float32))
    gamma = _dev(gamma, "gamma", (torch.float32,))
    beta = _dev(beta, "beta", (torch.float32,))
    D = x.shape[-1]
    x2 = x.reshape(-1, D)
    out_dtype = out_dtype or x.dtype
    if gather_rows is not None:
        gather_rows = _dev(gather_rows, "gather_rows", (torch.int32,))
        rows = gather_rows.numel()
        y = torch.empty(rows, D, dtype=out_dtype, device=x.device)
        pg = gather_rows.data_ptr()
    else:
        rows = x2.shape[0]
        y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
        pg = None
    check(lib.clipmi_layernorm(x2.data_ptr(), _DT[x.dtype], D, pg, gamma.data_ptr(), beta.data_ptr(), y.data_ptr(),
                               _DT[out_dtype], D, rows, D, eps, _stream()), "clipmi_layernorm")
    return y


def attention(qkv: torch.Tensor, n_seq: int, seq_len: int, n_head: int, causal: bool) -> torch.Tensor:
    """qkv fp16 [n_seq*seq_len, 3*64*n_head] -> fp16 [n_seq*seq_len, 64*n_head] (SURVEY a-5a)."""
    qkv = _dev(qkv, "qkv", (torch.float16,))
    D = 64 * n_head
    if qkv.shape != (n_seq * seq_len, 3 * D):
        raise ValueError(f"attention: qkv shape {tuple(qkv.shape)} != {(n_seq * seq_len, 3 * D)}")
    out = torch.empty(n_seq * seq_len, D, dtype=torch.float16, device=qkv.device)
    check(lib.clipmi_attention(qkv.data_ptr(), out.data_ptr(), n_seq, seq_len, n_head, int(bool(causal)), _stream()),
          "clipmi_attention")
    return out


def patchify(image: torch.Tensor, patch: int, kpad: Optional[int] = None) -> torch.Tensor:
    image = _dev(image, "image", (torch.float16, torch.float32))
    B, ch, R, R2 = image.shape
    if ch != 3 or R != R2:
        raise ValueError(f"patchify: expected [B,3,R,R], got {tuple(image.shape)}")
    kpad = kpad or (3 * patch * patch + 63) // 64 * 64
    g = R // patch
    col = torch.empty(B * g * g, kpad, dtype=torch.float16, device=image.device)
    check(lib.clipmi_patchify(image.data_ptr(), _DT[image.dtype], col.data_ptr(), B, R, patch, kpad, _stream()),
          "clipmi_patchify")
    return col


def l2_normalize(f: torch.Tensor, out_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """``f / f.norm(dim=-1, keepdim=True)`` in fp32 (reference zsclip.py:99); ``out_dtype=torch.float16`` rounds the
    quotient once (the exchange format of the multi-GPU path)."""
    f = _dev(f, "features", (torch.float16, torch.float32))
    rows, E = f.shape
    out = torch.empty(rows, E, dtype=out_dtype, device=f.device)
    with torch.cuda.device(f.device):
        check(lib.clipmi_l2_normalize_to(f.data_ptr(), _DT[f.dtype], out.data_ptr(), _DT[out_dtype], rows, E, _stream()),
              "clipmi_l2_normalize_to")
    return out


def logits_fused(img_n: torch.Tensor, txt_n: torch.Tensor, scale: float, dac_conf: Optional[torch.Tensor] = None,
                 want_conf_pred: bool = True):
    """(scale*img_n) @ txt_n^T  [+ DAC row scale]  [+ softmax top-1 conf / pred] on features that are ALREADY normalised
    (the gathered embeddings of the multi-GPU path, CoCoOp's shared image features): one launch of the fused tail."""
    logits, _, conf, pred = fused_tail(img_n, txt_n, scale, dac_conf, want_conf_pred, normalize=False)
    return logits, conf, pred


_TAIL_WS = collections.OrderedDict()   # (device index, stream) -> zeroed int32 ticket counters (the kernel leaves them zero); LRU, bounded
_TAIL_WS_MAX = 16                       # streams with a live workspace: short-lived per-thread streams must not pin buffers for ever


def _tail_workspace(device: torch.device, batch: int, classes: int):
    """Ticket counters of the fused tail: one buffer per (device, stream).  Launches on ONE stream run in order and each leaves the
    counters zero; two launches in flight on different streams must not share them (a foreign ticket would skip or double a row pass).
    Returns (key, buffer); the least recently used entries beyond _TAIL_WS_MAX are dropped (the caching allocator keeps a dropped buffer
    alive until the work queued on its stream is done), and `_tail_workspace_failed` drops the entry of a launch that raised."""
    need = lib.clipmi_fused_tail_workspace_bytes(batch, classes)
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _TAIL_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.zeros(max(need, 4096), dtype=torch.uint8, device=device)
        _TAIL_WS[key] = ws
    _TAIL_WS.move_to_end(key)
    while len(_TAIL_WS) > _TAIL_WS_MAX:
        _TAIL_WS.popitem(last=False)
    return key, ws


def _tail_workspace_failed(key) -> None:
    """A launch that raised may have left tickets behind: the next call on this stream starts from a freshly zeroed buffer."""
    _TAIL_WS.pop(key, None)


def fused_tail(img: torch.Tensor, txt_n: torch.Tensor, scale: float, dac_conf: Optional[torch.Tensor] = None,
               want_conf_pred: bool = True, normalize: bool = True, labels: Optional[torch.Tensor] = None,
               bins: Optional[torch.Tensor] = None, n_bins: int = 0):
    """The tail of the path in ONE launch (include/clipmi.h, clipmi_fused_tail): L2-normalise the image features,
    ``scale * img_n @ txt_n^T``, DAC row scale, softmax top-1 and -- when ``labels`` and ``bins`` are given -- the ECE bin
    accumulation.  Returns (logits, img_n, conf, pred); img_n is ``img`` itself when ``normalize`` is False."""
    img = _dev(img, "img", (torch.float32, torch.float16))
    txt_n = _dev(txt_n, "txt_n", (torch.float32,))
    if txt_n.device != img.device:
        raise RuntimeError(f"fused_tail: img on {img.device}, txt_n on {txt_n.device}")
    B, E = img.shape
    Cn = txt_n.shape[0]
    if txt_n.shape[1] != E:
        raise ValueError("fused_tail: feature widths differ")
    dac_conf, pd = _opt(dac_conf, "dac_conf", (torch.float32,))
    if dac_conf is not None and dac_conf.numel() != Cn:
        raise ValueError("fused_tail: dac_conf must have one entry per class")
    labels, pl = _opt(labels, "labels", (torch.int64,))
    bins, pb = _opt(bins, "bins", (torch.float64,))
    if (pb is None) != (pl is None):
        raise ValueError("fused_tail: labels and bins come together")
    if bins is not None and (bins.numel() != 3 * (n_bins + 1) or labels.numel() != B):
        raise ValueError("fused_tail: bins must hold 3*(n_bins+1) float64 and labels one entry per image")
    logits = torch.empty(B, Cn, dtype=torch.float32, device=img.device)
    img_n = torch.empty(B, E, dtype=torch.float32, device=img.device) if normalize else img
    conf = pred = None
    pc = pp = None
    if want_conf_pred or bins is not None:
        conf = torch.empty(B, dtype=torch.float32, device=img.device)
        pred = torch.empty(B, dtype=torch.int32, device=img.device)
        pc, pp = conf.data_ptr(), pred.data_ptr()
    key, ws = _tail_workspace(img.device, B, Cn)
    try:
        with torch.cuda.device(img.device):
            check(lib.clipmi_fused_tail(img.data_ptr(), _DT[img.dtype], int(normalize), txt_n.data_ptr(), float(scale), pd, logits.data_ptr(),
                                        img_n.data_ptr() if normalize else None, pc, pp, pl, pb, int(n_bins), ws.data_ptr(), ws.numel(),
                                        B, Cn, E, _stream()), "clipmi_fused_tail")
    except Exception:
        _tail_workspace_failed(key)
        raise
    return logits, img_n, conf, pred


def softmax_rows(logits: torch.Tensor, dac_conf: Optional[torch.Tensor] = None, want_conf_pred: bool = False):
    """probs = softmax(logits * dac_conf[argmax]) row-wise (vl_calibrator.py:83-109, DAC / plain branches); logits untouched."""
    logits = _dev(logits, "logits", (torch.float32,))
    B, Cn = logits.shape
    dac_conf, pd = _opt(dac_conf, "dac_conf", (torch.float32,))
    if dac_conf is not None and dac_conf.numel() != Cn:
        raise ValueError("softmax_rows: dac_conf must have one entry per class")
    probs = torch.empty_like(logits)
    conf = pred = None
    pc = pp = None
    if want_conf_pred:
        conf = torch.empty(B, dtype=torch.float32, device=logits.device)
        pred = torch.empty(B, dtype=torch.int32, device=logits.device)
        pc, pp = conf.data_ptr(), pred.data_ptr()
    check(lib.clipmi_softmax_rows(logits.data_ptr(), pd, probs.data_ptr(), pc, pp, B, Cn, _stream()), "clipmi_softmax_rows")
    return (probs, conf, pred) if want_conf_pred else probs


def ece_accumulate(conf: torch.Tensor, pred: torch.Tensor, labels: torch.Tensor, bins: torch.Tensor, n_bins: int) -> None:
    conf = _dev(conf, "conf", (torch.float32,))
    pred = _dev(pred, "pred", (torch.int32,))
    labels = _dev(labels, "labels", (torch.int64,))
    bins = _dev(bins, "bins", (torch.float64,))
    if bins.numel() != 3 * (n_bins + 1):
        raise ValueError("ece_accumulate: bins must hold 3*(n_bins+1) float64")
    check(lib.clipmi_ece_accumulate(conf.data_ptr(), pred.data_ptr(), labels.data_ptr(), conf.numel(), bins.data_ptr(),
                                    n_bins, _stream()), "clipmi_ece_accumulate")


# ---- CoCoOp glue (cocoop.py:154-199) -------------------------------------------------------------------------------
def cocoop_ctx(img_n: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor,
               ctx: torch.Tensor) -> torch.Tensor:
    """ctx + meta_net(img_n) per image -> fp32 [B, n_ctx, D]."""
    f32 = (torch.float32,)
    img_n, w1, b1, w2, b2, ctx = (_dev(t, n, f32) for t, n in ((img_n, "img_n"), (w1, "w1"), (b1, "b1"), (w2, "w2"), (b2, "b2"), (ctx, "ctx")))
    B, E = img_n.shape
    H, D, n_ctx = w1.shape[0], w2.shape[0], ctx.shape[0]
    if w1.shape != (H, E) or b1.shape != (H,) or w2.shape != (D, H) or b2.shape != (D,) or ctx.shape != (n_ctx, D):
        raise ValueError("cocoop_ctx: meta-net shapes do not chain")
    out = torch.empty(B, n_ctx, D, dtype=torch.float32, device=img_n.device)
    check(lib.clipmi_cocoop_ctx(img_n.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), ctx.data_ptr(),
                                out.data_ptr(), B, E, H, D, n_ctx, _stream()), "clipmi_cocoop_ctx")
    return out


def cocoop_prompts(base: torch.Tensor, ctx_shifted: torch.Tensor) -> torch.Tensor:
    """[C,L,D] base embeddings x [nb,n_ctx,D] shifted contexts -> fp16 prompts [nb*C, L, D]."""
    base = _dev(base, "base", (torch.float16, torch.float32))
    ctx_shifted = _dev(ctx_shifted, "ctx_shifted", (torch.float32,))
    Cn, L, D = base.shape
    nb, n_ctx, d2 = ctx_shifted.shape
    if d2 != D:
        raise ValueError("cocoop_prompts: widths differ")
    out = torch.empty(nb * Cn, L, D, dtype=torch.float16, device=base.device)
    check(lib.clipmi_cocoop_prompts(base.data_ptr(), _DT[base.dtype], ctx_shifted.data_ptr(), out.data_ptr(), nb, Cn, L, D, n_ctx,
                                    _stream()), "clipmi_cocoop_prompts")
    return out


def logits_per_image(img_n: torch.Tensor, txt: torch.Tensor, scale: float, dac_conf: Optional[torch.Tensor] = None,
                     want_conf_pred: bool = True, want_last_text: bool = True):
    """logits[b,c] = scale * <img_n[b], normalise(txt[b,c])>; txt [B,C,E] un-normalised."""
    img_n = _dev(img_n, "img_n", (torch.float32,))
    txt = _dev(txt, "txt", (torch.float32,))
    B, E = img_n.shape
    if txt.dim() != 3 or txt.shape[0] != B or txt.shape[2] != E:
        raise ValueError(f"logits_per_image: txt must be [B={B}, C, E={E}], got {tuple(txt.shape)}")
    Cn = txt.shape[1]
    dac_conf, pd = _opt(dac_conf, "dac_conf", (torch.float32,))
    if dac_conf is not None and dac_conf.numel() != Cn:
        raise ValueError("logits_per_image: dac_conf must have one entry per class")
    logits = torch.empty(B, Cn, dtype=torch.float32, device=img_n.device)
    conf = pred = last = None
    pc = pp = pl = None
    if want_conf_pred:
        conf = torch.empty(B, dtype=torch.float32, device=img_n.device)
        pred = torch.empty(B, dtype=torch.int32, device=img_n.device)
        pc, pp = conf.data_ptr(), pred.data_ptr()
    if want_last_text:
        last = torch.empty(Cn, E, dtype=torch.float32, device=img_n.device)
        pl = last.data_ptr()
    check(lib.clipmi_logits_per_image(img_n.data_ptr(), txt.data_ptr(), float(scale), pd, logits.data_ptr(), pc, pp, pl, B, Cn, E,
                                      _stream()), "clipmi_logits_per_image")
    return logits, conf, pred, last


def group_mean(x: torch.Tensor, group: int) -> torch.Tensor:
    """[G*group, E] -> [G, E] mean over consecutive groups of rows (ProDA's prompt-ensemble mean, proda.py:328-332)."""
    x = _dev(x, "x", (torch.float32,))
    rows, E = x.shape
    if group <= 0 or rows % group:
        raise ValueError(f"group_mean: {rows} rows do not split into groups of {group}")
    out = torch.empty(rows // group, E, dtype=torch.float32, device=x.device)
    check(lib.clipmi_group_mean(x.data_ptr(), out.data_ptr(), rows // group, group, E, _stream()), "clipmi_group_mean")
    return out


def adapter_blend(feats: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor, ratio: float) -> torch.Tensor:
    """ratio * relu(W2 relu(W1 f)) + (1 - ratio) * f  (CLIP-Adapter, clip_adapter.py:138-172)."""
    feats, w1, w2 = (_dev(t, n, (torch.float32,)) for t, n in ((feats, "feats"), (w1, "w1"), (w2, "w2")))
    B, E = feats.shape
    H = w1.shape[0]
    if w1.shape != (H, E) or w2.shape != (E, H):
        raise ValueError("adapter_blend: adapter shapes do not chain")
    out = torch.empty_like(feats)
    check(lib.clipmi_adapter_blend(feats.data_ptr(), w1.data_ptr(), w2.data_ptr(), float(ratio), out.data_ptr(), B, E, H, _stream()),
          "clipmi_adapter_blend")
    return out


def scale_add(a: torch.Tensor, b: torch.Tensor, alpha: float) -> torch.Tensor:
    """a + alpha * b (TaskRes, taskres.py:105-106)."""
    a, b = _dev(a, "a", (torch.float32,)), _dev(b, "b", (torch.float32,))
    if a.shape != b.shape:
        raise ValueError("scale_add: shapes differ")
    out = torch.empty_like(a)
    check(lib.clipmi_scale_add(a.data_ptr(), b.data_ptr(), float(alpha), out.data_ptr(), a.numel(), _stream()), "clipmi_scale_add")
    return out
