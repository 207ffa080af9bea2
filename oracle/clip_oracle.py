"""CPU ORACLE for the CLIP inference-and-calibration hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU restatement (plain torch CPU ops / numpy, fp32 unless a
dtype is passed) of the algorithm the reference implements for SURVEY.md §8(a) rows a-1..a-14.
It is *not* part of the product: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it, and only as the checker / timed baseline.
The product path (``clip_calibration_amd``) never imports this module and has no CPU fallback.

Parity pin: the reference ships no tests or golden vectors so parity is pinned on
outputs of the reference itself, produced in the build container by ``oracle/gen_golden.py``
(which imports /root/reference/clip/model.py, tools/metrics.py and
trainers/calibration/distanse_aware_calibration.py) and committed under ``tests/golden/``.
``tests/test_oracle_golden.py`` checks every function here against those fixtures.

Each function cites the reference file:line it follows.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# --------------------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------------------
def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    """fp32 statistics regardless of activation dtype, result cast back (clip/model.py:153-159)."""
    xf = x.float()
    mu = xf.mean(dim=-1, keepdim=True)
    var = ((xf - mu) ** 2).mean(dim=-1, keepdim=True)
    y = (xf - mu) * torch.rsqrt(var + eps) * w.float() + b.float()
    return y.to(x.dtype)


def quick_gelu(x: Tensor) -> Tensor:
    """x * sigmoid(1.702 x) (clip/model.py:162-164)."""
    return x * torch.sigmoid(1.702 * x)


def causal_mask(n: int) -> Tensor:
    """-inf strictly above the diagonal (clip/model.py:585-591)."""
    return torch.full((n, n), float("-inf")).triu_(1)


def multi_head_attention(x: Tensor, in_w: Tensor, in_b: Tensor, out_w: Tensor, out_b: Tensor,
                         n_head: int, mask: Optional[Tensor]) -> Tensor:
    """Packed-in-proj self attention on NLD input; what nn.MultiheadAttention(x,x,x) computes at
    clip/model.py:181-183 (SURVEY a-5a): q,k,v = split(x W_in^T + b_in); softmax(q k^T / sqrt(hd) + mask) v;
    merge heads; W_out."""
    n, l, d = x.shape
    hd = d // n_head
    qkv = x @ in_w.t() + in_b
    q, k, v = qkv.split(d, dim=-1)
    q = q.reshape(n, l, n_head, hd).transpose(1, 2)
    k = k.reshape(n, l, n_head, hd).transpose(1, 2)
    v = v.reshape(n, l, n_head, hd).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    if mask is not None:
        s = s + mask.to(s.dtype)
    p = torch.softmax(s.float(), dim=-1).to(x.dtype)
    o = (p @ v).transpose(1, 2).reshape(n, l, d)
    return o @ out_w.t() + out_b


def residual_block(x: Tensor, sd: SD, prefix: str, n_head: int, mask: Optional[Tensor]) -> Tensor:
    """x += MHA(LN1 x); x += c_proj(QuickGELU(c_fc(LN2 x)))  (clip/model.py:185-188)."""
    g = lambda k: sd[prefix + k].to(x.dtype)  # noqa: E731
    h = layer_norm(x, sd[prefix + "ln_1.weight"], sd[prefix + "ln_1.bias"])
    x = x + multi_head_attention(h, g("attn.in_proj_weight"), g("attn.in_proj_bias"),
                                 g("attn.out_proj.weight"), g("attn.out_proj.bias"), n_head, mask)
    h = layer_norm(x, sd[prefix + "ln_2.weight"], sd[prefix + "ln_2.bias"])
    h = quick_gelu(h @ g("mlp.c_fc.weight").t() + g("mlp.c_fc.bias"))
    x = x + (h @ g("mlp.c_proj.weight").t() + g("mlp.c_proj.bias"))
    return x


def _h(t: Tensor, dtype: torch.dtype) -> Tensor:
    """MaPLe prompt tokens pass through ``.half()`` before being spliced in (clip/model.py:306,323,459)."""
    return t.half().to(dtype)


def _n_layers(sd: SD, prefix: str) -> int:
    return len([k for k in sd if k.startswith(prefix + "resblocks.") and k.endswith(".attn.in_proj_weight")])


# --------------------------------------------------------------------------------------
# image tower (a-1 .. a-6, a-9 image side)
# --------------------------------------------------------------------------------------
def patch_embed(image: Tensor, conv_w: Tensor) -> Tensor:
    """Conv2d(3,W,k=p,s=p,bias=False) as a per-patch GEMM; output [B, grid*grid, W] row-major over (py,px)
    (clip/model.py:369,395-397).  K index = c*p*p + ky*p + kx."""
    b, c, r, _ = image.shape
    w, _, p, _ = conv_w.shape
    g = r // p
    patches = image.reshape(b, c, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(b, g * g, c * p * p)
    return patches @ conv_w.reshape(w, c * p * p).to(image.dtype).t()


def encode_image(sd: SD, image: Tensor, dtype: torch.dtype = torch.float32,
                 shared_ctx: Optional[Tensor] = None,
                 deep_prompts: Optional[Sequence[Tensor]] = None) -> Tensor:
    """VisionTransformer.forward (clip/model.py:394-424); with ``shared_ctx``/``deep_prompts`` the MaPLe
    variant (clip/model.py:447-478, block :287-312): shared_ctx [n_ctx,W] is appended after pos-emb, and for
    layers 1..len(deep_prompts) the last n_ctx tokens are overwritten by deep_prompts[layer-1]."""
    x = patch_embed(image.to(dtype), sd["visual.conv1.weight"])
    b = x.shape[0]
    cls = sd["visual.class_embedding"].to(dtype).expand(b, 1, -1)
    x = torch.cat([cls, x], dim=1) + sd["visual.positional_embedding"].to(dtype)
    n_ctx = 0
    if shared_ctx is not None:
        n_ctx = shared_ctx.shape[0]
        x = torch.cat([x, _h(shared_ctx, dtype).expand(b, -1, -1)], dim=1)
    x = layer_norm(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"])
    n_head = sd["visual.conv1.weight"].shape[0] // 64
    counter = 0
    for i in range(_n_layers(sd, "visual.transformer.")):
        if i > 0 and deep_prompts is not None and counter < len(deep_prompts):
            x = torch.cat([x[:, : x.shape[1] - n_ctx], _h(deep_prompts[counter], dtype).expand(b, -1, -1)], dim=1)
            counter += 1
        x = residual_block(x, sd, f"visual.transformer.resblocks.{i}.", n_head, None)
    x = layer_norm(x[:, 0, :], sd["visual.ln_post.weight"], sd["visual.ln_post.bias"])
    return x @ sd["visual.proj"].to(dtype)


def encode_image_resnet(sd: SD, image: Tensor, dtype: torch.dtype = torch.float32) -> Tensor:
    """ModifiedResNet.forward (clip/model.py:93-150): 3-conv stem (+BatchNorm, ReLU) and AvgPool2d(2); four stages of
    Bottlenecks (:10-56: 1x1 -> 3x3 -> AvgPool2d(stride) -> 1x1, BatchNorm after each conv, ReLU after the first two and after
    the residual add; the shortcut is AvgPool2d(stride) -> 1x1 -> BatchNorm when shapes change); AttentionPool2d (:58-90):
    tokens [mean | HW] + positional embedding, multi-head attention with the mean token as the query, c_proj."""
    import torch.nn.functional as F

    def bn(x, p):
        return F.batch_norm(x, sd[p + ".running_mean"].to(dtype), sd[p + ".running_var"].to(dtype), sd[p + ".weight"].to(dtype),
                            sd[p + ".bias"].to(dtype), False, 0.0, 1e-5)

    def conv(x, p, stride=1, pad=0):
        return F.conv2d(x, sd[p + ".weight"].to(dtype), None, stride, pad)

    x = image.to(dtype)
    x = F.relu(bn(conv(x, "visual.conv1", 2, 1), "visual.bn1"))
    x = F.relu(bn(conv(x, "visual.conv2", 1, 1), "visual.bn2"))
    x = F.relu(bn(conv(x, "visual.conv3", 1, 1), "visual.bn3"))
    x = F.avg_pool2d(x, 2)
    for li, stride0 in zip((1, 2, 3, 4), (1, 2, 2, 2)):
        bi = 0
        while f"visual.layer{li}.{bi}.conv1.weight" in sd:
            p = f"visual.layer{li}.{bi}"
            stride = stride0 if bi == 0 else 1
            out = F.relu(bn(conv(x, p + ".conv1"), p + ".bn1"))
            out = F.relu(bn(conv(out, p + ".conv2", 1, 1), p + ".bn2"))
            if stride > 1:
                out = F.avg_pool2d(out, stride)
            out = bn(conv(out, p + ".conv3"), p + ".bn3")
            identity = x
            if p + ".downsample.0.weight" in sd:
                identity = bn(conv(F.avg_pool2d(x, stride) if stride > 1 else x, p + ".downsample.0"), p + ".downsample.1")
            x = F.relu(out + identity)
            bi += 1
    b, c, h, w = x.shape
    tok = x.reshape(b, c, h * w).permute(0, 2, 1)                                    # [B, HW, C]
    tok = torch.cat([tok.mean(dim=1, keepdim=True), tok], dim=1) + sd["visual.attnpool.positional_embedding"].to(dtype)
    ap = "visual.attnpool."
    heads = c // 64
    q = tok[:, :1] @ sd[ap + "q_proj.weight"].to(dtype).t() + sd[ap + "q_proj.bias"].to(dtype)
    k = tok @ sd[ap + "k_proj.weight"].to(dtype).t() + sd[ap + "k_proj.bias"].to(dtype)
    v = tok @ sd[ap + "v_proj.weight"].to(dtype).t() + sd[ap + "v_proj.bias"].to(dtype)
    t = tok.shape[1]
    qh = q.reshape(b, 1, heads, 64).transpose(1, 2) * 64 ** -0.5
    kh, vh = k.reshape(b, t, heads, 64).transpose(1, 2), v.reshape(b, t, heads, 64).transpose(1, 2)
    att = torch.softmax(qh @ kh.transpose(-1, -2), dim=-1) @ vh                       # [B, heads, 1, 64]
    out = att.transpose(1, 2).reshape(b, c)
    return out @ sd[ap + "c_proj.weight"].to(dtype).t() + sd[ap + "c_proj.bias"].to(dtype)


# --------------------------------------------------------------------------------------
# text tower (a-7, a-8, a-9 text side)
# --------------------------------------------------------------------------------------
def text_transformer(sd: SD, x: Tensor, deep_prompts: Optional[Sequence[Tensor]] = None, n_ctx: int = 0) -> Tensor:
    """The 12 causal blocks on NLD embeddings (clip/model.py:334-359 with mask :585-591).  MaPLe text side
    (clip/model.py:313-328): for layers 1..len(deep_prompts) tokens 1..1+n_ctx are overwritten."""
    l = x.shape[1]
    mask = causal_mask(l)
    n_head = sd["ln_final.weight"].shape[0] // 64
    counter = 0
    for i in range(_n_layers(sd, "transformer.")):
        if i > 0 and deep_prompts is not None and counter < len(deep_prompts):
            ctx = _h(deep_prompts[counter], x.dtype).expand(x.shape[0], -1, -1)
            x = torch.cat([x[:, :1], ctx, x[:, 1 + n_ctx:]], dim=1)
            counter += 1
        x = residual_block(x, sd, f"transformer.resblocks.{i}.", n_head, mask)
    return x


def text_encoder(sd: SD, prompts: Tensor, tokenized: Tensor, dtype: torch.dtype = torch.float32,
                 deep_prompts: Optional[Sequence[Tensor]] = None, n_ctx: int = 0) -> Tensor:
    """TextEncoder.forward (trainers/classification/coop.py:56-67; MaPLe: maple.py:60-74): embeddings
    [C,77,D] + pos -> blocks -> ln_final -> EOT row (argmax of ids) -> @ text_projection."""
    x = prompts.to(dtype) + sd["positional_embedding"].to(dtype)
    x = text_transformer(sd, x, deep_prompts, n_ctx)
    x = layer_norm(x, sd["ln_final.weight"], sd["ln_final.bias"])
    eot = tokenized.argmax(dim=-1)
    return x[torch.arange(x.shape[0]), eot] @ sd["text_projection"].to(dtype)


def encode_text(sd: SD, ids: Tensor, dtype: torch.dtype = torch.float32) -> Tensor:
    """CLIP.encode_text (clip/model.py:600-613): token_embedding gather then the text encoder."""
    emb = sd["token_embedding.weight"][ids].to(dtype)
    return text_encoder(sd, emb, ids, dtype)


def ivlp_prompts(sd: SD):
    """Prompt tokens an IVLP / VPT model carries in its own state_dict (clip/model.py:191-226, 361-381): returns
    (vision shallow [n,W] or None, vision deep list for blocks 1.., text deep list for blocks 1.., text n_ctx).  Block i's
    ``VPT_shallow`` replaces the previous block's prompt tokens before block i runs -- the same splice as MaPLe's deep
    prompts (clip/model.py:231-252 vs :301-328), so the tower functions above take them unchanged."""
    def per_block(prefix):
        out, i = [], 1
        while f"{prefix}resblocks.{i}.VPT_shallow" in sd:
            out.append(sd[f"{prefix}resblocks.{i}.VPT_shallow"])
            i += 1
        return out
    deep_t = per_block("transformer.")
    return sd.get("visual.VPT"), per_block("visual.transformer."), deep_t, (deep_t[0].shape[0] if deep_t else 0)


def encode_image_ivlp(sd: SD, image: Tensor, dtype: torch.dtype = torch.float32) -> Tensor:
    """VisionTransformer.forward of an IVLP / VPT model (clip/model.py:394-424 with VPT_shallow = True)."""
    shallow, deep_v, _, _ = ivlp_prompts(sd)
    return encode_image(sd, image, dtype, shallow, deep_v if shallow is not None else None)


def encode_text_ivlp(sd: SD, ids: Tensor, dtype: torch.dtype = torch.float32) -> Tensor:
    """CLIP.encode_text of an IVLP model: the text blocks overwrite tokens 1..n_ctx from block 1 on (clip/model.py:240-252)."""
    _, _, deep_t, n_ctx = ivlp_prompts(sd)
    emb = sd["token_embedding.weight"][ids].to(dtype)
    return text_encoder(sd, emb, ids, dtype, deep_t or None, n_ctx)


def coop_prompts(sd: SD, ids: Tensor, ctx: Tensor, dtype: torch.dtype = torch.float32) -> Tensor:
    """PromptLearner.forward, class_token_position == 'end' (coop.py:128-144; maple.py:170-176):
    [SOS embedding | ctx (shared or per-class) | embeddings of the tokens after the n_ctx placeholders]."""
    emb = sd["token_embedding.weight"][ids].to(dtype)
    n_ctx = ctx.shape[-2]
    c = ctx.to(dtype)
    if c.dim() == 2:
        c = c.unsqueeze(0).expand(ids.shape[0], -1, -1)
    return torch.cat([emb[:, :1], c, emb[:, 1 + n_ctx:]], dim=1)


def cocoop_forward(sd: SD, pl: SD, image: Tensor, ids: Tensor, dtype: torch.dtype = torch.float32):
    """CoCoOp CustomCLIP.forward in eval mode (trainers/classification/cocoop.py:154-199).  ``pl`` holds ``ctx`` [n_ctx,D] and
    ``meta_net.linear{1,2}.{weight,bias}``; ids are those of "X X .. X name." prompts.  Per image: ctx + meta_net(f) is
    spliced into the C prompts, the text tower runs on them, the features are normalised and dotted with that image.
    Returns (logits [B,C], image_features_n [B,E], per-image un-normalised text features [B,C,E])."""
    f = l2_normalize(encode_image(sd, image, dtype))
    hid = torch.relu(f @ pl["meta_net.linear1.weight"].to(dtype).t() + pl["meta_net.linear1.bias"].to(dtype))
    bias = hid @ pl["meta_net.linear2.weight"].to(dtype).t() + pl["meta_net.linear2.bias"].to(dtype)
    scale = sd["logit_scale"].exp()
    logits, feats = [], []
    for b in range(f.shape[0]):
        prompts = coop_prompts(sd, ids, pl["ctx"].to(dtype) + bias[b], dtype)
        tf = text_encoder(sd, prompts, ids, dtype)
        feats.append(tf)
        logits.append(scale * f[b] @ l2_normalize(tf).t())
    return torch.stack(logits), f, torch.stack(feats)


def proda_classifier(sd: SD, ids: Tensor, ctx: Tensor, dtype: torch.dtype = torch.float32) -> Tensor:
    """ProDA's set_classifier (trainers/classification/proda.py:146-222, 316-333).  ctx [P, n_ctx, D]; prompt p of every class
    places the class-name tokens in FRONT of the context for p < P//4, in the MIDDLE (after n_ctx//2 context tokens) for
    P//4 <= p < P//2, and after the whole context otherwise (proda.py:110-114); ids are "X*n_ctx name ." prompts.  Every
    prompt goes through the text tower, is L2-normalised, and the class classifier is the plain mean over its P prompts.
    (The reference orders a class's prompts end|middle|front before averaging; the mean does not depend on it.)
    Pinning: proda.py imports dassl and cannot run here, so no reference-generated fixture covers the prompt ASSEMBLY;
    the text tower underneath is pinned by tests/golden/tiny*_clip.npz."""
    P, n_ctx, _ = ctx.shape
    emb = sd["token_embedding.weight"][ids].to(dtype)
    eot = ids.argmax(dim=-1)
    half = n_ctx // 2
    out = []
    for c in range(ids.shape[0]):
        nl = int(eot[c]) - n_ctx - 2
        sos, name, rest = emb[c, :1], emb[c, 1 + n_ctx: 1 + n_ctx + nl], emb[c, 1 + n_ctx + nl:]
        feats = []
        for p in range(P):
            cp = ctx[p].to(dtype)
            if P > 1 and p < P // 4:
                seq = torch.cat([sos, name, cp, rest])
            elif P > 1 and p < 2 * (P // 4):
                seq = torch.cat([sos, cp[:half], name, cp[half:], rest])
            else:
                seq = torch.cat([sos, cp, name, rest])
            feats.append(seq)
        tf = text_encoder(sd, torch.stack(feats), ids[c:c + 1].expand(P, -1), dtype)
        out.append(l2_normalize(tf).mean(dim=0))
    return torch.stack(out)


def maple_prompt_learner(sd: SD, ids: Tensor, pl: SD, dtype: torch.dtype = torch.float32):
    """MultiModalPromptLearner.forward (maple.py:170-187): returns (prompts, shared_ctx = proj(ctx),
    deep text prompts, deep visual prompts = per-depth Linear(512->768) of the text prompts)."""
    ctx = pl["ctx"].to(dtype)
    prompts = coop_prompts(sd, ids, ctx, dtype)
    shared = ctx @ pl["proj.weight"].to(dtype).t() + pl["proj.bias"].to(dtype)
    depth = len([k for k in pl if k.startswith("compound_prompts_text.")])
    deep_t = [pl[f"compound_prompts_text.{i}"].to(dtype) for i in range(depth)]
    deep_v = [deep_t[i] @ pl[f"compound_prompt_projections.{i}.weight"].to(dtype).t()
              + pl[f"compound_prompt_projections.{i}.bias"].to(dtype) for i in range(depth)]
    return prompts, shared, deep_t, deep_v


# --------------------------------------------------------------------------------------
# logits, DAC, softmax, ECE (a-10 .. a-12)
# --------------------------------------------------------------------------------------
def l2_normalize(f: Tensor) -> Tensor:
    """f / ||f||_2 per row (zsclip.py:99, coop.py:212-213)."""
    return f / f.norm(dim=-1, keepdim=True)


def clip_logits(image_features: Tensor, text_features: Tensor, scale: float | Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """(s * img_n) @ txt_n^T -- the scale multiplies the image features FIRST (zsclip.py:98-101,
    coop.py:212-217).  ``scale`` is exp(logit_scale) for classification trainers, 1.0 for the
    calibration base models (base_model/coop.py:222-224), the learnt exp(logit_scale) in
    CustomCLIPCalibration.forward (tempscaling.py:53-56).  Returns the trainer 3-tuple."""
    img_n = l2_normalize(image_features)
    txt_n = l2_normalize(text_features)
    return (scale * img_n) @ txt_n.t(), img_n, txt_n


def dac_fit(base_zs: np.ndarray, cur_zs: np.ndarray, base_tuned: np.ndarray, cur_tuned: np.ndarray, k: int) -> np.ndarray:
    """DistanseAwareCalibration.fit (distanse_aware_calibration.py:13-46): per current class,
    score = exp(-mean of the k smallest L2 distances to the base-class text features), for the zero-shot and the
    tuned feature sets; confidence = 1.0 if the nearest tuned distance < 0.05 else tuned_score / zs_score."""
    conf = np.empty(cur_zs.shape[0], dtype=np.float64)
    for i in range(cur_zs.shape[0]):
        d_zs = np.sort(np.linalg.norm(base_zs - cur_zs[i], axis=1))[:k]
        d_fs = np.sort(np.linalg.norm(base_tuned - cur_tuned[i], axis=1))[:k]
        zs_score = np.exp(-np.sum(d_zs) / k)
        fs_score = np.exp(-np.sum(d_fs) / k)
        conf[i] = 1.0 if d_fs[0] < 0.05 else fs_score / zs_score
    return conf


def dac_predict(logits: np.ndarray, class_confidence: np.ndarray) -> np.ndarray:
    """DistanseAwareCalibration.predict (distanse_aware_calibration.py:49-58): fp32; each sample's logits are
    multiplied by the confidence of its arg-max class."""
    lg = np.asarray(logits).astype(np.float32)
    cc = np.asarray(class_confidence).astype(np.float32)
    pred = lg.argmax(axis=1)
    return lg * cc[pred][:, None]


def softmax_probs(logits: np.ndarray) -> np.ndarray:
    """scipy.special.softmax(logits, axis=-1) in the input's dtype (vl_calibrator.py:91)."""
    lg = np.asarray(logits)
    m = lg.max(axis=-1, keepdims=True)
    e = np.exp(lg - m)
    return e / e.sum(axis=-1, keepdims=True)


def conf_pred(probs: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """preds = argmax, confs = probs[i, pred] (vl_evaluator.py:68, 83)."""
    preds = np.argmax(probs, axis=1)
    return probs[np.arange(probs.shape[0]), preds], preds


def ece(conf: np.ndarray, pred: np.ndarray, gt: np.ndarray, n_bins: int = 10) -> float:
    """tools/metrics.py:90-130.  Equal-width bins; bin id = digitize(conf, linspace(0,1,n+1)) - 1; per-bin mean
    accuracy / confidence over the members (0 when empty); weights from np.histogram.  Quirk reproduced: a
    confidence of exactly 1.0 digitizes to bin n (not averaged into any bin) while np.histogram's closed last
    edge counts it in bin n-1's weight."""
    conf = np.asarray(conf)
    pred = np.asarray(pred)
    gt = np.asarray(gt)
    edges = np.linspace(0, 1, n_bins + 1)
    which = np.digitize(conf, edges) - 1
    acc = np.zeros(n_bins)
    avg = np.zeros(n_bins)
    for b in range(n_bins):
        sel = which == b
        if sel.sum() > 0:
            acc[b] = np.mean(gt[sel] == pred[sel])
            avg[b] = np.mean(conf[sel])
    weights = np.histogram(conf, edges)[0] / len(conf)
    return float(np.sum(weights * np.abs(avg - acc)))


def mce(conf: np.ndarray, pred: np.ndarray, gt: np.ndarray, n_bins: int = 10) -> float:
    """tools/metrics.py:181-208 (the reference's weighted "maximal" calibration error): bins =
    digitize(conf, linspace(0,1,n+1)[1:-1]); max over non-empty bins of |mean correct - mean conf| * count / N."""
    conf = np.asarray(conf, dtype=np.float64)
    correct = (np.asarray(pred) == np.asarray(gt)).astype(np.float64)
    which = np.digitize(conf, np.linspace(0, 1, n_bins + 1)[1:-1])
    best = 0.0
    for b in np.unique(which):
        sel = which == b
        best = max(best, abs(correct[sel].mean() - conf[sel].mean()) * sel.sum() / len(conf))
    return float(best)


def quantile_bins(x: np.ndarray, n_bins: int) -> np.ndarray:
    """sklearn KBinsDiscretizer(n_bins, encode='ordinal', strategy='quantile').fit_transform as the reference calls it
    (tools/metrics.py:152, 228; sklearn is a third-party dependency, unpinned in requirements.txt, 1.7.2 in this image):
    edges = np.percentile(x, linspace(0, 100, n+1)) with linear interpolation; an edge within 1e-8 of its predecessor is
    removed; the ordinal code is searchsorted(inner edges, x, side='right'); a constant column maps to 0."""
    x = np.asarray(x)
    x = x if x.dtype in (np.float32, np.float64) else x.astype(np.float64)
    if x.min() == x.max():
        return np.zeros(len(x), dtype=np.int64)
    edges = [float(e) for e in np.percentile(x, np.linspace(0, 100, n_bins + 1))]
    kept = [edges[0]]
    for prev, e in zip(edges[:-1], edges[1:]):
        if e - prev > 1e-8:
            kept.append(e)
    inner = np.array(kept[1:-1], dtype=np.float64)
    return np.array([int(np.sum(inner <= v)) for v in x], dtype=np.int64)


def _group_gap(keys: List, conf: np.ndarray, correct: np.ndarray) -> float:
    """sum_g |mean(correct_g) - mean(conf_g)| * |g| / N over the distinct keys (the pandas groupby of metrics.py:159-162)."""
    total = 0.0
    for key in set(keys):
        sel = np.array([k == key for k in keys])
        total += abs(correct[sel].mean() - conf[sel].mean()) * sel.sum() / len(conf)
    return float(total)


def ace(conf: np.ndarray, pred: np.ndarray, gt: np.ndarray, n_bins: int = 10) -> float:
    """AdaptiveECE, tools/metrics.py:212-236: equal-mass bins of the confidence."""
    conf = np.asarray(conf)
    correct = (np.asarray(pred) == np.asarray(gt)).astype(np.float64)
    return _group_gap(list(quantile_bins(conf, n_bins)), conf.astype(np.float64), correct)


def piece(conf: np.ndarray, proximity: np.ndarray, pred: np.ndarray, gt: np.ndarray, dist_bins: int = 10, conf_bins: int = 10) -> float:
    """PIECE, tools/metrics.py:132-178: groups = (quantile bin of the proximity, digitize(conf, linspace(0,1,n+1)[1:-1]))."""
    conf = np.asarray(conf)
    correct = (np.asarray(pred) == np.asarray(gt)).astype(np.float64)
    kb = quantile_bins(np.asarray(proximity), dist_bins)
    cb = np.digitize(conf, np.linspace(0, 1, conf_bins + 1)[1:-1])
    return _group_gap(list(zip(kb.tolist(), cb.tolist())), conf.astype(np.float64), correct)


def macro_f1(pred: np.ndarray, gt: np.ndarray) -> float:
    """sklearn f1_score(labels, preds, average='macro', labels=np.unique(labels)) (vl_evaluator.py:77-82): the plain mean
    over classes present in gt of 2tp / (2tp + fp + fn), 0 where that denominator is 0."""
    pred, gt = np.asarray(pred), np.asarray(gt)
    scores = []
    for c in np.unique(gt):
        tp = np.sum((pred == c) & (gt == c))
        fp = np.sum((pred == c) & (gt != c))
        fn = np.sum((pred != c) & (gt == c))
        scores.append(2 * tp / (2 * tp + fp + fn) if (2 * tp + fp + fn) > 0 else 0.0)
    return float(np.mean(scores))


def knn_dists(val_base_class_features: np.ndarray, image_features_cur: np.ndarray, k: int) -> np.ndarray:
    """get_knn_dists (trainers/calibration/proximity.py:19-46): per query, the k smallest L2 distances to the reference
    rows, ascending (fp32, like the torch tensors the reference builds)."""
    refs = np.asarray(val_base_class_features, dtype=np.float32)
    q = np.asarray(image_features_cur, dtype=np.float32)
    out = np.empty((q.shape[0], k), dtype=np.float32)
    for i in range(q.shape[0]):
        out[i] = np.sort(np.linalg.norm(refs - q[i], axis=1))[:k]
    return out


def val_image_knn_dists(image_features_cur: np.ndarray, k: int) -> np.ndarray:
    """get_val_image_knn_dists (proximity.py:49-70): k+1 nearest within the set itself, the first (self) dropped."""
    return knn_dists(image_features_cur, image_features_cur, k + 1)[:, 1:]


def tokenize_ids(token_lists: List[List[int]], sot: int, eot: int, context_length: int = 77) -> np.ndarray:
    """clip.tokenize's packing (clip/clip.py:207-224): [SOT] + bpe + [EOT], zero padded to 77, error if too long."""
    out = np.zeros((len(token_lists), context_length), dtype=np.int64)
    for i, t in enumerate(token_lists):
        toks = [sot] + list(t) + [eot]
        if len(toks) > context_length:
            raise RuntimeError(f"Input {i} is too long for context length {context_length}")
        out[i, : len(toks)] = toks
    return out


# --------------------------------------------------------------------------------------
# whole-path drivers (what the trainers' model_inference returns, then calibration + metric)
# --------------------------------------------------------------------------------------
def zeroshot_inference(sd: SD, image: Tensor, text_features_n: Tensor, dtype: torch.dtype = torch.float32):
    """ZeroshotCLIP.model_inference (zsclip.py:97-102) with pre-normalised text features (zsclip.py:90-92)."""
    f = encode_image(sd, image, dtype)
    img_n = l2_normalize(f)
    logits = (sd["logit_scale"].exp().to(dtype) * img_n) @ text_features_n.t()
    return logits, img_n, text_features_n


def calibrated_ece(logits: np.ndarray, labels: np.ndarray, class_confidence: Optional[np.ndarray] = None,
                   n_bins: int = 10) -> Tuple[float, np.ndarray, np.ndarray]:
    """base_learner.py:141-144 -> VLCalibration.predict (vl_calibrator.py:83-109, DAC -> softmax branch) ->
    VLClassification.evaluate (vl_evaluator.py:59-92, ECE only).  logits arrive as float64 because the
    evaluator accumulates python lists (vl_evaluator.py:47)."""
    lg = np.asarray(logits, dtype=np.float64)
    if class_confidence is not None:
        lg = dac_predict(lg, class_confidence)
    probs = softmax_probs(lg)
    c, p = conf_pred(probs)
    return ece(c, p, np.asarray(labels), n_bins), c, p
