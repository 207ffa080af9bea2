"""Seeded synthetic CLIP checkpoints and inputs.

No OpenAI checkpoint is reachable offline, so every test, fixture and bench in
this repo runs on a *synthetic* state_dict that has the OpenAI key names and
shapes (the reference infers geometry from these shapes in
``clip/model.py:656-681``).  Values are rounded to fp16-representable numbers,
as the OpenAI JIT archives store fp16.

This module is host-side plumbing shared by the product path, the oracle, the
tests and ``bench.py``; it contains no model arithmetic.
"""
from __future__ import annotations

import dataclasses
import math
from typing import Dict, Optional

import torch


@dataclasses.dataclass(frozen=True)
class ClipGeometry:
    embed_dim: int
    image_resolution: int
    vision_layers: int
    vision_width: int
    vision_patch_size: int
    context_length: int
    vocab_size: int
    transformer_width: int
    transformer_heads: int
    transformer_layers: int

    @property
    def grid(self) -> int:
        return self.image_resolution // self.vision_patch_size

    @property
    def vision_tokens(self) -> int:
        return self.grid * self.grid + 1

    @property
    def vision_heads(self) -> int:
        return self.vision_width // 64


GEOMETRIES: Dict[str, ClipGeometry] = {
    # reference table clip/clip.py:29-39 (names only; shapes are the published CLIP ones)
    "ViT-B/16": ClipGeometry(512, 224, 12, 768, 16, 77, 49408, 512, 8, 12),
    "ViT-B/32": ClipGeometry(512, 224, 12, 768, 32, 77, 49408, 512, 8, 12),
    "ViT-L/14": ClipGeometry(768, 224, 24, 1024, 14, 77, 49408, 768, 12, 12),
    "ViT-L/14@336px": ClipGeometry(768, 336, 24, 1024, 14, 77, 49408, 768, 12, 12),
    # RN50 (clip/clip.py:30): ModifiedResNet (3, 4, 6, 3) x width 64 image tower -- the ViT fields are placeholders; text tower 512 / 8 / 12
    "RN50": ClipGeometry(1024, 224, 1, 64, 224, 77, 49408, 512, 8, 12),
    # committed-fixture geometry: 2 layers, 4x4 grid, 2 heads; EOT id = vocab-1
    "tiny": ClipGeometry(128, 64, 2, 128, 16, 77, 256, 128, 2, 2),
    # odd token count + 3 layers: exercises M-edge handling in the GEMM tiles
    "tiny3": ClipGeometry(64, 48, 3, 192, 16, 77, 512, 64, 1, 3),
}


def resnet_config_from_state_dict(sd: Dict[str, torch.Tensor]) -> Dict:
    """ModifiedResNet branch of the reference's shape inference (clip/model.py:666-672)."""
    if "visual.layer1.0.conv1.weight" not in sd or "visual.attnpool.positional_embedding" not in sd:
        raise ValueError("checkpoint has neither visual.proj (ViT) nor visual.layer1 / visual.attnpool (ModifiedResNet) keys")
    layers = tuple(len(set(k.split(".")[2] for k in sd if k.startswith(f"visual.layer{b}"))) for b in (1, 2, 3, 4))
    width = sd["visual.layer1.0.conv1.weight"].shape[0]
    out_w = round((sd["visual.attnpool.positional_embedding"].shape[0] - 1) ** 0.5)
    if out_w ** 2 + 1 != sd["visual.attnpool.positional_embedding"].shape[0]:
        raise ValueError("ModifiedResNet: attnpool.positional_embedding is not (w*w + 1) rows")
    return {"layers": layers, "width": int(width), "image_resolution": out_w * 32}


def geometry_from_state_dict(sd: Dict[str, torch.Tensor]) -> ClipGeometry:
    """Shape inference, same rules as reference ``build_model`` (clip/model.py:657-681)."""
    tw = sd["ln_final.weight"].shape[0]
    if "visual.proj" not in sd:     # ModifiedResNet image tower: the ViT fields are placeholders the RN path never reads
        rn = resnet_config_from_state_dict(sd)
        return ClipGeometry(embed_dim=sd["text_projection"].shape[1], image_resolution=rn["image_resolution"], vision_layers=1,
                            vision_width=64, vision_patch_size=rn["image_resolution"], context_length=sd["positional_embedding"].shape[0],
                            vocab_size=sd["token_embedding.weight"].shape[0], transformer_width=tw, transformer_heads=tw // 64,
                            transformer_layers=len(set(k.split(".")[2] for k in sd if k.startswith("transformer.resblocks"))))
    vw = sd["visual.conv1.weight"].shape[0]
    vl = len([k for k in sd if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
    ps = sd["visual.conv1.weight"].shape[-1]
    grid = round((sd["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    tl = len(set(k.split(".")[2] for k in sd if k.startswith("transformer.resblocks")))
    return ClipGeometry(
        embed_dim=sd["text_projection"].shape[1],
        image_resolution=ps * grid,
        vision_layers=vl,
        vision_width=vw,
        vision_patch_size=ps,
        context_length=sd["positional_embedding"].shape[0],
        vocab_size=sd["token_embedding.weight"].shape[0],
        transformer_width=tw,
        transformer_heads=tw // 64,
        transformer_layers=tl,
    )


def _r16(t: torch.Tensor) -> torch.Tensor:
    """Round to the nearest fp16-representable value, keep fp32 storage."""
    return t.half().float()


def synthetic_state_dict(geom: ClipGeometry | str = "ViT-B/16", seed: int = 0,
                         logit_scale: float = 4.6052, gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Seeded random checkpoint with OpenAI key names (302 keys for ViT-B/16).

    ``gain`` scales the residual-branch output projections; >1 gives the
    "stress" fixture with a large-magnitude residual stream (SURVEY §7 risks).
    """
    if isinstance(geom, str):
        geom = GEOMETRIES[geom]
    g = torch.Generator().manual_seed(seed)

    def randn(*shape, std=1.0):
        return _r16(torch.randn(*shape, generator=g) * std)

    sd: Dict[str, torch.Tensor] = {}

    def tower(prefix: str, width: int, layers: int):
        attn_std = width ** -0.5
        proj_std = (width ** -0.5) * ((2 * layers) ** -0.5) * gain
        fc_std = (2 * width) ** -0.5
        for i in range(layers):
            p = f"{prefix}resblocks.{i}."
            sd[p + "ln_1.weight"] = _r16(1.0 + 0.1 * torch.randn(width, generator=g))
            sd[p + "ln_1.bias"] = randn(width, std=0.05)
            sd[p + "attn.in_proj_weight"] = randn(3 * width, width, std=attn_std)
            sd[p + "attn.in_proj_bias"] = randn(3 * width, std=0.02)
            sd[p + "attn.out_proj.weight"] = randn(width, width, std=proj_std)
            sd[p + "attn.out_proj.bias"] = randn(width, std=0.02)
            sd[p + "ln_2.weight"] = _r16(1.0 + 0.1 * torch.randn(width, generator=g))
            sd[p + "ln_2.bias"] = randn(width, std=0.05)
            sd[p + "mlp.c_fc.weight"] = randn(4 * width, width, std=fc_std)
            sd[p + "mlp.c_fc.bias"] = randn(4 * width, std=0.02)
            sd[p + "mlp.c_proj.weight"] = randn(width, 4 * width, std=proj_std)
            sd[p + "mlp.c_proj.bias"] = randn(width, std=0.02)

    vw, ps = geom.vision_width, geom.vision_patch_size
    scale = vw ** -0.5
    sd["visual.conv1.weight"] = randn(vw, 3, ps, ps, std=(3 * ps * ps) ** -0.5)
    sd["visual.class_embedding"] = randn(vw, std=scale)
    sd["visual.positional_embedding"] = randn(geom.vision_tokens, vw, std=scale)
    sd["visual.ln_pre.weight"] = _r16(1.0 + 0.1 * torch.randn(vw, generator=g))
    sd["visual.ln_pre.bias"] = randn(vw, std=0.05)
    tower("visual.transformer.", vw, geom.vision_layers)
    sd["visual.ln_post.weight"] = _r16(1.0 + 0.1 * torch.randn(vw, generator=g))
    sd["visual.ln_post.bias"] = randn(vw, std=0.05)
    sd["visual.proj"] = randn(vw, geom.embed_dim, std=scale)

    tw = geom.transformer_width
    sd["token_embedding.weight"] = randn(geom.vocab_size, tw, std=0.02)
    sd["positional_embedding"] = randn(geom.context_length, tw, std=0.01)
    tower("transformer.", tw, geom.transformer_layers)
    sd["ln_final.weight"] = _r16(1.0 + 0.1 * torch.randn(tw, generator=g))
    sd["ln_final.bias"] = randn(tw, std=0.05)
    sd["text_projection"] = randn(tw, geom.embed_dim, std=tw ** -0.5)
    sd["logit_scale"] = torch.tensor(float(logit_scale))
    return sd


def outlier_state_dict(geom: ClipGeometry | str = "ViT-B/16", seed: int = 0, logit_scale: float = 4.6052,
                       n_outlier: int = 4, outlier_gain: float = 96.0, offset: float = 2.5) -> Dict[str, torch.Tensor]:
    """``synthetic_state_dict`` reshaped to the activation statistics of trained CLIP towers (SURVEY §7 risk: a handful of
    "massive" residual channels, LayerNorm gains spread over an order of magnitude, a common offset in the stream):

    * ``n_outlier`` residual channels per tower: their rows of ``attn.out_proj`` / ``mlp.c_proj`` in the first two blocks
      are scaled by ``outlier_gain`` and get biases of +-(6..12), so those channels of the residual stream sit 30-100x above
      the rest for the remaining depth; their LayerNorm gains are small (as in trained checkpoints), the others are
      log-uniform in [0.3, 3]; LayerNorm biases ~ N(0, 0.3);
    * ``ln_pre`` (image) / the positional embedding (text) add a common offset, so that mean^2 is comparable with E[x^2]
      in every row -- the cancellation a LayerNorm computed from (sum x, sum x^2) has to survive.

    Deterministic in (geom, seed); every tensor fp16-representable, like ``synthetic_state_dict``."""
    if isinstance(geom, str):
        geom = GEOMETRIES[geom]
    sd = synthetic_state_dict(geom, seed=seed, logit_scale=logit_scale)
    g = torch.Generator().manual_seed(7919 + seed)

    def tower(prefix: str, width: int, layers: int):
        ch = torch.randperm(width, generator=g)[:n_outlier]
        sign = torch.where(torch.rand(n_outlier, generator=g) < 0.5, -1.0, 1.0)
        for i in range(layers):
            p = f"{prefix}resblocks.{i}."
            for ln in ("ln_1", "ln_2"):
                gain = torch.exp(torch.empty(width).uniform_(-1.2, 1.1, generator=g))      # log-uniform in [0.3, 3]
                gain[ch] = 0.02 + 0.05 * torch.rand(n_outlier, generator=g)
                sd[p + ln + ".weight"] = _r16(gain)
                sd[p + ln + ".bias"] = _r16(0.3 * torch.randn(width, generator=g))
            if i < 2:
                for w, b in (("attn.out_proj.weight", "attn.out_proj.bias"), ("mlp.c_proj.weight", "mlp.c_proj.bias")):
                    sd[p + w][ch] = _r16(sd[p + w][ch] * outlier_gain)
                    sd[p + b][ch] = _r16(sign * (6.0 + 6.0 * torch.rand(n_outlier, generator=g)))
        return ch

    vch = tower("visual.transformer.", geom.vision_width, geom.vision_layers)
    tch = tower("transformer.", geom.transformer_width, geom.transformer_layers)
    sd["visual.ln_pre.bias"] = _r16(sd["visual.ln_pre.bias"] + offset)
    sd["positional_embedding"] = _r16(sd["positional_embedding"] + 0.04 * offset)   # token embeddings are ~0.02: the same ratio
    for name, ch in (("visual.ln_post", vch), ("ln_final", tch)):
        gain = sd[name + ".weight"].clone()
        gain[ch] = 0.05
        sd[name + ".weight"] = _r16(gain)
    return sd


def synthetic_resnet_state_dict(layers=(1, 1, 1, 1), width: int = 64, image_resolution: int = 64, text_geom: str = "tiny",
                                seed: int = 0, logit_scale: float = 4.6052) -> Dict[str, torch.Tensor]:
    """Seeded checkpoint with a ModifiedResNet image tower (key names of clip/model.py:10-150: ``visual.conv1.weight``,
    ``visual.bn1.running_mean``, ``visual.layer2.0.downsample.0.weight``, ``visual.attnpool.q_proj.bias`` ...) and the text tower
    of ``text_geom``.  BatchNorm running statistics are non-trivial so that the fold is exercised."""
    tg = GEOMETRIES[text_geom]
    sd = {k: v for k, v in synthetic_state_dict(tg, seed=seed, logit_scale=logit_scale).items() if not k.startswith("visual.")}
    g = torch.Generator().manual_seed(seed + 77)

    def conv(name, cout, cin, k):
        sd[name + ".weight"] = _r16(torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5)

    def bn(name, c):
        sd[name + ".weight"] = _r16(1.0 + 0.1 * torch.randn(c, generator=g))
        sd[name + ".bias"] = _r16(0.05 * torch.randn(c, generator=g))
        sd[name + ".running_mean"] = _r16(0.1 * torch.randn(c, generator=g))
        sd[name + ".running_var"] = _r16(1.0 + 0.2 * torch.rand(c, generator=g))
        sd[name + ".num_batches_tracked"] = torch.tensor(100, dtype=torch.long)

    conv("visual.conv1", width // 2, 3, 3); bn("visual.bn1", width // 2)
    conv("visual.conv2", width // 2, width // 2, 3); bn("visual.bn2", width // 2)
    conv("visual.conv3", width, width // 2, 3); bn("visual.bn3", width)
    inplanes = width
    for li, (mult, n, stride) in enumerate(zip((1, 2, 4, 8), layers, (1, 2, 2, 2)), start=1):
        planes = width * mult
        for bi in range(n):
            p = f"visual.layer{li}.{bi}"
            conv(p + ".conv1", planes, inplanes, 1); bn(p + ".bn1", planes)
            conv(p + ".conv2", planes, planes, 3); bn(p + ".bn2", planes)
            conv(p + ".conv3", planes * 4, planes, 1); bn(p + ".bn3", planes * 4)
            if bi == 0 and (stride > 1 or inplanes != planes * 4):
                conv(p + ".downsample.0", planes * 4, inplanes, 1); bn(p + ".downsample.1", planes * 4)
            inplanes = planes * 4
    ed = width * 32
    sd["visual.attnpool.positional_embedding"] = _r16(torch.randn((image_resolution // 32) ** 2 + 1, ed, generator=g) * ed ** -0.5)
    for nm, out in (("q_proj", ed), ("k_proj", ed), ("v_proj", ed), ("c_proj", tg.embed_dim)):
        sd[f"visual.attnpool.{nm}.weight"] = _r16(torch.randn(out, ed, generator=g) * ed ** -0.5)
        sd[f"visual.attnpool.{nm}.bias"] = _r16(0.02 * torch.randn(out, generator=g))
    return sd


def synthetic_token_ids(n_cls: int, geom: ClipGeometry | str = "ViT-B/16", seed: int = 0,
                        n_ctx_placeholders: int = 0) -> torch.Tensor:
    """ids ``[n_cls, 77]`` shaped like ``"a photo of a {name}."`` (zsclip.py:84-87) or, with
    ``n_ctx_placeholders`` > 0, like CoOp's ``"X X ... X {name}."`` (coop.py:101-120).

    Layout: ``[SOT, template..., name(1-3 tokens), '.', EOT, 0...]`` with SOT = vocab-2 and
    EOT = vocab-1, so that ``argmax(ids)`` is the EOT position as ``clip/model.py:611`` requires.
    """
    if isinstance(geom, str):
        geom = GEOMETRIES[geom]
    g = torch.Generator().manual_seed(1000 + seed)
    V = geom.vocab_size
    sot, eot = V - 2, V - 1
    lo, hi = (1000, 40000) if V > 40000 else (8, V - 8)
    # the real ids of "a photo of a" / "." / "X" in the CLIP BPE vocab, folded into tiny vocabs
    template = [320 % (V - 8), 1125 % (V - 8), 539 % (V - 8), 320 % (V - 8)]
    dot = 269 % (V - 8)
    x_tok = 343 % (V - 8)
    ids = torch.zeros(n_cls, geom.context_length, dtype=torch.long)
    for c in range(n_cls):
        k = 1 + int(torch.randint(0, 3, (1,), generator=g))
        name = torch.randint(lo, hi, (k,), generator=g).tolist()
        head = [x_tok] * n_ctx_placeholders if n_ctx_placeholders > 0 else template
        toks = [sot] + head + name + [dot, eot]
        ids[c, : len(toks)] = torch.tensor(toks)
    return ids


def synthetic_images(batch: int, geom: ClipGeometry | str = "ViT-B/16", seed: int = 0,
                     device: Optional[torch.device | str] = None) -> torch.Tensor:
    """``randn [B,3,R,R]`` fp32 ~ post-Normalize statistics."""
    if isinstance(geom, str):
        geom = GEOMETRIES[geom]
    g = torch.Generator().manual_seed(2000 + seed)
    x = torch.randn(batch, 3, geom.image_resolution, geom.image_resolution, generator=g)
    return x.to(device) if device is not None else x


def synthetic_labels(pred: torch.Tensor, n_cls: int, seed: int = 0, p_correct: float = 0.7) -> torch.Tensor:
    """Labels equal to ``pred`` with probability ``p_correct``, else uniform (non-degenerate ECE)."""
    g = torch.Generator().manual_seed(3000 + seed)
    pred = pred.detach().cpu().long()
    rnd = torch.randint(0, n_cls, pred.shape, generator=g)
    keep = torch.rand(pred.shape, generator=g) < p_correct
    return torch.where(keep, pred, rnd)


def flops_per_image(geom: ClipGeometry | str) -> float:
    """Algorithmic FLOP (2*MAC) of one image-tower forward, SURVEY §8(d) table (35.127 GFLOP for ViT-B/16)."""
    if isinstance(geom, str):
        geom = GEOMETRIES[geom]
    L, D = geom.vision_tokens, geom.vision_width
    hd = 64
    H = D // hd
    patch = 2 * (L - 1) * D * 3 * geom.vision_patch_size ** 2
    per_layer = 2 * L * D * 3 * D + 2 * 2 * H * L * L * hd + 2 * L * D * D + 2 * 2 * L * D * 4 * D
    proj = 2 * D * geom.embed_dim
    return float(patch + geom.vision_layers * per_layer + proj)


def flops_per_prompt(geom: ClipGeometry | str) -> float:
    if isinstance(geom, str):
        geom = GEOMETRIES[geom]
    L, D = geom.context_length, geom.transformer_width
    H = geom.transformer_heads
    hd = D // H
    per_layer = 2 * L * D * 3 * D + 2 * 2 * H * L * L * hd + 2 * L * D * D + 2 * 2 * L * D * 4 * D
    return float(geom.transformer_layers * per_layer + 2 * D * geom.embed_dim)


assert math.isclose(flops_per_image("ViT-B/16") / 1e9, 35.127, rel_tol=2e-3)
