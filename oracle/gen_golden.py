#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE in the build container.  TEST INFRASTRUCTURE.

Runs only where /root/reference exists (never on the GPU box).  It imports the reference's own
``clip/model.py``, ``tools/metrics.py`` and ``trainers/calibration/distanse_aware_calibration.py`` through
``importlib`` (bypassing ``clip/__init__.py`` which needs torchvision), feeds them seeded synthetic weights /
inputs from ``clip_calibration_amd.synthetic`` and stores inputs + expected outputs as small .npz fixtures.
No reference source text is stored -- only tensors.

    python oracle/gen_golden.py            # rewrites tests/golden/
"""
from __future__ import annotations

import importlib.util
import os
import warnings
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("CLIP_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")

from clip_calibration_amd import synthetic as syn  # noqa: E402


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _np(t):
    return t.detach().float().cpu().numpy()


PLAIN = {"trainer": "CoOp", "vision_depth": 0, "language_depth": 0, "vision_ctx": 0, "language_ctx": 0}
MAPLE = {"trainer": "MaPLe", "vision_depth": 0, "language_depth": 0, "vision_ctx": 0, "language_ctx": 0,
         "maple_length": 2}


def sd_checksum(sd):
    return float(sum(v.double().abs().sum().item() for v in sd.values()))


def ref_forward(ref_model, geom_name, seed, n_img, n_cls, capture=False, make_sd=None):
    """Run reference build_model + encode_image/encode_text/forward in fp32 (reading A) and fp16-on-CPU (reading B)."""
    sd = (make_sd or syn.synthetic_state_dict)(geom_name, seed=seed)
    images = syn.synthetic_images(n_img, geom_name, seed=seed)
    ids = syn.synthetic_token_ids(n_cls, geom_name, seed=seed)
    out = {"images": images.numpy(), "ids": ids.numpy(), "sd_checksum": np.float64(sd_checksum(sd)), "seed": seed}

    model16 = ref_model.build_model(dict(sd), dict(PLAIN))          # fp16 weights per convert_weights
    model32 = ref_model.build_model(dict(sd), dict(PLAIN)).float()   # what clip.load does on CPU

    caps = {}
    if capture:
        def hook(name):
            def f(_m, _i, o):
                caps[name] = _np(o)
            return f
        model32.visual.ln_pre.register_forward_hook(hook("v_ln_pre"))        # NLD
        for i, blk in enumerate(model32.visual.transformer.resblocks):
            blk.register_forward_hook(hook(f"v_block{i}"))                  # LND
        for i, blk in enumerate(model32.transformer.resblocks):
            blk.register_forward_hook(hook(f"t_block{i}"))                  # LND
        model32.visual.conv1.register_forward_hook(hook("v_conv1"))          # [B,W,g,g]

    with torch.no_grad():
        img32 = model32.encode_image(images)
        txt32 = model32.encode_text(ids)
        lpi32, _ = model32(images, ids)
        img16 = model16.encode_image(images)
        txt16 = model16.encode_text(ids)
        lpi16, _ = model16(images, ids)
    out.update(image_features=_np(img32), text_features=_np(txt32), logits=_np(lpi32),
               image_features_fp16=_np(img16), text_features_fp16=_np(txt16), logits_fp16=_np(lpi16))
    for k, v in caps.items():
        if k.startswith(("v_block", "t_block")):
            v = np.ascontiguousarray(v.transpose(1, 0, 2))  # LND -> NLD
        out["cap_" + k] = v
    return sd, model32, out


def vitl_goldens(ref_model):
    """ViT-L geometries (round 6; clip/clip.py:37-38, shape inference clip/model.py:660-665): the reference's fp32 and fp16-on-CPU outputs for
    ViT-L/14 (257 tokens) and ViT-L/14@336px (577 tokens: BASELINE configs[4]) on seeded weights, plain and with trained-CLIP-like outlier
    statistics -- what the long-sequence attention kernel and the split-row GEMMs of round 5 are pinned on.  Features + checksum only (weights
    and images regenerate from the seed).  ~1.5 TFLOP of fp32 CPU work per 336px case:  python oracle/gen_golden.py vitl"""
    nrm = lambda a: a / np.linalg.norm(a, axis=1, keepdims=True)
    for gname, stem in (("ViT-L/14", "vitl14"), ("ViT-L/14@336px", "vitl14_336")):
        for make_sd, suffix in ((None, "seed0"), (syn.outlier_state_dict, "outliers")):
            sd, m32, out = ref_forward(ref_model, gname, seed=0, n_img=2, n_cls=4, make_sd=make_sd)
            out.pop("images")
            if make_sd is not None:   # the residual statistics of the image tower, as in vitb16_outliers.npz
                stats = {}
                def stat_hook(i):
                    def f(_m, _i, o):
                        a = o.detach().abs().mean(dim=(0, 1))
                        stats[i] = (float(a.median()), float(a.max()), float(o.mean(-1).abs().mean()), float(o.std(-1).mean()))
                    return f
                for i, blk in enumerate(m32.visual.transformer.resblocks):
                    blk.register_forward_hook(stat_hook(i))
                with torch.no_grad():
                    m32.encode_image(syn.synthetic_images(1, gname, seed=0))
                out["residual_stats"] = np.array([stats[i] for i in sorted(stats)])
            np.savez_compressed(os.path.join(OUT, f"{stem}_{suffix}.npz"), **out)
            ref = nrm(out["image_features"]) @ nrm(out["text_features"]).T
            r16 = nrm(out["image_features_fp16"]) @ nrm(out["text_features_fp16"]).T
            print(f"wrote {stem}_{suffix}.npz; reference fp16-vs-fp32 cosine-logit distance {np.abs(r16 - ref).max():.2e}", flush=True)
            del sd, m32, out


def main():
    from sklearn.metrics import f1_score
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    ref_model = _load("ref_model", "clip/model.py")
    if len(sys.argv) > 1 and sys.argv[1] == "vitl":   # the ViT-L fixtures alone (the rest regenerates byte-identically without them)
        vitl_goldens(ref_model)
        return
    ref_metrics = _load("ref_metrics", "tools/metrics.py")
    ref_dac = _load("ref_dac", "trainers/calibration/distanse_aware_calibration.py")

    # ---------------- tiny geometry: weights + all intermediates ----------------
    for gname, fname in (("tiny", "tiny_clip.npz"), ("tiny3", "tiny3_clip.npz")):
        sd, m32, out = ref_forward(ref_model, gname, seed=0, n_img=3, n_cls=5, capture=True)
        if gname == "tiny":  # weights committed for the smallest geometry; the others are pinned by seed + checksum
            for k, v in sd.items():
                out["sd:" + k] = v.half().numpy() if v.dim() > 0 else v.numpy()

        # CoOp TextEncoder glue (trainers/classification/coop.py:56-67,128-144) evaluated with the reference's
        # own sub-modules: transformer / ln_final / text_projection / token_embedding.
        geom = syn.GEOMETRIES[gname]
        n_ctx = 4
        ids_c = syn.synthetic_token_ids(5, gname, seed=1, n_ctx_placeholders=n_ctx)
        g = torch.Generator().manual_seed(77)
        ctx = (0.02 * torch.randn(n_ctx, geom.transformer_width, generator=g)).half().float()
        with torch.no_grad():
            emb = m32.token_embedding(ids_c)
            prompts = torch.cat([emb[:, :1], ctx.unsqueeze(0).expand(5, -1, -1), emb[:, 1 + n_ctx:]], dim=1)
            x = prompts + m32.positional_embedding
            x = m32.transformer(x.permute(1, 0, 2)).permute(1, 0, 2)
            x = m32.ln_final(x)
            tf = x[torch.arange(5), ids_c.argmax(-1)] @ m32.text_projection
        out.update(coop_ids=ids_c.numpy(), coop_ctx=ctx.numpy(), coop_prompts=_np(prompts), coop_text_features=_np(tf))

        # CoCoOp glue (trainers/classification/cocoop.py:154-199) with the reference model's own sub-modules: meta-net
        # shift of the context per image, C prompts per image through the text tower, per-image normalise + dot.
        if gname == "tiny":
            g = torch.Generator().manual_seed(79)
            E, D = geom.embed_dim, geom.transformer_width
            mn = {"ctx": (0.02 * torch.randn(n_ctx, D, generator=g)).half().float(),
                  "meta_net.linear1.weight": (torch.randn(E // 16, E, generator=g) * E ** -0.5).half().float(),
                  "meta_net.linear1.bias": (0.1 * torch.randn(E // 16, generator=g)).half().float(),
                  "meta_net.linear2.weight": (torch.randn(D, E // 16, generator=g) * (E // 16) ** -0.5 * 0.1).half().float(),
                  "meta_net.linear2.bias": (0.02 * torch.randn(D, generator=g)).half().float()}
            images = torch.from_numpy(out["images"])
            with torch.no_grad():
                imf = m32.encode_image(images)
                imf = imf / imf.norm(dim=-1, keepdim=True)
                hid = torch.relu(torch.nn.functional.linear(imf, mn["meta_net.linear1.weight"], mn["meta_net.linear1.bias"]))
                bias = torch.nn.functional.linear(hid, mn["meta_net.linear2.weight"], mn["meta_net.linear2.bias"])
                ctx_shifted = mn["ctx"].unsqueeze(0) + bias.unsqueeze(1)
                emb = m32.token_embedding(ids_c)
                lg, tfs = [], []
                for ctx_i, imf_i in zip(ctx_shifted, imf):
                    pts = torch.cat([emb[:, :1], ctx_i.unsqueeze(0).expand(5, -1, -1), emb[:, 1 + n_ctx:]], dim=1)
                    x = pts + m32.positional_embedding
                    x = m32.transformer(x.permute(1, 0, 2)).permute(1, 0, 2)
                    x = m32.ln_final(x)
                    tf = x[torch.arange(5), ids_c.argmax(-1)] @ m32.text_projection
                    tfs.append(tf)
                    tf = tf / tf.norm(dim=-1, keepdim=True)
                    lg.append(m32.logit_scale.exp() * imf_i @ tf.t())
            out.update(cocoop_logits=_np(torch.stack(lg)), cocoop_text_features=_np(torch.stack(tfs)),
                       cocoop_ctx_shifted=_np(ctx_shifted))
            for k, v in mn.items():
                out["cocoop_pl:" + k] = v.numpy()

        # MaPLe: reference VisionTransformer_MaPLe / ResidualAttentionBlock_MaPLe (clip/model.py:259-331,427-478)
        if gname == "tiny":
            mm = ref_model.build_model(dict(sd), dict(MAPLE)).float()
            depth = 2  # PROMPT_DEPTH=2 -> one deep prompt ... use 2 deep prompts on a 2-layer tower: only layer 1 consumes
            g = torch.Generator().manual_seed(78)
            tw, vw = geom.transformer_width, geom.vision_width
            pl = {"ctx": (0.02 * torch.randn(2, tw, generator=g)).half().float(),
                  "proj.weight": (torch.randn(vw, tw, generator=g) * tw ** -0.5).half().float(),
                  "proj.bias": (0.02 * torch.randn(vw, generator=g)).half().float()}
            for i in range(depth):
                pl[f"compound_prompts_text.{i}"] = (0.02 * torch.randn(2, tw, generator=g)).half().float()
                pl[f"compound_prompt_projections.{i}.weight"] = (torch.randn(vw, tw, generator=g) * tw ** -0.5).half().float()
                pl[f"compound_prompt_projections.{i}.bias"] = (0.02 * torch.randn(vw, generator=g)).half().float()
            ids_m = syn.synthetic_token_ids(5, gname, seed=2, n_ctx_placeholders=2)
            images = torch.from_numpy(out["images"])
            with torch.no_grad():
                emb = mm.token_embedding(ids_m)
                prompts = torch.cat([emb[:, :1], pl["ctx"].unsqueeze(0).expand(5, -1, -1), emb[:, 3:]], dim=1)
                shared = torch.nn.functional.linear(pl["ctx"], pl["proj.weight"], pl["proj.bias"])
                deep_t = [pl[f"compound_prompts_text.{i}"] for i in range(depth)]
                deep_v = [torch.nn.functional.linear(deep_t[i], pl[f"compound_prompt_projections.{i}.weight"],
                                                     pl[f"compound_prompt_projections.{i}.bias"]) for i in range(depth)]
                x = prompts + mm.positional_embedding
                x = mm.transformer([x.permute(1, 0, 2), deep_t, 0])[0].permute(1, 0, 2)
                x = mm.ln_final(x)
                tfm = x[torch.arange(5), ids_m.argmax(-1)] @ mm.text_projection
                imf = mm.visual(images, shared, deep_v)
            out.update(maple_ids=ids_m.numpy(), maple_text_features=_np(tfm), maple_image_features=_np(imf),
                       maple_shared_ctx=_np(shared))
            for k, v in pl.items():
                out["maple_pl:" + k] = v.numpy()

        # IVLP / VPT blocks (clip/model.py:191-256, 361-424): per-layer prompt tokens that live INSIDE the model
        # (visual.VPT, *.resblocks.{i}.VPT_shallow).  The synthetic state_dict has none of them, so the reference's
        # non-strict fallback leaves them at their seeded random init; the fixture stores them next to the outputs.
        if gname == "tiny3":
            import contextlib, io
            for tag, dd in (("ivlp", {"trainer": "IVLP", "vision_depth": 3, "language_depth": 3, "vision_ctx": 2, "language_ctx": 2}),
                            ("vpt", {"trainer": "VPT", "vision_depth": 2, "language_depth": 0, "vision_ctx": 4, "language_ctx": 0})):
                torch.manual_seed(1234)
                with contextlib.redirect_stdout(io.StringIO()):
                    mi = ref_model.build_model(dict(sd), dict(dd)).float()
                ids_i = syn.synthetic_token_ids(5, gname, seed=3, n_ctx_placeholders=dd["language_ctx"])
                images = torch.from_numpy(out["images"])
                with torch.no_grad():
                    out[f"{tag}_image_features"] = _np(mi.encode_image(images))
                    out[f"{tag}_text_features"] = _np(mi.encode_text(ids_i))
                out[f"{tag}_ids"] = ids_i.numpy()
                for k, v in mi.state_dict().items():
                    if "VPT" in k:
                        out[f"{tag}_sd:{k}"] = v.float().numpy()
        np.savez_compressed(os.path.join(OUT, fname), **out)
        print("wrote", fname, {k: np.shape(v) for k, v in out.items() if not k.startswith("sd:")})

    # ---------------- ModifiedResNet image tower (clip/model.py:10-150): inputs by seed, outputs stored ----------------
    import contextlib, io
    rn_sd = syn.synthetic_resnet_state_dict((1, 2, 1, 1), 64, 64, "tiny", seed=0)
    with contextlib.redirect_stdout(io.StringIO()):
        rn = ref_model.build_model(dict(rn_sd), dict(PLAIN)).float()
    assert type(rn.visual).__name__ == "ModifiedResNet"
    rn_images = syn.synthetic_images(3, syn.ClipGeometry(128, 64, 1, 64, 64, 77, 256, 128, 2, 2), seed=5)
    rn_ids = syn.synthetic_token_ids(4, "tiny", seed=5)
    with torch.no_grad():
        rn_img = rn.encode_image(rn_images)
        rn_txt = rn.encode_text(rn_ids)
        rn_lpi, _ = rn(rn_images, rn_ids)
    np.savez_compressed(os.path.join(OUT, "resnet_tiny.npz"), image_features=_np(rn_img), text_features=_np(rn_txt), logits=_np(rn_lpi),
                        ids=rn_ids.numpy(), sd_checksum=np.float64(sd_checksum({k: v for k, v in rn_sd.items() if v.dtype.is_floating_point})),
                        n_keys=np.int64(len(rn.state_dict())))
    print("wrote resnet_tiny.npz", len(rn.state_dict()), "keys")

    # ---------------- full ViT-B/16 geometry: inputs by seed, outputs stored ----------------
    _, _, out = ref_forward(ref_model, "ViT-B/16", seed=0, n_img=2, n_cls=8)
    out.pop("images")  # regenerated from the seed; ids are tiny so they stay
    np.savez_compressed(os.path.join(OUT, "vitb16_seed0.npz"), **out)
    print("wrote vitb16_seed0.npz")

    # ---------------- full ViT-B/16 geometry with trained-CLIP-like activation statistics (massive residual channels, spread
    # LayerNorm gains, common offset): the reference in fp32 and in fp16-on-CPU; weights regenerated from the seed ----------------
    sd_o, m32_o, out = ref_forward(ref_model, "ViT-B/16", seed=0, n_img=4, n_cls=16, make_sd=syn.outlier_state_dict)
    out.pop("images")
    stats = {}
    def stat_hook(i):
        def f(_m, _i, o):
            a = o.detach().abs().mean(dim=(0, 1))
            stats[i] = (float(a.median()), float(a.max()), float(o.mean(-1).abs().mean()), float(o.std(-1).mean()))
        return f
    for i, blk in enumerate(m32_o.visual.transformer.resblocks):
        blk.register_forward_hook(stat_hook(i))
    with torch.no_grad():
        m32_o.encode_image(syn.synthetic_images(4, "ViT-B/16", seed=0))
    out["residual_stats"] = np.array([stats[i] for i in sorted(stats)])   # per block: median |x| per channel, max, mean |row mean|, mean row std
    np.savez_compressed(os.path.join(OUT, "vitb16_outliers.npz"), **out)
    print("wrote vitb16_outliers.npz; image-tower residual stream (median |x|, max |x| per channel, |row mean|, row std) per block:")
    print(np.round(out["residual_stats"], 2))
    i32, i16 = out["image_features"], out["image_features_fp16"]
    t32, t16 = out["text_features"], out["text_features_fp16"]
    nrm = lambda a: a / np.linalg.norm(a, axis=1, keepdims=True)
    print("  reference fp16-vs-fp32 cosine-logit error on this fixture:", np.abs(nrm(i16) @ nrm(t16).T - nrm(i32) @ nrm(t32).T).max())

    # ---------------- ECE (tools/metrics.py:90-130) ----------------
    rng = np.random.default_rng(0)
    cases = {}
    def add(name, conf, pred, gt, bins):
        cases[f"{name}:conf"] = conf
        cases[f"{name}:pred"] = pred
        cases[f"{name}:gt"] = gt
        cases[f"{name}:bins"] = np.int64(bins)
        cases[f"{name}:ece"] = np.float64(ref_metrics.ECE(conf, pred, gt, bins))
        cases[f"{name}:mce"] = np.float64(ref_metrics.MCE(conf, pred, gt, bins))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")            # sklearn: quantile_method FutureWarning, constant-feature warning
            cases[f"{name}:ace"] = np.float64(ref_metrics.AdaptiveECE(conf, pred, gt, bins))
            prox = np.exp(-np.abs(np.random.default_rng(len(conf)).normal(0.8, 0.2, len(conf))))
            cases[f"{name}:prox"] = prox
            cases[f"{name}:piece"] = np.float64(ref_metrics.PIECE(conf, prox, pred, gt, 10, bins))
            cases[f"{name}:f1"] = np.float64(f1_score(gt, pred, average="macro", labels=np.unique(gt)))
    n = 2000
    conf = rng.uniform(0.05, 1.0, n)
    pred = rng.integers(0, 50, n)
    gt = np.where(rng.uniform(size=n) < conf, pred, rng.integers(0, 50, n))
    add("uniform10", conf, pred, gt, 10)
    add("uniform15", conf, pred, gt, 15)
    conf2 = conf.copy(); conf2[:37] = 1.0          # the digitize/histogram right-edge quirk
    add("ones_quirk", conf2, pred, gt, 10)
    conf3 = rng.uniform(0.9, 0.999, 300)            # 9 empty bins
    add("empty_bins", conf3, pred[:300], gt[:300], 10)
    add("single", np.array([0.73]), np.array([3]), np.array([3]), 10)
    add("all_one", np.ones(16), pred[:16], pred[:16], 10)
    conf4 = rng.uniform(0, 1, 500).astype(np.float32)  # float32 confidences straddling bin edges
    conf4[:11] = np.linspace(0, 1, 11, dtype=np.float32)
    add("edges_f32", conf4, pred[:500], gt[:500], 10)
    np.savez_compressed(os.path.join(OUT, "ece_cases.npz"), **cases)
    print("wrote ece_cases.npz")

    # ---------------- DAC (distanse_aware_calibration.py) ----------------
    # predict() hard-codes .cuda(); there is no GPU in this container, so .cuda() is shimmed to the identity for
    # the duration of the call -- the arithmetic executed is still the reference's own lines 52-58.
    dac = {}
    torch_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        for name, (cb, cn, dim, k) in {"c50": (50, 50, 128, 5), "c19": (19, 18, 64, 5), "k3": (8, 5, 32, 3)}.items():
            def feats(n, base=None, jitter=0.0):
                f = rng.normal(size=(n, dim)) if base is None else base + jitter * rng.normal(size=base.shape)
                return f / np.linalg.norm(f, axis=1, keepdims=True)
            base_zs, cur_zs = feats(cb), feats(cn)
            base_tuned, cur_tuned = feats(cb, base_zs, 0.05), feats(cn, cur_zs, 0.05)
            cur_tuned[0] = base_tuned[1] + 1e-3            # "base class aware" branch (distance < 0.05)
            cal = ref_dac.DistanseAwareCalibration()
            cal.fit(base_zs, cur_zs, base_tuned, cur_tuned, k)
            logits = (100 * feats(64) @ cur_tuned.T)        # float64, as the evaluator's python lists give
            pred_out = cal.predict(logits.copy())
            dac.update({f"{name}:base_zs": base_zs, f"{name}:cur_zs": cur_zs, f"{name}:base_tuned": base_tuned,
                        f"{name}:cur_tuned": cur_tuned, f"{name}:k": np.int64(k),
                        f"{name}:class_confidence": cal.class_confidence, f"{name}:logits": logits,
                        f"{name}:scaled_logits": pred_out})
    finally:
        torch.Tensor.cuda = torch_cuda
    np.savez_compressed(os.path.join(OUT, "dac_cases.npz"), **dac)
    print("wrote dac_cases.npz")

    # ---------------- kNN proximity (trainers/calibration/proximity.py) ----------------
    # both functions hard-code .to('cuda'); shimmed to the identity as above
    ref_prox = _load("ref_prox", "trainers/calibration/proximity.py")
    torch_to = torch.Tensor.to
    torch.Tensor.to = lambda self, *a, **k: self if (a and a[0] == "cuda") else torch_to(self, *a, **k)
    knn = {}
    try:
        for name, (nq, nr, dim, k) in {"k5": (37, 150, 128, 5), "k10": (20, 64, 64, 10), "tiny": (3, 7, 64, 5)}.items():
            refs = rng.normal(size=(nr, dim)).astype(np.float32)
            refs /= np.linalg.norm(refs, axis=1, keepdims=True)
            q = rng.normal(size=(nq, dim)).astype(np.float32)
            q /= np.linalg.norm(q, axis=1, keepdims=True)
            q[0] = refs[3]                                   # an exact duplicate: distance 0
            knn.update({f"{name}:refs": refs, f"{name}:queries": q, f"{name}:k": np.int64(k),
                        f"{name}:knn": ref_prox.get_knn_dists(refs, q, k),
                        f"{name}:val_knn": ref_prox.get_val_image_knn_dists(refs, min(k, nr - 2))})
    finally:
        torch.Tensor.to = torch_to
    np.savez_compressed(os.path.join(OUT, "knn_cases.npz"), **knn)
    print("wrote knn_cases.npz")

    # ---------------- tokenizer (clip/simple_tokenizer.py; packing of clip/clip.py:207-224) ----------------
    # ftfy is not installed: stubbed to the identity (exact for ASCII).  Every merge-table lookup that HIT during the
    # run is recorded, so the fixture carries a sparse {pair: rank} table instead of OpenAI's 1.3 MB data file.
    import json, types
    sys.modules.setdefault("ftfy", types.SimpleNamespace(fix_text=lambda t: t))
    ref_tok_mod = _load("ref_tok", "clip/simple_tokenizer.py")
    tok = ref_tok_mod.SimpleTokenizer(os.path.join(REF, "clip", "bpe_simple_vocab_16e6.txt.gz"))

    class Recording(dict):
        def __init__(self, d):
            super().__init__(d)
            self.hits = {}
        def get(self, k, default=None):
            v = super().get(k, default)
            if v is not default and k in self:
                self.hits[k] = v
            return v
        def __contains__(self, k):
            return super().__contains__(k)
    tok.bpe_ranks = Recording(tok.bpe_ranks)
    names = ["accordion", "airplane", "anchor", "ant", "barrel", "bass", "beaver", "binocular", "bonsai", "brain",
             "brontosaurus", "buddha", "butterfly", "camera", "cannon", "car_side", "ceiling_fan", "cellphone", "chair",
             "chandelier", "cougar_body", "crab", "crayfish", "crocodile", "cup", "dalmatian", "dollar_bill", "dolphin",
             "dragonfly", "electric_guitar", "elephant", "emu", "euphonium", "ewer", "ferry", "flamingo", "garfield",
             "gerenuk", "gramophone", "grand_piano", "hawksbill", "headphone", "hedgehog", "helicopter", "ibis",
             "inline_skate", "joshua_tree", "kangaroo", "ketch", "lamp", "laptop", "llama", "lobster", "lotus", "mandolin",
             "mayfly", "menorah", "metronome", "minaret", "nautilus", "octopus", "okapi", "pagoda", "panda", "pigeon",
             "pizza", "platypus", "pyramid", "revolver", "rhino", "rooster", "saxophone", "schooner", "scissors",
             "scorpion", "sea_horse", "snoopy", "soccer_ball", "stapler", "starfish", "stegosaurus", "stop_sign",
             "strawberry", "sunflower", "tick", "trilobite", "umbrella", "watch", "water_lilly", "wheelchair", "wild_cat",
             "windsor_chair", "wrench", "yin_yang", "abyssinian", "american pit bull terrier", "boeing 737-800",
             "crème brûlée", "annual crop land", "2012 tesla model s sedan", "bird's nest", "it's 3 o'clock &amp; fine"]
    templates = ["a photo of a {}.", "a photo of a {}, a type of pet.", "{} texture.", "a centered satellite photo of {}.",
                 "a photo of a person doing {}.", "X X X X {}.", "X X X X X X X X X X X X X X X X {}."]
    prompts = [t.format(n.replace("_", " ")) for i, n in enumerate(names) for t in (templates[i % len(templates)], templates[0])]
    sot, eot = tok.encoder["<|startoftext|>"], tok.encoder["<|endoftext|>"]
    ids = [[sot] + tok.encode(p) + [eot] for p in prompts]
    fixture = {"prompts": prompts, "ids": ids, "sot": sot, "eot": eot,
               "merges": [[a, b, int(r)] for (a, b), r in sorted(tok.bpe_ranks.hits.items(), key=lambda kv: kv[1])],
               "decoded": [tok.decode(i[1:-1]) for i in ids[:10]]}
    with open(os.path.join(OUT, "tokenizer_cases.json"), "w") as f:
        json.dump(fixture, f)
    print("wrote tokenizer_cases.json:", len(prompts), "prompts,", len(fixture["merges"]), "merge ranks")


if __name__ == "__main__":
    main()
