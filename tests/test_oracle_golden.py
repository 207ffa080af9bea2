"""Pins oracle/clip_oracle.py against outputs of the reference itself (tests/golden, made by oracle/gen_golden.py)."""
import numpy as np
import pytest
import torch

from clip_calibration_amd import synthetic as syn
from oracle import clip_oracle as orc
from conftest import golden_state_dict, load_golden

TOL = dict(rtol=2e-5, atol=2e-5)   # fp32 vs fp32, different op order (manual attention vs torch SDPA)


def _sd_checksum(sd):
    return float(sum(v.double().abs().sum().item() for v in sd.values()))


def test_synthetic_generator_reproduces_committed_weights(tiny):
    sd = syn.synthetic_state_dict("tiny", seed=0)
    gsd = golden_state_dict(tiny)
    assert set(sd) == set(gsd)
    for k in sd:
        assert torch.equal(sd[k].float(), gsd[k].float()), k
    assert _sd_checksum(sd) == pytest.approx(float(tiny["sd_checksum"]), rel=1e-12)


@pytest.mark.parametrize("name", ["tiny_clip.npz", "tiny3_clip.npz"])
def test_towers_and_intermediates(name):
    g = load_golden(name)
    gname = "tiny" if name.startswith("tiny_") else "tiny3"
    sd = syn.synthetic_state_dict(gname, seed=0)
    assert _sd_checksum(sd) == pytest.approx(float(g["sd_checksum"]), rel=1e-12)
    images = torch.from_numpy(g["images"])
    ids = torch.from_numpy(g["ids"])

    # patch embed == conv1 output
    pe = orc.patch_embed(images, sd["visual.conv1.weight"])
    conv = torch.from_numpy(g["cap_v_conv1"]).flatten(2).permute(0, 2, 1)
    np.testing.assert_allclose(pe.numpy(), conv.numpy(), **TOL)

    # per-block captures, image tower
    x = torch.cat([sd["visual.class_embedding"].expand(images.shape[0], 1, -1), pe], 1) + sd["visual.positional_embedding"]
    x = orc.layer_norm(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"])
    np.testing.assert_allclose(x.numpy(), g["cap_v_ln_pre"], **TOL)
    nl = len([k for k in g if k.startswith("cap_v_block")])
    for i in range(nl):
        x = orc.residual_block(x, sd, f"visual.transformer.resblocks.{i}.", sd["visual.conv1.weight"].shape[0] // 64, None)
        np.testing.assert_allclose(x.numpy(), g[f"cap_v_block{i}"], rtol=1e-4, atol=1e-4)

    # text blocks
    t = sd["token_embedding.weight"][ids] + sd["positional_embedding"]
    mask = orc.causal_mask(t.shape[1])
    for i in range(len([k for k in g if k.startswith("cap_t_block")])):
        t = orc.residual_block(t, sd, f"transformer.resblocks.{i}.", sd["ln_final.weight"].shape[0] // 64, mask)
        np.testing.assert_allclose(t.numpy(), g[f"cap_t_block{i}"], rtol=1e-4, atol=1e-4)

    img = orc.encode_image(sd, images)
    txt = orc.encode_text(sd, ids)
    np.testing.assert_allclose(img.numpy(), g["image_features"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(txt.numpy(), g["text_features"], rtol=1e-4, atol=1e-4)
    logits, img_n, txt_n = orc.clip_logits(img, txt, sd["logit_scale"].exp())
    np.testing.assert_allclose(logits.numpy(), g["logits"], rtol=1e-4, atol=2e-3)   # logits ~ +-100
    np.testing.assert_allclose(img_n.norm(dim=-1).numpy(), 1.0, atol=1e-6)

    # reading (B): the reference's fp16-on-CPU path bounds how far any fp16 implementation sits from fp32
    cos32 = (logits / sd["logit_scale"].exp()).numpy()
    cos16 = g["logits_fp16"] / float(sd["logit_scale"].exp())
    assert np.abs(cos32 - cos16).max() < 5e-3


@pytest.mark.parametrize("name", ["tiny_clip.npz", "tiny3_clip.npz"])
def test_coop_text_encoder(name):
    g = load_golden(name)
    sd = syn.synthetic_state_dict("tiny" if name.startswith("tiny_") else "tiny3", seed=0)
    ids = torch.from_numpy(g["coop_ids"])
    ctx = torch.from_numpy(g["coop_ctx"])
    prompts = orc.coop_prompts(sd, ids, ctx)
    np.testing.assert_allclose(prompts.numpy(), g["coop_prompts"], rtol=0, atol=0)
    tf = orc.text_encoder(sd, prompts, ids)
    np.testing.assert_allclose(tf.numpy(), g["coop_text_features"], rtol=1e-4, atol=1e-4)


def test_maple(tiny):
    sd = syn.synthetic_state_dict("tiny", seed=0)
    pl = {k[len("maple_pl:"):]: torch.from_numpy(v) for k, v in tiny.items() if k.startswith("maple_pl:")}
    ids = torch.from_numpy(tiny["maple_ids"])
    images = torch.from_numpy(tiny["images"])
    prompts, shared, deep_t, deep_v = orc.maple_prompt_learner(sd, ids, pl)
    np.testing.assert_allclose(shared.numpy(), tiny["maple_shared_ctx"], rtol=1e-5, atol=1e-6)
    tf = orc.text_encoder(sd, prompts, ids, deep_prompts=deep_t, n_ctx=2)
    np.testing.assert_allclose(tf.numpy(), tiny["maple_text_features"], rtol=1e-4, atol=1e-4)
    imf = orc.encode_image(sd, images, shared_ctx=shared, deep_prompts=deep_v)
    np.testing.assert_allclose(imf.numpy(), tiny["maple_image_features"], rtol=1e-4, atol=1e-4)


def test_cocoop(tiny):
    g, sd = tiny, syn.synthetic_state_dict("tiny", seed=0)
    pl = {k.split(":", 1)[1]: torch.from_numpy(v) for k, v in g.items() if k.startswith("cocoop_pl:")}
    with torch.no_grad():
        logits, _, feats = orc.cocoop_forward(sd, pl, torch.from_numpy(g["images"]), torch.from_numpy(g["coop_ids"]))
    np.testing.assert_allclose(feats.numpy(), g["cocoop_text_features"], atol=2e-5)
    np.testing.assert_allclose(logits.numpy() / 100.0, g["cocoop_logits"] / 100.0, atol=2e-5)
    # the shift is per image: the three images see different text features
    assert np.abs(g["cocoop_text_features"][0] - g["cocoop_text_features"][1]).max() > 1e-3


@pytest.mark.parametrize("tag", ["ivlp", "vpt"])
def test_ivlp_vpt_blocks(tag):
    """f-4: per-layer prompt tokens of the IVLP / VPT design (clip/model.py:191-256) on the 3-layer geometry."""
    g = load_golden("tiny3_clip.npz")
    sd = syn.synthetic_state_dict("tiny3", seed=0)
    sd.update({k.split(":", 1)[1]: torch.from_numpy(v) for k, v in g.items() if k.startswith(tag + "_sd:")})
    shallow, deep_v, deep_t, n_ctx = orc.ivlp_prompts(sd)
    assert (len(deep_v), len(deep_t), n_ctx) == ((2, 2, 2) if tag == "ivlp" else (1, 0, 0))
    with torch.no_grad():
        img = orc.encode_image_ivlp(sd, torch.from_numpy(g["images"]))
        txt = orc.encode_text_ivlp(sd, torch.from_numpy(g[tag + "_ids"]))
    np.testing.assert_allclose(img.numpy(), g[tag + "_image_features"], atol=2e-5)
    np.testing.assert_allclose(txt.numpy(), g[tag + "_text_features"], atol=2e-5)
    # the prompts matter: the plain towers give something else on the image side
    assert np.abs(img.numpy() - g["image_features"]).max() > 1e-3


RN_GEOM = syn.ClipGeometry(128, 64, 1, 64, 64, 77, 256, 128, 2, 2)     # image side: only the resolution is read


def test_modified_resnet_tower():
    """f-4: ModifiedResNet image tower (clip/model.py:10-150) -- oracle against the reference's own build of the seeded checkpoint."""
    g = load_golden("resnet_tiny.npz")
    sd = syn.synthetic_resnet_state_dict((1, 2, 1, 1), 64, 64, "tiny", seed=0)
    assert len(sd) == int(g["n_keys"])                                 # the reference's ModifiedResNet CLIP has exactly these keys
    images = syn.synthetic_images(3, RN_GEOM, seed=5)
    with torch.no_grad():
        img = orc.encode_image_resnet(sd, images)
        txt = orc.encode_text(sd, torch.from_numpy(g["ids"]))
        logits, _, _ = orc.clip_logits(img, txt, sd["logit_scale"].exp())
    np.testing.assert_allclose(img.numpy(), g["image_features"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(txt.numpy(), g["text_features"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(logits.numpy() / 100.0, g["logits"] / 100.0, atol=2e-5)


def test_vitb16_full_geometry():
    g = load_golden("vitb16_seed0.npz")
    sd = syn.synthetic_state_dict("ViT-B/16", seed=0)
    assert len(sd) == 302
    assert _sd_checksum(sd) == pytest.approx(float(g["sd_checksum"]), rel=1e-12)
    images = syn.synthetic_images(2, "ViT-B/16", seed=0)
    ids = torch.from_numpy(g["ids"])
    assert torch.equal(ids, syn.synthetic_token_ids(8, "ViT-B/16", seed=0))
    img = orc.encode_image(sd, images)
    txt = orc.encode_text(sd, ids)
    np.testing.assert_allclose(img.numpy(), g["image_features"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(txt.numpy(), g["text_features"], rtol=2e-4, atol=2e-4)
    logits, _, _ = orc.clip_logits(img, txt, sd["logit_scale"].exp())
    np.testing.assert_allclose(logits.numpy() / 100.0, g["logits"] / 100.0, atol=2e-5)
    # the reference's own fp16-vs-fp32 noise floor on cosine logits: record it is < 1e-3
    assert np.abs(g["logits_fp16"] - g["logits"]).max() / float(sd["logit_scale"].exp()) < 1e-3


def test_vitb16_outlier_statistics_fixture():
    """ViT-B/16 geometry with trained-CLIP-like activation statistics (massive residual channels ~60 against a typical 2.5,
    LayerNorm gains over an order of magnitude, a common offset: mean^2 ~ E[x^2]/2): the oracle against the reference's own
    fp32 outputs, and the reference's fp16-vs-fp32 distance recorded as the noise floor of this fixture."""
    g = load_golden("vitb16_outliers.npz")
    sd = syn.outlier_state_dict("ViT-B/16", seed=0)
    assert _sd_checksum(sd) == pytest.approx(float(g["sd_checksum"]), rel=1e-12)
    st = g["residual_stats"]                       # per block: median |x| per channel, max, mean |row mean|, mean row std
    assert st.shape == (12, 4) and (st[1:, 1] > 20 * st[1:, 0]).all() and (st[:, 2] > 0.5 * st[:, 3]).all()
    images = syn.synthetic_images(4, "ViT-B/16", seed=0)
    ids = torch.from_numpy(g["ids"])
    img = orc.encode_image(sd, images)
    txt = orc.encode_text(sd, ids)
    n = lambda a: a / np.linalg.norm(a, axis=1, keepdims=True)
    ref = n(g["image_features"]) @ n(g["text_features"]).T
    assert np.abs(n(img.numpy()) @ n(txt.numpy()).T - ref).max() < 5e-5
    np.testing.assert_allclose(img.numpy(), g["image_features"], rtol=1e-3, atol=1e-3 * np.abs(g["image_features"]).max())
    ref16 = np.abs(n(g["image_features_fp16"]) @ n(g["text_features_fp16"]).T - ref).max()
    assert 1e-5 < ref16 < 1e-3


@pytest.mark.parametrize("gname,stem", [("ViT-L/14", "vitl14"), ("ViT-L/14@336px", "vitl14_336")])
@pytest.mark.parametrize("kind", ["seed0", "outliers"])
def test_vitl_geometries(gname, stem, kind):
    """Round 6: the ViT-L geometries (clip/clip.py:37-38; shapes inferred as in clip/model.py:660-665) -- 257 and 577 tokens, width 1024, 24 layers --
    against the reference's own fp32 outputs on seeded weights, plain and with trained-CLIP-like outlier statistics (oracle/gen_golden.py vitl).
    The oracle here; the HIP towers (ring attention, split-row GEMMs) in tests/test_gpu_model.py::test_golden_vitl."""
    g = load_golden(f"{stem}_{kind}.npz")
    sd = (syn.outlier_state_dict if kind == "outliers" else syn.synthetic_state_dict)(gname, seed=0)
    assert _sd_checksum(sd) == pytest.approx(float(g["sd_checksum"]), rel=1e-12)
    ids = torch.from_numpy(g["ids"])
    assert torch.equal(ids, syn.synthetic_token_ids(4, gname, seed=0))
    txt = orc.encode_text(sd, ids)
    n = lambda a: a / np.linalg.norm(a, axis=1, keepdims=True)
    assert np.abs(n(txt.numpy()) @ n(g["text_features"]).T - n(g["text_features"]) @ n(g["text_features"]).T).max() < 5e-5
    if kind == "outliers":
        st = g["residual_stats"]
        assert st.shape == (24, 4) and (st[2:, 1] > 10 * st[2:, 0]).all()
    if stem == "vitl14_336" and kind == "outliers":
        return                                        # (one 0.38 TFLOP fp32 image pass per geometry is enough for the CPU suite's budget)
    img = orc.encode_image(sd, syn.synthetic_images(1, gname, seed=0) if stem == "vitl14_336" else syn.synthetic_images(2, gname, seed=0))
    ref_img = g["image_features"][: img.shape[0]]
    ref = n(ref_img) @ n(g["text_features"]).T
    assert np.abs(n(img.numpy()) @ n(txt.numpy()).T - ref).max() < 5e-5
    np.testing.assert_allclose(img.numpy(), ref_img, rtol=1e-3, atol=1e-3 * np.abs(ref_img).max())
    ref16 = np.abs(n(g["image_features_fp16"]) @ n(g["text_features_fp16"]).T - n(g["image_features"]) @ n(g["text_features"]).T).max()
    assert 1e-6 < ref16 < 1e-3


def test_oracle_text_on_a_cut_context():
    """conftest.oracle_text_features (the GPU suite's cheaper text oracle for long class lists) == the oracle on the full 77-token context: causal
    mask + EOT-row read-out make everything behind the last EOT unreachable."""
    from conftest import oracle_text_features
    for gname, n in (("tiny", 9), ("ViT-B/16", 12)):
        sd = syn.synthetic_state_dict(gname, seed=0)
        ids = syn.synthetic_token_ids(n, gname, seed=4, n_ctx_placeholders=4)
        full, cut = orc.encode_text(sd, ids), oracle_text_features(sd, ids)
        assert int(ids.argmax(-1).max()) + 1 < ids.shape[1]
        np.testing.assert_allclose(cut.numpy(), full.numpy(), rtol=1e-5, atol=1e-5 * float(full.abs().max()))


def test_ece_cases():
    g = load_golden("ece_cases.npz")
    names = sorted({k.split(":")[0] for k in g})
    assert len(names) >= 7
    for n in names:
        got = orc.ece(g[f"{n}:conf"], g[f"{n}:pred"], g[f"{n}:gt"], int(g[f"{n}:bins"]))
        assert got == pytest.approx(float(g[f"{n}:ece"]), abs=1e-15), n
        # float32 confidences: pandas (the reference) averages them in float32, the oracle in float64
        mtol = 1e-12 if g[f"{n}:conf"].dtype == np.float64 else 2e-7
        assert orc.mce(g[f"{n}:conf"], g[f"{n}:pred"], g[f"{n}:gt"], int(g[f"{n}:bins"])) == pytest.approx(float(g[f"{n}:mce"]), abs=mtol), n
        bins = int(g[f"{n}:bins"])
        assert orc.ace(g[f"{n}:conf"], g[f"{n}:pred"], g[f"{n}:gt"], bins) == pytest.approx(float(g[f"{n}:ace"]), abs=mtol), n
        assert orc.piece(g[f"{n}:conf"], g[f"{n}:prox"], g[f"{n}:pred"], g[f"{n}:gt"], 10, bins) == pytest.approx(float(g[f"{n}:piece"]), abs=mtol), n
        assert orc.macro_f1(g[f"{n}:pred"], g[f"{n}:gt"]) == pytest.approx(float(g[f"{n}:f1"]), abs=1e-12), n


def test_dac_cases():
    g = load_golden("dac_cases.npz")
    for n in sorted({k.split(":")[0] for k in g}):
        conf = orc.dac_fit(g[f"{n}:base_zs"], g[f"{n}:cur_zs"], g[f"{n}:base_tuned"], g[f"{n}:cur_tuned"], int(g[f"{n}:k"]))
        np.testing.assert_allclose(conf, g[f"{n}:class_confidence"], rtol=1e-13)
        assert conf[0] == 1.0   # the "base class aware" branch
        out = orc.dac_predict(g[f"{n}:logits"], conf)
        assert out.dtype == np.float32
        np.testing.assert_array_equal(out, g[f"{n}:scaled_logits"])


def test_softmax_conf_pred_and_calibrated_ece():
    g = load_golden("dac_cases.npz")
    lg = g["c50:logits"]
    from scipy.special import softmax
    np.testing.assert_allclose(orc.softmax_probs(lg), softmax(lg, axis=-1), rtol=1e-13)
    labels = np.argmax(lg, 1)
    e, c, p = orc.calibrated_ece(lg, labels, g["c50:class_confidence"])
    assert 0 <= e <= 1 and c.shape == p.shape == (lg.shape[0],)


def test_tokenize_packing():
    ids = orc.tokenize_ids([[320, 1125, 539, 320, 48760, 269]], 49406, 49407)
    assert ids.shape == (1, 77) and ids[0, 0] == 49406 and ids[0, 7] == 49407 and ids[0, 8:].sum() == 0
    assert ids.argmax(-1)[0] == 7
    with pytest.raises(RuntimeError):
        orc.tokenize_ids([[5] * 80], 1, 2)


def test_knn_cases():
    g = load_golden("knn_cases.npz")
    for n in sorted({k.split(":")[0] for k in g}):
        k = int(g[f"{n}:k"])
        got = orc.knn_dists(g[f"{n}:refs"], g[f"{n}:queries"], k)
        np.testing.assert_allclose(got, g[f"{n}:knn"], rtol=2e-6, atol=2e-7)
        assert got[0, 0] < 1e-6                                   # the planted duplicate
        kv = g[f"{n}:val_knn"].shape[1]
        np.testing.assert_allclose(orc.val_image_knn_dists(g[f"{n}:refs"], kv), g[f"{n}:val_knn"], rtol=2e-6, atol=2e-7)
