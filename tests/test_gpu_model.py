"""Model-level parity on a real MI355X: the HIP towers behind ``build_model`` against (a) fixtures produced by the
reference itself (tests/golden) and (b) the CPU oracle on the same seeded inputs.

Tolerance (BASELINE.json north_star): logits within 1e-3 of the fp32 CPU path.  As SURVEY §7 derives, that bound is
meaningful on COSINE logits (the reference's own fp16 path sits 2.9e-4 from its fp32 path on cosines, i.e. 0.03 at
scale 100), so every check below compares cosine logits (= logits / exp(logit_scale)) with atol 1e-3, and scaled
logits with the same bound times the scale.  |dECE| < 1e-3 absolute."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

from clip_calibration_amd import synthetic as syn  # noqa: E402
from oracle import clip_oracle as orc  # noqa: E402  (checker only)

COS_TOL = 1e-3
PLAIN = {"trainer": "CoOp", "vision_depth": 0, "language_depth": 0, "vision_ctx": 0, "language_ctx": 0}


def _build(gname, dd=None, seed=0):
    from clip_calibration_amd.model import build_model
    sd = syn.synthetic_state_dict(gname, seed=seed)
    return sd, build_model(dict(sd), dict(dd or PLAIN)).cuda()


def _cos(a, b):
    a = a / np.linalg.norm(a, axis=-1, keepdims=True)
    b = b / np.linalg.norm(b, axis=-1, keepdims=True)
    return a @ b.T


def _feat_close(got, ref, what):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, what
    assert np.isfinite(got).all(), what
    # direction: every row within 1e-3 in cosine-logit terms against every reference row
    assert np.abs(_cos(got, ref) - _cos(ref, ref)).max() < COS_TOL, what
    # magnitude: un-normalised features agree to 0.5 % of the largest entry
    assert np.abs(got - ref).max() <= 5e-3 * np.abs(ref).max(), what


@pytest.mark.parametrize("gname,fname", [("tiny", "tiny_clip.npz"), ("tiny3", "tiny3_clip.npz")])
def test_golden_tiny(gname, fname):
    g = load_golden(fname)
    sd, model = _build(gname)
    images = torch.from_numpy(g["images"]).cuda()
    ids = torch.from_numpy(g["ids"]).cuda()
    with torch.no_grad():
        img = model.image_features_f32(images)
        txt = model.text_features_f32(ids)
        lpi, lpt = model(images, ids)
        enc_i = model.encode_image(images)
        enc_t = model.encode_text(ids)
    _feat_close(img.cpu().numpy(), g["image_features"], "image features")
    _feat_close(txt.cpu().numpy(), g["text_features"], "text features")
    assert enc_i.dtype == enc_t.dtype == model.dtype == torch.float16   # reference: output dtype = model.dtype
    scale = float(sd["logit_scale"].exp())
    assert np.abs(lpi.float().cpu().numpy() / scale - g["logits"] / scale).max() < COS_TOL + 2e-3   # fp16 output rounding of +-100
    assert np.abs(lpt.float().cpu().numpy().T / scale - g["logits"] / scale).max() < COS_TOL + 2e-3
    # fp32 logits through the fused kernel
    from clip_calibration_amd import ops
    lg, conf, pred = ops.logits_fused(ops.l2_normalize(img), ops.l2_normalize(txt), scale)
    assert np.abs(lg.cpu().numpy() / scale - g["logits"] / scale).max() < COS_TOL
    assert np.array_equal(pred.cpu().numpy(), g["logits"].argmax(1))


@pytest.mark.parametrize("what", ["ViT-B/32", "patch 8", "patch 32, fp32 stream", "patch 16, 3 images of 96 px"])
def test_image_tower_patch_sizes_vs_oracle(what, clipmi_option):
    """The patch embedding reads the NCHW image in the GEMM's loader (gemm_pp_kernel<IM2COL>: 64 / P row segments of one channel per K-step) for
    patch sizes 8, 16 and 32 -- 8, 4 and 2 segments per step, 1 / 4 / 16 K-steps per channel -- with ragged patch counts and both stream
    precisions; ViT-B/32 at full depth.  Image features against the fp32 oracle (clip/model.py:394-424)."""
    if what == "ViT-B/32":
        geom, n = syn.GEOMETRIES["ViT-B/32"], 6
    elif what == "patch 8":
        geom, n = syn.ClipGeometry(64, 40, 2, 128, 8, 77, 256, 128, 2, 2), 7          # 5 x 5 grid: 26 tokens
    elif what.startswith("patch 32"):
        geom, n = syn.ClipGeometry(64, 96, 2, 192, 32, 77, 256, 128, 2, 2), 5         # 3 x 3 grid: 10 tokens
        clipmi_option("residual_f16", 0)
    else:
        geom, n = syn.ClipGeometry(64, 96, 3, 128, 16, 77, 256, 128, 2, 2), 3         # 6 x 6 grid: 37 tokens
    from clip_calibration_amd.model import build_model
    sd = syn.synthetic_state_dict(geom, seed=2)
    model = build_model(dict(sd), dict(PLAIN)).cuda()
    images = syn.synthetic_images(n, geom, seed=4)
    with torch.no_grad():
        got = model.image_features_f32(images.cuda()).cpu().numpy()
        got16 = model.image_features_f32(images.half().cuda()).cpu().numpy()           # an fp16 image: no cast pass
        ref = orc.encode_image(sd, images).numpy()
        ref16 = orc.encode_image(sd, images.half().float()).numpy()
    _feat_close(got, ref, what)
    _feat_close(got16, ref16, what + " (fp16 image)")


@pytest.mark.parametrize("gname,fname", [("tiny", "tiny_clip.npz"), ("tiny3", "tiny3_clip.npz")])
def test_golden_coop_text_encoder_both_call_styles(gname, fname):
    g = load_golden(fname)
    sd, model = _build(gname)
    ids = torch.from_numpy(g["coop_ids"]).cuda()
    prompts = torch.from_numpy(g["coop_prompts"]).cuda()
    # (a) the mirror of trainers/classification/coop.py TextEncoder (fused device call)
    from clip_calibration_amd.trainers import TextEncoder
    tf = TextEncoder(model)(prompts, ids)
    _feat_close(tf.cpu().numpy(), g["coop_text_features"], "fused text encoder")
    # (b) the reference's own TextEncoder.forward statements, unchanged, on the attribute surface (coop.py:56-67)
    dtype = model.dtype
    with torch.no_grad():
        x = prompts.type(dtype) + model.positional_embedding.type(dtype)
        x = x.permute(1, 0, 2)
        x = model.transformer(x)
        x = x.permute(1, 0, 2)
        x = model.ln_final(x).type(dtype)
        x = x[torch.arange(x.shape[0]), ids.argmax(dim=-1)] @ model.text_projection
    got = x.float().cpu().numpy()
    assert np.abs(_cos(got, g["coop_text_features"]) - _cos(g["coop_text_features"], g["coop_text_features"])).max() < 3e-3  # fp16 activations at the boundary
    # CoOp prompt assembly mirror
    from clip_calibration_amd.trainers import PromptLearner
    pl = PromptLearner(model, ids, n_ctx=g["coop_ctx"].shape[0])
    with torch.no_grad():
        pl.ctx.copy_(torch.from_numpy(g["coop_ctx"]).cuda())
    assert np.array_equal(pl().detach().float().cpu().numpy(), torch.from_numpy(g["coop_prompts"]).half().float().numpy())


def test_golden_maple():
    g = load_golden("tiny_clip.npz")
    dd = dict(PLAIN, trainer="MaPLe", maple_length=2)
    sd, model = _build("tiny", dd)
    pl = {k[len("maple_pl:"):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("maple_pl:")}
    ids = torch.from_numpy(g["maple_ids"])
    images = torch.from_numpy(g["images"]).cuda()
    prompts, shared, deep_t, deep_v = orc.maple_prompt_learner(sd, ids, pl)   # tiny host projections
    with torch.no_grad():
        tf = model.text_encoder_f32(prompts.cuda(), ids.cuda(), [t.cuda() for t in deep_t], 2)
        imf = model.image_features_f32(images, shared.cuda(), [t.cuda() for t in deep_v])
        # reference call style: visual(image, shared_ctx, deep) and transformer([x, deep, 0]) (maple.py:64-66,207)
        imf2 = model.visual(images.half(), shared.cuda().half(), [t.cuda() for t in deep_v])
        x = (prompts.cuda().half() + model.positional_embedding.half()).permute(1, 0, 2)
        out = model.transformer([x, [t.cuda() for t in deep_t], 0])
    _feat_close(tf.cpu().numpy(), g["maple_text_features"], "maple text")
    _feat_close(imf.cpu().numpy(), g["maple_image_features"], "maple image")
    assert imf2.dtype == torch.float16 and torch.allclose(imf2.float(), imf, atol=2e-2, rtol=2e-2)
    assert isinstance(out, list) and out[0].shape == x.shape and out[2] == 1   # 2-layer tower consumes one deep prompt
    # the trainer mirror end to end
    from clip_calibration_amd.trainers import MaPLeCLIP
    mc = MaPLeCLIP(model, ids, n_ctx=2, prompt_depth=3)
    with torch.no_grad():
        mc.prompt_learner.ctx.copy_(pl["ctx"].cuda())
        mc.prompt_learner.proj.weight.copy_(pl["proj.weight"].cuda()); mc.prompt_learner.proj.bias.copy_(pl["proj.bias"].cuda())
        for i in range(2):
            mc.prompt_learner.compound_prompts_text[i].copy_(pl[f"compound_prompts_text.{i}"].cuda())
            mc.prompt_learner.compound_prompt_projections[i].weight.copy_(pl[f"compound_prompt_projections.{i}.weight"].cuda())
            mc.prompt_learner.compound_prompt_projections[i].bias.copy_(pl[f"compound_prompt_projections.{i}.bias"].cuda())
        logits, imgf, txtf = mc(images)
    want = 100.0 * _cos(g["maple_image_features"], g["maple_text_features"])
    assert np.abs(logits.cpu().numpy() - want).max() < 100 * COS_TOL


def test_golden_vitb16():
    g = load_golden("vitb16_seed0.npz")
    sd, model = _build("ViT-B/16")
    images = syn.synthetic_images(2, "ViT-B/16", seed=0).cuda()
    ids = torch.from_numpy(g["ids"]).cuda()
    with torch.no_grad():
        img = model.image_features_f32(images)
        txt = model.text_features_f32(ids)
    _feat_close(img.cpu().numpy(), g["image_features"], "ViT-B/16 image features")
    _feat_close(txt.cpu().numpy(), g["text_features"], "ViT-B/16 text features")
    cos = _cos(img.cpu().numpy(), txt.cpu().numpy())
    assert np.abs(cos - g["logits"] / 100.0).max() < COS_TOL
    # and we are no further from fp32 than the reference's own fp16 path is (x2 slack)
    ref16_err = np.abs(g["logits_fp16"] - g["logits"]).max() / 100.0
    assert np.abs(cos - g["logits"] / 100.0).max() < max(2 * ref16_err, 2e-4)
    # fp32-converted model (clip_model.float(), coop.py:243) gives the same answer and fp32 outputs
    model.float()
    with torch.no_grad():
        e = model.encode_image(images)
    assert e.dtype == torch.float32 and torch.allclose(e, img, atol=1e-6)


@pytest.mark.parametrize("mode", ["default", "ln_fold=0", "residual_f16=0", "residual_f16=1"])
def test_golden_vitb16_outlier_statistics(clipmi_option, mode):
    """Parity where it can fall over (SURVEY §7 risk; clip/model.py:153-159,185-188): ViT-B/16 depth with massive residual
    channels (~60 against a typical 2.5), LayerNorm gains spread over an order of magnitude and a common offset, so that
    mean^2 ~ E[x^2]/2 in every row -- against the REFERENCE's fp32 outputs on the same weights (tests/golden/
    vitb16_outliers.npz).  The default path (LayerNorm folded from sum x / sum x^2 partials + fp16 image stream), the
    separate-LayerNorm path, the fp32 stream and the all-fp16 stream must each stay within 1e-3 on cosine logits and within
    twice the reference's own fp16-vs-fp32 distance on this fixture."""
    from clip_calibration_amd.model import build_model
    g = load_golden("vitb16_outliers.npz")
    if mode != "default":
        name, value = mode.split("=")
        clipmi_option(name, int(value))
    sd = syn.outlier_state_dict("ViT-B/16", seed=0)
    model = build_model(dict(sd), dict(PLAIN)).cuda()
    images = syn.synthetic_images(4, "ViT-B/16", seed=0).cuda()
    ids = torch.from_numpy(g["ids"]).cuda()
    with torch.no_grad():
        img = model.image_features_f32(images).cpu().numpy()
        txt = model.text_features_f32(ids).cpu().numpy()
    assert np.isfinite(img).all() and np.isfinite(txt).all()
    ref = _cos(g["image_features"], g["text_features"])
    ref16_err = np.abs(_cos(g["image_features_fp16"], g["text_features_fp16"]) - ref).max()
    err_both = np.abs(_cos(img, txt) - ref).max()
    err_img = np.abs(_cos(img, g["text_features"]) - ref).max()
    err_txt = np.abs(_cos(g["image_features"], txt) - ref).max()
    print(f"[{mode}] cosine-logit error: both towers {err_both:.2e}, image side {err_img:.2e}, text side {err_txt:.2e}; "
          f"reference fp16 vs fp32 {ref16_err:.2e}")
    assert max(err_both, err_img, err_txt) < COS_TOL
    assert err_both <= 2 * ref16_err + 2e-5
    _feat_close(img, g["image_features"], "image features (outlier statistics)")
    _feat_close(txt, g["text_features"], "text features (outlier statistics)")


@pytest.mark.parametrize("gname,stem", [("ViT-L/14", "vitl14"), ("ViT-L/14@336px", "vitl14_336")])
@pytest.mark.parametrize("kind", ["seed0", "outliers"])
def test_golden_vitl(clipmi_option, gname, stem, kind):
    """The round-5 kernels of the long towers on REFERENCE outputs (tests/golden/vitl14*.npz, oracle/gen_golden.py vitl: clip/model.py's own fp32 and
    fp16 forwards on seeded weights; geometry by shape inference, clip/model.py:660-665, clip/clip.py:37-38): 257 / 577 tokens go through the ring
    attention kernel (rescale with slack), width-1024 GEMMs through the streamed kernels with a ragged last tile row launched apart.  Three modes --
    default, attn_ring = 0 (the streaming attention kernel), gemm_split_rows = 0 (one GEMM launch) -- each within 1e-3 on cosine logits of the
    reference's fp32 answer and no further from it than twice the reference's own fp16 path; with trained-CLIP-like outlier statistics too."""
    from clip_calibration_amd.model import build_model
    g = load_golden(f"{stem}_{kind}.npz")
    sd = (syn.outlier_state_dict if kind == "outliers" else syn.synthetic_state_dict)(gname, seed=0)
    model = build_model(dict(sd), dict(PLAIN)).cuda()
    images = syn.synthetic_images(2, gname, seed=0).cuda()
    ids = torch.from_numpy(g["ids"]).cuda()
    ref = _cos(g["image_features"], g["text_features"])
    ref16_err = np.abs(_cos(g["image_features_fp16"], g["text_features_fp16"]) - ref).max()
    for mode in ("default", "attn_ring=0", "gemm_split_rows=0", "cls_only_last_block=0"):
        if mode != "default":
            name, value = mode.split("=")
            clipmi_option(name, int(value))
        with torch.no_grad():
            img = model.image_features_f32(images).cpu().numpy()
            txt = model.text_features_f32(ids).cpu().numpy()
        if mode != "default":
            clipmi_option(name, 1)
        assert np.isfinite(img).all() and np.isfinite(txt).all()
        err_both = np.abs(_cos(img, txt) - ref).max()
        err_img = np.abs(_cos(img, g["text_features"]) - ref).max()
        print(f"[{gname} {kind} {mode}] cosine-logit error: both towers {err_both:.2e}, image side {err_img:.2e}; reference fp16 vs fp32 {ref16_err:.2e}")
        assert max(err_both, err_img) < COS_TOL
        # twice the reference's own fp16-vs-fp32 distance, with the floor test_golden_vitb16 uses: the distance is a maximum over 2 x 4 logits, and on
        # the plain ViT-L/14 fixture it happens to be 4.8e-5 (1.5e-4 .. 2.6e-4 on the other three); this path measures 1.2 .. 1.5e-4 there
        assert err_both <= max(2 * ref16_err + 2e-5, 2e-4)
        _feat_close(img, g["image_features"], f"{gname} image features ({kind}, {mode})")
        _feat_close(txt, g["text_features"], f"{gname} text features ({kind}, {mode})")


@pytest.mark.parametrize("B", [1, 5, 32])
def test_zeroshot_pipeline_vs_oracle(B):
    """BASELINE config 1: ViT-B/16, C=100 prompts, synthetic batch; logits, (conf, pred) and ECE vs the CPU path."""
    from clip_calibration_amd.trainers import ZeroshotCLIP
    from clip_calibration_amd.evaluator import DeviceCalibrationEvaluator
    C = 100
    sd, model = _build("ViT-B/16")
    ids = syn.synthetic_token_ids(C, "ViT-B/16", seed=0)
    images = syn.synthetic_images(B, "ViT-B/16", seed=B)
    zs = ZeroshotCLIP(model, ids)
    logits, imf, txf, conf, pred = zs.model_inference(images.cuda(), want_conf_pred=True)
    # oracle
    with torch.no_grad():
        txt_ref = orc.l2_normalize(orc.encode_text(sd, ids))
        lg_ref, img_ref, _ = orc.zeroshot_inference(sd, images, txt_ref)
    assert np.abs(txf.cpu().numpy() @ txt_ref.numpy().T - (txt_ref @ txt_ref.t()).numpy()).max() < COS_TOL
    assert np.abs(logits.cpu().numpy() - lg_ref.numpy()).max() < 100 * COS_TOL
    assert np.abs(imf.cpu().numpy().astype(np.float64) ** 2).sum(1) == pytest.approx(1.0, abs=1e-5)
    labels = syn.synthetic_labels(torch.from_numpy(lg_ref.numpy().argmax(1)), C, seed=B)
    ece_ref, c_ref, p_ref = orc.calibrated_ece(lg_ref.numpy(), labels.numpy())
    ev = DeviceCalibrationEvaluator(10)
    ev.process(conf, pred, labels.cuda())
    res = ev.evaluate()
    assert abs(res["ece"] / 100.0 - ece_ref) < 1e-3
    # predictions may legitimately flip only where the top-2 cosine gap is inside the tolerance
    flips = np.nonzero(pred.cpu().numpy() != p_ref)[0]
    for i in flips:
        top2 = np.sort(lg_ref.numpy()[i])[-2:]
        assert top2[1] - top2[0] < 2 * 100 * COS_TOL


def test_coop_per_batch_towers_on_two_streams_same_bits():
    """CoOpCLIP(cache_text_features=False) -- the reference's schedule, prompt learner + text tower on every batch (coop.py:208-210) --
    issues the text tower on a side stream beside the image tower (CustomCLIP.towers): independent work on separate workspaces, joined
    before the logits.  Outputs are bit-identical to the one-stream order, batch after batch (8 batches: a stale or half-written text
    feature row would show), and to a second model-level call after a ctx update."""
    from clip_calibration_amd.trainers import CoOpCLIP
    sd, model = _build("tiny")
    ids = syn.synthetic_token_ids(24, "tiny", seed=5, n_ctx_placeholders=4)
    a = CoOpCLIP(model, ids, n_ctx=4, seed=1, cache_text_features=False)
    b = CoOpCLIP(model, ids, n_ctx=4, seed=1, cache_text_features=False)
    b.overlap_towers = False
    for k in range(8):
        images = syn.synthetic_images(9, "tiny", seed=20 + k).cuda()
        if k == 4:
            with torch.no_grad():
                a.prompt_learner.ctx.mul_(1.5)
                b.prompt_learner.ctx.mul_(1.5)
        la, ia, ta, ca, pa = a(images, want_conf_pred=True)
        lb, ib, tb, cb, pb = b(images, want_conf_pred=True)
        torch.cuda.synchronize()
        assert torch.equal(la, lb) and torch.equal(ta, tb) and torch.equal(ia, ib) and torch.equal(ca, cb) and torch.equal(pa, pb), f"batch {k}"


@pytest.mark.parametrize("how", ["fresh", "rebind", "load_state_dict"])
def test_coop_two_stream_towers_first_call_after_a_bind(how):
    """The FIRST thing a freshly built (or re-bound) model sees is `CustomCLIP.towers()` with the text tower on its side stream: weight binding is lazy
    and packs BOTH towers' operands with torch ops on the current stream, so it must happen on the caller's stream before the fork -- otherwise the
    image tower, launched on the caller's stream right behind, could read vision operands that the side stream is still folding.  Twenty fresh binds,
    each against a model that was bound and warmed up on one stream; the bits must agree."""
    from clip_calibration_amd.trainers import CoOpCLIP
    sd, ref_model = _build("tiny")
    ids = syn.synthetic_token_ids(24, "tiny", seed=5, n_ctx_placeholders=4)
    ref = CoOpCLIP(ref_model, ids, n_ctx=4, seed=1, cache_text_features=False)
    ref.overlap_towers = False
    images = syn.synthetic_images(9, "tiny", seed=21).cuda()
    want = ref(images, want_conf_pred=True)
    torch.cuda.synchronize()
    _, model = _build("tiny")
    a = CoOpCLIP(model, ids, n_ctx=4, seed=1, cache_text_features=False)
    for trial in range(20):
        if how == "fresh":
            _, model = _build("tiny")
            a = CoOpCLIP(model, ids, n_ctx=4, seed=1, cache_text_features=False)
        elif how == "rebind":
            model.rebind()
        else:
            model.load_state_dict(model.state_dict())
        assert model._bound is None
        got = a(images, want_conf_pred=True)          # first call after the bind was dropped: goes through towers() on two streams
        torch.cuda.synchronize()
        for g, w in zip(got, want):
            assert torch.equal(g, w), f"{how}, trial {trial}"


@pytest.mark.parametrize("cached", [True, False])
def test_coop_dac_tempscaling_pipeline_vs_oracle(cached):
    """BASELINE config 3 (scaled down in C): CoOp prompts -> cached text features; DAC fit on the four text-feature
    sets; cosine base model + TempScaling scalar; DAC per-sample scale fused into the logits kernel.  cached = False is the
    reference's schedule (text tower on every batch, coop.py:208-210), which the mirror runs on the fp16 residual stream
    (per-call flag): same tolerances against the fp32 oracle."""
    from clip_calibration_amd.trainers import CoOpCLIP, ZeroshotCLIP, CustomCLIPCalibration
    from clip_calibration_amd.dac import DistanseAwareCalibration
    C, B, n_ctx = 40, 16, 16
    sd, model = _build("ViT-B/16")
    ids_zs_base = syn.synthetic_token_ids(C, "ViT-B/16", seed=10)
    ids_zs_new = syn.synthetic_token_ids(C, "ViT-B/16", seed=11)
    ids_coop_base = syn.synthetic_token_ids(C, "ViT-B/16", seed=10, n_ctx_placeholders=n_ctx)
    ids_coop_new = syn.synthetic_token_ids(C, "ViT-B/16", seed=11, n_ctx_placeholders=n_ctx)
    images = syn.synthetic_images(B, "ViT-B/16", seed=7)
    coop_new = CoOpCLIP(model, ids_coop_new, n_ctx=n_ctx, logit_scale=1.0, seed=3, cache_text_features=cached)   # cosine base model
    coop_base = CoOpCLIP(model, ids_coop_base, n_ctx=n_ctx, logit_scale=1.0, seed=3, cache_text_features=cached)
    from clip_calibration_amd import _lib
    assert coop_new._text_flags() == (_lib.CALL_DEFAULT if cached else _lib.CALL_STREAM_F16)
    ctx = coop_new.prompt_learner.ctx.detach().float().cpu()
    # oracle text features
    with torch.no_grad():
        t_new = orc.l2_normalize(orc.text_encoder(sd, orc.coop_prompts(sd, ids_coop_new, ctx), ids_coop_new))
        t_base = orc.l2_normalize(orc.text_encoder(sd, orc.coop_prompts(sd, ids_coop_base, ctx), ids_coop_base))
        z_new = orc.l2_normalize(orc.encode_text(sd, ids_zs_new))
        z_base = orc.l2_normalize(orc.encode_text(sd, ids_zs_base))
    got_new = coop_new.text_features()
    if cached:
        assert coop_new.text_features() is got_new                               # cached while ctx is unchanged
    else:
        again = coop_new.text_features()                                         # recomputed, same bits
        assert again is not got_new and torch.equal(again, got_new)
    assert np.abs(got_new.cpu().numpy() @ t_new.numpy().T - (t_new @ t_new.t()).numpy()).max() < COS_TOL
    # DAC fit on device-produced features vs oracle-produced features
    zs_new = ZeroshotCLIP(model, ids_zs_new); zs_base = ZeroshotCLIP(model, ids_zs_base)
    cal = DistanseAwareCalibration()
    cal.fit(zs_base.text_features.cpu().numpy(), zs_new.text_features.cpu().numpy(),
            coop_base.text_features().cpu().numpy(), got_new.cpu().numpy(), 5)
    conf_ref = orc.dac_fit(z_base.numpy(), z_new.numpy(), t_base.numpy(), t_new.numpy(), 5)
    np.testing.assert_allclose(cal.class_confidence, conf_ref, rtol=5e-3)
    # TempScaling wrapper: scale 100 on cosine logits, DAC fused
    calib = CustomCLIPCalibration(coop_new).cuda()
    dacc = cal.class_confidence_device("cuda")
    logits, imf, txf, conf, pred = calib(images.cuda(), dac_conf=dacc, want_conf_pred=True)
    with torch.no_grad():
        lg_ref, _, _ = orc.clip_logits(orc.encode_image(sd, images), t_new, np.exp(4.6052))
    lg_ref_dac = orc.dac_predict(lg_ref.numpy(), conf_ref)
    # the DAC factor itself carries the (rtol 5e-3) difference between device- and oracle-produced text features, which at
    # |logit| ~ 30 would eat the whole logit tolerance: check the row scaling with the factor the device path actually used
    lg_ref_own = orc.dac_predict(lg_ref.numpy(), cal.class_confidence)
    # a row whose two best classes are closer than the tolerance may pick the other one on the device and then carries THAT class's
    # factor on every element: such rows are compared under the device's own choice (and must really be ties)
    lr, pd = lg_ref.numpy(), pred.cpu().numpy()
    flipped = np.nonzero(pd != lr.argmax(1))[0]
    for r in flipped:
        top2 = np.sort(lr[r])[-2:]
        assert top2[1] - top2[0] < 2 * 100 * COS_TOL, f"row {r}: prediction differs from the oracle's outside a tie"
        lg_ref_own[r] = lr[r] * cal.class_confidence[pd[r]]
    assert np.abs(logits.cpu().numpy() - lg_ref_own).max() < 100 * COS_TOL * 1.5
    lg_ref_own = orc.dac_predict(lg_ref.numpy(), cal.class_confidence)
    assert np.abs(lg_ref_own - lg_ref_dac).max() < 5e-3 * np.abs(lg_ref_dac).max() * 1.01
    labels = syn.synthetic_labels(torch.from_numpy(lg_ref_dac.argmax(1)), C, seed=1)
    # ECE over 16 samples moves by 1/16 per changed prediction: the rows whose top two classes tie within the tolerance (checked above) are
    # left out on both sides
    keep = np.setdiff1d(np.arange(B), flipped)
    assert len(keep) >= B - 2, f"{len(flipped)} tied rows"
    ece_ref, _, _ = orc.calibrated_ece(lg_ref.numpy()[keep], labels.numpy()[keep], conf_ref)
    from clip_calibration_amd.metrics import ECE
    assert abs(ECE(conf.cpu().numpy()[keep], pred.cpu().numpy()[keep], labels.numpy()[keep]) - ece_ref) < 1e-3
    # ctx update invalidates the cache
    with torch.no_grad():
        coop_new.prompt_learner.ctx.add_(0.01)
    assert coop_new.text_features() is not got_new


def test_tempscaling_sgd_step_matches_reference_gradient():
    """tempscaling.py:146-158: loss = cross_entropy(logit_scale.exp() * img @ txt^T); one SGD step on the scalar.  The
    HIP path hands over constant cosine logits and torch differentiates only the scalar product: the gradient and the
    updated scalar must equal what the reference's own statements give on the oracle's features."""
    import torch.nn.functional as F
    from clip_calibration_amd.trainers import CoOpCLIP, CustomCLIPCalibration
    sd, model = _build("tiny")
    C, B = 9, 12
    ids = syn.synthetic_token_ids(C, "tiny", seed=21, n_ctx_placeholders=4)
    images = syn.synthetic_images(B, "tiny", seed=21)
    base = CoOpCLIP(model, ids, n_ctx=4, logit_scale=1.0, seed=5)          # cosine base model (base_model/coop.py:222-224)
    calib = CustomCLIPCalibration(base).cuda()
    labels = torch.arange(B) % C
    opt = torch.optim.SGD(calib.scale_learner.parameters(), lr=0.05)
    logits, _, _ = calib.forward_train(images.cuda())
    loss = F.cross_entropy(logits, labels.cuda())
    opt.zero_grad(); loss.backward()
    grad = float(calib.scale_learner.logit_scale.grad)
    opt.step()
    # the reference's statements on the oracle's features (fp32 CPU)
    with torch.no_grad():
        ctx = base.prompt_learner.ctx.detach().float().cpu()
        txt = orc.l2_normalize(orc.text_encoder(sd, orc.coop_prompts(sd, ids, ctx), ids))
        img = orc.l2_normalize(orc.encode_image(sd, images))
    s = torch.tensor(4.6052, requires_grad=True)
    ref_loss = F.cross_entropy(s.exp() * img @ txt.t(), labels)
    ref_loss.backward()
    assert abs(float(loss) - float(ref_loss)) < 2e-3 * max(1.0, abs(float(ref_loss)))
    assert grad == pytest.approx(float(s.grad), rel=2e-2, abs=1e-4) and np.sign(grad) == np.sign(float(s.grad))
    assert float(calib.scale_learner.logit_scale) == pytest.approx(4.6052 - 0.05 * float(s.grad), abs=0.05 * 2e-2 * abs(float(s.grad)) + 1e-5)
    assert all(p.grad is None for p in model.parameters())                      # nothing else is a leaf of that graph


# (BASELINE config 5's geometries -- ViT-L/14 at 224 and 336 px -- against the CPU oracle: superseded in round 6 by test_golden_vitl, the same towers
# against the REFERENCE's outputs on the same seeded weights, with the oracle held to those fixtures in tests/test_oracle_golden.py; 31 s of CPU
# oracle work less in the GPU suite.)


@pytest.mark.parametrize("batch", [256, 300, 160, 131])
def test_row_range_residual_kernel_in_tower(clipmi_option, batch):
    """gemm_rstream_kernel (option gemm_rstream = 1, the default for the fp16-stream residual GEMMs out-proj / c_proj where every row
    range splits into tiles of 8-10 pairs) against the one-tile-per-workgroup kernel (gemm_rstream = 0).  The row-range kernel adds
    the residual to the accumulators during the K loop -- the same fp32 terms in another order, one rounding to fp16: one element in a
    thousand of a GEMM's output moves by an fp16 ulp, and through 24 residual updates the image features move by the fp16 stream's own
    noise floor (1-2e-4 per component of the unit vector, against 1e-3 of tolerance).  Bit-identical run to run.  Batch 256 (85
    ranges of 18-19 pairs: tiles of 10 + 9 / 9 + 9), 300 and 160 (ranges of 22-23 / 11-12 pairs: the tile kernel serves them), 131
    (M = 25807: one tile of 9-10 pairs per workgroup, a ragged last pair)."""
    sd, model = _build("ViT-B/16")
    images = syn.synthetic_images(batch, "ViT-B/16", seed=9).cuda()
    with torch.no_grad():
        clipmi_option("gemm_rstream", 0)
        ref = model.image_features_f32(images).clone()
        clipmi_option("gemm_rstream", 1)
        a = model.image_features_f32(images).clone()
        b = model.image_features_f32(images).clone()
    assert torch.isfinite(a).all()
    assert torch.equal(a, b), f"run-to-run max diff {float((a - b).abs().max())}"
    an, rn = torch.nn.functional.normalize(a, dim=1), torch.nn.functional.normalize(ref, dim=1)
    assert float((an - rn).abs().max()) < 5e-4 and float((1 - (an * rn).sum(1)).abs().max()) < 2e-5
    with torch.no_grad():
        o = orc.encode_image(sd, images[:2].cpu()).numpy()
    _feat_close(a[:2].cpu().numpy(), o, "image tower with the row-range residual kernel")


@pytest.mark.parametrize("n_prompts", [1000, 700])
def test_row_range_residual_kernel_text_tower(n_prompts):
    """The same on the text tower's shapes with the fp16 stream asked for per call (CLIPMI_CALL_STREAM_F16): N = 512 is two column
    tiles (128 row ranges of 18-19 pairs), K = 512 and 2048."""
    from clip_calibration_amd import _lib
    sd, model = _build("ViT-B/16")
    ids = syn.synthetic_token_ids(n_prompts, "ViT-B/16", seed=4).cuda()
    with torch.no_grad():
        with _lib.option("gemm_rstream", 0):
            ref = model.text_features_f32(ids, flags=_lib.CALL_STREAM_F16).clone()
        a = model.text_features_f32(ids, flags=_lib.CALL_STREAM_F16).clone()
        b = model.text_features_f32(ids, flags=_lib.CALL_STREAM_F16).clone()
        f32 = model.text_features_f32(ids).clone()
    assert torch.isfinite(a).all()
    assert torch.equal(a, b), f"run-to-run max diff {float((a - b).abs().max())}"
    an, rn = torch.nn.functional.normalize(a, dim=1), torch.nn.functional.normalize(ref, dim=1)
    assert float((1 - (an * rn).sum(1)).abs().max()) < 2e-5
    assert not torch.equal(a, f32)                                # the flag does select the fp16 stream


@pytest.mark.parametrize("gname,batch", [("tiny", 5), ("ViT-B/16", 8), ("ViT-B/16", 70), ("ViT-L/14", 3), ("ViT-L/14@336px", 2)])
@pytest.mark.parametrize("fold,f16", [(1, 2), (1, 0), (0, 0)])
def test_class_rows_only_last_block(clipmi_option, gname, batch, fold, f16):
    """Option cls_only_last_block (the default since round 6): the image tower's last block computes K | V for every token and everything else --
    the Q third of the in-projection, attention (one query per head: attention_cls.hip), out-proj, ln_2, c_fc, c_proj -- for the class rows alone,
    the only rows ln_post reads (clip/model.py:419): the same GEMMs with M = batch and row stride L * D.  The features must equal the every-row
    computation up to the kernels' tile choice and the attention's summation order (fp32 probabilities here, fp16 on the matrix cores there), in
    every precision mode, and match the oracle like the every-row path."""
    if gname.startswith("ViT-L") and (fold, f16) != (1, 2):
        pytest.skip("the long towers: default precision mode only (suite time)")
    from clip_calibration_amd import _lib
    assert _lib.get_option("cls_only_last_block") == 1, "class rows only is the product default"
    clipmi_option("ln_fold", fold)
    clipmi_option("residual_f16", f16)
    sd, model = _build(gname)
    images = syn.synthetic_images(batch, gname, seed=5).cuda()
    with torch.no_grad():
        clipmi_option("cls_only_last_block", 0)
        full = model.image_features_f32(images).clone()
        clipmi_option("cls_only_last_block", 1)
        cls = model.image_features_f32(images).clone()
        cls2 = model.image_features_f32(images)
    assert torch.equal(cls, cls2)
    a, b = full.cpu().numpy(), cls.cpu().numpy()
    assert np.isfinite(b).all()
    scale = np.abs(a).max()
    assert np.abs(a - b).max() <= 2e-3 * scale, f"class-rows-only features differ: {np.abs(a - b).max()} vs scale {scale}"
    an, bn = a / np.linalg.norm(a, axis=1, keepdims=True), b / np.linalg.norm(b, axis=1, keepdims=True)
    assert np.abs((an * bn).sum(1) - 1.0).max() < 5e-6
    if batch <= 8 and not gname.startswith("ViT-L"):           # (the long towers against the reference: test_golden_vitl)
        with torch.no_grad():
            ref = orc.encode_image(sd, images.cpu()).numpy()
        _feat_close(b, ref, "class-rows-only last block")


def test_wide_tower_row_parameters_from_the_finalise_kernel(clipmi_option):
    """A vision width beyond 1024 (here 1280: five 256-column tiles per residual GEMM) gives the LayerNorm fold more row partials than the streamed
    kernel's LDS table holds (STREAM_RAW_PARTS = 4): the consumer GEMMs then read (rstd, mean * rstd) pairs that ln_finalize_kernel reduced once per GEMM,
    DMA'd per tile through the whole-array descriptor (gemm.hip `params`, round 6).  No CLIP geometry of the reference is that wide; two blocks of it
    at 40 images (in-proj 465 tiles, c_fc 620: the streamed kernel) against the tile kernels and against the oracle."""
    from clip_calibration_amd.model import build_model
    geom = syn.ClipGeometry(512, 224, 2, 1280, 16, 77, 49408, 512, 8, 1)
    sd = syn.synthetic_state_dict(geom, seed=0)
    model = build_model(dict(sd), dict(PLAIN)).cuda()
    images = syn.synthetic_images(40, geom, seed=4)
    with torch.no_grad():
        clipmi_option("cls_only_last_block", 0)
        clipmi_option("gemm_stream", 0)
        tile = model.image_features_f32(images.cuda()).clone()
        clipmi_option("gemm_stream", 1)
        stream = model.image_features_f32(images.cuda()).clone()
        clipmi_option("cls_only_last_block", 1)
        cls = model.image_features_f32(images.cuda()).clone()
        ref = orc.encode_image(sd, images[:2]).numpy()
    assert torch.isfinite(stream).all()
    # (the fold's fp32 expression is contracted differently in the two kernels: an fp16 ulp here and there on the GEMM outputs, as between any two tile shapes)
    sn, tn = torch.nn.functional.normalize(stream, dim=1), torch.nn.functional.normalize(tile, dim=1)
    assert float((1 - (sn * tn).sum(1)).abs().max()) < 2e-5 and float((stream - tile).abs().max()) <= 3e-3 * float(tile.abs().max())
    _feat_close(stream[:2].cpu().numpy(), ref, "wide tower, streamed GEMMs")
    _feat_close(cls[:2].cpu().numpy(), ref, "wide tower, class rows only")


@pytest.mark.parametrize("tower", ["maple", "ivlp", "vpt"])
def test_class_rows_only_last_block_hooked_towers(clipmi_option, tower):
    """The same on the towers that carry prompt tokens (clip/model.py:287-331, 447-478; 191-256): MaPLe's shared context + deep prompts
    (two tokens appended behind the patches, overwritten per layer), IVLP / VPT's learned per-layer prompts.  The class token stays row 0 of its
    sequence and every prompt token is a key of the last block's attention; class rows only == every row, and both match the reference's outputs."""
    from clip_calibration_amd.model import build_model
    if tower == "maple":
        g = load_golden("tiny_clip.npz")
        sd, model = _build("tiny", dict(PLAIN, trainer="MaPLe", maple_length=2))
        pl = {k[len("maple_pl:"):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("maple_pl:")}
        _, shared, _, deep_v = orc.maple_prompt_learner(sd, torch.from_numpy(g["maple_ids"]), pl)
        images = torch.from_numpy(g["images"]).cuda()
        run = lambda: model.image_features_f32(images, shared.cuda(), [t.cuda() for t in deep_v])
        want = g["maple_image_features"]
    else:
        dd = ({"trainer": "IVLP", "vision_depth": 3, "language_depth": 3, "vision_ctx": 2, "language_ctx": 2} if tower == "ivlp" else
              {"trainer": "VPT", "vision_depth": 2, "language_depth": 0, "vision_ctx": 4, "language_ctx": 0})
        g = load_golden("tiny3_clip.npz")
        model = build_model(dict(syn.synthetic_state_dict("tiny3", seed=0)), dict(dd))
        own = {k for k in model.state_dict() if "VPT" in k}
        model.load_state_dict({k: torch.from_numpy(g[f"{tower}_sd:{k}"]) for k in own}, strict=False)
        model = model.cuda()
        images = torch.from_numpy(g["images"]).cuda()
        run = lambda: model.image_features_f32(images)
        want = g[tower + "_image_features"]
    with torch.no_grad():
        clipmi_option("cls_only_last_block", 0)
        full = run().cpu().numpy()
        clipmi_option("cls_only_last_block", 1)
        cls = run().cpu().numpy()
    assert np.isfinite(cls).all()
    an, bn = full / np.linalg.norm(full, axis=1, keepdims=True), cls / np.linalg.norm(cls, axis=1, keepdims=True)
    assert np.abs((an * bn).sum(1) - 1.0).max() < 5e-6
    _feat_close(cls, want, f"{tower} image tower, class rows only in the last block")
    _feat_close(full, want, f"{tower} image tower, every row")


@pytest.mark.parametrize("gname", ["tiny", "ViT-B/16"])
def test_layernorm_fold_path(clipmi_option, gname):
    """Default path: ln_1 / ln_2 applied inside the GEMM epilogues (gamma folded into the weights, mean / rstd from
    per-tile row partials), bit-reproducible run to run; option ln_fold = 0 = separate LayerNorm kernels.  Both within the
    same tolerance of the oracle and of each other."""
    clipmi_option("ln_fold", 1)
    clipmi_option("residual_f16", 2)
    sd, model = _build(gname)
    images = syn.synthetic_images(3, gname, seed=3)
    ids = syn.synthetic_token_ids(6, gname, seed=3)
    with torch.no_grad():
        a = model.image_features_f32(images.cuda())
        b = model.image_features_f32(images.cuda())
        t = model.text_features_f32(ids.cuda())
        ref_i = orc.encode_image(sd, images).numpy()
        ref_t = orc.encode_text(sd, ids).numpy()
    assert torch.equal(a, b)
    _feat_close(a.cpu().numpy(), ref_i, "folded image tower")
    _feat_close(t.cpu().numpy(), ref_t, "folded text tower")
    # residual-stream precision: image tower fp16 by default (the reference's own GPU precision), fp32 on request
    clipmi_option("residual_f16", 0)
    with torch.no_grad():
        a32 = model.image_features_f32(images.cuda())
    assert not torch.equal(a, a32)
    _feat_close(a32.cpu().numpy(), ref_i, "folded image tower, fp32 stream")
    an, a32n, rn = (x / np.linalg.norm(x, axis=1, keepdims=True) for x in (a.cpu().numpy(), a32.cpu().numpy(), ref_i))
    assert np.abs(a32n @ rn.T - rn @ rn.T).max() <= np.abs(an @ rn.T - rn @ rn.T).max() + 2e-5      # fp32 stream is at least as close
    clipmi_option("ln_fold", 0)
    with torch.no_grad():
        c = model.image_features_f32(images.cuda())
        tc = model.text_features_f32(ids.cuda())
    assert not torch.equal(a, c)                                   # really a different code path
    _feat_close(c.cpu().numpy(), ref_i, "unfolded image tower")
    _feat_close(tc.cpu().numpy(), ref_t, "unfolded text tower")
    assert np.abs(_cos(a.cpu().numpy(), c.cpu().numpy()) - _cos(c.cpu().numpy(), c.cpu().numpy())).max() < 5e-4


def test_kgcoop_mirror_and_reference_style_zeroshot():
    """KgCoOp eval forward (kgcoop.py:246-259: CoOp with ctx taken from the embedding of "a photo of a", n_ctx = 4) and the
    reference's own ZeroshotCLIP.model_inference statements (zsclip.py:97-102) executed on the attribute surface."""
    from clip_calibration_amd.trainers import KgCoOpCLIP
    sd, model = _build("tiny")
    C = 6
    ids_ctx = syn.synthetic_token_ids(C, "tiny", seed=4, n_ctx_placeholders=4)
    ids_zs = syn.synthetic_token_ids(C, "tiny", seed=4)
    full = syn.synthetic_token_ids(1, "tiny", seed=4)            # [SOT, a, photo, of, a, name.., ., EOT, 0..]
    V = syn.GEOMETRIES["tiny"].vocab_size
    init = torch.zeros(1, full.shape[1], dtype=torch.long)       # what clip.tokenize("a photo of a") returns: zero-padded [1, 77]
    init[0, :6] = torch.tensor([V - 2] + full[0, 1:5].tolist() + [V - 1])
    kg = KgCoOpCLIP(model, ids_ctx, zeroshot_tokenized_prompts=ids_zs, ctx_init_ids=init)
    assert kg.prompt_learner.n_ctx == 4 and kg.prompt_learner.token_suffix.shape[1] == full.shape[1] - 5
    images = syn.synthetic_images(3, "tiny", seed=4)
    logits, imf, txf = kg(images.cuda())
    ctx = sd["token_embedding.weight"][init[0, 1:5]].half().float()
    with torch.no_grad():
        tf_ref = orc.l2_normalize(orc.text_encoder(sd, orc.coop_prompts(sd, ids_ctx, ctx), ids_ctx))
        lg_ref, _, _ = orc.clip_logits(orc.encode_image(sd, images), tf_ref, sd["logit_scale"].exp())
        zs_ref = orc.l2_normalize(orc.encode_text(sd, ids_zs))
    assert np.abs(logits.cpu().numpy() - lg_ref.numpy()).max() < 100 * COS_TOL
    assert np.abs(kg.ori_embedding.cpu().numpy() @ zs_ref.numpy().T - (zs_ref @ zs_ref.t()).numpy()).max() < COS_TOL
    # zsclip.py:97-102 verbatim, fp16 tensors as the reference would hold them
    clip_model = model
    text_features = clip_model.encode_text(ids_zs.cuda())
    text_features = text_features / text_features.norm(dim=-1, keepdim=True)
    image = images.cuda()
    image_features = clip_model.encode_image(image)
    image_features = image_features / image_features.norm(dim=-1, keepdim=True)
    logit_scale = clip_model.logit_scale.exp()
    lg = logit_scale * image_features @ text_features.t()
    with torch.no_grad():
        want, _, _ = orc.clip_logits(orc.encode_image(sd, images), orc.encode_text(sd, ids_zs), sd["logit_scale"].exp())
    assert lg.dtype == torch.float16
    assert np.abs(lg.float().detach().cpu().numpy() - want.numpy()).max() < 0.25   # fp16 end to end at scale 100: the reference's own noise floor
    # ln_final on a non-contiguous view and a 2-D input
    x = torch.randn(4, 77, 128, device="cuda").half().permute(1, 0, 2)
    y = clip_model.ln_final(x)
    ref = orc.layer_norm(x.float().cpu(), sd["ln_final.weight"], sd["ln_final.bias"])
    assert y.shape == x.shape and np.abs(y.float().cpu().numpy() - ref.numpy()).max() < 5e-3


@pytest.mark.parametrize("tag,dd", [
    ("ivlp", {"trainer": "IVLP", "vision_depth": 3, "language_depth": 3, "vision_ctx": 2, "language_ctx": 2}),
    ("vpt", {"trainer": "VPT", "vision_depth": 2, "language_depth": 0, "vision_ctx": 4, "language_ctx": 0})])
def test_golden_ivlp_vpt_designs(tag, dd):
    """f-4: build_model(..., design_details IVLP / VPT) creates the reference's per-layer prompt parameters
    (clip/model.py:191-256, 361-381), loads them by name, and both towers splice them -- against outputs of the
    reference's own IVLP model (tests/golden/tiny3_clip.npz)."""
    g = load_golden("tiny3_clip.npz")
    sd = syn.synthetic_state_dict("tiny3", seed=0)
    plain_keys = set(sd)
    from clip_calibration_amd.model import build_model
    model = build_model(dict(sd), dict(dd))                       # prompts missing from the dict: non-strict fallback
    own = {k for k in model.state_dict() if "VPT" in k}
    want = {k.split(":", 1)[1] for k in g if k.startswith(tag + "_sd:")}
    assert own == want and not (own & plain_keys)
    assert all(model.state_dict()[k].dtype == torch.float32 for k in own)       # convert_weights leaves them fp32
    model.load_state_dict({k: torch.from_numpy(g[f"{tag}_sd:{k}"]) for k in own}, strict=False)
    model = model.cuda()
    images, ids = torch.from_numpy(g["images"]).cuda(), torch.from_numpy(g[tag + "_ids"]).cuda()
    with torch.no_grad():
        img = model.encode_image(images).float().cpu().numpy()
        txt = model.encode_text(ids).float().cpu().numpy()
        img32 = model.image_features_f32(images).cpu().numpy()
    ri, rt = g[tag + "_image_features"], g[tag + "_text_features"]
    cos = lambda a, b: (a / np.linalg.norm(a, axis=1, keepdims=True)) @ (b / np.linalg.norm(b, axis=1, keepdims=True)).T
    assert np.abs(cos(img32, ri) - cos(ri, ri)).max() < COS_TOL
    assert np.abs(cos(txt, rt) - cos(rt, rt)).max() < 2 * COS_TOL             # through the fp16 return dtype
    assert np.abs(cos(img, ri) - cos(ri, ri)).max() < 2 * COS_TOL
    assert np.abs(img32 - g["image_features"]).max() > 1e-3                     # and not the plain tower's answer
    # the reference's TextEncoder statements (vpt.py:56-67) on the attribute surface
    x = model.token_embedding(ids).type(model.dtype) + model.positional_embedding.type(model.dtype)
    x = model.transformer(x.permute(1, 0, 2)).permute(1, 0, 2)
    x = model.ln_final(x).type(model.dtype)
    tf = (x[torch.arange(x.shape[0]), ids.argmax(dim=-1)] @ model.text_projection).float().detach().cpu().numpy()
    assert np.abs(cos(tf, rt) - cos(rt, rt)).max() < 5 * COS_TOL


def test_golden_cocoop_and_chunking():
    """f-4: CoCoOp eval forward (cocoop.py:154-199) -- meta-net shift, B*C prompts through the text tower, per-image
    normalise + dot -- against the fixture produced with the reference model's sub-modules, and across chunk sizes."""
    from clip_calibration_amd.trainers import CoCoOpCLIP
    g = load_golden("tiny_clip.npz")
    sd, model = _build("tiny")
    ids = torch.from_numpy(g["coop_ids"])
    images = torch.from_numpy(g["images"]).cuda()
    ref_txt = g["cocoop_text_features"]                                    # [B,C,E] un-normalised
    outs = []
    for per_call in (4096, 5, 7):                                          # all images at once / one image / ragged chunks
        co = CoCoOpCLIP(model, ids, n_ctx=4, prompts_per_call=per_call)
        co.prompt_learner.load_state_dict({k.split(":", 1)[1]: torch.from_numpy(v) for k, v in g.items() if k.startswith("cocoop_pl:")},
                                          strict=False)
        logits, imf, txf, conf, pred = co(images, want_conf_pred=True)
        outs.append(logits.cpu().numpy())
        assert np.abs(logits.cpu().numpy() - g["cocoop_logits"]).max() < 100 * COS_TOL
        ctx_shifted = co.prompt_learner(imf).cpu().numpy()
        assert np.abs(ctx_shifted - g["cocoop_ctx_shifted"]).max() < 1e-4
        last = ref_txt[-1] / np.linalg.norm(ref_txt[-1], axis=1, keepdims=True)
        assert np.abs(txf.cpu().numpy() @ last.T - last @ last.T).max() < COS_TOL      # the LAST image's text features
        probs = orc.softmax_probs(g["cocoop_logits"].astype(np.float64))
        rc, rp = orc.conf_pred(probs)
        # d conf <= |d logit| / 4 (softmax slope) with |d logit| < 100 * COS_TOL
        assert np.array_equal(pred.cpu().numpy(), rp) and np.abs(conf.cpu().numpy() - rc).max() < 100 * COS_TOL / 4
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])       # chunking is bitwise neutral


def test_cocoop_vs_oracle_with_dac_larger_batch():
    from clip_calibration_amd.trainers import CoCoOpCLIP
    sd, model = _build("tiny")
    B, C = 9, 13
    ids = syn.synthetic_token_ids(C, "tiny", seed=40, n_ctx_placeholders=4)
    images = syn.synthetic_images(B, "tiny", seed=40)
    co = CoCoOpCLIP(model, ids, n_ctx=4, prompts_per_call=40, seed=5)
    with torch.no_grad():                                                  # a meta-net that actually moves the context
        for p in co.prompt_learner.meta_net.parameters():
            p.copy_((torch.randn(p.shape, generator=torch.Generator().manual_seed(p.numel())) * 0.2).to(p))
    pl = {k: v.detach().float().cpu() for k, v in co.prompt_learner.state_dict().items()}
    dac = torch.linspace(0.6, 1.4, C)
    logits, imf, txf, conf, pred = co(images.cuda(), dac_conf=dac.cuda(), want_conf_pred=True)
    with torch.no_grad():
        r_logits, r_f, _ = orc.cocoop_forward(sd, pl, images, ids)
    r_scaled = orc.dac_predict(r_logits.numpy(), dac.numpy())
    # DAC scales a row by the factor of its ARG-MAX class: rows whose two best raw logits are closer than the logit
    # tolerance may legitimately pick the other factor -- judge those on the raw logits only
    top2 = np.sort(r_logits.numpy(), axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 2 * 100 * COS_TOL
    assert clear.sum() >= B // 2
    got = logits.cpu().numpy()
    assert np.abs(got[clear] - r_scaled[clear]).max() < 100 * COS_TOL * 1.5
    for b in np.nonzero(~clear)[0]:
        assert min(np.abs(got[b] - r_logits.numpy()[b] * f).max() for f in dac.numpy()[np.argsort(r_logits.numpy()[b])[-2:]]) < 100 * COS_TOL * 1.5
    rc, rp = orc.conf_pred(orc.softmax_probs(r_scaled.astype(np.float64)))
    assert np.array_equal(pred.cpu().numpy()[clear], rp[clear])
    assert np.abs(imf.cpu().numpy() @ r_f.numpy().T - (r_f @ r_f.t()).numpy()).max() < COS_TOL


def test_cocoop_text_stream_f16_at_depth():
    """CoCoOp runs its text tower -- the hot path there -- on the fp16 residual stream (text_stream_f16, default on; the
    reference's GPU precision, clip/model.py:186-187).  At ViT-B/16 depth: against the fp32-stream setting and against the
    oracle, in cosine-logit terms; a model set to residual_f16 = 0 is respected (bit-identical to text_stream_f16 = False).  The fp16
    stream is a per-call flag of the tower call (CLIPMI_CALL_STREAM_F16): the trainer writes no library option."""
    from clip_calibration_amd import _lib
    from clip_calibration_amd.trainers import CoCoOpCLIP
    sd, model = _build("ViT-B/16")
    B, C = 3, 24
    ids = syn.synthetic_token_ids(C, "ViT-B/16", seed=41, n_ctx_placeholders=4)
    images = syn.synthetic_images(B, "ViT-B/16", seed=41)
    outs = {}
    for f16 in (True, False):
        co = CoCoOpCLIP(model, ids, n_ctx=4, prompts_per_call=48, seed=5, text_stream_f16=f16)
        with torch.no_grad():
            for p in co.prompt_learner.meta_net.parameters():
                p.copy_((torch.randn(p.shape, generator=torch.Generator().manual_seed(p.numel())) * 0.05).to(p))
        outs[f16] = co(images.cuda())[0].cpu().numpy() / co.scale
        if f16:
            pl = {k: v.detach().float().cpu() for k, v in co.prompt_learner.state_dict().items()}
            model.set_option("residual_f16", 0)                                  # per MODEL: no process-wide state is touched
            fp32_everywhere = co(images.cuda())[0].cpu().numpy() / co.scale
            co.text_stream_f16 = False
            assert np.array_equal(co(images.cuda())[0].cpu().numpy() / co.scale, fp32_everywhere)
            model.set_option("residual_f16", -1)
            co.text_stream_f16 = True
    assert _lib.get_option("residual_f16") == 2 and model.get_option("residual_f16") == 2   # nothing global was written
    assert np.abs(outs[True] - outs[False]).max() < COS_TOL / 2
    assert np.abs(outs[True] - outs[False]).max() > 0                            # the switch does switch
    with torch.no_grad():
        r_logits, _, _ = orc.cocoop_forward(sd, pl, images, ids)
    ref = r_logits.numpy() / float(sd["logit_scale"].exp())
    assert np.abs(outs[True] - ref).max() < COS_TOL and np.abs(outs[False] - ref).max() < COS_TOL


def test_promptsrc_and_vpt_mirrors():
    """PromptSRC eval forward = CoOp splice through an IVLP-design CLIP (promptsrc.py:186-214); VPT = fixed hand-written
    text embeddings + a VPT-design image tower (vpt.py:94-116).  Oracle with the model's own prompt tokens."""
    from clip_calibration_amd.model import build_model
    from clip_calibration_amd.trainers import PromptSRCCLIP, VPTCLIP, CoOpCLIP
    sd = syn.synthetic_state_dict("tiny3", seed=0)
    images = syn.synthetic_images(4, "tiny3", seed=50)
    C = 6
    # --- PromptSRC
    dd = {"trainer": "IVLP", "vision_depth": 3, "language_depth": 2, "vision_ctx": 3, "language_ctx": 4}
    torch.manual_seed(7)
    model = build_model(dict(sd), dict(dd)).cuda()
    ids = syn.synthetic_token_ids(C, "tiny3", seed=50, n_ctx_placeholders=4)
    ps = PromptSRCCLIP(model, ids, n_ctx=4, seed=9)
    full = dict(sd); full.update({k: v.detach().float().cpu() for k, v in model.state_dict().items() if "VPT" in k})
    ctx = ps.prompt_learner.ctx.detach().float().cpu()
    logits, imf, txf = ps(images.cuda())
    _, _, deep_t, n_ctx_t = orc.ivlp_prompts(full)
    with torch.no_grad():
        r_t = orc.l2_normalize(orc.text_encoder(full, orc.coop_prompts(full, ids, ctx), ids, torch.float32, deep_t, n_ctx_t))
        r_l, r_i, _ = orc.clip_logits(orc.encode_image_ivlp(full, images), r_t, sd["logit_scale"].exp())
    assert np.abs(logits.cpu().numpy() - r_l.numpy()).max() < 100 * COS_TOL
    cached = ps.text_features()
    with torch.no_grad():
        model.transformer.resblocks[1].VPT_shallow.add_(0.05)            # a text-side model token changes -> cache must miss
    assert ps.text_features() is not cached
    with pytest.raises(ValueError):
        PromptSRCCLIP(_build("tiny3")[1], ids)                            # plain design: refused
    # --- VPT
    dd = {"trainer": "VPT", "vision_depth": 2, "language_depth": 0, "vision_ctx": 4, "language_ctx": 0}
    model = build_model(dict(sd), dict(dd)).cuda()
    ids_zs = syn.synthetic_token_ids(C, "tiny3", seed=51)
    vp = VPTCLIP(model, ids_zs)
    full = dict(sd); full.update({k: v.detach().float().cpu() for k, v in model.state_dict().items() if "VPT" in k})
    logits, imf, txf = vp(images.cuda())
    with torch.no_grad():
        r_l, _, _ = orc.clip_logits(orc.encode_image_ivlp(full, images), orc.encode_text(full, ids_zs), sd["logit_scale"].exp())
    assert np.abs(logits.cpu().numpy() - r_l.numpy()).max() < 100 * COS_TOL
    assert vp.fixed_embeddings is txf


def test_proda_and_prograd_mirrors():
    """ProDA: prompt-collection classifier with front / middle / end class positions (proda.py:146-222, 316-333);
    ProGrad: CoOp's forward (prograd.py:272-289) plus the zero-shot teacher."""
    from clip_calibration_amd.trainers import ProDACLIP, ProGradCLIP, CoOpCLIP
    from clip_calibration_amd.trainers.prograd import CLIP as ProGradTeacher
    sd, model = _build("tiny")
    C, n_ctx, P = 7, 4, 8
    ids = syn.synthetic_token_ids(C, "tiny", seed=60, n_ctx_placeholders=n_ctx)
    images = syn.synthetic_images(5, "tiny", seed=60)
    pd = ProDACLIP(model, ids, n_ctx=n_ctx, n_prompt=P, seed=4, prompts_per_call=20)      # ragged tower calls: 20 + 20 + 16
    with torch.no_grad():
        pd.prompt_learner.ctx.mul_(25.0)            # make the context matter (0.02-sigma init barely moves the features)
    assert pd.prompt_learner.pos.tolist() == [0, 0, 1, 1, 2, 2, 2, 2]
    ctx = pd.prompt_learner.ctx.detach().float().cpu()
    logits, imf, txf = pd(images.cuda())
    with torch.no_grad():
        r_cls = orc.proda_classifier(sd, ids, ctx)
        r_l, r_i, _ = orc.clip_logits(orc.encode_image(sd, images), r_cls, sd["logit_scale"].exp())
        # clip_logits normalises its text argument; ProDA's mean is NOT normalised: redo the last step by hand
        r_l = sd["logit_scale"].exp() * r_i @ r_cls.t()
    assert np.abs(txf.cpu().numpy() - r_cls.numpy()).max() < COS_TOL
    assert np.abs(logits.cpu().numpy() - r_l.numpy()).max() < 100 * COS_TOL
    assert float(txf.norm(dim=1).max()) < 1.0           # a mean of unit vectors
    # position matters: with every prompt at the end the classifier differs
    with torch.no_grad():
        end_only = orc.l2_normalize(orc.text_encoder(sd, orc.coop_prompts(sd, ids, ctx[0]), ids))
    assert np.abs(end_only.numpy() - r_cls.numpy()).max() > 1e-2
    # ProGrad
    assert ProGradCLIP is CoOpCLIP
    teacher = ProGradTeacher(model, syn.synthetic_token_ids(C, "tiny", seed=61))
    lg, _, tf = teacher(images.cuda())
    with torch.no_grad():
        want, _, _ = orc.clip_logits(orc.encode_image(sd, images), orc.encode_text(sd, syn.synthetic_token_ids(C, "tiny", seed=61)),
                                     sd["logit_scale"].exp())
    assert np.abs(lg.cpu().numpy() - want.numpy()).max() < 100 * COS_TOL


def test_clip_adapter_and_taskres_mirrors():
    """CLIP-Adapter: bias-free bottleneck blended into the image features (clip_adapter.py:138-187); TaskRes: base text
    features (mean over templates) + alpha * residual (taskres.py:96-210).  Checked against plain fp32 restatements."""
    from clip_calibration_amd.trainers import CLIPAdapterCLIP, TaskResCLIP
    sd, model = _build("tiny")
    C, n_ctx = 5, 4
    images = syn.synthetic_images(4, "tiny", seed=80)
    # --- CLIP-Adapter
    ids = syn.synthetic_token_ids(C, "tiny", seed=80, n_ctx_placeholders=n_ctx)
    ad = CLIPAdapterCLIP(model, ids, n_ctx=n_ctx, ratio=0.2, seed=6)
    with torch.no_grad():
        for p in ad.adapter.parameters():
            p.copy_((torch.randn(p.shape, generator=torch.Generator().manual_seed(p.numel())) * 0.3).to(p))
    w1, w2 = ad.adapter.fc[0].weight.detach().float().cpu(), ad.adapter.fc[2].weight.detach().float().cpu()
    ctx = ad.prompt_learner.ctx.detach().float().cpu()
    logits, imf, txf = ad(images.cuda())
    with torch.no_grad():
        f = orc.encode_image(sd, images)
        f = 0.2 * torch.relu(torch.relu(f @ w1.t()) @ w2.t()) + 0.8 * f                   # clip_adapter.py:170-172
        t = orc.l2_normalize(orc.text_encoder(sd, orc.coop_prompts(sd, ids, ctx), ids))
        want, fi, _ = orc.clip_logits(f, t, sd["logit_scale"].exp())
    assert np.abs(logits.cpu().numpy() - want.numpy()).max() < 100 * COS_TOL
    assert np.abs(imf.cpu().numpy() @ fi.numpy().T - (fi @ fi.t()).numpy()).max() < COS_TOL
    # --- TaskRes (3 templates per class)
    T = 3
    tmpl = torch.stack([syn.synthetic_token_ids(C, "tiny", seed=81 + i) for i in range(T)], dim=1)      # [C, T, 77]
    tr = TaskResCLIP(model, tmpl, alpha=0.5)
    res = torch.randn(C, syn.GEOMETRIES["tiny"].embed_dim, generator=torch.Generator().manual_seed(3)) * 0.05
    with torch.no_grad():
        tr.prompt_learner.text_feature_residuals.copy_(res.to(tr.prompt_learner.text_feature_residuals))
    logits, imf, txf = tr(images.cuda())
    with torch.no_grad():
        base = torch.stack([orc.encode_text(sd, tmpl[:, i]) for i in range(T)], dim=1).mean(dim=1)        # taskres.py:131
        want, _, tn = orc.clip_logits(orc.encode_image(sd, images), base + 0.5 * res, sd["logit_scale"].exp())
    assert np.abs(logits.cpu().numpy() - want.numpy()).max() < 100 * COS_TOL
    assert np.abs(txf.cpu().numpy() @ tn.numpy().T - (tn @ tn.t()).numpy()).max() < COS_TOL


def test_golden_modified_resnet_tower():
    """f-4: ModifiedResNet image tower (clip/model.py:10-150) -- BatchNorm folded into NHWC GEMM convolutions, attention pool --
    against the reference's own output on the seeded checkpoint and against the oracle on other inputs; zero-shot flow on top."""
    from clip_calibration_amd.model import build_model
    from clip_calibration_amd.trainers import ZeroshotCLIP
    g = load_golden("resnet_tiny.npz")
    sd = syn.synthetic_resnet_state_dict((1, 2, 1, 1), 64, 64, "tiny", seed=0)
    model = build_model(dict(sd), None).cuda()
    geom = syn.ClipGeometry(128, 64, 1, 64, 64, 77, 256, 128, 2, 2)
    images = syn.synthetic_images(3, geom, seed=5)
    ids = torch.from_numpy(g["ids"])
    with torch.no_grad():
        img = model.image_features_f32(images.cuda()).cpu().numpy()
        txt = model.text_features_f32(ids.cuda()).cpu().numpy()
    _feat_close(img, g["image_features"], "ModifiedResNet image tower vs reference")
    _feat_close(txt, g["text_features"], "text tower of the RN checkpoint")
    assert np.abs(img - g["image_features"]).max() < 2e-2 * np.abs(g["image_features"]).max()     # magnitudes too, not only directions
    zs = ZeroshotCLIP(model, ids)
    logits, imf, txf = zs.model_inference(images.cuda())
    assert np.abs(logits.cpu().numpy() - g["logits"]).max() < 100 * COS_TOL
    assert model.encode_image(images.cuda()).dtype == torch.float16
    # other batch sizes / inputs against the oracle; batch invariance
    more = syn.synthetic_images(9, geom, seed=6)
    with torch.no_grad():
        got = model.image_features_f32(more.cuda()).cpu().numpy()
        one = model.image_features_f32(more[4:5].cuda()).cpu().numpy()
        ref = orc.encode_image_resnet(sd, more).numpy()
    _feat_close(got, ref, "ModifiedResNet vs oracle, batch 9")
    assert np.abs(one - got[4:5]).max() < 1e-3 * np.abs(got).max()
    assert model.image_features_f32(more[:0].cuda()).shape == (0, 128)


def test_rn50_geometry_vs_oracle():
    """The published RN50 shape (layers 3-4-6-3, width 64, 224 px, 7x7 + 1 pooled tokens, embed 1024) on two images."""
    from clip_calibration_amd.model import build_model
    sd = syn.synthetic_resnet_state_dict((3, 4, 6, 3), 64, 224, "RN50", seed=1)
    model = build_model(dict(sd), None).cuda()
    assert model.visual.layers_cfg == (3, 4, 6, 3) and model.visual.attnpool.num_heads == 32 and model.visual.output_dim == 1024
    images = syn.synthetic_images(2, "RN50", seed=1)
    with torch.no_grad():
        got = model.image_features_f32(images.cuda()).cpu().numpy()
        ref = orc.encode_image_resnet(sd, images).numpy()
    _feat_close(got, ref, "RN50 image tower vs oracle")


@pytest.mark.parametrize("hooked", [False, True])
def test_image_tower_passes_same_bits(clipmi_option, hooked):
    """clipmi_encode_image runs a batch beyond one and a half passes as consecutive passes on the same workspace (option vision_pass:
    stream elements per pass; throughput per image peaks at 256 images for ViT-B/16 and falls 8-10 % for 512-1024 in one pass).  The
    passes are ordinary calls on their images: on this small geometry every pass selects the same kernels as the whole batch, so the
    features must be the same bits (at full size tile and kernel choice follow the row count, and passes differ from one pass as any two
    batch sizes do: test_gpu_fullsize) -- with MaPLe-style prompt tokens too, for a ragged remainder (11 = 4 + 4 + 3) and for one that
    joins the last pass (9 = 4 + 5); the workspace the library asks for shrinks to the largest pass."""
    from clip_calibration_amd import _lib
    sd, model = _build("tiny")
    g = model.geometry
    kw = {}
    if hooked:
        gen = torch.Generator().manual_seed(3)
        kw = dict(shared_ctx=(0.1 * torch.randn(2, g.vision_width, generator=gen)).cuda(),
                  deep_prompts=[(0.1 * torch.randn(2, g.vision_width, generator=gen)).cuda() for _ in range(1)])
    L = g.vision_tokens + (2 if hooked else 0)
    for B in (11, 9, 5):
        images = syn.synthetic_images(B, "tiny", seed=B).cuda()
        clipmi_option("vision_pass", 0)
        with torch.no_grad():
            whole = model.image_features_f32(images, **kw).clone()
        need_whole = _lib.lib.clipmi_vision_workspace_bytes(model._handle, B, 2 if hooked else 0)
        clipmi_option("vision_pass", 4 * L * g.vision_width)            # 4 images per pass
        need = _lib.lib.clipmi_vision_workspace_bytes(model._handle, B, 2 if hooked else 0)
        with torch.no_grad():
            parts = model.image_features_f32(images, **kw)
        assert torch.equal(parts, whole), f"B={B}: passes changed the features"
        assert need < need_whole if B >= 6 else need == need_whole            # 5 images: less than one and a half passes -> one pass


def test_timed_tower_pass_is_the_ordinary_pass():
    """clipmi_encode_image_timed (bench.py's `us_in_tower`): the same launches as clipmi_encode_image with a hipEvent behind each -- same
    bits out, one positive interval per launch (embedding launches + 5 per layer + ln_post + proj), and a batch beyond one pass is refused
    rather than timed as something else."""
    sd, model = _build("tiny")
    g = model.geometry
    images = syn.synthetic_images(6, "tiny", seed=2).cuda()
    with torch.no_grad():
        want = model.image_features_f32(images).clone()
        t = model.image_tower_launch_us(images)
    assert torch.equal(t["features"], want)
    assert len(t["blocks"]) == g.vision_layers and all(len(b) == 5 for b in t["blocks"]) and len(t["post"]) == 2 and len(t["embed"]) >= 1
    flat = t["embed"] + [u for b in t["blocks"] for u in b] + t["post"]
    assert all(u > 0.0 for u in flat) and abs(sum(flat) - t["total_us"]) < 1e-3
    from clip_calibration_amd import _lib
    with _lib.option("vision_pass", 2 * g.vision_tokens * g.vision_width):      # 2 images per pass: 6 images are three passes
        with pytest.raises(_lib.ClipmiError):
            model.image_tower_launch_us(images)


def test_mfma_ceiling_probe_runs():
    """clipmi_probe_mfma_f16 (bench.py's `ceiling.mfma_only`): the loop runs, fills its sink with finite sums that depend on the operands,
    and reports its clocks.  A rate is not asserted here -- bench.py measures it."""
    import ctypes as C
    from clip_calibration_amd._lib import check, lib
    waves = 4
    ops_ = (torch.randn(16, waves * 64, 8, device="cuda") * 0.25).half().contiguous()
    n_cus = C.c_int(0)
    sink = torch.full((1024 * waves * 64,), float("nan"), device="cuda")
    clk = torch.zeros(2, dtype=torch.int64, device="cuda")
    check(lib.clipmi_probe_mfma_f16(ops_.data_ptr(), sink.data_ptr(), clk.data_ptr(), waves, 10, C.byref(n_cus), torch.cuda.current_stream().cuda_stream), "probe")
    torch.cuda.synchronize()
    n = n_cus.value * waves * 64
    assert n_cus.value >= 1 and torch.isfinite(sink[:n]).all() and sink[:n].abs().max() > 0 and torch.isnan(sink[n:]).all()
    assert (clk > 0).all()
    assert lib.clipmi_probe_mfma_f16(ops_.data_ptr(), sink.data_ptr(), None, 9, 10, None, None) < 0      # more waves than the probe's register budget allows


@pytest.mark.parametrize("source", ["pageable", "pinned", "device"])
def test_device_batches_order_and_values(source):
    """runner.device_batches (the host -> device leg of the test loop, base_learner.py:84-88,175-182): batches arrive in order and
    intact from pageable, pinned and device-resident sources, including a ragged last batch, while a kernel queue keeps the compute
    stream busy (the consumer must wait on the copy's event, not on the host)."""
    from clip_calibration_amd.runner import device_batches
    g = torch.Generator().manual_seed(5)
    sizes = [8, 8, 8, 3]
    host = [(torch.randn(b, 3, 32, 32, generator=g), torch.randint(0, 10, (b,), generator=g)) for b in sizes]
    if source == "pinned":
        src = [(x.pin_memory(), y.pin_memory()) for x, y in host]
    elif source == "device":
        src = [(x.cuda(), y.cuda()) for x, y in host]
    else:
        src = host
    busy = torch.randn(1 << 24, device="cuda")
    seen = []
    for (x, y) in device_batches(iter(src), "cuda"):
        busy.mul_(1.0001)                                   # work in flight on the consumer's stream
        assert x.is_cuda and y.is_cuda
        seen.append((x.sum(dim=(1, 2, 3)), x.clone(), y.clone()))
    torch.cuda.synchronize()
    assert [s[1].shape[0] for s in seen] == sizes
    for (hx, hy), (_, dx, dy) in zip(host, seen):
        assert torch.equal(dx.cpu(), hx) and torch.equal(dy.cpu(), hy)


def test_runner_base_to_new_calibration_flow(tmp_path):
    """f-1..f-3 around the path, tiny geometry: base-val feature cache -> base_features.pt round trip -> text_feature_dict
    -> VLCalibration(DAC).fit -> test() with proximity; every number against the oracle's restatement of
    base_learner.py:59-152 / vl_calibrator.py:83-109 / vl_evaluator.py:59-102."""
    from clip_calibration_amd import checkpoint as ck, metrics, runner
    from clip_calibration_amd.calibrator import VLCalibration
    from clip_calibration_amd.trainers import CoOpCLIP, ZeroshotCLIP
    sd, model = _build("tiny")
    C, n_ctx, K = 12, 4, 3
    ids_zs_b, ids_zs_n = syn.synthetic_token_ids(C, "tiny", seed=20), syn.synthetic_token_ids(C, "tiny", seed=21)
    ids_cp_b = syn.synthetic_token_ids(C, "tiny", seed=20, n_ctx_placeholders=n_ctx)
    ids_cp_n = syn.synthetic_token_ids(C, "tiny", seed=21, n_ctx_placeholders=n_ctx)
    coop_b = CoOpCLIP(model, ids_cp_b, n_ctx=n_ctx, seed=2)
    coop_n = CoOpCLIP(model, ids_cp_n, n_ctx=n_ctx, seed=2)
    # f-3: the tuned context travels through a Dassl-style checkpoint
    ck.save_checkpoint(coop_b.prompt_learner.state_dict(), str(tmp_path), "prompt_learner", 50)
    with torch.no_grad():
        coop_n.prompt_learner.ctx.zero_()
    assert ck.load_model(coop_n.prompt_learner, str(tmp_path), "prompt_learner", epoch=50) == 50
    assert torch.equal(coop_n.prompt_learner.ctx, coop_b.prompt_learner.ctx)
    ctx = coop_n.prompt_learner.ctx.detach().float().cpu()
    zs_b, zs_n = ZeroshotCLIP(model, ids_zs_b), ZeroshotCLIP(model, ids_zs_n)

    val_images = syn.synthetic_images(20, "tiny", seed=30)
    test_images = syn.synthetic_images(37, "tiny", seed=31)
    with torch.no_grad():
        r_val_f, r_test_f = orc.encode_image(sd, val_images), orc.encode_image(sd, test_images)
        r_tb = orc.l2_normalize(orc.text_encoder(sd, orc.coop_prompts(sd, ids_cp_b, ctx), ids_cp_b))
        r_tn = orc.l2_normalize(orc.text_encoder(sd, orc.coop_prompts(sd, ids_cp_n, ctx), ids_cp_n))
        r_zb, r_zn = orc.l2_normalize(orc.encode_text(sd, ids_zs_b)), orc.l2_normalize(orc.encode_text(sd, ids_zs_n))
        scale = sd["logit_scale"].exp()
        r_val_logits, r_val_n, _ = orc.clip_logits(r_val_f, r_tb, scale)
        r_test_logits, r_test_n, _ = orc.clip_logits(r_test_f, r_tn, scale)
    val_labels = syn.synthetic_labels(r_val_logits.argmax(1), C, seed=2)
    loader = lambda im, lb, bs: [(im[i:i + bs], lb[i:i + bs]) for i in range(0, len(im), bs)]

    # save_base_val_features for the tuned and the zero-shot model, through the file format
    paths = {}
    for name, infer in (("CoOp", coop_b), ("ZeroshotCLIP", zs_b.model_inference)):
        d = runner.collect_base_val_features(infer, loader(val_images, val_labels, 8), image_k=K)
        paths[name] = ck.base_features_path(str(tmp_path / "temp"), "Caltech101", name, 16, "tiny", 1)
        ck.save_base_features(paths[name], **d)
    tuned, zsd = ck.load_base_features(paths["CoOp"]), ck.load_base_features(paths["ZeroshotCLIP"])
    assert np.abs(tuned["val_logits"] - r_val_logits.numpy()).max() < 100 * COS_TOL
    assert np.array_equal(tuned["val_labels"], val_labels.numpy())
    np.testing.assert_allclose(tuned["val_image_knn_dists"], orc.val_image_knn_dists(r_val_n.numpy(), K), atol=2e-3)
    tfd = runner.text_feature_dict(zsd, zs_n.text_features, tuned, coop_n.text_features())
    cal = VLCalibration(tuned, tfd, dac_flag=True, k_dac=5)
    cal.fit()
    r_cc = orc.dac_fit(r_zb.numpy(), r_zn.numpy(), r_tb.numpy(), r_tn.numpy(), 5)
    np.testing.assert_allclose(cal.dac_calibrator.class_confidence, r_cc, rtol=5e-3)
    with pytest.raises(NotImplementedError):
        VLCalibration(tuned, tfd, base_calibration_mode="bin_based")

    # VLCalibration.predict: numpy logits -> probabilities (DAC + softmax), reference contract
    r_scaled = orc.dac_predict(r_test_logits.numpy(), r_cc)
    r_probs = orc.softmax_probs(r_scaled.astype(np.float64))
    probs = cal.predict(r_test_logits.numpy(), np.ones(len(test_images)))
    assert probs.shape == r_probs.shape and np.abs(probs - r_probs).max() < 2e-3
    np.testing.assert_allclose(probs.sum(1), 1.0, atol=1e-5)
    with pytest.raises(AssertionError):
        cal.predict(r_test_logits.numpy(), np.ones(3))

    # test(): fused loop; labels drawn so that accuracy is neither 0 nor 1
    r_conf, r_pred = orc.conf_pred(r_probs)
    test_labels = syn.synthetic_labels(torch.from_numpy(r_pred), C, seed=3)
    res = runner.test(coop_n, loader(test_images, test_labels, 16), val_dict=tuned, calibrator=cal, image_k=K)
    r_prox = np.exp(-orc.knn_dists(r_val_n.numpy(), r_test_n.numpy(), K).mean(1))
    gt = test_labels.numpy()
    assert res["total"] == 37
    assert res["accuracy"] == pytest.approx(100.0 * np.mean(r_pred == gt), abs=1e-9)
    assert res["macro_f1"] == pytest.approx(100.0 * orc.macro_f1(r_pred, gt), abs=1e-9)
    assert res["confidence"] == pytest.approx(float(r_conf.mean()), abs=1e-3)
    # binned metrics: a sample within 1e-3 of a bin edge may change bins -> allow one sample's weight on top of |dconf|
    slack = 100.0 * (1e-3 + 2.0 / 37)
    assert abs(res["ece"] - 100.0 * orc.ece(r_conf, r_pred, gt, 10)) < slack
    assert abs(res["mce"] - 100.0 * orc.mce(r_conf, r_pred, gt, 10)) < slack
    assert abs(res["ace"] - 100.0 * orc.ace(r_conf, r_pred, gt, 10)) < slack
    assert abs(res["piece"] - 100.0 * orc.piece(r_conf, r_prox, r_pred, gt, 10, 10)) < 2 * slack
    assert list(res)[:8] == ["accuracy", "error_rate", "macro_f1", "confidence", "ece", "mce", "ace", "piece"]   # vl_evaluator.py:92-99


def test_stress_residual_magnitudes():
    """Scaled-up residual branches (real CLIP has large outlier channels): still finite and within tolerance."""
    from clip_calibration_amd.model import build_model
    sd = syn.synthetic_state_dict("tiny", seed=5, gain=6.0)
    model = build_model(dict(sd), dict(PLAIN)).cuda()
    images = syn.synthetic_images(4, "tiny", seed=5)
    with torch.no_grad():
        got = model.image_features_f32(images.cuda()).cpu().numpy()
        ref = orc.encode_image(sd, images).numpy()
    assert np.isfinite(got).all()
    assert np.abs(_cos(got, ref) - _cos(ref, ref)).max() < 2e-3


def test_empty_and_ragged_batches():
    """Edge cases the reference meets at the end of a DataLoader epoch: a last batch of any size, including none at all."""
    from clip_calibration_amd import ops
    from clip_calibration_amd.evaluator import DeviceCalibrationEvaluator
    from clip_calibration_amd.proximity import knn_dists_device
    from clip_calibration_amd.trainers import ZeroshotCLIP
    sd, model = _build("tiny")
    g = syn.GEOMETRIES["tiny"]
    ids = syn.synthetic_token_ids(3, "tiny", seed=70)
    zs = ZeroshotCLIP(model, ids)
    empty = torch.empty(0, 3, g.image_resolution, g.image_resolution, device="cuda")
    f = model.image_features_f32(empty)
    assert f.shape == (0, g.embed_dim)
    assert model.encode_image(empty).shape == (0, g.embed_dim) and model.encode_image(empty).dtype == model.dtype
    assert model.text_features_f32(ids[:0].cuda()).shape == (0, g.embed_dim)
    logits, imf, txf, conf, pred = zs.model_inference(empty, want_conf_pred=True)
    assert logits.shape == (0, 3) and conf.shape == (0,) and pred.shape == (0,)
    ev = DeviceCalibrationEvaluator(10, keep_samples=True)
    ev.process(conf, pred, torch.empty(0, dtype=torch.int64))
    assert float(ev.bins.sum()) == 0.0
    assert knn_dists_device(torch.empty(0, 128, device="cuda"), torch.randn(5, 128, device="cuda"), 3).shape == (0, 3)
    # batch sizes around the kernels' internal tile edges give the same rows as the big batch (batch invariance)
    images = syn.synthetic_images(19, "tiny", seed=70).cuda()
    full = model.image_features_f32(images).cpu().numpy()
    for lo, hi in ((0, 1), (1, 4), (4, 19), (7, 8)):
        part = model.image_features_f32(images[lo:hi]).cpu().numpy()
        assert np.abs(part - full[lo:hi]).max() < 2e-3 * np.abs(full).max(), (lo, hi)
    # errors, not garbage
    with pytest.raises(ValueError):
        model.image_features_f32(torch.zeros(2, 3, g.image_resolution + 16, g.image_resolution, device="cuda"))
    with pytest.raises(ValueError):
        model.text_features_f32(torch.zeros(2, g.context_length - 1, dtype=torch.int64, device="cuda"))
    with pytest.raises(Exception):
        knn_dists_device(torch.randn(4, 128, device="cuda"), torch.randn(5, 128, device="cuda"), 6)     # K > Nr
    with pytest.raises(Exception):
        model.image_features_f32(torch.zeros(1, 3, g.image_resolution, g.image_resolution))             # host tensor: no CPU path
