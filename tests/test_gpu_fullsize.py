"""BASELINE configs[1] at FULL size on a real MI355X (ViT-B/16, batch 256, 1000 prompts): size-independent properties of
the path, plus the oracle on a subset of the rows.  The oracle cannot run 256 images x 1000 prompts in a test's time, so
the full-size run is pinned by properties that do not depend on knowing the answer:

* determinism (bitwise) and batch invariance (an image's features do not depend on what else is in the batch);
* equivariance: permuting images permutes logit rows, permuting prompts permutes logit columns (bitwise);
* linearity of the logits in the scale (exact for a power of two), DAC with unit factors is the identity;
* conf / pred are the softmax top-1 of the returned logits (pred exact);
* ECE accumulators are additive over any split of the batch (counts exact) and invariant to sample order;
* a 16-image slice of the full batch agrees with the CPU oracle to the north-star tolerance.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from clip_calibration_amd import _lib, synthetic as syn  # noqa: E402
from oracle import clip_oracle as orc  # noqa: E402  (checker only)
from conftest import oracle_text_features  # noqa: E402

COS_TOL = 1e-3
G, B, C = "ViT-B/16", 256, 1000


@pytest.fixture(scope="module")
def full():
    from clip_calibration_amd.model import build_model
    from clip_calibration_amd.trainers import ZeroshotCLIP
    sd = syn.synthetic_state_dict(G, seed=0)
    model = build_model(dict(sd), {"trainer": "ZeroshotCLIP"}).cuda()
    ids = syn.synthetic_token_ids(C, G, seed=0)
    zs = ZeroshotCLIP(model, ids)
    images = syn.synthetic_images(B, G, seed=0, device="cuda")
    with torch.no_grad():
        out = zs.model_inference(images, want_conf_pred=True)
    return dict(sd=sd, model=model, ids=ids, zs=zs, images=images, out=out)


def test_determinism_and_batch_invariance(full):
    zs, images = full["zs"], full["images"]
    logits, imf, txf, conf, pred = full["out"]
    with torch.no_grad():
        again = zs.model_inference(images, want_conf_pred=True)
    assert all(torch.equal(a, b) for a, b in zip(full["out"], again))              # bitwise, run to run
    assert logits.shape == (B, C) and torch.isfinite(logits).all()
    np.testing.assert_allclose(imf.norm(dim=1).cpu().numpy(), 1.0, atol=1e-5)
    np.testing.assert_allclose(txf.norm(dim=1).cpu().numpy(), 1.0, atol=1e-5)
    # the same images in batches of 32 / 1 / ragged: same features up to the accumulation order of a different tile shape
    with torch.no_grad():
        parts = torch.cat([zs.model_inference(images[i:i + 32])[1] for i in range(0, B, 32)])
        single = zs.model_inference(images[77:78])[1]
        ragged = zs.model_inference(images[200:256 - 13])[1]
    assert (parts - imf).abs().max() < 2e-4
    assert (single - imf[77:78]).abs().max() < 2e-4
    assert (ragged - imf[200:243]).abs().max() < 2e-4


def test_large_batch_runs_as_passes(full):
    """A batch of one and a half passes or more (option vision_pass: 256 ViT-B/16 images per pass) is worked by clipmi_encode_image as
    consecutive passes, each an ordinary call on its images: 424 images = 256 + 168 must give the bits of those two calls, and differ from
    the first 256 of the fixture's batch not at all (the same call)."""
    model, images = full["model"], full["images"]
    more = syn.synthetic_images(168, G, seed=9, device="cuda")
    both = torch.cat([images, more])
    with torch.no_grad():
        whole = model.image_features_f32(both)
        first = model.image_features_f32(images)
        second = model.image_features_f32(more)
    assert torch.equal(whole[:B], first) and torch.equal(whole[B:], second)
    assert torch.isfinite(whole).all()


def test_equivariance_and_linearity(full):
    from clip_calibration_amd import ops
    from clip_calibration_amd.trainers import ZeroshotCLIP
    zs, images, ids, model = full["zs"], full["images"], full["ids"], full["model"]
    logits, imf, txf, conf, pred = full["out"]
    g = torch.Generator().manual_seed(5)
    pi, pc = torch.randperm(B, generator=g), torch.randperm(C, generator=g)
    with torch.no_grad():
        lp = zs.model_inference(images[pi.cuda()])[0]
        zs_c = ZeroshotCLIP(model, ids[pc])
        lc = zs_c.model_inference(images)[0]
    # a row's arithmetic does not depend on its position inside a 256-row tile, nor a prompt's on its column
    assert (lp - logits[pi.cuda()]).abs().max() < 100 * 2e-4
    assert (lc - logits[:, pc.cuda()]).abs().max() < 100 * 2e-4
    assert torch.equal(zs_c.text_features, zs.text_features[pc.cuda()]) or (zs_c.text_features - zs.text_features[pc.cuda()]).abs().max() < 2e-4
    # the fused tail alone: exact equivariance, exact scale linearity, unit DAC = identity
    l1, c1, p1 = ops.logits_fused(imf, txf, zs.scale, None, True)
    assert torch.equal(l1, logits) and torch.equal(p1, pred) and torch.equal(c1, conf)
    l2, _, _ = ops.logits_fused(imf[pi.cuda()].contiguous(), txf[pc.cuda()].contiguous(), zs.scale, None, True)
    assert torch.equal(l2, logits[pi.cuda()][:, pc.cuda()])
    l3, _, p3 = ops.logits_fused(imf, txf, 2.0 * zs.scale, None, True)
    assert torch.equal(l3, 2.0 * logits) and torch.equal(p3, pred)
    l4, c4, p4 = ops.logits_fused(imf, txf, zs.scale, torch.ones(C, device="cuda"), True)
    assert torch.equal(l4, logits) and torch.equal(p4, pred)
    # the DAC row pass sums the exponentials lane-strided over the re-scaled row, the plain one blockwise (logits.hip): same value, other order
    assert (c4 - conf).abs().max() <= 4e-7 * conf.abs().max()
    # conf / pred are the softmax top-1 of the logits that came back
    lg = logits.double().cpu().numpy()
    probs = orc.softmax_probs(lg)
    rc, rp = orc.conf_pred(probs)
    assert np.array_equal(pred.cpu().numpy(), rp)
    assert np.abs(conf.cpu().numpy() - rc).max() < 1e-5


def test_ece_accumulation_is_additive(full):
    from clip_calibration_amd.evaluator import DeviceCalibrationEvaluator
    from clip_calibration_amd.metrics import ECE, MCE
    logits, imf, txf, conf, pred = full["out"]
    labels = syn.synthetic_labels(pred, C, seed=1).cuda()
    whole = DeviceCalibrationEvaluator(10, keep_samples=True)
    whole.process(conf, pred, labels)
    pieces = DeviceCalibrationEvaluator(10)
    for lo, hi in ((0, 1), (1, 100), (100, 100), (100, 256)):
        pieces.process(conf[lo:hi], pred[lo:hi], labels[lo:hi])
    shuffled = DeviceCalibrationEvaluator(10)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(9)).cuda()
    shuffled.process(conf[perm].contiguous(), pred[perm].contiguous(), labels[perm].contiguous())
    bw = whole.bins.cpu().numpy().reshape(3, 11)
    for other in (pieces, shuffled):
        bo = other.bins.cpu().numpy().reshape(3, 11)
        assert np.array_equal(bw[0], bo[0]) and np.array_equal(bw[2], bo[2])       # counts and correct-counts exactly
        np.testing.assert_allclose(bw[1], bo[1], rtol=1e-12, atol=1e-12)
    assert bw[0].sum() == B
    res = whole.evaluate()
    c, p, l = conf.cpu().numpy().astype(np.float64), pred.cpu().numpy(), labels.cpu().numpy()
    assert res["ece"] == pytest.approx(100.0 * orc.ece(c, p, l, 10), abs=1e-9)
    assert res["mce"] == pytest.approx(100.0 * orc.mce(c, p, l, 10), abs=1e-9)
    assert res["ace"] == pytest.approx(100.0 * orc.ace(c, p, l, 10), abs=1e-6)
    assert res["ece"] == pytest.approx(100.0 * ECE(c, p, l, 10), abs=1e-9) and res["mce"] == pytest.approx(100.0 * MCE(c, p, l, 10), abs=1e-9)
    assert 0.0 <= res["accuracy"] <= 100.0 and 60.0 < res["accuracy"] < 80.0       # labels agree with pred w.p. 0.7 + 0.3/C


def test_slice_of_the_full_batch_against_the_oracle(full):
    """16 rows of the 256-image batch x all 1000 prompts against the CPU oracle (north-star tolerance)."""
    sd, ids, images = full["sd"], full["ids"], full["images"]
    logits, imf, txf, conf, pred = full["out"]
    rows = torch.arange(120, 136)
    with torch.no_grad():
        ri = orc.l2_normalize(orc.encode_image(sd, images[rows.cuda()].cpu()))
        rt = orc.l2_normalize(oracle_text_features(sd, ids))      # (the oracle on the context cut behind the last EOT: conftest.py)
        r_logits = (sd["logit_scale"].exp() * ri) @ rt.t()
    got = logits[rows.cuda()].cpu().numpy()
    scale = float(sd["logit_scale"].exp())
    assert np.abs(got - r_logits.numpy()).max() / scale < COS_TOL
    labels = syn.synthetic_labels(torch.from_numpy(r_logits.numpy().argmax(1)), C, seed=2)
    ece_ref, _, _ = orc.calibrated_ece(r_logits.numpy(), labels.numpy(), None)
    from clip_calibration_amd.metrics import ECE
    assert abs(ECE(conf[rows.cuda()].cpu().numpy(), pred[rows.cuda()].cpu().numpy(), labels.numpy()) - ece_ref) < 1e-3


def test_config3_coop_dac_full_size(full):
    """BASELINE configs[2]: CoOp (16 context tokens) base->new with DAC, 256 images x 500 classes.  Properties: cached text
    features are reused; the fused DAC equals the unfused logits followed by the stand-alone row pass (bitwise); a positive
    per-class factor never changes the prediction; TempScaling's scalar commutes with the matmul."""
    from clip_calibration_amd import ops
    from clip_calibration_amd.dac import DistanseAwareCalibration, scale_logits_
    from clip_calibration_amd.trainers import CoOpCLIP, CustomCLIPCalibration, ZeroshotCLIP
    model, images = full["model"], full["images"]
    Cn, n_ctx = 500, 16
    ids_new = syn.synthetic_token_ids(Cn, G, seed=21, n_ctx_placeholders=n_ctx)
    ids_base = syn.synthetic_token_ids(Cn, G, seed=20, n_ctx_placeholders=n_ctx)
    coop_new = CoOpCLIP(model, ids_new, n_ctx=n_ctx, logit_scale=1.0, seed=3)
    coop_base = CoOpCLIP(model, ids_base, n_ctx=n_ctx, logit_scale=1.0, seed=3)
    zs_new = ZeroshotCLIP(model, syn.synthetic_token_ids(Cn, G, seed=21))
    zs_base = ZeroshotCLIP(model, syn.synthetic_token_ids(Cn, G, seed=20))
    t_new = coop_new.text_features()
    assert coop_new.text_features() is t_new
    cal = DistanseAwareCalibration()
    cal.fit(zs_base.text_features.cpu().numpy(), zs_new.text_features.cpu().numpy(),
            coop_base.text_features().cpu().numpy(), t_new.cpu().numpy(), 5)
    cc = cal.class_confidence
    assert cc.shape == (Cn,) and np.isfinite(cc).all() and (cc > 0).all()
    dacc = cal.class_confidence_device("cuda")
    calib = CustomCLIPCalibration(coop_new).cuda()
    with torch.no_grad():
        fused = calib(images, dac_conf=dacc, want_conf_pred=True)
        plain = calib(images, want_conf_pred=True)
    assert torch.equal(fused[4], plain[4])                                        # positive factors keep the arg-max
    manual = plain[0].clone()
    scale_logits_(manual, dacc)
    assert torch.equal(manual, fused[0])                                          # fused DAC == logits then the row pass
    # numpy-contract predict on a slice agrees with the fused rows
    sl = cal.predict(plain[0][:32].cpu().numpy().astype(np.float64))
    assert np.array_equal(sl, fused[0][:32].cpu().numpy())
    # TempScaling scalar: cosine base logits x exp(4.6052) == logits at that scale
    cos = ops.logits_fused(plain[1], plain[2], 1.0, None, False)[0]
    assert (cos * float(np.exp(4.6052)) - plain[0]).abs().max() < 1e-3
    assert cos.abs().max() <= 1.0 + 1e-5


def test_config5_vit_l14_336_full_batch():
    """BASELINE configs[4] per-GPU shape: ViT-L/14@336px, batch 64 (577 tokens, 24 layers, width 1024).  Determinism, batch
    invariance, and four images of the batch (first, last, two interior -- one oracle pass of four images: ~40 s of CPU) against the
    CPU oracle."""
    from clip_calibration_amd.model import build_model
    gname = "ViT-L/14@336px"
    sd = syn.synthetic_state_dict(gname, seed=0)
    model = build_model(dict(sd), None).cuda()
    images = syn.synthetic_images(64, gname, seed=4, device="cuda")
    with torch.no_grad():
        a = model.image_features_f32(images)
        b = model.image_features_f32(images)
        parts = torch.cat([model.image_features_f32(images[i:i + 8]) for i in range(0, 64, 8)])
        picks = [0, 21, 37, 63]
        ref = orc.encode_image(sd, images[picks].cpu()).numpy()
        _lib.set_option("gemm_split_rows", 0)      # M = 36 928 = 144 x 256 + 64: c_fc's ragged last tile row is a launch of its own by default (gemm.hip launch_one)
        try:
            single = model.image_features_f32(images)
        finally:
            _lib.set_option("gemm_split_rows", 1)
    assert torch.equal(a, b) and torch.isfinite(a).all()
    an, pn = torch.nn.functional.normalize(a, dim=1), torch.nn.functional.normalize(parts, dim=1)
    assert (an - pn).abs().max() < 2e-4
    assert (an - torch.nn.functional.normalize(single, dim=1)).abs().max() < 2e-4      # 64 of 36 928 rows per layer through another kernel's QuickGELU contraction
    rn = ref / np.linalg.norm(ref, axis=1, keepdims=True)
    got = an.cpu().numpy()
    cos = got @ rn.T                                                                # [64, 4]
    for j, i in enumerate(picks):
        assert abs(float(cos[i, j]) - 1.0) < COS_TOL, (i, float(cos[i, j]))
        assert np.abs(np.delete(cos[:, j], i)).max() < 1.0 - 1e-3                   # and the other rows are other images


def test_config4_dataset_sweep_class_counts():
    """BASELINE configs[3]: the 11-dataset base/new sweep at the per-GPU batch of 128 changes only C (SURVEY §8(d):
    C_base / C_new of the 11 datasets).  Fused logits + DAC + softmax top-1 + ECE bins for every one of those class counts
    against numpy on the same normalised features."""
    from clip_calibration_amd import ops
    from clip_calibration_amd.metrics import bin_statistics
    counts = sorted({50, 19, 98, 51, 199, 24, 5, 500, 18, 198, 23})
    rng = np.random.default_rng(4)
    img = rng.normal(size=(128, 512)).astype(np.float32)
    img /= np.linalg.norm(img, axis=1, keepdims=True)
    imgd = torch.from_numpy(img).cuda()
    for Cn in counts:
        txt = rng.normal(size=(Cn, 512)).astype(np.float32)
        txt /= np.linalg.norm(txt, axis=1, keepdims=True)
        dac = rng.uniform(0.7, 1.3, Cn).astype(np.float32)
        logits, conf, pred = ops.logits_fused(imgd, torch.from_numpy(txt).cuda(), 100.0, torch.from_numpy(dac).cuda(), True)
        want = orc.dac_predict((100.0 * img.astype(np.float64)) @ txt.astype(np.float64).T, dac.astype(np.float64))
        assert np.abs(logits.cpu().numpy() - want).max() < 2e-3, Cn
        rc, rp = orc.conf_pred(orc.softmax_probs(want))
        raw = (img.astype(np.float64)) @ txt.astype(np.float64).T
        top2 = np.sort(raw, axis=1)[:, -2:]
        clear = (top2[:, 1] - top2[:, 0]) > 1e-5
        assert np.array_equal(pred.cpu().numpy()[clear], rp[clear]), Cn
        assert np.abs(conf.cpu().numpy()[clear] - rc[clear]).max() < 1e-3, Cn
        labels = torch.from_numpy(rng.integers(0, Cn, 128)).cuda()
        bins = torch.zeros(33, dtype=torch.float64, device="cuda")
        ops.ece_accumulate(conf, pred, labels, bins, 10)
        ref_bins = bin_statistics(conf.cpu().numpy(), pred.cpu().numpy(), labels.cpu().numpy(), 10)
        assert np.array_equal(bins.cpu().numpy().reshape(3, 11)[[0, 2]], ref_bins[[0, 2]]), Cn


@pytest.mark.parametrize("Cn", [199, 24])
def test_config4_per_rank_path_towers_and_tail(Cn):
    """BASELINE configs[3] as a PATH at its per-rank shape: 128 images per GPU through the ViT-B/16 image tower, a class list
    of the sweep's size (SUN397 base half: 199 prompts; DTD base half: 24) through the text tower, then what each rank of the
    8-GPU run executes after the all-gather -- fp16 embeddings, fused tail with DAC, softmax top-1, ECE bins.  Checked against
    the oracle on a 12-image slice (full towers on the CPU) and by batch invariance for the other rows."""
    from clip_calibration_amd import ops
    from clip_calibration_amd.model import build_model
    from clip_calibration_amd.trainers import ZeroshotCLIP
    from clip_calibration_amd.metrics import bin_statistics
    sd = syn.synthetic_state_dict("ViT-B/16", seed=0)
    model = build_model(dict(sd), None).cuda()
    images = syn.synthetic_images(128, "ViT-B/16", seed=9, device="cuda")
    ids = syn.synthetic_token_ids(Cn, "ViT-B/16", seed=9)
    zs = ZeroshotCLIP(model, ids)
    rng = np.random.default_rng(Cn)
    dac = torch.from_numpy(rng.uniform(0.8, 1.2, Cn).astype(np.float32)).cuda()
    labels = torch.from_numpy(rng.integers(0, Cn, 128)).cuda()
    bins = torch.zeros(33, dtype=torch.float64, device="cuda")
    with torch.no_grad():
        feats = model.image_features_f32(images)
        emb = ops.l2_normalize(feats, torch.float16)                       # what crosses xGMI
        logits, _, conf, pred = ops.fused_tail(emb, zs.text_features, zs.scale, dac, True, False, labels, bins, 10)
        sl = slice(40, 52)
        feats_sl = model.image_features_f32(images[sl])
        ref_i = orc.l2_normalize(orc.encode_image(sd, images[sl].cpu()))
        ref_t = orc.l2_normalize(orc.encode_text(sd, ids))
    fn_, fs_ = torch.nn.functional.normalize(feats[sl], dim=1), torch.nn.functional.normalize(feats_sl, dim=1)
    assert (fn_ - fs_).abs().max() < 2e-4                                  # batch invariance (tile selection depends on the batch)
    raw = (zs.scale * ref_i @ ref_t.t()).numpy()
    ref = orc.dac_predict(raw, dac.cpu().numpy())
    # the DAC factor of a row hangs on its argmax: where the oracle's top two are closer than the tolerance the device may
    # legitimately pick the other class (and scale the whole row by that class's factor) -- check such rows with the factor
    # of the class the device picked, the clear ones against the oracle outright
    top2 = np.sort(raw, axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 2 * zs.scale * COS_TOL
    got, got_pred = logits[sl].cpu().numpy(), pred[sl].cpu().numpy()
    own = raw * dac.cpu().numpy()[got_pred][:, None]
    assert np.abs(got[clear] - ref[clear]).max() < zs.scale * COS_TOL * 1.3        # DAC factors up to 1.2, fp16 embeddings
    assert np.abs(got - own).max() < zs.scale * COS_TOL * 1.3
    rc, rp = orc.conf_pred(orc.softmax_probs(ref.astype(np.float64)))
    assert np.array_equal(got_pred[clear], rp[clear]) and clear.sum() >= 1
    ref_bins = bin_statistics(conf.cpu().numpy(), pred.cpu().numpy(), labels.cpu().numpy(), 10)
    assert np.array_equal(bins.cpu().numpy().reshape(3, 11)[[0, 2]], ref_bins[[0, 2]])


def test_vit_l14_ragged_tile_rows_launched_apart():
    """ViT-L/14 at 128 images (M = 32 896 = 128 x 256 + 128): BOTH persistent GEMMs of a block have a ragged last tile row that opens a round of its own
    (in-proj 129 x 12 tiles = 6.05 rounds -- with the LayerNorm fold's row partials read at a row offset --, c_fc 129 x 16 = 8.06) and run it as a second
    launch (gemm_split_rows, gemm.hip launch_one).  Against the single launch: the in-projection's bias epilogue gives the same bits, c_fc's QuickGELU the
    usual 1-ulp contraction difference on 128 of 32 896 rows per layer -- the features stay within the fp16 stream's own noise."""
    from clip_calibration_amd.model import build_model
    gname = "ViT-L/14"
    model = build_model(dict(syn.synthetic_state_dict(gname, seed=0)), None).cuda()
    images = syn.synthetic_images(128, gname, seed=5, device="cuda")
    with torch.no_grad():
        apart = model.image_features_f32(images)
        again = model.image_features_f32(images)
        _lib.set_option("gemm_split_rows", 0)
        try:
            single = model.image_features_f32(images)
        finally:
            _lib.set_option("gemm_split_rows", 1)
    assert torch.equal(apart, again) and torch.isfinite(apart).all()
    d = (torch.nn.functional.normalize(apart, dim=1) - torch.nn.functional.normalize(single, dim=1)).abs().max()
    print(f"ViT-L/14 x 128: max |d feature| between one launch and remainder apart {float(d):.2e}")
    assert d < 2e-4

