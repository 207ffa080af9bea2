"""CPU-only checks: the C-ABI library loads, exports every symbol include/clipmi.h declares, fails loudly without a
GPU; host-side metric logic; the boundary's state_dict surface."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from clip_calibration_amd import _lib, metrics, synthetic as syn
from clip_calibration_amd.model import build_model
from conftest import load_golden
from oracle import clip_oracle as orc


def _header_functions():
    src = open(_lib.HEADER_PATH).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(clipmi_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    declared = _header_functions()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(_lib.lib, name), f"libclipmi.so does not export {name}"
    assert sorted(_lib.exported_symbols()) == declared, "python binding and header disagree"
    assert _lib.lib.clipmi_abi_version() == _lib.ABI_VERSION


def test_no_cpu_fallback():
    """Without a GPU every entry point must fail loudly (error code + message), never compute."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    rc = _lib.lib.clipmi_l2_normalize(ctypes.c_void_p(256), 1, ctypes.c_void_p(512), 4, 64, None)
    assert rc == _lib.ERR_HIP and _lib.last_error()
    sd = syn.synthetic_state_dict("tiny")
    model = build_model(dict(sd), {"trainer": "CoOp"})
    with pytest.raises(RuntimeError, match="no CPU"):
        model.encode_image(torch.zeros(1, 3, 64, 64))
    from clip_calibration_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.l2_normalize(torch.zeros(2, 16))


def test_argument_validation_needs_no_gpu():
    L = _lib.lib
    assert L.clipmi_gemm_f16(None, 64, None, 64, None, None, None, 64, 0, 4, 4, 64, 0, None) == _lib.ERR_ARG
    p = ctypes.c_void_p(4096)
    assert L.clipmi_gemm_f16(p, 48, p, 48, None, None, p, 8, 0, 8, 8, 48, 0, None) == _lib.ERR_SHAPE   # K % 64
    assert "K=48" in _lib.last_error()
    assert L.clipmi_gemm_f16(p, 64, p, 64, None, None, p, 8, 0, 8, 8, 64, 9, None) == _lib.ERR_ARG      # bad epilogue
    assert L.clipmi_layernorm(p, 1, 6, None, p, p, p, 1, 6, 1, 6, 1e-5, None) == _lib.ERR_SHAPE       # D % 4
    assert L.clipmi_logits(p, p, 1.0, None, p, None, None, 4, 4, 10, None) == _lib.ERR_SHAPE            # E % 16
    geo = _lib.Geometry(512, 224, 16, 770, 12, 77, 49408, 512, 12, 8)
    h = ctypes.c_void_p()
    assert L.clipmi_create(ctypes.byref(geo), ctypes.byref(h)) == _lib.ERR_SHAPE                          # width % 64
    geo = _lib.Geometry(512, 224, 16, 768, 12, 77, 49408, 512, 12, 8)
    assert L.clipmi_create(ctypes.byref(geo), ctypes.byref(h)) == _lib.OK
    assert L.clipmi_encode_image(h, p, 1, 1, None, p, p, 1 << 30, 0, None) == _lib.ERR_STATE             # unbound weights
    # per-model settings live on the handle (clipmi_model_set_option); -1 follows the process-wide default
    v = ctypes.c_int(-7)
    assert L.clipmi_model_get_option(h, b"residual_f16", ctypes.byref(v)) == _lib.OK and v.value == _lib.get_option("residual_f16")
    assert L.clipmi_model_set_option(h, b"residual_f16", 0) == _lib.OK
    assert L.clipmi_model_get_option(h, b"residual_f16", ctypes.byref(v)) == _lib.OK and v.value == 0
    assert _lib.get_option("residual_f16") == 2                                                          # the process-wide default is untouched
    assert L.clipmi_model_set_option(h, b"residual_f16", 7) == _lib.ERR_ARG and L.clipmi_model_set_option(h, b"gemm_band", 1) == _lib.ERR_ARG
    assert L.clipmi_model_set_option(h, b"residual_f16", -1) == _lib.OK
    assert L.clipmi_model_get_option(h, b"residual_f16", ctypes.byref(v)) == _lib.OK and v.value == 2
    assert L.clipmi_vision_workspace_bytes(h, 256, 0) == pytest.approx((22 * 768 + 64) * 256 * 197, rel=1e-3)   # 22*D B of activations + 64 B of LN-fold partials per token row
    # a batch of one and a half passes or more is worked in passes (option vision_pass, default 50432 * 768 stream elements = 256 images
    # here): the workspace is that of the largest pass (a pass plus a remainder of under a quarter pass), not of the whole batch
    ws = lambda b: L.clipmi_vision_workspace_bytes(h, b, 0)
    assert ws(320) > ws(256) and ws(383) > ws(320)           # less than one and a half passes: still one pass
    assert ws(384) == ws(256) and ws(1024) == ws(256)        # 256 + 128, 4 x 256
    assert ws(2048 + 40) == ws(296)                          # the last pass takes the short remainder: 256 + 40
    _lib.set_option("vision_pass", 0)
    try:
        assert ws(1024) == pytest.approx(4 * ws(256), rel=1e-3)
    finally:
        _lib.set_option("vision_pass", 50432 * 768)
    assert L.clipmi_destroy(h) == _lib.OK


def test_boundary_state_dict_surface():
    sd = syn.synthetic_state_dict("tiny")
    model = build_model(dict(sd), {"trainer": "CoOp", "vision_depth": 0, "language_depth": 0, "vision_ctx": 0, "language_ctx": 0})
    assert not model.training
    got = model.state_dict()
    assert list(got) and set(got) == set(sd)
    # dtype policy of convert_weights (clip/model.py:632-653)
    assert got["visual.conv1.weight"].dtype == torch.float16
    assert got["visual.transformer.resblocks.0.attn.in_proj_weight"].dtype == torch.float16
    assert got["transformer.resblocks.1.mlp.c_proj.bias"].dtype == torch.float16
    assert got["visual.proj"].dtype == got["text_projection"].dtype == torch.float16
    for k in ("visual.ln_pre.weight", "visual.class_embedding", "positional_embedding", "token_embedding.weight",
              "ln_final.bias", "logit_scale", "visual.transformer.resblocks.0.ln_1.weight"):
        assert got[k].dtype == torch.float32, k
    for k, v in sd.items():
        assert torch.equal(got[k].float(), v.float()), k   # synthetic weights are fp16-exact
    assert model.dtype == torch.float16
    assert model.float().dtype == torch.float32
    assert model.visual.input_resolution == 64 and model.visual.output_dim == 128
    assert model.ln_final.weight.shape[0] == 128 and model.positional_embedding.shape == (77, 128)
    assert float(model.logit_scale.detach().exp()) == pytest.approx(100.0, rel=1e-3)
    # frozen-parameter protocol used by the trainers (coop.py:252-258)
    for name, p in model.named_parameters():
        p.requires_grad_(False)
    missing_ok = dict(sd)
    del missing_ok["visual.proj"]
    del missing_ok["logit_scale"]
    m2 = build_model(missing_ok | {"visual.proj": sd["visual.proj"]}, None)   # non-strict fallback prints and continues
    assert float(m2.logit_scale) == pytest.approx(np.log(1 / 0.07))
    with pytest.raises(ValueError):
        build_model({k: v for k, v in sd.items() if k != "visual.proj"}, None)   # neither a ViT nor a ModifiedResNet checkpoint


def test_ctx_init_n_ctx_from_padded_and_unpadded_ids():
    """CTX_INIT (coop.py:82-90, kgcoop.py:102-112): n_ctx is the word count of the init prompt, whether the caller passes the
    zero-padded [1, 77] tensor clip.tokenize returns or an unpadded [1, n+2] slice; anything else is refused."""
    from clip_calibration_amd.trainers.coop import n_ctx_from_init_ids
    padded = torch.zeros(1, 77, dtype=torch.long)
    padded[0, :6] = torch.tensor([49406, 320, 1125, 539, 320, 49407])      # "a photo of a"
    assert n_ctx_from_init_ids(padded, 77) == 4
    assert n_ctx_from_init_ids(padded[:, :6], 77) == 4
    sixteen = torch.zeros(1, 77, dtype=torch.long)
    sixteen[0, :18] = torch.tensor([49406] + [343] * 16 + [49407])
    assert n_ctx_from_init_ids(sixteen, 77) == 16
    with pytest.raises(ValueError):
        n_ctx_from_init_ids(torch.tensor([[49406, 320, 1125, 539, 320, 999]]), 77)   # no EOT: argmax lands on SOT
    with pytest.raises(ValueError):
        n_ctx_from_init_ids(torch.tensor([[49406, 49407]]), 77)                       # empty init


def test_host_ece_matches_reference_goldens():
    g = load_golden("ece_cases.npz")
    for n in sorted({k.split(":")[0] for k in g}):
        conf, pred, gt, bins = g[f"{n}:conf"], g[f"{n}:pred"], g[f"{n}:gt"], int(g[f"{n}:bins"])
        # float32 confidences: the reference's np.mean accumulates in float32, this module in float64
        tol = 1e-12 if conf.dtype == np.float64 else 2e-7
        assert metrics.ECE(conf, pred, gt, bins) == pytest.approx(float(g[f"{n}:ece"]), abs=tol), n
        assert metrics.MCE(conf, pred, gt, bins) == pytest.approx(float(g[f"{n}:mce"]), abs=tol), n
        assert metrics.AdaptiveECE(conf, pred, gt, bins) == pytest.approx(float(g[f"{n}:ace"]), abs=tol), n
        assert metrics.PIECE(conf, g[f"{n}:prox"], pred, gt, 10, bins) == pytest.approx(float(g[f"{n}:piece"]), abs=tol), n
        assert metrics.macro_f1(pred, gt) == pytest.approx(float(g[f"{n}:f1"]), abs=1e-12), n
        which = metrics.digitize_bins(conf, bins)
        assert np.array_equal(which, np.digitize(conf, np.linspace(0, 1, bins + 1)) - 1)


def test_host_ece_property():
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=60, deadline=None)
    @given(st.lists(st.floats(0.0, 1.0, width=32), min_size=1, max_size=200), st.integers(1, 20), st.integers(0, 2 ** 31))
    def prop(confs, bins, seed):
        rng = np.random.default_rng(seed)
        conf = np.asarray(confs, dtype=np.float32)
        pred = rng.integers(0, 5, conf.size)
        gt = rng.integers(0, 5, conf.size)
        assert metrics.ECE(conf, pred, gt, bins) == pytest.approx(orc.ece(conf, pred, gt, bins), abs=1e-6)

    prop()


def test_dac_fit_host():
    from clip_calibration_amd.dac import DistanseAwareCalibration
    g = load_golden("dac_cases.npz")
    for n in ("c50", "c19", "k3"):
        cal = DistanseAwareCalibration()
        cal.fit(g[f"{n}:base_zs"], g[f"{n}:cur_zs"], g[f"{n}:base_tuned"], g[f"{n}:cur_tuned"], int(g[f"{n}:k"]))
        np.testing.assert_allclose(cal.class_confidence, g[f"{n}:class_confidence"], rtol=1e-13)


def test_tokenizer_matches_reference_fixture():
    """SURVEY f-3: own BPE implementation vs ids produced by the reference SimpleTokenizer (sparse merge table)."""
    import json
    import os
    from conftest import GOLDEN
    from clip_calibration_amd.tokenizer import ClipTokenizer, coop_prompts, zeroshot_prompts
    fx = json.load(open(os.path.join(GOLDEN, "tokenizer_cases.json")))
    tok = ClipTokenizer(merges={(a, b): r for a, b, r in fx["merges"]})
    assert (tok.sot, tok.eot) == (fx["sot"], fx["eot"]) == (49406, 49407)
    for p, want in zip(fx["prompts"], fx["ids"]):
        assert [tok.sot] + tok.encode(p) + [tok.eot] == want, p
    packed = tok.tokenize(fx["prompts"][:20])
    assert packed.shape == (20, 77) and packed.dtype == torch.int64
    for row, want in zip(packed, fx["ids"][:20]):
        assert row[: len(want)].tolist() == want and int(row[len(want):].sum()) == 0
        assert int(row.argmax()) == len(want) - 1                     # EOT = max id (clip/model.py:611)
    for i, text in enumerate(fx["decoded"]):
        assert tok.decode(fx["ids"][i][1:-1]) == text
    assert tok.encode("a photo of a accordion.") == [320, 1125, 539, 320, 48760, 269]   # SURVEY §8(c) known answer
    with pytest.raises(RuntimeError):
        tok.tokenize("accordion " * 80)
    assert tok.tokenize("accordion " * 80, truncate=True)[0, -1] == tok.eot
    assert zeroshot_prompts(["car_side"], "Caltech101") == ["a photo of a car side."]
    assert coop_prompts(["car_side"], 4) == ["X X X X car side."]
    with pytest.raises(FileNotFoundError):
        ClipTokenizer(bpe_path="/nonexistent/bpe.txt.gz")


def test_checkpoint_and_base_feature_formats(tmp_path):
    """f-3: Dassl prompt-learner checkpoint layout (coop.py:311-343) and the base_features.pt dict (base_learner.py:184-239)."""
    import torch.nn as nn
    from clip_calibration_amd import checkpoint as ck

    class PL(nn.Module):
        def __init__(self):
            super().__init__()
            self.ctx = nn.Parameter(torch.zeros(4, 8))
            self.register_buffer("token_prefix", torch.zeros(3, 1, 8))
            self.register_buffer("token_suffix", torch.zeros(3, 5, 8))
    src = PL()
    with torch.no_grad():
        src.ctx.copy_(torch.randn(4, 8))
        src.token_prefix.fill_(7.0)                       # a DIFFERENT class list's fixed tokens: must not be loaded
    path = ck.save_checkpoint(src.state_dict(), str(tmp_path), "prompt_learner", 50, val_result=12.5)
    assert path.endswith("prompt_learner/model.pth.tar-50") and (tmp_path / "prompt_learner" / "checkpoint").read_text().strip() == "model.pth.tar-50"
    dst = PL()
    assert ck.load_model(dst, str(tmp_path), "prompt_learner", epoch=50) == 50
    assert torch.equal(dst.ctx, src.ctx) and float(dst.token_prefix.abs().sum()) == 0.0
    with pytest.raises(FileNotFoundError):
        ck.load_model(dst, str(tmp_path), "prompt_learner")          # no model-best.pth.tar written
    p = ck.base_features_path(str(tmp_path / "temp"), "Caltech101", "CoOp", 16, "ViT-B/16", 1)
    assert p.endswith("Caltech101/CoOp/shots16/ViT-B/16/base/seed1/base_features.pt")
    rng = np.random.default_rng(0)
    d = dict(val_logits=rng.normal(size=(6, 3)).astype(np.float32), val_image_features=rng.normal(size=(6, 8)).astype(np.float32),
             val_text_features=rng.normal(size=(3, 8)).astype(np.float32), val_labels=rng.integers(0, 3, 6),
             val_image_knn_dists=rng.uniform(size=(6, 2)).astype(np.float32))
    ck.save_base_features(p, **d)
    back = ck.load_base_features(p)
    assert set(back) == set(ck.BASE_FEATURE_KEYS) and all(np.array_equal(back[k], d[k]) for k in d)
    torch.save({"val_logits": d["val_logits"]}, p)
    with pytest.raises(KeyError):
        ck.load_base_features(p)


def test_quantile_bins_match_sklearn():
    """metrics.quantile_bin_index restates sklearn's KBinsDiscretizer(quantile, ordinal) -- checked live where sklearn exists."""
    sk = pytest.importorskip("sklearn.preprocessing")
    import warnings
    rng = np.random.default_rng(3)
    cases = [rng.uniform(size=500), rng.uniform(size=37).astype(np.float32), np.round(rng.uniform(size=400), 1),   # heavy ties
             np.concatenate([np.ones(90), rng.uniform(size=10)]), np.full(12, 0.5), np.array([0.2, 0.9])]
    for x in cases:
        for nb in (3, 10, 15):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                want = sk.KBinsDiscretizer(n_bins=nb, encode="ordinal", strategy="quantile").fit_transform(x[:, None])[:, 0]
            assert np.array_equal(metrics.quantile_bin_index(x, nb), want.astype(np.int64))
            assert np.array_equal(orc.quantile_bins(x, nb), want.astype(np.int64))


def test_modified_resnet_boundary_surface():
    """f-4: a checkpoint without ``visual.proj`` builds the ModifiedResNet tower (clip/model.py:659-672) under the reference's key
    names and dtype policy; nothing runs without a GPU."""
    sd = syn.synthetic_resnet_state_dict((1, 2, 1, 1), 64, 64, "tiny", seed=0)
    g = load_golden("resnet_tiny.npz")
    m = build_model(dict(sd), None)
    keys = set(m.state_dict())
    assert keys == set(sd) and len(keys) == int(g["n_keys"])
    assert m.is_resnet and m.visual.input_resolution == 64 and m.visual.output_dim == 128 and m.visual.attnpool.num_heads == 32
    sdm = m.state_dict()
    assert sdm["visual.layer2.0.downsample.0.weight"].dtype == torch.float16 and sdm["visual.attnpool.q_proj.bias"].dtype == torch.float16
    assert sdm["visual.bn1.running_var"].dtype == torch.float32 and sdm["visual.attnpool.positional_embedding"].dtype == torch.float32
    assert m.dtype == torch.float16
    with pytest.raises(RuntimeError):
        m.encode_image(torch.zeros(1, 3, 64, 64))                       # no CPU path


MFMA_SOURCES = ("attention.hip", "logits.hip", "gemm.hip", "gemm_rstream.hip", "probe.hip")


def _hazard_scan():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("mfma_hazard_scan", os.path.join(root, "tools", "mfma_hazard_scan.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_no_valu_written_mfma_source_closer_than_four_wait_states():
    """Static guard for the hazard of profiles/r03_gpu_sharing.txt: on gfx950 a VGPR that a VALU instruction has written and a v_mfma reads as a
    source two wait states later (all hipcc guarantees) leaves the wave's own result right and clobbers a register quarter of a wave of another
    kernel resident on the same SIMD.  tools/mfma_hazard_scan.py runs a data-flow pass over the control-flow graph of every kernel of every
    translation unit with MFMAs (fall-through and branch edges, loops included): with the library's fences (CLIPMI_VALU_TO_MFMA_FENCE = four
    wait states by itself) no such pair is closer than four wait states.  The scan must have SEEN something: a file that does not compile or
    holds no MFMA kernel is an error, not a pass."""
    import shutil
    import subprocess
    import sys
    if shutil.which(os.environ.get("HIPCC", "hipcc")) is None:
        pytest.skip("no hipcc on this box: the ISA cannot be produced")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "clip_calibration_amd", "csrc")
    with_mfma = sorted(f for f in os.listdir(csrc) if f.endswith(".hip") and "mfma" in open(os.path.join(csrc, f)).read())
    assert with_mfma == sorted(MFMA_SOURCES), f"translation units with MFMAs changed: {with_mfma} -- extend MFMA_SOURCES"
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "mfma_hazard_scan.py"), *MFMA_SOURCES],
                       env=dict(os.environ, WAIT="4"), capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    lines = [ln for ln in r.stdout.splitlines() if "VALU write -> MFMA source sites" in ln]
    assert len(lines) == len(MFMA_SOURCES), r.stdout
    for ln in lines:
        m = re.search(r": 0 VALU write .* \((\d+) MFMA kernels, (\d+) MFMA instructions walked\)", ln)
        assert m and int(m.group(1)) >= 1 and int(m.group(2)) >= 32, ln
    # round 6: the GEMM translation units carry vector instructions BETWEEN their MFMAs now (the activation stream of gemm_stream_kernel): none of them
    # may read an MFMA result with a younger MFMA in flight (the second co-residency hazard) -- the scanner's second pass over every kernel of the file
    for src in ("gemm.hip", "gemm_rstream.hip"):
        assert re.search(re.escape(src) + r": 0 MFMA result -> vector / store read sites", r.stdout), r.stdout[-2000:]


def test_no_kernel_names_a_register_outside_its_allocation():
    """Round 6 (profiles/r06_hazard_root_cause.txt): damage to a NEIGHBOUR's registers is also what an out-of-allocation register write looks like.
    The scanner's third pass compares the highest VGPR / AGPR index any instruction of a kernel names with the kernel descriptor's allocation
    (.amdhsa_next_free_vgpr / .amdhsa_accum_offset), for every kernel of the two translation units with inline-asm MFMAs and tied tuples."""
    import shutil
    import subprocess
    import sys
    if shutil.which(os.environ.get("HIPCC", "hipcc")) is None:
        pytest.skip("no hipcc on this box: the ISA cannot be produced")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "mfma_hazard_scan.py"), "gemm_rstream.hip", "attention.hip"],
                       env=dict(os.environ, WAIT="4"), capture_output=True, text=True)
    lines = [ln for ln in r.stdout.splitlines() if "name a register outside their allocation" in ln]
    assert len(lines) == 2, r.stdout[-2000:] + r.stderr[-1000:]
    for ln in lines:
        m = re.search(r": 0 kernels name a register outside their allocation \((\d+) kernel descriptors checked", ln)
        assert m and int(m.group(1)) >= 1, ln
    hs = _hazard_scan()
    asm = ("_Z1kv:\n v_mfma_f32_16x16x32_f16 v[8:11], v[0:3], v[4:7], v[8:11]\n v_add_f32 v12, v8, v8\n s_endpgm\n"
           ".amdhsa_kernel _Z1kv\n .amdhsa_next_free_vgpr 12\n .amdhsa_accum_offset 12\n.end_amdhsa_kernel\n")
    (b,) = hs.register_bounds(asm, hs.split_kernels(asm))
    assert b[1] == 12 and not b[5]                      # v12 named, 12 registers allocated: caught
    (b,) = hs.register_bounds(asm.replace("next_free_vgpr 12", "next_free_vgpr 13").replace("accum_offset 12", "accum_offset 16"), hs.split_kernels(asm))
    assert b[5]


def test_no_close_vector_read_of_an_mfma_result_in_the_round5_kernels():
    """The second co-residency hazard (common.h CLIPMI_MFMA_TO_VALU_FENCE3; profiles/r05_vitl_attention.txt): a register an MFMA has written, read by the
    vector pipe behind hipcc's own `s_nop 10` with another MFMA issued in between -- `l += lacc[0]` right behind every softmax group made the first
    ring attention kernel corrupt 52-160 of 200 LayerNorm launches of a co-resident kernel.  The scanner's second pass (an MFMA in between counts 8
    wait states, everything else 1, `s_nop N` N + 1; sites below 32) must find none in the kernels written this round.  The older attention kernels
    keep such reads in their softmax-maximum and epilogue code (reported, not failed: three rounds of the hammer test beside them are clean)."""
    import shutil
    import subprocess
    import sys
    if shutil.which(os.environ.get("HIPCC", "hipcc")) is None:
        pytest.skip("no hipcc on this box: the ISA cannot be produced")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "mfma_hazard_scan.py"), "attention.hip"],
                       env=dict(os.environ, WAIT="4", KERNELS="attention_ring_kernel,attention_small_kernel", STRICT_READS="1"), capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert re.search(r": 0 MFMA result -> vector / store read sites", r.stdout), r.stdout
    hs = _hazard_scan()

    def reads(text):
        (name, insts, labels), = hs.split_kernels("_Z1kv:\n" + text)
        return hs.scan_kernel_reads(name, insts, labels)
    bad = ("v_mfma_f32_32x32x16_f16 v[64:79], v[0:3], v[4:7], v[64:79]\nv_mfma_f32_32x32x16_f16 v[16:31], v[8:11], v[4:7], v[16:31]\ns_nop 10\n"
           "v_add_f32 v100, v100, v64\nv_mfma_f32_32x32x16_f16 v[32:47], v[12:15], v[4:7], v[32:47]\ns_endpgm\n")
    got = reads(bad)
    assert len(got) == 1 and "v_add_f32" in got[0][2] and got[0][3] == 19          # 8 (the MFMA in between) + 11 (s_nop 10)
    latest = "v_mfma_f32_32x32x16_f16 v[64:79], v[0:3], v[4:7], v[64:79]\ns_nop 10\nv_max3_f32 v100, v64, v65, v100\ns_endpgm\n"
    assert reads(latest) == []                                                      # the latest MFMA's own result: hipcc's table holds
    far = ("v_mfma_f32_32x32x16_f16 v[64:79], v[0:3], v[4:7], v[64:79]\n" + "v_mfma_f32_32x32x16_f16 v[16:31], v[8:11], v[4:7], v[16:31]\n" * 4 +
           "v_add_f32 v100, v100, v64\ns_endpgm\n")
    assert reads(far) == []                                                         # four MFMAs in between: 32 wait states
    fenced = bad.replace("s_nop 10\n", "s_nop 10\ns_nop 15\ns_nop 15\n")
    assert reads(fenced) == []
    store = ("v_mfma_f32_32x32x16_f16 v[64:79], v[0:3], v[4:7], v[64:79]\nv_mfma_f32_32x32x16_f16 v[16:31], v[8:11], v[4:7], v[16:31]\n"
             "global_store_dwordx4 v[200:201], v[64:67], off\ns_endpgm\n")
    assert len(reads(store)) == 1                                                   # stores read registers too


def test_hazard_scan_follows_branches_and_fails_loudly():
    """The scanner itself, on hand-written ISA: (1) straight line -- `s_nop 1` between a conversion and the MFMA that reads it is two wait states:
    a site at WAIT = 4, none at WAIT = 2; (2) a VALU write at the BOTTOM of a loop feeding the MFMA at its head through the back-edge; (3) a
    forward branch that skips the padding; (4) AGPR sources; (5) a load into the register supersedes the write; (6) a compile failure and a
    file without MFMAs raise instead of reporting "0 sites"."""
    hs = _hazard_scan()

    def sites(text, wait):
        hs.WAIT = wait
        (name, insts, labels), = hs.split_kernels("_Z1kv:\n" + text)
        return hs.scan_kernel(name, insts, labels)[0]
    straight = "v_cvt_pk_f16_f32 v4, v0, v1\ns_nop 1\nv_mfma_f32_16x16x32_f16 v[8:11], v[4:7], v[12:15], v[8:11]\ns_endpgm\n"
    assert len(sites(straight, 4)) == 1 and sites(straight, 4)[0][3] == 2 and sites(straight, 2) == []
    loop = (".LBB0_1:\nv_mfma_f32_16x16x32_f16 v[8:11], v[4:7], v[12:15], v[8:11]\ns_nop 7\ns_nop 7\nv_exp_f32 v5, v20\n"
            "s_cbranch_scc1 .LBB0_1\ns_endpgm\n")
    got = sites(loop, 4)
    assert len(got) == 1 and "v_exp_f32 v5" in got[0][1] and got[0][3] == 1          # only reachable through the back-edge
    fwd = ("v_accvgpr_write_b32 a3, v1\ns_cbranch_vccz .LBB0_2\ns_nop 7\n.LBB0_2:\nv_mfma_f32_32x32x16_f16 a[16:31], v[4:7], a[0:3], a[16:31]\ns_endpgm\n")
    got = sites(fwd, 4)
    assert len(got) == 1 and "a3" in got[0][1] and got[0][3] == 1                      # the taken edge has one wait state, the fall-through nine
    reload = "v_mov_b32 v4, v0\nds_read_b128 v[4:7], v30\nv_mfma_f32_16x16x32_f16 v[8:11], v[4:7], v[12:15], v[8:11]\ns_endpgm\n"
    assert sites(reload, 4) == []
    hs.WAIT = 4
    import shutil
    if shutil.which("hipcc") is not None:
        with pytest.raises(hs.ScanError):
            hs.scan("does_not_exist.hip")
        with pytest.raises(hs.ScanError, match="no kernel with a v_mfma"):
            hs.scan("layernorm.hip")




def test_live_rows_is_host_logic():
    """CLIP.live_rows (dead-row elimination of the causal text tower, include/clipmi.h `seq_rows`): max(EOT) + 1 rounded up to 8, per NEW
    prompt set, never below a hook's prompt tokens, never above the context."""
    model = build_model(dict(syn.synthetic_state_dict("tiny")), {"trainer": "CoOp"})
    ids = syn.synthetic_token_ids(8, "tiny", seed=11, n_ctx_placeholders=4)
    last = int(ids.argmax(-1).max())
    r = model.live_rows(ids)
    assert r == (last + 8) // 8 * 8 and r == model.live_rows(ids[:]) and len(model._live_rows) == 1
    ids[0, :40] = 5
    ids[0, 40] = model.vocab_size - 1                      # in-place edit: the version counter moves, the bound is recomputed
    assert model.live_rows(ids) == 48 and model.live_rows(ids, n_ctx=60) == 61
    ids[1, 76] = model.vocab_size - 1
    ids[1, :76] = 5
    assert model.live_rows(ids) == 77
    for k in range(40):
        model.live_rows(syn.synthetic_token_ids(4, "tiny", seed=100 + k))
    assert len(model._live_rows) <= 16
    # tensors made under inference_mode have no version counter (reading it raises): the bound is computed per call, nothing is cached
    with torch.inference_mode():
        inf_ids = syn.synthetic_token_ids(8, "tiny", seed=11, n_ctx_placeholders=4).clone()
        n_before = len(model._live_rows)
        assert model.live_rows(inf_ids) == r and model.live_rows(inf_ids) == r and len(model._live_rows) == n_before
    # an entry does not keep its tensor alive, and an entry whose tensor is gone is not trusted
    import gc, weakref
    tmp = syn.synthetic_token_ids(4, "tiny", seed=999)
    model.live_rows(tmp)
    wr = weakref.ref(tmp)
    del tmp
    gc.collect()
    assert wr() is None
    model.text_dead_row_elimination = False
    assert model.live_rows(ids) == model.context_length
