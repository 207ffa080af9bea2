"""Two CLIP handles in one process, driven from two Python threads on two HIP streams -- what the reference does when
`get_text_features` builds a second CLIP beside the trainer's (trainers/classification/base_learner.py:262-272), each with its own
precision (cfg.TRAINER.<X>.PREC, trainers/classification/coop.py:243-245).  include/clipmi.h promises: calls on different handles may
run concurrently; precision is a per-handle setting / per-call flag, no launch path writes process-wide state."""
import threading

import pytest
import torch

from clip_calibration_amd import _lib, synthetic as syn
from clip_calibration_amd.model import build_model

pytestmark = pytest.mark.gpu


def _model(seed, residual_f16):
    m = build_model(dict(syn.synthetic_state_dict("ViT-B/16", seed=seed)), {"trainer": "CoOp"}).cuda()
    m.set_option("residual_f16", residual_f16)
    return m


def test_two_handles_two_threads_two_streams_bitwise():
    """Handle A: text tower on the fp32 residual stream (residual_f16 = 0), handle B: fp16 on both towers (1), different weights.
    Each thread runs its model's text and image towers ROUNDS times on its own stream while the other does the same; every result
    must equal the single-threaded result of that handle bit for bit, and the process-wide options must be untouched."""
    ids = syn.synthetic_token_ids(96, "ViT-B/16", seed=3).cuda()
    images = syn.synthetic_images(24, "ViT-B/16", seed=3).cuda()
    models = {"a": _model(0, 0), "b": _model(1, 1)}
    before = {k: _lib.get_option(k) for k in ("residual_f16", "ln_fold", "cls_only_last_block")}
    with torch.no_grad():
        want = {k: (m.text_features_f32(ids).clone(), m.image_features_f32(images).clone()) for k, m in models.items()}
        assert not torch.equal(want["a"][0], want["b"][0])
        # the settings really differ per handle: handle A with B's setting gives other bits
        models["a"].set_option("residual_f16", 1)
        other = models["a"].text_features_f32(ids).clone()
        models["a"].set_option("residual_f16", 0)
        assert not torch.equal(other, want["a"][0])
    torch.cuda.synchronize()
    ROUNDS, errors, start = 6, [], threading.Barrier(2)

    def work(key):
        try:
            m, stream = models[key], torch.cuda.Stream()
            start.wait()
            with torch.no_grad(), torch.cuda.stream(stream):
                for r in range(ROUNDS):
                    t = m.text_features_f32(ids)
                    i = m.image_features_f32(images)
                    # per-call flag on top of the handle's setting: the other precision for this call only
                    flip = m.text_features_f32(ids, flags=_lib.CALL_STREAM_F16 if key == "a" else _lib.CALL_STREAM_F32)
                    stream.synchronize()
                    if not (torch.equal(t, want[key][0]) and torch.equal(i, want[key][1])):
                        errors.append(f"{key}: round {r} differs from the single-threaded result")
                    if torch.equal(flip, want[key][0]):
                        errors.append(f"{key}: round {r}: the per-call flag did not switch the stream precision")
        except Exception as e:   # noqa: BLE001  (reported below: a thread's exception would otherwise vanish)
            errors.append(f"{key}: {type(e).__name__}: {e}")

    threads = [threading.Thread(target=work, args=(k,)) for k in models]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    assert {k: _lib.get_option(k) for k in before} == before          # nothing process-wide was written
    assert models["a"].get_option("residual_f16") == 0 and models["b"].get_option("residual_f16") == 1


def test_call_flag_f16_needs_the_fold():
    m = _model(0, -1)
    ids = syn.synthetic_token_ids(4, "ViT-B/16", seed=1).cuda()
    m.set_option("ln_fold", 0)
    with pytest.raises(_lib.ClipmiError):
        m.text_features_f32(ids, flags=_lib.CALL_STREAM_F16)
    m.set_option("ln_fold", -1)
    with torch.no_grad():
        assert torch.isfinite(m.text_features_f32(ids, flags=_lib.CALL_STREAM_F16)).all()


HAMMERS = {"vision attention (197 tokens) + tail": (24, 197, 12, False), "text attention (77 tokens, causal)": (500, 77, 8, True),
           "ViT-L/14 attention (257 tokens)": (16, 257, 16, False), "ViT-L/14@336 attention (577 tokens)": (8, 577, 16, False),
           "truncated text attention (24 tokens, causal)": (500, 24, 8, True)}


@pytest.mark.parametrize("hammer_kernel", list(HAMMERS))
def test_small_kernel_beside_mfma_kernels_on_a_second_stream(hammer_kernel):
    """Regression guard for CLIPMI_VALU_TO_MFMA_FENCE (common.h; profiles/r03_gpu_sharing.txt): an MFMA that reads a source operand VALU
    instructions have just written (the softmax's P, a re-materialised constant, rescaled accumulators, the tail's hi / lo split) needs wait
    states hipcc does not insert on gfx950 -- its own result is right, but a wave of another kernel resident on the same SIMD loses a
    register quarter.  While an attention kernel (and the fused tail) loops on a second stream, the LayerNorm kernel (one wave per row, 56
    registers: it fits beside them) must keep returning the right rows.  Without the fences: 28-32 of 300 launches wrong beside the vision
    kernel, 276 of 300 beside the text tower's, 24-26 of 300 beside the 257-token kernel.  Round 5 found the same damage from the other direction --
    a register an MFMA has just written, read by the vector pipe behind hipcc's own `s_nop 10` (CLIPMI_MFMA_TO_VALU_FENCE3): 52-160 of 200 launches
    wrong beside the first form of the ring attention kernel (257 / 577 tokens), hence its two hammers here; the one-item-per-wave kernel of the
    truncated text tower (four waves of 97 registers per workgroup: a victim fits on every SIMD) is the fifth."""
    from clip_calibration_amd import ops
    g = torch.Generator().manual_seed(0)
    M, K = 197 * 256, 768
    x = torch.randn(M, K, generator=g).cuda()
    gam, bet = torch.ones(K).cuda(), torch.zeros(K).cuda()
    truth = torch.nn.functional.layer_norm(x.double(), (K,)).float()
    n, l, h, causal = HAMMERS[hammer_kernel]
    qkv = torch.randn(n * l, 3 * 64 * h, generator=g).half().cuda()
    feat = torch.randn(256, 512, generator=g).cuda()
    txt = ops.l2_normalize(torch.randn(1000, 512, generator=g).cuda())
    with_tail = hammer_kernel.endswith("tail")
    stop = threading.Event()
    errors = []

    def hammer():
        try:
            s = torch.cuda.Stream()
            k = 0
            with torch.cuda.stream(s):
                while not stop.is_set():
                    ops.attention(qkv, n, l, h, causal)
                    if with_tail and k % 8 == 0:
                        ops.fused_tail(feat, txt, 100.0, None, True, True)
                    k += 1
                    if k % 100 == 0:
                        s.synchronize()
                s.synchronize()
        except Exception as e:      # surface it in the main thread
            errors.append(e)

    t = threading.Thread(target=hammer)
    t.start()
    try:
        bad = 0
        for _ in range(300):
            out = ops.layernorm(x, gam, bet)
            bad += int(bool(((out - truth).abs().amax(dim=1) > 1e-3).any()))
    finally:
        stop.set()
        t.join()
    assert not errors, errors
    assert bad == 0, f"{bad} of 300 LayerNorm launches returned a wrong row beside the MFMA kernels of the second stream"
