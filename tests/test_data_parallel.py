"""nn.DataParallel around the unchanged trainers (reference trainers/classification/coop.py:268-272, kgcoop.py:318-322, maple.py:273-277,
trainers/calibration/tempscaling.py:117-120: taken whenever torch.cuda.device_count() > 1).

DataParallel clones the module tree per device on every forward (``Module._replicate_for_data_parallel``, a shallow ``__dict__``
copy) and runs the clones in threads.  The clones of the tower proxies must never drive cuda:0's C handle with activations that
live on cuda:k: the owner answers such a call from a copy of itself resident on that device (model.py ``_resident``).  The CPU
tests pin the bookkeeping (no shared / freed handle, routing by the activation's device, invalidation on re-bind); the GPU tests
run real replicas through ``torch.nn.parallel.replicate`` / ``parallel_apply`` and a resident copy against its owner bit for bit.
A box with one GPU cannot execute cuda:1; what cannot be run here is exactly one line: ``torch.cuda.device(k)`` around the call."""
import copy
import gc
import pickle

import pytest
import torch
import torch.nn as nn

from clip_calibration_amd import synthetic as syn
from clip_calibration_amd.model import CLIP, build_model


class RefTextEncoder(nn.Module):
    """The reference's TextEncoder, attribute for attribute (coop.py:47-67): what ``nn.DataParallel`` wraps there."""

    def __init__(self, clip_model):
        super().__init__()
        self.transformer = clip_model.transformer
        self.positional_embedding = clip_model.positional_embedding
        self.ln_final = clip_model.ln_final
        self.text_projection = clip_model.text_projection
        self.dtype = clip_model.dtype

    def forward(self, prompts, tokenized_prompts):
        x = prompts + self.positional_embedding.type(self.dtype)
        x = x.permute(1, 0, 2)
        x = self.transformer(x)
        x = x.permute(1, 0, 2)
        x = self.ln_final(x).type(self.dtype)
        return x[torch.arange(x.shape[0]), tokenized_prompts.argmax(dim=-1)] @ self.text_projection


def _model(design="CoOp"):
    return build_model(dict(syn.synthetic_state_dict("tiny", seed=0)), {"trainer": design})


def test_replicas_never_share_the_owners_handle():
    m = _model()
    m._handle = 0xdead0                                  # pretend a bound handle (never dereferenced: no launch on the CPU)
    m._bound = ("packed",)
    try:
        r = m._replicate_for_data_parallel()
        assert r._handle is None and r._bound is None and r._ws == {} and r._device_copies == {}
        assert r._home() is m and m._home() is m
        rr = r._replicate_for_data_parallel()            # a clone of a clone still knows the owner
        assert rr._home() is m and rr._handle is None
        # the tower proxies a trainer holds (clip_model.visual / clip_model.transformer) keep pointing at the owner
        enc = RefTextEncoder(m)
        clone = enc._replicate_for_data_parallel()
        t = enc.transformer._replicate_for_data_parallel()
        assert t._owner is m and clone.dtype == m.dtype
        assert m.visual._replicate_for_data_parallel()._owner is m
        del r, rr
        gc.collect()                                     # a collected clone must not have destroyed anything
        assert m._handle == 0xdead0
    finally:
        m._handle = None


def test_calls_are_routed_by_the_activations_device(monkeypatch):
    """``_resident(device)``: the owner for its own device, ONE cached copy per other device, dropped when the weights are re-bound."""
    m = _model()
    made = []

    def fake_copy(self, device):
        made.append(device)
        return object()

    monkeypatch.setattr(CLIP, "_copy_to", fake_copy)
    monkeypatch.setattr(CLIP, "device", property(lambda self: torch.device("cuda", 0)))
    d0, d1, d2 = (torch.device("cuda", i) for i in range(3))
    assert m._resident(d0) is m and not made
    monkeypatch.setattr(torch.cuda, "device", lambda dev: __import__("contextlib").nullcontext())
    a = m._resident(d1)
    assert m._resident(d1) is a and made == [d1]
    replica = m._replicate_for_data_parallel()
    assert replica._resident(d1) is a and replica._resident(d0) is m     # clones route through the owner's table
    b = m._resident(d2)
    assert b is not a and made == [d1, d2]
    m.rebind()
    assert m._resident(d1) is not a and made == [d1, d2, d1]
    c = m._resident(d1)
    with torch.no_grad():
        next(m.parameters()).add_(0.0)                                    # an in-place update of ANY owner parameter (IVLP prompts under training) ...
    assert m._resident(d1) is not c and made == [d1, d2, d1, d1]          # ... retires the snapshot on cuda:1
    assert m._resident(d1) is m._resident(d1)
    m.load_state_dict(m.state_dict())
    assert m._device_copies == {}
    assert m._resident(torch.device("cpu")) is m                          # refused further down, by the op that sees the CPU tensor


def test_cpu_model_is_not_silently_moved():
    m = _model()                                         # lives on the CPU: a CUDA activation must not conjure a GPU copy
    assert m._resident(torch.device("cuda", 0)) is m
    assert m._elsewhere(torch.zeros(1)) is None


def test_copies_and_pickles_own_nothing_of_the_original():
    m = _model()
    m._handle = 0xbeef0
    try:
        c = copy.deepcopy(m)
        assert c._handle is None and c._bound is None and c._device_copies == {}
        assert c.visual._owner is c and c.transformer._owner is c
        assert c._copies_lock is not m._copies_lock and c._copies_lock is not None
        p = pickle.loads(pickle.dumps(m))
        assert p._handle is None and p.visual._owner is p and p.transformer._owner is p
        for (n1, a), (n2, b) in zip(m.state_dict().items(), p.state_dict().items()):
            assert n1 == n2 and torch.equal(a, b)
    finally:
        m._handle = None


def test_resnet_replica_takes_packed_operands_from_the_owner():
    m = build_model(dict(syn.synthetic_resnet_state_dict(seed=0)), {"trainer": "CoOp"})
    m.visual._packed = {"stale": 1}
    r = m.visual._replicate_for_data_parallel()
    assert r._packed is None and r.__dict__["_origin"] is m.visual
    assert r._replicate_for_data_parallel().__dict__["_origin"] is m.visual
    assert copy.deepcopy(m.visual)._packed is None and copy.deepcopy(m)._handle is None
    m.load_state_dict(m.state_dict())
    assert m.visual._packed is None and m.visual._packed_elsewhere == {}


# ---------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_resident_copy_equals_owner_bitwise():
    m = _model().cuda()
    m.set_option("residual_f16", 1)
    twin = m._copy_to(torch.device("cuda", 0))
    assert twin.get_option("residual_f16") == 1 and twin._handle != m._handle
    images = syn.synthetic_images(6, "tiny", seed=3).cuda()
    ids = syn.synthetic_token_ids(10, "tiny", seed=3).cuda()
    a, b = m.image_features_f32(images), twin.image_features_f32(images)
    ta, tb = m.text_features_f32(ids), twin.text_features_f32(ids)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(ta, tb)
    for (n1, p), (n2, q) in zip(m.named_parameters(), twin.named_parameters()):
        assert n1 == n2 and p.data_ptr() != q.data_ptr() and p.dtype == q.dtype


@pytest.mark.gpu
def test_replicated_text_encoder_and_visual_match_the_module():
    """``replicate`` + ``parallel_apply`` are the two halves of DataParallel.forward; with device_ids = [0, 0] two clones run in two
    threads on the one GPU there is.  Same bits as the un-replicated modules, and the owner still works after the clones are gone."""
    from torch.nn.parallel import parallel_apply, replicate
    m = _model().cuda()
    enc = RefTextEncoder(m).cuda()
    g = m.geometry
    prompts = (0.02 * torch.randn(12, g.context_length, g.transformer_width, generator=torch.Generator().manual_seed(1))).to("cuda", m.dtype)
    ids = syn.synthetic_token_ids(12, "tiny", seed=4).cuda()
    images = syn.synthetic_images(8, "tiny", seed=4).cuda().to(m.dtype)
    ref_t, ref_i = enc(prompts, ids), m.visual(images)
    torch.cuda.synchronize()
    clones = replicate(enc, [0, 0])
    assert all(c.transformer._owner is m for c in clones)
    outs = parallel_apply(clones, [(prompts[:6], ids[:6]), (prompts[6:], ids[6:])], devices=[0, 0])
    vis = parallel_apply(replicate(m.visual, [0, 0]), [(images[:4],), (images[4:],)], devices=[0, 0])
    whole = parallel_apply(replicate(m, [0]), [(images, ids)], devices=[0])[0]
    torch.cuda.synchronize()
    assert torch.equal(torch.cat(outs), ref_t) and torch.equal(torch.cat(vis), ref_i)
    lpi, _ = m(images, ids)
    assert torch.equal(whole[0], lpi)
    del clones, outs, vis, whole
    gc.collect()
    again = enc(prompts, ids)
    torch.cuda.synchronize()
    assert torch.equal(again, ref_t)


@pytest.mark.gpu
def test_wrong_device_is_refused_not_computed():
    """Operator level: a tensor that is not on the current device is refused with the remedy in the message (ops._dev)."""
    from clip_calibration_amd import ops
    m = _model().cuda()
    with pytest.raises(RuntimeError, match="no CPU"):
        m.visual(torch.zeros(1, 3, m.geometry.image_resolution, m.geometry.image_resolution))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.layernorm(torch.zeros(2, 64), m.ln_final.weight.float(), m.ln_final.bias.float())


@pytest.mark.gpu
def test_resnet_replicas_run_on_the_owners_packed_operands():
    from torch.nn.parallel import parallel_apply, replicate
    m = build_model(dict(syn.synthetic_resnet_state_dict(seed=0)), {"trainer": "CoOp"}).cuda()
    images = torch.randn(4, 3, m.visual.input_resolution, m.visual.input_resolution, generator=torch.Generator().manual_seed(2)).cuda().to(m.dtype)
    ref = m.visual(images)
    out = parallel_apply(replicate(m.visual, [0, 0]), [(images[:2],), (images[2:],)], devices=[0, 0])
    moved = m.visual._packed_on(torch.device("cuda", 0))
    torch.cuda.synchronize()
    assert torch.equal(torch.cat(out), ref) and moved is m.visual._packed
