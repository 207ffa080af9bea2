"""Dead-row elimination in the causal text tower (include/clipmi.h, ``seq_rows`` of clipmi_text_encoder / clipmi_encode_text).

Reference: the text blocks mask causally (clip/model.py:585-591) and only the EOT row leaves the tower (clip/model.py:611; coop.py:65),
so token rows behind the last prompt's EOT cannot reach any output.  The library computes ``seq_rows`` token positions per prompt
instead of all 77.  Tested here: the truncated tower against the full-length one (same library, same weights).  With the SAME GEMM
kernels on both sides (option gemm_variant pinned) the features agree to <= 1e-6 in cosine -- the arithmetic per element is the same.
With the cost model free to choose, 24 000 and 77 000 rows get different kernels (row-range residual GEMMs add the residual inside the
K loop, tile kernels in the epilogue: another fp32 summation order, then one fp16 rounding of the stream's shadow) and the two runs
differ as ANY two batch sizes do (<= 2e-4 on normalised features) -- both within the tolerance against the fp32
oracle.  Also: ragged EOTs, a prompt whose EOT is the very last token, the prompt hook of MaPLe, the fp16 stream, rows that must
never be read, and the bookkeeping of ``CLIP.live_rows`` (one read-back per NEW prompt set)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from clip_calibration_amd import _lib, ops, synthetic as syn  # noqa: E402
from clip_calibration_amd._lib import check, lib  # noqa: E402
from oracle import clip_oracle as orc  # noqa: E402  (checker only)

PLAIN = {"trainer": "CoOp", "vision_depth": 0, "language_depth": 0, "vision_ctx": 0, "language_ctx": 0}


def _build(gname, dd=None, seed=0):
    from clip_calibration_amd.model import build_model
    sd = syn.synthetic_state_dict(gname, seed=seed)
    return sd, build_model(dict(sd), dict(dd or PLAIN)).cuda()


def _cosine_gap(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double(), b.double()
    a = a / a.norm(dim=-1, keepdim=True)
    b = b / b.norm(dim=-1, keepdim=True)
    return float((a @ a.T - b @ a.T).abs().max())


def _ragged_ids(n, gname, seed, longest=None):
    """CoOp-shaped prompts with ragged EOT positions; ``longest``: move one prompt's EOT to that token index."""
    ids = syn.synthetic_token_ids(n, gname, seed=seed, n_ctx_placeholders=4)
    if longest is not None:
        g = syn.GEOMETRIES[gname]
        row = ids[n // 2].clone()
        e = int(row.argmax())
        filler = torch.randint(8, g.vocab_size - 8, (longest - e,), generator=torch.Generator().manual_seed(seed))
        ids[n // 2, e:longest] = filler
        ids[n // 2, longest] = g.vocab_size - 1
    return ids


def _raw_text_encoder(model, prompts, eot, seq_rows, flags=0):
    """clipmi_text_encoder through the C ABI with an explicit ``seq_rows`` (the mirrors derive it; here it is the test's variable)."""
    model._ensure_bound()
    Cn = prompts.shape[0]
    out = torch.empty(Cn, model.geometry.embed_dim, dtype=torch.float32, device="cuda")
    ws = torch.empty(lib.clipmi_text_workspace_bytes(model._handle, Cn, seq_rows), dtype=torch.uint8, device="cuda")
    check(lib.clipmi_text_encoder(model._handle, prompts.data_ptr(), _lib.F16 if prompts.dtype == torch.float16 else _lib.F32, eot.data_ptr(), Cn,
                                  seq_rows, None, out.data_ptr(), ws.data_ptr(), ws.numel(), flags, ops._stream()), "clipmi_text_encoder")
    return out


@pytest.mark.parametrize("gname,n,variant", [("tiny", 37, -1), ("ViT-B/16", 1000, 0), ("ViT-B/16", 1000, -1)])
def test_truncated_tower_equals_full_length_tower(gname, n, variant, clipmi_option):
    """encode_text at C = 1000 (BASELINE configs[1]'s class list) and a small ragged case: dead rows off vs on.  variant 0 pins every GEMM to
    the 128 x 128 tile kernel on both sides (same summation order: 1e-6); -1 leaves the cost model free (batch-size-class difference)."""
    sd, model = _build(gname)
    if variant >= 0:
        clipmi_option("gemm_variant", variant)
        clipmi_option("gemm_stream", 0)
        clipmi_option("gemm_rstream", 0)
    ids = _ragged_ids(n, gname, seed=3).cuda()
    rows = model.live_rows(ids)
    eot_max = int(ids.argmax(-1).max())
    assert eot_max < rows <= (eot_max + 8) // 8 * 8 < model.context_length
    with torch.no_grad():
        short = model.text_features_f32(ids)
        model.text_dead_row_elimination = False
        assert model.live_rows(ids) == model.context_length
        full = model.text_features_f32(ids)
    torch.cuda.synchronize()
    assert torch.isfinite(short).all()
    gap = _cosine_gap(short, full)
    print(f"[{gname} x {n}, gemm_variant {variant}] rows {rows} of {model.context_length}: cosine gap {gap:.2e}, "
          f"max |d feature| / max |feature| {float((short - full).abs().max()) / float(full.abs().max()):.2e}")
    same_kernels = variant >= 0 or gname == "tiny"
    assert gap <= (1e-6 if same_kernels else 2e-4)
    if gname == "ViT-B/16":      # both against the fp32 oracle on a slice (the 12-layer tower on the CPU: 24 prompts)
        with torch.no_grad():
            ref = orc.encode_text(sd, ids[:24].cpu()).numpy()
        for got in (short[:24].cpu().numpy(), full[:24].cpu().numpy()):
            assert np.abs(got - ref).max() <= 5e-3 * np.abs(ref).max()


def test_prompt_ending_on_the_last_token_keeps_every_row():
    sd, model = _build("tiny")
    ids = _ragged_ids(24, "tiny", seed=5, longest=76).cuda()
    assert int(ids.argmax(-1).max()) == 76 and model.live_rows(ids) == 77
    with torch.no_grad():
        a = model.text_features_f32(ids)
        model.text_dead_row_elimination = False
        b = model.text_features_f32(ids)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    ref = orc.encode_text(sd, ids.cpu()).numpy()
    got = a.cpu().numpy()
    assert np.abs(got - ref).max() <= 5e-3 * np.abs(ref).max()


@pytest.mark.parametrize("flags", [_lib.CALL_STREAM_F32, _lib.CALL_STREAM_F16])
def test_every_row_count_through_the_c_abi(flags):
    """seq_rows from the tightest bound (max EOT + 1, not a multiple of anything) to the whole context, 0 and out-of-range values: all
    the same features; against the oracle once."""
    sd, model = _build("tiny")
    g = model.geometry
    ids = _ragged_ids(19, "tiny", seed=7, longest=29)
    eot = ids.argmax(-1).to(torch.int32).cuda()
    prompts = model.token_embedding(ids.cuda()).to(torch.float16).contiguous()
    with torch.no_grad():
        full = _raw_text_encoder(model, prompts, eot, 0, flags)
        outs = {r: _raw_text_encoder(model, prompts, eot, r, flags) for r in (30, 31, 32, 40, 64, 76, 77, 78, 1000, -3)}
    torch.cuda.synchronize()
    tol = 1e-6 if flags == _lib.CALL_STREAM_F32 else 2e-4      # fp16 stream: tile choice moves the fp16 roundings of the stream's stats
    for r, o in outs.items():
        assert _cosine_gap(o, full) <= tol, r
    for r in (77, 78, 1000, -3):
        assert torch.equal(outs[r], full), r
    with torch.no_grad():
        ref = orc.text_encoder(sd, prompts.float().cpu(), ids).numpy()
    got = outs[30].cpu().numpy()
    assert np.abs(got - ref).max() <= 5e-3 * np.abs(ref).max()


def test_rows_behind_the_bound_are_never_read():
    """Poison (NaN) in every prompt row >= seq_rows and in the ids behind the EOT's bound: the features do not change."""
    sd, model = _build("tiny")
    ids = _ragged_ids(16, "tiny", seed=9).cuda()
    rows = model.live_rows(ids)
    eot = ids.argmax(-1).to(torch.int32)
    prompts = model.token_embedding(ids).to(torch.float32).contiguous()
    poisoned = prompts.clone()
    poisoned[:, rows:, :] = float("nan")
    with torch.no_grad():
        a = _raw_text_encoder(model, prompts, eot, rows)
        b = _raw_text_encoder(model, poisoned, eot, rows)
    torch.cuda.synchronize()
    assert torch.isfinite(b).all() and torch.equal(a, b)


def test_live_rows_bookkeeping():
    sd, model = _build("tiny")
    ids = _ragged_ids(8, "tiny", seed=11).cuda()
    r = model.live_rows(ids)
    assert r % 8 == 0 and r == model.live_rows(ids) and len(model._live_rows) == 1
    view = ids[:]                                         # same storage, same version: the same entry
    assert model.live_rows(view) == r and len(model._live_rows) == 1
    ids[0, 40] = model.vocab_size - 1                      # in-place edit: the version counter moves, the bound is recomputed
    ids[0, :40] = 5
    assert model.live_rows(ids) == 48
    assert model.live_rows(ids, n_ctx=60) == 61            # a hook's prompt tokens 1..n_ctx are always computed
    for k in range(40):                                    # bounded
        model.live_rows(_ragged_ids(4, "tiny", seed=100 + k))
    assert len(model._live_rows) <= 16
    model.text_dead_row_elimination = False
    assert model.live_rows(ids) == model.context_length


def test_text_paths_under_inference_mode():
    """Callers that tokenise or move ids inside torch.inference_mode() hand over tensors without a version counter (reading ``_version`` raises):
    encode_text / text_features_f32 / text_encoder_f32 compute the row bound per call instead of caching it -- same bits as outside."""
    sd, model = _build("tiny")
    ids = _ragged_ids(9, "tiny", seed=21)
    want = model.encode_text(ids.cuda())
    with torch.inference_mode():
        inf = ids.clone().cuda()
        with pytest.raises(RuntimeError):
            inf._version
        got = model.encode_text(inf)
        f32 = model.text_features_f32(inf)
        prompts = model.token_embedding(inf).type(model.dtype)
        enc = model.text_encoder_f32(prompts, inf)
    assert torch.equal(got, want) and torch.isfinite(f32).all() and torch.isfinite(enc).all()
    assert torch.equal(f32, model.text_features_f32(ids.cuda()))


def test_maple_hook_with_truncated_rows():
    """MaPLe's deep text prompts overwrite tokens 1..n_ctx of every block's input (clip/model.py:287-331): unaffected by the row bound."""
    dd = {"trainer": "MaPLe", "vision_depth": 0, "language_depth": 0, "vision_ctx": 0, "language_ctx": 0, "maple_length": 2}
    sd, model = _build("tiny", dd)
    g = model.geometry
    ids = _ragged_ids(14, "tiny", seed=13).cuda()
    gen = torch.Generator().manual_seed(3)
    prompts = model.token_embedding(ids).to(torch.float16)
    deep = [(0.02 * torch.randn(2, g.transformer_width, generator=gen)).cuda() for _ in range(g.transformer_layers - 1)]
    with torch.no_grad():
        short = model.text_encoder_f32(prompts, ids, deep, 2)
        model.text_dead_row_elimination = False
        full = model.text_encoder_f32(prompts, ids, deep, 2)
        none = model.text_encoder_f32(prompts, ids)
    torch.cuda.synchronize()
    assert _cosine_gap(short, full) <= 1e-6
    assert _cosine_gap(none, full) > 1e-4                  # the hook matters
    # a bound that would cut into the hook's tokens is refused, not computed
    model._ensure_bound()
    hook, keep = model._hook(2, None, deep, g.transformer_layers - 1)
    out = torch.empty(14, g.embed_dim, dtype=torch.float32, device="cuda")
    ws = torch.empty(lib.clipmi_text_workspace_bytes(model._handle, 14, 0), dtype=torch.uint8, device="cuda")
    eot = ids.argmax(-1).to(torch.int32)
    rc = lib.clipmi_text_encoder(model._handle, prompts.data_ptr(), _lib.F16, eot.data_ptr(), 14, 2, C.byref(hook), out.data_ptr(), ws.data_ptr(),
                                 ws.numel(), 0, ops._stream())
    assert rc == _lib.ERR_SHAPE and "n_ctx" in _lib.last_error()


def test_workspace_follows_the_row_count():
    sd, model = _build("tiny")
    model._ensure_bound()
    full = lib.clipmi_text_workspace_bytes(model._handle, 100, 0)
    assert lib.clipmi_text_workspace_bytes(model._handle, 100, 77) == full == lib.clipmi_text_workspace_bytes(model._handle, 100, 500)
    assert lib.clipmi_text_workspace_bytes(model._handle, 100, 24) < 0.4 * full
    # a workspace sized for fewer rows than the call asks for is refused
    ids = _ragged_ids(100, "tiny", seed=1).cuda()
    eot = ids.argmax(-1).to(torch.int32)
    prompts = model.token_embedding(ids).to(torch.float16).contiguous()
    out = torch.empty(100, model.geometry.embed_dim, dtype=torch.float32, device="cuda")
    ws = torch.empty(lib.clipmi_text_workspace_bytes(model._handle, 100, 24), dtype=torch.uint8, device="cuda")
    rc = lib.clipmi_text_encoder(model._handle, prompts.data_ptr(), _lib.F16, eot.data_ptr(), 100, 32, None, out.data_ptr(), ws.data_ptr(), ws.numel(), 0,
                                 ops._stream())
    assert rc == _lib.ERR_WORKSPACE


# ---- Level 1: clip_model.transformer(x) with the opt-in row hint (clipmi_text_blocks `seq_rows`) -----------------------------------
def test_transformer_row_hint_returns_the_live_rows_and_zeros_behind(clipmi_option):
    """``clip_model.transformer.live_rows = tokenized_prompts``: rows up to the bound are the rows of the full-length call, the rows behind
    are zero, and the reference's TextEncoder arithmetic on top (coop.py:56-67) gives the same features; without the hint nothing changes."""
    clipmi_option("gemm_variant", 0)
    clipmi_option("gemm_stream", 0)
    clipmi_option("gemm_rstream", 0)
    sd, model = _build("tiny")
    ids = _ragged_ids(21, "tiny", seed=17, longest=37).cuda()
    x = (model.token_embedding(ids) + model.positional_embedding).permute(1, 0, 2)       # LND, as coop.py:57-58 hands it over
    with torch.no_grad():
        full = model.transformer(x)
        model.transformer.live_rows = ids
        short = model.transformer(x)
        model.transformer.live_rows = 38                                                   # an int: taken as is
        tight = model.transformer(x)
        model.transformer.live_rows = None
        again = model.transformer(x)
    torch.cuda.synchronize()
    rows = model.live_rows(ids)
    assert rows == 40 and full.shape == short.shape == x.shape
    assert torch.equal(again, full)
    assert torch.equal(short[:rows], full[:rows]) and not short[rows:].any()
    assert torch.equal(tight[:38], full[:38]) and not tight[38:].any()
    eot = ids.argmax(-1)
    pick = lambda y: model.ln_final(y.permute(1, 0, 2)).float()[torch.arange(21), eot] @ model.text_projection.float()
    assert torch.equal(pick(short), pick(full))


def test_transformer_row_hint_with_the_maple_list_form():
    dd = {"trainer": "MaPLe", "vision_depth": 0, "language_depth": 0, "vision_ctx": 0, "language_ctx": 0, "maple_length": 2}
    sd, model = _build("tiny", dd)
    g = model.geometry
    ids = _ragged_ids(10, "tiny", seed=19).cuda()
    gen = torch.Generator().manual_seed(5)
    deep = [(0.02 * torch.randn(2, g.transformer_width, generator=gen)).cuda() for _ in range(g.transformer_layers - 1)]
    x = (model.token_embedding(ids) + model.positional_embedding).permute(1, 0, 2).half()
    with torch.no_grad():
        full, _, n_full = model.transformer([x, deep, 0])
        model.transformer.live_rows = ids
        short, _, n_short = model.transformer([x, deep, 0])
        model.transformer.live_rows = 1                     # below the hook's tokens: raised to 1 + n_ctx, never cut into them
        tiny, _, _ = model.transformer([x, deep, 0])
    torch.cuda.synchronize()
    rows = model.live_rows(ids, 2)
    assert n_full == n_short and short.dtype == x.dtype
    assert float((short[:rows].float() - full[:rows].float()).abs().max()) <= 2e-3 * float(full.float().abs().max()) and not short[rows:].any()
    assert float((tiny[:3].float() - full[:3].float()).abs().max()) <= 2e-3 * float(full.float().abs().max()) and not tiny[3:].any()


def test_text_blocks_in_place_only_without_a_row_bound():
    sd, model = _build("tiny")
    model._ensure_bound()
    g = model.geometry
    x = torch.randn(6, g.context_length, g.transformer_width, device="cuda")
    ws = torch.empty(lib.clipmi_text_workspace_bytes(model._handle, 6, 0), dtype=torch.uint8, device="cuda")
    args = (_lib.F32, 6)
    rc = lib.clipmi_text_blocks(model._handle, x.data_ptr(), x.data_ptr(), *args, 24, None, ws.data_ptr(), ws.numel(), 0, ops._stream())
    assert rc == _lib.ERR_ARG and "in place" in _lib.last_error()
    for r in (0, 77, 500):                                  # every row: in place is fine, as before
        check(lib.clipmi_text_blocks(model._handle, x.data_ptr(), x.data_ptr(), *args, r, None, ws.data_ptr(), ws.numel(), 0, ops._stream()), "text_blocks")
    torch.cuda.synchronize()
    assert torch.isfinite(x).all()
