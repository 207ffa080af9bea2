"""Operator-level parity of the HIP kernels (through the C ABI) against the CPU oracle.  Needs a real MI355X."""
import math

import numpy as np
import pytest
import torch

from clip_calibration_amd import _lib
from conftest import load_golden

pytestmark = pytest.mark.gpu

from oracle import clip_oracle as orc  # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def ops():
    from clip_calibration_amd import ops as _ops
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return _ops


def _cuda(t):
    return t.cuda()


# tolerance notes: operands are fp16-exact on both sides, accumulation is fp32 on both sides; what differs is the
# summation order and (for fp16 outputs) one final rounding of 2^-11 relative.
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 192, 128), (197 * 3, 768, 768), (1000, 2304, 768),
                                    (333, 512, 3072), (50432, 768, 768), (7, 64, 64), (1, 4, 64)])
@pytest.mark.parametrize("epi", ["none", "bias", "gelu", "residual"])
def test_gemm(ops, M, N, K, epi):
    if M == 50432 and epi not in ("gelu", "residual"):
        pytest.skip("full-size case runs once per output dtype")
    from clip_calibration_amd import _lib
    g = torch.Generator().manual_seed(M * 7 + N + K)
    a = (torch.randn(M, K, generator=g)).half()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).half()
    bias = torch.randn(N, generator=g) * 0.1
    res = torch.randn(M, N, generator=g)
    ref = a.float() @ w.float().t()
    if epi == "none":
        out = ops.gemm_f16(_cuda(a), _cuda(w), epilogue=_lib.EPI_NONE, out_dtype=torch.float32)
        tol = 2e-4
    elif epi == "bias":
        ref = ref + bias
        out = ops.gemm_f16(_cuda(a), _cuda(w), _cuda(bias), epilogue=_lib.EPI_BIAS, out_dtype=torch.float16)
        tol = 2e-3
    elif epi == "gelu":
        ref = orc.quick_gelu(ref + bias)
        out = ops.gemm_f16(_cuda(a), _cuda(w), _cuda(bias), epilogue=_lib.EPI_BIAS_QUICKGELU, out_dtype=torch.float16)
        tol = 2e-3
    else:
        ref = ref + bias + res
        r = _cuda(res)
        out = ops.gemm_f16(_cuda(a), _cuda(w), _cuda(bias), residual=r, epilogue=_lib.EPI_BIAS_RESIDUAL,
                           out_dtype=torch.float32, out=r)   # in place, as the towers use it
        tol = 2e-4
    got = out.float().cpu()
    scale = ref.abs().max().item() + 1e-6
    err = (got - ref).abs().max().item()
    assert err <= tol * scale, f"max err {err} vs scale {scale}"


@pytest.mark.parametrize("variant", ["0", "1", "a", "s", "r"])
@pytest.mark.parametrize("M,N,K,epi", [(200, 192, 128, "gelu"), (1000, 2304, 768, "bias"), (333, 512, 3072, "residual"),
                                        (50432, 768, 768, "residual"), (513, 260, 64, "none"), (5000, 3072, 768, "gelu"),
                                        (50432, 768, 768, "gelu"), (3000, 520, 640, "bias"), (40000, 2048, 512, "gelu"), (3000, 512, 512, "none")])
def test_gemm_tile_variants(ops, clipmi_option, variant, M, N, K, epi):
    """Every tile/pipeline variant the dispatcher can pick (option gemm_variant) computes the same thing."""
    clipmi_option("gemm_variant", _lib.gemm_variant_id(variant))
    test_gemm(ops, M, N, K, epi)


def test_gemm_stream_kernel_race_screen(ops, clipmi_option):
    """The ping-pong persistent GEMM (gemm_stream = 1, the default for multi-round fp16-out GEMMs; gemm.hip gemm_stream_kernel)
    hands LDS stages between two wave groups that run one phase apart, with counted vmcnt / lgkmcnt waits and LDS-DMA in flight
    across barriers.  A misplaced wait shows up as a rare wrong tile, so: the image tower's c_fc shape, 60 back-to-back launches
    on changing operands with an HBM-bound stream queued in front of every third (uneven load).  With the bias epilogue the
    kernel shares the one-tile-per-workgroup kernel's arithmetic (same K order, same MFMA) and must match it bit for bit; with
    QuickGELU the two compilations contract the element-wise math differently (a 1-ulp difference on ~3e-5 of the elements), so
    that arm checks run-to-run identity plus closeness."""
    g = torch.Generator().manual_seed(11)
    M, N, K = 50432, 3072, 768
    a0 = (torch.randn(M, K, generator=g) * 0.5).half().cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).half().cuda()
    bias = torch.randn(N, generator=g).cuda()
    filler = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    bad = 0
    for it in range(20):
        a = torch.roll(a0, shifts=it * 37, dims=0)
        epi = _lib.EPI_BIAS if it % 2 == 0 else _lib.EPI_BIAS_QUICKGELU
        clipmi_option("gemm_stream", 0)
        ref = ops.gemm_f16(a, w, bias, epilogue=epi, out_dtype=torch.float16)
        clipmi_option("gemm_stream", 1)
        first = None
        for rep in range(3):
            if (it + rep) % 3 == 0:
                filler.add_(1)
            out = ops.gemm_f16(a, w, bias, epilogue=epi, out_dtype=torch.float16)
            if epi == _lib.EPI_BIAS:
                bad += int(not torch.equal(out, ref))
            else:
                first = out.clone() if first is None else first
                ok = torch.equal(out, first) and float((out.float() - ref.float()).abs().max()) <= 2 ** -8 and int((out != ref).sum()) < 1e-4 * out.numel()
                bad += int(not ok)
    assert bad == 0, f"{bad} of 60 launches differ"


def test_gemm_stream_kernel_beyond_one_descriptor(ops):
    """gemm_stream_kernel addresses each operand through ONE buffer descriptor per launch (round 6): a matrix beyond 2 GiB -- the hidden
    activations of a text tower of a few hundred thousand token rows -- runs as consecutive launches over row ranges that fit (gemm.hip launch_one).
    270 000 x 4096 fp16 outputs = 2.2 GB: checked on rows around every range boundary and at both ends against an fp32 reference, and for
    untouched memory behind the last row."""
    g = torch.Generator().manual_seed(5)
    M, N, K = 270000, 4096, 512
    a = (torch.randn(M, K, generator=g) * 0.5).half().cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).half().cuda()
    bias = torch.randn(N, generator=g).cuda()
    out = ops.gemm_f16(a, w, bias, epilogue=_lib.EPI_BIAS_QUICKGELU, out_dtype=torch.float16)
    assert out.shape == (M, N) and out.numel() * 2 > 2 ** 31
    per_row = 2 * N
    step = (((2 ** 31 - 2 ** 25) // per_row) - 256) // 256 * 256
    rows = torch.cat([torch.arange(0, 300)] + [torch.arange(b - 300, min(M, b + 300)) for b in range(step, M, step)] + [torch.arange(M - 300, M)]).unique()
    ref = torch.nn.functional.linear(a[rows].float(), w.float(), bias)
    ref = ref * torch.sigmoid(1.702 * ref)
    assert float((out[rows].float() - ref).abs().max()) <= 2e-3 * float(ref.abs().max())
    assert torch.isfinite(out[::997]).all()


@pytest.mark.parametrize("M,N,K", [(16448, 3072, 768), (36928, 4096, 1024), (32896, 3072, 1024), (33000, 4096, 1024)])
def test_gemm_ragged_last_tile_row_as_its_own_launch(ops, clipmi_option, M, N, K):
    """gemm_split_rows (default 1): a last row of 256-row tiles with <= 128 live rows that would open a round of its own in the persistent kernel
    (65 x 12 tiles = 3.05 rounds of 256 workgroups; ViT-L/14@336 c_fc: 145 x 16 = 9.06; ViT-L/14 in-proj: 129 x 12 = 6.05) goes to the tile kernels as
    a second launch.  With the bias epilogue every kernel involved shares one arithmetic: bit for bit against the single launch; QuickGELU: the usual
    1-ulp contraction difference on a few elements; (33000, ...): 232 live rows in the last tile row -- not split, the same launch either way."""
    g = torch.Generator().manual_seed(M + N)
    a = (torch.randn(M, K, generator=g) * 0.5).half().cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).half().cuda()
    bias = torch.randn(N, generator=g).cuda()
    for epi in (_lib.EPI_BIAS, _lib.EPI_BIAS_QUICKGELU):
        clipmi_option("gemm_split_rows", 0)
        one = ops.gemm_f16(a, w, bias, epilogue=epi, out_dtype=torch.float16)
        clipmi_option("gemm_split_rows", 1)
        two = ops.gemm_f16(a, w, bias, epilogue=epi, out_dtype=torch.float16)
        again = ops.gemm_f16(a, w, bias, epilogue=epi, out_dtype=torch.float16)
        assert torch.equal(two, again)
        if epi == _lib.EPI_BIAS:
            assert torch.equal(one, two)
        else:
            assert float((one.float() - two.float()).abs().max()) <= 2 ** -8 and int((one != two).sum()) < 1e-4 * one.numel()
    rows = torch.cat([torch.arange(0, 300), torch.arange(M - 300, M)])
    ref = torch.nn.functional.linear(a[rows].float(), w.float(), bias)
    ref = ref * torch.sigmoid(1.702 * ref)
    assert float((two[rows].float() - ref).abs().max()) <= 2e-3 * float(ref.abs().max())


def test_gemm_pingpong_tile_kernel_race_screen(ops, clipmi_option):
    """gemm_pp_kernel (the 320 x 256 one-tile-per-workgroup kernel with the ping-pong main loop, gemm_variant 10: what the cost
    model picks for N = 768 at M = 50432) against the compiler-scheduled two-stage loop of gemm_f16_kernel (256 x 256, variant 1):
    same K order, same MFMA, same epilogue code -- bit-identical, here on the image tower's c_proj shape (48 K-steps, in-place
    fp32 residual epilogue) and the out-proj shape, 20 launches with an HBM-bound stream queued in front of every third."""
    g = torch.Generator().manual_seed(13)
    filler = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    bad = 0
    for (M, N, K) in [(50432, 768, 3072), (50432, 768, 768)]:
        a0 = (torch.randn(M, K, generator=g) * 0.5).half().cuda()
        w = (torch.randn(N, K, generator=g) * 0.05).half().cuda()
        bias = torch.randn(N, generator=g).cuda()
        res = torch.randn(M, N, generator=g).cuda()
        for it in range(10):
            a = torch.roll(a0, shifts=it * 53, dims=0)
            clipmi_option("gemm_variant", 1)
            ref = ops.gemm_f16(a, w, bias, residual=res.clone(), epilogue=_lib.EPI_BIAS_RESIDUAL, out_dtype=torch.float32)
            clipmi_option("gemm_variant", 10)
            if it % 3 == 0:
                filler.add_(1)
            out = ops.gemm_f16(a, w, bias, residual=res.clone(), epilogue=_lib.EPI_BIAS_RESIDUAL, out_dtype=torch.float32)
            bad += int(not torch.equal(out, ref))
    assert bad == 0, f"{bad} of 20 launches differ from the two-stage loop"


def _residual_ref(a, w, bias, x):
    """fp16(x + a @ w^T + bias) with ONE rounding of the fp32 sum, and the (sum, sumsq) partials per 256-column tile of the rounded rows."""
    y = (a.float() @ w.float().t() + bias + x.float()).half()
    yf = y.float()
    parts = [(yf[:, c:c + 256].sum(1), (yf[:, c:c + 256] ** 2).sum(1)) for c in range(0, y.shape[1], 256)]
    return y, parts


def _ulp_close(x, ref, what, frac=4e-3):
    """fp16 tensors that are the same fp32 sums added in a different order: equal except at rounding boundaries, never more than one
    fp16 ulp apart -- or, where terms of size O(1) cancel to almost nothing, a few fp32 roundings of those terms (1e-5 at K = 3072)."""
    d = (x.float() - ref.float()).abs()
    tol = torch.clamp(ref.float().abs() * (1.05 * 2.0 ** -10), min=1e-5)   # 1.05: |ref| * 2^-10 is the ulp only up to float rounding at a binade edge
    assert bool((d <= tol).all()), f"{what}: more than one fp16 ulp apart (max {float((d / tol).max()):.2f} x the tolerance)"
    assert float((x != ref).float().mean()) < frac, f"{what}: {float((x != ref).float().mean()):.2e} of the elements differ"


@pytest.mark.parametrize("M,N,K", [(50432, 768, 768), (25807, 768, 3072), (77000, 512, 512), (9000, 1024, 512), (30000, 520, 512),
                                    (3000, 520, 640), (64, 256, 512), (2100, 768, 128)])
def test_gemm_residual_f16_vs_reference(ops, clipmi_option, M, N, K):
    """clipmi_gemm_residual_f16 (the fp16-stream residual GEMM of a block, clip/model.py:186-187) in its three kernels -- persistent row
    ranges with the residual added to the accumulators DURING the K loop (gemm_variant 16; shapes it does not take fall back to the
    256 x 256 tiles), 320 x 256 ping-pong tiles (10), 256 x 256 tiles (1) -- against a plain fp32 computation rounded once.  The two
    tile kernels agree BIT FOR BIT (outputs and row partials); the row-range kernel adds the same terms in another order (bias, a few
    K-steps of products, the residual, the other products -- instead of products + bias + residual): equal except at rounding
    boundaries, never more than one fp16 ulp.  Shapes: ragged M (25807: one 9-10 pair tile per workgroup), a last column tile of 8
    columns (520), K = 512 (eight K-steps), and shapes only the tile kernels take (short ranges, four column tiles, K = 128)."""
    g = torch.Generator().manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g) * 0.5).half().cuda()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).half().cuda()
    bias = (torch.randn(N, generator=g) * 0.1).cuda()
    x0 = torch.randn(M, N, generator=g).half().cuda()
    outs = {}
    for v in (16, 10, 1):
        clipmi_option("gemm_variant", v)
        x = x0.clone()
        stats, parts = ops.gemm_residual_f16(a, w, bias, x)
        outs[v] = (x, stats[:parts].clone(), parts)
    assert outs[16][2] == outs[10][2] == outs[1][2] == (N + 255) // 256
    assert torch.equal(outs[10][0], outs[1][0]) and torch.equal(outs[10][1], outs[1][1]), "the two tile kernels differ"
    _ulp_close(outs[16][0], outs[10][0], "row-range kernel vs tile kernel")
    y, parts = _residual_ref(a.cpu(), w.cpu(), bias.cpu(), x0.cpu())
    for v in (16, 10):
        got = outs[v][0].cpu()
        _ulp_close(got, y, f"variant {v} vs fp32 reference", frac=2e-2)
        st = outs[v][1].cpu()
        gf = got.float()
        for t in range(len(parts)):
            ex = gf[:, t * 256:(t + 1) * 256]                                   # the partials are those of the kernel's OWN rounded rows
            np.testing.assert_allclose(st[t, :, 0].numpy(), ex.sum(1).numpy(), rtol=1e-4, atol=2e-3)
            np.testing.assert_allclose(st[t, :, 1].numpy(), (ex ** 2).sum(1).numpy(), rtol=1e-4, atol=2e-3)


def test_gemm_residual_stream_race_screen(ops, clipmi_option):
    """gemm_rstream_kernel hands LDS stages between two wave groups one part apart, keeps LDS-DMA, stores and residual loads in
    flight across barriers behind counted waits, and preloads the next tile's residual into accumulators that the current tile has
    just left.  A misplaced wait shows up as a rare wrong tile: the image tower's out-proj and c_proj shapes, 10 changing operand sets x
    3 launches each with an HBM-bound stream queued in front of every third (uneven load): every launch bit for bit equal to the
    first of its set, and within one fp16 ulp of the 320 x 256 tile kernel."""
    g = torch.Generator().manual_seed(17)
    filler = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    bad = 0
    for (M, N, K) in [(50432, 768, 768), (50432, 768, 3072)]:
        a0 = (torch.randn(M, K, generator=g) * 0.5).half().cuda()
        w = (torch.randn(N, K, generator=g) * 0.05).half().cuda()
        bias = torch.randn(N, generator=g).cuda()
        x0 = torch.randn(M, N, generator=g).half().cuda()
        for it in range(10):
            a = torch.roll(a0, shifts=it * 41, dims=0)
            clipmi_option("gemm_variant", 10)
            xr = x0.clone()
            ops.gemm_residual_f16(a, w, bias, xr)
            clipmi_option("gemm_variant", 16)                                    # the row-range kernel (the default dispatch takes it for K <= 1536)
            first = None
            for rep in range(3):
                if (it + rep) % 3 == 0:
                    filler.add_(1)
                x = x0.clone()
                st, p = ops.gemm_residual_f16(a, w, bias, x)
                if first is None:
                    first = (x, st[:p].clone())
                    _ulp_close(x, xr, "row-range kernel vs tile kernel")
                bad += int(not (torch.equal(x, first[0]) and torch.equal(st[:p], first[1])))
    assert bad == 0, f"{bad} of 60 launches differ from the first launch of their operand set"


def test_gemm_rejects_bad_shapes(ops):
    from clip_calibration_amd._lib import ClipmiError
    a = torch.zeros(8, 48, dtype=torch.float16, device="cuda")
    w = torch.zeros(8, 48, dtype=torch.float16, device="cuda")
    with pytest.raises(ClipmiError):
        ops.gemm_f16(a, w)          # K % 64 != 0
    with pytest.raises(RuntimeError):
        ops.gemm_f16(a.cpu(), w)    # CPU tensor: no fallback


@pytest.mark.parametrize("rows,D", [(1, 64), (5, 128), (197 * 4, 768), (77 * 3, 512), (33, 1024), (9, 192), (3, 2048)])
@pytest.mark.parametrize("dt_in,dt_out", [(torch.float32, torch.float16), (torch.float32, torch.float32),
                                          (torch.float16, torch.float16)])
def test_layernorm(ops, rows, D, dt_in, dt_out):
    g = torch.Generator().manual_seed(rows + D)
    x = (torch.randn(rows, D, generator=g) * 3 + 0.5).to(dt_in)
    gamma = 1 + 0.1 * torch.randn(D, generator=g)
    beta = 0.1 * torch.randn(D, generator=g)
    ref = orc.layer_norm(x.float(), gamma, beta)
    got = ops.layernorm(_cuda(x), _cuda(gamma), _cuda(beta), out_dtype=dt_out).float().cpu()
    tol = 2e-3 if dt_out == torch.float16 else 2e-5
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=tol, atol=tol * 4)


def test_layernorm_gather(ops):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(40, 128, generator=g)
    idx = torch.tensor([3, 39, 0, 17], dtype=torch.int32)
    gamma, beta = torch.ones(128), torch.zeros(128)
    ref = orc.layer_norm(x[idx.long()], gamma, beta)
    got = ops.layernorm(_cuda(x), _cuda(gamma), _cuda(beta), gather_rows=_cuda(idx)).cpu()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=2e-5, atol=1e-5)


def _attn_ref(qkv, n, l, h, causal):
    d = 64 * h
    q, k, v = qkv.float().reshape(n, l, 3 * d).split(d, dim=-1)
    q = q.reshape(n, l, h, 64).transpose(1, 2)
    k = k.reshape(n, l, h, 64).transpose(1, 2)
    v = v.reshape(n, l, h, 64).transpose(1, 2)
    s = q @ k.transpose(-1, -2) / 8.0
    if causal:
        s = s + orc.causal_mask(l)
    return (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(n * l, d)


@pytest.mark.parametrize("n,l,h,causal", [(2, 197, 12, False), (3, 77, 8, True), (2, 17, 2, False), (1, 10, 3, False),
                                          (2, 199, 12, False), (1, 257, 16, False), (1, 577, 4, False), (2, 77, 1, True),
                                          (1, 32, 1, True), (1, 1, 1, False), (1, 225, 2, True)])
def test_attention(ops, n, l, h, causal):
    g = torch.Generator().manual_seed(n * 1000 + l + h)
    qkv = (torch.randn(n * l, 3 * 64 * h, generator=g) * 1.5).half()
    ref = _attn_ref(qkv, n, l, h, causal)
    got = ops.attention(_cuda(qkv), n, l, h, causal).float().cpu()
    err = (got - ref).abs().max().item()
    # P and V are fp16 operands of the PV product: ~2^-11 relative on values of size O(1)
    assert err < 4e-3, f"max err {err}"


@pytest.mark.parametrize("n,l,h", [(3, 197, 12), (2, 199, 12), (1, 193, 2), (5, 200, 1), (40, 197, 12)])
def test_attention_vision_kernel_vs_persistent(ops, clipmi_option, n, l, h):
    """193..200-token non-causal attention (the image towers at 224 px; clip/model.py:181-183): the all-DMA kernel with a loader
    wave (attn_loader 1: fragment reads pinned ahead of their MFMAs by inline asm; 2, the default: its output rows stored
    non-temporal) and the persistent kernel (attn_loader 0) give the SAME bits -- only the operand transport and the instruction order differ -- and match the fp32
    reference.  40 sequences x 12 heads = several items per workgroup (both LDS buffers in use)."""
    clipmi_option("attn_loader", 1)
    g = torch.Generator().manual_seed(n * 1000 + l + h)
    qkv = (torch.randn(n * l, 3 * 64 * h, generator=g) * 1.5).half()
    got = ops.attention(_cuda(qkv), n, l, h, False)
    clipmi_option("attn_loader", 2)   # the same kernel with non-temporal output stores
    got_nt = ops.attention(_cuda(qkv), n, l, h, False)
    clipmi_option("attn_loader", 0)
    base = ops.attention(_cuda(qkv), n, l, h, False)
    assert torch.equal(got, base)
    assert torch.equal(got_nt, base)
    ref = _attn_ref(qkv[: 2 * l], 2 if n >= 2 else 1, l, h, False) if n >= 2 else _attn_ref(qkv, n, l, h, False)
    err = (got[: ref.shape[0]].float().cpu() - ref).abs().max().item()
    assert err < 4e-3, f"max err {err}"


@pytest.mark.parametrize("n,l,h", [(1, 577, 4), (2, 257, 16), (3, 225, 2), (1, 256, 1), (2, 300, 3), (1, 384, 2), (1, 512, 1), (2, 545, 2), (1, 576, 2),
                                   (1, 640, 1), (1, 1025, 1), (70, 257, 16), (40, 577, 16), (1, 352, 2), (1, 448, 1), (2, 480, 2), (1, 2560, 1), (2, 416, 2), (1, 2816, 1),
                                   (300, 577, 1)])
def test_attention_ring_kernel(ops, clipmi_option, n, l, h):
    """Non-causal attention over more than 224 tokens (ViT-L/14: 257, ViT-L/14@336: 577; clip/model.py:181-183): the ring kernel -- 128-key blocks
    through a three-slot LDS ring, query tiles in passes of eleven, a short last pass split over the waves by key tile and merged -- against the fp32
    reference and against the round-1 streaming kernel (attn_ring 0; another summation order: not the same bits).  Lengths that end a pass exactly
    (352: 11 tiles, 1025: 33 with one live row in the last), leave one row for the last tile (257, 577), split the last pass four ways (384: 1 tile;
    416: 2 tiles, two merge rounds) or two ways (448: 3 tiles; 480: 4 tiles and 512: 5 tiles, two rounds), leave waves idle (300, 545, 640), the
    longest sequences the kernel takes (2560, 2816 = 8 passes); (70, 257, 16), (40, 577, 16) and (300, 577, 1) give every workgroup several items
    (ring and Q prefetch across item seams)."""
    g = torch.Generator().manual_seed(n * 1000 + l + h)
    qkv = (torch.randn(n * l, 3 * 64 * h, generator=g) * 1.5).half()
    dq = _cuda(qkv)
    got = ops.attention(dq, n, l, h, False)
    again = ops.attention(dq, n, l, h, False)
    clipmi_option("attn_ring", 0)
    base = ops.attention(dq, n, l, h, False)
    assert torch.equal(got, again), "not deterministic"
    assert torch.isfinite(got.float()).all()
    # another summation order, then one fp16 rounding of the output: one unit in the last place (2^-10 relative) apart at most, plus the absolute floor
    d = ((got.float() - base.float()).abs() - base.float().abs() * 2.0 ** -10).max().item()
    assert d < 2e-3, f"ring vs streaming kernel: {d}"
    ns = min(n, 2)
    ref = _attn_ref(qkv[: ns * l], ns, l, h, False)
    for name, x in (("ring", got), ("stream", base)):
        err = (x[: ns * l].float().cpu() - ref).abs().max().item()
        assert err < 4e-3, f"{name}: max err {err}"
    if n > 2:      # and the LAST sequence (the tail of the persistent walk)
        ref = _attn_ref(qkv[(n - 1) * l:], 1, l, h, False)
        err = (got[(n - 1) * l:].float().cpu() - ref).abs().max().item()
        assert err < 4e-3, f"last sequence: max err {err}"


@pytest.mark.parametrize("n,l,h,causal", [(500, 24, 8, True), (1000, 16, 8, True), (3, 32, 2, False), (7, 31, 3, True), (5, 1, 2, False), (2, 9, 1, True),
                                          (260, 8, 12, False), (4100, 24, 1, True)])
def test_attention_small_kernel(ops, clipmi_option, n, l, h, causal):
    """Attention over at most 32 tokens -- the text tower after dead-row elimination (CoOp prompts: 24 rows, zero-shot templates: 16; causal mask,
    clip/model.py:585-591) --: one (sequence, head) item per WAVE, no workgroup barrier (attn_small 1, the default), against the persistent kernel
    (attn_small 0: the same arithmetic in the same order -> the same bits) and the fp32 reference.  Item counts from a fraction of one workgroup to
    several items per wave (the wave's LDS tiles are re-used: the reads of one item must be back before the next item's DMA)."""
    g = torch.Generator().manual_seed(n * 1000 + l + h)
    qkv = (torch.randn(n * l, 3 * 64 * h, generator=g) * 1.5).half()
    dq = _cuda(qkv)
    got = ops.attention(dq, n, l, h, causal)
    again = ops.attention(dq, n, l, h, causal)
    clipmi_option("attn_small", 0)
    base = ops.attention(dq, n, l, h, causal)
    assert torch.equal(got, again)
    assert torch.equal(got, base)
    ns = min(n, 3)
    ref = _attn_ref(qkv[: ns * l], ns, l, h, causal)
    assert (got[: ns * l].float().cpu() - ref).abs().max().item() < 4e-3
    ref = _attn_ref(qkv[(n - 1) * l:], 1, l, h, causal)
    assert (got[(n - 1) * l:].float().cpu() - ref).abs().max().item() < 4e-3


def test_attention_ring_peaked_rows(ops):
    """A dominant key in the LAST block and large queries: the running maximum moves late (rescale across blocks, and across the merged partials)."""
    n, l, h = 1, 577, 2
    g = torch.Generator().manual_seed(7)
    qkv = torch.randn(n * l, 3 * 64 * h, generator=g)
    qkv[:, :128] *= 5.0
    qkv[570, 128:256] *= 5.0
    qkv[3, 128:256] *= 4.0
    qkv = qkv.half()
    ref = _attn_ref(qkv, n, l, h, False)
    got = ops.attention(_cuda(qkv), n, l, h, False).float().cpu()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() < 8e-3


def test_attention_peaked_rows(ops):
    """softmax with a dominant key (forces large score ranges through the online rescale across key groups)."""
    n, l, h = 1, 197, 2
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(n * l, 3 * 64 * h, generator=g)
    qkv[:, :128] *= 6.0   # large queries -> peaked distributions
    qkv[150, 128:256] *= 5.0  # one dominant key in the last key group
    qkv = qkv.half()
    ref = _attn_ref(qkv, n, l, h, False)
    got = ops.attention(_cuda(qkv), n, l, h, False).float().cpu()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() < 8e-3


@pytest.mark.parametrize("B,R,P,dt", [(2, 224, 16, torch.float32), (3, 64, 16, torch.float16), (2, 28, 14, torch.float32),
                                      (1, 336, 14, torch.float32), (2, 64, 32, torch.float32)])
def test_patchify(ops, B, R, P, dt):
    g = torch.Generator().manual_seed(R)
    img = torch.randn(B, 3, R, R, generator=g).to(dt)
    col = ops.patchify(_cuda(img), P).cpu()
    gsz = R // P
    ref = img.float().reshape(B, 3, gsz, P, gsz, P).permute(0, 2, 4, 1, 3, 5).reshape(B * gsz * gsz, 3 * P * P).half()
    k = 3 * P * P
    assert col.shape == (B * gsz * gsz, (k + 63) // 64 * 64)
    assert torch.equal(col[:, :k], ref)
    assert (col[:, k:] == 0).all()


@pytest.mark.parametrize("B,R,P,D,n_ctx,dt_in,dt_out", [(2, 224, 16, 768, 0, torch.float32, torch.float16), (3, 64, 16, 128, 2, torch.float16, torch.float16),
                                                        (5, 64, 32, 64, 0, torch.float32, torch.float32), (1, 32, 8, 72, 0, torch.float32, torch.float16),
                                                        (7, 96, 16, 200, 3, torch.float16, torch.float32), (40, 224, 16, 768, 0, torch.float32, torch.float16)])
def test_patch_embed_vs_conv2d(B, R, P, D, n_ctx, dt_in, dt_out):
    """clipmi_patch_embed (conv1 as a GEMM whose loader reads the NCHW image itself, + positional embedding, token-row scatter; clip/model.py:395-401)
    against F.conv2d in fp32 on the same fp16-rounded pixels and weights.  Ragged M (patch count not a multiple of the 256-row tile),
    ragged N, patch sizes 8 / 16 / 32, fp16 and fp32 images and outputs; the class row and the prompt rows of every sequence are not touched."""
    import torch.nn.functional as F
    from clip_calibration_amd._lib import F16, F32, check, lib
    g = torch.Generator().manual_seed(B * 131 + R + P)
    G = R // P
    L = 1 + G * G + n_ctx
    image = torch.randn(B, 3, R, R, generator=g).to(dt_in)
    w = (torch.randn(D, 3, P, P, generator=g) * (3 * P * P) ** -0.5).half()
    pos = torch.randn(1 + G * G, D, generator=g) * 0.3
    x0 = torch.full((B * L, D), float("nan"), dtype=dt_out, device="cuda")
    wd = w.reshape(D, 3 * P * P).contiguous().cuda()
    idt = F32 if dt_in == torch.float32 else F16
    scratch = torch.empty(max(1, lib.clipmi_patch_embed_scratch_bytes(B, R, idt)), dtype=torch.uint8, device="cuda")
    assert (scratch.numel() > 1) == (dt_in == torch.float32)
    img_d, pos_d = image.cuda(), pos.cuda()
    check(lib.clipmi_patch_embed(img_d.data_ptr(), idt, scratch.data_ptr() if dt_in == torch.float32 else None, wd.data_ptr(), 3 * P * P, pos_d.data_ptr(),
                                 x0.data_ptr(), F32 if dt_out == torch.float32 else F16, B, R, P, D, L, torch.cuda.current_stream().cuda_stream),
          "clipmi_patch_embed")
    ref = F.conv2d(image.half().float(), w.float(), stride=P).reshape(B, D, G * G).permute(0, 2, 1) + pos[1:]
    got = x0.float().cpu().reshape(B, L, D)
    if B == 2:                                            # pos = NULL: the bare conv output (what the image tower asks for)
        x1 = torch.full_like(x0, float("nan"))
        check(lib.clipmi_patch_embed(img_d.data_ptr(), idt, scratch.data_ptr() if dt_in == torch.float32 else None, wd.data_ptr(), 3 * P * P, None,
                                     x1.data_ptr(), F32 if dt_out == torch.float32 else F16, B, R, P, D, L, torch.cuda.current_stream().cuda_stream),
              "clipmi_patch_embed")
        bare = x1.float().cpu().reshape(B, L, D)[:, 1:1 + G * G]
        assert (bare - (ref - pos[1:])).abs().max().item() < 2e-3 * max(1.0, ref.abs().max().item())
    assert torch.isnan(got[:, 0]).all() and (n_ctx == 0 or torch.isnan(got[:, 1 + G * G:]).all())       # class / prompt rows untouched
    err = (got[:, 1:1 + G * G] - ref).abs().max().item()
    assert err < (2e-3 if dt_out == torch.float16 else 2e-5) * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("B,L0,n_ctx,D,dt", [(3, 197, 0, 768, torch.float16), (2, 17, 2, 128, torch.float32), (5, 50, 4, 1024, torch.float16),
                                            (1, 5, 0, 1280, torch.float32)])
@pytest.mark.parametrize("outs", ["y", "y16", "both", "both+pos"])
def test_embed_ln_vs_layer_norm(B, L0, n_ctx, D, dt, outs):
    """clipmi_embed_ln: cat(class_embedding) + pos[0], the patch rows as the GEMM left them, MaPLe's shallow prompt rows, then ln_pre
    (clip/model.py:398-402,413,459-460) against F.layer_norm in fp32; the fold row sums are those of the output."""
    import torch.nn.functional as F
    from clip_calibration_amd._lib import F16, F32, check, lib
    g = torch.Generator().manual_seed(B * 7 + D)
    L = L0 + n_ctx
    x0 = (torch.randn(B * L, D, generator=g) * 2 + 0.5).to(dt)
    cls, pos = torch.randn(D, generator=g), torch.randn(L0, D, generator=g) * 0.2
    shallow = torch.randn(max(n_ctx, 1), D, generator=g)
    gamma, beta = torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g) * 0.1
    add_pos = outs.endswith("+pos")                       # the image tower's form: the GEMM left the bare conv output, pos[l] is added here
    outs = outs.split("+")[0]
    y = torch.full((B * L, D), float("nan"), device="cuda") if outs != "y16" else None
    y16 = torch.full((B * L, D), float("nan"), dtype=torch.float16, device="cuda") if outs != "y" else None
    stats = torch.full((B * L, 2), float("nan"), device="cuda") if outs != "y" else None
    dev = [t.cuda() for t in (x0, cls, pos, shallow, gamma, beta)]
    ptr = lambda t: None if t is None else t.data_ptr()
    check(lib.clipmi_embed_ln(dev[0].data_ptr(), F32 if dt == torch.float32 else F16, int(add_pos), dev[1].data_ptr(), dev[2].data_ptr(),
                              dev[3].data_ptr() if n_ctx else None, dev[4].data_ptr(), dev[5].data_ptr(), ptr(y), ptr(y16), ptr(stats), B, L, L0, D,
                              1e-5, torch.cuda.current_stream().cuda_stream), "clipmi_embed_ln")
    rows = x0.float().reshape(B, L, D).clone()
    if add_pos:
        rows[:, 1:L0] += pos[1:]
    rows[:, 0] = cls + pos[0]
    if n_ctx:
        rows[:, L0:] = shallow[:n_ctx]
    want = F.layer_norm(rows, (D,), gamma, beta, 1e-5).reshape(B * L, D)
    if y is not None:
        assert (y.cpu() - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())
    if y16 is not None:
        assert (y16.float().cpu() - want).abs().max().item() < 2e-3 * max(1.0, want.abs().max().item())
        st = stats.cpu()
        assert torch.allclose(st[:, 0], want.sum(1), rtol=1e-4, atol=1e-2) and torch.allclose(st[:, 1], (want * want).sum(1), rtol=1e-4, atol=1e-2)


def test_l2_and_logits_dac_conf_pred(ops):
    g = load_golden("dac_cases.npz")
    rng = np.random.default_rng(1)
    for B, C, E in [(64, 50, 128), (256, 1000, 512), (17, 37, 64), (5, 3, 16)]:
        img = torch.from_numpy(rng.normal(size=(B, E)).astype(np.float32))
        txt = torch.from_numpy(rng.normal(size=(C, E)).astype(np.float32))
        dacc = torch.from_numpy(rng.uniform(0.5, 1.5, size=C).astype(np.float32))
        img_n = ops.l2_normalize(_cuda(img))
        txt_n = ops.l2_normalize(_cuda(txt.half()))     # fp16 input path
        np.testing.assert_allclose(img_n.cpu().numpy(), orc.l2_normalize(img).numpy(), rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(txt_n.cpu().numpy(), orc.l2_normalize(txt.half().float()).numpy(), rtol=1e-6, atol=1e-7)
        ref_logits, _, _ = orc.clip_logits(img, txt.half().float(), 100.0)
        # plain logits
        lg, conf, pred = ops.logits_fused(img_n, txt_n, 100.0)
        np.testing.assert_allclose(lg.cpu().numpy(), ref_logits.numpy(), atol=2e-4)   # |logit| <= 100, fp32 dot of 512
        probs = orc.softmax_probs(lg.cpu().numpy().astype(np.float64))
        c_ref, p_ref = orc.conf_pred(probs)
        assert np.array_equal(pred.cpu().numpy(), p_ref)
        np.testing.assert_allclose(conf.cpu().numpy(), c_ref, rtol=2e-5)
        # DAC fused
        lg2, conf2, pred2 = ops.logits_fused(img_n, txt_n, 100.0, _cuda(dacc))
        ref2 = orc.dac_predict(lg.cpu().numpy(), dacc.numpy())
        np.testing.assert_allclose(lg2.cpu().numpy(), ref2, rtol=1e-6, atol=1e-6)
        c2, p2 = orc.conf_pred(orc.softmax_probs(ref2.astype(np.float64)))
        assert np.array_equal(pred2.cpu().numpy(), p2)
        np.testing.assert_allclose(conf2.cpu().numpy(), c2, rtol=2e-5)
    # the reference's own DAC outputs (golden): numpy in / numpy out predict()
    from clip_calibration_amd.dac import DistanseAwareCalibration
    for n in ("c50", "c19", "k3"):
        cal = DistanseAwareCalibration()
        cal.fit(g[f"{n}:base_zs"], g[f"{n}:cur_zs"], g[f"{n}:base_tuned"], g[f"{n}:cur_tuned"], int(g[f"{n}:k"]))
        np.testing.assert_allclose(cal.class_confidence, g[f"{n}:class_confidence"], rtol=1e-13)
        out = cal.predict(g[f"{n}:logits"])
        assert out.dtype == np.float32
        np.testing.assert_allclose(out, g[f"{n}:scaled_logits"], rtol=2e-7, atol=0)


@pytest.mark.parametrize("B,C,E", [(256, 1000, 512), (1, 1, 64), (16, 64, 64), (37, 1003, 512), (300, 50, 768), (2048, 1000, 512),
                                   (128, 397, 1024), (5, 3, 16), (1024, 199, 512), (64, 4200, 512)])
@pytest.mark.parametrize("dac", [False, True])
def test_fused_tail_bitwise_vs_unfused(ops, clipmi_option, B, C, E, dac):
    """clipmi_fused_tail (one launch: normalise + logits + DAC + softmax top-1 + ECE bins; zsclip.py:99-101,
    distanse_aware_calibration.py:49-58) against the same arithmetic as separate launches: bit-identical outputs, and the
    ticket counters are left zero.  Without DAC the row pass is BLOCKWISE in both forms (per 64-column block max / argmax / sum of
    exponentials, merged in ascending block order): the fused kernel forms it from one 16-byte partial per (row, column block) that
    each of its workgroups writes, the separate launch from the logits.  (5, 3, 16) is a shape the fused kernel does not take: both
    sides run the launches; (1024, 199, 512) is the gathered batch of BASELINE configs[3] against SUN397's base-class count (199 is
    not a multiple of 4: plain stores and the reference row pass inside the kernel); (64, 4200, 512): 66 column blocks.)"""
    rng = np.random.default_rng(B * 31 + C + E)
    img = _cuda(torch.from_numpy(rng.normal(size=(B, E)).astype(np.float32)) * 3.0)
    txt_n = ops.l2_normalize(_cuda(torch.from_numpy(rng.normal(size=(C, E)).astype(np.float32))))
    dacc = _cuda(torch.from_numpy(rng.uniform(0.5, 1.5, size=C).astype(np.float32))) if dac else None
    labels = _cuda(torch.from_numpy(rng.integers(0, C, size=B)))
    outs = []
    for unfused in (1, 0):
        clipmi_option("tail_unfused", unfused)
        bins = torch.zeros(3 * 11, dtype=torch.float64, device="cuda")
        lg, img_n, conf, pred = ops.fused_tail(img, txt_n, 100.0, dacc, True, True, labels, bins, 10)
        outs.append((lg, img_n, conf, pred, bins))
    (lg_u, in_u, cf_u, pr_u, bins_u), (lg_f, in_f, cf_f, pr_f, bins_f) = outs
    assert torch.equal(in_u, ops.l2_normalize(img)) and torch.equal(in_f, in_u)
    assert torch.equal(lg_f, lg_u) and torch.equal(pr_f, pr_u)
    assert torch.equal(cf_f, cf_u)                            # the row pass of the separate launches, verbatim
    from clip_calibration_amd.metrics import bin_statistics
    own = bin_statistics(cf_f.cpu().numpy(), pr_f.cpu().numpy(), labels.cpu().numpy(), 10)     # the fused bins against its own (conf, pred)
    got = bins_f.cpu().numpy().reshape(3, 11)
    assert np.array_equal(got[[0, 2]], own[[0, 2]])
    np.testing.assert_allclose(got[1], own[1], rtol=1e-13)
    assert torch.equal(bins_f[:11], bins_u[:11]) and torch.equal(bins_f[22:], bins_u[22:])      # counts, hits: exact
    np.testing.assert_allclose(bins_f[11:22].cpu().numpy(), bins_u[11:22].cpu().numpy(), rtol=1e-13)
    ws = ops._TAIL_WS[(torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)]
    assert int(ws[: 64 * 1024].view(torch.int32).abs().sum()) == 0               # the whole ticket-counter region
    # against the oracle (fp32 dot products of E terms, |logit| <= 100)
    ref, _, _ = orc.clip_logits(img.cpu(), txt_n.cpu(), 100.0)
    ref = ref.numpy()
    if dac:
        ref = orc.dac_predict(ref, dacc.cpu().numpy())
    np.testing.assert_allclose(lg_f.cpu().numpy(), ref, atol=3e-4, rtol=2e-6)
    # pre-normalised entry (the multi-GPU path after the all-gather): same logits from the normalised features
    lg_p, cf_p, pr_p = ops.logits_fused(in_f, txt_n, 100.0, dacc)
    assert torch.equal(lg_p, lg_f) and torch.equal(cf_p, cf_f) and torch.equal(pr_p, pr_f)      # same kernel, same partials


def test_fused_tail_handoff_under_load(ops):
    """The row pass runs in whichever workgroup of a 16-row block draws the last ticket and reads logits that workgroups
    on other CUs / XCDs stored a moment ago (agent-scope release / acquire).  Hammer it: 300 back-to-back launches on
    changing inputs with the consumer's lines warm (the previous launch's logits sit in the caches), each checked in
    full against the separate-launch path."""
    from clip_calibration_amd import _lib as L
    rng = np.random.default_rng(7)
    B, C, E = 2048, 1000, 512
    txt_n = ops.l2_normalize(_cuda(torch.from_numpy(rng.normal(size=(C, E)).astype(np.float32))))
    dacc = _cuda(torch.from_numpy(rng.uniform(0.5, 1.5, size=C).astype(np.float32)))
    base = _cuda(torch.from_numpy(rng.normal(size=(B, E)).astype(np.float32)))
    filler = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    bad = 0
    for it in range(300):
        img = torch.roll(base, shifts=it, dims=0) * (1.0 + 0.01 * it)
        L.set_option("tail_unfused", 0)
        if it % 3 == 0:
            filler.add_(1)                      # uneven load: an HBM-bound stream ahead of the launch
        lg, _, conf, pred = ops.fused_tail(img, txt_n, 100.0, dacc if it % 2 else None)
        L.set_option("tail_unfused", 1)
        lg_u, _, conf_u, pred_u = ops.fused_tail(img, txt_n, 100.0, dacc if it % 2 else None)
        L.set_option("tail_unfused", 0)
        bad += int(not (torch.equal(lg, lg_u) and torch.equal(pred, pred_u) and torch.equal(conf, conf_u)))
    assert bad == 0, f"{bad} of 300 launches differ from the separate-launch path"


def test_ece_device_accumulation(ops):
    from clip_calibration_amd.evaluator import DeviceCalibrationEvaluator
    from clip_calibration_amd.metrics import ECE
    g = load_golden("ece_cases.npz")
    for n in sorted({k.split(":")[0] for k in g}):
        conf, pred, gt, bins = g[f"{n}:conf"], g[f"{n}:pred"], g[f"{n}:gt"], int(g[f"{n}:bins"])
        conf32 = conf.astype(np.float32)
        want = orc.ece(conf32, pred, gt, bins)       # the device sees float32 confidences
        ev = DeviceCalibrationEvaluator(bins)
        half = len(conf) // 2                         # two calls: accumulation across batches
        for sl in (slice(0, half), slice(half, None)):
            if conf32[sl].size:
                ev.process(torch.from_numpy(conf32[sl]).cuda(), torch.from_numpy(pred[sl].astype(np.int32)).cuda(),
                           torch.from_numpy(gt[sl].astype(np.int64)).cuda())
        res = ev.evaluate()
        assert res["ece"] / 100.0 == pytest.approx(want, abs=1e-6), n
        assert res["total"] == len(conf)
        assert res["mce"] / 100.0 == pytest.approx(orc.mce(conf32, pred, gt, bins), abs=1e-6), n
        assert ECE(conf32, pred, gt, bins) == pytest.approx(want, abs=1e-6)


def test_knn_proximity(ops):
    """SURVEY f-2: kNN distances vs fixtures produced by the reference's own proximity.py, then a larger oracle case."""
    from clip_calibration_amd import proximity as prox
    g = load_golden("knn_cases.npz")
    for n in sorted({k.split(":")[0] for k in g}):
        k = int(g[f"{n}:k"])
        got = prox.get_knn_dists(g[f"{n}:refs"], g[f"{n}:queries"], k)
        assert got.shape == g[f"{n}:knn"].shape and got.dtype == np.float32
        np.testing.assert_allclose(got, g[f"{n}:knn"], rtol=3e-6, atol=3e-7)
        kv = g[f"{n}:val_knn"].shape[1]
        np.testing.assert_allclose(prox.get_val_image_knn_dists(g[f"{n}:refs"], kv), g[f"{n}:val_knn"], rtol=3e-6, atol=3e-7)
    rng = np.random.default_rng(5)
    refs = rng.normal(size=(1000, 512)).astype(np.float32)
    q = rng.normal(size=(203, 512)).astype(np.float32)
    np.testing.assert_allclose(prox.get_knn_dists(refs, q, 10), orc.knn_dists(refs, q, 10), rtol=3e-6)
    np.testing.assert_allclose(prox.proximity_from_knn(orc.knn_dists(refs, q, 5)), np.exp(-orc.knn_dists(refs, q, 5).mean(1)))


@pytest.mark.parametrize("B,C,dac", [(1, 1, False), (5, 37, True), (64, 1000, True), (7, 1000, False), (0, 10, True)])
def test_softmax_rows_vs_oracle(ops, B, C, dac):
    """clipmi_softmax_rows = VLCalibration.predict on the DAC / plain branches (vl_calibrator.py:83-109)."""
    rng = np.random.default_rng(B * 1000 + C)
    logits = (rng.normal(size=(B, C)) * 5).astype(np.float32)
    cc = rng.uniform(0.5, 1.5, C).astype(np.float32) if dac else None
    lg = torch.from_numpy(logits).cuda()
    before = lg.clone()
    probs, conf, pred = ops.softmax_rows(lg, None if cc is None else torch.from_numpy(cc).cuda(), want_conf_pred=True)
    assert torch.equal(lg, before)                                  # input untouched
    if B == 0:
        assert probs.shape == (0, C)
        return
    scaled = orc.dac_predict(logits.astype(np.float64), cc.astype(np.float64)) if dac else logits
    want = orc.softmax_probs(np.asarray(scaled, dtype=np.float64))
    wc, wp = orc.conf_pred(want)
    assert np.abs(probs.cpu().numpy() - want).max() < 1e-5
    assert np.array_equal(pred.cpu().numpy(), logits.argmax(1))     # DAC scales by a positive factor: argmax of the raw row
    assert np.abs(conf.cpu().numpy() - wc).max() < 1e-5
    # aliasing probs onto logits is allowed by the header
    from clip_calibration_amd._lib import check, lib
    check(lib.clipmi_softmax_rows(lg.data_ptr(), None if cc is None else torch.from_numpy(cc).cuda().data_ptr(), lg.data_ptr(),
                                  None, None, B, C, torch.cuda.current_stream().cuda_stream), "alias")
    assert np.abs(lg.cpu().numpy() - want).max() < 1e-5


def test_softmax_rows_masked_class_blocks(ops):
    """Caller-supplied logits with whole 64-column blocks at -inf (masked classes): the blockwise denominator must treat such a block
    as contributing nothing (exp(-inf - M) = 0, as torch.softmax does) instead of exp(-inf + inf) = NaN.  Rows: one masked block in the
    middle, a masked first block, a masked ragged last block, everything but one column masked."""
    C = 200
    rng = np.random.default_rng(5)
    logits = (rng.normal(size=(4, C)) * 3).astype(np.float32)
    logits[0, 64:128] = -np.inf
    logits[1, 0:64] = -np.inf
    logits[2, 192:200] = -np.inf
    logits[3, :] = -np.inf
    logits[3, 131] = 0.5
    probs, conf, pred = ops.softmax_rows(torch.from_numpy(logits).cuda(), None, want_conf_pred=True)
    want = torch.softmax(torch.from_numpy(logits).double(), dim=1).numpy()
    got = probs.cpu().numpy()
    assert np.isfinite(got).all() and np.isfinite(conf.cpu().numpy()).all()
    assert np.abs(got - want).max() < 1e-6
    assert np.array_equal(pred.cpu().numpy(), logits.argmax(1)) and pred[3].item() == 131 and abs(conf[3].item() - 1.0) < 1e-6


@pytest.mark.parametrize("B,H,W,C,Cout,relu", [(2, 8, 8, 64, 64, 1), (3, 7, 5, 128, 136, 1), (1, 14, 14, 256, 256, 0), (5, 3, 9, 64, 8, 1),
                                               (0, 4, 4, 64, 64, 1)])
def test_conv3x3_implicit_gemm(B, H, W, C, Cout, relu):
    """clipmi_conv3x3_nhwc (Bottleneck.conv2 + folded bn2 + ReLU, clip/model.py:20,46): implicit GEMM with hardware zero
    padding against a plain fp32 torch convolution of the same fp16-rounded operands."""
    import torch.nn.functional as F
    from clip_calibration_amd._lib import check, lib
    g = torch.Generator().manual_seed(B * 1000 + H * 100 + C)
    x = torch.randn(B, H, W, C, generator=g).half()
    w = (torch.randn(Cout, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).half()
    bias = torch.randn(Cout, generator=g) * 0.1
    xd = x.cuda().contiguous()
    wd = w.permute(0, 2, 3, 1).reshape(Cout, 9 * C).contiguous().cuda()      # tap-major: (ky*3+kx)*C + c
    bd = bias.cuda()
    out = torch.full((B * H * W, Cout), float("nan"), dtype=torch.float16, device="cuda")
    check(lib.clipmi_conv3x3_nhwc(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr(), B, H, W, C, Cout, relu,
                                  torch.cuda.current_stream().cuda_stream), "clipmi_conv3x3_nhwc")
    if B == 0:
        return
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), bias, 1, 1)
    if relu:
        ref = torch.relu(ref)
    ref = ref.permute(0, 2, 3, 1).reshape(B * H * W, Cout)
    got = out.float().cpu()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max() < 2e-3 * max(1.0, float(ref.abs().max()))
