#!/usr/bin/env python3
"""How many wait states does the VALU-write -> MFMA-source sequence need on gfx950?  (profiles/r03_gpu_sharing.txt; common.h
CLIPMI_VALU_TO_MFMA_FENCE).  One row of the table per BUILD of the library: `make -C clip_calibration_amd/csrc fence_sweep` compiles
attention.hip and logits.hip with the fence weakened (libclipmi_fence{n,0,1,2,3}.so: no fence, `s_nop 0` .. `s_nop 3`); this script is started
once per build, by the shell, with CLIPMI_LIBRARY pointing at it:

    for f in n 0 1 2 3; do CLIPMI_LIBRARY=$PWD/clip_calibration_amd/csrc/libclipmi_fence$f.so python tools/probes/hazard_fence_sweep.py $f; done

and counts, for each of three "hammer" kernels looping on a second stream, how many of N launches of the LayerNorm kernel (one wave per row,
56 registers, no LDS, no MFMA: it fits beside the hammer's waves on a SIMD) return a wrong row.  The distance column is the smallest number
of wait states tools/mfma_hazard_scan.py finds between such a write and its MFMA in that build (hipcc itself guarantees two)."""
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from clip_calibration_amd import ops  # noqa: E402

DIST = {"n": 2, "0": 3, "1": 4, "2": 5, "3": 6}     # WAIT=8 EXTRA=-DCLIPMI_FENCE_SNOP=k tools/mfma_hazard_scan.py attention.hip logits.hip
HAMMERS = {"vision attention (197 tokens) + fused tail": (24, 197, 12, False, True), "text attention (77 tokens, causal)": (500, 77, 8, True, False),
           "ViT-L/14 attention (257 tokens)": (16, 257, 16, False, False)}


def run(name, n_launch):
    n, l, h, causal, with_tail = HAMMERS[name]
    g = torch.Generator().manual_seed(0)
    M, K = 197 * 256, 768
    x = torch.randn(M, K, generator=g).cuda()
    gam, bet = torch.ones(K).cuda(), torch.zeros(K).cuda()
    truth = torch.nn.functional.layer_norm(x.double(), (K,)).float()
    qkv = torch.randn(n * l, 3 * 64 * h, generator=g).half().cuda()
    feat = torch.randn(256, 512, generator=g).cuda()
    txt = ops.l2_normalize(torch.randn(1000, 512, generator=g).cuda())
    stop, errors = threading.Event(), []

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
        except Exception as e:
            errors.append(e)
    t = threading.Thread(target=hammer)
    t.start()
    bad_launches = bad_rows = 0
    try:
        for _ in range(n_launch):
            out = ops.layernorm(x, gam, bet)
            wrong = int(((out - truth).abs().amax(dim=1) > 1e-3).sum())
            bad_launches += int(wrong > 0)
            bad_rows += wrong
    finally:
        stop.set()
        t.join()
    if errors:
        raise errors[0]
    return bad_launches, bad_rows


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "?"
    n_launch = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    fence = {"n": "no fence", "?": "product build"}.get(tag, f"s_nop {tag}")
    cells = []
    for name in HAMMERS:
        bl, br = run(name, n_launch)
        cells.append(f"{bl:3d} of {n_launch} launches ({br} rows)")
    print(f"fence {fence:13s} | smallest write -> MFMA distance {DIST.get(tag, '?')} | " + " | ".join(f"{n.split(' (')[0]}: {c}" for n, c in zip(HAMMERS, cells)), flush=True)


if __name__ == "__main__":
    main()
