#!/usr/bin/env python3
"""Interleaved A/B of BUILDS of the library on the image tower (round 6): one child process per build and round (CLIPMI_LIBRARY is read at import),
per round the tower's time (events on the launch stream) and the five per-layer kernels as the tower launches them (clipmi_profile_block).
    python tools/lib_tower_ab.py libclipmi_prev.so libclipmi.so        (names relative to clip_calibration_amd/csrc; GEOM, B, ROUNDS)
tools/build_prev_lib.sh builds libclipmi_prev.so from the sources of a git revision."""
import json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "clip_calibration_amd", "csrc")
CHILD = r'''
import json, os, sys, torch
sys.path.insert(0, %r)
from clip_calibration_amd import _lib, synthetic as syn
from clip_calibration_amd.model import build_model
G = os.environ.get("GEOM", "ViT-B/16"); B = int(os.environ.get("B", "256"))
_lib.set_option("cls_only_last_block", int(os.environ.get("CLS_ONLY", "0")))   # like with like: libraries before round 6 default to every row
for kv in filter(None, os.environ.get("OPTIONS", "").split(",")):              # e.g. OPTIONS=attn_loader=1
    _lib.set_option(kv.split("=")[0], int(kv.split("=")[1]))
model = build_model(dict(syn.synthetic_state_dict(G, seed=0)), None).cuda()
images = syn.synthetic_images(B, G, seed=0, device="cuda")
with torch.no_grad():
    for _ in range(3):
        f = model.image_features_f32(images)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            f = model.image_features_f32(images)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 4)
    ms = model.profile_block_ms(B, iters=10)
print(json.dumps({"tower_ms": sorted(ts)[2], "kernels_us": {k: v * 1e3 for k, v in ms.items()}, "checksum": float(f.double().abs().sum())}))
''' % ROOT
libs = sys.argv[1:] or ["libclipmi.so"]
res = {l: [] for l in libs}
for rnd in range(int(os.environ.get("ROUNDS", "4"))):
    for lib in libs:
        out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, CLIPMI_LIBRARY=os.path.join(CSRC, lib)), capture_output=True, text=True)
        try:
            res[lib].append(json.loads(out.stdout.strip().splitlines()[-1]))
        except (IndexError, ValueError):
            print(lib, "failed:", out.stderr[-800:], flush=True)
for lib, rs in res.items():
    if not rs:
        continue
    ks = rs[0]["kernels_us"].keys()
    print(f"{lib:26s} tower ms per round: " + " ".join(f"{r['tower_ms']:.3f}" for r in rs) + f" | median {statistics.median(r['tower_ms'] for r in rs):.3f} | " +
          " | ".join(f"{k} {statistics.median(r['kernels_us'][k] for r in rs):6.1f}" for k in ks) + f" | feature checksum {rs[0]['checksum']:.6f}", flush=True)
