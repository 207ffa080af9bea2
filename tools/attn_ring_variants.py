#!/usr/bin/env python3
"""Times the build-time scheduling variants of the ring attention kernel (make ring_variants: libclipmi_ring{N}.so, attention.hip CLIPMI_RING_VARIANT)
against the product library: one child process per library (CLIPMI_LIBRARY is read at import), three interleaved rounds, ViT-L/14@336 per-rank shape."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "clip_calibration_amd", "csrc")
CHILD = r'''
import os, sys, torch
sys.path.insert(0, %r)
from clip_calibration_amd import ops
n, l, h = [int(x) for x in os.environ.get("SHAPE", "64,577,16").split(",")]
qkv = torch.randn(n * l, 3 * 64 * h, device="cuda").half()
for _ in range(5):
    ops.attention(qkv, n, l, h, False)
ts = []
for rnd in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.attention(qkv, n, l, h, False)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 50)
print(f"{sorted(ts)[2]:.1f} {min(ts):.1f}")
''' % ROOT
libs = ["libclipmi.so"] + sorted(f for f in os.listdir(CSRC) if f.startswith("libclipmi_ring") and f.endswith(".so"))
res = {l: [] for l in libs}
for rnd in range(3):
    for lib in libs:
        out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, CLIPMI_LIBRARY=os.path.join(CSRC, lib)), capture_output=True, text=True)
        try:
            res[lib].append(float(out.stdout.split()[0]))
        except (IndexError, ValueError):
            print(lib, "failed:", out.stderr[-500:])
for lib, v in res.items():
    if v:
        print(f"{lib:28s} median-of-5 us per launch, three rounds: " + " ".join(f"{x:6.1f}" for x in v), flush=True)
