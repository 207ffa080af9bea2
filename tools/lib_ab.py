#!/usr/bin/env python3
"""Times one attention shape under several BUILDS of the library (CLIPMI_LIBRARY is read at import: one child process per build), interleaved rounds.
    python tools/lib_ab.py libclipmi.so libclipmi_x.so ...        (names relative to clip_calibration_amd/csrc; SHAPE=n,l,h; CAUSAL=0/1)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "clip_calibration_amd", "csrc")
CHILD = r'''
import os, sys, torch
sys.path.insert(0, %r)
from clip_calibration_amd import ops
n, l, h = [int(x) for x in os.environ.get("SHAPE", "64,577,16").split(",")]
causal = os.environ.get("CAUSAL", "0") == "1"
qkv = torch.randn(n * l, 3 * 64 * h, device="cuda").half()
for _ in range(5):
    ops.attention(qkv, n, l, h, causal)
ts = []
for rnd in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.attention(qkv, n, l, h, causal)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 50)
print(f"{sorted(ts)[2]:.1f}")
''' % ROOT
libs = sys.argv[1:] or ["libclipmi.so"]
res = {l: [] for l in libs}
for rnd in range(int(os.environ.get("ROUNDS", "3"))):
    for lib in libs:
        out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, CLIPMI_LIBRARY=os.path.join(CSRC, lib)), capture_output=True, text=True)
        try:
            res[lib].append(float(out.stdout.split()[0]))
        except (IndexError, ValueError):
            print(lib, "failed:", out.stderr[-500:])
for lib, v in res.items():
    print(f"{lib:28s} median-of-5 us per launch, per round: " + " ".join(f"{x:6.1f}" for x in v), flush=True)
