#!/usr/bin/env python3
"""Diagnostic (tuning build): where an item of the vision attention kernel spends its time.
query wave: 0 arrive at the barrier, 1 past it, 4 S of key tiles 0-3 done, 5 their softmax + P.V done, 6 S of key tiles 4-6 done, 2 their
softmax + P.V done, 3 outputs stored;  loader (wave 7): 0 arrive at its wait, 1 operands landed (vmcnt 0), 2 past the barrier, 3 next
item's DMA issued.  CLIPMI_LIBRARY = libclipmi_tuning.so or one of the `make attn_ablate` builds (tools/attn_ablate.sh)."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import _lib, ops
assert hasattr(_lib.lib, "clipmi_tuning_set_stamps"), "needs CLIPMI_LIBRARY=.../libclipmi_tuning.so"
_lib.lib.clipmi_tuning_set_stamps.argtypes = [ctypes.c_void_p]
n, l, h = 256, 197, 12
qkv = torch.randn(n * l, 3 * 64 * h, device="cuda").half()
for mode in [int(x) for x in os.environ.get("MODES", "2").split(",")]:
    _lib.set_option("attn_loader", mode)
    for _ in range(3):
        ops.attention(qkv, n, l, h, False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.attention(qkv, n, l, h, False)
    e1.record()
    torch.cuda.synchronize()
    print(f"{os.path.basename(os.environ.get('CLIPMI_LIBRARY', 'libclipmi.so'))}: {e0.elapsed_time(e1) * 20:.1f} us per launch (50 launches)")
    stamps = torch.zeros(n * h * 8 * 8, dtype=torch.int64, device="cuda")
    _lib.lib.clipmi_tuning_set_stamps(stamps.data_ptr())
    ops.attention(qkv, n, l, h, False)
    torch.cuda.synchronize()
    _lib.lib.clipmi_tuning_set_stamps(None)
    s = stamps.cpu().numpy().reshape(n * h, 8, 8).astype(np.float64) / 100.0   # us; [item][wave 0..6 query, 7 loader][stamp]
    ok = s[:, 0, 0] > 0
    s = s[ok]
    print(f"attn_loader={mode}: kernel span {s[:, :7, 3].max() - s[:, :7, 0].min():.1f} us, {ok.sum()} items")
    for w in range(7):
        t = s[:, w, :][:, [0, 1, 4, 5, 6, 2, 3]]
        if (t[:, 2] == 0).all():
            continue
        d = np.diff(t, axis=1).mean(axis=0)
        print(f"   query wave {w}: wait at barrier {d[0]:5.2f} | S(0-3) {d[1]:5.2f}  softmax+PV(0-3) {d[2]:5.2f}  S(4-6) {d[3]:5.2f}  softmax+PV(4-6) {d[4]:5.2f} | store {d[5]:5.2f} us")
    arrive = s[:, :, 0]                                    # arrival at the item's barrier, all 8 waves
    last = np.argmax(arrive, axis=1)
    print("   last wave to arrive at the barrier (share of items, waves 0..7):", np.round(np.bincount(last, minlength=8) / len(last), 2))
    e = np.diff(s[:, 7, :4], axis=1)
    print(f"   loader per item: wait for landing {np.median(e[:,0]):5.2f}  wait at barrier {np.median(e[:,1]):5.2f}  issue next {np.median(e[:,2]):5.2f} us")
