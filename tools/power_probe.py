#!/usr/bin/env python3
"""Diagnostic: is the image tower power-limited?  Runs one GEMM configuration (or the whole tower) back to back for a few seconds
while sampling `rocm-smi` (power, sclk) from a side thread, on random and on all-zero operands."""
import os, re, subprocess, sys, threading, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import _lib, ops

def sample(stop, out):
    while not stop.is_set():
        try:
            t = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True, timeout=5).stdout
            pw = re.findall(r"Power \(W\):\s*([\d.]+)", t) or re.findall(r"Socket Power.*?:\s*([\d.]+)", t)
            sc = re.findall(r"sclk clock level:.*?\((\d+)Mhz\)", t)
            out.append((time.time(), pw[:1], sc[:1]))
        except Exception as e:
            out.append((time.time(), str(e)[:60], None))
        time.sleep(0.15)

M, N, K = 256 * 197, 3072, 768
for fill in ("randn", "zeros"):
    a = (torch.randn(M, K, device="cuda") if fill == "randn" else torch.zeros(M, K, device="cuda")).half()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5 if fill == "randn" else torch.zeros(N, K, device="cuda")).half()
    bias = torch.randn(N, device="cuda") * 0.1
    out = torch.empty(M, N, dtype=torch.float16, device="cuda")
    stop, samples = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, samples)); th.start()
    torch.cuda.synchronize(); t0 = time.time(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < 4.0:
        for _ in range(50):
            ops.gemm_f16(a, w, bias, None, _lib.EPI_BIAS_QUICKGELU, torch.float16, out=out)
        n += 50
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    stop.set(); th.join()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"{fill}: {us:.1f} us per launch ({2.0*M*N*K/us/1e6:.0f} TF); rocm-smi samples (power W, sclk MHz):", [(s[1], s[2]) for s in samples[3:-1:3]][:8], flush=True)
print(subprocess.run(["rocm-smi", "--showmaxpower"], capture_output=True, text=True).stdout[-400:])
