#!/usr/bin/env python3
"""One-off sweep behind bench.py's CPU_BASELINE_THREADS: the CPU oracle (the timed `cpu_baseline` of the bench line, BASELINE configs[0]'s
shape: 32 images x 100 prompts, ViT-B/16, fp32) at several torch thread counts on the GPU box's host cores.  BASELINE.md section 4: two
warm-ups, median of the timed iterations.  No GPU is touched.

    python tools/cpu_baseline_threads.py [--threads 8 16 32 64] > profiles/r05_cpu_baseline_threads.txt
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from clip_calibration_amd import synthetic as syn  # noqa: E402
from oracle import clip_oracle as orc  # noqa: E402  (the timed baseline itself)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, nargs="+", default=[8, 16, 32, 64])
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--classes", type=int, default=100)
    ap.add_argument("--one", type=int, default=0, help="(internal) measure this thread count in THIS process and print one line")
    a = ap.parse_args()
    avail = len(os.sched_getaffinity(0))
    if not a.one:
        # a fresh process per thread count: torch.set_num_threads() on a live OpenMP pool is not the same thing as starting with that count
        # (the first version of this sweep, one process, read 14.7 images/s at 32 threads where bench.py's own process measures 35)
        import subprocess
        print(f"# host: {os.cpu_count()} logical CPUs, {avail} in this process's affinity mask; torch {torch.__version__}; oracle zeroshot_inference + "
              f"calibrated_ece, batch {a.batch} x {a.classes} prompts, 2 warm-ups, median of {a.iters}; one fresh process per thread count", flush=True)
        best = (0.0, None)
        for n in a.threads:
            if n > avail:
                print(f"threads {n:4d}: skipped (only {avail} CPUs available)", flush=True)
                continue
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", str(n), "--iters", str(a.iters), "--batch", str(a.batch),
                                  "--classes", str(a.classes)], capture_output=True, text=True, env=dict(os.environ, OMP_NUM_THREADS=str(n))).stdout.strip()
            print(out, flush=True)
            try:
                best = max(best, (float(out.split(":")[1].split()[0]), n))
            except (IndexError, ValueError):
                pass
        print(f"# fastest: {best[1]} threads, {best[0]:.2f} images/s")
        return
    a.threads = [a.one]
    sd = syn.synthetic_state_dict("ViT-B/16", seed=0)
    ids = syn.synthetic_token_ids(a.classes, "ViT-B/16", seed=0)
    images = syn.synthetic_images(a.batch, "ViT-B/16", seed=0)
    best = (0.0, None)
    with torch.no_grad():
        torch.set_num_threads(a.one)
        txt = orc.l2_normalize(orc.encode_text(sd, ids))
        for n in a.threads:
            if n > avail:
                print(f"threads {n:4d}: skipped (only {avail} CPUs available)")
                continue
            torch.set_num_threads(n)
            times = []
            for it in range(2 + a.iters):
                t0 = time.perf_counter()
                logits, _, _ = orc.zeroshot_inference(sd, images, txt)
                labels = syn.synthetic_labels(logits.argmax(1), a.classes, seed=0)
                orc.calibrated_ece(logits.numpy(), labels.numpy())
                if it >= 2:
                    times.append(time.perf_counter() - t0)
            v = a.batch / float(np.median(times))
            best = max(best, (v, n))
            print(f"threads {n:4d}: {v:7.2f} images/s (median {np.median(times):.3f} s/batch; all: {' '.join(f'{t:.3f}' for t in times)})", flush=True)


if __name__ == "__main__":
    main()
