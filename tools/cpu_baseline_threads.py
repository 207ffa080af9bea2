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
    a = ap.parse_args()
    avail = len(os.sched_getaffinity(0))
    print(f"# host: {os.cpu_count()} logical CPUs, {avail} in this process's affinity mask; torch {torch.__version__}; "
          f"oracle zeroshot_inference + calibrated_ece, batch {a.batch} x {a.classes} prompts, 2 warm-ups, median of {a.iters}")
    sd = syn.synthetic_state_dict("ViT-B/16", seed=0)
    ids = syn.synthetic_token_ids(a.classes, "ViT-B/16", seed=0)
    images = syn.synthetic_images(a.batch, "ViT-B/16", seed=0)
    best = (0.0, None)
    with torch.no_grad():
        torch.set_num_threads(min(avail, 32))
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
    print(f"# fastest: {best[1]} threads, {best[0]:.2f} images/s")


if __name__ == "__main__":
    main()
