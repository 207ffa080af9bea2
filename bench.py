#!/usr/bin/env python3
"""Headline benchmark: images/sec, ViT-B/16 224px zero-shot + ECE on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its N rank processes itself, as children)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W               (the same ranks under an external launcher)

One step = one pass of the hot path over one synthetic batch that is already resident in HBM:
  N = 1:  image tower (HIP)  ->  ONE fused tail launch: L2 normalise, scale * img @ txt^T, softmax top-1 (conf, pred),
          ECE bin accumulation (clipmi_fused_tail).
  N > 1:  image tower -> L2 normalise to fp16 -> ONE RCCL all-gather of the per-GPU embeddings over xGMI
          (clipmi_allgather) -> the same fused tail on the gathered batch (every rank).
Workload (default, the headline) = BASELINE configs[1]: ImageNet-1k zero-shot shape (1000 prompts), batch 256 per GPU,
seeded random ViT-B/16 weights (no checkpoints offline).  Text features are computed once before the timed region
(zsclip.py:90-92) and their time is reported separately.  `--workload coop_dac` = BASELINE configs[2] (second,
non-headline line): CoOp 16-shot prompts (n_ctx 16, 500 classes), DAC per-class factors fitted on base / new text
features, TempScaling scalar; reported with the text features cached and with the text tower re-run every batch (the
reference's behaviour, trainers/classification/coop.py:208-210).
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_F16_DENSE_PEAK_TFLOPS = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense bf16/fp16
HBM_PEAK_BYTES_PER_S = 8.0e12        # same guide: HBM3E 8 TB/s spec (6.3 TB/s measured achievable)
MFMA_PEAK_CLOCK_MHZ = 2400.0         # the clock the 2.5 PF figure is quoted at (same guide, peaks table)
CPU_BASELINE_THREADS = 16            # fastest of 8 / 16 / 32 / 64 / 128 on the GPU box host in two sweeps (profiles/r05_cpu_baseline_threads.txt)
CPU_BASELINE_WARMUPS = 2


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=("zeroshot", "coop_dac", "level1", "stream"), default="zeroshot",
                    help="zeroshot = the headline (BASELINE configs[1]); coop_dac = configs[2]; level1 = the reference caller's own statements "
                         "with only build_model swapped (INTEGRATION level 1); stream = host fp32 batches through runner.device_batches")
    ap.add_argument("--sustained-seconds", type=float, default=3.2, help="length of the second, sustained timed loop (0 = skip)")
    ap.add_argument("--batch", type=int, default=256, help="images per GPU per step")
    ap.add_argument("--classes", type=int, default=None, help="default: 1000 (zeroshot) / 500 (coop_dac)")
    ap.add_argument("--model", default="ViT-B/16")
    ap.add_argument("--exchange-f16", action="store_true",
                    help="N = 1 only: pass the embeddings through the fp16 exchange format of the N > 1 path (no gather), "
                         "so that a 1-process run is bitwise comparable with any N-process run on the same images")
    ap.add_argument("--images-seed", type=int, default=None, help="seed of the per-rank image shard (default: rank)")
    ap.add_argument("--virtual-ranks", type=int, default=1,
                    help="N = 1 only: process the image shards of this many ranks in ONE process (batch = ranks x --batch); with "
                         "--exchange-f16 the outputs equal rank 0's of the real N-process run bit for bit (self-test aid)")
    ap.add_argument("--dump", default=None, help="rank 0: save logits / conf / pred / ECE bins of the last step to this .npz")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-kernel profile (functional multi-rank self-tests)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the per-GPU shapes of BASELINE configs[2] / [3] / [4] (other_configs)")
    ap.add_argument("--cpu-batch", type=int, default=32)
    ap.add_argument("--cpu-classes", type=int, default=100, help="BASELINE configs[0]: Caltech101-sized prompt set")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    return ap.parse_args()


def cpu_baseline(sd, geom_name, n_cls, batch, budget_s):
    """The oracle (CPU restatement of the reference path, fp32) on BASELINE config[0]'s shape: image tower -> normalise ->
    logits -> softmax -> ECE.  Returns (images/s, cores, cores available, sample text, logits, images, labels)."""
    from clip_calibration_amd import synthetic as syn
    from oracle import clip_oracle as orc  # timed baseline + checker only

    # torch CPU kernels collapse when oversubscribed (256 logical CPUs on the GPU box -> 0.5 img/s): take the CPUs this process may
    # actually run on, capped at the thread count that was FASTEST in a sweep on the GPU box's host (tools/cpu_baseline_threads.py ->
    # profiles/r05_cpu_baseline_threads.txt; BENCH_CPU_THREADS overrides), and report both counts.
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, int(os.environ.get("BENCH_CPU_THREADS", CPU_BASELINE_THREADS))))
    torch.set_num_threads(cores)
    ids = syn.synthetic_token_ids(n_cls, geom_name, seed=0)
    images = syn.synthetic_images(batch, geom_name, seed=0)
    with torch.no_grad():
        t0 = time.perf_counter()
        txt = orc.l2_normalize(orc.encode_text(sd, ids))
        t_text = time.perf_counter() - t0
        times, logits = [], None
        t_start = time.perf_counter()
        it = 0
        while True:
            t0 = time.perf_counter()
            logits, _, _ = orc.zeroshot_inference(sd, images, txt)
            labels = syn.synthetic_labels(logits.argmax(1), n_cls, seed=0)
            orc.calibrated_ece(logits.numpy(), labels.numpy())
            dt = time.perf_counter() - t0
            it += 1
            if it > CPU_BASELINE_WARMUPS:  # two warm-ups, then the median of up to 5 timed iterations
                times.append(dt)
            if (time.perf_counter() - t_start > budget_s and len(times) >= 1) or len(times) >= 5:
                break
    med = float(np.median(times))
    sample = (f"{len(times)} timed iterations ({CPU_BASELINE_WARMUPS} warm-ups) of batch {batch} x {n_cls} prompts, fp32 oracle on {cores} host "
              f"threads, median {med:.3f} s/batch; text tower once {t_text:.2f} s (excluded)")
    return batch / med, cores, avail, sample, logits, images, labels


def timed_ms(fn, iters):
    """Mean milliseconds per call, torch events on the current stream (the stream every launch of the path goes to)."""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def kernel_roofline(model, syn, geom, model_name, B, images):
    """Per-kernel roofline measured live: hipEvents on the launch stream around every launch (clipmi_profile_block), the
    five per-layer kernels exactly as the tower launches them, on the activations the last step left in the workspace."""
    L, D = geom.vision_tokens, geom.vision_width
    M = B * L
    blk_ms = model.profile_block_ms(B, iters=24)
    blk_flop = {"in_proj": 2.0 * M * 3 * D * D, "attention": 4.0 * B * L * L * D, "out_proj": 2.0 * M * D * D,
                "c_fc": 2.0 * M * 4 * D * D, "c_proj": 2.0 * M * 4 * D * D}
    blk_shape = {"in_proj": {"M": M, "N": 3 * D, "K": D}, "attention": {"sequences": B, "tokens": L, "heads": D // 64},
                 "out_proj": {"M": M, "N": D, "K": D}, "c_fc": {"M": M, "N": 4 * D, "K": D}, "c_proj": {"M": M, "N": D, "K": 4 * D}}
    kernels = []
    for name in model.BLOCK_KERNELS:
        tf = blk_flop[name] / (blk_ms[name] * 1e-3) / 1e12
        kernels.append({"kernel": name, "shape": blk_shape[name], "flop_per_launch": blk_flop[name], "us_per_launch": 1e3 * blk_ms[name],
                        "achieved": tf, "frac": tf / MFMA_F16_DENSE_PEAK_TFLOPS})
        if name == "attention":
            # the kernel's own bound is its bytes, not either pipe (profiles/r03_attention_ablation.txt): q | k | v in once, rows out once, fp16
            byts = 2.0 * M * (3 * D + D)
            gbps = byts / (blk_ms[name] * 1e-3) / 1e9
            kernels[-1]["hbm"] = {"bound": "hbm", "algorithmic_bytes": byts, "achieved": gbps, "peak": (HBM_PEAK_BYTES_PER_S / 1e9), "unit": "GB/s",
                                  "frac": gbps / (HBM_PEAK_BYTES_PER_S / 1e9)}
    dom = max(kernels, key=lambda k: k["us_per_launch"])
    with torch.no_grad():
        tower_ms = timed_ms(lambda: model.image_features_f32(images), 10)
        # the same five kernels timed IN PLACE: hipEvents behind every launch of real tower passes (clipmi_encode_image_timed), each kernel
        # behind its real predecessor and on operands that predecessor has just written.  Mean over all layers of 3 passes (1 warm-up).
        model.image_tower_launch_us(images)
        passes = [model.image_tower_launch_us(images) for _ in range(3)]
    for i, k in enumerate(kernels):
        us = float(np.mean([blk[i] for p in passes for blk in p["blocks"]]))
        k["us_in_tower"] = us
        k["achieved_in_tower"] = k["flop_per_launch"] / (us * 1e-6) / 1e12
        k["frac_in_tower"] = k["achieved_in_tower"] / MFMA_F16_DENSE_PEAK_TFLOPS
    in_tower = {"embed_us": [float(np.mean([p["embed"][j] for p in passes])) for j in range(len(passes[0]["embed"]))],
                "post_us": [float(np.mean([p["post"][j] for p in passes])) for j in range(2)],
                "per_layer_us": float(np.mean([sum(blk) for p in passes for blk in p["blocks"]])),
                "sum_ms": float(np.mean([p["total_us"] for p in passes])) * 1e-3,
                "what": "one hipEvent behind every launch of a real pass: an interval = the kernel + the launch gap in front of it, so the "
                        "intervals add up to the pass (sum_ms; compare tower.ms, the same pass without events)"}
    tower_flop = syn.flops_per_image(model_name) * B
    tower_tf = tower_flop / (tower_ms * 1e-3) / 1e12
    # HBM-side bytes per launch of the dominant kernel come from PMC passes (rocprofv3 cannot run inside this process):
    # tools/measure_traffic.sh writes profiles/gemm_traffic.json stamped with the sha256 of the kernel source it measured;
    # reported only when shape AND source still match, otherwise null.
    traffic, note = None, "no PMC profile for this kernel source (tools/measure_traffic.sh regenerates it)"
    try:
        with open(os.path.join(ROOT, "profiles", "gemm_traffic.json")) as f:
            prof = json.load(f)
        h = hashlib.sha256()
        for name in ("gemm_common.h", "gemm.hip", "gemm_rstream.hip"):   # the GEMM kernels' sources (tools/traffic_json.py hashes the same)
            with open(os.path.join(ROOT, "clip_calibration_amd", "csrc", name), "rb") as f:
                h.update(f.read())
        sha = h.hexdigest()
        ent = prof.get("kernels", {}).get(dom["kernel"])
        if prof.get("gemm_hip_sha256") == sha and ent and ent["shape"] == dom["shape"]:
            traffic = ent["traffic_bytes"]
            note = ("HBM-side bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KB, rocprofv3 --pmc in separate passes with the gfx950 "
                    "FETCH_SIZE x2 correction (profiles/gemm_traffic.json, measured on these GEMM sources); algorithmic bytes = "
                    f"{ent.get('algorithmic_bytes')}")
        else:
            note = "profiles/gemm_traffic.json was measured on different GEMM sources: dropped"
    except (OSError, KeyError, ValueError):
        pass
    return {"bound": "mfma", "kernel": dom["kernel"] + " (dominant launch: one per layer, 12 per step)", "shape": dom["shape"],
            "flop_per_launch": dom["flop_per_launch"], "avg_launch_ms": dom["us_per_launch"] * 1e-3, "achieved": dom["achieved"],
            "peak": MFMA_F16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": dom["frac"], "traffic": traffic, "traffic_note": note,
            "kernels": kernels, "in_tower": in_tower,
            "tower": {"ms": tower_ms, "flop": tower_flop, "achieved": tower_tf, "frac": tower_tf / MFMA_F16_DENSE_PEAK_TFLOPS},
            "tower_frac": tower_tf / MFMA_F16_DENSE_PEAK_TFLOPS}


def ceilings(dev, local_rank, geom, B, seconds=1.0):
    """Two reference points for the fractions above, measured in THIS run on THIS box (neither is on the product path):
    mfma_only   -- clipmi_probe_mfma_f16: a register-only v_mfma_f32_16x16x32_f16 loop (no LDS, no memory) on N(0, 0.25^2) fp16 operands,
                   8 waves on every CU, back to back for `seconds` with the sysfs power sampler running: what the matrix pipe alone sustains
                   at the package power cap on toggling data;
    vendor_gemm -- torch.matmul (hipBLASLt) on the c_fc and c_proj shapes, no bias, no activation, no residual: a comparison only."""
    from clip_calibration_amd._lib import check, lib
    import ctypes as C
    waves, iters = 8, 8000
    n_cus = C.c_int(0)
    ops_ = (torch.randn(16, waves * 64, 8, device=dev) * 0.25).half().contiguous()
    sink = torch.empty(1024 * waves * 64, dtype=torch.float32, device=dev)
    clk = torch.zeros(2, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def launch():
        check(lib.clipmi_probe_mfma_f16(ops_.data_ptr(), sink.data_ptr(), clk.data_ptr(), waves, iters, C.byref(n_cus), st), "clipmi_probe_mfma_f16")
    one_ms = timed_ms(launch, 3)
    n = max(4, int(seconds * 1e3 / one_ms))
    sampler = PowerSampler(local_rank)
    sampler.start()
    ms = timed_ms(launch, n)
    power = sampler.stop()
    c = clk.cpu().numpy()
    flop = 2.0 * 2 * 64 * 64 * 32 * iters * waves * n_cus.value
    out = {"mfma_only": {"achieved": flop / (ms * 1e-3) / 1e12, "unit": "TFLOP/s", "frac_of_peak": flop / (ms * 1e-3) / 1e12 / MFMA_F16_DENSE_PEAK_TFLOPS,
                         "seconds": n * ms * 1e-3, "in_kernel_clock_mhz": float(c[0]) / (float(c[1]) / 100.0) if c[1] else None,
                         "power_w": power.get("avg_w"), "power_cap_w": power.get("cap_w"), "sclk_mhz_sysfs": power.get("sclk_mhz_avg"),
                         "what": f"v_mfma_f32_16x16x32_f16 from registers only, N(0, 0.25^2) fp16 operands, {waves} waves x {n_cus.value} CUs "
                                 "(clipmi_probe_mfma_f16)"}}
    L, D = geom.vision_tokens, geom.vision_width
    M = B * L
    vend = {}
    for name, (N_, K_) in (("c_fc", (4 * D, D)), ("c_proj", (D, 4 * D))):
        x = torch.randn(M, K_, device=dev).half()
        w = (torch.randn(N_, K_, device=dev) * 0.03).half()
        us = 1e3 * timed_ms(lambda: torch.matmul(x, w.t()), 20)
        vend[name] = {"us": us, "achieved": 2.0 * M * N_ * K_ / (us * 1e-6) / 1e12, "shape": {"M": M, "N": N_, "K": K_}}
        del x, w
    out["vendor_gemm"] = dict(vend, what="torch.matmul (hipBLASLt) fp16, randn activations, N(0, 0.03^2) weights, NO epilogue (no bias / QuickGELU / "
                                         "residual / LayerNorm fold): comparison only, never on the product path")
    return out


def environment():
    """Where this line was measured: kernel driver / VBIOS / ROCm versions from sysfs and the files of the image, the library's ABI and the
    VALU -> MFMA fence it was built with (common.h CLIPMI_FENCE_SNOP), so that a recurrence of the round-3 corruption can be tied to a box
    (profiles/r04_hazard_repro.txt).  Files only: nothing is exec'ed from this GPU-initialised process."""
    import glob
    import platform

    def read(path):
        try:
            with open(path) as f:
                return f.read().strip()
        except OSError:
            return None
    env = {"kernel": platform.release(), "amdgpu_driver": read("/sys/module/amdgpu/version"), "rocm": read("/opt/rocm/.info/version"),
           "torch": torch.__version__, "hip_runtime": getattr(torch.version, "hip", None)}
    cards = sorted(glob.glob("/sys/class/drm/card*/device/vbios_version"))
    env["vbios"] = sorted({v for v in (read(c) for c in cards) if v})
    try:
        pr = torch.cuda.get_device_properties(0)
        env["device"] = {"name": pr.name, "gcn_arch": getattr(pr, "gcnArchName", None), "cus": pr.multi_processor_count,
                         "hbm_gib": round(pr.total_memory / 2 ** 30, 1)}
    except Exception:
        pass
    try:
        from clip_calibration_amd import _lib
        env["libclipmi"] = {"abi": _lib.lib.clipmi_abi_version(), "sha256_16": hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()[:16]}
        src = open(os.path.join(ROOT, "clip_calibration_amd", "csrc", "common.h")).read()
        import re
        m = re.search(r"#\s*define\s+CLIPMI_FENCE_SNOP\s+(-?\d+)", src)
        env["libclipmi"]["valu_to_mfma_fence"] = f"s_nop {m.group(1)}" if m else "see common.h"
    except Exception as e:   # never let bookkeeping fail a measurement
        env["libclipmi"] = {"error": repr(e)}
    return env


def other_configs(dev, syn, model, build_model, images, Cn):
    """The other BASELINE configs at their per-GPU shapes, OUTSIDE `value` (a few seconds): the driver's record then carries a number for each.
      configs[3]  11-dataset sweep, batch 1024 over 8 GPUs   -> ViT-B/16 image tower at 128 images per GPU
      configs[4]  ViT-L/14@336px, batch 512 over 8 GPUs     -> ViT-L/14@336 image tower at 64 images per GPU
      configs[2]  CoOp 16-shot + DAC, reference schedule    -> both towers + tail per batch (text tower re-run every batch, coop.py:208-210)
    Image towers: torch events around 6 calls after 2 warm-ups; fraction = algorithmic flop / time / 2.5 PF."""
    from clip_calibration_amd.trainers import CoOpCLIP
    from clip_calibration_amd import ops
    out = {}

    from clip_calibration_amd import _lib

    def tower(m, name, x, iters=6):
        m.image_features_f32(x)
        ms = timed_ms(lambda: m.image_features_f32(x), iters)
        tf = syn.flops_per_image(name) * x.shape[0] / (ms * 1e-3) / 1e12
        with _lib.option("cls_only_last_block", 1):   # the product default (every number without the suffix: every row of every block, as `value`)
            m.image_features_f32(x)
            ms_c = timed_ms(lambda: m.image_features_f32(x), iters)
        return {"model": name, "batch_per_gpu": x.shape[0], "tower_ms": ms, "images_per_s_tower": x.shape[0] / (ms * 1e-3), "tower_tflops": tf,
                "tower_frac": tf / MFMA_F16_DENSE_PEAK_TFLOPS, "tower_ms_cls_only": ms_c, "images_per_s_tower_cls_only": x.shape[0] / (ms_c * 1e-3)}
    with torch.no_grad():
        out["configs[3] per-rank"] = tower(model, "ViT-B/16", images[:128].contiguous())
        # configs[2]: 500 classes, n_ctx 16, DAC factors (any positive vector times the row: the fit is host work outside the loop)
        ids = syn.synthetic_token_ids(500, "ViT-B/16", seed=11, n_ctx_placeholders=16)
        coop = CoOpCLIP(model, ids, n_ctx=16, logit_scale=1.0, seed=3, cache_text_features=False)
        dac = (1.0 + 0.1 * torch.rand(500, generator=torch.Generator().manual_seed(5))).to(dev)
        scale = float(np.exp(4.6052))

        def coop_step():
            feats, txt = coop.towers(images)
            return ops.fused_tail(feats, txt, scale, dac, True, True)
        coop_step()
        ms = timed_ms(coop_step, 6)
        with _lib.option("cls_only_last_block", 1):
            coop_step()
            ms_c = timed_ms(coop_step, 6)
        rows = model.live_rows(coop.tokenized_prompts)
        t_ms = timed_ms(lambda: coop.text_features(), 6)
        out["configs[2] per-batch schedule"] = {
            "model": "ViT-B/16", "batch_per_gpu": images.shape[0], "classes": 500, "n_ctx": 16, "ms_per_batch": ms,
            "images_per_s_text_recomputed_every_batch": images.shape[0] / (ms * 1e-3),
            "ms_per_batch_cls_only": ms_c, "images_per_s_text_recomputed_every_batch_cls_only": images.shape[0] / (ms_c * 1e-3),
            "text_rows_computed": rows, "text_rows_of_context": model.context_length, "text_tower_ms": t_ms,
            "text_tower_tflops_on_computed_rows": syn.flops_per_prompt("ViT-B/16") * 500 * rows / model.context_length / (t_ms * 1e-3) / 1e12,
            "what": "prompt learner + text tower (fp16 stream, side stream) + image tower + fused tail with DAC row scale, every batch; flop credited "
                    "for the computed token rows only (linear terms; the attention term is < 3 % of a prompt)"}
        del coop
        big = "ViT-L/14@336px"
        m2 = build_model(dict(syn.synthetic_state_dict(big, seed=0)), {"trainer": "ZeroshotCLIP"}).to(dev)
        x2 = syn.synthetic_images(64, big, seed=0, device=dev)
        out["configs[4] per-rank"] = tower(m2, big, x2, iters=4)
        del m2, x2
    torch.cuda.empty_cache()
    return out


class PowerSampler:
    """Package power, power cap and shader clock of the benchmarked GPU while a loop runs, read from sysfs hwmon files by a Python
    thread (files only: nothing is exec'ed from this GPU-initialised process).  amdgpu exposes power1_average / power1_input and
    power1_cap in microwatts and freq1_input (sclk) in Hz under /sys/class/drm/card*/device/hwmon/hwmon*/."""

    def __init__(self, device_index: int, period_s: float = 0.05):
        import glob
        import threading
        self.period, self._stop, self._thread = period_s, threading.Event(), None
        self.samples = {}   # hwmon dir -> [(power_w, sclk_mhz)]
        self.dirs, self.note = [], ""
        want = None
        try:
            pr = torch.cuda.get_device_properties(device_index)
            want = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        except Exception:   # older torch: no PCI ids on the properties object
            pass
        for h in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            if any(os.path.exists(os.path.join(h, f)) for f in ("power1_average", "power1_input")):
                self.dirs.append(h)
        if want:
            hit = [h for h in self.dirs if os.path.realpath(os.path.join(h, "..", "..")).endswith(want)]
            if hit:
                self.dirs, self.note = hit, f"pci {want}"
        if not self.dirs:
            self.note = "no readable amdgpu hwmon power file on this box"

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().split()[0])
        except (OSError, ValueError, IndexError):
            return None

    def _loop(self):
        while not self._stop.is_set():
            for h in self.dirs:
                pw = self._read(os.path.join(h, "power1_average"))
                if pw is None:
                    pw = self._read(os.path.join(h, "power1_input"))
                fr = self._read(os.path.join(h, "freq1_input"))
                self.samples.setdefault(h, []).append((None if pw is None else pw * 1e-6, None if fr is None else fr * 1e-6))
            self._stop.wait(self.period)

    def start(self):
        import threading
        if self.dirs:
            self._thread = threading.Thread(target=self._loop, daemon=True)
            self._thread.start()

    def stop(self):
        self._stop.set()
        if self._thread is not None:
            self._thread.join()
        if not self.samples:
            return {"avg_w": None, "cap_w": None, "sclk_mhz_avg": None, "note": self.note or "no samples"}
        best, best_w = None, -1.0
        for h, sm in self.samples.items():   # several cards visible and no PCI match: the one that drew the most power ran the loop
            ws = [a for a, _ in sm if a is not None]
            if ws and sum(ws) / len(ws) > best_w:
                best, best_w = h, sum(ws) / len(ws)
        if best is None:
            return {"avg_w": None, "cap_w": None, "sclk_mhz_avg": None, "note": "power files unreadable"}
        sm = self.samples[best][1:] or self.samples[best]   # the first sample predates the loop
        ws = [a for a, _ in sm if a is not None]
        fs = [b for _, b in sm if b is not None]
        cap = self._read(os.path.join(best, "power1_cap"))
        return {"avg_w": sum(ws) / len(ws) if ws else None, "max_w": max(ws) if ws else None, "cap_w": None if cap is None else cap * 1e-6,
                "sclk_mhz_avg": sum(fs) / len(fs) if fs else None, "sclk_mhz_min": min(fs) if fs else None, "samples": len(sm),
                "source": best + "/{power1_average|power1_input, freq1_input, power1_cap}" + (f" ({self.note})" if self.note else ""),
                "note": "sampled every 50 ms during the sustained loop; sysfs sclk reads up to ~10 % above the in-kernel clock of MFMA-dense loops "
                        "(MI355X_MICROARCH.md, DVFS give-back 6)"}


def level1_workload(args, dev, syn, model, sd):
    """INTEGRATION level 1: ONLY clip.build_model is swapped; every statement the reference's zero-shot trainer and evaluator execute per
    batch runs as the reference wrote it (trainers/classification/zsclip.py:97-102, trainers/classification/base_learner.py:84-88,
    evaluators/vl_evaluator.py:40-51): fp16 `encode_image` output, normalise and `logit_scale * img @ txt.t()` as torch ops, then four
    `.data.cpu().numpy().tolist()` per batch.  Reported beside the Level-2 mirror (the headline) to show what an unchanged caller gets."""
    B, Cn = args.batch, args.classes or 1000
    ids = syn.synthetic_token_ids(Cn, args.model, seed=0).to(dev)
    images = syn.synthetic_images(B, args.model, seed=0, device=dev)
    labels = torch.zeros(B, dtype=torch.int64, device=dev)
    with torch.no_grad():
        text_features = model.encode_text(ids)                                          # zsclip.py:90-92
        text_features = text_features / text_features.norm(dim=-1, keepdim=True)

        def model_inference(image):                                                     # zsclip.py:97-102, statement for statement
            image_features = model.encode_image(image)
            image_features = image_features / image_features.norm(dim=-1, keepdim=True)
            logit_scale = model.logit_scale.exp()
            logits = logit_scale * image_features @ text_features.t()
            return logits, image_features, text_features

        y_score, y_true, tf_l, if_l = [], [], [], []

        def process(mo, gt, image_features, text_feats):                               # vl_evaluator.py:40-51
            y_score.extend(mo.data.cpu().numpy().tolist())
            y_true.extend(gt.data.cpu().numpy().tolist())
            tf_l.extend(text_feats.data.cpu().numpy().tolist())
            if_l.extend(image_features.data.cpu().numpy().tolist())

        def loop(n, with_evaluator):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                out, imf, txf = model_inference(images)
                if with_evaluator:
                    process(out, labels, imf, txf)
                    for lst in (y_score, y_true, tf_l, if_l):
                        lst.clear()
            torch.cuda.synchronize()
            return time.perf_counter() - t0
        loop(args.warmup, False)
        t_inf = loop(args.steps, False)
        loop(1, True)
        n_ev = max(2, min(args.steps, 5))
        t_ev = loop(n_ev, True)
    return {"metric": "images/sec ViT-B/16 zero-shot, reference caller unchanged (INTEGRATION level 1; not the headline metric)",
            "value": B * args.steps / t_inf, "unit": "images/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * t_inf / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16",
            "data": "synthetic",
            "config": {"workload": f"reference zsclip.model_inference statements on the swapped build_model, {args.model}, batch {B}, {Cn} prompts", "batch_per_gpu": B, "classes": Cn},
            "level1": {"model_inference_images_per_s": B * args.steps / t_inf, "model_inference_ms": 1e3 * t_inf / args.steps,
                       "with_reference_evaluator_process_images_per_s": B * n_ev / t_ev, "with_reference_evaluator_process_ms": 1e3 * t_ev / n_ev,
                       "what": "model_inference: HIP image tower -> fp16 features -> torch normalise + matmul (fp16, 3 small torch kernels); "
                               "evaluator.process: four .data.cpu().numpy().tolist() per batch (logits [B,C], labels, text [C,E], image [B,E]) as "
                               "evaluators/vl_evaluator.py:40-51 does -- host-bound; the Level-2 mirror keeps all of it on the device"}}


def stream_workload(args, dev, syn, model):
    """Host fp32 batches -> pinned staging -> side-stream H2D (runner.device_batches), a batch ahead -> ZeroshotCLIP.model_inference with the
    device evaluator: the test loop of base_learner.py:84-88 with its input leg included (PCIe), against the resident-batch headline."""
    from clip_calibration_amd.evaluator import DeviceCalibrationEvaluator
    from clip_calibration_amd.runner import device_batches
    from clip_calibration_amd.trainers import ZeroshotCLIP
    B, Cn = args.batch, args.classes or 1000
    zs = ZeroshotCLIP(model, syn.synthetic_token_ids(Cn, args.model, seed=0))
    host = [(syn.synthetic_images(B, args.model, seed=k), torch.randint(0, Cn, (B,), generator=torch.Generator().manual_seed(k))) for k in range(4)]
    pinned = [(im.pin_memory(), lb.pin_memory()) for im, lb in host]      # what a DataLoader(pin_memory=True) hands over
    ev = DeviceCalibrationEvaluator(10, dev)

    def loader(src, n):
        for i in range(n):
            yield src[i % len(src)]

    def loop(n, mode):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode == "pinned":
            it = device_batches(loader(pinned, n), dev)
        elif mode == "pageable_prefetched":
            it = device_batches(loader(host, n), dev)
        else:                                                              # what parse_batch_test does: pageable .to() on the compute stream
            it = ((im.to(dev), lb.to(dev)) for im, lb in loader(host, n))
        for image, label in it:
            zs.model_inference(image, want_conf_pred=True, labels=label, evaluator=ev)
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    with torch.no_grad():
        loop(args.warmup, "pinned")
        t_staged = loop(args.steps, "pinned")
        loop(2, "pageable_prefetched")
        t_pp = loop(args.steps, "pageable_prefetched")
        loop(2, "reference")
        t_page = loop(args.steps, "reference")
        resident = host[0][0].to(dev), host[0][1].to(dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            zs.model_inference(resident[0], want_conf_pred=True, labels=resident[1], evaluator=ev)
        torch.cuda.synchronize()
        t_res = time.perf_counter() - t0
    bytes_per_batch = host[0][0].numel() * 4
    return {"metric": "images/sec ViT-B/16 zero-shot + ECE, inputs streamed from host memory (not the headline metric)",
            "value": B * args.steps / t_staged, "unit": "images/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * t_staged / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"pinned host fp32 batches of {B} (4 distinct, cycled) -> non_blocking H2D on a side stream, a batch ahead -> {args.model} zero-shot + ECE, {Cn} prompts",
                       "batch_per_gpu": B, "classes": Cn},
            "stream": {"pinned_prefetched_images_per_s": B * args.steps / t_staged, "pageable_prefetched_images_per_s": B * args.steps / t_pp,
                       "pageable_to_device_on_the_compute_stream_images_per_s": B * args.steps / t_page,
                       "resident_images_per_s": B * args.steps / t_res, "h2d_gbytes_per_s": bytes_per_batch * args.steps / t_staged / 1e9,
                       "bytes_per_image": bytes_per_batch // B, "pcie_gen5_x16_spec_gbytes_per_s": 63.0}}


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N rank processes OURSELVES, as children of this process, through
    `python -m torch.distributed.run` on this same script, and hand back the launcher's return code.  Called before anything here has
    touched the GPU (importing torch does not): a GPU-initialised process must not be the one that starts other programs on these boxes,
    and it is never an exec -- the parent stays, relays rank 0's JSON line (the children write to the inherited stdout) and exits with
    the child's code.  The reference's own multi-GPU mechanism is nn.DataParallel inside one process
    (trainers/classification/coop.py:268-272); one process per GPU is this repo's replacement for it."""
    import signal
    import subprocess
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    # The launcher asks for SIGTERM if this process dies, however it dies (PR_SET_PDEATHSIG = 1; torch.distributed.run answers SIGTERM by
    # stopping its workers).  It sets that itself, first thing in its own single-threaded start-up: a preexec_fn would run between fork and
    # exec of THIS process, which has imported torch and numpy and may own threads (subprocess documents that as unsafe).
    boot = ("import ctypes, runpy, signal, sys; ctypes.CDLL(None).prctl(1, signal.SIGTERM); sys.argv[0] = 'torch.distributed.run'; "
            "runpy.run_module('torch.distributed.run', run_name='__main__', alter_sys=True)")
    cmd = [sys.executable, "-c", boot, "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    # same process group as this process (no new session): a `kill -- -pgid` or a `timeout` aimed at bench.py reaches the ranks too
    child = subprocess.Popen(cmd, env=env)

    def forward(signum, _frame):
        child.send_signal(signal.SIGTERM)
    old = {s: signal.signal(s, forward) for s in (signal.SIGTERM, signal.SIGINT)}
    try:
        return child.wait()
    finally:
        for s, h in old.items():
            signal.signal(s, h)
        if child.poll() is None:
            child.kill()
            child.wait()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        # plain `python bench.py --gpus N`: this process becomes the launcher of its N ranks (nothing below this line runs here)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    if world > 1:
        print(f"[bench rank {rank} of {world}] started (local rank {local_rank})", file=sys.stderr, flush=True)
    assert torch.cuda.is_available(), f"bench.py needs a GPU (no CPU fallback) [rank {rank} of {world}]"
    # BENCH_SAME_GPU=1 is a functional self-test of the N>1 code path on a 1-GPU box: every rank uses cuda:0 and the
    # exchange goes through torch.distributed/gloo (RCCL refuses two ranks on one device).  Never a measurement.
    same_gpu = os.environ.get("BENCH_SAME_GPU") == "1"
    if same_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # the process group carries the RCCL unique id and the timing barrier (gloo, CPU tensors); its cuda:nccl half is only
        # touched if the library's own RCCL communicator cannot be built and the exchange falls back to torch.distributed
        dist.init_process_group("gloo" if same_gpu else "cpu:gloo,cuda:nccl", rank=rank, world_size=world)

    def in_turn(fn):
        """BENCH_SAME_GPU self-test only: the ranks share ONE device, so they take turns on it -- two processes interleaved on one GPU is not
        the configuration under test (one process per GPU is), and a bitwise comparison should not depend on it
        (profiles/r03_gpu_sharing.txt).  Everywhere else: just the call."""
        if not (same_gpu and world > 1):
            return fn()
        out = None
        for r in range(world):
            if r == rank:
                out = fn()
                torch.cuda.synchronize()
            dist.barrier()
        return out

    from clip_calibration_amd import ops, synthetic as syn
    from clip_calibration_amd.evaluator import DeviceCalibrationEvaluator
    from clip_calibration_amd.model import build_model
    from clip_calibration_amd.parallel import EmbeddingExchange
    from clip_calibration_amd.trainers import CoOpCLIP, ZeroshotCLIP

    if args.workload in ("level1", "stream"):
        if world != 1:
            raise SystemExit(f"--workload {args.workload} is a single-GPU line")
        sd_ = syn.synthetic_state_dict(args.model, seed=0)
        model_ = build_model(dict(sd_), {"trainer": "ZeroshotCLIP"}).to(dev)
        print(json.dumps(level1_workload(args, dev, syn, model_, sd_) if args.workload == "level1" else stream_workload(args, dev, syn, model_)))
        return
    geom = syn.GEOMETRIES[args.model]
    coop = args.workload == "coop_dac"
    # `value` and every number derived from it time the EVERY-ROW tower (every block computes every token row), as in rounds 1-5.  The product default
    # since round 6 -- the last image block computes K | V of every token and everything else for the class rows alone (clip/model.py:419 reads
    # nothing else) -- is timed on the same step as `value_cls_only`, with the flop it really issues.
    from clip_calibration_amd import _lib
    assert _lib.get_option("cls_only_last_block") == 1, "class rows only in the last image block is the product default"
    _lib.set_option("cls_only_last_block", 0)
    B = args.batch
    Cn = args.classes or (500 if coop else 1000)
    E = geom.embed_dim
    sd = syn.synthetic_state_dict(args.model, seed=0)
    model = build_model(dict(sd), {"trainer": "CoOp" if coop else "ZeroshotCLIP"}).to(dev)
    # N > 1 on real GPUs: the exchange MUST be the library's RCCL communicator over all ranks -- a run that fell back to torch.distributed
    # would look like a scaling measurement of something else.  backend="rccl" raises (on every rank alike) if it cannot be built.
    exchange = EmbeddingExchange(dev, backend="torch" if same_gpu else "rccl") if world > 1 else None
    if exchange is not None and not same_gpu and not (exchange.backend == "rccl" and exchange.rccl_ranks == world):
        raise SystemExit(f"bench: the exchange is {exchange.backend} over {exchange.rccl_ranks} ranks, not RCCL over {world}: refusing to time it")

    # ---- text side: computed once, outside the timed region (zsclip.py:90-92; CoOp: cached while ctx is unchanged)
    extra = {}
    dac_conf = None
    text_again = None
    if coop:
        from clip_calibration_amd.dac import DistanseAwareCalibration
        n_ctx = 16
        ids_new = syn.synthetic_token_ids(Cn, args.model, seed=11, n_ctx_placeholders=n_ctx)
        ids_base = syn.synthetic_token_ids(Cn, args.model, seed=10, n_ctx_placeholders=n_ctx)
        tuned_new = CoOpCLIP(model, ids_new, n_ctx=n_ctx, logit_scale=1.0, seed=3)     # cosine base model (base_model/coop.py:222-224)
        tuned_base = CoOpCLIP(model, ids_base, n_ctx=n_ctx, logit_scale=1.0, seed=3)
        zs_new = ZeroshotCLIP(model, syn.synthetic_token_ids(Cn, args.model, seed=11))
        zs_base = ZeroshotCLIP(model, syn.synthetic_token_ids(Cn, args.model, seed=10))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        text_features = tuned_new.text_features()
        torch.cuda.synchronize()
        text_s_cold = time.perf_counter() - t0
        cal = DistanseAwareCalibration()
        cal.fit(zs_base.text_features.cpu().numpy(), zs_new.text_features.cpu().numpy(),
                tuned_base.text_features().cpu().numpy(), text_features.cpu().numpy(), 5)
        dac_conf = cal.class_confidence_device(dev)
        scale = float(np.exp(4.6052))                                                   # TempScaling scalar (tempscaling.py:34)

        def text_fp32_stream():
            tuned_new._cache = None
            tuned_new._cache_key = None
            return tuned_new.text_features()
        # the reference's schedule: prompt learner + text tower on every batch (coop.py:208-210).  The mirror's cache_text_features=False
        # runs that per-batch tower call on the fp16 residual stream (per-call flag; the cached features above keep the fp32 stream)
        per_batch = CoOpCLIP(model, ids_new, n_ctx=n_ctx, logit_scale=1.0, seed=3, cache_text_features=False)
        per_batch.overlap_towers = os.environ.get("BENCH_OVERLAP_TOWERS", "1") == "1"   # text tower on a side stream beside the image tower
        text_again = per_batch.text_features
        with torch.no_grad():
            text_s = timed_ms(text_fp32_stream, 5) * 1e-3
            text_s_per_batch = timed_ms(text_again, 5) * 1e-3
        text_features = tuned_new.text_features()
    else:
        ids = syn.synthetic_token_ids(Cn, args.model, seed=0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        zs = in_turn(lambda: ZeroshotCLIP(model, ids))
        torch.cuda.synchronize()
        text_s_cold = time.perf_counter() - t0
        t0 = time.perf_counter()
        in_turn(lambda: zs.build_model(ids))
        torch.cuda.synchronize()
        text_s = time.perf_counter() - t0
        text_features, scale = zs.text_features, zs.scale

    img_seed = rank if args.images_seed is None else args.images_seed + rank
    shards = None
    if args.virtual_ranks > 1:
        assert world == 1 and args.exchange_f16, "--virtual-ranks is a single-process aid of the fp16 exchange path"
        # the towers run shard by shard, exactly as the ranks would (same batch -> same kernels), only the gather is a concatenation
        shards = [syn.synthetic_images(B, args.model, seed=img_seed + k, device=dev) for k in range(args.virtual_ranks)]
        images = shards[0]
    else:
        images = syn.synthetic_images(B, args.model, seed=img_seed, device=dev)   # resident in HBM before the timed region
    evaluator = DeviceCalibrationEvaluator(10, dev)
    n_bins = evaluator.n_bins
    f16_exchange = world > 1 or args.exchange_f16

    def tail(feats, labels, txt):
        bins = evaluator.bins if labels is not None else None
        if f16_exchange:
            emb = in_turn(lambda: ops.l2_normalize(feats, torch.float16))              # [B, E] fp16: 2*B*E bytes per rank on the wire
            if exchange is not None:
                emb = exchange.all_gather(emb)                                         # [world*B, E], rank-major
            return in_turn(lambda: ops.fused_tail(emb, txt, scale, dac_conf, True, False, labels, bins, n_bins))
        return ops.fused_tail(feats, txt, scale, dac_conf, True, True, labels, bins, n_bins)

    def step(labels, recompute_text=False):
        if recompute_text and world == 1 and shards is None:   # the reference's per-batch schedule: both towers of the batch (CoOpCLIP.towers)
            feats, txt = per_batch.towers(images)
            return tail(feats, labels, txt)
        txt = text_again() if recompute_text else text_features
        if shards is not None:
            emb = torch.cat([ops.l2_normalize(model.image_features_f32(x), torch.float16) for x in shards])
            bins = evaluator.bins if labels is not None else None
            return ops.fused_tail(emb, txt, scale, dac_conf, True, False, labels, bins, n_bins)
        return tail(in_turn(lambda: model.image_features_f32(images)), labels, txt)

    def run(labels, n_steps, recompute_text=False):
        evaluator.reset()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            out_ = step(labels, recompute_text)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, out_

    rows = world * B if exchange is not None else B * args.virtual_ranks
    with torch.no_grad():
        _, _, _, pred0 = step(None)                               # also sizes the workspaces
        # labels for every row the tail sees (the gathered batch when N > 1): 70 % agree with the prediction -> non-degenerate ECE
        labels = syn.synthetic_labels(pred0.cpu(), Cn, seed=1).to(dev)
        assert labels.numel() == rows
        for _ in range(args.warmup):
            step(labels)
        elapsed, last = run(labels, args.steps)
        res = evaluator.evaluate()
        bins_np = evaluator.bins.cpu().numpy().copy()
        if args.sustained_seconds > 0:
            # a second timed loop of the SAME step, long enough for the chip to reach its power / clock steady state (the K-step headline
            # region is ~0.2 s); `value` stays the K-step number
            n_s = max(args.steps, int(np.ceil(args.sustained_seconds / (elapsed / args.steps))))
            sampler = PowerSampler(local_rank) if rank == 0 else None
            if sampler:
                sampler.start()
            el_s, _ = run(labels, n_s)
            extra["sustained"] = {"seconds": el_s, "steps": n_s, "images_per_s": world * B * n_s / el_s, "ms_per_step": 1e3 * el_s / n_s,
                                  "ratio_to_value": (world * B * n_s / el_s) / (world * B * args.steps / elapsed)}
            if sampler:
                extra["power"] = sampler.stop()
        with _lib.option("cls_only_last_block", 1):   # the product default, same step, same K steps
            for _ in range(max(1, args.warmup // 2)):
                step(labels)
            elapsed_cls, _ = run(labels, args.steps)
        Lv, Dv = geom.vision_tokens, geom.vision_width
        flops_cls = syn.flops_per_image(args.model) - (Lv * (24 * Dv * Dv + 4 * Lv * Dv) - (4 * Lv * Dv * Dv + 20 * Dv * Dv + 4 * Lv * Dv))
        extra["value_cls_only"] = world * B * args.steps / elapsed_cls
        extra["cls_only"] = {
            "what": "the same step with cls_only_last_block = 1 (product default): the last image block computes K | V for every token, Q / attention / "
                    "out-proj / MLP for the class rows only (clip/model.py:419: ln_post reads x[:, 0, :])",
            "images_per_s": world * B * args.steps / elapsed_cls, "ms_per_step": 1e3 * elapsed_cls / args.steps,
            "gflop_per_image_issued": flops_cls / 1e9, "gflop_per_image_every_row": syn.flops_per_image(args.model) / 1e9,
            "tower_tflops_issued": flops_cls * world * B * args.steps / elapsed_cls / 1e12,
            "speedup_vs_value": elapsed / elapsed_cls}
        if coop:
            n_rt = max(2, args.steps // 2)
            for _ in range(2):
                step(labels, True)
            elapsed_rt, _ = run(labels, n_rt, True)
            res_rt = evaluator.evaluate()
            # how far the per-batch (fp16-stream) text features are from the cached (fp32-stream) ones, on this batch's logits: synthetic
            # weights give near-tied classes, so a few argmax flips (each inside the top-2 margin printed here) move accuracy / ECE
            lg16, lg32 = step(None, True)[0].float(), step(None)[0].float()     # logits as the evaluator sees them (DAC row scale applied)
            p16, p32 = lg16.argmax(1), lg32.argmax(1)
            top2 = lg32.topk(2, dim=1).values
            flipped = (p16 != p32).nonzero().flatten()
            t16, t32 = text_again().double(), text_features.double()             # both L2-normalised [C, E]
            stream_delta = {"max_text_feature_l2_distance": float((t16 - t32).norm(dim=1).max()),   # bounds every cosine-logit difference
                            "max_one_minus_cosine": float((1.0 - (t16 * t32).sum(1)).max()),
                            "max_abs_logit_diff_after_dac": float((lg16 - lg32).abs().max()), "logit_scale": scale,
                            "note": "a row whose argmax flips takes another class's DAC factor, which rescales the whole row: that is the logit "
                                    "difference; between the features themselves the distance bounds any cosine-logit difference",
                            "pred_flips": int(flipped.numel()), "rows": int(p32.numel()),
                            "max_top2_margin_of_flipped_rows": float((top2[flipped, 0] - top2[flipped, 1]).max()) if flipped.numel() else 0.0}
            # flop credited for the token rows the tower COMPUTES (dead-row elimination: rows behind the last EOT are never run); the
            # per-prompt figure is SURVEY 8(d)'s 5.960 GFLOP at 77 rows, linear terms scaled by rows / 77 (attention is < 3 % of it)
            text_rows = model.live_rows(per_batch.tokenized_prompts)
            tf = (lambda t: 5.960e9 * Cn * text_rows / model.context_length / t / 1e12) if args.model == "ViT-B/16" else (lambda t: None)
            extra["coop_dac"] = {
                "text_rows_computed": text_rows, "text_rows_of_context": model.context_length,
                "images_per_s_text_cached": world * B * args.steps / elapsed,
                "images_per_s_text_recomputed_every_batch": world * B * n_rt / elapsed_rt,
                "text_tower_prompts_per_s": Cn / text_s, "text_tower_ms": 1e3 * text_s, "text_tower_tflops": tf(text_s),
                "per_batch_text_tower": {"stream": "fp16 (CLIPMI_CALL_STREAM_F16; CoOpCLIP(cache_text_features=False))",
                                         "ms": 1e3 * text_s_per_batch, "prompts_per_s": Cn / text_s_per_batch, "tflops": tf(text_s_per_batch),
                                         "ece_percent": float(res_rt["ece"]), "accuracy_percent": float(res_rt["accuracy"]),
                                         "ece_percent_text_cached_fp32_stream": float(res["ece"]), "vs_cached_fp32_stream": stream_delta},
                "note": "the reference recomputes the text tower on every batch (trainers/classification/coop.py:208-210); "
                        "ctx is frozen at eval, so the features are cached here (precedent: proda.py:315-333)",
                "n_ctx": 16, "dac": "on (k = 5)", "tempscaling_logit_scale": 4.6052}

    if args.dump and rank == 0:
        np.savez(args.dump, logits=last[0].cpu().numpy(), conf=last[2].cpu().numpy(), pred=last[3].cpu().numpy(), bins=bins_np,
                 ece=res["ece"])

    tail_bytes = (2.0 if f16_exchange else 4.0) * rows * E + 4.0 * Cn * E + 4.0 * rows * Cn + 16.0 * rows
    workload = (f"BASELINE configs[2]: CoOp 16-shot {args.model} base->new eval + DAC + TempScaling scalar, {Cn} classes, batch {B} per GPU, "
                "text features cached" if coop else
                f"BASELINE configs[1]: zero-shot CLIP {args.model}, ImageNet-1k shape ({Cn} text prompts), batch {B} per GPU, "
                f"224x224 randn images, seeded random weights")
    out = {
        "metric": ("images/sec ViT-B/16 224px zero-shot + ECE" if not coop
                   else "images/sec ViT-B/16 CoOp base->new + DAC + ECE (BASELINE configs[2]; not the headline metric)"),
        "value": world * B * args.steps / elapsed,
        "unit": "images/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f16",
        "data": "synthetic",
        "config": {"workload": workload, "batch_per_gpu": B, "global_batch": world * B, "classes": Cn,
                   "rows_computed": "every token row of every block (cls_only_last_block = 0); the product default is value_cls_only",
                   "parallelism": (f"dp{world}: batch sharded, one {'RCCL (clipmi_allgather)' if exchange.backend == 'rccl' else ('gloo (same-GPU self-test)' if same_gpu else 'torch.distributed nccl (fallback)')} "
                                   f"all-gather of [B,{E}] fp16 embeddings per step, logits on the gathered batch on every rank")
                   if world > 1 else "single GPU"},
        "images_per_sec_per_gpu": B * args.steps / elapsed,
        "tower_tflops": syn.flops_per_image(args.model) * world * B * args.steps / elapsed / 1e12,
        "text_tower_once_s": text_s, "text_tower_first_call_s": text_s_cold,
        "ece_percent": res["ece"], "accuracy_percent": res["accuracy"],
    }
    out.update(extra)
    if world > 1:
        out["exchange"] = {"backend": exchange.backend, "rccl_ranks": exchange.rccl_ranks, "bytes_per_rank_per_step": 2 * B * E,
                           "world_size": world}

    if not args.no_roofline:
        with torch.no_grad():
            out["roofline"] = kernel_roofline(model, syn, geom, args.model, B, images)
            if rank == 0 and world == 1:
                out["ceiling"] = ceilings(dev, local_rank, geom, B)
                clk = out["ceiling"]["mfma_only"].get("in_kernel_clock_mhz")
                if clk:
                    # the datasheet's 2.5 PF is quoted at 2.4 GHz; under an MFMA-dense load this box holds `clk`: the same achieved rate
                    # against the peak AT THAT CLOCK (clock-normalised fraction, next to the datasheet one)
                    out["roofline"]["tower_frac_at_clock"] = out["roofline"]["tower"]["achieved"] / (MFMA_F16_DENSE_PEAK_TFLOPS * clk / MFMA_PEAK_CLOCK_MHZ)
                    out["roofline"]["frac_at_clock"] = out["roofline"]["achieved"] / (MFMA_F16_DENSE_PEAK_TFLOPS * clk / MFMA_PEAK_CLOCK_MHZ)
                if not coop and args.model == "ViT-B/16" and B >= 128 and not args.no_other_configs:
                    out["other_configs"] = other_configs(dev, syn, model, build_model, images, Cn)
            feats = model.image_features_f32(images)
            # device time of the tail: its launches are queued BEHIND a tower pass (10 ms of GPU work), so that the host's
            # ~40 us of Python per call never starves the stream and the events bracket back-to-back executions only
            n_tail = 40
            tail(feats, labels, text_features)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            model.image_features_f32(images)
            e0.record()
            for _ in range(n_tail):
                tail(feats, labels, text_features)
            e1.record()
            torch.cuda.synchronize()
            tail_ms = e0.elapsed_time(e1) / n_tail
            # Optional mode, NOT part of `value` (every timed step above computes every row of every block): the last block's
            # out-proj / MLP on the class rows only -- the only rows ln_post reads.  Reported with its feature difference.
            with _lib.option("cls_only_last_block", 1):
                feats_cls = model.image_features_f32(images).clone()
                cls_ms = timed_ms(lambda: model.image_features_f32(images), 10)
            fn_, cn_ = torch.nn.functional.normalize(feats, dim=1), torch.nn.functional.normalize(feats_cls, dim=1)
            out["class_rows_only_last_block"] = {
                "option": "cls_only_last_block=1 (the product default; every number above except value_cls_only / cls_only runs option 0)", "tower_ms": cls_ms,
                "tower_ms_every_row": out["roofline"]["tower"]["ms"], "images_per_s_tower_only": B / (cls_ms * 1e-3),
                "max_abs_cosine_diff_vs_every_row": float((1.0 - (fn_ * cn_).sum(1)).abs().max())}
        out["tail"] = {"what": "ONE launch (fused_tail_kernel): L2-normalise + scale*img@txt^T" + (" + DAC row scale" if coop else "") +
                               " + softmax top-1 (conf, pred) + ECE bin accumulation" +
                               (" [preceded by the fp16 normalise kernel" + (" and the all-gather]" if world > 1 else "]") if f16_exchange else ""),
                       "bound": "hbm", "bytes": tail_bytes, "us": 1e3 * tail_ms, "achieved": tail_bytes / (tail_ms * 1e-3) / 1e9, "unit": "GB/s",
                       "peak": HBM_PEAK_BYTES_PER_S / 1e9, "frac": tail_bytes / (tail_ms * 1e-3) / HBM_PEAK_BYTES_PER_S,
                       "bytes_note": "algorithmic: image features in + text features in + fp32 logits out + (conf, pred, label) per row"}

    if rank == 0 and world == 1 and not args.no_cpu_baseline and not coop:
        v, cores, avail, sample, lg_ref, cpu_images, cpu_labels = cpu_baseline(sd, args.model, args.cpu_classes, args.cpu_batch,
                                                                                args.cpu_seconds)
        out["cpu_baseline"] = {"value": v, "unit": "images/s", "cores": cores, "cores_available": avail, "kind": "port", "sample": sample}
        # parity on exactly that sample: HIP path vs oracle
        zs_cpu = ZeroshotCLIP(model, syn.synthetic_token_ids(args.cpu_classes, args.model, seed=0))
        with torch.no_grad():
            lg, _, _, conf, pred = zs_cpu.model_inference(cpu_images.to(dev), want_conf_pred=True)
        from clip_calibration_amd.metrics import ECE
        from oracle import clip_oracle as orc
        ece_ref, _, _ = orc.calibrated_ece(lg_ref.numpy(), cpu_labels.numpy())
        out["parity"] = {"max_abs_cosine_logit_err": float(np.abs(lg.cpu().numpy() - lg_ref.numpy()).max() / zs_cpu.scale),
                         "ece_delta": abs(ECE(conf.cpu().numpy(), pred.cpu().numpy(), cpu_labels.numpy()) - ece_ref),
                         "sample": f"{args.cpu_batch} images x {args.cpu_classes} prompts"}
    if rank == 0:
        out["env"] = environment()
        print(json.dumps(out))
    if exchange is not None:
        exchange.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
