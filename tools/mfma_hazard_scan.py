#!/usr/bin/env python3
"""Static check for the hazard of profiles/r03_gpu_sharing.txt: a VGPR written by a VALU instruction and read as a SOURCE operand by a v_mfma
fewer than WAIT wait states later (hipcc inserts none for some of these sequences on gfx950; the wave's own result is right, a co-resident wave's
registers are not).  Compiles a translation unit to ISA and walks every kernel; every instruction between the write and the MFMA counts one wait
state, `s_nop N` counts N + 1.

    python tools/mfma_hazard_scan.py attention.hip logits.hip gemm.hip gemm_rstream.hip        (exit status 1 if a site is found)
"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "clip_calibration_amd", "csrc")
WAIT = int(os.environ.get("WAIT", "2"))
REG = re.compile(r"\b([va])(?:\[(\d+):(\d+)\]|(\d+)\b)")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        lo, hi = (int(m.group(2)), int(m.group(3))) if m.group(2) is not None else (int(m.group(4)), int(m.group(4)))
        out.update((m.group(1), r) for r in range(lo, hi + 1))
    return out


def scan(src):
    asm = subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-S", "--cuda-device-only", src, "-o", "-"],
                         cwd=CSRC, capture_output=True, text=True).stdout
    sites, kernel, window = [], None, []      # window: (wait states since, written registers, text) of recent VALU writes
    for line in asm.splitlines():
        t = line.strip()
        m = re.match(r"^(_Z\S+):", t)
        if m:
            kernel, window = m.group(1), []
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        t = t.split(";")[0].strip()
        op, _, rest = t.partition(" ")
        ops = [o.strip() for o in rest.split(",")]
        if op.startswith("v_mfma"):
            srcs = set().union(*[regs(o) for o in ops[1:4]]) if len(ops) >= 4 else set()
            for waited, written, text in window:
                if waited < WAIT and written & srcs:
                    sites.append((kernel, text, t, waited))
        if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load", "ds_read", "ds_bpermute", "ds_swizzle")) and "lds" not in ops[-1:]:
            over = regs(ops[0])          # a later load into the register supersedes the VALU write (its own wait is a counter, not wait states)
            window = [(w, r - over, x) for (w, r, x) in window]
        step = int(rest.strip() or 0) + 1 if op == "s_nop" else 1
        window = [(w + step, r, x) for (w, r, x) in window if w + step < WAIT + 2]
        if op.startswith("v_") and not op.startswith(("v_mfma", "v_cmp", "v_cmpx", "v_nop")) and ops and ops[0]:
            written = regs(ops[0])
            if op.startswith(("v_permlane16_swap", "v_permlane32_swap", "v_swap")) and len(ops) > 1:
                written |= regs(ops[1])
            window.append((0, written, t))
    return sites


def main():
    bad = 0
    for src in sys.argv[1:]:
        sites = scan(src)
        print(f"{src}: {len(sites)} VALU write -> MFMA source sites with fewer than {WAIT} wait states")
        for kernel, w, m, waited in sites[:int(os.environ.get("SHOW", "12"))]:
            name = subprocess.run(["c++filt", kernel or "?"], capture_output=True, text=True).stdout.strip()[:90]
            print(f"   {name}\n      {w}\n      {m}      ({waited} wait states between)")
        bad += len(sites)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
