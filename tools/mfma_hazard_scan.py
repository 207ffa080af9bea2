#!/usr/bin/env python3
"""Static check for the hazard of profiles/r03_gpu_sharing.txt: a VGPR / AGPR written by a VALU instruction and read as a SOURCE operand by a
v_mfma fewer than WAIT wait states later (hipcc inserts none for some of these sequences on gfx950; the wave's own result is right, a co-resident
wave's registers are not).  Compiles a translation unit to ISA and runs a small data-flow pass over every kernel's control-flow graph: the set of
"recent VALU writes" (register -> wait states since) flows along fall-through edges AND along every s_branch / s_cbranch edge, forwards and
backwards, and is merged at labels with the SMALLEST distance -- so a write at the bottom of a loop that feeds an MFMA at its head through the
back-edge is seen.  Every instruction between the write and the MFMA counts one wait state, `s_nop N` counts N + 1.

    python tools/mfma_hazard_scan.py attention.hip logits.hip gemm.hip gemm_rstream.hip        (exit status 1 if a site is found, 2 if a
                                                                                                  file did not compile or holds no MFMA kernel)
    WAIT=16 EXTRA=-DCLIPMI_FENCE_SNOP=1 python tools/mfma_hazard_scan.py attention.hip         (EXTRA: more hipcc flags; every run also prints
                                                                                                  the SMALLEST distance it met below WAIT)
"""
import os
import re
import shutil
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "clip_calibration_amd", "csrc")
WAIT = int(os.environ.get("WAIT", "4"))
WAIT2 = int(os.environ.get("WAIT2", "32"))   # second check (round 5): MFMA-written register read by a vector / store instruction with another MFMA issued in between
MFMA_WAIT = 8                                # ... an MFMA in between counts as the 8 cycles it holds the vector issue port (MI355X_MICROARCH.md), anything else as 1
REG = re.compile(r"\b([va])(?:\[(\d+):(\d+)\]|(\d+)\b)")
LOADS = ("global_load", "buffer_load", "flat_load", "scratch_load", "ds_read", "ds_bpermute", "ds_swizzle", "ds_permute")
NO_FALLTHROUGH = ("s_branch", "s_endpgm", "s_setpc_b64", "s_swappc_b64")


class ScanError(RuntimeError):
    pass


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        lo, hi = (int(m.group(2)), int(m.group(3))) if m.group(2) is not None else (int(m.group(4)), int(m.group(4)))
        out.update((m.group(1), r) for r in range(lo, hi + 1))
    return out


def compile_to_isa(src, extra=()):
    hipcc = shutil.which(os.environ.get("HIPCC", "hipcc"))
    if hipcc is None:
        raise ScanError("hipcc not found")
    r = subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", *extra, "-S", "--cuda-device-only", src, "-o", "-"],
                       cwd=CSRC, capture_output=True, text=True)
    if r.returncode != 0:
        raise ScanError(f"hipcc failed on {src} (rc {r.returncode}):\n{r.stderr[-2000:]}")
    return r.stdout


def split_kernels(asm):
    """[(kernel symbol, [(op, operands, text)], {label: index of the instruction that follows it})]"""
    kernels, name, insts, labels = [], None, [], {}
    for line in asm.splitlines():
        t = line.strip()
        m = re.match(r"^(_Z\S+|[A-Za-z_]\w*):\s*(;.*)?$", t) if not t.startswith(".L") else None
        if m and not t.startswith("."):
            if name is not None:
                kernels.append((name, insts, labels))
            name, insts, labels = m.group(1), [], {}
            continue
        if name is None or not t or t.startswith((";", "//")):
            continue
        m = re.match(r"^(\.L\w+):", t)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if t.startswith("."):
            if t.startswith((".end_amdhsa_kernel", ".section", ".amdhsa_kernel")) and name is not None and insts:
                kernels.append((name, insts, labels))
                name, insts, labels = None, [], {}
            continue
        t = t.split(";")[0].strip()
        if not t:
            continue
        op, _, rest = t.partition(" ")
        insts.append((op, [o.strip() for o in rest.split(",")], t))
    if name is not None and insts:
        kernels.append((name, insts, labels))
    return kernels


def merge(dst, src):
    """dst, src: {register: (wait states since the write, text of the write)}; keeps the smaller distance.  True if dst changed."""
    changed = False
    for r, (w, x) in src.items():
        if r not in dst or w < dst[r][0]:
            dst[r] = (w, x)
            changed = True
    return changed


def scan_kernel(name, insts, labels):
    n = len(insts)
    state_in = [None] * (n + 1)          # None = not reached yet
    state_in[0] = {}
    sites, n_mfma = {}, 0
    dirty = True
    passes = 0
    while dirty:
        dirty = False
        passes += 1
        if passes > 64:
            raise ScanError(f"{name}: data flow did not settle")
        n_mfma = 0
        for i, (op, ops, text) in enumerate(insts):
            st = state_in[i]
            if st is None:
                continue
            if op.startswith("v_mfma") or op.startswith("v_smfmac"):
                n_mfma += 1
                if st and len(ops) >= 4:
                    srcs = set().union(*[regs(o) for o in ops[1:4]])
                    for r in srcs & st.keys():
                        w, x = st[r]
                        if w < WAIT:
                            sites[(x, text)] = min(w, sites.get((x, text), w))
            out = st
            if st or op.startswith("v_"):
                out = dict(st)
                if op.startswith(LOADS) and ops[-1:] != ["lds"] and " lds" not in text:
                    for r in regs(ops[0]):     # a later load into the register supersedes the VALU write (its own wait is a counter, not wait states)
                        out.pop(r, None)
                step = (int(ops[0] or 0) + 1) if op == "s_nop" else 1
                out = {r: (w + step, x) for r, (w, x) in out.items() if w + step < WAIT}
                if op.startswith("v_") and not op.startswith(("v_mfma", "v_smfmac", "v_cmp", "v_cmpx", "v_nop")) and ops and ops[0]:
                    written = regs(ops[0])
                    if op.startswith(("v_permlane16_swap", "v_permlane32_swap", "v_swap")) and len(ops) > 1:
                        written |= regs(ops[1])
                    for r in written:
                        out[r] = (0, text)
            targets = []
            if not op.startswith(NO_FALLTHROUGH):
                targets.append(i + 1)
            if op.startswith(("s_branch", "s_cbranch")):
                lab = ops[-1].strip()
                if lab in labels:
                    targets.append(labels[lab])
            for t in targets:
                if state_in[t] is None:
                    state_in[t] = dict(out)
                    if t <= i:
                        dirty = True
                elif merge(state_in[t], out) and t <= i:
                    dirty = True
    return [(name, x, m, w) for (x, m), w in sites.items()], n_mfma


def scan_kernel_reads(name, insts, labels):
    """Round 5's second hazard (common.h CLIPMI_MFMA_TO_VALU_FENCE3; profiles/r05_vitl_attention.txt): a register an MFMA has written, read by a vector (or
    LDS / global store) instruction fewer than WAIT2 wait states later WITH at least one other MFMA issued in between -- the shape of `acc2 = mfma(..);
    acc0 = mfma(..); s_nop 10; v_add x, x, acc2[0]` that corrupted co-resident waves.  (A read of the LATEST MFMA's result behind hipcc's own s_nop --
    scores into the softmax maximum -- is the ordinary case every kernel here has and is not reported.)  Same data flow as scan_kernel: state =
    {register: (wait states since the MFMA, other MFMAs since, text)}, merged at labels with the smallest distance."""
    n = len(insts)
    state_in = [None] * (n + 1)
    state_in[0] = {}
    sites = {}
    dirty, passes = True, 0
    while dirty:
        dirty = False
        passes += 1
        if passes > 64:
            raise ScanError(f"{name}: data flow did not settle")
        for i, (op, ops, text) in enumerate(insts):
            st = state_in[i]
            if st is None:
                continue
            is_mfma = op.startswith(("v_mfma", "v_smfmac"))
            reader = (op.startswith("v_") and not is_mfma and not op.startswith("v_nop")) or op.startswith(("ds_write", "global_store", "buffer_store", "scratch_store"))
            if reader and st:
                src_ops = ops[1:] if op.startswith("v_") else ops
                srcs = set().union(*[regs(o) for o in src_ops]) if src_ops else set()
                if op.startswith(("v_fmac", "v_mac", "v_pk_fmac", "v_dot2c")) and ops:
                    srcs |= regs(ops[0])
                for r in srcs & st.keys():
                    w, others, x = st[r]
                    if w < WAIT2 and others >= 1:
                        sites[(x, text)] = min(w, sites.get((x, text), w))
            out = dict(st)
            step = (int(ops[0] or 0) + 1) if op == "s_nop" else (MFMA_WAIT if is_mfma else 1)
            out = {r: (w + step, o + (1 if is_mfma else 0), x) for r, (w, o, x) in out.items() if w + step < WAIT2}
            if is_mfma and ops:
                for r in regs(ops[0]):
                    out[r] = (0, 0, text)
            elif ops and ops[0] and (op.startswith("v_") or op.startswith(LOADS)) and not op.startswith(("v_cmp", "v_cmpx")):
                for r in regs(ops[0]):      # overwritten by something else: no longer an MFMA result
                    out.pop(r, None)
            targets = []
            if not op.startswith(NO_FALLTHROUGH):
                targets.append(i + 1)
            if op.startswith(("s_branch", "s_cbranch")):
                lab = ops[-1].strip()
                if lab in labels:
                    targets.append(labels[lab])
            for t in targets:
                if state_in[t] is None:
                    state_in[t] = dict(out)
                    if t <= i:
                        dirty = True
                else:
                    changed = False
                    for r, v in out.items():
                        if r not in state_in[t] or v[0] < state_in[t][r][0] or (v[0] == state_in[t][r][0] and v[1] > state_in[t][r][1]):
                            state_in[t][r] = v
                            changed = True
                    if changed and t <= i:
                        dirty = True
    return [(name, x, m, w) for (x, m), w in sites.items()]


def register_bounds(asm, kernels):
    """Round 6 (verdict r5 item 6a): a co-resident wave's registers being damaged is also what an out-of-allocation register WRITE looks like -- an
    inline-asm operand or tied tuple reaching past what the kernel descriptor allocates.  For every kernel: the highest VGPR / AGPR index any
    instruction names against the descriptor (.amdhsa_next_free_vgpr, .amdhsa_accum_offset: arch VGPRs live below the accumulation offset, AGPRs in
    next_free_vgpr - accum_offset behind it).  Returns [(kernel, max v, max a, next_free_vgpr, accum_offset, ok)]."""
    desc = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", asm, re.S):
        nf = re.search(r"\.amdhsa_next_free_vgpr (\d+)", m.group(2))
        ao = re.search(r"\.amdhsa_accum_offset (\d+)", m.group(2))
        if nf:
            desc[m.group(1)] = (int(nf.group(1)), int(ao.group(1)) if ao else int(nf.group(1)))
    out = []
    for name, insts, _labels in kernels:
        if name not in desc:
            continue
        mv = ma = -1
        for _op, ops, _text in insts:
            for o in ops:
                for kind, r in regs(o):
                    if kind == "v":
                        mv = max(mv, r)
                    else:
                        ma = max(ma, r)
        nf, ao = desc[name]
        ok = mv < min(ao, nf) and (ma < 0 or ma < nf - ao) and nf <= 512 and ao <= 256   # (no AGPRs: next_free_vgpr counts the arch VGPRs alone)
        out.append((name, mv, ma, nf, ao, ok))
    return out


def scan(src, extra=()):
    """(sites, kernels with MFMAs, MFMA instructions) of one translation unit; raises ScanError when the file does not compile or holds no MFMA
    kernel at all (a guard that scans nothing must not pass)."""
    asm = compile_to_isa(src, extra)
    kernels = split_kernels(asm)
    scan.bounds = register_bounds(asm, kernels)
    sites, reads, with_mfma, total = [], [], 0, 0
    for name, insts, labels in kernels:
        s, n = scan_kernel(name, insts, labels)
        sites += s
        if n:
            reads += scan_kernel_reads(name, insts, labels)
        with_mfma += 1 if n else 0
        total += n
    if not with_mfma:
        raise ScanError(f"{src}: no kernel with a v_mfma found in {len(kernels)} functions -- nothing was checked")
    return sites, with_mfma, total, reads


def main():
    bad = 0
    for src in sys.argv[1:]:
        try:
            sites, nk, nm, reads = scan(src, tuple(os.environ.get("EXTRA", "").split()))
        except ScanError as e:
            print(f"{src}: SCAN FAILED: {e}")
            sys.exit(2)
        print(f"{src}: {len(sites)} VALU write -> MFMA source sites with fewer than {WAIT} wait states ({nk} MFMA kernels, {nm} MFMA instructions walked)"
              + (f"; smallest distance {min(w for _, _, _, w in sites)}" if sites else ""))
        for kernel, w, m, waited in sites[:int(os.environ.get("SHOW", "12"))]:
            name = subprocess.run(["c++filt", kernel or "?"], capture_output=True, text=True).stdout.strip()[:90]
            print(f"   {name}\n      {w}\n      {m}      ({waited} wait states between)")
        bad += len(sites)
        only = os.environ.get("KERNELS")         # second check: reported for the kernels named here (comma-separated substrings), all when unset
        if only:
            reads = [r for r in reads if any(k in r[0] for k in only.split(","))]
        print(f"{src}: {len(reads)} MFMA result -> vector / store read sites with another MFMA in between and fewer than {WAIT2} wait states"
              + (f"; smallest distance {min(w for _, _, _, w in reads)}" if reads else ""))
        for kernel, w, m, waited in reads[:int(os.environ.get("SHOW", "12"))]:
            name = subprocess.run(["c++filt", kernel or "?"], capture_output=True, text=True).stdout.strip()[:90]
            print(f"   {name}\n      {w}\n      {m}      ({waited} wait states between)")
        if os.environ.get("STRICT_READS", "0") == "1":
            bad += len(reads)
        bounds = scan.bounds
        over = [b for b in bounds if not b[5]]
        print(f"{src}: {len(over)} kernels name a register outside their allocation ({len(bounds)} kernel descriptors checked; the tightest: "
              + ", ".join(f"v{b[1]} of {min(b[3], b[4])}" for b in sorted(bounds, key=lambda b: min(b[3], b[4]) - b[1])[:3]) + ")")
        for b in over:
            print(f"   {b[0][:100]}: max v{b[1]} / a{b[2]} against next_free_vgpr {b[3]}, accum_offset {b[4]}")
        bad += len(over)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
