#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS table of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py gemm.hip [extra hipcc flags]
"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "clip_calibration_amd", "csrc")


def main():
    src = sys.argv[1]
    cmd = ["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
           "-Rpass-analysis=kernel-resource-usage", *sys.argv[2:], "-c", src, "-o", "/tmp/_kernel_resources.o"]
    out = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
    cur, rows = None, {}
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            rows[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[a-zA-Z/]+\])?: (\d+)", line)
        if m and cur:
            rows[cur][m.group(1).strip()] = int(m.group(2))
    print(f"{src}: {len(rows)} kernels")
    names = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.splitlines()
    worst = 0
    for mangled, name in sorted(zip(rows, names), key=lambda t: t[1]):
        v = rows[mangled]
        name = re.sub(r"clipmi::\(anonymous namespace\)::", "", name)
        name = re.sub(r"\((clipmi|void|float|int|_Float16|long|HIP_vector|unsigned).*$", "", name)[:120]
        worst = max(worst, v.get("ScratchSize", 0))
        print("%4dv %3da %4ds spillS=%d spillV=%d scratch=%3d lds=%6d occ=%d  %s" % (
            v.get("VGPRs", 0), v.get("AGPRs", 0), v.get("TotalSGPRs", 0), v.get("SGPRs Spill", 0), v.get("VGPRs Spill", 0),
            v.get("ScratchSize", 0), v.get("LDS Size", 0), v.get("Occupancy", 0), name))
    print("max scratch bytes/lane:", worst)


if __name__ == "__main__":
    main()
