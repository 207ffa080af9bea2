#!/bin/bash
# Energy attribution of the dominant GEMM launches (profiles/r03_energy_attribution.txt): every ablation of the c_fc / c_proj loops as its
# own process, power and clock sampled from sysfs.  Needs: make -C clip_calibration_amd/csrc tuning ablate
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
C="$ROOT/clip_calibration_amd/csrc"
run() { VARIANT="$1" CLIPMI_LIBRARY="$2" KNOB="$3" timeout -k 10 120 python3 "$ROOT/tools/energy_probe.py" 2>&1 | grep " | "; }
echo "variant                            | kernel   | time       | power   | sclk      | energy"
run "product build"                        "$C/libclipmi.so" 0
run "tuning build (stamps off)"            "$C/libclipmi_tuning.so" 0
run "no output stores"                     "$C/libclipmi_ablate1.so" 0
run "no LDS-DMA in the K loop"             "$C/libclipmi_ablate2.so" 0
run "no MFMAs"                             "$C/libclipmi_ablate4.so" 0
run "no LDS fragment reads"                "$C/libclipmi_ablate16.so" 0
run "no element-wise epilogue (c_fc)"      "$C/libclipmi_tuning.so" 8
run "A panel L2-resident (c_fc)"           "$C/libclipmi_tuning.so" 64
run "no stores + no epilogue (c_fc)"       "$C/libclipmi_ablate1.so" 8
