#!/bin/bash
# Where does an item of the vision attention kernel spend its time?  Launch time with parts of the item removed at build time
# (make -C clip_calibration_amd/csrc tuning attn_ablate), then the phase stamps of the tuning build.
set -e
cd "$(dirname "$0")/.."
L=clip_calibration_amd/csrc
run() { CLIPMI_LIBRARY=$L/$1 WHAT="$2" python tools/attn_ablate.py 2>&1 | grep -v amdgpu.ids; }
run libclipmi_tuning.so "the kernel"
run libclipmi_attn1.so "no v_exp"
run libclipmi_attn16.so "no row-sum MFMA"
run libclipmi_attn2.so "no P.V / row-sum MFMAs"
run libclipmi_attn4.so "no S MFMAs"
run libclipmi_attn6.so "no MFMAs"
run libclipmi_attn7.so "no MFMAs, no v_exp"
run libclipmi_attn38.so "no MFMAs, no max phase"
run libclipmi_attn70.so "no MFMAs, no LDS fragment reads"
run libclipmi_attn71.so "no MFMAs, no v_exp, no LDS fragment reads"
run libclipmi_attn8.so "query waves 4-6 idle"
run libclipmi_attn14.so "query waves 4-6 idle, no MFMAs"
run libclipmi_attn79.so "waves 4-6 idle, no MFMA / v_exp / LDS reads"
run libclipmi_attn256.so "no output stores"
run libclipmi_attn327.so "no MFMA / v_exp / LDS reads / stores"
run libclipmi_attn128.so "contiguous operands"
run libclipmi_attn384.so "contiguous operands, no output stores"
run libclipmi_attn199.so "contiguous operands, no MFMA / v_exp / LDS reads"
run libclipmi_attn455.so "contiguous operands, nothing but the DMA"
CLIPMI_LIBRARY=$L/libclipmi_tuning.so python tools/attn_stamps.py 2>&1 | grep -v amdgpu.ids
