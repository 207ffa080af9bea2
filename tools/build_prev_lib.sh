#!/bin/bash
# Builds clip_calibration_amd/csrc/libclipmi_prev.so from the csrc sources of a git revision (default HEAD): the "before" arm of tools/lib_tower_ab.py.
set -e
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
mkdir -p "$TMP/clip_calibration_amd/csrc" "$TMP/include"
git -C "$ROOT" archive "$REV" clip_calibration_amd/csrc include | tar -x -C "$TMP"
make -C "$TMP/clip_calibration_amd/csrc" -j6 libclipmi.so >/dev/null
cp "$TMP/clip_calibration_amd/csrc/libclipmi.so" "$ROOT/clip_calibration_amd/csrc/libclipmi_prev.so"
rm -rf "$TMP"
echo "built libclipmi_prev.so from $REV"
