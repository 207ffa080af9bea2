#!/bin/bash
# SQ counters per kernel of the default bench run (matrix-pipe busy cycles, wave cycles, wait buckets, LDS conflicts) ->
# gpurun_out/sq/summary.txt (copy into profiles/).  Counters only: no trace domains beside --pmc.  Two passes (8 SQ slots).
# Matrix-pipe utilisation of a kernel = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): pmc_summary prints the inputs.
set -e
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/sq"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/p1" -- \
    python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > "$OUT/p1.log" 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/p2" -- \
    python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > "$OUT/p2.log" 2>&1 || true
python3 "$ROOT/tools/pmc_summary.py" "$OUT" clipmi > "$OUT/summary.txt"
head -c 6000 "$OUT/summary.txt"
