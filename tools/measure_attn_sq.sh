#!/bin/bash
# SQ counters of the long-sequence attention kernel alone (tools/attn_ring_once.py) -> gpurun_out/attn_sq/summary.txt.  Counters only: no trace
# domains beside --pmc; three passes.  SQ_*_CYCLES count quad-cycles per SIMD-wave; SQ_VALU_MFMA_BUSY_CYCLES counts cycles (MI355X_MICROARCH.md).
set -e
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/attn_sq"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/p1" -- \
    python3 "$ROOT/tools/attn_ring_once.py" > "$OUT/p1.log" 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --output-format csv -d "$OUT/p2" -- \
    python3 "$ROOT/tools/attn_ring_once.py" > "$OUT/p2.log" 2>&1 || true
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d "$OUT/p3" -- \
    python3 "$ROOT/tools/attn_ring_once.py" > "$OUT/p3.log" 2>&1 || true
python3 "$ROOT/tools/pmc_summary.py" "$OUT" attention > "$OUT/summary.txt"
head -c 6000 "$OUT/summary.txt"
