#!/bin/bash
# Instruction counts per kernel of the image tower alone (tools/traffic_workload.py: ViT-B/16, batch 256, every dispatch has the bench shape) ->
# gpurun_out/sq2/summary.txt: SQ_INSTS_VALU / MFMA / LDS / SALU per dispatch.  Compare with what the ISA's straight path predicts: the ring attention
# kernel's "rare" rescale branch showed up this way (96 vector instructions per score tile against 56; profiles/r05_vitl_attention.txt).
# Counters only: no trace domains beside --pmc.  Run on the GPU box from the repo root.
set -e
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/sq2"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/p1" -- python3 "$ROOT/tools/traffic_workload.py" > "$OUT/p1.log" 2>&1
python3 "$ROOT/tools/pmc_summary.py" "$OUT" clipmi > "$OUT/summary.txt"
cat "$OUT/summary.txt"
