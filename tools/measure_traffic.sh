#!/bin/bash
# Regenerates profiles/gemm_traffic.json: HBM-side bytes per launch of the image tower's GEMM kernels from rocprofv3 PMC
# passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950: MI355X_MICROARCH.md "rocprofv3 PMC slots"), stamped with
# the sha256 of the gemm.hip they were measured on -- bench.py reports `roofline.traffic` only while that stamp matches.
# Run on the GPU box from the repo root:   bash tools/measure_traffic.sh        (counters only: no trace domains beside --pmc)
set -e
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/traffic"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d "$OUT/$ctr" -- python3 "$ROOT/tools/traffic_workload.py" > "$OUT/$ctr.log" 2>&1
done
python3 "$ROOT/tools/traffic_json.py" "$OUT/FETCH_SIZE" "$OUT/WRITE_SIZE" > "$ROOT/profiles/gemm_traffic.json"
cat "$ROOT/profiles/gemm_traffic.json"
