#!/bin/bash
# Round-end records beside the headline profile (tools/profile_round.sh): rocprofv3 kernel-trace summaries of the ViT-L/14@336px tower (BASELINE
# configs[4] per-rank shape, product default and every-row) and of the per-batch CoOp text tower, and the three non-headline bench lines.
# Usage on the GPU box, from the repo root:   bash tools/profile_extras.sh r06
set -e
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="${1:-extras}"
OUT="$ROOT/gpurun_out/$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/vitl" -- python3 "$ROOT/tools/vitl_tower_once.py" > "$OUT/vitl.log" 2>&1
find "$OUT/vitl" -name "*kernel_stats.csv" -exec cp {} "$OUT/vitl336_kernel_stats.csv" \;
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/text" -- python3 "$ROOT/tools/text_tower_once.py" > "$OUT/text.log" 2>&1
find "$OUT/text" -name "*kernel_stats.csv" -exec cp {} "$OUT/text_tower_kernel_stats.csv" \;
rm -rf "$OUT/vitl" "$OUT/text"
cd "$ROOT"
python3 bench.py --workload coop_dac > "$OUT/bench_coop_dac.json" 2> "$OUT/bench_coop_dac.err"
python3 bench.py --workload level1 > "$OUT/bench_level1.json" 2> "$OUT/bench_level1.err"
python3 bench.py --workload stream > "$OUT/bench_stream.json" 2> "$OUT/bench_stream.err"
python3 tools/text_bench.py > "$OUT/text_bench.txt" 2>&1
