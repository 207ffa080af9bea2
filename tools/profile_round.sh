#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench run -> gpurun_out/<tag>/ (copy the *_kernel_stats.csv into profiles/).
# Usage on the GPU box, from the repo root:   bash tools/profile_round.sh r02 [extra bench args]
set -e
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="${1:-prof}"; shift || true
OUT="$ROOT/gpurun_out/$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs "$@" > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof.log"
find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
head -n 14 "$OUT/kernel_stats.csv" | cut -c1-200
