#!/bin/bash
# kernel trace + stats of the default bench command: scripts/prof_bench.sh <outdir> [bench args]
OUT=$1; shift
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"   # the repo root: the GPU box exports it; elsewhere derived from this file
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
mkdir -p "$OUT"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --no-cpu-baseline "$@" > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.err"
echo rc=$?
f=$(ls $OUT/trace/*/*kernel_stats.csv | head -1)
cp "$f" "$OUT/rocprofv3_kernel_stats.csv"
head -6 "$OUT/rocprofv3_kernel_stats.csv"
