#!/bin/bash
# kernel trace of the evaluation pass: scripts/prof_forward.sh <outdir>
OUT=$1
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"   # the repo root: the GPU box exports it; elsewhere derived from this file
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
mkdir -p "$OUT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/fwd" -- python3 scripts/forward_timing.py > "$OUT/fwd.log" 2>&1
echo rc=$?
f=$(ls $OUT/fwd/*/*kernel_stats.csv | head -1)
cp "$f" "$OUT/forward_kernel_stats.csv"
head -12 "$OUT/forward_kernel_stats.csv"
